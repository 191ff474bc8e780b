#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02ob4
mkdir -p $out
run() { # tag lib args...
  local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "n_swap", d["config"]["n_swap"], "lds", d["config"]["data_in_lds"], "acc %.3f"%d["config"]["acceptance_rate_rank0"])')"
}
L=$PWD/tmp_exp/ob4.so
for nd in 2048 4096 16384 65536; do
run c2_${nd}_ob4 $L --config 2 --n-data $nd --waves 4 --steps 10
run c2_${nd}_ob8 $L --config 2 --n-data $nd --waves 8 --steps 10
done
run c2_16ch_ob4 $L --config 2 --chains-per-gpu 16 --waves 4
run c2_16ch_ob8 $L --config 2 --chains-per-gpu 16 --waves 8
run c2_256_ob8 $L --config 2 --chains-per-gpu 256 --waves 8
