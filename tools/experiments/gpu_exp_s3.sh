#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02s3
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 200 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"])')"; }
for rep in 1 2; do
run c3_idx $PWD/tmp_exp/s3idx.so --config 3
run c3_half $PWD/tmp_exp/s3half.so --config 3
run c3_now $PWD/apemost_amd/libapemost_hip.so --config 3
done
run c5_idx $PWD/tmp_exp/s3idx.so --config 5
run c3_idx_w2 $PWD/tmp_exp/s3idx.so --config 3 --waves 2
