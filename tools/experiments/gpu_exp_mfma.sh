#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02mfma
mkdir -p $out
run() { # tag lib args...
  local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "acc %.3f"%d["config"]["acceptance_rate_rank0"])')"
}
for rep in 1 2; do
run c2_mfma $PWD/tmp_exp/mfma.so --config 2
run c2_dpp $PWD/tmp_exp/dpp.so --config 2
run c4_mfma $PWD/tmp_exp/mfma.so --config 4
run c4_dpp $PWD/tmp_exp/dpp.so --config 4
done
APEMOST_HIP_LIB=$PWD/tmp_exp/mfma.so timeout -k 10 300 python -m pytest tests/test_gpu_one_barrier.py tests/test_gpu_parity.py -q -x -k "simplesin or pulse or one_barrier or trajectory" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
