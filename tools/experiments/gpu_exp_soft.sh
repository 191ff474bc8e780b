#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02soft
mkdir -p $out
L=$PWD/tmp_exp/soft.so
APEMOST_HIP_LIB=$L timeout -k 10 400 python -m pytest tests/test_gpu_one_barrier.py tests/test_gpu_parity.py tests/test_gpu_variants.py -q -x -k "one_barrier or (trajectory and not 1-) or launch_policies" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
run() { # tag lib args...
  local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "n_swap", d["config"]["n_swap"], "acc %.3f"%d["config"]["acceptance_rate_rank0"])')"
}
run c2 $L --config 2
run c2_r32 $L --config 2 --rounds-per-step 32
run c2_r512 $L --config 2 --rounds-per-step 512 --steps 10
run c4 $L --config 4
run c4_r32 $L --config 4 --rounds-per-step 32
run c4_s7 $L --config 4 --n-swap 7
run c2_256 $L --config 2 --chains-per-gpu 256
run c2_s1 $L --config 2 --n-swap 1
