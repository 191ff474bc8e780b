#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02prior
mkdir -p $out
L=$PWD/tmp_exp/prior.so
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "n_swap", d["config"]["n_swap"])')"; }
run c4 $L --config 4
run c4_old $PWD/apemost_amd/libapemost_hip.so --config 4
run c4w $L --config 4 --chains-per-gpu 2048 --rounds-per-step 32
run c4w_old $PWD/apemost_amd/libapemost_hip.so --config 4 --chains-per-gpu 2048 --rounds-per-step 32
run c5s $L --config 5 --chains-per-gpu 128 --n-data 1024 --n-swap 15 --rounds-per-step 64
run c5s_old $PWD/apemost_amd/libapemost_hip.so --config 5 --chains-per-gpu 128 --n-data 1024 --n-swap 15 --rounds-per-step 64
APEMOST_HIP_LIB=$L timeout -k 10 400 python -m pytest tests/test_gpu_one_barrier.py tests/test_gpu_parity.py tests/test_gpu_calibration.py -q -x -k "(pulse or maximum_parameter) and not 2- and not 6- and not sine3 and not simplesin" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
