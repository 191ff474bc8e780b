#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02coop
mkdir -p $out
L=$PWD/tmp_exp/coop.so
APEMOST_HIP_LIB=$L timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "launch_policies" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "n_swap", d["config"]["n_swap"])')"; }
run c2_512 $L --config 2 --chains-per-gpu 512
run c2_384 $L --config 2 --chains-per-gpu 384
run c2_512_single $L --config 2 --chains-per-gpu 512 --flags 1
run c2 $L --config 2
