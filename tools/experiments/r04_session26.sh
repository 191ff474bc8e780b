#!/bin/bash
# round 4, GPU session 26: nine-wave workgroups, the owner and the first producer change places (APEMOST_OB_WAVE_PERM=3:
# SIMD 0 = likelihood wave 0 + a producer + the helper, SIMD 1 = likelihood wave 1 + the owner).  w0 = product, w3, w3l0 =
# w3 without likelihood wave 0's priority.  Config 4.
set -o pipefail
out=gpurun_out/r04_s26
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
for v in w0 w3 w3l0; do run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/perm.txt
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_w3.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" > $out/pytest_w3.log 2>&1; echo "w3 parity rc $?"; tail -n 2 $out/pytest_w3.log
