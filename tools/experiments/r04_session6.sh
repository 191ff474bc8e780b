#!/bin/bash
# round 4, GPU session 6: which roles share a SIMD in the one-barrier workgroup (APEMOST_OB_WAVE_PERM 0 / 1 / 2,
# pt_onebarrier.h setup_common), configs 2 and 4; parity of the permuted builds against the oracle on the one-barrier
# tests (the tests take the library from APEMOST_HIP_LIB).
set -o pipefail
out=gpurun_out/r04_s6
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do for v in wp0 wp1 wp2; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 60 || exit 1
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/wave_perm.txt
for v in wp1 wp2; do
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_$v.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-simplesin]" "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" "tests/test_gpu_one_barrier.py::test_one_barrier_round_shapes[1-4]" "tests/test_gpu_one_barrier.py::test_one_barrier_round_shapes[15-4]" "tests/test_gpu_one_barrier.py::test_injected_ties_of_the_accept_comparison" > $out/pytest_$v.log 2>&1; echo "$v parity rc $?"; tail -2 $out/pytest_$v.log
done
# config 5's shard (2048 one-wave chains = eight per CU, the occupancy figure itself): multi-round launches through
# hipLaunchCooperativeKernel (APEMOST_COOP_ANY=1) against one round per launch
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export APEMOST_COOP_ANY=1; else unset APEMOST_COOP_ANY; fi
  timeout -k 10 300 python bench.py --config 5 --burn-in 100 --cpu-seconds 0 --steps 10 --warmup 2 --launches-per-step 20 --calib-iter-limit 400 > $out/b_c5_coop$v.log 2>&1 || { echo "c5 coop$v failed"; tail -5 $out/b_c5_coop$v.log; }
  echo "c5_coop$v $(tail -n 1 $out/b_c5_coop$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us" % (d["value"], d["roofline"]["launch_us"]))')"
done; done 2>&1 | tee $out/c5_coop.txt
unset APEMOST_COOP_ANY
