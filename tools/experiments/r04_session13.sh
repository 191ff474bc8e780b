#!/bin/bash
# round 4, GPU session 13: where the helper wavefront sits and what its kernels may use.
#   h0 = the product form (helper = wave 8: SIMD 0, beside likelihood wave 0 and the owner)
#   h1/h2/h3 = 1/2/3 placeholder wavefronts that end at once in front of it: the helper on SIMD 1/2/3, beside a producer
#   e3 = the nine-wave kernels compiled for three waves per SIMD (168 registers: the calibration kernel stops spilling)
# and, at config 2, the timing-only owner-slack build again on the round's final likelihood step (s_base / s_slack).
set -o pipefail
out=gpurun_out/r04_s13
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do for v in ${VARIANTS:-h0 h1 h2 h3 e3 e3h1}; do
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/helper_place.txt
for rep in 1 2; do for v in s_base s_slack; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --no-calibrate --launches-per-step 40 || exit 1
done; done 2>&1 | tee $out/slack_c2.txt
for v in ${PARITY:-h1 e3}; do
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_$v.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" "tests/test_gpu_one_barrier.py::test_injected_ties_of_the_accept_comparison[pulse]" "tests/test_gpu_parity.py::test_pulse_few_modes_paths_match_oracle" > $out/pytest_$v.log 2>&1; echo "$v parity rc $?"; tail -n 3 $out/pytest_$v.log
done
