#!/bin/bash
# round 4, GPU session 11: the pulse model's prepared rows in registers in the one-barrier likelihood waves (both rows with
# the step's first LDS reads, c_tau for both formed beside the add tree, a select behind the decision) against the row
# pointer chosen behind the decision (round 3).  r0 = off, r1 = rows of 8 (three modes), r2 = rows of 6 (two modes),
# r1w / r0w = the same under a budget of 256 registers (APEMOST_OB_WAVES_PER_EU=2: no spills).  Config 4's shard.
set -o pipefail
out=gpurun_out/r04_s11
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2 3; do for v in ${VARIANTS:-r0 r1 r2 r0w r1w}; do
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/row_regs.txt
for v in ${PARITY:-r1}; do
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_$v.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" "tests/test_gpu_one_barrier.py::test_injected_ties_of_the_accept_comparison[pulse]" "tests/test_gpu_one_barrier.py::test_one_barrier_sampling_through_the_range_guard_matches_oracle[pulse-1e+30-1e+200]" "tests/test_gpu_one_barrier.py::test_one_barrier_sampling_through_the_range_guard_matches_oracle[pulse-1.0-1e-250]" "tests/test_gpu_parity.py::test_pulse_few_modes_paths_match_oracle" > $out/pytest_$v.log 2>&1; echo "$v parity rc $?"; tail -3 $out/pytest_$v.log
done
