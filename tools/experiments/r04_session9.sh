#!/bin/bash
# round 4, GPU session 9: the three data loops behind the register-held pass of a one-barrier likelihood wave.  When the
# whole vector is one pass held in registers (configs 2 and 4) all three loop tests fail -- but each is a per-lane test
# (v_cmp -> s_and_saveexec -> s_cbranch_execz -> s_or exec) on the step's dependent chain; APEMOST_OB_SKIP_LOOPS=1 puts one
# uniform test in front of them.  sl0b / sl2: + the full-mask DPP moves without a zeroed destination (mov_dpp: eight
# v_mov_b32 fewer per cross-lane sum); sl3: + the partial sums first in the step's LDS queue (APEMOST_OB_PART_FIRST).
set -o pipefail
out=gpurun_out/r04_s9
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2 3; do for v in ${VARIANTS:-sl0 sl1 sl0b sl2 sl3}; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 60 || exit 1
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/skip_loops.txt
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_${PARITY:-sl3}.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-simplesin]" "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" "tests/test_gpu_one_barrier.py::test_one_barrier_round_shapes[1-4]" "tests/test_gpu_one_barrier.py::test_one_barrier_round_shapes[15-4]" "tests/test_gpu_one_barrier.py::test_injected_ties_of_the_accept_comparison" "tests/test_gpu_one_barrier.py::test_one_barrier_redraw_path_and_circular_parameters" > $out/pytest_sl3.log 2>&1; echo "sl3 parity rc $?"; tail -2 $out/pytest_sl3.log
