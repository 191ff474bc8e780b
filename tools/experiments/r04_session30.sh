#!/bin/bash
# round 4, GPU session 30: the pulse model's c = 1 / (2 pi lifetime)^2 of both prepared rows formed beside the decision
# (APEMOST_PULSE_SPEC_TAU) in the one-barrier likelihood waves.  w0 = product, t1 = with it and the redraw flag's test behind
# the add tree, t2 = with it alone, t3 = the flag's test behind the tree alone.  Config 4, with and without the helper.
set -o pipefail
out=gpurun_out/r04_s30
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
for v in ${VARIANTS:-w0 t1 t2 t3}; do run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/spec.txt
for v in ${VARIANTS:-w0 t1 t2 t3}; do APEMOST_OB_HELPER=0 run c4nh_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done 2>&1 | tee -a $out/spec.txt
for v in ${PARITY:-t1}; do
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_$v.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" "tests/test_gpu_one_barrier.py::test_injected_ties_of_the_accept_comparison[pulse]" > $out/pytest_$v.log 2>&1; echo "$v parity rc $?"; tail -n 2 $out/pytest_$v.log
done
