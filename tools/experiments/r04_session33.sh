#!/bin/bash
# round 4, GPU session 33: the sample row as ONE store per step (lane 62 carries prob - prior) in the one-barrier round
# kernel.  base = the product library, one = development builds with it.  Configs 2 and 4; parity of the rows.
set -o pipefail
out=gpurun_out/r04_s33
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2 3; do
run c2_base $PWD/apemost_amd/libapemost_hip.so --config 2 --launches-per-step 40 || exit 1
run c2_one $PWD/tmp_exp/r04_s_one.so --config 2 --launches-per-step 40 || exit 1
run c4_base $PWD/apemost_amd/libapemost_hip.so --config 4 --launches-per-step 200 || exit 1
run c4_one $PWD/tmp_exp/r04_p_one.so --config 4 --launches-per-step 200 || exit 1
done 2>&1 | tee $out/one_store.txt
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_s_one.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-simplesin]" "tests/test_gpu_one_barrier.py::test_one_barrier_config2_bench_shape_matches_oracle" > $out/pytest_s.log 2>&1; echo "simplesin parity rc $?"; tail -n 2 $out/pytest_s.log
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_p_one.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" > $out/pytest_p.log 2>&1; echo "pulse parity rc $?"; tail -n 2 $out/pytest_p.log
