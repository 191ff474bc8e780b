#!/bin/bash
# round 4, GPU session 22 (= 21 with the 64-bit compare intrinsic fixed): one Philox evaluation per candidate set instead of two (ph1), and the owner's prepared
# proposals with the comparisons' own lane masks instead of ballots of booleans (as).  Config 2: s_base (before), s_ph1,
# s_ph1as; config 4: f3 (before), f3_ph1, f3_ph1as.  Then the producers' phases (stamped twin) and parity of the new forms.
set -o pipefail
out=gpurun_out/r04_s22
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
for v in s_base s_ph1 s_ph1as; do run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 40 || exit 1; done
for v in f3 f3_ph1 f3_ph1as; do run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/ph1.txt
export APEMOST_STAMP_LIB=$PWD/tmp_exp/r04_stamps_ph.so APEMOST_STAMP_PHASES=1
timeout -k 10 120 python tools/ob_profile.py simplesin 128 1024 4 > $out/phases_c2.txt 2>&1; cat $out/phases_c2.txt
timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/phases_c4.txt 2>&1; cat $out/phases_c4.txt
unset APEMOST_STAMP_LIB APEMOST_STAMP_PHASES
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_s_ph1as.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-simplesin]" "tests/test_gpu_one_barrier.py::test_one_barrier_config2_bench_shape_matches_oracle" > $out/pytest_s.log 2>&1; echo "simplesin parity rc $?"; tail -n 3 $out/pytest_s.log
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_f3_ph1as.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" > $out/pytest_p.log 2>&1; echo "pulse parity rc $?"; tail -n 3 $out/pytest_p.log
