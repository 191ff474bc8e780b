#!/bin/bash
# round 4, GPU session 20: the candidate producers look at the redraw flag behind their phase's work (pfl) instead of in
# front of it (base / f3); on top of it the TIMING-ONLY owner without the prepared proposals (noatt) as the ceiling of
# moving them to another wave; phases of the stamped twin.  Config 2: s_*, config 4: f3*.
set -o pipefail
out=gpurun_out/r04_s20
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
for v in s_base s_pfl s_pfl_noatt; do run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --no-calibrate --launches-per-step 40 || exit 1; done
for v in f3 f3_pfl f3_pfl_noatt; do run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --no-calibrate --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/pfl.txt
for v in s_base s_pfl; do run c2cal_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 40 || exit 1; done 2>&1 | tee -a $out/pfl.txt
for v in f3 f3_pfl; do run c4cal_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done 2>&1 | tee -a $out/pfl.txt
export APEMOST_STAMP_LIB=$PWD/tmp_exp/r04_stamps_ph.so APEMOST_STAMP_PHASES=1
timeout -k 10 120 python tools/ob_profile.py simplesin 128 1024 4 > $out/phases_c2.txt 2>&1; cat $out/phases_c2.txt
timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/phases_c4.txt 2>&1; cat $out/phases_c4.txt
APEMOST_HIP_LIB=$PWD/tmp_exp/r04_f3_pfl.so timeout -k 10 300 python -m pytest -x -q "tests/test_gpu_one_barrier.py::test_one_barrier_equals_two_phase_kernel_and_oracle[4-pulse]" "tests/test_gpu_one_barrier.py::test_one_barrier_redraw_path_and_circular_parameters" > $out/pytest_pfl.log 2>&1; echo "pfl parity rc $?"; tail -n 3 $out/pytest_pfl.log
