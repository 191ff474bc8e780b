#!/bin/bash
# round 4, GPU session 18: TIMING ONLY -- the owner without the two prepared proposals of the next step
# (-DAPEMOST_EXP_NO_ATTEMPTS: the rows are never rewritten, results are garbage): what a step would cost if another
# wavefront prepared them, before that wavefront's own cost.  Config 2 (s_base / s_noatt), config 4 (f3 / f3_noatt).
set -o pipefail
out=gpurun_out/r04_s18
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
run c2_s_base $PWD/tmp_exp/r04_s_base.so --config 2 --no-calibrate --launches-per-step 40 || exit 1
run c2_s_noatt $PWD/tmp_exp/r04_s_noatt.so --config 2 --no-calibrate --launches-per-step 40 || exit 1
run c4_f3 $PWD/tmp_exp/r04_f3.so --config 4 --no-calibrate --launches-per-step 200 || exit 1
run c4_f3_noatt $PWD/tmp_exp/r04_f3_noatt.so --config 4 --no-calibrate --launches-per-step 200 || exit 1
done 2>&1 | tee $out/noatt.txt
