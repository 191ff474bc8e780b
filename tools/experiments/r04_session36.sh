#!/bin/bash
# round 4, GPU session 36: the candidate producers at s_setprio 1 / 2 / 3 (they share SIMDs 1-3 with likelihood waves that
# have slack at config 2, and their logarithm phase is one of the three waves the step waits for there).  Configs 2 and 4.
set -o pipefail
out=gpurun_out/r04_s36
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
run c2_base $PWD/apemost_amd/libapemost_hip.so --config 2 --launches-per-step 40 || exit 1
for v in pp1 pp2 pp3; do run c2_$v $PWD/tmp_exp/r04_s_$v.so --config 2 --launches-per-step 40 || exit 1; done
run c4_base $PWD/apemost_amd/libapemost_hip.so --config 4 --launches-per-step 200 || exit 1
for v in pp1 pp2 pp3; do run c4_$v $PWD/tmp_exp/r04_p_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/prod_prio.txt
