#!/bin/bash
# round 4, GPU session 7: what the pair-range guard of the pulse likelihoods costs at config 4, and whether testing it
# behind the cross-lane sum (guard2) gets it back.  p_nosplit = the guard in front of the sum (sessions 3-6).
set -o pipefail
out=gpurun_out/r04_s7
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2 3; do for v in p_nosplit guard2 noguard; do
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/guard.txt
