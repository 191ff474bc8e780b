#!/bin/bash
# round 4, GPU session 14: the one-barrier kernel with the helper wavefront (config 4): who waits for whom now.
#   role profile of the stamped twin (tools/ob_profile.py), the owner's priority (p3 = product ... p0), the helper's
#   (p3h1..3), and the timing-only owner-slack build on top of the helper.
set -o pipefail
out=gpurun_out/r04_s14
mkdir -p $out
timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/roles_c4_helper.txt 2>&1; cat $out/roles_c4_helper.txt
APEMOST_OB_HELPER=0 timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/roles_c4_nohelper.txt 2>&1; cat $out/roles_c4_nohelper.txt
timeout -k 10 120 python tools/ob_profile.py simplesin 128 1024 4 > $out/roles_c2.txt 2>&1; cat $out/roles_c2.txt
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do for v in ${VARIANTS:-p3 p2 p1 p0 p3h1 p3h2 p3h3 slack}; do
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/prio.txt
