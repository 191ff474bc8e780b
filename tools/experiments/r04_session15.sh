#!/bin/bash
# round 4, GPU session 15: the helper's step as one basic block per number of modes, the prior's division by the number
# of modes as product + two FMAs (SmallDivisor).  GPU suite, role profiles (stamped twin), config 4 with / without helper.
set -o pipefail
out=gpurun_out/r04_s15
mkdir -p $out
( while true; do date +%T >> $out/heartbeat.txt; sleep 45; done ) & hb=$!
trap "kill $hb" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "gpu suite rc $rc"; tail -n 4 $out/pytest_gpu.log
[ $rc = 0 ] || exit 1
timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/roles_c4_helper.txt 2>&1; cat $out/roles_c4_helper.txt
APEMOST_OB_HELPER=0 timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/roles_c4_nohelper.txt 2>&1; cat $out/roles_c4_nohelper.txt
run() { local tag=$1; shift
  timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s, %s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0), d["roofline"]["kernel"]))')"; }
for rep in 1 2; do
run c4_helper --config 4 --launches-per-step 200 || exit 1
APEMOST_OB_HELPER=0 run c4_nohelper --config 4 --launches-per-step 200 || exit 1
done 2>&1 | tee $out/ab.txt
run c2 --config 2 | tee -a $out/ab.txt
run c5 --config 5 --steps 3 --warmup 1 | tee -a $out/ab.txt
