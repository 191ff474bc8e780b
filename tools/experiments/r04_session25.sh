#!/bin/bash
# round 4, GPU session 25: the product library after sessions 13-24 (helper budget and priority, one Philox evaluation per candidate set): GPU suite, smoke, all four configs.
# (pulse / pulse_vrot, ladders of at most one chain per CU).  The whole GPU suite and smoke() on the product library, then
# config 4 with the helper (default) and without (APEMOST_OB_HELPER=0), and configs 2, 3, 5 as they stand.
set -o pipefail
out=gpurun_out/r04_s25
mkdir -p $out
( while true; do date +%T >> $out/heartbeat.txt; sleep 45; done ) & hb=$!
trap "kill $hb" EXIT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "gpu suite rc $rc"; tail -n 4 $out/pytest_gpu.log
[ $rc = 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $out/smoke.log; exit 1; }
tail -n 1 $out/smoke.log
run() { local tag=$1; shift
  timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s, %s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0), d["roofline"]["kernel"]))')"; }
for rep in 1 2; do
run c4_helper --config 4 --launches-per-step 200 || exit 1
APEMOST_OB_HELPER=0 run c4_nohelper --config 4 --launches-per-step 200 || exit 1
done 2>&1 | tee $out/ab.txt
run c2 --config 2 | tee -a $out/ab.txt
run c3 --config 3 --steps 4 | tee -a $out/ab.txt
run c5 --config 5 --steps 3 --warmup 1 | tee -a $out/ab.txt
