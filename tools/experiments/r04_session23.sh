#!/bin/bash
# round 4, GPU session 23: with one Philox evaluation per candidate set (ph1) the producers are no longer as long as the
# owner at config 2 -- what the owner-side experiments say now: both prepared proposals through attempts2 (ph1m), the
# timing-only owner without them (noatt) and with slack (slack).  Config 4: f3 (before) / f3_ph1.
set -o pipefail
out=gpurun_out/r04_s23
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
for v in s_base s_ph1 s_ph1m; do run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 40 || exit 1; done
for v in s_ph1_noatt s_ph1_slack; do run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --no-calibrate --launches-per-step 40 || exit 1; done
for v in f3 f3_ph1; do run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/ph1m.txt
run c2_s_ph1_nocal $PWD/tmp_exp/r04_s_ph1.so --config 2 --no-calibrate --launches-per-step 40 | tee -a $out/ph1m.txt
