#!/bin/bash
# round 4, GPU session 24: config 4 and the candidate sets' Philox evaluation: f3 = before, f3_ph0 = the current source
# with two evaluations in the one-barrier producers (what the pulse models get by default), f3_ph1 = one.
set -o pipefail
out=gpurun_out/r04_s24
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2 3; do
for v in f3 f3_ph0 f3_ph1; do run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/ph.txt
