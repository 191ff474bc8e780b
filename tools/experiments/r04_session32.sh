#!/bin/bash
# round 4, GPU session 32: what the sample rows cost the one-barrier kernels (bench.py with and without --no-samples), and the
# helper wavefront's gain by number of modes and ladder size (tools/helper_rate.py).
set -o pipefail
out=gpurun_out/r04_s32
mkdir -p $out
run() { local tag=$1; shift
  timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us" % (d["value"], d["roofline"]["launch_us"]))')"; }
for rep in 1 2; do
run c2 --config 2 --launches-per-step 40
run c2_nosamples --config 2 --launches-per-step 40 --no-samples
run c4 --config 4 --launches-per-step 200
run c4_nosamples --config 4 --launches-per-step 200 --no-samples
done 2>&1 | tee $out/samples.txt
