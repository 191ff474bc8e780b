#!/bin/bash
# round 4, GPU session 35: session 34's scheduler strategies on the one-wave kernels of configs 3 and 5 (run phase only,
# --no-calibrate: the calibration is the same likelihood loop)
set -o pipefail
out=gpurun_out/r04_s35
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 6 --warmup 2 --no-calibrate "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us" % (d["value"], d["roofline"]["launch_us"]))')"; }
for rep in 1 2; do
for v in base ilp clause bias0; do run c3_$v $PWD/tmp_exp/r04_s3_$v.so --config 3 --launches-per-step 60 || exit 1; done
for v in base ilp clause bias0; do run c5_$v $PWD/tmp_exp/r04_pv_$v.so --config 5 --launches-per-step 20 || exit 1; done
done 2>&1 | tee $out/sched.txt
