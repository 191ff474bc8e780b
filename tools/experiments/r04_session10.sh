#!/bin/bash
# round 4, GPU session 10: mov_dpp (no zeroed destination) against update_dpp(0, ...) in the one-wave kernels of configs 3
# and 5 (wave_allreduce_sum, once per step), same box, three runs each.
set -o pipefail
out=gpurun_out/r04_s10
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us" % (d["value"], d["roofline"]["launch_us"]))')"; }
for rep in 1 2 3; do for v in z0 z1; do
run c3_$v $PWD/tmp_exp/r04_$v.so --config 3 --launches-per-step 100 --burn-in 400 --calib-iter-limit 400 || exit 1
run c5_$v $PWD/tmp_exp/r04_$v.so --config 5 --launches-per-step 20 --burn-in 100 --calib-iter-limit 400 || exit 1
done; done 2>&1 | tee $out/dpp_onewave.txt
