#!/bin/bash
# round 4, GPU session 16: SIMD 0 of the nine-wave workgroup holds likelihood wave 0, the owner and the helper.
#   p3  = the kernels before the helper's restructuring (session 14's build)      n3 = after (SmallDivisor, one block per mode count)
#   n3s0 = n3 without the branch on the mode count in prior_only()
#   lXoY = likelihood wave 0 at s_setprio X, the owner at Y (product: lik 0, owner 3)
# config 2 (simplesin, eight-wave workgroups): s_base, s_l3o3, s_l3o1, s_l2o3
set -o pipefail
out=gpurun_out/r04_s16
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do for v in ${VARIANTS:-p3 n3 n3s0 l3o3 l3o1 l3o0 l2o1 l1o0}; do
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/prio_c4.txt
for rep in 1 2; do for v in s_base s_l3o3 s_l3o1 s_l2o3; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 40 || exit 1
done; done 2>&1 | tee $out/prio_c2.txt
