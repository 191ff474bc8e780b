#!/bin/bash
# round 4, GPU session 34: the compiler's instruction scheduler, since a dozen instructions' placement decides these
# kernels: -mllvm -amdgpu-sched-strategy=max-ilp (ilp) / max-memory-clause (clause), -amdgpu-schedule-metric-bias=0
# (bias0), -enable-post-misched=0 (nopost) against the product library.  Configs 2 and 4.
set -o pipefail
out=gpurun_out/r04_s34
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do
run c2_base $PWD/apemost_amd/libapemost_hip.so --config 2 --launches-per-step 40 || exit 1
for v in ilp clause bias0 nopost; do run c2_$v $PWD/tmp_exp/r04_s_$v.so --config 2 --launches-per-step 40 || exit 1; done
run c4_base $PWD/apemost_amd/libapemost_hip.so --config 4 --launches-per-step 200 || exit 1
for v in ilp clause bias0 nopost; do run c4_$v $PWD/tmp_exp/r04_p_$v.so --config 4 --launches-per-step 200 || exit 1; done
done 2>&1 | tee $out/sched.txt
