#!/bin/bash
# round 4, GPU session 3: where the pulse prior goes (config 4's shard, pulse 256 x 1024, 4 likelihood waves).
#   p_nosplit : round 3's kernel, the owner computes the prior (APEMOST_OB_SPLIT=0)
#   p_seq     : helper = the producer in its Philox phase; decision first, then the chosen row's prior
#   p_both    : helper = that producer; both rows' priors side by side, select behind the decision
#   p_wave    : helper = a ninth wavefront (both rows);  p_wave_seq: the same, decision first
# then per-role busy ticks of the stamped twins, the timing-only owner-slack experiment at config 2 (DESIGN.md 9:
# what a step costs when nobody waits for the owner's whole program), and the GPU tests of this session's code.
set -o pipefail
out=gpurun_out/r04_s3
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do for v in p_nosplit p_seq p_both p_wave p_wave_seq; do
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/ab_c4.txt
for v in st_nosplit st_seq st_both st_wave; do
echo "== $v"; APEMOST_STAMP_LIB=$PWD/tmp_exp/r04_$v.so timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 || exit 1
done 2>&1 | tee $out/roles_c4.txt
for rep in 1 2; do for v in base slack3 slack1 slack0; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --no-calibrate --launches-per-step 40 || exit 1
done; done 2>&1 | tee $out/slack_c2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_user_model.py tests/test_gpu_one_barrier.py tests/test_gpu_calibration.py -x -q -s > $out/pytest.log 2>&1
rc=$?
tail -6 $out/pytest.log
exit $rc
