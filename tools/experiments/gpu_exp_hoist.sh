#!/bin/bash
# A/B script behind a number in DESIGN.md 16 (APEMOST_HOIST_CAND: the owner's read of the next tick's candidates
# in the step's first batch of LDS reads).  tmp_exp/h0.so, h1.so: apemost_amd.build.build_dev([0, 1], [1, 4, 8], ...).
set -o pipefail
out=gpurun_out/r03hoist
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 200 python bench.py --cpu-seconds 0 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], "calib %.4f" % d["calibration"]["wall_s"])')"; }
for rep in 1 2; do for v in ${VARIANTS:-h0 h1}; do
[ -n "$SKIP_C2" ] || run c2_$v $PWD/tmp_exp/$v.so --config 2
run c4_$v $PWD/tmp_exp/$v.so --config 4
done; done
