#!/bin/bash
# A/B script behind numbers in DESIGN.md 15 (round 3, config 2: the one-barrier step's two dependent chains).
# The libraries it compares (tmp_exp/*.so, not tracked) are development builds:
# apemost_amd.build.build_dev([0], [1, 4], out=..., extra=[-D switches]) with the switches of pt_device.h
# (APEMOST_THR_SHORTCUT, APEMOST_SIN_FOLD_N, APEMOST_EVEN_ODD_WAVES).
set -o pipefail
out=gpurun_out/r03c2chain
mkdir -p $out
run() { # tag lib args...
  local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "acc %.3f"%d["config"]["acceptance_rate_rank0"], "calib %.4f" % d["calibration"]["wall_s"])')"
}
for rep in 1 2; do
for v in "$@"; do
run c2_$v $PWD/tmp_exp/$v.so --config 2
done
done
