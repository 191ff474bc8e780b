#!/bin/bash
# A/B script behind numbers in DESIGN.md 15 (round 3: the pulse model over a common denominator, APEMOST_PULSE_ND,
# and the register budget of its one-wave kernel, APEMOST_PULSE_MIN_WAVES).  tmp_exp/*.so (not tracked) are
# development builds: apemost_amd.build.build_dev([1], [1, 2, 4, 8], out=..., extra=[-D switches]).
set -o pipefail
out=gpurun_out/r03pulsend
mkdir -p $out
run() { # tag lib args...
  local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 200 python bench.py --cpu-seconds 0 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "acc %.3f"%d["config"]["acceptance_rate_rank0"], "calib %.4f" % d["calibration"]["wall_s"])')"
}
for v in "$@"; do
run c4_$v $PWD/tmp_exp/$v.so --config 4
run c4whole_$v $PWD/tmp_exp/$v.so --config 4 --chains-per-gpu 2048 --rounds-per-step 32 --launches-per-step 20
done
