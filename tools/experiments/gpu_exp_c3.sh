#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02c3
mkdir -p $out
for w in 2 1 2; do
  APEMOST_HIP_LIB=$PWD/tmp_exp/c3w2.so timeout -k 10 150 python bench.py --config 3 --cpu-seconds 0 --burn-in 200 --waves $w > $out/b_$w.log 2>&1 || { echo "w$w failed"; tail -5 $out/b_$w.log; exit 1; }
  echo "waves $w $(tail -n 1 $out/b_$w.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"].get("kernel"))')"
done
