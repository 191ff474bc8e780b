#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02law
mkdir -p $out
for v in 7 0 1 2 4 7 0; do
  for c in 2 4; do
    APEMOST_HIP_LIB=$PWD/tmp_exp/law$v.so timeout -k 10 120 python -c "
import sys, runpy
from apemost_amd import capi
capi.ABI_VERSION = 1 if '$v' == 'old' else 2
sys.argv = ['bench.py', '--config', '$c', '--cpu-seconds', '0', '--burn-in', '200']
runpy.run_path('bench.py', run_name='__main__')" > $out/b_${v}_$c.log 2>&1 || { echo "law$v c$c failed"; tail -5 $out/b_${v}_$c.log; exit 1; }
    echo "law$v c$c $(tail -n 1 $out/b_${v}_$c.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"
  done
done
