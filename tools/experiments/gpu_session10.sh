#!/bin/bash
set -o pipefail
out=gpurun_out/r02s10
mkdir -p $out
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $out/session.log; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc" | tee -a $out/session.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; tail -20 $out/$name.log; exit 1; fi; return 0; }
for c in 3 2 4 5; do
  step bench_c$c 200 python bench.py --config $c --cpu-seconds 0 --burn-in 200
  tail -n 1 $out/bench_c$c.log | cut -c1-200
done




step pytest 1000 python -m pytest tests -q -m gpu
tail -6 $out/pytest.log
