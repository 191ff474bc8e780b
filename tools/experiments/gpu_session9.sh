#!/bin/bash
set -o pipefail
out=gpurun_out/r02s9
mkdir -p $out
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $out/session.log; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc" | tee -a $out/session.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; tail -20 $out/$name.log; exit 1; fi; return 0; }
step variants 600 python -m pytest tests/test_gpu_variants.py -q -x
tail -15 $out/variants.log
for c in 2 3 4 5; do
  step bench_c$c 200 python bench.py --config $c --cpu-seconds 0 --burn-in 200
  tail -n 1 $out/bench_c$c.log | cut -c1-200
done
step pytest 1000 python -m pytest tests -q -m gpu
tail -6 $out/pytest.log
step bench_c3_w2 200 python bench.py --config 3 --cpu-seconds 0 --burn-in 200 --waves 2
tail -n 1 $out/bench_c3_w2.log | cut -c1-200
