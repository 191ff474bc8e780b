#!/bin/bash
# round 4, GPU session 28: the final product library -- GPU suite, smoke(), the driver's default bench line.
set -o pipefail
out=gpurun_out/r04_s28
mkdir -p $out
( while true; do date +%T >> $out/heartbeat.txt; sleep 45; done ) & hb=$!
trap "kill $hb" EXIT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "gpu suite rc $rc"; tail -n 4 $out/pytest_gpu.log
[ $rc = 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $out/smoke.log; exit 1; }
tail -n 1 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err && tail -n 1 $out/bench_default.json | cut -c1-400
