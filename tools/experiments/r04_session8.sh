#!/bin/bash
# round 4, GPU session 8 (final state): the whole -m gpu suite, smoke(), the un-profiled bench lines of every config.
set -o pipefail
out=gpurun_out/r04_final
mkdir -p $out
( while sleep 45; do date >> $out/heartbeat.txt; done ) &
hb=$!
trap "kill $hb" EXIT
timeout -k 10 800 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1
rc=$?
tail -6 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
python __graft_entry__.py smoke 2>&1 | tail -2
for cfg in 2 3 4 5; do
  extra=""
  [ $cfg = 5 ] && extra="--burn-in 100"
  timeout -k 10 400 python bench.py --config $cfg $extra > $out/bench_c$cfg.json 2> $out/bench_c$cfg.err \
    && tail -1 $out/bench_c$cfg.json | cut -c1-200
done
