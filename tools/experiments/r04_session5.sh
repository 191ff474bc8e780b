#!/bin/bash
# round 4, GPU session 5: the round's profiles (tools/profile_round4.sh: every BASELINE config, four rocprofv3 passes
# each, no torch in the profiled process) and the un-profiled bench lines of the same build.
#   r04_session5.sh <suffix> "<configs to profile>" "<configs to bench>"
# (a heartbeat file: config 5's passes are minutes without a line of output, and a silent run is taken to be hung)
set -o pipefail
mkdir -p gpurun_out/r04_final
( while sleep 45; do date >> gpurun_out/r04_final/heartbeat.txt; done ) &
hb=$!
trap "kill $hb" EXIT
benches="${3:-2 3 4 5}"; [ "$benches" = "-" ] && benches=""
for cfg in $benches; do
  extra=""
  [ $cfg = 5 ] && extra="--burn-in 100"
  timeout -k 10 400 python bench.py --config $cfg $extra > gpurun_out/r04_final/bench_c$cfg.json 2> gpurun_out/r04_final/bench_c$cfg.err \
    && tail -1 gpurun_out/r04_final/bench_c$cfg.json | cut -c1-260
done
tools/profile_round4.sh ${1:-a} "${2:-2 3 4 5}" > gpurun_out/r04_final/profile_${1:-a}.log 2>&1
tail -3 gpurun_out/r04_final/profile_${1:-a}.log
