#!/bin/bash
# final profile pass of a round: every BASELINE config, four rocprofv3 passes each
set -o pipefail
tagp=${1:-r02}
for c in 2 3 4 5; do
  echo "== profiling config $c"
  timeout -k 10 400 bash tools/profile_config.sh $c ${tagp}_c${c}_final2 || { echo "config $c failed or timed out"; exit 1; }
done
