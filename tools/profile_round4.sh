#!/bin/bash
# The round-4 profiles, one GPU call: every BASELINE config through tools/profile_config.sh (kernel stats,
# FETCH/WRITE_SIZE, one SQ pass; the calibration launches summarised beside the stepping launches and their phase's
# issue roofline registered in profiles/pmc_traffic.json).  The profiled process holds no torch (bench.py --no-torch).
set -o pipefail
suffix=${1:-a}
for cfg in ${2:-2 3 4 5}; do
  extra=""
  [ $cfg = 5 ] && extra="--burn-in 100"
  tools/profile_config.sh $cfg r04_c${cfg}_$suffix $extra || echo "config $cfg profile failed"
done
echo profiles done
