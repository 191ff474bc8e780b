#!/bin/bash
# The round-4 profiles, one GPU call: every BASELINE config through tools/profile_config.sh (kernel stats,
# FETCH/WRITE_SIZE, one SQ pass; the calibration launches summarised beside the stepping launches and their phase's
# issue roofline registered in profiles/pmc_traffic.json).  The profiled process holds no torch (bench.py --no-torch).
set -o pipefail
suffix=${1:-a}
for cfg in ${2:-2 3 4 5}; do
  extra=""
  # (config 5's shard -- 2048 one-wave workgroups, eight per CU: the occupancy figure itself, one more than the
  # residency estimate admits -- steps one round per launch: a bench "launch" of 32 rounds is 32 dispatches there, and
  # every dispatch under --pmc costs ~0.1 s of counter read-back: 3 batches per step, not 60)
  [ $cfg = 5 ] && extra="--burn-in 100 --launches-per-step 3"
  tools/profile_config.sh $cfg r04_c${cfg}_$suffix $extra || echo "config $cfg profile failed"
done
echo profiles done
