#!/bin/bash
# Profiles one BASELINE config of bench.py on the MI355X box and leaves the raw rocprofv3 output
# under gpurun_out/<tag>/ (scratch); tools/summarize_profile.py turns it into profiles/<tag>_*.
#   tools/profile_config.sh <config> <tag> [extra bench.py args]
# Passes (counters in their own runs, kernel-trace only, as the pool requires):
#   stats : --kernel-trace --stats          -> per-kernel durations
#   fetch : --pmc FETCH_SIZE                -> HBM read side   (TCC: FETCH_SIZE takes 3 of 4 slots)
#   write : --pmc WRITE_SIZE                -> HBM write side
#   sq    : --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
#                 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY   (8 SQ slots)
set -o pipefail
cfg=$1; tag=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
# --no-torch: the profiled process holds one HIP / HSA runtime (with torch's bundled one beside the tool's, the
# passes of config 4 -- cooperative launches -- aborted inside exit(), profiles/README.md); -e again since then
set -e
# (60 launches per bench step: a few hundred dispatches of the stepping kernel per pass keep the counter files small;
# the driver's default is several hundred launches per step so that 20 steps are 5 s of GPU time)
common="--config $cfg --cpu-seconds 0 --steps 4 --warmup 1 --launches-per-step 60 --no-torch $*"
python3 bench.py $common > $out/bench_short.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $common > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py $common > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py $common > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $out/sq -- python3 bench.py $common > $out/sq.log 2>&1
python3 tools/summarize_profile.py $tag $cfg
# the calibration launches of the same runs (markov_chain_calibrate before the timed steps)
python3 tools/summarize_profile.py $tag $cfg pt_calibrate calib
echo "profiled config $cfg -> $out"
