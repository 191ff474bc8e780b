#!/bin/bash
# The round-3 profiles, one GPU call: every BASELINE config through tools/profile_config.sh
# (kernel stats, FETCH/WRITE_SIZE, one SQ pass; the calibration launches are summarised beside the
# stepping launches).  Config 4 runs with the engine's own geometry (eight likelihood waves, placed by
# hipLaunchCooperativeKernel) and leaves /proc/self/maps of every pass beside its log: in round 2 the
# profiled process of that shape died inside exit() after rocprofv3's finalisation, and the frames of
# such a trace can only be assigned to libraries with the map of the same process.
set -o pipefail
suffix=${1:-a}
tools/profile_config.sh 2 r03_c2_$suffix || echo "config 2 profile failed"
tools/profile_config.sh 3 r03_c3_$suffix || echo "config 3 profile failed"
mkdir -p gpurun_out/r03_c4_$suffix
tools/profile_config.sh 4 r03_c4_$suffix --maps-dump gpurun_out/r03_c4_$suffix/maps.txt || echo "config 4 profile failed"
tools/profile_config.sh 5 r03_c5_$suffix --burn-in 100 || echo "config 5 profile failed"
echo profiles done
