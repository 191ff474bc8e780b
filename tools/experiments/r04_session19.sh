#!/bin/bash
# round 4, GPU session 19: the candidate producers' step by phase (stamped twin with -DAPEMOST_STAMP_PHASES), configs 2 and 4
set -o pipefail
out=gpurun_out/r04_s19
mkdir -p $out
export APEMOST_STAMP_LIB=$PWD/tmp_exp/r04_stamps_ph.so APEMOST_STAMP_PHASES=1
timeout -k 10 120 python tools/ob_profile.py simplesin 128 1024 4 > $out/phases_c2.txt 2>&1; cat $out/phases_c2.txt
timeout -k 10 120 python tools/ob_profile.py pulse 256 1024 4 > $out/phases_c4.txt 2>&1; cat $out/phases_c4.txt
