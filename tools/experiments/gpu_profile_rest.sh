#!/bin/bash
set -o pipefail
timeout -k 10 300 bash tools/profile_config.sh 2 r02_c2_final3 || { echo "config 2 failed"; exit 1; }
timeout -k 10 300 bash tools/profile_config.sh 4 r02_c4_final3 || { echo "config 4 failed"; exit 1; }
timeout -k 10 700 bash tools/profile_config.sh 5 r02_c5_final3 --burn-in 100 || { echo "config 5 failed"; exit 1; }
