#!/bin/bash
# GPU session 2 of round 2
set -o pipefail
out=gpurun_out/r02s2
mkdir -p $out
step() { # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "== $name" | tee -a $out/session.log
    timeout -k 10 $secs "$@" > $out/$name.log 2>&1
    local rc=$?
    echo "== $name rc=$rc" | tee -a $out/session.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; exit 1; fi
    return 0
}
step pytest 900 python -m pytest tests -q -m gpu
tail -15 $out/pytest.log
step bench_c2 120 python bench.py --cpu-seconds 0
step bench_c3 120 python bench.py --config 3 --cpu-seconds 0
step bench_c4 120 python bench.py --config 4 --cpu-seconds 0
step bench_c5 200 python bench.py --config 5 --cpu-seconds 0 --burn-in 200
step sink_rate 300 python tools/sink_rate.py
step profile_c5 600 tools/profile_config.sh 5 r02_c5_base --burn-in 200
step profile_c3 420 tools/profile_config.sh 3 r02_c3_prefetch
tail -2 $out/bench_c*.log $out/sink_rate.log
cat $out/session.log
