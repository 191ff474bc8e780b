#!/bin/bash
# GPU session 6 of round 2: final state -- all tests, sink rates, profiles of every config
set -o pipefail
out=gpurun_out/r02s6
mkdir -p $out
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $out/session.log; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc" | tee -a $out/session.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; tail -20 $out/$name.log; exit 1; fi; return 0; }
step pytest 900 python -m pytest tests -q -m gpu
tail -6 $out/pytest.log
step sink_rate 300 python tools/sink_rate.py 600000
step ob_profile 120 python tools/ob_profile.py simplesin 128 1024 8
step bench_c2 200 python bench.py
step profile_c2 420 tools/profile_config.sh 2 r02_c2_final
step profile_c3 420 tools/profile_config.sh 3 r02_c3_final
step profile_c4 420 tools/profile_config.sh 4 r02_c4_final
step profile_c5 600 tools/profile_config.sh 5 r02_c5_final --burn-in 200
grep -v amdgpu $out/sink_rate.log $out/ob_profile.log
tail -n 1 $out/bench_c2.log | cut -c1-600
cat $out/session.log
