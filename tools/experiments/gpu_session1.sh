#!/bin/bash
# GPU session 1 of round 2: full GPU test suite, launch-policy comparison, probes, profiles of every config.
# A step that is killed at its limit (124/137) ends the session: no further GPU step is started.
set -o pipefail
out=gpurun_out/r02s1
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
step pytest 600 python -m pytest tests -q -m gpu -x
tail -5 $out/pytest.log
step fp64_probe 60 ./tools/fp64_probe.exe
step bench_c2 120 python bench.py
step bench_c2_coop 120 python bench.py --flags 2 --cpu-seconds 0
step bench_c2_single 120 python bench.py --flags 1 --cpu-seconds 0
step bench_c4_full 120 python bench.py --config 4 --chains-per-gpu 2048 --cpu-seconds 0
step stamps_w0 120 python tools/stamp_profile.py 8
APEMOST_STAMP_WAVE=-1 step stamps_tl 120 python tools/stamp_profile.py 8
for c in 2 3 4 5; do
    step profile_c$c 420 tools/profile_config.sh $c r02_c${c}_base
done
tail -3 $out/bench_c2*.log $out/bench_c4_full.log
cat $out/session.log
