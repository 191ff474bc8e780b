#!/bin/bash
# GPU session 3 of round 2: the one-barrier round kernel (first run on hardware), then everything
set -o pipefail
out=gpurun_out/r02s3
mkdir -p $out
step() { # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "== $name" | tee -a $out/session.log
    timeout -k 10 $secs "$@" > $out/$name.log 2>&1
    local rc=$?
    echo "== $name rc=$rc" | tee -a $out/session.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; tail -20 $out/$name.log; exit 1; fi
    return 0
}
# a new kernel with its own barrier protocol: smallest case first, short leash
step ob_first 90 python -m pytest tests/test_gpu_one_barrier.py -x -q -k "test_one_barrier_equals_two_phase_kernel_and_oracle and simplesin"
tail -5 $out/ob_first.log
step ob_all 300 python -m pytest tests/test_gpu_one_barrier.py -q
tail -8 $out/ob_all.log
step pytest 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_one_barrier.py
tail -8 $out/pytest.log
step bench_c2 120 python bench.py --cpu-seconds 0
step bench_c2_classic 120 python bench.py --cpu-seconds 0 --flags 4
step bench_c3 120 python bench.py --config 3 --cpu-seconds 0
step bench_c4 120 python bench.py --config 4 --cpu-seconds 0
step bench_c4_classic 120 python bench.py --config 4 --cpu-seconds 0 --flags 4
step bench_c5 200 python bench.py --config 5 --cpu-seconds 0 --burn-in 200
step sink_rate 300 python tools/sink_rate.py
for f in bench_c2 bench_c2_classic bench_c3 bench_c4 bench_c4_classic bench_c5 sink_rate; do echo "--- $f"; tail -n 4 $out/$f.log | cut -c1-400; done
cat $out/session.log
