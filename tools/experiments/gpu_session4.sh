#!/bin/bash
set -o pipefail
out=gpurun_out/r02s4
mkdir -p $out
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $out/session.log; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc" | tee -a $out/session.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; tail -20 $out/$name.log; exit 1; fi; return 0; }
step ob_first 90 python -m pytest tests/test_gpu_one_barrier.py -x -q -k "test_one_barrier_equals_two_phase_kernel_and_oracle and simplesin"
step ob_all 300 python -m pytest tests/test_gpu_one_barrier.py -q
tail -3 $out/ob_all.log
step ob_profile 120 python tools/ob_profile.py simplesin 128 1024 8
step ob_profile4 120 python tools/ob_profile.py pulse 256 1024 4
step bench_c2 120 python bench.py --cpu-seconds 0
step bench_c4 120 python bench.py --config 4 --cpu-seconds 0
grep -v amdgpu $out/ob_profile.log $out/ob_profile4.log
for f in bench_c2 bench_c4; do tail -n 1 $out/$f.log | cut -c1-330; done
