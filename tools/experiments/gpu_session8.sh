#!/bin/bash
set -o pipefail
out=gpurun_out/r02s8
mkdir -p $out
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $out/session.log; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc" | tee -a $out/session.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $out/session.log; tail -20 $out/$name.log; exit 1; fi; return 0; }
step bench_dist1 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --cpu-seconds 0 --force-dist
tail -n 3 $out/bench_dist1.log | cut -c1-300
step nccl_probe 300 python tools/nccl_probe.py
tail -n 6 $out/nccl_probe.log | cut -c1-300
step bench_c5 200 python bench.py --config 5 --cpu-seconds 0 --burn-in 200
tail -n 1 $out/bench_c5.log | cut -c1-260
step pytest 900 python -m pytest tests -q -m gpu
tail -4 $out/pytest.log
