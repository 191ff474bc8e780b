#!/bin/bash
out=gpurun_out/r02s5
mkdir -p $out
for v in estrin horner; do
  APEMOST_HIP_LIB=$PWD/apemost_amd/libapemost_hip_dev_$v.so timeout -k 10 100 python bench.py --cpu-seconds 0 --steps 60 > $out/$v.log 2>&1 || true
  echo "$v: $(tail -n 1 $out/$v.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.4g" % d["value"])')"
done
APEMOST_HIP_LIB=$PWD/apemost_amd/libapemost_hip_dev_estrin.so timeout -k 10 100 python -m pytest tests/test_gpu_one_barrier.py -q -k "simplesin and 8" 2>&1 | tail -2
timeout -k 10 100 python tools/ob_profile.py simplesin 128 1024 8 | grep -v amdgpu
