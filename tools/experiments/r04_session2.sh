#!/bin/bash
# round 4, GPU session 2: -m gpu suite (threshold split: the pulse prior in a helper wave), bench lines of configs 2
# and 4, and the timing-only owner-slack experiment at config 2 (tmp_exp/r04_*.so: build_dev([0],[4], extra=...)).
set -o pipefail
out=gpurun_out/r04_s2
mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1
rc=$?
tail -8 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err && tail -1 $out/bench_c2.json | cut -c1-300
timeout -k 10 200 python bench.py --config 4 > $out/bench_c4.json 2> $out/bench_c4.err && tail -1 $out/bench_c4.json | cut -c1-300
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 120 python bench.py --cpu-seconds 0 --no-calibrate --steps 10 --warmup 2 --launches-per-step 40 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"])')"; }
for rep in 1 2; do for v in base slack3 slack1 slack0; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2
done; done
