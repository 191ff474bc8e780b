#!/bin/bash
# round 4, GPU session 1: the whole -m gpu suite with the round's first changes (calibration record barrier,
# LogProdT pair guard + reference-order fallback, full-size 1e-12 likelihoods, injected ties), then the bench
# lines of configs 2 and 4 on this box.
set -o pipefail
mkdir -p gpurun_out/r04_s1
python -m pytest tests -m gpu -x -q -s > gpurun_out/r04_s1/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/r04_s1/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r04_s1/bench_c2.json 2> gpurun_out/r04_s1/bench_c2.err && tail -1 gpurun_out/r04_s1/bench_c2.json | cut -c1-400
python bench.py --config 4 > gpurun_out/r04_s1/bench_c4.json 2> gpurun_out/r04_s1/bench_c4.err && tail -1 gpurun_out/r04_s1/bench_c4.json | cut -c1-400
