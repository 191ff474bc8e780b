#!/bin/bash
# A/B script behind a number in DESIGN.md.  The libraries it compares (tmp_exp/*.so, not tracked) are development
# builds: apemost_amd.build.build_dev(models, waves, out=..., extra=[-D switches]) of the commit that quotes the number.
set -o pipefail
out=gpurun_out/r02w8
mkdir -p $out
run() { local tag=$1; shift
  timeout -k 10 150 python bench.py --cpu-seconds 0 --burn-in 200 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"], d["config"]["waves_per_chain"], "n_swap", d["config"]["n_swap"])')"; }
for w in 4 8; do
run pulse128_w$w --config 4 --chains-per-gpu 128 --n-swap 15 --rounds-per-step 64 --waves $w
run pulse64_w$w --config 4 --chains-per-gpu 64 --n-swap 31 --rounds-per-step 32 --waves $w
run vrot128_w$w --config 5 --chains-per-gpu 128 --n-data 1024 --n-swap 15 --rounds-per-step 64 --waves $w
run vrot256_w$w --config 5 --chains-per-gpu 256 --n-data 1024 --n-swap 7 --rounds-per-step 64 --waves $w
run sine3_128_w$w --config 3 --chains-per-gpu 128 --n-data 1024 --n-swap 15 --rounds-per-step 64 --waves $w
run sine3_256_w$w --config 3 --chains-per-gpu 256 --n-data 1024 --n-swap 7 --rounds-per-step 64 --waves $w
done
