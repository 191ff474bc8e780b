#!/bin/bash
# A/B script behind a number in DESIGN.md 16 (APEMOST_SCALAR_ADVANCE: the data loop's base address on the scalar unit).
# tmp_exp/sa0.so, sa1.so: apemost_amd.build.build_dev([2, 3], [1, 2, 4, 8], out=..., extra=[-DAPEMOST_SCALAR_ADVANCE=0|1]).
set -o pipefail
out=gpurun_out/r03scalaradv
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 300 python bench.py --cpu-seconds 0 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; exit 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g"%d["value"])')"; }
for rep in 1 2; do for v in ${VARIANTS:-sa0 sa1}; do
run c3_$v $PWD/tmp_exp/$v.so --config 3 --no-calibrate
run ${OTHER:-c5}_$v $PWD/tmp_exp/$v.so --config ${OTHERCFG:-5} --no-calibrate
done; done
