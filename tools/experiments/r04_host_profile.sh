#!/bin/bash
# one run that can falsify the attribution of the exit-time SIGSEGV under rocprofv3
# (profiles/README.md "The rocprofv3 abort at config 4").  A 300-chain pulse ladder -- 8-wave one-barrier
# workgroups, two per CU on some CUs: multi-round launches through hipLaunchCooperativeKernel -- run through the
# C HOST EXECUTABLE: no torch in the process, one HIP / HSA runtime.  The program is what follows `--`.
# Runs ONCE; do not loop it.
set -o pipefail
out=$PWD/gpurun_out/r04_hostprof
work=$out/work
mkdir -p $work
python - "$work" <<'PY'
import sys
import numpy as np
sys.path.insert(0, ".")
from apemost_amd import workloads as wl
w = wl.pulse(n_data=1024, n_chain=300)
open(sys.argv[1] + "/data", "w").write("".join("%.17e\t%.17e\n" % tuple(r) for r in w.data))
names = ["lifetime", "offset", "f1", "h1", "f2", "h2"]
open(sys.argv[1] + "/params", "w").write("".join("%.15e\t%.15e\t%.15e\t%s\t-1\n" % (s0, lo, hi, nm)
                                                  for s0, lo, hi, nm in zip(w.start, w.pmin, w.pmax, names)))
PY
exe=$out/pulse300.exe
make -s -C apemost_amd/host OUT=$exe APP=$PWD/apemost_amd/host/examples/pulse_model.c \
  "CCFLAGS=-DN_BETA=300 -DMAX_ITERATIONS=60000 -DBURN_IN_ITERATIONS=400 -DSKIP_CALIBRATE_ALLCHAINS" || exit 1
cd $work
export APEMOST_DUMP=binary,thin:100 APEMOST_SEED=5
timeout -k 10 120 $exe calibrate_first > $out/first.log 2>&1 || { echo calibrate_first failed; tail -5 $out/first.log; exit 1; }
timeout -k 10 200 $exe calibrate_rest > $out/rest.log 2>&1 || { echo calibrate_rest failed; tail -5 $out/rest.log; exit 1; }
# un-profiled run first: its exit code is the control
timeout -k 10 120 $exe run > $out/run_plain.log 2>&1; echo "plain run: exit $?" | tee $out/exit_codes.txt
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- $exe run > $out/run_prof.log 2>&1
echo "rocprofv3 run: exit $?" | tee -a $out/exit_codes.txt
tail -3 $out/run_prof.log
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -6
