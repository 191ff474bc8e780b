#!/bin/bash
# A/B of the one-wave calibration kernel at BASELINE config 5's shard (pulse_vrot, 2048 chains x 65 536 points), a
# short calibration (ITER_LIMIT 2000: every chain stops after 2200 sweeps = 36.5 M likelihood evaluations, no tail):
#   A  -DAPEMOST_NO_OFFSET_SHORTCUT : single-parameter updates of the additive parameter walk the data vector too
#   B  the product source
# Prints wall seconds of the calibration and the steps/s of the stepping kernel behind it; with `sq` as first
# argument also one SQ pass per variant (VALU-active / parked shares of the calibration launches).
# Round 3 on one MI355X: A 5.92 s -> B 5.18 s with the logarithm's table still read through the vector cache;
# table in LDS + stage-major terms + 256-register bound: A 4.5 s, B 3.93 s.  (The historical variants quoted in
# DESIGN.md 15 -- the kernel with its record in registers, 264 + 8 AGPRs, 5.03 s; the same forced to one wave per
# SIMD, 7.47 s -- were builds of earlier commits of this round.)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python3 - <<'PY'
import os
from apemost_amd import build as b
b.build_dev([2], [1, 2, 4, 8], out=os.path.join(b.HERE, "libapemost_hip_devA.so"), extra=["-DAPEMOST_NO_OFFSET_SHORTCUT"])
b.build_dev([2], [1, 2, 4, 8], out=os.path.join(b.HERE, "libapemost_hip_devB.so"))
PY
args="--config 5 --burn-in 100 --calib-iter-limit 2000 --cpu-seconds 0 --steps 4 --warmup 1"
for L in A B A B; do
  APEMOST_HIP_LIB=$PWD/apemost_amd/libapemost_hip_dev$L.so python3 bench.py $args 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['calibration']
print('$L', 'calibration %.3f s' % k['wall_s'], k['evaluations_counted_on_device'], 'evaluations,', k['segments'], 'segments; stepping kernel %.3g steps/s' % b['value'])"
done
if [ "$1" = "sq" ]; then
  for L in A B; do
    out=gpurun_out/calib_ab/$L; mkdir -p $out
    APEMOST_HIP_LIB=$PWD/apemost_amd/libapemost_hip_dev$L.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
        --kernel-trace --output-format csv -d $out/sq -- python3 bench.py --config 5 --burn-in 100 --calib-iter-limit 600 --cpu-seconds 0 --steps 2 --warmup 1 > $out/sq.log 2>&1
  done
  python3 - <<'PY'
import csv, glob
for L in "AB":
    rows = []
    for f in glob.glob("gpurun_out/calib_ab/%s/sq/*/*_counter_collection.csv" % L):
        rows += list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        if "pt_calibrate_kernel" not in r["Kernel_Name"]:
            continue
        d = by.setdefault(r["Dispatch_Id"], {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    for i, d in sorted(by.items(), key=lambda x: int(x[0])):
        wc = d.get("SQ_WAVE_CYCLES", 1)
        print(L, i, "%.0f us" % d["us"], "VALU-active %.3f" % (d.get("SQ_ACTIVE_INST_VALU", 0) / wc), "parked %.3f" % (d.get("SQ_WAIT_ANY", 0) / wc),
              "issue-stall %.3f" % (d.get("SQ_WAIT_INST_ANY", 0) / wc), "VALU instructions per wave %.0f" % (d.get("SQ_INSTS_VALU", 0) / max(d.get("SQ_WAVES", 1), 1)))
PY
fi
rm -f apemost_amd/libapemost_hip_devA.so apemost_amd/libapemost_hip_devB.so
