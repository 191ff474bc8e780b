#!/bin/bash
# round 4, GPU session 4: the owner's priority by phase (APEMOST_OWNER_PRIO_PHASE 0..3, configs 2 and 4), the whole
# -m gpu suite, the one host-executable profile (r04_host_profile.sh), HIP_VISIBLE_DEVICES probe.
set -o pipefail
out=gpurun_out/r04_s4
mkdir -p $out
run() { local tag=$1 lib=$2; shift 2
  APEMOST_HIP_LIB=$lib timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 "$@" > $out/b_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/b_$tag.log; return 1; }
  echo "$tag $(tail -n 1 $out/b_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g steps/s, launch %.1f us, calibration %.3f s" % (d["value"], d["roofline"]["launch_us"], d.get("calibration", {}).get("wall_s", 0)))')"; }
for rep in 1 2; do for v in pp0 pp1 pp2 pp3; do
run c2_$v $PWD/tmp_exp/r04_$v.so --config 2 --launches-per-step 60 || exit 1
run c4_$v $PWD/tmp_exp/r04_$v.so --config 4 --launches-per-step 200 || exit 1
done; done 2>&1 | tee $out/prio_phase.txt
timeout -k 10 700 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1
rc=$?
tail -6 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
HIP_VISIBLE_DEVICES=0,0 python -c "
from apemost_amd import capi
import ctypes as C
n = C.c_int(0)
print('HIP_VISIBLE_DEVICES=0,0 -> device_count rc', capi.lib().apemost_hip_device_count(C.byref(n)), 'count', n.value)
" > $out/visible_devices_probe.txt 2>&1
cat $out/visible_devices_probe.txt
bash tools/experiments/r04_host_profile.sh
