#!/bin/bash
# round 4, GPU session 29: soak -- the final kernels over a minute of timed GPU work per config (the default line times ~12 s)
set -o pipefail
out=gpurun_out/r04_s29
mkdir -p $out
( while true; do date +%T >> $out/heartbeat.txt; sleep 45; done ) & hb=$!
trap "kill $hb" EXIT
for cfg in 2 4 3; do
  timeout -k 10 500 python bench.py --config $cfg --cpu-seconds 0 --steps 200 --warmup 5 > $out/soak_c$cfg.json 2> $out/soak_c$cfg.err || { echo "config $cfg failed"; tail -3 $out/soak_c$cfg.err; exit 1; }
  tail -n 1 $out/soak_c$cfg.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("config %d: %d bench steps of %.1f ms (%.0f s timed): %.4g steps/s, acceptance %.3f" % (d["baseline_config"], d["steps"], d["ms_per_step"], d["steps"]*d["ms_per_step"]/1e3, d["value"], d["config"]["acceptance_rate_rank0"]))'
done | tee $out/soak.txt
