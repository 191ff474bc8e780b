#!/bin/bash
export TMPDIR=/tmp
for L in C B D; do
  out=gpurun_out/r03_ab/$L; mkdir -p $out
  export APEMOST_HIP_LIB=$PWD/apemost_amd/libapemost_hip_dev$L.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq -- python3 bench.py --config 5 --burn-in 100 --calib-iter-limit 600 --cpu-seconds 0 --steps 2 --warmup 1 > $out/sq.log 2>&1
done
python3 - <<'PY'
import csv,glob,statistics
for L in "CBD":
    rows=[]
    for f in glob.glob("gpurun_out/r03_ab/%s/sq/*/*_counter_collection.csv"%L):
        rows+=list(csv.DictReader(open(f)))
    by={}
    for r in rows:
        if "pt_calibrate_kernel" not in r["Kernel_Name"]: continue
        d=by.setdefault(r["Dispatch_Id"],{"us":(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,"vgpr":r["VGPR_Count"],"agpr":r["Accum_VGPR_Count"],"lds":r["LDS_Block_Size"]})
        d[r["Counter_Name"]]=d.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    for i,d in sorted(by.items(),key=lambda x:int(x[0])):
        wc=d.get("SQ_WAVE_CYCLES",1)
        print(L,i,"us %.0f"%d["us"],"vgpr",d["vgpr"],"agpr",d["agpr"],"valu_active %.3f"%(d.get("SQ_ACTIVE_INST_VALU",0)/wc),"wait_any %.3f"%(d.get("SQ_WAIT_ANY",0)/wc),"issue_wait %.3f"%(d.get("SQ_WAIT_INST_ANY",0)/wc),"insts/wave %.0f"%(d.get("SQ_INSTS_VALU",0)/max(d.get("SQ_WAVES",1),1)),"busy %.3g wavecyc %.3g"%(d.get("SQ_BUSY_CYCLES",0),wc))
PY
