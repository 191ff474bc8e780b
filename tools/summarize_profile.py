#!/usr/bin/env python3
"""Condenses the raw rocprofv3 output of tools/profile_config.sh (gpurun_out/<tag>/) into the small
files kept under profiles/: <tag>_kernel_stats.csv (the --stats summary as rocprofv3 wrote it) and
<tag>_counters.json (per-launch medians of the dominant kernel: HBM bytes with the gfx950
correction of MI355X_MICROARCH.md, SQ counters and the ratios read from them)."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_rows(d):
    rows = []
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def dominant(rows, family="pt_round"):
    """the kernel with the largest total duration among the kernels of one family"""
    tot = {}
    for r in rows:
        k = r["Kernel_Name"]
        if family in k:
            tot[k] = tot.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return max(tot, key=tot.get) if tot else None


def per_launch(rows, kernel, min_us=0.0):
    """counter -> list of per-dispatch values (dispatches of `kernel`; the 0-step finalise launches
    are dropped by duration)"""
    by = {}
    durs = {}
    for r in rows:
        if r["Kernel_Name"] != kernel:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        durs[r["Dispatch_Id"]] = d
        by.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    if not durs:
        return {}, 0.0, 0
    med = statistics.median(durs.values())
    keep = {i for i, d in durs.items() if d >= 0.5 * med}
    out = {c: [v for i, v in vals.items() if i in keep] for c, vals in by.items()}
    return out, statistics.median([durs[i] for i in keep]), len(keep)


def family_totals(rows, family):
    """every dispatch of the kernels of one family together: launches, total duration, total SQ_INSTS_VALU, and the
    same per kernel -- the issue roofline of a PHASE made of many different launches (the calibration:
    workgroup shapes change as chains finish)"""
    dur, insts, name = {}, {}, {}
    for r in rows:
        k = r["Kernel_Name"]
        if family not in k:
            continue
        i = r["Dispatch_Id"]
        dur[i] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name[i] = k
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            insts[i] = insts.get(i, 0.0) + float(r["Counter_Value"])
    if not dur or not insts:
        return None
    per = {}
    for i, k in name.items():
        e = per.setdefault(k, {"launches": 0, "total_us": 0.0, "valu_wave_insts": 0.0})
        e["launches"] += 1
        e["total_us"] += dur[i]
        e["valu_wave_insts"] += insts.get(i, 0.0)
    total_us, total_insts = sum(dur.values()), sum(insts.values())
    peak = 1024 * 2.4e9 / 4      # wave-instructions per second: 1024 SIMDs, one per four cycles (bench.py)
    for e in per.values():
        e["valu_issue_frac"] = e["valu_wave_insts"] / (e["total_us"] * 1e-6) / peak
    return {"launches": len(dur), "total_us": total_us, "valu_wave_insts": total_insts,
            "valu_issue_frac": total_insts / (total_us * 1e-6) / peak,
            "note": "SQ_INSTS_VALU summed over every launch of the family / their summed durations / (1024 SIMDs x 2.4 GHz / 4)",
            "by_kernel": per}


def main():
    """summarize_profile.py <tag> <config> [family [suffix]]: family = substring that selects the kernels
    (default pt_round; pt_calibrate for the calibration launches of the same runs), suffix = inserted
    into the output file name"""
    tag, cfg = sys.argv[1], sys.argv[2]
    family = sys.argv[3] if len(sys.argv) > 3 else "pt_round"
    suffix = ("_" + sys.argv[4]) if len(sys.argv) > 4 else ""
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    out = {"config": int(cfg), "tag": tag,
           "command": "tools/profile_config.sh %s %s  (rocprofv3 ... -- python3 bench.py --config %s --cpu-seconds 0 --steps 4 --warmup 1 --launches-per-step 60 --no-torch)" % (cfg, tag, cfg)}
    try:
        out["bench_line"] = json.loads(open(os.path.join(src, "bench_short.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    kernel = None
    for name in ("sq", "fetch", "write"):
        rows = counter_rows(os.path.join(src, name))
        if not rows:
            continue
        kernel = kernel or dominant(rows, family)
        vals, med_us, n = per_launch(rows, kernel)
        out.setdefault("kernel", kernel)
        out[name] = {"launches": n, "median_launch_us": med_us,
                     "median": {c: statistics.median(v) for c, v in vals.items() if v}}
        r0 = next((r for r in rows if r["Kernel_Name"] == kernel), None)
        if r0:
            out["registers"] = {"vgpr": int(r0["VGPR_Count"]), "agpr": int(r0["Accum_VGPR_Count"]),
                                "sgpr": int(r0["SGPR_Count"]), "lds_bytes": int(r0["LDS_Block_Size"]),
                                "scratch": int(r0["Scratch_Size"]), "workgroup": int(r0["Workgroup_Size"]),
                                "grid": int(r0["Grid_Size"])}
    f = out.get("fetch", {}).get("median", {}).get("FETCH_SIZE")
    w = out.get("write", {}).get("median", {}).get("WRITE_SIZE")
    if f is not None and w is not None:
        # FETCH_SIZE/WRITE_SIZE are in KB; gfx950 tallies wide coalesced reads at half their bytes
        out["hbm_bytes_per_launch"] = (2 * f + w) * 1024
        out["hbm_correction"] = "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B), WRITE_SIZE exact; KB -> bytes"
    if family != "pt_round":
        tot = family_totals(counter_rows(os.path.join(src, "sq")), family)
        if tot:
            out["phase_issue_roofline"] = tot
    sq = out.get("sq", {}).get("median", {})
    if sq:
        wc = sq.get("SQ_WAVE_CYCLES", 0)
        out["sq_ratios"] = {
            "valu_active_share_of_wave_cycles": sq.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else None,
            "issue_stall_share_of_wave_cycles": sq.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
            "parked_share_of_wave_cycles (s_waitcnt/barrier)": sq.get("SQ_WAIT_ANY", 0) / wc if wc else None,
            "valu_insts_per_wave": sq.get("SQ_INSTS_VALU", 0) / sq["SQ_WAVES"] if sq.get("SQ_WAVES") else None,
            "lds_insts_per_wave": sq.get("SQ_INSTS_LDS", 0) / sq["SQ_WAVES"] if sq.get("SQ_WAVES") else None,
            "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, summed over all waves",
        }
    if sq.get("SQ_INSTS_VALU"):
        out["valu_wave_insts_per_launch"] = sq["SQ_INSTS_VALU"]
    with open(os.path.join(dst, tag + suffix + "_counters.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    # what bench.py quotes as roofline.traffic_profiled / the instruction count of its issue roofline
    bl = out.get("bench_line", {}).get("config")
    if family == "pt_round" and bl and "hbm_bytes_per_launch" in out and len(sys.argv) <= 3:
        reg = os.path.join(dst, "pmc_traffic.json")
        key = "%s/%d/%d/%d/%d/%s" % (out["bench_line"]["metric"].split(" on ")[1].split(",")[0], bl["chains_per_gpu"], bl["n_data"],
                                     bl["n_swap"], bl["rounds_per_step"], "samples")
        entries = [e for e in json.load(open(reg)) if e.get("workload_key") != key or e.get("waves_per_chain", bl["waves_per_chain"]) != bl["waves_per_chain"]]
        entries.append({"tag": tag + "_counters.json", "source": out["command"] + ", MI355X", "kernel": out.get("kernel"),
                        "workload_key": key, "waves_per_chain": bl["waves_per_chain"],
                        "hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "correction": out.get("hbm_correction"),
                        "valu_wave_insts_per_launch": out.get("valu_wave_insts_per_launch"),
                        # bench.py compares this with the sources it runs on (profile_kernel_sources_changed)
                        "kernel_sources_sha1": _bench_module().kernel_sources_sha1()})
        json.dump(entries, open(reg, "w"), indent=1)
    # the calibration phase of the same runs: what bench.py quotes as calibration.valu_issue_frac_profiled
    if family == "pt_calibrate" and bl and "phase_issue_roofline" in out:
        reg = os.path.join(dst, "pmc_traffic.json")
        key = "%s/%d/%d/calibration" % (out["bench_line"]["metric"].split(" on ")[1].split(",")[0], bl["chains_per_gpu"], bl["n_data"])
        entries = [e for e in json.load(open(reg)) if e.get("workload_key") != key]
        t = out["phase_issue_roofline"]
        entries.append({"tag": tag + suffix + "_counters.json", "source": out["command"] + ", MI355X", "workload_key": key,
                        "launches": t["launches"], "seconds_profiled": t["total_us"] * 1e-6, "valu_wave_insts": t["valu_wave_insts"],
                        "valu_issue_frac": t["valu_issue_frac"], "kernel_sources_sha1": _bench_module().kernel_sources_sha1()})
        json.dump(entries, open(reg, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("kernel", "registers", "hbm_bytes_per_launch", "sq_ratios", "phase_issue_roofline") if k in out}, indent=1))


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("apemost_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    main()
