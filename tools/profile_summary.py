#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs that tools/profile.sh left under gpurun_out/prof_<tag>/ into one text table
(per kernel: calls, total/avg duration, PMC counter sums).  Usage: profile_summary.py gpurun_out/prof_r01 > profiles/r01_summary.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:48]


def find(pass_name, suffix):
    hits = glob.glob(os.path.join(root, pass_name, "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


print(f"# rocprofv3 summary of {root}")
f = find("trace", "kernel_stats.csv")
if f:
    print("\n## kernel-trace --stats (kernel_stats.csv)")
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':34s} {'calls':>7s} {'total_ms':>12s} {'avg_us':>12s} {'pct':>7s}")
    for r in rows:
        name = short(r.get("Name", ""))
        if "k_" not in name:
            continue
        print(f"{name:34s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:12.3f} {float(r['AverageNs']) / 1e3:12.2f} {float(r['Percentage']):7.2f}")
f = find("trace", "kernel_trace.csv")
if f:
    agg = defaultdict(lambda: [0, 0.0, 0, 0, 0])
    for r in csv.DictReader(open(f)):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a[2] = int(r.get("VGPR_Count", 0) or 0)
        a[3] = int(r.get("SGPR_Count", 0) or 0)
        a[4] = int(r.get("LDS_Block_Size", 0) or 0)
    print("\n## kernel_trace.csv (per kernel: dispatches, total ms, VGPR, SGPR, LDS)")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if k.startswith("k_"):
            print(f"{k:34s} {a[0]:7d} {a[1]:12.3f} ms  avg {a[1] / a[0] * 1e3:10.2f} us  vgpr {a[2]:4d} sgpr {a[3]:4d} lds {a[4]:6d}")

for p in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(p):
        continue
    f = find(os.path.basename(p), "counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    if "--per-dispatch" in sys.argv:
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            per[(int(r["Dispatch_Id"]), short(r["Kernel_Name"]))][r["Counter_Name"]] = float(r["Counter_Value"])
        print(f"\n## {os.path.basename(p)} per dispatch")
        for (d, k), v in sorted(per.items()):
            if k.startswith("k_"):
                print(f"{d:5d} {k:26s} " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
        continue
    print(f"\n## {os.path.basename(p)} (counter sums over all dispatches of the kernel)")
    for k in sorted(agg, key=lambda kk: -sum(agg[kk].values())):
        if not k.startswith("k_"):
            continue
        print(f"{k:34s} dispatches {len(calls[k]):6d}  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(agg[k].items())))


# ---- --limiters OUT.json [--config JSON]: per kernel, what binds it, each figure recomputable from the table above ----
def limiters(root, out_path, config, source, frames=1):
    import json
    stats = {}
    f = find("trace", "kernel_trace.csv")
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        a = stats.setdefault(k, {"launches": 0, "ms": 0.0})
        a["launches"] += 1
        a["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    pmc = defaultdict(dict)
    for p in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        f = find(os.path.basename(p), "counter_collection.csv") if os.path.isdir(p) else None
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            pmc[k][r["Counter_Name"]] = pmc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    CLOCK_HZ, SIMDS, HBM_PEAK, L2_PEAK = 2.4e9, 1024, 8.0e12, 34.5e12      # MI355X_MICROARCH.md
    # kernel durations inside the GRBM pass itself (counter collection serialises dispatches and changes timing a little)
    grbm_ms = {}
    fg = find("pmc_grbm", "kernel_trace.csv")
    if fg:
        for r in csv.DictReader(open(fg)):
            grbm_ms[short(r["Kernel_Name"])] = grbm_ms.get(short(r["Kernel_Name"]), 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = {}
    for k, a in stats.items():
        c = pmc.get(k)
        if not c or a["ms"] < 1.0 or not k.startswith("k_"):
            continue
        t = a["ms"] * 1e-3
        e = {"launches": a["launches"], "avg_launch_ms": round(a["ms"] / a["launches"], 4)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH doubled (gfx950 tallies 128-B requests at 64 B: guide, "HBM").  The
            # doubling is calibrated for wide coalesced reads only: for these gathers it is an upper bound.
            hbm = 2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = round(hbm / a["launches"])
            e["hbm_traffic_frac"] = round(hbm / t / HBM_PEAK, 4)
        if "TCC_REQ_sum" in c:
            e["l2_bw_frac"] = round(c["TCC_REQ_sum"] * 128 / t / L2_PEAK, 4)       # <= 128 B per request: upper bound
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        clock = CLOCK_HZ
        if "GRBM_GUI_ACTIVE" in c:
            # rocprofv3 reports the sum over the 8 XCDs; per launch it reads high below ~0.3 ms, so take the kernel's total
            # (MI355X_MICROARCH.md "DVFS give-back").  The pmc_grbm pass has its own timing: use that pass's durations.
            t_g = grbm_ms.get(k, a["ms"]) * 1e-3
            clock = c["GRBM_GUI_ACTIVE"] / 8.0 / t_g
            e["clock_GHz_measured"] = round(clock / 1e9, 3)
        if "SQ_INSTS_VALU" in c:
            # a wave64 VALU instruction occupies its SIMD-32 for 2 clocks; 1024 SIMDs at the clock measured in the same
            # profile (GRBM_GUI_ACTIVE / 8 / kernel time), or the 2.4 GHz peak clock when that pass is missing
            e["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 2 / (SIMDS * clock * t), 4)
            e["valu_issue_frac_at_peak_clock"] = round(c["SQ_INSTS_VALU"] * 2 / (SIMDS * CLOCK_HZ * t), 4)
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            e["lanes_active"] = round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), 4)
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            e["mem_wait_share"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
            e["issue_wait_share"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        if "SQ_INSTS_SALU" in c and "SQ_INSTS_VALU" in c:
            e["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 4)
            # a CU has ONE scalar unit for its four SIMDs (MI355X_MICROARCH.md: "4 SIMD-32 vector units, 1 scalar unit"), one
            # instruction per clock: 256 CUs at the measured clock
            e["salu_issue_frac"] = round(c["SQ_INSTS_SALU"] / (256 * clock * t), 4)
        cand = {n: e[n] for n in ("hbm_traffic_frac", "l2_bw_frac", "valu_issue_frac", "salu_issue_frac") if n in e}
        if cand:
            e["binding"] = max(cand, key=cand.get)      # the throughput limit closest to its ceiling
        out[k] = e
    json.dump({"config": config, "frames_profiled": frames, "source": source, "peaks": {"hbm_Bps": HBM_PEAK, "l2_Bps": L2_PEAK, "simds": SIMDS, "clock_Hz": CLOCK_HZ},
               "kernels": out}, open(out_path, "w"), indent=1)


if "--limiters" in sys.argv:
    import json as _json
    i = sys.argv.index("--limiters")
    cfg = _json.loads(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else None
    src = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else root
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 1
    limiters(root, sys.argv[i + 1], cfg, src, frames)
