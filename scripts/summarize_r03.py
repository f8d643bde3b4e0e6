#!/usr/bin/env python3
"""Turns gpurun_out/prof_r03_c<C>_<D>/ (rocprofv3 CSVs of scripts/profile_r03.sh) into the small
files kept under profiles/:
   profiles/r03_kernel_stats_config<C>_<D>_<run>.csv    rocprofv3 --kernel-trace --stats, verbatim
                                                        (run = default: as bench.py issues the steps;
                                                        1percall: config 1 with one step per call)
   profiles/r03_timeline_config<C>_<D>_<run>.json       span of the dominant kernel's trace (first
                                                        start -> last end) / launches: the per-step
                                                        time of overlapped launches, reproducible
   profiles/r03_pmc_config<C>_<D>.json                  per-launch averages of every PMC counter for
                                                        the dominant kernel + corrected HBM traffic,
                                                        LDS busy share, L2 hit rate
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a
wide (16 B/lane) streaming read -> doubled; WRITE_SIZE is exact; both are in KiB."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
SKIP = ("k_diag", "rocclr", "at::", "k_walked", "elementwise", "distribution", "k_pack", "k_unpack",
        "k_visits")


def newest(pattern):
    found = sorted(glob.glob(pattern), key=os.path.getmtime)
    return found[-1] if found else None


def dominant(stats_csv):
    best = None
    for row in csv.DictReader(open(stats_csv)):
        if any(s in row["Name"] for s in SKIP):
            continue
        if best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"]):
            best = row
    return best


for src in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_r03_c*"))):
    tag = os.path.basename(src)[len("prof_r03_c"):]          # e.g. 1_syn256
    cfg, dfa = tag.split("_", 1)
    name = "config%s_%s" % (cfg, dfa)
    dom = None
    for run in ("stats_default", "stats_1percall"):
        f = newest(os.path.join(src, run, "*", "*_kernel_stats.csv"))
        if not f:
            continue
        shutil.copy(f, os.path.join(dst, "r03_kernel_stats_%s_%s.csv" % (name, run[6:])))
        d = dominant(f)
        if run == "stats_default":
            dom = d
        tr = newest(os.path.join(src, run, "*", "*_kernel_trace.csv"))
        if tr and d:
            rows = [r for r in csv.DictReader(open(tr)) if r["Kernel_Name"] == d["Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            # runs of launches without a host-side pause (> 1 ms between a start and the latest end)
            runs, cur, last_end = [], [], 0
            for r in rows:
                if cur and int(r["Start_Timestamp"]) - last_end > 1_000_000:
                    runs.append(cur)
                    cur = []
                cur.append(r)
                last_end = max(last_end, int(r["End_Timestamp"]))
            if cur:
                runs.append(cur)
            out = []
            for run_rows in runs:
                if len(run_rows) < 10:
                    continue
                span = max(int(r["End_Timestamp"]) for r in run_rows) - int(run_rows[0]["Start_Timestamp"])
                avg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in run_rows) / len(run_rows)
                out.append({"launches": len(run_rows), "span_ns": span,
                            "ns_per_launch_in_span": round(span / len(run_rows), 1),
                            "average_kernel_duration_ns": round(avg, 1),
                            "overlapped": bool(avg > 1.15 * span / len(run_rows))})
            json.dump({"kernel": d["Name"].split("(redgpu::DevDfa")[0].strip(), "runs": out,
                       "note": "runs of the dominant kernel in the trace (split at host pauses > 1 ms): "
                               "span / launches is the per-step time; in an overlapped run (several "
                               "streams) it is smaller than the average kernel duration"},
                      open(os.path.join(dst, "r03_timeline_%s_%s.json" % (name, run[6:])), "w"), indent=1)
    if not dom:
        continue
    kname = dom["Name"]
    summary = {"kernel": kname.split("(redgpu::DevDfa")[0].strip(), "average_ns": float(dom["AverageNs"]),
               "counters": {}, "launches": {}}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        f = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"] == kname:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            v = v[len(v) // 4:]
            summary["counters"][k] = sum(v) / len(v)
            summary["launches"][k] = len(v)
    c = summary["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        summary["hbm_traffic_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        summary["hbm_traffic_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 FETCH_SIZE counts half "
                                       "of a 16-B/lane streaming read; separate --pmc passes")
    if c.get("SQ_INSTS_LDS"):
        summary["lds_cycles_per_wave_instruction"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"]
        summary["lds_conflict_cycles_per_wave_instruction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"]
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_LDS_IDX_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; LDS cycles are summed over 256 CUs
        summary["lds_busy_share_of_launch"] = (c["SQ_LDS_IDX_ACTIVE"] / 256) / (c["GRBM_GUI_ACTIVE"] / 8)
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and \
            c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        summary["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("TCC_REQ_sum"):
        summary["l2_requests_per_us"] = c["TCC_REQ_sum"] / (float(dom["AverageNs"]) / 1e3)
    # input bytes of one launch of the dominant kernel, from the shape of the config
    # (config 3: the seeded 2^23 ragged lines of 32..256 B bench.py generates - its
    # roofline.input_bytes_per_launch)
    shape = {"1": (1 << 20) * 64, "2": (1 << 21) * 4096, "3": 1208234691,
             "4": (1 << 16) * (1 << 16)}.get(cfg)
    bpl = 20 if (cfg == "1" and "multi" in kname) else 1
    summary["batches_per_launch"] = bpl
    if shape:
        summary["input_bytes_per_launch"] = shape * bpl
        if c.get("SQ_INSTS_VALU"):
            # a wave-instruction does one lane-line's byte step for 64 lanes
            summary["valu_per_input_byte"] = c["SQ_INSTS_VALU"] * 64 / (shape * bpl)
        if c.get("SQ_INSTS_SALU"):
            summary["salu_per_input_byte"] = c["SQ_INSTS_SALU"] * 64 / (shape * bpl)
        if c.get("SQ_INSTS_LDS"):
            summary["lds_instructions_per_input_byte"] = c["SQ_INSTS_LDS"] * 64 / (shape * bpl)
        if "hbm_traffic_bytes_per_launch" in summary:
            summary["hbm_traffic_over_input"] = summary["hbm_traffic_bytes_per_launch"] / (shape * bpl)
    summary["schema"] = ("r03: kernel, average_ns, counters{} (per-launch averages), launches{}, "
                         "hbm_traffic_bytes_per_launch, lds_*, l2_*, batches_per_launch, "
                         "input_bytes_per_launch, valu/salu/lds per input byte - the same keys for every kernel")
    json.dump(summary, open(os.path.join(dst, "r03_pmc_%s.json" % name), "w"), indent=1)
    print(name, json.dumps({k: v for k, v in summary.items() if k not in ("counters", "launches")}))
