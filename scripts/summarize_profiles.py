#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into the small files kept under profiles/:
   profiles/<tag>_kernel_stats_<run>.csv   rocprofv3 --kernel-trace --stats summary, verbatim
   profiles/<tag>_pmc_summary.json         per-launch averages of every PMC counter for the
                                           dominant kernel + the corrected HBM traffic figure
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a
wide (16 B/lane) streaming read -> doubled; WRITE_SIZE is exact; both are in KiB."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
for run in ("stats_default", "stats_1stream"):
    found = sorted(glob.glob(os.path.join(src, run, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if found:  # the newest run only (gpurun merges into gpurun_out/, older runs' files stay)
        shutil.copy(found[-1], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, run[6:])))
summary = {"kernel": None, "counters": {}, "launches": {}}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_stream" in row["Kernel_Name"]:
                summary["kernel"] = row["Kernel_Name"].split("(redgpu::DevDfa")[0].strip()
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            summary["counters"][k] = sum(v) / len(v)
            summary["launches"][k] = len(v)
c = summary["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    summary["hbm_traffic_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    summary["hbm_traffic_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 FETCH_SIZE counts half "
                                   "of a 16-B/lane streaming read; separate --pmc passes")
if "SQ_INSTS_LDS" in c and c.get("SQ_INSTS_LDS"):
    summary["lds_cycles_per_wave_instruction"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"]
    summary["lds_conflict_cycles_per_wave_instruction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"]
json.dump(summary, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1))
