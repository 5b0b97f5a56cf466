"""Aggregates rocprofv3 counter_collection CSVs (per dispatch) into one JSON per pass: kernel -> {dispatches, counter: mean per dispatch},
plus the kernel-trace resource columns (VGPR / LDS / grid).  Usage: summarize_pmc.py <gpurun_out dir> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(out_dir, tag + "_pmc*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            meta.setdefault(k, {c: r.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")})
    res = {}
    for k in agg:
        n = len(disp[k])
        res[k] = {"dispatches": n, **{c: v / n for c, v in agg[k].items()}, "resources": meta[k]}
    json.dump(res, open(d + ".json", "w"), indent=1)
    for f in files:
        os.remove(f)
    print("summarised", d, len(res), "kernels")
