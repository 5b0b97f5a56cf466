"""Turns the raw rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/ (usage: collect_profiles.py <stats_dir> <pmc_prefix>)."""
import csv, collections, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats_dir, pmc_prefix = sys.argv[1], sys.argv[2]
math_name = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
rows = list(csv.DictReader(open(glob.glob(os.path.join(ROOT, "gpurun_out", stats_dir, "*", "*_kernel_stats.csv"))[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ddim   (MI355X, default arithmetic " + math_name + ", B=32 FFHQ-128 RL step)",
       "# 7 training steps are in the trace (2 warm-up + 4 timed + 1 per-op profile pass); durations in ns; Percentage of total GPU kernel time",
       f"# total kernel time {tot / 1e6:.1f} ms", f"{'Name':90s} {'Calls':>7s} {'TotalNs':>14s} {'AvgNs':>12s} {'Pct':>6s}"]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    out.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {r['TotalDurationNs']:>14s} {float(r['AverageNs']):12.0f} {float(r['Percentage']):6.2f}")
open(os.path.join(ROOT, "profiles", "r01_kernel_stats.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[2:12]))


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"]][0] += 1; a[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return a


af = agg(glob.glob(os.path.join(ROOT, "gpurun_out", pmc_prefix + "FETCH_SIZE", "*", "*_counter_collection.csv"))[0])
aw = agg(glob.glob(os.path.join(ROOT, "gpurun_out", pmc_prefix + "WRITE_SIZE", "*", "*_counter_collection.csv"))[0])
o = {"math": math_name, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim",
     "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size (MI355X_MICROARCH.md, HBM "
              "section): fetch_bytes = 2 x FETCH_SIZE x 1024. Calibration inside this very trace: gn_apply_stream_kernel reads and writes tensors of equal size "
              "and reports WRITE_SIZE ~ 2 x FETCH_SIZE.", "kernels": {}}
for k in af:
    n = af[k][0]; fk = af[k][1] / n; wk = aw[k][1] / max(aw[k][0], 1)
    o["kernels"][k] = {"dispatches": n, "fetch_size_kib_avg": round(fk, 1), "write_size_kib_avg": round(wk, 1), "hbm_bytes_per_launch": round((2 * fk + wk) * 1024)}
json.dump(o, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
lines = ["# " + o["source"], "# " + o["units"], f"{'kernel':70s} {'disp':>6s} {'FETCH KiB':>12s} {'WRITE KiB':>12s} {'HBM MB/launch':>14s}"]
for k, v in sorted(o["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:25]:
    lines.append(f"{k[:70]:70s} {v['dispatches']:6d} {v['fetch_size_kib_avg']:12.1f} {v['write_size_kib_avg']:12.1f} {v['hbm_bytes_per_launch'] / 1e6:14.1f}")
open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[2:8]))
