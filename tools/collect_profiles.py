"""Turns the raw outputs of one tools/gpu_round.sh visit (gpurun_out/<tag>_*) into the committed summaries under profiles/.
usage: collect_profiles.py <tag> <round-label, e.g. r02>"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
cmd = "python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ddim"

# ---- kernel stats
rows = list(csv.DictReader(open(glob.glob(os.path.join(G, tag + "_stats", "*", "*_kernel_stats.csv"))[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"# rocprofv3 --kernel-trace --stats -- {cmd}   (MI355X, default arithmetic f16x3, B=32 FFHQ-128 RL step)",
       "# 7 training steps are in the trace (2 warm-up + 4 timed + 1 per-op profile pass); durations in ns; Pct of total GPU kernel time",
       f"# total kernel time {tot / 1e6:.1f} ms", f"{'Name':90s} {'Calls':>7s} {'TotalNs':>14s} {'AvgNs':>12s} {'Pct':>6s}"]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:50]:
    out.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {r['TotalDurationNs']:>14s} {float(r['AverageNs']):12.0f} {float(r['Percentage']):6.2f}")
open(os.path.join(P, f"{rnd}_kernel_stats.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[2:14]))

# ---- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
F = json.load(open(os.path.join(G, tag + "_pmc_FETCH_SIZE.json")))
W = json.load(open(os.path.join(G, tag + "_pmc_WRITE_SIZE.json")))
o = {"math": "f16x3", "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim",
     "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size (MI355X_MICROARCH.md, HBM "
              "section): fetch_bytes = 2 x FETCH_SIZE x 1024.", "kernels": {}}
for k in F:
    fk, wk = F[k].get("FETCH_SIZE", 0.0), W.get(k, {}).get("WRITE_SIZE", 0.0)
    o["kernels"][k] = {"dispatches": F[k]["dispatches"], "fetch_size_kib_avg": round(fk, 1), "write_size_kib_avg": round(wk, 1), "hbm_bytes_per_launch": round((2 * fk + wk) * 1024)}
json.dump(o, open(os.path.join(P, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
lines = ["# " + o["source"], "# " + o["units"], f"{'kernel':70s} {'disp':>6s} {'FETCH KiB':>12s} {'WRITE KiB':>12s} {'HBM MB/launch':>14s}"]
for k, v in sorted(o["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:25]:
    lines.append(f"{k[:70]:70s} {v['dispatches']:6d} {v['fetch_size_kib_avg']:12.1f} {v['write_size_kib_avg']:12.1f} {v['hbm_bytes_per_launch'] / 1e6:14.1f}")
open(os.path.join(P, f"{rnd}_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")

# ---- registers / scratch / LDS per kernel: from the COMPILER's resource remarks (tests/test_kernel_resources_cpu.resource_table), not from the
# kernel trace -- rocprofv3 reports Accum_VGPR_Count = 0 and LDS_Block_Size = 0 for kernels with a unified register file / dynamic LDS (round 3's
# table showed AGPR 0 and LDS 0 for conv3x3r, which hid a 68-byte-per-lane spill)
import subprocess
sys.path.insert(0, ROOT)
from tests.test_kernel_resources_cpu import HOT, resource_table
from pdae_amd import build as _B
RES = {}
for src in HOT:
    if not os.path.exists(os.path.join(_B.CSRC, src)):
        continue
    for k in resource_table(src):
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        RES[name] = k
# dynamic LDS of the kernels that size it at launch (bytes per workgroup; launch code of the respective .hip file)
DYN_LDS = {"conv3x3y_kernel<4": 2 * 2 * 31104 + 4 * 2 * 32 * 36 * 4 + 1024, "conv3x3y_kernel<2": 2 * 2 * 31104 + 4 * 2 * 32 * 36 * 4 + 1024, "conv3x3y_kernel<1": 2 * 1 * 31104 + 4 * 2 * 32 * 36 * 4 + 1024,
           "conv3x3r_kernel<4": 2 * 2 * 30720 + 4 * 32 * 36 * 4, "conv3x3r_kernel<1": 2 * 1 * 30720 + 4 * 32 * 36 * 4, "conv3x3r_kernel<2": 2 * 2 * 30720 + 4 * 32 * 36 * 4,
           "conv3x3p_kernel<4, 8": 2 * 200 * 80, "conv3x3p_kernel<4, 16": 2 * 360 * 80, "conv3x3v_kernel<4": 2 * (2 * 2 * (180 * 32 + 32) + 2 * 128 * 64) * 2, "conv3x3w_kernel<4, false": (2 * 180 * 32 + 2 * 128 * 64) * 2 + 256 * 16, "conv3x3w_kernel<4, true": (2 * 200 * 32 + 2 * 128 * 64) * 2 + 256 * 16,
           "conv1x1_kernel<4": 2 * 128 * 72 * 2}


def res_of(kname):
    r = RES.get(kname)
    if r is None:
        return None
    lds = r["lds"]
    for pre, b in DYN_LDS.items():
        if kname.startswith("void " + pre):
            lds = max(lds, b)
    if kname.startswith("void conv3x3y_kernel<") and ", 1>(" in kname:        # 8-row tiles (RH = 1): 10 x 36 positions per plane instead of 18 x 36
        planes = 1 if kname.startswith("void conv3x3y_kernel<1") else 2
        lds = 2 * planes * 17280 + 4 * 2 * 32 * 36 * 4 + 1024
    return r["vgpr"], r["agpr"], r["scratch"], lds


# ---- SQ counters of the MFMA kernels
A = json.load(open(os.path.join(G, tag + "_pmcA.json")))
B = json.load(open(os.path.join(G, tag + "_pmcB.json")))
hdr = ["# rocprofv3 --pmc <8 SQ counters + GRBM_GUI_ACTIVE> --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim   (two passes)",
       "# per-launch means.  MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)  [busy cycles = 32 per 32x32x16 MFMA];",
       "# WAIT_ANY = parked in s_waitcnt / barrier, WAIT_INST_ANY = issue stalls (MFMA pipe / dependencies), ACTIVE = instruction issue -- fractions of SQ_WAVE_CYCLES;",
       "# LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; VGPR / AGPR / scratch bytes per lane / LDS bytes per workgroup (static + dynamic) from hipcc's",
       "# -Rpass-analysis=kernel-resource-usage remarks of the shipped sources (the kernel trace reports 0 for AGPRs and dynamic LDS); '-' = not a hot source",
       f"{'kernel':58s} {'disp':>5s} {'MFMAutil':>8s} {'WAIT_ANY':>8s} {'WAIT_INST':>9s} {'ACTIVE':>7s} {'LDSconf':>7s} {'VALU/MFMA':>9s} {'LDS/MFMA':>8s} {'VGPR':>5s} {'AGPR':>5s} {'scr':>4s} {'LDS':>7s} {'WG':>4s}"]
for k, v in sorted(A.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["dispatches"]):
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    b = B.get(k, {})
    wc = v["SQ_WAVE_CYCLES"]
    util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
    nm = max(b.get("SQ_INSTS_MFMA", 0.0), 1.0)
    r = v["resources"]
    cr = res_of(k)
    cols = (f"{cr[0]:5d} {cr[1]:5d} {cr[2]:4d} {cr[3]:7d}" if cr else f"{'-':>5s} {'-':>5s} {'-':>4s} {'-':>7s}")
    hdr.append(f"{k[:58]:58s} {v['dispatches']:5d} {util:8.3f} {v['SQ_WAIT_ANY'] / wc:8.3f} {v['SQ_WAIT_INST_ANY'] / wc:9.3f} {v['SQ_ACTIVE_INST_ANY'] / wc:7.3f} "
               f"{v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):7.3f} {(b.get('SQ_INSTS_VALU', 0) - nm) / nm:9.2f} {b.get('SQ_INSTS_LDS', 0) / nm:8.2f} "
               f"{cols} {r['Workgroup_Size']:>4s}")
open(os.path.join(P, f"{rnd}_pmc_sq.txt"), "w").write("\n".join(hdr) + "\n")
print("\n".join(hdr[4:12]))

# ---- bench line, test log
for src, dst in ((tag + "_bench.json", f"{rnd}_bench_default.json"), (tag + "_bench.log", f"{rnd}_bench_default_run.log"), (tag + "_tests.txt", f"{rnd}_gpu_tests.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
