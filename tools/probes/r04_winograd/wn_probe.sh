#!/bin/bash
# Winograd F(2x2,3x3) kernel: timing probes with cycle / clock / wait counters on one box: product, then every pdae_amd/lib/probe_wn_*/ build
# (tools/probe_build.py wn_).  One shape (128x128 128->128, B=32; SHAPE_C=256 for 256->128), forward only.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/wn_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import torch
from pdae_amd import hip as H
N, S, Cin, Cout = 32, 128, int(os.environ.get("SHAPE_C", "128")), 128
x = torch.randn(N, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5; b = torch.randn(Cout, device="cuda")
y = torch.empty(N, S, S, Cout, device="cuda")
c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
wp = torch.empty(H.wino_wprep_bytes(c) // 4, device="cuda"); H.wino_wprep(c, w, wp)
f = lambda: H.wino_fwd(c, x, wp, b, y)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("ms", e0.elapsed_time(e1) / 10)
PY
export R
for lib in product $(ls -d pdae_amd/lib/probe_wn_* 2>/dev/null); do
  n=$(basename $lib)
  [ $lib = product ] && unset PDAE_HIP_LIB || export PDAE_HIP_LIB=$R/$lib/libpdae_hip.so
  ms=$(timeout 60 python /tmp/wn_one.py 2>/dev/null | grep ms)
  (cd /tmp && timeout 90 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/wp_$n -- python /tmp/wn_one.py > /dev/null 2>&1)
  python - <<PY
import csv, glob
info = {}
for f in glob.glob("$O/wp_$n/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino" in r["Kernel_Name"] and "wprep" not in r["Kernel_Name"]:
            info[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
cnt = {}
for f in glob.glob("$O/wp_$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] in info:
            cnt.setdefault(r["Dispatch_Id"], {}); cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = sorted(((info[d], c) for d, c in cnt.items()), key=lambda r: r[0])
if rows:
    us, c = rows[len(rows) // 2]; cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("%-18s un-profiled $ms | profiled us=%.1f Mcyc=%.3f GHz=%.2f mfma_util=%.3f wait_any=%.3f wait_inst=%.3f (lds %.3f) active=%.3f" % ("$n", us, cyc / 1e6, cyc / us / 1e3,
          c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
else:
    print("$n: no data, $ms")
PY
  rm -rf $O/wp_$n
done
