#!/bin/bash
# conv3x3y (Winograd along x, one wave per SIMD) timing probes with counters on one box: product, then every pdae_amd/lib/probe_y_*/ build.  128x128 128->128, B=32, plain forward.
export TMPDIR=/tmp PDAE_W1=2
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/y_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import torch
from pdae_amd import hip as H
N, S, Cin, Cout = 32, 128, 128, 128
x = torch.randn(N, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5; b = torch.randn(Cout, device="cuda")
y = torch.empty(N, S, S, Cout, device="cuda")
c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
op = H.op_conv_fwd(c, x, None, w, b, y, wp=wp)
for _ in range(40): H.run(op)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): H.run(op)
e1.record(); torch.cuda.synchronize()
print("ms", e0.elapsed_time(e1) / 10)
PY
export R
for lib in product $(ls -d pdae_amd/lib/probe_y_* 2>/dev/null); do
  n=$(basename $lib)
  [ $lib = product ] && unset PDAE_HIP_LIB || export PDAE_HIP_LIB=$R/$lib/libpdae_hip.so
  ms=$(timeout 60 python $R/tools/y_one.py 2>/dev/null | grep ms)
  (cd /tmp && timeout 90 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/yp_$n -- python $R/tools/y_one.py > /dev/null 2>&1)
  python - <<PY
import csv, glob
info = {}
for f in glob.glob("$O/yp_$n/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3y" in r["Kernel_Name"]:
            info[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
cnt = {}
for f in glob.glob("$O/yp_$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] in info:
            cnt.setdefault(r["Dispatch_Id"], {}); cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = sorted(((info[d], c) for d, c in cnt.items()), key=lambda r: r[0])
if rows:
    us, c = rows[len(rows) // 2]; cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("%-18s un-profiled $ms | profiled us=%.1f GHz=%.2f mfma_util=%.3f wait_any=%.3f wait_inst=%.3f (lds %.3f) active=%.3f" % ("$n", us, cyc / us / 1e3,
          c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
else:
    print("$n: no data, $ms")
PY
  rm -rf $O/yp_$n
done
