#!/bin/bash
# cycles / clock / wait breakdown of the 3x3 patch kernels for the product and probe builds on ONE shape set.  Usage: tools/patch_pmc.sh <tag> [libdirs...]
tag=$1; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for lib in product "$@"; do
  n=$(basename $lib)
  [ $lib = product ] && unset PDAE_HIP_LIB || export PDAE_HIP_LIB=$R/$lib/libpdae_hip.so
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/${tag}_${n} -- python $R/tools/patch_bench.py > $O/${tag}_${n}.log 2>&1)
  python - <<PY
import csv, glob, collections
info = {}
for f in glob.glob("$O/${tag}_${n}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"] and "wprep" not in r["Kernel_Name"]:
            info[r["Dispatch_Id"]] = (r["Kernel_Name"][5:33], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(dict)
for f in glob.glob("$O/${tag}_${n}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] in info:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
groups = collections.defaultdict(list)
for d, c in cnt.items():
    groups[(info[d][0], round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1e6))].append((info[d][1], c))
for key in sorted(groups):
    rows = sorted(groups[key], key=lambda x: x[0]); us, c = rows[len(rows) // 2]
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("%-16s" % "$n", key, "n=%d us=%.1f Mcyc=%.3f GHz=%.2f mfma_util=%.3f wait_any=%.3f wait_inst=%.3f active=%.3f" % (len(rows), us, cyc / 1e6, cyc / us / 1e3,
          c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
  rm -rf $O/${tag}_${n}
done
