#!/bin/bash
# SQ counter passes over tools/q_bench.py (both patch kernels on the large shapes).  Usage: tools/q_pmc.sh <tag>; results in gpurun_out/<tag>_pmc{A,B}.json
tag=$1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/${tag}_pmcA -- python $R/tools/q_bench.py > $O/${tag}_pmcA.log 2>&1); echo "pmcA rc=$?"
(cd /tmp && timeout 300 rocprofv3 --output-format csv --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/${tag}_pmcB -- python $R/tools/q_bench.py > $O/${tag}_pmcB.log 2>&1); echo "pmcB rc=$?"
python $R/tools/summarize_pmc.py $O $tag
python - <<PY
import json
for p in "AB":
    try: d = json.load(open("$O/${tag}_pmc%s.json" % p))
    except Exception as e: print(e); continue
    for k, v in d.items():
        if "conv3x3" in k and "wprep" not in k:
            print(p, k[:60], {c: round(x, 1) for c, x in v.items() if c not in ("resources",)})
PY
find $O -name "*.db" -delete 2>/dev/null
