#!/bin/bash
# SQ counter passes over the conv3x3v probe (128^2 128->128, B = 32): tools/v_pmc.sh <tag> [variant ...]   ("" = product build)
tag=$1; shift; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
vars=${@:-product}
for v in $vars; do
  lib=""; [ "$v" != product ] && lib=$R/pdae_amd/lib/probe_$v/libpdae_hip.so
  run() { name=$1; shift; (cd /tmp && PDAE_HIP_LIB=$lib W3_ONLY=1 timeout 300 rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d $O/${tag}_${v}_pmc$name -- python $R/tools/w3_probe.py 1 > $O/${tag}_${v}_pmc$name.log 2>&1); echo "$v pass $name rc=$?"; }
  run A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  run B SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
  run C SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  python $R/tools/summarize_pmc.py $O ${tag}_${v} || true
done
find $O -name "*.db" -size +20M -delete 2>/dev/null
