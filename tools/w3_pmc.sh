#!/bin/bash
# SQ counter passes over the weight-gradient probe (tools/w3_probe.py): what the conv3x3w waves wait for.  Output: gpurun_out/<tag>_w3_pmc{A,B,C}.json
tag=$1; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d $O/${tag}_w3_pmc$name -- python $R/tools/w3_probe.py > $O/${tag}_w3_pmc$name.log 2>&1); echo "pass $name rc=$?"; }
run A SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
run B SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run C SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
python $R/tools/summarize_pmc.py $O ${tag}_w3 || true
