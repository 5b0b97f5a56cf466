#!/bin/bash
# HBM fetch of the conv3x3y launches with the XCD-contiguous tile order off / on (PDAE_Y_XCD = 0 / 2): one FETCH_SIZE pass each over one training step
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in 0 2; do
  (cd /tmp && PDAE_Y_XCD=$v timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/xcd${v}_pmc_FETCH_SIZE -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim --no-legs > $O/xcd${v}.log 2>&1); echo "xcd $v rc=$?"
  python $R/tools/summarize_pmc.py $O xcd${v} || true
done
