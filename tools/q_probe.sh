#!/bin/bash
# conv3x3q timing probes on one box: product library, then every pdae_amd/lib/probe_q_*/ build (q column only is meaningful for the probes)
python tools/q_bench.py ${1:-32} 2>&1 | grep -v amdgpu.ids | sed 's/^/product     /'
for d in pdae_amd/lib/probe_q_*; do
  n=$(basename $d)
  PDAE_HIP_LIB=$d/libpdae_hip.so python tools/q_bench.py ${1:-32} 2>&1 | grep -v amdgpu.ids | awk -v n=$n '{printf "%-12s", n; print}' | cut -c1-400
done
