#!/bin/bash
# conv3x3v probe ladder: product, no matrix work, no staging, no loads, matrix waves alone.  tools/v_probe.sh [1|gn]  (1: 128^2 128->128; gn: 128^2 256->128 plain + GroupNorm-recomputing)
sel=${1:-1}
for v in product v_nomma v_nostage v_noload v_mmaonly; do
  lib=""; [ "$v" != product ] && lib=$PWD/pdae_amd/lib/probe_$v/libpdae_hip.so
  echo "$v: $(PDAE_HIP_LIB=$lib W3_ONLY=$sel python tools/w3_probe.py 1 2>/dev/null)"
done
echo "conv3x3w: $(W3_ONLY=$sel python tools/w3_probe.py 0 2>/dev/null)"
