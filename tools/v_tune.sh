#!/bin/bash
# conv3x3v tuning A/B on the probe shapes (same box): product vs a variant library
for i in 1 2; do
echo "product : $(python tools/w3_probe.py 1 2>/dev/null)"
for v in "$@"; do echo "$v : $(PDAE_HIP_LIB=$PWD/pdae_amd/lib/probe_$v/libpdae_hip.so python tools/w3_probe.py 1 2>/dev/null)"; done
done
