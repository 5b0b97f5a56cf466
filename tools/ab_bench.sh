#!/bin/bash
# Same-box A/B of two library builds on the training-step bench.  Usage: tools/ab_bench.sh <baseline lib dir under pdae_amd/lib> [rounds]
base=$1; rounds=${2:-2}
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in $(seq $rounds); do PDAE_HIP_LIB=pdae_amd/lib/$base/libpdae_hip.so one $base; one product; done
