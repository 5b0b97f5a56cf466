#!/bin/bash
# Same-box A/B of an engine switch on the training-step bench.  Usage: tools/ab_env.sh VAR [rounds]   (VAR=0 against the default)
var=$1; rounds=${2:-2}
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['final_loss'])"; }
for i in $(seq $rounds); do env $var=0 bash -c "$(declare -f one); one ${var}_off"; one ${var}_on; done
