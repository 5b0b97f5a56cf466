#!/bin/bash
# Same-box A/B of engine switches on the training-step bench.  Usage: tools/ab_env2.sh "VAR=0" ["VAR2=0" ...]  (each against the default), 2 rounds
one() { env $2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do one default ""; for v in "$@"; do one "$v" "$v"; done; done
