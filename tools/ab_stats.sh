one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do PDAE_FUSE_GN_STATS=0 one stats_off; one stats_on; done
PDAE_FUSE_GN_STATS=0 python tools/ddim_ops.py 2>/dev/null | head -1
python tools/ddim_ops.py 2>/dev/null | head -1
