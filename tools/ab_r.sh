one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['final_loss'], d['roofline']['avg_launch_ms'])"; }
for i in 1 2; do one r512; PDAE_P3R_MIN=256 one r256; PDAE_P3R_MIN=224 one r224; done
