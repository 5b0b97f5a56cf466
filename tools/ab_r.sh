one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['final_loss'], d['roofline']['avg_launch_ms'])"; }
for i in 1 2; do PDAE_P3R=0 one r_off; one r_on; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ddim', d['ddim100'])"
PDAE_P3R=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ddim r_off', d['ddim100'])"
