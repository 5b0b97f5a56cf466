run() { env "$@" python bench.py --no-cpu-baseline --no-ddim --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['roofline_wgrad']['kernel_ms_per_step'])"; }
for i in 1 2; do
run PDAE_FUSE_GN_TRAIN=1
run PDAE_FUSE_GN_TRAIN=0
run PDAE_FUSE_GN_TRAIN=1 PDAE_FUSE_GN_TRAIN_MAXCOUT=256
done
