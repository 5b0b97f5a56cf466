#!/bin/bash
# Same-box A/B of the round-6 changes on the FFHQ-128 training step (B = 32): every line = one `python bench.py --no-cpu-baseline --no-ddim --no-legs` run.
# r5 = the round-5 behaviour through the switches (conv3x3w everywhere, encoder on the caller's stream, GroupNorm-recomputing weight gradients,
# tile = block index in conv3x3y); then one switch at a time towards the round-6 defaults; r6 = defaults.
run() { env "$@" python bench.py --no-cpu-baseline --no-ddim --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-92s step %.3f ms  wgrad %.3f ms/step frac %.4f  conv %.4f' % ('$*', d['ms_per_step'], d['roofline_wgrad']['kernel_ms_per_step'], d['roofline_wgrad']['frac'], d['roofline']['frac']))"; }
R5="PDAE_W3V=0 PDAE_SIDE_ENC=0 PDAE_FUSE_GN_TRAIN=1 PDAE_Y_XCD=0"
for i in 1 2 3; do
run $R5
run PDAE_W3V=1 PDAE_SIDE_ENC=0 PDAE_FUSE_GN_TRAIN=1 PDAE_Y_XCD=0
run PDAE_W3V=1 PDAE_SIDE_ENC=1 PDAE_FUSE_GN_TRAIN=1 PDAE_Y_XCD=0
run PDAE_W3V=1 PDAE_SIDE_ENC=1 PDAE_FUSE_GN_TRAIN=0 PDAE_Y_XCD=0
run PDAE_W3V=1 PDAE_SIDE_ENC=1 PDAE_FUSE_GN_TRAIN=0 PDAE_Y_XCD=2
done
