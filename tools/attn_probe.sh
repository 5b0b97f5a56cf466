timeout 100 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -2
timeout 60 python tools/attn_bench.py
for v in ${@:-at_nonn at_nont at_nosched}; do PDAE_HIP_LIB=$PWD/pdae_amd/lib/probe_$v/libpdae_hip.so timeout 60 python tools/attn_bench.py; done
