for s in 0 20 40 80; do TVL_GEMM_STAGGER_US=$s python tools/bench_layer_gemms.py tiles=0 rounds=5 2>&1 | grep -E "us  |seven" | sed "s/^/stagger $s: /"; done
