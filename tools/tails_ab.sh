for e in "" "TVL_GEMM_M16=0" "TVL_ATTN_H2=0" "TVL_DQKV_H2=0" "TVL_GEMM_H2=0"; do
  echo "== env: $e"
  env $e python -m pytest tests/test_net_parity.py -q -s -k "full_size_tp3 and (maple_n4_d9_newlast_tails or vpt_n10_d1_tails or maple_n4_d9_newlast-)" 2>&1 | grep -E "PARITY|AssertionError: |passed|failed" | cut -c1-260
done
