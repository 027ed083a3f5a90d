#!/bin/bash
# time every tile of the full-line GEMM family (and the round-1 dispatch, --k64 1) on the hot shapes
SH=${SH:-'packed,probe,square,qkv fwd,fc1 fwd,fc2 fwd,lm_head,vitL,prefill,vit '}
for k in ${KS:-1 0 2 3 4 5 6 7 8 9 10 11 12 13 14 15}; do
  echo "=== --k64 $k"
  python tools/gemm_bench.py --k64 $k --only "$SH" 2>/dev/null
done
