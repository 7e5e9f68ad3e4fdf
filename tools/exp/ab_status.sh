#!/bin/bash
# A/B of the fused quad kernel on the strip in the steady state, without and with the status words: tools/exp/ab_status.sh [variant.so ...]
for v in "$@"; do
  echo "== $v"; FEMSHELL_HIP_LIB=$PWD/$v FUSED_BENCH_PREWARM=60 FUSED_BENCH_STATUS=1 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only\|fused with" || exit 1
done
echo "== in-tree"; FUSED_BENCH_PREWARM=60 FUSED_BENCH_STATUS=1 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only\|fused with"
