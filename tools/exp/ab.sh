#!/bin/bash
# A/B of the fused quad kernel on the strip (steady state: 60 untimed launches first): tools/exp/ab.sh [variant.so ...]  (the in-tree library runs last)
for v in "$@"; do
  echo "== $v"; FEMSHELL_HIP_LIB=$PWD/$v FUSED_BENCH_PREWARM=60 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only" || exit 1
done
echo "== in-tree"; FUSED_BENCH_PREWARM=60 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only"
