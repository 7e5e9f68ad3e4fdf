#!/bin/bash
# A/B of the fused quad kernel's stiffness and mass operators on the strip (steady state): tools/exp/ab_mass.sh [variant.so ...]
for v in "$@"; do
  echo "== $v"; FEMSHELL_HIP_LIB=$PWD/$v FUSED_BENCH_PREWARM=60 FUSED_BENCH_MASS=1 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only\|fused mass" || exit 1
done
echo "== in-tree"; FUSED_BENCH_PREWARM=60 FUSED_BENCH_MASS=1 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only\|fused mass"
