# strip: stiffness and mass operators of the fused quad kernel under different patch-id tiles (FEMSHELL_PATCH_TILE)
for t in ${TILES:-1x1 4x8 2x8 3x8 1x16 2x4 4x4 1x8}; do echo "== tile $t"; FEMSHELL_PATCH_TILE=$t FUSED_BENCH_PREWARM=60 FUSED_BENCH_MASS=1 FUSED_BENCH_ONLY=40 timeout -k 10 120 python -u tools/fused_bench.py 2>&1 | grep "fused only\|fused mass" || exit 1; done
