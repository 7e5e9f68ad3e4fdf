#!/bin/bash
# builds tools/exp/variants/lib_timing.so: the library with -DFS_FUSED_TIMING (tools/exp/fused_timing.py)
set -e
cd "$(dirname "$0")/../fem_shell_amd/csrc"
mkdir -p ../../tools/exp/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include -DFS_FUSED_TIMING "$@" -c femshell_hip.hip -o /tmp/femshell_hip_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/femshell_hip_timing.o femshell_fused_hex.o femshell_fused_tri.o femshell_fused_nl.o femshell_symbolic.o femshell_sparse.o femshell_stress.o femshell_fused_plan.o femshell_fused_plan_dev.o -o ../../tools/exp/variants/lib_timing.so -lgomp
