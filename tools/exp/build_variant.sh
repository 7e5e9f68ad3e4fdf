#!/bin/bash
# tools/exp/build_variant.sh NAME [-DFLAG ...]: tools/exp/variants/NAME.so = the library with femshell_hip.hip rebuilt under the flags
set -e
name=$1; shift
cd "$(dirname "$0")/../../fem_shell_amd/csrc"
mkdir -p ../../tools/exp/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include "$@" -c femshell_hip.hip -o /tmp/femshell_hip_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/femshell_hip_$name.o femshell_fused_hex.o femshell_fused_tri.o femshell_fused_nl.o femshell_symbolic.o femshell_sparse.o femshell_stress.o femshell_fused_plan.o femshell_fused_plan_dev.o -o ../../tools/exp/variants/$name.so -lgomp
