#!/bin/bash
# usage: tools/bench_short.sh [lib.so]  -> prints value, ms/step and the fused kernel's time and roofline fraction
[ -n "$1" ] && export FEMSHELL_HIP_LIB=$1
timeout -k 10 200 python -u bench.py --no-pmc --no-legs --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('el/s %.4g  ms/step %.3f  %s %.3f ms frac %.3f' % (d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac']))"
