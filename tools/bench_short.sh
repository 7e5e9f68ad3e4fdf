#!/bin/bash
# usage: tools/bench_short.sh [lib.so]  -> prints value, ms/step, K-kernel ms, assembly ms
[ -n "$1" ] && export FEMSHELL_HIP_LIB=$1
timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('el/s %.4g  ms/step %.3f  K %.3f ms  asm %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('k_kernel_ms', d['roofline']['kernel_ms']), d['assembly']['kernel_ms']))"
