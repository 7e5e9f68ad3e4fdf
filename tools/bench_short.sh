#!/bin/bash
# usage: tools/bench_short.sh [lib.so]  -> prints value, ms/step, K-kernel ms, assembly ms
[ -n "$1" ] && export FEMSHELL_HIP_LIB=$1
timeout -k 10 200 python -u bench.py --no-pmc --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=[d['roofline'],d['roofline_other_kernel']]; print('el/s %.4g  ms/step %.3f  ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.3f ms frac %.3f' % (x['kernel'][:12], x['kernel_ms'], x['frac']) for x in r))"
