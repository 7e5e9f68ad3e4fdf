#!/bin/bash
# kernel trace + stats of a short bench run (summary copied to profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_trace
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || echo "trace run failed"
ls $OUT/*/* | head
