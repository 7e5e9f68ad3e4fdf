#!/bin/bash
# kernel trace + stats of (a) the default bench command and (b) the 1.25 M strip alone (summaries are copied to
# profiles/ afterwards by tools/trace_summary.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_trace
rm -rf $OUT && mkdir -p $OUT/full $OUT/strip
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -- python3 bench.py --no-pmc > $OUT/full/bench.json 2> $OUT/full/bench.err || { echo "trace run (full) failed"; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strip -- python3 bench.py --no-pmc --strip-only --no-legs --no-cpu-baseline > $OUT/strip/bench.json 2> $OUT/strip/bench.err || { echo "trace run (strip) failed"; exit 1; }
ls $OUT/*/*/* | head -20
