#!/bin/bash
# PMC passes (each its own rocprofv3 run, <= slot limits): the fused kernel on the 1.25 M strip alone, then the legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FEMSHELL_BENCH_NO_PREWARM=1     # counters, not times: no need for steady clocks
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT/strip $OUT/legs
run() { sub=$1; name=$2; shift 2; extra=""; [ $sub = strip ] && extra="--no-legs"; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$sub/$name -- python3 bench.py --no-pmc --strip-only $extra --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$sub/$name.json 2> $OUT/$sub/$name.err || { echo "pass $sub/$name failed"; exit 1; }; }
for sub in strip legs; do
run $sub sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run $sub sq2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE
run $sub fetch FETCH_SIZE
run $sub write WRITE_SIZE
done
ls $OUT/*/* | head
