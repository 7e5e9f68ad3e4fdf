#!/bin/bash
# PMC passes of the Newton kernels (each its own rocprofv3 run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_newton
rm -rf $OUT && mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/exp/kt_one.py > $OUT/$name.out 2> $OUT/$name.err || { echo "pass $name failed"; exit 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/exp/kt_one.py > $OUT/trace.out 2> $OUT/trace.err || { echo "trace failed"; exit 1; }
ls $OUT/*/* | head -30
