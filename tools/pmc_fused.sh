#!/bin/bash
# PMC passes over the fused kernel alone (tools/fused_bench.py, FUSED_BENCH_ONLY): each pass its own rocprofv3 run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_fused
rm -rf $OUT && mkdir -p $OUT
export FUSED_BENCH_ONLY=3
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/fused_bench.py > $OUT/$name.log 2> $OUT/$name.err || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE
run sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_fused/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fused_quad_k_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
