#!/bin/bash
# SQ instruction / wait counters of a secondary workload's kernels (two rocprofv3 passes, program directly behind `--`):
#     tools/pmc_sq_config.sh <outdir under gpurun_out> <bench.py arguments, e.g. --config 3>
set -e
out=gpurun_out/$1
shift
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$out/sq1" -o p -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > "$out/sq1.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$out/sq2" -o p -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > "$out/sq2.log" 2>&1
for p in sq1 sq2; do python tools/pmc_insts.py $(ls $out/$p/*counter_collection.csv | head -1); done > "$out/sq.txt"
