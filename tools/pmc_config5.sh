#!/bin/bash
# SQ / TCC counters of the config-5 kernel (sls_admm_kernel), separate passes, program directly behind `--`:
#     tools/pmc_config5.sh <dim 1|3> <outdir under gpurun_out>
set -e
dim=$1; out=gpurun_out/$2
mkdir -p "$out"
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$out/$1" -o p -- python3 bench.py --config5 --config5-dim "$dim" --no-cpu-baseline > "$out/$1.log" 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run sq2 "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum"
