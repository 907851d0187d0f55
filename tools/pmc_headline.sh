#!/bin/bash
# HBM traffic per launch of the headline workload's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (TCC slots),
# then the kernel-trace statistics of an unprofiled-counter run; the program sits directly behind `--`.
#     tools/pmc_headline.sh <outdir under gpurun_out> <tree tag>
set -e
out=gpurun_out/$1; tag=$2
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$out/write.log" 2>&1
python tools/pmc_traffic.py $(ls $out/fetch/*counter_collection.csv | head -1) $(ls $out/write/*counter_collection.csv | head -1) "$tag" > "$out/pmc_traffic.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 bench.py --no-cpu-baseline > "$out/stats_bench.json" 2> "$out/stats.log"
