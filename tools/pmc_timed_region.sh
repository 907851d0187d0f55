#!/bin/bash
# HBM traffic per launch of the kernels of bench.py's TIMED REGION alone (no general-layout / shared-LTI leg behind it):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes; the program sits directly behind `--`.
#     tools/pmc_timed_region.sh <outdir under gpurun_out> <tree tag>
set -e
out=gpurun_out/$1; tag=$2
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o f -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --only-timed-region > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o w -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --only-timed-region > "$out/write.log" 2>&1
python tools/pmc_traffic.py $(ls $out/fetch/*counter_collection.csv | head -1) $(ls $out/write/*counter_collection.csv | head -1) "$tag" > "$out/pmc_traffic.json"
