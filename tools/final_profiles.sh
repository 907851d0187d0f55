#!/bin/bash
# Everything profiles/rNN_* is taken from, in one gpurun call:  tools/final_profiles.sh <outdir under gpurun_out> <tree tag>
set -e
out=gpurun_out/$1; tag=$2
mkdir -p "$out"
tools/pmc_headline.sh "$1/pmc" "$tag"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
for c in 3 4; do
  python bench.py --config $c > "$out/config${c}_bench.json" 2> "$out/config${c}.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/config${c}_stats" -o s -- python3 bench.py --config $c --no-cpu-baseline > /dev/null 2> "$out/config${c}_stats.log"
done
python bench.py --config5 --config5-dim 1 > "$out/config5_bench.jsonl" 2> "$out/config5.err"
python bench.py --config5 --config5-dim 3 >> "$out/config5_bench.jsonl" 2>> "$out/config5.err"
python bench.py --isls-admm > "$out/isls_admm_bench.json" 2> "$out/isls_admm.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/isls_admm_stats" -o s -- python3 bench.py --isls-admm --no-cpu-baseline > /dev/null 2> "$out/isls_admm_stats.log"
echo done
