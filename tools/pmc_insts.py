#!/usr/bin/env python3
"""Per-kernel averages of SQ instruction counters from a rocprofv3 --pmc pass (counter_collection.csv):
    python tools/pmc_insts.py gpurun_out/pmc_sq/s_counter_collection.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("isls::"):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k[:100])
    print("   ", {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", len(next(iter(d.values()))))
