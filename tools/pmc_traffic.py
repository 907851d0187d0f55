#!/usr/bin/env python3
"""Condense rocprofv3 PMC passes into per-kernel HBM traffic (bytes per launch).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \
        "$(git rev-parse --short HEAD)" > profiles/r01_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so read bytes = 2 * FETCH_SIZE KiB;
WRITE_SIZE is exact.  The two counters need separate passes (TCC slots)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("isls::"):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out[k] = {"launches_sampled": max(nf, nw), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                  "read_bytes": 2.0 * f * 1024.0, "write_bytes": w * 1024.0, "hbm_bytes": 2.0 * f * 1024.0 + w * 1024.0}
    tree = sys.argv[3] if len(sys.argv) > 3 else None           # commit (or description) of the source tree that was measured
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py default workload", "tree": tree,
               "correction": "read = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B), write = WRITE_SIZE",
               "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
