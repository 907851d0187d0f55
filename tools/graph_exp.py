#!/usr/bin/env python3
"""Experiment: one outer iteration of the headline workload (linearize, expand, outer driver, accept, reduce) replayed from a
HIP graph against the eager launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ilqr-admm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

from two_stream_exp import make, step


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = make(4096, 100, 5, 20, 0, dev)
    for _ in range(3):
        step(eng)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        step(eng)
    torch.cuda.synchronize()
    print(f"eager: {1e3 * (time.perf_counter() - t0) / K:.4f} ms per outer iteration", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(eng)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph: {1e3 * (time.perf_counter() - t0) / K:.4f} ms per outer iteration", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
