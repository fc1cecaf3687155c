#!/usr/bin/env python3
"""Per-launch time of the fp32 job-list kernel (csrc/mlp_f32.hip) as a function of the GEMM's shape: a graph of 40
back-to-back launches of ONE forward job out = tanh(x[rows, K] @ w[N, K].T + b), for a grid of (rows, N, K).
Separates the fixed cost of a dependent launch from what the reduction length / the weight volume add.
    python tools/probes/f32_job_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_distributed_amd.ops import mlp_f32_engine as m


def bench(rows, N, K, reps=40, dgrad=False):
    dev = torch.device("cuda:0")
    x = torch.randn(rows, (K + 3) // 4 * 4, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.zeros(N, device=dev)
    out = torch.zeros(rows, (N + 3) // 4 * 4, device=dev)
    if dgrad:
        j = m.dgrad_job(x, w.t().contiguous(), None, out, rows, K, N)
    else:
        j = m.fwd_job(x, w, b, out, rows, N, K)
    jl = m.JobList([j])
    jl.run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            jl.run()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * reps) * 1e6


def main():
    m._L()
    res = []
    for rows, N, K in [(64, 200, 3), (64, 200, 16), (64, 200, 64), (64, 200, 128), (64, 200, 300), (64, 200, 448),
                       (64, 16, 300), (64, 64, 300), (64, 800, 300), (16, 200, 300), (256, 200, 300),
                       (192, 300, 401)]:
        us = bench(rows, N, K)
        res.append({"rows": rows, "N": N, "K": K, "us": round(us, 2)})
        print(res[-1], flush=True)
    for rows, N, K in [(128, 401, 300), (64, 300, 200)]:
        us = bench(rows, N, K, dgrad=True)
        res.append({"dgrad": True, "rows": rows, "N": N, "K": K, "us": round(us, 2)})
        print(res[-1], flush=True)


if __name__ == "__main__":
    main()
