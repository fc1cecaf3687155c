"""fc1's forward and data-gradient GEMMs at batch 4096 (BASELINE config 5) per tile variant of the GEMM entry points:
64 x 64 tiles (mode 1) against the 128-row tiles (modes 2 / 3, csrc/gemm.hip pda_gemm_tile_mode), bf16 and
split-bf16 (fp32-class) operands; every result is checked against an fp32 / fp64 PyTorch product first.

    python tools/probes/gemm_tiles_bench.py [--batch 4096] [--reps 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops import nn
    torch.manual_seed(0)
    B, N, K = a.batch, 512, 3136
    dev = "cuda"
    x = torch.relu(torch.randn(B, K, device=dev))
    ws = [torch.randn(N, K, device=dev) / K ** 0.5 for _ in range(2)]
    dy = torch.randn(B, N, device=dev)
    out = {"batch": B, "rows": []}
    # ---- bf16 ----
    xb, wb, dyb = x.bfloat16(), [w.bfloat16() for w in ws], dy.bfloat16()
    ref_f = [xb.float() @ w.float().t() for w in wb]
    ref_d = (dyb.float() @ wb[0].float()) * (xb.float() > 0)
    for mode in (1, 2, 3):
        nn.gemm_tile_mode(mode)
        slabs = [torch.full((1, B, N), 7.0, device=dev) for _ in range(2)]
        fwd = lambda: nn.linear_fwd_slab([xb, xb], wb, slabs, B, N, K, K, K, N, B * N, K)     # noqa: E731
        fwd()
        err_f = max(rel(s[0], r) for s, r in zip(slabs, ref_f))
        dx = torch.zeros(B, K, dtype=torch.bfloat16, device=dev)
        dg = lambda: nn.linear_dgrad(dyb, wb[0], xb, dx, B, N, K, N, K, K, K, act=nn.ACT_RELU)   # noqa: E731
        dg()
        err_d = rel(dx, ref_d)
        t_f, t_d = timed(fwd, a.reps), timed(dg, a.reps)
        out["rows"].append({"operands": "bf16", "tile_mode": mode, "fc1_fwd_us": round(t_f, 1), "fc1_dgrad_us": round(t_d, 1),
                            "fwd_tflops": round(2 * 2 * B * N * K / t_f * 1e-6, 1),
                            "dgrad_tflops": round(2 * B * N * K / t_d * 1e-6, 1),
                            "fwd_rel_err": err_f, "dgrad_rel_err": err_d})
        print(out["rows"][-1], flush=True)
    # ---- split-bf16 (three products per pair) ----
    xp, wp, dyp = nn.Planes.from_float(x), [nn.Planes.from_float(w) for w in ws], nn.Planes.from_float(dy)
    ref_f = [x.double() @ w.double().t() for w in ws]
    ref_d = (dy.double() @ ws[0].double()) * (x > 0)
    for mode in (1, 2, 3):
        nn.gemm_tile_mode(mode)
        slabs = [torch.full((1, B, N), 7.0, device=dev) for _ in range(2)]
        fwd = lambda: nn.x3_linear_fwd([xp, xp], wp, slabs, B, N, K, K, K, N, ksplit=K, slice_stride=B * N)   # noqa: E731
        fwd()
        err_f = max(rel(s[0], r) for s, r in zip(slabs, ref_f))
        dx = nn.Planes(B, K, device=dev)
        dg = lambda: nn.x3_linear_dgrad(dyp, wp[0], xp.hi, dx, B, N, K, N, K, K, K, act=nn.ACT_RELU)  # noqa: E731
        dg()
        err_d = rel(dx.value(), ref_d)
        t_f, t_d = timed(fwd, a.reps), timed(dg, a.reps)
        out["rows"].append({"operands": "split-bf16 x3", "tile_mode": mode, "fc1_fwd_us": round(t_f, 1),
                            "fc1_dgrad_us": round(t_d, 1),
                            "fwd_tflops_incl_x3": round(3 * 2 * 2 * B * N * K / t_f * 1e-6, 1),
                            "dgrad_tflops_incl_x3": round(3 * 2 * B * N * K / t_d * 1e-6, 1),
                            "fwd_rel_err": err_f, "dgrad_rel_err": err_d})
        print(out["rows"][-1], flush=True)
    # ---- the large tiles with the reduction split in slices (more workgroups = more loads in flight per CU) ----
    for slices in (1, 2, 4):
        ks = nn.round_ksplit((K + slices - 1) // slices)
        ns = nn.num_slices(K, ks)
        nn.gemm_tile_mode(3)
        slabs = [torch.full((ns, B, N), 7.0, device=dev) for _ in range(2)]
        fwd = lambda: nn.linear_fwd_slab([xb, xb], wb, slabs, B, N, K, K, K, N, B * N, ks)     # noqa: E731
        fwd()
        e_b = rel(slabs[0].sum(0), xb.float() @ wb[0].float().t())
        t_b = timed(fwd, a.reps)
        nn.gemm_tile_mode(2)
        fwd3 = lambda: nn.x3_linear_fwd([xp, xp], wp, slabs, B, N, K, K, K, N, ksplit=ks, slice_stride=B * N)   # noqa: E731
        fwd3()
        e_x = rel(slabs[0].sum(0), ref_f[0])
        t_x = timed(fwd3, a.reps)
        out["rows"].append({"large_tiles_k_slices": ns, "ksplit": ks, "bf16_fc1_fwd_us": round(t_b, 1),
                            "x3_fc1_fwd_us": round(t_x, 1), "bf16_rel_err": e_b, "x3_rel_err": e_x})
        print(out["rows"][-1], flush=True)
    nn.gemm_tile_mode(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
