"""Time the native DDPG learner step (ops/mlp_engine.py) on HBM-resident replay: fused three-launch step
(csrc/mlp_fused.hip, --precision bf16) or the fp32 job-list engine (csrc/mlp_f32.hip, --precision fp32).

    python tools/bench_mlp_step.py [--batch 64] [--steps 4000] [--graph-steps 10] [--obs 3] [--act 1]
Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--graph-steps", type=int, default=None,
                    help="updates per hipGraph replay (default: Options.graph_steps = what the launcher-run learner uses, "
                         "10; fp32 engine: 5 / 10 / 20 -> 36.2 / 35.7 / 35.4 us per update)")
    ap.add_argument("--obs", type=int, default=3)
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--replay", type=int, default=100000)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--phases", action="store_true", help="print the fused kernel's phase timeline (wall_clock64 stamps)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="bf16: fused bf16 kernel (csrc/mlp_fused.hip); fp32: fp32-input MFMA job lists (csrc/mlp_f32.hip)")
    a = ap.parse_args()
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.models.ddpg_mlp_model import DdpgMlpModel
    from pytorch_distributed_amd.ops import mlp_engine as eng
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(0)
    opt = Options(config=7)
    D, A, n, B = a.obs, a.act, a.replay, a.batch
    model = DdpgMlpModel(opt.model_params, 1.0, (1, D), A, A).cuda()
    target = DdpgMlpModel(opt.model_params, 1.0, (1, D), A, A).cuda()
    if a.precision == "fp32":
        from pytorch_distributed_amd.ops.mlp_f32_engine import DdpgMlpEngineF32 as engine_cls
    else:
        engine_cls = eng.DdpgMlpEngine
    e = engine_cls(model, target, B, lr=1e-3, tau=1e-3, leak=False, tail_offsets=(0, A, A + 1, A + 2), seed=1)
    s0, s1 = torch.randn(n, D, device="cuda"), torch.randn(n, D, device="cuda")
    tails = torch.zeros(n, A + 4, device="cuda")
    tails[:, :A] = torch.rand(n, A, device="cuda") * 2 - 1
    tails[:, A] = torch.randn(n, device="cuda")
    tails[:, A + 1] = 0.99
    if a.phases:
        e.prof = torch.zeros(32, dtype=torch.int64, device="cuda")
    e.bind_replay(s0, s1, tails, None)
    e.set_size(n)
    graph_steps = int(a.graph_steps if a.graph_steps is not None else opt.graph_steps)
    prog = eng.MlpStepProgram(e, None, use_graph=not a.no_graph, staged=False, graph_steps=graph_steps)
    for _ in range(a.warmup):
        prog.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        prog.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if a.phases and e.fused and a.precision == "bf16":
        st = e.prof.cpu().tolist()
        t0_ = st[31]
        names = ["zero-lds", "stage", "actor.0", "actor.2", "actor.4+critic.0", "critic.1", "head", "dh2", "dcat",
                 "dha2", "dha1"]
        marks = [s_ for s_ in st[:30] if s_ > 0]
        prev = t0_
        for i, m in enumerate(marks):
            print("phase %-18s %6.2f us" % (names[i] if i < len(names) else "?", (m - prev) / 100.0))
            prev = m
        print("phase %-18s %6.2f us   total %6.2f us (100 MHz wall clock)" % ("spill", (st[30] - prev) / 100.0,
                                                                            (st[30] - t0_) / 100.0))
    print(json.dumps({"bench": "ddpg_mlp_step", "precision": a.precision, "fused": bool(e.fused), "batch": B, "obs": D, "act": A,
                      "graph_steps": prog.graph_steps, "hip_graph": not a.no_graph,
                      "us_per_step": round(1e6 * dt / a.steps, 2), "updates_per_sec": round(a.steps / dt, 1),
                      "final_critic_loss": e.mean_losses()[1]}), flush=True)


if __name__ == "__main__":
    main()
