#!/usr/bin/env python3
"""A bare learner-step runner for hardware-counter passes.

    rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES ... -- python3 tools/pmc_step.py --batch 128 --steps 20

Builds the native DQN engine on raw tensors (no replay object, no multiprocessing
primitives, so this process never starts a child: a counter-collecting rocprofv3 has
initialised the GPU before Python runs, and nothing may be exec'd after that) and runs
`--steps` eager learner steps on synthetic Atari-shaped data.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--replay", type=int, default=8192)
    ap.add_argument("--per", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true",
                    help="graph-replayed steps (five updates per hipGraph, as the learner runs them) instead of eager ones; "
                         "--steps is then rounded up to whole replays after one eager step")
    a = ap.parse_args()
    import torch
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
    from pytorch_distributed_amd.ops.dqn_engine import DqnCnnEngine, DqnStepProgram
    from pytorch_distributed_amd.ops.segtree import DeviceSumMinTree
    from pytorch_distributed_amd.utils.options import ModelParams
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    mp_ = ModelParams(config=2)
    model = DqnCnnModel(mp_, 255.0, (4, 84, 84), 6, 1).to(dev)
    target = DqnCnnModel(mp_, 255.0, (4, 84, 84), 6, 1).to(dev)
    m = a.replay
    s0 = torch.randint(0, 256, (m, 4, 84, 84), dtype=torch.uint8, device=dev)
    s1 = torch.randint(0, 256, (m, 4, 84, 84), dtype=torch.uint8, device=dev)
    tails = torch.zeros(m, 5, device=dev)
    tails[:, 0] = torch.randint(0, 6, (m,), device=dev).float()
    tails[:, 1] = torch.randn(m, device=dev).clamp_(-1, 1)
    tails[:, 2] = 0.99 ** 5
    tree, per = None, None
    if a.per:
        tree = DeviceSumMinTree(m, dev)
        tree.update(torch.arange(m, device=dev, dtype=torch.int32), torch.rand(m, device=dev) + 0.05)
        per = {"eps": 1e-6, "alpha": 0.5, "beta": 0.4}
    eng = DqnCnnEngine(model, target, a.batch, per=per, precision=a.precision).bind_replay(s0, s1, tails, tree)
    eng.set_size(m)
    prog = DqnStepProgram(eng, None, use_graph=a.graph, graph_steps=5 if a.graph else 1)
    steps = a.steps if not a.graph else 1 + 5 * ((a.steps + 4) // 5)
    for _ in range(steps):
        prog.step()
    torch.cuda.synchronize()
    print("pmc_step: %d %s steps of batch %d, loss %.5f" % (steps, "graph-replayed" if a.graph else "eager", a.batch,
                                                             eng.mean_loss()), flush=True)


if __name__ == "__main__":
    main()
