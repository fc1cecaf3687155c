"""Diagnostic: parameters after 5 graph steps with the fc-tail Adam inside the wgrad-pair launch (PDA_PAIR_ADAM=1)
vs the one-launch optimizer, plus a repeat of the latter (run-to-run determinism of the step itself)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_dqn_engine as T     # noqa: E402
from pytorch_distributed_amd.ops import dqn_engine as eng   # noqa: E402

B, mem_n = 128, 512
s0, s1, tails = T.make_batch(B, 6)
g = torch.Generator(device="cuda").manual_seed(2)
frames = torch.randint(0, 256, (mem_n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
rt = tails[torch.arange(mem_n, device="cuda") % B].contiguous()
outs = []
for flag in ("0", "0", "1", "1"):
    os.environ["PDA_PAIR_ADAM"] = flag
    model, target = T.make_models(seed=7)
    e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=1000, seed=11)
    e.bind_replay(frames, frames.flip(0).contiguous(), rt, None)
    e.set_size(mem_n)
    prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=int(os.environ.get("GS", "2")))
    for _ in range(int(os.environ.get("NS", "5"))):
        prog.step()
    torch.cuda.synchronize()
    outs.append((flag, e.online.flat.clone(), e.grads.clone(), e.tail_start, e.idx.clone()))
ref = outs[0]
for flag, flat, grads, ts, idx in outs[1:]:
    print("flag", flag, "idx equal", torch.equal(idx, ref[4]),
          "param head maxdiff %.3e tail %.3e" % ((flat[:ts] - ref[1][:ts]).abs().max().item(),
                                                (flat[ts:] - ref[1][ts:]).abs().max().item()),
          "grad head %.3e tail %.3e" % ((grads[:ts] - ref[2][:ts]).abs().max().item(),
                                        (grads[ts:] - ref[2][ts:]).abs().max().item()), flush=True)
