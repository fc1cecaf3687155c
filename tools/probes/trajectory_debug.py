"""Per-update drift of the native learner against the fp32 torch learner on the same replay rows (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_trajectory as T
from pytorch_distributed_amd.ops import dqn_engine as eng

precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, n, steps, seed, lr, tu = 128, 2048, 12, 77, 1e-4, 10
s0, s1, tails = T.make_replay(n)
model, target = T.make_models(seed=3)
ref = T.TorchLearner(model, lr, tu)
ref64 = T.TorchLearner(model, lr, tu)     # same learner, parameters perturbed by 1e-6 relative: how fast do two fp32 runs part?
with torch.no_grad():
    for p in ref64.model.parameters():
        p.mul_(1.0 + 1e-6 * torch.randn_like(p))
    ref64.target.load_state_dict(ref64.model.state_dict())
init = ref.flat().clone()
e = eng.DqnCnnEngine(model, target, B, lr=lr, target_update=tu, seed=seed, precision=precision)
e.bind_replay(s0, s1, tails, None); e.set_size(n)
prog = eng.DqnStepProgram(e, None, use_graph=False, staged=False, graph_steps=1)
names = [(nm, p.numel()) for nm, p in ref.model.named_parameters()]
for t in range(steps):
    prog.step(); torch.cuda.synchronize()
    rows = T.uniform_rows(e.seed, t, B, n)
    td = ref.step(s0, s1, tails, rows)
    ref64.step(s0, s1, tails, rows)
    if e.x3:
        worst = max(((w.value() - w.hi.float()).abs().max() / w.hi.float().abs().max()).item() for w in e.online.w[:4])
        m = model.critic[1][0].weight.detach().reshape(512, 64, 49).permute(0, 2, 1).reshape(512, -1)
        sh = ((e.online.w[3].value() - m).abs().max() / m.abs().max()).item()
    else:
        worst = sh = 0.0
    gref = torch.cat([p.grad.reshape(-1) for p in ref.model.parameters()])
    gnat = e.grads_torch_layout()
    off, per = 0, []
    for nm, k in names:
        a, b = gnat[off:off + k], gref[off:off + k]
        per.append("%s %.1e" % (nm.replace("critic.", ""), ((a - b).norm() / b.norm().clamp_min(1e-30)).item()))
        off += k
    print("t=%d drift %.4f (perturbed torch twin: %.4f) shadow err %.1e  td err %.2e | grad rel err: %s" % (
          t, T.drift(e.online.flat, ref.flat(), init), T.drift(ref64.flat(), ref.flat(), init), sh,
          (e.td - td).abs().max().item(), ", ".join(per[:1] + per[6:7])), flush=True)
