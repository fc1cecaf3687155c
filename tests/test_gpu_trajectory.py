"""End-to-end checks of the BENCHED learner path (HBM replay read through the index vector, index draw folded
into the optimizer launch, split-K slabs summed by Adam / the head kernel, five updates per hipGraph replay):

* trajectory: N native updates == N updates of an fp32 PyTorch learner (the reference's arithmetic,
  /root/reference/core/single_processes/dqn_learner.py:50-101) fed the SAME replay rows -- the uniform sampler is a
  counter hash that the test re-computes on the host; the prioritized sampler's rows and importance weights are read
  back from the engine after every one-step replay;
* determinism: two runs of the same program end with bit-identical parameters, Adam moments and priorities
  (SURVEY.md section 5.2 "deterministic-mode flag for DP tests": it is simply how the step runs now).
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1


def hash_u32(x):
    """csrc/common.h: hash_u32"""
    x &= MASK
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & MASK
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & MASK
    x ^= x >> 33
    return x & 0xFFFFFFFF


def uniform_rows(seed, ctr, B, n):
    """Replay rows of update number `ctr` (csrc/replay_kernels.hip: uniform_idx_kernel and the sampler blocks of
    the optimizer launch draw from the same counter hash)."""
    return [(hash_u32(seed ^ ((ctr * GOLD + i + 1) & MASK)) * n) >> 32 for i in range(B)]


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops import dqn_engine
    return dqn_engine


def make_models(seed=0, A=6):
    from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(seed)
    opt = Options(config=2)
    return (DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda(),
            DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda())


def make_replay(n, A=6, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    s0 = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    s1 = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    tails = torch.zeros(n, 5, device="cuda")
    tails[:, 0] = torch.randint(0, A, (n,), device="cuda", generator=g).float()
    tails[:, 1] = torch.randn(n, device="cuda", generator=g).clamp_(-1, 1)
    tails[:, 2] = 0.99 ** 5
    tails[:, 3] = (torch.rand(n, device="cuda", generator=g) < 0.05).float()
    return s0, s1, tails


class TorchLearner(object):
    """fp32 PyTorch DQN learner on given replay rows: the reference loop's arithmetic."""

    def __init__(self, model, lr, target_update):
        self.model = copy.deepcopy(model)
        self.target = copy.deepcopy(model)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.target_update, self.steps = target_update, 0

    def step(self, s0, s1, tails, rows, weights=None):
        r = torch.as_tensor(rows, device="cuda", dtype=torch.long)
        a, rew, gam, term = (tails[r, i:i + 1] for i in range(4))
        with torch.no_grad():
            y = rew + gam * self.target(s1[r]).max(1, keepdim=True)[0] * (1 - term)
        pred = self.model(s0[r]).gather(1, a.long())
        w = torch.ones_like(pred) if weights is None else weights.reshape(-1, 1)
        loss = (w * (pred - y) ** 2).mean()
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        if self.steps % self.target_update == 0:        # helpers.update_target_model(model, target, period, step):
            self.target.load_state_dict(self.model.state_dict())       # hard copy AFTER the update, on the period
        self.steps += 1
        return (pred - y).detach().reshape(-1)

    def flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self.model.parameters()])


def drift(native_flat, torch_flat, init_flat):
    """|native - torch| relative to how far the torch learner moved: 0 = same trajectory."""
    moved = (torch_flat - init_flat).norm().item()
    return (native_flat - torch_flat).norm().item() / max(moved, 1e-30)


def perturbed_twin(model, lr, tu, rel=1e-6):
    """A second fp32 torch learner whose initial parameters differ by `rel` relative noise: how far two runs of
    the REFERENCE arithmetic part on this problem calibrates what "the same trajectory" can mean (Adam's
    normalised steps turn any difference in a near-zero gradient into a full step of difference, and the
    differences then feed back through the TD targets)."""
    twin = TorchLearner(model, lr, tu)
    with torch.no_grad():
        g = torch.Generator(device="cuda").manual_seed(99)
        for p in twin.model.parameters():
            p.mul_(1.0 + rel * torch.randn(p.shape, device="cuda", generator=g))
        twin.target.load_state_dict(twin.model.state_dict())
    return twin


# Bounds, in units of the distance the torch learner moved in the same updates.  Measured (tools/probes/
# trajectory_debug.py): a torch twin perturbed by 1e-6 parts by 0.08 in 21 updates; the fp32-precision engine by 0.12
# (0.01 after the FIRST update: sign flips of near-zero gradients under Adam, then the twin's growth rate); bf16 0.22.
@pytest.mark.parametrize("precision,bound", [("fp32", 0.16), ("bf16", 0.35)])
def test_uniform_trajectory_of_the_benched_path_matches_torch(eng, precision, bound):
    B, n, steps, seed, lr, tu = 128, 2048, 21, 77, 1e-4, 10
    s0, s1, tails = make_replay(n)
    model, target = make_models(seed=3)
    ref = TorchLearner(model, lr, tu)
    twin = perturbed_twin(model, lr, tu)
    init = ref.flat().clone()
    e = eng.DqnCnnEngine(model, target, B, lr=lr, target_update=tu, seed=seed, precision=precision)
    e.bind_replay(s0, s1, tails, None)
    e.set_size(n)
    prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=5)
    assert prog.graph_steps == 5 and prog.fused_tail and e.deterministic
    for _ in range(steps):              # 1 eager update + 4 replays of 5
        prog.step()
    torch.cuda.synchronize()
    assert prog.graphs is not None and e.step_count == steps and prog.ahead == 0
    for t in range(steps):
        rows = uniform_rows(e.seed, t, B, n)
        ref.step(s0, s1, tails, rows)
        twin.step(s0, s1, tails, rows)
    # the engine's index vector now holds the draw for update number `steps`
    assert e.idx.cpu().tolist() == uniform_rows(e.seed, steps, B, n)
    d = drift(e.online.flat, ref.flat(), init)
    big = ((e.online.flat - ref.flat()).abs() > 0.5 * lr).float().mean().item()
    d_twin = drift(twin.flat(), ref.flat(), init)
    print("%s: trajectory drift after %d updates = %.4f of the distance moved (a torch twin perturbed by 1e-6: %.4f); "
          "%.3f %% of the parameters differ by more than half an Adam step" % (precision, steps, d, d_twin, 100 * big))
    assert d < bound, d
    # the hard target update of update 20 (period 10) fell inside the last replay: target == online on both sides
    assert torch.equal(e.target.flat, e.online.flat)
    tflat = torch.cat([p.detach().reshape(-1) for p in ref.target.parameters()])
    assert torch.equal(tflat, ref.flat())


def test_prioritized_trajectory_matches_torch_on_the_rows_the_engine_drew(eng):
    from pytorch_distributed_amd.ops.segtree import DeviceSumMinTree
    B, n, steps, lr, tu = 64, 1024, 12, 1e-4, 6
    s0, s1, tails = make_replay(n, seed=5)
    model, target = make_models(seed=4)
    ref = TorchLearner(model, lr, tu)
    init = ref.flat().clone()
    tree = DeviceSumMinTree(n, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(9)
    tree.update(torch.arange(n, device="cuda", dtype=torch.int32), torch.rand(n, device="cuda", generator=g) + 0.05)
    per = {"eps": 1e-6, "alpha": 0.5, "beta": 0.4}
    e = eng.DqnCnnEngine(model, target, B, lr=lr, target_update=tu, seed=31, per=per, precision="fp32")
    e.bind_replay(s0, s1, tails, tree)
    e.set_size(n)
    prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=1)
    # The optimizer launch of a step also repairs the tree and draws the NEXT step's rows (trees of <= 3 levels), so
    # the rows of a step are what e.idx holds BEFORE it runs; the very first draw is made here instead of inside it.
    assert e.presamples(False)
    e.sample()
    prog.presampled = True
    for t in range(steps):
        torch.cuda.synchronize()
        idx_t = e.idx.clone()
        rows, w = idx_t.cpu().tolist(), e.isw.clone()
        prog.step()
        torch.cuda.synchronize()
        assert 0 <= min(rows) and max(rows) < n and float(w.max()) <= 1.0 + 1e-6 and float(w.min()) > 0
        td = ref.step(s0, s1, tails, rows, w)
        # priorities written back: leaf = (|td| + eps)^alpha of the engine's own TD error
        want = (e.td.abs() + per["eps"]) ** per["alpha"]
        got = tree.sum[idx_t.long()]
        assert ((got - want).abs() < 1e-5 * (1 + want)).float().mean().item() > 0.95   # duplicates: either writer
        assert (e.td - td).abs().max().item() < 5e-3 * max(1.0, td.abs().max().item()), t
    d = drift(e.online.flat, ref.flat(), init)
    print("prioritized: trajectory drift after %d updates = %.4f of the distance moved" % (steps, d))
    assert d < 0.12, d
    assert abs(tree.total().item() - tree.sum[:n].sum().item()) < 1e-3 * tree.total().item()


@pytest.mark.parametrize("precision,per", [("bf16", False), ("fp32", False), ("bf16", True), ("fp32", True)])
def test_two_runs_of_the_step_program_are_bit_identical(eng, precision, per):
    from pytorch_distributed_amd.ops.segtree import DeviceSumMinTree
    B, n, steps = 128, 2048, 16
    s0, s1, tails = make_replay(n, seed=8)
    outs = []
    for _ in range(2):
        model, target = make_models(seed=6)
        tree, per_cfg = None, None
        if per:
            tree = DeviceSumMinTree(n, torch.device("cuda"))
            g = torch.Generator(device="cuda").manual_seed(2)
            tree.update(torch.arange(n, device="cuda", dtype=torch.int32), torch.rand(n, device="cuda", generator=g) + 0.05)
            per_cfg = {"eps": 1e-6, "alpha": 0.5, "beta": 0.4}
        e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=5, seed=13, per=per_cfg, precision=precision)
        e.bind_replay(s0, s1, tails, tree)
        e.set_size(n)
        prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=5)
        for _ in range(steps):
            prog.step()
        torch.cuda.synchronize()
        outs.append((e.online.flat.clone(), e.target.flat.clone(), e.adam.m.clone(), e.adam.v.clone(),
                     e.online.shadow.clone(), tree.sum.clone() if tree is not None else torch.zeros(1),
                     float(np.float32(e.mean_loss()))))
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert torch.equal(a, b)
    assert outs[0][-1] == outs[1][-1]
