"""Native DQN learner step (ops/dqn_engine.py) vs the fp32 PyTorch step of the same
model (agents/learners.py:DqnTorchCore arithmetic) on a real MI355X."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops import dqn_engine
    return dqn_engine


def make_models(A=6, seed=0):
    from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(seed)
    opt = Options(config=2)
    m = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda()
    t = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda()
    return m, t


def make_batch(B, A, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    s0 = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    s1 = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    tails = torch.zeros(B, 5, device="cuda")
    tails[:, 0] = torch.randint(0, A, (B,), device="cuda", generator=g).float()
    tails[:, 1] = torch.randn(B, device="cuda", generator=g)
    tails[:, 2] = 0.99 ** 5
    tails[:, 3] = (torch.rand(B, device="cuda", generator=g) < 0.1).float()
    return s0, s1, tails


def torch_reference(model, target, s0, s1, tails, isw=None, double=False):
    model.zero_grad(set_to_none=True)
    a, r, g, t = tails[:, 0:1], tails[:, 1:2], tails[:, 2:3], tails[:, 3:4]
    with torch.no_grad():
        if double:
            na = model(s1).argmax(1, keepdim=True)
            nq = target(s1).gather(1, na)
        else:
            nq = target(s1).max(1, keepdim=True)[0]
        y = r + g * nq * (1 - t)
    qall = model(s0)
    pred = qall.gather(1, a.long())
    w = isw.reshape(-1, 1) if isw is not None else torch.ones_like(pred)
    loss = (w * (pred - y) ** 2).mean()
    loss.backward()
    return qall.detach(), (pred - y).detach().reshape(-1), loss.item()


class _Bf16STE(torch.autograd.Function):
    """Round to bf16 in forward, identity in backward."""
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


def bf16_emulated_q(model, s):
    """The engine's forward arithmetic in torch: bf16 weights and activations,
    fp32 accumulation, fp32 head."""
    r = _Bf16STE.apply
    c, f = model.critic[0], model.critic[1]
    x = F.conv2d(s.float(), r(c[0].weight), None, stride=4) * (1.0 / 255.0) + c[0].bias[None, :, None, None]
    x = r(F.relu(x))
    x = r(F.relu(F.conv2d(x, r(c[2].weight), c[2].bias, stride=2)))
    x = r(F.relu(F.conv2d(x, r(c[4].weight), c[4].bias, stride=1)))
    h = F.relu(F.linear(x.flatten(1), r(f[0].weight), f[0].bias))
    return F.linear(h, f[2].weight, f[2].bias)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,double,A", [(128, False, 6), (32, True, 18)])
def test_forward_backward_matches_torch(eng, B, double, A):
    model, target = make_models(A)
    e = eng.DqnCnnEngine(model, target, B, double=double, target_update=10 ** 9)
    with torch.no_grad():       # the engine starts with target == online: make them differ
        for p in target.parameters():
            p.add_(0.01 * torch.randn_like(p))
    e.target.refresh()
    ref_model, tgt_ref = copy.deepcopy(model), copy.deepcopy(target)
    s0, s1, tails = make_batch(B, A)
    e.make_staging(5)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    e.forward_head(staged=True)
    e.backward_fc()
    e.backward_conv(staged=True)
    torch.cuda.synchronize()
    q_ref, td_ref, loss_ref = torch_reference(ref_model, tgt_ref, s0, s1, tails, double=double)
    assert rel_err(e.q, q_ref) < 2e-2
    assert (e.td - td_ref).abs().max().item() < 3e-2 * max(1.0, td_ref.abs().max().item())
    assert abs(e.loss_acc.sum().item() - loss_ref) < 3e-2 * max(1.0, abs(loss_ref))
    names = [n for n, _ in ref_model.named_parameters()]
    off = 0
    errs = {}
    grads = e.grads_torch_layout()
    for name, p in ref_model.named_parameters():
        got = grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        errs[name] = rel_err(got, p.grad)
    print("grad rel errs:", {k: round(v, 4) for k, v in errs.items()})
    # bf16 activations/gradients: the error grows with depth of the backward chain
    for name, err in errs.items():
        limit = 0.12 if name.startswith("critic.0") else 6e-2
        assert err < limit, "%s: rel err %g (all: %s)" % (name, err, errs)
    assert names[0] == "critic.0.0.weight"
    # tight check against a torch model with the same rounding points: isolates kernel
    # arithmetic from bf16 relu-mask flips
    ref_model.zero_grad(set_to_none=True)
    a = tails[:, 0:1].long()
    y = (e.q.gather(1, a) - e.td[:, None]).detach()      # the engine's own TD target
    pred = bf16_emulated_q(ref_model, s0).gather(1, a)
    ((pred - y) ** 2).mean().backward()
    off, errs2 = 0, {}
    for name, p in ref_model.named_parameters():
        got = grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        errs2[name] = rel_err(got, p.grad)
    print("grad rel errs vs bf16-emulated torch:", {k: round(v, 4) for k, v in errs2.items()})
    for name, err in errs2.items():
        assert err < 2.5e-2, "%s: rel err %g vs bf16-emulated reference (all: %s)" % (name, err, errs2)


def test_adam_and_target_update_match_torch(eng):
    model, target = make_models()
    ref = copy.deepcopy(model)
    e = eng.DqnCnnEngine(model, target, 32, lr=1e-3, target_update=2)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    torch.manual_seed(3)
    for it in range(3):
        gr = torch.randn_like(e.grads) * 0.1
        e.grads.copy_(e.online.to_kernel_layout(gr))     # gradients live in kernel layout
        off = 0
        for p in ref.parameters():
            p.grad = gr[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        opt.step()
        e.update(staged=True, tick=True)
    flat_ref = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    assert (e.online.flat - flat_ref).abs().max().item() < 1e-5
    # bf16 shadows follow the master weights, in kernel layout
    w2 = model.critic[0][2].weight.detach()
    from pytorch_distributed_amd.ops import nn as pnn
    assert torch.equal(e.online.w[1], pnn.pack_conv_weight(w2))
    w4 = model.critic[1][0].weight.detach()
    assert torch.equal(e.online.w[3], pnn.pack_fc_after_conv(w4, 64, 49))
    e.hard_target_update()
    assert torch.equal(e.target.flat, e.online.flat)
    assert torch.equal(e.target.w[3], e.online.w[3])


def test_overfits_fixed_batch_with_graph(eng):
    model, target = make_models()
    B = 64
    e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=10 ** 9)
    s0, s1, tails = make_batch(B, 6)
    tails[:, 3] = 1.0       # terminal: the target is just the reward -> a pure regression problem
    e.make_staging(5)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    prog = eng.DqnStepProgram(e, None, use_graph=True, staged=True)
    prog.step()
    first = e.mean_loss()
    for _ in range(300):
        prog.step()
    e.mean_loss()
    for _ in range(10):
        prog.step()
    last = e.mean_loss()
    assert prog.graphs is not None
    assert np.isfinite(last) and last < 0.2 * first, (first, last)


def test_device_replay_uniform_and_per_steps(eng):
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.utils.options import Options
    from pytorch_distributed_amd.launcher import prepare_options
    for cfg in (2, 4):
        opt = prepare_options(Options(config=cfg, overrides={"memory_params.memory_size": 4096, "graph_steps": 1,
                                                             "agent_params.batch_size": 64}))
        mem = DeviceMemory(opt.memory_params)
        mem.attach(torch.device("cuda"))
        n = 1000
        g = torch.Generator(device="cuda").manual_seed(5)
        s = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
        mem.feed_batch(s, torch.randint(0, 6, (n, 1), device="cuda"), torch.randn(n, 1, device="cuda"),
                       torch.full((n, 1), 0.95, device="cuda"), s.flip(0), torch.zeros(n, 1, device="cuda"),
                       priorities=torch.rand(n, device="cuda") + 0.1)
        model, target = make_models()
        core = eng.DqnNativeCore(opt, model, target, mem)
        for _ in range(5):
            core.step()
        torch.cuda.synchronize()
        # (the optimizer launch of an update also draws the next update's rows: the rows of the sixth update are
        # what the engine holds after the fifth)
        assert core.engine.presamples(False)
        idx_last = core.engine.idx.clone()
        core.step()
        torch.cuda.synchronize()
        idx = idx_last.cpu().numpy()
        assert idx.min() >= 0 and idx.max() < n, (idx.min(), idx.max())
        assert len(np.unique(idx)) > 16
        loss = core.loss_values()["critic_loss"]
        assert np.isfinite(loss)
        if cfg == 4:
            tree = mem.tree
            leaves = tree.sum[:n]
            want = (core.engine.td.abs() + mem.priority_eps) ** mem.priority_exponent
            got = leaves[idx_last.long()]
            # duplicates inside a batch may have been written by either sample
            close = (got - want).abs() < 1e-3 * (1 + want.abs())
            assert close.float().mean().item() > 0.9
            assert abs(tree.total().item() - tree.sum[:4096].sum().item()) < 1e-2 * tree.total().item()
            assert abs(tree.min_value().item() - tree.min[:4096].min().item()) < 1e-6


def test_multi_step_graph_runs_ahead_and_keeps_target_updates_aligned(eng):
    """graph_steps > 1: one replay performs K updates (K cut down to a divisor of the target-update period), the
    following K - 1 step() calls return at once, and the hard target update still lands on its period."""
    B = 32
    s0, s1, tails = make_batch(B, 6)
    model, target = make_models(seed=3)
    e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=6, seed=5)
    mem_n = 256
    g = torch.Generator(device="cuda").manual_seed(1)
    frames = torch.randint(0, 256, (mem_n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    rt = torch.zeros(mem_n, tails.shape[1], device="cuda")
    rt[:, :] = tails[0]
    e.bind_replay(frames, frames.flip(0).contiguous(), rt, None)
    e.set_size(mem_n)
    prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=4)
    assert prog.graph_steps == 3            # 6 % 4 != 0 -> 3
    prog.step()                             # eager, update 0 (+ target update)
    assert e.step_count == 1 and prog.ahead == 0
    prog.step()                             # captures and replays updates 1..3
    assert e.step_count == 4 and prog.ahead == 2
    before = e.online.flat.clone()
    prog.step(); prog.step()                # already done
    torch.cuda.synchronize()
    assert torch.equal(before, e.online.flat) and prog.ahead == 0
    prog.step()                             # updates 4..6: the last one is due for the hard target update
    torch.cuda.synchronize()
    assert e.step_count == 7
    assert torch.equal(e.target.flat, e.online.flat)
    assert np.isfinite(e.mean_loss())


def test_engine_knobs_are_resolved_once_at_construction(eng, monkeypatch):
    """Tuning knobs come from Options.engine_knobs / PDA_<NAME> through ops/knobs.py when the engine is BUILT; the
    step itself never reads the environment.  A different fc1 split changes the slab count, not the result."""
    from pytorch_distributed_amd.ops.knobs import resolve
    B = 128
    s0, s1, tails = make_batch(B, 6)
    mem_n = 512
    g = torch.Generator(device="cuda").manual_seed(2)
    frames = torch.randint(0, 256, (mem_n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    rt = tails[torch.arange(mem_n, device="cuda") % B].contiguous()
    outs = []
    for blocks in (256, 64):
        monkeypatch.setenv("PDA_FC1_BLOCKS", str(blocks))
        model, target = make_models(seed=7)
        e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=1000, seed=11)
        assert e.knobs.fc1_blocks == blocks and resolve().fc1_blocks == blocks
        monkeypatch.setenv("PDA_FC1_BLOCKS", "999")     # too late: the engine holds its knobs
        e.bind_replay(frames, frames.flip(0).contiguous(), rt, None)
        e.set_size(mem_n)
        prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=2)
        for _ in range(5):
            prog.step()
        torch.cuda.synchronize()
        assert e.knobs.fc1_blocks == blocks and e.step_count == 5
        outs.append((e.acts.h1_slices, e.online.flat.clone(), e.grads.clone()))
    (n0, p0, g0), (n1, p1, g1) = outs
    assert n0 > n1 >= 1
    assert (p0 - p1).abs().max().item() < 1e-4 and rel_err(g1, g0) < 1e-3
    with pytest.raises(ValueError):
        resolve(dp_graph="sideways")


def test_batched_inference_matches_model(eng):
    model, _ = make_models()
    model.eval()
    inf = eng.DqnCnnInference(model, max_batch=256)
    obs = torch.randint(0, 256, (200, 4, 84, 84), dtype=torch.uint8, device="cuda")
    a, q, mq = inf.act(obs, None)
    with torch.no_grad():
        ref = model(obs)
    assert rel_err(inf.q_values(200), ref) < 2e-2
    top2 = ref.topk(2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 0.02 * ref.abs().max()
    assert torch.equal(a[clear], ref.argmax(1)[clear])
    assert (mq - ref.max(1)[0]).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    # epsilon = 1 -> uniformly random actions, q reported for the action taken
    eps = torch.ones(200, device="cuda")
    a2, q2, _ = inf.act(obs, eps)
    assert len(torch.unique(a2)) == 6
    assert (q2 - inf.q_values(200).gather(1, a2[:, None])[:, 0]).abs().max().item() < 1e-5


def test_dp_step_program_one_rank_rccl_equals_single(eng):
    """The data-parallel step program (side-stream RCCL all-reduces, split Adam) with one rank
    must reproduce the single-GPU program exactly: same kernels, identity collectives."""
    import socket
    from pytorch_distributed_amd.parallel.comm import TorchComm
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    comm = TorchComm(0, 1, torch.device("cuda:0"), init_method="tcp://127.0.0.1:%d" % port)
    try:
        B = 32
        s0, s1, tails = make_batch(B, 6)
        tails[:, 3] = 1.0
        flats = []
        for dp in (False, True):
            model, target = make_models(seed=7)
            e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=3, seed=5)
            e.make_staging(5)
            e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
            prog = eng.DqnStepProgram(e, comm if dp else None, use_graph=True, staged=True, force_dp=dp)
            assert prog.dp == dp
            for _ in range(8):
                prog.step()
            torch.cuda.synchronize()
            assert prog.graphs is not None
            if dp:
                print("dp graph mode:", prog.dp_graph_mode)
                assert prog.dp_graph_mode in ("whole", "split")
            flats.append((e.online.flat.clone(), e.target.flat.clone(), e.mean_loss()))
        # fc1's forward split-K sums with fp32 atomics, and the conv split-K slabs are added in a different
        # association order by the reduce launch (DP) and by the optimizer's folded sum (single rank), so
        # the two runs differ in the last bits and Adam (g / |g|) amplifies that on near-zero gradients:
        # compare with a tolerance
        for i in (0, 1):
            d = (flats[0][i] - flats[1][i]).abs()
            assert d.max().item() <= 8 * 3e-4 and (d > 2e-5).float().mean().item() < 5e-2, \
                (d.max().item(), (d > 2e-5).float().mean().item())
        assert abs(flats[0][2] - flats[1][2]) < 2e-2 * max(1.0, abs(flats[0][2]))
    finally:
        comm.close()
