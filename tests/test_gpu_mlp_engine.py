"""Native DDPG / MLP-DQN learner steps (ops/mlp_engine.py) vs the fp32 PyTorch
arithmetic of agents/learners.py (DdpgTorchCore / DqnTorchCore) on a real MI355X."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops import mlp_engine
    return mlp_engine


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def make_ddpg(D=3, A=1, seed=0):
    from pytorch_distributed_amd.models.ddpg_mlp_model import DdpgMlpModel
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(seed)
    opt = Options(config=3)
    m = DdpgMlpModel(opt.model_params, 1.0, (1, D), A, A).cuda()
    t = DdpgMlpModel(opt.model_params, 1.0, (1, D), A, A).cuda()
    with torch.no_grad():       # the reference init makes the last layers ~0: give the test some signal
        for net in (m, t):
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
    return m, t


def ddpg_batch(B, D, A, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    s0 = torch.randn(B, D, device="cuda", generator=g)
    s1 = torch.randn(B, D, device="cuda", generator=g)
    tails = torch.zeros(B, A + 4, device="cuda")
    tails[:, :A] = torch.rand(B, A, device="cuda", generator=g) * 2 - 1
    tails[:, A] = torch.randn(B, device="cuda", generator=g)
    tails[:, A + 1] = 0.99 ** 5
    tails[:, A + 2] = (torch.rand(B, device="cuda", generator=g) < 0.1).float()
    return s0, s1, tails


def ddpg_reference(model, target, s0, s1, tails, A, leak, isw=None, actor_l2=0.0):
    """The reference's two backward passes (ddpg_learner.py:66-86)."""
    model.zero_grad(set_to_none=True)
    a, r, g, t = tails[:, :A], tails[:, A:A + 1], tails[:, A + 1:A + 2], tails[:, A + 2:A + 3]
    if not leak:
        for p in model.critic.parameters():
            p.requires_grad_(False)
    z = model.actor[:-1](s0.flatten(1))
    q_pi = model.forward_critic(s0, torch.tanh(z))
    actor_loss = -q_pi.mean() + actor_l2 * (z ** 2).sum(1).mean()
    actor_loss.backward()
    for p in model.critic.parameters():
        p.requires_grad_(True)
    with torch.no_grad():
        _, nq = target(s1)
        y = r + g * nq * (1 - t)
    pred = model.forward_critic(s0, a)
    w = isw.reshape(-1, 1) if isw is not None else torch.ones_like(pred)
    critic_loss = (w * (pred - y) ** 2).mean()
    critic_loss.backward()
    return q_pi.detach().reshape(-1), (pred - y).detach().reshape(-1), actor_loss.item(), critic_loss.item()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("B,D,A,leak", [(64, 3, 1, True), (64, 3, 1, False), (128, 17, 6, True), (40, 5, 2, False)])
def test_ddpg_forward_backward_matches_torch(eng, B, D, A, leak, fused):
    model, target = make_ddpg(D, A)
    e = eng.DdpgMlpEngine(model, target, B, leak=leak, tail_offsets=(0, A, A + 1, A + 2))
    assert e.fused, "the fused DDPG kernel must support the reference's layer sizes"
    with torch.no_grad():
        for p in target.parameters():
            p.add_(0.02 * torch.randn_like(p))
    e.target.refresh()
    ref, tref = copy.deepcopy(model), copy.deepcopy(target)
    s0, s1, tails = ddpg_batch(B, D, A)
    e.make_staging(A + 4)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    if fused:       # one launch for forward + head + data gradients, one for the six weight gradients
        e.fused_forward_backward(staged=True)
        e.wgrad_all()
    else:
        e.forward_head(staged=True)
        e.backward(staged=True)
    torch.cuda.synchronize()
    q_pi, td, al, cl = ddpg_reference(ref, tref, s0, s1, tails, A, leak)
    assert rel_err(e.q[:B], q_pi) < 3e-2
    assert (e.td - td).abs().max().item() < 3e-2 * max(1.0, td.abs().max().item())
    la, lc = e.loss_acc.tolist()
    assert abs(la - al) < 3e-2 * max(1.0, abs(al)) and abs(lc - cl) < 3e-2 * max(1.0, abs(cl))
    off, errs = 0, {}
    for name, p in ref.named_parameters():
        got = e.grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        errs[name] = rel_err(got, p.grad)
    print("ddpg grad rel errs:", {k: round(v, 4) for k, v in errs.items()})
    for name, err in errs.items():
        assert err < 6e-2, "%s: rel err %g (all: %s)" % (name, err, errs)


def test_ddpg_fused_kernel_equals_the_layer_by_layer_launches(eng):
    """Same bf16 operands, same fp32 accumulation order per 32-wide k-step: the fused launch must reproduce the
    unfused buffers (activations, data gradients, flat parameter gradient) to rounding of the last bf16 bit."""
    B, D, A = 64, 3, 1
    s0, s1, tails = ddpg_batch(B, D, A)
    outs = []
    for fused in (False, True):
        model, target = make_ddpg(D, A, seed=3)
        e = eng.DdpgMlpEngine(model, target, B, leak=True, tail_offsets=(0, A, A + 1, A + 2))
        e.make_staging(A + 4)
        e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
        if fused:
            e.fused_forward_backward(staged=True)
            e.wgrad_all()
        else:
            e.forward_head(staged=True)
            e.backward(staged=True)
        torch.cuda.synchronize()
        outs.append({"q": e.q.clone(), "td": e.td.clone(), "h2": e.h2[:2 * B].float(), "cat": e.cat_on.float(),
                     "dh2": e.dh2.float(), "dcat": e.dcat.float(), "dha2": e.dha2.float(), "dha1": e.dha1.float(),
                     "grads": e.grads.clone()})
    a, b = outs
    for k in a:
        err = rel_err(b[k], a[k])
        print("fused vs unfused %-6s rel err %.2e" % (k, err))
        assert err < 2e-3, (k, err)


def test_ddpg_actor_preact_penalty_matches_torch(eng):
    """actor_l2 * mean |z|^2 on the pre-tanh actor output (what keeps the tanh policy out of saturation for good)."""
    B, D, A, lam = 64, 3, 2, 0.5
    model, target = make_ddpg(D, A, seed=7)
    e = eng.DdpgMlpEngine(model, target, B, leak=False, tail_offsets=(0, A, A + 1, A + 2), actor_l2=lam)
    ref, tref = copy.deepcopy(model), copy.deepcopy(target)
    s0, s1, tails = ddpg_batch(B, D, A)
    e.make_staging(A + 4)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    e.fused_forward_backward(staged=True)
    e.wgrad_all()
    torch.cuda.synchronize()
    _, _, al, _ = ddpg_reference(ref, tref, s0, s1, tails, A, False, actor_l2=lam)
    assert abs(e.loss_acc[0].item() - al) < 3e-2 * max(1.0, abs(al))
    off = 0
    for name, p in ref.named_parameters():
        got = e.grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        if name.startswith("actor"):
            assert rel_err(got, p.grad) < 3e-2, (name, rel_err(got, p.grad))


def test_ddpg_fused_step_is_three_launches_and_keeps_the_transposed_shadows(eng):
    """Uniform sampling from device replay: fused forward/backward, grouped weight gradients, optimizer (+ Polyak +
    transposed shadows + next index draw).  After a few steps the transposed copies equal the masters' transposes
    and the target follows by Polyak."""
    B, D, A, n = 64, 3, 1, 512
    model, target = make_ddpg(D, A, seed=5)
    e = eng.DdpgMlpEngine(model, target, B, lr=1e-3, tau=0.01, leak=False, tail_offsets=(0, A, A + 1, A + 2), seed=9)
    g = torch.Generator(device="cuda").manual_seed(4)
    s0 = torch.randn(n, D, device="cuda", generator=g)
    s1 = torch.randn(n, D, device="cuda", generator=g)
    tails = torch.zeros(n, A + 4, device="cuda")
    tails[:, :A] = torch.rand(n, A, device="cuda", generator=g) * 2 - 1
    tails[:, A] = torch.randn(n, device="cuda", generator=g)
    tails[:, A + 1] = 0.99
    e.bind_replay(s0, s1, tails, None)
    e.set_size(n)
    tgt0 = e.target.flat.clone()
    prog = eng.MlpStepProgram(e, None, use_graph=True, staged=False, graph_steps=1)
    assert prog.fused and prog.extras is not None
    seen = []
    for _ in range(6):
        prog.step()
        torch.cuda.synchronize()
        seen.append(e.idx.clone())
    assert prog.graphs is not None and prog.presampled
    assert all(0 <= int(i.min()) and int(i.max()) < n for i in seen)
    assert not torch.equal(seen[-1], seen[-2])          # a fresh draw per step, made by the optimizer launch
    for name in ("critic.1.0", "actor.2"):
        w = dict(model.named_parameters())[name + ".weight"].detach()
        nn_, kk = w.shape
        assert torch.equal(e.online.wT[name][:, :nn_], w.t().to(torch.bfloat16)), name
        assert torch.equal(e.online.w[name][:, :kk], w.to(torch.bfloat16)), name
    assert (e.target.flat - tgt0).abs().max().item() > 0
    assert torch.equal(e.target.w["critic.1.0"][:, :e.CK],
                       target.critic[1][0].weight.detach().to(torch.bfloat16))
    import ctypes
    assert ctypes.sizeof(eng._DdpgFusedArgs) < 4096


def test_ddpg_steps_track_torch_core(eng):
    """A few full native steps (graph-captured) against Adam + Polyak in torch on the same batch."""
    B, D, A = 64, 3, 1
    model, target = make_ddpg(D, A)
    ref, tref = copy.deepcopy(model), copy.deepcopy(target)
    e = eng.DdpgMlpEngine(model, target, B, lr=1e-3, clip=40.0, tau=0.05, tail_offsets=(0, A, A + 1, A + 2))
    tref.load_state_dict(ref.state_dict())      # the engine hard-copies online -> target at start
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    s0, s1, tails = ddpg_batch(B, D, A)
    e.make_staging(A + 4)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    prog = eng.MlpStepProgram(e, None, use_graph=True, staged=True)
    for it in range(4):
        prog.step()
        ddpg_reference(ref, tref, s0, s1, tails, A, True)
        for p in ref.parameters():
            p.grad.clamp_(-40.0, 40.0)
        opt.step()
        with torch.no_grad():
            for tp, p in zip(tref.parameters(), ref.parameters()):
                tp.mul_(0.95).add_(0.05 * p)
    torch.cuda.synchronize()
    assert prog.graphs is not None
    flat_ref = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    flat_tref = torch.cat([p.detach().reshape(-1) for p in tref.parameters()])
    # 4 Adam steps of lr 1e-3 move each weight by <= 4e-3; sign flips of tiny bf16-noisy gradients bound the gap
    assert (e.online.flat - flat_ref).abs().max().item() < 8.5e-3
    assert (e.online.flat - flat_ref).abs().mean().item() < 6e-4
    assert (e.target.flat - flat_tref).abs().max().item() < 2e-3
    # state_dict still sees the trained parameters (views of the flat buffer)
    sd = model.state_dict()
    assert torch.equal(sd["actor.0.weight"].reshape(-1), e.online.flat[:sd["actor.0.weight"].numel()])
    # shadows follow the masters
    w = e.online.w["critic.1.0"]
    assert torch.equal(w[:, :e.CK], model.critic[1][0].weight.detach().to(torch.bfloat16))
    assert w[:, e.CK:].abs().max().item() == 0.0


def test_ddpg_actor_warmup_native(eng):
    """actor_warmup: the head kernel switches the actor loss off while the update count is below it."""
    B, D, A = 64, 3, 1
    model, target = make_ddpg(D, A)
    e = eng.DdpgMlpEngine(model, target, B, lr=1e-3, clip=40.0, tau=0.05, tail_offsets=(0, A, A + 1, A + 2),
                          leak=False, actor_warmup=3)
    s0, s1, tails = ddpg_batch(B, D, A)
    e.make_staging(A + 4)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    prog = eng.MlpStepProgram(e, None, use_graph=True, staged=True)
    n_actor = sum(p.numel() for p in model.actor.parameters())
    flat0 = e.online.flat.clone()
    for _ in range(2):
        prog.step()
    torch.cuda.synchronize()
    assert torch.equal(e.online.flat[:n_actor], flat0[:n_actor])
    assert (e.online.flat[n_actor:] - flat0[n_actor:]).abs().max().item() > 0
    for _ in range(3):
        prog.step()
    torch.cuda.synchronize()
    assert (e.online.flat[:n_actor] - flat0[:n_actor]).abs().max().item() > 0


def test_ddpg_overfits_and_per_device_replay(eng):
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.utils.options import Options
    from pytorch_distributed_amd.launcher import prepare_options
    opt = prepare_options(Options(config=7, overrides={"memory_params.memory_size": 4096,
                                                       "memory_params.enable_per": True,
                                                       "agent_params.batch_size": 64}))
    mem = DeviceMemory(opt.memory_params)
    mem.attach(torch.device("cuda"))
    n, D, A = 1000, int(np.prod(opt.state_shape)), int(opt.action_shape)
    g = torch.Generator(device="cuda").manual_seed(5)
    s = torch.randn((n,) + tuple(opt.state_shape), device="cuda", generator=g)
    reward = s.reshape(n, -1)[:, :1].clone()        # learnable: r is a function of the state
    mem.feed_batch(s, torch.rand(n, A, device="cuda") * 2 - 1, reward,
                   torch.full((n, 1), 0.95, device="cuda"), s.flip(0), torch.ones(n, 1, device="cuda"),
                   priorities=torch.rand(n, device="cuda") + 0.1)
    from pytorch_distributed_amd.models.ddpg_mlp_model import DdpgMlpModel
    torch.manual_seed(0)
    model = DdpgMlpModel(opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, opt.action_shape).cuda()
    target = DdpgMlpModel(opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, opt.action_shape).cuda()
    opt.agent_params.lr = 1e-3
    # with the reference's gradient leak the actor loss also pushes the critic's Q up without
    # bound on a fixed data set; the regression check wants the critic loss alone
    opt.agent_params.compat_ddpg_grad_leak = False
    core = eng.MlpNativeCore("ddpg", opt, model, target, mem)
    core.step()
    first = core.loss_values()["critic_loss"]
    for _ in range(400):
        core.step()
    core.loss_values()
    for _ in range(20):
        core.step()
    torch.cuda.synchronize()
    last = core.loss_values()
    idx = core.engine.idx.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < n
    assert np.isfinite(last["critic_loss"]) and np.isfinite(last["actor_loss"])
    assert last["critic_loss"] < 0.5 * first, (first, last)      # terminal transitions: pure regression on r
    tree = mem.tree
    want = (core.engine.td.abs() + mem.priority_eps) ** mem.priority_exponent
    got = tree.sum[:n][core.engine.idx.long()]
    close = (got - want).abs() < 1e-3 * (1 + want.abs())
    assert close.float().mean().item() > 0.9
    assert abs(tree.total().item() - tree.sum[:4096].sum().item()) < 1e-2 * tree.total().item()


def test_ddpg_batched_actor_matches_model(eng):
    model, _ = make_ddpg(11, 3)
    inf = eng.DdpgActorInference(model, max_batch=300)
    obs = torch.randn(257, 11, device="cuda")
    got = inf.act(obs).clone()
    with torch.no_grad():
        ref = model.forward_actor(obs)
    assert (got - ref).abs().max().item() < 2e-2
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.5)
    inf.refresh_weights()
    with torch.no_grad():
        ref2 = model.forward_actor(obs)
    assert (inf.act(obs) - ref2).abs().max().item() < 2e-2


@pytest.mark.parametrize("double", [False, True])
def test_dqn_mlp_forward_backward_matches_torch(eng, double):
    from pytorch_distributed_amd.models.dqn_mlp_model import DqnMlpModel
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(0)
    opt = Options(config=6)
    B, D, A = 32, 10, 4
    model = DqnMlpModel(opt.model_params, 2.0, (1, D), A, 1).cuda()
    target = DqnMlpModel(opt.model_params, 2.0, (1, D), A, 1).cuda()
    e = eng.DqnMlpEngine(model, target, B, double=double, target_update=10 ** 9)
    with torch.no_grad():
        for p in target.parameters():
            p.add_(0.02 * torch.randn_like(p))
    e.target.refresh()
    ref, tref = copy.deepcopy(model), copy.deepcopy(target)
    g = torch.Generator(device="cuda").manual_seed(2)
    s0 = torch.randn(B, D, device="cuda", generator=g)
    s1 = torch.randn(B, D, device="cuda", generator=g)
    tails = torch.zeros(B, 5, device="cuda")
    tails[:, 0] = torch.randint(0, A, (B,), device="cuda", generator=g).float()
    tails[:, 1] = torch.randn(B, device="cuda", generator=g)
    tails[:, 2] = 0.99 ** 3
    tails[:, 3] = (torch.rand(B, device="cuda", generator=g) < 0.2).float()
    e.make_staging(5)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    e.forward_head(staged=True)
    e.backward(staged=True)
    torch.cuda.synchronize()
    a, r, gm, t = tails[:, 0:1], tails[:, 1:2], tails[:, 2:3], tails[:, 3:4]
    with torch.no_grad():
        if double:
            nq = tref(s1).gather(1, ref(s1).argmax(1, keepdim=True))
        else:
            nq = tref(s1).max(1, keepdim=True)[0]
        y = r + gm * nq * (1 - t)
    qall = ref(s0)
    loss = ((qall.gather(1, a.long()) - y) ** 2).mean()
    loss.backward()
    assert rel_err(e.q, qall.detach()) < 2e-2
    assert abs(e.loss_acc.sum().item() - loss.item()) < 3e-2 * max(1.0, loss.item())
    off, errs = 0, {}
    for name, p in ref.named_parameters():
        got = e.grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        errs[name] = rel_err(got, p.grad)
    print("dqn-mlp grad rel errs:", {k: round(v, 4) for k, v in errs.items()})
    # relu masks are taken from bf16-input activations: units within rounding of zero flip
    # against the fp32 reference, and every flip adds or drops a whole contribution below it
    for name, err in errs.items():
        limit = 6e-2 if name.startswith(("critic.4", "critic.6")) else 0.12
        assert err < limit, "%s: rel err %g (all: %s)" % (name, err, errs)
