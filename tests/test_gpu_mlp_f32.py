"""Distributed DDPG at the reference's precision (ops/mlp_f32_engine.py, csrc/mlp_f32.hip: fp32 operands on the fp32-input
MFMA) against the reference learner's two backward passes (/root/reference/core/single_processes/ddpg_learner.py:66-86)
run in fp64 -- and beside plain fp32 PyTorch's own distance from that fp64 step."""
import copy

import pytest
import torch

from test_gpu_mlp_engine import ddpg_batch, make_ddpg, rel_err

pytestmark = pytest.mark.gpu


def reference_step(model, target, s0, s1, tails, A, leak, actor_l2, dtype):
    """The reference's two backward passes (/root/reference/core/single_processes/ddpg_learner.py:66-86) in `dtype`
    (the model's own forward casts its input to fp32, so the layers are applied here directly)."""
    model, target = copy.deepcopy(model).to(dtype), copy.deepcopy(target).to(dtype)
    s0, s1, tails = s0.to(dtype), s1.to(dtype), tails.to(dtype)

    def critic(net, s, act):
        return net.critic[1](torch.cat((net.critic[0](s), act), 1))
    a, r, g, t = tails[:, :A], tails[:, A:A + 1], tails[:, A + 1:A + 2], tails[:, A + 2:A + 3]
    if not leak:
        for p in model.critic.parameters():
            p.requires_grad_(False)
    z = model.actor[:-1](s0)
    q_pi = critic(model, s0, torch.tanh(z))
    actor_loss = -q_pi.mean() + actor_l2 * (z ** 2).sum(1).mean()
    actor_loss.backward()
    for p in model.critic.parameters():
        p.requires_grad_(True)
    with torch.no_grad():
        y = r + g * critic(target, s1, target.actor(s1)) * (1 - t)
    pred = critic(model, s0, a)
    critic_loss = ((pred - y) ** 2).mean()
    critic_loss.backward()
    return model, q_pi.detach().reshape(-1), (pred - y).detach().reshape(-1), actor_loss.item(), critic_loss.item()


@pytest.fixture(scope="module")
def f32():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops import mlp_f32_engine
    return mlp_f32_engine


@pytest.mark.parametrize("B,D,A,leak,l2", [(64, 3, 1, True, 0.0), (64, 3, 1, False, 0.01), (128, 17, 6, True, 0.0),
                                           (40, 5, 2, False, 0.02)])
def test_ddpg_fp32_engine_matches_the_fp64_reference_step(f32, B, D, A, leak, l2):
    model, target = make_ddpg(D, A)
    e = f32.DdpgMlpEngineF32(model, target, B, leak=leak, tail_offsets=(0, A, A + 1, A + 2), actor_l2=l2)
    with torch.no_grad():
        for p in target.parameters():
            p.add_(0.02 * torch.randn_like(p))
    s0, s1, tails = ddpg_batch(B, D, A)
    e.make_staging(A + 4)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    e.fused_forward_backward(staged=True)
    e.wgrad_all()
    torch.cuda.synchronize()
    ref64, q_pi, td, al, cl = reference_step(model, target, s0, s1, tails, A, leak, l2, torch.float64)
    ref32 = reference_step(model, target, s0, s1, tails, A, leak, l2, torch.float32)[0]
    assert rel_err(e.q[:B].double(), q_pi) < 1e-5
    assert (e.td.double() - td).abs().max().item() < 1e-5 * max(1.0, td.abs().max().item())
    la, lc = e.loss_acc.tolist()
    assert abs(la - al) < 1e-5 * max(1.0, abs(al)) and abs(lc - cl) < 1e-5 * max(1.0, abs(cl))
    off, errs, errs32 = 0, {}, {}
    for (name, p64), (_, p32) in zip(ref64.named_parameters(), ref32.named_parameters()):
        got = e.grads[off:off + p64.numel()].view_as(p64)
        off += p64.numel()
        errs[name] = ((got.double() - p64.grad).norm() / p64.grad.norm().clamp_min(1e-30)).item()
        errs32[name] = ((p32.grad.double() - p64.grad).norm() / p64.grad.norm().clamp_min(1e-30)).item()
    print("fp32 engine grad rel errs vs fp64:", {k: float("%.2g" % v) for k, v in errs.items()})
    print("fp32 torch  grad rel errs vs fp64:", {k: float("%.2g" % v) for k, v in errs32.items()})
    for name, err in errs.items():
        # true fp32 arithmetic: within 1e-5 of fp64 and within 8x of PyTorch fp32's own distance from it
        assert err < 1e-5, "%s: rel err %g (all: %s)" % (name, err, errs)
        assert err <= 8.0 * max(errs32[name], 2e-7), "%s: %g vs torch fp32 %g" % (name, err, errs32[name])


def test_precision_option_selects_the_engine_and_is_never_ignored(f32):
    """precision='fp32' (the default) -> the fp32 engine; 'bf16' -> the bf16 fused engine; anything the native MLP
    cores cannot honour is an error, not a silent bf16 run."""
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.ops import mlp_engine
    from pytorch_distributed_amd.utils.options import Options
    for prec, cls in (("fp32", f32.DdpgMlpEngineF32), ("bf16", mlp_engine.DdpgMlpEngine)):
        opt = prepare_options(Options(config=7, overrides={"precision": prec, "memory_params.memory_size": 4096}))
        mem = DeviceMemory(opt.memory_params)
        mem.use_pump = False
        mem.attach(torch.device("cuda:0"))
        model, target = make_ddpg(3, 1)
        core = mlp_engine.make_mlp_native_core("ddpg", opt, model, target, mem, None)
        assert isinstance(core.engine, cls), (prec, type(core.engine))
        n = 512
        shape = (n,) + tuple(opt.state_shape)
        mem.feed_batch(torch.randn(shape, device="cuda"), torch.rand(n, 1, device="cuda"), torch.randn(n, 1, device="cuda"),
                       torch.full((n, 1), 0.99, device="cuda"), torch.randn(shape, device="cuda"),
                       torch.zeros(n, 1, device="cuda"))
        before = core.flat_params().clone()
        for _ in range(12):     # eager first step, then graph replays of five updates
            core.step()
        torch.cuda.synchronize()
        assert (core.flat_params() - before).abs().max().item() > 0
        vals = core.loss_values()
        assert all(v == v for v in vals.values()), vals
        mem.close()
    opt = prepare_options(Options(config=7, overrides={"precision": "fp8"}))
    with pytest.raises(ValueError):
        mlp_engine.make_mlp_native_core("ddpg", opt, *make_ddpg(3, 1), DeviceMemory(opt.memory_params), None)


@pytest.mark.parametrize("double", [False, True])
@pytest.mark.parametrize("B,D,A", [(32, 10, 4), (100, 5, 18)])
def test_dqn_mlp_fp32_engine_matches_the_fp64_reference_step(f32, double, B, D, A):
    """MLP-DQN in fp32 (DqnMlpEngineF32) against the reference learner's step (/root/reference/core/single_processes/
    dqn_learner.py:60-96) in fp64, with norm_val != 1 (the 1 / norm_val rides in the first layer's epilogue and in its
    weight gradient)."""
    from pytorch_distributed_amd.models.dqn_mlp_model import DqnMlpModel
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(0)
    opt = Options(config=6)
    model = DqnMlpModel(opt.model_params, 2.0, (1, D), A, 1).cuda()
    target = DqnMlpModel(opt.model_params, 2.0, (1, D), A, 1).cuda()
    with torch.no_grad():
        for p in target.parameters():
            p.add_(0.02 * torch.randn_like(p))
    e = f32.DqnMlpEngineF32(model, target, B, double=double, target_update=10 ** 9)
    e.target.refresh()
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
    e.fused_forward_backward(staged=True)
    e.wgrad_all()
    torch.cuda.synchronize()

    def step(dtype):
        ref, tref = copy.deepcopy(model).to(dtype), copy.deepcopy(target).to(dtype)
        x0, x1, tl = s0.to(dtype) / 2.0, s1.to(dtype) / 2.0, tails.to(dtype)
        a, r, gm, t = tl[:, 0:1], tl[:, 1:2], tl[:, 2:3], tl[:, 3:4]
        with torch.no_grad():
            if double:
                nq = tref.critic(x1).gather(1, ref.critic(x1).argmax(1, keepdim=True))
            else:
                nq = tref.critic(x1).max(1, keepdim=True)[0]
            y = r + gm * nq * (1 - t)
        qall = ref.critic(x0)
        loss = ((qall.gather(1, a.long()) - y) ** 2).mean()
        loss.backward()
        return ref, qall.detach(), loss.item()
    ref64, q64, loss64 = step(torch.float64)
    ref32 = step(torch.float32)[0]
    assert rel_err(e.q.double(), q64) < 1e-5
    assert abs(e.loss_acc.sum().item() - loss64) < 1e-5 * max(1.0, loss64)
    off = 0
    for (name, p64), (_, p32) in zip(ref64.named_parameters(), ref32.named_parameters()):
        got = e.grads[off:off + p64.numel()].view_as(p64)
        off += p64.numel()
        err = ((got.double() - p64.grad).norm() / p64.grad.norm().clamp_min(1e-30)).item()
        err32 = ((p32.grad.double() - p64.grad).norm() / p64.grad.norm().clamp_min(1e-30)).item()
        assert err < 1e-5 and err <= 8.0 * max(err32, 2e-7), "%s: %g (torch fp32: %g)" % (name, err, err32)


def test_dqn_mlp_precision_selects_the_engine(f32):
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.models.dqn_mlp_model import DqnMlpModel
    from pytorch_distributed_amd.ops import mlp_engine
    from pytorch_distributed_amd.utils.options import Options
    for prec, cls in (("fp32", f32.DqnMlpEngineF32), ("bf16", mlp_engine.DqnMlpEngine)):
        opt = prepare_options(Options(config=6, overrides={"precision": prec, "memory_type": "device", "device": "cuda",
                                                           "memory_params.replay_device": "cuda",
                                                           "memory_params.memory_size": 4096}))
        mem = DeviceMemory(opt.memory_params)
        mem.use_pump = False
        mem.attach(torch.device("cuda:0"))
        mk = lambda: DqnMlpModel(opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, 1).cuda()
        core = mlp_engine.make_mlp_native_core("dqn", opt, mk(), mk(), mem, None)
        assert core is not None and isinstance(core.engine, cls), (prec, core and type(core.engine))
        n = 512
        shape = (n,) + tuple(opt.state_shape)
        mem.feed_batch(torch.randn(shape, device="cuda"),
                       torch.randint(0, opt.action_space, (n, 1), device="cuda").float(),
                       torch.randn(n, 1, device="cuda"), torch.full((n, 1), 0.97, device="cuda"),
                       torch.randn(shape, device="cuda"), torch.zeros(n, 1, device="cuda"))
        before = core.flat_params().clone()
        for _ in range(12):
            core.step()
        torch.cuda.synchronize()
        assert (core.flat_params() - before).abs().max().item() > 0
        vals = core.loss_values()
        assert all(v == v for v in vals.values()), vals
        mem.close()
