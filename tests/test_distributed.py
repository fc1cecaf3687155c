"""Data-parallel learner path without a GPU: FakeComm (in-process), TorchComm over
gloo with 2 spawned ranks, and a 2-learner job through the launcher.  The same
TorchComm class runs RCCL on the MI355X node (backend "nccl")."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pytorch_distributed_amd.agents.learners import DqnTorchCore, DdpgTorchCore
from pytorch_distributed_amd.launcher import prepare_options, run
from pytorch_distributed_amd.parallel.comm import FakeComm, SingleComm, TorchComm, make_comm
from pytorch_distributed_amd.utils.factory import ModelsDict
from pytorch_distributed_amd.utils.options import Options


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class FixedBatchMemory(object):
    """Replays a fixed list of batches; what a rank's replay shard would hand out."""
    enable_per = False

    def __init__(self, batches):
        self.batches, self.i = batches, 0

    def sample_with_priorities(self, batch_size):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        assert b[0].shape[0] == batch_size
        return b, None, torch.ones(batch_size, 1)

    def update_priorities(self, inds, td):
        pass


def _dqn_batches(n, B, D, A, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        out.append((torch.randn(B, 1, D, generator=g), torch.randint(0, A, (B, 1), generator=g).float(),
                    torch.randn(B, 1, generator=g), torch.full((B, 1), 0.97), torch.randn(B, 1, D, generator=g),
                    (torch.rand(B, 1, generator=g) < 0.2).float()))
    return out


def _same_params(a, b, lr=1e-3, steps=5):
    """Equal up to fp32 summation order.  Adam divides by |g|, so the handful of elements whose
    gradient is ~1e-8 may move differently by up to lr per step; everything else must agree."""
    d = (a - b).abs()
    return d.max().item() <= 2.5 * lr * steps and (d > 2e-6).float().mean().item() < 1e-3


def _split(batch, world):
    parts = [torch.chunk(t, world, dim=0) for t in batch]
    return [tuple(p[r] for p in parts) for r in range(world)]


def _mlp_opt(B):
    opt = Options(config=6, overrides={"device": "cpu", "agent_params.batch_size": B, "agent_params.lr": 1e-3,
                                       "agent_params.target_model_update": 2})
    return prepare_options(opt)


def test_single_comm_is_identity():
    c = SingleComm()
    t = torch.arange(4.0)
    c.all_reduce_mean_(t)
    assert torch.equal(t, torch.arange(4.0)) and c.all_agree(True) and not c.all_agree(False)


def test_fake_comm_collectives():
    comms = FakeComm.create(4)
    ts = [torch.full((3,), float(r)) for r in range(4)]
    for c, t in zip(comms, ts):
        c.all_reduce_mean_(t)
    assert all(torch.allclose(t, torch.full((3,), 1.5)) for t in ts)
    ts = [torch.full((3,), float(r + 1)) for r in range(4)]
    for c, t in zip(comms, ts):
        c.broadcast_(t, src=0)
    assert all(torch.equal(t, torch.ones(3)) for t in ts)


@pytest.mark.parametrize("clip", [float("inf"), 0.02])
def test_dp_two_ranks_equal_one_rank_with_double_batch(clip):
    """Synchronous DP: 2 ranks x B/2 with the gradient mean == 1 rank x B (reference has only Hogwild) -- also with
    a finite value clip that bites (0.02): every back-end clips the rank-AVERAGED gradient, i.e. after the
    all-reduce, so data parallelism stays equivalent to the large batch."""
    B, world, steps = 16, 2, 5
    opt = _mlp_opt(B)
    opt.agent_params.clip_grad = clip
    D, A = int(np.prod(opt.state_shape)), opt.action_space
    full = _dqn_batches(steps, B, D, A, seed=3)
    torch.manual_seed(0)
    proto = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space,
                                       opt.action_shape)
    single = DqnTorchCore(opt, copy.deepcopy(proto), copy.deepcopy(proto), FixedBatchMemory(full), SingleComm())
    opt_half = _mlp_opt(B // world)
    opt_half.agent_params.clip_grad = clip
    comms = FakeComm.create(world)
    ranks = []
    for r in range(world):
        shard = [_split(b, world)[r] for b in full]
        ranks.append(DqnTorchCore(opt_half, copy.deepcopy(proto), copy.deepcopy(proto), FixedBatchMemory(shard),
                                  comms[r]))
    clipped = 0
    # FakeComm finishes a collective when the last rank arrives, so run the ranks' steps in
    # lock-step threads-free fashion: interleave by monkeypatching _finish to defer after all backward passes
    for _ in range(steps):
        single.step()
        pending = []
        for core in ranks:
            orig = core._finish
            core._finish = lambda td, inds, _c=core, **kw: pending.append((_c, td, inds, kw))
            core.step()
            core._finish = orig
        for core, td, inds, kw in pending:
            core.comm.all_reduce_mean_(core.model.flat_grads)
        clipped += int((ranks[0].model.flat_grads.abs() > clip).sum())
        for core, td, inds, kw in pending:
            comm, core.comm = core.comm, SingleComm()
            core._finish(td, inds, **kw)
            core.comm = comm
    assert clip == float("inf") or clipped > 0      # the finite clip really bit
    ref = single.model.flat_params
    for core in ranks:
        assert _same_params(core.model.flat_params, ref), (core.model.flat_params - ref).abs().max()
        assert _same_params(core.target_model.flat_params, single.target_model.flat_params)


def _gloo_worker(rank, world, port, out_q):
    comm = TorchComm(rank, world, torch.device("cpu"), init_method="tcp://127.0.0.1:%d" % port)
    t = torch.full((1000,), float(rank + 1))
    comm.all_reduce_mean_(t)
    ok_mean = bool(torch.allclose(t, torch.full((1000,), (world + 1) / 2.0)))
    s = torch.arange(10.0) * (rank + 1)
    comm.all_reduce_sum_(s)
    ok_sum = bool(torch.allclose(s, torch.arange(10.0) * sum(range(1, world + 1))))
    w = torch.full((64,), float(rank * 7 + 3))
    comm.broadcast_(w, src=0)
    ok_bc = bool(torch.equal(w, torch.full((64,), 3.0)))
    agree_all = comm.all_agree(True)
    agree_one = comm.all_agree(rank == 0)
    comm.barrier()
    # a DP learner step through the real process group: 2 ranks x 8 == 1 rank x 16
    B = 16
    opt = _mlp_opt(B // world)
    D, A = int(np.prod(opt.state_shape)), opt.action_space
    full = _dqn_batches(3, B, D, A, seed=11)
    torch.manual_seed(0)
    proto = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space,
                                       opt.action_shape)
    core = DqnTorchCore(opt, copy.deepcopy(proto), copy.deepcopy(proto),
                        FixedBatchMemory([_split(b, world)[rank] for b in full]), comm)
    for _ in range(3):
        core.step()
    flat = core.model.flat_params.clone()
    ref_core = DqnTorchCore(_mlp_opt(B), copy.deepcopy(proto), copy.deepcopy(proto), FixedBatchMemory(full),
                            SingleComm())
    for _ in range(3):
        ref_core.step()
    ok_dp = _same_params(flat, ref_core.model.flat_params)
    out_q.put((rank, ok_mean, ok_sum, ok_bc, agree_all, agree_one, ok_dp))
    comm.close()


def test_torch_comm_gloo_two_ranks():
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_mean, ok_sum, ok_bc, agree_all, agree_one, ok_dp in results:
        assert ok_mean and ok_sum and ok_bc and ok_dp, (rank, ok_mean, ok_sum, ok_bc, ok_dp)
        assert agree_all is True and agree_one is False


def test_make_comm_world_one_is_single():
    assert isinstance(make_comm(0, 1, torch.device("cpu")), SingleComm)


def test_two_learner_job_through_launcher(tmp_root):
    """num_learners=2: two learner processes form a gloo group, rank 0 publishes."""
    from tests.test_integration import FAST
    o = dict(FAST)
    o.update({"root_dir": tmp_root, "timestamp": "dp2", "num_learners": 2, "num_actors": 2,
              "dist_port": free_port(), "agent_params.steps": 25, "agent_params.learn_start": 60})
    opt = Options(config=6, overrides=o)
    sup = run(opt)
    assert all(c == 0 for c in sup.exitcodes), sup.exitcodes
    assert os.path.exists(opt.model_name)
    # the job itself proved that two distinct learner ranks met in one group, and that they ended in step
    import json
    facts = json.load(open(os.path.join(opt.log_dir, "learner_ranks.json")))
    assert facts["rccl_ranks"] == 2 and facts["ranks_distinct"] and facts["rank_id_min_max_sum"] == [0, 1, 1]
    assert len(set(r["pid"] for r in facts["rank_devices"])) == 2
    agree = json.load(open(os.path.join(opt.log_dir, "learner_rank_agreement.json")))
    assert agree["steps"] == 25 and agree["param_max_abs_diff_summed_over_ranks"] == 0.0


def test_ddpg_dp_equivalence_fake_comm():
    B, world = 8, 2
    opt = prepare_options(Options(config=3, overrides={"device": "cpu", "agent_params.batch_size": B,
                                                       "agent_params.lr": 1e-3}))
    D, A = int(np.prod(opt.state_shape)), int(opt.action_shape)
    g = torch.Generator().manual_seed(5)
    full = [(torch.randn(B, 1, D, generator=g), torch.rand(B, A, generator=g) * 2 - 1, torch.randn(B, 1, generator=g),
             torch.full((B, 1), 0.95), torch.randn(B, 1, D, generator=g), torch.zeros(B, 1))]
    torch.manual_seed(1)
    proto = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space,
                                       opt.action_shape)
    single = DdpgTorchCore(opt, copy.deepcopy(proto), copy.deepcopy(proto), FixedBatchMemory(full), SingleComm())
    single.step()
    opt_h = prepare_options(Options(config=3, overrides={"device": "cpu", "agent_params.batch_size": B // world,
                                                         "agent_params.lr": 1e-3}))
    comms = FakeComm.create(world)
    cores, pending = [], []
    for r in range(world):
        core = DdpgTorchCore(opt_h, copy.deepcopy(proto), copy.deepcopy(proto),
                             FixedBatchMemory([_split(full[0], world)[r]]), comms[r])
        core._finish_real = core._finish
        core._finish = lambda td, inds, _c=core, **kw: pending.append((_c, td, inds, kw))
        core.step()
        cores.append(core)
    for core, td, inds, kw in pending:
        core.comm.all_reduce_mean_(core.model.flat_grads)
    for core, td, inds, kw in pending:
        core.comm = SingleComm()
        core._finish_real(td, inds, **kw)
    # value-clip happens per rank before the mean in DP, after it on one rank: identical while no
    # gradient element reaches the clip (40), which holds for this data
    for core in cores:
        assert _same_params(core.model.flat_params, single.model.flat_params)


def test_ddpg_actor_warmup_trains_only_the_critic():
    """agent_params.actor_warmup_steps: during the warm-up the actor loss is not applied (the actor keeps its
    initial weights, the critic trains); afterwards both move."""
    B = 8
    opt = prepare_options(Options(config=3, overrides={"device": "cpu", "agent_params.batch_size": B,
                                                       "agent_params.lr": 1e-3, "agent_params.actor_warmup_steps": 3,
                                                       "agent_params.compat_ddpg_grad_leak": False}))
    D, A = int(np.prod(opt.state_shape)), int(opt.action_shape)
    g = torch.Generator().manual_seed(5)
    full = [(torch.randn(B, 1, D, generator=g), torch.rand(B, A, generator=g) * 2 - 1, torch.randn(B, 1, generator=g),
             torch.full((B, 1), 0.95), torch.randn(B, 1, D, generator=g), torch.zeros(B, 1))]
    torch.manual_seed(1)
    proto = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space,
                                       opt.action_shape)
    core = DdpgTorchCore(opt, copy.deepcopy(proto), copy.deepcopy(proto), FixedBatchMemory(full), SingleComm())
    actor0 = torch.cat([p.detach().reshape(-1).clone() for p in core.model.actor.parameters()])
    critic0 = torch.cat([p.detach().reshape(-1).clone() for p in core.model.critic.parameters()])
    for _ in range(2):
        core.step()
    actor1 = torch.cat([p.detach().reshape(-1) for p in core.model.actor.parameters()])
    critic1 = torch.cat([p.detach().reshape(-1) for p in core.model.critic.parameters()])
    assert torch.equal(actor0, actor1)
    assert (critic0 - critic1).abs().max().item() > 0
    for _ in range(3):
        core.step()
    actor2 = torch.cat([p.detach().reshape(-1) for p in core.model.actor.parameters()])
    assert (actor0 - actor2).abs().max().item() > 0


def test_ddpg_actor_preact_penalty_pulls_a_saturated_actor_back():
    """agent_params.actor_preact_l2: with the actor's output layer blown up so that tanh saturates (|a| = 1 in fp32,
    d(-Q)/dz = 0), the plain actor loss leaves the actor where it is; with the penalty it moves, towards smaller |z|."""
    B = 8
    moved = {}
    for lam in (0.0, 0.01):
        opt = prepare_options(Options(config=3, overrides={"device": "cpu", "agent_params.batch_size": B,
                                                           "agent_params.lr": 1e-3, "agent_params.actor_preact_l2": lam,
                                                           "agent_params.compat_ddpg_grad_leak": False}))
        D, A = int(np.prod(opt.state_shape)), int(opt.action_shape)
        g = torch.Generator().manual_seed(5)
        full = [(torch.randn(B, 1, D, generator=g), torch.rand(B, A, generator=g) * 2 - 1,
                 torch.randn(B, 1, generator=g), torch.full((B, 1), 0.95), torch.randn(B, 1, D, generator=g),
                 torch.zeros(B, 1))]
        torch.manual_seed(1)
        proto = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space,
                                           opt.action_shape)
        with torch.no_grad():
            proto.actor[4].bias.fill_(40.0)             # z ~ 40: tanh(z) == 1.0 exactly in fp32
        core = DdpgTorchCore(opt, copy.deepcopy(proto), copy.deepcopy(proto), FixedBatchMemory(full), SingleComm())
        z0 = core.model.actor[:-1](full[0][0].flatten(1)).detach()
        assert torch.tanh(z0).abs().min().item() == 1.0
        a0 = torch.cat([p.detach().reshape(-1).clone() for p in core.model.actor.parameters()])
        for _ in range(3):
            core.step()
        a1 = torch.cat([p.detach().reshape(-1) for p in core.model.actor.parameters()])
        z1 = core.model.actor[:-1](full[0][0].flatten(1)).detach()
        moved[lam] = ((a1 - a0).abs().max().item(), (z0.abs().mean() - z1.abs().mean()).item())
    assert moved[0.0][0] == 0.0                           # stuck for good without it
    assert moved[0.01][0] > 0 and moved[0.01][1] > 0      # |z| shrinks with it


def _weight_sync_worker(rank, port, out_q):
    """2 learner ranks + 1 inference rank in one gloo world: sub-group collectives + weight broadcast."""
    from pytorch_distributed_amd.parallel.weight_sync import WeightPublisher, WeightReceiver
    learners = 2
    comm = make_comm(rank, learners, torch.device("cpu"), port=port, extra_ranks=1)
    n = 1000
    ok = True
    if rank < learners:
        t = torch.full((8,), float(rank + 1))
        comm.all_reduce_sum_(t)                 # only the two learners take part
        ok = ok and bool(torch.equal(t, torch.full((8,), 3.0))) and comm.all_agree(True)
        if rank == 0:
            pub = WeightPublisher(comm, n, torch.device("cpu"))
            for k in range(1, 6):
                pub.publish(torch.full((n,), float(k)))
            pub.close()
        out_q.put((rank, ok, 0))
    else:
        recv = WeightReceiver(comm, n, torch.device("cpu"))
        flat = torch.zeros(n)
        seen = []
        import time
        deadline = time.time() + 60
        while not recv.closed and time.time() < deadline:
            if recv.poll_into(flat):
                seen.append(float(flat[0]))
            time.sleep(0.001)
        if recv.poll_into(flat):
            seen.append(float(flat[0]))
        recv.join(10)
        ok = recv.closed and recv.version == 5 and seen[-1] == 5.0 and seen == sorted(seen) \
            and bool(torch.equal(flat, torch.full((n,), 5.0)))
        out_q.put((rank, ok, recv.version))
    comm.close()


def test_weight_broadcast_to_inference_rank_gloo():
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_weight_sync_worker, args=(r, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results


def test_collective_weight_sync_job_through_launcher(tmp_root):
    """Batched inference server as rank L of the learners' group: weights arrive by broadcast (gloo here,
    RCCL when the server owns a GPU), actions still flow through the mailbox."""
    from tests.test_integration import FAST
    o = dict(FAST)
    o.update({"root_dir": tmp_root, "timestamp": "wsync", "num_actors": 2, "actor_inference": "batched",
              "weight_sync": "collective", "dist_port": free_port(),
              "agent_params.steps": 40, "agent_params.learn_start": 60})
    opt = Options(config=6, overrides=o)
    sup = run(opt)
    assert all(c == 0 for c in sup.exitcodes), sup.exitcodes
    assert os.path.exists(opt.model_name)
