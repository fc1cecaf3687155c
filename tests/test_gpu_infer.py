"""Native batched-inference loop (ops/csrc/infer_server.hip through ops/infer.py) on a real
MI355X: mailbox protocol + graph-captured policy forward vs the fp32 PyTorch model."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    return ops


def _cnn_setup(slots, A=6, seed=0):
    from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
    from pytorch_distributed_amd.ops.infer import make_native_server
    from pytorch_distributed_amd.parallel.inference_server import InferenceMailbox
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(seed)
    opt = Options(config=2)
    model = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda().eval()
    mailbox = InferenceMailbox(slots, (4, 84, 84), torch.uint8, 1, discrete=True)
    server = make_native_server(opt, model, mailbox, torch.device("cuda:0"))
    return opt, model, mailbox, server


def _post(mailbox, slot, obs, k):
    from pytorch_distributed_amd.parallel import hostlib
    mailbox.obs[slot] = torch.as_tensor(obs)
    hostlib.lib().pda_atomic_store(mailbox._req_ptr(slot), k)


def test_serve_once_matches_model_greedy(native):
    _, model, mailbox, server = _cnn_setup(24)
    rng = np.random.RandomState(0)
    slots = [0, 1, 2, 7, 13, 14, 23]            # two merged runs and two singletons
    obs = {s: rng.randint(0, 256, size=(4, 84, 84)).astype(np.uint8) for s in slots}
    for s in slots:
        _post(mailbox, s, obs[s], 1)
    assert server.serve_once() == len(slots)
    assert server.serve_once() == 0             # everything answered
    with torch.no_grad():
        q = model(torch.as_tensor(np.stack([obs[s] for s in slots])).cuda()).cpu()
    out = mailbox.out.numpy()
    for i, s in enumerate(slots):
        top2 = q[i].topk(2).values
        if float(top2[0] - top2[1]) > 0.02:     # bf16 forward: only check clear winners
            assert int(out[s, 0]) == int(q[i].argmax())
        assert abs(out[s, 2] - float(q[i].max())) < 0.03 * max(1.0, float(q[i].abs().max()))
        assert abs(out[s, 1] - float(q[i][int(out[s, 0])])) < 0.03 * max(1.0, float(q[i].abs().max()))
        assert int(mailbox.seq[s, 8]) == 1      # resp == req
    # untouched slots stay unanswered
    assert int(mailbox.seq[5, 8]) == 0
    # epsilon = 1: actions become uniform over A
    mailbox.eps[:] = 1.0
    counts = np.zeros(6)
    for k in range(2, 42):
        for s in range(24):
            _post(mailbox, s, obs[slots[0]], k)
        assert server.serve_once() == 24
        for s in range(24):
            counts[int(mailbox.out[s, 0])] += 1
    assert counts.min() > 0.5 * counts.sum() / 6
    st = server.stats()
    assert st["requests"] == len(slots) + 40 * 24 and st["max_batch"] == 24
    server.close()


def test_loop_thread_serves_concurrent_actors(native):
    _, model, mailbox, server = _cnn_setup(8, seed=1)
    server.start()
    rng = np.random.RandomState(1)
    frames = rng.randint(0, 256, size=(8, 5, 4, 84, 84)).astype(np.uint8)
    results = {}

    def actor(slot):
        res = []
        for t in range(40):
            a, q, mq = mailbox.request(slot, frames[slot, t % 5], timeout_s=20.0)
            res.append((int(a[0][0]), q, mq))
        results[slot] = res

    threads = [threading.Thread(target=actor, args=(s,)) for s in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(60)
    server.stop()
    assert sorted(results) == list(range(8))
    with torch.no_grad():
        q = model(torch.as_tensor(frames.reshape(40, 4, 84, 84)).cuda()).cpu().reshape(8, 5, -1)
    for s in range(8):
        for t, (a, qa, mq) in enumerate(results[s]):
            ref = q[s, t % 5]
            assert abs(mq - float(ref.max())) < 0.03 * max(1.0, float(ref.abs().max()))
            assert abs(qa - mq) < 1e-6          # eps = 0: greedy, q[a] == max q
    st = server.stats()
    assert st["requests"] == 8 * 40
    server.close()


def test_refresh_from_global_model(native):
    from pytorch_distributed_amd.utils.param_store import publish_params, share_model
    import copy
    opt, model, mailbox, server = _cnn_setup(8, seed=2)
    global_model = share_model(copy.deepcopy(model).cpu())
    obs = np.random.RandomState(2).randint(0, 256, size=(4, 84, 84)).astype(np.uint8)
    _post(mailbox, 0, obs, 1)
    server.serve_once()
    before = float(mailbox.out[0, 2])
    with torch.no_grad():
        new = global_model.flat_params.clone()
        new[-6:] += 5.0                         # fc2 bias: every q moves by +5
    publish_params(global_model, new)
    server.refresh_from(global_model)
    _post(mailbox, 0, obs, 2)
    server.serve_once()
    assert abs(float(mailbox.out[0, 2]) - (before + 5.0)) < 0.05
    server.close()


def test_ddpg_actor_program(native):
    from pytorch_distributed_amd.models.ddpg_mlp_model import DdpgMlpModel
    from pytorch_distributed_amd.ops.infer import make_native_server
    from pytorch_distributed_amd.parallel.inference_server import InferenceMailbox
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(3)
    opt = Options(config=7)
    D, A = 17, 6
    model = DdpgMlpModel(opt.model_params, 1.0, (D,), A, A).cuda().eval()
    with torch.no_grad():       # the reference init makes the last layer ~0: spread the outputs
        model.actor[4].weight.mul_(200.0)
    mailbox = InferenceMailbox(12, (D,), torch.float32, A, discrete=False)
    server = make_native_server(opt, model, mailbox, torch.device("cuda:0"))
    rng = np.random.RandomState(3)
    obs = rng.randn(12, D).astype(np.float32)
    for s in (0, 3, 4, 11):
        _post(mailbox, s, obs[s], 1)
    assert server.serve_once() == 4
    with torch.no_grad():
        ref = model.forward_actor(torch.as_tensor(obs).cuda()).cpu().numpy()
    for s in (0, 3, 4, 11):
        np.testing.assert_allclose(mailbox.out[s, :A].numpy(), ref[s], atol=0.03)
        assert float(mailbox.out[s, A]) == 0.0 and float(mailbox.out[s, A + 1]) == 0.0
    server.close()


def test_pumped_ingestion_into_hbm_replay(native):
    """actors' feed() -> shm ring -> native pump thread -> pinned staging -> async H2D -> unpack kernel."""
    import time
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.utils.options import Options
    opt = prepare_options(Options(config=4, overrides={"memory_params.memory_size": 1024,
                                                       "memory_params.ingest_chunk": 64,
                                                       "memory_params.ingest_ring_slots": 256}))
    mem = DeviceMemory(opt.memory_params)
    rng = np.random.RandomState(5)
    n = 300
    s0 = rng.randint(0, 256, size=(n, 4, 84, 84)).astype(np.uint8)
    s1 = rng.randint(0, 256, size=(n, 4, 84, 84)).astype(np.uint8)
    mem.attach(torch.device("cuda:0"))
    assert mem.pump is not None
    fed = 0
    moved = 0
    deadline = time.time() + 30
    while moved < n and time.time() < deadline:
        while fed < n and fed - moved < 200:
            mem.feed((s0[fed], [fed % 6], [float(fed)], [0.95], s1[fed], float(fed % 7 == 0)), priority=1.0 + fed)
            fed += 1
        moved += mem.drain()
    assert moved == n and mem.size == n
    torch.cuda.synchronize()
    assert torch.equal(mem.state0s[:n].cpu(), torch.as_tensor(s0))
    assert torch.equal(mem.state1s[:n].cpu(), torch.as_tensor(s1))
    assert torch.equal(mem.rewards[:n, 0].cpu(), torch.arange(n, dtype=torch.float32))
    assert torch.equal(mem.actions[:n, 0].cpu(), (torch.arange(n) % 6).float())
    leaves = mem.tree.get(torch.arange(n, device="cuda")).cpu()
    want = (torch.arange(n, dtype=torch.float32) + 1.0 + mem.priority_eps) ** mem.priority_exponent
    assert torch.allclose(leaves, want, rtol=1e-4)
    assert abs(float(mem.tree.total()) - float(want.sum())) < 1e-2 * float(want.sum())
    mem.close()
