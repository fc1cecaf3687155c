"""Replay memories on the host: ring wrap, uniform / prioritized sampling,
the shm ingestion ring, and the device-replay code path with replay_device=cpu."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pytorch_distributed_amd.utils.options import Options
from pytorch_distributed_amd.launcher import prepare_options
from pytorch_distributed_amd.memories.shared_memory import SharedMemory
from pytorch_distributed_amd.memories.device_memory import DeviceMemory
from pytorch_distributed_amd.parallel.ingest import ShmRing


def make_opts(config=6, **ov):
    o = Options(config=config, overrides=ov)
    return prepare_options(o)


def tr(i, shape, dtype=np.float32):
    return (np.full(shape, i, dtype), [i % 3], [float(i)], [0.5], np.full(shape, i + 1, dtype), bool(i % 2))


def test_shared_memory_wraps_and_samples():
    o = make_opts(**{"memory_params.memory_size": 8})
    mem = SharedMemory(o.memory_params)
    assert mem.size == 0
    for i in range(11):
        mem.feed(tr(i, o.state_shape))
    assert mem.size == 8 and mem.pos.value == 3 and mem.full.value
    s0, a, r, g, s1, t = mem.sample(16)
    assert s0.shape == (16, 1, 1, 5) and a.shape == (16, 1) and r.shape == (16, 1)
    assert torch.equal(s1[:, 0, 0, 0], s0[:, 0, 0, 0] + 1)
    assert set(r[:, 0].tolist()) <= set(float(i) for i in range(3, 11))
    assert torch.equal(t[:, 0], (r[:, 0] % 2))


def test_shared_memory_uint8_states():
    o = make_opts(config=1, **{"memory_params.memory_size": 4})
    mem = SharedMemory(o.memory_params)
    mem.feed(tr(7, o.state_shape, np.uint8))
    s0, *_ = mem.sample(2)
    assert s0.dtype == torch.uint8 and s0.shape == (2, 4, 84, 84) and int(s0[0, 0, 0, 0]) == 7


def test_shared_memory_per():
    o = make_opts(**{"memory_params.memory_size": 64, "memory_params.enable_per": True})
    mem = SharedMemory(o.memory_params)
    for i in range(64):
        mem.feed(tr(i, o.state_shape), priority=100.0 if i == 5 else 0.01)
    batch, inds, w = mem.sample_with_priorities(256)
    frac = float((np.asarray(inds) == 5).mean())
    assert frac > 0.4                          # sqrt(100)/(sqrt(100)+63*sqrt(.01)) ~ 0.61
    assert w.shape == (256, 1) and float(w.max()) <= 1.0 + 1e-6
    assert float(w[np.asarray(inds) == 5].max()) < 0.2
    mem.update_priorities(np.array([5]), torch.tensor([0.01]))
    _, inds, _ = mem.sample_with_priorities(256)
    assert float((np.asarray(inds) == 5).mean()) < 0.1


def _producer(ring, pid, n):
    for i in range(n):
        rec = np.full(24, pid, dtype=np.uint8)
        rec[:8] = np.frombuffer(np.int64(i).tobytes(), dtype=np.uint8)
        assert ring.push_segments([rec[:8], rec[8:]])


def test_ring_many_producers_no_loss_no_dup():
    ring = ShmRing(64, 24)      # small ring: forces wrap + back-pressure
    ctx = mp.get_context("spawn")
    nprod, per = 4, 500
    procs = [ctx.Process(target=_producer, args=(ring, p, per)) for p in range(nprod)]
    for p in procs:
        p.start()
    seen = {p: [] for p in range(nprod)}
    buf = np.zeros((16, 24), dtype=np.uint8)
    total = 0
    import time
    deadline = time.time() + 60
    while total < nprod * per and time.time() < deadline:
        n = ring.pop_many(buf, 16)
        for row in buf[:n]:
            seen[int(row[8])].append(int(np.frombuffer(row[:8].tobytes(), dtype=np.int64)[0]))
        total += n
    for p in procs:
        p.join(10)
    assert total == nprod * per
    for p in range(nprod):
        assert seen[p] == list(range(per))     # per-producer FIFO, nothing lost or duplicated
    st = ring.stats()
    assert st["pushed"] == st["popped"] == nprod * per and st["backlog"] == 0


@pytest.mark.parametrize("per", [False, True])
def test_device_memory_cpu_path(per):
    o = make_opts(config=1, **{"memory_params.memory_size": 32, "memory_params.enable_per": per,
                               "memory_params.ingest_ring_slots": 16, "memory_params.ingest_chunk": 8,
                               "memory_params.replay_device": "cpu"})
    mem = DeviceMemory(o.memory_params)
    for i in range(10):
        mem.feed(tr(i, o.state_shape, np.uint8), priority=float(i + 1))
    assert mem.size == 0                     # nothing resident until a learner drains
    mem.attach("cpu")
    assert mem.drain() == 10 and mem.size == 10
    assert int(mem.state0s[3, 0, 0, 0]) == 3 and int(mem.state1s[3, 2, 5, 5]) == 4
    assert float(mem.rewards[7, 0]) == 7.0 and float(mem.gamma1s[7, 0]) == 0.5 and float(mem.actions[7, 0]) == 1.0
    assert float(mem.terminal1s[7, 0]) == 1.0 and float(mem.init_priorities[7]) == 8.0
    (s0, a, r, g, s1, t), idx, w = mem.sample_with_priorities(64)
    assert s0.dtype == torch.uint8 and s0.shape == (64, 4, 84, 84)
    assert int(idx.max()) < 10
    assert torch.equal(s0[:, 0, 0, 0].float(), r[:, 0])
    if per:
        assert float(mem.tree.total()) == pytest.approx(sum((i + 1 + 1e-6) ** 0.5 for i in range(10)), rel=1e-5)
        mem.update_priorities(idx[:1], torch.tensor([1e4]))
        _, idx2, w2 = mem.sample_with_priorities(64)
        assert float((idx2 == idx[0]).float().mean()) > 0.5 and float(w2.min()) < 0.5
    for i in range(10, 50):                  # wrap the 32-slot ring
        mem.feed(tr(i, o.state_shape, np.uint8), priority=1.0)
        if i % 8 == 0:
            mem.drain()
    mem.drain()
    assert mem.size == 32 and mem.full and mem.pos == 50 % 32
    assert int(mem.state0s[49 % 32, 0, 0, 0]) == 49


def test_device_memory_feed_batch_float_states():
    o = make_opts(config=3, **{"memory_params.memory_size": 16, "memory_params.replay_device": "cpu"})
    o.memory_type = "device"
    mem = DeviceMemory(o.memory_params).attach("cpu")
    n = 20
    s = torch.arange(n).float().reshape(n, 1, 1, 1).expand(n, 1, 1, 3)
    mem.feed_batch(s, torch.zeros(n, 1), torch.arange(n).float(), torch.full((n,), 0.9), s + 1, torch.zeros(n))
    assert mem.size == 16 and mem.pos == 4
    s0, a, r, g, s1, t = mem.sample(8)
    assert torch.equal(s0[:, 0, 0, 0], r[:, 0]) and torch.equal(s1, s0 + 1)
