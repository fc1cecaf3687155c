"""n-step assembler vs a brute-force oracle, using the reference author's debug
trick (dqn_actor.py.bk-per): states carry their step index, rewards equal the
step index, so every emitted transition can be checked by value."""
import numpy as np
import pytest

from pytorch_distributed_amd.agents.nstep import NStepAssembler
from pytorch_distributed_amd.agents.actors import actor_epsilon


def run_episode(nstep, gamma, length, per=False):
    asm = NStepAssembler(nstep, gamma, per)
    asm.reset(0)
    out = []
    for k in range(1, length + 1):
        term = (k == length)
        # q(s_{k-1}) known when acting from s_{k-1}: use q = 100 + (k-1), maxq = 200 + (k-1)
        out += [("push", k) + t for t in asm.push(action=k - 1, reward=float(k), state1=k, terminal1=term,
                                                 qvalue=100.0 + k - 1, max_qvalue=200.0 + k - 1)]
    out += [("flush", length) + t for t in asm.flush(max_qvalue_last=200.0 + length)]
    return out


def oracle(nstep, gamma, length):
    """Expected (s0, a0, R, g, sN, term) list per SURVEY.md 3.2."""
    exp = []
    for k in range(2, length + 1):                 # after the k-th step, states = s_{max(0,k-n-1)}..s_k
        first = max(0, k - (nstep + 1))
        last = k - 1                               # lagged: end state is states[-2]
        horizon = last - first
        R = sum((first + 1 + i) * gamma ** i for i in range(horizon))
        exp.append((first, first, R, gamma ** horizon, last, False))
    # flush
    n_states = min(length + 1, nstep + 2)
    if n_states >= nstep + 2:
        first = length - nstep
    else:
        first = 0
    horizon = length - first
    R = sum((first + 1 + i) * gamma ** i for i in range(horizon))
    exp.append((first, first, R, gamma ** horizon, length, True))
    return exp


@pytest.mark.parametrize("nstep", [1, 3, 5])
@pytest.mark.parametrize("length", [1, 2, 4, 9])
def test_assembler_matches_oracle(nstep, length):
    gamma = 0.9
    got = run_episode(nstep, gamma, length)
    exp = oracle(nstep, gamma, length)
    assert len(got) == len(exp)
    for (_, _, tr, _), e in zip(got, exp):
        s0, a0, R, g, sN, term = tr
        assert (s0, a0, sN, bool(term)) == (e[0], e[1], e[4], e[5])
        assert R[0] == pytest.approx(e[2]) and g[0] == pytest.approx(e[3])


def test_horizon_never_exceeds_nstep_in_steady_state():
    got = run_episode(5, 0.99, 30)
    for tag, k, tr, _ in got:
        s0, _, _, g, sN, _ = tr
        assert sN - s0 <= 5 and g[0] == pytest.approx(0.99 ** (sN - s0))


def test_per_initial_priority():
    gamma, nstep = 0.5, 2
    got = run_episode(nstep, gamma, 6, per=True)
    for tag, k, tr, prio in got:
        s0, a0, R, g, sN, term = tr
        # q0 = Q(s0, a0) = 100 + s0 ; max_q(sN) = 200 + sN
        assert prio == pytest.approx(abs(R[0] + g[0] * (200.0 + sN) - (100.0 + s0)))


def test_windows_do_not_straddle_episodes():
    asm = NStepAssembler(3, 0.9)
    asm.reset("a0")
    asm.push(0, 1.0, "a1", True)
    asm.flush()
    asm.reset("b0")
    assert asm.push(0, 1.0, "b1", False) == []
    (tr, _), = asm.push(0, 1.0, "b2", False)
    assert tr[0] == "b0" and tr[4] == "b1"


def test_epsilon_ladder():
    assert actor_epsilon(1, 1, 0.4, 7) == 0.1
    n = 8
    vals = [actor_epsilon(i, n, 0.4, 7) for i in range(1, n + 1)]
    assert vals[0] == pytest.approx(0.4) and vals[-1] == pytest.approx(0.4 ** 8)
    for i, v in enumerate(vals):
        assert v == pytest.approx(0.4 ** (1 + 7 * i / (n - 1)))


@pytest.mark.parametrize("nstep,per,dtype", [(1, False, "uint8"), (5, True, "uint8"), (3, True, "float32")])
def test_native_feeder_matches_python_assembler(nstep, per, dtype):
    """C n-step assembler (pda_nstep_*) == NStepAssembler + DeviceMemory.feed, record for record."""
    import torch
    from pytorch_distributed_amd.agents.nstep import NativeFeeder, PythonFeeder
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.utils.options import Options
    cfg = 4 if dtype == "uint8" else 7
    opt = prepare_options(Options(config=cfg, overrides={"device": "cpu", "memory_params.memory_size": 64,
                                                         "memory_params.ingest_ring_slots": 256,
                                                         "memory_params.replay_device": "cpu"}))
    mems = [DeviceMemory(opt.memory_params), DeviceMemory(opt.memory_params)]
    gamma = 0.97
    feeders = [PythonFeeder(mems[0], nstep, gamma, per), NativeFeeder(mems[1], nstep, gamma, per)]
    rng = np.random.RandomState(nstep)
    shape = tuple(opt.state_shape)
    A = int(opt.action_shape)

    def state():
        if dtype == "uint8":
            return rng.randint(0, 256, size=shape).astype(np.uint8)
        return rng.randn(*shape).astype(np.float32)

    counts = [0, 0]
    for length in (1, 2, 3, nstep + 1, nstep + 2, 11):
        s = state()
        for f in feeders:
            f.reset(s)
        for t in range(length):
            a = rng.randint(0, 3, size=(1, A)).astype(np.float64)
            r, s1, term = float(rng.randn()), state(), bool(t == length - 1 and length % 2)
            q, mq = float(rng.randn()), float(rng.randn())
            for i, f in enumerate(feeders):
                counts[i] += f.push(a, r, s1, term, q, mq)
        mq_last = float(rng.randn())
        for i, f in enumerate(feeders):
            counts[i] += f.flush(mq_last if per else None)
    assert counts[0] == counts[1] > 0
    c = mems[0].codec
    bufs = []
    for m in mems:
        out = torch.zeros(counts[0] + 4, c.record_bytes, dtype=torch.uint8)
        assert m.ring.pop_many(out, counts[0] + 4) == counts[0]
        bufs.append(out[:counts[0]].numpy())
    a, b = bufs
    assert np.array_equal(a[:, :c.tail_offset], b[:, :c.tail_offset])            # both states, byte for byte
    ta = a[:, c.tail_offset:c.tail_offset + 4 * c.tail_floats].copy().view(np.float32)
    tb = b[:, c.tail_offset:c.tail_offset + 4 * c.tail_floats].copy().view(np.float32)
    np.testing.assert_allclose(ta, tb, rtol=1e-6, atol=1e-6)
    if per:
        assert (ta[:, -1] > 0).any()
