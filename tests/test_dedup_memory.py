"""Frame-deduplicated replay (memories/dedup_memory.py, parallel/csrc/frame_runs.h) on the CPU: the native run packer
against the Python n-step assembler (the reference's bookkeeping, core/single_processes/dqn_actor.py:55-61,110-163),
the sampling window when planes run out before transition rows do, and prioritized sampling over it."""
import numpy as np
import pytest
import torch

from pytorch_distributed_amd.agents.nstep import NStepAssembler, make_feeder
from pytorch_distributed_amd.launcher import prepare_options
from pytorch_distributed_amd.utils.factory import MemoriesDict
from pytorch_distributed_amd.utils.options import Options


def make_memory(memory_type="device-dedup", size=512, **mp):
    ov = {"device": "cpu", "memory_type": memory_type, "memory_params.memory_size": size}
    ov.update({"memory_params." + k: v for k, v in mp.items()})
    opt = prepare_options(Options(config=2, overrides=ov))
    return opt, MemoriesDict[opt.memory_type](opt.memory_params)


def sliding_state(rng, prev):
    plane = rng.randint(0, 256, (1, 84, 84)).astype(np.uint8)
    return np.repeat(plane, 4, 0) if prev is None else np.concatenate([prev[1:], plane])


def play(feeder, rng, episodes, max_len, oracle=None, slide=True, per=False):
    out = []
    for _ in range(episodes):
        s = sliding_state(rng, None)
        feeder.reset(s)
        if oracle is not None:
            oracle.reset(s)
        steps = int(rng.randint(1, max_len))
        for t in range(steps):
            s = sliding_state(rng, s if slide else None)
            a, r, term = [int(rng.randint(6))], float(rng.randn()), t == steps - 1
            q, mq = (float(rng.randn()), float(rng.randn())) if per else (None, None)
            feeder.push(a, r, s, term, q, mq)
            if oracle is not None:
                out += oracle.push(a, r, s, term, q, mq)
        mq = float(rng.randn()) if per else None
        feeder.flush(mq)
        if oracle is not None:
            out += oracle.flush(mq)
    feeder.ship()
    return out


def drain_all(memory):
    while memory.drain():
        pass


def check_rows(memory, rows, expected):
    s0, a, r, g, s1, t = memory._gather(torch.tensor(rows, dtype=torch.int32))
    for i, (tr, _) in enumerate(expected):
        assert np.array_equal(s0[i].numpy(), tr[0]) and np.array_equal(s1[i].numpy(), tr[4]), i
        assert float(a[i]) == tr[1][0] and abs(float(r[i]) - tr[2][0]) < 1e-5
        assert abs(float(g[i]) - tr[3][0]) < 1e-6 and float(t[i]) == float(tr[5])


@pytest.mark.parametrize("nstep", [1, 5])
def test_runs_hold_what_the_python_assembler_emits(nstep):
    _, m = make_memory()
    rng = np.random.RandomState(nstep)
    f = make_feeder(m, nstep, 0.99)
    assert f.dedup
    expected = play(f, rng, 14, 60, NStepAssembler(nstep, 0.99))
    st = f.run_stats()
    assert st["transitions"] == len(expected) > 100 and st["runs"] >= 3
    m.attach("cpu")
    drain_all(m)
    assert m.trans_linear == len(expected) and m.window() == (len(expected), 0, 512)
    check_rows(m, list(range(len(expected))), expected)
    # one plane per step plus the episode starts and run restarts: far below the 8 planes of two full stacks
    assert m.stats()["planes_per_transition"] < 1.6


def test_states_that_do_not_slide_are_sent_in_full():
    _, m = make_memory()
    rng = np.random.RandomState(3)
    f = make_feeder(m, 3, 0.99)
    expected = play(f, rng, 3, 12, NStepAssembler(3, 0.99), slide=False)
    m.attach("cpu")
    drain_all(m)
    check_rows(m, list(range(len(expected))), expected)
    assert m.stats()["planes_per_transition"] > 3.5


def test_single_feed_goes_in_as_a_run_of_its_own():
    _, m = make_memory()
    rng = np.random.RandomState(5)
    trs = []
    for _ in range(5):
        s0, s1 = rng.randint(0, 256, (2, 4, 84, 84)).astype(np.uint8)
        tr = (s0, [2], [0.5], [0.9], s1, 0.0)
        m.feed(tr, 0.3)
        trs.append((tr, 0.3))
    m.attach("cpu")
    drain_all(m)
    check_rows(m, list(range(5)), trs)


def test_window_shrinks_when_planes_run_out_first_and_never_shows_overwritten_frames():
    # short episodes (4 planes each at their start) on a plane ring sized for ~1 plane per row
    _, m = make_memory(size=256, dedup_plane_slack=0.5, dedup_plane_cap=32, dedup_chunk_runs=2)
    rng = np.random.RandomState(7)
    f = make_feeder(m, 2, 0.99)
    m.attach("cpu")
    oracle = NStepAssembler(2, 0.99)
    expected = []
    for _ in range(30):
        expected += play(f, rng, 4, 8, oracle)
        drain_all(m)
        rows, first, ring = m.window()
        assert 0 < rows <= 256 and ring == 256
        lin = list(range(m.trans_linear - rows, m.trans_linear))
        check_rows(m, [j % 256 for j in lin], [expected[j] for j in lin])
    assert m.trans_linear == len(expected) > 256
    assert m.window()[0] < 256              # the plane ring, not the row ring, bounds what can be sampled
    idx, w = m.sample_indices(4096)
    rows, first, ring = m.window()
    assert bool((((idx.long() - first) % ring) < rows).all())


def test_prioritized_rows_outside_the_window_have_no_mass():
    _, m = make_memory("device-dedup-per", size=256, dedup_plane_slack=0.5, dedup_plane_cap=32, dedup_chunk_runs=2)
    rng = np.random.RandomState(11)
    f = make_feeder(m, 2, 0.99, enable_per=True)
    m.attach("cpu")
    for _ in range(40):
        play(f, rng, 4, 8, per=True)
        drain_all(m)
    rows, first, ring = m.window()
    assert rows < 256 and m.dropped_early > 0
    inside = ((torch.arange(256) - first) % ring) < rows
    mass = m.tree.get(torch.arange(256))
    assert bool((mass[~inside] == 0).all()) and bool((mass[inside] > 0).all())
    assert torch.isfinite(m.tree.min_value()) and float(m.tree.min_value()) > 0
    (s0, a, r, g, s1, t), idx, w = m.sample_with_priorities(512)
    assert bool(inside[idx.long()].all()) and bool(torch.isfinite(w).all())


def test_float_states_are_refused():
    with pytest.raises(ValueError):
        opt = prepare_options(Options(config=7, overrides={"device": "cpu", "memory_type": "device-dedup"}))
        MemoriesDict[opt.memory_type](opt.memory_params)
