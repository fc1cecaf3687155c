"""HIP radix-64 sum/min tree (ops/csrc/replay_kernels.hip through ops/segtree.py) against a NumPy
oracle: property-based sequences of batched priority updates (with duplicate indices) followed by
stratified prefix-sum sampling and importance weights (kernels K7/K8)."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st, HealthCheck

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree_cls():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops.segtree import DeviceSumMinTree
    return DeviceSumMinTree


def check_samples(p, uniforms, idx, w, beta):
    """Every drawn leaf must own the mass point it was drawn for (up to fp32 prefix-sum rounding)."""
    B = len(uniforms)
    cum = np.cumsum(p.astype(np.float64))
    total = cum[-1]
    v = (np.arange(B) + uniforms.astype(np.float64)) / B * total
    tol = 1e-5 * total + 1e-6
    lo = np.where(idx > 0, cum[np.maximum(idx - 1, 0)], 0.0)
    hi = cum[idx]
    assert np.all(p[idx] > 0), "sampled a zero-priority leaf"
    assert np.all(v >= lo - tol) and np.all(v <= hi + tol), (v, lo, hi)
    pmin = p[p > 0].min()
    np.testing.assert_allclose(w, (pmin / p[idx]) ** beta, rtol=2e-3)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.sampled_from([1, 7, 64, 65, 1000, 4097, 70000]), seed=st.integers(0, 2 ** 16),
       rounds=st.integers(1, 4), batch=st.sampled_from([1, 32, 128, 1024]), beta=st.sampled_from([0.0, 0.4, 1.0]))
def test_update_then_sample_matches_oracle(tree_cls, n, seed, rounds, batch, beta):
    rng = np.random.RandomState(seed)
    tree = tree_cls(n, torch.device("cuda"))
    p = np.zeros(n, dtype=np.float32)
    for _ in range(rounds):
        k = min(n, int(rng.randint(1, 2 * batch + 1)))
        idx = rng.randint(0, n, size=k)                        # duplicates on purpose
        val = (rng.rand(k).astype(np.float32) + 1e-3) ** 2
        # duplicates: the kernel's last writer is unspecified, so make equal indices carry equal values
        first = {}
        for j, i in enumerate(idx):
            val[j] = first.setdefault(int(i), val[j])
        p[idx] = val
        tree.update(torch.as_tensor(idx, dtype=torch.int32, device="cuda"), torch.as_tensor(val, device="cuda"))
    torch.cuda.synchronize()
    assert abs(float(tree.total()) - float(p.astype(np.float64).sum())) < 1e-4 * max(1.0, float(p.sum()))
    assert float(tree.min_value()) == pytest.approx(float(p[p > 0].min()), rel=1e-6)
    u = rng.rand(batch).astype(np.float32)
    idx_out, w_out = tree.sample(torch.as_tensor(u, device="cuda"), beta)
    torch.cuda.synchronize()
    check_samples(p, u, idx_out.cpu().numpy().astype(np.int64), w_out.cpu().numpy(), beta)


def test_two_million_leaves_depth_and_distribution(tree_cls):
    """The BASELINE config-4 size: 2M leaves = 4 levels; sampling frequency follows the priorities."""
    n = 2 * 1024 * 1024
    tree = tree_cls(n, torch.device("cuda"))
    assert tree.depth == 4
    g = torch.Generator(device="cuda").manual_seed(0)
    prio = torch.rand(n, device="cuda", generator=g) + 0.01
    prio[: n // 2] *= 3.0                      # the first half carries 3/4 of the mass... (0.51*3 : 0.51)
    tree.update(torch.arange(n, device="cuda", dtype=torch.int32), prio)
    hits = 0
    B, reps = 4096, 8
    for r in range(reps):
        u = torch.rand(B, device="cuda", generator=g)
        idx, w = tree.sample(u, 0.4)
        hits += int((idx < n // 2).sum())
        assert float(w.max()) <= 1.0 + 1e-6 and float(w.min()) > 0
    frac = hits / (B * reps)
    expected = float(prio[: n // 2].double().sum() / prio.double().sum())
    assert abs(frac - expected) < 0.02
    # partial update of a few leaves repairs all ancestors
    some = torch.randint(0, n, (4096,), device="cuda", generator=g, dtype=torch.int32)
    newv = torch.full((4096,), 5.0, device="cuda")
    tree.update(some, newv)
    prio[some.long()] = 5.0
    assert abs(float(tree.total()) - float(prio.double().sum())) < 1e-4 * float(prio.double().sum())


@pytest.mark.gpu
def test_sample_rng_equals_sample_of_the_same_counter_hash():
    """The prioritized step draws its stratum uniforms inside the descent kernel (sample_rng): the same indices and
    weights as uniform_f32 + sample, one launch less."""
    import ctypes
    from pytorch_distributed_amd.ops import lib
    from pytorch_distributed_amd.ops.segtree import DeviceSumMinTree
    dev = torch.device("cuda:0")
    n, B = 50000, 128
    tree = DeviceSumMinTree(n, dev)
    tree.update(torch.arange(n, device=dev, dtype=torch.int32), torch.rand(n, device=dev) + 0.05)
    counter = torch.tensor([7], dtype=torch.int64, device=dev)
    seed = 0x1234567
    L = lib()
    L.pda_uniform_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    L.pda_uniform_f32.restype = None
    u = torch.zeros(B, device=dev)
    L.pda_uniform_f32(u.data_ptr(), B, seed, counter.data_ptr(), torch.cuda.current_stream().cuda_stream)
    i0, w0 = tree.sample(u, 0.4)
    i1 = torch.zeros(B, dtype=torch.int32, device=dev)
    w1 = torch.zeros(B, device=dev)
    tree.sample_rng(B, 0.4, seed, counter, i1, w1)
    torch.cuda.synchronize()
    assert torch.equal(i0, i1) and torch.equal(w0, w1)
    assert (u > 0).all() and (u < 1).all() and i1.unique().numel() > B // 2
