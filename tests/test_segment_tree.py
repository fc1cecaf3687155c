"""Sum/min trees: host oracle vs brute force, and the radix-64 device-layout
tree (CPU fallback here; the HIP kernels run the same checks in test_gpu_*)."""
import numpy as np
import pytest
import torch

from pytorch_distributed_amd.utils.segment_tree import SumMinTree, SegmentTree
from pytorch_distributed_amd.ops.segtree import DeviceSumMinTree, level_sizes


def brute_find(p, v):
    c = np.cumsum(p)
    return int(min(np.searchsorted(c, v, side="left"), len(p) - 1))


def test_reference_smoke_script():
    # the inline script at the top of the reference's utils/segment_tree.py
    tree = SegmentTree(4)
    for i, v in enumerate([0.1, 0.3, 0.4, 0.2]):
        tree.append(i, v)
    assert tree.total_sum() == pytest.approx(1.0)
    assert tree.find(0.5)[1] == 2 and tree.find(0.15)[1] == 1
    assert tree.get(2) == 2


@pytest.mark.parametrize("n", [1, 5, 64, 1000])
def test_host_tree_matches_bruteforce(n):
    rng = np.random.RandomState(n)
    tree = SumMinTree(n)
    p = np.zeros(n)
    for _ in range(5):
        k = rng.randint(1, n + 1)
        idx = rng.randint(0, n, size=k)
        val = rng.randint(1, 100, size=k).astype(np.float64)
        tree.update(idx, val)
        for i, v in zip(idx, val):      # last write wins
            p[i] = v
        assert tree.total_sum() == pytest.approx(p.sum())
        nz = p[p > 0]
        vs = rng.uniform(0, p.sum(), size=50)
        got = tree.find(vs)
        for v, g in zip(vs, got):
            assert g == brute_find(p, v)
    w = tree.importance_weights(np.arange(n)[p > 0], 0.4)
    assert w.max() == pytest.approx((tree.min() / nz.min()) ** 0.4) or n == 1


def test_stratified_sampling_distribution():
    rng = np.random.RandomState(0)
    n = 256
    tree = SumMinTree(n)
    p = rng.uniform(0.1, 2.0, size=n)
    tree.update(np.arange(n), p)
    counts = np.zeros(n)
    for _ in range(400):
        np.add.at(counts, tree.sample_stratified(64, rng), 1)
    freq = counts / counts.sum()
    assert np.abs(freq - p / p.sum()).max() < 0.004


def test_level_sizes():
    assert level_sizes(50000) == [50048, 832, 13]
    assert level_sizes(2 * 1024 * 1024) == [2097152, 32768, 512, 8]
    assert level_sizes(10) == [64]


@pytest.mark.parametrize("n", [10, 64, 65, 5000])
def test_device_layout_tree_cpu_matches_host_oracle(n):
    rng = np.random.RandomState(n)
    host = SumMinTree(n, dtype=np.float64)
    dev = DeviceSumMinTree(n, "cpu")
    for _ in range(4):
        k = rng.randint(1, n + 1)
        idx = np.unique(rng.randint(0, n, size=k))
        val = rng.randint(1, 50, size=idx.size).astype(np.float64)     # integers: exact in fp32
        host.update(idx, val)
        dev.update(torch.as_tensor(idx), torch.as_tensor(val, dtype=torch.float32))
        assert float(dev.total()) == pytest.approx(host.total_sum())
        assert float(dev.min_value()) == pytest.approx(host.min())
    B = 32
    u = torch.rand(B)
    idx, w = dev.sample(u, 0.4)
    targets = (np.arange(B) + u.numpy().astype(np.float64)) / B * host.total_sum()
    expect = host.find(targets.astype(np.float32).astype(np.float64))
    assert np.array_equal(idx.numpy(), expect)
    assert np.allclose(w.numpy(), host.importance_weights(expect, 0.4), rtol=1e-5)
