"""Host-side sum/min segment trees for prioritized replay.

The reference ships a list-based recursive sum tree that fails at import and is
never used (utils/segment_tree.py:3-69).  This is the finished host version:
array-backed, batched, with a min tree for importance weights.  It backs the
CPU prioritized replay (memories/shared_memory.py) and is the oracle for the
HIP segment-tree kernels (ops/segtree.py), which use a radix-64 layout sized
for 64-lane wavefronts but must return the same leaf for the same prefix mass.

Layout: 1-based binary heap over `capacity` (power of two) leaves; node 1 is
the root, leaves live at [capacity, 2*capacity).
"""
import numpy as np


def next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


class SumMinTree(object):
    def __init__(self, num_leaf_nodes, dtype=np.float64, sum_buf=None, min_buf=None):
        self.data_size = int(num_leaf_nodes)
        self.capacity = next_pow2(self.data_size)
        n = 2 * self.capacity
        self.sum_nodes = np.zeros(n, dtype=dtype) if sum_buf is None else sum_buf
        self.min_nodes = np.full(n, np.inf, dtype=dtype) if min_buf is None else min_buf
        assert self.sum_nodes.shape[0] == n and self.min_nodes.shape[0] == n

    # -- updates ---------------------------------------------------------
    def update(self, data_inds, values):
        """Set leaves `data_inds` to `values` (last write wins on duplicates)
        and repair every ancestor, level by level."""
        data_inds = np.atleast_1d(np.asarray(data_inds, dtype=np.int64))
        values = np.broadcast_to(np.atleast_1d(np.asarray(values, dtype=self.sum_nodes.dtype)), data_inds.shape)
        assert data_inds.size == 0 or (data_inds.min() >= 0 and data_inds.max() < self.data_size)
        nodes = data_inds + self.capacity
        self.sum_nodes[nodes] = values
        self.min_nodes[nodes] = values
        nodes = np.unique(nodes // 2)
        while nodes.size and nodes[0] >= 1:
            left, right = 2 * nodes, 2 * nodes + 1
            self.sum_nodes[nodes] = self.sum_nodes[left] + self.sum_nodes[right]
            self.min_nodes[nodes] = np.minimum(self.min_nodes[left], self.min_nodes[right])
            if nodes[0] == 1:
                break
            nodes = np.unique(nodes // 2)

    # -- queries ---------------------------------------------------------
    def total_sum(self):
        return float(self.sum_nodes[1])

    def min(self):
        return float(self.min_nodes[1])

    def get(self, data_inds):
        return self.sum_nodes[np.asarray(data_inds, dtype=np.int64) + self.capacity]

    def find(self, values):
        """For each prefix mass v return the first leaf whose inclusive prefix
        sum reaches v (go left when v <= left-subtree mass, as the reference's
        _retrieve does, utils/segment_tree.py:50-57)."""
        values = np.atleast_1d(np.asarray(values, dtype=self.sum_nodes.dtype)).copy()
        nodes = np.ones(values.shape, dtype=np.int64)
        while nodes[0] < self.capacity:
            left = 2 * nodes
            left_mass = self.sum_nodes[left]
            go_right = values > left_mass
            values = np.where(go_right, values - left_mass, values)
            nodes = left + go_right.astype(np.int64)
        data_inds = nodes - self.capacity
        return np.minimum(data_inds, self.data_size - 1)

    def sample_stratified(self, batch_size, rng=np.random):
        """One draw from each of `batch_size` equal-mass strata."""
        total = self.total_sum()
        u = (np.arange(batch_size) + rng.uniform(size=batch_size)) / batch_size * total
        return self.find(u)

    def importance_weights(self, data_inds, beta):
        """w_i = (N p_i / sum)^-beta / max_j w_j = (p_min / p_i)^beta."""
        p = self.get(data_inds)
        return (self.min() / np.maximum(p, 1e-30)) ** beta


class SegmentTree(SumMinTree):
    """Reference-shaped facade (append / find / get / total_sum) over SumMinTree
    (reference utils/segment_tree.py:13-69)."""

    def __init__(self, num_leaf_nodes):
        super().__init__(num_leaf_nodes)
        self.data = [None] * self.data_size
        self.data_ind = 0
        self.full = False

    def append(self, data, value):
        self.data[self.data_ind] = data
        self.update([self.data_ind], [value])
        self.data_ind += 1
        if self.data_ind == self.data_size:
            self.full = True
            self.data_ind = 0

    def find(self, value):
        if np.ndim(value) > 0:
            return super().find(value)
        data_ind = int(super().find([value])[0])
        node_ind = data_ind + self.capacity
        return (float(self.sum_nodes[node_ind]), data_ind, node_ind)

    def get(self, data_ind):
        if np.ndim(data_ind) > 0:
            return super().get(data_ind)
        return self.data[data_ind % self.data_size]
