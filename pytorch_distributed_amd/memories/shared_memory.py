"""Host replay in POSIX shared memory, for the no-GPU plumbing config and as the
behavioural baseline (reference core/memories/shared_memory.py:10-75): ring
buffers of uint8/float32 states plus float32 action/reward/gamma/terminal
columns, shared by every spawned process, one mp.Lock.

Differences from the reference: `sample` does not cast states to float32 under
the lock (models normalise on their own device), sampling uses a per-process
RandomState, and `enable_per` is real: a shared-memory sum/min tree
(utils/segment_tree.py) holds (|priority| + eps)^alpha, sampling is stratified
and returns importance weights (p_min / p_i)^beta.
"""
import numpy as np
import torch
from ..utils import mpctx as mp

from ..utils.segment_tree import SumMinTree, next_pow2
from .memory import Memory


class SharedMemory(Memory):
    def __init__(self, args):
        super().__init__(args)
        self.pos = mp.Value("l", 0)
        self.full = mp.Value("b", False)
        self.total_fed = mp.Value("l", 0)

        m = self.memory_size
        self.state0s = torch.zeros((m,) + tuple(self.state_shape), dtype=self.state_dtype)
        self.state1s = torch.zeros((m,) + tuple(self.state_shape), dtype=self.state_dtype)
        self.actions = torch.zeros(m, self.action_shape)
        self.rewards = torch.zeros(m, self.reward_shape)
        self.gamma1s = torch.zeros(m, self.gamma_shape)
        self.terminal1s = torch.zeros(m, self.terminal_shape)
        for t in (self.state0s, self.actions, self.rewards, self.gamma1s, self.state1s, self.terminal1s):
            t.share_memory_()

        self._tree = None
        if self.enable_per:
            cap = next_pow2(m)
            self.sum_nodes = torch.zeros(2 * cap, dtype=torch.float64).share_memory_()
            self.min_nodes = torch.full((2 * cap,), float("inf"), dtype=torch.float64).share_memory_()

        self.memory_lock = mp.Lock()
        self._rng = None

    # shared tensors re-map in each process: build the numpy views lazily
    @property
    def tree(self):
        if self._tree is None:
            self._tree = SumMinTree(self.memory_size, sum_buf=self.sum_nodes.numpy(),
                                    min_buf=self.min_nodes.numpy())
        return self._tree

    @property
    def rng(self):
        if self._rng is None:
            import os
            self._rng = np.random.RandomState((os.getpid() * 2654435761) % (2 ** 31))
        return self._rng

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_tree"] = None
        d["_rng"] = None
        return d

    @property
    def size(self):
        if self.full.value:
            return self.memory_size
        return self.pos.value

    def _priority_to_leaf(self, priority):
        return (np.abs(priority) + self.priority_eps) ** self.priority_exponent

    def _feed(self, experience, priority=0.0):
        state0, action, reward, gamma1, state1, terminal1 = experience
        p = self.pos.value
        self.state0s[p] = torch.as_tensor(np.asarray(state0)).to(self.state_dtype)
        self.actions[p] = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(-1))
        self.rewards[p] = torch.as_tensor(np.asarray(reward, dtype=np.float32).reshape(-1))
        self.gamma1s[p] = torch.as_tensor(np.asarray(gamma1, dtype=np.float32).reshape(-1))
        self.state1s[p] = torch.as_tensor(np.asarray(state1)).to(self.state_dtype)
        self.terminal1s[p] = float(terminal1)
        if self.enable_per:
            self.tree.update([p], [self._priority_to_leaf(priority)])
        self.pos.value = p + 1
        self.total_fed.value += 1
        if self.pos.value == self.memory_size:
            self.full.value = True
            self.pos.value = 0

    def _gather(self, inds):
        inds = torch.as_tensor(inds, dtype=torch.long)
        return (self.state0s[inds], self.actions[inds], self.rewards[inds],
                self.gamma1s[inds], self.state1s[inds], self.terminal1s[inds])

    def feed(self, experience, priority=0.0):
        with self.memory_lock:
            return self._feed(experience, priority)

    def sample(self, batch_size):
        with self.memory_lock:
            if self.enable_per:
                inds = self.tree.sample_stratified(batch_size, self.rng)
            else:
                inds = self.rng.randint(0, self.size, size=batch_size)
            return self._gather(inds)

    def sample_with_priorities(self, batch_size):
        if not self.enable_per:
            return super().sample_with_priorities(batch_size)
        with self.memory_lock:
            inds = self.tree.sample_stratified(batch_size, self.rng)
            inds = np.minimum(inds, max(self.size - 1, 0))
            weights = self.tree.importance_weights(inds, self.priority_weight)
            batch = self._gather(inds)
        return batch, inds, torch.as_tensor(weights, dtype=torch.float32).reshape(-1, 1)

    def update_priorities(self, indices, td_abs):
        if not self.enable_per or indices is None:
            return
        td_abs = np.asarray(torch.as_tensor(td_abs).detach().cpu().reshape(-1), dtype=np.float64)
        with self.memory_lock:
            self.tree.update(np.asarray(indices), self._priority_to_leaf(td_abs))
