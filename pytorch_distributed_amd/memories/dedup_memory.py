"""Frame-deduplicated HBM replay: every frame is stored once.

The reference keeps both frame stacks of a transition in full (core/memories/shared_memory.py:19-24: 2 x 4 x 84 x 84
bytes per transition, 113 GB for 2 M), and so does DeviceMemory.  Consecutive states of one env share all but one
frame, so here the actors send RUNS (parallel/csrc/frame_runs.h: the planes of consecutive steps once, in order,
plus the transitions as two window offsets into them; packed by the n-step assembler in parallel/csrc/shm_ring.cpp)
and the learner keeps

    planes  uint8 [F, plane_bytes]     a ring of planes; a run's planes are contiguous in it
    tails   fp32  [N, tail_floats]     action | reward | gamma | terminal | initial priority, as DeviceMemory
    ref0 / refN  int32 [N]             first plane of the s0 / sN window of a transition

A state is the `window` consecutive planes that start at its ref -- the same contiguous bytes the Nature-CNN
kernels read from a row of full stacks, found at `planes + ref[idx[b]] * plane_bytes` instead of
`states + idx[b] * stack_bytes` (ops/nn.py FrameWindows, ops/csrc/nature_cnn.h frame_index).  With runs of 96 planes
and n-step 5 that is ~1.12 planes per transition instead of 8: 2 M transitions take 16 GB, and the ingest ring,
the pinned buffers and the H2D copies carry a seventh of the bytes.

Both rings are FIFO in arrival order but advance at different rates (episode starts and run restarts add planes
without transitions), so a transition can outlive its planes.  The host keeps, per ingested chunk, where its planes and
transitions started, and publishes the SAMPLING WINDOW -- the newest transitions whose planes are all still in the
ring (less a safety margin of two chunks, for index draws already queued) -- as {rows, first row, ring length} in
`window_dev`, which the uniform draw kernels read (ops/csrc/common.h window_pick) and the unpack kernel updates in
stream order.  With prioritized replay the rows that fall out of the window early get zero mass.  Sized by
`dedup_plane_slack` (planes per transition row; default 1.25) the window is the whole ring.
"""
from collections import deque

import numpy as np
import torch

from ..ops import replay as replay_ops
from ..ops.nn import FrameWindows
from ..parallel.ingest import RunCodec, ShmRing
from .device_memory import DeviceMemory

MAX_RUNS_PER_CHUNK = 16     # ops/csrc/replay_kernels.hip PDA_MAX_RUNS


def _pow2_floor(v):
    p = 1
    while p * 2 <= v:
        p *= 2
    return p


class DedupDeviceMemory(DeviceMemory):
    dedup = True

    def _make_ring(self, args):
        c = self.codec
        self.run_codec = RunCodec(self.state_shape, c.action_shape, c.num_tasks,
                                  plane_cap=int(getattr(args, "dedup_plane_cap", 96)))
        if self.state_dtype != torch.uint8:
            raise ValueError("frame-deduplicated replay stores uint8 frame stacks (memory_type %r with float states)"
                             % getattr(args, "memory_type", "device-dedup"))
        budget = int(float(getattr(args, "dedup_ring_mb", 64)) * (1 << 20))
        slots = max(8, min(1024, _pow2_floor(max(budget // self.run_codec.record_bytes, 1))))
        self.ring = ShmRing(slots, self.run_codec.record_bytes)
        self.ingest_chunk = max(1, min(MAX_RUNS_PER_CHUNK, int(getattr(args, "dedup_chunk_runs", 8))))
        self.plane_slack = float(getattr(args, "dedup_plane_slack", 1.25))
        self.max_age_ms = int(getattr(args, "dedup_max_age_ms", 250))

    # ---- actor face ------------------------------------------------------------
    def feed(self, experience, priority=0.0):
        """One transition as a run of its own (2 x window planes): what a producer without the run packer sends.
        The actors use agents/nstep.py NativeFeeder, which packs whole runs."""
        state0, action, reward, gamma1, state1, terminal1 = experience
        rc, c = self.run_codec, self.codec
        w = rc.window
        s0 = np.ascontiguousarray(state0, dtype=np.uint8).reshape(w, -1)
        s1 = np.ascontiguousarray(state1, dtype=np.uint8).reshape(w, -1)
        rec = rc.pack(np.concatenate([s0, s1]), c.tail(action, reward, gamma1, terminal1, priority)[None], [0], [w])
        if not self.ring.push_segments([rec[:rc.planes_offset + 2 * w * rc.plane_stride]]):
            raise RuntimeError("replay ingestion ring stayed full for 30 s (is a learner draining it?)")

    @property
    def size(self):
        if self.attached:
            return self.valid_rows
        return self.shared_size.value

    # ---- learner face --------------------------------------------------------------
    def _alloc_storage(self, m, dev):
        rc = self.run_codec
        margin = 2 * self.ingest_chunk * rc.plane_cap
        self.num_planes = int(m * self.plane_slack) + 2 * margin + rc.window
        self.planes = torch.zeros(self.num_planes, rc.plane_stride, dtype=torch.uint8, device=dev)
        self.ref0 = torch.zeros(m, dtype=torch.int32, device=dev)
        self.refN = torch.zeros(m, dtype=torch.int32, device=dev)
        self.window_dev = torch.zeros(4, dtype=torch.int32, device=dev)
        self._alloc_tails(m, dev)
        self._margin = margin
        self.planes_linear = 0          # planes written so far, skipped ring tails included
        self.trans_linear = 0           # transitions written so far
        self.valid_rows, self.first_row = 0, 0
        self._chunks = deque()          # (planes_linear, trans_linear) at the start of every chunk still (partly) alive
        self.dropped_early = 0          # rows that left the sampling window before they were overwritten
        self.planes_received = 0

    def frame_windows(self):
        """(s0, sN) frame sources for the Nature-CNN kernels (ops/nn.py FrameWindows)."""
        w = self.run_codec.window
        return FrameWindows(self.planes, self.ref0, w), FrameWindows(self.planes, self.refN, w)

    def _chunk_meta(self, host_records, n):
        rec = host_records.numpy()
        return [self.run_codec.header(rec[r]) for r in range(n)]

    def _place(self, counts):
        """Plane ring / transition ring placement of a chunk of runs, and the sampling window once it is in."""
        F, N = self.num_planes, self.memory_size
        self._chunks.append((self.planes_linear, self.trans_linear))
        plane_base, trans_pos = [], []
        for n_p, n_t in counts:
            at = self.planes_linear % F
            if at + n_p > F:                        # a run stays contiguous: skip the ring's tail
                self.planes_linear += F - at
                at = 0
            plane_base.append(at)
            trans_pos.append(self.trans_linear % N)
            self.planes_linear += n_p
            self.trans_linear += n_t
            self.planes_received += n_p
        # chunks whose planes are (about to be) overwritten: everything older than F - margin planes
        horizon = self.planes_linear + self._margin - F
        while len(self._chunks) > 1 and self._chunks[0][0] < horizon:
            self._chunks.popleft()
        oldest_alive = self._chunks[0][1] if self._chunks[0][0] >= horizon else self.trans_linear
        first = max(self.trans_linear - N, oldest_alive)
        return plane_base, trans_pos, first

    def ingest_records(self, records_dev, n, meta=None):
        """Unpack n run records (already on the replay device); meta: their (planes, transitions) counts."""
        counts = meta if meta is not None else self._chunk_meta(records_dev.cpu(), n)
        N = self.memory_size
        old_first, old_total = self.trans_linear - self.valid_rows, self.trans_linear
        plane_base, trans_pos, first = self._place(counts)
        rows = self.trans_linear - first
        n_t = self.trans_linear - old_total
        fused_leaves = self.tree is not None and self.tree.native
        c = self.codec
        replay_ops.unpack_runs(records_dev, n, self.run_codec, counts, plane_base, trans_pos, self.planes, self.tails,
                               self.ref0, self.refN, self.window_dev, (rows, first % N, N),
                               leaves=(self.tree.sum, self.tree.min, c.action_shape + 3 * c.num_tasks, self.priority_eps,
                                       self.priority_exponent) if fused_leaves else None)
        if self.tree is not None:
            if n_t and fused_leaves:
                # the unpack kernel wrote the new rows' leaves; one single-workgroup launch repairs above them
                self.tree.repair_range(old_total % N, n_t)
            elif n_t:
                pos = ((torch.arange(n_t, device=self.device) + old_total) % N).to(torch.int32)
                self.tree.update(pos, self._leaf_priority(self.init_priorities[pos.long()]))
            # rows that left the window without being overwritten by this chunk keep no mass
            lo, hi = max(old_first, self.trans_linear - N), min(first, old_total)
            if hi > lo:
                self.tree.clear((torch.arange(lo, hi, device=self.device) % N).to(torch.int32))
                self.dropped_early += hi - lo
        self.valid_rows, self.first_row = rows, first % N
        self.pos, self.full = self.trans_linear % N, self.trans_linear >= N
        self.total_fed.value += n_t
        if self.rank == 0:
            self.shared_size.value = self.valid_rows

    def window(self):
        """(rows, first row, ring length) of the sampling window (what window_dev holds on the device)."""
        return self.valid_rows, self.first_row, self.memory_size

    def feed_batch(self, state0s, actions, rewards, gamma1s, state1s, terminal1s, priorities=None):
        """Direct device-side ingestion of n transitions (benchmark fill, tests): each goes in as a run of its own
        planes -- s0's window, then sN's (no sharing between the transitions of a batch)."""
        n, rc, N = int(state0s.shape[0]), self.run_codec, self.memory_size
        w = rc.window
        counts = [(2 * w, 1)] * n
        for lo in range(0, n, 4096):
            hi = min(n, lo + 4096)
            k = hi - lo
            plane_base, trans_pos, first = self._place(counts[lo:hi])
            pb = torch.tensor(plane_base, device=self.device, dtype=torch.long)
            tp = torch.tensor(trans_pos, device=self.device, dtype=torch.long)
            rows = (pb[:, None] + torch.arange(2 * w, device=self.device)[None, :]).reshape(-1)
            both = torch.cat([state0s[lo:hi].reshape(k, w, -1), state1s[lo:hi].reshape(k, w, -1)], dim=1)
            self.planes[rows] = both.reshape(k * 2 * w, -1).to(torch.uint8)
            self.ref0[tp] = pb.to(torch.int32)
            self.refN[tp] = (pb + w).to(torch.int32)
            self.actions[tp] = actions[lo:hi].reshape(k, -1).float()
            self.rewards[tp] = rewards[lo:hi].reshape(k, -1).float()
            self.gamma1s[tp] = gamma1s[lo:hi].reshape(k, -1).float()
            self.terminal1s[tp] = terminal1s[lo:hi].reshape(k, -1).float()
            pr = torch.ones(k, device=self.device) if priorities is None else priorities[lo:hi].float()
            self.init_priorities[tp] = pr
            if self.tree is not None:
                self.tree.update(tp.to(torch.int32), self._leaf_priority(pr))
            self.valid_rows, self.first_row = self.trans_linear - first, first % N
        self.window_dev.copy_(torch.tensor([self.valid_rows, self.first_row, N, 0], dtype=torch.int32))
        self.pos, self.full = self.trans_linear % N, self.trans_linear >= N
        self.total_fed.value += n
        if self.rank == 0:
            self.shared_size.value = self.valid_rows

    def fill_synthetic(self, n, nstep=5, seed=0, num_actions=1, gamma_n=0.95):
        """n transitions over random planes laid out as the actors' runs are (benchmarks): one plane per step, s0 at
        plane p, sN at plane p + nstep; random actions / clipped rewards / 1 % terminals / priorities in
        [0.05, 1.05).  Needs an empty memory."""
        assert self.trans_linear == 0 and n <= self.memory_size
        rc, N = self.run_codec, self.memory_size
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        need = n + nstep + rc.window
        assert need <= self.num_planes
        for lo in range(0, need, 1 << 14):
            hi = min(need, lo + (1 << 14))
            self.planes[lo:hi] = torch.randint(0, 256, (hi - lo, rc.plane_stride), device=self.device, generator=g,
                                               dtype=torch.uint8)
        p = torch.arange(n, device=self.device, dtype=torch.int32)
        self.ref0[:n], self.refN[:n] = p, p + nstep
        self.actions[:n] = torch.randint(0, max(int(num_actions), 1), (n, self.actions.shape[1]), device=self.device,
                                         generator=g).float()
        self.rewards[:n] = torch.randn(n, self.rewards.shape[1], device=self.device, generator=g).clamp_(-1, 1)
        self.gamma1s[:n] = float(gamma_n)
        self.terminal1s[:n] = (torch.rand(n, self.terminal1s.shape[1], device=self.device, generator=g) < 0.01).float()
        self.init_priorities[:n] = torch.rand(n, device=self.device, generator=g) + 0.05
        if self.tree is not None:
            for lo in range(0, n, 1 << 16):
                hi = min(n, lo + (1 << 16))
                self.tree.update(p[lo:hi], self._leaf_priority(self.init_priorities[lo:hi]))
        self._chunks.append((0, 0))
        self.planes_linear, self.trans_linear, self.planes_received = need, n, need
        self.valid_rows, self.first_row = n, 0
        self.window_dev.copy_(torch.tensor([n, 0, N, 0], dtype=torch.int32))
        self.pos, self.full = n % N, n >= N
        self.total_fed.value += n
        if self.rank == 0:
            self.shared_size.value = n

    # ---- sampling --------------------------------------------------------------------
    def sample_indices(self, batch_size, out_idx=None, out_weight=None):
        if self.tree is not None:
            return super().sample_indices(batch_size, out_idx, out_weight)
        r = torch.randint(0, max(self.valid_rows, 1), (batch_size,), device=self.device, generator=self._gen)
        idx = ((r + self.first_row) % self.memory_size).to(torch.int32)
        w = torch.ones(batch_size, device=self.device)
        if out_idx is not None:
            out_idx.copy_(idx)
            out_weight.copy_(w)
            return out_idx, out_weight
        return idx, w

    def states(self, ref, idx):
        """The frame stacks [B, C, ...] of transitions idx through one of the two ref tables."""
        w = self.run_codec.window
        rows = ref[idx.long()].long()[:, None] + torch.arange(w, device=self.device)[None, :]
        flat = self.planes[rows.reshape(-1)][:, :self.run_codec.plane_bytes]
        return flat.reshape((int(idx.numel()),) + tuple(self.state_shape))

    def _gather(self, idx):
        li = idx.long()
        return (self.states(self.ref0, idx), self.actions[li], self.rewards[li], self.gamma1s[li],
                self.states(self.refN, idx), self.terminal1s[li])

    def stats(self):
        t = max(self.trans_linear, 1)
        return {"transitions": self.trans_linear, "planes": self.planes_received,
                "planes_per_transition": self.planes_received / t, "plane_ring": self.num_planes,
                "rows_in_window": self.valid_rows, "dropped_early": self.dropped_early,
                "bytes_per_transition": self.planes_received * self.run_codec.plane_stride / t}
