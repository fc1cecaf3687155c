"""HBM-resident replay fed by many CPU actors without host round-trips on the
learner path.

The reference keeps replay in host shm behind one lock, casts sampled states to
float32 on the CPU and ships 28.9 MB per step over PCIe from unpinned memory
(core/memories/shared_memory.py:59-75, dqn_learner.py:63-64).  Here:

  actors --feed()--> lock-free shm ring (parallel/ingest.py, C++)
         learner: drain() pops a chunk into pinned memory, one async H2D copy,
         one unpack kernel scatters it into the HBM ring (uint8 states stay uint8)
         sample_indices(): uniform draw, or wave-per-sample descent of the HIP
         radix-64 sum tree with importance weights (memory_type "device-per")
         the conv1 kernel then reads frames *through the index vector*:
         no gather copy, no float inflation, no H2D in the step.

One object, two faces: pickled to actor processes it only exposes feed()/size;
in a learner process `attach(device, rank, world)` allocates that rank's shard
(memory_size / world transitions; 2M x 2 x 28 KB = 113 GB fits one MI355X).
With replay_device="cpu" the same code runs on host tensors (tests, no-GPU).
"""
import numpy as np
import torch
from ..utils import mpctx as mp

from ..ops import replay as replay_ops
from ..ops.segtree import DeviceSumMinTree
from ..parallel.ingest import ShmRing, TransitionCodec
from .memory import Memory


class DeviceMemory(Memory):
    def __init__(self, args):
        super().__init__(args)
        self.replay_device = getattr(args, "replay_device", "cpu")
        self.codec = TransitionCodec(self.state_shape, self.state_dtype, self.action_shape, self.reward_shape)
        self._make_ring(args)
        self.ingest_device_slabs = int(getattr(args, "ingest_device_slabs", 32))
        self.shared_size = mp.Value("l", 0)     # rank 0's resident transitions, visible to every process
        self.total_fed = mp.Value("l", 0)
        self.attached = False
        self._gen = None

    def _make_ring(self, args):
        self.ring = ShmRing(int(getattr(args, "ingest_ring_slots", 4096)), self.codec.record_bytes)
        self.ingest_chunk = int(getattr(args, "ingest_chunk", 256))

    # ---- actor face ------------------------------------------------------------
    def feed(self, experience, priority=0.0):
        state0, action, reward, gamma1, state1, terminal1 = experience
        c = self.codec
        s0 = np.ascontiguousarray(state0, dtype=c.state_np).reshape(-1)
        s1 = np.ascontiguousarray(state1, dtype=c.state_np).reshape(-1)
        pad = c.state_stride - c.state_bytes
        tail = c.tail(action, reward, gamma1, terminal1, priority)
        if pad:
            z = np.zeros(pad, dtype=np.uint8)
            s0 = np.concatenate([s0.view(np.uint8), z])
            s1 = np.concatenate([s1.view(np.uint8), z])
        ok = self.ring.push_segments([s0, s1, tail])
        if not ok:
            raise RuntimeError("replay ingestion ring stayed full for 30 s (is a learner draining it?)")

    @property
    def size(self):
        if self.attached:
            return self.memory_size if self.full else self.pos
        return self.shared_size.value

    def __getstate__(self):
        if self.attached:
            raise RuntimeError("an attached DeviceMemory cannot be pickled; pass it to children before attach()")
        return self.__dict__.copy()

    # ---- learner face --------------------------------------------------------------
    def attach(self, device, rank=0, world_size=1):
        self.device = torch.device(device if self.replay_device != "cpu" else "cpu")
        self.rank, self.world_size = rank, world_size
        self.memory_size = max(self.memory_size // world_size, 1)
        self._alloc_storage(self.memory_size, self.device)
        self._alloc_ingest(self.device, self.ring.payload_bytes)
        self.tree = DeviceSumMinTree(self.memory_size, self.device) if self.enable_per else None
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(1234 + rank)
        self.pos, self.full = 0, False
        self.attached = True
        return self

    def _alloc_tails(self, m, dev):
        c = self.codec
        self.tails = torch.zeros(m, c.tail_floats, dtype=torch.float32, device=dev)
        a, k = c.action_shape, c.num_tasks
        self.actions = self.tails[:, :a]
        self.rewards = self.tails[:, a:a + k]
        self.gamma1s = self.tails[:, a + k:a + 2 * k]
        self.terminal1s = self.tails[:, a + 2 * k:a + 3 * k]
        self.init_priorities = self.tails[:, a + 3 * k]

    def _alloc_storage(self, m, dev):
        self.state0s = torch.zeros((m,) + tuple(self.state_shape), dtype=self.state_dtype, device=dev)
        self.state1s = torch.zeros((m,) + tuple(self.state_shape), dtype=self.state_dtype, device=dev)
        self._alloc_tails(m, dev)

    def _alloc_ingest(self, dev, record_bytes):
        # ingestion staging: a small ring of pinned host buffers, each with its device twin and an
        # event, so drain() never waits for the GPU (a buffer is reused only once its copy landed)
        nbuf = 4 if dev.type == "cuda" else 1
        self.staging, self.staging_dev, self.staging_evt = [], [], []
        for _ in range(nbuf):
            host = torch.zeros(self.ingest_chunk, record_bytes, dtype=torch.uint8)
            if dev.type == "cuda":
                host = host.pin_memory()
                self.staging_dev.append(torch.zeros_like(host, device=dev))
                self.staging_evt.append(None)
            else:
                self.staging_dev.append(host)
                self.staging_evt.append(None)
            self.staging.append(host)
        self._stage_next = 0
        # Pumped path: the H2D copies run on a stream of their own, so a pinned buffer is free again as soon as its
        # DMA is done -- NOT after the learner stream, which the host keeps up to ~100 updates (tens of ms) ahead of
        # the GPU, has reached it: with 4 x 256 records in flight per such latency the actors were throttled to
        # ~32 k transitions/s (profiles/r2_node_*.json).  Only the unpack kernel is ordered into the learner stream
        # (rows change between two updates, never under one); it reads from a pool of device slabs (0.46 GB of the
        # 288 GB) that are recycled once their unpack has run.
        self._copy_stream = None
        self._slabs, self._slab_free, self._slab_busy = [], [], []
        if dev.type == "cuda":
            self._copy_stream = torch.cuda.Stream(device=dev)
            nslab = max(2, self.ingest_device_slabs)
            self._slabs = [torch.zeros(self.ingest_chunk, record_bytes, dtype=torch.uint8, device=dev)
                           for _ in range(nslab)]
            self._slab_free = list(range(nslab))
            self._copy_evt = [torch.cuda.Event() for _ in range(nslab)]
            self._unpack_evt = [torch.cuda.Event() for _ in range(nslab)]
        # GPU replay: a native pump thread fills the pinned buffers (parallel/csrc/shm_ring.cpp), the
        # learner thread only enqueues DMAs.  Host replay keeps the simple in-line pop.
        self.pump = None
        self._in_flight = []        # [(buffer index, event)] DMAs not yet known complete
        self._copying = []          # [(slab, records, meta)] chunks whose H2D copy was issued, unpack still due
        if dev.type == "cuda" and getattr(self, "use_pump", True):
            import ctypes
            from ..parallel import hostlib
            L = hostlib.lib()
            ptrs = (ctypes.c_void_p * nbuf)(*[t.data_ptr() for t in self.staging])
            self.pump = L.pda_pump_create(self.ring.base, nbuf, ptrs, self.ingest_chunk, record_bytes, 300)
            if not self.pump:
                raise RuntimeError("ingestion pump could not attach to the ring")
            self._pump_n = ctypes.c_int(0)
            import atexit
            atexit.register(self.close)     # the pump thread must stop before the ring's mapping goes away
        self._pos_base = torch.arange(self.ingest_chunk, device=dev, dtype=torch.int64)

    def _chunk_meta(self, host_records, n):
        """What ingest_records needs to know about a chunk from its HOST copy (read before the DMA is issued)."""
        return None

    def _leaf_priority(self, raw):
        return (raw.abs() + self.priority_eps) ** self.priority_exponent

    def ingest_records(self, records_dev, n, meta=None):
        """Scatter n packed records (already on the replay device) into the ring."""
        m = self.memory_size
        c = self.codec
        if self.tree is None:
            # uniform replay: the kernel derives every record's ring slot from (pos, m) itself -- no index tensor
            replay_ops.unpack_records(records_dev, n, None, self.state0s, self.state1s, self.tails,
                                      c.state_bytes, c.state_stride, c.tail_floats, pos_start=self.pos)
        elif self.tree.native:
            # prioritized, on the GPU: the unpack kernel also writes the new rows' leaves (|p| + eps)^alpha, one
            # single-workgroup launch repairs the tree above them (~12 launches per chunk before)
            replay_ops.unpack_records(records_dev, n, None, self.state0s, self.state1s, self.tails,
                                      c.state_bytes, c.state_stride, c.tail_floats, pos_start=self.pos,
                                      leaves=(self.tree.sum, self.tree.min, c.action_shape + 3 * c.num_tasks,
                                              self.priority_eps, self.priority_exponent))
            self.tree.repair_range(self.pos, n)
        else:
            base = self._pos_base[:n] if n <= self._pos_base.numel() else torch.arange(n, device=self.device)
            positions = ((base + self.pos) % m).to(torch.int32)
            replay_ops.unpack_records(records_dev, n, positions, self.state0s, self.state1s, self.tails,
                                      c.state_bytes, c.state_stride, c.tail_floats)
            self.tree.update(positions, self._leaf_priority(self.init_priorities[positions.long()]))
        self.pos += n
        if self.pos >= m:
            self.full = True
            self.pos -= m
        self.total_fed.value += n
        if self.rank == 0:
            self.shared_size.value = self.size

    def drain(self, max_chunks=4):
        """Move up to max_chunks*ingest_chunk transitions ring -> HBM.  Returns count.
        Never blocks on the GPU: the pop lands in a pinned buffer whose previous H2D copy has
        completed (event query), the copy and the unpack kernel are enqueued on the current
        stream behind whatever the learner already queued there."""
        if self.pump is not None:
            return self._drain_pumped(max_chunks)
        moved = 0
        cuda = self.device.type == "cuda"
        for _ in range(max_chunks):
            i = self._stage_next
            evt = self.staging_evt[i]
            if evt is not None and not evt.query():
                break                       # every staging buffer still in flight: try again next step
            n = self.ring.pop_many(self.staging[i], self.ingest_chunk)
            if n == 0:
                break
            meta = self._chunk_meta(self.staging[i], n)
            if cuda:
                self.staging_dev[i][:n].copy_(self.staging[i][:n], non_blocking=True)
                if evt is None:
                    evt = self.staging_evt[i] = torch.cuda.Event()
                evt.record()
            self.ingest_records(self.staging_dev[i], n, meta)
            self._stage_next = (i + 1) % len(self.staging)
            moved += n
            if n < self.ingest_chunk:
                break
        return moved

    def _drain_pumped(self, max_chunks):
        import ctypes
        from ..parallel import hostlib
        L = hostlib.lib()
        while self._in_flight and self._in_flight[0][1].query():        # DMA done: the pinned buffer goes back
            L.pda_pump_release(self.pump, self._in_flight.pop(0)[0])
        while self._slab_busy and self._unpack_evt[self._slab_busy[0]].query():     # unpacked: the slab is free
            self._slab_free.append(self._slab_busy.pop(0))
        moved = 0
        learner_stream = torch.cuda.current_stream(self.device)
        # 1) chunks whose H2D copy HAS LANDED get their unpack kernel on the learner stream.  The learner stream never
        #    waits for a DMA that is still in flight: a 14.5 MB chunk takes ~0.3 ms over PCIe, i.e. three whole learner
        #    updates of stall per chunk when the unpack was ordered behind the copy it had just issued (round 2:
        #    7.4 k updates/s inside the job against 10.4 k alone)
        while self._copying and self._copy_evt[self._copying[0][0]].query():
            d, n, meta = self._copying.pop(0)
            self.chunks_unpacked = getattr(self, "chunks_unpacked", 0) + 1
            learner_stream.wait_event(self._copy_evt[d])    # complete already: an ordering edge, not a stall
            self.ingest_records(self._slabs[d], n, meta)
            self._unpack_evt[d].record(learner_stream)
            self._slab_busy.append(d)
            moved += n
        # 2) new chunks start their copies on the copy stream
        for _ in range(max_chunks):
            if not self._slab_free:
                break               # every slab waits for the learner stream: back-pressure on the ring
            i = L.pda_pump_take(self.pump, ctypes.byref(self._pump_n))
            if i < 0:
                break
            n = self._pump_n.value
            d = self._slab_free.pop(0)
            meta = self._chunk_meta(self.staging[i], n)
            with torch.cuda.stream(self._copy_stream):
                self._slabs[d][:n].copy_(self.staging[i][:n], non_blocking=True)
                self._copy_evt[d].record()
            self._in_flight.append((i, self._copy_evt[d]))
            self._copying.append((d, n, meta))
        return moved

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if getattr(self, "pump", None):
            from ..parallel import hostlib
            hostlib.lib().pda_pump_destroy(self.pump)
            self.pump = None

    def feed_batch(self, state0s, actions, rewards, gamma1s, state1s, terminal1s, priorities=None):
        """Direct device-side ingestion of n transitions already on the replay
        device (used by the on-GPU actor pool and the benchmark fill)."""
        n = state0s.shape[0]
        m = self.memory_size
        pos = ((torch.arange(n, device=self.device) + self.pos) % m)
        self.state0s[pos] = state0s.to(self.state_dtype)
        self.state1s[pos] = state1s.to(self.state_dtype)
        self.actions[pos] = actions.reshape(n, -1).float()
        self.rewards[pos] = rewards.reshape(n, -1).float()
        self.gamma1s[pos] = gamma1s.reshape(n, -1).float()
        self.terminal1s[pos] = terminal1s.reshape(n, -1).float()
        if priorities is None:
            priorities = torch.ones(n, device=self.device)
        self.init_priorities[pos] = priorities.float()
        if self.tree is not None:
            self.tree.update(pos.to(torch.int32), self._leaf_priority(priorities.float()))
        self.pos += n
        if self.pos >= m:
            self.full = True
            self.pos %= m
        self.total_fed.value += n
        if self.rank == 0:
            self.shared_size.value = self.size

    # ---- sampling --------------------------------------------------------------------
    def sample_indices(self, batch_size, out_idx=None, out_weight=None):
        """(idx int32 (B,), weights fp32 (B,)) on the replay device, no host sync."""
        if self.tree is not None:
            u = torch.rand(batch_size, device=self.device, generator=self._gen)
            return self.tree.sample(u, self.priority_weight, out_idx, out_weight)
        idx = torch.randint(0, max(self.size, 1), (batch_size,), device=self.device,
                            generator=self._gen, dtype=torch.int32)
        w = torch.ones(batch_size, device=self.device)
        if out_idx is not None:
            out_idx.copy_(idx)
            out_weight.copy_(w)
            return out_idx, out_weight
        return idx, w

    def _gather(self, idx):
        li = idx.long()
        return (replay_ops.gather_states(self.state0s, idx), self.actions[li], self.rewards[li],
                self.gamma1s[li], replay_ops.gather_states(self.state1s, idx), self.terminal1s[li])

    def sample(self, batch_size):
        idx, _ = self.sample_indices(batch_size)
        return self._gather(idx)

    def sample_with_priorities(self, batch_size):
        idx, w = self.sample_indices(batch_size)
        return self._gather(idx), idx, w.reshape(-1, 1)

    def update_priorities(self, indices, td_abs):
        if self.tree is None or indices is None:
            return
        self.tree.update(indices.to(torch.int32), self._leaf_priority(td_abs.detach().reshape(-1).float()))
