"""Collectives for data-parallel learners and weight distribution.

The reference has no collectives at all: extra learners are Hogwild writers on
one shared CUDA model.  Here learners are synchronous
data-parallel ranks, one process per MI355X, and everything they exchange is a
single flat fp32 buffer:

* `all_reduce_mean_(flat_grads)`   one RCCL all-reduce per step (6.75 MB for
  the Nature CNN -- latency-bound on xGMI, so exactly one launch, no bucketing);
  optionally issued on a side stream so it overlaps the tail of backward;
* `broadcast_(flat_params, src)`   initial weight sync and learner -> inference
  rank weight publish;
* `all_agree(flag)`                MIN-reduce of a readiness flag so every rank
  enters (and leaves) the collective loop on the same step.

`TorchComm` wraps torch.distributed ("nccl" is RCCL on ROCm; "gloo" on CPU).
`FakeComm` is an in-process stand-in (world of W logical ranks sharing tensors)
so the DP arithmetic is unit-testable with no process group at all;
`SingleComm` is the world_size == 1 degenerate path.
"""
import os

import torch

try:
    import torch.distributed as dist
except Exception:   # pragma: no cover
    dist = None


class SingleComm(object):
    rank, world_size = 0, 1

    def all_reduce_mean_(self, flat, async_op=False):
        return None

    def all_reduce_sum_(self, flat):
        return flat

    def broadcast_(self, flat, src=0):
        return flat

    def all_agree(self, flag):
        return bool(flag)

    def agree_max(self, code):
        """MAX over ranks of a small non-negative status code (0 = carry on): every rank learns the worst news."""
        return int(code)

    def barrier(self):
        pass

    def wait(self, handle):
        pass

    def close(self):
        pass


class TorchComm(SingleComm):
    """torch.distributed-backed communicator (RCCL on GPU, gloo on CPU)."""

    def __init__(self, rank, world_size, device, init_method=None, backend=None, timeout_s=600, extra_ranks=0):
        """`extra_ranks` more processes join the same process group after the `world_size` learner
        ranks (the inference server receiving weight broadcasts, parallel/weight_sync.py).  The
        learners' collectives then run on a sub-group of ranks [0, world_size) and
        `self.weight_group` = {0, world_size, ...} carries the learner -> inference broadcast."""
        import datetime
        self.rank, self.world_size = rank, world_size
        self.total_ranks = world_size + int(extra_ranks)
        self.group = None
        self.weight_group = None
        self.device = torch.device(device)
        if backend is None:
            backend = "nccl" if self.device.type == "cuda" else "gloo"
        self.backend = backend
        self.owns_group = False
        if not dist.is_initialized():
            kwargs = dict(backend=backend, rank=rank, world_size=self.total_ranks,
                          timeout=datetime.timedelta(seconds=timeout_s))
            if init_method is not None:
                kwargs["init_method"] = init_method
            if backend == "nccl":
                torch.cuda.set_device(self.device)
                try:
                    kwargs["device_id"] = self.device
                except Exception:
                    pass
            dist.init_process_group(**kwargs)
            self.owns_group = True
        if extra_ranks:     # every process of the world creates both groups, in this order
            self.group = dist.new_group(ranks=list(range(world_size)))
            self.weight_group = dist.new_group(ranks=[0] + list(range(world_size, self.total_ranks)))
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _all_reduce(self, flat, op):
        """dist.all_reduce; a GPU tensor under the gloo back-end (rehearsals of the multi-rank learner
        on one card, tests) is staged through the host here: gloo's own device path stalls when the
        ranks share a GPU (both ranks inside the second all-reduce of a step, measured on this pool),
        and a host-side reduce is what that back-end does underneath anyway."""
        if self.backend == "gloo" and flat.is_cuda:
            host = flat.detach().to("cpu")      # waits for what is queued on the current stream
            dist.all_reduce(host, op=op, group=self.group)
            flat.copy_(host)
            return flat
        dist.all_reduce(flat, op=op, group=self.group)
        return flat

    def all_reduce_mean_(self, flat, async_op=False):
        """In-place mean over ranks.  With async_op the reduce runs on a side
        stream after everything already queued on the current stream; call
        wait(handle) before consuming `flat`."""
        if self.comm_stream is not None and async_op:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self._all_reduce(flat, dist.ReduceOp.SUM)
                flat.mul_(1.0 / self.world_size)
            return self.comm_stream
        self._all_reduce(flat, dist.ReduceOp.SUM)
        flat.mul_(1.0 / self.world_size)
        return None

    def wait(self, handle):
        if handle is not None:
            torch.cuda.current_stream(self.device).wait_stream(handle)

    def all_reduce_sum_(self, flat):
        return self._all_reduce(flat, dist.ReduceOp.SUM)

    def broadcast_(self, flat, src=0):
        if self.backend == "gloo" and flat.is_cuda:
            host = flat.detach().to("cpu")
            dist.broadcast(host, src=src, group=self.group)
            flat.copy_(host)
            return flat
        dist.broadcast(flat, src=src, group=self.group)
        return flat

    def all_agree(self, flag):
        self._flag.fill_(1 if flag else 0)
        self._all_reduce(self._flag, dist.ReduceOp.MIN)
        return bool(self._flag.item())

    def agree_max(self, code):
        self._flag.fill_(int(code))
        self._all_reduce(self._flag, dist.ReduceOp.MAX)
        return int(self._flag.item())

    def verify_ranks(self):
        """Proof that `world_size` DISTINCT ranks take part in this group's collectives: MIN / MAX / SUM
        all-reduces of the rank id, and an all-gather of (rank, device index, device uuid hash).
        Returns JSON-friendly facts (bench.py prints them; the launcher test asserts on them)."""
        red = self.device if self.backend == "nccl" else torch.device("cpu")
        me = torch.tensor([self.rank], dtype=torch.int64, device=red)
        lo, hi, tot = me.clone(), me.clone(), me.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        dev_index, uuid = -1, 0
        if self.device.type == "cuda":
            dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            props = torch.cuda.get_device_properties(dev_index)
            ident = str(getattr(props, "uuid", "")) or "%s/%d" % (props.name, dev_index)
            uuid = int.from_bytes(ident.encode()[-7:], "little") if ident else 0
        mine = torch.tensor([self.rank, dev_index, uuid, os.getpid()], dtype=torch.int64, device=red)
        rows = [torch.zeros_like(mine) for _ in range(self.world_size)]
        dist.all_gather(rows, mine, group=self.group)
        rows = [[int(v) for v in r.tolist()] for r in rows]
        n = self.world_size
        return {"rccl_ranks": dist.get_world_size(group=self.group), "collective_backend": self.backend,
                "rank_id_min_max_sum": [int(lo.item()), int(hi.item()), int(tot.item())],
                "ranks_distinct": (sorted(r[0] for r in rows) == list(range(n)) and int(lo.item()) == 0
                                   and int(hi.item()) == n - 1 and int(tot.item()) == n * (n - 1) // 2
                                   and len(set(r[3] for r in rows)) == n),
                "rank_devices": [{"rank": r[0], "device_index": r[1], "device_uuid_hash": r[2], "pid": r[3]}
                                 for r in rows],
                "devices_distinct": len(set((r[1], r[2]) for r in rows)) == n}

    def barrier(self):
        if self.device.type == "cuda" and self.backend != "gloo":
            dist.barrier(group=self.group, device_ids=[self.device.index if self.device.index is not None else 0])
        else:
            dist.barrier(group=self.group)

    def close(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()


class FakeComm(SingleComm):
    """W logical ranks inside one process.  Each rank registers its tensor for a
    collective; the last arrival performs it for everybody.  Used by unit tests
    to check DP == large-batch equivalence without a process group."""

    class _World(object):
        def __init__(self, world_size):
            self.world_size = world_size
            self.pending = {}

    def __init__(self, world, rank):
        self.world, self.rank, self.world_size = world, rank, world.world_size

    @classmethod
    def create(cls, world_size):
        w = cls._World(world_size)
        return [cls(w, r) for r in range(world_size)]

    def _collect(self, key, tensor, fn):
        slot = self.world.pending.setdefault(key, [])
        slot.append(tensor)
        if len(slot) == self.world_size:
            fn(slot)
            del self.world.pending[key]

    def all_reduce_mean_(self, flat, async_op=False):
        def fn(ts):
            total = torch.stack([t.detach().clone() for t in ts]).sum(0) / len(ts)
            for t in ts:
                t.copy_(total)
        self._collect("arm", flat, fn)

    def all_reduce_sum_(self, flat):
        def fn(ts):
            total = torch.stack([t.detach().clone() for t in ts]).sum(0)
            for t in ts:
                t.copy_(total)
        self._collect("ars", flat, fn)
        return flat

    def broadcast_(self, flat, src=0):
        def fn(ts):
            for t in ts:
                t.copy_(ts[src])
        self._collect("bc", flat, fn)     # ranks must call in rank order
        return flat

    def agree_max(self, code):
        return int(code)        # (logical ranks of one process: nothing to disagree about)


def make_comm(rank, world_size, device, port=None, backend=None, extra_ranks=0):
    """Learner rank `rank` of `world_size`; `extra_ranks` non-learner processes (inference server)
    share the process group for weight broadcasts (they call make_comm with rank >= world_size)."""
    if world_size + extra_ranks <= 1:
        return SingleComm()
    port = port or int(os.environ.get("PDA_DIST_PORT", "29533"))
    # a posted weight broadcast legitimately waits as long as the learner has nothing to publish
    return TorchComm(rank, world_size, device, init_method="tcp://127.0.0.1:%d" % port, backend=backend,
                     extra_ranks=extra_ranks, timeout_s=86400 if extra_ranks else 600)


def comm_from_env(device, backend=None, force=False):
    """Build a communicator from torchrun's RANK/WORLD_SIZE/MASTER_* env.  `force` builds a real
    process group even for one rank (rehearses the RCCL path on a single GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1 and not force:
        return SingleComm()
    init = None
    if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
        init = "tcp://127.0.0.1:%d" % int(os.environ.get("PDA_DIST_PORT", "29533"))
    return TorchComm(rank, max(world, 1), device, init_method=init, backend=backend)
