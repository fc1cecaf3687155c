"""Learner -> inference-server weight distribution as a collective broadcast.

The reference lets every actor copy out of one CUDA-IPC-shared model while Adam
writes into it (dqn_actor.py:177-178, dqn_learner.py:87).  With batched actor
inference there is exactly ONE consumer of fresh weights per inference GPU, so the
hand-off is one broadcast of the flat parameter buffer from learner rank 0 to the
inference rank(s) every `param_sync_freq` learner steps: RCCL over xGMI when the
inference server owns a GPU of its own, gloo on the CPU plumbing config.  (When
the server shares the learner's GPU the hand-off is either the host seqlock buffer,
utils/param_store.py, or nothing at all: inference_placement="learner" reads the
learner's live shadow weights.)

Wire format: one fp32 buffer [version, params...]; version < 0 is the shutdown
marker.  The publisher never blocks the learner (async collective on the comm
stream, double-buffered); the receiver is a thread that always has one broadcast
posted, so the two sides need no step alignment.
"""
import threading

import torch

try:
    import torch.distributed as dist
except Exception:   # pragma: no cover
    dist = None


class WeightPublisher(object):
    """Learner rank 0.  `comm` is the learners' TorchComm built with extra_ranks > 0."""

    def __init__(self, comm, numel, device):
        self.comm, self.group = comm, comm.weight_group
        self.bufs = [torch.zeros(numel + 1, dtype=torch.float32, device=device) for _ in range(2)]
        self.works = [None, None]
        self.count = 0
        self.cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None

    def _send(self, i):
        self.works[i] = dist.broadcast(self.bufs[i], src=0, group=self.group, async_op=True)

    def publish(self, flat):
        i = self.count & 1
        if self.works[i] is not None:
            self.works[i].wait()            # the buffer's previous broadcast has left
        self.count += 1
        buf = self.bufs[i]
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                buf[1:].copy_(flat.detach().reshape(-1), non_blocking=True)
                buf[:1].fill_(float(self.count))
                self._send(i)
        else:
            buf[1:].copy_(flat.detach().reshape(-1))
            buf[0] = float(self.count)
            self._send(i)

    def close(self):
        for w in self.works:
            if w is not None:
                w.wait()
        self.bufs[0][:1].fill_(-1.0)
        self._send(0)
        self.works[0].wait()
        if self.cuda:
            torch.cuda.synchronize()


class WeightReceiver(object):
    """Inference rank: a thread that keeps one broadcast posted and stores the newest snapshot."""

    def __init__(self, comm, numel, device):
        self.comm, self.group = comm, comm.weight_group
        self.device = torch.device(device)
        self.buf = torch.zeros(numel + 1, dtype=torch.float32, device=device)
        self.latest = torch.zeros(numel, dtype=torch.float32, device=device)
        self.version = 0
        self.applied = 0
        self.closed = False
        self.lock = threading.Lock()
        self.thread = threading.Thread(target=self._run, name="weight-receiver", daemon=True)
        self.thread.start()

    def _run(self):
        cuda = self.device.type == "cuda"
        if cuda:
            torch.cuda.set_device(self.device)
        while True:
            dist.broadcast(self.buf, src=0, group=self.group)
            if cuda:
                torch.cuda.current_stream().synchronize()
            v = float(self.buf[0].item())
            if v < 0:
                self.closed = True
                return
            with self.lock:
                self.latest.copy_(self.buf[1:])
                if cuda:
                    torch.cuda.current_stream().synchronize()
                self.version = int(v)

    def poll_into(self, flat_params):
        """Copy the newest snapshot into `flat_params` if there is one; returns True when it did."""
        if self.version == self.applied:
            return False
        with self.lock:
            flat_params.copy_(self.latest)
            self.applied = self.version
        return True

    def join(self, timeout=None):
        self.thread.join(timeout)


def resolve_weight_sync(args):
    """"shm": seqlocked host buffer (utils/param_store.py); "collective": broadcast over the
    learners' process group (RCCL on GPUs, gloo on CPU).  auto -> collective when the inference
    server runs as its own process on a GPU that no learner uses."""
    mode = getattr(args, "weight_sync", "auto")
    if mode != "auto":
        return mode
    if args.actor_inference != "batched" or args.device != "cuda":
        return "shm"
    from .inference_server import resolve_placement
    if resolve_placement(args) != "process":
        return "shm"
    try:
        ngpu = torch.cuda.device_count()
    except Exception:
        ngpu = 0
    return "collective" if ngpu > args.num_learners else "shm"


def inference_device(args):
    """Device of the inference-server process: the first GPU after the learners' when weights
    arrive by RCCL broadcast (two RCCL ranks cannot share a GPU), else the first learner's GPU."""
    if args.device != "cuda":
        return torch.device(args.device)
    if resolve_weight_sync(args) == "collective":
        return torch.device("cuda:%d" % ((args.gpu_ind + args.num_learners) % max(torch.cuda.device_count(), 1)))
    return torch.device("cuda:%d" % args.gpu_ind)
