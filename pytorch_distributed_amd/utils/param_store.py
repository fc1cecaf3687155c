"""Learner -> actor/evaluator weight hand-off through host shared memory.

The reference shares one CUDA-resident nn.Module with every process through
CUDA-IPC handles and lets readers copy out of it while Adam writes into it
(main.py:44-46, dqn_actor.py:177-178, dqn_learner.py:87): torn snapshots are
possible.  Here `global_model` is a CPU module whose parameters are views of
ONE flat shared-memory buffer guarded by a seqlock version counter: the learner
publishes a whole flat snapshot (one D2H copy from its flat HBM buffer), readers
retry until they see an even, unchanged version.  No GPU IPC handles cross
process boundaries.  Multi-GPU inference ranks receive the same flat buffer by
RCCL broadcast instead (parallel/comm.py).
"""
import time

import torch
from . import mpctx as mp


def share_model(model):
    """Flatten `model` (CPU) into one shared-memory buffer and attach a
    version counter.  Returns the model (now the `global_model`)."""
    model.cpu()
    flat = model.flatten_(with_grad=False)
    flat.share_memory_()
    for p in model.parameters():
        p.requires_grad_(False)
    model.param_version = mp.Value("l", 0)
    return model


@torch.no_grad()
def publish_params(global_model, source):
    """source: an nn.Module with the same parameter layout, or a flat tensor."""
    flat_src = source if torch.is_tensor(source) else None
    ver = global_model.param_version
    with ver.get_lock():
        ver.value += 1                      # odd: write in progress
        if flat_src is not None:
            global_model.flat_params.copy_(flat_src.detach().reshape(-1), non_blocking=False)
        else:
            for gp, p in zip(global_model.parameters(), source.parameters()):
                gp.copy_(p.detach())
        ver.value += 1                      # even: snapshot complete


class AsyncParamPublisher(object):
    """Non-blocking publish for a learner whose parameters live on a GPU.

    publish_params() is a synchronous 6.75 MB D2H copy into pageable shared memory: ~1.5 ms of
    learner stall per call.  Here the learner thread only enqueues a D2H copy into a pinned
    buffer on a side stream; a helper thread waits for that copy and moves the snapshot into the
    seqlocked shared buffer.  If the previous snapshot is still in flight the call is skipped
    (the next one carries newer weights anyway); `min_interval` bounds the publish rate."""

    def __init__(self, global_model, device, min_interval=0.02):
        import threading
        self.global_model = global_model
        self.device = torch.device(device)
        self.min_interval = float(min_interval)
        self.last = 0.0
        self.cuda = self.device.type == "cuda"
        n = global_model.flat_params.numel()
        self.pinned = torch.zeros(n, dtype=torch.float32)
        self.busy = False
        self.published = 0
        self.skipped = 0
        if self.cuda:
            self.pinned = self.pinned.pin_memory()
            self.stream = torch.cuda.Stream(device=self.device)
            self.event = torch.cuda.Event(enable_timing=True)
            self.event0 = torch.cuda.Event(enable_timing=True)
            self.copy_ms = 0.0
            self.cv = threading.Condition()
            self.stopping = False
            self.thread = threading.Thread(target=self._run, name="param-publisher", daemon=True)
            self.thread.start()

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            with self.cv:
                while not self.busy and not self.stopping:
                    self.cv.wait()
                if self.stopping and not self.busy:
                    return
            # (polling, not event.synchronize(): a thread parked inside the runtime's blocking wait while the learner's
            # step graphs are replaying stretched those replays 2.8x -- 8.3 k instead of 9.5 k updates/s inside the job,
            # profiles/r3_node_learner_loop.txt)
            while not self.event.query():
                time.sleep(0.0003)
            self.copy_ms += self.event0.elapsed_time(self.event)
            publish_params(self.global_model, self.pinned)
            with self.cv:
                self.busy = False
                self.published += 1
                self.cv.notify_all()

    def publish(self, flat, force=False):
        """Returns True when a snapshot was taken."""
        if not self.cuda:
            publish_params(self.global_model, flat)
            self.published += 1
            return True
        now = time.time()
        with self.cv:
            if force:
                while self.busy:
                    self.cv.wait()
            elif self.busy or now - self.last < self.min_interval:
                self.skipped += 1
                return False
            self.last = now
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                self.event0.record(self.stream)
                self.pinned.copy_(flat.detach().reshape(-1), non_blocking=True)
                self.event.record(self.stream)
            self.busy = True
            self.cv.notify_all()
        return True

    def close(self):
        """Waits for the last snapshot to land in the shared buffer."""
        if not self.cuda:
            return
        with self.cv:
            while self.busy:
                self.cv.wait()
            self.stopping = True
            self.cv.notify_all()
        self.thread.join(10)
        print("[publisher] %d snapshots (%d requests skipped), %.3f ms per device-to-host copy"
              % (self.published, self.skipped, self.copy_ms / max(self.published, 1)), flush=True)


@torch.no_grad()
def sync_params(local_model, global_model, max_tries=50):
    """Copy a consistent snapshot of global_model into local_model; returns the
    version that was read."""
    ver = getattr(global_model, "param_version", None)
    for _ in range(max_tries):
        v0 = ver.value if ver is not None else 0
        if v0 % 2 == 1:
            time.sleep(0.0005)
            continue
        for lp, gp in zip(local_model.parameters(), global_model.parameters()):
            lp.copy_(gp)
        if ver is None or ver.value == v0:
            return v0
    return -1   # gave up on a clean snapshot; weights are usable but may be torn
