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
