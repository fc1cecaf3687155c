"""Small shared helpers: the Experience record and target-network updates.

Capability parity with reference utils/helpers.py:8-34 (Experience tuple,
reset_experience, update_target_model hard/soft, ensure_global_grads).  The
target update here works on flat parameter buffers when the model exposes one
(a single fused HIP launch, see ops/optim.py) and falls back to a per-tensor
torch path on CPU.
"""
from collections import namedtuple

import torch

# A transition: from `state0`, `action` yields `reward` and lands in `state1`,
# which may be terminal.  Envs return it with state0=None (as the reference does).
Experience = namedtuple("Experience", "state0, action, reward, state1, terminal1")


def reset_experience():
    return Experience(state0=None, action=None, reward=None, state1=None, terminal1=False)


def is_soft_update(target_model_update):
    return target_model_update < 1.0


@torch.no_grad()
def update_target_model(model, target_model, target_model_update=1.0, learner_step=0):
    """Polyak average when `target_model_update` < 1, else a hard copy every
    `target_model_update` learner steps (reference utils/helpers.py:19-25)."""
    if is_soft_update(target_model_update):
        tau = float(target_model_update)
        src = list(model.parameters())
        dst = list(target_model.parameters())
        if src and src[0].is_cuda:
            from ..ops import optim as _optim
            if _optim.polyak_params_(dst, src, tau):
                return
        torch._foreach_mul_(dst, 1.0 - tau)
        torch._foreach_add_(dst, src, alpha=tau)
    elif learner_step % int(target_model_update) == 0:
        for tp, p in zip(target_model.parameters(), model.parameters()):
            tp.copy_(p)
        for tb, b in zip(target_model.buffers(), model.buffers()):
            tb.copy_(b)


def ensure_global_grads(local_model, global_model, local_device=None, global_device=torch.device("cpu")):
    """A3C-style gradient hand-off kept for API parity (reference
    utils/helpers.py:28-34, unused there too).  Data-parallel learners in this
    framework all-reduce a flat gradient bucket instead (parallel/comm.py)."""
    for lp, gp in zip(local_model.parameters(), global_model.parameters()):
        if lp.grad is None:
            continue
        gp.grad = lp.grad.detach().to(gp.device)


def flatten_state_dict(sd):
    """state_dict -> single fp32 vector (parameter order = insertion order)."""
    return torch.cat([v.detach().reshape(-1).float().cpu() for v in sd.values()])
