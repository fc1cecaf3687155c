"""Checkpoint I/O.  Format is the reference's: a plain `state_dict`
(OrderedDict[str, Tensor]) torch.save'd to ./models/MACHINE_TIMESTAMP.pth
(reference evaluators.py:97-100, main.py:45, testers.py:25), with identical
keys/shapes so files move between the two code bases.  Written as CPU tensors,
atomically (tmp + rename), and loaded with map_location="cpu",
weights_only=True so a file saved on a GPU box opens anywhere.

An optional sidecar `<name>.resume.pth` carries optimizer moments, target net
and step counters for true resume (the reference can only warm-start).
"""
import os

import torch


def save_state_dict(model_or_sd, path):
    sd = model_or_sd.state_dict() if hasattr(model_or_sd, "state_dict") else model_or_sd
    cpu_sd = type(sd)((k, v.detach().to("cpu", copy=True).contiguous()) for k, v in sd.items())
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp.%d" % os.getpid()
    torch.save(cpu_sd, tmp)
    os.replace(tmp, path)
    return path


def load_state_dict(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def save_resume(path, **tensors_and_scalars):
    tmp = path + ".tmp.%d" % os.getpid()
    torch.save(tensors_and_scalars, tmp)
    os.replace(tmp, path)


def load_resume(path):
    return torch.load(path, map_location="cpu", weights_only=True)
