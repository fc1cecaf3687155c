"""Alias of pytorch_distributed_amd.models.model (reference import path core/model.py)."""
from pytorch_distributed_amd.models.model import *  # noqa: F401,F403
from pytorch_distributed_amd.models.model import Model  # noqa: F401
