"""Alias of pytorch_distributed_amd.memories.memory (reference import path core/memory.py)."""
from pytorch_distributed_amd.memories.memory import *  # noqa: F401,F403
from pytorch_distributed_amd.memories.memory import Memory  # noqa: F401
