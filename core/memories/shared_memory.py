"""Alias of pytorch_distributed_amd.memories.shared_memory (reference import path core/memories/shared_memory.py)."""
from pytorch_distributed_amd.memories.shared_memory import *  # noqa: F401,F403
from pytorch_distributed_amd.memories.shared_memory import SharedMemory  # noqa: F401
