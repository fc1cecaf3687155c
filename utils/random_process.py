"""Alias of pytorch_distributed_amd.utils.random_process (reference import path utils/random_process.py)."""
from pytorch_distributed_amd.utils.random_process import *  # noqa: F401,F403
