"""Alias of pytorch_distributed_amd.utils.helpers (reference import path utils/helpers.py)."""
from pytorch_distributed_amd.utils.helpers import *  # noqa: F401,F403
