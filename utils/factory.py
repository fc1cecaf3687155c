"""Alias of pytorch_distributed_amd.utils.factory (reference import path utils/factory.py)."""
from pytorch_distributed_amd.utils.factory import *  # noqa: F401,F403
