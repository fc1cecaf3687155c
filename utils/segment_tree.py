"""Alias of pytorch_distributed_amd.utils.segment_tree (reference import path utils/segment_tree.py)."""
from pytorch_distributed_amd.utils.segment_tree import *  # noqa: F401,F403
