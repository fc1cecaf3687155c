"""Alias of pytorch_distributed_amd.agents.logs (reference import path core/single_processes/logs.py)."""
from pytorch_distributed_amd.agents.logs import *  # noqa: F401,F403

