"""Alias of pytorch_distributed_amd.agents.testers (reference import path core/single_processes/testers.py)."""
from pytorch_distributed_amd.agents.testers import *  # noqa: F401,F403
from pytorch_distributed_amd.agents.testers import tester  # noqa: F401
