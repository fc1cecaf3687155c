"""Alias of pytorch_distributed_amd.agents.evaluators (reference import path core/single_processes/evaluators.py)."""
from pytorch_distributed_amd.agents.evaluators import *  # noqa: F401,F403
from pytorch_distributed_amd.agents.evaluators import evaluator  # noqa: F401
