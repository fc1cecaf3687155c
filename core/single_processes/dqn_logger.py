"""Alias of pytorch_distributed_amd.agents.loggers (reference import path core/single_processes/dqn_logger.py)."""
from pytorch_distributed_amd.agents.loggers import *  # noqa: F401,F403
from pytorch_distributed_amd.agents.loggers import dqn_logger  # noqa: F401
