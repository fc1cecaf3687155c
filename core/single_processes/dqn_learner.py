"""Alias of pytorch_distributed_amd.agents.learners (reference import path core/single_processes/dqn_learner.py)."""
from pytorch_distributed_amd.agents.learners import *  # noqa: F401,F403
from pytorch_distributed_amd.agents.learners import dqn_learner  # noqa: F401
