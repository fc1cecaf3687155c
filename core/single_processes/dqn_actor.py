"""Alias of pytorch_distributed_amd.agents.actors (reference import path core/single_processes/dqn_actor.py)."""
from pytorch_distributed_amd.agents.actors import *  # noqa: F401,F403
from pytorch_distributed_amd.agents.actors import dqn_actor  # noqa: F401
