"""Alias of pytorch_distributed_amd.agents.actors (reference import path core/single_processes/ddpg_actor.py)."""
from pytorch_distributed_amd.agents.actors import *  # noqa: F401,F403
from pytorch_distributed_amd.agents.actors import ddpg_actor  # noqa: F401
