"""Alias of pytorch_distributed_amd.envs.env (reference import path core/env.py)."""
from pytorch_distributed_amd.envs.env import *  # noqa: F401,F403
from pytorch_distributed_amd.envs.env import Env  # noqa: F401
