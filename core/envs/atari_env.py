"""Alias of pytorch_distributed_amd.envs.atari_env (reference import path core/envs/atari_env.py)."""
from pytorch_distributed_amd.envs.atari_env import *  # noqa: F401,F403
from pytorch_distributed_amd.envs.atari_env import AtariEnv, make_atari_env  # noqa: F401
