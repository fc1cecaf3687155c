"""Alias of pytorch_distributed_amd.utils.options (reference import path utils/options.py)."""
from pytorch_distributed_amd.utils.options import *  # noqa: F401,F403
from pytorch_distributed_amd.utils.options import CONFIGS, PRESETS, Options, Params, EnvParams, MemoryParams, ModelParams, AgentParams  # noqa: F401
