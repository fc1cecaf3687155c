"""Alias of pytorch_distributed_amd.models.dqn_mlp_model (reference import path core/models/dqn_mlp_model.py)."""
from pytorch_distributed_amd.models.dqn_mlp_model import *  # noqa: F401,F403
from pytorch_distributed_amd.models.dqn_mlp_model import DqnMlpModel  # noqa: F401
