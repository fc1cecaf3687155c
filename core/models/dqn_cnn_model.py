"""Alias of pytorch_distributed_amd.models.dqn_cnn_model (reference import path core/models/dqn_cnn_model.py)."""
from pytorch_distributed_amd.models.dqn_cnn_model import *  # noqa: F401,F403
from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel  # noqa: F401
