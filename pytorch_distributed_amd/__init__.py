"""pytorch_distributed_amd: an MI355X-native distributed deep-RL framework.

Distributed DQN (Ape-X style, with finished prioritized replay) and Distributed
DDPG (D4PG style), re-designed for one-process-per-GPU learners on MI355X
(gfx950): HBM-resident replay, hand-written HIP/MFMA kernels for the hot ops,
RCCL data-parallel learners, and a batched GPU actor-inference server.

Public API mirrors the reference's factory/options layout (utils.options,
utils.factory, main.py mode=1/2); see SURVEY.md for the capability map.
"""

__version__ = "0.1.0"
