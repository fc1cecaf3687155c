"""Name -> class/function registries, same twelve dict names as the reference
(utils/factory.py:5-43), with more entries: DqnMlpModel is registered, there are
continuous-control / synthetic / toy envs, and HBM-resident replay memories.
"""
from ..agents.logs import GlobalLogs, ActorLogs, DQNLearnerLogs, DDPGLearnerLogs, EvaluatorLogs
from ..agents.loggers import dqn_logger, ddpg_logger
from ..agents.actors import dqn_actor, ddpg_actor
from ..agents.learners import dqn_learner, ddpg_learner
from ..agents.evaluators import evaluator
from ..agents.testers import tester
from ..envs.atari_env import make_atari_env, AtariEnv
from ..envs.synthetic_env import SyntheticAtariEnv, make_synthetic_env
from ..envs.pendulum_env import PendulumEnv, ChainEnv, CountingEnv
from ..memories.shared_memory import SharedMemory
from ..memories.device_memory import DeviceMemory
from ..memories.dedup_memory import DedupDeviceMemory
from ..models.dqn_cnn_model import DqnCnnModel
from ..models.dqn_mlp_model import DqnMlpModel
from ..models.ddpg_mlp_model import DdpgMlpModel

GlobalLogsDict = {"dqn": GlobalLogs, "ddpg": GlobalLogs}
ActorLogsDict = {"dqn": ActorLogs, "ddpg": ActorLogs}
LearnerLogsDict = {"dqn": DQNLearnerLogs, "ddpg": DDPGLearnerLogs}
EvaluatorLogsDict = {"dqn": EvaluatorLogs, "ddpg": EvaluatorLogs}

LoggersDict = {"dqn": dqn_logger, "ddpg": ddpg_logger}
ActorsDict = {"dqn": dqn_actor, "ddpg": ddpg_actor}
LearnersDict = {"dqn": dqn_learner, "ddpg": ddpg_learner}
EvaluatorsDict = {"dqn": evaluator, "ddpg": evaluator}
TestersDict = {"dqn": tester, "ddpg": tester}

EnvsDict = {"atari": make_atari_env,        # ALE when installed, synthetic frames otherwise
            "atari-ale": AtariEnv,          # ALE or ImportError
            "synthetic": make_synthetic_env,    # noise frames | game="catch" | game="minipong"
            "pendulum": PendulumEnv,
            "chain": ChainEnv,
            "counting": CountingEnv}

MemoriesDict = {"shared": SharedMemory,         # host shm ring, uniform
                "shared-per": SharedMemory,     # host shm ring + sum/min tree
                "device": DeviceMemory,         # HBM-resident ring, uniform
                "device-per": DeviceMemory,     # HBM-resident ring + HIP segment tree
                "device-dedup": DedupDeviceMemory,      # HBM-resident, every frame stored once (memories/dedup_memory.py)
                "device-dedup-per": DedupDeviceMemory,  # ... + HIP segment tree
                "none": None}

ModelsDict = {"dqn-cnn": DqnCnnModel,
              "dqn-mlp": DqnMlpModel,
              "ddpg-mlp": DdpgMlpModel}
