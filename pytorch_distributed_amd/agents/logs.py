"""Shared counters/accumulators the worker processes exchange with the logger
(reference core/single_processes/logs.py:3-32: GlobalLogs, ActorLogs,
DQNLearnerLogs, DDPGLearnerLogs, EvaluatorLogs -- all mp.Value).

GlobalLogs additionally carries a `stop` flag (set by the supervisor when a
learner dies, so nobody spins forever on `learner_step < steps`) and per-role
heartbeat timestamps.
"""
import time

from ..utils import mpctx as mp


class GlobalLogs(object):
    def __init__(self):
        self.actor_step = mp.Value("l", 0)
        self.learner_step = mp.Value("l", 0)
        self.stop = mp.Value("b", False)
        self.learner_heartbeat = mp.Value("d", 0.0)
        self.actor_heartbeat = mp.Value("d", 0.0)
        self.frames_ingested = mp.Value("l", 0)

    def running(self, max_steps):
        return (not self.stop.value) and self.learner_step.value < max_steps

    def beat(self, role):
        getattr(self, role + "_heartbeat").value = time.time()


class ActorLogs(object):
    def __init__(self):
        self.total_steps = mp.Value("l", 0)
        self.total_reward = mp.Value("d", 0.0)
        self.nepisodes = mp.Value("l", 0)
        self.nepisodes_solved = mp.Value("l", 0)


class DQNLearnerLogs(object):
    def __init__(self):
        self.critic_loss = mp.Value("d", 0.0)
        self.loss_counter = mp.Value("l", 0)


class DDPGLearnerLogs(object):
    def __init__(self):
        self.actor_loss = mp.Value("d", 0.0)
        self.critic_loss = mp.Value("d", 0.0)
        self.loss_counter = mp.Value("l", 0)


class EvaluatorLogs(object):
    def __init__(self):
        self.total_steps = mp.Value("l", 0)
        self.total_reward = mp.Value("d", 0.0)
        self.nepisodes = mp.Value("l", 0)
        self.nepisodes_solved = mp.Value("l", 0)
        self.logger_lock = mp.Value("b", False)
