"""Shared counters/accumulators the worker processes exchange with the logger
(reference core/single_processes/logs.py:3-32: GlobalLogs, ActorLogs,
DQNLearnerLogs, DDPGLearnerLogs, EvaluatorLogs -- all mp.Value).

GlobalLogs additionally carries a `stop` flag (set by the supervisor when a
learner dies, so nobody spins forever on `learner_step < steps`), one heartbeat
slot per process that the supervisor reads (launcher.py: a stale learner stops
the job, a stale actor is restarted), and its hot counters are sharded: the
reference takes an mp.Value lock for every env step of every actor
(dqn_actor.py:166-167), which serialises 256 envs on one semaphore; here every
process adds to a cache line of its own and readers sum the lines.
"""
import time

from ..utils import mpctx as mp

MAX_PROCS = 1024        # heartbeat slots / counter shards (process_ind of the launcher's process table)
_STRIDE = 8             # longs per shard: one 64-byte line each, no false sharing


class ShardedCounter(object):
    """A shared integer with the surface the reference uses (`.value`, `.get_lock()`) whose increments need no
    lock: `add(shard, n)` touches only the caller's own cache line, `.value` sums all lines.  Assigning `.value`
    (resume) rewrites the total; do that only while nobody is adding."""

    def __init__(self, shards=MAX_PROCS):
        self.shards = int(shards)
        self._cells = mp.RawArray("l", self.shards * _STRIDE)
        self._lock = mp.Lock()          # only for API parity: `with counter.get_lock(): counter.value += 1`

        self._view = None               # numpy view of the cells, rebuilt in every process

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_view"] = None
        return d

    def _np(self):
        if self._view is None:
            import numpy as np
            self._view = np.frombuffer(self._cells, dtype=np.int64)[::_STRIDE]
        return self._view

    def add(self, shard, n=1):
        self._np()[int(shard) % self.shards] += n

    @property
    def value(self):
        return int(self._np().sum())

    @value.setter
    def value(self, total):
        # shard 0 absorbs a set / a locked `+=` (the reference's idiom keeps working)
        self._np()[0] += int(total) - self.value

    def get_lock(self):
        return self._lock


class PlainValue(object):
    """A shared scalar with mp.Value's surface (`.value`, `.get_lock()`) whose READS take no lock: an aligned 8-byte
    load is atomic, and there is one writer at a time (learner rank 0 for `learner_step`; whoever raises `stop` only
    ever stores True).  `learner_step` used to be an mp.Value: every `.value` read acquires its semaphore, every
    actor reads it once per env step (`running()`), 100 k times a second in all, and the learner's own
    `with get_lock(): value += 1` queued behind them -- ~100 us per update, a quarter of the learner's rate inside
    the job (profiles/r3_node_learner_loop.txt).  The lock still exists for writers that want the reference's idiom."""

    def __init__(self, typecode, initial):
        self._raw = mp.RawValue(typecode, initial)
        self._lock = mp.Lock()

    @property
    def value(self):
        return self._raw.value

    @value.setter
    def value(self, v):
        self._raw.value = v

    def get_lock(self):
        return self._lock


class GlobalLogs(object):
    def __init__(self):
        self.actor_step = ShardedCounter()
        self.learner_step = PlainValue("l", 0)
        self.stop = PlainValue("b", False)
        self.frames_ingested = ShardedCounter(64)
        self.heartbeats = mp.RawArray("d", MAX_PROCS)       # time.time() of each process's last sign of life

    def running(self, max_steps):
        return (not self.stop.value) and self.learner_step.value < max_steps

    def beat(self, role, process_ind=0):
        """Mark process `process_ind` alive (role kept for readability at the call sites)."""
        self.heartbeats[int(process_ind) % MAX_PROCS] = time.time()

    def last_beat(self, process_ind):
        return self.heartbeats[int(process_ind) % MAX_PROCS]


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
