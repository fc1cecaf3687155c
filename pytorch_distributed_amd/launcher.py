"""Process orchestration (reference main.py:12-118): discover shapes from a
dummy env, build the shared replay / shared model / log structs, spawn
1 logger + N actors + L learners + 1 evaluator (mode=1) or a tester (mode=2)
with the reference's process-index layout (0 logger, 1..N actors, N+1..N+L
learners, N+L+1 evaluator, N+L+2 tester), then supervise.

Added: device selection (no hard-coded 'cuda'), learner ranks = one process per
GPU joined in an RCCL group, an optional batched inference-server process, and a
supervisor loop -- the reference just join()s, so a dead learner leaves every
other process spinning forever.  Here dead actors are restarted
and a dead learner raises the shared stop flag.
"""
import os
import time

import torch
import torch.multiprocessing as mp

from .utils.checkpoint import load_state_dict
from .parallel.inference_server import resolve_placement
from .utils.param_store import share_model


def prepare_options(opt):
    """Fill in the env-derived shapes (reference main.py:22-31, 36-41)."""
    from .utils.factory import EnvsDict
    env_prototype = EnvsDict[opt.env_type]
    dummy_env = env_prototype(opt.env_params, 0)
    opt.norm_val = dummy_env.norm_val
    opt.state_shape = dummy_env.state_shape
    opt.action_shape = dummy_env.action_shape
    opt.action_space = dummy_env.action_space
    opt.reward_shape = opt.agent_params.num_tasks
    opt.gamma_shape = opt.agent_params.num_tasks
    opt.terminal_shape = opt.agent_params.num_tasks
    del dummy_env
    mp_ = opt.memory_params
    mp_.norm_val = opt.norm_val
    mp_.state_shape = opt.state_shape
    mp_.action_shape = opt.action_shape
    mp_.reward_shape = opt.reward_shape
    mp_.gamma_shape = opt.gamma_shape
    mp_.terminal_shape = opt.terminal_shape
    return opt


def build_global_model(opt):
    from .utils.factory import ModelsDict
    model = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape,
                                       opt.action_space, opt.action_shape)
    if opt.model_file is not None:      # finetune from model_file
        model.load_state_dict(load_state_dict(opt.model_file))
    return share_model(model)


class Supervisor(object):
    """Keeps the process table.  The reference just join()s (main.py:117-118): a crashed actor goes unnoticed and a
    crashed learner leaves every other process spinning on `learner_step < steps` forever.  Here:

    * a dead actor is restarted (actors are stateless); a dead learner / inference process raises the shared stop
      flag and the job ends with that exit code;
    * heartbeats are CONSUMED: every worker stamps its slot of `global_logs.heartbeats` about once a second; a
      learner / inference process whose stamp is older than `watchdog_learner_s` is treated as hung (stop flag, then
      terminated), a stale actor is terminated and restarted;
    * shutdown is bounded: once the job is over -- steps reached, stop flag, or a fatal exit -- the remaining
      processes get `shutdown_grace_s` seconds and are then terminated (a learner rank blocked in a collective
      whose peer died never looks at the stop flag again) and, if need be, killed."""

    FATAL_ROLES = ("learner", "inference")

    def __init__(self, opt, global_logs, ctx):
        self.opt, self.global_logs, self.ctx = opt, global_logs, ctx
        self.procs = []     # [role, index, (target, args), Process]
        self.restarts = 0
        self.hung = []      # (role, index) of processes the watchdog gave up on
        self.failed = None  # (role, index, exitcode | "hung") of the first fatal failure
        self.started_at = {}
        self.watchdog_learner_s = float(getattr(opt, "watchdog_learner_s", 120.0))
        self.watchdog_actor_s = float(getattr(opt, "watchdog_actor_s", 120.0))
        self.shutdown_grace_s = float(getattr(opt, "shutdown_grace_s", 20.0))

    def start(self, role, index, target, args):
        p = self.ctx.Process(target=target, args=args, name="%s-%d" % (role, index))
        p.start()
        self.started_at[(role, index)] = time.time()
        self.procs.append([role, index, (target, args), p])
        return p

    def _restart(self, entry, why):
        role, index, (target, args), p = entry
        print("[supervisor] %s %d %s: restarting" % (role, index, why), flush=True)
        if p.is_alive():
            p.terminate()
            p.join(5)
            if p.is_alive():
                p.kill()
        if self.global_logs is not None:
            self.global_logs.beat(role, index)      # the new incarnation gets a full period to show up
        np_ = self.ctx.Process(target=target, args=args, name="%s-%d" % (role, index))
        np_.start()
        self.started_at[(role, index)] = time.time()
        entry[3] = np_
        self.restarts += 1

    def _stale(self, role, index, limit):
        """True when the process has beaten before (or has had `limit` seconds to start) and its last beat is
        older than `limit` seconds."""
        if limit <= 0 or self.global_logs is None:
            return False
        last = self.global_logs.last_beat(index)
        ref = last if last > 0 else self.started_at.get((role, index), time.time())
        return time.time() - ref > limit

    def poll(self):
        """One supervision pass.  Returns False when everything has exited."""
        alive = False
        steps = self.opt.agent_params.steps
        gl = self.global_logs
        for entry in self.procs:
            role, index, _, p = entry
            running = gl is not None and gl.running(steps)
            if p.is_alive():
                alive = True
                if not running or not self.opt.supervise:
                    continue
                if role == "actor" and self._stale(role, index, self.watchdog_actor_s):
                    self._restart(entry, "has not beaten for %.0f s (hung)" % self.watchdog_actor_s)
                elif role in self.FATAL_ROLES and self._stale(role, index, self.watchdog_learner_s):
                    print("[supervisor] %s %d has not beaten for %.0f s (hung): stopping the job"
                          % (role, index, self.watchdog_learner_s), flush=True)
                    self.hung.append((role, index))
                    self.failed = self.failed or (role, index, "hung")
                    gl.stop.value = True
                continue
            if p.exitcode not in (0, None) and running:
                if role == "actor" and self.opt.supervise:
                    self._restart(entry, "died (exit %s)" % p.exitcode)
                    alive = True
                elif role in self.FATAL_ROLES:
                    print("[supervisor] %s %d died (exit %s): stopping the job" % (role, index, p.exitcode), flush=True)
                    self.failed = self.failed or (role, index, p.exitcode)
                    gl.stop.value = True
        return alive

    def join(self, poll_s=0.5):
        """Supervise until the job is over, then give what is left a bounded time to leave."""
        steps = self.opt.agent_params.steps
        deadline = None
        while self.poll():
            if deadline is None and self.global_logs is not None and not self.global_logs.running(steps):
                deadline = time.time() + self.shutdown_grace_s
            if deadline is not None and time.time() > deadline:
                break
            time.sleep(poll_s)
        for role, index, _, p in self.procs:
            if p.is_alive():
                print("[supervisor] %s %d still running %.0f s after the end of the job: terminating"
                      % (role, index, self.shutdown_grace_s), flush=True)
                p.terminate()
        t_end = time.time() + 10
        for _, _, _, p in self.procs:
            p.join(max(0.1, t_end - time.time()))
            if p.is_alive():
                p.kill()
                p.join(5)
        for role, index, _, p in self.procs:       # a learner that left with an error after raising the stop flag itself
            if role in self.FATAL_ROLES and p.exitcode not in (0, None) and self.failed is None:
                self.failed = (role, index, p.exitcode)
        return [p.exitcode for _, _, _, p in self.procs]

    @property
    def ok(self):
        """The job ran to its end: no fatal failure, no watchdog kill."""
        return self.failed is None and not self.hung


def run(opt):
    """Launch the configured job and block until it finishes.  Returns the
    Supervisor (exit codes, restart count) for tests."""
    from .utils.factory import (GlobalLogsDict, ActorLogsDict, LearnerLogsDict, EvaluatorLogsDict,
                                LoggersDict, ActorsDict, LearnersDict, EvaluatorsDict, TestersDict,
                                EnvsDict, MemoriesDict, ModelsDict)
    ctx = mp.get_context("spawn")
    torch.manual_seed(opt.seed)
    prepare_options(opt)
    env_prototype = EnvsDict[opt.env_type]
    memory_prototype = MemoriesDict[opt.memory_type]
    model_prototype = ModelsDict[opt.model_type]
    os.makedirs(os.path.dirname(opt.model_name), exist_ok=True)
    os.makedirs(opt.log_dir, exist_ok=True)

    if opt.mode == 2:
        sup = Supervisor(opt, None, ctx)
        sup.start("tester", opt.num_actors + opt.num_learners + 2, TestersDict[opt.agent_type],
                  (opt.num_actors + opt.num_learners + 2, opt, env_prototype, model_prototype))
        for _, _, _, p in sup.procs:
            p.join()
        return sup

    from .parallel.weight_sync import resolve_weight_sync
    if (opt.num_learners > 1 or resolve_weight_sync(opt) == "collective") and not hasattr(opt, "dist_port"):
        opt.dist_port = 29500 + (os.getpid() % 2000)
    global_memory = memory_prototype(opt.memory_params)
    global_model = build_global_model(opt)
    global_optimizer = None     # kept for signature parity (reference main.py:49)
    global_logs = GlobalLogsDict[opt.agent_type]()
    actor_logs = ActorLogsDict[opt.agent_type]()
    learner_logs = LearnerLogsDict[opt.agent_type]()
    evaluator_logs = EvaluatorLogsDict[opt.agent_type]()

    if opt.actor_inference == "batched":
        from .parallel.inference_server import InferenceMailbox
        global_model.mailbox = InferenceMailbox(opt.num_actors * max(1, int(opt.num_envs_per_actor)), opt.state_shape,
                                                global_memory.state_dtype if hasattr(global_memory, "state_dtype")
                                                else torch.float32,
                                                opt.action_shape, discrete=(opt.agent_type == "dqn"))

    sup = Supervisor(opt, global_logs, ctx)
    sup.start("logger", 0, LoggersDict[opt.agent_type],
              (0, opt, global_logs, actor_logs, learner_logs, evaluator_logs))
    for i in range(opt.num_actors):
        sup.start("actor", i + 1, ActorsDict[opt.agent_type],
                  (i + 1, opt, global_logs, actor_logs, env_prototype, model_prototype, global_memory, global_model))
    for i in range(opt.num_learners):
        ind = opt.num_actors + i + 1
        sup.start("learner", ind, LearnersDict[opt.agent_type],
                  (ind, opt, global_logs, learner_logs, model_prototype, global_memory, global_model, global_optimizer))
    ind = opt.num_actors + opt.num_learners + 1
    sup.start("evaluator", ind, EvaluatorsDict[opt.agent_type],
              (ind, opt, global_logs, evaluator_logs, env_prototype, model_prototype, global_model))
    if opt.actor_inference == "batched" and resolve_placement(opt) == "process":
        from .parallel.inference_server import inference_server
        sup.start("inference", ind + 2, inference_server,
                  (ind + 2, opt, global_logs, model_prototype, global_model))
    sup.exitcodes = sup.join()
    return sup
