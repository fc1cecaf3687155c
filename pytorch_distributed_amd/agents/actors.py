"""Actor processes: step an env, assemble n-step transitions, feed the replay
(reference core/single_processes/dqn_actor.py:9-192 and ddpg_actor.py:9-172;
same signatures).

Two ways to pick actions:
* actor_inference="local"   -- a private model copy, re-synced from the shared
  `global_model` every `actor_sync_freq` steps (what the reference does, but on
  the host: eight batch-1 GPU contexts are the wrong shape for one MI355X);
* actor_inference="batched" -- observations go to the GPU inference server
  (parallel/inference_server.py) through a shared-memory mailbox; one batched
  forward serves every actor, with per-actor epsilon applied in the kernel
  epilogue.
"""
import os
import time

import numpy as np
import torch

from ..utils.helpers import reset_experience
from ..utils.param_store import sync_params
from .nstep import make_feeder


def actor_epsilon(process_ind, num_actors, eps, eps_alpha):
    """Ape-X ladder eps^(1 + i/(N-1) * alpha); 0.1 for a single actor
    (reference dqn_actor.py:33-36)."""
    if num_actors <= 1:
        return 0.1
    return eps ** (1.0 + (process_ind - 1) / (num_actors - 1) * eps_alpha)


def _maybe_fault(args, role, process_ind, step):
    spec = getattr(args, "inject_fault", None)
    if not spec:
        return
    try:
        who, when = spec.split("@")
        r, idx = who.split(":")
        if r == "hang-" + role and int(idx) == process_ind and step == int(when):
            import time
            while True:                 # alive but silent: what the supervisor's watchdog is for
                time.sleep(1.0)
        if r == role and int(idx) == process_ind and step == int(when):
            marker = os.path.join(args.log_dir, "fault_%s_%d.done" % (role, process_ind))
            if not os.path.exists(marker):      # only the first incarnation dies
                os.makedirs(args.log_dir, exist_ok=True)
                open(marker, "w").close()
                os._exit(17)
    except ValueError:
        pass


class _EnvWorker(object):
    """One env of an actor process with its n-step feeder, episode bookkeeping and (batched
    inference) mailbox slot.  begin() returns the observation to act on, finish() takes the action."""

    def __init__(self, args, kind, env, feeder, slot, eps, random_process, enable_per):
        self.args, self.kind, self.env, self.feeder, self.slot = args, kind, env, feeder, slot
        self.eps, self.random_process, self.enable_per = eps, random_process, enable_per
        self.early_stop = args.env_params.early_stop
        self.experience = reset_experience()
        self.flag_reset = True      # terminal1 or episode_steps reached early_stop
        self.episode_steps, self.episode_reward = 0, 0.0
        self.total_steps, self.total_reward, self.nepisodes, self.nepisodes_solved = 0, 0.0, 0, 0

    def begin(self):
        if self.flag_reset:
            self.episode_steps, self.episode_reward = 0, 0.0
            self.experience = self.env.reset()
            assert self.experience.state1 is not None
            self.feeder.reset(self.experience.state1)
            self.flag_reset = False
        return self.experience.state1

    def finish(self, action, qvalue, max_qvalue, last_q_fn):
        """Step the env with `action`; last_q_fn(state) -> max_a Q(state) is only called when an
        episode ends with PER on (priority of the closing transition)."""
        if self.kind == "ddpg":
            action = action + self.random_process.sample()
        self.experience = exp = self.env.step(action)
        self.feeder.push(exp.action, exp.reward, exp.state1, exp.terminal1, qvalue, max_qvalue)
        if exp.terminal1:
            self.nepisodes_solved += 1
            self.flag_reset = True
        if self.early_stop and (self.episode_steps + 1) >= self.early_stop:
            self.flag_reset = True
        if self.flag_reset:
            self.feeder.flush(last_q_fn(exp.state1) if self.enable_per else None)
        self.episode_steps += 1
        self.episode_reward += exp.reward
        if self.flag_reset:
            self.nepisodes += 1
            self.total_steps += self.episode_steps
            self.total_reward += self.episode_reward

    def drain_stats(self):
        out = (self.total_steps, self.total_reward, self.nepisodes, self.nepisodes_solved)
        self.total_steps, self.total_reward, self.nepisodes, self.nepisodes_solved = 0, 0.0, 0, 0
        return out


def _actor_loop(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
                global_memory, global_model, kind):
    print("---------------------------->", process_ind, "actor")
    ap, mp_ = args.agent_params, args.memory_params
    E = max(1, int(args.num_envs_per_actor))    # the reference asserts 1 (atari_env.py:15); here E envs per process
    rng = np.random.RandomState(args.seed + 7919 * process_ind)

    mailbox = getattr(global_model, "mailbox", None) if args.actor_inference == "batched" else None
    local_model = None
    local_device = torch.device(getattr(args, "actor_device", "cpu"))
    if mailbox is None:
        local_model = model_prototype(args.model_params, args.norm_val, args.state_shape,
                                      args.action_space, args.action_shape).to(local_device)
        sync_params(local_model, global_model)
        local_model.eval()
    torch.set_grad_enabled(False)
    torch.set_num_threads(1)
    nice = int(getattr(args, "actor_nice", 0))
    if nice > 0:        # env workers yield the cores to the learner / inference threads when oversubscribed
        try:
            os.nice(nice)
        except OSError:
            pass

    enable_per = bool(mp_.enable_per) and kind == "dqn"
    workers = []
    for e in range(E):
        index = (process_ind - 1) * E + e           # global env index: mailbox slot and rung of the eps ladder
        env = env_prototype(args.env_params, process_ind if E == 1 else 1 + index, E)
        env.train()
        eps, random_process = None, None
        if kind == "dqn":
            eps = actor_epsilon(1 + index, args.num_actors * E, ap.eps, ap.eps_alpha)
            if mailbox is not None:
                mailbox.set_epsilon(index, eps)
        else:
            x0 = None
            if getattr(ap, "ou_stationary_start", False):
                # start the OU state at its stationary spread sigma / sqrt(2 theta) instead of 0 (the reference's):
                # from 0 the noise is ~0.03 sqrt(t), so with N actors the first learn_start transitions all carry
                # |action| << 1, the critic has no idea of Q's shape along the action axis, and the tanh actor runs
                # into saturation in whichever direction the first gradients point (measured: tools/probes/ddpg_recipe.py)
                x0 = rng.normal(0.0, 0.3 / np.sqrt(2 * 0.15), size=args.action_space)
            random_process = ap.random_process(size=args.action_space, theta=0.15, sigma=0.3, x0=x0,
                                               n_steps_annealing=mp_.memory_size * 100)
        feeder = make_feeder(global_memory, ap.nstep, ap.gamma, enable_per,
                             native=getattr(args, "native_nstep", True))
        workers.append(_EnvWorker(args, kind, env, feeder, index, eps, random_process, enable_per))

    def local_act(w, state):
        if kind == "dqn":
            return local_model.get_action(state, enable_per, w.eps, device=local_device, rng=rng)
        return local_model.get_action(state, 0.0, device=local_device)

    def last_q(w):
        if mailbox is not None:
            return lambda state: mailbox.request(w.slot, state)[2]
        return lambda state: local_model.get_action(state, True, w.eps, device=local_device, rng=rng)[2]

    # optional lower bound on the replay ratio (the counterpart of the learner's max_replay_ratio): the actors wait
    # while they are more than learn_start + learner_step / min_replay_ratio env steps in, so a slow learner (CPU
    # runs) sees the same data-to-update ratio a fast one does.  0 = uncoupled, as in the reference.
    min_ratio = float(getattr(ap, "min_replay_ratio", 0.0) or 0.0)

    def ahead_of_learner():
        if min_ratio <= 0:
            return False
        allowed = ap.learn_start + 16 * E * args.num_actors + global_logs.learner_step.value / min_ratio
        return global_logs.actor_step.value > allowed

    step = 0
    pending = None
    if mailbox is not None:     # every env has one request in flight; the server batches across actors
        pending = [mailbox.send(w.slot, w.begin()) for w in workers]
    global_logs.beat("actor", process_ind)
    last_beat = time.time()
    while global_logs.running(ap.steps):
        if ahead_of_learner():
            time.sleep(0.001)
            if time.time() - last_beat > 1.0:
                global_logs.beat("actor", process_ind)
                last_beat = time.time()
            continue
        for j, w in enumerate(workers):
            if mailbox is not None:
                action, qvalue, max_qvalue = mailbox.wait(w.slot, pending[j])
            else:
                action, qvalue, max_qvalue = local_act(w, w.begin())
            w.finish(action, qvalue, max_qvalue, last_q(w))
            if mailbox is not None:
                pending[j] = mailbox.send(w.slot, w.begin())    # in flight while the other envs are stepped

            # (the reference locks one shared mp.Value here for every env step of every actor,
            # dqn_actor.py:166-167; each process adds to its own cache line instead)
            global_logs.actor_step.add(process_ind)
            step += 1
            if local_model is not None and step % ap.actor_sync_freq == 0:
                sync_params(local_model, global_model)
            if step % ap.actor_freq == 0:   # push local stats to the logger, then reset them
                tot = [0, 0.0, 0, 0]
                for ww in workers:
                    for i, v in enumerate(ww.drain_stats()):
                        tot[i] += v
                with actor_logs.nepisodes.get_lock():
                    actor_logs.total_steps.value += tot[0]
                    actor_logs.total_reward.value += tot[1]
                    actor_logs.nepisodes.value += tot[2]
                    actor_logs.nepisodes_solved.value += tot[3]
            _maybe_fault(args, "actor", process_ind, step)
        now = time.time()
        if now - last_beat > 1.0:       # sign of life for the supervisor's watchdog (launcher.py)
            global_logs.beat("actor", process_ind)
            last_beat = now


def _profiled(fn, process_ind, args):
    """PDA_ACTOR_PROF=1: actor 1 runs under cProfile and leaves <log_dir>/actor_profile.txt (where an env step's host
    time goes: env, n-step feeder, mailbox round trip, bookkeeping)."""
    if process_ind != 1 or os.environ.get("PDA_ACTOR_PROF", "0") != "1":
        return fn()
    import cProfile
    import io
    import pstats
    prof = cProfile.Profile()
    try:
        return prof.runcall(fn)
    finally:
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(30)
        with open(os.path.join(args.log_dir, "actor_profile.txt"), "w") as f:
            f.write(out.getvalue())


def dqn_actor(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
              global_memory, global_model):
    _profiled(lambda: _actor_loop(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
                                  global_memory, global_model, "dqn"), process_ind, args)


def ddpg_actor(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
               global_memory, global_model):
    _actor_loop(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
                global_memory, global_model, "ddpg")
