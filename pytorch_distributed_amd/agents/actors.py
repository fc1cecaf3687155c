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
        if r == role and int(idx) == process_ind and step == int(when):
            marker = os.path.join(args.log_dir, "fault_%s_%d.done" % (role, process_ind))
            if not os.path.exists(marker):      # only the first incarnation dies
                os.makedirs(args.log_dir, exist_ok=True)
                open(marker, "w").close()
                os._exit(17)
    except ValueError:
        pass


def _actor_loop(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
                global_memory, global_model, kind):
    print("---------------------------->", process_ind, "actor")
    ap, mp_ = args.agent_params, args.memory_params
    env = env_prototype(args.env_params, process_ind, args.num_envs_per_actor)
    env.train()
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
    if kind == "dqn":
        eps = actor_epsilon(process_ind, args.num_actors, ap.eps, ap.eps_alpha)
        if mailbox is not None:
            mailbox.set_epsilon(process_ind - 1, eps)
    else:
        random_process = ap.random_process(size=args.action_space, theta=0.15, sigma=0.3,
                                           n_steps_annealing=mp_.memory_size * 100)

    def act(state):
        if kind == "dqn":
            if mailbox is not None:
                return mailbox.request(process_ind - 1, state)
            return local_model.get_action(state, enable_per, eps, device=local_device, rng=rng)
        noise = random_process.sample()
        if mailbox is not None:
            a, q, mq = mailbox.request(process_ind - 1, state)
            return a + noise, q, mq
        return local_model.get_action(state, noise, device=local_device)

    feeder = make_feeder(global_memory, ap.nstep, ap.gamma, enable_per,
                         native=getattr(args, "native_nstep", True))
    experience = reset_experience()
    step = 0
    episode_steps, episode_reward = 0, 0.0
    total_steps, total_reward, nepisodes, nepisodes_solved = 0, 0.0, 0, 0
    flag_reset = True   # terminal1 or episode_steps reached early_stop
    while global_logs.running(ap.steps):
        if flag_reset:
            episode_steps, episode_reward = 0, 0.0
            experience = env.reset()
            assert experience.state1 is not None
            feeder.reset(experience.state1)
            flag_reset = False

        action, qvalue, max_qvalue = act(experience.state1)
        experience = env.step(action)

        feeder.push(experience.action, experience.reward, experience.state1, experience.terminal1,
                    qvalue, max_qvalue)

        if experience.terminal1:
            nepisodes_solved += 1
            flag_reset = True
        if args.env_params.early_stop and (episode_steps + 1) >= args.env_params.early_stop:
            flag_reset = True

        if flag_reset:
            max_q_last = None
            if enable_per:      # one extra forward of the final state for its priority
                if mailbox is not None:
                    _, _, max_q_last = mailbox.request(process_ind - 1, experience.state1)
                else:
                    _, _, max_q_last = local_model.get_action(experience.state1, True, eps,
                                                              device=local_device, rng=rng)
            feeder.flush(max_q_last)

        with global_logs.actor_step.get_lock():
            global_logs.actor_step.value += 1
        step += 1
        episode_steps += 1
        episode_reward += experience.reward
        if flag_reset:
            nepisodes += 1
            total_steps += episode_steps
            total_reward += episode_reward

        if local_model is not None and step % ap.actor_sync_freq == 0:
            sync_params(local_model, global_model)

        if step % ap.actor_freq == 0:   # push local stats to the logger, then reset them
            with actor_logs.nepisodes.get_lock():
                actor_logs.total_steps.value += total_steps
                actor_logs.total_reward.value += total_reward
                actor_logs.nepisodes.value += nepisodes
                actor_logs.nepisodes_solved.value += nepisodes_solved
            total_steps, total_reward, nepisodes, nepisodes_solved = 0, 0.0, 0, 0
            global_logs.beat("actor")
        _maybe_fault(args, "actor", process_ind, step)


def dqn_actor(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
              global_memory, global_model):
    _actor_loop(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
                global_memory, global_model, "dqn")


def ddpg_actor(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
               global_memory, global_model):
    _actor_loop(process_ind, args, global_logs, actor_logs, env_prototype, model_prototype,
                global_memory, global_model, "ddpg")
