"""Evaluator process: every `evaluator_freq` seconds pull a weight snapshot,
play `evaluator_nepisodes` greedy episodes in an eval-mode env, publish the
totals for the logger, and write the checkpoint -- the only checkpoint writer,
as in the reference (core/single_processes/evaluators.py:9-102).
"""
import time

import torch

from ..utils.checkpoint import save_state_dict
from ..utils.helpers import reset_experience
from ..utils.param_store import sync_params


def run_episodes(env, model, nepisodes, early_stop, device=torch.device("cpu"), verbose=None):
    """Greedy rollouts; returns (total_steps, total_reward, nepisodes, nepisodes_solved)."""
    experience = reset_experience()
    episode_steps, episode_reward = 0, 0.0
    total_steps, total_reward, done_eps, solved = 0, 0.0, 0, 0
    flag_reset = True
    while done_eps < nepisodes:
        if flag_reset:
            episode_steps, episode_reward = 0, 0.0
            experience = env.reset()
            assert experience.state1 is not None
            flag_reset = False
        action, _, _ = model.get_action(experience.state1, device=device)
        experience = env.step(action)
        if experience.terminal1:
            solved += 1
            flag_reset = True
        if early_stop and (episode_steps + 1) >= early_stop:
            flag_reset = True
        episode_steps += 1
        episode_reward += experience.reward
        if flag_reset:
            done_eps += 1
            total_steps += episode_steps
            total_reward += episode_reward
            if verbose:
                print(verbose, done_eps)
    return total_steps, total_reward, done_eps, solved


def evaluator(process_ind, args, global_logs, evaluator_logs, env_prototype, model_prototype, global_model):
    print("---------------------------->", process_ind, "evaluator")
    ap = args.agent_params
    env = env_prototype(args.env_params, process_ind)
    env.eval()
    local_device = torch.device(getattr(args, "actor_device", "cpu"))
    local_model = model_prototype(args.model_params, args.norm_val, args.state_shape,
                                  args.action_space, args.action_shape).to(local_device)
    sync_params(local_model, global_model)
    local_model.eval()
    torch.set_grad_enabled(False)
    torch.set_num_threads(max(1, min(2, torch.get_num_threads())))     # the env workers need the cores
    poll = float(getattr(ap, "poll_interval", 5.0))

    def evaluate_and_save():
        sync_params(local_model, global_model)
        total_steps, total_reward, nepisodes, solved = run_episodes(
            env, local_model, ap.evaluator_nepisodes, args.env_params.early_stop, local_device)
        with evaluator_logs.logger_lock.get_lock():
            evaluator_logs.total_steps.value = total_steps
            evaluator_logs.total_reward.value = total_reward
            evaluator_logs.nepisodes.value = nepisodes
            evaluator_logs.nepisodes_solved.value = solved
            evaluator_logs.logger_lock.value = True
        print("Saving model " + args.model_name + " ...")
        save_state_dict(local_model, args.model_name)
        print("Saved  model " + args.model_name + ".")
        # optional "solved": end the job once an evaluation reaches the target score (bench/pong_wallclock.py measures
        # the wall-clock to that point; the reference only ever stops at `steps`)
        goal = getattr(ap, "stop_at_eval_reward", None)
        if goal is not None and nepisodes > 0 and total_reward / nepisodes >= float(goal):
            print("[evaluator] avg reward %.2f >= %.2f: stopping the job" % (total_reward / nepisodes, float(goal)), flush=True)
            global_logs.stop.value = True

    last_eval_time = time.time()
    while global_logs.running(ap.steps):
        time.sleep(poll)
        if time.time() - last_eval_time > ap.evaluator_freq:
            evaluate_and_save()
            last_eval_time = time.time()
    if not global_logs.stop.value:
        evaluate_and_save()     # final weights always get evaluated and checkpointed
