"""Tester process (mode=2): load ./models/MACHINE_TIMESTAMP.pth on the CPU, play
`tester_nepisodes` greedy episodes, print the summary (reference
core/single_processes/testers.py:8-83).
"""
import torch

from ..utils.checkpoint import load_state_dict
from .evaluators import run_episodes


def tester(process_ind, args, env_prototype, model_prototype, result=None):
    print("---------------------------->", process_ind, "tester")
    ap = args.agent_params
    env = env_prototype(args.env_params, process_ind)
    env.eval()
    local_device = torch.device("cpu")
    local_model = model_prototype(args.model_params, args.norm_val, args.state_shape,
                                  args.action_space, args.action_shape).to(local_device)
    local_model.load_state_dict(load_state_dict(args.model_file))
    local_model.eval()
    torch.set_grad_enabled(False)

    total_steps, total_reward, nepisodes, solved = run_episodes(
        env, local_model, ap.tester_nepisodes, args.env_params.early_stop, local_device,
        verbose="Testing Episode ")
    print("nepisodes:", nepisodes)
    print("avg_steps:", total_steps / nepisodes)
    print("avg_reward:", total_reward / nepisodes)
    print("nepisodes_solved:", solved)
    print("repisodes_solved:", solved / nepisodes)
    if result is not None:      # optional mp.Array for tests
        result[0], result[1], result[2] = total_steps / nepisodes, total_reward / nepisodes, solved / nepisodes
