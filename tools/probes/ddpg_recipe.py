"""Synchronous single-process DDPG on PendulumEnv with the package's model, OU process and n-step assembler and the
reference learner's arithmetic (agents/learners.py DdpgTorchCore): a fast way to see which hyper-parameters decide
whether the reference's DDPG configuration learns, before the same settings go through the asynchronous job.

    python tools/probes/ddpg_recipe.py --lr 1e-4 --leak 1 --reward-scale 1 --seeds 0 1 2 3
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(a, seed):
    from pytorch_distributed_amd.agents.nstep import NStepAssembler
    from pytorch_distributed_amd.envs.pendulum_env import PendulumEnv
    from pytorch_distributed_amd.models.ddpg_mlp_model import DdpgMlpModel
    from pytorch_distributed_amd.utils.options import Options
    from pytorch_distributed_amd.utils.random_process import OrnsteinUhlenbeckProcess
    torch.manual_seed(seed)
    np.random.seed(seed)
    torch.set_num_threads(1)
    opt = Options(config=3, overrides={"device": "cpu", "seed": seed})
    env, eval_env = PendulumEnv(opt.env_params, 1, 1), PendulumEnv(opt.env_params, 99, 1)
    model = DdpgMlpModel(opt.model_params, 1.0, [1, 1, 3], 1, 1)
    target = DdpgMlpModel(opt.model_params, 1.0, [1, 1, 3], 1, 1)
    target.load_state_dict(model.state_dict())
    if a.critic_lr:
        optim = torch.optim.Adam([{"params": model.actor.parameters(), "lr": a.lr},
                                  {"params": model.critic.parameters(), "lr": a.critic_lr}])
    else:
        optim = torch.optim.Adam(model.parameters(), lr=a.lr)
    noise = OrnsteinUhlenbeckProcess(size=1, theta=0.15, sigma=a.sigma, n_steps_annealing=50000 * 100, seed=seed)
    asm = NStepAssembler(a.nstep, 0.99, False)
    N = a.memory
    S0, A, R, G, S1, T = (np.zeros((N, 3), np.float32), np.zeros((N, 1), np.float32), np.zeros((N, 1), np.float32),
                          np.zeros((N, 1), np.float32), np.zeros((N, 3), np.float32), np.zeros((N, 1), np.float32))
    pos, size = 0, 0

    def feed(items):
        nonlocal pos, size
        for (s0, act, r, g, s1, t), _ in items:
            S0[pos], A[pos], R[pos], G[pos], S1[pos], T[pos] = (np.asarray(s0).reshape(-1), np.asarray(act).reshape(-1),
                                                               r[0] * a.reward_scale, g[0], np.asarray(s1).reshape(-1),
                                                               float(t))
            pos, size = (pos + 1) % N, min(size + 1, N)

    sat = []

    def evaluate(n=10):
        tot, big, cnt = 0.0, 0, 0
        for _ in range(n):
            e = eval_env.reset()
            for _ in range(200):
                act, _, _ = model.get_action(e.state1)
                big += int(abs(float(act.reshape(-1)[0])) > 0.98)
                cnt += 1
                e = eval_env.step(act)
                tot += e.reward
        sat.append(round(big / cnt, 2))
        return tot / n

    rng = np.random.RandomState(seed)
    updates, evals, debt = 0, [], 0.0
    e = env.reset()
    asm.reset(e.state1)
    ep_t = 0
    for step in range(1, a.steps + 1):
        act, _, _ = model.get_action(e.state1)
        e = env.step(act + noise.sample())
        feed(asm.push(e.action, e.reward, e.state1, e.terminal1))
        ep_t += 1
        if ep_t >= 200:
            feed(asm.flush())
            e = env.reset()
            asm.reset(e.state1)
            ep_t = 0
        if size > a.learn_start:
            debt += a.ratio
            while debt >= 1.0:
                debt -= 1.0
                idx = rng.randint(0, size, a.batch)
                s0, ac, r, g, s1, t = (torch.from_numpy(x[idx]) for x in (S0, A, R, G, S1, T))
                optim.zero_grad()
                if not a.leak:
                    for p in model.critic.parameters():
                        p.requires_grad_(False)
                _, q_pi = model(s0)
                if updates >= a.actor_warmup:
                    (-q_pi.mean()).backward()
                if not a.leak:
                    for p in model.critic.parameters():
                        p.requires_grad_(True)
                with torch.no_grad():
                    _, nq = target(s1)
                    y = r + g * nq * (1 - t)
                loss = torch.nn.functional.mse_loss(model.forward_critic(s0, ac), y)
                loss.backward()
                torch.nn.utils.clip_grad_value_(model.parameters(), a.clip)
                optim.step()
                with torch.no_grad():
                    for pt, p in zip(target.parameters(), model.parameters()):
                        pt.mul_(1 - a.tau).add_(p, alpha=a.tau)
                updates += 1
        if step % a.eval_every == 0:
            evals.append(round(evaluate(), 1))
    return evals, sat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--critic-lr", type=float, default=0.0)
    ap.add_argument("--leak", type=int, default=1)
    ap.add_argument("--reward-scale", type=float, default=1.0)
    ap.add_argument("--nstep", type=int, default=5)
    ap.add_argument("--tau", type=float, default=1e-3)
    ap.add_argument("--ratio", type=float, default=1.0, help="learner updates per env step")
    ap.add_argument("--sigma", type=float, default=0.3)
    ap.add_argument("--clip", type=float, default=40.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--memory", type=int, default=50000)
    ap.add_argument("--learn-start", type=int, default=250)
    ap.add_argument("--actor-warmup", type=int, default=0)
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--eval-every", type=int, default=2000)
    ap.add_argument("--seeds", type=int, nargs="*", default=[0])
    a = ap.parse_args()
    for seed in a.seeds:
        ev, sat = run(a, seed)
        print(json.dumps({"seed": seed, "saturated": sat, "lr": a.lr, "critic_lr": a.critic_lr, "leak": a.leak, "rs": a.reward_scale,
                          "nstep": a.nstep, "tau": a.tau, "ratio": a.ratio, "last3": ev[-3:], "evals": ev}), flush=True)


if __name__ == "__main__":
    main()
