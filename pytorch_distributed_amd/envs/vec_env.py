"""VecEnv: N independent envs stepped in lockstep inside one process, with
auto-reset, returning batched arrays -- the shape the batched GPU inference
server consumes (num_envs_per_actor > 1; the reference asserts it is 1,
core/envs/atari_env.py:15).
"""
import numpy as np


class VecEnv(object):
    def __init__(self, env_prototype, args, process_ind=0, num_envs=1):
        self.envs = [env_prototype(args, process_ind * num_envs + i, 1) for i in range(num_envs)]
        self.num_envs = num_envs
        e = self.envs[0]
        self.state_shape = e.state_shape
        self.action_shape = e.action_shape
        self.action_space = e.action_space
        self.norm_val = e.norm_val
        self.discrete = e.discrete
        self.early_stop = e.early_stop
        self.episode_steps = np.zeros(num_envs, dtype=np.int64)

    def train(self):
        for e in self.envs:
            e.train()

    def eval(self):
        for e in self.envs:
            e.eval()

    def reset(self):
        self.episode_steps[:] = 0
        return np.stack([e.reset().state1 for e in self.envs], axis=0)

    def step(self, actions):
        """actions: (N, action_shape).  Returns (state1 (N,...), reward (N,),
        terminal (N,), truncated (N,), next_obs (N,...)): `state1` is the true
        successor, `next_obs` is what to act on next (the reset state when the
        episode ended)."""
        s1, rew, term, trunc, nxt = [], [], [], [], []
        for i, e in enumerate(self.envs):
            exp = e.step(np.asarray(actions[i]).reshape(1, -1))
            self.episode_steps[i] += 1
            t = bool(exp.terminal1)
            tr = bool(self.early_stop and self.episode_steps[i] >= self.early_stop)
            s1.append(exp.state1)
            rew.append(exp.reward)
            term.append(t)
            trunc.append(tr)
            if t or tr:
                nxt.append(e.reset().state1)
                self.episode_steps[i] = 0
            else:
                nxt.append(exp.state1)
        return (np.stack(s1, 0), np.asarray(rew, dtype=np.float32), np.asarray(term),
                np.asarray(trunc), np.stack(nxt, 0))
