"""Built-in Atari-shaped environments that need no ROMs.

`SyntheticAtariEnv` emits (4, 84, 84) uint8 frame stacks with Pong's action
count, +-1 rewards and episode boundaries: the measurement config named in
BASELINE.json ("synthetic observations of Atari shape").  With game="catch" the
same class renders a small learnable pixel game (a ball falls, a 3-action
paddle has to be under it) so a CNN agent can be sanity-trained without ALE.
The Env API matches AtariEnv (reference core/envs/atari_env.py:60-129).
"""
from collections import deque

import numpy as np

from .env import Env


class SyntheticAtariEnv(Env):
    _POOL = 32      # distinct pre-rendered noise frames per env

    def __init__(self, args, process_ind=0, num_envs_per_process=1):
        super().__init__(args, process_ind, num_envs_per_process)
        self.hei = self.state_hei or 84
        self.wid = self.state_wid or 84
        self.catch = (self.game == "catch")
        self.num_actions = 3 if self.catch else int(getattr(args, "synthetic_num_actions", 6))
        self.mean_len = int(getattr(args, "synthetic_episode_len", 1000))
        self.pool = self.rng.randint(0, 256, size=(self._POOL, self.hei, self.wid)).astype(np.uint8)
        self.frames = deque(maxlen=self.state_cha)
        self.t = 0
        self._blank()

    def _blank(self):
        self.frames.clear()
        for _ in range(self.state_cha):
            self.frames.append(np.zeros((self.hei, self.wid), dtype=np.uint8))

    @property
    def norm_val(self):
        return 255.0

    @property
    def state_shape(self):
        return [self.state_cha, self.hei, self.wid]

    @property
    def action_shape(self):
        return 1

    @property
    def action_space(self):
        return self.num_actions

    # -- catch game ------------------------------------------------------
    _GRID = 12      # catch is played on a 12x12 grid of 7x7-pixel cells

    def _catch_reset(self):
        self.ball_x = self.rng.randint(self._GRID)
        self.ball_y = 0
        self.paddle = self._GRID // 2

    def _catch_frame(self):
        f = np.zeros((self.hei, self.wid), dtype=np.uint8)
        c = self.hei // self._GRID
        f[self.ball_y * c:(self.ball_y + 1) * c, self.ball_x * c:(self.ball_x + 1) * c] = 255
        lo = max(self.paddle - 1, 0) * c
        hi = min(self.paddle + 2, self._GRID) * c
        f[(self._GRID - 1) * c:self._GRID * c, lo:hi] = 128
        return f

    # -- Env API ---------------------------------------------------------
    def _state(self):
        return np.stack(self.frames, axis=0)

    def reset(self):
        self._reset_experience()
        self._blank()
        self.t = 0
        if self.catch:
            self._catch_reset()
            self.frames.append(self._catch_frame())
        else:
            self.frames.append(self.pool[self.rng.randint(self._POOL)])
        self.exp_state1 = self._state()
        return self._get_experience()

    def step(self, action):
        a = int(np.asarray(action).reshape(-1)[0])
        self.exp_action = np.asarray(action).reshape(self.action_shape)
        self.t += 1
        if self.catch:
            self.paddle = int(np.clip(self.paddle + (a - 1), 0, self._GRID - 1))
            self.ball_y += 1
            done = self.ball_y >= self._GRID - 1
            self.exp_reward = 0.0
            if done:
                self.exp_reward = 1.0 if abs(self.ball_x - self.paddle) <= 1 else -1.0
            self.exp_terminal1 = bool(done)
            self.frames.append(self._catch_frame())
        else:
            u = self.rng.uniform()
            self.exp_reward = 1.0 if u < 0.01 else (-1.0 if u < 0.02 else 0.0)
            self.exp_terminal1 = bool(self.rng.uniform() < 1.0 / self.mean_len)
            self.frames.append(self.pool[(self.t * 7 + a) % self._POOL])
        self.exp_state1 = self._state()
        return self._get_experience()


def make_synthetic_env(args, process_ind=0, num_envs_per_process=1):
    """EnvsDict["synthetic"]: the built-in pixel games by `game` -- "minipong" (envs/minipong.py: a real horizon,
    first to 21), "catch" (a 12-step toy), anything else: Atari-shaped noise frames for throughput runs."""
    if args.game == "minipong":
        from .minipong import MiniPongEnv
        return MiniPongEnv(args, process_ind, num_envs_per_process)
    return SyntheticAtariEnv(args, process_ind, num_envs_per_process)
