"""Pure-numpy continuous-control and toy environments (gym is not a dependency).

`PendulumEnv`   classic torque-limited pendulum swing-up: obs (cos th, sin th,
                th_dot), one action in [-2, 2] (the model emits tanh in [-1, 1]
                and the env scales it), reward -(th^2 + .1 th_dot^2 + .001 u^2),
                200-step episodes.  The reference lists Distributed DDPG but
                registers no continuous env (utils/factory.py:33-34); this one
                makes config 3 runnable out of the box.
`ChainEnv`      5-state corridor for DQN-MLP sanity tests (right = +1 at the end).
`CountingEnv`   deterministic fake: state is filled with the step index, reward
                equals the step index -- the "state = deque length" trick from
                the reference's dqn_actor.py.bk-per debug prints, used by the
                n-step oracle tests.
"""
import numpy as np

from .env import Env


class PendulumEnv(Env):
    max_speed = 8.0
    max_torque = 2.0
    dt = 0.05
    g = 10.0
    m = 1.0
    l = 1.0

    def __init__(self, args, process_ind=0, num_envs_per_process=1):
        super().__init__(args, process_ind, num_envs_per_process)
        self.th = 0.0
        self.thdot = 0.0
        self.t = 0
        self.last_u = 0.0

    @property
    def discrete(self):
        return False

    @property
    def state_shape(self):
        return [1, 1, 3]

    @property
    def action_shape(self):
        return 1

    @property
    def action_space(self):
        return 1

    def _obs(self):
        return np.array([[[np.cos(self.th), np.sin(self.th), self.thdot / self.max_speed]]], dtype=np.float32)

    def reset(self):
        self._reset_experience()
        self.th = self.rng.uniform(-np.pi, np.pi)
        self.thdot = self.rng.uniform(-1.0, 1.0)
        self.t = 0
        self.exp_state1 = self._obs()
        return self._get_experience()

    def step(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        self.exp_action = np.asarray(action, dtype=np.float32).reshape(self.action_shape)
        u = float(np.clip(a[0], -1.0, 1.0)) * self.max_torque
        th_norm = ((self.th + np.pi) % (2 * np.pi)) - np.pi
        cost = th_norm ** 2 + 0.1 * self.thdot ** 2 + 0.001 * u ** 2
        self.thdot = self.thdot + (3 * self.g / (2 * self.l) * np.sin(self.th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
        self.thdot = float(np.clip(self.thdot, -self.max_speed, self.max_speed))
        self.th = self.th + self.thdot * self.dt
        self.t += 1
        self.last_u = u
        self.exp_reward = float(-cost)
        self.exp_terminal1 = False      # time-limit only (actor applies early_stop)
        self.exp_state1 = self._obs()
        return self._get_experience()


class ChainEnv(Env):
    """States 0..n-1 one-hot; action 1 moves right, 0 moves left; reaching the
    right end pays +1 and terminates, every other step pays -0.01."""
    n = 5

    def __init__(self, args, process_ind=0, num_envs_per_process=1):
        super().__init__(args, process_ind, num_envs_per_process)
        self.pos = 0

    @property
    def state_shape(self):
        return [1, 1, self.n]

    @property
    def action_shape(self):
        return 1

    @property
    def action_space(self):
        return 2

    def _obs(self):
        s = np.zeros((1, 1, self.n), dtype=np.float32)
        s[0, 0, self.pos] = 1.0
        return s

    def reset(self):
        self._reset_experience()
        self.pos = 0
        self.exp_state1 = self._obs()
        return self._get_experience()

    def step(self, action):
        a = int(np.asarray(action).reshape(-1)[0])
        self.exp_action = np.asarray(action).reshape(self.action_shape)
        self.pos = min(self.pos + 1, self.n - 1) if a == 1 else max(self.pos - 1, 0)
        done = self.pos == self.n - 1
        self.exp_reward = 1.0 if done else -0.01
        self.exp_terminal1 = bool(done)
        self.exp_state1 = self._obs()
        return self._get_experience()


class CountingEnv(Env):
    """Deterministic: the k-th state of an episode is filled with k, the k-th
    reward is k, episodes last `episode_len` steps."""
    episode_len = 9

    def __init__(self, args, process_ind=0, num_envs_per_process=1):
        super().__init__(args, process_ind, num_envs_per_process)
        self.k = 0
        self.width = self.state_wid or 4

    @property
    def state_shape(self):
        return [1, 1, self.width]

    @property
    def action_shape(self):
        return 1

    @property
    def action_space(self):
        return 3

    def _obs(self):
        return np.full((1, 1, self.width), float(self.k), dtype=np.float32)

    def reset(self):
        self._reset_experience()
        self.k = 0
        self.exp_state1 = self._obs()
        return self._get_experience()

    def step(self, action):
        self.exp_action = np.asarray(action).reshape(self.action_shape)
        self.k += 1
        self.exp_reward = float(self.k)
        self.exp_terminal1 = bool(self.k >= self.episode_len)
        self.exp_state1 = self._obs()
        return self._get_experience()
