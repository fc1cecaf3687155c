"""Env base class: reset()/step(action) -> Experience, train()/eval(), and the
shape properties the launcher reads off a dummy env (reference core/env.py:5-76).
"""
import numpy as np

from ..utils.helpers import Experience


class Env(object):
    def __init__(self, args, process_ind=0, num_envs_per_process=1):
        self.process_ind = process_ind                    # distinct seeds per process
        self.num_envs_per_process = num_envs_per_process
        self.num_envs_per_actor = args.num_envs_per_actor

        self.mode = args.mode
        self.seed = args.seed + process_ind * args.num_envs_per_actor

        self.env_type = args.env_type
        self.game = args.game
        self.state_cha = args.state_cha
        self.state_hei = args.state_hei
        self.state_wid = args.state_wid
        self.early_stop = args.early_stop

        self.training = True
        self.action_repetition = 4
        self.rng = np.random.RandomState(self.seed)
        self._reset_experience()

    def train(self):    # loss of life counts as terminal
        self.training = True
        self.action_repetition = 4

    def eval(self):     # standard terminal signal
        self.training = False
        self.action_repetition = 4

    def _reset_experience(self):
        self.exp_state0 = None      # always None coming out of an env
        self.exp_action = None
        self.exp_reward = None
        self.exp_state1 = None
        self.exp_terminal1 = None

    def _preprocess_state(self, state):
        return state

    def _get_experience(self):
        return Experience(state0=self.exp_state0,
                          action=self.exp_action,
                          reward=self.exp_reward,
                          state1=self._preprocess_state(self.exp_state1),
                          terminal1=self.exp_terminal1)

    @property
    def norm_val(self):             # max state value, used to normalise model inputs
        return 1.0

    @property
    def state_shape(self):
        raise NotImplementedError("not implemented in base class")

    @property
    def action_shape(self):         # discrete: 1; continuous: action dim
        raise NotImplementedError("not implemented in base class")

    @property
    def action_space(self):         # discrete: #actions; continuous: action dim
        raise NotImplementedError("not implemented in base class")

    @property
    def discrete(self):
        return True

    def render(self):
        pass

    def visual(self):
        pass

    def close(self):
        pass

    def reset(self):
        raise NotImplementedError("not implemented in base class")

    def step(self, action):
        raise NotImplementedError("not implemented in base class")
