"""Exploration noise processes for continuous control.

Capability parity with reference utils/random_process.py:10-49: an annealed
Gaussian base and the Ornstein-Uhlenbeck process the DDPG actor uses
(theta=0.15, sigma=0.3).  An actor process that steps several envs
(agents/actors.py: one _EnvWorker per env) gives each env its own process.
"""
import math

import numpy as np


class RandomProcess(object):
    def reset_states(self):
        pass


class AnnealedGaussianProcess(RandomProcess):
    """sigma decays linearly from `sigma` to `sigma_min` over n_steps_annealing;
    with sigma_min=None it stays constant (the reference's DDPG setting)."""

    def __init__(self, mu, sigma, sigma_min, n_steps_annealing):
        self.mu = mu
        self.sigma = sigma
        self.n_steps = 0
        if sigma_min is None:
            self.slope, self.intercept, self.sigma_min = 0.0, sigma, sigma
        else:
            self.slope = -float(sigma - sigma_min) / float(n_steps_annealing)
            self.intercept, self.sigma_min = sigma, sigma_min

    @property
    def current_sigma(self):
        return max(self.sigma_min, self.slope * float(self.n_steps) + self.intercept)


class OrnsteinUhlenbeckProcess(AnnealedGaussianProcess):
    def __init__(self, theta, mu=0.0, sigma=1.0, dt=1e-2, x0=None, size=1,
                 sigma_min=None, n_steps_annealing=1000, seed=None):
        super().__init__(mu=mu, sigma=sigma, sigma_min=sigma_min, n_steps_annealing=n_steps_annealing)
        self.theta, self.dt, self.x0, self.size = theta, dt, x0, size
        self.rng = np.random.RandomState(seed) if seed is not None else np.random
        self.reset_states()

    def sample(self):
        drift = self.theta * (self.mu - self.x_prev) * self.dt
        diffusion = self.current_sigma * math.sqrt(self.dt) * self.rng.normal(size=self.size)
        self.x_prev = self.x_prev + drift + diffusion
        self.n_steps += 1
        return self.x_prev

    def reset_states(self):
        self.x_prev = np.array(self.x0, dtype=np.float64) if self.x0 is not None else np.zeros(self.size)
