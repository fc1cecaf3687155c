"""DQN critic for low-dimensional observations: D -> 256 -> 256 -> 256 -> A with
ReLU (reference core/models/dqn_mlp_model.py:15-26, keys `critic.{0,2,4,6}.*`).
The reference leaves this model unregistered and with an older one-value
get_action; here it is registered as "dqn-mlp" and returns the same
(action, qvalue, max_qvalue) triple the actors unpack.
"""
import numpy as np
import torch
import torch.nn as nn

from .model import Model


class DqnMlpModel(Model):
    def __init__(self, args, norm_val, input_dims, output_dims, action_dims):
        super().__init__(args, norm_val, input_dims, output_dims, action_dims)
        self.hidden_dims = 256
        in_dim = int(np.prod(self.input_dims))
        self.critic = nn.Sequential(
            nn.Linear(in_dim, self.hidden_dims), nn.ReLU(),
            nn.Linear(self.hidden_dims, self.hidden_dims), nn.ReLU(),
            nn.Linear(self.hidden_dims, self.hidden_dims), nn.ReLU(),
            nn.Linear(self.hidden_dims, self.output_dims),
        )
        self._reset()

    def _init_weights(self):
        pass    # torch defaults, as in the reference

    def forward(self, input):
        return self.critic(input.float().flatten(1) / self.norm_val)

    @torch.no_grad()
    def get_action(self, input, enable_per=False, eps=0.0, device=torch.device("cpu"), rng=np.random):
        action, qvalue, max_qvalue = None, None, None
        need_forward = True
        if eps > 0.0 and rng.uniform() < eps:
            action = rng.randint(self.output_dims, size=(1, self.action_dims))
            need_forward = enable_per
        if need_forward:
            x = torch.as_tensor(np.asarray(input)).unsqueeze(0).to(device)
            qvalues = self.forward(x)
            max_q, max_a = qvalues.max(dim=1)
            max_qvalue = max_q.item()
            if action is None:
                qvalue, action = max_qvalue, np.array([[max_a.item()]])
            elif enable_per:
                qvalue = qvalues[0, int(action[0][0])].item()
        return action, qvalue, max_qvalue
