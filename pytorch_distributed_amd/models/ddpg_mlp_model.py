"""DDPG actor/critic MLPs.  Actor D->300->200->A with tanh after every layer;
critic D->400 (tanh), then [400 ++ action]->300 (tanh)->1.  Xavier-uniform
hidden layers, U(-3e-3, 3e-3) output layers, zero biases.  Keys `actor.{0,2,4}.*`,
`critic.0.0.*`, `critic.1.{0,2}.*` (reference core/models/ddpg_mlp_model.py:
16-78).  get_action handles A-dimensional actions (the reference's `.item()`
only works for A == 1).
"""
import numpy as np
import torch
import torch.nn as nn

from .model import Model


class DdpgMlpModel(Model):
    def __init__(self, args, norm_val, input_dims, output_dims, action_dims):
        super().__init__(args, norm_val, input_dims, output_dims, action_dims)
        in_dim = int(np.prod(self.input_dims))
        self.actor = nn.Sequential(
            nn.Linear(in_dim, 300), nn.Tanh(),
            nn.Linear(300, 200), nn.Tanh(),
            nn.Linear(200, self.output_dims), nn.Tanh(),
        )
        self.critic = nn.ModuleList()
        self.critic.append(nn.Sequential(nn.Linear(in_dim, 400), nn.Tanh()))
        self.critic.append(nn.Sequential(
            nn.Linear(400 + self.output_dims, 300), nn.Tanh(),
            nn.Linear(300, 1),
        ))
        self._reset()

    def _init_weights(self):
        bound = 3e-3
        hidden = (self.actor[0], self.actor[2], self.critic[0][0], self.critic[1][0])
        last = (self.actor[4], self.critic[1][2])
        for layer in hidden:
            nn.init.xavier_uniform_(layer.weight.data)
            nn.init.constant_(layer.bias.data, 0)
        for layer in last:
            nn.init.uniform_(layer.weight.data, -bound, bound)
            nn.init.constant_(layer.bias.data, 0)

    def forward_actor(self, input):
        return self.actor(input.float().flatten(1))

    def forward_critic(self, input, action):
        qx = self.critic[0](input.float().flatten(1))
        return self.critic[1](torch.cat((qx, action.float()), 1))

    def forward(self, input):
        action = self.forward_actor(input)
        return action, self.forward_critic(input, action)

    @torch.no_grad()
    def get_action(self, input, noise=0.0, device=torch.device("cpu")):
        x = torch.as_tensor(np.asarray(input)).unsqueeze(0).to(device)
        action = self.forward_actor(x).cpu().numpy().astype(np.float64)   # (1, A)
        return action + noise, 0.0, 0.0
