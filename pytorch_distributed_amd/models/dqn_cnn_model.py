"""Nature-CNN DQN critic.  Same layers, state-dict keys (`critic.0.{0,2,4}.*`,
`critic.1.{0,2}.*`) and get_action contract as the reference
(core/models/dqn_cnn_model.py:16-78); 1,687,206 parameters at 6 actions.

The module itself runs plain torch ops (CPU plumbing, evaluator, tester, and
the fp32 oracle for the kernel tests).  On MI355X the learner and the batched
actor server bind a `ops.dqn_engine.DqnCnnEngine` to this module's flat
parameter buffer: uint8 frames go straight from HBM replay into an MFMA
implicit-GEMM conv1 (the /255 is folded into its epilogue), activations stay
bf16 NHWC, and backward/Adam run as hand-written HIP kernels.
"""
import numpy as np
import torch
import torch.nn as nn

from .model import Model


class DqnCnnModel(Model):
    def __init__(self, args, norm_val, input_dims, output_dims, action_dims):
        super().__init__(args, norm_val, input_dims, output_dims, action_dims)
        self.critic = nn.ModuleList()
        self.critic.append(nn.Sequential(
            nn.Conv2d(self.input_dims[0], 32, kernel_size=8, stride=4), nn.ReLU(),
            nn.Conv2d(32, 64, kernel_size=4, stride=2), nn.ReLU(),
            nn.Conv2d(64, 64, kernel_size=3, stride=1), nn.ReLU(),
        ))
        conv_out = self._get_conv_out_size(self.input_dims)
        self.critic.append(nn.Sequential(
            nn.Linear(conv_out, 512), nn.ReLU(),
            nn.Linear(512, self.output_dims),
        ))
        # The reference defines an orthogonal init but never calls it
        # (dqn_cnn_model.py:33), so torch's default init is what trains there.
        if getattr(args, "init", "default") == "orthogonal":
            self._reset()

    def _get_conv_out_size(self, input_dims):
        h, w = input_dims[1], input_dims[2]
        for k, s in ((8, 4), (4, 2), (3, 1)):
            h, w = (h - k) // s + 1, (w - k) // s + 1
        return 64 * h * w

    def _init_weights(self):
        gain = nn.init.calculate_gain("relu")
        for layer in (self.critic[0][0], self.critic[0][2], self.critic[0][4],
                      self.critic[1][0], self.critic[1][2]):
            nn.init.orthogonal_(layer.weight.data, gain)
            nn.init.constant_(layer.bias.data, 0)

    def forward(self, input):
        x = input.float() / self.norm_val
        return self.critic[1](self.critic[0](x).flatten(1))

    @torch.no_grad()
    def get_action(self, input, enable_per=False, eps=0.0, device=torch.device("cpu"), rng=np.random):
        """eps-greedy.  Returns (action np[[a]], q(s,a), max_a q(s,a)); the
        forward is skipped for a random action unless PER needs the q-values."""
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
