"""Offline DDPG on a fixed Pendulum data set: fp32 torch core vs native HIP engine (diagnostic)."""
import copy
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pytorch_distributed_amd.launcher import prepare_options
from pytorch_distributed_amd.memories.device_memory import DeviceMemory
from pytorch_distributed_amd.utils.options import Options
from pytorch_distributed_amd.utils.factory import EnvsDict, ModelsDict
from pytorch_distributed_amd.agents.learners import DdpgTorchCore
from pytorch_distributed_amd.ops.mlp_engine import MlpNativeCore

leak = len(sys.argv) > 1 and sys.argv[1] == "leak"
opt = prepare_options(Options(config=7, overrides={"memory_params.memory_size": 32768, "agent_params.lr": 1e-3,
                                                   "agent_params.compat_ddpg_grad_leak": leak}))
dev = torch.device("cuda")
env = EnvsDict[opt.env_type](opt.env_params, 0)
rng = np.random.RandomState(0)
S0, A, R, S1, T = [], [], [], [], []
n = 20000
exp = env.reset()
t = 0
while len(S0) < n:
    a = rng.uniform(-1, 1, size=(1, 1))
    s0 = exp.state1
    exp = env.step(a)
    S0.append(s0); A.append(a.reshape(-1)); R.append([exp.reward]); S1.append(exp.state1); T.append([0.0])
    t += 1
    if t >= 200:
        exp = env.reset(); t = 0
mems = []
for _ in range(2):
    m = DeviceMemory(opt.memory_params); m.use_pump = False; m.attach(dev)
    m.feed_batch(torch.tensor(np.array(S0)).to(dev), torch.tensor(np.array(A)).float().to(dev),
                 torch.tensor(R).float().to(dev), torch.full((n, 1), 0.99, device=dev),
                 torch.tensor(np.array(S1)).to(dev), torch.tensor(T).float().to(dev))
    mems.append(m)
torch.manual_seed(0)
proto = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, opt.action_shape)
cores = {"torch": DdpgTorchCore(opt, copy.deepcopy(proto).to(dev), copy.deepcopy(proto).to(dev), mems[0]),
         "native": MlpNativeCore("ddpg", opt, copy.deepcopy(proto).to(dev), copy.deepcopy(proto).to(dev), mems[1])}
probe = torch.tensor(np.array(S0[:512])).to(dev)
for it in range(1, 6001):
    for c in cores.values():
        c.step()
    if it % 500 == 0:
        line = "step %5d" % it
        for name, c in cores.items():
            lv = c.loss_values()
            with torch.no_grad():
                act, q = c.model(probe)
            line += " | %s critic %.4f actor %.3f  |pi| %.3f  Q %.2f" % (name, lv["critic_loss"], lv["actor_loss"],
                                                                       act.abs().mean().item(), q.mean().item())
        print(line, flush=True)
