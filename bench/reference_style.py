"""Baseline measurement: the reference's learner and actor hot loops, re-created
step for step (same data path, same op sequence) and timed on the local GPU.

The reference publishes no throughput, so the number to beat is
what its own loop achieves on this hardware.  This script re-implements, from
the description in SURVEY.md section 3, what
  core/single_processes/dqn_learner.py:50-101  (learner step)
  core/memories/shared_memory.py:59-75         (locked uniform sample, CPU uint8 -> fp32 cast)
  core/models/dqn_cnn_model.py:58-78           (batch-1 actor forward, two .item() syncs)
do per iteration, on synthetic Atari-shaped data with the Nature CNN in fp32
eager PyTorch:

  learner: lock; randint indices; fancy-index 2 x (B,4,84,84) uint8 out of host
           shared memory; .float() on the CPU; unlock; 5 H2D copies from
           pageable memory; target forward + max; online forward + gather;
           MSE; backward; clip_grad_value_; Adam; hard target copy / 250;
           loss.item() every 100 steps.
  actor:   np state -> FloatTensor -> unsqueeze -> .to(cuda); forward; max;
           two .item() calls.

    python bench/reference_style.py [--steps 300] [--out bench/reference_measured.json]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class NatureCnn(nn.Module):
    def __init__(self, actions=6, norm_val=255.0):
        super().__init__()
        self.norm_val = norm_val
        self.conv = nn.Sequential(nn.Conv2d(4, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(),
                                  nn.Conv2d(64, 64, 3, 1), nn.ReLU())
        self.fc = nn.Sequential(nn.Linear(3136, 512), nn.ReLU(), nn.Linear(512, actions))

    def forward(self, x):
        return self.fc(self.conv(x / self.norm_val).flatten(1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--replay", type=int, default=50000)
    ap.add_argument("--actor-steps", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "bench", "reference_measured.json"))
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    np.random.seed(0)

    M, B = a.replay, a.batch
    # host replay in shared memory, uint8 states stored twice (s0 and sN), float32 tails
    state0s = torch.randint(0, 256, (M, 4, 84, 84), dtype=torch.uint8).share_memory_()
    state1s = torch.randint(0, 256, (M, 4, 84, 84), dtype=torch.uint8).share_memory_()
    actions = torch.randint(0, 6, (M, 1)).float().share_memory_()
    rewards = torch.randn(M, 1).clamp_(-1, 1).share_memory_()
    gamma1s = torch.full((M, 1), 0.99 ** 5).share_memory_()
    terminal1s = (torch.rand(M, 1) < 0.01).float().share_memory_()
    lock = threading.Lock()

    model, target = NatureCnn().to(dev), NatureCnn().to(dev)
    target.load_state_dict(model.state_dict())
    optim = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0.0)
    crit = nn.MSELoss()
    model.train()

    def learner_step(step):
        with lock:
            inds = np.random.randint(0, M, B)
            s0 = state0s[inds].float()
            s1 = state1s[inds].float()
            ac, rw, gm, tm = actions[inds], rewards[inds], gamma1s[inds], terminal1s[inds]
        optim.zero_grad()
        s0 = s0.cuda(non_blocking=True)
        s1 = s1.cuda(non_blocking=True)
        with torch.no_grad():
            nq = target(s1).max(dim=1, keepdim=True)[0]
            y = rw.cuda(non_blocking=True) + gm.cuda(non_blocking=True) * nq * (1 - tm.cuda(non_blocking=True))
        q = model(s0).gather(1, ac.long().cuda(non_blocking=True))
        loss = crit(q, y)
        loss.backward()
        nn.utils.clip_grad_value_(model.parameters(), float("inf"))
        optim.step()
        if step % 250 == 0:
            target.load_state_dict(model.state_dict())
        if step % 100 == 0:
            return loss.item()
        return None

    for i in range(a.warmup):
        learner_step(i + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        learner_step(i + 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ups = a.steps / dt

    # actor-side inference as the reference does it: batch 1, two host syncs per env step
    model.eval()
    state = np.random.randint(0, 256, (4, 84, 84)).astype(np.uint8)
    with torch.no_grad():
        for _ in range(50):
            q = model(torch.from_numpy(state).float().unsqueeze(0).to(dev))
            mq, ma = q.max(dim=1)
            mq.item(), ma.item()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.actor_steps):
            q = model(torch.from_numpy(state).float().unsqueeze(0).to(dev))
            mq, ma = q.max(dim=1)
            mq.item(), ma.item()
        dta = time.perf_counter() - t1
    aps = a.actor_steps / dta

    out = {"learner_updates_per_sec": round(ups, 2), "learner_ms_per_step": round(1e3 / ups, 3),
           "actor_inference_steps_per_sec_per_actor": round(aps, 1),
           "actor_inference_frames_per_sec_8_actors_upper_bound": round(aps * 8 * 4, 1),
           "batch": B, "replay": M, "dtype": "fp32", "device": torch.cuda.get_device_name(0),
           "torch": torch.__version__,
           "what": "reference-style loop (host shm replay, CPU uint8->fp32 cast under a lock, pageable H2D, "
                   "eager fp32 PyTorch, Adam, hard target update/250); see bench/reference_style.py"}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
