"""Replay memory base: feed(experience, priority) / sample(batch_size) / size
(reference core/memory.py:4-32 and core/memories/shared_memory.py:39-75).

A transition is the 6-tuple (state0, action, reward, gamma^n, stateN, terminalN).
Two additions finish the reference's prioritized-replay TODO:
`sample_with_priorities` returns (batch, indices, importance weights) and
`update_priorities` writes |TD error| back.  Uniform memories return unit
weights and ignore the write-back, so learners have a single code path.
"""
import torch


class Memory(object):
    def __init__(self, args):
        self.state_shape = args.state_shape
        self.action_shape = args.action_shape
        self.reward_shape = args.reward_shape       # num_tasks
        self.gamma_shape = args.gamma_shape         # num_tasks
        self.terminal_shape = args.terminal_shape   # num_tasks

        self.memory_size = args.memory_size
        self.tensortype = args.tensortype
        self.state_dtype = torch.uint8 if args.tensortype == torch.ByteTensor else torch.float32

        self.enable_per = bool(getattr(args, "enable_per", False))
        self.priority_exponent = getattr(args, "priority_exponent", 0.5)
        self.priority_weight = getattr(args, "priority_weight", 0.4)
        self.priority_eps = getattr(args, "priority_eps", 1e-6)

        self.pos = 0
        self.full = False

    @property
    def size(self):
        if self.full:
            return self.memory_size
        return self.pos

    def feed(self, experience, priority=0.0):
        raise NotImplementedError("not implemented in base class")

    append = feed   # the reference's base class calls it `append`, subclasses `feed`

    def sample(self, batch_size):
        raise NotImplementedError("not implemented in base class")

    def sample_with_priorities(self, batch_size):
        batch = self.sample(batch_size)
        return batch, None, torch.ones(batch_size, 1)

    def update_priorities(self, indices, td_abs):
        pass
