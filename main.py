"""Launcher -- never needs editing (reference main.py).  Configure in
pytorch_distributed_amd/utils/options.py (or utils/options.py, the same module),
with KEY=VALUE arguments, or through PDA_OVERRIDES:

    python main.py                                  # mode=1 train, config 0
    python main.py config=3 num_actors=8            # Distributed DDPG on Pendulum
    python main.py mode=2                           # test ./models/MACHINE_TIMESTAMP.pth
"""
import sys

import torch.multiprocessing as mp

from pytorch_distributed_amd.launcher import run
from pytorch_distributed_amd.utils.options import Options

if __name__ == "__main__":
    mp.set_start_method("spawn", force=True)
    opt = Options(argv=[a for a in sys.argv[1:] if "=" in a])
    sup = run(opt)
    sys.exit(0 if getattr(sup, "ok", True) else 1)      # a job that lost a learner (or was stopped by one) is a failure
