"""Scalar sink for the logger process: tensorboard when a writer is importable
(tensorboardX or torch.utils.tensorboard), and always a JSONL file
`<log_dir>/scalars.jsonl` so runs on a box without tensorboard keep their
curves (the reference hard-requires tensorboardX, dqn_logger.py:2,15).
"""
import json
import os
import time


def _make_tb_writer(log_dir):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(log_dir)
    except Exception:
        pass
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir)
    except Exception:
        return None


class Board(object):
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.tb = _make_tb_writer(log_dir)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self.fh = open(self.path, "a", buffering=1)

    def add_scalar(self, tag, value, step):
        if self.tb is not None:
            self.tb.add_scalar(tag, value, step)
        self.fh.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "time": time.time()}) + "\n")

    def close(self):
        if self.tb is not None:
            self.tb.close()
        self.fh.close()


def read_scalars(log_dir):
    path = os.path.join(log_dir, "scalars.jsonl")
    if not os.path.exists(path):
        return []
    with open(path) as fh:
        return [json.loads(line) for line in fh if line.strip()]
