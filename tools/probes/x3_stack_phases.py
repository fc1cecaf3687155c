"""Phase timeline (shader clock cycles of workgroup 0, thread 0) of the fp32-precision stack kernels, batch 128."""
import ctypes
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pytorch_distributed_amd.ops import nn, lib

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = lib()
L.pda_cnn_stack_prof.argtypes = [ctypes.c_void_p]
L.pda_cnn_stack_prof.restype = None


class View(object):
    def __init__(self, hi, plane):
        self.hi, self.plane = hi, plane

    def data_ptr(self):
        return self.hi.data_ptr()


def shadow(shapes):
    total = sum(r * c for r, c in shapes)
    buf = (torch.randn(2, total, device=dev) / 20).to(torch.bfloat16)
    out, off = [], 0
    for r, c in shapes:
        out.append(View(buf[0, off:off + r * c].view(r, c), total))
        off += r * c
    return out, buf


mem = torch.randint(0, 256, (4096, 4, 84, 84), dtype=torch.uint8, device=dev)
idx = torch.randint(0, 4096, (B,), dtype=torch.int32, device=dev)
nets = [shadow([(32, 256), (64, 512), (64, 576)]) for _ in range(2)]
wT = shadow([(512, 64), (576, 64)])
bs = [torch.zeros(32, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)]
a1 = [nn.Planes(B * 400, 32, device=dev), None]
a2 = [nn.Planes(B * 81, 64, device=dev), None]
a3 = [nn.Planes(B * 49, 64, device=dev) for _ in range(2)]
da3 = nn.Planes.from_float(torch.randn(B * 49, 64, device=dev))
da2, da1 = nn.Planes(B * 81, 64, device=dev), nn.Planes(B * 400, 32, device=dev)


def fwd():
    nn.x3_cnn_fwd_stack([mem, mem], [idx, idx], [n[0][0] for n in nets], [bs[0]] * 2, [n[0][1] for n in nets],
                        [bs[1]] * 2, [n[0][2] for n in nets], [bs[2]] * 2, a1, a2, a3, B, 84, 84, 4, 1 / 255.)


def bwd():
    nn.x3_cnn_dgrad_stack(da3, a2[0].hi, a1[0].hi, wT[0][1], wT[0][0], da2, da1, B)


for name, fn, labels in (
        ("forward stack", fwd, ["start", "loads issued + offsets", "patch committed", "conv1", "-", "a1 out + pins",
                                "conv2", "-", "a2 out + pins", "conv3"]),
        ("dgrad stack", bwd, ["start", "w3 fragments issued", "inputs issued", "grids + inputs", "conv3^T", "da2 out",
                              "conv2^T", "da1 out"])):
    prof = torch.zeros(128, dtype=torch.int64, device=dev)
    for rep in range(3):
        prof.zero_()
        L.pda_cnn_stack_prof(ctypes.c_void_p(prof.data_ptr()))
        fn()
        torch.cuda.synchronize()
        L.pda_cnn_stack_prof(ctypes.c_void_p(0))
    p = prof.cpu()[:16].tolist()
    if name == "forward stack":
        print("   prologue: w1 chunks + frames issued %d, w2 issued %d, w1 staged through LDS %d (cycles since start)"
              % tuple(p[i] - p[0] for i in (10, 12, 13)))
    print(name, "(batch %d), cycles since start / since previous stamp:" % B)
    prev = p[0]
    for i, lab in enumerate(labels):
        if lab == "-" or p[i] == 0:
            continue
        print("   %-26s %8d %8d" % (lab, p[i] - p[0], p[i] - prev))
        prev = p[i]
