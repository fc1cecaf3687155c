"""Phase timeline (shader clock cycles, workgroup 0 thread 0) of the bf16 forward stack kernel at batch 128."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pytorch_distributed_amd.ops import nn, lib
dev = torch.device("cuda"); bf = torch.bfloat16
B = 128
mem = torch.randint(0, 256, (4096, 4, 84, 84), dtype=torch.uint8, device=dev)
w1 = (torch.randn(32, 256, device=dev) / 16).to(bf); b1 = torch.zeros(32, device=dev)
w2 = (torch.randn(64, 512, device=dev) / 22).to(bf); b2 = torch.zeros(64, device=dev)
w3 = (torch.randn(64, 576, device=dev) / 24).to(bf); b3 = torch.zeros(64, device=dev)
L = lib()
L.pda_cnn_stack_prof.argtypes = [ctypes.c_void_p]; L.pda_cnn_stack_prof.restype = None
idx = torch.randint(0, 4096, (B,), dtype=torch.int32, device=dev)
a1 = torch.zeros(B * 400, 32, dtype=bf, device=dev); a2 = torch.zeros(B * 81, 64, dtype=bf, device=dev)
a3 = torch.zeros(B * 49, 64, dtype=bf, device=dev)
prof = torch.zeros(128, dtype=torch.int64, device=dev)
for rep in range(3):
    prof.zero_()
    L.pda_cnn_stack_prof(ctypes.c_void_p(prof.data_ptr()))
    nn.cnn_fwd_stack([mem, mem], [idx, idx], [w1] * 2, [b1] * 2, [w2] * 2, [b2] * 2, [w3] * 2, [b3] * 2,
                     [a1, None], [a2, None], [a3, a3], B, 84, 84, 4, 1 / 255.)
    torch.cuda.synchronize()
    L.pda_cnn_stack_prof(ctypes.c_void_p(0))
p = prof.cpu()[:16].tolist()
names = ["loop top", "committed", "conv1 done", "barrier", "a1 out", "pinned", "conv2 done", "barrier", "a2 out", "conv3 done"]
print("bf16 forward stack, batch %d; prologue (cycles before the loop top): kernel entry %d, loads issued %d, w1 staged %d" % (B, p[10] - p[0], p[11] - p[0], p[12] - p[0]))
for i, n in enumerate(names):
    print("   %-12s %8d %8d" % (n, p[i] - p[0], p[i] - p[max(i - 1, 0)]))

# ---- data-gradient stack (+ conv1 weight gradient)
da3 = torch.randn(B * 49, 64, device=dev).to(bf)
da2 = torch.zeros_like(a2); da1 = torch.zeros_like(a1)
ns = nn.cnn_dgrad_stack_slices(B)
slab = torch.zeros(ns, 32 * 256, device=dev); bslab = torch.zeros(ns, 32, device=dev)
for rep in range(3):
    prof.zero_()
    L.pda_cnn_stack_prof(ctypes.c_void_p(prof.data_ptr()))
    nn.cnn_dgrad_stack(da3, a2, a1, w3, w2, da2, da1, B, w1=(mem, idx, slab, bslab, 1 / 255.))
    torch.cuda.synchronize()
    L.pda_cnn_stack_prof(ctypes.c_void_p(0))
p = prof.cpu()[:16].tolist()
names = ["start", "weights staged (LDS)", "fragments + offsets", "inputs committed", "conv3^T", "da2 out", "conv2^T",
         "da1 out", "conv1 wgrad"]
print("bf16 dgrad stack + conv1 wgrad, batch %d" % B)
prev = p[0]
for i, n in enumerate(names):
    print("   %-22s %8d %8d" % (n, p[i] - p[0], p[i] - prev))
    prev = p[i]
