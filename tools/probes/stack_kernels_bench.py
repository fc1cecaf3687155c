"""Time the per-image conv stack kernels (forward, data-gradient chain, conv1 wgrad) alone at several batch sizes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pytorch_distributed_amd.ops import nn

dev = torch.device("cuda")
bf = torch.bfloat16
mem = torch.randint(0, 256, (4096, 4, 84, 84), dtype=torch.uint8, device=dev)
w1 = (torch.randn(32, 256, device=dev) / 16).to(bf); b1 = torch.zeros(32, device=dev)
w2 = (torch.randn(64, 512, device=dev) / 22).to(bf); b2 = torch.zeros(64, device=dev)
w3 = (torch.randn(64, 576, device=dev) / 24).to(bf); b3 = torch.zeros(64, device=dev)


def timeit(fn, n=20, reps=20):
    """us per launch, from replays of a hipGraph holding n launches (python launch cost excluded)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


for B in (1, 128, 256, 512, 1024, 4096):
    idx = torch.randint(0, 4096, (B,), dtype=torch.int32, device=dev)
    a1 = torch.zeros(B * 400, 32, dtype=bf, device=dev); a2 = torch.zeros(B * 81, 64, dtype=bf, device=dev)
    a3 = torch.zeros(B * 49, 64, dtype=bf, device=dev)
    da3 = torch.randn(B * 49, 64, device=dev).to(bf)
    da2 = torch.zeros_like(a2); da1 = torch.zeros_like(a1)
    slab = torch.zeros(nn.conv1_wgrad_slices(B), 32 * 256, device=dev); bslab = torch.zeros(nn.conv1_wgrad_slices(B), 32, device=dev)
    f1 = lambda: nn.cnn_fwd_stack([mem], [idx], [w1], [b1], [w2], [b2], [w3], [b3], [a1], [a2], [a3], B, 84, 84, 4, 1 / 255.)
    f2 = lambda: nn.cnn_fwd_stack([mem, mem], [idx, idx], [w1] * 2, [b1] * 2, [w2] * 2, [b2] * 2, [w3] * 2, [b3] * 2,
                                  [a1, None], [a2, None], [a3, a3], B, 84, 84, 4, 1 / 255.)
    f3 = lambda: nn.cnn_dgrad_stack(da3, a2, a1, w3, w2, da2, da1, B)
    f4 = lambda: nn.conv_wgrad_u8(da1, mem, idx, slab, bslab, B, 84, 84, 4, 8, 8, 4, 32, 1 / 255., ksplit=0,
                                  slice_stride=32 * 256, bslice_stride=32)
    ns = nn.cnn_wgrad_stack_slices(B)
    s2, sb2 = torch.zeros(ns, 64 * 512, device=dev), torch.zeros(ns, 64, device=dev)
    s3, sb3 = torch.zeros(ns, 64 * 576, device=dev), torch.zeros(ns, 64, device=dev)
    f5 = lambda: nn.cnn_wgrad_stack(a1, a2, da2, da3, s2, sb2, s3, sb3, B)
    ks = nn.round_ksplit(max(128, (8 * B * 81 + 9 * B * 49 + 511) // 512))
    n2, n3 = nn.num_slices(B * 81, ks), nn.num_slices(B * 49, ks)
    p2, pb2 = torch.zeros(n2, 64 * 512, device=dev), torch.zeros(n2, 64, device=dev)
    p3, pb3 = torch.zeros(n3, 64 * 576, device=dev), torch.zeros(n3, 64, device=dev)
    f6 = lambda: nn.conv_wgrad_pair([da2, da3], [a1, a2], [p2, p3], [pb2, pb3], B,
                                    [(20, 20, 32, 4, 4, 2), (9, 9, 64, 3, 3, 1)], 64, ks, [64 * 512, 64 * 576], [64, 64])
    print("B=%5d  fwd(1 group) %7.1f us  fwd(2 groups) %7.1f us  dgrad stack %7.1f us  conv1 wgrad %7.1f us  "
          "conv2+3 wgrad: per-image %7.1f us (%d slices), GEMM pair %7.1f us (%d + %d slices)"
          % (B, timeit(f1), timeit(f2), timeit(f3), timeit(f4), timeit(f5), ns, timeit(f6), n2, n3), flush=True)


# ---- phase stamps of workgroup (0, 0) (shader clock cycles), forward stack
import ctypes
from pytorch_distributed_amd.ops import lib
L = lib()
L.pda_cnn_stack_prof.argtypes = [ctypes.c_void_p]
L.pda_cnn_stack_prof.restype = None
for B in (128, 1024):
    idx = torch.randint(0, 4096, (B,), dtype=torch.int32, device=dev)
    a1 = torch.zeros(B * 400, 32, dtype=bf, device=dev); a2 = torch.zeros(B * 81, 64, dtype=bf, device=dev)
    a3 = torch.zeros(B * 49, 64, dtype=bf, device=dev)
    prof = torch.zeros(128, dtype=torch.int64, device=dev)
    for rep in range(2):
        L.pda_cnn_stack_prof(ctypes.c_void_p(prof.data_ptr()))
        nn.cnn_fwd_stack([mem, mem], [idx, idx], [w1] * 2, [b1] * 2, [w2] * 2, [b2] * 2, [w3] * 2, [b3] * 2,
                         [a1, None], [a2, None], [a3, a3], B, 84, 84, 4, 1 / 255.)
        torch.cuda.synchronize()
        L.pda_cnn_stack_prof(ctypes.c_void_p(0))
    p = prof.cpu()[:64].view(4, 16)
    names = ["loop top", "committed", "conv1 done", "barrier", "a1 out", "pinned", "conv2 done", "barrier", "a2 out", "conv3 done"]
    for it in range(4):
        if p[it, 0] == 0:
            continue
        row = p[it, :10] - p[0, 0]
        print("B=%d image %d: " % (B, it) + "  ".join("%s %d" % (n, int(v)) for n, v in zip(names, row)), flush=True)
