"""Phase stamps (shader clock cycles of workgroup 0, thread 0, LAST image of that workgroup) of the fp32-precision stack
kernels in their large-batch variants: multi-image forward, data-gradient stack with the conv1 weight-gradient tail.

    python tools/probes/x3_stack_phases_large.py [batch=4096]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) < 2:
    sys.argv.append("4096")
src = open(os.path.join(ROOT, "tools", "probes", "x3_stack_phases.py")).read().split("for name, fn, labels in")[0]
exec(src)       # the buffers and fwd() / bwd() of the batch-128 probe, at this batch

ns = nn.x3_cnn_dgrad_stack_slices(B)
gw = torch.zeros(ns, 32 * 256, device=dev)
gb = torch.zeros(ns, 32, device=dev)


def bwd_w1():
    nn.x3_cnn_dgrad_stack(da3, a2[0].hi, a1[0].hi, wT[0][1], wT[0][0], da2, da1, B, w1=(mem, idx, gw, gb, 1 / 255.))


for name, fn, labels in (
        ("forward stack", fwd, {1: "image top", 2: "patch committed + w1 fragments", 3: "conv1", 5: "a1 out + pins",
                                6: "conv2", 8: "a2 out + pins + next frames issued", 9: "conv3"}),
        ("dgrad stack + conv1 wgrad tail", bwd_w1,
         {0: "image top", 1: "w3 fragments issued", 2: "inputs issued", 3: "grids + inputs", 4: "conv3^T", 5: "da2 out",
          6: "conv2^T", 7: "da1 out", 8: "conv1 wgrad tail"})):
    prof = torch.zeros(128, dtype=torch.int64, device=dev)
    for rep in range(3):
        prof.zero_()
        L.pda_cnn_stack_prof(ctypes.c_void_p(prof.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        L.pda_cnn_stack_prof(ctypes.c_void_p(0))
    p = prof.cpu()[:16].tolist()
    keys = sorted(k for k in labels if p[k])
    print("%s (batch %d, launch %.1f us): cycles since the image's first stamp / since the previous stamp" %
          (name, B, 1e3 * e0.elapsed_time(e1)))
    prev = p[keys[0]]
    for k in keys:
        print("   %-40s %8d %8d" % (labels[k], p[k] - p[keys[0]], p[k] - prev))
        prev = p[k]
