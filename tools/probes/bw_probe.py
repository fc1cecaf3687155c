import torch
dev = "cuda"
def timeit(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
for mb in (7, 14, 27, 54, 216):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    t3 = timeit(lambda: torch.add(x, y, out=z))
    print("copy %3d MB -> %3d MB: %6.1f us = %.2f TB/s;   z = x + y (3 x %d MB): %6.1f us = %.2f TB/s" % (mb, mb, t, 2 * mb / 1e6 * 1.048576 / t * 1e6 / 1e6, mb, t3, 3 * mb * 1.048576 / t3 / 1e6 * 1e6 / 1e6), flush=True)
