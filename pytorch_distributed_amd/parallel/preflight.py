"""Pre-flight check for graph-captured collectives.

The data-parallel learner step is fastest as ONE hipGraph that contains the RCCL
all-reduces on a forked stream (ops/dqn_engine.py:DqnStepProgram).  A runtime that
refuses to capture a collective leaves the capturing process unusable (the capture
stays "invalidated" for the thread), and a replay that hangs cannot be interrupted
from inside.  So the question "does whole-step capture work on this node?" is
answered by a throw-away child process per rank, BEFORE the parent touches the
GPU: the children form their own process group on a neighbouring port, capture an
all-reduce on a forked stream, replay it and check the sum.  The parent only
reads the exit code (with a timeout) and sets PDA_DP_GRAPH=whole, or leaves it at
auto: the process group's collectives then stay host-driven ("split" graphs), while a
step whose all-reduce is the peer kernel -- a plain launch -- is still ONE graph.  (A
failed or timed-out pre-flight used to answer "split", which also split the peer
kernel's step: half the throughput for a question that never concerned it.)

    parent:  mode = preflight.choose_dp_graph_mode()      # before any HIP call
    child:   python -m pytorch_distributed_amd.parallel.preflight
"""
import os
import subprocess
import sys
import time


def _child():
    import datetime
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PDA_PREFLIGHT_BACKEND", "nccl")
    dev = torch.device("cuda:%d" % (local % max(torch.cuda.device_count(), 1)))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60), device_id=dev if backend == "nccl" else None)
    buf = torch.ones(1 << 20, dtype=torch.float32, device=dev)
    small = torch.ones(1 << 12, dtype=torch.float32, device=dev)
    dist.all_reduce(buf)            # eager first: communicator and channels set up outside the capture
    dist.all_reduce(small)
    torch.cuda.synchronize()
    buf.fill_(1.0)
    small.fill_(1.0)
    main_s, side_s = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    e0, e1, e2, e3 = (torch.cuda.Event() for _ in range(4))
    gr = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=main_s, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        buf.mul_(1.0)
        e0.record(cur)
        with torch.cuda.stream(side_s):
            side_s.wait_event(e0)
            dist.all_reduce(buf)
            e1.record(side_s)
        small.mul_(1.0)
        e2.record(cur)
        with torch.cuda.stream(side_s):
            side_s.wait_event(e2)
            dist.all_reduce(small)
            e3.record(side_s)
        cur.wait_event(e1)
        buf.mul_(0.5)
        cur.wait_event(e3)
        small.mul_(0.5)
    reps = 3
    for _ in range(reps):
        gr.replay()
    torch.cuda.synchronize()
    want = (world * 0.5) ** reps
    ok = abs(float(buf[12345].item()) - want) < 1e-3 * want and abs(float(small[77].item()) - want) < 1e-3 * want
    dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 3


def choose_dp_graph_mode(timeout_s=150, backend="nccl", port_offset=23):
    """Run the child for this rank; returns "whole" (collectives of the process group may be captured) or "auto"
    (they may not: ops/dqn_program.py:_capture_dp then splits the graph around them, unless the all-reduce is the
    peer kernel) -- or the knob's value when it was set explicitly.  Call before the parent initialises the GPU."""
    from ..ops.knobs import resolve
    forced = resolve().dp_graph
    forced = None if forced == "auto" else forced
    if forced in ("whole", "split"):
        return forced
    env = dict(os.environ)
    env["PDA_PREFLIGHT_BACKEND"] = backend
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + port_offset)
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RESTART_COUNT",
              "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_ERROR_FILE"):
        env.pop(k, None)        # a plain env:// TCP store of its own, not torchrun's agent store
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    t0 = time.time()
    try:
        proc = subprocess.Popen([sys.executable, "-m", "pytorch_distributed_amd.parallel.preflight"],
                                env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    except OSError:
        return "auto"
    try:
        out, _ = proc.communicate(timeout=timeout_s)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        rc = -9
    mode = "whole" if rc == 0 else "auto"
    if rc != 0 and int(os.environ.get("RANK", "0")) == 0:
        tail = out.decode(errors="replace").strip().splitlines()[-3:] if out else []
        print("[preflight] graph-captured collectives unavailable (rc=%s, %.0fs): %s" %
              (rc, time.time() - t0, " | ".join(tail)), file=sys.stderr, flush=True)
    return mode


if __name__ == "__main__":
    sys.exit(_child())
