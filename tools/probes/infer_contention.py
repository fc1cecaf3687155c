#!/usr/bin/env python3
"""What batched actor inference costs the learner when both share one MI355X -- in ONE process, so that
`rocprofv3 --kernel-trace -- python tools/probes/infer_contention.py` can see every kernel (the real job is a tree
of spawned processes, which the profiler cannot follow on this pool).

The learner is the benchmark's (config 2, Nature-CNN DQN, batch 128, HBM replay, five updates per hipGraph); the
inference server is the native loop thread hosted in this process exactly as `inference_placement="learner"` does
(high-priority stream, the learner's live shadow weights); a feeder thread keeps `--slots` mailbox slots posted the
way the actors do (one request in flight per env, `--think-us` of "env step" between an answer and the next request).

    python tools/probes/infer_contention.py [--slots 16] [--think-us 0] [--seconds 3] [--precision fp32]

Prints one JSON line: learner updates/s alone, with the server idle, and with the feeders running; batches/s and
us/batch of the server; the slowdown per inference batch.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _push_forever(mem, stop, seed):
    """Feeder process: pushes synthetic transitions into the replay's shm ring as fast as the ring takes them."""
    import numpy as np
    rng = np.random.RandomState(seed)
    s0 = rng.randint(0, 256, (4, 84, 84), dtype=np.uint8)
    while not stop.value:
        mem.feed((s0, np.array([1.0]), [0.5], [0.95], s0, 0.0), 1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--think-us", type=float, default=0.0)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--window-us", type=int, default=40)
    ap.add_argument("--feeders", type=int, default=2, help="feeder threads (slots are split between them)")
    ap.add_argument("--ingest-threads", type=int, default=0,
                    help="feeder PROCESSES (started before this process touches the GPU) that push transitions into the "
                         "replay's shm ring as fast as they can; the learner loop drains it into HBM between updates, "
                         "as in the job")
    ap.add_argument("--no-inference", action="store_true")
    ap.add_argument("--only", default=None, choices=[None, "alone", "inference"],
                    help="run a single phase (for rocprofv3 --kernel-trace --stats: per-kernel times of that phase)")
    a = ap.parse_args()

    import numpy as np
    import torch
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.memories.device_memory import DeviceMemory
    from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
    from pytorch_distributed_amd.ops.dqn_engine import DqnNativeCore
    from pytorch_distributed_amd.ops.infer import make_native_server
    from pytorch_distributed_amd.parallel.inference_server import InferenceMailbox
    from pytorch_distributed_amd.utils.options import Options

    opt = prepare_options(Options(config=2, overrides={"precision": a.precision, "memory_params.memory_size": 16384,
                                                       "inference_window_us": a.window_us}))
    mem = DeviceMemory(opt.memory_params)
    feeders, stop_flag = [], None
    if a.ingest_threads > 0:        # children first: nothing GPU-related exists in this process yet
        from pytorch_distributed_amd.utils import mpctx
        stop_flag = mpctx.RawValue("b", False)
        feeders = [mpctx.Process(target=_push_forever, args=(mem, stop_flag, 11 + i), daemon=True)
                   for i in range(a.ingest_threads)]
        for p_ in feeders:
            p_.start()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mem.use_pump = a.ingest_threads > 0
    mem.attach(dev)
    n = 4096
    for _ in range(4):
        s = torch.randint(0, 256, (n + 1, 4, 84, 84), dtype=torch.uint8, device=dev)
        mem.feed_batch(s[:-1], torch.randint(0, 6, (n, 1), device=dev), torch.randn(n, 1, device=dev).clamp_(-1, 1),
                       torch.full((n, 1), 0.95, device=dev), s[1:], torch.zeros(n, 1, device=dev))
    model = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), 6, 1).to(dev)
    target = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), 6, 1).to(dev)
    core = DqnNativeCore(opt, model, target, mem)

    ingested = [0]

    def learner_rate(seconds):
        for _ in range(50):
            core.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(100):
                if a.ingest_threads > 0:
                    ingested[0] += mem.drain()
                core.step()
            k += 100
            torch.cuda.synchronize()
        return k / (time.perf_counter() - t0)

    out = {"precision": a.precision, "slots": a.slots, "think_us": a.think_us, "window_us": a.window_us}
    if a.only != "inference":
        out["learner_alone"] = round(learner_rate(a.seconds), 1)
    if a.only == "alone":
        print(json.dumps(out), flush=True)
        mem.close()
        return

    if a.ingest_threads > 0:
        n0, t0 = ingested[0], time.perf_counter()
        out["learner_with_ingest"] = round(learner_rate(a.seconds), 1)
        out["ingest_transitions_per_sec"] = round((ingested[0] - n0) / (time.perf_counter() - t0), 1)
        if a.no_inference:
            stop_flag.value = True
            print(json.dumps(out), flush=True)
            for p_ in feeders:
                p_.join(3)
                if p_.is_alive():
                    p_.terminate()
            mem.close()
            return

    mailbox = InferenceMailbox(a.slots, (4, 84, 84), torch.uint8, 1, discrete=True)
    for i in range(a.slots):
        mailbox.set_epsilon(i, 0.1)
    server = make_native_server(opt, model, mailbox, dev, high_priority=True, weights=core.engine.online)
    server.start()
    if a.only is None:
        out["learner_with_idle_server"] = round(learner_rate(a.seconds), 1)

    stop = threading.Event()
    frames = np.random.randint(0, 256, (a.slots, 4, 84, 84), dtype=np.uint8)

    def feed(lo, hi):
        pend = {i: mailbox.send(i, frames[i]) for i in range(lo, hi)}
        think = a.think_us * 1e-6
        while not stop.is_set():
            for i in range(lo, hi):
                mailbox.wait(i, pend[i])
                if think > 0:
                    t = time.perf_counter() + think
                    while time.perf_counter() < t:
                        pass
                pend[i] = mailbox.send(i, frames[i])

    per = (a.slots + a.feeders - 1) // a.feeders
    threads = [threading.Thread(target=feed, args=(k * per, min(a.slots, (k + 1) * per)), daemon=True)
               for k in range(a.feeders)]
    for t in threads:
        t.start()
    time.sleep(0.3)
    st0 = server.stats()
    t0 = time.perf_counter()
    out["learner_with_inference"] = round(learner_rate(a.seconds), 1)
    dt = time.perf_counter() - t0
    st1 = server.stats()
    stop.set()
    mailbox.closed[0] = 1
    for t in threads:
        t.join(2)
    server.stop()
    nb = st1["batches"] - st0["batches"]
    out["inference_batches_per_sec"] = round(nb / dt, 1)
    out["inference_requests_per_sec"] = round((st1["requests"] - st0["requests"]) / dt, 1)
    out["mean_batch"] = round((st1["requests"] - st0["requests"]) / max(nb, 1), 2)
    out["us_per_batch_host"] = round((st1["busy_ns"] - st0["busy_ns"]) / 1e3 / max(nb, 1), 1)
    if "learner_alone" in out:
        la, li = out["learner_alone"], out["learner_with_inference"]
        out["learner_us_per_step"] = [round(1e6 / la, 2), round(1e6 / li, 2)]
        out["learner_us_lost_per_inference_batch"] = round((1e6 / li - 1e6 / la) * li / max(nb / dt, 1), 2)
    print(json.dumps(out), flush=True)
    server.close()
    if stop_flag is not None:
        stop_flag.value = True
        for p_ in feeders:
            p_.join(3)
            if p_.is_alive():
                p_.terminate()
    mem.close()


if __name__ == "__main__":
    main()
