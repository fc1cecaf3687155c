"""Headline benchmark: Distributed-DQN learner throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms run N ranks.  Under torchrun (RANK / WORLD_SIZE in the environment) this process IS one
rank.  Without them and with --gpus N > 1, this process becomes a launcher that never touches the GPU:
it starts N fresh children (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), waits for them
with a timeout, forwards rank 0's single JSON line and exits non-zero if any rank failed -- the
equivalent of the reference's `num_learners` loop (main.py:82-94), with real ranks instead of Hogwild
writers.

Config (BASELINE.json config 2, the reference's README Pong setting): Nature-CNN DQN, 6 actions, batch
128 per learner GPU, n-step 5, Adam lr 1e-4, hard target update every 250 steps, uniform replay of
50 000 transitions resident in HBM as uint8, synthetic Atari-shaped observations (4x84x84 uint8),
random-init weights.  One learner process per GPU; N > 1 = synchronous data parallel (weak scaling: 128
transitions per GPU per step, all-reduce of the flat gradient over RCCL / the xGMI peer all-reduce).

Every timed step is a complete learner update: replay sampling, target + online forward, TD
target/loss, full backward, (all-reduce,) Adam, target update.  The timed window is EXACTLY --steps
updates between barrier + synchronize pairs; because the default window is short, it is repeated until
at least --min-time seconds have been measured and the MEDIAN window is reported (min / max beside it).
Rank 0 prints ONE JSON line.  `value` = learner updates/s summed over GPUs (one update = one batch of
128 transitions through one optimizer step on one GPU) at the precision named in `dtype`;
`fp32_value` / `bf16_value` are the same config on the fp32-equivalent (split-bf16 x3 MFMA products,
fp32 accumulation: gradients within 1e-3 of an fp32 PyTorch step) and on the plain bf16 engine.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# reference-style learner loop (host shm replay -> fp32 cast -> H2D -> eager fp32 PyTorch),
# measured on one MI355X with bench/reference_style.py; see BASELINE.md
BASELINE_FILE = os.path.join(ROOT, "bench", "reference_measured.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=128, help="transitions per GPU per step")
    ap.add_argument("--replay", type=int, default=50000)
    ap.add_argument("--config", type=int, default=2, help="options.CONFIGS row (2 uniform, 4 Ape-X PER)")
    ap.add_argument("--memory-type", default=None,
                    help="replay kind override, e.g. device-dedup / device-dedup-per (frame-deduplicated HBM replay: "
                         "every frame stored once, memories/dedup_memory.py)")
    ap.add_argument("--compute", default="native", choices=["native", "torch"])
    ap.add_argument("--precision", default="both", choices=["both", "fp32", "bf16"],
                    help="native engine arithmetic: fp32 = split-bf16 x3 products (fp32-equivalent, the headline), "
                         "bf16 = plain bf16 operands; both = time the two and report both")
    ap.add_argument("--device", default=None, help="cuda | cpu (default: cuda when present)")
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="repeat the --steps window until this many seconds are measured (0 = one window)")
    ap.add_argument("--max-windows", type=int, default=400)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default=None, help="process-group backend override (default: nccl = RCCL)")
    ap.add_argument("--allreduce", default=None, choices=[None, "auto", "rccl", "peer"],
                    help="gradient all-reduce of the data-parallel step (default auto: xGMI peer kernel when every "
                         "rank can map its peers' buffers, RCCL otherwise)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel step program even with one rank (rehearsal of the collective path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks on cuda:0 (rehearsal of the multi-rank program on a one-GPU box)")
    ap.add_argument("--graph-steps", type=int, default=None,
                    help="learner updates per hipGraph replay (default: the option's, 5; 1 = one graph per step)")
    ap.add_argument("--actor-bench", action="store_true", help="also time batched actor inference (untimed extra)")
    ap.add_argument("--node", default="auto", choices=["auto", "on", "off"],
                    help="after the timed learner windows, also run the whole job (logger + actors + inference server + "
                         "learner + evaluator, bench/node_throughput.py) for a bounded number of learner steps and "
                         "report env frames/s + learner updates/s measured inside it under \"node\" "
                         "(auto: on a GPU with --gpus 1)")
    ap.add_argument("--node-steps", type=int, default=200000,
                    help="learner steps of the --node job (200 k = >= 20 s of steady learning = >= 10 logger windows of "
                         "2 s, whose mean / std / min / max are reported)")
    ap.add_argument("--node-actors", type=int, default=8)
    ap.add_argument("--node-envs-per-actor", type=int, default=16,
                    help="envs per actor process, one inference request in flight per env: an actor process at 4 envs "
                         "spends 63 %% of its time waiting for answers; 4 / 8 / 16 / 32 envs give 0.68 / 0.98 / 1.43 / 1.82 M "
                         "env frames/s on 16 host cores with the learner at 9.88 / 9.74 / 9.75 / 9.63 k updates/s "
                         "(profiles/r3_node_envs_per_actor_sweep.txt)")
    ap.add_argument("--node-memory", default="device-dedup",
                    help="replay kind of the --node job: device-dedup (every frame stored once: 1.11 planes = 7.9 KB per "
                         "transition in the job instead of 56.5 KB; 307 s soak profiles/r3_node_soak_dedup_3M_steps.json) "
                         "| device (both frame stacks in full) | config (what the options row names)")
    ap.add_argument("--node-timeout", type=float, default=240.0)
    ap.add_argument("--ddpg", default="auto", choices=["auto", "on", "off"],
                    help="also time the Distributed-DDPG learner step (config 7 shapes: 3-dim obs, 1-dim action, batch "
                         "64, HBM replay) on the fp32 job-list engine and the fused bf16 engine: \"ddpg_value\" "
                         "(fp32, the reference's precision) and \"ddpg_bf16_value\", updates/s (auto: 1 GPU, native)")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"],
                    help="also run the other BASELINE.json learner configs as short child benches and report them as "
                         "extra keys: \"per_value\" (config 4: Ape-X prioritized replay, batch 128) and "
                         "\"b4096_value\" (config 5: batch 4096, prioritized), fp32-class engine, updates/s "
                         "(auto: 1 GPU, native, --config 2)")
    ap.add_argument("--launcher", action="store_true", help="start the rank(s) as child processes even for --gpus 1")
    ap.add_argument("--launch-timeout", type=float, default=900.0)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: a parent that never initialises the GPU
# ---------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    """Start a.gpus rank processes of this script, forward rank 0's JSON line.  Returns the exit code."""
    n = int(a.gpus)
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / peer mappings across processes
    env0["MASTER_ADDR"] = "127.0.0.1"
    env0["MASTER_PORT"] = str(_free_port())
    env0["WORLD_SIZE"] = str(n)
    env0["PDA_BENCH_LAUNCHED"] = "1"
    procs = []
    for r in range(n):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()          # rank 0's pipe is drained as it fills, so it can never block on a full pipe
    deadline = time.time() + float(a.launch_timeout)
    rc, failed = 0, None
    pending = list(range(n))
    while pending and failed is None:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.remove(r)
                if code != 0:
                    failed = (r, code)
        if time.time() > deadline:
            failed = (-1, 124)
        if pending and failed is None:
            time.sleep(0.1)
    if failed is not None:
        rc = failed[1] if failed[1] > 0 else 1
        who = "timeout after %.0f s" % a.launch_timeout if failed[0] < 0 else "rank %d exited with %s" % failed
        print("[bench] %s: stopping the other ranks" % who, file=sys.stderr, flush=True)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=15)
    out = b"".join(chunks).decode(errors="replace")
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    for ln in out.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if rc == 0 and len(lines) != 1:
        print("[bench] expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        rc = 1
    for ln in lines:
        print(ln, flush=True)
    return rc


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------
def _median(xs):
    s = sorted(xs)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def run_node_job(a, precision, device):
    """bench/node_throughput.py as a child process (its launcher spawns the job's processes); -> its JSON dict,
    or {"error": ...}: the learner number of this bench stays valid whatever happens here."""
    import signal
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench", "node_throughput.py"), "--config", str(a.config),
           "--steps", str(a.node_steps), "--actors", str(a.node_actors),
           "--envs-per-actor", str(a.node_envs_per_actor), "--precision", precision,
           "--device", "cuda" if str(device).startswith("cuda") else "cpu"]
    if a.node_memory != "config" and a.config in (2, 8):
        cmd += ["--set", "memory_type=%s" % a.node_memory]
    elif a.node_memory != "config" and a.config in (4, 5, 9):
        cmd += ["--set", "memory_type=%s-per" % a.node_memory]
    try:
        env = dict(os.environ)
        env["PDA_LEARNER_PROF"] = "1"       # where the in-job learner's time goes (agents/learners.py _LoopProfile)
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT,
                             start_new_session=True, env=env)
        try:
            stdout, _ = p.communicate(timeout=a.node_timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)        # the session this call started, nothing else
            p.communicate()
            return {"error": "node job did not finish within %.0f s" % a.node_timeout}
        for line in reversed(stdout.splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                keep = ("num_actors", "num_envs_per_actor", "actor_inference", "precision", "learner_steps", "wall_s",
                        "learner_updates_per_sec_steady", "env_frames_per_sec_mean", "env_frames_per_sec_peak",
                        "actor_steps_per_sec_mean", "inference", "exitcodes", "actor_restarts", "env", "memory",
                        "learner_updates_per_sec_windows", "env_frames_per_sec_windows", "logger_window_s",
                        "learner_loop_profile", "replay_stats")
                d = {k: d.get(k) for k in keep}
                d["env_frames_per_sec"] = d.pop("env_frames_per_sec_mean")
                d["learner_updates_per_sec_in_job"] = d.pop("learner_updates_per_sec_steady")
                return d
        return {"error": "node job printed no result (exit %s)" % p.returncode}
    except Exception as exc:
        return {"error": repr(exc)}


def run_ddpg_step(precision, timeout=120.0):
    """tools/bench_mlp_step.py in a child process -> its JSON line (dict), or {"error": ...}."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_mlp_step.py"), "--precision", precision,
           "--steps", "20000", "--warmup", "500"]
    try:
        p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "ddpg step bench did not finish within %.0f s" % timeout}
    for line in reversed(p.stdout.decode(errors="replace").splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": "ddpg step bench printed no result (exit %s): %s" % (p.returncode,
                                                                          p.stderr.decode(errors="replace")[-300:])}


def run_child_bench(extra_args, timeout=240.0):
    """This script once more, as a child, for another config -> its JSON line (dict) or {"error": ...}."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--node", "off", "--ddpg", "off", "--extras", "off",
           "--precision", "fp32"] + list(extra_args)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "child bench did not finish within %.0f s" % timeout}
    for line in reversed(p.stdout.decode(errors="replace").splitlines()):
        if line.startswith("{"):
            r = json.loads(line)
            r["exit_code"] = p.returncode
            return r
    return {"error": "child bench printed no result (exit %s): %s" % (p.returncode,
                                                                      p.stderr.decode(errors="replace")[-300:])}


def timed_windows(core, comm, steps, min_time, max_windows, sync, reduce_max):
    """Repeat [barrier+sync | `steps` updates | sync+barrier] until min_time seconds are measured.
    Every rank sees the same (MAX over ranks) window times, so all take the same decision."""
    times = []
    while True:
        sync()
        comm.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            core.step()
        sync()
        comm.barrier()
        sync()
        times.append(reduce_max(time.perf_counter() - t0))
        if sum(times) >= min_time or len(times) >= max_windows:
            return times


def run_rank(a):
    # stdout carries ONE line, the JSON: libraries write there too (RCCL prints its version banner on the first
    # communicator), so file descriptor 1 points at stderr from here on and the result goes out through a copy
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    if os.environ.get("PDA_BENCH_FAULTHANDLER"):     # debugging aid: dump every thread's stack if we stall
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PDA_BENCH_FAULTHANDLER"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PDA_BENCH_FAIL_RANK") == str(rank) and world > 1:     # launcher test hook: this rank dies
        os._exit(17)
    if world != a.gpus:
        # a caller that asked for N GPUs must never get a number measured on another count
        raise SystemExit("[bench] --gpus %d but WORLD_SIZE=%d: launch with matching values" % (a.gpus, world))
    use_cuda = (a.device or "cuda") == "cuda" and torch.cuda.is_available()
    if a.compute == "native" and not use_cuda:
        raise SystemExit("[bench] --compute native needs a GPU (use --compute torch --device cpu for the CPU path)")
    dp = world > 1 or a.force_dp
    backend = a.backend or ("nccl" if use_cuda else "gloo")
    if dp and backend == "nccl" and a.compute == "native" and not a.no_graph:
        # ask a throw-away child whether this node captures RCCL collectives into a hipGraph
        # (must happen before this process initialises the GPU); sets the DP step's graph mode
        from pytorch_distributed_amd.parallel.preflight import choose_dp_graph_mode
        os.environ["PDA_DP_GRAPH"] = choose_dp_graph_mode()
    if a.allreduce:
        os.environ["PDA_DP_ALLREDUCE"] = a.allreduce
    if use_cuda:
        device = torch.device("cuda:%d" % (0 if a.share_gpu else local))
        torch.cuda.set_device(device)
        sync = torch.cuda.synchronize
    else:
        device = torch.device("cpu")
        torch.set_num_threads(max(1, min(4, (os.cpu_count() or 4) // max(world, 1))))
        sync = lambda: None     # noqa: E731

    from pytorch_distributed_amd.utils.options import Options
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.utils.factory import MemoriesDict, ModelsDict
    from pytorch_distributed_amd.parallel.comm import comm_from_env
    from pytorch_distributed_amd.agents.learners import make_core

    comm = comm_from_env(device, backend=backend, force=a.force_dp)
    precisions = ["fp32", "bf16"] if a.precision == "both" else [a.precision]
    if a.compute != "native":
        precisions = ["fp32"]

    def options(precision):
        opt = prepare_options(Options(config=a.config, overrides={
            **({"memory_type": a.memory_type} if a.memory_type else {}),
            "device": "cuda" if use_cuda else "cpu", "compute": a.compute, "num_learners": world,
            "force_dp": a.force_dp, "precision": precision,
            "memory_params.replay_device": "cuda" if use_cuda else "cpu",
            "memory_params.memory_size": a.replay * world,      # each rank keeps a `replay`-sized shard
            "agent_params.batch_size": a.batch}))
        if a.graph_steps is not None:
            opt.graph_steps = int(a.graph_steps)
        else:
            # the learner replays several updates per hipGraph and runs ahead of its caller (options.graph_steps):
            # pick a count that divides --steps, so that EXACTLY that many updates execute inside a timed window
            opt.graph_steps = max(d for d in range(1, int(opt.graph_steps) + 1) if a.steps % d == 0)
        return opt

    opt = options(precisions[0])
    ap_ = opt.agent_params
    memory = MemoriesDict[opt.memory_type](opt.memory_params)
    memory.attach(device, rank=rank, world_size=world)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    gamma_n = ap_.gamma ** ap_.nstep
    filled = 0
    if getattr(memory, "dedup", False):     # frame-deduplicated replay: one random plane per step, as the actors' runs
        memory.fill_synthetic(memory.memory_size, nstep=ap_.nstep, seed=1000 + rank, num_actions=opt.action_space,
                              gamma_n=gamma_n)
        filled = memory.memory_size
        s = None
    while filled < memory.memory_size:      # synthetic transitions, written straight into the replay
        n = min(4096, memory.memory_size - filled)
        s = torch.randint(0, 256, (n + 1,) + tuple(opt.state_shape), dtype=torch.uint8, device=device, generator=gen)
        memory.feed_batch(s[:-1], torch.randint(0, opt.action_space, (n, 1), device=device, generator=gen),
                          torch.randn(n, 1, device=device, generator=gen).clamp_(-1, 1),
                          torch.full((n, 1), gamma_n, device=device), s[1:],
                          (torch.rand(n, 1, device=device, generator=gen) < 0.01).float(),
                          priorities=torch.rand(n, device=device, generator=gen) + 0.05)
        filled += n
    del s

    red_dev = device if backend == "nccl" else torch.device("cpu")

    def reduce_max(x):
        if world <= 1:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=getattr(comm, "group", None))
        return float(t.item())

    results, cores = {}, {}
    for precision in precisions:
        o = opt if precision == precisions[0] else options(precision)
        torch.manual_seed(o.seed)
        model = ModelsDict[o.model_type](o.model_params, o.norm_val, o.state_shape, o.action_space,
                                         o.action_shape).to(device)
        target = ModelsDict[o.model_type](o.model_params, o.norm_val, o.state_shape, o.action_space,
                                          o.action_shape).to(device)
        if world > 1:
            comm.broadcast_(model.flatten_(), src=0)
        core = make_core("dqn", o, model, target, memory, comm, device)
        if a.no_graph and hasattr(core, "program"):
            core.program.use_graph = False
        native = type(core).__name__ == "DqnNativeCore"
        if a.compute == "native" and not native:
            raise RuntimeError("native compute requested but the HIP engine is not in use")
        for _ in range(a.warmup):
            core.step()
        prog = getattr(core, "program", None)
        while prog is not None and getattr(prog, "ahead", 0) > 0:   # end the warm-up on a replay boundary (untimed)
            core.step()
        times = timed_windows(core, comm, a.steps, a.min_time, a.max_windows, sync, reduce_max)
        results[precision] = {"times": times, "loss": core.loss_values().get("critic_loss", float("nan")),
                              "engine": type(core).__name__}
        cores[precision] = (core, model, o)

    head = precisions[0]
    core, model, opt = cores[head]
    prog = getattr(core, "program", None)
    extra = {}
    if world > 1 or a.force_dp:
        if hasattr(comm, "verify_ranks"):
            extra.update(comm.verify_ranks())
        # after the timed region: every rank must hold the same parameters
        mine = core.flat_params().detach().clone()
        ref = mine.clone()
        comm.broadcast_(ref, src=0)
        extra["dp_param_max_abs_diff_across_ranks"] = reduce_max(float((mine - ref).abs().max().item()))
        extra["dp_graph"] = getattr(prog, "dp_graph_mode", None)
        extra["dp_allreduce"] = getattr(prog, "allreduce_kind", getattr(comm, "backend", None))
        # the peer kernel's spins are bounded: a wait that ran out sets an error word instead of hanging the GPU
        errs = 0
        for pr in [getattr(c[0], "program", None) for c in cores.values()]:
            for ar in (getattr(pr, "ar_tail", None), getattr(pr, "ar_head", None)):
                if ar is not None and hasattr(ar, "error"):
                    errs += int(bool(ar.error()))
        extra["dp_allreduce_errors"] = int(reduce_max(float(errs)))
        # self-diagnosis for the scaling runs: what the start-up measurement chose for the fc-tail bucket (and the
        # candidates' times), and where the step's communication goes (per-bucket all-reduce, exposed waits)
        extra["dp_wire_choice"] = getattr(prog, "comm_report", None)
        extra["dp_grad_allreduce_dtype"] = getattr(prog, "grad_dtype", None)
        if hasattr(prog, "measure_comm") and os.environ.get("PDA_BENCH_COMM_PROBE", "1") != "0":
            comm_us = prog.measure_comm(steps=20)
            extra["dp_comm_us"] = {k: (round(reduce_max(v), 2) if isinstance(v, float) else v)
                                   for k, v in comm_us.items()}
        if extra["dp_allreduce_errors"] and rank == 0:
            print("[bench] the peer all-reduce reported timed-out waits: the DP result is NOT valid; rerun with "
                  "--allreduce rccl", file=sys.stderr, flush=True)
    if a.actor_bench and rank == 0 and a.compute == "native":
        from pytorch_distributed_amd.ops.dqn_engine import DqnCnnInference
        for nact in (8, 256):
            inf = DqnCnnInference(model, max_batch=nact)
            obs = torch.randint(0, 256, (nact,) + tuple(opt.state_shape), dtype=torch.uint8, device=device)
            eps = torch.full((nact,), 0.1, device=device)
            for _ in range(20):
                inf.act(obs, eps)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(200):
                inf.act(obs, eps)[0].cpu()      # actions return to the host every env step
            dt = time.perf_counter() - t1
            extra["actor_inference_frames_per_sec_%d_envs" % nact] = round(200 * nact * 4 / dt, 1)  # 4 frames / step

    out = None
    if rank == 0:
        def ups_of(p):
            return world * a.steps / _median(results[p]["times"])
        times = results[head]["times"]
        med = _median(times)
        ups = world * a.steps / med
        base = None
        if os.path.exists(BASELINE_FILE):
            with open(BASELINE_FILE) as f:
                base = json.load(f).get("learner_updates_per_sec")
        native = results[head]["engine"] == "DqnNativeCore"
        dtype = {"fp32": "fp32-equivalent (split-bf16 x3 MFMA products, fp32 accumulate, fp32 master weights)",
                 "bf16": "bf16"}[head] if native else "fp32"
        out = {
            "metric": "learner updates/sec (Distributed DQN, Nature-CNN, batch %d per GPU; summed over GPUs)" % a.batch,
            "value": round(ups, 2), "unit": "updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * med / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(ups / base, 2) if base else None,
            "baseline_note": "reference publishes no throughput; baseline = reference-style learner loop "
                             "(host shm replay, fp32 eager PyTorch) measured on one MI355X, bench/reference_style.py",
            "dtype": dtype, "data": "synthetic",
            "timed_windows": len(times), "timed_seconds": round(sum(times), 4),
            "ms_per_step_min": round(1e3 * min(times) / a.steps, 4),
            "ms_per_step_max": round(1e3 * max(times) / a.steps, 4),
            "samples_per_sec": round(ups * a.batch, 1), "final_loss": round(float(results[head]["loss"]), 6),
            "config": {"model": "dqn-cnn (Nature CNN, 6 actions, 1,687,206 params)", "global_batch": a.batch * world,
                       "seq_len": None, "obs": "4x84x84 uint8", "nstep": ap_.nstep, "replay": a.replay,
                       "replay_kind": ("prioritized (HIP sum/min tree)" if memory.enable_per else "uniform") +
                                      (", frame-deduplicated" if getattr(memory, "dedup", False) else ""),
                       "optimizer": "adam lr=%g" % ap_.lr, "target_update": ap_.target_model_update,
                       "parallelism": "dp%d" % world, "engine": results[head]["engine"],
                       "precision": head if native else "fp32",
                       "collectives": (getattr(comm, "backend", None) if (world > 1 or a.force_dp) else None),
                       "dp_graph": getattr(prog, "dp_graph_mode", None),
                       "dp_grad_allreduce_dtype": getattr(prog, "grad_dtype", None),
                       "hip_graph": bool(native and not a.no_graph),
                       "steps_per_graph": getattr(prog, "graph_steps", None)},
        }
        if native:
            for p in precisions:
                out["%s_value" % p] = round(ups_of(p), 2)
                out["%s_final_loss" % p] = round(float(results[p]["loss"]), 6)
        out.update(extra)

    cores.clear()
    if hasattr(memory, "close"):
        memory.close()
    comm.close()
    node_on = a.node == "on" or (a.node == "auto" and use_cuda and world == 1)
    if node_on and rank == 0:
        # whole-job metric (BASELINE.json: learner updates/s + env frames/s of the node): the launcher-run job
        # of config 2 in child processes; this rank's GPU work is over and its replay is released
        del memory
        if use_cuda:
            torch.cuda.empty_cache()
        out["node"] = run_node_job(a, precisions[0], device)
    if rank == 0 and (a.ddpg == "on" or (a.ddpg == "auto" and use_cuda and world == 1 and a.compute == "native")):
        # the second agent family (reference core/single_processes/ddpg_learner.py): one learner update = actor +
        # critic forward/backward, both Adam updates, both Polyak target updates, on HBM-resident replay
        for key, prec in (("ddpg_value", "fp32"), ("ddpg_bf16_value", "bf16")):
            r = run_ddpg_step(prec)
            out[key] = r.get("updates_per_sec")
            out[key.replace("value", "us_per_step")] = r.get("us_per_step")
            if "error" in r:
                out[key + "_error"] = r["error"]
        out["ddpg_config"] = {"model": "ddpg-mlp (actor 3-300-200-1, critic 3-400 | 401-300-1)", "batch": 64,
                              "replay": "HBM-resident, uniform", "unit": "updates/s",
                              "dtype": {"ddpg_value": "fp32 (v_mfma_f32_16x16x4_f32: exact fp32 products and sums)",
                                        "ddpg_bf16_value": "bf16 operands, fp32 accumulate"}}
    if rank == 0 and (a.extras == "on" or (a.extras == "auto" and use_cuda and world == 1 and a.compute == "native"
                                           and a.config == 2)):
        # BASELINE.json configs 4 and 5 (learner side): each measured for <= ~3 s in a child of its own
        for key, extra in (("per_value", ["--config", "4", "--batch", "128", "--steps", "1000", "--warmup", "200",
                                          "--min-time", "1.5"]),
                           ("b4096_value", ["--config", "5", "--batch", "4096", "--steps", "50", "--warmup", "10",
                                            "--min-time", "1.5"])):
            r = run_child_bench(extra)
            out[key] = r.get("value")
            out[key.replace("_value", "_ms_per_step")] = r.get("ms_per_step")
            out[key.replace("_value", "_exit_code")] = r.get("exit_code")
            if "error" in r:
                out[key + "_error"] = r["error"]
        out["extras_config"] = {"per_value": "config 4 learner: prioritized replay (HIP sum/min tree, 50 k transitions "
                                             "HBM-resident), batch 128, fp32-class engine, updates/s",
                                "b4096_value": "config 5 learner: batch 4096, prioritized replay, fp32-class engine, "
                                               "updates/s (x 4096 = samples/s)"}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    in_group = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_group and (a.gpus > 1 or a.launcher):
        sys.exit(launch_ranks(a, argv))
    run_rank(a)


if __name__ == "__main__":
    main()
