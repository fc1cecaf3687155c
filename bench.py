"""Headline benchmark: Distributed-DQN learner throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Config (BASELINE.json config 2, the reference's README Pong setting): Nature-CNN
DQN, 6 actions, batch 128 per learner GPU, n-step 5, Adam lr 1e-4, hard target
update every 250 steps, uniform replay of 50 000 transitions resident in HBM as
uint8, synthetic Atari-shaped observations (4x84x84 uint8), random-init weights.
One learner process per GPU; N > 1 = synchronous data parallel (weak scaling:
128 transitions per GPU per step, RCCL all-reduce of the flat gradient).

Every timed step is a complete learner update: replay sampling, target + online
forward, TD target/loss, full backward, (all-reduce,) Adam, target update.
Rank 0 prints ONE JSON line.  `value` = learner updates/s summed over GPUs (one
update = one batch of 128 transitions through one optimizer step on one GPU).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# reference-style learner loop (host shm replay -> fp32 cast -> H2D -> eager fp32 PyTorch),
# measured on one MI355X with bench/reference_style.py; see BASELINE.md
BASELINE_FILE = os.path.join(ROOT, "bench", "reference_measured.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=128, help="transitions per GPU per step")
    ap.add_argument("--replay", type=int, default=50000)
    ap.add_argument("--config", type=int, default=2, help="options.CONFIGS row (2 uniform, 4 Ape-X PER)")
    ap.add_argument("--compute", default="native", choices=["native", "torch"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default=None, help="process-group backend override (default: nccl = RCCL)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel step program even with one rank (rehearsal of the collective path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks on cuda:0 (gloo rehearsal of the multi-rank program on a one-GPU box)")
    ap.add_argument("--graph-steps", type=int, default=None,
                    help="learner updates per hipGraph replay (default: the option's, 5; 1 = one graph per step)")
    ap.add_argument("--actor-bench", action="store_true", help="also time batched actor inference (untimed extra)")
    a = ap.parse_args()

    if os.environ.get("PDA_BENCH_FAULTHANDLER"):     # debugging aid: dump every thread's stack if we stall
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PDA_BENCH_FAULTHANDLER"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print("[bench] WORLD_SIZE=%d but --gpus %d: using WORLD_SIZE" % (world, a.gpus), file=sys.stderr)
    if (world > 1 or a.force_dp) and (a.backend in (None, "nccl")) and a.compute == "native":
        # ask a throw-away child whether this node captures RCCL collectives into a hipGraph
        # (must happen before this process initialises the GPU); sets the DP step's graph mode
        from pytorch_distributed_amd.parallel.preflight import choose_dp_graph_mode
        os.environ["PDA_DP_GRAPH"] = choose_dp_graph_mode()
    device = torch.device("cuda:%d" % (0 if a.share_gpu else local))
    torch.cuda.set_device(device)

    from pytorch_distributed_amd.utils.options import Options
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.utils.factory import MemoriesDict, ModelsDict
    from pytorch_distributed_amd.parallel.comm import comm_from_env
    from pytorch_distributed_amd.agents.learners import make_core

    comm = comm_from_env(device, backend=a.backend, force=a.force_dp)
    opt = prepare_options(Options(config=a.config, overrides={
        "device": "cuda", "compute": a.compute, "num_learners": world, "force_dp": a.force_dp,
        "memory_params.memory_size": a.replay * world,      # each rank keeps a `replay`-sized shard
        "agent_params.batch_size": a.batch}))
    if a.graph_steps is not None:
        opt.graph_steps = int(a.graph_steps)
    else:
        # the learner replays several updates per hipGraph and runs ahead of its caller (options.graph_steps): pick a
        # count that divides --steps, so that EXACTLY that many updates execute inside the timed window
        opt.graph_steps = max(d for d in range(1, int(opt.graph_steps) + 1) if a.steps % d == 0)
    ap_ = opt.agent_params
    torch.manual_seed(opt.seed)

    def build():
        return ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape,
                                          opt.action_space, opt.action_shape).to(device)
    model, target = build(), build()
    if world > 1:
        comm.broadcast_(model.flatten_(), src=0)

    memory = MemoriesDict[opt.memory_type](opt.memory_params)
    memory.attach(device, rank=rank, world_size=world)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    gamma_n = ap_.gamma ** ap_.nstep
    filled = 0
    while filled < memory.memory_size:      # synthetic transitions, written straight into HBM
        n = min(4096, memory.memory_size - filled)
        s = torch.randint(0, 256, (n + 1,) + tuple(opt.state_shape), dtype=torch.uint8, device=device, generator=gen)
        memory.feed_batch(s[:-1], torch.randint(0, opt.action_space, (n, 1), device=device, generator=gen),
                          torch.randn(n, 1, device=device, generator=gen).clamp_(-1, 1),
                          torch.full((n, 1), gamma_n, device=device), s[1:],
                          (torch.rand(n, 1, device=device, generator=gen) < 0.01).float(),
                          priorities=torch.rand(n, device=device, generator=gen) + 0.05)
        filled += n
    del s

    core = make_core("dqn", opt, model, target, memory, comm, device)
    if a.no_graph and hasattr(core, "program"):
        core.program.use_graph = False
    native = type(core).__name__ == "DqnNativeCore"
    if a.compute == "native" and not native:
        raise RuntimeError("native compute requested but the HIP engine is not in use")

    for _ in range(a.warmup):
        core.step()
    prog = getattr(core, "program", None)
    while prog is not None and getattr(prog, "ahead", 0) > 0:   # end the warm-up on a replay boundary (untimed extras)
        core.step()
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        core.step()
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        # (host tensors under gloo: its device path stalls when the rehearsal's ranks share one GPU)
        red_dev = device if getattr(comm, "backend", None) == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = core.loss_values().get("critic_loss", float("nan"))

    extra = {}
    if world > 1:       # after the timed region: every rank must hold the same parameters
        mine = core.flat_params().detach().clone()
        ref = mine.clone()
        comm.broadcast_(ref, src=0)
        diff = (mine - ref).abs().max().reshape(1).to(red_dev)
        torch.distributed.all_reduce(diff, op=torch.distributed.ReduceOp.MAX)
        extra["dp_param_max_abs_diff_across_ranks"] = float(diff.item())
    if a.actor_bench and rank == 0 and native:
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

    if rank == 0:
        ups = world * a.steps / elapsed
        base = None
        if os.path.exists(BASELINE_FILE):
            with open(BASELINE_FILE) as f:
                base = json.load(f).get("learner_updates_per_sec")
        out = {
            "metric": "learner updates/sec (Distributed DQN, Nature-CNN, batch 128 per GPU; summed over GPUs)",
            "value": round(ups, 2), "unit": "updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(ups / base, 2) if base else None,
            "baseline_note": "reference publishes no throughput; baseline = reference-style learner loop "
                             "(host shm replay, fp32 eager PyTorch) measured on one MI355X, bench/reference_style.py",
            "dtype": "bf16" if native else "fp32", "data": "synthetic",
            "samples_per_sec": round(ups * a.batch, 1), "final_loss": round(float(loss), 6),
            "config": {"model": "dqn-cnn (Nature CNN, 6 actions, 1,687,206 params)", "global_batch": a.batch * world,
                       "seq_len": None, "obs": "4x84x84 uint8", "nstep": ap_.nstep, "replay": a.replay,
                       "replay_kind": "prioritized (HIP sum/min tree)" if memory.enable_per else "uniform",
                       "optimizer": "adam lr=%g" % ap_.lr, "target_update": ap_.target_model_update,
                       "parallelism": "dp%d" % world, "engine": type(core).__name__,
                       "collectives": (getattr(comm, "backend", None) if getattr(getattr(core, "program", None), "dp", False)
                                       else None),
                       "dp_graph": getattr(getattr(core, "program", None), "dp_graph_mode", None),
                       "dp_grad_allreduce_dtype": getattr(getattr(core, "program", None), "grad_dtype", None),
                       "hip_graph": bool(native and not a.no_graph),
                       "steps_per_graph": getattr(getattr(core, "program", None), "graph_steps", None)},
        }
        out.update(extra)
        print(json.dumps(out), flush=True)
    if hasattr(memory, "close"):
        memory.close()
    comm.close()


if __name__ == "__main__":
    main()
