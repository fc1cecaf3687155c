"""End-to-end node throughput: the whole multi-process job (logger, N CPU actors,
learner(s), evaluator, batched inference server) run through the launcher for a
bounded number of learner steps, then learner updates/s and env frames/s read
back from the logger's own scalars (the BASELINE.json metric; the reference
never computes it).

    python bench/node_throughput.py --config 2 --steps 60000 --actors 8
    python bench/node_throughput.py --config 7 --steps 20000          # DDPG, HBM replay

Prints ONE JSON line.  One env step = 4 emulator frames (action repeat 4,
reference core/env.py:31), so frames/s = 4 x actor steps/s for the Atari-shaped
envs and 1 x for Pendulum.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60000)
    ap.add_argument("--actors", type=int, default=None)
    ap.add_argument("--learners", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--replay", type=int, default=None)
    ap.add_argument("--learn-start", type=int, default=None)
    ap.add_argument("--inference", default=None, choices=[None, "local", "batched"])
    ap.add_argument("--device", default=None)
    ap.add_argument("--envs-per-actor", type=int, default=None,
                    help="envs stepped by each actor process (2 = one inference request in flight while the other "
                         "env is stepped)")
    ap.add_argument("--window-us", type=int, default=None, help="inference batching window (option inference_window_us)")
    ap.add_argument("--precision", default=None, choices=[None, "fp32", "bf16"])
    ap.add_argument("--set", nargs="*", default=[], help="extra KEY=VALUE option overrides")
    a = ap.parse_args()

    import torch.multiprocessing as mp
    mp.set_start_method("spawn", force=True)
    from pytorch_distributed_amd.launcher import run
    from pytorch_distributed_amd.utils.board import read_scalars
    from pytorch_distributed_amd.utils.options import Options, parse_overrides

    root = tempfile.mkdtemp(prefix="pda_node_")
    ov = {"root_dir": root, "timestamp": "node%d" % a.config, "agent_params.steps": a.steps,
          "agent_params.logger_freq": 2.0, "agent_params.poll_interval": 0.5,
          "agent_params.evaluator_freq": 5.0, "agent_params.evaluator_nepisodes": 1,
          "env_params.early_stop": 500, "num_learners": a.learners}
    if a.actors is not None:
        ov["num_actors"] = a.actors
    if a.batch is not None:
        ov["agent_params.batch_size"] = a.batch
    if a.replay is not None:
        ov["memory_params.memory_size"] = a.replay
    if a.learn_start is not None:
        ov["agent_params.learn_start"] = a.learn_start
    if a.inference is not None:
        ov["actor_inference"] = a.inference
    if a.device is not None:
        ov["device"] = a.device
    if a.envs_per_actor is not None:
        ov["num_envs_per_actor"] = a.envs_per_actor
    if a.window_us is not None:
        ov["inference_window_us"] = a.window_us
    if a.precision is not None:
        ov["precision"] = a.precision
    ov.update(parse_overrides(a.set))
    opt = Options(config=a.config, overrides=ov)

    t0 = time.time()
    sup = run(opt)
    wall = time.time() - t0
    rows = read_scalars(opt.log_dir)

    def series(tag):
        return [(r["time"], r["step"], r["value"]) for r in rows if r["tag"] == tag]

    ups = series("throughput/learner_updates_per_sec")
    fps = series("throughput/actor_frames_per_sec")
    # windows in which the learner was running the whole time (skip replay warm-up and the last partial one)
    busy = [v for (_, s, v), (_, s_prev, _) in zip(ups[1:], ups[:-1]) if s_prev > 0 and v > 0]
    if len(busy) > 1:
        busy = busy[:-1]
    frame_mult = 4 if opt.env_type in ("atari", "synthetic") else 1
    steps_per_s = [v for _, _, v in fps if v > 0]
    # the same steady windows for the actors (the learner's windows are the logger's windows)
    steady_fps = [frame_mult * v for (_, s, v), (_, s_prev, _) in zip(fps[1:], fps[:-1]) if s_prev > 0 and v > 0]
    if len(steady_fps) > 1:
        steady_fps = steady_fps[:-1]

    def spread(xs):
        if not xs:
            return None
        m = sum(xs) / len(xs)
        sd = (sum((x - m) ** 2 for x in xs) / max(len(xs) - 1, 1)) ** 0.5
        return {"windows": len(xs), "mean": round(m, 1), "std": round(sd, 1), "min": round(min(xs), 1),
                "max": round(max(xs), 1)}
    loop_profile = None
    try:
        with open(os.path.join(opt.log_dir, "learner_loop_profile.json")) as f:
            loop_profile = json.load(f)
    except (OSError, ValueError):
        pass
    infer = None
    try:
        with open(os.path.join(opt.log_dir, "inference_stats.json")) as f:
            infer = json.load(f)
    except (OSError, ValueError):
        pass
    replay_stats = None
    try:
        with open(os.path.join(opt.log_dir, "replay_stats.json")) as f:
            replay_stats = json.load(f)
    except (OSError, ValueError):
        pass
    out = {
        "replay_stats": replay_stats,
        "bench": "node_throughput", "config": a.config, "agent": opt.agent_type, "env": opt.env_type,
        "model": opt.model_type, "memory": opt.memory_type, "num_actors": opt.num_actors,
        "num_envs_per_actor": opt.num_envs_per_actor,
        "num_learners": opt.num_learners, "actor_inference": opt.actor_inference, "device": opt.device,
        "compute": opt.compute, "precision": opt.precision, "inference": infer, "batch_size": opt.agent_params.batch_size, "nstep": opt.agent_params.nstep,
        "learner_steps": a.steps, "wall_s": round(wall, 1),
        "learner_updates_per_sec_steady": round(sum(busy) / len(busy), 1) if busy else None,
        "learner_updates_per_sec_windows": spread(busy), "env_frames_per_sec_windows": spread(steady_fps),
        "logger_window_s": opt.agent_params.logger_freq, "learner_loop_profile": loop_profile,
        "learner_updates_per_sec_peak": round(max([v for _, _, v in ups] or [0]), 1),
        "actor_steps_per_sec_mean": round(sum(steps_per_s) / len(steps_per_s), 1) if steps_per_s else None,
        "env_frames_per_sec_mean": round(frame_mult * sum(steps_per_s) / len(steps_per_s), 1) if steps_per_s else None,
        "env_frames_per_sec_peak": round(frame_mult * max(steps_per_s), 1) if steps_per_s else None,
        "exitcodes": sup.exitcodes, "actor_restarts": sup.restarts,
        "evaluator_reports": len(series("evaluator/avg_reward")),
        "checkpoint_written": os.path.exists(opt.model_name),
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
