"""Wall-clock to Pong score: the third leg of the BASELINE.json metric ("learner updates/sec + env frames/sec (node);
wall-clock to Pong score 20"), measured on the built-in MiniPong (envs/minipong.py) because no ALE exists on this pool.

The reference's only published result is the README Pong curve (/root/reference/README.md:5-10): Distributed DQN,
8 actors, n-step 5, evaluator average reward (2 greedy games) against LEARNER STEPS; it reaches about +20 after
35-45 k learner steps.  This runs the same job -- config 8: the reference's hyper-parameters untouched
(/root/reference/utils/options.py:112-141) -- until an evaluation averages >= --goal (default +18 of +-21), and reports
the wall-clock from job start and from the first learner update, the learner steps and env steps it took, and the
evaluation curve.

    python bench/pong_wallclock.py --core fp32 --seed 1        # native engine, fp32-class (split-bf16 x3)
    python bench/pong_wallclock.py --core bf16 --seed 1        # native engine, bf16 operands
    python bench/pong_wallclock.py --core torch --seed 1       # compute=torch: eager fp32 PyTorch-ROCm (the oracle core)

Prints the curve and ONE JSON line.
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
    ap.add_argument("--core", default="fp32", choices=["fp32", "bf16", "torch"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--goal", type=float, default=18.0)
    ap.add_argument("--max-steps", type=int, default=600000, help="learner steps after which the run gives up")
    ap.add_argument("--actors", type=int, default=8)
    ap.add_argument("--envs-per-actor", type=int, default=None)
    ap.add_argument("--eval-episodes", type=int, default=2)
    ap.add_argument("--eval-every", type=float, default=2.0, help="seconds between two evaluations (reference: 30)")
    ap.add_argument("--set", nargs="*", default=[])
    a = ap.parse_args()
    import torch.multiprocessing as mp
    mp.set_start_method("spawn", force=True)
    from pytorch_distributed_amd.launcher import run
    from pytorch_distributed_amd.utils.board import read_scalars
    from pytorch_distributed_amd.utils.options import Options, parse_overrides
    root = tempfile.mkdtemp(prefix="pda_pong_")
    ov = {"root_dir": root, "timestamp": "pong", "seed": a.seed, "num_actors": a.actors,
          "compute": "torch" if a.core == "torch" else "native", "precision": "bf16" if a.core == "bf16" else "fp32",
          "agent_params.steps": a.max_steps, "agent_params.stop_at_eval_reward": a.goal,
          "agent_params.logger_freq": 2.0, "agent_params.poll_interval": 0.25,
          "agent_params.evaluator_freq": a.eval_every, "agent_params.evaluator_nepisodes": a.eval_episodes}
    if a.envs_per_actor is not None:
        ov["num_envs_per_actor"] = a.envs_per_actor
    ov.update(parse_overrides(a.set))
    opt = Options(config=8, overrides=ov)
    t0 = time.time()
    sup = run(opt)
    wall = time.time() - t0
    rows = read_scalars(opt.log_dir)
    evals = [(r["time"], r["step"], r["value"]) for r in rows if r["tag"] == "evaluator/avg_reward"]
    nfr = [(r["time"], r["step"], r["value"]) for r in rows if r["tag"] == "actor/total_nframes"]
    ups = [(r["time"], r["step"], r["value"]) for r in rows if r["tag"] == "throughput/learner_updates_per_sec"]
    t_learn = next((t for t, s, _ in ups if s > 0), None)      # first logger window in which the learner had started
    for t, s, v in evals:
        print("t = %6.1f s   learner step %7d   eval avg reward %6.1f" % (t - t0, s, v))
    hit = next(((t, s, v) for t, s, v in evals if v >= a.goal), None)
    env_steps_at_hit = None
    if hit is not None and nfr:
        before = [v for t, _, v in nfr if t <= hit[0]]
        env_steps_at_hit = int(before[-1]) if before else None
    steady = [v for _, s, v in ups if s > 0 and v > 0]
    fps = [r["value"] for r in rows if r["tag"] == "throughput/actor_frames_per_sec" and r["step"] > 0 and r["value"] > 0]
    out = {"bench": "pong_wallclock", "env": "minipong", "core": a.core, "seed": a.seed, "goal": a.goal,
           "reached": hit is not None,
           "wall_s_from_job_start": round(hit[0] - t0, 1) if hit else None,
           "wall_s_from_first_update": round(hit[0] - t_learn, 1) if (hit and t_learn) else None,
           "learner_steps_at_goal": int(hit[1]) if hit else None, "env_steps_at_goal": env_steps_at_hit,
           "eval_at_goal": round(hit[2], 2) if hit else None, "evaluations": len(evals),
           "best_eval": max([v for _, _, v in evals] or [float("nan")]),
           "learner_updates_per_sec": round(sum(steady) / len(steady), 1) if steady else None,
           "env_steps_per_sec": round(sum(fps) / len(fps), 1) if fps else None,
           "max_replay_ratio": opt.agent_params.max_replay_ratio,
           "job_wall_s": round(wall, 1), "num_actors": opt.num_actors, "num_envs_per_actor": opt.num_envs_per_actor,
           "nstep": opt.agent_params.nstep, "batch_size": opt.agent_params.batch_size, "lr": opt.agent_params.lr,
           "target_model_update": opt.agent_params.target_model_update, "learn_start": opt.agent_params.learn_start,
           "memory_size": opt.memory_params.memory_size, "exitcodes": sup.exitcodes,
           "curve": [(round(t - t0, 1), int(s), round(v, 1)) for t, s, v in evals]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
