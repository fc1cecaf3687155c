"""Learning sanity on the GPU ("learning sanity" tier): the whole multi-process job
with the native HIP learner has to actually learn.

    python bench/learning_sanity.py --config 7 --steps 150000      # DDPG (native MLP kernels) on Pendulum
    python bench/learning_sanity.py --config 6 --steps 20000       # DQN-MLP on the 5-state chain

Prints the evaluator's reward curve (greedy episodes, against learner steps) and ONE JSON line.
Pendulum counts as learned when the best evaluation reaches avg reward > -400 (random: about -1200;
a solved swing-up: -150 .. -300)."""
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
    ap.add_argument("--config", type=int, default=7)
    ap.add_argument("--steps", type=int, default=150000)
    ap.add_argument("--actors", type=int, default=8)
    ap.add_argument("--set", nargs="*", default=[])
    a = ap.parse_args()
    import torch.multiprocessing as mp
    mp.set_start_method("spawn", force=True)
    from pytorch_distributed_amd.launcher import run
    from pytorch_distributed_amd.utils.board import read_scalars
    from pytorch_distributed_amd.utils.options import Options, parse_overrides
    root = tempfile.mkdtemp(prefix="pda_learn_")
    ov = {"root_dir": root, "timestamp": "learn%d" % a.config, "num_actors": a.actors,
          "agent_params.steps": a.steps, "agent_params.logger_freq": 2.0, "agent_params.poll_interval": 0.5,
          "agent_params.evaluator_freq": 3.0, "agent_params.evaluator_nepisodes": 3}
    ov.update(parse_overrides(a.set))
    opt = Options(config=a.config, overrides=ov)
    t0 = time.time()
    sup = run(opt)
    wall = time.time() - t0
    rows = read_scalars(opt.log_dir)
    curve = [(r["step"], round(r["value"], 1)) for r in rows if r["tag"] == "evaluator/avg_reward"]
    for step, val in curve:
        print("eval @ learner step %7d: avg reward %8.1f" % (step, val))
    best = max([v for _, v in curve] or [float("nan")])
    # what the final checkpoint's policy looks like (initial weights give |pi| ~ 1e-3, a saturated actor 1.0)
    extra = {}
    try:
        import torch
        from pytorch_distributed_amd.launcher import prepare_options
        from pytorch_distributed_amd.utils.checkpoint import load_state_dict
        from pytorch_distributed_amd.utils.factory import ModelsDict
        prepare_options(opt)
        model = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space,
                                           opt.action_shape)
        init = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        model.load_state_dict(load_state_dict(opt.model_name))
        now = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        extra["param_l2_norm"] = round(float(now.norm()), 4)
        if opt.agent_type == "ddpg":
            obs = torch.randn((256,) + tuple(opt.state_shape)).clamp_(-1, 1)
            extra["mean_abs_action"] = round(float(model.forward_actor(obs).abs().mean()), 4)
    except Exception as exc:      # diagnostics only
        extra["checkpoint_probe_error"] = repr(exc)
    out = {"bench": "learning_sanity", "config": a.config, "agent": opt.agent_type, "env": opt.env_type,
           "model": opt.model_type, "memory": opt.memory_type, "compute": opt.compute, "device": opt.device,
           "learner_steps": a.steps, "wall_s": round(wall, 1), "first_eval": curve[0][1] if curve else None,
           "best_eval": best, "last_eval": curve[-1][1] if curve else None, "evaluations": len(curve),
           "exitcodes": sup.exitcodes}
    out.update(extra)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
