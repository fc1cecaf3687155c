"""End-to-end plumbing with real spawned processes and no GPU (BASELINE.json
config 1: Distributed DQN, 2 actors, nstep=1, CPU learner) plus DDPG, mode=2
testing, batched inference, device-replay ingestion and actor restart."""
import os

import pytest
import torch

from pytorch_distributed_amd.launcher import run
from pytorch_distributed_amd.utils.board import read_scalars
from pytorch_distributed_amd.utils.checkpoint import load_state_dict
from pytorch_distributed_amd.utils.options import Options

FAST = {"device": "cpu", "agent_params.poll_interval": 0.2, "agent_params.logger_freq": 0.3,
        "agent_params.evaluator_freq": 0.5, "agent_params.evaluator_nepisodes": 1,
        "agent_params.actor_freq": 20, "agent_params.learner_freq": 5, "agent_params.actor_sync_freq": 20,
        "param_sync_freq": 5, "env_params.early_stop": 40, "env_params.synthetic_episode_len": 20}


def opts(config, root, **ov):
    o = dict(FAST)
    o.update({"root_dir": root, "timestamp": "t%d" % config})
    o.update(ov)
    return Options(config=config, overrides=o)


def check_common(opt, sup, min_steps):
    assert all(c == 0 for c in sup.exitcodes), sup.exitcodes
    assert os.path.exists(opt.model_name)
    tags = {s["tag"] for s in read_scalars(opt.log_dir)}
    assert {"learner/critic_loss", "throughput/learner_updates_per_sec",
            "throughput/actor_frames_per_sec", "evaluator/avg_reward"} <= tags
    steps = [s["step"] for s in read_scalars(opt.log_dir)]
    assert max(steps) >= min_steps


def test_config1_dqn_cpu_plumbing_then_mode2(tmp_root):
    opt = opts(1, tmp_root, **{"agent_params.steps": 30, "agent_params.learn_start": 40,
                               "agent_params.batch_size": 16, "memory_params.memory_size": 500})
    assert opt.num_actors == 2 and opt.agent_params.nstep == 1
    sup = run(opt)
    check_common(opt, sup, 30)
    sd = load_state_dict(opt.model_name)
    assert sd["critic.1.0.weight"].shape == (512, 3136)
    tags = {s["tag"] for s in read_scalars(opt.log_dir)}
    assert "actor/avg_reward" in tags and "actor/total_nframes" in tags
    # mode=2: only flip the mode, the tester finds the checkpoint by signature
    opt2 = opts(1, tmp_root, mode=2, **{"agent_params.tester_nepisodes": 2})
    assert opt2.model_file == opt.model_name
    sup2 = run(opt2)
    assert [p.exitcode for *_, p in sup2.procs] == [0]


def test_ddpg_pendulum_plumbing(tmp_root):
    opt = opts(3, tmp_root, num_actors=2, **{"agent_params.steps": 40, "agent_params.learn_start": 30,
                                             "agent_params.batch_size": 16})
    sup = run(opt)
    check_common(opt, sup, 40)
    assert "learner/actor_loss" in {s["tag"] for s in read_scalars(opt.log_dir)}
    assert "actor.4.weight" in load_state_dict(opt.model_name)


def test_device_replay_per_batched_inference_and_finetune(tmp_root):
    # device-per memory (host tensors here) fed through the shm ring, actions from the
    # batched inference server, PER on, double DQN
    opt = opts(4, tmp_root, num_actors=3, **{"agent_params.steps": 25, "agent_params.learn_start": 30,
                                             "agent_params.batch_size": 8, "agent_params.enable_double": True,
                                             "memory_params.memory_size": 256, "memory_params.ingest_ring_slots": 64,
                                             "memory_params.ingest_chunk": 16})
    assert opt.actor_inference == "batched" and opt.memory_params.enable_per
    sup = run(opt)
    check_common(opt, sup, 25)
    # finetune: a second run starts from the first run's checkpoint (main.py:45 in the reference)
    opt_ft = opts(1, tmp_root, timestamp="ft", model_file=opt.model_name,
                  **{"agent_params.steps": 3, "agent_params.learn_start": 10, "agent_params.batch_size": 4,
                     "agent_params.lr": 0.0, "memory_params.memory_size": 100})
    run(opt_ft)
    a, b = load_state_dict(opt.model_name), load_state_dict(opt_ft.model_name)
    assert all(torch.equal(a[k], b[k]) for k in a)      # lr=0: weights carried over unchanged


def test_frame_deduplicated_replay_whole_job(tmp_root):
    """memory_type device-dedup-per (host tensors here): the actors' n-step assemblers send runs of planes, the learner
    samples frame windows of its plane ring (memories/dedup_memory.py); 2 envs per actor = 4 interleaved streams."""
    import json
    opt = opts(4, tmp_root, num_actors=2, num_envs_per_actor=2, timestamp="dedup", memory_type="device-dedup-per",
               **{"agent_params.steps": 25, "agent_params.learn_start": 40, "agent_params.batch_size": 8,
                  "memory_params.memory_size": 256, "memory_params.dedup_plane_cap": 32,
                  "memory_params.dedup_max_age_ms": 2000, "memory_params.dedup_chunk_runs": 2})
    from pytorch_distributed_amd.utils.factory import MemoriesDict
    assert MemoriesDict[opt.memory_type].__name__ == "DedupDeviceMemory"
    sup = run(opt)
    check_common(opt, sup, 25)
    with open(os.path.join(opt.log_dir, "replay_stats.json")) as f:
        st = json.load(f)
    assert st["transitions"] >= 40 and st["planes_per_transition"] < 3.0 and st["rows_in_window"] > 0


def test_dead_actor_on_frame_deduplicated_replay_is_restarted_and_the_job_finishes(tmp_root):
    """An actor that dies takes its open run (and possibly a claimed ring slot) with it: the supervisor restarts it, its
    new n-step assemblers start new runs, and the learner keeps ingesting the other streams' runs meanwhile."""
    opt = opts(2, tmp_root, num_actors=2, num_envs_per_actor=2, timestamp="dedupfault", memory_type="device-dedup",
               inject_fault="actor:1@30",
               **{"agent_params.steps": 40, "agent_params.learn_start": 60, "agent_params.batch_size": 8,
                  "memory_params.memory_size": 256, "memory_params.dedup_plane_cap": 32,
                  "memory_params.dedup_max_age_ms": 200, "memory_params.dedup_chunk_runs": 2})
    sup = run(opt)
    assert sup.restarts == 1
    assert os.path.exists(opt.model_name)
    steps = [s["step"] for s in read_scalars(opt.log_dir)]
    assert max(steps) >= 40


def test_supervisor_restarts_dead_actor(tmp_root):
    opt = opts(6, tmp_root, inject_fault="actor:1@15", **{"agent_params.steps": 150})
    sup = run(opt)
    assert sup.restarts == 1
    assert os.path.exists(opt.model_name)


def test_resume_sidecar_continues_training(tmp_root):
    """True resume (absent in the reference): optimizer moments, target net and counters come back."""
    from pytorch_distributed_amd.utils.checkpoint import load_resume
    base = {"agent_params.learn_start": 60}
    opt = opts(6, tmp_root, timestamp="res", **dict(base, **{"agent_params.steps": 30}))
    run(opt)
    side = opt.model_name[:-4] + ".resume.pth"
    a = load_resume(side)
    assert a["learner_step"] == 30 and a["opt_step"] == 30 and a["exp_avg"].abs().sum() > 0
    opt2 = opts(6, tmp_root, timestamp="res", resume=True, **dict(base, **{"agent_params.steps": 50}))
    sup = run(opt2)
    assert all(c == 0 for c in sup.exitcodes), sup.exitcodes
    b = load_resume(side)
    assert b["learner_step"] == 50 and b["opt_step"] == 50          # 20 more optimizer steps, not 50
    assert not torch.equal(a["params"], b["params"])


def test_multi_env_actors_with_batched_inference(tmp_root):
    """num_envs_per_actor > 1 (the reference asserts 1): each actor process drives 3 envs with one
    inference request in flight per env; 2 x 3 mailbox slots, a 6-rung epsilon ladder."""
    opt = opts(4, tmp_root, num_actors=2, num_envs_per_actor=3, timestamp="venv",
               **{"agent_params.steps": 25, "agent_params.learn_start": 40, "agent_params.batch_size": 8,
                  "memory_params.memory_size": 256, "memory_params.ingest_ring_slots": 64,
                  "memory_params.ingest_chunk": 16})
    assert opt.env_params.num_envs_per_actor == 3
    sup = run(opt)
    check_common(opt, sup, 25)


# ---- failure handling (the reference's main.py:117-118 only join()s) -----------------------------
def test_learner_rank_death_tears_the_job_down_in_bounded_time(tmp_root):
    """Two data-parallel learner ranks over gloo; rank 1 dies after its 5th update.  Rank 0 is then inside (or on
    its way into) an all-reduce that can never complete and never looks at the stop flag again: the supervisor must
    notice the death, raise the flag, give the rest a bounded grace period, terminate what is left and report
    non-zero exit codes -- instead of hanging until the collective's own timeout (600 s)."""
    import time
    opt = opts(6, tmp_root, num_learners=2, timestamp="dead", inject_fault="learner:1@5", shutdown_grace_s=4.0,
               **{"agent_params.steps": 100000, "agent_params.learn_start": 40})
    t0 = time.time()
    sup = run(opt)
    wall = time.time() - t0
    assert wall < 90, wall
    assert sup.failed is not None and sup.failed[0] == "learner" and sup.failed[2] == 17
    assert not sup.ok
    codes = {(role, index): p.exitcode for role, index, _, p in sup.procs}
    learners = [c for (role, _), c in codes.items() if role == "learner"]
    assert 17 in learners and all(c is not None for c in codes.values())      # nobody left running
    assert any(c != 0 for c in learners)


def test_watchdog_stops_the_job_when_the_learner_hangs(tmp_root):
    """Heartbeats are consumed: a learner that is alive but silent (injected hang) is declared hung after
    watchdog_learner_s, the job stops and the hung process is terminated."""
    import time
    opt = opts(6, tmp_root, timestamp="hang", inject_fault="hang-learner:0@8", watchdog_learner_s=3.0,
               shutdown_grace_s=3.0, **{"agent_params.steps": 100000, "agent_params.learn_start": 40})
    t0 = time.time()
    sup = run(opt)
    assert time.time() - t0 < 90
    assert sup.hung and sup.hung[0][0] == "learner" and not sup.ok
    assert sup.failed == ("learner", sup.hung[0][1], "hung")
    assert all(p.exitcode is not None for *_, p in sup.procs)


def test_watchdog_restarts_a_hung_actor(tmp_root):
    opt = opts(6, tmp_root, timestamp="hanga", inject_fault="hang-actor:1@15", watchdog_actor_s=3.0,
               **{"agent_params.steps": 2500, "agent_params.learn_start": 40})
    sup = run(opt)
    assert sup.restarts >= 1 and sup.ok
    assert all(c == 0 for (role, *_), c in zip(sup.procs, sup.exitcodes) if role != "actor"), sup.exitcodes


def test_sharded_step_counter_keeps_the_reference_idiom():
    from pytorch_distributed_amd.agents.logs import GlobalLogs
    g = GlobalLogs()
    g.actor_step.add(3)
    g.actor_step.add(700, 5)
    with g.actor_step.get_lock():       # dqn_actor.py:166-167 still works
        g.actor_step.value += 1
    assert g.actor_step.value == 7
    g.actor_step.value = 100            # resume
    assert g.actor_step.value == 100
    g.beat("actor", 12)
    assert g.last_beat(12) > 0 and g.last_beat(13) == 0
