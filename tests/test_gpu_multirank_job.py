"""The multi-learner JOB (launcher -> logger, actors, N learner ranks, evaluator) rehearsed on one card.

The reference's `num_learners` loop (/root/reference/main.py:82-94) starts Hogwild writers on one shared CUDA model.
Here the learners are synchronous data-parallel ranks, and three properties of the job around them are pinned:

* load balance: batched actor inference is hosted by EVERY learner rank (each serves its share of the mailbox slots
  from its own live weights), so rank 0 is not the straggler of the synchronous step;
* fail-safe: the peer all-reduce's bounded waits raise an error word; the production learner loop folds that word into
  its periodic agreement and either moves every rank to the process group's all-reduce (after re-syncing from rank 0)
  or ends the job with a non-zero exit code -- it never trains on with garbage sums;
* the ranks stay bit-identical (learner_rank_agreement.json).

One GPU per box on the build pool: the ranks share cuda:0 (allow_shared_gpu), the process group is gloo (RCCL refuses
two ranks on one device) and the gradient all-reduce is the hipIpc peer kernel, exactly as it would be across GPUs.
"""
import glob
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_job(root, *overrides, timeout=110, learners=2, steps=400):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_DP_GRAPH", "PDA_OVERRIDES"):
        env.pop(k, None)
    env["PDA_DP_ALLREDUCE"] = "peer"                # fail loudly if the peer regions cannot be mapped
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["config=2", "root_dir=%s" % root, "timestamp=mr", "num_actors=4", "num_envs_per_actor=2",
            "num_learners=%d" % learners, "allow_shared_gpu=True", "dist_backend=gloo", "precision=bf16",
            "agent_params.steps=%d" % steps, "agent_params.learn_start=300", "agent_params.batch_size=32",
            "memory_params.memory_size=4096", "agent_params.logger_freq=1.0", "agent_params.poll_interval=0.3",
            "agent_params.evaluator_freq=3.0", "agent_params.evaluator_nepisodes=1", "env_params.early_stop=60",
            "env_params.synthetic_episode_len=30", "shutdown_grace_s=8.0"] + list(overrides)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args, cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    log_dir = os.path.join(root, "logs", "mi355x_mr")
    text = out.stdout.decode(errors="replace")
    return out.returncode, text, log_dir


def _json(path):
    with open(path) as f:
        return json.load(f)


def test_every_learner_rank_hosts_its_share_of_batched_inference(tmp_root):
    rc, text, log_dir = _run_job(tmp_root)
    assert rc == 0, text[-3000:]
    assert "[learner 0] hosts batched inference for mailbox slots [0, 4)" in text, text[-3000:]
    assert "[learner 1] hosts batched inference for mailbox slots [4, 8)" in text, text[-3000:]
    st0 = _json(os.path.join(log_dir, "inference_stats.json"))
    st1 = _json(os.path.join(log_dir, "inference_stats_rank1.json"))
    assert st0["where"] == "learner" and st1["where"] == "learner"
    assert st0["requests"] > 100 and st1["requests"] > 100          # both ranks really served their actors
    assert st0["live_slots"] <= 4 and st1["live_slots"] <= 4
    agree = _json(os.path.join(log_dir, "learner_rank_agreement.json"))
    assert agree["param_max_abs_diff_summed_over_ranks"] == 0.0, agree
    comm = _json(os.path.join(log_dir, "learner_comm.json"))
    assert comm["allreduce"].startswith("peer") and comm["peer_allreduce_timeouts"] == 0
    assert not glob.glob(os.path.join(log_dir, "fault_*"))


def test_peer_allreduce_timeout_moves_every_rank_to_the_process_group(tmp_root):
    """Rank 1 stops storing its gradient slices at its 40th update (what a dead link looks like): rank 0's waits run
    out, its error word reaches every rank through the next agreement, all ranks re-sync from rank 0 and carry on
    over the process group's all-reduce; at the end the ranks are bit-identical again."""
    rc, text, log_dir = _run_job(tmp_root, "inject_fault=peer-withhold:1@40", "stop_agree_every=8",
                                 "dp_allreduce_on_error=rccl")
    assert rc == 0, text[-3000:]
    assert "peer all-reduce: a wait timed out" in text, text[-3000:]
    comm = _json(os.path.join(log_dir, "learner_comm.json"))
    assert comm["peer_allreduce_timeouts"] >= 1 and comm["allreduce"] == "rccl" and not comm["stopped_on_comm_error"]
    agree = _json(os.path.join(log_dir, "learner_rank_agreement.json"))
    assert agree["steps"] == 400 and agree["param_max_abs_diff_summed_over_ranks"] == 0.0, agree


def test_peer_allreduce_timeout_can_stop_the_job_with_a_nonzero_exit(tmp_root):
    rc, text, log_dir = _run_job(tmp_root, "inject_fault=peer-withhold:0@40", "stop_agree_every=8",
                                 "dp_allreduce_on_error=stop", steps=100000)
    assert rc != 0, text[-3000:]
    assert "peer all-reduce: a wait timed out" in text and "stopping the job" in text, text[-3000:]
    comm = _json(os.path.join(log_dir, "learner_comm.json"))
    assert comm["stopped_on_comm_error"] and comm["peer_allreduce_timeouts"] == 1
