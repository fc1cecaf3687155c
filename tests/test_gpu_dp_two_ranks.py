"""Two data-parallel learner ranks of the native (HIP) step program on ONE card.

RCCL refuses two ranks on one device, so the rehearsal runs over gloo (host-staged all-reduces,
parallel/comm.py) with both ranks on cuda:0: what it checks is the multi-rank step program itself --
bucket order, split optimizer, per-rank replay shards and RNG streams -- by requiring the parameters
of the two ranks to be identical after the run.  (Reference: num_learners > 1 is Hogwild on shared
CUDA tensors, main.py:83-94 / dqn_learner.py:27-39; here it is synchronous data parallelism.)
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_dp_bench_keeps_ranks_identical():
    env = dict(os.environ)
    env.pop("PDA_DP_GRAPH", None)
    env["PDA_BENCH_FAULTHANDLER"] = "100"       # a stalled rank dumps its stack and exits
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # no torchrun: bench.py starts its two ranks itself (fresh children of a parent that never touches the GPU)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
           "--backend", "gloo", "--share-gpu", "--replay", "4096", "--min-time", "0", "--precision", "bf16",
           "--allreduce", "rccl", "--launch-timeout", "100"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=115)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints the one JSON line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 256
    assert res["config"]["engine"] == "DqnNativeCore" and res["config"]["collectives"] == "gloo"
    assert res["dp_param_max_abs_diff_across_ranks"] == 0.0
    assert res["rccl_ranks"] == 2 and res["ranks_distinct"] is True and res["rank_id_min_max_sum"] == [0, 1, 1]
    assert res["final_loss"] == res["final_loss"] and res["value"] > 0      # finite


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_two_rank_dp_with_the_peer_allreduce_in_one_graph(precision):
    """Same rehearsal with the xGMI peer all-reduce kernel (hipIpc-mapped regions, ops/csrc/peer_allreduce.hip):
    nothing host-driven is left in the step, so ten whole data-parallel updates -- collectives included -- replay as
    ONE hipGraph even though the process group is gloo."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_DP_GRAPH", "PDA_DP_ALLREDUCE"):
        env.pop(k, None)
    env["PDA_BENCH_FAULTHANDLER"] = "100"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
           "--backend", "gloo", "--share-gpu", "--replay", "4096", "--min-time", "0", "--precision", precision,
           "--allreduce", "peer", "--launch-timeout", "100"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=115)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    res = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["config"]["engine"] == "DqnNativeCore" and res["config"]["precision"] == precision
    # the wire format of the fc-tail bucket is chosen by a start-up measurement on the real bucket (knob
    # dp_grad_dtype=measure): the fp32-class engine only considers the exact fp32 wire
    choice = res["dp_wire_choice"]
    assert res["dp_allreduce"] == choice["chosen"] and res["dp_graph"] == "whole"
    exact = {"peer-fp32", "peer-fp32p"}         # plain and packed fp32 wires: both carry exact fp32 sums
    if precision == "fp32":
        assert set(choice["us"]) == exact and res["dp_grad_allreduce_dtype"] == "fp32"
    else:
        assert set(choice["us"]) == exact | {"peer-bf16"}
    assert all(v > 0 for v in choice["us"].values())
    assert res["dp_allreduce"] == min(sorted(choice["us"]), key=choice["us"].get)
    # ... and the step reports where its communication goes
    cu = res["dp_comm_us"]
    assert cu["steps"] == 20 and cu["allreduce_tail_us"] > 0 and cu["allreduce_head_us"] > 0
    assert cu["exposed_tail_us"] >= 0 and cu["exposed_head_us"] >= 0 and cu["step_us"] > cu["allreduce_tail_us"]
    assert res["config"]["steps_per_graph"] == 10
    assert res["dp_param_max_abs_diff_across_ranks"] == 0.0
    assert res["ranks_distinct"] is True
    assert res["final_loss"] == res["final_loss"] and res["value"] > 0
    print(json.dumps({k: res[k] for k in ("value", "ms_per_step", "dp_allreduce", "dp_graph", "dp_wire_choice",
                                          "dp_comm_us")}))


@pytest.mark.gpu
def test_eight_rank_dp_step_on_one_card():
    """The node size the data-parallel step is built for: eight ranks (here eight processes on one card), the peer
    all-reduce inside the captured step, every rank bit-identical afterwards."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_DP_GRAPH", "PDA_DP_ALLREDUCE"):
        env.pop(k, None)
    env["PDA_BENCH_FAULTHANDLER"] = "110"
    env["PDA_PEER_WORKGROUPS"] = "64"       # eight ranks' all-reduce launches resident on the one card together
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
           "--backend", "gloo", "--share-gpu", "--replay", "2048", "--min-time", "0", "--precision", "bf16",
           "--allreduce", "peer", "--launch-timeout", "110"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=118)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    res = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 8 and res["config"]["global_batch"] == 8 * 128
    assert res["dp_allreduce"] in ("peer-fp32", "peer-fp32p", "peer-bf16") and res["dp_graph"] == "whole"
    assert res["dp_wire_choice"]["chosen"] == res["dp_allreduce"]
    assert res["dp_param_max_abs_diff_across_ranks"] == 0.0 and res["dp_allreduce_errors"] == 0
    assert res["rccl_ranks"] == 8 and res["ranks_distinct"] is True and res["rank_id_min_max_sum"] == [0, 7, 28]
    assert res["final_loss"] == res["final_loss"] and res["value"] > 0


@pytest.mark.gpu
def test_two_rank_dp_on_frame_deduplicated_prioritized_replay():
    """Config 9's combination in the same rehearsal: every rank samples frame windows of its own plane ring
    (memories/dedup_memory.py) through its own sum tree; the data-parallel step is unchanged."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_DP_GRAPH", "PDA_DP_ALLREDUCE"):
        env.pop(k, None)
    env["PDA_BENCH_FAULTHANDLER"] = "100"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
           "--backend", "gloo", "--share-gpu", "--replay", "4096", "--min-time", "0", "--precision", "fp32",
           "--allreduce", "peer", "--launch-timeout", "100", "--config", "4", "--memory-type", "device-dedup-per",
           "--node", "off", "--ddpg", "off", "--extras", "off"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=115)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    res = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and "frame-deduplicated" in res["config"]["replay_kind"]
    assert res["dp_param_max_abs_diff_across_ranks"] == 0.0 and res["dp_allreduce_errors"] == 0
    assert res["final_loss"] == res["final_loss"] and res["value"] > 0
