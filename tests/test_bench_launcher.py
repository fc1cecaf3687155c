"""bench.py starts its own ranks: `python bench.py --gpus N` with no torchrun environment must launch N
fresh rank processes from a parent that never touches the GPU, forward rank 0's one JSON line, prove N
distinct participants and fail loudly when a rank dies (the reference's `num_learners` loop,
/root/reference/main.py:82-94, here as real data-parallel ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_DP_GRAPH"):
        env.pop(k, None)
    return env


def _run(args, timeout=300):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_bench_launches_two_gloo_ranks_itself():
    out = _run(["--gpus", "2", "--compute", "torch", "--backend", "gloo", "--device", "cpu", "--steps", "4",
                "--warmup", "1", "--replay", "512", "--batch", "16", "--min-time", "0"])
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["warmup"] == 1
    assert res["rccl_ranks"] == 2 and res["ranks_distinct"] is True
    assert res["rank_id_min_max_sum"] == [0, 1, 1]
    assert len({d["pid"] for d in res["rank_devices"]}) == 2
    assert res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 32
    assert res["config"]["collectives"] == "gloo"
    assert res["dp_param_max_abs_diff_across_ranks"] == 0.0
    assert res["value"] > 0 and res["timed_windows"] == 1


def test_bench_repeats_short_windows_and_reports_spread():
    out = _run(["--gpus", "1", "--launcher", "--compute", "torch", "--device", "cpu", "--steps", "1", "--warmup", "0",
                "--replay", "256", "--batch", "8", "--min-time", "0.3", "--max-windows", "50"])
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    res = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 1 and res["steps"] == 1
    assert res["timed_windows"] >= 2 and res["timed_seconds"] >= 0.3 or res["timed_windows"] == 50
    assert res["ms_per_step_min"] <= res["ms_per_step"] <= res["ms_per_step_max"]


def test_bench_refuses_mismatched_world_size():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--compute", "torch",
                          "--device", "cpu", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0 and b"WORLD_SIZE" in out.stderr
    assert not [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]


def test_bench_launcher_fails_when_a_rank_dies():
    env = _env()
    env["PDA_BENCH_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--compute", "torch",
                          "--backend", "gloo", "--device", "cpu", "--steps", "2", "--warmup", "0", "--replay", "256",
                          "--batch", "8", "--min-time", "0", "--launch-timeout", "120"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
    assert out.returncode != 0
    assert b"rank 1 exited" in out.stderr
