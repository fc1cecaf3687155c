"""Learning evidence on pixels: the whole job (8 actors x 4 envs, batched inference, HBM replay, learner, evaluator) on
the built-in MiniPong (envs/minipong.py, config 8) must carry the evaluator's average from -21 (random play) into the
winning band, within a wall-clock bound, for every step back-end.

The reference's one published result is a Pong learning curve (/root/reference/README.md:5-10); there is no ALE on
this pool, MiniPong is the stand-in (parity with the emulator unpinned).  Measured (profiles/r3_minipong_wallclock.txt):
the native cores reach an evaluation >= +15 within 6.7-10.1 s of job start on six of six runs, the eager-torch job within
59-89 s; the thresholds here leave room for evaluation noise (two games per evaluation, +-3 points) and box load.
The torch core takes minutes and only runs with PDA_SLOW_TESTS=1.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(core, seed, goal, max_steps, timeout, extra=()):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDA_OVERRIDES"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench", "pong_wallclock.py"), "--core", core, "--seed", str(seed),
           "--goal", str(goal), "--max-steps", str(max_steps)] + list(extra)
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0, text[-3000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-3000:]
    return json.loads(lines[-1]), text


@pytest.mark.parametrize("core", ["fp32", "bf16"])
def test_native_cores_learn_minipong_within_a_minute(core):
    res, text = _run(core, seed=1, goal=12.0, max_steps=450000, timeout=170)
    assert res["reached"], text[-3000:]
    assert res["eval_at_goal"] >= 12.0 and res["wall_s_from_job_start"] < 60.0, res
    assert res["learner_updates_per_sec"] > 3000 and res["env_steps_per_sec"] > 30000, res


def test_learns_minipong_from_full_frame_stacks_too():
    """Config 8 uses the frame-deduplicated replay (memories/dedup_memory.py: the actors' runs, the plane ring and the
    kernels' frame windows feed the learner in the tests above); the same job on the plain HBM replay, which stores both
    frame stacks of every transition, with the 4 envs per actor process the first measurements used."""
    res, text = _run("fp32", seed=2, goal=12.0, max_steps=450000, timeout=170,
                     extra=["--set", "memory_type=device", "--envs-per-actor", "4"])
    assert res["reached"], text[-3000:]
    assert res["eval_at_goal"] >= 12.0 and res["wall_s_from_job_start"] < 60.0, res


@pytest.mark.skipif(os.environ.get("PDA_SLOW_TESTS", "0") != "1", reason="minutes of wall clock: PDA_SLOW_TESTS=1")
def test_torch_core_learns_minipong_too():
    res, text = _run("torch", seed=1, goal=12.0, max_steps=200000, timeout=420)
    assert res["reached"], text[-3000:]
    assert res["wall_s_from_job_start"] < 400.0, res
