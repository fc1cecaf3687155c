"""Race detection for the native host runtime: the lock-free ring, the ingestion
pump, the mailbox protocol and the frame-deduplicated run packer are exercised by a multi-threaded C++ driver built with
ThreadSanitizer (host code only; GPU sanitizers are not available on this pool)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ring_stress.cpp")


def _build(tmp_path, flags, name):
    exe = str(tmp_path / name)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread"] + flags + ["-o", exe, SRC]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        pytest.skip("cannot build with %s here: %s" % (flags, res.stdout[-400:]))
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_host_runtime_under_sanitizer(tmp_path, sanitizer):
    exe = _build(tmp_path, ["-fsanitize=%s" % sanitizer], "ring_stress_" + sanitizer)
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=0 report_signal_unsafe=0"
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    out = res.stdout
    if "FATAL: ThreadSanitizer" in out or "Sanitizer: failed to" in out or "unexpected memory mapping" in out \
            or "runtime does not come first" in out:
        pytest.skip("sanitizer runtime cannot start in this container: " + out[-300:])
    assert res.returncode == 0, out[-3000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out, out[-3000:]
    assert "ring_stress ok" in out
    assert "runs ok" in out         # the frame-deduplicated run packer (4 streams -> one ring -> checking consumer)
