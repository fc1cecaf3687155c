"""Build ops/libpda_hip.so: every .hip source in ops/csrc compiled for gfx950 and
linked into one shared library with a C ABI.

    python -m pytorch_distributed_amd.ops.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs on the CPU dev box; the .so is
git-ignored but travels to the GPU box with the tree.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libpda_hip.so")
ARCH = "gfx950"
# -ffast-math for reassociation / approximate transcendentals, but NOT finite-math-only: kernels use +-INFINITY as
# real values (arg-max sentinels in rl_kernels.hip, +inf leaves of the min tree), which finite-math-only would make
# undefined behaviour.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-ffast-math", "-fno-finite-math-only"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LAST_REPORT = {}       # source basename -> "compiled" | "reused" for the most recent build() (+ "link")


def build(force=False, verbose=False, extra_flags=()):
    """Compile what is stale (everything with force=True, or PDA_BUILD_FORCE=1) and link.  LAST_REPORT says, per
    object, whether it was compiled from source in this call or reused from an earlier one."""
    force = force or os.environ.get("PDA_BUILD_FORCE", "0") == "1"
    LAST_REPORT.clear()
    os.makedirs(OBJ_DIR, exist_ok=True)
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    headers += sorted(glob.glob(os.path.join(HERE, "..", "parallel", "csrc", "*.h")))   # mailbox protocol
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in sources:
        obj = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        # (the flags are part of the recipe: an object built with other flags is stale too)
        stamp = obj + ".flags"
        flags = " ".join(FLAGS + list(extra_flags))
        same_flags = os.path.exists(stamp) and open(stamp).read() == flags
        if force or not same_flags or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj])
            LAST_REPORT[os.path.basename(src)] = "compiled"
            if os.path.exists(stamp):
                os.remove(stamp)
        else:
            LAST_REPORT[os.path.basename(src)] = "reused"

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), res.stdout))
        if verbose and res.stdout.strip():
            print(res.stdout)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    for cmd in jobs:
        with open(cmd[-1] + ".flags", "w") as fh:
            fh.write(" ".join(FLAGS + list(extra_flags)))
    LAST_REPORT["link"] = "linked" if (force or jobs or _stale(LIB, objs)) else "reused"
    if force or jobs or _stale(LIB, objs):
        tmp = LIB + ".tmp%d" % os.getpid()
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs + ["-lpthread"])
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print("built", path, LAST_REPORT)
