#!/usr/bin/env python3
"""Per-kernel time + hardware counters of the native DQN learner step, as text under profiles/.

    python tools/profile_step.py --batch 128 --tag r1_b128          # on the GPU box (through gpurun)

Runs rocprofv3 once for the kernel trace and once per counter group (rocprofv3 does not split
counters over passes; a group holds at most 8 SQ / 4 TCC-slot counters), every pass under its own
hard timeout, on tools/pmc_step.py -- a runner that never starts a child process.  The summaries
(tools/rocpd_summary.py tables + derived MFMA utilisation, LDS conflict rate, HBM bytes per
kernel) are written to gpurun_out/<tag>/ and copied to profiles/<tag>_*.txt.
"""
import argparse
import glob
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# SQ_VALU_MFMA_BUSY_CYCLES per matrix-pipe busy cycle (summed over the chip's SIMDs).  Calibrated on a launch whose
# MFMA count is known exactly (the fc1 forward GEMM, cross-check line at the end of every counter file): the counter
# reads 16.00 per v_mfma_f32_16x16x32_bf16, which holds its SIMD's matrix pipe for 16 cycles => 1 tick per busy cycle.
MFMA_COUNTER_PER_CYCLE = 1.0
MFMA_CYCLES_16x16x32 = 16.0

GROUPS = {
    "sq": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU",
           "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY"],
    "fetch": ["FETCH_SIZE", "TCC_HIT_sum"],
    "write": ["WRITE_SIZE", "TCC_MISS_sum", "GRBM_GUI_ACTIVE"],
}


def run(cmd, timeout, log):
    with open(log, "w") as fh:
        try:
            return subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, timeout=timeout, cwd=ROOT).returncode
        except subprocess.TimeoutExpired:
            return 124


def counters_by_kernel(db):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    out = {}
    for k, c, n, s in rows:
        out.setdefault(k, {})[c] = (n, s)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tag", default="step")
    ap.add_argument("--per", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true", help="profile graph-replayed steps (tools/pmc_step.py --graph)")
    a = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out", a.tag)      # text summaries and pass logs (merged back by gpurun)
    raw = os.path.join("/tmp", "pda_prof", a.tag)       # rocprofv3 databases: large, stay on the GPU box
    os.makedirs(out, exist_ok=True)
    os.makedirs(raw, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    env_tmp = "/tmp"
    os.environ["TMPDIR"] = env_tmp
    target = [sys.executable, os.path.join(ROOT, "tools", "pmc_step.py"), "--batch", str(a.batch),
              "--steps", str(a.steps), "--precision", a.precision] + (["--per"] if a.per else []) + \
        (["--graph"] if a.graph else [])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocpd_summary import short

    # pass 1: kernel trace
    d = os.path.join(raw, "trace")
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "t", "--"] + target, 240,
             os.path.join(out, "trace.log"))
    print("kernel-trace pass rc=%d" % rc, flush=True)
    if rc != 0:
        return rc
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    summary = os.path.join(ROOT, "profiles", "%s_kernels.txt" % a.tag)
    with open(summary, "w") as fh:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db, "--steps", str(a.steps),
                        "--top", "30"], stdout=fh, cwd=ROOT)
    shutil.copy(summary, os.path.join(out, os.path.basename(summary)))
    con = sqlite3.connect(db)
    durs = dict(con.execute("select name, sum(duration) from kernels group by name").fetchall())
    calls = dict(con.execute("select name, count(*) from kernels group by name").fetchall())

    # counter passes
    merged = {}
    for gname, ctrs in GROUPS.items():
        d = os.path.join(raw, "pmc_" + gname)
        rc = run(["timeout", "-s", "KILL", "120", "rocprofv3", "--pmc"] + ctrs + ["-d", d, "-o", "p", "--"] + target,
                 180, os.path.join(out, "pmc_%s.log" % gname))
        print("pmc pass %s rc=%d" % (gname, rc), flush=True)
        if rc != 0:
            return rc       # a failed counter pass ends the run: nothing further on the GPU
        dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        for k, v in counters_by_kernel(dbs[0]).items():
            merged.setdefault(k, {}).update(v)

    path = os.path.join(ROOT, "profiles", "%s_counters.txt" % a.tag)
    with open(path, "w") as fh:
        fh.write("# hardware counters of the native DQN learner step, batch %d, %d %s steps (tools/profile_step.py)\n"
                 % (a.batch, a.steps, "graph-replayed" if a.graph else "eager"))
        fh.write("# mfma_busy%% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel time * 2.4 GHz): the counter adds, over every\n"
                 "#   SIMD of the chip, the cycles its matrix pipe is busy (cross-check at the end: it reads 16.00 per\n"
                 "#   v_mfma_f32_16x16x32_bf16, which holds its pipe for 16 cycles); at the 2.4 GHz ceiling clock this is a LOWER\n"
                 "#   bound of the pipe's duty cycle (the chip clocks lower under load) and cannot exceed 100 %%.\n"
                 "#   mfma_cnt = SQ_VALU_MFMA_BUSY_CYCLES / 16 = 16x16x32 MFMA wave-instructions per launch.\n")
        fh.write("# lds_conflict%% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;  wait%% = SQ_WAIT_ANY / SQ_WAVE_CYCLES\n")
        fh.write("# hbm MB/launch = (2*FETCH_SIZE + WRITE_SIZE) KB / 1024 per dispatch (gfx950 FETCH_SIZE counts half of a wide read)\n")
        fh.write("%-72s %6s %9s %8s %10s %10s %8s %8s %9s %9s %8s\n" % (
            "kernel", "calls", "us/call", "waves", "mfma_busy%", "mfma_cnt", "lds_cf%", "wait%", "rd_MB", "wr_MB",
            "L2hit%"))
        for k in sorted(merged, key=lambda n: -durs.get(n, 0)):
            c = merged[k]
            g = lambda name: (c.get(name, (0, 0.0))[1])
            n = max(c.get("SQ_WAVES", (1, 0))[0], 1)
            ncalls = max(calls.get(k, n), 1)
            us_call = durs.get(k, 0) / 1e3 / ncalls
            mfma_raw = g("SQ_VALU_MFMA_BUSY_CYCLES") / n          # per launch, summed over the chip
            mf = 100.0 * mfma_raw / MFMA_COUNTER_PER_CYCLE / (1024.0 * us_call * 2400.0) if us_call else 0.0
            ldsa = g("SQ_LDS_IDX_ACTIVE")
            cf = 100.0 * g("SQ_LDS_BANK_CONFLICT") / ldsa if ldsa else 0.0
            wc = g("SQ_WAVE_CYCLES")
            wt = 100.0 * g("SQ_WAIT_ANY") / wc if wc else 0.0
            hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
            l2 = 100.0 * hit / (hit + miss) if hit + miss else 0.0
            fh.write("%-72s %6d %9.2f %8.0f %10.1f %10.0f %8.1f %8.1f %9.2f %9.2f %8.1f\n" % (
                short(k, 72), calls.get(k, n), us_call, g("SQ_WAVES") / n,
                mf, mfma_raw / MFMA_CYCLES_16x16x32, cf, wt, 2.0 * g("FETCH_SIZE") / n / 1024.0, g("WRITE_SIZE") / n / 1024.0, l2))
        # hand cross-check of the MFMA counter on a launch whose instruction count is known exactly: the fc1 forward
        # GEMM (64x64 tiles, 4 waves x 2x2 fragments of 16x16x32): per launch
        #   groups * ceil(B/64) * (512/64) tiles * sum over k-slices of ceil(len/64) k-tiles * 2 k-steps * 16 MFMA per tile
        for k, c in merged.items():
            if ("EpiAtomicF32" in k or "EpiSlabF32" in k) and "SrcDenseK" in k and "SrcSplit" not in k \
                    and a.precision == "bf16":
                import math
                B, F, HID, groups = a.batch, 3136, 512, 2
                tiles = groups * math.ceil(B / 64) * (HID // 64)
                target_blocks = int(os.environ.get("PDA_FC1_BLOCKS", "256"))
                slices = max(1, target_blocks // max(tiles, 1))
                ks = max(128, math.ceil(F / slices))
                ks = (ks + 63) // 64 * 64
                ktiles = sum(math.ceil((min(F, s0 + ks) - s0) / 64) for s0 in range(0, F, ks))
                expect = tiles * ktiles * 2 * 16
                nn_ = max(c.get("SQ_WAVES", (1, 0))[0], 1)
                got = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1] / nn_
                fh.write("# cross-check (%s): %d MFMA wave-instructions per launch by hand; counter %.0f = %.2f per "
                         "instruction (16 pipe cycles each => the counter ticks %.2f x per busy cycle)\n"
                         % (short(k, 60), expect, got, got / expect, got / expect / 16.0))
    shutil.copy(path, os.path.join(out, os.path.basename(path)))
    print(open(path).read())
    return 0


if __name__ == "__main__":
    sys.exit(main())
