#!/usr/bin/env python3
"""Instruction mix and s_waitcnt vmcnt sequence per kernel of a gfx950 assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o x.s file.hip && python tools/probes/asm_mix.py x.s

What the round-2 GEMM rework was steered by: run-time integer divisions in the operand loaders (v_rcp / v_mul_hi /
v_mul_lo counts), divergent address branches (s_cbranch) and the prefetch ring being drained (vmcnt(0) inside the k-loop).
"""
import re
import sys


def main(path, pat=""):
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:\s+; @", l)]
    for n, i in enumerate(starts):
        body = []
        for b in txt[i + 1:starts[n + 1] if n + 1 < len(starts) else len(txt)]:
            if b.startswith(".Lfunc_end"):
                break
            body.append(b)
        name = re.sub(r"Li(\d+)E", r"\1,", txt[i].split(":")[0])
        if pat not in name:
            continue
        cnt = lambda p: sum(1 for b in body if re.match(r"\s+" + p, b))
        vm = [re.search(r"vmcnt\((\d+)\)", b).group(1) for b in body if "vmcnt" in b]
        print(name[:170])
        print("  instrs %d  mfma %d  branches %d  v_rcp %d  v_mul_lo %d  v_mul_hi %d  global_load %d  ds_read %d  vmcnt: %s"
              % (cnt(r"[vs]_|ds_|global_|buffer_"), cnt("v_mfma"), cnt("s_cbranch"), cnt("v_rcp"), cnt("v_mul_lo"),
                 cnt("v_mul_hi"), cnt("global_load"), cnt("ds_read"), ",".join(vm)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
