#!/usr/bin/env python3
"""Lists, per kernel of the library, the places where hipcc (ROCm 7.2) materialises a wave mask as a 0/1 VGPR and turns it back
into a mask later:

    v_cndmask_b32_e64 vX, 0, 1, s[a:b]        ...        v_cmp_ne_u32_e64 s[c:d], 1, vX

The pair is harmless when both instructions run under the same exec mask.  It is NOT when the first runs under a partial exec mask
inside a divergent loop and the second outside it: lanes that were inactive at the first keep a stale vX, and a wave-uniform flag
("this world has no CSG object") comes back true for them.  That is how PointLightTracer's mirror loop walked a CSG tree that does
not exist and faulted; `uniform_count()` (rt_device.h) keeps such counts in scalar registers.  Run after touching
loops whose trip count is a Scene field:

    python scripts/isa_audit.py            # compiles raytracer_amd/csrc/*.hip to assembly (about a minute), prints pairs per kernel
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracer_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden -S --cuda-device-only".split()


def audit(asm_text):
    out, kernel, pending = {}, None, {}
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending = m.group(1), {}
            continue
        if kernel is None:
            continue
        m = re.match(r"\s+v_cndmask_b32_e64 (v\d+), 0, 1, s\[", line)
        if m:
            pending[m.group(1)] = True
            continue
        m = re.match(r"\s+v_cmp_ne_u32_e64 s\[\d+:\d+\], 1, (v\d+)", line)
        if m and pending.get(m.group(1)):
            out[kernel] = out.get(kernel, 0) + 1
            continue
        m = re.match(r"\s+v_\w+ (v\d+),", line)  # the register is overwritten by something else: the pair is broken
        if m and m.group(1) in pending:
            pending.pop(m.group(1))
    return out


def main():
    total = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name in sorted(os.listdir(SRC)):
            if not name.endswith(".hip"):
                continue
            asm = os.path.join(tmp, name + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", os.path.join(ROOT, "include"), os.path.join(SRC, name), "-o", asm], check=True,
                           stderr=subprocess.DEVNULL)
            for kernel, n in sorted(audit(open(asm).read()).items()):
                demangled = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
                print(f"{name}: {n:3d} mask -> 0/1 VGPR -> mask pair(s) in {demangled[:110]}")
                total += n
    print(f"{total} pair(s) in all; each is safe only if both ends run under the same exec mask (see the docstring)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
