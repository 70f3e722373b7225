#!/usr/bin/env python3
"""Reference renders of the LARGE synthetic scenes (the worlds the wide path kernel is built for), made by RUNNING THE
UNMODIFIED REFERENCE (oracle/_ref/rt_ref) on the scene-DSL text raytracer_amd/synth.py emits:

  synth_large_images.npz   path 10/3/5, serial PCG(42, 54):  N=512 at 128x72 AA=2,  N=4096 at 48x27 AA=2
                           + the reference's flat renders of both at 160x90 AA=1 (bit-level pins of the scan)

Minutes of CPU (the 4096-sphere render is ~6e7 closest-hit queries over 4098 objects, single-threaded by construction).
Only runnable where /root/reference exists.  Re-run with:  python tests/golden/make_synth_large_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import REF_BIN, read_pfm  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402

JOBS = {
    "path_synth512_128x72_aa2_serial": (512, ["--mode", "path", "--width", 128, "--height", 72, "--aa", 2, "--n_rays", 10, "--roulette", 3,
                                              "--max-depth", 5, "--seq", 54]),
    "path_synth4096_48x27_aa2_serial": (4096, ["--mode", "path", "--width", 48, "--height", 27, "--aa", 2, "--n_rays", 10, "--roulette", 3,
                                               "--max-depth", 5, "--seq", 54]),
    "flat_synth512_160x90_aa1": (512, ["--mode", "flat", "--width", 160, "--height", 90, "--aa", 1]),
    "flat_synth4096_160x90_aa1": (4096, ["--mode", "flat", "--width", 160, "--height", 90, "--aa", 1]),
}


def main():
    tmp = tempfile.mkdtemp(prefix="rtsynthlarge")
    texts = {}
    for n in (512, 4096):
        texts[n] = os.path.join(tmp, f"synth_{n}.txt")
        with open(texts[n], "w") as f:
            f.write(synthetic_scene(n)[1])

    def job(item):
        name, (n, args) = item
        pfm = os.path.join(tmp, name + ".pfm")
        out = subprocess.run([REF_BIN, "render", texts[n], "-o", pfm, *map(str, args)], check=True, capture_output=True, text=True).stdout
        print(name, json.loads(out), flush=True)
        return name, read_pfm(pfm)

    with ThreadPoolExecutor(4) as ex:
        images = dict(ex.map(job, JOBS.items()))
    np.savez_compressed(os.path.join(HERE, "synth_large_images.npz"), **images)


if __name__ == "__main__":
    main()
