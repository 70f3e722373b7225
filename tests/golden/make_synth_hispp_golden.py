#!/usr/bin/env python3
"""High-spp reference renders of the headline worlds (N=512 and N=4096 random spheres + floor + sky), made by RUNNING THE
UNMODIFIED REFERENCE (oracle/_ref/rt_ref) on the scene-DSL text raytracer_amd/synth.py emits, path 10/3/5, serial PCG(42, 54):

  synth_hispp_images.npz    path_synth512_64x36_aa6_serial   (36 spp; ~9 CPU-minutes)
                            path_synth4096_24x14_aa1_serial_mean36 / _std36: mean and per-pixel standard deviation of 36 one-sample
                            renders (AA=1: the ray through the pixel centre, no jitter) with --seq-number 54 ... 89, each its own
                            short run of the reference (~1.5 CPU-minutes: 336 samples x ~12 000 rays over 4098 objects; ~1 CPU-hour in
                            all), so that the set can be made in pieces; the device side is rendered the same way.
  synth_hispp_meta.json     the reference's own elapsed seconds per render

They are the targets of the RMSE-bound tests of the large-world kernels (tests/test_synth_large.py).
Only runnable where /root/reference exists.   python tests/golden/make_synth_hispp_golden.py [512 | 409654 ... 409689 | 0 (assemble only)] ...   (parts already made are skipped)
Each render is written to its own .npy first, so the two can be made in separate runs; the .npz is assembled from what exists.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import REF_BIN, read_pfm  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402

N4096_SEEDS = list(range(54, 54 + 36))
JOBS = {
    512: ("path_synth512_64x36_aa6_serial", ["--mode", "path", "--width", 64, "--height", 36, "--aa", 6, "--n_rays", 10, "--roulette", 3, "--max-depth", 5, "--seq", 54]),
}
for _seq in N4096_SEEDS:  # job key 4096 * 100 + seq
    JOBS[4096 * 100 + _seq] = (f"path_synth4096_24x14_aa1_serial_seq{_seq}", ["--mode", "path", "--width", 24, "--height", 14, "--aa", 1, "--n_rays", 10, "--roulette", 3,
                                                                           "--max-depth", 5, "--seq", _seq])
PARTS = os.path.join(HERE, "_hispp_parts")


def main():
    which = [int(a) for a in sys.argv[1:]] or list(JOBS)  # job keys: 512, 409654 ... 409657; "assemble" jobs: none
    os.makedirs(PARTS, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="rthispp")
    for n in which:
        if n == 0:
            continue
        name, args = JOBS[n]
        if os.path.exists(os.path.join(PARTS, name + ".npy")):
            continue
        n_spheres = n if n < 10000 else n // 100
        text = os.path.join(tmp, f"synth_{n}.txt")
        with open(text, "w") as f:
            f.write(synthetic_scene(n_spheres)[1])
        pfm = os.path.join(tmp, name + ".pfm")
        out = subprocess.run([REF_BIN, "render", text, "-o", pfm, *map(str, args)], check=True, capture_output=True, text=True).stdout
        info = json.loads(out)
        print(name, info, flush=True)
        np.save(os.path.join(PARTS, name + ".npy"), read_pfm(pfm))
        with open(os.path.join(PARTS, name + ".json"), "w") as f:
            json.dump(info, f)
    images, meta = {}, {}
    name = JOBS[512][0]
    if os.path.exists(os.path.join(PARTS, name + ".npy")):
        images[name] = np.load(os.path.join(PARTS, name + ".npy"))
        meta[name] = json.load(open(os.path.join(PARTS, name + ".json")))
    parts = [JOBS[4096 * 100 + q][0] for q in N4096_SEEDS]
    if all(os.path.exists(os.path.join(PARTS, q + ".npy")) for q in parts):
        stack = np.stack([np.load(os.path.join(PARTS, q + ".npy")) for q in parts]).astype(np.float64)
        images["path_synth4096_24x14_aa1_serial_mean36"] = stack.mean(0).astype(np.float32)
        images["path_synth4096_24x14_aa1_serial_std36"] = stack.std(0, ddof=1).astype(np.float32)
        meta["path_synth4096_24x14_aa1_serial_mean36"] = {"seeds": N4096_SEEDS, "elapsed_s": sum(json.load(open(os.path.join(PARTS, q + ".json")))["elapsed_s"] for q in parts)}
    else:
        print("N=4096: %d of %d parts present" % (sum(os.path.exists(os.path.join(PARTS, q + ".npy")) for q in parts), len(parts)))
    np.savez_compressed(os.path.join(HERE, "synth_hispp_images.npz"), **images)
    with open(os.path.join(HERE, "synth_hispp_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
