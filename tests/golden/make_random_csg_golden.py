#!/usr/bin/env python3
"""Fixtures that pin the oracle (and through it the GPU) on worlds the reference's samples do not cover: random CSG
trees of all four operations with rotated / non-uniformly scaled leaves and nodes, ellipsoids, several planes, checkered
pigments, several lights, perspective and orthogonal cameras.  Made by RUNNING THE UNMODIFIED REFERENCE
(oracle/_ref/rt_ref) on scene-DSL text written here:

  random_csg_<k>.scene.json   the World/Camera the reference's parser builds from the text, flattened
  random_csg_images.npz       the reference's own renders of scenes 0-5: onoff / flat (AA 1, 2) / pointlight / path (serial PCG)
  (scenes 6-23 are flattened only: more worlds for the GPU-against-oracle comparison)

Only runnable where /root/reference exists.  Re-run with:  python tests/golden/make_random_csg_golden.py
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

N_SCENES = 6    # with the reference's renders
N_EXTRA = 18    # flattened by the reference's parser only: GPU-against-oracle cases (tests/test_random_csg.py)
RENDERS = {
    "onoff_96x64_aa1": ["--mode", "onoff", "--width", 96, "--height", 64, "--aa", 1],
    "flat_96x64_aa1": ["--mode", "flat", "--width", 96, "--height", 64, "--aa", 1],
    "flat_72x48_aa2": ["--mode", "flat", "--width", 72, "--height", 48, "--aa", 2],
    "pointlight_96x64_aa1": ["--mode", "pointlight", "--width", 96, "--height", 64, "--aa", 1],
    "path_36x24_aa1_serial": ["--mode", "path", "--width", 36, "--height", 24, "--aa", 1, "--n_rays", 2, "--roulette", 2,
                              "--max-depth", 3, "--seq", 11],
}


def num(x) -> str:
    return np.format_float_positional(np.float32(x), unique=True, trim="0")


def vec(v):
    return "[%s, %s, %s]" % tuple(num(x) for x in v)


def col(c):
    return "<%s, %s, %s>" % tuple(num(x) for x in c)


def transformation(rng, spread, smin, smax, rotate=True):
    parts = ["translation(%s)" % vec(rng.uniform(-spread, spread, 3))]
    if rotate:
        for axis in rng.permutation(3)[: int(rng.integers(0, 3))]:
            parts.append("rotation_%s(%s)" % ("xyz"[axis], num(rng.uniform(-180, 180))))
    s = rng.uniform(smin, smax, 3)
    if rng.random() < 0.3:
        s[:] = s[0]
    parts.append("scaling(%s)" % vec(s))
    return " * ".join(parts)


def scene_text(k: int) -> str:
    rng = np.random.default_rng(1000 + k)
    out = ["# random CSG world %d (tests/golden/make_random_csg_golden.py)" % k]
    mats = []
    for i in range(6):
        kind = "specular" if i == 5 else "diffuse"
        if i in (1, 3):
            pig = "checkered(%s, %s, %d)" % (col(rng.uniform(0.1, 0.9, 3)), col(rng.uniform(0.1, 0.9, 3)), int(rng.integers(2, 7)))
        else:
            pig = "uniform(%s)" % col(rng.uniform(0.1, 0.9, 3))
        emit = col(rng.uniform(0.0, 0.6, 3)) if i == 2 else "<0, 0, 0>"
        out.append("material m%d(%s(%s), uniform(%s))" % (i, kind, pig, emit))
        mats.append("m%d" % i)
    out.append("material sky(diffuse(uniform(<0, 0, 0>)), uniform(<0.3, 0.4, 0.9>))")
    out.append("plane(translation([0, 0, -1.5]), m1)")
    out.append("sphere(scaling([60, 60, 60]), sky)")
    if k % 2:
        out.append("plane(translation([6, 0, 0]) * rotation_y(%s), m3)" % num(rng.uniform(60, 120)))
    names = 0

    def leaf():
        nonlocal names
        names += 1
        name = "s%d" % names
        m = mats[int(rng.integers(0, len(mats)))]
        if rng.random() < 0.25:
            out.append("norender plane %s(%s, %s)" % (name, transformation(rng, 0.4, 0.8, 1.2), m))
        else:
            out.append("norender sphere %s(%s, %s)" % (name, transformation(rng, 0.7, 0.5, 1.3), m))
        return name

    def tree(depth):
        nonlocal names
        if depth == 0 or rng.random() < 0.25:
            return leaf()
        a, b = tree(depth - 1), tree(depth - 1)
        names += 1
        name = "c%d" % names
        op = ["union", "intersection", "difference", "fusion"][int(rng.integers(0, 4))]
        t = "identity" if rng.random() < 0.3 else transformation(rng, 0.3, 0.8, 1.2)
        out.append("norender csg %s(%s, %s, %s, %s)" % (name, a, b, op, t))
        return name

    for j in range(2 + k % 2):
        a, b = tree(1 + (k + j) % 3), tree(1 + (k + 2 * j) % 2)
        op = ["difference", "intersection", "union", "fusion"][(k + j) % 4]
        place = "translation(%s) * rotation_z(%s)" % (vec([1.0 + 1.6 * j, -2.0 + 2.0 * j + rng.uniform(-0.3, 0.3), rng.uniform(-0.4, 0.4)]),
                                                      num(rng.uniform(-90, 90)))
        out.append("csg(%s, %s, %s, %s)" % (a, b, op, place))
    for j in range(3):
        out.append("sphere(%s, %s)" % (transformation(rng, 2.5, 0.2, 0.7), mats[int(rng.integers(0, len(mats)))]))
    out.append("point_light([-2, 3, 4], <1, 0.9, 0.8>, 0)")
    out.append("point_light(%s, %s, %s)" % (vec(rng.uniform(-3, 3, 3) + [0, 0, 4]), col(rng.uniform(0.2, 0.8, 3)), num(rng.uniform(0.5, 3))))
    if k % 3 == 2:
        out.append("camera(orthogonal, translation([-4, 0, 0.5]) * rotation_y(%s) * scaling([1, 4, 4]), 1.5)" % num(rng.uniform(5, 15)))
    else:
        out.append("camera(perspective, translation([-4, %s, 0.6]) * rotation_z(%s), exact_asp_ratio, %s)"
                   % (num(rng.uniform(-0.5, 0.5)), num(rng.uniform(-10, 10)), num(rng.uniform(1.0, 2.0))))
    return "\n".join(out) + "\n"


def main():
    tmp = tempfile.mkdtemp(prefix="rtcsg")

    def ref(*args):
        return subprocess.run([REF_BIN, *map(str, args)], check=True, capture_output=True, text=True).stdout

    images = {}
    pfm = os.path.join(tmp, "o.pfm")
    for k in range(N_SCENES):
        txt = os.path.join(tmp, "random_csg_%d.txt" % k)
        with open(txt, "w") as f:
            f.write(scene_text(k))
        ref("flatten", txt, os.path.join(HERE, "random_csg_%d.scene.json" % k))
        for name, args in RENDERS.items():
            info = json.loads(ref("render", txt, "-o", pfm, *args))
            images["%d_%s" % (k, name)] = read_pfm(pfm)
            print(k, name, info)
    np.savez_compressed(os.path.join(HERE, "random_csg_images.npz"), **images)
    for k in range(N_SCENES, N_SCENES + N_EXTRA):
        txt = os.path.join(tmp, "random_csg_%d.txt" % k)
        with open(txt, "w") as f:
            f.write(scene_text(k))
        ref("flatten", txt, os.path.join(HERE, "random_csg_%d.scene.json" % k))


if __name__ == "__main__":
    main()
