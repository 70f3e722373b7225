#!/usr/bin/env python3
"""Fixture for the tone-mapping tail, made by RUNNING THE UNMODIFIED REFERENCE (oracle/_ref/rt_ref tonemap = HdrImage's own
average_luminosity / normalize_image / clamp_image) on two of the golden renders in images.npz:

  tonemap.npz:  <key>_avg  the average luminosity, <key>_hdr  the normalised + clamped pixels, for alpha = 0.18;
                <key>_dark_hdr  the same with the fixed dark-mode luminosity 0.1 (src/raytracer.cpp:180)

Only runnable where /root/reference exists.  Re-run with:  python tests/golden/make_tonemap_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from oracle_lib import REF_BIN  # noqa: E402

KEYS = ["flat_demo_path_160x120_aa2", "path_csg_example_96x72_aa6_serial"]


def write_pfm(path, img):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img[::-1], dtype="<f4").tobytes())


def main():
    images = np.load(os.path.join(HERE, "images.npz"))
    tmp = tempfile.mkdtemp(prefix="rttone")
    out = {}
    for key in KEYS:
        img = images[key]
        pfm, binf = os.path.join(tmp, "i.pfm"), os.path.join(tmp, "o.bin")
        write_pfm(pfm, img)
        info = json.loads(subprocess.run([REF_BIN, "tonemap", pfm, binf, "0.18"], check=True, capture_output=True, text=True).stdout)
        out[key + "_avg"] = np.float32(info["avg_luminosity"])
        out[key + "_hdr"] = np.fromfile(binf, dtype=np.float32).reshape(img.shape)
        subprocess.run([REF_BIN, "tonemap", pfm, binf, "0.18", "0.1"], check=True, capture_output=True, text=True)
        out[key + "_dark_hdr"] = np.fromfile(binf, dtype=np.float32).reshape(img.shape)
        print(key, info)
    np.savez_compressed(os.path.join(HERE, "tonemap.npz"), **out)


if __name__ == "__main__":
    main()
