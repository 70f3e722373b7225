#!/usr/bin/env python3
"""End-to-end time of BASELINE configs C1-C4 through the C++ route (reference parser -> flattener -> C-ABI -> HdrImage):
oracle/_ref/raytracer_gpu <scene> --repeat 5.  Reports, per config, the first call (HIP runtime + code object + upload of
the World + frame) and a frame on the resident World (fire_all_rays alone: flatten + compare + render + one D2H copy),
beside the kernel time of the same frame.  The reference's sample files do not travel to the GPU box: the scene text is
written back from the flattened fixtures in tests/golden/ (raytracer_amd/scene_text.py; same objects, same order).

    python scripts/cpp_end_to_end.py > profiles/r3/cpp_end_to_end.jsonl
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_gpu")
sys.path.insert(0, ROOT)
from raytracer_amd.scene import FlatScene  # noqa: E402
from raytracer_amd.scene_text import to_scene_text  # noqa: E402

CONFIGS = [
    ("C1", "demo_onoff", ["--mode", "onoff", "--width", 1280, "--height", 960, "--aa", 1]),
    ("C2-flat", "demo_path", ["--mode", "flat", "--width", 1280, "--height", 960, "--aa", 2]),
    ("C2-pointlight", "demo_path", ["--mode", "pointlight", "--width", 1280, "--height", 960, "--aa", 2]),
    ("C3", "demo_path", ["--mode", "path", "--width", 1280, "--height", 960, "--aa", 2]),
    ("C4", "csg_example", ["--mode", "path", "--width", 1920, "--height", 1080, "--aa", 4]),
]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        for cid, scene, args in CONFIGS:
            out, txt = os.path.join(tmp, "o.pfm"), os.path.join(tmp, scene + ".txt")
            with open(txt, "w") as f:
                f.write(to_scene_text(FlatScene.from_json(os.path.join(ROOT, "tests", "golden", scene + ".scene.json"))))
            res = subprocess.run([BIN, txt, "-o", out, "--one-gpu", "--repeat", "5", *map(str, args)],
                                 capture_output=True, text=True, timeout=300)
            if res.returncode != 0:
                print(json.dumps({"id": cid, "error": res.stderr[-500:]}))
                continue
            info = json.loads(res.stdout)
            print(json.dumps({"id": cid, "scene": scene, "args": " ".join(map(str, args)), "first_call_ms": info["elapsed_s"] * 1e3,
                              "resident_frame_ms": info["resident_frame_ms"], "kernel_ms": info["kernel_ms"], "rays": info["rays"],
                              "kernel": info["kernel"]}), flush=True)


if __name__ == "__main__":
    main()
