#!/usr/bin/env python3
"""Times BASELINE.json's configs C1-C4 (the reference's sample scenes, from the flattened fixtures in tests/golden/)
and, optionally, synthetic scenes, on one GPU.  Prints one JSON object per line.

    python scripts/run_configs.py [--synth N WIDTH HEIGHT AA]...
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from raytracer_amd.scene import FlatScene  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(label, scene, mode, w, h, aa, reps=3):
    world = DeviceWorld(scene)
    tracer = GpuImageTracer(w, h, scene.camera, aa)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    tracer.fire_all_rays(world, mode, out=out)          # warm-up + stats
    st = tracer.last_stats
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tracer.fire_all_rays(world, mode, out=out, want_stats=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    world.close()
    print(json.dumps({"config": label, "mode": mode.mode, "size": f"{w}x{h} AA={aa}", "objects": scene.count("n_objects"), "kernel": st.kernel.decode(),
                      "rays": int(st.rays), "samples": int(st.samples), "capped_samples": int(st.capped_samples),
                      "frame_ms": best * 1e3, "Mrays_per_s": st.rays / best / 1e6,
                      "mean_radiance": float(out.mean()),
                      "frame_sha256": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synth", nargs=4, type=int, action="append", default=[], metavar=("N", "W", "H", "AA"))
    ap.add_argument("--skip-samples", action="store_true")
    ap.add_argument("--bvh", action="store_true", help="synthetic scenes: RT_FLAG_ACCEL_BVH")
    ap.add_argument("--scan-all", action="store_true", help="synthetic scenes: RT_FLAG_SCAN_ALL_TILES (no cluster bounds)")
    args = ap.parse_args()
    if not args.skip_samples:
        onoff = FlatScene.from_json(os.path.join(GOLDEN, "demo_onoff.scene.json"))
        demo = FlatScene.from_json(os.path.join(GOLDEN, "demo_path.scene.json"))
        csg = FlatScene.from_json(os.path.join(GOLDEN, "csg_example.scene.json"))
        run("C1 demo_onoff", onoff, RenderMode("onoff"), 1280, 960, 1)
        run("C2 demo_path", demo, RenderMode("flat"), 1280, 960, 2)
        run("C2 demo_path", demo, RenderMode("pointlight"), 1280, 960, 2)
        run("C3 demo_path", demo, RenderMode("path"), 1280, 960, 2)
        run("C4 csg_example", csg, RenderMode("path"), 1920, 1080, 4)
    for n, w, h, aa in args.synth:
        scene, _ = synthetic_scene(n)
        run(f"synthetic {n} spheres" + (" (BVH)" if args.bvh else ""), scene, RenderMode("path", bvh=args.bvh, scan_all_tiles=args.scan_all), w, h, aa, reps=1)


if __name__ == "__main__":
    main()
