#!/usr/bin/env python3
"""Does a fused (FMA) path kernel render the same EXPECTED image as the unfused one (= the reference's arithmetic)?
Image means over several seeds, per kernel variant, with their standard errors.

    python scripts/bias_check.py synth:4096:48x27 synth:512:128x72 scene:demo_path:320x240 scene:csg_example:320x180
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.scene import FlatScene  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode  # noqa: E402

for spec in sys.argv[1:]:
    kind, name, size = spec.split(":")
    w, h = map(int, size.split("x"))
    scene = synthetic_scene(int(name))[0] if kind == "synth" else FlatScene.from_json(os.path.join(ROOT, "tests", "golden", f"{name}.scene.json"))
    world = DeviceWorld(scene)
    tr = GpuImageTracer(w, h, scene.camera, 2)
    print(spec, flush=True)
    variants = [("default", {}), ("exact", dict(exact=True))]
    if kind == "synth":
        variants += [("valu-only", dict(valu_only=True)), ("objspace", dict(object_space_spheres=True)), ("bvh", dict(bvh=True))]
    for label, kw in variants:
        ms, rays = [], []
        for seed in range(100, 100 + int(os.environ.get("BIAS_SEEDS", "8"))):
            img = tr.fire_all_rays(world, RenderMode("path", n_rays=10, roulette=3, max_depth=5, seq_number=seed, **kw)).astype(np.float64)
            ms.append(img.mean())
            rays.append(int(tr.last_stats.rays))
            print(f"    [{label} seed {seed}: {tr.last_stats.kernel_ms:.0f} ms]", file=sys.stderr, flush=True)
        ms = np.array(ms)
        print(f"  {label:9s} {tr.last_stats.kernel.decode():34s} mean {ms.mean():.6f} +- {ms.std(ddof=1) / np.sqrt(len(ms)):.6f}  rays {np.mean(rays):.5g}", flush=True)
    world.close()
