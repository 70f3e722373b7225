#!/usr/bin/env python3
"""Debug: ray counts and image differences of oracle / scalar exact / scalar fast / wide on synthetic scenes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode  # noqa: E402

W, H, AA = 192, 108, 2
for n in [int(x) for x in sys.argv[1:]] or [64, 128, 200]:
    scene, _ = synthetic_scene(n)
    ref, st = O.render(scene, O.make_cfg(W, H, "path", spp_edge=AA))
    print(f"N={n} oracle rays={st.rays} mean={ref.mean():.6f}", flush=True)
    world = DeviceWorld(scene)
    t = GpuImageTracer(W, H, scene.camera, AA)
    for label, mode in (("exact", RenderMode("path", exact=True)), ("fast/auto", RenderMode("path")),
                        ("objspace", RenderMode("path", object_space_spheres=True))):
        img = t.fire_all_rays(world, mode)
        d = np.abs(img - ref).max(axis=2)
        print(f"   {label:10s} rays={t.last_stats.rays} ({t.last_stats.rays / st.rays - 1:+.3%}) mean={img.mean():.6f} "
              f"pixels>1e-4: {np.mean(d > 1e-4):.3%}  >1e-2: {np.mean(d > 1e-2):.3%}", flush=True)
    world.close()
