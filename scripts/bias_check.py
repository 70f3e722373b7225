import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from raytracer_amd.synth import synthetic_scene
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
ref = dict(np.load('/root/repo/tests/golden/synth_large_images.npz'))
for n, w, h in ((4096, 48, 27), (512, 128, 72)):
    scene = synthetic_scene(n)[0]
    world = DeviceWorld(scene)
    tr = GpuImageTracer(w, h, scene.camera, 2)
    r = ref[f"path_synth{n}_{w}x{h}_aa2_serial"].astype(np.float64)
    print(n, "reference mean", r.mean(), "per channel", r.mean(axis=(0, 1)))
    for name, kw in (("wide", {}),):
        ms, rays = [], []
        for seed in range(100, 106):
            img = tr.fire_all_rays(world, RenderMode("path", n_rays=10, roulette=3, max_depth=5, seq_number=seed, **kw)).astype(np.float64)
            ms.append(img.mean()); rays.append(int(tr.last_stats.rays))
        ms = np.array(ms)
        print(f"  {name:10s}", flush=True) if False else print(f"  {name:10s} mean {ms.mean():.5f} +- {ms.std(ddof=1) / np.sqrt(len(ms)):.5f} (std {ms.std(ddof=1):.5f}) rays {np.mean(rays):.4g}", flush=True)
    world.close()
