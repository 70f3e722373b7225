#!/usr/bin/env python3
"""A whole frame of a heavy configuration on ONE GPU, rendered as SLICES row-cyclic slices one after the other (each slice
is exactly what one of SLICES GPUs would render, raytracer_amd/dist.py) so that progress is printed every slice.

    python scripts/full_frame.py --spheres 4096 --width 1920 --height 1080 --aa 4 [--bvh] [--slices 16]

Prints one JSON line: rays, kernel seconds (sum and slowest slice), Mrays/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from raytracer_amd.synth import synthetic_scene  # noqa: E402
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode, RowBands  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spheres", type=int, default=4096)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--aa", type=int, default=4)
    ap.add_argument("--slices", type=int, default=16)
    ap.add_argument("--bvh", action="store_true")
    ap.add_argument("--slice-range", type=int, nargs=2, default=None, metavar=("FIRST", "END"), help="only slices FIRST..END-1 (a frame too long for one run)")
    args = ap.parse_args()
    scene, _ = synthetic_scene(args.spheres)
    world = DeviceWorld(scene)
    tracer = GpuImageTracer(args.width, args.height, scene.camera, args.aa)
    mode = RenderMode("path", n_rays=10, roulette=3, max_depth=5, seq_number=54, bvh=args.bvh)
    rays = 0
    kernel_ms = []
    total = 0.0
    t_start = time.perf_counter()
    first, end = args.slice_range or (0, args.slices)
    for k in range(first, end):
        rows = RowBands.cyclic(args.height, k, args.slices, band_rows=1)
        out = torch.empty((rows.n_local_rows, args.width, 3), dtype=torch.float32, device="cuda")
        tracer.fire_all_rays(world, mode, rows, out=out)
        torch.cuda.synchronize()
        rays += int(tracer.last_stats.rays)
        kernel_ms.append(float(tracer.last_stats.kernel_ms))
        total += float(out.sum())
        print(f"[full_frame] slice {k + 1}/{args.slices}: {kernel_ms[-1] / 1e3:.1f} s, {rays} rays so far", file=sys.stderr, flush=True)
    wall = time.perf_counter() - t_start
    world.close()
    ks = sum(kernel_ms) / 1e3
    print(json.dumps({"config": f"synthetic {args.spheres} spheres" + (" (BVH)" if args.bvh else ""), "size": f"{args.width}x{args.height} AA={args.aa}",
                      "slices": args.slices, "slice_range": [first, end], "rays": rays, "kernel_s_one_gpu": ks, "slowest_slice_s": max(kernel_ms) / 1e3,
                      "fastest_slice_s": min(kernel_ms) / 1e3, "wall_s": wall, "Mrays_per_s": rays / ks / 1e6,
                      "mean_radiance_of_slices": total / (args.width * args.height * 3 * (end - first) / args.slices)}), flush=True)


if __name__ == "__main__":
    main()
