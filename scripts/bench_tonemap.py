#!/usr/bin/env python3
"""Times the tone-mapping tail on a framebuffer resident in HBM (HIP events on torch's current stream) and prints
achieved HBM GB/s against algorithmic bytes: luminosity pass 12 B/pixel read; map 12 B read + 3 B written.

    python scripts/bench_tonemap.py [--width 3840 --height 2160]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd.tracer import average_luminosity, tonemap  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    n = args.width * args.height
    img = torch.rand((args.height, args.width, 3), device="cuda") * 2.0
    avg = average_luminosity(img)
    rows = []
    for label, fn, nbytes in (
        ("average_luminosity (fixed-point sum; includes the 8-byte D2H + sync)", lambda: average_luminosity(img), 12 * n),
        ("average_luminosity (serial binary32 order, one wave)", lambda: average_luminosity(img, serial_sum=True), 16 * n),
        ("tonemap map only (avg given): normalize + clamp + gamma 2.2 -> u8", lambda: tonemap(img, 0.18, 2.2, avg_luminosity=avg), 15 * n),
        ("tonemap map only, gamma 1", lambda: tonemap(img, 0.18, 1.0, avg_luminosity=avg), 15 * n),
        ("tonemap whole tail (luminosity + map)", lambda: tonemap(img, 0.18, 2.2), 27 * n),
    ):
        ms = timed(fn)
        rows.append({"step": label, "ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBS})
    print(json.dumps({"image": f"{args.width}x{args.height}", "pixels": n, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
