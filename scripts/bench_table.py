#!/usr/bin/env python3
"""Prints the figures DESIGN.md 7's table quotes from profiles/r3/a_bench_line.json (so that they are copied, not retyped)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r3", "a_bench_line.json")))
r = d["roofline"]
print(f"headline: value {d['value']:.1f} Mrays/s, {d['ms_per_step']:.1f} ms per step, frame estimate {d['frame_ms_estimate'] / 1e3:.1f} s, kernel {r['kernel']}")
print(f"  frac (fp32 flops executed) {r['frac']:.4f}, frac_algorithmic {r['frac_algorithmic']:.2f}, valu_issue {r['valu_issue']['frac']:.3f}, "
      f"mfma frac {r['mfma']['frac']:.4f}, tile products per 32 rays {r['mfma']['tile_products_per_32_rays']:.1f}, traffic {r['traffic'] / 1e9:.1f} GB")
print(f"  share0_sha256 {d['share0_sha256'][:16]}, cpu baseline {d['cpu_baseline']['value']:.4f} Mrays/s ({d['cpu_baseline']['kind']}), port {d['cpu_baseline'].get('port_value', 0):.4f}")
for c in d["configs"]:
    rr = c["roofline"]
    vi = (rr.get("valu_issue") or {}).get("frac")
    print(f"{c['id']}: {c['kernel_ms']:.3f} ms, {c['Mrays_per_s']:.0f} Mrays/s, rays {c['rays'] / 1e9:.3f} G, {c['kernel']}, frac {rr['frac'] if rr['frac'] is None else round(rr['frac'], 4)}, "
          f"frac_algorithmic {rr.get('frac_algorithmic', 0):.2f}" + (f", valu_issue {vi:.3f}" if vi else ""))
