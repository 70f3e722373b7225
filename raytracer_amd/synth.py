"""Synthetic N-random-sphere benchmark scene (SURVEY.md §8d / BASELINE.json configs[4]).

The scene is produced twice from the same binary32 numbers:
  * as scene-DSL text (the reference's own input format, EBNF.md), so that the unmodified reference /
    its parser read the very same scene, and
  * as a FlatScene, flattened exactly like raytracer_amd/host/flatten.hpp flattens the parsed World
    (material ids in order of first use by a shape, two pigments per material).
tests/test_synth.py checks the two routes give identical arrays (through oracle/_ref/rt_ref where the
reference exists, and against a committed fixture elsewhere).

Layout: floor plane z=-2 (checkered, as samples/demo_path_tracing.txt), emissive sky sphere r=50, then
N spheres with centres uniform in x[-2,8] y[-6,6] z[-2+r,2], r uniform in [0.05,0.15], materials from a
palette of 16 (12 diffuse, 3 specular, 1 emissive).  All draws: random_float of PCG(2026, N)
(reference include/random.hpp), in the order written below.  Camera: perspective at (-3,0,0).
"""
from __future__ import annotations

import numpy as np

from .scene import SceneBuilder, compose, scaling, translation

_f32 = np.float32
_MULT = 6364136223846793005
_MASK = (1 << 64) - 1


class _PCG:
    """PCG32 XSH-RR exactly as reference include/random.hpp:41-70 (host-side scene generation only)."""

    def __init__(self, init_state=42, init_seq=54):
        self.state = 0
        self.inc = ((init_seq << 1) | 1) & _MASK
        self.random()
        self.state = (self.state + init_state) & _MASK
        self.random()

    def random(self) -> int:
        old = self.state
        self.state = (old * _MULT + self.inc) & _MASK
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xFFFFFFFF

    def random_float(self) -> np.float32:
        return _f32(_f32(self.random()) * _f32(2.0 ** -32))


def _num(x) -> str:
    """Shortest decimal that stof() reads back to exactly this binary32, no exponent (the reference lexer
    accepts digits, '.', 'e', 'E' but not a sign inside a number: include/scenefiles.hpp:415-426)."""
    return np.format_float_positional(_f32(x), unique=True, trim="0")


def _lerp(lo, hi, u) -> np.float32:
    return _f32(_f32(lo) + _f32(_f32(_f32(hi) - _f32(lo)) * u))


def synthetic_scene(n_spheres: int):
    """-> (FlatScene with camera, scene-DSL text)"""
    pcg = _PCG(2026, n_spheres)
    rf = pcg.random_float
    palette = []  # (name, brdf, brdf colour, emission colour)
    for k in range(12):
        c = tuple(_lerp(0.1, 0.9, rf()) for _ in range(3))
        palette.append((f"diffuse_{k}", "diffuse", c, (_f32(0), _f32(0), _f32(0))))
    for k in range(3):
        g = _lerp(0.5, 0.9, rf())
        palette.append((f"mirror_{k}", "specular", (g, g, g), (_f32(0), _f32(0), _f32(0))))
    palette.append(("lamp", "diffuse", (_f32(0), _f32(0), _f32(0)), (_f32(1.0), _f32(0.9), _f32(0.7))))

    b = SceneBuilder()
    lines = ["# synthetic benchmark scene: %d random spheres + floor + sky (raytracer_amd/synth.py)" % n_spheres, ""]

    def color(c):
        return "<%s, %s, %s>" % tuple(_num(x) for x in c)

    def vec(v):
        return "[%s, %s, %s]" % tuple(_num(x) for x in v)

    lines.append("material floor_material(diffuse(checkered(<0.3, 0.5, 0.1>, <0.1, 0.2, 0.5>, 4)), uniform(<0, 0, 0>))")
    lines.append("material sky_material(diffuse(uniform(<0, 0, 0>)), uniform(<0.2, 0.3, 1>))")
    for name, brdf, c, e in palette:
        lines.append(f"material {name}({brdf}(uniform({color(c)})), uniform({color(e)}))")
    lines.append("")

    floor = b.material("diffuse", b.pigment_checkered((_f32(0.3), _f32(0.5), _f32(0.1)), (_f32(0.1), _f32(0.2), _f32(0.5)), 4),
                       b.pigment_uniform((0, 0, 0)))
    b.plane(compose(translation([0, 0, -2])), floor)
    lines.append("plane(translation([0, 0, -2]), floor_material)")
    sky = b.material("diffuse", b.pigment_uniform((0, 0, 0)), b.pigment_uniform((_f32(0.2), _f32(0.3), _f32(1))))
    b.sphere(compose(scaling([50, 50, 50])), sky)
    lines.append("sphere(scaling([50, 50, 50]), sky_material)")

    material_ids = {}  # palette index -> material id, assigned at first use (what the C++ flattener does)
    for _ in range(n_spheres):
        r = _lerp(0.05, 0.15, rf())
        x = _lerp(-2.0, 8.0, rf())
        y = _lerp(-6.0, 6.0, rf())
        z = _lerp(_f32(_f32(-2.0) + r), 2.0, rf())
        k = min(int(_f32(rf() * _f32(16.0))), 15)
        if k not in material_ids:
            name, brdf, c, e = palette[k]
            material_ids[k] = b.material(brdf, b.pigment_uniform(c), b.pigment_uniform(e))
        b.sphere(compose(translation([x, y, z]), scaling([r, r, r])), material_ids[k])
        lines.append(f"sphere(translation({vec((x, y, z))}) * scaling({vec((r, r, r))}), {palette[k][0]})")
    lines.append("")
    lines.append("camera(perspective, translation([-3, 0, 0]), exact_asp_ratio, 1)")
    b.set_camera("perspective", compose(translation([-3, 0, 0])), None, 1.0)
    return b.build(), "\n".join(lines) + "\n"
