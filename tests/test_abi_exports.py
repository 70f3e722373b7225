"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU and exports every
symbol include/raytracer_amd.h declares; struct layouts agree between the header and the ctypes mirror."""
import ctypes as C
import os
import re
import subprocess

import pytest

from raytracer_amd import abi
from raytracer_amd.lib import LIB_PATH, dll

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raytracer_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"RT_API\s+[\w\s\*]+?\b(rt_\w+)\s*\(", text)))


def test_header_declares_what_the_mirror_binds():
    assert declared_symbols() == sorted(abi.PROTOTYPES)


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(LIB_PATH), "build with __graft_entry__.build()"
    lib = dll()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.rt_last_error() == b""


def test_invalid_arguments_are_rejected_without_a_gpu():
    lib = dll()
    assert lib.rt_scene_create(None, 0, None) == abi.RT_ERR_INVALID_ARGUMENT
    assert b"NULL" in lib.rt_last_error()


def test_struct_layouts_match_the_c_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    src = tmp_path / "layout.c"
    fields = {
        "rt_scene_desc": ["n_shapes", "shape_inv", "n_csg_nodes", "csg_inv", "objects", "material_emission_pigment",
                          "pigment_tex_height", "texels", "light_radius"],
        "rt_camera": ["kind", "fwd", "asp_ratio", "has_asp_ratio", "distance"],
        "rt_render_cfg": ["width", "mode", "seq_number", "background", "ambient", "max_mirror_bounces", "rng_mode",
                          "band_rows", "n_local_rows", "flags"],
        "rt_stats": ["rays", "samples", "capped_samples", "kernel_ms", "grid", "block", "kernel"],
        "rt_hit": ["object", "shape", "t", "point", "normal", "uv"],
        "rt_ray": ["origin", "direction", "tmin", "tmax"],
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "raytracer_amd.h"', 'int main(void){']
    for st, fs in fields.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            lines.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    lines.append('return 0;}')
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.splitlines():
        name, val = line.split()
        if "." in name:
            st, f = name.split(".")
            assert getattr(getattr(abi, st), f).offset == int(val), name
        else:
            assert C.sizeof(getattr(abi, name)) == int(val), name


def _create_status(scene):
    import ctypes as C
    handle = C.c_void_p()
    desc = scene.desc()
    st = dll().rt_scene_create(C.byref(desc), 0, C.byref(handle))
    return st, dll().rt_last_error().decode()


def test_scene_must_be_a_forest_and_within_csg_limits_checked_before_any_device_call():
    """rt_scene_create validates on the host first, so these refusals need no GPU.  A World is a forest (every Object has
    one owner: reference include/shapes.hpp:384-385,525): shared shapes / nodes are refused — the per-shape tables with the
    CSG ancestors folded in rely on it — and so is a 40-level DAG (node i = ~(i-1) op ~(i-1)), which would otherwise be
    flattened into 2^40 program entries, and a chain deeper than the evaluator's limit, without recursing into either."""
    import numpy as np
    from raytracer_amd.scene import SceneBuilder, Transformation

    def base():
        b = SceneBuilder()
        m = b.material("diffuse", b.pigment_uniform((1, 1, 1)))
        return b, m

    # a leaf shared by two CSG objects (bypassing SceneBuilder's own check, as a C caller could)
    b, m = base()
    s0, s1, s2 = (b.sphere(Transformation(), m, norender=True) for _ in range(3))
    b.csg(s0, s1, "union")
    b.csg(s2, b.sphere(Transformation(), m, norender=True), "difference")
    scene = b.build()
    scene.csg_right = np.ascontiguousarray(np.array([s1, s1], dtype=np.int32))
    st, msg = _create_status(scene)
    assert st == abi.RT_ERR_INVALID_ARGUMENT and "more than once" in msg
    # a shape that is both a top-level object and a CSG leaf
    b, m = base()
    top = b.sphere(Transformation(), m)
    b.csg(b.sphere(Transformation(), m, norender=True), b.sphere(Transformation(), m, norender=True), "union")
    scene = b.build()
    scene.csg_left = np.ascontiguousarray(np.array([top], dtype=np.int32))
    st, msg = _create_status(scene)
    assert st == abi.RT_ERR_INVALID_ARGUMENT and "more than once" in msg
    # DAG: node i uses node i-1 on both sides
    b, m = base()
    b.csg(b.sphere(Transformation(), m, norender=True), b.sphere(Transformation(), m, norender=True), "union", norender=True)
    scene = b.build()
    n = 40
    scene.csg_op = np.zeros(n, dtype=np.uint8)
    scene.csg_left = np.ascontiguousarray(np.array([0] + [~(i - 1) for i in range(1, n)], dtype=np.int32))
    scene.csg_right = np.ascontiguousarray(np.array([1] + [~(i - 1) for i in range(1, n)], dtype=np.int32))
    scene.csg_fwd = np.ascontiguousarray(np.tile(Transformation().fwd, n))
    scene.csg_inv = np.ascontiguousarray(np.tile(Transformation().inv, n))
    scene.objects = np.ascontiguousarray(np.array([~(n - 1)], dtype=np.int32))
    st, msg = _create_status(scene)
    assert st == abi.RT_ERR_INVALID_ARGUMENT and "more than once" in msg
    # a chain of 50 000 nodes (a tree, but far deeper than RT_MAX_CSG_DEPTH): refused without recursion
    n = 50_000
    b, m = base()
    scene = b.build()
    scene.shape_kind = np.zeros(n + 1, dtype=np.uint8)
    scene.shape_material = np.zeros(n + 1, dtype=np.uint32)
    scene.shape_fwd = np.ascontiguousarray(np.tile(Transformation().fwd, n + 1))
    scene.shape_inv = np.ascontiguousarray(np.tile(Transformation().inv, n + 1))
    scene.csg_op = np.zeros(n, dtype=np.uint8)
    scene.csg_left = np.ascontiguousarray(np.array([0] + [~(i - 1) for i in range(1, n)], dtype=np.int32))
    scene.csg_right = np.ascontiguousarray(np.arange(1, n + 1, dtype=np.int32))
    scene.csg_fwd = np.ascontiguousarray(np.tile(Transformation().fwd, n))
    scene.csg_inv = np.ascontiguousarray(np.tile(Transformation().inv, n))
    scene.objects = np.ascontiguousarray(np.array([~(n - 1)], dtype=np.int32))
    st, msg = _create_status(scene)
    assert st == abi.RT_ERR_INVALID_ARGUMENT and "RT_MAX_CSG_DEPTH" in msg
    # SceneBuilder refuses the same things early
    b, m = base()
    s = b.sphere(Transformation(), m, norender=True)
    b.csg(s, b.sphere(Transformation(), m, norender=True), "union")
    with pytest.raises(ValueError):
        b.csg(s, b.sphere(Transformation(), m, norender=True), "union")
