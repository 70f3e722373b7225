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
