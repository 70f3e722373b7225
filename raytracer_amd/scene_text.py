"""FlatScene -> scene-DSL text (the reference's input language, EBNF.md), for worlds whose transformations are
`translation(t) * scaling(d)` (diagonal linear part): every sample scene of the reference and the synthetic benchmark
scenes.  Used to hand a flattened fixture to the C++ route (reference parser -> flattener -> C-ABI), e.g. by
scripts/cpp_end_to_end.py on the GPU box, where the reference's own sample files do not exist.

The emitted text describes the same objects in the same order; parsing it rebuilds the same matrices up to the signs
of zeros, a rotation about z (the orbiting camera of the samples) up to its last bits (it is meant for timing and smoke
runs, not as a parity fixture).  Other rotations, image pigments: ValueError.
"""
from __future__ import annotations

import numpy as np

from . import abi
from .scene import FlatScene

_f32 = np.float32
_OPS = {abi.RT_CSG_UNION: "union", abi.RT_CSG_INTERSECTION: "intersection", abi.RT_CSG_DIFFERENCE: "difference", abi.RT_CSG_FUSION: "fusion"}


def _num(x) -> str:
    return np.format_float_positional(_f32(x), unique=True, trim="0")


def _vec(v) -> str:
    return "[%s, %s, %s]" % tuple(_num(x) for x in v)


def _color(c) -> str:
    return "<%s, %s, %s>" % tuple(_num(x) for x in c)


def _transformation(fwd) -> str:
    m = np.asarray(fwd, dtype=_f32)
    lin, t = m[:9].reshape(3, 3), m[9:12]
    if np.any(lin[~np.eye(3, dtype=bool)] != 0):
        c, sn = float(lin[0, 0]), float(lin[1, 0])
        rot_z = np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1]])
        if abs(c * c + sn * sn - 1) < 1e-5 and np.allclose(lin, rot_z, atol=1e-6):  # rotation_z(a) * translation(v): a camera orbit
            v = rot_z.T @ t.astype(np.float64)
            head = f"rotation_z({_num(np.degrees(np.arctan2(sn, c)))})"
            return head + (f" * translation({_vec(v)})" if np.any(t != 0) else "")
        raise ValueError("only translation * scaling (or rotation_z * translation) transformations can be written back as scene text")
    parts = []
    if np.any(t != 0):
        parts.append(f"translation({_vec(t)})")
    if np.any(np.diag(lin) != 1):
        parts.append(f"scaling({_vec(np.diag(lin))})")
    return " * ".join(parts) if parts else "identity"


def to_scene_text(scene: FlatScene) -> str:
    lines = ["# written by raytracer_amd/scene_text.py from a flattened World", ""]

    def pigment(pg):
        kind = int(scene.pigment_kind[pg])
        if kind == abi.RT_PIGMENT_UNIFORM:
            return f"uniform({_color(scene.pigment_color1[3 * pg: 3 * pg + 3])})"
        if kind == abi.RT_PIGMENT_CHECKERED:
            return (f"checkered({_color(scene.pigment_color1[3 * pg: 3 * pg + 3])}, {_color(scene.pigment_color2[3 * pg: 3 * pg + 3])}, "
                    f"{int(scene.pigment_n_intervals[pg])})")
        raise ValueError("image pigments need their PFM file: not written back")

    for m in range(scene.count("n_materials")):
        brdf = "specular" if scene.material_brdf[m] == abi.RT_BRDF_SPECULAR else "diffuse"
        lines.append(f"material m{m}({brdf}({pigment(int(scene.material_brdf_pigment[m]))}), {pigment(int(scene.material_emission_pigment[m]))})")
    lines.append("")
    top = set(int(r) for r in scene.objects)

    def shape(i, name=None):
        kind = "sphere" if scene.shape_kind[i] == abi.RT_SHAPE_SPHERE else "plane"
        t = _transformation(scene.shape_fwd[12 * i: 12 * i + 12])
        head = f"norender {kind} {name}" if name else kind
        return f"{head}({t}, m{int(scene.shape_material[i])})"

    def name_of(ref):
        return f"s{ref}" if ref >= 0 else f"c{~ref}"

    # CSG children first (norender, named), in node order: children precede their parents
    used = set()
    for node in range(scene.count("n_csg_nodes")):
        for ref in (int(scene.csg_left[node]), int(scene.csg_right[node])):
            if ref >= 0 and ref not in used:
                lines.append(shape(ref, name_of(ref)))
                used.add(ref)
        if ~node not in top:
            lines.append(f"norender csg {name_of(~node)}({name_of(int(scene.csg_left[node]))}, {name_of(int(scene.csg_right[node]))}, "
                         f"{_OPS[int(scene.csg_op[node])]}, {_transformation(scene.csg_fwd[12 * node: 12 * node + 12])})")
    for ref in scene.objects:  # World::objects in order (it decides ties)
        ref = int(ref)
        if ref >= 0:
            lines.append(shape(ref))
        else:
            node = ~ref
            lines.append(f"csg({name_of(int(scene.csg_left[node]))}, {name_of(int(scene.csg_right[node]))}, {_OPS[int(scene.csg_op[node])]}, "
                         f"{_transformation(scene.csg_fwd[12 * node: 12 * node + 12])})")
    for k in range(scene.count("n_lights")):
        lines.append(f"point_light({_vec(scene.light_point[3 * k: 3 * k + 3])}, {_color(scene.light_color[3 * k: 3 * k + 3])}, {_num(scene.light_radius[k])})")
    cam = scene.camera
    if cam is not None:
        asp = "exact_asp_ratio" if cam.asp_ratio is None else _num(cam.asp_ratio)
        t = _transformation(cam.transformation.fwd)
        if cam.kind == abi.RT_CAMERA_PERSPECTIVE:
            lines.append(f"camera(perspective, {t}, {asp}, {_num(cam.distance)})")
        else:
            lines.append(f"camera(orthogonal, {t}, {asp})")
    return "\n".join(lines) + "\n"
