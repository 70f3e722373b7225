"""Flattened scene container + a small builder that mirrors the reference's scene vocabulary.

`FlatScene` holds exactly the struct-of-arrays that `rt_scene_desc` (include/raytracer_amd.h) takes.
It is filled either from a JSON dump written by the C++ flattener (raytracer_amd/host/flatten.hpp,
which walks the reference's parsed `World`), or through `SceneBuilder`, whose transformations
compose in binary32 with the reference's operation order so that both routes give identical bits:

  Transformation::operator* / HomMatrix::compose   reference include/geometry.hpp:403-426, 509-513
  translation / scaling / rotation_{x,y,z}          include/geometry.hpp:520-562
  degs_to_rads                                       include/geometry.hpp:618
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field

import numpy as np

from . import abi

_f32 = np.float32
_libm = C.CDLL("libm.so.6")
_libm.cosf.restype = C.c_float
_libm.cosf.argtypes = [C.c_float]
_libm.sinf.restype = C.c_float
_libm.sinf.argtypes = [C.c_float]
PI_F = _f32(3.14159265358979323846)


class Transformation:
    """hom_matrix + inverse_hom_matrix, 12 binary32 each: 3x3 row-major linear part then translation."""

    __slots__ = ("fwd", "inv")

    def __init__(self, fwd=None, inv=None):
        ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=_f32)
        self.fwd = ident.copy() if fwd is None else np.asarray(fwd, dtype=_f32).copy()
        self.inv = ident.copy() if inv is None else np.asarray(inv, dtype=_f32).copy()

    @staticmethod
    def _compose(a, b):
        """HomMatrix::compose: T_A * T_B = [A.B, A.t_B + t_A] with the reference's left-to-right sums."""
        out = np.empty(12, dtype=_f32)
        for i in range(3):
            for j in range(3):
                out[3 * i + j] = _f32(_f32(a[3 * i] * b[j]) + _f32(a[3 * i + 1] * b[3 + j])) + _f32(a[3 * i + 2] * b[6 + j])
        for i in range(3):
            applied = _f32(_f32(a[3 * i] * b[9]) + _f32(a[3 * i + 1] * b[10])) + _f32(a[3 * i + 2] * b[11])
            out[9 + i] = _f32(applied + a[9 + i])
        return out

    def __mul__(self, other: "Transformation") -> "Transformation":
        return Transformation(self._compose(self.fwd, other.fwd), self._compose(other.inv, self.inv))


def identity() -> Transformation:
    return Transformation()


def compose(*parts: Transformation) -> Transformation:
    """`a * b * c` as the scene parser evaluates it: starts from the identity and multiplies left to
    right (reference include/scenefiles.hpp:731-785), which also fixes the signs of zeros."""
    result = Transformation()
    for p in parts:
        result = result * p
    return result


def translation(vec) -> Transformation:
    v = np.asarray(vec, dtype=_f32)
    t = Transformation()
    t.fwd[9:12] = v
    t.inv[9:12] = -v
    return t


def scaling(diag) -> Transformation:
    d = np.asarray(diag, dtype=_f32)
    t = Transformation()
    for i in range(3):
        t.fwd[4 * i] = d[i]
        t.inv[4 * i] = _f32(1.0) / d[i]
    return t


def degs_to_rads(angle_degs) -> np.float32:
    return _f32(_f32(_f32(angle_degs) * PI_F) / _f32(180.0))


def _rotation(rows) -> Transformation:
    m = np.array(rows, dtype=_f32)
    t = Transformation()
    t.fwd[:9] = m.reshape(9)
    t.inv[:9] = m.T.reshape(9)
    return t


def rotation_x(theta_rad) -> Transformation:
    c, s = _f32(_libm.cosf(float(theta_rad))), _f32(_libm.sinf(float(theta_rad)))
    return _rotation([[1, 0, 0], [0, c, -s], [0, s, c]])


def rotation_y(theta_rad) -> Transformation:
    c, s = _f32(_libm.cosf(float(theta_rad))), _f32(_libm.sinf(float(theta_rad)))
    return _rotation([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotation_z(theta_rad) -> Transformation:
    c, s = _f32(_libm.cosf(float(theta_rad))), _f32(_libm.sinf(float(theta_rad)))
    return _rotation([[c, -s, 0], [s, c, 0], [0, 0, 1]])


@dataclass
class Camera:
    """Camera (reference include/cameras.hpp:73-160) as rt_camera."""
    kind: int = abi.RT_CAMERA_PERSPECTIVE
    transformation: Transformation = field(default_factory=Transformation)
    asp_ratio: float | None = None
    distance: float = 1.0

    def to_abi(self) -> abi.rt_camera:
        c = abi.rt_camera()
        c.kind = self.kind
        c.fwd[:] = [float(x) for x in self.transformation.fwd]
        c.has_asp_ratio = 0 if self.asp_ratio is None else 1
        c.asp_ratio = 0.0 if self.asp_ratio is None else float(self.asp_ratio)
        c.distance = float(self.distance)
        return c


_FIELDS = [
    # name, dtype, items per record
    ("shape_kind", np.uint8, 1), ("shape_material", np.uint32, 1), ("shape_fwd", _f32, 12), ("shape_inv", _f32, 12),
    ("csg_op", np.uint8, 1), ("csg_left", np.int32, 1), ("csg_right", np.int32, 1), ("csg_fwd", _f32, 12),
    ("csg_inv", _f32, 12),
    ("objects", np.int32, 1),
    ("material_brdf", np.uint8, 1), ("material_brdf_pigment", np.uint32, 1), ("material_emission_pigment", np.uint32, 1),
    ("pigment_kind", np.uint8, 1), ("pigment_color1", _f32, 3), ("pigment_color2", _f32, 3),
    ("pigment_n_intervals", np.int32, 1), ("pigment_tex_offset", np.uint32, 1), ("pigment_tex_width", np.int32, 1),
    ("pigment_tex_height", np.int32, 1),
    ("texels", _f32, 3),
    ("light_point", _f32, 3), ("light_color", _f32, 3), ("light_radius", _f32, 1),
]
_COUNTS = {
    "n_shapes": "shape_kind", "n_csg_nodes": "csg_op", "n_objects": "objects", "n_materials": "material_brdf",
    "n_pigments": "pigment_kind", "n_texels": "texels", "n_lights": "light_radius",
}
_CTYPE = {np.uint8: C.c_uint8, np.uint32: C.c_uint32, np.int32: C.c_int32, _f32: C.c_float}


class FlatScene:
    """The arrays behind an rt_scene_desc (flat, C-contiguous numpy) and, optionally, the camera."""

    def __init__(self, arrays: dict | None = None, camera: Camera | None = None):
        arrays = arrays or {}
        for name, dtype, _ in _FIELDS:
            setattr(self, name, np.ascontiguousarray(np.asarray(arrays.get(name, []), dtype=dtype).reshape(-1)))
        self.camera = camera

    # -- construction --------------------------------------------------------------------------
    @classmethod
    def from_json(cls, path) -> "FlatScene":
        with open(path) as f:
            d = json.load(f)
        cam = None
        if "camera" in d:
            c = d["camera"]
            t = Transformation(fwd=np.array(c["fwd"], dtype=_f32))
            cam = Camera(kind=int(c["kind"]), transformation=t,
                         asp_ratio=(_f32(c["asp_ratio"]) if c["has_asp_ratio"] else None),
                         distance=_f32(c["distance"]))
        return cls({k: v for k, v in d.items() if k != "camera"}, cam)

    def to_json(self, path) -> None:
        d = {}
        for name, dtype, _ in _FIELDS:
            a = getattr(self, name)
            d[name] = [float(np.format_float_scientific(x, unique=True)) for x in a] if dtype is _f32 else [int(x) for x in a]
        if self.camera is not None:
            c = self.camera
            d["camera"] = {"kind": c.kind, "has_asp_ratio": 0 if c.asp_ratio is None else 1,
                           "asp_ratio": 0.0 if c.asp_ratio is None else float(c.asp_ratio),
                           "distance": float(c.distance), "fwd": [float(x) for x in c.transformation.fwd]}
        with open(path, "w") as f:
            json.dump(d, f)

    # -- sizes ---------------------------------------------------------------------------------
    def count(self, what: str) -> int:
        name = _COUNTS[what]
        per = next(n for f, _, n in _FIELDS if f == name)
        return len(getattr(self, name)) // per

    # -- C view --------------------------------------------------------------------------------
    def desc(self) -> abi.rt_scene_desc:
        """rt_scene_desc whose pointers alias this object's arrays (keep `self` alive while in use)."""
        d = abi.rt_scene_desc()
        for name, dtype, _ in _FIELDS:
            a = getattr(self, name)
            setattr(d, name, a.ctypes.data_as(C.POINTER(_CTYPE[dtype])))
        for cnt in _COUNTS:
            setattr(d, cnt, self.count(cnt))
        return d

    def equals(self, other: "FlatScene") -> bool:
        return all(getattr(self, n).tobytes() == getattr(other, n).tobytes() for n, _, _ in _FIELDS)


class SceneBuilder:
    """Builds a FlatScene with the reference's vocabulary: materials, spheres, planes, CSG, lights.

    Mirrors what Scene::parse_scene does with a scene file (reference include/scenefiles.hpp:934-1043):
    `add_*` appends to World::objects in call order; `norender=True` returns the reference without
    adding it to the world (scenefiles.hpp:1055-1065), for use as a CSG child.
    """

    def __init__(self):
        self._a = {name: [] for name, _, _ in _FIELDS}
        self.camera: Camera | None = None

    def pigment_uniform(self, color):
        return ("uniform", tuple(color))

    def pigment_checkered(self, c1, c2, n_intervals):
        return ("checkered", tuple(c1), tuple(c2), int(n_intervals))

    def pigment_image(self, rgb):
        """rgb: (H, W, 3) float32, row 0 = top (HdrImage::pixels order)."""
        return ("image", np.ascontiguousarray(rgb, dtype=_f32))

    def _add_pigment(self, p) -> int:
        a = self._a
        idx = len(a["pigment_kind"])
        c1, c2, n, off, w, h = (0, 0, 0), (0, 0, 0), 0, 0, 0, 0
        if p[0] == "uniform":
            kind, c1 = abi.RT_PIGMENT_UNIFORM, p[1]
        elif p[0] == "checkered":
            kind, c1, c2, n = abi.RT_PIGMENT_CHECKERED, p[1], p[2], p[3]
        elif p[0] == "image":
            kind = abi.RT_PIGMENT_IMAGE
            img = p[1]
            h, w = img.shape[:2]
            off = len(a["texels"]) // 3
            a["texels"].extend(img.reshape(-1).tolist())
        else:
            raise ValueError(f"unknown pigment {p[0]!r}")
        a["pigment_kind"].append(kind)
        a["pigment_color1"].extend(c1)
        a["pigment_color2"].extend(c2)
        a["pigment_n_intervals"].append(n)
        a["pigment_tex_offset"].append(off)
        a["pigment_tex_width"].append(w)
        a["pigment_tex_height"].append(h)
        return idx

    def material(self, brdf: str, brdf_pigment, emitted_pigment=None) -> int:
        """brdf: 'diffuse' | 'specular'.  Returns the material index."""
        kinds = {"diffuse": abi.RT_BRDF_DIFFUSE, "specular": abi.RT_BRDF_SPECULAR}
        a = self._a
        idx = len(a["material_brdf"])
        a["material_brdf"].append(kinds[brdf])
        a["material_brdf_pigment"].append(self._add_pigment(brdf_pigment))
        a["material_emission_pigment"].append(self._add_pigment(emitted_pigment or ("uniform", (0, 0, 0))))
        return idx

    def _shape(self, kind, t: Transformation, material: int, norender: bool) -> int:
        a = self._a
        idx = len(a["shape_kind"])
        a["shape_kind"].append(kind)
        a["shape_material"].append(material)
        a["shape_fwd"].extend(t.fwd.tolist())
        a["shape_inv"].extend(t.inv.tolist())
        if not norender:
            a["objects"].append(idx)
        return idx

    def sphere(self, t: Transformation, material: int, norender=False) -> int:
        return self._shape(abi.RT_SHAPE_SPHERE, t, material, norender)

    def plane(self, t: Transformation, material: int, norender=False) -> int:
        return self._shape(abi.RT_SHAPE_PLANE, t, material, norender)

    def csg(self, left: int, right: int, op: str, t: Transformation | None = None, norender=False) -> int:
        ops = {"union": abi.RT_CSG_UNION, "intersection": abi.RT_CSG_INTERSECTION,
               "difference": abi.RT_CSG_DIFFERENCE, "fusion": abi.RT_CSG_FUSION}
        t = t or Transformation()
        a = self._a
        # a norender object is MOVED into the CSG node that names it (scenefiles.hpp:1055-1065): one owner per object
        for child in (left, right):
            if child in a["objects"] or child in a["csg_left"] or child in a["csg_right"] or left == right:
                raise ValueError("csg(): a child must be a norender object that no other CSG node uses (one owner per object)")
        node = len(a["csg_op"])
        a["csg_op"].append(ops[op])
        a["csg_left"].append(left)
        a["csg_right"].append(right)
        a["csg_fwd"].extend(t.fwd.tolist())
        a["csg_inv"].extend(t.inv.tolist())
        ref = ~node
        if not norender:
            a["objects"].append(ref)
        return ref

    def point_light(self, point, color, radius=0.0):
        a = self._a
        a["light_point"].extend(point)
        a["light_color"].extend(color)
        a["light_radius"].append(radius)

    def set_camera(self, kind: str, t: Transformation, asp_ratio=None, distance=1.0):
        k = {"perspective": abi.RT_CAMERA_PERSPECTIVE, "orthogonal": abi.RT_CAMERA_ORTHOGONAL}[kind]
        self.camera = Camera(k, t, asp_ratio, distance)

    def build(self) -> FlatScene:
        return FlatScene(self._a, self.camera)
