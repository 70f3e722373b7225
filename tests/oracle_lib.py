"""Loader + thin wrappers for oracle/librt_oracle.so (the CPU restatement; test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi
from raytracer_amd.scene import FlatScene, Camera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librt_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "rt_ref")

_u64x2 = C.c_uint64 * 2
_f3 = C.c_float * 3

_PROTOS = {
    "rto_render": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.POINTER(abi.rt_camera), C.POINTER(abi.rt_render_cfg),
                             C.c_void_p, C.POINTER(abi.rt_stats)]),
    "rto_trace_rays": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "rto_offset_if_visible": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "rto_object_all_hits": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.c_int32, C.POINTER(abi.rt_ray), C.c_void_p, C.c_int]),
    "rto_object_hit": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.c_int32, C.POINTER(abi.rt_ray), C.POINTER(abi.rt_hit)]),
    "rto_is_point_inside": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.c_int32, _f3]),
    "rto_pcg_init": (None, [_u64x2, C.c_uint64, C.c_uint64]),
    "rto_pcg_random": (C.c_uint32, [_u64x2]),
    "rto_pcg_random_float": (C.c_float, [_u64x2]),
    "rto_pcg_advance": (None, [_u64x2, C.c_uint64]),
    "rto_node_root_stream": (None, [_u64x2, C.c_uint64, C.c_uint64]),
    "rto_node_child_stream": (None, [_u64x2, C.c_uint32]),
    "rto_aa_jitter": (None, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rto_scatter_ray": (None, [C.c_int, _u64x2, _f3, _f3, _f3, C.c_int, C.POINTER(abi.rt_ray)]),
    "rto_pigment": (None, [C.POINTER(abi.rt_scene_desc), C.c_uint32, C.c_float, C.c_float, _f3]),
    "rto_onb": (None, [_f3, _f3, _f3]),
    "rto_camera_ray": (None, [C.POINTER(abi.rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                              C.POINTER(abi.rt_ray)]),
    "rto_shade": (C.c_int, [C.POINTER(abi.rt_scene_desc), C.POINTER(abi.rt_render_cfg), C.POINTER(abi.rt_ray), C.c_int,
                            _u64x2, _f3, C.POINTER(C.c_uint64)]),
    "rto_hom_apply": (None, [C.c_float * 12, C.c_float * 12, C.c_int, _f3, _f3]),
    "rto_average_luminosity": (C.c_float, [C.c_void_p, C.c_size_t, C.c_float]),
    "rto_tonemap": (C.c_float, [C.c_void_p, C.c_size_t, C.POINTER(abi.rt_tonemap_cfg), C.c_void_p, C.c_void_p]),
}

_dll = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "librt_oracle.so"], check=True)


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _dll = abi.bind(C.CDLL(ORACLE_SO), _PROTOS)
    return _dll


def make_cfg(width, height, mode="flat", spp_edge=1, n_rays=10, roulette=3, max_depth=5, seq_number=54,
             background=(0, 0, 0), ambient=(0.1, 0.1, 0.1), max_mirror_bounces=64,
             rng_mode=abi.RT_RNG_PER_NODE_STREAMS, band_rows=None, band_stride=None, band_offset=0,
             n_local_rows=None) -> abi.rt_render_cfg:
    cfg = abi.rt_render_cfg()
    cfg.width, cfg.height, cfg.spp_edge = width, height, spp_edge
    cfg.mode = abi.MODES[mode] if isinstance(mode, str) else mode
    cfg.n_rays, cfg.roulette, cfg.max_depth, cfg.seq_number = n_rays, roulette, max_depth, seq_number
    cfg.background[:] = background
    cfg.ambient[:] = ambient
    cfg.max_mirror_bounces = max_mirror_bounces
    cfg.rng_mode = rng_mode
    cfg.band_rows = height if band_rows is None else band_rows
    cfg.band_stride = cfg.band_rows if band_stride is None else band_stride
    cfg.band_offset = band_offset
    cfg.n_local_rows = height if n_local_rows is None else n_local_rows
    return cfg


def render(scene: FlatScene, cfg: abi.rt_render_cfg, camera: Camera | None = None):
    """-> (image float32 [n_local_rows, W, 3], rt_stats)"""
    cam = (camera or scene.camera).to_abi()
    out = np.zeros((cfg.n_local_rows, cfg.width, 3), dtype=np.float32)
    stats = abi.rt_stats()
    desc = scene.desc()
    rc = dll().rto_render(C.byref(desc), C.byref(cam), C.byref(cfg), out.ctypes.data, C.byref(stats))
    if rc != abi.RT_OK:
        raise RuntimeError(f"rto_render failed: {rc}")
    return out, stats


def trace_rays(scene: FlatScene, rays: np.ndarray) -> np.ndarray:
    rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
    hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
    desc = scene.desc()
    dll().rto_trace_rays(C.byref(desc), rays.ctypes.data, len(rays), hits.ctypes.data)
    return hits


def offset_if_visible(scene: FlatScene, viewer, surface, normal) -> np.ndarray:
    viewer = np.ascontiguousarray(viewer, dtype=np.float32).reshape(-1, 3)
    surface = np.ascontiguousarray(surface, dtype=np.float32).reshape(-1, 3)
    normal = np.ascontiguousarray(normal, dtype=np.float32).reshape(-1, 3)
    vis = np.zeros(len(viewer), dtype=np.uint8)
    desc = scene.desc()
    dll().rto_offset_if_visible(C.byref(desc), viewer.ctypes.data, surface.ctypes.data, normal.ctypes.data,
                                len(viewer), vis.ctypes.data)
    return vis


def mk_ray(origin, direction, tmin=1e-5, tmax=np.inf) -> abi.rt_ray:
    r = abi.rt_ray()
    r.origin[:] = [float(x) for x in origin]
    r.direction[:] = [float(x) for x in direction]
    r.tmin, r.tmax = tmin, tmax
    return r


def object_all_hits(scene: FlatScene, ref: int, ray: abi.rt_ray) -> np.ndarray:
    hits = np.zeros(2 * abi.RT_MAX_CSG_LEAVES, dtype=abi.HIT_DTYPE)
    desc = scene.desc()
    n = dll().rto_object_all_hits(C.byref(desc), ref, C.byref(ray), hits.ctypes.data, len(hits))
    return hits[:n]


def object_hit(scene: FlatScene, ref: int, ray: abi.rt_ray):
    h = abi.rt_hit()
    desc = scene.desc()
    ok = dll().rto_object_hit(C.byref(desc), ref, C.byref(ray), C.byref(h))
    return h if ok else None


def is_point_inside(scene: FlatScene, ref: int, p) -> bool:
    desc = scene.desc()
    return bool(dll().rto_is_point_inside(C.byref(desc), ref, _f3(*[float(x) for x in p])))


class PCG:
    """PCG (reference include/random.hpp) through the oracle."""

    def __init__(self, init_state=42, init_seq=54):
        self.st = _u64x2()
        dll().rto_pcg_init(self.st, init_state, init_seq)

    @property
    def state(self):
        return self.st[0]

    @property
    def inc(self):
        return self.st[1]

    def random(self):
        return dll().rto_pcg_random(self.st)

    def random_float(self):
        return dll().rto_pcg_random_float(self.st)

    def advance(self, n):
        dll().rto_pcg_advance(self.st, n)


def aa_jitter(k: np.ndarray):
    k = np.ascontiguousarray(k, dtype=np.uint64)
    u = np.zeros(len(k), dtype=np.float32)
    v = np.zeros(len(k), dtype=np.float32)
    dll().rto_aa_jitter(k.ctypes.data, len(k), u.ctypes.data, v.ctypes.data)
    return u, v


def shade(scene: FlatScene, cfg: abi.rt_render_cfg, ray: abi.rt_ray, depth=0, pcg: PCG | None = None):
    pcg = pcg or PCG(42, cfg.seq_number)
    out = _f3()
    rays = C.c_uint64()
    desc = scene.desc()
    dll().rto_shade(C.byref(desc), C.byref(cfg), C.byref(ray), depth, pcg.st, out, C.byref(rays))
    return np.array(out[:], dtype=np.float32), rays.value


def camera_ray(camera: Camera, width, height, col, row, u_pixel=0.5, v_pixel=0.5) -> abi.rt_ray:
    cam = camera.to_abi()
    r = abi.rt_ray()
    dll().rto_camera_ray(C.byref(cam), width, height, col, row, u_pixel, v_pixel, C.byref(r))
    return r


def scatter_ray(brdf_kind, pcg: PCG, in_dir, point, normal, depth) -> abi.rt_ray:
    r = abi.rt_ray()
    f = lambda v: _f3(*[float(x) for x in v])
    dll().rto_scatter_ray(brdf_kind, pcg.st, f(in_dir), f(point), f(normal), depth, C.byref(r))
    return r


def pigment(scene: FlatScene, idx, u, v):
    out = _f3()
    desc = scene.desc()
    dll().rto_pigment(C.byref(desc), idx, u, v, out)
    return np.array(out[:], dtype=np.float32)


def onb(e3):
    e1, e2 = _f3(), _f3()
    dll().rto_onb(_f3(*[float(x) for x in e3]), e1, e2)
    return np.array(e1[:], dtype=np.float32), np.array(e2[:], dtype=np.float32)


def hom_apply(fwd, inv, what, vec):
    out = _f3()
    m = (C.c_float * 12)(*[float(x) for x in fwd])
    mi = (C.c_float * 12)(*[float(x) for x in inv])
    dll().rto_hom_apply(m, mi, {"vec": 0, "point": 1, "normal": 2}[what], _f3(*[float(x) for x in vec]), out)
    return np.array(out[:], dtype=np.float32)


def read_pfm(path) -> np.ndarray:
    """PFM -> float32 [H, W, 3], row 0 = top (HdrImage::pixels order; reference include/colors.hpp:328-376)."""
    with open(path, "rb") as f:
        data = f.read()
    magic, size, endian, body = data.split(b"\n", 3)
    assert magic == b"PF"
    w, h = (int(x) for x in size.split())
    dt = "<f4" if float(endian) < 0 else ">f4"
    a = np.frombuffer(body, dtype=dt, count=w * h * 3).reshape(h, w, 3)
    return np.ascontiguousarray(a[::-1]).astype(np.float32)


def average_luminosity(image: np.ndarray, delta=1e-10) -> float:
    img = np.ascontiguousarray(image, dtype=np.float32)
    return dll().rto_average_luminosity(img.ctypes.data, img.size // 3, delta)


def tonemap(image: np.ndarray, alpha=0.18, gamma=2.2, avg_luminosity=None, delta=1e-10):
    """-> (ldr uint8, hdr float32 (normalised + clamped), average luminosity used)"""
    img = np.ascontiguousarray(image, dtype=np.float32)
    cfg = abi.rt_tonemap_cfg()
    cfg.alpha, cfg.gamma, cfg.delta = alpha, gamma, delta
    cfg.has_avg_luminosity = 0 if avg_luminosity is None else 1
    cfg.avg_luminosity = 0.0 if avg_luminosity is None else float(avg_luminosity)
    ldr = np.empty(img.shape, dtype=np.uint8)
    hdr = np.empty(img.shape, dtype=np.float32)
    avg = dll().rto_tonemap(img.ctypes.data, img.size // 3, C.byref(cfg), hdr.ctypes.data, ldr.ctypes.data)
    return ldr, hdr, avg
