"""ctypes mirror of include/raytracer_amd.h (the C-ABI of the gfx950 tracing backend).

Pure declarations: struct layouts, enums and function prototypes.  Used by the product loader
(`raytracer_amd.lib`) and, with a different shared object, by the tests' oracle loader.
"""
import ctypes as C

RT_OK = 0
RT_ERR_INVALID_ARGUMENT = 1
RT_ERR_HIP = 2
RT_ERR_NO_DEVICE = 3
RT_ERR_OUT_OF_MEMORY = 4

RT_SHAPE_SPHERE, RT_SHAPE_PLANE = 0, 1
RT_CSG_UNION, RT_CSG_INTERSECTION, RT_CSG_DIFFERENCE, RT_CSG_FUSION = 0, 1, 2, 3
RT_BRDF_DIFFUSE, RT_BRDF_SPECULAR = 0, 1
RT_PIGMENT_UNIFORM, RT_PIGMENT_CHECKERED, RT_PIGMENT_IMAGE = 0, 1, 2
RT_CAMERA_ORTHOGONAL, RT_CAMERA_PERSPECTIVE = 0, 1
RT_MODE_ONOFF, RT_MODE_FLAT, RT_MODE_POINTLIGHT, RT_MODE_PATH = 0, 1, 2, 3
RT_RNG_PER_NODE_STREAMS, RT_RNG_SERIAL_REFERENCE = 0, 1
RT_FLAG_EXACT_ARITHMETIC = 1
RT_FLAG_OBJECT_SPACE_SPHERES = 2
RT_FLAG_ACCEL_BVH = 4
RT_FLAG_VALU_ONLY = 8
RT_FLAG_SCAN_ALL_TILES = 16
RT_MAX_CSG_DEPTH = 8
RT_MAX_CSG_LEAVES = 32
RT_NOALLOC_SCRATCH_BYTES = 128 * 1024 * 1024

MODES = {"onoff": RT_MODE_ONOFF, "flat": RT_MODE_FLAT, "pointlight": RT_MODE_POINTLIGHT, "path": RT_MODE_PATH}

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("n_shapes", C.c_uint32), ("shape_kind", _u8p), ("shape_material", _u32p),
        ("shape_fwd", _f32p), ("shape_inv", _f32p),
        ("n_csg_nodes", C.c_uint32), ("csg_op", _u8p), ("csg_left", _i32p), ("csg_right", _i32p),
        ("csg_fwd", _f32p), ("csg_inv", _f32p),
        ("n_objects", C.c_uint32), ("objects", _i32p),
        ("n_materials", C.c_uint32), ("material_brdf", _u8p), ("material_brdf_pigment", _u32p),
        ("material_emission_pigment", _u32p),
        ("n_pigments", C.c_uint32), ("pigment_kind", _u8p), ("pigment_color1", _f32p),
        ("pigment_color2", _f32p), ("pigment_n_intervals", _i32p), ("pigment_tex_offset", _u32p),
        ("pigment_tex_width", _i32p), ("pigment_tex_height", _i32p),
        ("n_texels", C.c_uint32), ("texels", _f32p),
        ("n_lights", C.c_uint32), ("light_point", _f32p), ("light_color", _f32p), ("light_radius", _f32p),
    ]


class rt_camera(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("fwd", C.c_float * 12), ("asp_ratio", C.c_float),
                ("has_asp_ratio", C.c_uint32), ("distance", C.c_float)]


class rt_render_cfg(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp_edge", C.c_int32), ("mode", C.c_uint32),
        ("n_rays", C.c_int32), ("roulette", C.c_int32), ("max_depth", C.c_int32),
        ("seq_number", C.c_uint64), ("background", C.c_float * 3), ("ambient", C.c_float * 3),
        ("max_mirror_bounces", C.c_int32), ("rng_mode", C.c_uint32),
        ("band_rows", C.c_int32), ("band_stride", C.c_int32), ("band_offset", C.c_int32),
        ("n_local_rows", C.c_int32), ("flags", C.c_uint32),
    ]


class rt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("capped_samples", C.c_uint64),
                ("kernel_ms", C.c_float), ("grid", C.c_uint32), ("block", C.c_uint32), ("kernel", C.c_char * 48)]


class rt_hit(C.Structure):
    _fields_ = [("object", C.c_int32), ("shape", C.c_int32), ("t", C.c_float), ("point", C.c_float * 3),
                ("normal", C.c_float * 3), ("uv", C.c_float * 2)]


class rt_ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("tmin", C.c_float),
                ("tmax", C.c_float)]


# numpy dtypes with the same layout as rt_ray / rt_hit (both are packed 4-byte fields)
import numpy as _np

RAY_DTYPE = _np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("tmin", "<f4"), ("tmax", "<f4")])
HIT_DTYPE = _np.dtype([("object", "<i4"), ("shape", "<i4"), ("t", "<f4"), ("point", "<f4", 3),
                       ("normal", "<f4", 3), ("uv", "<f4", 2)])
assert RAY_DTYPE.itemsize == C.sizeof(rt_ray) and HIT_DTYPE.itemsize == C.sizeof(rt_hit)

# Every symbol include/raytracer_amd.h declares: name -> (restype, argtypes)
RT_TONEMAP_SERIAL_SUM = 1


class rt_tonemap_cfg(C.Structure):
    _fields_ = [("alpha", C.c_float), ("gamma", C.c_float), ("has_avg_luminosity", C.c_uint32), ("avg_luminosity", C.c_float),
                ("delta", C.c_float), ("flags", C.c_uint32)]


RT_MULTI_PEER_COPIES, RT_MULTI_RCCL_SELF, RT_MULTI_MAX_TIMED_RANKS = 1, 2, 16


class rt_multi_timing(C.Structure):
    _fields_ = [("n_ranks", C.c_int32), ("slowest_render_ms", C.c_float), ("gather_ms", C.c_float),
                ("render_ms", C.c_float * RT_MULTI_MAX_TIMED_RANKS), ("rays", C.c_uint64 * RT_MULTI_MAX_TIMED_RANKS)]


PROTOTYPES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p, C.c_void_p,
                            C.POINTER(rt_stats)]),
    "rt_render_noalloc": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p, C.c_void_p,
                                    C.POINTER(rt_stats)]),
    "rt_render_host": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p,
                                 C.POINTER(rt_stats)]),
    "rt_render_frame_host": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(rt_stats)]),
    "rt_multi_create": (C.c_int, [C.POINTER(rt_scene_desc), C.POINTER(C.c_int), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_multi_rank_count": (C.c_int, [C.c_void_p]),
    "rt_multi_uses_rccl": (C.c_int, [C.c_void_p]),
    "rt_multi_render": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p, C.POINTER(rt_stats),
                                  C.POINTER(rt_multi_timing)]),
    "rt_multi_render_host": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p, C.POINTER(rt_stats),
                                       C.POINTER(rt_multi_timing)]),
    "rt_multi_render_frame_host": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.POINTER(rt_stats), C.POINTER(rt_multi_timing)]),
    "rt_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rt_offset_if_visible": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rt_aa_jitter": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rt_scatter_rays": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rt_average_luminosity": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]),
    "rt_tonemap": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(rt_tonemap_cfg), C.c_void_p, C.c_void_p, C.c_void_p,
                             C.POINTER(C.c_float)]),
    "rt_tonemap_host": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(rt_tonemap_cfg), C.c_void_p, C.POINTER(C.c_float)]),
}


def bind(dll, prototypes=None):
    """Attach restype/argtypes for every prototype; raises AttributeError if a symbol is missing."""
    for name, (res, args) in (prototypes or PROTOTYPES).items():
        fn = getattr(dll, name)
        fn.restype = res
        fn.argtypes = args
    return dll
