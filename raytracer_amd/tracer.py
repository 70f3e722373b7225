"""Host-side mirror of the reference's ImageTracer / Renderer interface over the C-ABI.

    reference (C++)                                     here
    -----------------------------------------------     ------------------------------------------
    World                       include/shapes.hpp:520   DeviceWorld(FlatScene, device)
    ImageTracer{image,camera,spp}   cameras.hpp:166-187  GpuImageTracer(width, height, camera, spp_edge)
    build_renderer(mode_cfg, world) raytracer.cpp:333    RenderMode("path", n_rays=..., ...)
    tracer.fire_all_rays(renderer)  cameras.hpp:221      tracer.fire_all_rays(world, mode) -> image

Every call goes through libraytracer_amd.so; nothing here computes pixels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi
from .lib import check, dll
from .scene import Camera, FlatScene


@dataclass
class RenderMode:
    """ModeCfg + PathCfg (reference src/raytracer.cpp:40-51) and the constants build_renderer passes."""
    mode: str = "flat"
    n_rays: int = 10           # --n_rays
    roulette: int = 3          # --roulette
    max_depth: int = 5         # --max-depth
    seq_number: int = 54       # --seq-number
    background: tuple = (0.0, 0.0, 0.0)          # BLACK, raytracer.cpp:337,348,352
    ambient: tuple = (0.1, 0.1, 0.1)             # DARK_GREY, renderers.hpp:103
    max_mirror_bounces: int = 64                 # the reference has no cap (renderers.hpp:129-152)
    exact: bool = False                          # path mode: unfused reference-order arithmetic in the scan
    object_space_spheres: bool = False           # path mode, large worlds: never use the world-space sphere form
    bvh: bool = False                            # path mode: closest sphere through a BVH instead of the linear scan
    valu_only: bool = False                      # path mode, large worlds: the sphere filter on the vector ALU, no MFMA instruction
    scan_all_tiles: bool = False                 # path mode, worlds of many sphere tiles: every ray against every tile (no cluster bounds)


@dataclass
class RowBands:
    """Rows of the image one call renders: local row lr -> (lr // band_rows) * band_stride + band_offset + lr % band_rows."""
    band_rows: int
    band_stride: int
    band_offset: int
    n_local_rows: int

    @staticmethod
    def whole(height: int) -> "RowBands":
        return RowBands(height, height, 0, height)

    @staticmethod
    def cyclic(height: int, rank: int, world: int, band_rows: int = 16) -> "RowBands":
        """Bands of band_rows rows dealt round-robin to ranks: path cost is strongly row dependent.
        raytracer_amd.dist uses band_rows = 1."""
        n = sum(min(band_rows, height - start) for start in range(rank * band_rows, height, world * band_rows))
        return RowBands(band_rows, band_rows * world, band_rows * rank, n)

    def image_rows(self) -> np.ndarray:
        lr = np.arange(self.n_local_rows)
        return (lr // self.band_rows) * self.band_stride + self.band_offset + lr % self.band_rows


class DeviceWorld:
    """A flattened World resident in HBM (rt_scene*)."""

    def __init__(self, scene: FlatScene, device: int = 0):
        self.scene = scene
        self.device = device
        self._handle = C.c_void_p()
        desc = scene.desc()
        check(dll().rt_scene_create(C.byref(desc), device, C.byref(self._handle)))

    def close(self):
        if self._handle:
            dll().rt_scene_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._handle:
            raise RuntimeError("DeviceWorld is closed")
        return self._handle

    # -- batch operators (parity entry points) ---------------------------------------------------
    def ray_intersection(self, rays: np.ndarray) -> np.ndarray:
        """World::ray_intersection for a batch (rays: abi.RAY_DTYPE) -> abi.HIT_DTYPE."""
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
        check(dll().rt_trace_rays(self.handle, rays.ctypes.data, len(rays), hits.ctypes.data))
        return hits

    def offset_if_visible(self, viewer, surface, normal) -> np.ndarray:
        viewer = np.ascontiguousarray(viewer, dtype=np.float32).reshape(-1, 3)
        surface = np.ascontiguousarray(surface, dtype=np.float32).reshape(-1, 3)
        normal = np.ascontiguousarray(normal, dtype=np.float32).reshape(-1, 3)
        vis = np.zeros(len(viewer), dtype=np.uint8)
        check(dll().rt_offset_if_visible(self.handle, viewer.ctypes.data, surface.ctypes.data, normal.ctypes.data,
                                         len(viewer), vis.ctypes.data))
        return vis


class MultiDeviceWorld:
    """One replica of a flattened World per GPU of the node (rt_multi*): the frame's rows are dealt round-robin to the
    devices and gathered on the first one with RCCL (peer copies when ranks share a device or peer_copies=True)."""

    def __init__(self, scene: FlatScene, devices=None, peer_copies: bool = False, rccl_self: bool = False):
        self.scene = scene
        devices = list(devices) if devices is not None else []
        self._handle = C.c_void_p()
        desc = scene.desc()
        arr = (C.c_int * max(len(devices), 1))(*devices)
        flags = (abi.RT_MULTI_PEER_COPIES if peer_copies else 0) | (abi.RT_MULTI_RCCL_SELF if rccl_self else 0)
        check(dll().rt_multi_create(C.byref(desc), arr, len(devices), flags, C.byref(self._handle)))
        self.n_ranks = dll().rt_multi_rank_count(self._handle)
        self.uses_rccl = bool(dll().rt_multi_uses_rccl(self._handle))
        self.first_device = devices[0] if devices else 0

    def close(self):
        if self._handle:
            dll().rt_multi_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._handle:
            raise RuntimeError("MultiDeviceWorld is closed")
        return self._handle


def aa_jitter(sample_index, device: int = 0):
    k = np.ascontiguousarray(sample_index, dtype=np.uint64)
    u = np.zeros(len(k), dtype=np.float32)
    v = np.zeros(len(k), dtype=np.float32)
    check(dll().rt_aa_jitter(device, k.ctypes.data, len(k), u.ctypes.data, v.ctypes.data))
    return u, v


def scatter_rays(brdf_kind: int, pcg_state_inc, in_dir, normal, device: int = 0, exact: bool = True):
    """BRDF::scatter_ray directions for a batch; returns (directions [n,3], advanced pcg [n,2]).
    exact=True: the reference's formulas; False: the forms the default (FMA) path kernels use."""
    pcg = np.ascontiguousarray(pcg_state_inc, dtype=np.uint64).reshape(-1, 2).copy()
    in_dir = np.ascontiguousarray(in_dir, dtype=np.float32).reshape(-1, 3)
    normal = np.ascontiguousarray(normal, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(normal)
    check(dll().rt_scatter_rays(device, brdf_kind, abi.RT_FLAG_EXACT_ARITHMETIC if exact else 0, pcg.ctypes.data, in_dir.ctypes.data, normal.ctypes.data, len(normal),
                                out.ctypes.data))
    return out, pcg


class GpuImageTracer:
    """ImageTracer with the per-pixel loop on the GPU."""

    def __init__(self, width: int, height: int, camera: Camera, samples_per_pixel_edge: int = 1):
        if width <= 0 or height <= 0:
            raise ValueError("HdrImage dimensions must be positive")  # colors.hpp:386-389
        self.width, self.height = width, height
        self.camera = camera
        self.samples_per_pixel_edge = samples_per_pixel_edge
        self.last_stats: abi.rt_stats | None = None

    def make_cfg(self, mode: RenderMode, rows: RowBands | None = None) -> abi.rt_render_cfg:
        rows = rows or RowBands.whole(self.height)
        cfg = abi.rt_render_cfg()
        cfg.width, cfg.height, cfg.spp_edge = self.width, self.height, self.samples_per_pixel_edge
        cfg.mode = abi.MODES[mode.mode]
        cfg.n_rays, cfg.roulette, cfg.max_depth, cfg.seq_number = mode.n_rays, mode.roulette, mode.max_depth, mode.seq_number
        cfg.background[:] = mode.background
        cfg.ambient[:] = mode.ambient
        cfg.max_mirror_bounces = mode.max_mirror_bounces
        cfg.rng_mode = abi.RT_RNG_PER_NODE_STREAMS
        cfg.band_rows, cfg.band_stride, cfg.band_offset = rows.band_rows, rows.band_stride, rows.band_offset
        cfg.n_local_rows = rows.n_local_rows
        cfg.flags = (abi.RT_FLAG_EXACT_ARITHMETIC if mode.exact else 0) | \
            (abi.RT_FLAG_OBJECT_SPACE_SPHERES if mode.object_space_spheres else 0) | \
            (abi.RT_FLAG_ACCEL_BVH if mode.bvh else 0) | (abi.RT_FLAG_VALU_ONLY if mode.valu_only else 0) | \
            (abi.RT_FLAG_SCAN_ALL_TILES if mode.scan_all_tiles else 0)
        return cfg

    def fire_all_rays(self, world: DeviceWorld, mode: RenderMode, rows: RowBands | None = None, out=None,
                      stream=None, want_stats: bool = True):
        """Renders this tracer's rows.

        out=None      -> numpy float32 [n_local_rows, W, 3] (device framebuffer + D2H inside the library)
        out=<tensor>  -> a CUDA/HIP torch tensor of that shape on world.device, filled in place (stays in HBM);
                         `stream` = a raw hipStream_t (int) or None for torch's current stream.
        """
        cfg = self.make_cfg(mode, rows)
        if cfg.n_local_rows == 0:  # a rank with no band of this frame (more ranks than bands)
            self.last_stats = abi.rt_stats() if want_stats else None
            return np.empty((0, self.width, 3), dtype=np.float32) if out is None else out
        cam = self.camera.to_abi()
        stats = abi.rt_stats()
        stats_p = C.byref(stats) if want_stats else None
        if out is None:
            img = np.empty((cfg.n_local_rows, self.width, 3), dtype=np.float32)
            check(dll().rt_render_host(world.handle, C.byref(cam), C.byref(cfg), img.ctypes.data, stats_p))
            self.last_stats = stats if want_stats else None
            return img
        import torch
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                and tuple(out.shape) == (cfg.n_local_rows, self.width, 3) and out.device.index == world.device):
            raise ValueError("out must be a contiguous float32 CUDA tensor [n_local_rows, W, 3] on the world's device")
        if stream is None:
            stream = torch.cuda.current_stream(out.device).cuda_stream
        check(dll().rt_render(world.handle, C.byref(cam), C.byref(cfg), out.data_ptr(), stream, stats_p))
        self.last_stats = stats if want_stats else None
        return out


    def fire_all_rays_multi(self, world: MultiDeviceWorld, mode: RenderMode, out=None, progress=None, n_reports: int = 8):
        """The whole frame on every device of `world`.  out=None -> numpy [H, W, 3]; out=<tensor on the first device> stays
        in HBM.  Sets last_stats (summed over ranks) and last_timing (abi.rt_multi_timing).  progress (host output only):
        the reference's callback, called up to n_reports times with the finished fraction of the frame's samples over all
        ranks (rt_multi_render_frame_host)."""
        cfg = self.make_cfg(mode)
        cam = self.camera.to_abi()
        stats, timing = abi.rt_stats(), abi.rt_multi_timing()
        if out is None and progress is not None:
            img = np.empty((self.height, self.width, 3), dtype=np.float32)
            cb = C.CFUNCTYPE(None, C.c_float, C.c_void_p)(lambda f, _u: progress(f))
            check(dll().rt_multi_render_frame_host(world.handle, C.byref(cam), C.byref(cfg), img.ctypes.data, n_reports,
                                                   C.cast(cb, C.c_void_p), None, C.byref(stats), C.byref(timing)))
        elif out is None:
            img = np.empty((self.height, self.width, 3), dtype=np.float32)
            check(dll().rt_multi_render_host(world.handle, C.byref(cam), C.byref(cfg), img.ctypes.data, C.byref(stats), C.byref(timing)))
        else:
            import torch
            if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (self.height, self.width, 3)
                    and out.device.index == world.first_device):
                raise ValueError("out must be a contiguous float32 CUDA tensor [H, W, 3] on the first device of the world")
            check(dll().rt_multi_render(world.handle, C.byref(cam), C.byref(cfg), out.data_ptr(), C.byref(stats), C.byref(timing)))
            img = out
        self.last_stats, self.last_timing = stats, timing
        return img

    def fire_all_rays_host(self, world: DeviceWorld, mode: RenderMode, n_slices: int = 8, progress=None):
        """rt_render_frame_host: the frame into host memory with the reference's progress callback (values in (0, 1])."""
        cfg = self.make_cfg(mode)
        cam = self.camera.to_abi()
        stats = abi.rt_stats()
        img = np.empty((self.height, self.width, 3), dtype=np.float32)
        cb_type = C.CFUNCTYPE(None, C.c_float, C.c_void_p)
        cb = cb_type((lambda f, _u: progress(f)) if progress else 0)
        check(dll().rt_render_frame_host(world.handle, C.byref(cam), C.byref(cfg), img.ctypes.data, n_slices,
                                         C.cast(cb, C.c_void_p) if progress else None, None, C.byref(stats)))
        self.last_stats = stats
        return img


# -- the tone-mapping tail (reference src/raytracer.cpp:176-186, include/colors.hpp:469-516) -------------------------
DEFAULT_DELTA_LOG = 1e-10           # include/colors.hpp:41
DEFAULT_AVG_LUMINOSITY_DARK_MODE = 0.1  # include/colors.hpp:43-44 (the value --dark-mode passes to normalize_image)


def make_tonemap_cfg(alpha=0.18, gamma=2.2, avg_luminosity=None, delta=DEFAULT_DELTA_LOG, serial_sum=False) -> abi.rt_tonemap_cfg:
    cfg = abi.rt_tonemap_cfg()
    cfg.alpha, cfg.gamma, cfg.delta = alpha, gamma, delta
    cfg.has_avg_luminosity = 0 if avg_luminosity is None else 1
    cfg.avg_luminosity = 0.0 if avg_luminosity is None else float(avg_luminosity)
    cfg.flags = abi.RT_TONEMAP_SERIAL_SUM if serial_sum else 0
    return cfg


def average_luminosity(image, delta=DEFAULT_DELTA_LOG, serial_sum=False, stream=None) -> float:
    """HdrImage::average_luminosity of a float32 [H, W, 3] CUDA/HIP tensor (stays in HBM)."""
    import torch
    if not (image.is_cuda and image.dtype == torch.float32 and image.is_contiguous() and image.shape[-1] == 3):
        raise ValueError("image must be a contiguous float32 CUDA tensor [..., 3]")
    if stream is None:
        stream = torch.cuda.current_stream(image.device).cuda_stream
    avg = C.c_float()
    check(dll().rt_average_luminosity(image.device.index, image.data_ptr(), image.numel() // 3, delta,
                                      abi.RT_TONEMAP_SERIAL_SUM if serial_sum else 0, stream, C.byref(avg)))
    return avg.value


def tonemap(image, alpha=0.18, gamma=2.2, avg_luminosity=None, serial_sum=False, want_hdr=False, stream=None):
    """normalize_image + clamp_image + 8-bit gamma transform.

    image: float32 [H, W, 3] — a CUDA/HIP tensor (result: uint8 tensor on the same device, nothing leaves HBM) or a numpy
    array (host round trip through rt_tonemap_host).  Returns (ldr uint8 [H, W, 3], average luminosity used[, hdr])."""
    cfg = make_tonemap_cfg(alpha, gamma, avg_luminosity, serial_sum=serial_sum)
    avg = C.c_float()
    if isinstance(image, np.ndarray):
        img = np.ascontiguousarray(image, dtype=np.float32)
        ldr = np.empty(img.shape, dtype=np.uint8)
        check(dll().rt_tonemap_host(0, img.ctypes.data, img.size // 3, C.byref(cfg), ldr.ctypes.data, C.byref(avg)))
        return ldr, avg.value
    import torch
    if not (image.is_cuda and image.dtype == torch.float32 and image.is_contiguous() and image.shape[-1] == 3):
        raise ValueError("image must be a contiguous float32 CUDA tensor [..., 3]")
    if stream is None:
        stream = torch.cuda.current_stream(image.device).cuda_stream
    ldr = torch.empty(image.shape, dtype=torch.uint8, device=image.device)
    hdr = torch.empty_like(image) if want_hdr else None
    check(dll().rt_tonemap(image.device.index, image.data_ptr(), image.numel() // 3, C.byref(cfg), ldr.data_ptr(),
                           hdr.data_ptr() if want_hdr else None, stream, C.byref(avg)))
    return (ldr, avg.value, hdr) if want_hdr else (ldr, avg.value)
