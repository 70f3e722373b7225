"""rt_render_noalloc (the entry point that allocates nothing: its scratch is a file-scope device array) against
rt_render, through the C-ABI.  Bar: bit-identical frames and equal ray counts — sample values are 64-bit fixed-point
sums that do not depend on the grid, so the smaller grid a fixed workspace may force changes nothing.

Two calls in a row, the second with a smaller input: the array keeps the first call's contents and the second call
must clear what has to start at zero.  Collected last (file name) so that a failure here hides no other test.
"""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.lib import dll
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode, RowBands

pytestmark = pytest.mark.gpu


def _render(entry, world, tracer, mode, rows=None):
    import torch
    cfg = tracer.make_cfg(mode, rows)
    cam = tracer.camera.to_abi()
    out = torch.full((cfg.n_local_rows, tracer.width, 3), float("nan"), dtype=torch.float32, device=f"cuda:{world.device}")
    stats = abi.rt_stats()
    stream = torch.cuda.current_stream(out.device).cuda_stream
    st = getattr(dll(), entry)(world.handle, C.byref(cam), C.byref(cfg), out.data_ptr(), stream, C.byref(stats))
    msg = dll().rt_last_error().decode()
    torch.cuda.synchronize()
    return st, msg, out.cpu().numpy(), stats


def _pair(world, tracer, mode, rows=None):
    st, msg, got, gs = _render("rt_render_noalloc", world, tracer, mode, rows)
    assert st == abi.RT_OK, msg
    st, msg, want, ws = _render("rt_render", world, tracer, mode, rows)
    assert st == abi.RT_OK, msg
    assert not np.isnan(got).any()
    assert np.array_equal(got, want), f"{int((got != want).any(axis=-1).sum())} pixels differ"
    assert (gs.rays, gs.samples, gs.capped_samples) == (ws.rays, ws.samples, ws.capped_samples)
    return got


def test_noalloc_equals_rt_render_two_calls_second_smaller(scenes):
    world = DeviceWorld(scenes["csg_example"])
    path = RenderMode("path", n_rays=4, roulette=2, max_depth=3)
    big = GpuImageTracer(160, 96, scenes["csg_example"].camera, 2)
    small = GpuImageTracer(48, 32, scenes["csg_example"].camera, 2)
    first = _pair(world, big, path)
    second = _pair(world, small, path)           # smaller input right after: stale samples / counters / stacks in the array
    assert first.shape == (96, 160, 3) and second.shape == (32, 48, 3)
    _pair(world, small, RenderMode("flat"))      # a deterministic mode, AA through the sample buffer
    _pair(world, GpuImageTracer(64, 40, scenes["csg_example"].camera, 1), RenderMode("onoff"))  # spp 1: no sample buffer
    _pair(world, big, path, RowBands.cyclic(96, 1, 3, band_rows=4))  # a rank's share of a sharded frame
    world.close()


def test_noalloc_wide_and_bvh_kernels():
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(600)[0]
    world = DeviceWorld(scene)
    tr = GpuImageTracer(64, 40, scene.camera, 2)
    kw = dict(n_rays=3, roulette=2, max_depth=3)
    _pair(world, tr, RenderMode("path", **kw))
    _pair(world, tr, RenderMode("path", bvh=True, **kw))
    _pair(world, GpuImageTracer(32, 20, scene.camera, 1), RenderMode("path", **kw))
    world.close()


def test_noalloc_rejects_what_does_not_fit(scenes):
    """Beyond the workspace: a clear error, no allocation, and the entry point still works afterwards."""
    world = DeviceWorld(scenes["demo_path"])
    cam = scenes["demo_path"].camera
    # sample buffer alone: 3840 * 2160 * 16 * 12 bytes = 1.6 GB
    st, msg, _, _ = _render("rt_render_noalloc", world, GpuImageTracer(3840, 2160, cam, 4), RenderMode("flat"))
    assert st == abi.RT_ERR_INVALID_ARGUMENT and "sample buffer" in msg and str(abi.RT_NOALLOC_SCRATCH_BYTES) in msg
    # node stacks: depth 64 x 256 lanes x 64 bytes (a 16-word node record) = 1 MiB per workgroup, one per CU does not fit
    st, msg, _, _ = _render("rt_render_noalloc", world, GpuImageTracer(256, 256, cam, 1), RenderMode("path", n_rays=2, max_depth=64))
    assert st == abi.RT_ERR_INVALID_ARGUMENT and "node stacks" in msg
    _pair(world, GpuImageTracer(40, 30, cam, 2), RenderMode("path", n_rays=3, roulette=2, max_depth=3))
    # the same frame in row slices fits
    tr = GpuImageTracer(1280, 960, cam, 2)
    rows = RowBands.cyclic(960, 0, 8, band_rows=1)
    _pair(world, tr, RenderMode("flat"), rows)
    world.close()
