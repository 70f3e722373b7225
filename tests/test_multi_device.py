"""rt_multi_* (one frame on several GPUs of one node, through the C-ABI) and rt_render_frame_host.

The multi-device frame must equal the single-call frame BIT FOR BIT for any number of ranks: a sample's random streams
depend only on its global index and sample values are integer sums.  On the one-GPU test box the ranks are "virtual":
the same device listed k times (bands travel by peer copies, since an RCCL communicator cannot hold a device twice), and
the RCCL transport itself is exercised with one rank that sends its band to itself (RT_MULTI_RCCL_SELF).  With two or
more real devices the same test also runs one rank per device over RCCL.
"""
import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, MultiDeviceWorld, RenderMode

pytestmark = pytest.mark.gpu

PATH = dict(n_rays=4, roulette=2, max_depth=3)


@pytest.mark.parametrize("scene_name,w,h,aa", [("demo_path", 70, 45, 2), ("csg_example", 64, 37, 1)])
def test_virtual_ranks_on_one_device_equal_the_single_call(scenes, scene_name, w, h, aa):
    scene = scenes[scene_name]
    single = DeviceWorld(scene)
    tr = GpuImageTracer(w, h, scene.camera, aa)
    want = {m: tr.fire_all_rays(single, RenderMode(m, **(PATH if m == "path" else {}))) for m in ("path", "flat", "onoff")}
    rays = {}
    for m in want:
        tr.fire_all_rays(single, RenderMode(m, **(PATH if m == "path" else {})))
        rays[m] = int(tr.last_stats.rays)
    single.close()
    for devices in ([0], [0, 0], [0, 0, 0], [0] * 7):
        multi = MultiDeviceWorld(scene, devices)
        assert multi.n_ranks == len(devices) and not multi.uses_rccl
        for m, img in want.items():
            got = tr.fire_all_rays_multi(multi, RenderMode(m, **(PATH if m == "path" else {})))
            assert np.array_equal(got, img), (m, devices)
            assert int(tr.last_stats.rays) == rays[m] and int(tr.last_stats.samples) == w * h * aa * aa
            t = tr.last_timing
            assert t.n_ranks == len(devices) and sum(t.rays[: len(devices)]) == rays[m]
            assert t.slowest_render_ms == max(t.render_ms[: len(devices)]) > 0 and t.gather_ms >= 0
        multi.close()


def test_rccl_transport_self_send_and_device_tensor(scenes):
    """One rank whose band goes through ncclSend / ncclRecv (to itself): the RCCL path of the gather on a single GPU."""
    import torch
    scene = scenes["demo_path"]
    single = DeviceWorld(scene)
    tr = GpuImageTracer(96, 54, scene.camera, 2)
    want = tr.fire_all_rays(single, RenderMode("path", **PATH))
    single.close()
    multi = MultiDeviceWorld(scene, [0], rccl_self=True)
    assert multi.uses_rccl and multi.n_ranks == 1
    out = torch.empty((54, 96, 3), dtype=torch.float32, device="cuda:0")
    for _ in range(2):  # communicators and buffers are reused from frame to frame
        out.fill_(float("nan"))
        tr.fire_all_rays_multi(multi, RenderMode("path", **PATH), out=out)
        assert np.array_equal(out.cpu().numpy(), want)
    multi.close()


def test_every_visible_device_over_rccl(scenes):
    """n_devices = 0: one rank per visible GPU, gathered over RCCL when there are at least two."""
    import torch
    scene = scenes["csg_example"]
    single = DeviceWorld(scene)
    tr = GpuImageTracer(80, 50, scene.camera, 2)
    want = tr.fire_all_rays(single, RenderMode("path", **PATH))
    single.close()
    multi = MultiDeviceWorld(scene)
    assert multi.n_ranks == torch.cuda.device_count()
    assert multi.uses_rccl == (multi.n_ranks > 1)
    assert np.array_equal(tr.fire_all_rays_multi(multi, RenderMode("path", **PATH)), want)
    multi.close()


def test_multi_rejects_bad_arguments(scenes):
    from raytracer_amd.lib import RtError
    scene = scenes["demo_path"]
    with pytest.raises(RtError):
        MultiDeviceWorld(scene, [99])
    multi = MultiDeviceWorld(scene, [0, 0])
    tr = GpuImageTracer(16, 16, scene.camera, 1)
    with pytest.raises(RtError):
        tr.fire_all_rays_multi(multi, RenderMode("path", n_rays=0))
    cfg = tr.make_cfg(RenderMode("flat"))
    cfg.n_local_rows = 8  # a band description: the multi-device call deals the rows itself
    import ctypes as C
    from raytracer_amd.lib import dll
    cam = scene.camera.to_abi()
    img = np.zeros((16, 16, 3), dtype=np.float32)
    assert dll().rt_multi_render_host(multi.handle, C.byref(cam), C.byref(cfg), img.ctypes.data, None, None) == abi.RT_ERR_INVALID_ARGUMENT
    multi.close()


def test_frame_host_slices_and_progress(scenes):
    """rt_render_frame_host: same pixels as the single call, one progress report per slice, increasing, ending at 1."""
    scene = scenes["demo_path"]
    world = DeviceWorld(scene)
    tr = GpuImageTracer(90, 61, scene.camera, 2)
    for mode in (RenderMode("path", **PATH), RenderMode("pointlight"), RenderMode("onoff")):
        want = tr.fire_all_rays(world, mode)
        rays = int(tr.last_stats.rays)
        seen = []
        got = tr.fire_all_rays_host(world, mode, n_slices=8, progress=seen.append)
        assert np.array_equal(got, want)
        # deterministic modes: one report per slice; path mode: one launch, up to 8 reports read off its sample queue
        assert (len(seen) == 8 if mode.mode != "path" else 1 <= len(seen) <= 8)
        assert seen == sorted(set(seen)) and seen[-1] == 1.0 and seen[0] > 0
        assert int(tr.last_stats.rays) == rays and tr.last_stats.kernel_ms > 0
        assert np.array_equal(tr.fire_all_rays_host(world, mode), want)  # no callback: one launch
        assert int(tr.last_stats.rays) == rays
    world.close()


@pytest.mark.parametrize("devices,rccl_self", [([0], False), ([0, 0, 0], False), ([0], True)])
def test_multi_frame_host_progress(scenes, devices, rccl_self):
    """rt_multi_render_frame_host: the multi-device frame with the reference's progress callback.  Same pixels and ray
    count as the single call; reports increase, lie in (0, 1] and end at exactly 1; gather_ms is a HIP-event duration."""
    scene = scenes["demo_path"]
    single = DeviceWorld(scene)
    tr = GpuImageTracer(120, 80, scene.camera, 2)
    multi = MultiDeviceWorld(scene, devices, rccl_self=rccl_self)
    for mode in (RenderMode("path", **PATH), RenderMode("pointlight")):
        want = tr.fire_all_rays(single, mode)
        rays = int(tr.last_stats.rays)
        seen = []
        got = tr.fire_all_rays_multi(multi, mode, progress=seen.append, n_reports=6)
        assert np.array_equal(got, want), (mode.mode, devices)
        assert int(tr.last_stats.rays) == rays
        assert 1 <= len(seen) <= 6 and seen == sorted(set(seen)) and seen[0] > 0 and seen[-1] == 1.0
        t = tr.last_timing
        assert t.n_ranks == len(devices) and sum(t.rays[: len(devices)]) == rays and 0 <= t.gather_ms < 1000
        # the counters' head is cleared per frame: a second frame reports from the start again
        seen2 = []
        assert np.array_equal(tr.fire_all_rays_multi(multi, mode, progress=seen2.append, n_reports=6), want)
        assert seen2 and seen2[-1] == 1.0 and seen2 == sorted(set(seen2))
    multi.close()
    single.close()
