"""Animation batches: one flattened scene, one camera per frame.

The reference renders an animation by re-parsing the scene with `--define-float clock=<angle>` per frame
(scripts/generate_image.sh); only the camera depends on `clock` (samples/demo_path_tracing.txt).  Here the scene stays in
HBM and `orbit_camera` rebuilds that camera.  Pinned by the reference's own parse and render of the scene at clock = 30.
"""
import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd.dist import frames_for_rank, orbit_camera


def test_orbit_camera_equals_reference_parse(scenes):
    cam = orbit_camera(scenes["demo_path"].camera, 30.0)
    ref = scenes["demo_path_clock30"].camera
    assert np.array_equal(cam.transformation.fwd, ref.transformation.fwd) and cam.kind == ref.kind and cam.distance == ref.distance
    assert scenes["demo_path"].equals(scenes["demo_path_clock30"])   # nothing but the camera depends on the clock


def test_frames_are_dealt_round_robin():
    got = sorted(f for r in range(3) for f in frames_for_rank(10, r, 3))
    assert got == list(range(10)) and list(frames_for_rank(10, 1, 3)) == [1, 4, 7]


def test_oracle_frame_at_clock_30(scenes, golden_images):
    img, _ = O.render(scenes["demo_path"], O.make_cfg(160, 120, "flat"), orbit_camera(scenes["demo_path"].camera, 30.0))
    assert np.array_equal(img, golden_images["flat_demo_path_clock30_160x120_aa1"])


@pytest.mark.gpu
def test_gpu_animation_reuses_one_device_world(scenes, golden_images):
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    world = DeviceWorld(scenes["demo_path"])
    base = scenes["demo_path"].camera
    frames = {}
    for clock in (0.0, 30.0, 60.0):
        frames[clock] = GpuImageTracer(160, 120, orbit_camera(base, clock), 1).fire_all_rays(world, RenderMode("flat"))
    world.close()
    assert np.array_equal(frames[30.0], golden_images["flat_demo_path_clock30_160x120_aa1"])
    assert np.array_equal(frames[0.0], golden_images["flat_demo_path_160x120_aa1"])
    assert not np.array_equal(frames[60.0], frames[30.0])
