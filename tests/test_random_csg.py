"""Random CSG worlds (tests/golden/make_random_csg_golden.py): the reference's own renders of scenes with nested CSG trees
of all four operations, rotated and non-uniformly scaled leaves and nodes, ellipsoids, checkered pigments, two lights,
perspective and orthogonal cameras.  CPU: the oracle reproduces them bit for bit (path mode with the reference's serial
generator included).  GPU: the deterministic modes reproduce them bit for bit through the C-ABI; path mode follows the
oracle on the same per-node streams."""
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import FlatScene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_SCENES = 6
CASES = [("onoff", 96, 64, 1), ("flat", 96, 64, 1), ("flat", 72, 48, 2), ("pointlight", 96, 64, 1)]


@pytest.fixture(scope="module")
def images():
    return np.load(os.path.join(GOLDEN, "random_csg_images.npz"))


def scene(k):
    return FlatScene.from_json(os.path.join(GOLDEN, f"random_csg_{k}.scene.json"))


def test_fixtures_exercise_what_they_claim():
    ops, depth_seen, kinds = set(), 0, set()
    for k in range(N_SCENES):
        s = scene(k)
        ops |= set(int(x) for x in s.csg_op)
        kinds.add(int(s.camera.kind))
        assert s.count("n_csg_nodes") >= 2 and s.count("n_lights") == 2
    assert ops == {abi.RT_CSG_UNION, abi.RT_CSG_INTERSECTION, abi.RT_CSG_DIFFERENCE, abi.RT_CSG_FUSION}
    assert kinds == {abi.RT_CAMERA_PERSPECTIVE, abi.RT_CAMERA_ORTHOGONAL}


@pytest.mark.parametrize("k", range(N_SCENES))
def test_oracle_reproduces_the_reference_renders(images, k):
    s = scene(k)
    for mode, w, h, aa in CASES:
        got, _ = O.render(s, O.make_cfg(w, h, mode, spp_edge=aa))
        want = images[f"{k}_{mode}_{w}x{h}_aa{aa}"]
        assert np.array_equal(got, want), (k, mode, int((got != want).any(axis=-1).sum()))
    got, _ = O.render(s, O.make_cfg(36, 24, "path", spp_edge=1, n_rays=2, roulette=2, max_depth=3, seq_number=11,
                                    rng_mode=abi.RT_RNG_SERIAL_REFERENCE))
    assert np.array_equal(got, images[f"{k}_path_36x24_aa1_serial"])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(N_SCENES))
def test_gpu_reproduces_the_reference_renders(images, k):
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    s = scene(k)
    world = DeviceWorld(s)
    try:
        for mode, w, h, aa in CASES:
            got = GpuImageTracer(w, h, s.camera, aa).fire_all_rays(world, RenderMode(mode))
            want = images[f"{k}_{mode}_{w}x{h}_aa{aa}"]
            assert np.array_equal(got, want), (k, mode, int((got != want).any(axis=-1).sum()))
        # path mode: same per-node streams as the oracle; the unfused kernel may leave the oracle's path where a libm ulp
        # in a scattered direction flips a later hit decision, the fused kernel more often — same estimator either way
        kw = dict(n_rays=3, roulette=2, max_depth=3, seq_number=11)
        want, ostats = O.render(s, O.make_cfg(60, 40, "path", spp_edge=2, **kw))
        t = GpuImageTracer(60, 40, s.camera, 2)
        exact = t.fire_all_rays(world, RenderMode("path", exact=True, **kw))
        # Scene 1 holds a large mirror ellipsoid far from the floor: a reflected ray (tmin 1e-5) re-hits its own surface
        # at t = 1.1e-5 ... 4e-5 or not at all depending on the last bit of the incoming direction, which comes from the
        # diffuse scattering formulas (device libm vs glibc, 1 ulp).  Walked side by side (scripts/dbg_paths.py): every
        # diverging sample there is such a self-intersection decision; closest hits for identical rays are identical
        # (test below).  7 % of its samples take another branch of the same estimator; elsewhere < 0.1 %.
        assert (np.abs(exact - want).max(axis=-1) > 1e-4).mean() <= (0.30 if k == 1 else 0.02)
        assert abs(int(t.last_stats.rays) - int(ostats.rays)) <= 0.02 * ostats.rays
        assert abs(float(exact.mean()) - float(want.mean())) <= 0.01 * float(want.mean())
        fast = t.fire_all_rays(world, RenderMode("path", **kw))
        assert (np.abs(fast - want).max(axis=-1) > 1e-4).mean() <= (0.60 if k == 1 else 0.10)
        assert abs(float(fast.mean()) - float(want.mean())) <= 0.03 * float(want.mean())
    finally:
        world.close()


def _secondary_like_rays(s, n=6000, seed=3):
    """Rays as the path tracer produces them below the first bounce: origins ON the surfaces (the oracle's hit points of
    random probe rays), random directions, tmin 1e-3 (diffuse) or 1e-5 (mirror)."""
    rng = np.random.default_rng(seed)
    probe = np.zeros(n, dtype=abi.RAY_DTYPE)
    probe["origin"] = rng.uniform([-4, -4, -1], [6, 4, 4], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    probe["direction"], probe["tmin"], probe["tmax"] = d, 1e-5, np.inf
    first = O.trace_rays(s, probe)
    ok = first["object"] >= 0
    rays = np.zeros(int(ok.sum()), dtype=abi.RAY_DTYPE)
    rays["origin"] = first["point"][ok]
    rays["direction"] = rng.normal(size=(len(rays), 3)).astype(np.float32) * rng.uniform(0.2, 3.0, (len(rays), 1)).astype(np.float32)
    rays["tmin"] = np.where(rng.random(len(rays)) < 0.5, np.float32(1e-3), np.float32(1e-5))
    rays["tmax"] = np.inf
    return np.concatenate([probe, rays])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(N_SCENES))
def test_gpu_closest_hits_equal_the_oracle_on_secondary_like_rays(k):
    """World::ray_intersection for rays that start on surfaces, where self-intersection decisions sit on the last bits:
    the device's heap-free CSG evaluation must take every one of them as the oracle's interval lists do."""
    from raytracer_amd.tracer import DeviceWorld
    s = scene(k)
    rays = _secondary_like_rays(s)
    want = O.trace_rays(s, rays)
    world = DeviceWorld(s)
    got = world.ray_intersection(rays)
    world.close()
    assert np.array_equal(got["object"], want["object"]), int((got["object"] != want["object"]).sum())
    hit = want["object"] >= 0
    assert hit.sum() > 3000
    assert np.array_equal(got["t"][hit], want["t"][hit])
    assert np.array_equal(got["point"][hit], want["point"][hit])
    assert np.array_equal(got["normal"][hit], want["normal"][hit])
    np.testing.assert_allclose(got["uv"][hit], want["uv"][hit], rtol=0, atol=1e-6)


# ---- scenes 6 .. 23: flattened by the reference's parser, no reference renders; the GPU against the (pinned) oracle ----
EXTRA = range(N_SCENES, N_SCENES + 18)


@pytest.mark.gpu
@pytest.mark.parametrize("k", EXTRA)
def test_gpu_equals_the_oracle_on_more_random_worlds(k):
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    s = scene(k)
    world = DeviceWorld(s)
    try:
        for mode, w, h, aa in [("onoff", 64, 40, 1), ("flat", 64, 40, 2), ("pointlight", 64, 40, 1)]:
            t = GpuImageTracer(w, h, s.camera, aa)
            got = t.fire_all_rays(world, RenderMode(mode))
            want, ostats = O.render(s, O.make_cfg(w, h, mode, spp_edge=aa))
            assert np.array_equal(got, want), (k, mode, int((got != want).any(axis=-1).sum()))
            assert int(t.last_stats.rays) == int(ostats.rays)
        rays = _secondary_like_rays(s, n=3000, seed=k)
        want = O.trace_rays(s, rays)
        got = world.ray_intersection(rays)
        hit = want["object"] >= 0
        assert np.array_equal(got["object"], want["object"]) and np.array_equal(got["t"][hit], want["t"][hit])
        assert np.array_equal(got["point"][hit], want["point"][hit]) and np.array_equal(got["normal"][hit], want["normal"][hit])
        # path mode, fused kernel with its tables in LDS: the estimator's mean
        kw = dict(n_rays=3, roulette=2, max_depth=3, seq_number=k)
        want, _ = O.render(s, O.make_cfg(48, 32, "path", spp_edge=2, **kw))
        fast = GpuImageTracer(48, 32, s.camera, 2).fire_all_rays(world, RenderMode("path", **kw))
        assert abs(float(fast.mean()) - float(want.mean())) <= 0.03 * float(want.mean())
    finally:
        world.close()
