"""Parity of the HIP path (through the C-ABI, raytracer_amd/libraytracer_amd.so) with
 (a) the golden vectors rendered by the unmodified reference (tests/golden), and
 (b) the CPU oracle (oracle/rt_oracle.c) on the same seeded inputs.

Bars (BASELINE.json north_star):
  * onoff hit mask: identical.
  * t / world point / normal of World::ray_intersection, shadow tests, AA jitter: bit-exact (fp32
    add/mul/div/sqrt are correctly rounded on both sides, no FMA contraction in the predicates).
  * flat / pointlight images: bit-exact except pixels whose (u,v) falls within a few ulp of a
    checker edge, because atan2f/acosf of glibc and of the ROCm device library differ in the last
    bit; those pixels are counted and bounded (FLIP_BUDGET) and every other pixel is within FP_TOL.
  * path mode: per-pixel agreement with the oracle run on the same per-sample streams (outliers:
    paths that diverge at a discontinuity through an ulp-level difference in sinf/cosf/acosf/powf),
    and RMSE against the reference's own high-spp serial render within the Monte-Carlo bound.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode, RowBands, aa_jitter

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FP_TOL = 2e-6      # |gpu - reference| per channel for pixels that took the same branches
FLIP_BUDGET = 2e-3  # fraction of pixels allowed to sit on a uv discontinuity


@pytest.fixture(scope="module")
def worlds(scenes):
    w = {k: DeviceWorld(v) for k, v in scenes.items()}
    yield w
    for x in w.values():
        x.close()


def gpu_render(worlds, scenes, scene, mode, w, h, aa=1, rows=None, **kw):
    tr = GpuImageTracer(w, h, scenes[scene].camera, aa)
    img = tr.fire_all_rays(worlds[scene], RenderMode(mode=mode, **kw), rows)
    return img, tr.last_stats


def assert_image_close(img, want, key):
    assert img.shape == want.shape
    diff = np.abs(img - want).max(axis=-1)
    flips = diff > FP_TOL
    frac = flips.mean()
    assert frac <= FLIP_BUDGET, f"{key}: {flips.sum()} pixels differ by more than {FP_TOL} (max {diff.max()})"
    return int(flips.sum()), int((diff > 0).sum())


def test_onoff_c1_mask_identical(worlds, scenes):
    bits = np.load(os.path.join(GOLDEN, "onoff_demo_onoff_1280x960.maskbits.npy"))
    mask = np.unpackbits(bits)[: 1280 * 960].reshape(960, 1280).astype(bool)
    img, stats = gpu_render(worlds, scenes, "demo_onoff", "onoff", 1280, 960)
    assert set(np.unique(img)) <= {0.0, 1.0}
    assert np.array_equal(img[..., 0] > 0, mask)      # identical mask, no documented exceptions needed
    assert int((img[..., 0] > 0).sum()) == 22712
    assert stats.rays == 1280 * 960 and stats.samples == 1280 * 960


@pytest.mark.parametrize("key,scene,mode,w,h,aa", [
    ("onoff_demo_onoff_160x120_aa1", "demo_onoff", "onoff", 160, 120, 1),
    ("onoff_demo_onoff_160x120_aa2", "demo_onoff", "onoff", 160, 120, 2),
    ("onoff_csg_example_240x135_aa1", "csg_example", "onoff", 240, 135, 1),
    ("onoff_csg_example_240x135_aa2", "csg_example", "onoff", 240, 135, 2),
    ("pointlight_demo_onoff_160x120_aa1", "demo_onoff", "pointlight", 160, 120, 1),
    ("pointlight_demo_onoff_160x120_aa2", "demo_onoff", "pointlight", 160, 120, 2),
])
def test_uv_free_images_bit_identical(worlds, scenes, golden_images, key, scene, mode, w, h, aa):
    """No pigment in these renders reads (u,v): nothing transcendental is involved -> exact."""
    img, _ = gpu_render(worlds, scenes, scene, mode, w, h, aa)
    if mode == "pointlight":
        # cos_theta / normalisation are +,*,/,sqrt only -> exact as well
        assert np.array_equal(img, golden_images[key]), np.abs(img - golden_images[key]).max()
    else:
        assert np.array_equal(img, golden_images[key])


@pytest.mark.parametrize("key,scene,mode,w,h,aa", [
    ("flat_demo_path_160x120_aa1", "demo_path", "flat", 160, 120, 1),
    ("flat_demo_path_160x120_aa2", "demo_path", "flat", 160, 120, 2),
    ("pointlight_demo_path_160x120_aa1", "demo_path", "pointlight", 160, 120, 1),
    ("pointlight_demo_path_160x120_aa2", "demo_path", "pointlight", 160, 120, 2),
    ("flat_csg_example_240x135_aa1", "csg_example", "flat", 240, 135, 1),
    ("flat_csg_example_240x135_aa2", "csg_example", "flat", 240, 135, 2),
    ("flat_demo_path_clock30_160x120_aa1", "demo_path_clock30", "flat", 160, 120, 1),
])
def test_deterministic_images_vs_reference(worlds, scenes, golden_images, key, scene, mode, w, h, aa):
    img, _ = gpu_render(worlds, scenes, scene, mode, w, h, aa)
    flips, nonzero = assert_image_close(img, golden_images[key], key)
    print(f"{key}: {flips} uv-edge pixels, {nonzero} pixels with any difference")


def test_flat_and_pointlight_c2_full_size_vs_oracle(worlds, scenes):
    """BASELINE config 2: demo_path_tracing, flat and pointlight, 1280x960, AA=2 (oracle = CPU restatement,
    itself bit-identical to the reference on the 160x120 goldens)."""
    for mode in ("flat", "pointlight"):
        img, stats = gpu_render(worlds, scenes, "demo_path", mode, 1280, 960, 2)
        want, ostats = O.render(scenes["demo_path"], O.make_cfg(1280, 960, mode, spp_edge=2, max_mirror_bounces=64))
        assert_image_close(img, want, f"C2 {mode}")
        assert stats.capped_samples == ostats.capped_samples
        assert abs(int(stats.rays) - int(ostats.rays)) <= 64


@pytest.mark.parametrize("scene", ["demo_onoff", "demo_path", "csg_example"])
def test_world_ray_intersection_vs_reference(worlds, golden_traces, scene):
    rays = golden_traces[f"{scene}_rays"]
    r = np.zeros(len(rays), dtype=abi.RAY_DTYPE)
    r["origin"], r["direction"], r["tmin"], r["tmax"] = rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7]
    hits = worlds[scene].ray_intersection(r)
    assert np.array_equal(hits["object"], golden_traces[f"{scene}_object"])
    rec = golden_traces[f"{scene}_rec"]
    hit = hits["object"] >= 0
    assert np.array_equal(hits["t"][hit], rec[hit, 0])
    assert np.array_equal(hits["point"][hit], rec[hit, 1:4])
    assert np.array_equal(hits["normal"][hit], rec[hit, 4:7])
    np.testing.assert_allclose(hits["uv"][hit], rec[hit, 7:9], rtol=0, atol=1e-6)  # atan2f / acosf: last-bit class


@pytest.mark.parametrize("scene", ["demo_onoff", "demo_path", "csg_example"])
def test_offset_if_visible_vs_reference(worlds, golden_traces, scene):
    tri = golden_traces[f"{scene}_vis_in"]
    vis = worlds[scene].offset_if_visible(tri[:, 0:3], tri[:, 3:6], tri[:, 6:9])
    assert np.array_equal(vis, golden_traces[f"{scene}_vis_out"])


def test_aa_jitter_exact():
    k = np.concatenate([np.arange(0, 64), np.array([10**6, 2**31 - 1, 2**31, 2**33 + 12345, 132_710_399, 2**40 + 7])]).astype(np.uint64)
    u, v = aa_jitter(k)
    ou, ov = O.aa_jitter(k)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)
    pcg = O.PCG()  # serial draws of PCG(42,54): sample k uses draws 2k, 2k+1
    serial = [pcg.random_float() for _ in range(128)]
    assert np.array_equal(u[:64], np.array(serial[0::2], dtype=np.float32))
    assert np.array_equal(v[:64], np.array(serial[1::2], dtype=np.float32))


@pytest.mark.parametrize("scene,w,h,aa,n_rays,roulette,max_depth,seq", [
    ("demo_path", 64, 48, 1, 10, 3, 5, 54),
    ("csg_example", 64, 48, 1, 10, 3, 5, 54),
    ("demo_path", 32, 24, 2, 3, 2, 4, 7),
    ("csg_example", 32, 24, 2, 3, 2, 4, 7),
    ("demo_path", 24, 16, 1, 1, 2, 40, 11),   # n_rays == 1: tail-call frames, deep recursion
])
@pytest.mark.parametrize("exact", [True, False])
def test_path_same_streams_vs_oracle(worlds, scenes, scene, w, h, aa, n_rays, roulette, max_depth, seq, exact):
    """Same per-sample PCG streams on both sides -> the same paths; only ulp-level libm differences remain
    (exact=True: unfused scan arithmetic; exact=False: the default fused-multiply-add scan of path mode)."""
    img, stats = gpu_render(worlds, scenes, scene, "path", w, h, aa, n_rays=n_rays, roulette=roulette, max_depth=max_depth,
                            seq_number=seq, exact=exact)
    cfg = O.make_cfg(w, h, "path", spp_edge=aa, n_rays=n_rays, roulette=roulette, max_depth=max_depth, seq_number=seq,
                     rng_mode=abi.RT_RNG_PER_NODE_STREAMS)
    want, ostats = O.render(scenes[scene], cfg)
    diff = np.abs(img - want).max(axis=-1)
    outliers = diff > 1e-4
    print(f"exact={exact}: {int(outliers.sum())} of {outliers.size} pixels differ by > 1e-4, {int((diff == 0).sum())} identical")
    assert outliers.mean() <= 0.02, f"{outliers.sum()} of {outliers.size} pixels differ (max {diff.max()})"
    # a path that diverges near the root takes its whole subtree (up to ~10^max_depth = 111 111 rays) with it: on these
    # 3 072-sample frames a handful of diverged pixels move the count by several per cent (measured: 11 pixels, 15 %;
    # the full-size C3 / C4 frames of the fused kernels are within 0.2 % of the unfused ones)
    assert abs(int(stats.rays) - int(ostats.rays)) <= (0.08 if exact else 0.20) * ostats.rays
    # exact: only libm-ulp divergences; fused: a few pixels follow another path (other samples of the same estimator)
    assert abs(float(img.mean()) - float(want.mean())) <= (2e-3 if exact else 1e-2) * float(want.mean())


@pytest.mark.parametrize("scene", ["demo_path", "csg_example"])
def test_path_statistical_vs_reference_high_spp(worlds, scenes, golden_images, scene):
    """RMSE bound against the reference's own serial-PCG render at 36 spp (96x72, AA=6).

    Two independent unbiased estimates with per-pixel variance s^2/36 each differ with mean square 2*s^2/36.
    s^2 is measured from a second GPU render with another --seq-number; the bound is 1.25x that expectation,
    plus the means must agree to 1 %.
    """
    ref = golden_images[f"path_{scene}_96x72_aa6_serial"]
    a, _ = gpu_render(worlds, scenes, scene, "path", 96, 72, 6, seq_number=54)
    b, _ = gpu_render(worlds, scenes, scene, "path", 96, 72, 6, seq_number=1234)
    mse_gpu_gpu = float(np.mean((a - b) ** 2))
    mse_gpu_ref = float(np.mean((a - ref) ** 2))
    assert mse_gpu_ref <= 1.25 * mse_gpu_gpu + 1e-12, (mse_gpu_ref, mse_gpu_gpu)
    assert abs(float(a.mean()) - float(ref.mean())) <= 0.01 * float(ref.mean())
    print(f"{scene}: rmse(gpu,ref)={mse_gpu_ref ** 0.5:.5f} rmse(gpu,gpu')={mse_gpu_gpu ** 0.5:.5f} mean={a.mean():.5f}/{ref.mean():.5f}")


def test_row_bands_bit_equal_to_whole_image(worlds, scenes):
    """1-GPU image == image assembled from cyclic row bands of 2 and 3 'ranks' (stream ids depend only on the
    global sample index), for a deterministic mode with AA and for path mode."""
    for mode, kw in (("flat", {}), ("path", dict(n_rays=4, roulette=2, max_depth=3))):
        whole, _ = gpu_render(worlds, scenes, "demo_path", mode, 80, 50, 2, **kw)
        for world_size in (2, 3):
            out = np.zeros_like(whole)
            for rank in range(world_size):
                rows = RowBands.cyclic(50, rank, world_size, band_rows=16)
                part, _ = gpu_render(worlds, scenes, "demo_path", mode, 80, 50, 2, rows=rows, **kw)
                out[rows.image_rows()] = part
            assert np.array_equal(out, whole), (mode, world_size)


def test_render_into_device_tensor_matches_host_path(worlds, scenes):
    import torch
    tr = GpuImageTracer(96, 64, scenes["csg_example"].camera, 2)
    host = tr.fire_all_rays(worlds["csg_example"], RenderMode("flat"))
    out = torch.empty((64, 96, 3), dtype=torch.float32, device="cuda:0")
    tr.fire_all_rays(worlds["csg_example"], RenderMode("flat"), out=out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), host)


def test_error_behaviour(worlds, scenes):
    from raytracer_amd.lib import RtError
    tr = GpuImageTracer(16, 16, scenes["demo_path"].camera, 1)
    with pytest.raises(RtError):
        tr.fire_all_rays(worlds["demo_path"], RenderMode("pointlight", max_mirror_bounces=0))
    with pytest.raises(RtError):
        tr.fire_all_rays(worlds["demo_path"], RenderMode("path", n_rays=0))
    with pytest.raises(ValueError):
        GpuImageTracer(0, 16, scenes["demo_path"].camera)
    bad = scenes["demo_path"]
    import copy
    broken = copy.deepcopy(bad)
    broken.shape_material = broken.shape_material.copy()
    broken.shape_material[0] = 99
    with pytest.raises(RtError):
        DeviceWorld(broken)


# ---- the large-world path kernels (worlds with >= 32 register spheres: spheres in registers, rays through LDS) ----
# sphere counts cover: the smallest worlds that take them (33, 40, 45: one full tile / row and a bit), leftover scanned per
# lane (77 -> 78 = 64 + 14), partly empty last tile / row (100, 90), several waves per workgroup (600: 2, 1100 and 4000: 4)

def _ellipsoid_world(n, seed=5):
    """n random ellipsoids (rotated, non-uniformly scaled: the general affine register form) + floor + sky."""
    from raytracer_amd.scene import SceneBuilder, compose, rotation_x, rotation_z, scaling, translation
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    floor = b.material("diffuse", b.pigment_checkered((0.3, 0.5, 0.1), (0.1, 0.2, 0.5), 4))
    sky = b.material("diffuse", b.pigment_uniform((0, 0, 0)), b.pigment_uniform((0.2, 0.3, 1.0)))
    mats = [b.material("diffuse", b.pigment_uniform(tuple(rng.uniform(0.2, 0.9, 3)))) for _ in range(6)]
    mats.append(b.material("specular", b.pigment_uniform((0.7, 0.7, 0.7))))
    b.plane(compose(translation([0, 0, -1])), floor)
    b.sphere(compose(scaling([40, 40, 40])), sky)
    for _ in range(n):
        c = rng.uniform([-1, -3, -0.8], [5, 3, 1.5])
        sc = rng.uniform(0.08, 0.3, 3)
        t = compose(translation(c), rotation_z(rng.uniform(0, 3)), rotation_x(rng.uniform(0, 3)), scaling(sc))
        b.sphere(t, mats[int(rng.integers(len(mats)))])
    b.set_camera("perspective", compose(translation([-3, 0, 0.3])), None, 1.0)
    return b.build()


@pytest.mark.parametrize("kind,n", [("iso", 600), ("iso", 1100), ("general", 700), ("iso", 77), ("iso", 100), ("general", 90), ("iso", 4000),
                                    ("iso", 33), ("iso", 40), ("general", 45)])
def test_path_wide_kernel_same_streams_vs_oracle(kind, n):
    """Same per-sample streams on the wide kernel and on the CPU oracle -> the same paths (FMA / world-space
    sphere arithmetic moves t by ulps, so a few paths diverge at discontinuities)."""
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(n)[0] if kind == "iso" else _ellipsoid_world(n)
    world = DeviceWorld(scene)
    w, h, aa, kw = 40, 24, 1, dict(n_rays=3, roulette=2, max_depth=3, seq_number=9)
    tr = GpuImageTracer(w, h, scene.camera, aa)
    img = tr.fire_all_rays(world, RenderMode("path", **kw))
    stats = tr.last_stats
    exact = tr.fire_all_rays(world, RenderMode("path", exact=True, **kw))  # the per-lane kernel, unfused arithmetic
    objsp = tr.fire_all_rays(world, RenderMode("path", object_space_spheres=True, **kw))  # wide kernel, object-space form
    world.close()
    want, ostats = O.render(scene, O.make_cfg(w, h, "path", spp_edge=aa, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))
    for name, got in (("wide", img), ("wide-object-space", objsp), ("exact", exact)):
        diff = np.abs(got - want).max(axis=-1)
        outliers = diff > 1e-4
        print(f"{kind} n={n} {name}: {int(outliers.sum())} of {outliers.size} pixels differ by > 1e-4 (max {diff.max():.3g})")
        # the unfused kernel follows the oracle's paths except for libm-ulp effects; the FMA kernels also leave them where
        # the filter ranks two near-equal candidates the other way or a scattered direction moves by an ulp (measured:
        # n=600 0.9 %, 1100 1.8 %, 4000 2.4 %; before the winner's record was recomputed in the reference's arithmetic: 2 / 6 / 17 %)
        limit = 0.03 if name == "exact" else (0.04 if n < 2000 else 0.06)
        assert outliers.mean() <= limit
        # diverged pixels are other samples of the same estimator: the 960-pixel mean moves by ~sqrt(#diverged) * 0.3 / 960
        assert abs(float(got.mean()) - float(want.mean())) <= (0.01 if n < 2000 else 0.03) * float(want.mean())
    assert abs(int(stats.rays) - int(ostats.rays)) <= 0.08 * ostats.rays


def test_path_wide_kernel_row_bands_and_determinism():
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(512)[0]
    world = DeviceWorld(scene)
    kw = dict(n_rays=3, roulette=2, max_depth=3)
    tr = GpuImageTracer(48, 40, scene.camera, 2)
    whole = tr.fire_all_rays(world, RenderMode("path", **kw))
    again = tr.fire_all_rays(world, RenderMode("path", **kw))
    assert np.array_equal(whole, again)  # sample -> lane assignment is dynamic, the image is not
    out = np.zeros_like(whole)
    for rank in range(3):
        rows = RowBands.cyclic(40, rank, 3, band_rows=16)
        out[rows.image_rows()] = tr.fire_all_rays(world, RenderMode("path", **kw), rows)
    world.close()
    assert np.array_equal(out, whole)


# ---- optional BVH mode (RT_FLAG_ACCEL_BVH): same hits as the linear scan, fewer tests ----

@pytest.mark.parametrize("kind,n", [("iso", 2), ("iso", 5), ("iso", 70), ("iso", 600), ("general", 300), ("iso", 4000), ("iso", 5000)])
def test_bvh_mode_finds_the_linear_scan_winner(kind, n):
    """With the unfused arithmetic every sphere test gives the same t through the hierarchy as in the linear scan and
    candidates compare as (t, object index), so the two path renders must be IDENTICAL, ray count included.
    n = 2, 5: a root that is a leaf / one split; n = 5000: the heap outgrows LDS (deep levels read from HBM); the sky
    sphere of the synthetic scenes is one of the spheres kept out of the tree."""
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(n)[0] if kind == "iso" else _ellipsoid_world(n)
    world = DeviceWorld(scene)
    kw = dict(n_rays=3, roulette=2, max_depth=3, seq_number=9)
    tr = GpuImageTracer(48, 30, scene.camera, 2)
    linear = tr.fire_all_rays(world, RenderMode("path", exact=True, **kw))
    rays_linear = int(tr.last_stats.rays)
    bvh = tr.fire_all_rays(world, RenderMode("path", exact=True, bvh=True, **kw))
    rays_bvh = int(tr.last_stats.rays)
    fast = tr.fire_all_rays(world, RenderMode("path", bvh=True, **kw))
    world.close()
    assert np.array_equal(bvh, linear) and rays_bvh == rays_linear
    # the hierarchy against the CPU oracle (same per-ray streams), at the bar the exact linear kernel meets in
    # test_path_wide_kernel_same_streams_vs_oracle (there 1.4 % of SAMPLES at n = 4000; a pixel here is 4 samples): <= 3 % /
    # 8 % of pixels leave the oracle's paths (libm ulps in a scattered direction), mean within 1 %
    want, ostats = O.render(scene, O.make_cfg(48, 30, "path", spp_edge=2, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))
    odiff = np.abs(bvh - want).max(axis=-1)
    print(f"{kind} n={n} bvh exact vs oracle: {int((odiff > 1e-4).sum())} of {odiff.size} pixels differ by > 1e-4")
    assert (odiff > 1e-4).mean() <= (0.03 if n < 2000 else 0.08)
    assert abs(float(bvh.mean()) - float(want.mean())) <= 0.01 * float(want.mean())
    assert abs(rays_bvh - int(ostats.rays)) <= 0.08 * ostats.rays
    odiff = np.abs(fast - want).max(axis=-1)
    assert (odiff > 1e-4).mean() <= (0.25 if n < 2000 else 0.60)
    assert abs(float(fast.mean()) - float(want.mean())) <= 0.03 * float(want.mean())
    # the fused arithmetic leaves the unfused paths at some hit decisions (per SAMPLE: 2 % at n=600, 17 % at n=4000, see
    # above; a pixel here is 4 samples); the estimator is the same
    diff = np.abs(fast - linear).max(axis=-1)
    assert (diff > 1e-4).mean() <= (0.25 if n < 2000 else 0.60)
    assert abs(float(fast.mean()) - float(linear.mean())) <= 0.03 * float(linear.mean())
