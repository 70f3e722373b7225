"""Hit-level check of the FAST closest-hit scans (the tile kernel with its MFMA filter, the register-row kernel in the object-space form, and the BVH kernel)
against the unfused scan and the CPU oracle, ray by ray — not through image statistics.

Trick: path mode with n_rays = 1, max-depth = 1 and no Russian roulette traces exactly two rays per sample, and with
every object a mirror of reflectance (0, 1, 0) that EMITS its own object number (id, id, id) the sample's value is
    red = id(first hit),   green = id(first hit) + id(second hit)
— small integers, exact in the 64-bit fixed-point accumulators and in binary32 — so the image IS the table of winners
of World::ray_intersection (reference include/shapes.hpp:547-561) for a camera ray (tmin 1e-5) and a mirror ray leaving
a surface (SpecularBRDF::scatter_ray, include/materials.hpp:252-261).

Bars:
  * unfused kernel == oracle: identical ids everywhere (no transcendental function is involved);
  * fused kernels vs unfused: the first-hit id may differ only where the decision is within rounding — one of the two
    candidates is grazed (|reduced discriminant| within the fp32 error bound of its terms) or both are hit at distances
    that agree to 1e-5 relative — checked per differing pixel in binary64; and such pixels are < 0.5 % of the frame;
  * second-hit ids (given the same first hit) differ on < 1 % of pixels (the mirror direction itself moves by an ulp),
    not counting rays that re-hit their own sphere on one side only (a last-bit matter of the hit point, see below).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import SceneBuilder, compose, scaling, translation
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode

pytestmark = pytest.mark.gpu

W, H = 192, 108


def id_world(n, seed=11, shift=(0.0, 0.0, 0.0)):
    """`shift` moves the whole world and its camera away from the origin (binary32 coordinates throughout)."""
    rng = np.random.default_rng(seed)
    shift = np.asarray(shift, dtype=np.float64)

    def at(p):
        return [float(np.float32(x)) for x in np.asarray(p, dtype=np.float64) + shift]

    b = SceneBuilder()
    geo = []  # per object: None (not a small sphere) or (centre, radius)

    def mat(obj_id, reflect=(0.0, 1.0, 0.0)):
        return b.material("specular", b.pigment_uniform(reflect), b.pigment_uniform((obj_id, obj_id, obj_id)))

    b.plane(compose(translation(at([0, 0, -2]))), mat(1))
    geo.append(None)
    b.sphere(compose(translation(at([0, 0, 0])), scaling([50, 50, 50])), mat(2, (0.0, 0.0, 0.0)))  # the sky reflects nothing: the path ends there
    geo.append(None)
    for k in range(n):
        r = float(np.float32(rng.uniform(0.05, 0.15)))
        c = at([float(np.float32(x)) for x in rng.uniform([-2, -6, -2 + r], [8, 6, 2])])
        b.sphere(compose(translation(c), scaling([r, r, r])), mat(3 + k))
        geo.append((np.array(c, dtype=np.float64), r))
    b.set_camera("perspective", compose(translation(at([-3, 0, 0]))), None, 1.0)
    return b.build(), geo


def ids_of(img):
    first = img[..., 0]
    second = img[..., 1] - img[..., 0]
    assert np.array_equal(first, np.round(first)) and np.array_equal(second, np.round(second)) and np.all(img[..., 2] == first)
    return first.astype(np.int64), second.astype(np.int64)


def camera_rays(scene):
    cam = scene.camera.to_abi()
    o = np.zeros((H, W, 3))
    d = np.zeros((H, W, 3))
    r = abi.rt_ray()
    for row in range(H):
        for col in range(W):
            O.dll().rto_camera_ray(C.byref(cam), W, H, col, row, 0.5, 0.5, C.byref(r))
            o[row, col], d[row, col] = r.origin[:], r.direction[:]
    return o, d


def within_rounding(geo, obj_a, obj_b, o, d):
    """Is the choice between objects a and b (ids, 1-based) for the ray (o, d) a matter of fp32 rounding?"""
    ts = []
    for obj in (obj_a, obj_b):
        g = geo[obj - 1]
        if g is None:
            ts.append(None)
            continue
        c, r = g
        oc = o - c
        a, bq, cc = d @ d, oc @ d, oc @ oc - r * r
        disc = bq * bq - a * cc
        if abs(disc) <= 32 * 2.0 ** -24 * (bq * bq + a * (oc @ oc)):  # grazed: the sign of disc is within the error of its terms
            return True
        ts.append((-bq - np.sqrt(disc)) / a if disc > 0 else None)
    return ts[0] is not None and ts[1] is not None and abs(ts[0] - ts[1]) <= 1e-5 * abs(ts[1])


@pytest.mark.parametrize("n", [600, 1000, 2500, 4000])  # 19, 32, 79 and 125 sphere tiles: W = 2 with a partial pass, W = 2 full, W = 4 with an uneven last pass, W = 4
def test_fast_scans_pick_the_reference_winner(n):
    scene, geo = id_world(n)
    kw = dict(n_rays=1, roulette=5, max_depth=1, seq_number=3)
    world = DeviceWorld(scene)
    tr = GpuImageTracer(W, H, scene.camera, 1)
    renders = {}
    for name, mode in (("unfused", RenderMode("path", exact=True, **kw)), ("wide-world-space", RenderMode("path", **kw)),
                       ("wide-object-space", RenderMode("path", object_space_spheres=True, **kw)), ("bvh", RenderMode("path", bvh=True, **kw)),
                       ("bvh-unfused", RenderMode("path", bvh=True, exact=True, **kw))):
        renders[name] = ids_of(tr.fire_all_rays(world, mode))
        if name == "wide-world-space":
            assert tr.last_stats.kernel.decode().startswith("k_render_path_tile")
    world.close()
    want = ids_of(O.render(scene, O.make_cfg(W, H, "path", spp_edge=1, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))[0])
    ref1, ref2 = renders["unfused"]
    assert np.array_equal(ref1, want[0]) and np.array_equal(ref2, want[1])          # the unfused kernel IS the oracle, hit by hit
    assert np.array_equal(renders["bvh-unfused"][0], ref1) and np.array_equal(renders["bvh-unfused"][1], ref2)
    assert len(np.unique(ref1)) > min(n, 300) // 3                                   # the frame really sees many objects
    o, d = camera_rays(scene)
    for name in ("wide-world-space", "wide-object-space", "bvh"):
        got1, got2 = renders[name]
        bad = np.argwhere(got1 != ref1)
        print(f"n={n} {name}: {len(bad)} first-hit ids differ, {int(((got2 != ref2) & (got1 == ref1)).sum())} second-hit ids differ, of {W * H}")
        assert len(bad) <= 0.005 * W * H
        for row, col in bad:
            assert within_rounding(geo, int(got1[row, col]), int(ref1[row, col]), o[row, col], d[row, col]), \
                f"{name}: pixel ({col},{row}) picks object {got1[row, col]} where the reference picks {ref1[row, col]}, not a rounding case"
        # second hits: a mirror ray (tmin 1e-5) re-hits ITS OWN sphere when rounding left its origin a hair inside.  In the
        # reference's arithmetic that is common for small far spheres — the root t = (-b - sqrt(b^2 - ac)) / a cancels
        # ~4 digits, the hit point is off by ~1e-4 radii along the ray and the exit root of the mirror ray lands right
        # at tmin = 1e-5: the oracle itself re-hits on 22 % of the reflections off the small spheres here (11 % of all
        # pixels).  Which side of tmin it falls is a last-bit matter — and the brightness of a crowded scene depends on it
        # (each re-hit costs a factor rho) — which is why the large-world kernels recompute the winner's hit record and
        # test the sphere a ray leaves in the reference's own operation order (rt_device.h self_sphere_exact): with that
        # the fused kernels reproduce these decisions too (measured: 0-1 flips of 20 736; 900-2 700 before).
        second = (got2 != ref2) & (got1 == ref1)
        self_hit = second & ((got2 == got1) | (ref2 == ref1))
        print(f"  of the second-hit differences {int(self_hit.sum())} are own-sphere re-hits on one side")
        assert (second & ~self_hit).mean() <= 0.005 and self_hit.mean() <= 0.01


def test_world_far_from_the_origin_takes_the_object_space_kernel():
    """The world-space filter's rounding error grows with |O|^2 + |C|^2 (rt_api.hip world_filter_ok): a world 400 units from
    the origin with spheres of radius 0.05-0.15 must not use it.  The object-space kernel it takes instead names the unfused
    kernel's winners (which IS the oracle, see above) on all but a few silhouette pixels, at this distance too."""
    scene, _ = id_world(600, shift=(300.0, -250.0, 60.0))
    kw = dict(n_rays=1, roulette=5, max_depth=1, seq_number=3)
    world = DeviceWorld(scene)
    tr = GpuImageTracer(W, H, scene.camera, 1)
    ref1, ref2 = ids_of(tr.fire_all_rays(world, RenderMode("path", exact=True, **kw)))
    got1, got2 = ids_of(tr.fire_all_rays(world, RenderMode("path", **kw)))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_wide<GEN")
    world.close()
    want = ids_of(O.render(scene, O.make_cfg(W, H, "path", spp_edge=1, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))[0])
    assert np.array_equal(ref1, want[0]) and np.array_equal(ref2, want[1])
    assert len(np.unique(ref1)) > 100
    print(f"far world: {int((got1 != ref1).sum())} first-hit ids differ, {int(((got2 != ref2) & (got1 == ref1)).sum())} second-hit ids differ, of {W * H}")
    assert (got1 != ref1).mean() <= 0.005
    assert ((got2 != ref2) & (got1 == ref1)).mean() <= 0.02
