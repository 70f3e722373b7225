"""The cluster-culled tile kernel (raytracer_amd/csrc/rt_tile_cull.h) against the same filter run over every tile
(RT_FLAG_SCAN_ALL_TILES -> k_render_path_tile) and against the CPU oracle.

A tile of 32 spheres is multiplied only with the rays whose line meets the bounding sphere of its cluster; a ray whose line
misses the bound misses the spheres, and a product element does not depend on which rays share its block — so the keys, the
winners, the random streams and therefore the FRAMES must be identical bit for bit, ray counts included.  That is a hard check
of the bounds' conservativeness (one missed candidate anywhere in a frame's ~10^8 rays changes the hash).

Equal distances: the reference keeps the first of equally distant objects (strict `<` in World::ray_intersection,
include/shapes.hpp:547-561).  The clusters reorder the spheres in space, so a key carries the sphere's position in OBJECT
order; a world of exact duplicates (same centre and radius, different emission) checks it hit by hit against the oracle.
"""
import hashlib

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import SceneBuilder, compose, scaling, translation
from raytracer_amd.synth import synthetic_scene
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode

pytestmark = pytest.mark.gpu


def sha(img):
    return hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()


# 1616 spheres: 51 tiles, the last one half empty and a last cluster group that is not full; 2048: 64 tiles; 4096: the
# headline world (128 tiles); 9000: more clusters than one group of 32 per wave holds (282 tiles)
@pytest.mark.parametrize("n,w,h,aa", [(1616, 160, 90, 2), (2048, 160, 90, 2), (4096, 192, 108, 2), (9000, 96, 54, 1)])
def test_culled_frame_is_the_scan_all_frame(n, w, h, aa):
    scene, _ = synthetic_scene(n)
    world = DeviceWorld(scene)
    tr = GpuImageTracer(w, h, scene.camera, aa)
    kw = dict(n_rays=10, roulette=3, max_depth=5, seq_number=54)
    culled = tr.fire_all_rays(world, RenderMode("path", **kw))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_tile_cull<")
    rays_culled = int(tr.last_stats.rays)
    full = tr.fire_all_rays(world, RenderMode("path", scan_all_tiles=True, **kw))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_tile<")
    world.close()
    assert rays_culled == int(tr.last_stats.rays)
    assert sha(culled) == sha(full)
    assert float(culled.mean()) > 0.05  # a real picture


def test_small_worlds_keep_the_plain_tile_kernel():
    scene, _ = synthetic_scene(512)  # 16 tiles: bounds of so few clusters cull little
    world = DeviceWorld(scene)
    tr = GpuImageTracer(64, 36, scene.camera, 1)
    tr.fire_all_rays(world, RenderMode("path", n_rays=2, roulette=2, max_depth=2, seq_number=1))
    world.close()
    assert tr.last_stats.kernel.decode().startswith("k_render_path_tile<")


def duplicates_world(n_distinct, copies, seed=5):
    """Every sphere `copies` times in the same place, the copies far apart in object order; each object emits its own number and
    reflects nothing, so a flat-like path image (max_depth = 0) is the table of first hits."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()

    def mat(obj_id):
        return b.material("diffuse", b.pigment_uniform((0.0, 0.0, 0.0)), b.pigment_uniform((obj_id, obj_id, obj_id)))

    b.plane(compose(translation([0, 0, -2])), mat(1))
    b.sphere(compose(scaling([50, 50, 50])), mat(2))
    spheres = []
    for _ in range(n_distinct):
        r = float(np.float32(rng.uniform(0.08, 0.2)))
        c = [float(np.float32(x)) for x in rng.uniform([-2, -6, -2 + r], [8, 6, 2])]
        spheres.append((c, r))
    k = 0
    for _ in range(copies):
        for c, r in spheres:
            b.sphere(compose(translation(c), scaling([r, r, r])), mat(3 + k))
            k += 1
    b.set_camera("perspective", compose(translation([-3, 0, 0])), None, 1.0)
    return b.build()


def test_equal_distances_keep_the_first_object():
    n_distinct, copies = 1024, 2  # 2048 spheres = 64 tiles; object 3 + i and object 3 + 1024 + i coincide
    scene = duplicates_world(n_distinct, copies)
    w, h = 192, 108
    kw = dict(n_rays=1, roulette=5, max_depth=0, seq_number=3)
    world = DeviceWorld(scene)
    tr = GpuImageTracer(w, h, scene.camera, 1)
    culled = tr.fire_all_rays(world, RenderMode("path", **kw))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_tile_cull<")
    full = tr.fire_all_rays(world, RenderMode("path", scan_all_tiles=True, **kw))
    exact = tr.fire_all_rays(world, RenderMode("path", exact=True, **kw))
    world.close()
    want = O.render(scene, O.make_cfg(w, h, "path", spp_edge=1, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))[0]
    assert np.array_equal(exact, want)  # the unfused kernel is the oracle
    ids = culled[..., 0]
    assert np.array_equal(ids, np.round(ids))
    on_spheres = ids >= 3
    assert on_spheres.mean() > 0.2
    assert np.all(ids[on_spheres] < 3 + n_distinct)  # never the later copy
    assert np.array_equal(culled, full)
    # against the oracle: the filters may rank two DIFFERENT near-equal spheres differently (tests/test_hit_level.py prices that)
    assert (culled[..., 0] != want[..., 0]).mean() <= 0.005


def test_world_with_csg_objects_runs_the_culled_kernel_with_the_evaluators():
    """1536 random spheres around a lens (the intersection of two spheres) and a holed ball (a difference): the CSG = true build of the
    culled kernel.  Same frame as scanning all tiles; first hits against the oracle."""
    rng = np.random.default_rng(17)
    b = SceneBuilder()

    def mat(obj_id):
        return b.material("diffuse", b.pigment_uniform((0.0, 0.0, 0.0)), b.pigment_uniform((obj_id, obj_id, obj_id)))

    b.plane(compose(translation([0, 0, -2])), mat(1))
    b.sphere(compose(scaling([50, 50, 50])), mat(2))
    left = b.sphere(compose(translation([0.0, -0.25, 0.0]), scaling([0.6, 0.6, 0.6])), mat(3), norender=True)
    right = b.sphere(compose(translation([0.0, 0.25, 0.0]), scaling([0.6, 0.6, 0.6])), mat(4), norender=True)
    b.csg(left, right, "intersection")
    ball = b.sphere(compose(translation([1.0, 1.5, 0.5]), scaling([0.7, 0.7, 0.7])), mat(5), norender=True)
    hole = b.sphere(compose(translation([0.5, 1.5, 0.5]), scaling([0.5, 0.5, 0.5])), mat(6), norender=True)
    b.csg(ball, hole, "difference")
    for k in range(1536):
        r = float(np.float32(rng.uniform(0.05, 0.15)))
        c = [float(np.float32(x)) for x in rng.uniform([-2, -6, -2 + r], [8, 6, 2])]
        b.sphere(compose(translation(c), scaling([r, r, r])), mat(7 + k))
    b.set_camera("perspective", compose(translation([-3, 0, 0])), None, 1.0)
    scene = b.build()
    w, h = 192, 108
    world = DeviceWorld(scene)
    tr = GpuImageTracer(w, h, scene.camera, 1)
    kw = dict(n_rays=1, roulette=5, max_depth=0, seq_number=3)
    culled = tr.fire_all_rays(world, RenderMode("path", **kw))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_tile_cull<")
    full = tr.fire_all_rays(world, RenderMode("path", scan_all_tiles=True, **kw))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_tile<")
    # and a real path frame (the streams go through the CSG surfaces too)
    path_kw = dict(n_rays=3, roulette=2, max_depth=3, seq_number=9)
    a = tr.fire_all_rays(world, RenderMode("path", **path_kw))
    rays_a = int(tr.last_stats.rays)
    c = tr.fire_all_rays(world, RenderMode("path", scan_all_tiles=True, **path_kw))
    assert rays_a == int(tr.last_stats.rays) and np.array_equal(a, c)
    world.close()
    assert np.array_equal(culled, full)
    want = O.render(scene, O.make_cfg(w, h, "path", spp_edge=1, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))[0]
    ids = culled[..., 0]
    assert ((ids == 3) | (ids == 4)).sum() > 50 and ((ids == 5) | (ids == 6)).sum() > 50  # both CSG objects are in the picture
    assert (ids != want[..., 0]).mean() <= 0.005
