"""Known answers of the reference's own unit tests, restated against the oracle (CPU) and, for the
hot-path operators, against the HIP library through the C-ABI (`-m gpu`).

Only the VALUES come from the reference (inputs and expected outputs of test/*.cpp, cited per test);
the tolerance is the reference's DEFAULT_ERROR_TOLERANCE = 1e-5 (include/colors.hpp:40) unless stated.
This is what pins the oracle beyond the rendered golden images: SURVEY.md §8c item 6.
"""
import math

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import (Camera, SceneBuilder, Transformation, compose, rotation_x, rotation_y, rotation_z,
                                 scaling, translation)

TOL = 1e-5
PI = np.float32(math.pi)
VEC_X, VEC_Y, VEC_Z = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)


def neg(v):
    return tuple(-x for x in v)


def close(a, b, tol=TOL):
    return np.all(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) < tol)


def neutral(b: SceneBuilder) -> int:
    """make_neutral_material(): diffusive black BRDF, black emission (include/materials.hpp)."""
    return b.material("diffuse", b.pigment_uniform((0, 0, 0)))


def one_shape(kind: str, t: Transformation):
    b = SceneBuilder()
    m = neutral(b)
    ref = getattr(b, kind)(t, m)
    return b.build(), ref


def ray_at(r: abi.rt_ray, t):
    return np.array(r.origin[:]) + t * np.array(r.direction[:])


def check_hit(h, point, normal, uv, t):
    assert h is not None
    assert close(h.point[:], point) and close(h.normal[:], normal) and close(h.uv[:], uv) and abs(h.t - t) < TOL


# ---------------------------------------------------------------------------------------------------
# test/random_test.cpp:18-29
# ---------------------------------------------------------------------------------------------------
def test_pcg_default_sequence():
    pcg = O.PCG()
    assert pcg.state == 1753877967969059832 and pcg.inc == 109
    for expected in (2707161783, 2068313097, 3122475824, 2211639955, 3215226955, 3421331566):
        assert pcg.random() == expected


def test_pcg_advance_equals_stepping():
    """pcg_advance is ours (not in the reference): it must equal n plain steps of the reference generator."""
    a, b = O.PCG(42, 54), O.PCG(42, 54)
    for _ in range(1000):
        a.random()
    b.advance(1000)
    assert a.state == b.state and a.random() == b.random()


# ---------------------------------------------------------------------------------------------------
# test/geometry_test.cpp:114-134 (Transformation * Vec / Point / Normal) and :76-112 (composition)
# ---------------------------------------------------------------------------------------------------
def test_transformation_applied_to_vec_point_normal():
    fwd = [1, 2, 3, 5, 6, 7, 9, 9, 8, 4, 8, 7]
    inv = [-3.75, 2.75, -1, 5.75, -4.75, 2, -2.25, 2.25, -1, 0, 1, -2]
    assert close(O.hom_apply(fwd, inv, "vec", (1, 2, 3)), (14, 38, 51))
    assert close(O.hom_apply(fwd, inv, "point", (1, 2, 3)), (18, 46, 58))
    assert close(O.hom_apply(fwd, inv, "normal", (3, 2, 4)), (-8.75, 7.75, -3))


def test_transformation_composition():
    # T1 is the matrix of the test above it in the reference file; the product's expected value is at :104-109
    t1 = Transformation([1, 2, 3, 5, 6, 7, 9, 9, 8, 4, 8, 7], [-3.75, 2.75, -1, 5.75, -4.75, 2, -2.25, 2.25, -1, 0, 1, -2])
    t2 = Transformation([2, 6, 4, 0, 3, 5, 1, 2, 1, 3, 2, 6], [-1.75, 0.5, 4.5, 1.25, -0.5, -2.5, -0.75, 0.5, 1.5, -22.75, 12.25, -7.75])
    p = t1 * t2
    assert close(p.fwd, [5, 18, 17, 17, 62, 57, 26, 97, 89, 29, 77, 100])
    assert close(p.inv, [-0.6875, 2.9375, -1.75, -1.9375, 0.1875, 0.25, 2.3125, -1.0625, 0.25, -31.25, 16.75, -10.25])


# ---------------------------------------------------------------------------------------------------
# test/cameras_test.cpp:54-116 (cameras), :135-167 (ImageTracer::fire_ray), :170-193 (stratified sampling)
# ---------------------------------------------------------------------------------------------------
def cam_fire(cam: Camera, u, v):
    # Camera::fire_ray(u, v) == ImageTracer::fire_ray on a 1x1 image with u_pixel = u, v_pixel = 1 - v
    return O.camera_ray(cam, 1, 1, 0, 0, u, 1.0 - v)


@pytest.mark.parametrize("kind", [abi.RT_CAMERA_ORTHOGONAL, abi.RT_CAMERA_PERSPECTIVE])
def test_camera_corner_rays(kind):
    cam = Camera(kind, Transformation(), 2.0, 1.0)
    rays = [cam_fire(cam, 0, 0), cam_fire(cam, 1, 0), cam_fire(cam, 0, 1), cam_fire(cam, 1, 1)]
    d0 = np.array(rays[0].direction[:])
    for r in rays[1:]:
        cross = np.cross(d0, np.array(r.direction[:]))
        if kind == abi.RT_CAMERA_ORTHOGONAL:
            assert abs(np.dot(cross, cross)) < TOL          # parallel rays
        else:
            assert close(r.origin[:], rays[0].origin[:])    # common origin
    assert close(ray_at(rays[0], 1), (0, 2, -1))
    assert close(ray_at(rays[1], 1), (0, -2, -1))
    assert close(ray_at(rays[2], 1), (0, 2, 1))
    assert close(ray_at(rays[3], 1), (0, -2, 1))


def test_camera_transformations():
    t = translation(neg((0, 2, 0))) * rotation_z(np.float32(0.5) * PI)
    assert close(ray_at(cam_fire(Camera(abi.RT_CAMERA_ORTHOGONAL, t, 1.0), 0.5, 0.5), 1), (0, -2, 0))
    assert close(ray_at(cam_fire(Camera(abi.RT_CAMERA_PERSPECTIVE, t, 1.0, 1.0), 0.5, 0.5), 1), (0, -2, 0))
    t3 = translation((0, 0, -3)) * rotation_y(np.float32(0.5) * PI)
    assert close(ray_at(cam_fire(Camera(abi.RT_CAMERA_PERSPECTIVE, t3, 1.0, 1.0), 0.5, 0.5), 1), (0, 0, -3))


def test_image_tracer_uv_submapping_and_orientation():
    cam = Camera(abi.RT_CAMERA_PERSPECTIVE, Transformation(), 2.0, 1.0)
    r1 = O.camera_ray(cam, 4, 2, 0, 0, 2.5, 1.5)
    r2 = O.camera_ray(cam, 4, 2, 2, 1)
    assert close(r1.origin[:], r2.origin[:]) and close(r1.direction[:], r2.direction[:])
    assert close(ray_at(O.camera_ray(cam, 4, 2, 0, 0, 0.0, 0.0), 1), (0, 2, 1))      # issue #4 regression
    assert close(ray_at(O.camera_ray(cam, 4, 2, 3, 1, 1.0, 1.0), 1), (0, -2, -1))


def test_stratified_sampling_lands_inside_the_pixel():
    """cameras_test.cpp:170-193: 1x1 image, 10 samples per edge -> 100 rays, all inside the screen."""
    cam = Camera(abi.RT_CAMERA_ORTHOGONAL, Transformation(), 1.0)
    u, v = O.aa_jitter(np.arange(100, dtype=np.uint64))
    assert np.all((u >= 0) & (u < 1) & (v >= 0) & (v < 1))
    spp = 10
    k = 0
    for i in range(spp):
        for j in range(spp):
            r = O.camera_ray(cam, 1, 1, 0, 0, (i + float(u[k])) / spp, (j + float(v[k])) / spp)
            p = ray_at(r, 1)
            assert abs(p[0]) < TOL and -1 < p[1] < 1 and -1 < p[2] < 1
            k += 1
    # colour normalisation: a constant-colour world gives that colour back after the mean over 100 samples
    b = SceneBuilder()
    m = b.material("diffuse", b.pigment_uniform((0, 0, 0)), b.pigment_uniform((1, 2, 3)))
    b.plane(translation((5, 0, 0)) * rotation_y(-PI / np.float32(2)), m)
    img, st = O.render(b.build(), O.make_cfg(1, 1, "flat", spp_edge=10), cam)
    assert st.samples == 100 and close(img[0, 0], (1, 2, 3))


# ---------------------------------------------------------------------------------------------------
# test/shapes_test.cpp:29-158 (Sphere)
# ---------------------------------------------------------------------------------------------------
def test_sphere_outer_and_inner_hits():
    s, ref = one_shape("sphere", Transformation())
    check_hit(O.object_hit(s, ref, O.mk_ray((0, 0, 2), neg(VEC_Z))), (0, 0, 1), VEC_Z, (0, 0), 1)
    check_hit(O.object_hit(s, ref, O.mk_ray((3, 0, 0), neg(VEC_X))), (1, 0, 0), VEC_X, (0, 0.5), 2)
    check_hit(O.object_hit(s, ref, O.mk_ray((0, 0, 0), VEC_X)), (1, 0, 0), neg(VEC_X), (0, 0.5), 1)


def test_sphere_translation():
    s, ref = one_shape("sphere", translation((10, 0, 0)))
    check_hit(O.object_hit(s, ref, O.mk_ray((10, 0, 2), neg(VEC_Z))), (10, 0, 1), VEC_Z, (0, 0), 1)
    check_hit(O.object_hit(s, ref, O.mk_ray((13, 0, 0), neg(VEC_X))), (11, 0, 0), VEC_X, (0, 0.5), 2)
    assert O.object_hit(s, ref, O.mk_ray((0, 0, 2), neg(VEC_Z))) is None
    assert O.object_hit(s, ref, O.mk_ray((-10, 0, 2), neg(VEC_Z))) is None


def test_sphere_normals_under_scaling_and_flipping():
    s, ref = one_shape("sphere", scaling((2, 1, 1)))
    h = O.object_hit(s, ref, O.mk_ray((1, 1, 0), (-1, -1, 0)))
    n = np.array(h.normal[:])
    assert close(n / np.linalg.norm(n), np.array([1, 4, 0]) / math.sqrt(17))
    s, ref = one_shape("sphere", scaling((1, -1, 1)))
    assert close(O.object_hit(s, ref, O.mk_ray((0, 2, 0), neg(VEC_Y))).normal[:], VEC_Y)


@pytest.mark.parametrize("origin,direction,uv", [
    ((2, 0, 0), neg(VEC_X), (0, 0.5)), ((0, 2, 0), neg(VEC_Y), (0.25, 0.5)), ((-2, 0, 0), VEC_X, (0.5, 0.5)),
    ((0, -2, 0), VEC_Y, (0.75, 0.5)), ((2, 0, 0.5), neg(VEC_X), (0, 1 / 3)), ((2, 0, -0.5), neg(VEC_X), (0, 2 / 3))])
def test_sphere_surface_coordinates(origin, direction, uv):
    s, ref = one_shape("sphere", Transformation())
    assert close(O.object_hit(s, ref, O.mk_ray(origin, direction)).uv[:], uv)


# ---------------------------------------------------------------------------------------------------
# test/shapes_test.cpp:172-233 (Plane)
# ---------------------------------------------------------------------------------------------------
def test_plane_hits():
    s, ref = one_shape("plane", Transformation())
    check_hit(O.object_hit(s, ref, O.mk_ray((0, 0, 1), neg(VEC_Z))), (0, 0, 0), VEC_Z, (0, 0), 1)
    for d in (VEC_Z, VEC_X, VEC_Y):
        assert O.object_hit(s, ref, O.mk_ray((0, 0, 1), d)) is None


def test_plane_rotation():
    s, ref = one_shape("plane", rotation_y(PI / np.float32(2)))
    check_hit(O.object_hit(s, ref, O.mk_ray((1, 0, 0), neg(VEC_X))), (0, 0, 0), VEC_X, (0, 0), 1)
    for d in (VEC_X, VEC_Y, VEC_Z):
        assert O.object_hit(s, ref, O.mk_ray((1, 0, 0), d)) is None


def test_plane_surface_coordinates_are_periodic():
    s, ref = one_shape("plane", Transformation())
    assert close(O.object_hit(s, ref, O.mk_ray((0.25, 0.75, 1), neg(VEC_Z))).uv[:], (0.25, 0.75))
    assert close(O.object_hit(s, ref, O.mk_ray((4.25, 7.75, 1), neg(VEC_Z))).uv[:], (0.25, 0.75))


# ---------------------------------------------------------------------------------------------------
# test/shapes_test.cpp:239-393 (CSGObject)
# ---------------------------------------------------------------------------------------------------
CSG_RAYS = [((-2, 0, 0), VEC_X), ((0, 0, -2), VEC_Z), ((1, 0, -2), VEC_Z)]
CSG_EXPECTED = {
    "union": [[1, 2, 3, 4], [1, 3], [1, 3]],
    "intersection": [[2, 3], [], []],
    "difference": [[1, 2], [1, 3], []],
    "fusion": [[1, 4], [1, 3], [1, 3]],
}


def two_sphere_csg(op):
    b = SceneBuilder()
    m = neutral(b)
    s1 = b.sphere(Transformation(), m, norender=True)
    s2 = b.sphere(translation(VEC_X), m, norender=True)
    ref = b.csg(s1, s2, op)
    return b.build(), ref


@pytest.mark.parametrize("op", list(CSG_EXPECTED))
def test_csg_two_spheres(op):
    s, ref = two_sphere_csg(op)
    for (o, d), ts in zip(CSG_RAYS, CSG_EXPECTED[op]):
        hits = O.object_all_hits(s, ref, O.mk_ray(o, d))
        assert len(hits) == len(ts)
        assert close(hits["t"], ts)


def triple_csg():
    b = SceneBuilder()
    m = neutral(b)
    s1 = b.sphere(Transformation(), m, norender=True)
    s2 = b.sphere(translation(VEC_X), m, norender=True)
    pl = b.plane(translation((0, 0, -0.5)), m, norender=True)
    inter = b.csg(s1, s2, "intersection", norender=True)
    ref = b.csg(inter, pl, "difference")
    return b.build(), ref


TRIPLE_CASES = [((-2, 0, 0), VEC_X, [2, 3]), ((0, 0, -2), VEC_Z, []), ((1, 0, -2), VEC_Z, []),
                # the hit at t = 2 - sqrt(3)/2 is missed by design of is_point_inside (shapes_test.cpp:362-371)
                ((0.5, 0, 2), neg(VEC_Z), [2.5])]


def test_csg_triple():
    s, ref = triple_csg()
    for o, d, ts in TRIPLE_CASES:
        hits = O.object_all_hits(s, ref, O.mk_ray(o, d))
        assert len(hits) == len(ts) and close(hits["t"], ts)


def hemisphere():
    b = SceneBuilder()
    m = neutral(b)
    sp = b.sphere(Transformation(), m, norender=True)
    pl = b.plane(scaling((1, 1, -1)), m, norender=True)
    ref = b.csg(sp, pl, "intersection", translation((2, 0, 0)))
    return b.build(), ref


def test_csg_transformation():
    s, ref = hemisphere()
    assert len(O.object_all_hits(s, ref, O.mk_ray((0, 0, 2), neg(VEC_Z)))) == 0
    hits = O.object_all_hits(s, ref, O.mk_ray((2, 0, 2), neg(VEC_Z)))
    assert len(hits) == 2 and close(hits["t"], [1, 2])
    assert not O.is_point_inside(s, ref, (0, 0, 0.5))
    assert O.is_point_inside(s, ref, (2, 0, 0.5))
    assert not O.is_point_inside(s, ref, (2, 0, -0.5))


# ---------------------------------------------------------------------------------------------------
# test/shapes_test.cpp:407-453 (World)
# ---------------------------------------------------------------------------------------------------
def world_two_spheres(x1, x2):
    b = SceneBuilder()
    m = neutral(b)
    b.sphere(translation((x1, 0, 0)), m)
    b.sphere(translation((x2, 0, 0)), m)
    return b.build()


def rays_array(specs):
    rays = np.zeros(len(specs), dtype=abi.RAY_DTYPE)
    for i, (o, d) in enumerate(specs):
        rays[i]["origin"], rays[i]["direction"], rays[i]["tmin"], rays[i]["tmax"] = o, d, 1e-5, np.inf
    return rays


WORLD_RAYS = [((0, 0, 0), VEC_X), ((10, 0, 0), neg(VEC_X))]
VISIBLE_CASES = [  # viewer, surface point, normal, visible
    ((0, 0, 0), (10, 0, 0), (-1, 0, 0), 0), ((0, 0, 0), (5, 0, 0), (-1, 0, 0), 0),
    ((4, 0, 0), (5, 0, 0), (-1, 0, 0), 1), ((0, 0, 0), (0.5, 0, 0), (-1, 0, 0), 1),
    ((0, 0, 0), (0, 10, 0), (0, -1, 0), 1), ((0, 0, 0), (0, 0, 10), (0, 0, -1), 1)]


def check_world_hits(hits):
    assert np.all(hits["object"] >= 0)
    assert close(hits["point"][0], (1, 0, 0)) and close(hits["point"][1], (9, 0, 0))


def test_world_ray_intersection():
    check_world_hits(O.trace_rays(world_two_spheres(2, 10), rays_array(WORLD_RAYS)))


def test_world_offset_if_visible():
    s = world_two_spheres(2, 8)
    vis = O.offset_if_visible(s, [c[0] for c in VISIBLE_CASES], [c[1] for c in VISIBLE_CASES], [c[2] for c in VISIBLE_CASES])
    assert vis.tolist() == [c[3] for c in VISIBLE_CASES]


# ---------------------------------------------------------------------------------------------------
# test/materials_test.cpp:18-111
# ---------------------------------------------------------------------------------------------------
def pigment_scene(p_of):
    b = SceneBuilder()
    b.material("diffuse", p_of(b))
    return b.build()


def test_uniform_pigment():
    s = pigment_scene(lambda b: b.pigment_uniform((1, 2, 3)))
    for uv in ((0, 0), (0, 1), (1, 0), (1, 1)):
        assert close(O.pigment(s, 0, *uv), (1, 2, 3))


def test_checkered_pigment():
    s = pigment_scene(lambda b: b.pigment_checkered((1, 2, 3), (10, 20, 30), 2))
    assert close(O.pigment(s, 0, 0.25, 0.25), (1, 2, 3))
    assert close(O.pigment(s, 0, 0.25, 0.75), (10, 20, 30))
    assert close(O.pigment(s, 0, 0.75, 0.25), (10, 20, 30))
    assert close(O.pigment(s, 0, 0.75, 0.75), (1, 2, 3))


IMAGE_2X2 = np.array([[[1, 2, 3], [2, 3, 1]], [[2, 1, 3], [3, 2, 1]]], dtype=np.float32)  # [row][col]
IMAGE_2X2_CASES = [((0, 0), (1, 2, 3)), ((0.5, 0), (2, 3, 1)), ((0, 0.5), (2, 1, 3)), ((0.5, 0.5), (3, 2, 1))]


def test_image_pigment():
    s = pigment_scene(lambda b: b.pigment_image(IMAGE_2X2))
    for uv, c in IMAGE_2X2_CASES:
        assert close(O.pigment(s, 0, *uv), c)


def test_specular_brdf_reflection():
    in_dir = np.array([0, -1, -1], dtype=np.float32) / np.float32(math.sqrt(2))
    r = O.scatter_ray(abi.RT_BRDF_SPECULAR, O.PCG(), in_dir, (0, 0, 0), (0, 0, 1), 1)
    d = np.array(r.direction[:])
    assert close(d / np.linalg.norm(d), np.array([0, -1, 1]) / math.sqrt(2))


def test_diffusive_brdf_scatters_into_the_normal_hemisphere():
    """Not a reference test value: a property of DiffusiveBRDF::scatter_ray (materials.hpp) — unit directions,
    cos-weighted about the normal, so E[cos] = 2/3."""
    pcg = O.PCG()
    n = np.array([1, 2, -2], dtype=np.float32) / 3
    cos = []
    for _ in range(4000):
        d = np.array(O.scatter_ray(abi.RT_BRDF_DIFFUSE, pcg, (0, 0, -1), (0, 0, 0), n, 1).direction[:])
        assert abs(np.linalg.norm(d) - 1) < 1e-4
        cos.append(float(np.dot(d, n)))
    assert min(cos) > -1e-6 and abs(np.mean(cos) - 2 / 3) < 0.02


# ---------------------------------------------------------------------------------------------------
# test/renderers_test.cpp:18-205
# ---------------------------------------------------------------------------------------------------
ORTHO = Camera(abi.RT_CAMERA_ORTHOGONAL, Transformation(), None)


def small_sphere_world(color):
    b = SceneBuilder()
    m = b.material("diffuse", b.pigment_uniform(color))
    b.sphere(translation((2, 0, 0)) * scaling((0.2, 0.2, 0.2)), m)
    return b.build()


def check_3x3(img, centre):
    expected = np.zeros((3, 3, 3), dtype=np.float32)
    expected[1, 1] = centre
    assert close(img, expected)


def test_onoff_example():
    img, _ = O.render(small_sphere_world((1, 1, 1)), O.make_cfg(3, 3, "onoff"), ORTHO)
    check_3x3(img, (1, 1, 1))


def test_flat_example():
    img, _ = O.render(small_sphere_world((1, 2, 3)), O.make_cfg(3, 3, "flat"), ORTHO)
    check_3x3(img, (1, 2, 3))


def pointlight_example_world():
    b = SceneBuilder()
    m = b.material("diffuse", b.pigment_uniform((0.2, 0, 0)), b.pigment_uniform((0, 0.3, 0)))
    b.plane(translation(VEC_X) * rotation_y(-PI / np.float32(2)), m)
    b.plane(translation(VEC_Y) * rotation_x(PI / np.float32(2)), m)
    for y in (2, -2, -3):
        b.point_light((0, y, 0), (1, 1, 1))
    return b.build()


POINTLIGHT_EXPECTED = (0.2 * (1 / math.sqrt(5) + 1 / math.sqrt(10)) / math.pi, 0.3, 0.1)


def test_pointlight_example():
    cfg = O.make_cfg(1, 1, "pointlight", ambient=(0, 0, 0.1))
    img, _ = O.render(pointlight_example_world(), cfg, ORTHO)
    assert close(img[0, 0], POINTLIGHT_EXPECTED)


def pointlight_reflection_world(with_mirror: bool):
    b = SceneBuilder()
    black, grey = (0, 0, 0), (0.5, 0.5, 0.5)
    mirror = b.material("specular", b.pigment_uniform(grey), b.pigment_uniform(black))
    grey_m = b.material("diffuse", b.pigment_uniform(grey), b.pigment_uniform(black))
    sphere_m = b.material("diffuse", b.pigment_uniform(black), b.pigment_uniform(black))
    b.sphere(translation((-0.5, 0, 0)) * scaling((0.1, 0.1, 0.1)), sphere_m)
    b.point_light((-1, 0, 0), (1, 1, 1))
    b.plane(translation((2, 0, 0)) * rotation_y(-PI / np.float32(2)), grey_m)
    if with_mirror:
        b.plane(translation(VEC_X) * rotation_z(-PI / np.float32(4)) * rotation_y(-PI / np.float32(2)), mirror)
        b.plane(translation((0, 2, 0)) * rotation_x(PI / np.float32(2)), grey_m)
    return b.build()


REFLECTION_EXPECTED = 0.5 * (np.array([0, 0, 0.1]) + 0.5 / math.pi * math.cos(math.pi / 4))


def test_pointlight_reflections():
    cfg = O.make_cfg(1, 1, "pointlight", ambient=(0, 0, 0.1))
    ray = O.mk_ray((0, 0, 0), VEC_X)
    c1, _ = O.shade(pointlight_reflection_world(False), cfg, ray)
    assert close(c1, (0, 0, 0.1))
    c2, _ = O.shade(pointlight_reflection_world(True), cfg, ray)
    assert close(c2, REFLECTION_EXPECTED)


def furnace_world(reflectance, luminosity):
    b = SceneBuilder()
    m = b.material("diffuse", b.pigment_uniform((reflectance, 0, 0)), b.pigment_uniform((luminosity, 0, 0)))
    b.sphere(Transformation(), m)
    return b.build()


def furnace_cases(n):
    pcg = O.PCG()
    return [(np.float32(pcg.random_float()) * np.float32(0.9), np.float32(pcg.random_float())) for _ in range(n)]


def test_path_tracer_furnace():
    """renderers_test.cpp:167-205: n_rays 1, roulette limit 200, max depth 200, one PCG across all 100 cases."""
    cfg = O.make_cfg(1, 1, "path", n_rays=1, roulette=200, max_depth=200, rng_mode=abi.RT_RNG_SERIAL_REFERENCE)
    pcg = O.PCG()
    ray = O.mk_ray((0, 0, 0), VEC_X)
    for reflectance, luminosity in furnace_cases(100):
        c, rays = O.shade(furnace_world(reflectance, luminosity), cfg, ray, pcg=pcg)
        assert rays == 201
        assert close(c, (luminosity / (np.float32(1) - reflectance), 0, 0))


# ===================================================================================================
# The same known answers through the HIP library (C-ABI): hot-path operators only
# ===================================================================================================
@pytest.fixture(scope="module")
def gpu():
    from raytracer_amd import tracer
    return tracer


def gpu_hits(gpu, scene, specs):
    w = gpu.DeviceWorld(scene)
    try:
        return w.ray_intersection(rays_array(specs))
    finally:
        w.close()


def gpu_render(gpu, scene, camera, width, height, mode, spp=1):
    w = gpu.DeviceWorld(scene)
    try:
        return gpu.GpuImageTracer(width, height, camera, spp).fire_all_rays(w, mode)
    finally:
        w.close()


@pytest.mark.gpu
def test_gpu_world_ray_intersection_and_visibility(gpu):
    check_world_hits(gpu_hits(gpu, world_two_spheres(2, 10), WORLD_RAYS))
    w = gpu.DeviceWorld(world_two_spheres(2, 8))
    vis = w.offset_if_visible([c[0] for c in VISIBLE_CASES], [c[1] for c in VISIBLE_CASES], [c[2] for c in VISIBLE_CASES])
    w.close()
    assert vis.tolist() == [c[3] for c in VISIBLE_CASES]


@pytest.mark.gpu
def test_gpu_sphere_and_plane_records(gpu):
    s, _ = one_shape("sphere", Transformation())
    h = gpu_hits(gpu, s, [((0, 0, 2), neg(VEC_Z)), ((3, 0, 0), neg(VEC_X)), ((0, 0, 0), VEC_X), ((0, -2, 0), VEC_Y),
                          ((2, 0, 0.5), neg(VEC_X))])
    assert close(h["t"], [1, 2, 1, 1, h["t"][4]])
    assert close(h["normal"][:3], [VEC_Z, VEC_X, neg(VEC_X)])
    assert close(h["uv"], [(0, 0), (0, 0.5), (0, 0.5), (0.75, 0.5), (0, 1 / 3)])
    s, _ = one_shape("sphere", scaling((2, 1, 1)))
    n = gpu_hits(gpu, s, [((1, 1, 0), (-1, -1, 0))])["normal"][0]
    assert close(n / np.linalg.norm(n), np.array([1, 4, 0]) / math.sqrt(17))
    s, _ = one_shape("plane", rotation_y(PI / np.float32(2)))
    h = gpu_hits(gpu, s, [((1, 0, 0), neg(VEC_X)), ((1, 0, 0), VEC_X), ((1, 0, 0), VEC_Y), ((1, 0, 0), VEC_Z)])
    assert h["object"].tolist()[1:] == [-1, -1, -1] and abs(h["t"][0] - 1) < TOL and close(h["normal"][0], VEC_X)
    s, _ = one_shape("plane", Transformation())
    assert close(gpu_hits(gpu, s, [((4.25, 7.75, 1), neg(VEC_Z))])["uv"][0], (0.25, 0.75))


@pytest.mark.gpu
@pytest.mark.parametrize("op", list(CSG_EXPECTED))
def test_gpu_csg_first_hits(gpu, op):
    """World::ray_intersection over a CSG object returns the first valid hit of all_ray_intersections."""
    s, _ = two_sphere_csg(op)
    h = gpu_hits(gpu, s, CSG_RAYS)
    for i, ts in enumerate(CSG_EXPECTED[op]):
        if ts:
            assert h["object"][i] == 0 and abs(h["t"][i] - ts[0]) < TOL
        else:
            assert h["object"][i] == -1


@pytest.mark.gpu
def test_gpu_csg_triple_and_transformation(gpu):
    s, _ = triple_csg()
    h = gpu_hits(gpu, s, [(o, d) for o, d, _ in TRIPLE_CASES])
    for i, (_, _, ts) in enumerate(TRIPLE_CASES):
        assert (h["object"][i] == 0 and abs(h["t"][i] - ts[0]) < TOL) if ts else h["object"][i] == -1
    s, _ = hemisphere()
    h = gpu_hits(gpu, s, [((0, 0, 2), neg(VEC_Z)), ((2, 0, 2), neg(VEC_Z))])
    assert h["object"][0] == -1 and abs(h["t"][1] - 1) < TOL


@pytest.mark.gpu
def test_gpu_onoff_flat_examples(gpu):
    check_3x3(gpu_render(gpu, small_sphere_world((1, 1, 1)), ORTHO, 3, 3, gpu.RenderMode("onoff")), (1, 1, 1))
    check_3x3(gpu_render(gpu, small_sphere_world((1, 2, 3)), ORTHO, 3, 3, gpu.RenderMode("flat")), (1, 2, 3))


@pytest.mark.gpu
def test_gpu_image_pigment(gpu):
    """ImagePigment (materials_test.cpp:56-68) seen through a flat render of a textured plane facing the camera."""
    b = SceneBuilder()
    m = b.material("diffuse", b.pigment_image(IMAGE_2X2))
    b.plane(Transformation(), m)
    scene = b.build()
    for (u, v), c in IMAGE_2X2_CASES:
        h = gpu_hits(gpu, scene, [((u + 0.125, v + 0.125, 1), neg(VEC_Z))])
        assert close(h["uv"][0], (u + 0.125, v + 0.125))
        cam = Camera(abi.RT_CAMERA_ORTHOGONAL, translation((u + 0.125, v + 0.125, 2)) * rotation_y(PI / np.float32(2)), 1.0)
        img = gpu_render(gpu, scene, cam, 1, 1, gpu.RenderMode("flat"))
        ref, _ = O.render(scene, O.make_cfg(1, 1, "flat"), cam)
        assert close(img[0, 0], c) and np.array_equal(img, ref)


@pytest.mark.gpu
def test_gpu_pointlight_examples(gpu):
    mode = gpu.RenderMode("pointlight", ambient=(0, 0, 0.1))
    img = gpu_render(gpu, pointlight_example_world(), ORTHO, 1, 1, mode)
    assert close(img[0, 0], POINTLIGHT_EXPECTED)
    # the reflection test fires Ray{origin, VEC_X}: an orthogonal camera moved to x = +1 fires exactly that ray
    cam = Camera(abi.RT_CAMERA_ORTHOGONAL, translation((1, 0, 0)), 1.0)
    assert close(gpu_render(gpu, pointlight_reflection_world(False), cam, 1, 1, mode)[0, 0], (0, 0, 0.1))
    assert close(gpu_render(gpu, pointlight_reflection_world(True), cam, 1, 1, mode)[0, 0], REFLECTION_EXPECTED)


@pytest.mark.gpu
def test_gpu_stratified_sampling_mean(gpu):
    b = SceneBuilder()
    m = b.material("diffuse", b.pigment_uniform((0, 0, 0)), b.pigment_uniform((1, 2, 3)))
    b.plane(translation((5, 0, 0)) * rotation_y(-PI / np.float32(2)), m)
    t = gpu.GpuImageTracer(1, 1, Camera(abi.RT_CAMERA_ORTHOGONAL, Transformation(), 1.0), 10)
    w = gpu.DeviceWorld(b.build())
    img = t.fire_all_rays(w, gpu.RenderMode("flat"))
    w.close()
    assert t.last_stats.samples == 100 and close(img[0, 0], (1, 2, 3))


@pytest.mark.gpu
def test_gpu_path_tracer_furnace(gpu):
    """Furnace test on the device path kernel: L = Le / (1 - rho) whatever the random directions are."""
    cam = Camera(abi.RT_CAMERA_ORTHOGONAL, translation((1, 0, 0)), 1.0)   # fires Ray{(0,0,0), VEC_X}
    mode = gpu.RenderMode("path", n_rays=1, roulette=200, max_depth=200)
    for reflectance, luminosity in furnace_cases(12):
        t = gpu.GpuImageTracer(1, 1, cam, 1)
        w = gpu.DeviceWorld(furnace_world(reflectance, luminosity))
        img = t.fire_all_rays(w, mode)
        w.close()
        assert t.last_stats.rays == 201
        assert close(img[0, 0], (luminosity / (np.float32(1) - reflectance), 0, 0), 2e-5)


# ---------------------------------------------------------------------------------------------------
# RT_RNG_PER_NODE_STREAMS (ours: one stream per ray) must be the same estimator as the reference's serial PCG
# ---------------------------------------------------------------------------------------------------
def test_furnace_with_node_streams():
    """The furnace identity holds whatever the random numbers are, so it pins the node-stream evaluation
    (throughput form + fixed-point accumulation) of the oracle to 2^-32-level accuracy."""
    cam = Camera(abi.RT_CAMERA_ORTHOGONAL, translation((1, 0, 0)), 1.0)   # fires Ray{(0,0,0), VEC_X}
    cfg = O.make_cfg(1, 1, "path", n_rays=1, roulette=200, max_depth=200)
    for reflectance, luminosity in furnace_cases(20):
        img, st = O.render(furnace_world(reflectance, luminosity), cfg, cam)
        assert st.rays == 201
        assert close(img[0, 0], (luminosity / (np.float32(1) - reflectance), 0, 0), 2e-5)


def test_node_streams_agree_with_serial_reference_statistically(scenes, golden_images):
    """Same scene, same settings: the node-stream image and the reference's serial-PCG image (golden, 36 spp) are two
    independent estimates of one mean.  Their difference must look like Monte-Carlo noise: image means within 1 %,
    and the mean squared difference no larger than the sum of the two estimators' variances (measured from
    independent node-stream renders with other seq_numbers)."""
    scene = scenes["demo_path"]
    ref = golden_images["path_demo_path_96x72_aa6_serial"].astype(np.float64)
    imgs = [O.render(scene, O.make_cfg(96, 72, "path", spp_edge=6, seq_number=s))[0].astype(np.float64) for s in (54, 55, 56)]
    a, b, c = imgs
    var_one = 0.5 * np.mean((b - c) ** 2)                 # variance of one 36-spp estimate, image average
    mse = np.mean((a - ref) ** 2)                         # ~ 2 * var_one if both are unbiased
    assert abs(a.mean() - ref.mean()) < 0.01 * ref.mean()
    assert mse < 2.6 * var_one, (mse, var_one)
    assert mse > 1.4 * var_one, (mse, var_one)            # and they are not the same stream in disguise


def test_node_stream_keys_are_distinct():
    st = O._u64x2()
    seen = set()
    for g in range(64):
        O.dll().rto_node_root_stream(st, 54, g)
        root = st[0]
        assert st[1] == 109
        for k in range(10):
            st[0] = root
            O.dll().rto_node_child_stream(st, k)
            seen.add(st[0])
        seen.add(root)
    assert len(seen) == 64 * 11
