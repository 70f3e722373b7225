"""CSG trees that are the intersection of all their leaves (samples/csg_example.txt is one): the fused path kernels evaluate
them in closed form — the ray is inside the solid between the last entry and the first exit, and only those two boundaries can
be valid hits (rt_device.h, CsgTop::tt == 2) — instead of testing all 2 L boundary hits against the truth tables.  Trees that are
the union of all their leaves (tt == 3: the reference keeps every hit of a union) take the smallest candidate.  The host decides
from the truth tables, so a tree with one other operation must keep the general evaluator.

Every leaf has its own emitting material and max_depth is 0, so a path-mode pixel is the emission of the leaf whose boundary
the camera ray hits first: the image is the table of winners, free of Monte-Carlo noise.  The fused kernel must name the winners
of the exact kernel (the reference's interval lists, itself pinned to the oracle here) except where the decision is within
rounding: boundary pixels."""
import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import SceneBuilder, compose, degs_to_rads, rotation_x, rotation_y, rotation_z, scaling, translation

pytestmark = pytest.mark.gpu

W, H = 160, 120


def chain(b, leaves, ops):
    """((l0 op0 l1) op1 l2) ... as nested two-child nodes, the last one rendered."""
    node = leaves[0]
    for k, leaf in enumerate(leaves[1:]):
        last = k == len(leaves) - 2
        node = b.csg(node, leaf, ops[k], compose(translation([0.1, 0.05, -0.2])) if last else None, norender=not last)
    return node


def build(kind):
    b = SceneBuilder()
    black = b.pigment_uniform((0, 0, 0))

    def mat(k):
        return b.material("diffuse", black, b.pigment_uniform((0.1 + 0.13 * k, 1.0 - 0.11 * k, 0.3 + 0.05 * (k % 3))))

    b.sphere(compose(scaling([40, 40, 40])), mat(7))  # sky: every ray ends somewhere
    if kind == "axis_aligned":  # translations and axis-aligned scalings only (Scene::all_diag), 4 leaves like the sample scene
        leaves = [b.sphere(compose(translation([0, 0.2, 0])), mat(0), True), b.sphere(compose(translation([0, -0.6, 0.1]), scaling([1.2, 1.2, 1.2])), mat(1), True),
                  b.plane(compose(scaling([1, 1, -1])), mat(2), True), b.plane(compose(translation([0, 0, 0.6])), mat(3), True)]
        ops = ["intersection"] * 3
    elif kind == "rotated":  # rotated and non-uniformly scaled leaves: the general transforms
        leaves = [b.sphere(compose(rotation_z(degs_to_rads(30)), scaling([1.4, 0.8, 1.0])), mat(0), True),
                  b.sphere(compose(translation([0.3, -0.4, 0.2]), rotation_x(degs_to_rads(20)), scaling([1.0, 1.3, 0.7])), mat(1), True),
                  b.plane(compose(rotation_x(degs_to_rads(25)), translation([0, 0, 0.4])), mat(2), True)]
        ops = ["intersection"] * 2
    elif kind == "six_leaves":  # a sphere cut by five half-spaces: the 6-leaf evaluator
        leaves = [b.sphere(compose(scaling([1.3, 1.3, 1.3])), mat(0), True),
                  b.plane(compose(translation([0, 0, 0.7])), mat(1), True), b.plane(compose(translation([0, 0, -0.6]), scaling([1, 1, -1])), mat(2), True),
                  b.plane(compose(translation([0, 0.8, 0]), rotation_x(degs_to_rads(-90))), mat(3), True),
                  b.plane(compose(translation([0, -0.7, 0]), rotation_x(degs_to_rads(90))), mat(4), True),
                  b.sphere(compose(translation([0.5, 0, 0]), scaling([1.5, 1.5, 1.5])), mat(5), True)]
        ops = ["intersection"] * 5
    elif kind == "coincident":  # two leaves share their boundary: strict compares make those hits invalid in the reference
        leaves = [b.sphere(compose(translation([0, 0, 0])), mat(0), True), b.sphere(compose(translation([0, 0, 0])), mat(1), True),
                  b.plane(compose(translation([0, 0, 0.3])), mat(2), True)]
        ops = ["intersection"] * 2
    elif kind == "union":  # every hit of every leaf is valid (shapes.hpp:460-490): the first one wins
        leaves = [b.sphere(compose(translation([0, 0.5, 0])), mat(0), True), b.sphere(compose(translation([0, -0.5, 0.2]), scaling([0.9, 0.9, 0.9])), mat(1), True),
                  b.sphere(compose(translation([-0.5, 0, 0.6]), scaling([0.5, 0.5, 0.5])), mat(2), True),
                  b.sphere(compose(translation([0.2, 0, -0.5]), rotation_z(degs_to_rads(40)), scaling([1.1, 0.6, 0.5])), mat(3), True)]
        ops = ["union"] * 3
    elif kind == "union_of_intersection":  # (a and b) or c: neither closed form applies
        leaves = [b.sphere(compose(translation([0, 0.3, 0])), mat(0), True), b.sphere(compose(translation([0, -0.5, 0.1])), mat(1), True),
                  b.sphere(compose(translation([-0.3, 0, 0.8]), scaling([0.6, 0.6, 0.6])), mat(2), True)]
        ops = ["intersection", "union"]
    else:  # "mixed": one difference among the intersections — NOT a pure intersection, the general evaluator must run
        leaves = [b.sphere(compose(translation([0, 0.2, 0])), mat(0), True), b.sphere(compose(translation([0, -0.6, 0.1])), mat(1), True),
                  b.sphere(compose(translation([-0.7, 0, 0.3]), scaling([0.6, 0.6, 0.6])), mat(2), True), b.plane(compose(translation([0, 0, 0.6])), mat(3), True)]
        ops = ["intersection", "difference", "intersection"]
    chain(b, leaves, ops)
    # looking down at the solid from the side, so that caps, cuts and sphere faces all show
    b.set_camera("perspective", compose(rotation_z(degs_to_rads(35)), rotation_y(degs_to_rads(28)), translation([-3.0, 0, 0])), None, 2.6)
    return b.build()


@pytest.mark.parametrize("kind", ["axis_aligned", "rotated", "six_leaves", "coincident", "mixed", "union", "union_of_intersection"])
def test_fused_evaluator_names_the_exact_kernels_winners(kind):
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    scene = build(kind)
    world = DeviceWorld(scene)
    tr = GpuImageTracer(W, H, scene.camera, 1)
    kw = dict(n_rays=1, roulette=99, max_depth=0, seq_number=5)
    exact = tr.fire_all_rays(world, RenderMode("path", exact=True, **kw))
    fused = tr.fire_all_rays(world, RenderMode("path", **kw))
    assert tr.last_stats.kernel.decode().startswith("k_render_path_wave<FMA")
    world.close()
    want, _ = O.render(scene, O.make_cfg(W, H, "path", spp_edge=1, rng_mode=abi.RT_RNG_PER_NODE_STREAMS, **kw))
    assert np.array_equal(exact, want)  # no scattering: nothing for a libm ulp to flip
    csg_pixels = (np.abs(want - want[0, 0]).max(axis=-1) > 0).sum()  # pixels that do not show the sky
    assert csg_pixels > 0.06 * W * H, csg_pixels
    differ = np.abs(fused - exact).max(axis=-1) > 0
    off = differ.sum()
    print(f"{kind}: {csg_pixels} pixels on the solid, {off} differ between the fused and the exact kernel")
    colours = lambda img: {tuple(c) for c in np.unique(img.reshape(-1, 3), axis=0)}  # noqa: E731
    if kind != "coincident":
        assert off <= 0.02 * csg_pixels  # silhouette and edge pixels where two boundaries are within rounding
        assert colours(exact) == colours(fused)  # every leaf that shows in the exact image shows in the fused one
    else:
        # Two leaves with the SAME boundary.  The reference asks whether the hit point, mapped into the other sphere's frame, is
        # strictly inside it: a coin flip on the last bits (speckle).  The fused evaluators compare ray parameters, which are
        # bit-equal here, so such a hit is never valid — in the general evaluator and in the closed form alike (DESIGN.md 4.3:
        # a point within rounding of another leaf's boundary may fall on either side).  So the only pixels that may differ
        # are those where the exact kernel accepted a hit on one of the coincident spheres, and the fused image shows no such hit.
        sphere_colours = {tuple(np.float32(c) for c in (0.1 + 0.13 * k, 1.0 - 0.11 * k, 0.3 + 0.05 * (k % 3))) for k in (0, 1)}
        on_sphere = np.array([[tuple(px) in sphere_colours for px in row] for row in exact])
        assert not (differ & ~on_sphere).any()
        assert not (colours(fused) & sphere_colours)
        assert colours(fused) <= colours(exact) and len(colours(fused)) >= 2  # the sky and the plane's cap
