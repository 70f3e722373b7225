"""ImagePigment at render level (reference: include/materials.hpp:102-138; DSL image("file.pfm"), include/scenefiles.hpp:685-690).

Fixtures: tests/golden/make_image_pigment_golden.py — renders made by the UNMODIFIED reference of scenes whose materials read a
37 x 23 texture of random texels: on spheres (uv from atan2 / acos), on the floor plane (uv = fractional part), as BRDF pigment
and as emission pigment; a second world with 40 such spheres, which the large-world kernel renders; and two one-ray scenes
that hit u == 1.0f / v == 1.0f exactly, where int(u * w) == w and only the clamp of materials.hpp:126-133 picks the texel.

CPU: the oracle reproduces every render bit for bit (path mode with the reference's serial generator included).
GPU (through the C-ABI): the clamp cases exactly; flat / pointlight within the uv budget of the device's atan2f / acosf (a texel
index flips where a hit sits within an ulp of a texel edge); path mode statistically against the reference's 36-spp renders.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import FlatScene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ("image_pigment", "image_pigment_many")


@pytest.fixture(scope="module")
def images():
    return dict(np.load(os.path.join(GOLDEN, "image_pigment_images.npz")))


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLDEN, "image_pigment_meta.json")) as f:
        return json.load(f)


def scene(name):
    return FlatScene.from_json(os.path.join(GOLDEN, name + ".scene.json"))


def test_fixtures_hold_the_texture(images):
    s = scene("image_pigment")
    tex = images["texture"]
    assert tex.shape == (23, 37, 3)
    kinds = set(int(k) for k in s.pigment_kind)
    assert abi.RT_PIGMENT_IMAGE in kinds
    # the flattened texel pool is the texture, row 0 = top (HdrImage's order after reading the PFM)
    off = int(s.pigment_tex_offset[list(s.pigment_kind).index(abi.RT_PIGMENT_IMAGE)])
    assert np.array_equal(np.asarray(s.texels).reshape(-1, 3)[off:off + 23 * 37].reshape(23, 37, 3), tex)
    assert scene("image_pigment_many").count("n_objects") == 42


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_the_reference_renders(images, name):
    s = scene(name)
    for mode, w, h, aa in (("flat", 160, 120, 1), ("flat", 120, 90, 2)):
        got, _ = O.render(s, O.make_cfg(w, h, mode, spp_edge=aa))
        assert np.array_equal(got, images[f"{name}_{mode}_{w}x{h}_aa{aa}"]), (name, mode, aa)
    got, _ = O.render(s, O.make_cfg(48, 36, "path", spp_edge=1, n_rays=3, roulette=2, max_depth=4, seq_number=9,
                                    rng_mode=abi.RT_RNG_SERIAL_REFERENCE))
    assert np.array_equal(got, images[f"{name}_path_48x36_aa1_serial"])
    if name == "image_pigment":
        got, _ = O.render(s, O.make_cfg(160, 120, "pointlight", spp_edge=1))
        assert np.array_equal(got, images["image_pigment_pointlight_160x120_aa1"])


@pytest.mark.parametrize("which,axis", [("u", 1), ("v", 0)])
def test_oracle_clamp_cases(images, meta, which, axis):
    """u == 1.0f (a sphere hit where atan2 / 2 pi + 1 rounds to 1) and v == 1.0f (a plane hit at y = -1e-10): the texel is the last
    column / row, the one the reference's clamp returns."""
    name = f"image_pigment_clamp_{which}"
    got, _ = O.render(scene(name), O.make_cfg(1, 1, "flat"))
    assert np.array_equal(got, images[name + "_flat_1x1"])
    row, col = meta[name]["texel_row_col"]
    assert (col if which == "u" else row) == images["texture"].shape[axis] - 1
    assert np.array_equal(got[0, 0], images["texture"][row, col])


def test_nan_coordinate_is_defined_as_index_zero():
    """acos of a local z that rounding left a hair beyond 1 gives v = NaN at a sphere's pole (about one hit in 1e8; it aborts the
    reference's 36-spp render of the first scene at seed 54, see make_image_pigment_golden.py): the reference then reads
    pixels[] out of bounds.  The oracle (and the device code) define the case: index 0."""
    s = scene("image_pigment")
    pg = list(s.pigment_kind).index(abi.RT_PIGMENT_IMAGE)
    tex = np.asarray(s.texels).reshape(-1, 3)
    off, w = int(s.pigment_tex_offset[pg]), int(s.pigment_tex_width[pg])
    assert np.array_equal(O.pigment(s, pg, 0.5, float("nan")), tex[off + int(0.5 * w)])
    assert np.array_equal(O.pigment(s, pg, float("nan"), float("nan")), tex[off])


# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["u", "v"])
def test_gpu_clamp_cases(images, which):
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    name = f"image_pigment_clamp_{which}"
    s = scene(name)
    world = DeviceWorld(s)
    got = GpuImageTracer(1, 1, s.camera, 1).fire_all_rays(world, RenderMode("flat"))
    world.close()
    assert np.array_equal(got, images[name + "_flat_1x1"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_deterministic_modes_within_the_uv_budget(images, name):
    """Everything but (u, v) of a sphere hit is bit-exact on the device; atan2f / acosf are the device's libm (1e-6 off glibc's at
    most, tests/test_gpu_parity.py), so a hit within that of a texel edge reads the neighbouring texel.  Bar: at most 0.5 % of
    the pixels differ at all."""
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    s = scene(name)
    world = DeviceWorld(s)
    cases = [("flat", 160, 120, 1), ("flat", 120, 90, 2)] + ([("pointlight", 160, 120, 1)] if name == "image_pigment" else [])
    try:
        for mode, w, h, aa in cases:
            got = GpuImageTracer(w, h, s.camera, aa).fire_all_rays(world, RenderMode(mode))
            want = images[f"{name}_{mode}_{w}x{h}_aa{aa}"]
            differ = (got != want).any(axis=-1)
            print(f"{name} {mode} AA={aa}: {int(differ.sum())} of {differ.size} pixels differ, max {np.abs(got - want).max():.3g}")
            assert differ.mean() <= 0.005
            assert abs(float(got.mean()) - float(want.mean())) <= 1e-4 * float(want.mean())
    finally:
        world.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kernel_prefix", [("image_pigment", "k_render_path_wave<"), ("image_pigment_many", "k_render_path_tile")])
def test_gpu_path_statistical_vs_reference_36spp(images, meta, name, kernel_prefix):
    """RMSE bound against the reference's 36-spp serial-PCG render (as tests/test_gpu_parity.py::test_path_statistical_vs_reference_high_spp):
    two independent unbiased 36-spp estimates differ with mean square 2 s^2 / 36; s^2 comes from a second device render with another
    seed; bound 1.25 x, means within 1.5 %.  The first scene runs the small-world kernel, the 42-object one the large-world kernel
    (its shading reads the texels)."""
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    s = scene(name)
    info = meta[f"{name}_path_64x48_aa6_serial"]
    kw = dict(n_rays=info["n_rays"], roulette=info["roulette"], max_depth=info["max_depth"])
    ref = images[f"{name}_path_64x48_aa6_serial"]
    world = DeviceWorld(s)
    tr = GpuImageTracer(64, 48, s.camera, 6)
    a = tr.fire_all_rays(world, RenderMode("path", seq_number=54, **kw))
    assert tr.last_stats.kernel.decode().startswith(kernel_prefix), tr.last_stats.kernel
    b = tr.fire_all_rays(world, RenderMode("path", seq_number=1234, **kw))
    world.close()
    mse_gpu_gpu = float(np.mean((a - b) ** 2))
    mse_gpu_ref = float(np.mean((a - ref) ** 2))
    print(f"{name}: rmse(gpu,ref)={mse_gpu_ref ** 0.5:.5f} rmse(gpu,gpu')={mse_gpu_gpu ** 0.5:.5f} mean={a.mean():.5f}/{ref.mean():.5f}")
    assert mse_gpu_ref <= 1.25 * mse_gpu_gpu + 1e-12
    assert abs(float(a.mean()) - float(ref.mean())) <= 0.015 * float(ref.mean())


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_path_same_streams_vs_oracle(name):
    """Same per-ray streams as the oracle: the exact-arithmetic kernel follows the oracle's paths except where a libm ulp flips a
    decision (here also a texel index); the fused kernels more often."""
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    s = scene(name)
    kw = dict(n_rays=3, roulette=2, max_depth=3, seq_number=11)
    want, _ = O.render(s, O.make_cfg(60, 40, "path", spp_edge=2, **kw))
    world = DeviceWorld(s)
    tr = GpuImageTracer(60, 40, s.camera, 2)
    exact = tr.fire_all_rays(world, RenderMode("path", exact=True, **kw))
    fast = tr.fire_all_rays(world, RenderMode("path", **kw))
    world.close()
    for label, img, budget in (("exact", exact, 0.03), ("fused", fast, 0.10)):
        off = (np.abs(img - want).max(axis=-1) > 1e-4).mean()
        print(f"{name} {label}: {100 * off:.2f} % of pixels off the oracle's path")
        assert off <= budget
        assert abs(float(img.mean()) - float(want.mean())) <= 0.02 * float(want.mean())
