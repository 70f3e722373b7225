"""End-to-end test of the C++ drop-in (raytracer_amd/host/gpu_image_tracer.hpp + flatten.hpp).

oracle/_ref/raytracer_gpu is the reference's own pipeline — its scene parser, World, HdrImage::write_pfm,
compiled from the unmodified reference headers — with GpuImageTracer::fire_all_rays in place of
ImageTracer::fire_all_rays (src/raytracer.cpp:140-153).  The test feeds it a scene file written here in the
reference's scene language (CSG tree, mirror, checkered floor, a point light) and compares the PFM it writes
with the same scene built through the Python SceneBuilder and rendered through the same library: the two
host routes must give identical bits, in every mode.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import SceneBuilder, Transformation, compose, rotation_z, degs_to_rads, scaling, translation

GPU_BIN = os.path.join(O.ORACLE_DIR, "_ref", "raytracer_gpu")

SCENE_TEXT = """
# written for tests/test_cpp_adapter.py
float lens_radius(0.8)
material ground(diffuse(checkered(<0.9, 0.8, 0.2>, <0.1, 0.1, 0.3>, 2)), uniform(<0, 0, 0>))
material sky(diffuse(uniform(<0, 0, 0>)), uniform(<0.6, 0.7, 1>))
material mirror(specular(uniform(<0.7, 0.7, 0.7>)), uniform(<0, 0, 0>))
material red(diffuse(uniform(<0.8, 0.2, 0.1>)), uniform(<0, 0, 0>))
material glow(diffuse(uniform(<0.1, 0.1, 0.1>)), uniform(<0.9, 0.6, 0.2>))

plane(translation([0, 0, -1]), ground)
sphere(scaling([40, 40, 40]), sky)
norender sphere a(scaling([lens_radius, lens_radius, lens_radius]), red)
norender sphere b(translation([0.5, 0, 0]) * scaling([lens_radius, lens_radius, lens_radius]), glow)
norender plane cut(translation([0, 0, 0.25]), red)
norender csg lens(a, b, intersection, identity)
csg(lens, cut, difference, translation([1, 0.5, 0]) * rotation_z(30))
sphere(translation([1.5, -1.2, -0.4]) * scaling([0.6, 0.6, 0.6]), mirror)
point_light([-2, 3, 4], <1, 0.9, 0.8>, 0)
point_light([0, -4, 2], <0.3, 0.3, 0.5>, 2.5)
camera(perspective, translation([-3, 0, 0.3]), exact_asp_ratio, 1.5)
"""


def twin_scene():
    b = SceneBuilder()
    f = np.float32
    ground = b.material("diffuse", b.pigment_checkered((f(0.9), f(0.8), f(0.2)), (f(0.1), f(0.1), f(0.3)), 2))
    sky = b.material("diffuse", b.pigment_uniform((0, 0, 0)), b.pigment_uniform((f(0.6), f(0.7), f(1))))
    red = b.material("diffuse", b.pigment_uniform((f(0.8), f(0.2), f(0.1))))
    glow = b.material("diffuse", b.pigment_uniform((f(0.1), f(0.1), f(0.1))), b.pigment_uniform((f(0.9), f(0.6), f(0.2))))
    mirror = b.material("specular", b.pigment_uniform((f(0.7), f(0.7), f(0.7))))
    r = f(0.8)
    b.plane(compose(translation([0, 0, -1])), ground)
    b.sphere(compose(scaling([40, 40, 40])), sky)
    a = b.sphere(compose(scaling([r, r, r])), red, norender=True)
    bb = b.sphere(compose(translation([0.5, 0, 0]), scaling([r, r, r])), glow, norender=True)
    cut = b.plane(compose(translation([0, 0, 0.25])), red, norender=True)
    lens = b.csg(a, bb, "intersection", compose(Transformation()), norender=True)
    b.csg(lens, cut, "difference", compose(translation([1, 0.5, 0]), rotation_z(degs_to_rads(30))))
    b.sphere(compose(translation([1.5, -1.2, -0.4]), scaling([0.6, 0.6, 0.6])), mirror)
    b.point_light((-2, 3, 4), (f(1), f(0.9), f(0.8)), 0.0)
    b.point_light((0, -4, 2), (f(0.3), f(0.3), f(0.5)), 2.5)
    b.set_camera("perspective", compose(translation([-3, 0, 0.3])), None, 1.5)
    return b.build()


def run_gpu_bin(tmp_path, text, *args):
    txt, pfm = tmp_path / "scene.txt", tmp_path / "out.pfm"
    txt.write_text(text)
    res = subprocess.run([GPU_BIN, str(txt), "-o", str(pfm), *map(str, args)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    return O.read_pfm(pfm), json.loads(res.stdout)


needs_bin = pytest.mark.skipif(not os.path.exists(GPU_BIN), reason="oracle/_ref/raytracer_gpu was not built (needs /root/reference)")


@pytest.mark.skipif(not os.path.exists(O.REF_BIN) or not os.path.isdir("/root/reference"), reason="needs the reference build")
def test_twin_scene_equals_reference_parse(tmp_path):
    """CPU: the scene text parsed by the reference and flattened by flatten.hpp == the SceneBuilder twin, up to the
    numbering of materials (the flattener numbers them in order of first use)."""
    from raytracer_amd.scene import FlatScene
    txt, js = tmp_path / "s.txt", tmp_path / "s.json"
    txt.write_text(SCENE_TEXT)
    subprocess.run([O.REF_BIN, "flatten", str(txt), str(js)], check=True, capture_output=True)
    ref, twin = FlatScene.from_json(js), twin_scene()
    for name in ("shape_kind", "shape_fwd", "shape_inv", "csg_op", "csg_left", "csg_right", "csg_fwd", "csg_inv", "objects",
                 "light_point", "light_color", "light_radius"):
        assert getattr(ref, name).tobytes() == getattr(twin, name).tobytes(), name
    assert np.array_equal(ref.camera.transformation.fwd, twin.camera.transformation.fwd)
    for mode in ("flat", "pointlight"):
        a, _ = O.render(ref, O.make_cfg(64, 48, mode, spp_edge=2))
        b, _ = O.render(twin, O.make_cfg(64, 48, mode, spp_edge=2))
        assert np.array_equal(a, b)


@needs_bin
@pytest.mark.gpu
@pytest.mark.parametrize("mode,aa", [("onoff", 1), ("flat", 2), ("pointlight", 2), ("path", 2)])
def test_cpp_route_equals_python_route(tmp_path, mode, aa):
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    w, h = 96, 64
    img_cpp, info = run_gpu_bin(tmp_path, SCENE_TEXT, "--mode", mode, "--width", w, "--height", h, "--aa", aa, "--n_rays", 4,
                                "--roulette", 2, "--max-depth", 4, "--seq", 9)
    scene = twin_scene()
    world = DeviceWorld(scene)
    tracer = GpuImageTracer(w, h, scene.camera, aa)
    img_py = tracer.fire_all_rays(world, RenderMode(mode, n_rays=4, roulette=2, max_depth=4, seq_number=9))
    world.close()
    assert info["rays"] == tracer.last_stats.rays
    assert np.array_equal(img_cpp, img_py)
    if mode != "path":  # and both equal the CPU oracle, bit for bit, in the deterministic modes
        ref, _ = O.render(scene, O.make_cfg(w, h, mode, spp_edge=aa))
        assert np.array_equal(img_py, ref)


@needs_bin
@pytest.mark.gpu
def test_cpp_route_reports_errors_like_the_reference(tmp_path):
    """A scene the parser rejects ends with the parser's message on stderr and EXIT_FAILURE (src/raytracer.cpp:132-137)."""
    txt = tmp_path / "bad.txt"
    txt.write_text("sphere(identity, no_such_material)\n")
    res = subprocess.run([GPU_BIN, str(txt), "-o", str(tmp_path / "o.pfm")], capture_output=True, text=True, timeout=60)
    assert res.returncode != 0 and res.stderr.strip()


@needs_bin
@pytest.mark.gpu
def test_cpp_route_with_the_bvh_flag(tmp_path):
    """`--bvh` (GpuRenderMode::flags = RT_FLAG_ACCEL_BVH) on the synthetic scene text parsed by the reference's own parser:
    same frame and ray count as the Python route with the flag, and as the C++ route without it (the FMA kernels test a
    sphere with the same arithmetic through the hierarchy and in the wide scan)."""
    from raytracer_amd.synth import synthetic_scene
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
    scene, text = synthetic_scene(96)
    args = ("--mode", "path", "--width", 64, "--height", 40, "--aa", 2, "--n_rays", 3, "--roulette", 2, "--max-depth", 3, "--seq", 5)
    img_bvh, info_bvh = run_gpu_bin(tmp_path, text, *args, "--bvh")
    img_lin, info_lin = run_gpu_bin(tmp_path, text, *args)
    world = DeviceWorld(scene)
    tracer = GpuImageTracer(64, 40, scene.camera, 2)
    img_py = tracer.fire_all_rays(world, RenderMode("path", n_rays=3, roulette=2, max_depth=3, seq_number=5, bvh=True))
    world.close()
    assert np.array_equal(img_bvh, img_py) and info_bvh["rays"] == tracer.last_stats.rays
    assert np.array_equal(img_bvh, img_lin) and info_bvh["rays"] == info_lin["rays"]


@needs_bin
@pytest.mark.gpu
def test_cpp_route_resident_world_and_repeat(tmp_path):
    """--repeat renders the frame again on the World that is already in HBM (GpuImageTracer re-uploads only when the
    flattened World changes): same pixels, and the binary reports the end-to-end time of a resident frame."""
    args = ("--mode", "path", "--width", 96, "--height", 64, "--aa", 2, "--n_rays", 4, "--roulette", 2, "--max-depth", 4, "--seq", 9)
    once, info1 = run_gpu_bin(tmp_path, SCENE_TEXT, *args)
    again, info2 = run_gpu_bin(tmp_path, SCENE_TEXT, *args, "--repeat", 3, "--one-gpu")
    assert np.array_equal(once, again) and info1["rays"] == info2["rays"]
    assert 0 < info2["resident_frame_ms"] < 1e3 * info2["elapsed_s"]
    assert info2["kernel"].startswith("k_render_path_wave") and info2["gpus"] == 1


@needs_bin
@pytest.mark.gpu
def test_cpp_route_over_several_ranks_with_progress(tmp_path):
    """GpuImageTracer's multi-GPU route (rank_devices / all_devices -> rt_multi_render_frame_host) on the one-GPU box: three
    ranks on device 0.  Same PFM bytes and ray count as the single-device route; the progress callback's fractions increase
    and end at 1."""
    args = ("--mode", "path", "--width", 120, "--height", 70, "--aa", 2, "--n_rays", 4, "--roulette", 2, "--max-depth", 4, "--seq", 11)
    one, info1 = run_gpu_bin(tmp_path, SCENE_TEXT, *args)
    txt = tmp_path / "multi.txt"
    txt.write_text(SCENE_TEXT)
    out = tmp_path / "multi.pfm"
    res = subprocess.run([GPU_BIN, str(txt), "-o", str(out), *map(str, args), "--ranks", "0,0,0", "--progress"], capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    info3 = json.loads(res.stdout.strip().splitlines()[-1])
    assert info3["gpus"] == 3 and info3["rays"] == info1["rays"]
    assert np.array_equal(O.read_pfm(out), one)
    seen = [float(line.split()[1]) for line in res.stderr.splitlines() if line.startswith("progress ")]
    assert seen and seen == sorted(set(seen)) and seen[-1] == 1.0 and seen[0] > 0
    # --all-gpus on a box with one GPU is the single-device route
    same, info = run_gpu_bin(tmp_path, SCENE_TEXT, *args, "--all-gpus")
    assert np.array_equal(same, one) and info["gpus"] == 1
