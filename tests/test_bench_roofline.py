"""bench.py's roofline block for every kernel family the dispatcher can name, with and without a committed counter profile — on the
CPU, so that a slip in the bookkeeping cannot take the driver's bench run down."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from raytracer_amd.scene import FlatScene  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("kernel,profile", [
    ("k_render_path_tile_cull<W=4,BF16x3>", "n4096_share"),
    ("k_render_path_tile_cull<W=4,BF16x3>", None),
    ("k_render_path_tile<S=8,W=4,BF16x3>", "n4096_share_scan_all"),
    ("k_render_path_tile<S=8,W=1,BF16x3>", None),
    ("k_render_path_tile<S=8,W=4>", None),
    ("k_render_path_wide<ISO,K=8,W=8>", None),
    ("k_render_path_wide<GEN,K=4,W=8>", None),
    ("k_render_path<FMA>", None),
    ("k_render_path<EXACT>", None),
])
def test_large_world_blocks(kernel, profile):
    scene, _ = synthetic_scene(4096)
    census = bench.scene_census(scene)
    prof = bench.load_profile(profile) if profile else None
    assert profile is None or prof is not None, f"profiles/r3/{profile}_pmc.json is missing"
    ms = 4500.0 if "tile_cull" in kernel else 11000.0  # about what a 2.5e10-ray share takes on these kernels
    out = bench.roofline_block(census, kernel, 2.5e10, ms, prof)
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in out
    assert out["frac_algorithmic"] > 0
    if prof is not None:
        assert out["frac"] is not None and 0 < out["frac"] <= 1 and out["traffic"] > 0
    if "tile_cull" in kernel and prof is not None:
        assert 0 < out["mfma"]["frac"] < 1 and 0 < out["valu_issue"]["frac"] < 1.3
        assert out["mfma"]["tile_products_per_32_rays"] < out["mfma"]["scan_all_tile_products_per_32_rays"]
    if "tile_cull" in kernel and prof is None:
        assert out["frac"] is None  # data-dependent flops are never taken from a formula


@pytest.mark.parametrize("fixture,kernel,profile", [
    ("demo_path", "k_render_path_wave<FMA,NO_CSG>", "c3"),
    ("csg_example", "k_render_path_wave<FMA,CSG_TT4>", "c4"),
    ("demo_onoff", "k_render_simple<ONOFF>", None),
])
def test_small_world_blocks(fixture, kernel, profile):
    scene = FlatScene.from_json(os.path.join(GOLDEN, f"{fixture}.scene.json"))
    prof = bench.load_profile(profile) if profile else None
    out = bench.roofline_block(bench.scene_census(scene), kernel, 1e9, 20.0, prof)
    assert out["bound"] == "valu" and 0 < out["frac"] < 1
    if prof is not None:
        assert 0.5 < out["valu_issue"]["frac"] < 1.5


def test_bvh_block_carries_no_fraction():
    scene, _ = synthetic_scene(512)
    out = bench.roofline_block(bench.scene_census(scene), "k_render_path_bvh<FMA,T=256,LEAF_LDS>", 1e9, 100.0, None, bvh=True)
    assert out["frac"] is None
