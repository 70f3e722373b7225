"""Pins the CPU restatement (oracle/rt_oracle.c) against outputs of the UNMODIFIED reference
(tests/golden/*, produced by tests/golden/make_golden.py through oracle/_ref/rt_ref).

With RT_RNG_SERIAL_REFERENCE the oracle consumes random numbers exactly like the reference, so even
path-mode images must be bit-identical here (same glibc libm as the fixtures were rendered with).
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SERIAL = abi.RT_RNG_SERIAL_REFERENCE


def test_onoff_c1_mask_bit_identical(scenes):
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))["onoff_demo_onoff_1280x960"]
    bits = np.load(os.path.join(GOLDEN, "onoff_demo_onoff_1280x960.maskbits.npy"))
    mask = np.unpackbits(bits)[: 1280 * 960].reshape(960, 1280).astype(bool)
    img, stats = O.render(scenes["demo_onoff"], O.make_cfg(1280, 960, "onoff"))
    assert set(np.unique(img)) <= {0.0, 1.0}
    assert np.array_equal(img[..., 0] > 0, mask)
    assert int(mask.sum()) == meta["hit_pixels"] == 22712
    assert stats.rays == 1280 * 960


@pytest.mark.parametrize("key,scene,mode,w,h,aa", [
    ("flat_demo_path_160x120_aa1", "demo_path", "flat", 160, 120, 1),
    ("flat_demo_path_160x120_aa2", "demo_path", "flat", 160, 120, 2),
    ("pointlight_demo_path_160x120_aa1", "demo_path", "pointlight", 160, 120, 1),
    ("pointlight_demo_path_160x120_aa2", "demo_path", "pointlight", 160, 120, 2),
    ("onoff_demo_onoff_160x120_aa1", "demo_onoff", "onoff", 160, 120, 1),
    ("onoff_demo_onoff_160x120_aa2", "demo_onoff", "onoff", 160, 120, 2),
    ("pointlight_demo_onoff_160x120_aa1", "demo_onoff", "pointlight", 160, 120, 1),
    ("pointlight_demo_onoff_160x120_aa2", "demo_onoff", "pointlight", 160, 120, 2),
    ("flat_csg_example_240x135_aa1", "csg_example", "flat", 240, 135, 1),
    ("flat_csg_example_240x135_aa2", "csg_example", "flat", 240, 135, 2),
    ("onoff_csg_example_240x135_aa1", "csg_example", "onoff", 240, 135, 1),
    ("onoff_csg_example_240x135_aa2", "csg_example", "onoff", 240, 135, 2),
    ("flat_demo_path_clock30_160x120_aa1", "demo_path_clock30", "flat", 160, 120, 1),
])
def test_deterministic_modes_bit_identical(golden_images, scenes, key, scene, mode, w, h, aa):
    for rng_mode in (SERIAL, abi.RT_RNG_PER_NODE_STREAMS):  # AA jitter: serial draw order == jump-ahead
        img, _ = O.render(scenes[scene], O.make_cfg(w, h, mode, spp_edge=aa, rng_mode=rng_mode, max_mirror_bounces=64))
        assert np.array_equal(img, golden_images[key]), key


@pytest.mark.parametrize("key,scene,w,h,aa,n_rays,roulette,max_depth,seq", [
    ("path_demo_path_64x48_aa1_serial", "demo_path", 64, 48, 1, 10, 3, 5, 54),
    ("path_csg_example_64x48_aa1_serial", "csg_example", 64, 48, 1, 10, 3, 5, 54),
    ("path_demo_path_32x24_aa2_serial_seq7", "demo_path", 32, 24, 2, 3, 2, 4, 7),
    ("path_csg_example_32x24_aa2_serial_seq7", "csg_example", 32, 24, 2, 3, 2, 4, 7),
])
def test_path_mode_serial_rng_bit_identical(golden_images, scenes, key, scene, w, h, aa, n_rays, roulette, max_depth, seq):
    cfg = O.make_cfg(w, h, "path", spp_edge=aa, n_rays=n_rays, roulette=roulette, max_depth=max_depth, seq_number=seq,
                     rng_mode=SERIAL)
    img, stats = O.render(scenes[scene], cfg)
    assert np.array_equal(img, golden_images[key]), key
    assert stats.rays > w * h * aa * aa


@pytest.mark.parametrize("scene", ["demo_onoff", "demo_path", "csg_example"])
def test_world_ray_intersection_bit_identical(golden_traces, scenes, scene):
    rays = golden_traces[f"{scene}_rays"]
    r = np.zeros(len(rays), dtype=abi.RAY_DTYPE)
    r["origin"], r["direction"], r["tmin"], r["tmax"] = rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7]
    hits = O.trace_rays(scenes[scene], r)
    assert np.array_equal(hits["object"], golden_traces[f"{scene}_object"])
    rec = golden_traces[f"{scene}_rec"]
    hit = hits["object"] >= 0
    assert hit.sum() > 100
    assert np.array_equal(hits["t"][hit], rec[hit, 0])
    assert np.array_equal(hits["point"][hit], rec[hit, 1:4])
    assert np.array_equal(hits["normal"][hit], rec[hit, 4:7])
    assert np.array_equal(hits["uv"][hit], rec[hit, 7:9])


@pytest.mark.parametrize("scene", ["demo_onoff", "demo_path", "csg_example"])
def test_offset_if_visible_identical(golden_traces, scenes, scene):
    tri = golden_traces[f"{scene}_vis_in"]
    vis = O.offset_if_visible(scenes[scene], tri[:, 0:3], tri[:, 3:6], tri[:, 6:9])
    assert np.array_equal(vis, golden_traces[f"{scene}_vis_out"])
    assert 0 < vis.sum() < len(vis)


def test_csg_all_ray_intersections_tlists(golden_traces, scenes):
    rays = golden_traces["csg_example_rays"]
    rows = golden_traces["csg_example_tlists"]
    sc = scenes["csg_example"]
    nonempty = 0
    for row in rows:
        i, o, n = int(row[0]), int(row[1]), int(row[2])
        ray = O.mk_ray(rays[i, 0:3], rays[i, 3:6], float(rays[i, 6]), float(rays[i, 7]))
        hits = O.object_all_hits(sc, int(sc.objects[o]), ray)
        assert len(hits) == n
        want = row[3:3 + n].astype(np.float32)
        assert np.array_equal(hits["t"], want)
        nonempty += (n > 0 and o == 2)
    assert nonempty > 20
