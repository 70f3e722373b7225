"""Tone-mapping tail: HdrImage::average_luminosity / normalize_image / clamp_image and the
8-bit transform of write_ldr_image (reference include/colors.hpp:469-516, src/raytracer.cpp:176-186).

CPU: the oracle against the reference's own unit-test values (test/colors_test.cpp:206-244) and against fixtures made
by running the reference's HdrImage methods on two golden renders (tests/golden/make_tonemap_golden.py).
GPU: rt_average_luminosity / rt_tonemap through the C-ABI against the oracle.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ["flat_demo_path_160x120_aa2", "path_csg_example_96x72_aa6_serial"]


@pytest.fixture(scope="module")
def tone():
    return dict(np.load(os.path.join(GOLDEN, "tonemap.npz")))


def two_pixel_image():
    return np.array([[[5, 10, 15], [500, 1000, 1500]]], dtype=np.float32)


# ---- test/colors_test.cpp:206-244 -------------------------------------------------------------------------------------
def test_average_luminosity_known_answers():
    assert O.average_luminosity(two_pixel_image(), delta=0.0) == 100.0
    assert abs(O.average_luminosity(np.zeros((1, 1, 3), dtype=np.float32)) - 1e-10) < 1e-5


def test_normalize_image_known_answers():
    _, hdr, avg = O.tonemap(two_pixel_image(), alpha=10.0, gamma=1.0)
    # undo the clamp x/(1+x) to look at normalize_image alone
    norm = hdr.astype(np.float64) / (1.0 - hdr.astype(np.float64))
    assert np.allclose(norm[0, 0], (0.5, 1.0, 1.5), atol=1e-5) and np.allclose(norm[0, 1], (50, 100, 150), rtol=1e-4)


def test_clamp_image_known_answers():
    img = np.array([[[2e3, 4e5, 6e1], [1e2, 3e4, 5e7]]], dtype=np.float32)
    _, hdr, _ = O.tonemap(img, alpha=1.0, gamma=1.0, avg_luminosity=1.0)
    assert np.all((hdr >= 0) & (hdr <= 1))


def test_ldr_transform():
    """write_ldr_image's lambda: round(pow(x, 1/gamma) * 255) (colors.hpp:513-516)."""
    x = np.linspace(0, 4, 3999, dtype=np.float32).reshape(-1, 1, 3)
    ldr, hdr, _ = O.tonemap(x, alpha=1.0, gamma=2.2, avg_luminosity=1.0)
    want = np.round(np.power(hdr.astype(np.float64), 1 / np.float32(2.2)) * 255)
    assert np.max(np.abs(ldr.astype(np.int32) - want)) <= 1 and np.mean(ldr != want) < 1e-3


# ---- the reference's HdrImage methods run on golden renders ----------------------------------------------------------------
@pytest.mark.parametrize("key", KEYS)
def test_oracle_equals_reference_methods(golden_images, tone, key):
    img = golden_images[key]
    assert np.float32(O.average_luminosity(img)) == tone[key + "_avg"]
    _, hdr, avg = O.tonemap(img, alpha=0.18, gamma=1.0)
    assert np.array_equal(hdr, tone[key + "_hdr"])
    _, dark, _ = O.tonemap(img, alpha=0.18, gamma=1.0, avg_luminosity=0.1)
    assert np.array_equal(dark, tone[key + "_dark_hdr"])


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("key", KEYS)
def test_gpu_average_luminosity(golden_images, tone, key):
    import torch
    from raytracer_amd.tracer import average_luminosity
    img = torch.from_numpy(golden_images[key]).cuda()
    want = float(tone[key + "_avg"])
    serial = average_luminosity(img, serial_sum=True)     # the reference's summation order in binary32
    exact = average_luminosity(img)                       # order-independent fixed-point sum
    # device log10f may differ from glibc's in the last bit of a term; the sums agree far below the 8-bit step
    assert abs(serial - want) <= 2e-6 * want
    assert abs(exact - want) <= 2e-5 * want
    assert average_luminosity(img) == exact               # reproducible


@pytest.mark.gpu
@pytest.mark.parametrize("key", KEYS)
def test_gpu_tonemap_against_reference_floats_and_oracle_bytes(golden_images, tone, key):
    import torch
    from raytracer_amd.tracer import tonemap
    img_np = golden_images[key]
    img = torch.from_numpy(img_np).cuda()
    avg = float(tone[key + "_avg"])
    # same average -> normalize + clamp are two IEEE operations per component: bit-identical to the reference's floats
    ldr1, used, hdr = tonemap(img, alpha=0.18, gamma=1.0, avg_luminosity=avg, want_hdr=True)
    assert used == np.float32(avg) and np.array_equal(hdr.cpu().numpy(), tone[key + "_hdr"])
    want1, _, _ = O.tonemap(img_np, alpha=0.18, gamma=1.0, avg_luminosity=avg)
    assert np.array_equal(ldr1.cpu().numpy(), want1)                        # gamma 1: no pow, bytes identical
    # gamma 2.2: device powf vs glibc powf -> a component may land on the other side of a rounding boundary
    ldr2, _ = tonemap(img, alpha=0.18, gamma=2.2, avg_luminosity=avg)
    want2, _, _ = O.tonemap(img_np, alpha=0.18, gamma=2.2, avg_luminosity=avg)
    d = np.abs(ldr2.cpu().numpy().astype(np.int32) - want2.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) <= 1e-3
    # whole tail with the device's own average (dark mode and normal mode), host entry point
    ldr3, avg3 = tonemap(img_np, alpha=0.18, gamma=2.2)
    want3, _, _ = O.tonemap(img_np, alpha=0.18, gamma=2.2)
    d = np.abs(ldr3.astype(np.int32) - want3.astype(np.int32))
    assert abs(avg3 - avg) <= 2e-5 * avg and d.max() <= 1 and np.mean(d != 0) <= 5e-3
    ldr4, _ = tonemap(img_np, alpha=0.18, gamma=1.0, avg_luminosity=0.1)
    assert np.array_equal(ldr4, O.tonemap(img_np, alpha=0.18, gamma=1.0, avg_luminosity=0.1)[0])


@pytest.mark.gpu
def test_gpu_tonemap_odd_sizes_and_errors():
    import torch
    from raytracer_amd.lib import RtError
    from raytracer_amd.tracer import tonemap
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 63, 1001):   # 3n not a multiple of 4: the tail path of the map kernel
        img = rng.uniform(0, 3, size=(1, n, 3)).astype(np.float32)
        ldr, _ = tonemap(torch.from_numpy(img).cuda(), alpha=0.5, gamma=1.0, avg_luminosity=0.7)
        assert np.array_equal(ldr.cpu().numpy(), O.tonemap(img, alpha=0.5, gamma=1.0, avg_luminosity=0.7)[0])
        a = O.average_luminosity(img)
        from raytracer_amd.tracer import average_luminosity
        assert abs(average_luminosity(torch.from_numpy(img).cuda(), serial_sum=True) - a) <= 2e-6 * a
    with pytest.raises(RtError):
        tonemap(np.ones((2, 2, 3), dtype=np.float32), gamma=0.0)
