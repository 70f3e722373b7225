"""BRDF::scatter_ray on the device (rt_scatter_rays, through the C-ABI) against the CPU oracle's restatement of
DiffusiveBRDF::scatter_ray (reference include/materials.hpp:199-211, PCG::random_phong include/random.hpp:75-84,
ONB include/geometry.hpp:591-599) and SpecularBRDF::scatter_ray (include/materials.hpp:252-261), ray by ray on the same
(normal, generator state) pairs.

Bars, per component of the (unit-length) direction:
  * specular, reference formulas: bit-exact (sqrt, /, *, + only);
  * diffuse, reference formulas: 4e-7 absolute = 3.4 ulp of 1.0 (device acosf / sinf / cosf differ from glibc's in the
    last bit; products of two such factors);
  * the forms the default FMA path kernels run (rsq, sqrt(x) / sqrt(1 - x), v_sin / v_cos): 2e-6 absolute;
and the generator state after the call is exactly the oracle's (two draws for diffuse, none for specular).
Edge cases: random_float() == 1.0f for either draw (include/random.hpp:67-70 can return it), normals with z ~ -1 (the
negative branch of the Duff et al. basis), un-normalised normals over six decades of length.
"""
import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.tracer import scatter_rays

MULT = 6364136223846793005
MASK = (1 << 64) - 1
INC = (54 << 1) | 1
STATE_ONE = ((1 << 59) - 1) ^ ((1 << 41) - 1)  # XSH-RR of this state is 0xFFFFFFFF -> random_float() == 1.0f
STATE_THEN_ONE = ((STATE_ONE - INC) * pow(MULT, -1, 1 << 64)) & MASK  # its successor is STATE_ONE: the SECOND draw is 1.0f


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    normal = rng.normal(size=(n, 3)).astype(np.float32)
    normal *= (10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)      # HitRecord normals are not normalised
    k = n // 10
    normal[:k] = np.column_stack([rng.normal(scale=1e-3, size=k), rng.normal(scale=1e-3, size=k), -np.ones(k)]).astype(np.float32)
    normal[k] = (0, 0, -1)
    normal[k + 1] = (0, 0, 1)
    normal[k + 2] = (1e-20, 0, -1)
    in_dir = rng.normal(size=(n, 3)).astype(np.float32)
    state = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    state[0], state[1] = STATE_ONE, STATE_THEN_ONE
    state[k], state[k + 1] = STATE_ONE, STATE_THEN_ONE                         # 1.0f draws on the z = -1 / +1 normals too
    pcg = np.column_stack([state, np.full(n, INC, dtype=np.uint64)])
    return pcg, in_dir, normal


def _oracle(brdf, pcg, in_dir, normal):
    out = np.zeros((len(normal), 3), dtype=np.float32)
    state = np.zeros(len(normal), dtype=np.uint64)
    tmin = np.zeros(len(normal), dtype=np.float32)
    for i in range(len(normal)):
        g = O.PCG()
        g.st[0], g.st[1] = int(pcg[i, 0]), int(pcg[i, 1])
        r = O.scatter_ray(brdf, g, in_dir[i], (0.0, 0.0, 0.0), normal[i], 1)
        out[i] = r.direction[:]
        state[i] = g.st[0]
        tmin[i] = r.tmin
    return out, state, tmin


def test_one_point_zero_draws_exist():
    """The crafted states really give random_float() == 1.0f (first draw, second draw)."""
    g = O.PCG()
    g.st[0], g.st[1] = STATE_ONE, INC
    assert g.random_float() == 1.0
    g.st[0], g.st[1] = STATE_THEN_ONE, INC
    g.random_float()
    assert g.random_float() == 1.0


@pytest.mark.gpu
def test_diffuse_scatter_vs_oracle():
    pcg, in_dir, normal = _cases(10_000, 1)
    want, state, tmin = _oracle(abi.RT_BRDF_DIFFUSE, pcg, in_dir, normal)
    assert np.all(tmin == np.float32(1e-3))
    got, pcg_after = scatter_rays(abi.RT_BRDF_DIFFUSE, pcg, in_dir, normal, exact=True)
    assert np.array_equal(pcg_after[:, 0], state) and np.array_equal(pcg_after[:, 1], pcg[:, 1])
    err = np.abs(got - want)
    print(f"diffuse, reference formulas: max |d| = {err.max():.3g}, identical components {np.mean(err == 0):.3f}")
    assert err.max() <= 4e-7
    fast, pcg_fast = scatter_rays(abi.RT_BRDF_DIFFUSE, pcg, in_dir, normal, exact=False)
    assert np.array_equal(pcg_fast[:, 0], state)
    ferr = np.abs(fast - want)
    print(f"diffuse, FMA-kernel forms: max |d| = {ferr.max():.3g}, mean {ferr.mean():.3g}")
    assert ferr.max() <= 2e-6
    # what the estimator needs from the sample: unit length, on the normal's side (cos theta = sqrt(x) >= 0)
    n_hat = normal.astype(np.float64) / np.linalg.norm(normal.astype(np.float64), axis=1, keepdims=True)
    for d in (got, fast):
        assert np.abs(np.linalg.norm(d.astype(np.float64), axis=1) - 1).max() <= 2e-6
        assert (np.einsum("ij,ij->i", d.astype(np.float64), n_hat) >= -2e-6).all()
    # x == 1.0f: the direction is the normal itself
    assert np.abs(got[0] - n_hat[0]).max() <= 4e-7 and np.abs(fast[0] - n_hat[0]).max() <= 2e-6


@pytest.mark.gpu
def test_specular_scatter_vs_oracle():
    pcg, in_dir, normal = _cases(10_000, 2)
    want, state, tmin = _oracle(abi.RT_BRDF_SPECULAR, pcg, in_dir, normal)
    assert np.all(tmin == np.float32(1e-5)) and np.array_equal(state, pcg[:, 0])   # consumes no draw
    got, pcg_after = scatter_rays(abi.RT_BRDF_SPECULAR, pcg, in_dir, normal, exact=True)
    assert np.array_equal(pcg_after, pcg)
    assert np.array_equal(got, want)                                                   # bit-exact
    fast, _ = scatter_rays(abi.RT_BRDF_SPECULAR, pcg, in_dir, normal, exact=False)
    ferr = np.abs(fast - want)
    print(f"specular, FMA-kernel forms: max |d| = {ferr.max():.3g}")
    assert ferr.max() <= 2e-6


@pytest.mark.gpu
def test_scatter_rejects_bad_arguments():
    from raytracer_amd.lib import dll
    z = np.zeros((1, 3), dtype=np.float32)
    p = np.zeros((1, 2), dtype=np.uint64)
    assert dll().rt_scatter_rays(0, 7, 0, p.ctypes.data, z.ctypes.data, z.ctypes.data, 1, z.ctypes.data) == abi.RT_ERR_INVALID_ARGUMENT
    assert dll().rt_scatter_rays(0, 0, 64, p.ctypes.data, z.ctypes.data, z.ctypes.data, 1, z.ctypes.data) == abi.RT_ERR_INVALID_ARGUMENT
    assert dll().rt_scatter_rays(0, 0, 0, p.ctypes.data, z.ctypes.data, z.ctypes.data, 0, z.ctypes.data) == abi.RT_OK
