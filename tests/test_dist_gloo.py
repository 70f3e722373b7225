"""N>1 path on CPU: world_size-2, -3 and -8 `gloo` runs of the row sharding + gather that bench.py uses
(raytracer_amd/dist.py: whole frames by rank_rows / gather_rows, and the timed loop's shares by share_rows /
gather_shares), with the oracle rendering each rank's rows.  The assembled frame must equal the
single-process frame bit for bit (every sample's random stream depends only on its global index) and the
ranks' ray counts must add up to the whole frame's.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd.dist import max_local_rows, rank_rows, share_rows
from raytracer_amd.synth import synthetic_scene
from raytracer_amd.tracer import RowBands

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("height,world", [(1080, 1), (1080, 2), (1080, 8), (50, 3), (16, 4), (7, 2)])
def test_cyclic_bands_partition_the_rows(height, world):
    seen = np.concatenate([rank_rows(height, r, world).image_rows() for r in range(world)])
    assert sorted(seen.tolist()) == list(range(height))
    assert max_local_rows(height, world) == max(rank_rows(height, r, world).n_local_rows for r in range(world))
    # balanced to within one band
    counts = [rank_rows(height, r, world).n_local_rows for r in range(world)]
    assert max(counts) - min(counts) <= 16


def test_whole_is_the_one_rank_case():
    assert RowBands.whole(33).image_rows().tolist() == list(range(33))
    assert RowBands.cyclic(33, 0, 1).image_rows().tolist() == list(range(33))


@pytest.mark.parametrize("height,world,n_shares", [(1080, 1, 32), (1080, 8, 32), (1080, 4, 32), (50, 2, 8), (5, 2, 8)])
def test_shares_of_a_run_partition_the_rows(height, world, n_shares):
    """bench.py's timed loop: n_shares / world steps, every rank one share per step, are exactly one frame."""
    steps = n_shares // world
    seen = np.concatenate([share_rows(height, j, r, world, n_shares).image_rows() for j in range(steps) for r in range(world)])
    assert sorted(seen.tolist()) == list(range(height))
    counts = [share_rows(height, j, r, world, n_shares).n_local_rows for j in range(steps) for r in range(world)]
    assert max(counts) - min(counts) <= 1 and max(counts) == max_local_rows(height, n_shares)
    # the same step never hands one share to two ranks
    for j in range(steps):
        assert len({share_rows(height, j, r, world, n_shares).band_offset for r in range(world)}) == world


@pytest.mark.parametrize("world,mode,width,height,aa,n_shares", [
    (2, "path", 40, 50, 2, 0), (3, "flat", 32, 40, 1, 0), (2, "path", 24, 9, 1, 0),
    (8, "path", 32, 45, 1, 0),      # the node's rank count; 45 rows: ranks get 6 or 5
    (8, "flat", 24, 6, 1, 0),       # more ranks than rows: two ranks gather nothing
    (8, "path", 24, 40, 1, 16),     # bench.py's loop: 2 steps of 8 shares
    (2, "path", 24, 33, 1, 8),      # shares of unequal height (5 or 4 rows)
])
def test_sharded_frame_equals_single_process_frame(tmp_path, world, mode, width, height, aa, n_shares):
    out = tmp_path / "frame.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(out), mode, str(width),
           str(height), str(aa)] + ([str(n_shares)] if n_shares else [])
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.load(out)
    scene, _ = synthetic_scene(64)
    ref, stats = O.render(scene, O.make_cfg(width, height, mode, spp_edge=aa, n_rays=3, roulette=2, max_depth=3))
    assert np.array_equal(got["image"], ref)
    assert int(got["rays"][0]) == stats.rays
