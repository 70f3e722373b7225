"""N>1 path on CPU: world_size-2 and -3 `gloo` runs of the row-band sharding + gather that bench.py uses
(raytracer_amd/dist.py), with the oracle rendering each rank's bands.  The assembled frame must equal the
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
from raytracer_amd.dist import max_local_rows, rank_rows
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


@pytest.mark.parametrize("world,mode,width,height,aa", [(2, "path", 40, 50, 2), (3, "flat", 32, 40, 1), (2, "path", 24, 9, 1)])
def test_sharded_frame_equals_single_process_frame(tmp_path, world, mode, width, height, aa):
    out = tmp_path / "frame.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(out), mode, str(width),
           str(height), str(aa)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.load(out)
    scene, _ = synthetic_scene(64)
    ref, stats = O.render(scene, O.make_cfg(width, height, mode, spp_edge=aa, n_rays=3, roulette=2, max_depth=3))
    assert np.array_equal(got["image"], ref)
    assert int(got["rays"][0]) == stats.rays
