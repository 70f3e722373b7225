"""bench.py's N > 1 path with the DEVICE renderer: 1 and 2 ranks sharing the one GPU of the test box (gloo, host-staged gather —
RCCL cannot hold one device twice), started by bench.py itself exactly as `python bench.py --gpus N` does.  Every rank count
walks the same 4-share frame completely (warm-up step + timed steps = 4 / N steps): the assembled frame must be the same file,
and the line must carry what the driver reads.  (tests/test_dist_gloo.py covers the same host logic on CPU with up to 8 ranks.)"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(tmp_path, n):
    pfm = tmp_path / f"frame_{n}.pfm"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(4 // n - 1), "--warmup", "1", "--spheres", "128",
           "--width", "320", "--height", "180", "--aa", "2", "--shares", "4", "--no-cpu-baseline", "--no-configs", "--backend", "gloo",
           "--save-pfm", str(pfm)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    return line, hashlib.sha256(pfm.read_bytes()).hexdigest()


@pytest.mark.timeout(600)
def test_one_and_two_ranks_assemble_the_same_frame(tmp_path):
    one, frame1 = run_bench(tmp_path, 1)
    two, frame2 = run_bench(tmp_path, 2)
    assert frame1 == frame2
    assert one["share0_sha256"] == two["share0_sha256"]
    for line, n in ((one, 1), (two, 2)):
        assert line["n_gpus"] == n and line["scaling"] == "weak" and line["unit"] == "Mrays/s" and line["value"] > 0
        assert line["steps"] == 4 // n - 1 and line["config"]["shares_per_frame"] == 4
        k = line["kernel_ms_per_step"]
        assert 0 < k["min_over_ranks"] <= k["max_over_ranks"]
        assert line["roofline"]["kernel"].startswith("k_render_path_tile")
    # two ranks render two shares per step: twice the rays per step of one rank (shares differ by a few per cent)
    assert 1.8 < two["config"]["rays_per_step"] / one["config"]["rays_per_step"] < 2.2
