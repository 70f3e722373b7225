"""One rank of the world_size-N CPU rehearsal of the sharded frame (tests/test_dist_gloo.py launches it).

Same host logic as bench.py's N>1 path — rank_rows -> render this rank's cyclic rows -> gather_rows to
rank 0, or (with a sixth argument n_shares) bench.py's timed loop: n_shares / world steps of share_rows -> render ->
gather_shares — with the `gloo` backend and the CPU oracle standing in for the device renderer (it takes the same
rt_render_cfg band fields as rt_render).  Rank 0 saves the assembled image.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from raytracer_amd.dist import gather_rows, gather_shares, max_local_rows, rank_rows, share_rows  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402


def main():
    out_path, mode, width, height, aa = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    scene, _ = synthetic_scene(64)
    n_shares = int(sys.argv[6]) if len(sys.argv) > 6 else 0

    def render(rows):
        cfg = O.make_cfg(width, height, mode, spp_edge=aa, n_rays=3, roulette=2, max_depth=3, band_rows=rows.band_rows,
                         band_stride=rows.band_stride, band_offset=rows.band_offset, n_local_rows=rows.n_local_rows)
        if rows.n_local_rows > 0:
            img, stats = O.render(scene, cfg)
            return torch.from_numpy(img), int(stats.rays)
        return torch.zeros((0, width, 3), dtype=torch.float32), 0

    full = torch.full((height, width, 3), float("nan"), dtype=torch.float32) if rank == 0 else None
    if n_shares:  # bench.py's loop: one share per rank and step; n_shares / world steps are the frame
        assert n_shares % world == 0
        padded = torch.empty((max_local_rows(height, n_shares), width, 3), dtype=torch.float32)
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        rays = 0
        for step in range(n_shares // world):
            local, r = render(share_rows(height, step, rank, world, n_shares))
            rays += r
            gather_shares(local, [share_rows(height, step, q, world, n_shares) for q in range(world)], full, padded, parts)
    else:
        local, rays = render(rank_rows(height, rank, world))
        padded = torch.empty((max_local_rows(height, world), width, 3), dtype=torch.float32)
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        gather_rows(local, height, full, padded=padded, parts=parts)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(out_path, image=full.numpy(), rays=total.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
