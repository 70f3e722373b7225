"""Row-band sharding of one frame over the ranks of one node, and the gather to rank 0.

The path shards by pixel rows with no exchange while rendering (every pixel sample is independent and its
random stream depends only on its global index), so the only collective is one gather of the finished
bands: each rank sends its compact [n_local_rows, W, 3] buffer to rank 0 over RCCL
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests), rank 0 scatters the rows into the
row-major HdrImage layout.  Rows are dealt round-robin one by one (BAND_ROWS = 1): neighbouring rows cost alike, so every
rank gets the same mix (1080 rows over 8 ranks: 135 each; 16-row bands would give 9 bands to some ranks and 8 to
others, 6 % apart before the content is even looked at), because path-mode cost is strongly row
dependent (sky rows cost ~1 ray per sample, object rows hundreds).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .tracer import RowBands

BAND_ROWS = 1


def rank_rows(height: int, rank: int, world: int, band_rows: int = BAND_ROWS) -> RowBands:
    return RowBands.cyclic(height, rank, world, band_rows)


def max_local_rows(height: int, world: int, band_rows: int = BAND_ROWS) -> int:
    return max(rank_rows(height, r, world, band_rows).n_local_rows for r in range(world))


def share_rows(height: int, step: int, rank: int, world: int, n_shares: int) -> RowBands:
    """One frame cut into n_shares row-cyclic shares (share k = rows k, k + n_shares, ...), dealt to the ranks step by step:
    in step `step` rank `rank` renders share (step * world + rank) mod n_shares.  Every share is the same mix of the frame's
    rows (they differ by < 2 % in cost), so the ranks of a step are balanced and a run of n_shares / world steps is the frame."""
    return RowBands.cyclic(height, (step * world + rank) % n_shares, n_shares, 1)


def gather_shares(local: torch.Tensor, rows_of_rank: list, full_out: torch.Tensor | None, padded: torch.Tensor, parts: list | None):
    """Gathers the ranks' buffers of one step (rank r rendered rows_of_rank[r], any RowBands) to rank 0 and places their rows
    in full_out.  padded: [max rows, W, 3] staging on every rank; parts: on rank 0 a list of `world` such tensors."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        idx = torch.as_tensor(rows_of_rank[0].image_rows(), device=full_out.device, dtype=torch.long)
        full_out.index_copy_(0, idx, local[: rows_of_rank[0].n_local_rows])
        return full_out
    padded[: local.shape[0]].copy_(local)
    if rank == 0:
        dist.gather(padded, gather_list=parts, dst=0)
        for r in range(world):
            idx = torch.as_tensor(rows_of_rank[r].image_rows(), device=full_out.device, dtype=torch.long)
            full_out.index_copy_(0, idx, parts[r][: rows_of_rank[r].n_local_rows].to(full_out.device))
        return full_out
    dist.gather(padded, gather_list=None, dst=0)
    return None


def gather_rows(local: torch.Tensor, height: int, full_out: torch.Tensor | None, band_rows: int = BAND_ROWS,
                padded: torch.Tensor | None = None, parts: list | None = None):
    """Gathers every rank's band buffer to rank 0 and assembles the image there.

    local:    [n_local_rows, W, 3] float32 on this rank's device
    full_out: [height, W, 3] on rank 0 (ignored elsewhere)
    padded / parts: optional preallocated staging ([max_rows, W, 3]; on rank 0 a list of `world` such
                    tensors) so that a timed loop allocates nothing.
    Returns full_out on rank 0, None elsewhere.  With world size 1 it is a copy.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        full_out.copy_(local)
        return full_out
    max_rows = max_local_rows(height, world, band_rows)
    if padded is None:
        padded = torch.empty((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]].copy_(local)
    if rank == 0:
        if parts is None:
            parts = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, gather_list=parts, dst=0)
        for r in range(world):
            rows = rank_rows(height, r, world, band_rows)
            idx = torch.as_tensor(rows.image_rows(), device=full_out.device, dtype=torch.long)
            full_out.index_copy_(0, idx, parts[r][: rows.n_local_rows])
        return full_out
    dist.gather(padded, gather_list=None, dst=0)
    return None


# -- animation batches ---------------------------------------------------------------------------
def frames_for_rank(n_frames: int, rank: int, world: int) -> range:
    """Frame-per-GPU parallelism for an animation (the reference's scripts/generate_image.sh loop over
    `--define-float clock=...`): frames are dealt round-robin, every rank keeps ONE flattened scene in HBM and only the
    camera changes from frame to frame (samples/demo_path_tracing.txt:3,40 — `rotation_z(clock) * translation(...)`)."""
    return range(rank, n_frames, world)


def orbit_camera(base, clock_degs: float):
    """The camera of the reference's animation: rotation_z(clock) * <camera transformation of the scene at clock = 0>,
    composed in binary32 exactly as the scene parser does (include/scenefiles.hpp:731-785)."""
    from .scene import Camera, compose, degs_to_rads, rotation_z
    return Camera(base.kind, compose(rotation_z(degs_to_rads(clock_degs)), base.transformation), base.asp_ratio, base.distance)
