"""Multi-GPU MSM / commitment: the index range is sharded over ranks (one process per GPU), every rank runs the
full single-GPU pipeline on its slice, and the per-rank affine partial sums are exchanged with ONE all_gather of
(2*E + 1) 64-bit words per rank (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) and added by
every rank (RCCL has no user-defined reduction, so curve points cannot ride an all_reduce).
The reference has no counterpart: its only parallelism is rayon on one host (groth16/src/msm.rs:17-20)."""
from __future__ import annotations

import numpy as np

from .lib import KG_G2


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; slices differ by at most one element (== kg_shard_range, the cut the
    single-process multi-device entries kg_commit_sharded / kg_sharded_key_* use)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_comm_streams = {}


def _comm_stream(torch, device):
    """One exchange queue per device: the 9-word all_gather must not queue behind the next MSM's accumulation on the
    compute stream (a collective issued on the compute stream would wait 1 ms for it and leave a bubble after it)."""
    key = (device.type, device.index)
    if key not in _comm_streams:
        _comm_streams[key] = torch.cuda.Stream(device=device)
    return _comm_streams[key]


def combine_partials(ctx, curve: int, partial_xy: np.ndarray, partial_inf: int, group=None, device=None):
    """all_gather the ranks' affine partial sums and add them (every rank gets the total).
    partial_xy: 2*E uint64 words (E = 4, or 8 for G2); returns (xy, inf)."""
    import contextlib
    import torch
    import torch.distributed as dist
    e2 = 16 if curve == KG_G2 else 8
    world = dist.get_world_size(group)
    mine = np.zeros(e2 + 1, dtype=np.int64)
    mine[:e2] = np.ascontiguousarray(partial_xy, dtype=np.uint64).view(np.int64)[:e2]
    mine[e2] = int(partial_inf)
    t = torch.from_numpy(mine)
    on_gpu = device is not None and torch.device(device).type == "cuda"
    with (torch.cuda.stream(_comm_stream(torch, torch.device(device))) if on_gpu else contextlib.nullcontext()):
        if device is not None:
            t = t.to(device)
        out = torch.empty(world * (e2 + 1), dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        h = out.cpu().numpy().view(np.uint64).reshape(world, e2 + 1)
    return ctx.points_sum_affine(curve, np.ascontiguousarray(h[:, :e2]), (h[:, e2] != 0).astype(np.uint8))


def sharded_commit(ctx, curve: int, d_bases: int, d_inf: int, d_scalars: int, n_local: int, group=None, device=None):
    """commit over the union of all ranks' local slices (device pointers to THIS rank's slice)."""
    xy, inf = ctx.commit(curve, d_bases, d_inf, d_scalars, n_local)
    return combine_partials(ctx, curve, xy, inf, group=group, device=device)
