"""N > 1 path on CPU: world_size 2 over gloo.  The exchange + host-side point addition of kogarashi_amd.dist are
the product's; the per-rank partial (a GPU kernel in production) is supplied by the oracle here, since this
container has no GPU."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x4B6F676172617368


class HostOnlyCtx:
    """points_sum_affine is a host function of the C ABI and needs no device context."""

    def points_sum_affine(self, curve, pts, inf):
        import ctypes as C
        from kogarashi_amd import lib as L
        so = L.load()
        pts = np.ascontiguousarray(pts, dtype=np.uint64)
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        xy = np.zeros(16 if curve == 2 else 8, dtype=np.uint64)
        oi = C.c_uint8(0)
        rc = so.kg_points_sum_affine(None, curve, pts.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p), C.c_size_t(len(inf)),
                                     xy.ctypes.data_as(C.c_void_p), C.byref(oi))
        assert rc == 0
        return xy, int(oi.value)


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from kogarashi_amd import dist as kd
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = kd.shard_range(n, rank, world)
        bases = O.gen_bases(0, SEED + 1, lo, hi - lo, threads=2)
        scal = O.gen_scalars(0, SEED + 2, lo, hi - lo)
        if rank == 1:
            scal[:] = 0                                   # one rank contributes the identity
        xy, inf = O.to_affine("g1", O.msm("g1", bases, scal))
        total_xy, total_inf = kd.combine_partials(HostOnlyCtx(), 0, xy, inf)
        q.put((rank, total_xy.tobytes(), total_inf))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    from kogarashi_amd import dist as kd
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            parts = [kd.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1


def test_two_rank_commit_over_gloo(oracle):
    O = oracle
    n, world = 600, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                     # a free rendezvous port on the loop-back interface (as bench.py's launcher picks one)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from kogarashi_amd import dist as kd
    lo, hi = kd.shard_range(n, 0, world)
    bases = O.gen_bases(0, SEED + 1, 0, n)
    scal = O.gen_scalars(0, SEED + 2, 0, n)
    scal[hi:] = 0
    want_xy, want_inf = O.to_affine("g1", O.msm("g1", bases, scal))
    for _, xy, inf in res:
        assert inf == want_inf and xy == want_xy.tobytes()
