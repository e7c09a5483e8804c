"""kg_msm_host_scalars / kg_commit_host_scalars: the per-call shape of the reference's call sites -- the bases are fixed and
resident (groth16/src/params.rs:6-28 CRS vectors, nova/src/pedersen.rs:6-13 generators), the scalars are a fresh host slice
(groth16/src/msm.rs:6 `coeffs`, pedersen.rs:15 `m`).  The scalars travel in index slices, each sorted and accumulated while the
next one is on the bus; every result must be the point kg_msm gives on the same pairs, and the oracle's."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x4B6F676172617368
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import kogarashi_amd as K
    c = K.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("cv,curve,sfd,w", [("g1", 0, 0, 8), ("gk", 1, 1, 8), ("g2", 2, 0, 16)])
def test_host_scalars_match_oracle_at_every_slice_count(ctx, oracle, cv, curve, sfd, w):
    """empty, one pair, below the first cut (one slice), 2 slices (2^17), 4 slices (2^19: G1 only -- the G2 oracle is slow); ragged lengths;
    identity bases at the slice seams; registered, unregistered and offset-into-registered bases; zero scalars"""
    O = oracle
    sizes = [0, 1, 7, 1000, (1 << 17) + 5] + ([(1 << 19) + 3] if curve == 0 else [])
    nmax = max(sizes)
    if curve == 2:                                  # G2 bases: k_i * G2 from the device (checked against the oracle in test_gpu_groth16.py)
        dk = ctx.upload(O.gen_scalars(0, SEED + 899, 0, nmax))
        dxy, dinf = ctx.empty((nmax, 16)), ctx.empty((nmax,), dtype=np.uint8)
        ctx.fixed_base_mul(2, dk.ptr, nmax, dxy.ptr, dinf.ptr)
        bases = dxy.numpy()
    else:
        bases = O.gen_bases(curve, SEED + 900 + curve, 0, nmax)
    scal = O.gen_scalars(sfd, SEED + 901 + curve, 0, nmax)
    scal[[0, 5, nmax // 3]] = 0
    inf = np.zeros(nmax, dtype=np.uint8)
    inf[[2, nmax // 3, nmax // 3 + 1, nmax // 2, nmax - 1]] = 1
    db, di = ctx.upload(bases), ctx.upload(inf)
    for registered in (False, True):
        if registered:
            ctx.bases_register(curve, db.ptr, di.ptr, nmax)
        for n in sizes:
            got = ctx.msm_host_scalars(curve, db.ptr, di.ptr, scal[:n], n)
            ds = ctx.upload(scal[:max(n, 1)])
            assert (got == ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)).all(), (cv, n, registered)
            if not registered and (curve != 2 or n <= 1000):
                want_xy, want_inf = O.to_affine(cv, O.msm(cv, bases[:n], scal[:n], inf[:n], threads=8))
                xy, oi = ctx.commit_host_scalars(curve, db.ptr, di.ptr, scal[:n], n)
                assert oi == want_inf and (want_inf or (xy == want_xy).all()), (cv, n)
                assert oi or (got[:w] == xy).all()
        # a whole-point offset into the array (params.a[cs.l()..]): served from the resident copy when registered
        off, n = 4097, (1 << 17) + 5 - 4097
        got = ctx.msm_host_scalars(curve, db.ptr + off * w * 8, di.ptr + off, scal[:n], n)
        ds = ctx.upload(scal[:n])
        assert (got == ctx.msm(curve, db.ptr + off * w * 8, di.ptr + off, ds.ptr, n)).all()
    ctx.bases_unregister(db.ptr)


def test_host_scalars_2_20_is_the_resident_point_and_the_oracles(ctx, oracle):
    """BASELINE configs[1] through the host-scalar entry: bases registered once, 2^20 scalars from pageable host memory per call"""
    import kogarashi_amd as K
    O, n = oracle, 1 << 20
    db, ds = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED, 0, n, db.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 1, 0, n, ds.ptr)
    ctx.bases_register(K.KG_G1, db.ptr, 0, n)
    hs = ds.numpy()
    want = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    for _ in range(3):                                    # cached upload buffers, both scalar-side sets
        assert (ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, hs, n) == want).all()
    oxy, oinf = O.to_affine("g1", O.msm("g1", db.numpy(), hs, None, threads=14))
    assert not oinf and (want[:8] == oxy).all()
    # a witness-like vector (hot buckets in every slice)
    from kogarashi_amd import synthetic as syn
    syn.witness_like(hs, 31)
    ds2 = ctx.upload(hs)
    assert (ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, hs, n) == ctx.msm(K.KG_G1, db.ptr, 0, ds2.ptr, n)).all()
    ctx.bases_unregister(db.ptr)


def test_host_scalars_after_pending_device_work_and_between_tickets(ctx, oracle):
    """stream semantics: bases produced on the context's stream right before the call (kg_fixed_base_mul) are complete when the
    slices read them; tickets of kg_msm_begin may be in flight around the call (disjoint result slots)"""
    import kogarashi_amd as K
    O, n = oracle, (1 << 17) + 77
    k = ctx.upload(O.gen_scalars(0, SEED + 910, 0, n))
    scal = O.gen_scalars(0, SEED + 911, 0, n)
    ds = ctx.upload(scal)
    db, di = ctx.empty((n, 8)), ctx.empty((n,), dtype=np.uint8)
    ob = ctx.upload(O.gen_bases(0, SEED + 912, 0, 4096))
    ctx.msm_begin(K.KG_G1, ob.ptr, 0, ds.ptr, 4096, 0)
    ctx.fixed_base_mul(K.KG_G1, k.ptr, n, db.ptr, di.ptr)             # no sync: the MSM below follows it in stream order
    got = ctx.msm_host_scalars(K.KG_G1, db.ptr, di.ptr, scal, n)
    t0 = ctx.msm_end(K.KG_G1, 0)
    assert (got == ctx.msm(K.KG_G1, db.ptr, di.ptr, ds.ptr, n)).all()
    assert (t0 == ctx.msm(K.KG_G1, ob.ptr, 0, ds.ptr, 4096)).all()


def test_forced_slice_counts_give_the_same_point():
    """KG_HOST_SLICES / KG_HOST_FIRST_DIV (tuning.h): 1..8 slices, first slice down to a sixteenth of a share, ragged length"""
    script = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import kogarashi_amd as K
from oracle import oracle as O
SEED = 0x4B6F676172617368
ctx = K.Context(0)
n = (1 << 16) + 12345
b = O.gen_bases(0, SEED + 920, 0, n); s = O.gen_scalars(0, SEED + 921, 0, n)
db, ds = ctx.upload(b), ctx.upload(s)
ctx.bases_register(K.KG_G1, db.ptr, 0, n)
want = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
assert (ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, s, n) == want).all()
assert (ctx.msm_host(K.KG_G1, b, None, s, n) == want).all()
assert (ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, s[:5], 5) == ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, 5)).all()
print("ok")
""" % ROOT
    for knobs in ({"KG_HOST_SLICES": "1"}, {"KG_HOST_SLICES": "3", "KG_HOST_FIRST_DIV": "1"}, {"KG_HOST_SLICES": "8", "KG_HOST_FIRST_DIV": "16"},
                  {"KG_HOST_SLICES": "5", "KG_HOST_FIRST_DIV": "3"}):
        r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, **knobs), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (knobs, r.stderr[-2000:])


@pytest.mark.parametrize("log_n", [16, 18, 19])
def test_window_tables_serve_short_blocking_calls_and_host_scalars(ctx, oracle, log_n):
    """Registered bases with window tables (kg_bases_precompute): a blocking kg_msm / kg_msm_host_scalars of up to 2^18 pairs goes through the
    tables (merged sort), a longer one runs in window groups against the resident copy -- either way the point of the plain call, and the oracle's."""
    import kogarashi_amd as K
    O, n = oracle, 1 << log_n
    db, ds = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 40 + log_n, 0, n, db.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 41 + log_n, 0, n, ds.ptr)
    hs = ds.numpy()
    flags = np.zeros(n, dtype=np.uint8)
    flags[[3, n // 3 - 1, n // 3, n // 3 + 1, n // 2, n - 1]] = 1      # identity bases, some at the seams of the index slices (the table rows carry them)
    di = ctx.upload(flags)
    plain = ctx.msm(K.KG_G1, db.ptr, di.ptr, ds.ptr, n)
    ctx.bases_register(K.KG_G1, db.ptr, di.ptr, n)
    try:
        ctx.bases_precompute(db.ptr)
        ctx.sync()
        for _ in range(2):
            assert (ctx.msm(K.KG_G1, db.ptr, di.ptr, ds.ptr, n) == plain).all()
            assert (ctx.msm_host_scalars(K.KG_G1, db.ptr, di.ptr, hs, n) == plain).all()      # 2^19: two index slices, each through the table's rows at its offset
        xy, inf = ctx.commit_host_scalars(K.KG_G1, db.ptr, di.ptr, hs, n)
        # a shorter call against the same array, and one at an offset into it
        m = n - 4097
        assert (ctx.msm_host_scalars(K.KG_G1, db.ptr, di.ptr, hs[:m], m) == ctx.msm(K.KG_G1, db.ptr, di.ptr, ds.ptr, m)).all()
        assert (ctx.msm_host_scalars(K.KG_G1, db.ptr + 4097 * 64, di.ptr + 4097, hs[:m], m) == ctx.msm(K.KG_G1, db.ptr + 4097 * 64, di.ptr + 4097, ds.ptr, m)).all()
    finally:
        ctx.bases_unregister(db.ptr)
    oxy, oinf = O.to_affine("g1", O.msm("g1", db.numpy(), hs, flags, threads=8))
    assert not oinf and not inf and (xy == oxy).all() and (plain[:8] == oxy).all()
