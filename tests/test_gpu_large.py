"""BASELINE.json's full sizes: oracle equality where the oracle finishes in seconds (MSM 2^20, NTT 2^20) and
size-independent properties beyond that (split-sum of an MSM, transform round trips, linearity)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x4B6F676172617368


@pytest.fixture(scope="module")
def ctx():
    import kogarashi_amd as K
    c = K.Context(0)
    yield c
    c.close()


def test_msm_2_20_matches_oracle_and_splits(ctx, oracle):
    import kogarashi_amd as K
    O, n = oracle, 1 << 20
    db, ds = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 1, 0, n, db.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 2, 0, n, ds.ptr)
    full = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    want_xy, want_inf = O.to_affine("g1", O.msm("g1", db.numpy(), ds.numpy(), None, threads=17))
    assert not want_inf and (full[:8] == want_xy).all()
    h = 333333                                           # ragged split
    a = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, h)
    b = ctx.msm(K.KG_G1, db.ptr + 64 * h, 0, ds.ptr + 32 * h, n - h)
    xy, inf = ctx.points_sum_affine(K.KG_G1, np.stack([a[:8], b[:8]]), np.zeros(2, dtype=np.uint8))
    assert inf == 0 and (xy == full[:8]).all()


def test_ntt_2_20_matches_oracle(ctx, oracle):
    import kogarashi_amd as K
    O, k = oracle, 20
    v = O.gen_scalars(0, SEED + 3, 0, 1 << k)
    fo, fg = O.Fft(k), K.Fft(k, ctx=ctx)
    assert (fg.dft(v) == fo.dft(v, threads=16)).all()
    assert (fg.coset_idft(v) == fo.coset_idft(v, threads=16)).all()


@pytest.mark.parametrize("k", [22, 24])
def test_ntt_round_trips_and_linearity(ctx, k):
    import kogarashi_amd as K
    n = 1 << k
    a, b, s = ctx.empty((n, 4)), ctx.empty((n, 4)), ctx.empty((n, 4))
    ctx.gen_scalars(K.KG_FR, SEED + 10 + k, 0, n, a.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 20 + k, 0, n, b.ptr)
    ctx.field_vec_op(K.KG_FR, "add", a.ptr, b.ptr, s.ptr, n)
    a0 = a.numpy()
    for inv, coset in ((False, False), (False, True)):
        ctx.ntt(a.ptr, k, inv, coset)
        ctx.ntt(a.ptr, k, not inv, coset)
        assert (a.numpy() == a0).all()                  # idft(dft(v)) = v, coset_idft(coset_dft(v)) = v
    ctx.ntt(a.ptr, k, False, False)
    ctx.ntt(b.ptr, k, False, False)
    ctx.ntt(s.ptr, k, False, False)
    ctx.field_vec_op(K.KG_FR, "add", a.ptr, b.ptr, a.ptr, n)
    assert (a.numpy() == s.numpy()).all()               # dft(a + b) = dft(a) + dft(b)


@pytest.mark.parametrize("curve,sfd", [(0, 0), (1, 1)])
def test_commit_2_24_properties(ctx, curve, sfd):
    """BASELINE.json configs[4] (Nova's commitment key: 2^24 bases) on one GPU, both curves: no oracle at this size, so
    size-independent properties of commit(m) = affine(sum g_i * m_i): the split sum over a ragged cut, and linearity in
    the scalars, commit(a + b) = commit(a) + commit(b) (the vector sum through kg_field_vec_op)."""
    import kogarashi_amd as K
    n = 1 << 24
    fld = K.KG_FR if sfd == 0 else K.KG_FQ
    g, a, b, s = ctx.empty((n, 8)), ctx.empty((n, 4)), ctx.empty((n, 4)), ctx.empty((n, 4))
    ctx.gen_bases(curve, SEED + 40 + curve, 0, n, g.ptr)
    ctx.gen_scalars(fld, SEED + 41, 0, n, a.ptr)
    ctx.gen_scalars(fld, SEED + 42, 0, n, b.ptr)
    ca, ia = ctx.commit(curve, g.ptr, 0, a.ptr, n)
    cb, ib = ctx.commit(curve, g.ptr, 0, b.ptr, n)
    assert not ia and not ib
    # split sum
    h = 5000001
    c1, i1 = ctx.commit(curve, g.ptr, 0, a.ptr, h)
    c2, i2 = ctx.commit(curve, g.ptr + 64 * h, 0, a.ptr + 32 * h, n - h)
    xy, inf = ctx.points_sum_affine(curve, np.stack([c1, c2]), np.array([i1, i2], dtype=np.uint8))
    assert inf == 0 and (xy == ca).all()
    # linearity
    ctx.field_vec_op(fld, "add", a.ptr, b.ptr, s.ptr, n)
    cs, is_ = ctx.commit(curve, g.ptr, 0, s.ptr, n)
    xy, inf = ctx.points_sum_affine(curve, np.stack([ca, cb]), np.array([ia, ib], dtype=np.uint8))
    assert not is_ and inf == 0 and (xy == cs).all()
