"""BASELINE.json's full sizes: oracle equality where the oracle finishes in seconds (G1 MSM 2^20, G2 MSM 2^16 / 2^18,
NTT 2^20 / 2^22, the Groth16 proof at 2^18 constraints) and size-independent properties beyond that (split-sum of an
MSM, transform round trips, linearity)."""
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


@pytest.mark.parametrize("curve,sfd,cv", [(0, 0, "g1"), (1, 1, "gk")])
def test_commit_2_24_matches_the_oracles_pippenger(ctx, oracle, curve, sfd, cv):
    """BASELINE.json configs[4] at full size against the oracle: commit(m) = affine(msm_curve_addition(g, m)) -- the
    restatement of groth16/src/msm.rs:6-48 runs the 2^24 pairs in ~10 s (one thread per window, c = 19) and is itself
    equal to the naive fold of nova/src/pedersen.rs:15-20 where both are feasible (tests/test_oracle_pinning.py)."""
    import kogarashi_amd as K
    n = 1 << 24
    fld = K.KG_FR if sfd == 0 else K.KG_FQ
    g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(curve, SEED + 40 + curve, 0, n, g.ptr)
    ctx.gen_scalars(fld, SEED + 41, 0, n, m.ptr)
    got_xy, got_inf = ctx.commit(curve, g.ptr, 0, m.ptr, n)
    want_xy, want_inf = oracle.to_affine(cv, oracle.msm(cv, g.numpy(), m.numpy(), None, threads=14))
    assert got_inf == want_inf == 0 and (got_xy == want_xy).all()
    # and with the key resident (kg_bases_register), as PedersenCommitment { g } is
    ctx.bases_register(curve, g.ptr, 0, n)
    reg_xy, reg_inf = ctx.commit(curve, g.ptr, 0, m.ptr, n)
    # and as pedersen.rs:15-20 is called: the key resident, the 2^24 scalars a host slice (eight index slices, uploads under the accumulations)
    host_xy, host_inf = ctx.commit_host_scalars(curve, g.ptr, 0, m.numpy(), n)
    ctx.bases_unregister(g.ptr)
    assert reg_inf == 0 and (reg_xy == want_xy).all()
    assert host_inf == 0 and (host_xy == want_xy).all()


def test_commit_2_24_of_a_witness_like_vector_matches_the_oracles_pippenger(ctx, oracle):
    """The same commitment of a vector distributed like a folded R1CS witness (half ones, a fifth zeros, a tenth -1: nova/src/relaxed_r1cs/
    witness.rs:56-70 meets bit decompositions): the wide windows (c = 20, nine-bit fine field) with hot buckets -- 8 M entries in one bucket, its
    tasks written by the whole workgroup (k_len_scatter), cut finer (bucket_task_len) and folded by up to 256 shares (k_hot_sum / k_hot_fold)."""
    import kogarashi_amd as K
    from kogarashi_amd import synthetic as syn
    n = 1 << 24
    g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 40, 0, n, g.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 43, 0, n, m.ptr)
    hm = m.numpy()
    syn.witness_like(hm, 29)
    m = ctx.upload(hm)
    want_xy, want_inf = oracle.to_affine("g1", oracle.msm("g1", g.numpy(), hm, None, threads=14))
    got_xy, got_inf = ctx.commit(K.KG_G1, g.ptr, 0, m.ptr, n)
    assert got_inf == want_inf == 0 and (got_xy == want_xy).all()
    ctx.bases_register(K.KG_G1, g.ptr, 0, n)
    reg_xy, reg_inf = ctx.commit(K.KG_G1, g.ptr, 0, m.ptr, n)
    ctx.bases_unregister(g.ptr)
    assert reg_inf == 0 and (reg_xy == want_xy).all()


def test_ntt_2_22_all_variants_match_oracle(ctx, oracle):
    """BASELINE.json configs[2] against the oracle's restatement of Fft<Fr> (fft.rs:92-127), all four transforms."""
    import kogarashi_amd as K
    O, k = oracle, 22
    n = 1 << k
    d = ctx.empty((n, 4))
    ctx.gen_scalars(K.KG_FR, SEED + 60, 0, n, d.ptr)
    v = d.numpy()
    fo = O.Fft(k)
    work = ctx.empty((n, 4))
    for name, inv, coset in (("dft", False, False), ("idft", True, False), ("coset_dft", False, True), ("coset_idft", True, True)):
        ctx.copy_d2d(work.ptr, d.ptr, n * 32)
        ctx.ntt(work.ptr, k, inv, coset)
        want = getattr(fo, name)(v, threads=16)
        assert (work.numpy() == want).all(), name


def _g2_bases(ctx, oracle, n, seed):
    """n G2 points k_i * G2 (cofactor != 1: SURVEY.md 8d draws G2 bases as generator multiples), made on the device
    (kg_fixed_base_mul == the oracle's scalar_point: tests/test_gpu_groth16.py) with two identities mixed in."""
    import kogarashi_amd as K
    dk = ctx.empty((n, 4))
    ctx.gen_scalars(K.KG_FR, seed, 0, n, dk.ptr)
    ctx.write(dk.ptr + 32 * 11, np.zeros((1, 4), dtype=np.uint64))          # k = 0: an identity base with its flag set
    dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
    ctx.fixed_base_mul(K.KG_G2, dk.ptr, n, dxy.ptr, dinf.ptr)
    return dxy, dinf


def _witness_like(O, scal, seed):
    """Groth16 witnesses are 0/1-heavy: half ones, a fifth zeros, some -1 and small values"""
    n = len(scal)
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 10, n)
    one = O.f_consts(0)["r"]
    scal[kind < 5] = one
    scal[(kind >= 5) & (kind < 7)] = 0
    scal[kind == 7] = O.f_neg(0, one)
    scal[kind == 8] = O.f_to_mont(0, np.array([5, 0, 0, 0], dtype=np.uint64))
    return scal


@pytest.mark.parametrize("log_n,skew", [(16, False), (16, True), (18, False), (18, True)])
def test_msm_g2_at_scale_matches_oracle(ctx, oracle, log_n, skew):
    """G2 through the two-pass sort (n >= 2^16), the two-wave k_acc_tasks<Fq2> and the lane-pair halving reduction
    (k_halve<Fp2S>), against the oracle's msm_curve_addition on the same bases -- plain and registered."""
    import kogarashi_amd as K
    O, n = oracle, 1 << log_n
    dxy, dinf = _g2_bases(ctx, O, n, SEED + 70 + log_n)
    scal = O.gen_scalars(0, SEED + 71 + log_n, 0, n)
    if skew:
        scal = _witness_like(O, scal, log_n)
    ds = ctx.upload(scal)
    bases, inf = dxy.numpy(), dinf.numpy()
    assert inf[11] == 1 and inf.sum() == 1
    wxy, winf = O.to_affine("g2", O.msm("g2", bases, scal, inf, threads=18))
    got = ctx.msm(K.KG_G2, dxy.ptr, dinf.ptr, ds.ptr, n)
    assert not winf and got[16:].any() and (got[:16] == wxy).all()
    ctx.bases_register(K.KG_G2, dxy.ptr, dinf.ptr, n)
    try:
        assert (ctx.msm(K.KG_G2, dxy.ptr, dinf.ptr, ds.ptr, n) == got).all()
    finally:
        ctx.bases_unregister(dxy.ptr)


def test_groth16_2_18_matches_oracle(ctx, oracle):
    """BASELINE.json configs[3]: Groth16 prove at 2^18 constraints.  CRS from the device setup (api.groth16_setup), proof
    from Prover.create_proof (blocking) and from two proofs in flight, against the oracle's create_proof (prover.rs:20-99)
    run on the host with the same CRS, witness and (r, s): the three affine points must be bit-identical."""
    import kogarashi_amd as K
    from kogarashi_amd import synthetic as syn
    from kogarashi_amd.api import groth16_setup
    O = oracle
    m = 1 << 18
    cc = syn.ChainCircuit(m)
    P = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=ctx)
    # the CRS itself at full size: the oracle's restatement of zksnark.rs gives every SCALAR of the 2^18-constraint setup in seconds; its
    # generator multiples (the slow part on a CPU) are checked on 96 sampled entries per vector, the edges and the instance wires included
    sc = O.groth16_setup_scalars(O.R1cs(cc.a, cc.b, cc.c, cc.x, cc.w), syn.fixed_toxic())
    rng = np.random.default_rng(18)
    for name, key, curve in (("h", "h", 0), ("l", "l", 0), ("a", "a", 0), ("b_g1", "b", 0), ("b_g2", "b", 2), ("ic", "ic", 0)):
        k = sc[key]
        assert len(P[name]) == len(k), name
        idx = np.unique(np.concatenate([[0, 1, 2, len(k) - 1], rng.integers(0, len(k), 92)])) if len(k) > 4 else np.arange(len(k))
        xy, inf = O.fixed_base_mul(curve, k[idx], threads=8)
        assert (P[name][idx] == xy).all() and (P[name + "_inf"][idx] == inf).all(), name
    r, s = syn.fixed_rs()
    prover = K.Prover(P, m, cc.l, cc.m_l_1, ctx=ctx)
    got = prover.create_proof(cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w, r, s)
    Pc = {name: P[name] for name in ("h", "l", "a", "b_g1", "b_g2")}
    Pc.update({name + "_inf": (P[name + "_inf"] if P[name + "_inf"].any() else None) for name in ("h", "l", "a", "b_g1", "b_g2")})
    Pc["vk_g1"], Pc["vk_g2"] = P["vk_g1"], P["vk_g2"]
    empty = (np.zeros(m + 1, dtype=np.uint64),) * 3
    cs = O.R1cs(empty, empty, empty, cc.x, cc.w)
    want = O.groth16_prove(cs, Pc, r, s, threads=32, evals=(cc.a_eval, cc.b_eval, cc.c_eval))
    for g, w_, name in zip(got[:3], want[:3], "ABC"):
        assert (g == w_).all(), name
    assert (got[3] == want[3]).all() and not got[3].any()
    jobs = [(cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w, r, s)] * 5
    for pr in prover.create_proofs(jobs):
        assert all((pr[i] == got[i]).all() for i in range(4))
    # the same CRS with window tables (kg_bases_precompute on all five vectors): merged bucket sets, identical proofs
    del prover
    tabled = K.Prover(P, m, cc.l, cc.m_l_1, ctx=ctx, window_tables=True)
    assert tabled.window_tables
    pr = tabled.create_proof(cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w, r, s)
    assert all((pr[i] == got[i]).all() for i in range(4))
    for pr in tabled.create_proofs(jobs):
        assert all((pr[i] == got[i]).all() for i in range(4))


def test_commit_between_2_23_and_2_24_is_the_sum_of_its_parts(ctx, oracle):
    """The wide window (c = 20, 13 windows, window groups) serves every length from 2^23 to 2^24: a ragged 12 000 001-pair commitment (blocking and
    through a ticket) against the sum of two parts that take other paths (2^22 + 7: c = 17 in three groups; the rest: index slices of the
    c = 17 sort or the wide window again), and 2^23 pairs against the oracle's Pippenger."""
    import kogarashi_amd as K
    n = 12_000_001
    g, a = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 60, 0, n, g.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 61, 0, n, a.ptr)
    whole, iw = ctx.commit(K.KG_G1, g.ptr, 0, a.ptr, n)
    ctx.msm_begin(K.KG_G1, g.ptr, 0, a.ptr, n, 0)
    tick = ctx.msm_end(K.KG_G1, 0)
    h = (1 << 22) + 7
    c1, i1 = ctx.commit(K.KG_G1, g.ptr, 0, a.ptr, h)
    c2, i2 = ctx.commit(K.KG_G1, g.ptr + 64 * h, 0, a.ptr + 32 * h, n - h)
    xy, inf = ctx.points_sum_affine(K.KG_G1, np.stack([c1, c2]), np.array([i1, i2], dtype=np.uint8))
    assert not iw and inf == 0 and (xy == whole).all()
    assert (ctx.msm(K.KG_G1, g.ptr, 0, a.ptr, n) == tick).all()
    m = 1 << 23
    got, gi = ctx.commit(K.KG_G1, g.ptr, 0, a.ptr, m)
    oxy, oinf = oracle.to_affine("g1", oracle.msm("g1", g.numpy()[:m], a.numpy()[:m], None, threads=14))
    assert not gi and not oinf and (got == oxy).all()
