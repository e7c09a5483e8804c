"""The multi-device entries of the C ABI (kg_commit_sharded, kg_msm_sharded, kg_sharded_key_*): one process, several
contexts, index-range sharding.  The GPU box has one device, so the contexts share device 0 -- the code path (a host thread
per context, per-context work spaces and streams, host-side sum of the partials) is the one an 8-GPU node runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x4B6F676172617368


@pytest.fixture(scope="module")
def ctxs():
    import kogarashi_amd as K
    cs = [K.Context(0) for _ in range(3)]
    yield cs
    for c in cs:
        c.close()


@pytest.mark.parametrize("cv,curve,sfd", [("g1", 0, 0), ("gk", 1, 1)])
@pytest.mark.parametrize("n_ctx", [1, 2, 3])
def test_commit_sharded_matches_single_context(ctxs, oracle, cv, curve, sfd, n_ctx):
    from kogarashi_amd import lib as L
    O, n = oracle, 20011
    bases = O.gen_bases(curve, SEED + 800 + curve, 0, n)
    scal = O.gen_scalars(sfd, SEED + 801 + curve, 0, n)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[0, 7777, n - 1]] = 1
    c0 = ctxs[0]
    db, di, ds = c0.upload(bases), c0.upload(inf), c0.upload(scal)
    want_xy, want_inf = c0.commit(curve, db.ptr, di.ptr, ds.ptr, n)
    oxy, oinf = O.to_affine(cv, O.msm(cv, bases, scal, inf, threads=8))
    assert want_inf == oinf == 0 and (want_xy == oxy).all()
    use = ctxs[:n_ctx]
    keep, pb, pi, ps, nl = [], [], [], [], []
    for r, c in enumerate(use):
        lo, hi = L.shard_range(n, r, n_ctx)
        arrs = (c.upload(bases[lo:hi]), c.upload(inf[lo:hi]), c.upload(scal[lo:hi]))
        keep.append(arrs)
        pb.append(arrs[0].ptr); pi.append(arrs[1].ptr); ps.append(arrs[2].ptr); nl.append(hi - lo)
    xy, oi = L.commit_sharded(use, curve, pb, pi, ps, nl)
    assert oi == 0 and (xy == want_xy).all()
    out = L.msm_sharded(use, curve, pb, pi, ps, nl)
    assert (out == c0.msm(curve, db.ptr, di.ptr, ds.ptr, n)).all()
    # without flag arrays (NULL), and with an empty slice
    xy2, oi2 = L.commit_sharded(use, curve, pb, None, ps, nl)
    w2 = c0.commit(curve, db.ptr, 0, ds.ptr, n)
    assert oi2 == w2[1] and (xy2 == w2[0]).all()
    nl0 = list(nl)
    nl0[-1] = 0
    cut = n - nl[-1]
    xy3, oi3 = L.commit_sharded(use, curve, pb, pi, ps, nl0)
    w3 = c0.commit(curve, db.ptr, di.ptr, ds.ptr, cut) if cut else (None, 1)
    assert oi3 == w3[1] and (oi3 or (xy3 == w3[0]).all())


def test_sharded_identity_and_cancellation(ctxs, oracle):
    """partials that cancel across devices (P on one, -P on the other) and all-zero scalars give the identity"""
    from kogarashi_amd import lib as L
    O = oracle
    b = O.gen_bases(0, SEED + 810, 0, 4)
    one = O.f_consts(0)["r"]
    s_pos = np.tile(one, (4, 1))
    s_neg = np.tile(O.f_neg(0, one), (4, 1))
    use = ctxs[:2]
    k0 = (use[0].upload(b), use[0].upload(s_pos))
    k1 = (use[1].upload(b), use[1].upload(s_neg))
    xy, inf = L.commit_sharded(use, 0, [k0[0].ptr, k1[0].ptr], None, [k0[1].ptr, k1[1].ptr], [4, 4])
    assert inf == 1
    out = L.msm_sharded(use, 0, [k0[0].ptr, k1[0].ptr], None, [k0[1].ptr, k1[1].ptr], [4, 4])
    assert not out[:4].any() and (out[4:8] == O.f_consts(1)["r"]).all() and not out[8:].any()      # (0, 1, 0)
    xy, inf = L.commit_sharded(use, 0, [0, 0], None, [0, 0], [0, 0])
    assert inf == 1


@pytest.mark.parametrize("cv,curve,sfd,w", [("g1", 0, 0, 8), ("gk", 1, 1, 8), ("g2", 2, 0, 16)])
def test_sharded_key_commit(ctxs, oracle, cv, curve, sfd, w):
    """PedersenCommitment spread over contexts: key of 2^k + 1 generators (pedersen.rs:10-13), commits of full, shorter
    (zip semantics) and one-element vectors against kg_commit on one context and the oracle's naive fold."""
    from kogarashi_amd import lib as L
    O = oracle
    n = (1 << 11) + 1 if curve != 2 else 300
    if curve == 2:
        bases, _ = O.fixed_base_mul(2, O.gen_scalars(0, SEED + 820, 0, n))
    else:
        bases = O.gen_bases(curve, SEED + 821, 0, n)
    inf = np.zeros(n, dtype=np.uint8)
    inf[5] = 1
    c0 = ctxs[0]
    db, di = c0.upload(bases), c0.upload(inf)
    for n_ctx in (1, 2, 3):
        key = L.ShardedKey(ctxs[:n_ctx], curve, bases, inf)
        assert len(key) == n
        for cnt in (n, n - 700 if n > 1000 else n - 70, 1, 2):
            m = O.gen_scalars(sfd, SEED + 822 + cnt, 0, cnt)
            ds = c0.upload(m)
            want = c0.commit(curve, db.ptr, di.ptr, ds.ptr, cnt)
            got = key.commit(m)
            assert got[1] == want[1] and (got[0] == want[0]).all(), (n_ctx, cnt)
        m = O.gen_scalars(sfd, SEED + 830, 0, n + 50)             # longer than the key: the extra scalars are ignored
        want = c0.commit(curve, db.ptr, di.ptr, c0.upload(m[:n]).ptr, n)
        got = key.commit(m)
        assert got[1] == want[1] and (got[0] == want[0]).all()
        if curve != 2:
            small = 40
            oxy, oinf = O.commit_naive(cv, bases[:small], m[:small], inf[:small])
            got = key.commit(m[:small])
            assert got[1] == oinf and (got[0] == oxy).all()
        key.close()


@pytest.mark.parametrize("n_ctx", [1, 2, 3, 4])
@pytest.mark.parametrize("m", [37, 1024])
def test_groth16_proof_over_several_contexts(oracle, n_ctx, m):
    """kg_groth16_prove_sharded (the MSMs of prover.rs:51-65 spread task-parallel, G2 query | G1 queries |
    transforms + h) against the single-context proof and the oracle; the contexts share device 0 on the one-GPU box."""
    import kogarashi_amd as K
    O = oracle
    cs = O.chain_r1cs(m, O.gen_scalars(0, SEED + 700 + m, 0, 1)[0])
    params = O.groth16_params(cs, O.gen_scalars(0, SEED + 701, 0, 5), threads=8)
    r, s = O.gen_scalars(0, SEED + 702, 0, 2)
    a, b, c = cs.evaluate()
    want = O.groth16_prove(cs, params, r, s, evals=(a, b, c))
    params["vk_g2"] = params["vk_g2"][:2]
    ctxs = [K.Context(0) for _ in range(n_ctx)]
    try:
        sp = K.ShardedProver(params, cs.m, cs.l, cs.m_l_1, ctxs)
        for _ in range(2):                                  # twice: the contexts' jobs and buffers are reusable
            got = sp.create_proof(a, b, c, cs.x, cs.w, r, s)
            for g, w_, name in zip(got[:3], want[:3], "ABC"):
                assert (g == w_).all(), (name, n_ctx)
            assert (got[3] == want[3]).all()
        single = K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=ctxs[0]).create_proof(a, b, c, cs.x, cs.w, r, s)
        assert all((g == w_).all() for g, w_ in zip(single[:3], want[:3]))
        del sp
    finally:
        for c_ in ctxs:
            c_.close()


def test_sharded_proof_rejects_identity_delta_and_bad_arguments(oracle):
    import kogarashi_amd as K
    from kogarashi_amd.lib import ProverSubVersionCrsAttack, KogarashiError
    O = oracle
    cs = O.chain_r1cs(4, O.gen_scalars(0, SEED + 710, 0, 1)[0])
    params = O.groth16_params(cs, O.gen_scalars(0, SEED + 711, 0, 5))
    a, b, c = cs.evaluate()
    r, s = O.gen_scalars(0, SEED + 712, 0, 2)
    want = O.groth16_prove(cs, params, r, s, evals=(a, b, c))
    params["vk_g2"] = params["vk_g2"][:2]
    ctxs = [K.Context(0) for _ in range(2)]
    try:
        bad = dict(params, delta_g2_inf=1)
        with pytest.raises(ProverSubVersionCrsAttack):
            K.ShardedProver(bad, cs.m, cs.l, cs.m_l_1, ctxs).create_proof(a, b, c, cs.x, cs.w, r, s)
        sp = K.ShardedProver(params, cs.m, cs.l, cs.m_l_1, ctxs)
        sp.ctxs = [ctxs[0], ctxs[0]]                        # the same context twice
        with pytest.raises(KogarashiError):
            sp.create_proof(a, b, c, cs.x, cs.w, r, s)
        sp.ctxs = ctxs
        got = sp.create_proof(a, b, c, cs.x, cs.w, r, s)    # and the contexts are still usable afterwards
        assert all((g == w_).all() for g, w_ in zip(got[:3], want[:3]))
        del sp
    finally:
        for c_ in ctxs:
            c_.close()
