"""Groth16 prover and fixed-base multiples on the GPU vs the oracle's restatement of groth16/src/{zksnark,prover}.rs.
The proof is unique given (CRS, witness, r, s); parity is bit equality of the three affine points."""
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


@pytest.mark.parametrize("curve,sfd,w", [(0, 0, 8), (1, 1, 8), (2, 0, 16)])
def test_fixed_base_mul(ctx, oracle, curve, sfd, w):
    O = oracle
    n = 70 if curve != 2 else 24
    k = O.gen_scalars(sfd, SEED + 300 + curve, 0, n)
    k[0] = 0
    k[1] = O.f_consts(sfd)["r"]                   # 1 * G
    k[2] = O.f_neg(sfd, O.f_consts(sfd)["r"])     # -1 * G
    dk = ctx.upload(k)
    dxy, dinf = ctx.empty((n, w)), ctx.empty((n,), dtype=np.uint8)
    ctx.fixed_base_mul(curve, dk.ptr, n, dxy.ptr, dinf.ptr)
    want_xy, want_inf = O.fixed_base_mul(curve, k)
    assert (dinf.numpy() == want_inf).all() and want_inf[0] == 1
    assert (dxy.numpy() == want_xy).all()


@pytest.mark.parametrize("m", [1, 2, 5, 16, 100, 1024])
def test_groth16_proof_matches_oracle(ctx, oracle, m):
    import kogarashi_amd as K
    O = oracle
    t0 = O.gen_scalars(0, SEED + 400 + m, 0, 1)[0]
    cs = O.chain_r1cs(m, t0)
    toxic = O.gen_scalars(0, SEED + 401, 0, 5)
    params = O.groth16_params(cs, toxic, threads=8)
    r, s = O.gen_scalars(0, SEED + 402, 0, 2)
    a, b, c = cs.evaluate()
    want = O.groth16_prove(cs, params, r, s, evals=(a, b, c))
    params["vk_g2"] = params["vk_g2"][:2]
    prover = K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=ctx)
    got = prover.create_proof(a, b, c, cs.x, cs.w, r, s)
    for g, w_, name in zip(got[:3], want[:3], "ABC"):
        assert (g == w_).all(), name
    assert (got[3] == want[3]).all()


def test_groth16_rejects_identity_delta(ctx, oracle):
    import kogarashi_amd as K
    from kogarashi_amd.lib import ProverSubVersionCrsAttack
    O = oracle
    cs = O.chain_r1cs(4, O.gen_scalars(0, SEED + 410, 0, 1)[0])
    params = O.groth16_params(cs, O.gen_scalars(0, SEED + 411, 0, 5))
    params["vk_g2"] = params["vk_g2"][:2]
    params["delta_g1_inf"] = 1
    a, b, c = cs.evaluate()
    r, s = O.gen_scalars(0, SEED + 412, 0, 2)
    with pytest.raises(ProverSubVersionCrsAttack):
        K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=ctx).create_proof(a, b, c, cs.x, cs.w, r, s)


def test_r1cs_evaluate_matches_oracle(ctx, oracle):
    """cs.evaluate() (zkstd/src/r1cs.rs:137-142) as a CSR SpMV on the device, on a random sparse system plus the chain."""
    O = oracle
    rng = np.random.default_rng(3)
    m, nv = 3000, 500
    counts = rng.integers(0, 6, m)
    counts[7] = 0                                   # empty row -> 0
    rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    col = rng.integers(0, nv, int(rp[-1])).astype(np.uint64)
    val = O.gen_scalars(0, SEED + 500, 0, int(rp[-1]))
    z = O.gen_scalars(0, SEED + 501, 0, nv)
    want = np.empty((m, 4), dtype=np.uint64)
    import ctypes as C
    O.lib().kgo_r1cs_evaluate(O._p(rp), O._p(col), O._p(val), C.c_size_t(m), O._p(z), O._p(want))
    drp, dcol, dval, dz, dout = ctx.upload(rp), ctx.upload(col), ctx.upload(val), ctx.upload(z), ctx.empty((m, 4))
    ctx.r1cs_evaluate(drp.ptr, dcol.ptr, dval.ptr, m, dz.ptr, dout.ptr)
    assert (dout.numpy() == want).all()
    cs = O.chain_r1cs(64, O.gen_scalars(0, SEED + 502, 0, 1)[0])
    zz = np.concatenate([cs.x, cs.w])
    dz = ctx.upload(zz)
    for (rp, col, val), want in zip((cs.a, cs.b, cs.c), cs.evaluate()):
        d = [ctx.upload(np.ascontiguousarray(t)) for t in (rp, col, val)]
        out = ctx.empty((cs.m, 4))
        ctx.r1cs_evaluate(d[0].ptr, d[1].ptr, d[2].ptr, cs.m, dz.ptr, out.ptr)
        assert (out.numpy() == want).all()


class _FrOps:
    """five host-side scalar operations for groth16_setup, served by the oracle in this test"""

    def __init__(self, O):
        self.O = O

    def one(self):
        return self.O.f_consts(0)["r"]

    def inv(self, x):
        return self.O.f_invert(0, x)

    def mul(self, x, y):
        return self.O.f_mul(0, x, y)

    def sub(self, x, y):
        return self.O.f_sub(0, x, y)

    def pow2k(self, x, k):
        for _ in range(k):
            x = self.O.f_square(0, x)
        return x


@pytest.mark.parametrize("m", [1, 2, 6, 64, 1000, 1 << 14])
def test_device_setup_matches_oracle_and_proves(ctx, oracle, m):
    """ZkSnark::setup behind the C ABI (kg_groth16_setup_bn254: powers, idft, device-side transposition, transposed products,
    axpy / scale, generator multiples) equals the oracle's restatement of zksnark.rs element for element; the proof made with it
    matches too.  m = 1, 2: the smallest circuits (n = 2); 2^14: the windowed generator tables, long transposed rows."""
    import kogarashi_amd as K
    from kogarashi_amd.api import groth16_setup
    O = oracle
    cs = O.chain_r1cs(m, O.gen_scalars(0, SEED + 600 + m, 0, 1)[0])
    toxic = O.gen_scalars(0, SEED + 601, 0, 5)
    want = O.groth16_params(cs, toxic, threads=8)
    got = groth16_setup(cs.a, cs.b, cs.c, cs.m, cs.l, cs.m_l_1, toxic, _FrOps(O), ctx=ctx)
    for name in ("h", "l", "a", "b_g1", "b_g2", "ic"):
        assert (got[name] == want[name]).all(), name
        assert (got[name + "_inf"] == want[name + "_inf"]).all(), name
    assert (got["vk_g1"] == want["vk_g1"]).all() and (got["vk_g2"] == want["vk_g2"][:2]).all()
    assert (got["gamma_g2"] == want["vk_g2"][2]).all()
    r, s = O.gen_scalars(0, SEED + 602, 0, 2)
    a, b, c = cs.evaluate()
    proof = K.Prover(got, cs.m, cs.l, cs.m_l_1, ctx=ctx).create_proof(a, b, c, cs.x, cs.w, r, s)
    ref = O.groth16_prove(cs, want, r, s, evals=(a, b, c))
    assert all((g == w_).all() for g, w_ in zip(proof[:3], ref[:3]))


@pytest.mark.parametrize("m,l,m_l_1,seed", [(37, 3, 50, 1), (500, 2, 300, 2), (1, 1, 0, 3), (260, 5, 700, 4)])
def test_device_setup_on_random_sparse_systems(ctx, oracle, m, l, m_l_1, seed):
    """The setup's transposition and transposed products on matrices that are NOT the chain circuit: random rows of 0..6 entries (empty rows,
    columns nobody uses, the same column twice in a row), one wire that most rows use (a heavy column: the wave-aggregated atomics), one
    row that names every wire (a long row), more wires than constraints and the other way round -- every CRS element against the
    oracle's restatement of zksnark.rs:131-194 (eval / eval_at_tau fold the entries in storage order; the device in any order)."""
    from kogarashi_amd.api import groth16_setup
    O = oracle
    rng = np.random.default_rng(seed)
    nv = l + m_l_1

    def matrix(k):
        rp, col = [0], []
        for i in range(m):
            cnt = int(rng.integers(0, 7))
            if nv and i == m // 2 and k == 1:
                cols = list(range(nv))                                 # the long row
            elif nv:
                cols = [int(c) for c in rng.integers(0, nv, cnt)]
                if cnt >= 2 and rng.random() < 0.3:
                    cols[1] = cols[0]                                  # a duplicate entry
                if rng.random() < 0.7 and k != 2:
                    cols.append(0)                                     # the heavy column
            else:
                cols = []
            col += cols
            rp.append(len(col))
        val = O.gen_scalars(0, SEED + 700 + 10 * seed + k, 0, max(len(col), 1))[: len(col)]
        return (np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint64), np.ascontiguousarray(val).reshape(-1, 4))
    a, b, c = matrix(0), matrix(1), matrix(2)
    x = O.gen_scalars(0, SEED + 790 + seed, 0, max(l, 1))[:l]
    w = O.gen_scalars(0, SEED + 795 + seed, 0, max(m_l_1, 1))[:m_l_1]
    cs = O.R1cs(a, b, c, x, w)
    toxic = O.gen_scalars(0, SEED + 799 + seed, 0, 5)
    want = O.groth16_params(cs, toxic, threads=8)
    got = groth16_setup(a, b, c, m, l, m_l_1, toxic, ctx=ctx)
    for name in ("h", "l", "a", "b_g1", "b_g2", "ic"):
        assert got[name].shape == want[name].shape and (got[name] == want[name]).all(), name
        assert (got[name + "_inf"] == want[name + "_inf"]).all(), name
    assert (got["vk_g1"] == want["vk_g1"]).all() and (got["vk_g2"] == want["vk_g2"][:2]).all() and (got["gamma_g2"] == want["vk_g2"][2]).all()


def test_setup_status_codes(ctx, oracle):
    """gamma = 0 or delta = 0 -> Error::ProverInversionFailed (zksnark.rs:37-38); alpha = 0 is legal (alpha_g1 is the identity, flagged);
    missing output arrays and m = 0 are KG_ERR_BAD_ARG; nothing aborts and the context keeps working"""
    import kogarashi_amd as K
    from kogarashi_amd.api import groth16_setup
    from kogarashi_amd.lib import ProverInversionFailed, Groth16Crs, KogarashiError
    O = oracle
    cs = O.chain_r1cs(6, O.gen_scalars(0, SEED + 610, 0, 1)[0])
    toxic = O.gen_scalars(0, SEED + 611, 0, 5)
    for z in (2, 3):
        bad = toxic.copy(); bad[z] = 0
        with pytest.raises(ProverInversionFailed):
            groth16_setup(cs.a, cs.b, cs.c, cs.m, cs.l, cs.m_l_1, bad, ctx=ctx)
    za = toxic.copy(); za[0] = 0
    got, want = groth16_setup(cs.a, cs.b, cs.c, cs.m, cs.l, cs.m_l_1, za, ctx=ctx), O.groth16_params(cs, za, threads=2)
    for name in ("h", "l", "a", "b_g1", "b_g2", "ic"):
        assert (got[name] == want[name]).all() and (got[name + "_inf"] == want[name + "_inf"]).all(), name
    d = [ctx.upload(np.ascontiguousarray(x, dtype=np.uint64)) for x in cs.a]
    mat = (d[0].ptr, d[1].ptr, d[2].ptr)
    with pytest.raises(KogarashiError, match="bad argument"):
        ctx.groth16_setup(mat, mat, mat, cs.m, cs.l, cs.m_l_1, toxic, Groth16Crs(), 0, 0)       # no output arrays
    with pytest.raises(KogarashiError, match="bad argument"):
        groth16_setup(cs.a, cs.b, cs.c, 0, cs.l, cs.m_l_1, toxic, ctx=ctx)
    # the matrices' CONTENTS are checked on the device -- a column beyond the l + m_l_1 variables, or row pointers that run
    # backwards, would send the transposition's counters out of their arrays: KG_ERR_BAD_ARG instead, and the context keeps working
    nv = cs.l + cs.m_l_1
    bad_col = (cs.b[0], cs.b[1].copy(), cs.b[2]); bad_col[1][3] = nv
    with pytest.raises(KogarashiError, match="bad argument"):
        groth16_setup(cs.a, bad_col, cs.c, cs.m, cs.l, cs.m_l_1, toxic, ctx=ctx)
    bad_ptr = (cs.a[0].copy(), cs.a[1], cs.a[2]); bad_ptr[0][2], bad_ptr[0][3] = bad_ptr[0][3], bad_ptr[0][2] - 1 if bad_ptr[0][2] else 0
    if (np.diff(bad_ptr[0].astype(np.int64)) < 0).any():
        with pytest.raises(KogarashiError, match="bad argument"):
            groth16_setup(bad_ptr, cs.b, cs.c, cs.m, cs.l, cs.m_l_1, toxic, ctx=ctx)
    again = groth16_setup(cs.a, cs.b, cs.c, cs.m, cs.l, cs.m_l_1, za, ctx=ctx)
    assert (again["a"] == want["a"]).all()


@pytest.mark.parametrize("m,njobs", [(300, 5), (6000, 4)])
def test_groth16_two_proofs_in_flight_match_blocking_calls(ctx, oracle, m, njobs):
    """kg_groth16_prove_begin / _end with two proofs in flight (different witnesses and blinding scalars, alternating
    tickets) return exactly the proofs of the blocking call; an identity delta surfaces at the matching _end.  The context is in
    its default mode (inputs ordered behind the caller's queue): h's chain of these short proofs runs on a queue of the library's own so
    that the proofs overlap; 6000 constraints: the one-launch MSMs with their scalars converted once, windows split over workgroups."""
    import kogarashi_amd as K
    from kogarashi_amd.lib import ProverSubVersionCrsAttack
    O = oracle
    css = [O.chain_r1cs(m, O.gen_scalars(0, SEED + 450 + j, 0, 1)[0]) for j in range(njobs)]
    full = O.groth16_params(css[0], O.gen_scalars(0, SEED + 451, 0, 5), threads=8)
    params = dict(full)
    params["vk_g2"] = full["vk_g2"][:2]
    prover = K.Prover(params, css[0].m, css[0].l, css[0].m_l_1, ctx=ctx)
    jobs = []
    for j, cs in enumerate(css):
        a, b, c = cs.evaluate()
        r, s = O.gen_scalars(0, SEED + 460 + j, 0, 2)
        jobs.append((a, b, c, cs.x, cs.w, r, s))
    want = [prover.create_proof(*job) for job in jobs]
    got = list(prover.create_proofs(jobs))
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        for k in range(4):
            assert (g[k] == w_[k]).all()
    # the oracle agrees with the first one (the blocking path is compared in test_groth16_proof_matches_oracle)
    ref = O.groth16_prove(css[0], full, jobs[0][5], jobs[0][6], evals=jobs[0][:3])
    assert all((got[0][k] == ref[k]).all() for k in range(3))
    bad = dict(params); bad["delta_g1_inf"] = 1
    with pytest.raises(ProverSubVersionCrsAttack):
        list(K.Prover(bad, css[0].m, css[0].l, css[0].m_l_1, ctx=ctx).create_proofs(jobs[:2]))


@pytest.mark.parametrize("seed", [int(x) for x in __import__("os").environ.get("KG_SOAK_SEEDS", "99,7,2026").split(",")])
def test_interleaved_calls_share_one_context(ctx, oracle, seed):
    """Soak: blocking MSMs on three curves, MSMs in flight, NTTs, blocking proofs and proofs in flight, registered (one with a window table) and
    plain base arrays, issued in a seeded random order on ONE context -- the calls share result slots, run-space sets,
    reduction queues and the sort work space, so every result is compared with the value the same call gave alone."""
    import kogarashi_amd as K
    O = oracle
    rng = np.random.default_rng(seed)
    # fixtures
    msm_in = {}
    for name, curve, sfd, n in (("g1a", 0, 0, 70000), ("g1b", 0, 0, 3000), ("gk", 1, 1, 20000)):
        b = O.gen_bases(curve, SEED + 800 + n, 0, n); s = O.gen_scalars(sfd, SEED + 801 + n, 0, n)
        msm_in[name] = (curve, n, ctx.upload(b), ctx.upload(s))
    n2 = 900
    dk = ctx.upload(O.gen_scalars(0, SEED + 810, 0, n2)); dxy = ctx.empty((n2, 16)); dinf = ctx.empty((n2,), dtype=np.uint8)
    ctx.fixed_base_mul(2, dk.ptr, n2, dxy.ptr, dinf.ptr)
    msm_in["g2"] = (2, n2, dxy, ctx.upload(O.gen_scalars(0, SEED + 811, 0, n2)))
    ctx.bases_register(0, msm_in["g1a"][2].ptr, 0, msm_in["g1a"][1])
    cs = O.chain_r1cs(700, O.gen_scalars(0, SEED + 820, 0, 1)[0])
    params = O.groth16_params(cs, O.gen_scalars(0, SEED + 821, 0, 5), threads=8)
    params["vk_g2"] = params["vk_g2"][:2]
    prover = K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=ctx)
    a, b_, c = cs.evaluate()
    r, s_ = O.gen_scalars(0, SEED + 822, 0, 2)
    proof_args = (a, b_, c, cs.x, cs.w, r, s_)
    up = lambda v: ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
    dproof = [up(v) for v in proof_args[:5]]
    ntt_v = O.gen_scalars(0, SEED + 830, 0, 1 << 14)
    fft = K.Fft(14, ctx=ctx)
    # reference values, each call alone
    want_msm = {k: ctx.msm(c_, b.ptr, 0, s.ptr, n) for k, (c_, n, b, s) in msm_in.items()}
    want_proof = prover.create_proof(*proof_args)
    want_ntt = fft.coset_dft(ntt_v)
    ctx.bases_precompute(msm_in["g1a"][2].ptr)      # from here on g1a's MSMs take the merged sort over its window table
    try:
        pending_msm, pending_proof = {}, {}
        for step in range(60):
            op = int(rng.integers(0, 6))
            if op == 0:
                k = list(msm_in)[int(rng.integers(0, 4))]
                c_, n, b, s = msm_in[k]
                assert (ctx.msm(c_, b.ptr, 0, s.ptr, n) == want_msm[k]).all(), (step, k)
            elif op == 1:
                t = int(rng.integers(0, 4))
                if t in pending_msm:
                    k = pending_msm.pop(t)
                    assert (ctx.msm_end(msm_in[k][0], t) == want_msm[k]).all(), (step, "end", k)
                else:
                    k = list(msm_in)[int(rng.integers(0, 4))]
                    c_, n, b, s = msm_in[k]
                    ctx.msm_begin(c_, b.ptr, 0, s.ptr, n, t)
                    pending_msm[t] = k
            elif op == 2:
                assert (fft.coset_dft(ntt_v) == want_ntt).all(), step
            elif op == 3:
                got = prover.create_proof(*proof_args) if 0 not in pending_proof else None
                if got is not None:
                    assert all((got[i] == want_proof[i]).all() for i in range(4)), step
            else:
                t = int(rng.integers(0, 2))
                if t in pending_proof:
                    pending_proof.pop(t)
                    got = ctx.groth16_prove_end(t)
                    assert all((got[i] == want_proof[i]).all() for i in range(4)), (step, "proof end")
                else:
                    ctx.groth16_prove_begin(prover.crs, *[d.ptr for d in dproof], r, s_, t)
                    pending_proof[t] = True
        for t, k in pending_msm.items():
            assert (ctx.msm_end(msm_in[k][0], t) == want_msm[k]).all()
        for t in pending_proof:
            got = ctx.groth16_prove_end(t)
            assert all((got[i] == want_proof[i]).all() for i in range(4))
    finally:
        ctx.bases_unregister(msm_in["g1a"][2].ptr)


def test_context_closes_with_a_proof_in_flight(oracle):
    """kg_ctx_destroy with a proof begun and never collected: the worker threads are joined, nothing hangs or crashes,
    and a new context afterwards proves correctly."""
    import kogarashi_amd as K
    O = oracle
    cs = O.chain_r1cs(200, O.gen_scalars(0, SEED + 470, 0, 1)[0])
    full = O.groth16_params(cs, O.gen_scalars(0, SEED + 471, 0, 5), threads=8)
    params = dict(full); params["vk_g2"] = full["vk_g2"][:2]
    a, b, c = cs.evaluate()
    r, s = O.gen_scalars(0, SEED + 472, 0, 2)
    want = O.groth16_prove(cs, full, r, s, evals=(a, b, c))
    c1 = K.Context(0)
    p1 = K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=c1)
    up = lambda v: c1.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
    d = [up(v) for v in (a, b, c, cs.x, cs.w)]
    c1.groth16_prove_begin(p1.crs, *[x.ptr for x in d], r, s, 1)
    del p1
    c1.close()
    c2 = K.Context(0)
    try:
        got = K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=c2).create_proof(a, b, c, cs.x, cs.w, r, s)
        assert all((got[k] == want[k]).all() for k in range(3))
    finally:
        c2.close()


def test_proofs_in_flight_with_complete_inputs(oracle):
    """kg_ctx_set_inputs_complete(1) lets a proof's transform chains, witness sort and h chain start without waiting for the
    main queue (per-ticket polynomial buffers): several proofs over two circuits sizes, two in flight, must equal the
    blocking proofs and the oracle's."""
    import kogarashi_amd as K
    O = oracle
    ctx = K.Context(0)
    try:
        provers = []
        for m in (300, 5000):
            cs = O.chain_r1cs(m, O.gen_scalars(0, SEED + 1300 + m, 0, 1)[0])
            params = O.groth16_params(cs, O.gen_scalars(0, SEED + 1301, 0, 5), threads=8)
            a, b, c = cs.evaluate()
            rs = O.gen_scalars(0, SEED + 1302 + m, 0, 6)
            want = [O.groth16_prove(cs, params, rs[2 * i], rs[2 * i + 1], evals=(a, b, c)) for i in range(3)]
            params["vk_g2"] = params["vk_g2"][:2]
            provers.append((K.Prover(params, cs.m, cs.l, cs.m_l_1, ctx=ctx), (a, b, c, cs.x, cs.w), rs, want))
        ctx.set_inputs_complete(True)
        for prover, (a, b, c, x, w), rs, want in provers:
            jobs = [(a, b, c, x, w, rs[2 * (i % 3)], rs[2 * (i % 3) + 1]) for i in range(7)]
            for i, got in enumerate(prover.create_proofs(jobs)):
                for g, w_ in zip(got[:3], want[i % 3][:3]):
                    assert (g == w_).all(), i
            got = prover.create_proof(a, b, c, x, w, rs[0], rs[1])
            assert all((g == w_).all() for g, w_ in zip(got[:3], want[0][:3]))
    finally:
        ctx.close()


@pytest.mark.parametrize("curve,sfd,w,n", [(0, 0, 8, 5000), (1, 1, 8, 3001), (2, 0, 16, 1200)])
def test_fixed_base_mul_windowed(ctx, oracle, curve, sfd, w, n):
    """n >= 512 takes the windowed path (32 x 255 generator multiples, 32 mixed additions per scalar, batched conversion to
    affine with one inversion per 8 points): element for element the oracle's double-and-add, including k = 0 (identity flag),
    +-1, bytes of zero, the largest scalar and identities at a batch's edges."""
    O = oracle
    k = O.gen_scalars(sfd, SEED + 1500 + curve, 0, n)
    one = O.f_consts(sfd)["r"]
    k[0] = 0; k[7] = 0; k[8] = 0; k[n - 1] = 0                       # identities inside and at the edges of inversion batches
    k[1] = one
    k[2] = O.f_neg(sfd, one)                                          # p - 1: the largest scalar
    k[3] = O.f_to_mont(sfd, np.array([0, 0, 1, 0], dtype=np.uint64))  # 2^128: a single non-zero byte
    k[4] = O.f_to_mont(sfd, np.array([0xFF00FF00FF00FF00, 0, 0xFF, 0], dtype=np.uint64))
    dk = ctx.upload(k)
    dxy, dinf = ctx.empty((n, w)), ctx.empty((n,), dtype=np.uint8)
    ctx.fixed_base_mul(curve, dk.ptr, n, dxy.ptr, dinf.ptr)
    want_xy, want_inf = O.fixed_base_mul(curve, k)
    assert (dinf.numpy() == want_inf).all() and want_inf[[0, 7, 8, n - 1]].all() and want_inf.sum() == 4
    assert (dxy.numpy() == want_xy).all()
    ctx.fixed_base_mul(curve, dk.ptr, n, dxy.ptr, dinf.ptr)          # second call: cached table
    assert (dxy.numpy() == want_xy).all()



@pytest.mark.parametrize("m", [300, 5000, 70000])
def test_proof_from_witness_evaluates_on_the_device(ctx, oracle, m):
    """kg_groth16_prove_r1cs_bn254 / _begin: the constraint matrices take the place of cs.evaluate()'s three vectors (each
    transform chain starts with its matrix-vector product) -- the proofs equal those made from host-side evaluations, blocking
    and in flight, in both input-ordering modes."""
    import kogarashi_amd as K
    O = oracle
    if m <= 5000:
        cs = O.chain_r1cs(m, O.gen_scalars(0, SEED + 1500 + m, 0, 1)[0])
        params = O.groth16_params(cs, O.gen_scalars(0, SEED + 1501, 0, 5), threads=8)
        params["vk_g2"] = params["vk_g2"][:2]
        shape, x, w, l, m_l_1 = (cs.a, cs.b, cs.c), cs.x, cs.w, cs.l, cs.m_l_1
        a_ev, b_ev, c_ev = cs.evaluate()
    else:                                            # device setup for the larger circuit
        from kogarashi_amd import synthetic as syn
        from kogarashi_amd.api import groth16_setup
        cc = syn.ChainCircuit(m)
        params = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=ctx)
        shape, x, w, l, m_l_1 = (cc.a, cc.b, cc.c), cc.x, cc.w, cc.l, cc.m_l_1
        a_ev, b_ev, c_ev = cc.a_eval, cc.b_eval, cc.c_eval
    r, s = O.gen_scalars(0, SEED + 1502, 0, 2)
    prover = K.Prover(params, m, l, m_l_1, ctx=ctx)
    want = prover.create_proof(a_ev, b_ev, c_ev, x, w, r, s)
    prover.attach_constraint_system(*shape)
    got = prover.create_proof_from_witness(x, w, r, s)
    assert all((got[i] == want[i]).all() for i in range(4))
    up = lambda v: ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
    dx, dw = up(x), up(w)
    ptrs = [tuple(d.ptr for d in trip) for trip in prover._cs]
    for complete in (False, True):
        ctx.set_inputs_complete(complete)
        try:
            ctx.groth16_prove_r1cs_begin(prover.crs, ptrs[0], ptrs[1], ptrs[2], dx.ptr, dw.ptr, r, s, 0)
            ctx.groth16_prove_r1cs_begin(prover.crs, ptrs[0], ptrs[1], ptrs[2], dx.ptr, dw.ptr, r, s, 1)
            for t in (0, 1):
                g = ctx.groth16_prove_end(t)
                assert all((g[i] == want[i]).all() for i in range(4)), (complete, t)
        finally:
            ctx.set_inputs_complete(False)
