"""Nova's per-step hot path on the device: the cross term T of a folding step (nova/src/prover.rs:53-90) and its commitment
(prover.rs:35), on both cycle curves' scalar fields, against the oracle's restatement."""
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


def random_shape(O, fd, m, nvars, seed, max_row=6):
    """three random sparse matrices over z = (u | x | w) with empty rows, a long row and repeated columns"""
    rng = np.random.default_rng(seed)
    out = []
    for j in range(3):
        counts = rng.integers(0, max_row + 1, m)
        counts[3 % m] = 0
        if m > 10:
            counts[7] = 200                              # one long row (8 lanes stride over it)
        rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        col = rng.integers(0, nvars, int(rp[-1])).astype(np.uint64)
        val = O.gen_scalars(fd, seed + 10 + j, 0, max(int(rp[-1]), 1))[: int(rp[-1])]
        one = O.f_consts(fd)["r"]
        val[::5] = one                                   # R1CS coefficients are mostly +-1
        val[1::7] = O.f_neg(fd, one)
        out.append((rp, col, np.ascontiguousarray(val)))
    return out


@pytest.mark.parametrize("fd,curve", [(0, 0), (1, 1)])
@pytest.mark.parametrize("m", [1, 5, 1000, 20000])
def test_cross_term_matches_oracle(ctx, oracle, fd, curve, m):
    import kogarashi_amd as K
    O = oracle
    lx, lw = 3, max(m // 2, 4)
    nvars = 1 + lx + lw
    shape = random_shape(O, fd, m, nvars, SEED + 900 + m + fd)
    u1 = O.gen_scalars(fd, SEED + 901, 0, 1)[0]
    u2 = O.f_consts(fd)["r"]                             # the fresh instance has u = 1 (prover.rs:61)
    x1, w1 = O.gen_scalars(fd, SEED + 902, 0, lx), O.gen_scalars(fd, SEED + 903, 0, lw)
    x2, w2 = O.gen_scalars(fd, SEED + 904, 0, lx), O.gen_scalars(fd, SEED + 905, 0, lw)
    w1[0] = 0
    z1 = np.concatenate([u1[None], x1, w1])
    z2 = np.concatenate([u2[None], x2, w2])
    want = O.nova_cross_term(fd, shape[0], shape[1], shape[2], z1, z2, u1, u2)
    g = O.gen_bases(curve, SEED + 906, 0, min(m, 64) + 1)
    ck = K.PedersenCommitment(g, curve=curve, ctx=ctx)
    prover = K.NovaProver(shape, ck, ctx=ctx)
    got = prover.compute_cross_term(u1, x1, w1, u2, x2, w2)
    assert (got == want).all()
    # the six products one by one through kg_r1cs_prod (SparseMatrix::prod) agree with the oracle as well
    dz = ctx.upload(z1)
    for trip, dev in zip(shape, prover._dev):
        out = ctx.empty((m, 4))
        ctx.r1cs_prod(fd, dev[0].ptr, dev[1].ptr, dev[2].ptr, m, dz.ptr, out.ptr)
        assert (out.numpy() == O.matrix_prod(fd, trip, z1)).all()
    # commit_t = ck.commit(&t): the reference's naive fold over min(len) pairs
    t, (xy, inf) = prover.commit_t(u1, x1, w1, u2, x2, w2)
    assert (t == want).all()
    cv = "g1" if curve == 0 else "gk"
    n = min(m, len(g))
    wxy, winf = O.commit_naive(cv, g[:n], want[:n])
    assert inf == winf and (inf or (xy == wxy).all())


def test_cross_term_of_a_satisfied_pair_folds(ctx, oracle):
    """The identity folding relies on: for satisfying (z1, E1, u1) and (z2, 0, 1), the folded z = z1 + r z2 with
    E = E1 + r T satisfies AZ o BZ = u CZ + E (nova/src/relaxed_r1cs.rs is_sat) -- checked with device ops only."""
    import kogarashi_amd as K
    O, fd = oracle, 0
    cs = O.chain_r1cs(512, O.gen_scalars(0, SEED + 950, 0, 1)[0])
    cs2 = O.chain_r1cs(512, O.gen_scalars(0, SEED + 951, 0, 1)[0])
    one = O.f_consts(0)["r"]
    # chain_r1cs columns are over (x | w) with x[0] = 1: the same as (u | x' | w) with u = 1, x' = x[1:]
    shape = (cs.a, cs.b, cs.c)
    g = O.gen_bases(0, SEED + 952, 0, 513)
    prover = K.NovaProver(shape, K.PedersenCommitment(g, curve=0, ctx=ctx), ctx=ctx)
    t = prover.compute_cross_term(one, cs.x[1:], cs.w, one, cs2.x[1:], cs2.w)
    r = O.gen_scalars(0, SEED + 953, 0, 1)[0]
    z1, z2 = np.concatenate([cs.x, cs.w]), np.concatenate([cs2.x, cs2.w])
    dz1, dz2, dz = ctx.upload(z1), ctx.upload(z2), ctx.empty(z1.shape)
    ctx.field_vec_axpy(fd, dz1.ptr, r, dz2.ptr, dz.ptr, len(z1))                    # z = z1 + r z2 (witness.rs:56-70)
    u = O.f_add(0, one, r)
    outs = []
    for rp, col, val in shape:
        d = [ctx.upload(np.ascontiguousarray(x)) for x in (rp, col, val)]
        o = ctx.empty((cs.m, 4))
        ctx.r1cs_prod(fd, d[0].ptr, d[1].ptr, d[2].ptr, cs.m, dz.ptr, o.ptr)
        outs.append(o)
    az, bz, cz = outs
    lhs, rhs, dt = ctx.empty((cs.m, 4)), ctx.empty((cs.m, 4)), ctx.upload(t)
    ctx.field_vec_op(fd, "mul", az.ptr, bz.ptr, lhs.ptr, cs.m)
    ctx.field_vec_scale(fd, cz.ptr, u, rhs.ptr, cs.m)
    ctx.field_vec_axpy(fd, rhs.ptr, r, dt.ptr, rhs.ptr, cs.m)                       # u CZ + r T   (E1 = 0)
    assert (lhs.numpy() == rhs.numpy()).all()
