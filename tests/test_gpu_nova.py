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
            counts[7] = 200                              # one long row: past 48 entries a row gets a wave of its own (work list)
        if m > 100:
            counts[20 + j:60:7] = 254                    # range-check-like rows, at different constraints in A, B and C
            counts[61] = 5000 if j == 1 else counts[61]
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


@pytest.mark.parametrize("fd,curve,cv", [(0, 0, "g1"), (1, 1, "gk")])
def test_folding_step_matches_big_integer_fold(ctx, oracle, pyoracle, fd, curve, cv):
    """NovaProver.prove = nova::Prover::prove with the challenge supplied (prover.rs:24-50): two folding steps in a row --
    the second starts from a relaxed instance with u != 1, E != 0 and a non-identity commit_e -- against the oracle's cross
    term and commitments, big-integer folds of witness and instance (pyoracle point arithmetic), and the relaxed
    satisfiability identity AZ o BZ = u CZ + E of the folded pair evaluated with device ops."""
    import kogarashi_amd as K
    from helpers import CURVES, I, L, np_to_pt, pt_to_np
    O, P = oracle, pyoracle
    cur = CURVES[cv][1]
    p = cur.n                                        # the curve's scalar field = the circuit's field
    one = O.f_consts(fd)["r"]
    sat = fd == 0                                    # the oracle's chain circuit lives over Fr; the Fq run folds unsatisfied random data
    if sat:
        m = 300
        css = [O.chain_r1cs(m, O.gen_scalars(0, SEED + 1100 + j, 0, 1)[0]) for j in range(3)]
        shape = (css[0].a, css[0].b, css[0].c)
        xs = [cs.x[1:] for cs in css]                # chain_r1cs columns are (x | w) with x[0] = 1 = the one-wire u
        ws = [cs.w for cs in css]
    else:
        m, lx, lw = 300, 2, 200
        shape = random_shape(O, fd, m, 1 + lx + lw, SEED + 1101)
        xs = [O.gen_scalars(fd, SEED + 1102 + j, 0, lx) for j in range(3)]
        ws = [O.gen_scalars(fd, SEED + 1105 + j, 0, lw) for j in range(3)]
    g = O.gen_bases(curve, SEED + 1110, 0, m + 1)
    ck = K.PedersenCommitment(g, curve=curve, ctx=ctx)
    prover = K.NovaProver(shape, ck, ctx=ctx)
    commit = lambda v: ck.commit(v)
    to_int = lambda a: [P.from_mont(I(row), p) for row in np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)]
    to_np = lambda vals: np.array([L(P.to_mont(v % p, p)) for v in vals], dtype=np.uint64).reshape(-1, 4)
    pt = lambda c: np_to_pt(cur, c[0], c[1])

    def fresh(j):                                    # R1csInstance / R1csWitness j (u = 1)
        return {"commit_w": commit(ws[j]), "x": xs[j]}, {"w": ws[j]}

    i1 = {"commit_w": commit(ws[0]), "commit_e": (np.zeros(8, dtype=np.uint64), 1), "u": one, "x": xs[0]}
    w1 = {"w": ws[0], "e": np.zeros((m, 4), dtype=np.uint64)}
    for step, seed in enumerate((1, 2)):
        i2, w2 = fresh(seed)
        r = O.gen_scalars(fd, SEED + 1120 + seed, 0, 1)[0]
        inst, wit, commit_t = prover.prove(i1, w1, i2, w2, r)
        # cross term and its commitment: the oracle's restatement
        z1 = np.concatenate([np.asarray(i1["u"]).reshape(1, 4), i1["x"], w1["w"]])
        z2 = np.concatenate([one.reshape(1, 4), i2["x"], w2["w"]])
        t = O.nova_cross_term(fd, shape[0], shape[1], shape[2], z1, z2, np.asarray(i1["u"]).reshape(4), one)
        nc = min(m, len(g))
        wxy, winf = O.commit_naive(cv, g[:nc], t[:nc])
        assert commit_t[1] == winf and (winf or (commit_t[0] == wxy).all())
        # big-integer folds
        ri = to_int(r)[0]
        assert (wit["w"] == to_np([a + ri * b for a, b in zip(to_int(w1["w"]), to_int(w2["w"]))])).all()
        assert (wit["e"] == to_np([a + ri * b for a, b in zip(to_int(w1["e"]), to_int(t))])).all()
        assert (inst["u"] == to_np([to_int(i1["u"])[0] + ri])[0]).all()
        assert (inst["x"] == to_np([a + ri * b for a, b in zip(to_int(i1["x"]), to_int(i2["x"]))])).all()
        for name, a, b in (("commit_w", i1["commit_w"], i2["commit_w"]), ("commit_e", i1["commit_e"], commit_t)):
            want = cur.add(pt(a), cur.mul(pt(b), ri))
            got = inst[name]
            assert (want is None) == bool(got[1]) and (want is None or (got[0] == pt_to_np(cur, want)).all()), (step, name)
        # the commitments of the folded witness are the folded commitments
        assert commit(wit["w"])[1] == inst["commit_w"][1] and (commit(wit["w"])[0] == inst["commit_w"][0]).all()
        assert commit(wit["e"])[1] == inst["commit_e"][1] and (commit(wit["e"])[0] == inst["commit_e"][0]).all()
        i1_next, w1_next = inst, wit
        if not sat:
            i1, w1 = i1_next, w1_next
            continue
        # the folded pair satisfies the relaxed relation (device ops only)
        z = np.concatenate([inst["u"].reshape(1, 4), inst["x"], wit["w"]])
        dz = ctx.upload(z)
        outs = []
        for dev in prover._dev:
            o = ctx.empty((m, 4))
            ctx.r1cs_prod(fd, dev[0].ptr, dev[1].ptr, dev[2].ptr, m, dz.ptr, o.ptr)
            outs.append(o)
        lhs, rhs, de = ctx.empty((m, 4)), ctx.empty((m, 4)), ctx.upload(wit["e"])
        ctx.field_vec_op(fd, "mul", outs[0].ptr, outs[1].ptr, lhs.ptr, m)
        ctx.field_vec_scale(fd, outs[2].ptr, inst["u"], rhs.ptr, m)
        ctx.field_vec_op(fd, "add", rhs.ptr, de.ptr, rhs.ptr, m)
        assert (lhs.numpy() == rhs.numpy()).all(), step
        i1, w1 = i1_next, w1_next


@pytest.mark.parametrize("fd", [0, 1])
def test_matrix_vector_product_with_every_row_class(ctx, oracle, fd):
    """kg_r1cs_prod classes its rows: up to 48 entries a lane each, 49 .. 4096 a wave each (a range check's bit sum holds 254: a circuit has
    many), beyond that a 1024-lane workgroup each (the constant-one wire's row of a transposed system).  One matrix with all three --
    300 rows of 254 entries, rows of exactly 48 / 49 / 4096 / 4097 entries, two rows of 70 000 -- against the oracle's SparseMatrix::prod."""
    O = oracle
    rng = np.random.default_rng(77 + fd)
    m, nvars = 2000, 5000
    counts = rng.integers(0, 7, m)
    counts[100:400] = 254
    counts[[10, 11, 12, 13]] = [48, 49, 4096, 4097]
    counts[[500, 1999]] = 70000
    counts[501] = 0
    rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    col = rng.integers(0, nvars, int(rp[-1])).astype(np.uint64)
    val = O.gen_scalars(fd, SEED + 950 + fd, 0, int(rp[-1]))
    z = O.gen_scalars(fd, SEED + 952 + fd, 0, nvars)
    want = O.matrix_prod(fd, (rp, col, np.ascontiguousarray(val)), z)
    d = [ctx.upload(rp), ctx.upload(col), ctx.upload(val)]
    dz, out = ctx.upload(z), ctx.empty((m, 4))
    for _ in range(2):                                   # twice: the work lists are rebuilt per call
        ctx.r1cs_prod(fd, d[0].ptr, d[1].ptr, d[2].ptr, m, dz.ptr, out.ptr)
        assert (out.numpy() == want).all()
