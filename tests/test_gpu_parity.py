"""GPU parity tests proper: every call goes through the C ABI (kogarashi_amd.lib -> libkogarashi_amd.so) and is
compared bit-for-bit with the oracle (oracle/kg_oracle.c, the restatement of the reference's CPU path) on the same
seeded inputs.  Field outputs: canonical Montgomery limbs.  MSM / commit outputs: affine points."""
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


def aff(O, cv, proj):
    xy, inf = O.to_affine(cv, proj)
    return (None if inf else xy.tobytes())


def gpu_aff(out, nb):
    """ABI projective output (x, y, z) with z in {0, 1}: -> affine bytes / None"""
    z = out[2 * nb:3 * nb]
    return None if not z.any() else out[:2 * nb].tobytes()


@pytest.mark.parametrize("fd", [0, 1])
def test_field_vector_ops(ctx, oracle, fd):
    O, n = oracle, 4096
    a, b = O.gen_scalars(fd, SEED + 1, 0, n), O.gen_scalars(fd, SEED + 2, 0, n)
    c = O.f_consts(fd)
    pm1 = c["p"].copy(); pm1[0] -= 1
    a[0] = 0; b[1] = 0; a[2] = c["r"]; a[3] = O.f_to_mont(fd, pm1); b[3] = a[3]; a[4] = a[5]; b[4] = a[5]
    da, db, do = ctx.upload(a), ctx.upload(b), ctx.empty((n, 4))
    exp = {"add": O.f_add, "sub": O.f_sub, "mul": O.f_mul}
    for op, f in exp.items():
        ctx.field_vec_op(fd, op, da.ptr, db.ptr, do.ptr, n)
        got = do.numpy()
        want = np.stack([f(fd, a[i], b[i]) for i in range(n)])
        assert (got == want).all(), op
    for op, f in {"square": O.f_square, "neg": O.f_neg, "double": O.f_double, "from_mont": O.f_from_mont}.items():
        ctx.field_vec_op(fd, op, da.ptr, 0, do.ptr, n)
        got = do.numpy()
        want = np.stack([f(fd, a[i]) for i in range(n)])
        assert (got == want).all(), op
    m = 64
    ctx.field_vec_op(fd, "invert", da.ptr, 0, do.ptr, m)
    got = do.numpy()[:m]
    want = np.stack([O.f_invert(fd, a[i]) if a[i].any() else np.zeros(4, dtype=np.uint64) for i in range(m)])
    assert (got == want).all()
    # a * a^-1 = 1 (field_test!, zkstd/src/macros/field/test.rs)
    ctx.field_vec_op(fd, "mul", da.ptr, do.ptr, do.ptr, m)
    got = do.numpy()[:m]
    assert all((got[i] == c["r"]).all() for i in range(m) if a[i].any())
    s = O.gen_scalars(fd, SEED + 3, 0, 1)[0]
    ctx.field_vec_scale(fd, da.ptr, s, do.ptr, n)
    assert (do.numpy() == np.stack([O.f_mul(fd, a[i], s) for i in range(n)])).all()
    # to_mont_form (represent.rs:30-32) of canonical integers, and its round trip with montgomery_reduce
    ints = np.stack([O.f_from_mont(fd, a[i]) for i in range(n)])
    di = ctx.upload(ints)
    ctx.field_vec_op(fd, "to_mont", di.ptr, 0, do.ptr, n)
    assert (do.numpy() == np.stack([O.f_to_mont(fd, ints[i]) for i in range(n)])).all() and (do.numpy() == a).all()
    ctx.field_vec_op(fd, "from_mont", do.ptr, 0, do.ptr, n)
    assert (do.numpy() == ints).all()
    # start * base^i (the scan tables of fft.rs:35-41, the powers of tau of zksnark.rs:44-49)
    npow = 3000
    dp = ctx.empty((npow, 4))
    ctx.field_powers(fd, a[5], b[5], dp.ptr, npow)
    want, cur = [], a[5]
    for _ in range(npow):
        want.append(cur)
        cur = O.f_mul(fd, cur, b[5])
    assert (dp.numpy() == np.stack(want)).all()
    ctx.field_powers(fd, c["r"], np.zeros(4, dtype=np.uint64), dp.ptr, 4)            # 0^0 = 1, then zeros
    assert (dp.numpy()[0] == c["r"]).all() and not dp.numpy()[1:4].any()
    # Nova fold W1 + r * W2 (nova/src/relaxed_r1cs/witness.rs:56-70)
    ctx.field_vec_axpy(fd, da.ptr, s, db.ptr, do.ptr, n)
    assert (do.numpy() == np.stack([O.f_add(fd, a[i], O.f_mul(fd, s, b[i])) for i in range(n)])).all()


def test_generators_match_oracle(ctx, oracle):
    O = oracle
    for fd in (0, 1):
        d = ctx.empty((1000, 4))
        ctx.gen_scalars(fd, SEED + 9, 17, 1000, d.ptr)
        assert (d.numpy() == O.gen_scalars(fd, SEED + 9, 17, 1000)).all()
    for curve in (0, 1):
        d = ctx.empty((300, 8))
        ctx.gen_bases(curve, SEED + 10, 5, 300, d.ptr)
        assert (d.numpy() == O.gen_bases(curve, SEED + 10, 5, 300)).all()


def edge_mix(O, cv, curve, sfd, n, seed):
    bases = O.gen_bases(curve, seed, 0, n)
    scal = O.gen_scalars(sfd, seed + 1, 0, n)
    inf = np.zeros(n, dtype=np.uint8)
    if n >= 8:
        inf[2] = 1                                   # identity base
        scal[1] = 0                                  # zero scalar
        bases[3] = bases[0]                          # duplicate point
        bases[5] = bases[4]                          # P and (-1)*P below
        scal[6] = scal[7]; bases[6] = bases[7]       # same point, same scalar
        one = O.f_consts(sfd)["r"]
        scal[4] = one                                # scalar 1
        scal[5] = O.f_neg(sfd, one)                  # scalar -1
    return bases, scal, inf


@pytest.mark.parametrize("cv,curve,sfd", [("g1", 0, 0), ("gk", 1, 1)])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 31, 32, 33, 257, 1024, 5000])
def test_msm_matches_oracle(ctx, oracle, cv, curve, sfd, n):
    O = oracle
    bases, scal, inf = edge_mix(O, cv, curve, sfd, n, SEED + 20 + n)
    want = aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
    got = ctx.msm_host(curve, bases, inf, scal, n)
    assert gpu_aff(got, 4) == want
    # without flags
    want2 = aff(O, cv, O.msm(cv, bases, scal, None, threads=8))
    assert gpu_aff(ctx.msm_host(curve, bases, None, scal, n), 4) == want2
    # these lengths run as the short-input kernel by default (tests/test_gpu_small.py): the long pipeline on the same inputs
    ctx.set_msm_small(0)
    try:
        assert gpu_aff(ctx.msm_host(curve, bases, inf, scal, n), 4) == want
        assert gpu_aff(ctx.msm_host(curve, bases, None, scal, n), 4) == want2
    finally:
        ctx.set_msm_small(32768)


def test_msm_special_sums(ctx, oracle):
    """all-zero scalars, all-identity bases, P + (-P), k*P + (r-k)*P, empty input (msm.rs edge behaviour)."""
    O = oracle
    n = 64
    bases = O.gen_bases(0, SEED + 40, 0, n)
    scal = O.gen_scalars(0, SEED + 41, 0, n)
    ident = lambda out: gpu_aff(out, 4) is None
    assert ident(ctx.msm_host(0, bases, None, np.zeros_like(scal), n))
    assert ident(ctx.msm_host(0, bases, np.ones(n, dtype=np.uint8), scal, n))
    assert ident(ctx.msm_host(0, bases[:0], None, scal[:0], 0))
    b2 = np.stack([bases[0], bases[0]]); s2 = np.stack([scal[0], O.f_neg(0, scal[0])])
    assert ident(ctx.msm_host(0, b2, None, s2, 2))
    nb = bases[0].copy(); nb[4:] = O.f_neg(1, bases[0, 4:])
    assert ident(ctx.msm_host(0, np.stack([bases[0], nb]), None, np.stack([scal[0], scal[0]]), 2))
    # identity output is (0, 1, 0) in Montgomery form (group.rs:106-110)
    out = ctx.msm_host(0, bases, None, np.zeros_like(scal), n)
    assert not out[:4].any() and (out[4:8] == O.f_consts(1)["r"]).all() and not out[8:].any()


@pytest.mark.parametrize("c", [2, 5, 8, 11, 13])
def test_msm_window_widths(ctx, oracle, c):
    O, n = oracle, 700
    bases, scal, inf = edge_mix(O, "g1", 0, 0, n, SEED + 60)
    want = aff(O, "g1", O.msm("g1", bases, scal, inf, threads=8))
    ctx.set_msm_window(c)
    try:
        assert gpu_aff(ctx.msm_host(0, bases, inf, scal, n), 4) == want
    finally:
        ctx.set_msm_window(0)


def test_msm_g2_matches_oracle(ctx, oracle):
    O, n = oracle, 48
    g = O.generator("g2")
    one = np.concatenate([O.f_consts(1)["r"], np.zeros(4, dtype=np.uint64)])
    gen_proj = np.concatenate([g, one])
    ks = O.gen_scalars(0, SEED + 70, 0, n)
    bases = np.stack([O.to_affine("g2", O.scalar_point("g2", gen_proj, ks[i]))[0] for i in range(n)])
    scal = O.gen_scalars(0, SEED + 71, 0, n)
    inf = np.zeros(n, dtype=np.uint8); inf[3] = 1; scal[2] = 0; bases[5] = bases[4]
    want = aff(O, "g2", O.msm("g2", bases, scal, inf, threads=8))
    assert gpu_aff(ctx.msm_host(2, bases, inf, scal, n), 8) == want


def test_commit_matches_naive_pedersen(ctx, oracle):
    """nova PedersenCommitment::commit (naive NAF scalar muls in the reference) vs the device MSM, both curves."""
    import kogarashi_amd as K
    O = oracle
    for name, cv, curve, sfd in (("g1", "g1", 0, 0), ("grumpkin", "gk", 1, 1)):
        n = 40
        g = O.gen_bases(curve, SEED + 80, 0, n + 1)           # key size 2^k + 1 (pedersen.rs:10-13)
        m = O.gen_scalars(sfd, SEED + 81, 0, n)
        want_xy, want_inf = O.commit_naive(cv, g, m)
        pc = K.PedersenCommitment(g, curve=name, ctx=ctx)
        xy, inf = pc.commit(m)
        assert inf == want_inf and (xy == want_xy).all()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 9, 10, 12, 13, 16, 17, 18])
def test_ntt_matches_oracle(ctx, oracle, k):
    import kogarashi_amd as K
    O = oracle
    n = 1 << k
    v = O.gen_scalars(0, SEED + 100 + k, 0, n)
    fo, fg = O.Fft(k), K.Fft(k, ctx=ctx)
    assert (fg.dft(v) == fo.dft(v, threads=8)).all()
    assert (fg.idft(v) == fo.idft(v, threads=8)).all()
    assert (fg.coset_dft(v) == fo.coset_dft(v, threads=8)).all()
    assert (fg.coset_idft(v) == fo.coset_idft(v, threads=8)).all()
    assert (fg.divide_by_z_on_coset(v) == fo.divide_by_z_on_coset(v)).all()
    # ragged input: shorter than n is zero padded (prepare_fft)
    assert (fg.dft(v[: n // 2 + 1]) == fo.dft(v[: n // 2 + 1])).all()
    # idft(dft(v)) == v (fft_transformation_test, fft.rs:246-257)
    assert (fg.idft(fg.dft(v)) == v).all()


@pytest.mark.parametrize("c", [0, 3, 7, 14])
def test_msm_skewed_scalars(ctx, oracle, c):
    """Witness-like scalars (mostly 0 / 1 / small / -1, groth16 aux vectors): a few buckets hold almost every point,
    which exercises the multi-round partial-sum path of the load-balanced accumulation."""
    O, n = oracle, 20000
    rng = np.random.default_rng(7)
    bases = O.gen_bases(0, SEED + 90, 0, n)
    scal = O.gen_scalars(0, SEED + 91, 0, n)
    one = O.f_consts(0)["r"]
    kind = rng.integers(0, 10, n)
    small = [O.f_to_mont(0, np.array([v, 0, 0, 0], dtype=np.uint64)) for v in range(0, 9)]
    for i in range(n):
        if kind[i] < 4:
            scal[i] = one
        elif kind[i] < 6:
            scal[i] = 0
        elif kind[i] < 8:
            scal[i] = small[int(rng.integers(2, 9))]
        elif kind[i] == 8:
            scal[i] = O.f_neg(0, one)
    bases[100:200] = bases[100]                      # one point repeated 100 times
    want = aff(O, "g1", O.msm("g1", bases, scal, None, threads=8))
    ctx.set_msm_window(c)
    try:
        assert gpu_aff(ctx.msm_host(0, bases, None, scal, n), 4) == want
    finally:
        ctx.set_msm_window(0)


def test_msm_degenerate_inputs_hit_the_exceptional_branches(ctx, oracle):
    """One base repeated with one scalar: every addition after the first meets an equal point (doubling branch of
    add_mixed / add_xyzz, weierstrass.rs:75-81,114-120) and every bucket list is maximally skewed; alternating P, -P
    cancels to the identity through the inverse-point branch."""
    O = oracle
    n = 6000
    P = O.gen_bases(0, SEED + 95, 0, 1)[0]
    bases = np.tile(P, (n, 1))
    k = O.gen_scalars(0, SEED + 96, 0, 1)[0]
    scal = np.tile(k, (n, 1))
    want = aff(O, "g1", O.msm("g1", bases, scal, None, threads=8))
    assert gpu_aff(ctx.msm_host(0, bases, None, scal, n), 4) == want
    one = O.f_consts(0)["r"]
    scal1 = np.tile(one, (n, 1))
    want = aff(O, "g1", O.msm("g1", bases, scal1, None, threads=8))          # n * P
    assert gpu_aff(ctx.msm_host(0, bases, None, scal1, n), 4) == want
    negP = P.copy(); negP[4:] = O.f_neg(1, P[4:])
    bases[1::2] = negP
    assert gpu_aff(ctx.msm_host(0, bases, None, scal, n), 4) is None        # k*P - k*P + ... = identity
    bases[-1] = P                                                            # odd one out
    scal[-1] = one
    want = aff(O, "g1", O.msm("g1", bases, scal, None, threads=8))
    assert gpu_aff(ctx.msm_host(0, bases, None, scal, n), 4) == want


def test_msm_begin_end_pipeline_matches_blocking_calls(ctx, oracle):
    """kg_msm_begin / kg_msm_end with several MSMs in flight (different sizes, curves and tickets) give exactly the
    results of the blocking kg_msm."""
    import kogarashi_amd as K
    O = oracle
    jobs = []
    for i, (curve, sfd, n) in enumerate([(0, 0, 3000), (1, 1, 700), (0, 0, 20000), (0, 0, 1), (1, 1, 5000), (0, 0, 0)]):
        b = O.gen_bases(curve, SEED + 200 + i, 0, max(n, 1))[:n]
        s = O.gen_scalars(sfd, SEED + 210 + i, 0, max(n, 1))[:n]
        jobs.append((curve, n, ctx.upload(b) if n else None, ctx.upload(s) if n else None))
    want = [ctx.msm(c, b.ptr if b else 0, 0, s.ptr if s else 0, n) for c, n, b, s in jobs]
    got = [None] * len(jobs)
    depth = 3
    for i, (c, n, b, s) in enumerate(jobs):
        if i >= depth:
            got[i - depth] = ctx.msm_end(jobs[i - depth][0], (i - depth) % 4)
        ctx.msm_begin(c, b.ptr if b else 0, 0, s.ptr if s else 0, n, i % 4)
    for i in range(max(0, len(jobs) - depth), len(jobs)):
        got[i] = ctx.msm_end(jobs[i][0], i % 4)
    for g, w in zip(got, want):
        assert (g == w).all()


@pytest.mark.parametrize("cv,curve,sfd", [("g1", 0, 0), ("gk", 1, 1), ("g2", 2, 0)])
def test_registered_bases_give_identical_results(ctx, oracle, cv, curve, sfd):
    """kg_bases_register converts a resident base array once; every MSM over it or over a whole-point offset into it
    (params.a[cs.l()..] style) must equal the unregistered call and the oracle.  A pointer that does not fall on a
    point boundary of a registered array, or a range that runs past its end, is not served from it."""
    O = oracle
    if curve == 2:                                  # G2 bases: k_i * G2 from the device (checked against the oracle elsewhere)
        n = 1500
        dk = ctx.upload(O.gen_scalars(0, SEED + 299, 0, n))
        dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
        ctx.fixed_base_mul(2, dk.ptr, n, dxy.ptr, dinf.ptr)
        b = dxy.numpy()
    else:
        n = 6000
        b = O.gen_bases(curve, SEED + 300, 0, n)
    inf = np.zeros(n, dtype=np.uint8); inf[[5, 77]] = 1
    s = O.gen_scalars(sfd, SEED + 301, 0, n)
    db, di, ds = ctx.upload(b), ctx.upload(inf), ctx.upload(s)
    stride = b.shape[1] * 8
    nb = 8 if curve == 2 else 4
    cases = [(0, n), (1234, n - 1234), (17, 1000), (n - 1, 1)]
    assert b.shape[1] * 8 == stride and stride == (128 if curve == 2 else 64)
    plain = [ctx.msm(curve, db.ptr + o * stride, di.ptr + o, ds.ptr, m) for o, m in cases]
    ctx.bases_register(curve, db.ptr, di.ptr, n)
    try:
        for (o, m), want in zip(cases, plain):
            got = ctx.msm(curve, db.ptr + o * stride, di.ptr + o, ds.ptr, m)
            assert (got == want).all()
        o, m = cases[1]
        assert gpu_aff(plain[1], nb) == aff(O, cv, O.msm(cv, b[o:], s[:m], inf=inf[o:], threads=8))
    finally:
        ctx.bases_unregister(db.ptr)
    assert (ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n) == plain[0]).all()


@pytest.mark.parametrize("cv,curve,sfd,n,c,skew", [
    ("g1", 0, 0, 1 << 16, 0, False), ("g1", 0, 0, 70001, 12, False), ("gk", 1, 1, 100000, 16, False),
    ("g1", 0, 0, 1 << 17, 0, True), ("g1", 0, 0, 150000, 13, True), ("gk", 1, 1, 90000, 16, True),
    ("g2", 2, 0, 66000, 0, False), ("g2", 2, 0, 70001, 13, True),
    # windows of 19 and 20 bits: nine-bit fine field, eight-byte intermediate entries (the 2^24-pair commitment's sort)
    ("g1", 0, 0, 70001, 19, False), ("g1", 0, 0, 1 << 17, 20, False), ("gk", 1, 1, 90000, 20, True), ("g1", 0, 0, 150000, 19, True),
    ("g2", 2, 0, 66000, 20, False),
])
def test_msm_two_pass_sort_sizes(ctx, oracle, cv, curve, sfd, n, c, skew):
    """n >= 2^16 with c >= 12 takes the two-pass bucket sort (bucket group, then bucket inside the group; groups cut
    into 8192-entry segments).  Uniform scalars fill every group evenly; the witness-like mix piles half of the entries
    into one group of window 0 (many segments of one group) and leaves most other groups empty."""
    O = oracle
    if curve == 2:                                  # G2 bases: k_i * G2 made on the device (== the oracle's: test_fixed_base_mul)
        dk = ctx.upload(O.gen_scalars(0, SEED + 499 + n, 0, n))
        dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
        ctx.fixed_base_mul(2, dk.ptr, n, dxy.ptr, dinf.ptr)
        bases = dxy.numpy()
    else:
        bases = O.gen_bases(curve, SEED + 500 + n, 0, n)
    scal = O.gen_scalars(sfd, SEED + 501 + n, 0, n)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[3, n - 1]] = 1
    if skew:
        rng = np.random.default_rng(n)
        kind = rng.integers(0, 10, n)
        one = O.f_consts(sfd)["r"]
        scal[kind < 5] = one
        scal[(kind >= 5) & (kind < 7)] = 0
        scal[kind == 7] = O.f_neg(sfd, one)
        scal[kind == 8] = O.f_to_mont(sfd, np.array([5, 0, 0, 0], dtype=np.uint64))
        bases[1000:1500] = bases[1000]
    want = aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
    ctx.set_msm_window(c)
    try:
        assert gpu_aff(ctx.msm_host(curve, bases, inf, scal, n), 8 if curve == 2 else 4) == want
        if c >= 19:                                 # and the blocking device call (window groups: four for these widths)
            db, di, ds = ctx.upload(bases), ctx.upload(inf), ctx.upload(scal)
            assert gpu_aff(ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n), 8 if curve == 2 else 4) == want
    finally:
        ctx.set_msm_window(0)


def test_msm_two_pass_sort_degenerate_scalars(ctx, oracle):
    """All-zero scalars leave every bucket group empty (no segments, no tasks): the identity.  One scalar value
    everywhere piles each window's entries into a single bucket (one group, many segments, multi-round partial sums)."""
    O, n = oracle, (1 << 16) + 5
    bases = O.gen_bases(0, SEED + 600, 0, n)
    zeros = np.zeros((n, 4), dtype=np.uint64)
    assert gpu_aff(ctx.msm_host(0, bases, None, zeros, n), 4) is None
    k = O.gen_scalars(0, SEED + 601, 0, 1)[0]
    scal = np.tile(k, (n, 1))
    want = aff(O, "g1", O.msm("g1", bases, scal, None, threads=8))
    assert gpu_aff(ctx.msm_host(0, bases, None, scal, n), 4) == want


def test_combine_partials_over_rccl_single_rank(ctx, oracle):
    """The exchange step of the sharded MSM on the device path: a one-rank RCCL group (backend "nccl") all_gathers the
    partial sum on the dedicated exchange stream and returns it unchanged; an identity partial stays the identity."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from kogarashi_amd import dist as kdist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        P = oracle.gen_bases(0, SEED + 700, 0, 1)[0]
        xy, inf = kdist.combine_partials(ctx, 0, P, 0, device=dev)
        assert not inf and (xy == P).all()
        xy, inf = kdist.combine_partials(ctx, 0, np.zeros(8, dtype=np.uint64), 1, device=dev)
        assert inf
    finally:
        dist.destroy_process_group()


def test_msm_randomised_configurations(ctx, oracle):
    """Seeded sweep over (curve, n, forced window, scalar mix, identity density, duplicate density): 60 small MSMs against
    the oracle.  Scalar mixes: uniform, small (< 2^20), near the group order (p - small), powers of two, 0/1-heavy --
    they move the load between windows and hit the signed-digit borrow at every window boundary."""
    O = oracle
    rng = np.random.default_rng(20260101)
    for case in range(60):
        cv, curve, sfd = [("g1", 0, 0), ("gk", 1, 1)][int(rng.integers(0, 2))]
        n = int(rng.choice([1, 2, 5, 17, 64, 100, 333, 1000, 2500, 4097]))
        c = int(rng.choice([0, 0, 2, 3, 5, 8, 11, 13]))
        bases = O.gen_bases(curve, SEED + 900 + case, 0, n)
        scal = O.gen_scalars(sfd, SEED + 901 + case, 0, n)
        mix = int(rng.integers(0, 5))
        pm = O.f_consts(sfd)["p"]
        for i in range(n):
            if mix == 1:
                v = int(rng.integers(0, 1 << 20))
                scal[i] = O.f_to_mont(sfd, np.array([v, 0, 0, 0], dtype=np.uint64))
            elif mix == 2:
                v = pm.copy(); v[0] -= np.uint64(int(rng.integers(1, 1 << 16)))            # p - small
                scal[i] = O.f_to_mont(sfd, v)
            elif mix == 3:
                e = int(rng.integers(0, 253))
                v = np.zeros(4, dtype=np.uint64); v[e // 64] = np.uint64(1) << np.uint64(e % 64)
                scal[i] = O.f_to_mont(sfd, v)
            elif mix == 4 and rng.random() < 0.8:
                scal[i] = O.f_to_mont(sfd, np.array([int(rng.integers(0, 2)), 0, 0, 0], dtype=np.uint64))
        inf = (rng.random(n) < [0.0, 0.02, 0.3][int(rng.integers(0, 3))]).astype(np.uint8)
        if n > 4 and rng.random() < 0.5:                   # duplicate points
            dup = rng.integers(0, n, size=max(1, n // 10))
            bases[dup] = bases[0]
        want = aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
        ctx.set_msm_window(c)
        try:
            got = gpu_aff(ctx.msm_host(curve, bases, inf, scal, n), 4)
        finally:
            ctx.set_msm_window(0)
        assert got == want, (case, cv, n, c, mix)


def test_work_space_growth_with_msms_in_flight(oracle):
    """A fresh context whose MSM sizes grow while earlier MSMs are still in flight: every growth re-allocates sort or
    run space that a pending ticket may still be using (the library must wait for it), and destroying the context with
    a ticket never collected must not hang or crash."""
    import kogarashi_amd as K
    O = oracle
    c = K.Context(0)
    try:
        sizes = [100, 70000, 5, 200000, 3000, 66000]
        data = []
        for i, n in enumerate(sizes):
            b = O.gen_bases(0, SEED + 950 + i, 0, n); s = O.gen_scalars(0, SEED + 960 + i, 0, n)
            data.append((n, c.upload(b), c.upload(s), aff(O, "g1", O.msm("g1", b, s, None, threads=8))))
        # begin i, begin i+1 (bigger: re-allocates), end i, ...
        c.msm_begin(0, data[0][1].ptr, 0, data[0][2].ptr, data[0][0], 0)
        for i in range(1, len(sizes)):
            n, b, s, _ = data[i]
            c.msm_begin(0, b.ptr, 0, s.ptr, n, i & 3)
            got = c.msm_end(0, (i - 1) & 3)
            assert gpu_aff(got, 4) == data[i - 1][3], i - 1
            assert gpu_aff(c.msm(0, b.ptr, 0, s.ptr, n), 4) == data[i][3], ("blocking", i)
        assert gpu_aff(c.msm_end(0, (len(sizes) - 1) & 3), 4) == data[-1][3]
        c.msm_begin(0, data[3][1].ptr, 0, data[3][2].ptr, data[3][0], 2)      # never collected
    finally:
        c.close()


def test_bad_arguments_are_status_codes_not_crashes(ctx):
    """The ABI never aborts: null pointers, unknown curves / fields / ops, out-of-range tickets, sizes and
    window widths come back as KG_ERR_BAD_ARG (-2) -- called through the raw ctypes handle, past the Python checks."""
    import ctypes as C
    L, h = ctx._lib, ctx._h
    n = 64
    d = ctx.empty((n, 8))
    s = ctx.empty((n, 4))
    out = (C.c_uint64 * 24)()
    vp = lambda x: C.c_void_p(x)
    BAD = -2
    assert L.kg_msm(h, 7, vp(d.ptr), None, vp(s.ptr), C.c_size_t(n), out) == BAD              # unknown curve
    assert L.kg_msm(h, 0, None, None, vp(s.ptr), C.c_size_t(n), out) == BAD                   # null bases
    assert L.kg_msm(h, 0, vp(d.ptr), None, vp(s.ptr), C.c_size_t(n), None) == BAD             # null output
    assert L.kg_msm(None, 0, vp(d.ptr), None, vp(s.ptr), C.c_size_t(n), out) == BAD           # null context
    assert L.kg_msm_begin(h, 0, vp(d.ptr), None, vp(s.ptr), C.c_size_t(n), 4) == BAD          # ticket out of range
    assert L.kg_msm_end(h, 0, -1, out) == BAD
    assert L.kg_ntt_bn254_fr(h, vp(s.ptr), 0, 0, 0) == BAD                                    # log_n out of range
    assert L.kg_ntt_bn254_fr(h, vp(s.ptr), 29, 0, 0) == BAD
    assert L.kg_ntt_bn254_fr(h, None, 4, 0, 0) == BAD
    assert L.kg_field_vec_op(h, 9, 0, vp(s.ptr), vp(s.ptr), vp(s.ptr), C.c_size_t(n)) == BAD  # unknown field
    assert L.kg_field_vec_op(h, 0, 0, vp(s.ptr), None, vp(s.ptr), C.c_size_t(n)) == BAD       # binary op without b
    assert L.kg_msm_set_window(h, 21) == BAD                                            # 0 (automatic) .. 20
    assert L.kg_bases_register(h, 5, vp(d.ptr), None, C.c_size_t(n)) == BAD
    assert L.kg_groth16_prove_end(h, 0, out, None) == BAD                                     # null flags
    assert L.kg_groth16_prove_end(h, 1, out, (C.c_uint8 * 3)()) == BAD                        # nothing begun on ticket 1
    assert L.kg_groth16_prove_begin(h, None, None, None, None, None, None, None, None, 0) == BAD
    # zero-length inputs are valid: the identity
    assert L.kg_msm(h, 0, None, None, None, C.c_size_t(0), out) == 0 and not any(out[8:12])
    assert ctx.msm(0, d.ptr, 0, s.ptr, 0) is not None


def test_pipeline_with_complete_inputs_matches_blocking_calls(oracle):
    """kg_ctx_set_inputs_complete(1): the scalar side of MSM i+1 (digit extraction, sort, base conversion on the scalar queue)
    runs under the accumulation of MSM i, the two scalar-side spaces alternate, gathers and reductions run on the
    reduction queues, host finishes on worker threads.  Four tickets in flight over MSMs of different sizes, curves and
    window widths (one- and two-pass sorts, skewed scalars with several partial-sum rounds) must give exactly the blocking
    results, again and again over the same buffers."""
    import kogarashi_amd as K
    O = oracle
    ctx = K.Context(0)
    try:
        jobs = []
        for j, (curve, sfd, n) in enumerate([(0, 0, 70000), (1, 1, 3000), (0, 0, 1 << 17), (0, 0, 257), (1, 1, 66000), (0, 0, 5)]):
            b = O.gen_bases(curve, SEED + 1200 + j, 0, n)
            s_ = O.gen_scalars(sfd, SEED + 1210 + j, 0, n)
            if j in (2, 4):                            # witness-like skew: many partial sums per bucket
                s_[::2] = O.f_consts(sfd)["r"]
                s_[1::5] = 0
            inf = np.zeros(n, dtype=np.uint8)
            inf[n // 2] = 1
            jobs.append((curve, ctx.upload(b), ctx.upload(inf), ctx.upload(s_), n))
        want = [ctx.msm(c, db.ptr, di.ptr, ds.ptr, n) for c, db, di, ds, n in jobs]
        ctx.set_inputs_complete(True)
        for rep in range(3):
            order = list(range(len(jobs))) * 2
            got = [None] * len(order)
            depth = 4
            for i, jx in enumerate(order):
                c, db, di, ds, n = jobs[jx]
                ctx.msm_begin(c, db.ptr, di.ptr, ds.ptr, n, i % 4)
                if i >= depth - 1:
                    k = i - depth + 1
                    got[k] = ctx.msm_end(jobs[order[k]][0], k % 4)
            for k in range(max(len(order) - depth + 1, 0), len(order)):
                got[k] = ctx.msm_end(jobs[order[k]][0], k % 4)
            for k, jx in enumerate(order):
                assert (got[k] == want[jx]).all(), (rep, k, jx)
        # blocking calls and commits in the same mode
        for (c, db, di, ds, n), w in zip(jobs, want):
            assert (ctx.msm(c, db.ptr, di.ptr, ds.ptr, n) == w).all()
        ctx.set_inputs_complete(False)
        # default (stream-ordered) mode: an MSM right behind the kernel that produces its scalars, no synchronisation
        c, db, di, ds, n = jobs[0]
        d2 = ctx.empty((n, 4))
        ctx.field_vec_op(0, "double", ds.ptr, 0, d2.ptr, n)
        dbl = ctx.msm(c, db.ptr, di.ptr, d2.ptr, n)
        two = ctx.points_sum_affine(c, np.stack([want[0][:8], want[0][:8]]), np.zeros(2, dtype=np.uint8))
        assert (dbl[:8] == two[0]).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("cv,curve,sfd,n", [("g1", 0, 0, (1 << 19) + 77), ("gk", 1, 1, 70001), ("g2", 2, 0, 66000)])
def test_msm_host_slices_match_device_call(ctx, oracle, cv, curve, sfd, n):
    """kg_msm_host cuts large inputs into index slices that are uploaded, sorted and accumulated as a pipeline and adds the
    slices' sums: identical to kg_msm over the same arrays resident on the device (and to the oracle), with identity
    flags, zero scalars and a ragged length; repeated calls reuse the cached device buffers."""
    O = oracle
    if curve == 2:
        dk = ctx.upload(O.gen_scalars(0, SEED + 1400, 0, n))
        dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
        ctx.fixed_base_mul(2, dk.ptr, n, dxy.ptr, dinf.ptr)
        bases = dxy.numpy()
    else:
        bases = O.gen_bases(curve, SEED + 1401, 0, n)
    scal = O.gen_scalars(sfd, SEED + 1402, 0, n)
    scal[::1000] = 0
    inf = np.zeros(n, dtype=np.uint8)
    inf[[1, n // 4, n // 2 + 1, n - 1]] = 1
    db, di, ds = ctx.upload(bases), ctx.upload(inf), ctx.upload(scal)
    want = ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)
    for _ in range(3):
        assert (ctx.msm_host(curve, bases, inf, scal, n) == want).all()
    assert (ctx.msm_host(curve, bases, None, scal, n) == ctx.msm(curve, db.ptr, 0, ds.ptr, n)).all()
    m = n // 3                                          # a shorter call afterwards (fewer slices, same cached buffers)
    assert (ctx.msm_host(curve, bases, inf, scal, m) == ctx.msm(curve, db.ptr, di.ptr, ds.ptr, m)).all()
    nb = 8 if curve == 2 else 4
    assert gpu_aff(want, nb) == aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))


def _experiments_lib():
    """libkogarashi_amd_exp.so: the product's sources with -DKG_EXPERIMENTS (python -m kogarashi_amd.build --experiments; built by
    __graft_entry__.build()) -- the three kernels that lost their A/B runs exist there only, and so do their parity tests"""
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kogarashi_amd", "libkogarashi_amd_exp.so")
    if not os.path.exists(so):
        pytest.skip("libkogarashi_amd_exp.so is not built (python -m kogarashi_amd.build --experiments)")
    return so


def test_g2_accumulation_on_lane_pairs_gives_the_same_sum(oracle):
    """KG_G2_PAIR_ACC=1 (read once per process): k_acc_tasks<Fp2S> -- a lane pair per task, c0 on the even lane, c1 on the odd
    one, partial sums written in the one-lane layout -- against the oracle, in a child process; identity bases and zero
    scalars included, registered and per-call bases"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import kogarashi_amd as K
from oracle import oracle as O
S = 0x4B6F676172617368
ctx = K.Context(0)
n = 70001
dk = ctx.empty((n, 4)); ctx.gen_scalars(K.KG_FR, S + 950, 0, n, dk.ptr)
ctx.write(dk.ptr + 32 * 5, np.zeros((1, 4), dtype=np.uint64))
dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
ctx.fixed_base_mul(K.KG_G2, dk.ptr, n, dxy.ptr, dinf.ptr)
ds = ctx.empty((n, 4)); ctx.gen_scalars(K.KG_FR, S + 951, 0, n, ds.ptr)
ctx.write(ds.ptr + 32 * 9, np.zeros((1, 4), dtype=np.uint64))
want = O.to_affine("g2", O.msm("g2", dxy.numpy(), ds.numpy(), dinf.numpy(), threads=8))
got = ctx.msm(K.KG_G2, dxy.ptr, dinf.ptr, ds.ptr, n)
assert not want[1] and (got[:16] == want[0]).all()
ctx.bases_register(K.KG_G2, dxy.ptr, dinf.ptr, n)
got = ctx.msm(K.KG_G2, dxy.ptr, dinf.ptr, ds.ptr, n)
assert (got[:16] == want[0]).all()
print("ok")
""" % root
    exp = _experiments_lib()
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, KG_G2_PAIR_ACC="1", KG_LIB_PATH=exp), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("cv,curve,sfd,n", [("g1", 0, 0, (1 << 17) + 5), ("gk", 1, 1, 140001)])
def test_blocking_msm_in_window_groups_is_the_same_point(ctx, oracle, cv, curve, sfd, n):
    """kg_msm pipelines a blocking call against itself by window groups (kg_msm_set_groups; groth16/src/msm.rs:6-48 is a blocking
    call): the windows are sorted, accumulated and reduced group by group, top windows first, and the host's double-and-add chain runs
    through the groups in that order.  Every plan -- none, two, three, four groups, per-call bases and a registered array -- must give
    the oracle's point on an edge mix (identity bases, zero / one / minus-one scalars, repeated points)."""
    O = oracle
    bases, scal, inf = edge_mix(O, cv, curve, sfd, n, SEED + 700 + n)
    want = aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
    db, di, ds = ctx.upload(bases), ctx.upload(inf), ctx.upload(scal)
    try:
        for g in (1, 2, 3, 4, 0):
            ctx.set_msm_groups(g)
            assert gpu_aff(ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n), 4) == want, g
        ctx.bases_register(curve, db.ptr, di.ptr, n)
        for g in (2, 4):
            ctx.set_msm_groups(g)
            assert gpu_aff(ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n), 4) == want, ("registered", g)
        # a prefix of the registered array (zip semantics of msm_curve_addition), still in groups
        m = n - 1234
        want_m = aff(O, cv, O.msm(cv, bases[:m], scal[:m], inf[:m], threads=8))
        assert gpu_aff(ctx.msm(curve, db.ptr, di.ptr, ds.ptr, m), 4) == want_m
    finally:
        ctx.set_msm_groups(0)
        ctx.bases_unregister(db.ptr)
    import ctypes as C
    assert ctx._lib.kg_msm_set_groups(ctx._h, 5) == -2 and ctx._lib.kg_msm_set_groups(ctx._h, -1) == -2


def test_hot_bucket_tree_matches_the_lane_by_lane_rounds(oracle):
    """A 0/1-heavy scalar vector puts tens of thousands of entries into one bucket: its partial sums go through k_hot_sum / k_hot_fold
    (sixteen workgroup-wide trees per hot bucket, then one wave).  The older path -- rounds of sixteen partial sums per lane -- is still in
    the library (KG_HOT_SUM=0) for hot-bucket counts beyond one launch: both must give the oracle's point, pipelined and blocking."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import kogarashi_amd as K
from kogarashi_amd import synthetic as syn
from oracle import oracle as O
ctx = K.Context(0)
n = (1 << 18) + 77
bases = O.gen_bases(0, 4242, 0, n)
scal = O.gen_scalars(0, 4243, 0, n)
syn.witness_like(scal, 5)
want_xy, want_inf = O.to_affine("g1", O.msm("g1", bases, scal, None, threads=8))
db, ds = ctx.upload(bases), ctx.upload(scal)
for _ in range(2):
    got = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    assert not want_inf and (got[:8] == want_xy).all()
ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, 0); ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, 1)
assert (ctx.msm_end(K.KG_G1, 0)[:8] == want_xy).all() and (ctx.msm_end(K.KG_G1, 1)[:8] == want_xy).all()
print("ok")
""" % root
    for hot in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, KG_HOT_SUM=hot), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (hot, r.stderr[-2000:])


def test_first_sort_pass_variants_give_the_oracles_point(oracle):
    """The first pass of the two-pass sort exists in two kernels and several shapes -- k_group_scatter (1024-entry tiles, entries held in
    registers: KG_GS_TILE=0) and k_group_scatter_big (a tile walked twice; 4096 / 8192 entries, 256 / 512 / 1024 threads; 1024 threads
    and quarter-size conversion workgroups where a blocking call's sort runs alone, KG_SORT_ALONE=0 turns that off).  Every one of them
    must sort the same lists: blocking (window groups), pipelined and witness-like scalars against the oracle.  The accumulation's
    experiment kernel (KG_ACC_PREFETCH=1) rides along: same inputs, same point."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import kogarashi_amd as K
from kogarashi_amd import synthetic as syn
from oracle import oracle as O
ctx = K.Context(0)
for n, skew in (((1 << 17) + 4099, False), ((1 << 16) + 1, True)):
    bases = O.gen_bases(0, 777, 0, n)
    scal = O.gen_scalars(0, 778, 0, n)
    if skew: syn.witness_like(scal, 9)
    want_xy, want_inf = O.to_affine("g1", O.msm("g1", bases, scal, None, threads=8))
    db, ds = ctx.upload(bases), ctx.upload(scal)
    got = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    assert not want_inf and (got[:8] == want_xy).all(), (n, "blocking")
    ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, 0); ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, 1)
    assert (ctx.msm_end(K.KG_G1, 0)[:8] == want_xy).all() and (ctx.msm_end(K.KG_G1, 1)[:8] == want_xy).all(), (n, "pipelined")
print("ok")
""" % root
    for knobs in ({"KG_GS_TILE": "0"}, {"KG_GS_TILE": "8192", "KG_GS_NT": "512", "KG_GS_NT0": "256"}, {"KG_GS_TILE": "4096", "KG_GS_NT": "1024"},
                  {"KG_SORT_ALONE": "0"}, {"KG_ACC_PREFETCH": "1"},        # k_acc_tasks_q (bases through LDS, gathers shared by lane quads)
                  {"KG_HOT_SHIFT": "0"}, {"KG_HOT_SHIFT": "1"}):          # hot buckets cut no finer / twice finer than the others (default: four times)
        env = dict(os.environ, **knobs)
        if knobs.get("KG_GS_TILE") == "0" or "KG_ACC_PREFETCH" in knobs:      # kernels of the -DKG_EXPERIMENTS build only
            env["KG_LIB_PATH"] = _experiments_lib()
            script_k = script.replace('print("ok")', 'from kogarashi_amd.lib import experiments_built\nassert experiments_built()\nprint("ok")')
        else:
            script_k = script
        r = subprocess.run([sys.executable, "-c", script_k], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (knobs, r.stderr[-2000:])


def test_skewed_scalar_vectors_over_the_three_curves():
    """tools/dbg/stress_skew.py as a test: sixteen seeded configurations -- all scalars equal, one dominant value, a handful of values,
    witness-like, half zeros -- at ragged sizes 2^12..2^18 over G1 / Grumpkin / G2, blocking and pipelined, against the oracle.  The hot-bucket
    machinery (tasks written by the workgroup, the finer cut, the shares that follow the fullest bucket) meets every one of them."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dbg", "stress_skew.py"), "16"], env=dict(os.environ, KG_STRESS_SEED="21"),
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "failures: 0" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_service_queues_are_placed_off_the_main_queues_pipe(ctx):
    """capi.cpp place_queues: in a fresh process (none but the runtime's own streams, GPU_MAX_HW_QUEUES as the suite sets it) the probe
    finds exactly the candidates j and j + 4 on the main queue's compute pipe (2 <= code <= 5), also with never-used streams created
    first; in THIS process -- hundreds of streams and contexts behind it -- any answer is a status-clean code 0..5 (1 = no clear
    picture: creation order, which only costs speed)."""
    import os, subprocess, sys
    assert 0 <= ctx.queue_placement() <= 5
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = "import sys; sys.path.insert(0, %r)\nimport kogarashi_amd as K\nc = K.Context(0); print('placement', c.queue_placement()); c.close()" % root
    for env in ({}, {"KG_STREAM_PAD": "1,2"}):
        # (the probe's own switch is set explicitly: the suite also runs under non-default knob sets, tools/dbg/r5_knob_suites.sh)
        seen = []
        for attempt in range(3):                    # the probe is a timing measurement on a box whose host cores are shared: it repeats an
            r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, KG_QUEUE_PLACEMENT="1", **env), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-1500:]      # inconclusive pass itself (four times), and so does this test (three processes)
            seen.append(int(r.stdout.split("placement")[1].split()[0]))
            if 2 <= seen[-1] <= 5:
                break
        assert 2 <= seen[-1] <= 5 and all(v == 1 for v in seen[:-1]), (env, seen)
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, KG_QUEUE_PLACEMENT="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and int(r.stdout.split("placement")[1].split()[0]) == 0
