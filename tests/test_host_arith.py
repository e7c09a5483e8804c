"""The device arithmetic (fp29.h / curve.h) compiled for the host, checked against the oracle.

Every case runs twice: with the plain types the kernels use, and with the bound-tracking FpChecked shadow,
which aborts the process if any 64-bit column, 32-bit limb or Montgomery value bound could overflow."""
import ctypes as C
import random

import numpy as np
import pytest

from helpers import CURVES, U8P, np_to_pt, p32, pt_to_np

SEED = 0x4B6F676172617368


def field_ops(H, field, checked, op, a, b):
    o = np.empty_like(a)
    H.ht_field_ops(field, checked, op, p32(a.view(np.uint32)), p32(b.view(np.uint32)), p32(o.view(np.uint32)), C.c_size_t(a.shape[0]))
    return o


@pytest.mark.parametrize("fd", [0, 1])
def test_prime_field_ops_match_oracle(hostlib, oracle, fd):
    O, n = oracle, 1500
    a, b = O.gen_scalars(fd, SEED + 1, 0, n), O.gen_scalars(fd, SEED + 2, 0, n)
    c = O.f_consts(fd)
    a[0] = 0; b[1] = 0; a[2] = c["r"]; a[3] = O.f_neg(fd, c["r"]); b[3] = a[3]; a[4] = a[5]; b[4] = a[5]
    pm1 = c["p"].copy(); pm1[0] -= 1
    a[6] = O.f_to_mont(fd, pm1); b[6] = a[6]            # (p-1)*(p-1)
    ops = {0: lambda x, y: O.f_mul(fd, x, y), 1: lambda x, y: O.f_square(fd, x), 2: lambda x, y: O.f_add(fd, x, y),
           3: lambda x, y: O.f_sub(fd, x, y), 4: lambda x, y: O.f_neg(fd, x), 5: lambda x, y: O.f_double(fd, x),
           7: lambda x, y: O.f_add(fd, O.f_mul(fd, x, y), O.f_square(fd, x)), 8: lambda x, y: x}
    for op, f in ops.items():
        exp = np.stack([f(a[i], b[i]) for i in range(n)])
        for chk in (0, 1):
            got = field_ops(hostlib, fd, chk, op, a, b)
            assert (got == exp).all(), (fd, op, chk)
    m = 40
    exp = np.stack([O.f_invert(fd, a[i]) if a[i].any() else np.zeros(4, dtype=np.uint64) for i in range(m)])
    for chk in (0, 1):
        assert (field_ops(hostlib, fd, chk, 6, a[:m].copy(), b[:m].copy()) == exp).all()
    # the binary-GCD inversion the kernels use (fp_inv.h) against the oracle's invert on every element, edge values included
    # (0 -> 0, 1, -1, p - 1 squared's operand, equal operands)
    exp = np.stack([O.f_invert(fd, a[i]) if a[i].any() else np.zeros(4, dtype=np.uint64) for i in range(n)])
    assert (field_ops(hostlib, fd, 0, 9, a.copy(), b.copy()) == exp).all()


@pytest.mark.parametrize("fd", [0, 1])
def test_host_field_type_matches_oracle(hostlib, oracle, fd):
    """host_fp.h (4 x 64-bit Montgomery words: the type of the MSM's host finish and of the prover's assembly) against the oracle: the
    no-carry product on random operands and on the edges (0, 1, -1, (p - 1)^2, operands with all-ones low words), sums, differences, inverses"""
    O, n = oracle, 3000
    a, b = O.gen_scalars(fd, SEED + 5, 0, n), O.gen_scalars(fd, SEED + 6, 0, n)
    c = O.f_consts(fd)
    pm1 = c["p"].copy(); pm1[0] -= 1                    # p - 1 as a raw word pattern: the largest operand the type holds
    a[0] = 0; b[1] = 0; a[2] = c["r"]; a[3] = O.f_neg(fd, c["r"]); b[3] = a[3]; a[4] = pm1; b[4] = pm1; a[5] = pm1; b[6] = pm1
    a[7] = np.array([2**64 - 1, 2**64 - 1, 2**64 - 1, 0], dtype=np.uint64); b[7] = a[7]
    a[8] = np.array([2**64 - 1, 2**64 - 1, 2**64 - 1, int(c["p"][3]) - 1], dtype=np.uint64); b[8] = a[8]
    u64p = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint64))
    ops = {0: lambda x, y: O.f_mul(fd, x, y), 1: lambda x, y: O.f_square(fd, x), 2: lambda x, y: O.f_add(fd, x, y),
           3: lambda x, y: O.f_sub(fd, x, y), 5: lambda x, y: O.f_double(fd, x)}
    for op, f in ops.items():
        got = np.empty_like(a)
        hostlib.ht_hostfp_ops(fd, op, u64p(a), u64p(b), u64p(got), C.c_size_t(n))
        assert (got == np.stack([f(a[i], b[i]) for i in range(n)])).all(), (fd, op)
    m = 24
    got = np.empty_like(a[:m])
    am, bm = a[:m].copy(), b[:m].copy()
    hostlib.ht_hostfp_ops(fd, 4, u64p(am), u64p(bm), u64p(got), C.c_size_t(m))
    assert (got == np.stack([O.f_invert(fd, a[i]) if a[i].any() else np.zeros(4, dtype=np.uint64) for i in range(m)])).all()


@pytest.mark.parametrize("fd", [0, 1])
def test_montgomery_domain_conversions(hostlib, oracle, fd):
    O, n = oracle, 500
    a = O.gen_scalars(fd, SEED + 3, 0, n)
    a[0] = 0
    o = np.empty_like(a)
    hostlib.ht_ref_to_int(fd, p32(a.view(np.uint32)), p32(o.view(np.uint32)), C.c_size_t(n))
    assert (o == np.stack([O.f_from_mont(fd, a[i]) for i in range(n)])).all()
    o2 = np.empty_like(a)
    hostlib.ht_int_to_ref(fd, p32(o.view(np.uint32)), p32(o2.view(np.uint32)), C.c_size_t(n))
    assert (o2 == a).all()


def test_fq2_ops_match_oracle(hostlib, oracle):
    O, n = oracle, 600
    a = np.concatenate([O.gen_scalars(1, SEED + 3, 0, n), O.gen_scalars(1, SEED + 4, 0, n)], axis=1)
    b = np.concatenate([O.gen_scalars(1, SEED + 5, 0, n), O.gen_scalars(1, SEED + 6, 0, n)], axis=1)
    a[0] = 0; b[1, :4] = 0; a[2, 4:] = 0; b[3] = a[3]
    for op, f in {0: lambda x, y: O.f2_mul(x, y), 1: lambda x, y: O.f2_square(x)}.items():
        exp = np.stack([f(a[i], b[i]) for i in range(n)])
        for chk in (0, 1):
            assert (field_ops(hostlib, 2, chk, op, a, b) == exp).all(), (op, chk)
    exp = np.stack([O.f2_invert(a[i]) for i in range(3, 15)])
    for chk in (0, 1):
        assert (field_ops(hostlib, 2, chk, 6, a[3:15].copy(), b[3:15].copy()) == exp).all()
    exp = np.stack([O.f2_invert(a[i]) for i in range(3, 200)])
    assert (field_ops(hostlib, 2, 0, 9, a[3:200].copy(), b[3:200].copy()) == exp).all()          # Fq2 through the binary GCD


def curve_sum(H, cid, cur, chk, mode, pts, inf):
    arr = np.stack([pt_to_np(cur, pt) for pt in pts])
    infa = np.array(inf, dtype=np.uint8)
    out = np.zeros(arr.shape[1], dtype=np.uint64)
    r = H.ht_curve_sum(cid, chk, mode, p32(arr.view(np.uint32)), infa.ctypes.data_as(U8P), C.c_size_t(len(pts)), p32(out.view(np.uint32)))
    return np_to_pt(cur, out, r)


@pytest.mark.parametrize("name", ["g1", "gk", "g2"])
def test_xyzz_point_formulas_match_bigint_oracle(hostlib, pyoracle, name):
    """Group-law cases of the reference's curve_test! (zkstd/src/macros/curve/weierstrass/test.rs:2-224) plus the
    exceptional branches of weierstrass.rs (equal points, inverse points, identities in the input)."""
    _check_point_formulas(hostlib, pyoracle, name)


def test_g2_formulas_with_the_lane_pair_product(hostlib_pm, pyoracle):
    """The same G2 cases with every Fq2 product computed by mul2pm (a*b + sigma*c*d over signed columns, the routine the
    lane-pair type of csrc/fp2s.h runs on the device): values against the oracle, worst-case bounds by FpChecked."""
    _check_point_formulas(hostlib_pm, pyoracle, "g2")


def _check_point_formulas(hostlib, pyoracle, name):
    P = pyoracle
    cid, cur, _ = CURVES[name]
    rnd = random.Random(5)
    if cur.ext:
        base = [cur.mul(cur.gen, rnd.randrange(1, cur.n)) for _ in range(5)]
    else:
        base = [P.base_at(cur, SEED + 100, i) for i in range(10)]
    A, B = base[0], base[1]
    cases = [base, [A, A], [A, cur.neg(A)], [A, A, A], [A, B, cur.neg(A), cur.neg(B)], [A, B, cur.add(A, B)],
             [A, B, cur.neg(cur.add(A, B))], [cur.add(A, A), A, A], [cur.add(A, A), cur.neg(A), cur.neg(A)], [A],
             [A, B, A, B, A, B, A], base + base, base + [cur.neg(x) for x in base]]
    for ci, pts in enumerate(cases):
        for some_inf in (False, True):
            inf = [0] * len(pts)
            if some_inf and len(pts) > 2:
                inf[1] = 1; inf[-1] = 1
            exp = None
            for pt, f in zip(pts, inf):
                if not f:
                    exp = cur.add(exp, pt)
            for chk in (0, 1):
                for mode in (0, 1, 2, 7, 8, 9, 12, 14):     # 9: the streaming formulas in place on their first operand; 12 / 14: the lane-cooperative addition (coop_add.h), temporaries in LDS / in registers
                    assert curve_sum(hostlib, cid, cur, chk, mode, pts, inf) == exp, (name, ci, some_inf, chk, mode)
                assert curve_sum(hostlib, cid, cur, chk, 13, pts, inf) == cur.mul(exp, 4), (name, ci, "coop dbl")
                assert curve_sum(hostlib, cid, cur, chk, 15, pts, inf) == cur.mul(exp, 4), (name, ci, "coop dbl, register form")
                assert curve_sum(hostlib, cid, cur, chk, 5, pts, inf) == cur.neg(exp)
                assert curve_sum(hostlib, cid, cur, chk, 3, pts, inf) == cur.add(exp, exp), (name, ci, "dbl")
    # mode 11: the short-input kernel's shape (msm_small_kernels.h) -- task sums by add_mixed_signed whose X is not value-reduced, doubled as
    # they are (a plane of a two-bucket range is a copy of a task sum), the planes added by a tree: sum_i 2^i (P_2i - P_2i+1)
    for pts in (base[:2], base[:4], (base + base)[:10], [A, A, B, cur.neg(B)], [A, cur.neg(A), B, B]):
        for some_inf in (False, True):
            inf = [0] * len(pts)
            if some_inf:
                inf[1] = 1
            exp = None
            for i in range(0, len(pts) - 1, 2):
                t = None
                if not inf[i]:
                    t = cur.add(t, pts[i])
                if not inf[i + 1]:
                    t = cur.add(t, cur.neg(pts[i + 1]))
                exp = cur.add(exp, cur.mul(t, 1 << (i // 2)) if t is not None else None)
            for chk in (0, 1):
                assert curve_sum(hostlib, cid, cur, chk, 11, pts, inf) == exp, (name, "small", len(pts), some_inf, chk)
    for chk in (0, 1):
        assert curve_sum(hostlib, cid, cur, chk, 4, [A] * 5, [0] * 5) == cur.mul(A, 32)
        assert curve_sum(hostlib, cid, cur, chk, 10, [A] * 5, [0] * 5) == cur.mul(A, 32)       # double_xyzz_stream in place


def test_ntt_butterfly_network_bounds_and_values(hostlib, oracle):
    """The register-resident 3-stage DIT network of the NTT kernel (ntt_core.h): values vs the oracle's field ops, and
    worst-case bounds via FpChecked over 3 chained passes (K grows by up to 4 per stage, limbs stay lazy inside a pass)."""
    O = oracle
    rng = np.random.default_rng(11)
    for trial in range(20):
        data = O.gen_scalars(0, SEED + 700 + trial, 0, 8)
        if trial == 0:
            data[:] = 0xFFFFFFFFFFFFFFFF                      # non-canonical all-ones input (value < 2^256)
        tw = O.gen_scalars(0, SEED + 800 + trial, 0, 7)
        rounds, trivial = 3, trial % 2
        # reference: same network with fully reduced oracle arithmetic, on values taken mod p
        # raw 256-bit loads stand for their residue mod p: from_mont(to_mont(v)) = v mod p
        x = [data[k].copy() for k in range(8)]
        x = [O.f_from_mont(0, O.f_to_mont(0, v)) for v in x]
        for r in range(rounds):
            for t in (1, 2, 3):
                half = 1 << (t - 1)
                for pi in range(4):
                    k0 = ((pi >> (t - 1)) << t) | (pi & (half - 1)); k1 = k0 + half
                    if trivial and r == 0 and (k0 & (half - 1)) == 0:        # w = 1: product skipped
                        tt = x[k1]
                    else:
                        w = tw[0] if t == 1 else (tw[1 + (k0 & 1)] if t == 2 else tw[3 + (k0 & 3)])
                        tt = O.f_mul(0, x[k1], w)
                    a = x[k0]
                    x[k0], x[k1] = O.f_add(0, a, tt), O.f_sub(0, a, tt)
        want = np.stack(x)
        for chk in (0, 1):
            out = np.zeros((8, 4), dtype=np.uint64)
            hostlib.ht_ntt_network(chk, p32(data.view(np.uint32)), p32(tw.view(np.uint32)), rounds, trivial, p32(out.view(np.uint32)))
            assert (out == want).all(), (trial, chk)


@pytest.mark.parametrize("fd", [0, 1])
def test_cross_term_row_bounds_and_values(hostlib, oracle, fd):
    """vecops.h: the lazy row products and the cross-term combination of kg_nova_cross_term (nova/src/prover.rs:53-90) with
    the bound-tracking type (no column / limb / value overflow for ANY operands) and against the oracle's restatement."""
    O = oracle
    c = O.f_consts(fd)
    pm1 = O.f_to_mont(fd, (lambda v: (v.__setitem__(0, v[0] - 1), v)[1])(c["p"].copy()))
    for cnt, seed in ((1, 1), (3, 2), (9, 3), (40, 4)):
        va, vb, vc = (O.gen_scalars(fd, SEED + 700 + 10 * seed + j, 0, cnt) for j in range(3))
        z1, z2 = O.gen_scalars(fd, SEED + 710 + seed, 0, cnt), O.gen_scalars(fd, SEED + 720 + seed, 0, cnt)
        u1, u2 = O.gen_scalars(fd, SEED + 730 + seed, 0, 1)[0], c["r"]
        if seed == 4:                                   # worst-case magnitudes: everything p - 1
            va[:], vb[:], vc[:], z1[:], z2[:] = pm1, pm1, pm1, pm1, pm1
            u1 = pm1
        rp = np.array([0, cnt], dtype=np.uint64)
        col = np.arange(cnt, dtype=np.uint64)
        want = O.nova_cross_term(fd, (rp, col, va), (rp, col, vb), (rp, col, vc), z1, z2, u1, u2)[0]
        for chk in (0, 1):
            out = np.empty(4, dtype=np.uint64)
            a32 = lambda x: p32(np.ascontiguousarray(x).view(np.uint32))
            hostlib.ht_cross_term(fd, chk, a32(va), a32(vb), a32(vc), a32(z1), a32(z2), C.c_size_t(cnt), a32(u1), a32(u2), p32(out.view(np.uint32)))
            assert (out == want).all(), (fd, cnt, chk)


@pytest.mark.parametrize("c", range(2, 17))
def test_signed_window_digits_recompose_the_scalar(hostlib, c):
    """msm_digits.h (the short-input MSM's digit rule, the same as the long pipeline's window_digit): for every width c the signed digits of
    k + H recompose k, every |digit| is a bucket number <= 2^(c-1), and the top window is unsigned -- on random canonical scalars of both
    fields and the edge values 0, 1, p - 1, 2^(c-1) boundaries."""
    rnd = random.Random(100 + c)
    r_mod = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    q_mod = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    ks = [0, 1, 2, r_mod - 1, q_mod - 1, (1 << (c - 1)), (1 << (c - 1)) - 1, (1 << c) - 1, (1 << 253) + 12345, (1 << 254) - 1]
    ks += [rnd.randrange(r_mod) for _ in range(200)] + [rnd.randrange(q_mod) for _ in range(100)]
    ks += [sum(((1 << (c - 1)) - (j & 1)) << (j * c) for j in range(254 // c)) % q_mod]            # digits at the sign boundary in every window
    arr = np.array([[(k >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for k in ks], dtype=np.uint32)
    dig = np.zeros((len(ks), 128), dtype=np.int32)
    hostlib.ht_small_digits.restype = C.c_int
    W = hostlib.ht_small_digits(p32(arr), C.c_size_t(len(ks)), c, dig.ctypes.data_as(C.POINTER(C.c_int32)))
    assert W == (255 + c - 1) // c
    for k, row in zip(ks, dig):
        assert sum(int(row[w]) << (w * c) for w in range(W)) == k
        assert all(abs(int(row[w])) <= 1 << (c - 1) for w in range(W)) and row[W - 1] >= 0


@pytest.mark.parametrize("fd,name", [(0, "g1"), (1, "gk")])
def test_glv_decomposition_recomposes_the_scalar(hostlib, pyoracle, fd, name):
    """msm_digits.h glv_decompose_with: k = k1 + k2 lambda (mod n) with |k1|, |k2| < 2^126.6 for random scalars and the edges (0, 1, n - 1,
    lambda, n - lambda, 2^253, the largest 254-bit value below n) -- lambda and the lattice from tools/gen/glv_consts.py, recomputed here"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("glv_consts", os.path.join(root, "tools", "gen", "glv_consts.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    curve = pyoracle.G1 if fd == 0 else pyoracle.GRUMPKIN
    n = curve.n
    lam = G.consts(curve)[0]
    rnd = random.Random(77 + fd)
    ks = [0, 1, 2, n - 1, n - 2, n // 2, n // 3, lam, n - lam, 1 << 253, n - 3, (1 << 128) - 1, 1 << 128, (1 << 127) + 5] + [rnd.randrange(n) for _ in range(20000)]
    kw = np.array([[(k >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for k in ks], dtype=np.uint32)
    k1 = np.zeros((len(ks), 4), dtype=np.uint32); k2 = np.zeros_like(k1); neg = np.zeros((len(ks), 2), dtype=np.uint8)
    hostlib.ht_glv_decompose(fd, p32(kw), C.c_size_t(len(ks)), p32(k1), p32(k2), neg.ctypes.data_as(U8P))
    val = lambda w: sum(int(w[j]) << (32 * j) for j in range(4))
    for i, k in enumerate(ks):
        a, b = val(k1[i]), val(k2[i])
        assert a < 3 << 125 and b < 3 << 125, (i, hex(k))      # 3/4 (|a1| + |a2|) < 2^126.4: what keeps every width's top digit in range
        sa, sb = (-a if neg[i, 0] else a), (-b if neg[i, 1] else b)
        assert (sa + sb * lam - k) % n == 0, (i, hex(k))
    # every window width: with W = ceil(128 / c) windows and the bias H of msm_digits.h the unsigned top window of |k_i| + H stays within its
    # 2^(c-1) buckets (an entry beyond them would belong to no workgroup of the short-input kernel)
    worst = max(max(val(k1[i]), val(k2[i])) for i in range(len(ks)))
    for c in range(2, 11):
        W = (128 + c - 1) // c
        H = sum(1 << (w * c + c - 1) for w in range(W - 1))
        assert (worst + H) >> ((W - 1) * c) <= 1 << (c - 1), c
        assert ((3 << 125) + H) >> ((W - 1) * c) <= 1 << (c - 1), c        # and for the proven bound


@pytest.mark.parametrize("c", [2, 3, 4, 5, 8, 10])
def test_glv_digit_path_recomposes_the_scalar(hostlib, pyoracle, c):
    """what k_msm_small / k_small_prep do to a scalar with halved scalars, on the host: decompose, bias, cut W = ceil(128 / c) windows per half,
    fold the half's sign into the digit's -- sum_w d1_w 2^(cw) + lambda sum_w d2_w 2^(cw) = k (mod n), every |digit| a bucket number <= 2^(c-1)"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("glv_consts", os.path.join(root, "tools", "gen", "glv_consts.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    for fd, curve in ((0, pyoracle.G1), (1, pyoracle.GRUMPKIN)):
        n = curve.n
        lam = G.consts(curve)[0]
        rnd = random.Random(500 + 10 * c + fd)
        ks = [0, 1, n - 1, n // 2, lam, n - lam, (1 << 253) + 1, n - 3] + [rnd.randrange(n) for _ in range(3000)]
        kw = np.array([[(k >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for k in ks], dtype=np.uint32)
        dig = np.zeros((len(ks), 2, 64), dtype=np.int32)
        hostlib.ht_glv_digits.restype = C.c_int
        W = hostlib.ht_glv_digits(fd, p32(kw), C.c_size_t(len(ks)), c, dig.ctypes.data_as(C.POINTER(C.c_int32)))
        assert W == (128 + c - 1) // c
        for k, row in zip(ks, dig):
            k1 = sum(int(row[0][w]) << (w * c) for w in range(W))
            k2 = sum(int(row[1][w]) << (w * c) for w in range(W))
            assert (k1 + k2 * lam - k) % n == 0
            assert all(abs(int(row[e][w])) <= 1 << (c - 1) for e in range(2) for w in range(W))
