"""Pins the CPU oracle (oracle/kg_oracle.c) before it is trusted as the checker:
 1. every constant the reference holds for this path (quoted here as data, with file:line),
 2. the reference's own algebraic test properties (field_test!, curve_test!, msm test, fft tests),
 3. the committed golden vectors from the independent big-integer implementation (tests/golden/)."""
import json
import os

import numpy as np
import pytest

from helpers import CURVES, I, L, np_to_pt, pt_to_np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 0x4B6F676172617368


def gold(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


# constants exactly as written in the reference sources (little-endian u64 limbs)
REF = {
    "fr_modulus": [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029],   # bn254/src/fr.rs:11-16
    "fr_r": [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f],         # fr.rs:25-30
    "fr_r2": [0x1bb8e645ae216da7, 0x53fe3ab1e35c59e3, 0x8c49833d53bb8085, 0x0216d0b17f4e44a5],        # fr.rs:34-39
    "fr_r3": [0x5e94d8e1b4bf0040, 0x2a489cbe1cfbb6b8, 0x893cc664a19fcfed, 0x0cf8594b7fcc657c],        # fr.rs:43-48
    "fr_inv": 0xc2e1f593efffffff,                                                                      # fr.rs:51
    "fr_root_raw": [0xd34f1ed960c37c9c, 0x3215cf6dd39329c8, 0x98865ea93dd31f74, 0x03ddb9f5166d18b7],  # fr.rs:60-65
    "fq_modulus": [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029],   # bn254/src/fq.rs:10-15
    "fq_r": [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f],         # fq.rs:20-25
    "fq_r2": [0xf32cfc5b538afa89, 0xb5e71911d44501fb, 0x47ab1eff0a417ff6, 0x06d89f71cab8351f],        # fq.rs:28-33
    "fq_r3": [0xb1cd6dafda1530df, 0x62f210e6a7283db6, 0xef7f0b0c0ada0afb, 0x20fd6e902d592544],        # fq.rs:36-41
    "fq_inv": 0x87d20782e4866389,                                                                      # fq.rs:44
    "grumpkin_gen_y": [0x11b2dff1448c41d8, 0x23d3446f21c77dc3, 0xaa7b8cf435dfafbb, 0x14b34cf69dc25d68],  # grumpkin/src/params.rs:5-10
    "grumpkin_b": [0xdd7056026000005a, 0x223fa97acb319311, 0xcc388229877910c0, 0x034394632b724eaa],     # grumpkin/src/params.rs:13-18
}


def test_reference_constants(oracle, pyoracle):
    O, P = oracle, pyoracle
    for fd, tag, p in ((0, "fr", P.R_MOD), (1, "fq", P.Q_MOD)):
        c = O.f_consts(fd)
        assert [int(v) for v in c["p"]] == REF[f"{tag}_modulus"] and I(c["p"]) == p
        assert [int(v) for v in c["r"]] == REF[f"{tag}_r"] and I(c["r"]) == (1 << 256) % p
        assert [int(v) for v in c["r2"]] == REF[f"{tag}_r2"] and I(c["r2"]) == pow(2, 512, p)
        assert [int(v) for v in c["r3"]] == REF[f"{tag}_r3"] and I(c["r3"]) == pow(2, 768, p)
        assert c["inv"] == REF[f"{tag}_inv"] == (-pow(p, -1, 1 << 64)) % (1 << 64)
        # Fr::one() == R through the oracle's own to_mont_form (represent.rs:10-12)
        assert (O.f_to_mont(fd, L(1)) == c["r"]).all()
    # test_root_of_unity (bn254/src/fr.rs:315-320): ROOT_OF_UNITY^(2^28) == 1, and its order is exactly 2^28
    root = O.f_to_mont(0, np.array(REF["fr_root_raw"], dtype=np.uint64))
    x = root
    for i in range(28):
        if i == 27:
            assert not (x == O.f_consts(0)["r"]).all()
        x = O.f_square(0, x)
    assert (x == O.f_consts(0)["r"]).all()
    assert I(REF["fr_root_raw"]) == pow(7, (P.R_MOD - 1) >> 28, P.R_MOD)      # GENERATOR^t (fr.rs:55-59)
    # generators / curve constants (bn254/src/params.rs:8-57, grumpkin/src/params.rs:4-19)
    g1 = O.generator("g1")
    assert (g1[:4] == O.f_consts(1)["r"]).all() and (g1[4:] == O.f_to_mont(1, L(2))).all()
    gk = O.generator("gk")
    assert (gk[:4] == O.f_consts(0)["r"]).all() and [int(v) for v in gk[4:]] == REF["grumpkin_gen_y"]
    b, b3 = O.curve_b("gk")
    assert [int(v) for v in b] == REF["grumpkin_b"] and P.from_mont(I(b), P.R_MOD) == P.R_MOD - 17
    for cv in ("g1", "gk", "g2"):
        assert O.is_on_curve(cv, O.generator(cv))
        assert np_to_pt(CURVES[cv][1], O.generator(cv), 0) == CURVES[cv][1].gen


@pytest.mark.parametrize("fd,tag", [(0, "fr"), (1, "fq")])
def test_field_golden(oracle, pyoracle, fd, tag):
    O, P = oracle, pyoracle
    p = P.R_MOD if fd == 0 else P.Q_MOD
    m = lambda h: L(P.to_mont(int(h, 16), p))
    for row in gold("field")[tag]:
        a, b = m(row["a"]), m(row["b"])
        assert (O.f_add(fd, a, b) == m(row["add"])).all() and (O.f_sub(fd, a, b) == m(row["sub"])).all()
        assert (O.f_mul(fd, a, b) == m(row["mul"])).all() and (O.f_square(fd, a) == m(row["sqr"])).all()
        assert (O.f_neg(fd, a) == m(row["neg"])).all() and (O.f_double(fd, a) == m(row["dbl"])).all()
        inv = O.f_invert(fd, a)
        assert (inv is None) == (row["inv"] is None) and (inv is None or (inv == m(row["inv"])).all())
        assert I(O.f_from_mont(fd, a)) == int(row["a"], 16)


def gpt(cur, v):
    if v is None:
        return None
    from oracle.pyoracle import Fq2
    x = [int(h, 16) for h in v]
    return (Fq2(x[0], x[1]), Fq2(x[2], x[3])) if cur.ext else (x[0], x[1])


@pytest.mark.parametrize("cv", ["g1", "gk", "g2"])
def test_point_golden_and_group_laws(oracle, cv):
    O = oracle
    _, cur, sfd = CURVES[cv]
    g = gold("points")[cv]
    nb = 8 if cur.ext else 4
    one = np.concatenate([O.f_consts(1 if cv != "gk" else 0)["r"], np.zeros(nb - 4, dtype=np.uint64)])
    proj = lambda pt: np.concatenate([pt_to_np(cur, pt), one]) if pt is not None else np.concatenate([np.zeros(nb, dtype=np.uint64), one, np.zeros(nb, dtype=np.uint64)])
    for row in g["add"]:
        p, q, s = gpt(cur, row["p"]), gpt(cur, row["q"]), gpt(cur, row["sum"])
        got = O.add_affine(cv, pt_to_np(cur, p), p is None, pt_to_np(cur, q), q is None)
        assert np_to_pt(cur, *O.to_affine(cv, got)) == s
        got = O.add_mixed(cv, pt_to_np(cur, p), p is None, proj(q))
        assert np_to_pt(cur, *O.to_affine(cv, got)) == s
        got = O.add_projective(cv, proj(p), proj(q))
        assert np_to_pt(cur, *O.to_affine(cv, got)) == s
    # curve_test!: 7g + 16g = 23g via NAF scalar_point (zkstd/src/macros/curve/weierstrass/test.rs)
    gen = proj(cur.gen)
    k = lambda v: O.f_to_mont(sfd, L(v))
    s7, s16, s23 = (O.scalar_point(cv, gen, k(v)) for v in (7, 16, 23))
    assert np_to_pt(cur, *O.to_affine(cv, s23)) == gpt(cur, g["gen_times_23"])
    assert np_to_pt(cur, *O.to_affine(cv, O.add_projective(cv, s7, s16))) == gpt(cur, g["seven_plus_sixteen"]) == gpt(cur, g["gen_times_23"])
    # 8a == double^3(a)
    d = gen
    for _ in range(3):
        d = O.double_projective(cv, d)
    assert O.to_affine(cv, d)[0].tobytes() == O.to_affine(cv, O.scalar_point(cv, gen, k(8)))[0].tobytes()


@pytest.mark.parametrize("cv", ["g1", "gk", "g2"])
def test_msm_golden(oracle, pyoracle, cv):
    O, P = oracle, pyoracle
    _, cur, sfd = CURVES[cv]
    sp = P.R_MOD if sfd == 0 else P.Q_MOD
    for case in gold("msm")[cv]:
        bases = np.stack([pt_to_np(cur, gpt(cur, b)) for b in case["bases"]])
        scal = np.stack([L(P.to_mont(int(h, 16), sp)) for h in case["scalars"]])
        inf = np.array(case["inf"], dtype=np.uint8)
        for threads in (1, 4):
            got = O.msm(cv, bases, scal, inf, threads=threads)
            assert np_to_pt(cur, *O.to_affine(cv, got)) == gpt(cur, case["sum"]), (cv, case["n"])
        if len(case["scalars"]) == len(case["bases"]):
            xy, oi = O.commit_naive(cv, bases, scal, inf)
            assert np_to_pt(cur, xy, oi) == gpt(cur, case["sum"])


def test_msm_equals_naive_sum_like_reference_test(oracle):
    """multi_scalar_multiplication_test (groth16/src/msm.rs:118-135): Pippenger == sum of independent scalar muls, n = 32."""
    O = oracle
    n = 32
    bases, scal = O.gen_bases(0, SEED + 5, 0, n), O.gen_scalars(0, SEED + 6, 0, n)
    xy, inf = O.commit_naive("g1", bases, scal)
    axy, ainf = O.to_affine("g1", O.msm("g1", bases, scal))
    assert inf == ainf == 0 and (xy == axy).all()


def test_ntt_golden_and_fft_properties(oracle, pyoracle):
    O, P = oracle, pyoracle
    m = lambda hs: np.stack([L(P.to_mont(int(h, 16), P.R_MOD)) for h in hs])
    for case in gold("ntt"):
        f = O.Fft(case["k"])
        v = m(case["v"])
        assert (f.dft(v) == m(case["dft"])).all() and (f.idft(v) == m(case["idft"])).all()
        assert (f.coset_dft(v) == m(case["coset_dft"])).all() and (f.coset_idft(v) == m(case["coset_idft"])).all()
        assert (f.divide_by_z_on_coset(v) == m(case["divide_by_z_on_coset"])).all()
        assert (f.dft(v, threads=4) == m(case["dft"])).all()
    # fft_transformation_test (fft.rs:246-257), k = 10
    v = O.gen_scalars(0, SEED + 7, 0, 1 << 10)
    f = O.Fft(10)
    assert (f.idft(f.dft(v)) == v).all()
    # fft_multiplication_test (fft.rs:259-291): FFT product == schoolbook, k = 5
    a, b = O.gen_scalars(0, SEED + 8, 0, 16), O.gen_scalars(0, SEED + 9, 0, 16)
    f5 = O.Fft(5)
    prod = f5.idft(O.fr_vec("mul", f5.dft(a), f5.dft(b)))
    ai = [P.from_mont(I(x), P.R_MOD) for x in a]
    bi = [P.from_mont(I(x), P.R_MOD) for x in b]
    want = [0] * 32
    for i, x in enumerate(ai):
        for j, y in enumerate(bi):
            want[i + j] = (want[i + j] + x * y) % P.R_MOD
    assert [P.from_mont(I(x), P.R_MOD) for x in prod] == want


def test_groth16_oracle_proof_verifies_in_the_exponent(oracle, pyoracle):
    """Pins the oracle's restatement of ZkSnark::setup + Prover::create_proof without a pairing: with the toxic waste
    known, the discrete logs of A, B, C follow from the witness by big-integer arithmetic alone; the proof points must be
    those multiples of the generators, and a*b = alpha*beta + (sum_i ic_i x_i)*gamma + c*delta must hold (the Groth16
    verification equation, groth16/src/proof.rs, read in the exponent)."""
    O, P = oracle, pyoracle
    p = P.R_MOD
    fm = lambda x: P.from_mont(I(x), p)
    for m in (4, 16):
        cs = O.chain_r1cs(m, O.gen_scalars(0, SEED + 30 + m, 0, 1)[0])
        a, b, c = cs.evaluate()
        assert (O.fr_vec("mul", a, b) == c).all()                                  # is_sat (r1cs.rs:60-76)
        toxic = O.gen_scalars(0, SEED + 31, 0, 5)
        prm = O.groth16_params(cs, toxic, threads=4)
        r, s = O.gen_scalars(0, SEED + 32, 0, 2)
        A, B, C, inf = O.groth16_prove(cs, prm, r, s)
        assert not inf.any()
        alpha, beta, gamma, delta, tau = [fm(x) for x in toxic]
        sc = prm["scalars"]
        z = [fm(v) for v in np.concatenate([cs.x, cs.w])]
        a_s, b_s, l_s, ic_s, h_s = ([fm(v) for v in sc[k]] for k in ("a", "b", "l", "ic", "h"))
        k = max(1, (m - 1).bit_length())
        av, bv, cv = ([fm(v) for v in t] for t in (a, b, c))
        ac, bc, cc = (P.coset_dft(P.idft(t, k), k) for t in (av, bv, cv))
        q = P.coset_idft(P.divide_by_z_on_coset([(x * y - w) % p for x, y, w in zip(ac, bc, cc)], k), k)
        rr, ss = fm(r), fm(s)
        a_dl = (alpha + sum(zi * ai for zi, ai in zip(z, a_s)) + rr * delta) % p
        b_dl = (beta + sum(zi * bi for zi, bi in zip(z, b_s)) + ss * delta) % p
        c_dl = (sum(zi * li for zi, li in zip(z[cs.l:], l_s)) + sum(qi * hi for qi, hi in zip(q, h_s)) + ss * a_dl + rr * b_dl - rr * ss * delta) % p
        assert np_to_pt(P.G1, A, 0) == P.G1.mul(P.G1.gen, a_dl)
        assert np_to_pt(P.G2, B, 0) == P.G2.mul(P.G2.gen, b_dl)
        assert np_to_pt(P.G1, C, 0) == P.G1.mul(P.G1.gen, c_dl)
        assert a_dl * b_dl % p == (alpha * beta + sum(zi * ici for zi, ici in zip(z[:cs.l], ic_s)) * gamma + c_dl * delta) % p


def test_big_golden_msm_2_10_and_ntt_2_10(oracle, pyoracle):
    """tests/golden/big.json: BASELINE.json configs[0] (G1 MSM, 2^10 pairs) and the k = 10 transforms, inputs from the
    seeded streams (pinned by digest), outputs from the big-integer oracle."""
    from helpers import digest_limbs
    O, P = oracle, pyoracle
    g = gold("big")
    c = g["msm_2_10"]
    bases = O.gen_bases(0, int(c["seed_bases"], 16), 0, c["n"])
    scal = O.gen_scalars(0, int(c["seed_scalars"], 16), 0, c["n"])
    assert digest_limbs(bases, P.Q_MOD) == c["bases_digest"] and digest_limbs(scal, P.R_MOD) == c["scalars_digest"]
    for threads in (1, 8):
        assert np_to_pt(P.G1, *O.to_affine("g1", O.msm("g1", bases, scal, None, threads=threads))) == gpt(P.G1, c["sum"])
    c = g["ntt_2_10"]
    v = O.gen_scalars(0, int(c["seed"], 16), 0, 1 << c["k"])
    assert digest_limbs(v, P.R_MOD) == c["input_digest"]
    f = O.Fft(c["k"])
    for name in ("dft", "idft", "coset_dft", "coset_idft"):
        out = getattr(f, name)(v, threads=4)
        assert digest_limbs(out, P.R_MOD) == c[name]["digest"], name
        for i, h in zip(c[name]["sample_index"], c[name]["sample"]):
            assert P.from_mont(I(out[i]), P.R_MOD) == int(h, 16)


def groth16_tiny_case(P):
    """inputs of the fixed-(r, s) proof in tests/golden/big.json as Montgomery limbs"""
    c = gold("big")["groth16_tiny"]
    m = lambda h: L(P.to_mont(int(h, 16), P.R_MOD))
    return c, m(c["t0"]), np.stack([m(h) for h in c["toxic"]]), m(c["r"]), m(c["s"])


def test_big_golden_groth16_tiny(oracle, pyoracle):
    """One Groth16 proof with fixed toxic waste and (r, s) (shape of groth16/src/lib.rs:29-77): the C restatement's setup
    scalars, h coefficients' MSM inputs and proof must equal the integer-only fixture."""
    O, P = oracle, pyoracle
    c, t0, toxic, r, s = groth16_tiny_case(P)
    cs = O.chain_r1cs(c["m"], t0)
    prm = O.groth16_params(cs, toxic, threads=2)
    for name in ("h", "l", "a", "b", "ic"):
        want = [int(h, 16) for h in c["crs_scalars"][name]]
        assert [P.from_mont(I(v), P.R_MOD) for v in prm["scalars"][name]] == want, name
    A, B, C, inf = O.groth16_prove(cs, prm, r, s)
    assert not inf.any()
    assert np_to_pt(P.G1, A, 0) == gpt(P.G1, c["proof"]["a"])
    assert np_to_pt(P.G2, B, 0) == gpt(P.G2, c["proof"]["b"])
    assert np_to_pt(P.G1, C, 0) == gpt(P.G1, c["proof"]["c"])


# Public known answers of the SAME curve, independent of the reference and of this repository: the reference's bn254 is Ethereum's
# alt_bn128 (G1 generator (1, 2) on y^2 = x^3 + 3, bn254/src/params.rs:8-13; the G2 generator of params.rs:15-42 is EIP-197's), so the
# test vectors of the EIP-196 / EIP-197 precompiles apply to it.
EIP196_2G = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
             9918110051302171585080402603319702774565515993150576347155970296011118125764)        # ecMul((1, 2), 2) = ecAdd((1, 2), (1, 2))
EIP197_G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,       # x: real, imaginary
              11559732032986387107991004021392285783925812861821192530917403151452391805634),
             (8495653923123431417604973247489272438418190587263600148770280649306958101930,        # y: real, imaginary
              4082367875863433681332203403145435568316851327593401208105741076214120093531))


def test_public_alt_bn128_known_answers(oracle, pyoracle):
    O, P = oracle, pyoracle
    q, r = P.Q_MOD, P.R_MOD
    # the reference's G2 generator limbs (bn254/src/params.rs:15-42, canonical integers given to to_mont_form) ARE EIP-197's
    ref_g2 = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
              (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))
    assert ref_g2 == EIP197_G2
    mont = lambda v: O.f_to_mont(0, L(v))
    k = np.stack([mont(1), mont(2), mont(r - 1), mont(0)])
    # the oracle's generator multiples (restatement of `g * scalar`, zksnark.rs:57): 1 G, 2 G, (r - 1) G = -G, 0 G
    xy, inf = O.fixed_base_mul(0, k, threads=1)
    pts = [np_to_pt(CURVES["g1"][1], xy[i], inf[i]) for i in range(4)]
    assert pts[0] == (1, 2) and pts[1] == EIP196_2G and pts[2] == (1, q - 2) and pts[3] is None
    xy2, inf2 = O.fixed_base_mul(2, k[:1], threads=1)
    g2 = np_to_pt(CURVES["g2"][1], xy2[0], inf2[0])
    assert ((g2[0].a, g2[0].b), (g2[1].a, g2[1].b)) == EIP197_G2
    # the same through the restatement of msm_curve_addition (groth16/src/msm.rs:6-48) and through the independent big-integer curve
    g = pt_to_np(CURVES["g1"][1], (1, 2)).reshape(1, 8)
    got = O.to_affine("g1", O.msm("g1", np.repeat(g, 3, axis=0), np.stack([mont(1), mont(1), mont(r - 1)]), None, threads=1))
    assert np_to_pt(CURVES["g1"][1], got[0], got[1]) == (1, 2)                       # G + G - G
    got = O.to_affine("g1", O.msm("g1", g, np.stack([mont(2)]), None, threads=1))
    assert np_to_pt(CURVES["g1"][1], got[0], got[1]) == EIP196_2G
    cur = CURVES["g1"][1]
    assert cur.add((1, 2), (1, 2)) == EIP196_2G and cur.mul((1, 2), 2) == EIP196_2G and cur.mul((1, 2), r) is None
