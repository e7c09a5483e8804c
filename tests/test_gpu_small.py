"""The short-input MSM (csrc/msm_small.hip: one launch up to 1536 pairs, two or three above; blocking kg_msm / kg_msm_host / kg_commit take it
for n <= 32768 by default, kg_msm_begin for n <= 8192) against the oracle's restatement of msm_curve_addition (groth16/src/msm.rs:6-48) -- at the lengths the
reference's own tests and bench use (msm.rs:118-135: 32 pairs; bn254/benches: 2^10), every length 1 .. 64, every window width and
bucket-range shape the knob admits, all three curves, the edge mixes, maximally skewed inputs, and against the long pipeline."""
import numpy as np
import pytest

from test_gpu_parity import SEED, aff, edge_mix, gpu_aff

pytestmark = pytest.mark.gpu
SMALL_DEFAULT = 32768       # KG_SMALL_MAX


@pytest.fixture(scope="module")
def ctx():
    import kogarashi_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _g2_bases(O, n, seed):
    g = O.generator("g2")
    one = np.concatenate([O.f_consts(1)["r"], np.zeros(4, dtype=np.uint64)])
    gen_proj = np.concatenate([g, one])
    ks = O.gen_scalars(0, seed, 0, n)
    return np.stack([O.to_affine("g2", O.scalar_point("g2", gen_proj, ks[i]))[0] for i in range(n)])


def test_every_length_up_to_64_matches_the_oracle(ctx, oracle):
    """n = 1 .. 64 (msm.rs:118-135 tests 32 pairs): resident arrays through kg_msm, host arrays through kg_msm_host; one flagged
    identity and one zero scalar from n = 8 (edge_mix)."""
    O = oracle
    bases_all, scal_all, inf_all = edge_mix(O, "g1", 0, 0, 64, SEED + 900)
    db, ds, di = ctx.upload(bases_all), ctx.upload(scal_all), ctx.upload(inf_all)
    for n in range(1, 65):
        want = aff(O, "g1", O.msm("g1", bases_all[:n], scal_all[:n], inf_all[:n], threads=1))
        assert gpu_aff(ctx.msm(0, db.ptr, di.ptr, ds.ptr, n), 4) == want, n
        if n % 7 == 0:
            assert gpu_aff(ctx.msm_host(0, bases_all[:n], inf_all[:n], scal_all[:n], n), 4) == want, n


@pytest.mark.parametrize("cv,curve,sfd,nb", [("g1", 0, 0, 4), ("gk", 1, 1, 4), ("g2", 2, 0, 8)])
@pytest.mark.parametrize("n", [33, 700, 1024, 2100, 4096, 7000])
def test_three_curves_at_the_plans_lengths(ctx, oracle, cv, curve, sfd, nb, n):
    """every row of the length -> shape table (msm_small_plan), the split windows from 2^11 pairs included, on G1, Grumpkin and G2"""
    O = oracle
    if curve == 2:
        if n > 1100:
            pytest.skip("G2 bases come from the oracle's scalar multiplication: a second per hundred")
        bases = _g2_bases(O, n, SEED + 910 + n)
        scal = O.gen_scalars(0, SEED + 911 + n, 0, n)
        inf = np.zeros(n, dtype=np.uint8)
        inf[3] = 1; scal[2] = 0; bases[5] = bases[4]; bases[9] = bases[8]; scal[9] = O.f_neg(0, scal[8])
    else:
        bases, scal, inf = edge_mix(O, cv, curve, sfd, n, SEED + 920 + n)
    want = aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
    assert gpu_aff(ctx.msm_host(curve, bases, inf, scal, n), nb) == want
    assert gpu_aff(ctx.msm_host(curve, bases, None, scal, n), nb) == aff(O, cv, O.msm(cv, bases, scal, None, threads=8))


def test_every_shape_gives_the_long_pipelines_point(ctx, oracle):
    """window widths 2 .. 10 and every bucket range the knob admits (kg_msm_set_small), at four lengths: the same affine point as the long
    pipeline (the knob at 0 pairs) and the oracle"""
    O = oracle
    for n in (37, 300, 1000, 3000):
        bases, scal, inf = edge_mix(O, "g1", 0, 0, n, SEED + 930 + n)
        want = aff(O, "g1", O.msm("g1", bases, scal, inf, threads=8))
        ctx.set_msm_small(0)
        try:
            assert gpu_aff(ctx.msm_host(0, bases, inf, scal, n), 4) == want
        finally:
            ctx.set_msm_small(SMALL_DEFAULT)
        for c in range(2, 11):
            for r in sorted({-1, 0, 1, min(c - 1, 3), min(c - 1, 5), min(c - 1, 7)}):
                ctx.set_msm_small(SMALL_DEFAULT, c, r)
                try:
                    assert gpu_aff(ctx.msm_host(0, bases, inf, scal, n), 4) == want, (n, c, r)
                finally:
                    ctx.set_msm_small(SMALL_DEFAULT, 0, -1)


def test_longest_inputs_of_the_short_path(ctx, oracle):
    """the lengths around the rows of the shape table and around 8192 (the longest input of the form whose workgroups convert the scalars
    themselves: a 13-bit index field), and a lower limit set through the knob"""
    O = oracle
    for n in (1536, 1537, 3072, 3073, 6144, 6145, 8191, 8192, 8193):
        bases, scal, inf = edge_mix(O, "gk", 1, 1, n, SEED + 940 + n)
        want = aff(O, "gk", O.msm("gk", bases, scal, inf, threads=8))
        assert gpu_aff(ctx.msm_host(1, bases, inf, scal, n), 4) == want, n
        ctx.set_msm_small(4096)
        try:
            assert gpu_aff(ctx.msm_host(1, bases, inf, scal, n), 4) == want, n
        finally:
            ctx.set_msm_small(SMALL_DEFAULT)


def test_the_form_with_scalars_converted_once_up_to_2_15_pairs(ctx, oracle):
    """beyond 2048 pairs the scalars are converted by a launch of their own and the workgroups cut digits from the word planes (the KT
    form; its lists hold 4096 entries in LDS and spill to global memory beyond): the lengths up to the 15-bit index field, the
    automatic shape and a few forced ones, uniform scalars and ONE scalar repeated (every entry of a window in one workgroup: the spill)"""
    O = oracle
    ctx.set_msm_small(32768)
    try:
        for n in (2049, 5000, 12000, 16384, 32767, 32768):
            bases, scal, inf = edge_mix(O, "g1", 0, 0, n, SEED + 970 + n)
            want = aff(O, "g1", O.msm("g1", bases, scal, inf, threads=8))
            assert gpu_aff(ctx.msm_host(0, bases, inf, scal, n), 4) == want, n
            for c, r in ((8, 3), (9, 4), (5, 1)):
                ctx.set_msm_small(32768, c, r)
                assert gpu_aff(ctx.msm_host(0, bases, inf, scal, n), 4) == want, (n, c, r)
            ctx.set_msm_small(32768, 0, -1)
        n = 20000
        bases = O.gen_bases(1, SEED + 975, 0, n)
        k = O.gen_scalars(1, SEED + 976, 0, 1)[0]
        scal = np.tile(k, (n, 1))
        assert gpu_aff(ctx.msm_host(1, bases, None, scal, n), 4) == aff(O, "gk", O.msm("gk", bases, scal, None, threads=8))
        one = np.tile(O.f_consts(1)["r"], (n, 1))
        assert gpu_aff(ctx.msm_host(1, bases, None, one, n), 4) == aff(O, "gk", O.msm("gk", bases, one, None, threads=8))
    finally:
        ctx.set_msm_small(SMALL_DEFAULT, 0, -1)


def test_g2_up_to_its_longest_short_input(ctx, oracle):
    """G2 beyond the lengths the oracle's scalar multiplication makes bases for in seconds: generator multiples from kg_fixed_base_mul
    (== the oracle's scalar_point, tests/test_gpu_groth16.py), one identity with its flag; 2049 .. 20480 pairs (the form with converted
    scalars; 20480 is msm_small_plan's longest G2 input, 20481 the long pipeline's), witness-like scalars at one length"""
    import kogarashi_amd as K
    from test_gpu_large import _g2_bases as dev_g2_bases, _witness_like
    O = oracle
    for n in (1537, 2049, 6000, 16385, 20480, 20481):
        dxy, dinf = dev_g2_bases(ctx, O, n, SEED + 980 + n)
        scal = O.gen_scalars(0, SEED + 981 + n, 0, n)
        if n == 6000:
            scal = _witness_like(O, scal, 6)
        ds = ctx.upload(scal)
        wxy, winf = O.to_affine("g2", O.msm("g2", dxy.numpy(), scal, dinf.numpy(), threads=8))
        got = ctx.msm(K.KG_G2, dxy.ptr, dinf.ptr, ds.ptr, n)
        assert not winf and (got[:16] == wxy).all(), n
        ctx.set_msm_small(0)
        try:
            assert (ctx.msm(K.KG_G2, dxy.ptr, dinf.ptr, ds.ptr, n) == got).all(), n
        finally:
            ctx.set_msm_small(SMALL_DEFAULT)


def test_both_forms_over_each_others_lengths(oracle):
    """KG_SMALL_KT_FROM moves the length from which the scalars are converted once: at 8192 the workgroups convert them themselves up to
    8192 pairs (the form the short lengths run, with its 13-bit index field full), at 1 the word planes serve every length from 2 pairs"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import kogarashi_amd as K
from oracle import oracle as O
from test_gpu_parity import aff, edge_mix, gpu_aff
ctx = K.Context(0)
for cv, curve, sfd, lens in (("g1", 0, 0, LENS), ("gk", 1, 1, LENS[1::2])):
    for n in lens:
        bases, scal, inf = edge_mix(O, cv, curve, sfd, n, 5150 + n)
        want = aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
        assert gpu_aff(ctx.msm_host(curve, bases, inf, scal, n), 4) == want, (cv, n)
        for c, r in SHAPES:
            ctx.set_msm_small(32768, c, r)
            assert gpu_aff(ctx.msm_host(curve, bases, inf, scal, n), 4) == want, (cv, n, c, r)
        ctx.set_msm_small(32768, 0, -1)
print("ok")
""" % (root, os.path.join(root, "tests"))
    for kt_from, lens, shapes in (("8192", "[2049, 3000, 4097, 6145, 8191, 8192]", "[(5, 1), (8, 3), (3, 0)]"),
                                  ("1", "[2, 3, 33, 257, 1024, 2048]", "[(2, 1), (5, 2), (8, 3)]")):
        r = subprocess.run([sys.executable, "-c", script.replace("LENS", lens).replace("SHAPES", shapes)], env=dict(os.environ, KG_SMALL_KT_FROM=kt_from),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (kt_from, r.stdout[-500:], r.stderr[-2000:])


def test_skewed_and_degenerate_inputs(ctx, oracle):
    """one base and one scalar repeated (every entry of a window in ONE bucket: the task cutting and the merge tree carry it; every
    addition after the first meets an equal point: the doubling branch), alternating P / -P (the inverse branch: identity), all-zero
    scalars, all-identity bases, scalars 1 and -1 only, and the empty input"""
    O = oracle
    one, ident = O.f_consts(0)["r"], (lambda out: gpu_aff(out, 4) is None)
    for n in (48, 600, 2500, 4096):
        P = O.gen_bases(0, SEED + 950, 0, 1)[0]
        bases = np.tile(P, (n, 1))
        k = O.gen_scalars(0, SEED + 951, 0, 1)[0]
        scal = np.tile(k, (n, 1))
        assert gpu_aff(ctx.msm_host(0, bases, None, scal, n), 4) == aff(O, "g1", O.msm("g1", bases, scal, None, threads=8)), n
        scal1 = np.tile(one, (n, 1))
        assert gpu_aff(ctx.msm_host(0, bases, None, scal1, n), 4) == aff(O, "g1", O.msm("g1", bases, scal1, None, threads=8)), n
        negP = P.copy(); negP[4:] = O.f_neg(1, P[4:])
        bases[1::2] = negP
        assert ident(ctx.msm_host(0, bases, None, scal, n)), n
        rb = O.gen_bases(0, SEED + 952, 0, n)
        pm = np.tile(one, (n, 1)); pm[::3] = O.f_neg(0, one)
        assert gpu_aff(ctx.msm_host(0, rb, None, pm, n), 4) == aff(O, "g1", O.msm("g1", rb, pm, None, threads=8)), n
        assert ident(ctx.msm_host(0, rb, None, np.zeros_like(pm), n))
        assert ident(ctx.msm_host(0, rb, np.ones(n, dtype=np.uint8), pm, n))
    out = ctx.msm_host(0, rb[:0], None, pm[:0], 0)
    assert not out[:4].any() and (out[4:8] == O.f_consts(1)["r"]).all() and not out[8:].any()        # (0, 1, 0): group.rs:106-110


def test_calls_in_flight_and_commitments(ctx, oracle):
    """kg_msm_begin / kg_msm_end over four tickets of different short lengths, kg_commit and kg_commit_host_scalars: the blocking call's points"""
    O = oracle
    n = 3000
    bases, scal, inf = edge_mix(O, "g1", 0, 0, n, SEED + 960)
    db, ds, di = ctx.upload(bases), ctx.upload(scal), ctx.upload(inf)
    lens = [17, 900, 3000, 64, 2049, 1, 1024, 300]
    want = [ctx.msm(0, db.ptr, di.ptr, ds.ptr, m) for m in lens]
    for i, m in enumerate(lens):
        assert gpu_aff(want[i], 4) == aff(O, "g1", O.msm("g1", bases[:m], scal[:m], inf[:m], threads=8)), m
    for rep in range(2):
        for i, m in enumerate(lens):
            ctx.msm_begin(0, db.ptr, di.ptr, ds.ptr, m, i % 4)
            if i >= 3:
                assert (ctx.msm_end(0, (i - 3) % 4) == want[i - 3]).all()
        for i in range(len(lens) - 3, len(lens)):
            assert (ctx.msm_end(0, i % 4) == want[i]).all()
    xy, oi = ctx.commit(0, db.ptr, di.ptr, ds.ptr, 1000)
    w = ctx.msm(0, db.ptr, di.ptr, ds.ptr, 1000)
    assert not oi and (xy == w[:8]).all()
    ctx.bases_register(0, db.ptr, di.ptr, n)
    try:
        xy2, oi2 = ctx.commit_host_scalars(0, db.ptr, di.ptr, scal, 1000)
        assert not oi2 and (xy2 == xy).all()
        assert (ctx.msm(0, db.ptr, di.ptr, ds.ptr, 1000) == w).all()
    finally:
        ctx.bases_unregister(db.ptr)


def test_bad_shapes_are_status_codes(ctx):
    L, h = ctx._lib, ctx._h
    for args in ((40000, 0, -1), (-1, 0, -1), (4096, 1, -1), (4096, 11, -1), (4096, 0, 8), (4096, 0, -2)):
        assert L.kg_msm_set_small(h, *args) == -2, args
    assert L.kg_msm_set_small(None, 4096, 0, -1) == -2
    assert L.kg_msm_set_small(h, -2, 0, -1) == 0 and L.kg_msm_set_small(h, SMALL_DEFAULT, 0, -1) == 0


def test_randomised_lengths_shapes_and_patterns_against_the_long_pipeline():
    """tools/dbg/stress_small.py: 300 random (curve, length 1 .. 32768, window of the stored bases, scalar pattern, identity flags) cases,
    each through the automatic shape and two random (c, r) shapes, blocking and three in flight, against the long pipeline's point -- with
    the scalars halved by the endomorphism where the plan says so, never, and wherever possible"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # KG_SMALL_GLV: 1 = the default (every scalar as two 127-bit halves k1 + k2 lambda where it pays: msm_digits.h), 0 = never (the form
    # calls in flight and the proofs' MSMs beyond 256 pairs keep), 2 = wherever the entries' index field allows (up to 16384 pairs)
    for glv, seed in (("1", "606"), ("0", "607"), ("2", "608")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "dbg", "stress_small.py"), "300", seed], env=dict(os.environ, KG_SMALL_GLV=glv),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "mismatches: 0" in r.stdout, (glv, r.stdout[-1500:], r.stderr[-1500:])
