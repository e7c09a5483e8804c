"""Window tables for registered bases (kg_bases_precompute): an MSM over an array that carries them sorts the digits of all
windows into one set of buckets ("merged" sort) and gathers 2^(c*w) * P_i from the table.  Results must equal the plain
call bit for bit and the oracle as affine points -- on uniform scalars, on witness-like skew (one huge bucket: several
partial-sum rounds), with identity bases, zero scalars and repeated points, for every curve."""
import numpy as np
import pytest

from test_gpu_parity import SEED, aff, gpu_aff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kogarashi_amd as K
    c = K.Context(0)
    yield c
    c.close()


def make_bases(ctx, O, curve, n, seed):
    if curve == 2:                                  # G2 bases: k_i * G2 made on the device (== the oracle's: test_fixed_base_mul)
        dk = ctx.upload(O.gen_scalars(0, seed, 0, n))
        dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
        ctx.fixed_base_mul(2, dk.ptr, n, dxy.ptr, dinf.ptr)
        return dxy.numpy()
    return O.gen_bases(curve, seed, 0, n)


def skewed(O, sfd, scal, n):
    rng = np.random.default_rng(n)
    kind = rng.integers(0, 10, n)
    one = O.f_consts(sfd)["r"]
    scal[kind < 5] = one
    scal[(kind >= 5) & (kind < 7)] = 0
    scal[kind == 7] = O.f_neg(sfd, one)
    scal[kind == 8] = O.f_to_mont(sfd, np.array([5, 0, 0, 0], dtype=np.uint64))


@pytest.mark.parametrize("cv,curve,sfd,n,skew", [
    ("g1", 0, 0, 1 << 16, False), ("g1", 0, 0, 70001, True), ("g1", 0, 0, (1 << 17) + 3, False), ("g1", 0, 0, 1 << 18, True),
    ("gk", 1, 1, 90000, False), ("gk", 1, 1, 1 << 17, True), ("g2", 2, 0, 66000, False), ("g2", 2, 0, 70001, True),
])
def test_table_msm_matches_plain_call_and_oracle(ctx, oracle, cv, curve, sfd, n, skew):
    O = oracle
    nb = 8 if curve == 2 else 4
    bases = make_bases(ctx, O, curve, n, SEED + 900 + n)
    scal = O.gen_scalars(sfd, SEED + 901 + n, 0, n)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[3, n - 1]] = 1
    scal[7] = 0
    bases[11] = bases[10]
    if skew:
        skewed(O, sfd, scal, n)
        bases[1000:1500] = bases[1000]
    db, di, ds = ctx.upload(bases), ctx.upload(inf), ctx.upload(scal)
    plain = ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)
    ctx.bases_register(curve, db.ptr, di.ptr, n)
    try:
        registered = ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)
        ctx.bases_precompute(db.ptr)
        ctx.bases_precompute(db.ptr, n)               # idempotent
        tabled = ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)
        again = ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)
        # a sub-range of the array is served from the plain resident copy
        stride = bases.shape[1] * 8
        sub = ctx.msm(curve, db.ptr + 100 * stride, di.ptr + 100, ds.ptr, n - 100)
    finally:
        ctx.bases_unregister(db.ptr)
    assert (registered == plain).all()
    assert (tabled == plain).all() and (again == plain).all()
    assert gpu_aff(tabled, nb) == aff(O, cv, O.msm(cv, bases, scal, inf, threads=8))
    assert (sub == ctx.msm(curve, db.ptr + 100 * stride, di.ptr + 100, ds.ptr, n - 100)).all()


def test_table_msm_degenerate_scalars(ctx, oracle):
    """all-zero scalars (no entries at all); one scalar everywhere (every window's entries in one bucket of the shared set)"""
    O, n = oracle, (1 << 16) + 5
    bases = O.gen_bases(0, SEED + 950, 0, n)
    db = ctx.upload(bases)
    ctx.bases_register(0, db.ptr, 0, n)
    ctx.bases_precompute(db.ptr)
    try:
        dz = ctx.upload(np.zeros((n, 4), dtype=np.uint64))
        assert gpu_aff(ctx.msm(0, db.ptr, 0, dz.ptr, n), 4) is None
        k = O.gen_scalars(0, SEED + 951, 0, 1)[0]
        scal = np.tile(k, (n, 1))
        ds = ctx.upload(scal)
        assert gpu_aff(ctx.msm(0, db.ptr, 0, ds.ptr, n), 4) == aff(O, "g1", O.msm("g1", bases, scal, None, threads=8))
    finally:
        ctx.bases_unregister(db.ptr)


def test_table_pipeline_and_commit(oracle):
    """kg_msm_begin / _end four deep with complete inputs, and kg_commit, over an array with tables"""
    import kogarashi_amd as K
    O, n = oracle, 1 << 17
    ctx = K.Context(0)
    try:
        bases = O.gen_bases(0, SEED + 960, 0, n)
        db = ctx.upload(bases)
        scal = [O.gen_scalars(0, SEED + 961 + i, 0, n) for i in range(6)]
        ds = [ctx.upload(s) for s in scal]
        want = [ctx.msm(0, db.ptr, 0, d.ptr, n) for d in ds]
        assert gpu_aff(want[0], 4) == aff(O, "g1", O.msm("g1", bases, scal[0], None, threads=8))
        ctx.bases_register(0, db.ptr, 0, n)
        ctx.bases_precompute(db.ptr)
        ctx.sync()
        ctx.set_inputs_complete(True)
        got = [None] * 6
        for i in range(6 + 4):
            if i >= 4:
                got[i - 4] = ctx.msm_end(0, (i - 4) % 4)
            if i < 6:
                ctx.msm_begin(0, db.ptr, 0, ds[i].ptr, n, i % 4)
        ctx.set_inputs_complete(False)
        for g, w in zip(got, want):
            assert (g == w).all()
        xy, inf = ctx.commit(0, db.ptr, 0, ds[1].ptr, n)
        assert not inf and xy.tobytes() == gpu_aff(want[1], 4)
    finally:
        ctx.close()


def test_precompute_argument_checks(ctx, oracle):
    import kogarashi_amd as K
    O = oracle
    n = 5000
    db = ctx.upload(O.gen_bases(0, SEED + 970, 0, n))
    with pytest.raises(K.KogarashiError):
        ctx.bases_precompute(db.ptr)                  # not registered
    ctx.bases_register(0, db.ptr, 0, n)
    try:
        with pytest.raises(K.KogarashiError):
            ctx.bases_precompute(db.ptr)              # 5000-scalar MSMs: below the offered range
        with pytest.raises(K.KogarashiError):
            ctx.bases_precompute(db.ptr, 100)         # shorter than the array
        with pytest.raises(K.KogarashiError):
            ctx.bases_precompute(db.ptr, (1 << 20) + 1)
        ctx.bases_precompute(db.ptr, 1 << 16)         # a short array that meets a long scalar vector (the prover's l): allowed
        ds = ctx.upload(O.gen_scalars(0, SEED + 971, 0, n))
        got = ctx.msm(0, db.ptr, 0, ds.ptr, n)        # its own 5000-pair MSM does not use the table
    finally:
        ctx.bases_unregister(db.ptr)
    assert (got == ctx.msm(0, db.ptr, 0, ds.ptr, n)).all()


_FMT_SCRIPT = r"""
import sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import kogarashi_amd as K
from oracle import oracle as O
SEED = 0x4B6F676172617368
ctx = K.Context(0)
out = {}
for cv, curve, sfd, n in (("g1", 0, 0, 70001), ("gk", 1, 1, 5000), ("g2", 2, 0, 66000)):
    if curve == 2:
        dk = ctx.upload(O.gen_scalars(0, SEED + 980, 0, n))
        dxy, dinf = ctx.empty((n, 16)), ctx.empty((n,), dtype=np.uint8)
        ctx.fixed_base_mul(2, dk.ptr, n, dxy.ptr, dinf.ptr)
        bases = dxy.numpy()
    else:
        bases = O.gen_bases(curve, SEED + 981 + curve, 0, n)
    scal = O.gen_scalars(sfd, SEED + 982 + curve, 0, n)
    inf = np.zeros(n, dtype=np.uint8); inf[[4, n - 2]] = 1
    db, di, ds = ctx.upload(bases), ctx.upload(inf), ctx.upload(scal)
    res = [ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n)]                 # per-call conversion
    ctx.bases_register(curve, db.ptr, di.ptr, n)
    res.append(ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n))             # resident copy
    if n >= 1 << 16:
        ctx.bases_precompute(db.ptr)
        res.append(ctx.msm(curve, db.ptr, di.ptr, ds.ptr, n))         # window table
    ctx.bases_unregister(db.ptr)
    out[cv] = [r.tolist() for r in res]
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("env", [{"KG_FMT64_MIN_LOG": "0", "KG_TABLE64": "1"}, {"KG_FMT64_MIN_LOG": "30", "KG_TABLE64": "0"}])
def test_resident_formats_agree(env):
    """The 64-byte resident form (arrays of >= 2^22 points, window tables) and the 72-byte one give the same sums: the same
    MSMs (per-call conversion, registered, with tables; identity flags set) in two processes that force one form each."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = []
    for e in (env, {}):
        p = subprocess.run([sys.executable, "-c", _FMT_SCRIPT, root], capture_output=True, text=True, env={**os.environ, **e}, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        runs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert runs[0] == runs[1]
    for cv, res in runs[0].items():
        assert all(r == res[0] for r in res), cv
