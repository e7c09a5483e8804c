"""Every transform length the ABI accepts beyond the sizes tests/test_gpu_parity.py and test_gpu_large.py compare with
the oracle: the odd lengths 2^19, 2^21, 2^23 and 2^25 against the oracle's Fft<Fr> (fft.rs:92-127), 2^26 against the
(oracle-checked) 2^25 path through the decimation identity, 2^27 and 2^28 (S = 28, bn254/src/fr.rs:53) through impulse
responses, round trips and linearity evaluated on the device.  Three-step plans (ntt_tile.h) start at 2^23."""
import numpy as np
import pytest

from helpers import L

pytestmark = pytest.mark.gpu
SEED = 0x4B6F676172617368


@pytest.fixture(scope="module")
def ctx():
    import kogarashi_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _tensor(n_words):
    import torch
    return torch.empty(n_words, dtype=torch.int64, device="cuda:0")


def _mont(pyoracle, x):
    return L(pyoracle.to_mont(x % pyoracle.R_MOD, pyoracle.R_MOD))


@pytest.mark.parametrize("k", [19, 21, 23])
def test_odd_sizes_match_oracle(ctx, oracle, k):
    import kogarashi_amd as K
    n = 1 << k
    d, work = ctx.empty((n, 4)), ctx.empty((n, 4))
    ctx.gen_scalars(K.KG_FR, SEED + 200 + k, 0, n, d.ptr)
    v = d.numpy()
    fo = oracle.Fft(k)
    for name, inv, coset in (("dft", False, False), ("idft", True, False), ("coset_dft", False, True), ("coset_idft", True, True)):
        ctx.copy_d2d(work.ptr, d.ptr, n * 32)
        ctx.ntt(work.ptr, k, inv, coset)
        assert (work.numpy() == getattr(fo, name)(v, threads=16)).all(), (k, name)


def test_2_25_matches_oracle(ctx, oracle):
    import kogarashi_amd as K
    k = 25
    n = 1 << k
    d = ctx.empty((n, 4))
    ctx.gen_scalars(K.KG_FR, SEED + 225, 0, n, d.ptr)
    v = d.numpy()
    ctx.ntt(d.ptr, k, False, False)
    got = d.numpy()
    want = oracle.Fft(k).dft(v, threads=16)
    assert (got == want).all()
    del want
    ctx.ntt(d.ptr, k, True, False)
    assert (d.numpy() == v).all()


def test_2_26_by_decimation_against_2_25(ctx, pyoracle):
    """X[i] = E[i] + w^i O[i], X[i + n/2] = E[i] - w^i O[i] with E, O the 2^25-point transforms of the even / odd samples
    (the first butterfly level of fft.rs:195-218 written out), w = ROOT_OF_UNITY^(2^(28-26))."""
    import torch
    import kogarashi_amd as K
    k = 26
    n, h = 1 << k, 1 << (k - 1)
    x = _tensor(4 * n)
    ctx.gen_scalars(K.KG_FR, SEED + 226, 0, n, x.data_ptr())
    ctx.sync()
    xv = x.view(n, 4)
    e, o = xv[0::2].contiguous(), xv[1::2].contiguous()
    torch.cuda.synchronize()
    ctx.ntt(x.data_ptr(), k, False, False)
    ctx.ntt(e.data_ptr(), k - 1, False, False)
    ctx.ntt(o.data_ptr(), k - 1, False, False)
    pw = _tensor(4 * h)
    ctx.field_powers(K.KG_FR, _mont(pyoracle, 1), _mont(pyoracle, pyoracle.fr_omega(k)), pw.data_ptr(), h)
    ctx.field_vec_op(K.KG_FR, "mul", o.data_ptr(), pw.data_ptr(), o.data_ptr(), h)
    ctx.field_vec_op(K.KG_FR, "add", e.data_ptr(), o.data_ptr(), pw.data_ptr(), h)      # pw <- E + w^i O
    ctx.field_vec_op(K.KG_FR, "sub", e.data_ptr(), o.data_ptr(), e.data_ptr(), h)       # e  <- E - w^i O
    ctx.sync()
    assert torch.equal(x[: 4 * h], pw) and torch.equal(x[4 * h:], e.view(-1))


@pytest.mark.parametrize("k", [24, 27, 28])
def test_huge_sizes_by_device_side_properties(ctx, pyoracle, k):
    """impulse responses against kg_field_powers (dft of a delta_j0 + b delta_j1 is a w^(i j0) + b w^(i j1), every output
    index), both round trips and linearity on random vectors; all comparisons on the device."""
    import torch
    import kogarashi_amd as K
    P = pyoracle
    n = 1 << k
    w = P.fr_omega(k)
    x, y, t = _tensor(4 * n), _tensor(4 * n), _tensor(4 * n)
    j0, j1 = 1, (n // 3) | 1
    a, b = 0x1234567 + k, P.R_MOD - 99
    x.zero_()
    torch.cuda.synchronize()
    ctx.write(x.data_ptr() + 32 * j0, _mont(P, a).reshape(1, 4))
    ctx.write(x.data_ptr() + 32 * j1, _mont(P, b).reshape(1, 4))
    ctx.ntt(x.data_ptr(), k, False, False)
    ctx.field_powers(K.KG_FR, _mont(P, a), _mont(P, pow(w, j0, P.R_MOD)), y.data_ptr(), n)
    ctx.field_powers(K.KG_FR, _mont(P, b), _mont(P, pow(w, j1, P.R_MOD)), t.data_ptr(), n)
    ctx.field_vec_op(K.KG_FR, "add", y.data_ptr(), t.data_ptr(), y.data_ptr(), n)
    ctx.sync()
    assert torch.equal(x, y), "impulse response"
    # round trips (fft_transformation_test, fft.rs:246-257) and linearity on random data
    ctx.gen_scalars(K.KG_FR, SEED + 300 + k, 0, n, x.data_ptr())
    ctx.gen_scalars(K.KG_FR, SEED + 400 + k, 0, n, y.data_ptr())
    ctx.sync()
    t.copy_(x)
    torch.cuda.synchronize()
    for coset in (False, True):
        ctx.ntt(t.data_ptr(), k, False, coset)
        ctx.ntt(t.data_ptr(), k, True, coset)
        ctx.sync()
        assert torch.equal(t, x), ("round trip", coset)
    ctx.field_vec_op(K.KG_FR, "add", x.data_ptr(), y.data_ptr(), t.data_ptr(), n)
    for v in (x, y, t):
        ctx.ntt(v.data_ptr(), k, False, False)
    ctx.field_vec_op(K.KG_FR, "add", x.data_ptr(), y.data_ptr(), x.data_ptr(), n)
    ctx.sync()
    assert torch.equal(x, t), "linearity"


_PLAN_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import kogarashi_amd as K
from oracle import oracle as O
ctx = K.Context(0)
for k in [int(a) for a in os.environ.get("KG_TEST_NTT_SIZES", "13,18,20").split(",")]:
    v = O.gen_scalars(0, 0x4B6F676172617368 + 900 + k, 0, 1 << k)
    fo, fg = O.Fft(k), K.Fft(k, ctx=ctx)
    assert (fg.dft(v) == fo.dft(v, threads=8)).all(), ("dft", k)
    assert (fg.coset_idft(v) == fo.coset_idft(v, threads=8)).all(), ("coset_idft", k)
    assert (fg.idft(fg.coset_dft(v)) == fo.idft(fo.coset_dft(v, threads=8), threads=8)).all(), ("mixed", k)
print("ok")
"""


@pytest.mark.parametrize("env", [{"KG_NTT_STEPS": "3"}, {"KG_NTT_STEPS": "3", "KG_NTT_TILE": "10"}, {"KG_NTT_TILE": "11"},
                                 {"KG_NTT_DIRECT_MAX_LOG": "0"}, {"KG_NTT_DIRECT_MAX_LOG": "0", "KG_NTT_STEPS": "3"},
                                 {"KG_NTT_STEPS": "2", "KG_TEST_NTT_SIZES": "22"}, {"KG_NTT_TILE": "11", "KG_TEST_NTT_SIZES": "22"},
                                 {"KG_NTT_DIRECT_MAX_LOG": "0", "KG_TEST_NTT_SIZES": "22"}])
def test_every_plan_knob_gives_the_same_transform(env):
    """the plan knobs are read once per process: three-step plans, other tile sizes, the table-free inter-step twiddles
    (two-level composition instead of the direct table) and the two-step form of 2^22 (4096-element tiles; the automatic plan
    there has three steps) against the oracle, each in a child process"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _PLAN_SCRIPT % root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (env, r.stderr[-2000:])


@pytest.mark.parametrize("k", [11, 12, 17, 20])
def test_extreme_values(ctx, oracle, k):
    """inputs that push the lazy sums hardest: every element p - 1, alternating 0 / p - 1, a single non-zero element --
    one-step, 1024-, and 2048-element-tile plans, all four transforms"""
    import kogarashi_amd as K
    O = oracle
    n = 1 << k
    c = O.f_consts(0)
    pm1 = c["p"].copy()
    pm1[0] -= 1
    top = O.f_to_mont(0, pm1)                              # p - 1 in the ABI's Montgomery form
    fo, fg = O.Fft(k), K.Fft(k, ctx=ctx)
    cases = [np.tile(top, (n, 1)), np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)]
    cases[1][0::2] = top
    cases[2][n - 1] = top
    for v in cases:
        for name in ("dft", "idft", "coset_dft", "coset_idft"):
            assert (getattr(fg, name)(v) == getattr(fo, name)(v, threads=8)).all(), (k, name)
