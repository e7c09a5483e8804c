"""The NTT kernel's per-thread bodies (csrc/ntt_tile.h) run on the host, thread by thread and pass by pass
(tests/host/hosttest_ntt.cpp): whole transforms through the device's own plans, step arguments and table formulas against
the oracle's restatement of Fft<Fr> (groth16/src/fft.rs:92-127), once more with the bound-tracking field type (no column,
limb or value bound of fp29.h can overflow in any pass), and the LDS bank conflicts of every pass counted."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x4B6F676172617368
VARIANTS = (("dft", 0, 0), ("idft", 1, 0), ("coset_dft", 0, 1), ("coset_idft", 1, 1))


@pytest.fixture(scope="module")
def nttlib():
    d = os.path.join(ROOT, "tests", "host")
    so = os.path.join(d, "libhosttest_ntt.so")
    srcs = [os.path.join(d, "hosttest_ntt.cpp")] + [
        os.path.join(ROOT, "kogarashi_amd", "csrc", f) for f in ("fp29.h", "fp29_checked.h", "fp_consts.h", "ntt_core.h", "ntt_tile.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, srcs[0]])
    return C.CDLL(so)


def _run(lib, oracle, k, steps, inv, coset, checked, seed=SEED):
    v = oracle.gen_scalars(0, seed + k, 0, 1 << k)
    d = np.ascontiguousarray(v.copy())
    cyc, ideal = C.c_long(), C.c_long()
    rc = lib.ht_ntt(int(checked), k, steps, inv, coset, d.ctypes.data_as(C.c_void_p), C.byref(cyc), C.byref(ideal))
    assert rc == 0, f"tile shape of 2^{k} (steps={steps}) is not instantiated"
    return v, d, cyc.value, ideal.value


@pytest.mark.parametrize("k", list(range(1, 15)))
def test_every_plan_matches_the_oracle_and_keeps_its_bounds(nttlib, oracle, k):
    f = oracle.Fft(k)
    for name, inv, coset in VARIANTS:
        v, got, _, _ = _run(nttlib, oracle, k, 0, inv, coset, checked=False)
        want = getattr(f, name)(v)
        assert (got == want).all(), (k, name)
        _, got_c, _, _ = _run(nttlib, oracle, k, 0, inv, coset, checked=True)       # aborts the process on a violated bound
        assert (got_c == want).all(), (k, name, "checked")


@pytest.mark.parametrize("k,steps", [(16, 0), (17, 0), (18, 0), (19, 0), (18, 3), (19, 3)])
def test_larger_plans(nttlib, oracle, k, steps):
    """two-step plans with 1024- and 2048-element tiles, and the three-step plans (column step in place, step B's table)"""
    f = oracle.Fft(k)
    for name, inv, coset in (VARIANTS if k <= 17 else VARIANTS[:1] + VARIANTS[3:]):
        v, got, cyc, ideal = _run(nttlib, oracle, k, steps, inv, coset, checked=(k <= 18))
        assert (got == getattr(f, name)(v, threads=8)).all(), (k, steps, name)
        assert cyc <= 2 * ideal, "more than two-way LDS bank conflicts on average in a tile of >= 1024 elements"


@pytest.mark.slow
@pytest.mark.parametrize("k,steps", [(20, 0), (21, 0), (22, 0), (22, 2), (23, 0)])
def test_big_tiles(nttlib, oracle, k, steps):
    """the 2048- and 4096-element tiles (2^10- and 2^11-point column and row steps), the automatic three-step plan of 2^22
    (8 + 8 + 6) beside its two-step form, and a plan above the direct tables"""
    f = oracle.Fft(k)
    v, got, cyc, ideal = _run(nttlib, oracle, k, steps, 0, 0, checked=True)
    assert (got == f.dft(v, threads=8)).all()
    assert cyc <= 2 * ideal


def test_plans_use_only_instantiated_shapes(nttlib):
    """every (DFT size, tile width) ntt_plan can ask for -- 2^1..2^28, automatic and three-step plans, every forced tile size
    (a forced size falls back to the automatic one where the shape does not exist) -- is a shape the kernel dispatch knows"""
    import re
    hdr = open(os.path.join(ROOT, "kogarashi_amd", "csrc", "ntt_tile.h")).read()
    both = set(re.findall(r"X\((\d+), (\d+)\)", re.search(r"#define KG_NTT_SHAPES\(X\)(.*)", hdr).group(1)))
    col = both | set(re.findall(r"X\((\d+), (\d+)\)", re.search(r"#define KG_NTT_SHAPES_COL_ONLY\(X\)(.*)", hdr).group(1)))
    row = both | set(re.findall(r"X\((\d+), (\d+)\)", re.search(r"#define KG_NTT_SHAPES_ROW_ONLY\(X\)(.*)", hdr).group(1)))
    out = (C.c_int * 9)()
    for steps in (0, 2, 3):
        for tile in (0, 10, 11, 12):
            for k in range(1, 29):
                c = nttlib.ht_ntt_plan(k, steps, tile, out)
                assert 1 <= c <= 3 and sum(out[3 * i] for i in range(c)) == k
                for i in range(c):
                    shape = (str(out[3 * i]), str(out[3 * i + 1]))
                    assert shape in (row if out[3 * i + 2] else col), (k, steps, tile, shape)
                    assert bool(out[3 * i + 2]) == (i == c - 1)


@pytest.mark.parametrize("k,steps,barriers", [(12, 0, 3), (18, 0, 3), (18, 3, 4), (20, 0, 5)])
def test_wave_private_passes_replace_workgroup_barriers(nttlib, oracle, k, steps, barriers):
    """the synchronisation plan of ntt_tile.h: passes whose exchange stays inside a wave's 256-element block run without a
    workgroup barrier between them.  The emulation aborts if such a pass touches an LDS word outside its wave's block (the
    discipline that makes the missing barrier safe) and counts what is left: e.g. 3 barriers per 2^18 transform instead of 8."""
    v, got, _, _ = _run(nttlib, oracle, k, steps, 0, 1, checked=False)
    assert (got == oracle.Fft(k).coset_dft(v, threads=8)).all()
    pp, bb = C.c_long(), C.c_long()
    nttlib.ht_ntt_sync_counts(C.byref(pp), C.byref(bb))
    assert bb.value == barriers and pp.value >= 3, (pp.value, bb.value)


@pytest.mark.parametrize("k,steps,tile", [(13, 0, 11), (16, 0, 11), (18, 3, 11), (20, 0, 10), (21, 0, 11)])
def test_forced_tile_sizes(nttlib, oracle, k, steps, tile):
    """KG_NTT_TILE plans: the 2048-element tiles of small factors, single-column tiles of 2^10- and 2^11-point factors"""
    v = oracle.gen_scalars(0, SEED + 50 + k, 0, 1 << k)
    d = np.ascontiguousarray(v.copy())
    assert nttlib.ht_ntt_tile(1 if k <= 18 else 0, k, steps, tile, 0, 1, d.ctypes.data_as(C.c_void_p), None, None) == 0
    assert (d == oracle.Fft(k).coset_dft(v, threads=8)).all()
