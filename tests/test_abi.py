"""The C-ABI library loads without a GPU and exports every symbol include/kogarashi_amd.h declares; the product
refuses to run without a device (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kogarashi_amd import build, lib as L
    build.build()
    return L


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "kogarashi_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(kg_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    so = lib.load()
    missing = [s for s in declared if not hasattr(so, s)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == declared
    assert so.kg_version() == 6
    assert so.kg_strerror(-1) == b"no HIP device"


def test_ntt_plan_is_reported_without_a_device(lib):
    """kg_ntt_plan (what bench.py prices the transform's roofline with): the factors of every accepted size multiply to n, a
    tile holds at least one DFT and at most 4096 elements, and sizes outside 2^1..2^28 are refused."""
    for k in range(1, 29):
        plan = lib.ntt_plan(k)
        assert 1 <= len(plan) <= 3 and sum(m for m, _ in plan) == k, (k, plan)
        assert all(1 <= m <= 11 and m <= t <= 12 for m, t in plan), (k, plan)
        assert len(plan) == (1 if k <= 11 else 2 if k <= 21 else 3), (k, plan)
    assert lib.ntt_plan(22) == [(8, 10), (8, 10), (6, 10)]         # the bench's transform
    assert lib.ntt_plan(0) == [] and lib.ntt_plan(29) == []


def test_no_built_artefact_is_tracked():
    """the history stays source-only: no ELF object, library or microbenchmark binary among the tracked files"""
    import subprocess
    try:
        files = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("not a git checkout")
    elf = []
    for f in files:
        path = os.path.join(ROOT, f)
        if os.path.isfile(path):
            with open(path, "rb") as fh:
                if fh.read(4) == b"\x7fELF":
                    elf.append(f)
    assert not elf, elf


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(lib.KogarashiError):
        lib.Context(0)
    import kogarashi_amd as K
    with pytest.raises(lib.KogarashiError):
        K.msm_curve_addition([[0] * 8], [[0] * 4])


def test_product_does_not_import_the_oracle():
    """oracle/ is the checker only: nothing in the product package may import, include or link it."""
    pkg = os.path.join(ROOT, "kogarashi_amd")
    bad = re.compile(r"^\s*(import\s+oracle|from\s+oracle|from\s+\.\.?oracle|#\s*include\s*[<\"].*oracle)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(txt), f
                assert "liboracle" not in txt, f


def test_host_point_sums_without_a_device(lib):
    """kg_points_sum_affine is the host half of the sharded commit (kogarashi_amd/dist.py): it needs no GPU, so its
    group-law edge cases are checked here against the oracle for all three curves."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    so = lib.load()
    seed = 0x4B6F676172617368

    def host_sum(curve, pts, inf):
        pts = np.ascontiguousarray(pts, dtype=np.uint64)
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        xy = np.zeros(pts.shape[1], dtype=np.uint64)
        oi = C.c_uint8(0)
        assert so.kg_points_sum_affine(None, curve, pts.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p), C.c_size_t(len(inf)),
                                       xy.ctypes.data_as(C.c_void_p), C.byref(oi)) == 0
        return xy, int(oi.value)

    for cv, curve, sfd in (("g1", 0, 0), ("gk", 1, 1), ("g2", 2, 0)):
        k = O.gen_scalars(sfd, seed + 900 + curve, 0, 6)
        pts, _ = O.fixed_base_mul(curve, k)
        nb = pts.shape[1] // 2
        base_fd = 0 if cv == "gk" else 1
        neg0 = pts[0].copy()
        for j in range(nb // 4):
            neg0[nb + 4 * j: nb + 4 * j + 4] = O.f_neg(base_fd, pts[0, nb + 4 * j: nb + 4 * j + 4])
        cases = [(pts, np.zeros(6, dtype=np.uint8)), (np.stack([pts[0], pts[0], pts[0]]), np.zeros(3, dtype=np.uint8)),
                 (np.stack([pts[0], neg0]), np.zeros(2, dtype=np.uint8)), (pts, np.array([0, 1, 0, 1, 1, 0], dtype=np.uint8)),
                 (pts[:2], np.ones(2, dtype=np.uint8))]
        for arr, inf in cases:
            want = None
            for p, f in zip(arr, inf):
                if f:
                    continue
                if want is None:
                    one = np.concatenate([O.f_consts(base_fd)["r"], np.zeros(nb - 4, dtype=np.uint64)])
                    want = np.concatenate([p, one])
                else:
                    want = O.add_mixed(cv, p, 0, want)
            got_xy, got_inf = host_sum(curve, arr, inf)
            if want is None:
                assert got_inf == 1
            else:
                wxy, winf = O.to_affine(cv, want)
                assert got_inf == winf and (winf or (got_xy == wxy).all()), (cv, len(arr))


def test_knob_table_is_one_table_and_the_readme_quotes_it(lib):
    """every KG_* variable the library reads lives in kogarashi_amd/csrc/tuning.h (no getenv anywhere else), kg_tuning_describe walks it,
    and README.md's table is the generated one (tools/gen_knob_table.py)"""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "kogarashi_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".h", ".hip", ".cpp")) and f not in ("tuning.cpp",):
            txt = open(os.path.join(csrc, f)).read()
            for m in re.finditer(r'getenv\("([A-Z0-9_]+)"\)', txt):
                assert m.group(1) == "GPU_MAX_HW_QUEUES", (f, m.group(1))        # kg_init / kg_hw_queue_setting: the runtime's variable, not a knob
    from kogarashi_amd.lib import tuning_table
    rows = tuning_table()
    assert len(rows) >= 30 and len({r["env"] for r in rows}) == len(rows)
    assert all(r["env"].startswith("KG_") and r["doc"] for r in rows)
    if not any(k.startswith("KG_") and k not in ("KG_LIB_PATH", "KG_BENCH_SELFTEST", "KG_BENCH_PMC") for k in os.environ):
        assert all(r["value"] == r["default"] for r in rows)
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_knob_table.py"), "--check"]).returncode == 0, "run python tools/gen_knob_table.py"
