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
    assert so.kg_version() == 1
    assert so.kg_strerror(-1) == b"no HIP device"


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
