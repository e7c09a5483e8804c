"""The compiled-language host side: include/kogarashi_amd.hpp mirrors the reference's call sites (msm_curve_addition, Fft,
PedersenCommitment, Prover) in C++ over the C ABI -- the reference is Rust and no Rust toolchain exists in the image
(rust/ holds that shim as files).  tests/host/abi_cpp_test.cpp drives the header against the oracle's C restatement; this test
builds it with g++ (no HIP, no Python in the loop: the binary links libkogarashi_amd.so and liboracle.so only) and runs it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from oracle import oracle as O
    O.build()
    exe = str(tmp_path / "abi_cpp_test")
    lib_dir, ora_dir = os.path.join(ROOT, "kogarashi_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "abi_cpp_test.cpp"),
                           "-L" + lib_dir, "-lkogarashi_amd", "-L" + ora_dir, "-loracle", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + ora_dir])
    return exe


def test_cpp_mirror_compiles_against_the_abi(tmp_path):
    """no GPU needed: the header and the test program compile and link against the built library (every entry point the
    mirror uses exists with the signature it expects)"""
    if not os.path.exists(os.path.join(ROOT, "kogarashi_amd", "libkogarashi_amd.so")):
        pytest.skip("library not built")
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_mirror_matches_the_oracle(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
