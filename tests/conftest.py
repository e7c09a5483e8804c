import os
import subprocess
import sys

import pytest

# what kg_init() would set: the GPU tests run the configuration the bench measures (read when HIP initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def pyoracle():
    from oracle import pyoracle as P
    return P


@pytest.fixture(scope="session")
def hostlib():
    """The device arithmetic templates compiled for the host (tests/host/hosttest.cpp)."""
    import ctypes
    d = os.path.join(ROOT, "tests", "host")
    so = os.path.join(d, "libhosttest.so")
    srcs = [os.path.join(d, "hosttest.cpp")] + [
        os.path.join(ROOT, "kogarashi_amd", "csrc", f) for f in ("fp29.h", "fp29_checked.h", "curve.h", "fp_consts.h", "fp_inv.h", "ntt_core.h", "vecops.h", "msm_digits.h", "glv_consts.h", "coop_add.h", "host_fp.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, srcs[0]])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def hostlib_pm():
    """Same templates with Fq2 products routed through mul2pm, the signed double product a lane pair shares on the
    device (csrc/fp2s.h): the bound checker then sees it with the operands of every G2 formula, in both sign modes."""
    import ctypes
    d = os.path.join(ROOT, "tests", "host")
    so = os.path.join(d, "libhosttest_pm.so")
    srcs = [os.path.join(d, "hosttest.cpp")] + [
        os.path.join(ROOT, "kogarashi_amd", "csrc", f) for f in ("fp29.h", "fp29_checked.h", "curve.h", "fp_consts.h", "fp_inv.h", "ntt_core.h", "vecops.h", "msm_digits.h", "glv_consts.h", "coop_add.h", "host_fp.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DKG_FP2_MUL_VIA_PM", "-o", so, srcs[0]])
    return ctypes.CDLL(so)
