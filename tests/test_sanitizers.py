"""AddressSanitizer + UBSan builds of the CPU-side code (GPU sanitizers are not available on the pool): the oracle's C
restatement and the host build of the device templates (field / curve formulas, the NTT kernel bodies) run a small
workload each in a child process with the sanitizer runtime preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.slow


def _asan_env():
    lib = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(lib) or not os.path.exists(lib):
        pytest.skip("libasan.so not found")
    return dict(os.environ, LD_PRELOAD=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", KG_ORACLE_SO="liboracle_asan.so")


def _run(code, env):
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    out = _run("""
import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle as O
O._LIB = C.CDLL('oracle/liboracle_asan.so'); O._LIB.kgo_fft_new.restype = C.c_void_p
S = 0x4B6F676172617368
for cv, cid, fd in (('g1', 0, 0), ('gk', 1, 1)):
    b, s = O.gen_bases(cid, S, 0, 300, threads=2), O.gen_scalars(fd, S + 1, 0, 300)
    inf = np.zeros(300, dtype=np.uint8); inf[5] = 1; s[7] = 0
    a = O.to_affine(cv, O.msm(cv, b, s, inf, threads=3))
    n = O.commit_naive(cv, b, s, inf)
    assert a[1] == n[1] and (a[0] == n[0]).all()
v = O.gen_scalars(0, S + 2, 0, 1 << 9)
f = O.Fft(9)
assert (f.idft(f.dft(v, threads=4), threads=4) == v).all() and (f.coset_idft(f.coset_dft(v)) == v).all()
cs = O.chain_r1cs(64, O.gen_scalars(0, S + 3, 0, 1)[0])
print('ok')
""", _asan_env())
    assert "ok" in out


def test_device_templates_on_the_host_under_asan_ubsan(tmp_path):
    host = os.path.join(ROOT, "tests", "host")
    flags = ["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    builds = [subprocess.Popen(flags + ["-o", str(tmp_path / "ht.so"), os.path.join(host, "hosttest.cpp")]),
              subprocess.Popen(flags + ["-DKG_NTT_HOST_FEW", "-o", str(tmp_path / "htn.so"), os.path.join(host, "hosttest_ntt.cpp")])]
    assert all(b.wait() == 0 for b in builds)
    out = _run(f"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle as O
S = 0x4B6F676172617368
ht, htn = C.CDLL({str(tmp_path / 'ht.so')!r}), C.CDLL({str(tmp_path / 'htn.so')!r})
p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
a, b = O.gen_scalars(0, S, 0, 64), O.gen_scalars(0, S + 1, 0, 64)
o = np.zeros_like(a)
for chk in (0, 1):
    for op in range(9):
        ht.ht_field_ops(0, chk, op, p32(a.view(np.uint32)), p32(b.view(np.uint32)), p32(o.view(np.uint32)), C.c_size_t(64))
pts = O.gen_bases(0, S + 2, 0, 40, threads=2)
xy = np.zeros(8, dtype=np.uint64)
for mode in (0, 1, 8):
    ht.ht_curve_sum(0, 1, mode, p32(pts.view(np.uint32)), None, C.c_size_t(40), p32(xy.view(np.uint32)))
for k, steps in ((5, 0), (11, 0), (12, 0), (13, 0)):
    v = O.gen_scalars(0, S + k, 0, 1 << k)
    d = np.ascontiguousarray(v.copy())
    assert htn.ht_ntt(int(k < 13), k, steps, 0, 1, d.ctypes.data_as(C.c_void_p), None, None) == 0
    assert (d == O.Fft(k).coset_dft(v, threads=4)).all()
print('ok')
""", dict(_asan_env(), KG_ORACLE_SO=""))
    assert "ok" in out


def test_worker_pool_under_tsan_and_asan(tmp_path):
    """kogarashi_amd/csrc/worker_pool.h on its own (tests/host/pool_test.cpp): sequential tasks reuse ONE thread, tasks that wait for tasks
    of the same pool, no growth with the number of calls, a refused thread start as std::system_error, WaitAll on unwind -- under
    ThreadSanitizer and under AddressSanitizer + UBSan."""
    src = os.path.join(ROOT, "tests", "host", "pool_test.cpp")
    for name, san in (("tsan", "-fsanitize=thread"), ("asan", "-fsanitize=address,undefined")):
        exe = str(tmp_path / f"pool_{name}")
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", san, "-pthread", src, "-o", exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "pool_test ok" in r.stdout, (name, r.stdout[-500:], r.stderr[-3000:])
