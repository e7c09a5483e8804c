"""ctypes binding of libkogarashi_amd.so (the C ABI in include/kogarashi_amd.h).

There is deliberately NO fallback: if the HIP library is missing or no gfx950 device is visible, every entry
point raises.  Device memory is handled through the ABI's own kg_malloc / kg_memcpy_* (or through torch
tensors' data_ptr() in bench.py); numpy arrays cross the boundary as host pointers."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libkogarashi_amd.so")
SO_PATH = os.environ.get("KG_LIB_PATH", SO_PATH)     # A/B experiments: an alternative build of the same ABI

KG_FR, KG_FQ = 0, 1
KG_G1, KG_GRUMPKIN, KG_G2 = 0, 1, 2
OPS = {"add": 0, "sub": 1, "mul": 2, "square": 3, "neg": 4, "double": 5, "invert": 6, "from_mont": 7, "to_mont": 8}
EXPORTS = [
    "kg_version", "kg_init", "kg_hw_queue_setting", "kg_device_count", "kg_strerror", "kg_ctx_create", "kg_ctx_destroy", "kg_last_error", "kg_ctx_set_stream",
    "kg_ctx_sync", "kg_malloc", "kg_free", "kg_memcpy_h2d", "kg_memcpy_d2h", "kg_memcpy_d2d", "kg_field_vec_op",
    "kg_field_vec_scale", "kg_ntt_bn254_fr", "kg_fr_divide_by_z_on_coset", "kg_msm", "kg_msm_host", "kg_commit",
    "kg_points_sum_affine", "kg_msm_set_window", "kg_msm_set_groups", "kg_ctx_queue_placement", "kg_gen_scalars", "kg_gen_bases", "kg_profile_enable", "kg_profile_last",
    "kg_fixed_base_mul", "kg_groth16_prove_bn254", "kg_r1cs_evaluate", "kg_field_vec_axpy", "kg_field_powers", "kg_msm_begin", "kg_msm_end", "kg_profile_summary", "kg_bases_register", "kg_bases_unregister", "kg_groth16_prove_begin", "kg_groth16_prove_end",
    "kg_msm_pick_window", "kg_shard_range", "kg_commit_sharded", "kg_msm_sharded", "kg_sharded_key_create", "kg_sharded_key_destroy",
    "kg_sharded_key_len", "kg_sharded_key_commit", "kg_r1cs_prod", "kg_nova_cross_term", "kg_ctx_set_inputs_complete", "kg_bases_precompute", "kg_msm_table_window", "kg_groth16_prove_r1cs_bn254", "kg_groth16_prove_r1cs_begin", "kg_groth16_prove_sharded", "kg_ntt_plan",
    "kg_msm_host_scalars", "kg_commit_host_scalars", "kg_tuning_describe", "kg_mem_info", "kg_groth16_setup_bn254", "kg_experiments_built", "kg_msm_host_slices",
    "kg_msm_set_small", "kg_ctx_worker_threads", "kg_ctx_queue_placement2", "kg_ctx_trim",
]


class Groth16Crs(C.Structure):
    """kg_groth16_crs of include/kogarashi_amd.h"""
    _fields_ = [("m", C.c_size_t), ("l", C.c_size_t), ("m_l_1", C.c_size_t),
                ("d_h", C.c_void_p), ("d_h_inf", C.c_void_p), ("d_l", C.c_void_p), ("d_l_inf", C.c_void_p),
                ("d_a", C.c_void_p), ("d_a_inf", C.c_void_p), ("d_b_g1", C.c_void_p), ("d_b_g1_inf", C.c_void_p),
                ("d_b_g2", C.c_void_p), ("d_b_g2_inf", C.c_void_p),
                ("alpha_g1", C.c_uint64 * 8), ("beta_g1", C.c_uint64 * 8), ("delta_g1", C.c_uint64 * 8),
                ("beta_g2", C.c_uint64 * 16), ("delta_g2", C.c_uint64 * 16),
                ("delta_g1_inf", C.c_uint8), ("delta_g2_inf", C.c_uint8)]


class Csr(C.Structure):
    """kg_csr of include/kogarashi_amd.h"""
    _fields_ = [("d_row_ptr", C.c_void_p), ("d_col", C.c_void_p), ("d_val", C.c_void_p)]


class KogarashiError(RuntimeError):
    pass


class ProverInversionFailed(KogarashiError):
    """groth16::Error::ProverInversionFailed (groth16/src/zksnark.rs:37-38): gamma or delta is zero"""


class ProverSubVersionCrsAttack(KogarashiError):
    """groth16::Error::ProverSubVersionCrsAttack (groth16/src/error.rs:2-8, prover.rs:67-69)"""


_lib = None


def load():
    """Loads the shared library; raises if it has not been built (python -m kogarashi_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise KogarashiError(f"{SO_PATH} is missing: build it with `python -m kogarashi_amd.build` (hipcc, gfx950)")
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64 and
        # loads them by path.  If this library came first it would bind /opt/rocm's copies, torch would then load a
        # second runtime and report "No HIP GPUs are available".  With torch imported first the soname is already
        # resolved and both share the runtime (torch tensors and kg_malloc memory live in one address space).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(SO_PATH)
        _lib.kg_strerror.restype = C.c_char_p
        _lib.kg_last_error.restype = C.c_char_p
        _lib.kg_last_error.argtypes = [C.c_void_p]
        _lib.kg_sharded_key_len.restype = C.c_size_t
        _lib.kg_sharded_key_len.argtypes = [C.c_void_p]
        _lib.kg_sharded_key_destroy.argtypes = [C.c_void_p]
        _lib.kg_sharded_key_destroy.restype = None
    return _lib


def init() -> bool:
    """kg_init: one hardware queue per library queue (GPU_MAX_HW_QUEUES=16 unless set).  Explicit and optional: call it
    before anything initialises HIP (importing torch does not; the first torch.cuda call or Context does).  The package
    never changes the environment by itself."""
    return bool(load().kg_init())


def hw_queue_setting() -> int:
    return int(load().kg_hw_queue_setting())


def msm_table_window(msm_len: int) -> int:
    """window width of the tables kg_bases_precompute builds for MSMs of msm_len scalars (0: none offered); no device needed"""
    return int(load().kg_msm_table_window(C.c_size_t(msm_len)))


def ntt_plan(log_n: int) -> list[tuple[int, int]]:
    """kg_ntt_plan: [(log2 DFT length, log2 tile elements)] per step of a 2^log_n transform (no device needed)"""
    m, t = (C.c_uint32 * 3)(), (C.c_uint32 * 3)()
    s = int(load().kg_ntt_plan(C.c_uint32(log_n), m, t))
    return [(int(m[i]), int(t[i])) for i in range(s)]


def tuning_table() -> list[dict]:
    """kg_tuning_describe: every environment knob of the library -- name, description, default, value in this process"""
    lib = load()
    rows = []
    for i in range(int(lib.kg_tuning_describe(-1, None, None, None, None))):
        env, doc, d, v = C.c_char_p(), C.c_char_p(), C.c_int(), C.c_int()
        if lib.kg_tuning_describe(i, C.byref(env), C.byref(doc), C.byref(d), C.byref(v)) != 0:
            raise KogarashiError("kg_tuning_describe")
        rows.append({"env": env.value.decode(), "doc": doc.value.decode(), "default": d.value, "value": v.value})
    return rows


def msm_host_slices(n: int, scalars_only: bool = True) -> list[int]:
    """kg_msm_host_slices: boundaries lo[0..K] of the index slices a host-array MSM of n pairs runs in (no device needed)"""
    lo = (C.c_size_t * 9)()
    k = int(load().kg_msm_host_slices(C.c_size_t(n), int(bool(scalars_only)), lo))
    if k < 0:
        raise KogarashiError("kg_msm_host_slices")
    return [int(lo[i]) for i in range(k + 1)] if k else [0]


def experiments_built() -> bool:
    """kg_experiments_built: does the loaded library carry the -DKG_EXPERIMENTS kernels?"""
    return bool(load().kg_experiments_built())


def msm_pick_window(n: int) -> int:
    """kg_msm_pick_window: the automatic window width for n pairs (no device needed)"""
    return int(load().kg_msm_pick_window(C.c_size_t(n)))


def _vp(x):
    return C.c_void_p(int(x) if x else 0)


class Context:
    """One kg_ctx: a GPU, a stream, cached twiddles and MSM work space."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.kg_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise KogarashiError(f"kg_ctx_create(device={device}) failed: {self._lib.kg_strerror(rc).decode()} "
                                 "(the HIP extension needs a visible MI355X; there is no CPU path)")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise KogarashiError(f"{what}: {self._lib.kg_strerror(rc).decode()} [{self._lib.kg_last_error(self._h).decode()}]")

    # ---- plumbing ------------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr: int):
        self._chk(self._lib.kg_ctx_set_stream(self._h, _vp(hip_stream_ptr)), "kg_ctx_set_stream")

    def set_inputs_complete(self, on: bool = True):
        self._chk(self._lib.kg_ctx_set_inputs_complete(self._h, int(bool(on))), "kg_ctx_set_inputs_complete")

    def sync(self):
        self._chk(self._lib.kg_ctx_sync(self._h), "kg_ctx_sync")

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self._lib.kg_malloc(self._h, C.c_size_t(nbytes), C.byref(p)), "kg_malloc")
        return p.value

    def trim(self):
        """kg_ctx_trim: give the device memory kg_free has kept back to the driver"""
        self._chk(self._lib.kg_ctx_trim(self._h), "kg_ctx_trim")

    def mem_info(self) -> tuple[int, int]:
        """(free, total) bytes of the context's device (kg_mem_info)"""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._chk(self._lib.kg_mem_info(self._h, C.byref(f), C.byref(t)), "kg_mem_info")
        return int(f.value), int(t.value)

    def free(self, dptr: int):
        self._chk(self._lib.kg_free(self._h, _vp(dptr)), "kg_free")

    def upload(self, arr: np.ndarray) -> "DeviceArray":
        arr = np.ascontiguousarray(arr)
        d = DeviceArray(self, arr.nbytes, arr.shape, arr.dtype)
        if arr.nbytes:
            self._chk(self._lib.kg_memcpy_h2d(self._h, _vp(d.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)), "kg_memcpy_h2d")
        return d

    def write(self, dptr: int, arr: np.ndarray):
        """host array -> device memory at dptr"""
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            self._chk(self._lib.kg_memcpy_h2d(self._h, _vp(dptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)), "kg_memcpy_h2d")

    def copy_d2d(self, dst: int, src: int, nbytes: int):
        """device -> device on the context's stream (asynchronous, ordered with the kernels)"""
        self._chk(self._lib.kg_memcpy_d2d(self._h, _vp(dst), _vp(src), C.c_size_t(nbytes)), "kg_memcpy_d2d")

    def empty(self, shape, dtype=np.uint64) -> "DeviceArray":
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return DeviceArray(self, nbytes, tuple(shape), np.dtype(dtype))

    def download(self, d: "DeviceArray") -> np.ndarray:
        out = np.empty(d.shape, dtype=d.dtype)
        if out.nbytes:
            self._chk(self._lib.kg_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), _vp(d.ptr), C.c_size_t(out.nbytes)), "kg_memcpy_d2h")
        return out

    # ---- raw entry points (device pointers as ints) ------------------------------------------------
    def field_vec_op(self, field: int, op: str, a: int, b: int, out: int, n: int):
        self._chk(self._lib.kg_field_vec_op(self._h, field, OPS[op], _vp(a), _vp(b), _vp(out), C.c_size_t(n)), "kg_field_vec_op")

    def field_vec_scale(self, field: int, a: int, s: np.ndarray, out: int, n: int):
        s = np.ascontiguousarray(s, dtype=np.uint64)
        self._chk(self._lib.kg_field_vec_scale(self._h, field, _vp(a), s.ctypes.data_as(C.c_void_p), _vp(out), C.c_size_t(n)), "kg_field_vec_scale")

    def field_powers(self, field: int, start: np.ndarray, base: np.ndarray, out: int, n: int):
        start = np.ascontiguousarray(start, dtype=np.uint64)
        base = np.ascontiguousarray(base, dtype=np.uint64)
        self._chk(self._lib.kg_field_powers(self._h, field, start.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p), _vp(out), C.c_size_t(n)), "kg_field_powers")

    def field_vec_axpy(self, field: int, a: int, s: np.ndarray, b: int, out: int, n: int):
        s = np.ascontiguousarray(s, dtype=np.uint64)
        self._chk(self._lib.kg_field_vec_axpy(self._h, field, _vp(a), s.ctypes.data_as(C.c_void_p), _vp(b), _vp(out), C.c_size_t(n)), "kg_field_vec_axpy")

    def ntt(self, data: int, log_n: int, inverse: bool, coset: bool):
        self._chk(self._lib.kg_ntt_bn254_fr(self._h, _vp(data), C.c_uint32(log_n), int(bool(inverse)), int(bool(coset))), "kg_ntt_bn254_fr")

    def divide_by_z_on_coset(self, data: int, log_n: int):
        self._chk(self._lib.kg_fr_divide_by_z_on_coset(self._h, _vp(data), C.c_uint32(log_n)), "kg_fr_divide_by_z_on_coset")

    def msm(self, curve: int, bases: int, inf: int, scalars: int, n: int) -> np.ndarray:
        out = np.zeros(24 if curve == KG_G2 else 12, dtype=np.uint64)
        self._chk(self._lib.kg_msm(self._h, curve, _vp(bases), _vp(inf), _vp(scalars), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)), "kg_msm")
        return out

    def bases_register(self, curve: int, bases: int, inf: int, n: int):
        self._chk(self._lib.kg_bases_register(self._h, curve, _vp(bases), _vp(inf), C.c_size_t(n)), "kg_bases_register")

    def bases_precompute(self, bases: int, msm_len: int = 0):
        """window tables for a registered array (kg_bases_precompute); msm_len = length of the MSMs it meets (0: its own)"""
        self._chk(self._lib.kg_bases_precompute(self._h, _vp(bases), C.c_size_t(msm_len)), "kg_bases_precompute")

    def bases_unregister(self, bases: int):
        self._chk(self._lib.kg_bases_unregister(self._h, _vp(bases)), "kg_bases_unregister")

    def msm_begin(self, curve: int, bases: int, inf: int, scalars: int, n: int, ticket: int):
        self._chk(self._lib.kg_msm_begin(self._h, curve, _vp(bases), _vp(inf), _vp(scalars), C.c_size_t(n), int(ticket)), "kg_msm_begin")

    def msm_end(self, curve: int, ticket: int) -> np.ndarray:
        out = np.zeros(24 if curve == KG_G2 else 12, dtype=np.uint64)
        self._chk(self._lib.kg_msm_end(self._h, curve, int(ticket), out.ctypes.data_as(C.c_void_p)), "kg_msm_end")
        return out

    def msm_host(self, curve: int, bases: np.ndarray, inf, scalars: np.ndarray, n: int) -> np.ndarray:
        out = np.zeros(24 if curve == KG_G2 else 12, dtype=np.uint64)
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        ip = None
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            ip = inf.ctypes.data_as(C.c_void_p)
        self._chk(self._lib.kg_msm_host(self._h, curve, bases.ctypes.data_as(C.c_void_p), ip, scalars.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(n), out.ctypes.data_as(C.c_void_p)), "kg_msm_host")
        return out

    def msm_host_scalars(self, curve: int, bases: int, inf: int, scalars: np.ndarray, n: int) -> np.ndarray:
        """kg_msm_host_scalars: device bases (registered or not), HOST scalars -- the per-call shape of the reference's call sites"""
        out = np.zeros(24 if curve == KG_G2 else 12, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        self._chk(self._lib.kg_msm_host_scalars(self._h, curve, _vp(bases), _vp(inf), scalars.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                                out.ctypes.data_as(C.c_void_p)), "kg_msm_host_scalars")
        return out

    def commit_host_scalars(self, curve: int, bases: int, inf: int, scalars: np.ndarray, n: int):
        xy = np.zeros(16 if curve == KG_G2 else 8, dtype=np.uint64)
        oi = C.c_uint8(0)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        self._chk(self._lib.kg_commit_host_scalars(self._h, curve, _vp(bases), _vp(inf), scalars.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                                   xy.ctypes.data_as(C.c_void_p), C.byref(oi)), "kg_commit_host_scalars")
        return xy, int(oi.value)

    def commit(self, curve: int, bases: int, inf: int, scalars: int, n: int):
        xy = np.zeros(16 if curve == KG_G2 else 8, dtype=np.uint64)
        oi = C.c_uint8(0)
        self._chk(self._lib.kg_commit(self._h, curve, _vp(bases), _vp(inf), _vp(scalars), C.c_size_t(n), xy.ctypes.data_as(C.c_void_p), C.byref(oi)), "kg_commit")
        return xy, int(oi.value)

    def points_sum_affine(self, curve: int, pts: np.ndarray, inf: np.ndarray):
        pts = np.ascontiguousarray(pts, dtype=np.uint64)
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        xy = np.zeros(16 if curve == KG_G2 else 8, dtype=np.uint64)
        oi = C.c_uint8(0)
        self._chk(self._lib.kg_points_sum_affine(self._h, curve, pts.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p),
                                                 C.c_size_t(len(inf)), xy.ctypes.data_as(C.c_void_p), C.byref(oi)), "kg_points_sum_affine")
        return xy, int(oi.value)

    def set_msm_window(self, c: int):
        self._chk(self._lib.kg_msm_set_window(self._h, int(c)), "kg_msm_set_window")

    def queue_placement(self) -> int:
        """how the service queues were dealt over the compute pipes (kg_ctx_queue_placement2): 0 probe off, 1 creation order (no clear picture),
        2 + j probed"""
        p = C.c_int(0)
        self._chk(self._lib.kg_ctx_queue_placement2(self._h, C.byref(p)), "kg_ctx_queue_placement2")
        return int(p.value)

    def set_msm_groups(self, groups: int):
        """window groups of a blocking MSM: 0 automatic, 1 none, 2..4 (kg_msm_set_groups)"""
        self._chk(self._lib.kg_msm_set_groups(self._h, int(groups)), "kg_msm_set_groups")

    def worker_threads(self) -> int:
        """host worker threads the context has started so far (kg_ctx_worker_threads)"""
        v = C.c_int(0)
        self._chk(self._lib.kg_ctx_worker_threads(self._h, C.byref(v)), "kg_ctx_worker_threads")
        return int(v.value)

    def set_msm_small(self, max_pairs: int = -2, c: int = 0, r: int = -1):
        """the short-input MSM (kg_msm_set_small): longest MSM it takes (-2: keep, 0: never), window width (0: by length), log2 of the
        buckets per workgroup (-1: by length)"""
        self._chk(self._lib.kg_msm_set_small(self._h, int(max_pairs), int(c), int(r)), "kg_msm_set_small")

    def gen_scalars(self, field: int, seed: int, start: int, n: int, out: int):
        self._chk(self._lib.kg_gen_scalars(self._h, field, C.c_uint64(seed), C.c_size_t(start), C.c_size_t(n), _vp(out)), "kg_gen_scalars")

    def gen_bases(self, curve: int, seed: int, start: int, n: int, out: int):
        self._chk(self._lib.kg_gen_bases(self._h, curve, C.c_uint64(seed), C.c_size_t(start), C.c_size_t(n), _vp(out)), "kg_gen_bases")

    def r1cs_evaluate(self, row_ptr: int, col: int, val: int, m: int, z: int, out: int):
        self._chk(self._lib.kg_r1cs_evaluate(self._h, _vp(row_ptr), _vp(col), _vp(val), C.c_size_t(m), _vp(z), _vp(out)), "kg_r1cs_evaluate")

    def r1cs_prod(self, field: int, row_ptr: int, col: int, val: int, m: int, z: int, out: int):
        self._chk(self._lib.kg_r1cs_prod(self._h, field, _vp(row_ptr), _vp(col), _vp(val), C.c_size_t(m), _vp(z), _vp(out)), "kg_r1cs_prod")

    def nova_cross_term(self, field: int, a, b, c, m: int, z1: int, z2: int, u1: np.ndarray, u2: np.ndarray, out: int):
        """kg_nova_cross_term; a, b, c: (row_ptr, col, val) device pointers"""
        mk = lambda t: Csr(_vp(t[0]), _vp(t[1]), _vp(t[2]))
        ca, cb, cc = mk(a), mk(b), mk(c)
        u1 = np.ascontiguousarray(u1, dtype=np.uint64)
        u2 = np.ascontiguousarray(u2, dtype=np.uint64)
        self._chk(self._lib.kg_nova_cross_term(self._h, field, C.byref(ca), C.byref(cb), C.byref(cc), C.c_size_t(m), _vp(z1), _vp(z2),
                                               u1.ctypes.data_as(C.c_void_p), u2.ctypes.data_as(C.c_void_p), _vp(out)), "kg_nova_cross_term")

    def fixed_base_mul(self, curve: int, k: int, n: int, out_xy: int, out_inf: int):
        self._chk(self._lib.kg_fixed_base_mul(self._h, curve, _vp(k), C.c_size_t(n), _vp(out_xy), _vp(out_inf)), "kg_fixed_base_mul")

    def groth16_prove(self, crs: "Groth16Crs", a_eval: int, b_eval: int, c_eval: int, x: int, w: int, r: np.ndarray, s: np.ndarray):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        s = np.ascontiguousarray(s, dtype=np.uint64)
        out = np.zeros(32, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        rc = self._lib.kg_groth16_prove_bn254(self._h, C.byref(crs), _vp(a_eval), _vp(b_eval), _vp(c_eval), _vp(x), _vp(w),
                                              r.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p))
        if rc == -6:
            raise ProverSubVersionCrsAttack("delta is the identity")
        self._chk(rc, "kg_groth16_prove_bn254")
        return out[:8].copy(), out[8:24].copy(), out[24:].copy(), inf

    def groth16_prove_begin(self, crs: "Groth16Crs", a_eval: int, b_eval: int, c_eval: int, x: int, w: int, r: np.ndarray, s: np.ndarray, ticket: int):
        """kg_groth16_prove_begin: enqueue a proof (ticket 0 or 1); crs and the device inputs stay alive until the end."""
        r = np.ascontiguousarray(r, dtype=np.uint64)
        s = np.ascontiguousarray(s, dtype=np.uint64)
        rc = self._lib.kg_groth16_prove_begin(self._h, C.byref(crs), _vp(a_eval), _vp(b_eval), _vp(c_eval), _vp(x), _vp(w),
                                              r.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), int(ticket))
        self._chk(rc, "kg_groth16_prove_begin")

    def groth16_prove_r1cs(self, crs: "Groth16Crs", a, b, c, x: int, w: int, r: np.ndarray, s: np.ndarray):
        """kg_groth16_prove_r1cs_bn254: a, b, c = (row_ptr, col, val) device pointers of the constraint matrices (CSR over x || w)"""
        mk = lambda t: Csr(_vp(t[0]), _vp(t[1]), _vp(t[2]))
        ca, cb, cc = mk(a), mk(b), mk(c)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        s = np.ascontiguousarray(s, dtype=np.uint64)
        out = np.zeros(32, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        rc = self._lib.kg_groth16_prove_r1cs_bn254(self._h, C.byref(crs), C.byref(ca), C.byref(cb), C.byref(cc), _vp(x), _vp(w),
                                                   r.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                                                   out.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p))
        if rc == -6:
            raise ProverSubVersionCrsAttack("delta is the identity")
        self._chk(rc, "kg_groth16_prove_r1cs_bn254")
        return out[:8].copy(), out[8:24].copy(), out[24:].copy(), inf

    def groth16_prove_r1cs_begin(self, crs: "Groth16Crs", a, b, c, x: int, w: int, r: np.ndarray, s: np.ndarray, ticket: int):
        mk = lambda t: Csr(_vp(t[0]), _vp(t[1]), _vp(t[2]))
        ca, cb, cc = mk(a), mk(b), mk(c)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        s = np.ascontiguousarray(s, dtype=np.uint64)
        rc = self._lib.kg_groth16_prove_r1cs_begin(self._h, C.byref(crs), C.byref(ca), C.byref(cb), C.byref(cc), _vp(x), _vp(w),
                                                   r.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), int(ticket))
        self._chk(rc, "kg_groth16_prove_r1cs_begin")

    def groth16_prove_end(self, ticket: int):
        out = np.zeros(32, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        rc = self._lib.kg_groth16_prove_end(self._h, int(ticket), out.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p))
        if rc == -6:
            raise ProverSubVersionCrsAttack("delta is the identity")
        self._chk(rc, "kg_groth16_prove_end")
        return out[:8].copy(), out[8:24].copy(), out[24:].copy(), inf

    def groth16_setup(self, a, b, c, m: int, l: int, m_l_1: int, toxic: np.ndarray, crs: "Groth16Crs", ic: int, ic_inf: int):
        """kg_groth16_setup_bn254: a, b, c = (row_ptr, col, val) device pointers; crs carries the output arrays' device pointers and
        comes back ready for groth16_prove; returns (gamma_g2, vk_inf[6])"""
        mk = lambda t: Csr(_vp(t[0]), _vp(t[1]), _vp(t[2]))
        ca, cb, cc = mk(a), mk(b), mk(c)
        toxic = np.ascontiguousarray(toxic, dtype=np.uint64).reshape(5, 4)
        gamma_g2 = np.zeros(16, dtype=np.uint64)
        vk_inf = np.zeros(6, dtype=np.uint8)
        rc = self._lib.kg_groth16_setup_bn254(self._h, C.byref(ca), C.byref(cb), C.byref(cc), C.c_size_t(m), C.c_size_t(l), C.c_size_t(m_l_1),
                                              toxic.ctypes.data_as(C.c_void_p), C.byref(crs), _vp(ic), _vp(ic_inf),
                                              gamma_g2.ctypes.data_as(C.c_void_p), vk_inf.ctypes.data_as(C.c_void_p))
        if rc == -7:
            raise ProverInversionFailed("gamma or delta is zero")
        self._chk(rc, "kg_groth16_setup_bn254")
        return gamma_g2, vk_inf

    def profile_enable(self, on: bool = True):
        self._chk(self._lib.kg_profile_enable(self._h, int(on)), "kg_profile_enable")

    def profile_last(self) -> dict:
        """{phase: summed ms} since profile_enable"""
        return {k: v[0] for k, v in self.profile_summary().items()}

    def profile_summary(self) -> dict:
        """{phase: (summed ms, occurrences)} since profile_enable"""
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        cnt = (C.c_int * 32)()
        n = self._lib.kg_profile_summary(self._h, names, ms, cnt, 32)
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(max(n, 0))}


class DeviceArray:
    """A device allocation owned by a Context (freed on garbage collection)."""

    def __init__(self, ctx: Context, nbytes: int, shape, dtype):
        self.ctx, self.nbytes, self.shape, self.dtype = ctx, nbytes, tuple(shape), np.dtype(dtype)
        self.ptr = ctx.malloc(max(nbytes, 1))

    def __del__(self):
        try:
            if self.ptr and self.ctx._h:
                self.ctx.free(self.ptr)
        except Exception:
            pass
        self.ptr = 0

    def numpy(self) -> np.ndarray:
        return self.ctx.download(self)


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """kg_shard_range: contiguous slice [lo, hi) of `rank` (no device needed)"""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    rc = load().kg_shard_range(C.c_size_t(n), int(rank), int(world), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise KogarashiError(f"kg_shard_range: {load().kg_strerror(rc).decode()}")
    return int(lo.value), int(hi.value)


def _ptr_array(ctype, values):
    return (ctype * len(values))(*[ctype(int(v) if v else 0) for v in values])


def commit_sharded(ctxs, curve: int, bases, infs, scalars, n_local):
    """kg_commit_sharded over `ctxs`: per-context device pointers (ints) and pair counts; returns (xy, inf)."""
    k = len(ctxs)
    h = (C.c_void_p * k)(*[c._h for c in ctxs])
    pb, ps = _ptr_array(C.c_void_p, bases), _ptr_array(C.c_void_p, scalars)
    pi = _ptr_array(C.c_void_p, infs) if infs is not None else None
    nl = (C.c_size_t * k)(*[int(v) for v in n_local])
    xy = np.zeros(16 if curve == KG_G2 else 8, dtype=np.uint64)
    oi = C.c_uint8(0)
    ctxs[0]._chk(load().kg_commit_sharded(h, k, curve, pb, pi, ps, nl, xy.ctypes.data_as(C.c_void_p), C.byref(oi)), "kg_commit_sharded")
    return xy, int(oi.value)


def msm_sharded(ctxs, curve: int, bases, infs, scalars, n_local) -> np.ndarray:
    k = len(ctxs)
    h = (C.c_void_p * k)(*[c._h for c in ctxs])
    pb, ps = _ptr_array(C.c_void_p, bases), _ptr_array(C.c_void_p, scalars)
    pi = _ptr_array(C.c_void_p, infs) if infs is not None else None
    nl = (C.c_size_t * k)(*[int(v) for v in n_local])
    out = np.zeros(24 if curve == KG_G2 else 12, dtype=np.uint64)
    ctxs[0]._chk(load().kg_msm_sharded(h, k, curve, pb, pi, ps, nl, out.ctypes.data_as(C.c_void_p)), "kg_msm_sharded")
    return out


def groth16_prove_sharded(ctxs, crss, a_eval, b_eval, c_eval, x, w, r: np.ndarray, s: np.ndarray):
    """kg_groth16_prove_sharded: one proof over the contexts `ctxs`, task-parallel (G2 query | G1 queries | transforms + h).
    crss: one Groth16Crs per context (its own device pointers); a_eval .. w: per-context device pointers (0 where a context
    does not read the vector).  Returns (A, B, C, inf) like Context.groth16_prove."""
    k = len(ctxs)
    h = (C.c_void_p * k)(*[c._h for c in ctxs])
    pc = (C.POINTER(Groth16Crs) * k)(*[C.pointer(c) for c in crss])
    arrs = [_ptr_array(C.c_void_p, v) for v in (a_eval, b_eval, c_eval, x, w)]
    r = np.ascontiguousarray(r, dtype=np.uint64)
    s = np.ascontiguousarray(s, dtype=np.uint64)
    out = np.zeros(32, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    rc = load().kg_groth16_prove_sharded(h, k, pc, *arrs, r.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p))
    if rc == -6:
        raise ProverSubVersionCrsAttack("delta is the identity")
    ctxs[0]._chk(rc, "kg_groth16_prove_sharded")
    return out[:8].copy(), out[8:24].copy(), out[24:].copy(), inf


class ShardedKey:
    """kg_sharded_key: a commitment key (nova/src/pedersen.rs:6-13) resident across several contexts"""

    def __init__(self, ctxs, curve: int, bases: np.ndarray, inf=None):
        self.ctxs, self.curve = list(ctxs), curve
        w = 16 if curve == KG_G2 else 8
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
        ip = None
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            ip = inf.ctypes.data_as(C.c_void_p)
        k = len(self.ctxs)
        h = (C.c_void_p * k)(*[c._h for c in self.ctxs])
        key = C.c_void_p()
        self.ctxs[0]._chk(load().kg_sharded_key_create(h, k, curve, bases.ctypes.data_as(C.c_void_p), ip, C.c_size_t(len(bases)), C.byref(key)),
                          "kg_sharded_key_create")
        self._k = key

    def __len__(self):
        return int(load().kg_sharded_key_len(self._k))

    def commit(self, m: np.ndarray):
        m = np.ascontiguousarray(m, dtype=np.uint64).reshape(-1, 4)
        xy = np.zeros(16 if self.curve == KG_G2 else 8, dtype=np.uint64)
        oi = C.c_uint8(0)
        self.ctxs[0]._chk(load().kg_sharded_key_commit(self._k, m.ctypes.data_as(C.c_void_p), C.c_size_t(len(m)), xy.ctypes.data_as(C.c_void_p), C.byref(oi)),
                          "kg_sharded_key_commit")
        return xy, int(oi.value)

    def close(self):
        if getattr(self, "_k", None):
            load().kg_sharded_key_destroy(self._k)
            self._k = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
