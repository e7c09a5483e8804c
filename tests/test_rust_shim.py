"""The Rust shim (rust/) cannot be compiled in this image (no rustc / cargo): these checks keep it mechanically in step
with the C ABI and with the reference tree it patches."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as G  # noqa: E402

RUST = os.path.join(ROOT, "rust")
HEADER = open(os.path.join(ROOT, "include", "kogarashi_amd.h")).read()


def test_ffi_block_is_generated_from_the_header():
    fns = G.parse(HEADER)
    declared = sorted(set(re.findall(r"\b(kg_[a-z0-9_]+)\s*\(", G.strip_comments(HEADER))))
    assert sorted(f[0] for f in fns) == declared                      # the parser sees every function of the header
    assert open(G.OUT).read() == G.render(fns), "run python tools/gen_rust_ffi.py"
    by_name = {f[0]: f for f in fns}
    # spot checks of the type mapping
    assert by_name["kg_commit_sharded"][1][0] == ("ctxs", "*const *mut KgCtx")
    assert by_name["kg_commit_sharded"][1][3] == ("d_bases", "*const *const u64")
    assert by_name["kg_strerror"][2] == "*const c_char" and by_name["kg_ctx_destroy"][2] is None
    assert by_name["kg_malloc"][1][2] == ("d_ptr", "*mut *mut c_void")


def test_glue_calls_match_the_header():
    """every sys::kg_* call in rust/kogarashi-amd names a header function and passes its number of arguments"""
    arity = {f[0]: len(f[1]) for f in G.parse(HEADER)}
    seen = set()
    for dirpath, _, files in os.walk(os.path.join(RUST, "kogarashi-amd", "src")):
        for f in files:
            src = open(os.path.join(dirpath, f)).read()
            for m in re.finditer(r"sys::(kg_[a-z0-9_]+)\s*\(", src):
                name, i, depth, args, cur = m.group(1), m.end(), 1, 0, ""
                while depth:
                    ch = src[i]
                    depth += ch in "([{"
                    depth -= ch in ")]}"
                    if ch == "," and depth == 1:
                        args += 1
                        cur = ""
                    elif depth:
                        cur += ch
                    i += 1
                n_args = args + (1 if cur.strip() else 0)
                assert name in arity, (f, name)
                assert n_args == arity[name], (f, name, n_args, arity[name])
                seen.add(name)
    for needed in ("kg_msm_host", "kg_msm_host_scalars", "kg_groth16_setup_bn254", "kg_field_vec_axpy", "kg_ntt_bn254_fr", "kg_fr_divide_by_z_on_coset", "kg_sharded_key_create", "kg_sharded_key_commit",
                   "kg_groth16_prove_bn254", "kg_groth16_prove_r1cs_bn254", "kg_groth16_prove_sharded", "kg_r1cs_evaluate", "kg_bases_register",
                   "kg_bases_unregister", "kg_bases_precompute", "kg_nova_cross_term"):
        assert needed in seen, needed


def test_crs_struct_mirrors_the_header():
    c = re.search(r"typedef struct \{(.*?)\} kg_groth16_crs;", G.strip_comments(HEADER), re.S).group(1)
    c_fields = []
    for decl in c.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            c_fields.append(re.search(r"([A-Za-z_0-9]+)\s*(\[\d+\])?$", part.strip()).group(1))
    r = re.search(r"pub struct KgGroth16Crs \{(.*?)\n\}", open(os.path.join(RUST, "kogarashi-amd-sys", "src", "lib.rs")).read(), re.S).group(1)
    r_fields = re.findall(r"pub ([a-z0-9_]+):", r)
    assert r_fields == c_fields
    # and the ctypes mirror the tests run through
    from kogarashi_amd.lib import Groth16Crs
    assert [f[0] for f in Groth16Crs._fields_] == c_fields


def test_constants_match_the_header():
    lib_rs = open(os.path.join(RUST, "kogarashi-amd-sys", "src", "lib.rs")).read()
    consts = {k: int(v) for k, v in re.findall(r"pub const (KG_[A-Z0-9_]+): i32 = (-?\d+);", lib_rs)}
    hdr = G.strip_comments(HEADER)
    for name, val in re.findall(r"\b(KG_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr):
        assert consts.get(name) == int(val), name


@pytest.mark.skipif(not os.path.isdir("/root/reference/groth16"), reason="the reference tree is only present in the build container")
def test_patches_apply_to_the_reference(tmp_path):
    for d in ("groth16", "nova", "zkstd", "bn254"):
        subprocess.check_call(["cp", "-r", os.path.join("/root/reference", d), str(tmp_path / d)])
    patches = sorted(p for p in os.listdir(os.path.join(RUST, "patches")) if p.endswith(".diff"))
    assert len(patches) >= 11
    for p in patches:
        with open(os.path.join(RUST, "patches", p)) as f:
            subprocess.run(["patch", "-p1", "-s"], stdin=f, cwd=str(tmp_path), check=True)
    # the call sites the patches name exist afterwards
    assert "kogarashi_amd::msm(bases, coeffs)" in (tmp_path / "groth16/src/msm.rs").read_text()
    assert "kogarashi_amd::pedersen::commit" in (tmp_path / "nova/src/pedersen.rs").read_text()
    prover = (tmp_path / "groth16/src/prover.rs").read_text()
    assert "kogarashi_amd::groth16::resident" in prover
    # the matrices are handed over as a closure: cs.matrices() (three O(nnz) clones) runs for the first proof against a CRS only
    assert "crs.prove_cs_with(|| cs.matrices()," in prover and "let (a, b, c) = cs.matrices();" not in prover.split("let fft = Fft")[0]
    assert (tmp_path / "groth16/src/fft.rs").read_text().count("gpu::transform") == 5
    assert "kogarashi_amd::nova::cross_term" in (tmp_path / "nova/src/prover.rs").read_text()
    assert "pub fn to_csr" in (tmp_path / "zkstd/src/matrix.rs").read_text()
    zk = (tmp_path / "groth16/src/zksnark.rs").read_text()
    # the device setup sits AFTER the five toxic scalars are drawn and the two inversions have had their chance to fail
    assert zk.index("let tau = Fr::random(&mut r);") < zk.index("delta.invert().ok_or(Error::ProverInversionFailed)?") < zk.index("kogarashi_amd::groth16::setup(")
    assert zk.index("kogarashi_amd::groth16::setup(") < zk.index("let mut h = vec![G1Affine::ADDITIVE_IDENTITY")
    assert (tmp_path / "nova/src/relaxed_r1cs/witness.rs").read_text().count("kogarashi_amd::nova::axpy(") == 2


# std / core items stabilised after the reference's pinned toolchain (rust-toolchain: nightly-2022-11-14, i.e. 1.67-dev);
# the glue crate must compile there, and no compiler is available here to say so
TOO_NEW = [
    r"\bOnceLock\b", r"\bLazyLock\b", r"\bLazyCell\b", r"\bOnceCell\b",          # 1.70 / 1.80
    r"\.is_some_and\(", r"\.is_ok_and\(", r"\.is_err_and\(", r"\.is_none_or\(",      # 1.70 / 1.82
    r"\.inspect_err\(", r"\.div_ceil\(", r"\.next_multiple_of\(",                   # 1.76 / 1.73
    r"\.checked_ilog2\(", r"\.ilog2\(", r"\.ilog10\(", r"\.ilog\(",                  # 1.67 (release after the pin)
    r"\bstd::iter::repeat_n\b", r"\.array_chunks\b", r"\bhint::black_box\b",
    r"\bCStr::from_bytes_until_nul\b", r"\bc\"",                                    # 1.69 / C-string literals 1.77
]


def test_glue_avoids_std_items_newer_than_the_pinned_toolchain():
    pin = open("/root/reference/rust-toolchain").read().strip() if os.path.exists("/root/reference/rust-toolchain") else "nightly-2022-11-14"
    assert pin == "nightly-2022-11-14"
    for crate in ("kogarashi-amd", "kogarashi-amd-sys"):
        for dirpath, _, files in os.walk(os.path.join(RUST, crate)):
            for f in files:
                if not f.endswith(".rs"):
                    continue
                for ln, line in enumerate(open(os.path.join(dirpath, f)).read().splitlines(), 1):
                    code = line.split("//")[0]
                    for pat in TOO_NEW:
                        assert not re.search(pat, code), f"{crate}/{f}:{ln}: `{pat}` is not available on {pin}"


def test_process_wide_state_uses_const_mutexes():
    """every `static` of the glue is a `Mutex<Option<..>>` initialised by the const `Mutex::new(None)`"""
    n = 0
    for f in os.listdir(os.path.join(RUST, "kogarashi-amd", "src")):
        for m in re.finditer(r"^\s*static\s+([A-Z_]+)\s*:\s*([^=]+)=\s*([^;]+);", open(os.path.join(RUST, "kogarashi-amd", "src", f)).read(), re.M):
            n += 1
            assert m.group(2).strip().startswith("Mutex<Option<") and m.group(3).strip().startswith("Mutex::new(None)"), (f, m.group(0))
    assert n >= 4


def test_resident_matrices_are_keyed_by_a_hash_of_all_their_content():
    csr = open(os.path.join(RUST, "kogarashi-amd", "src", "csr.rs")).read()
    assert "pub(crate) fn content_hash" in csr and "for_each_entry" in csr and ".take(8)" not in csr
    assert "pub fn covers" in csr
    for f, needle in (("nova.rs", "content_hash([a, b, c], l)"), ("groth16.rs", "content_hash([&a, &b, &c], l)")):
        src = open(os.path.join(RUST, "kogarashi-amd", "src", f)).read()
        assert needle in src and ".covers(" in src, f
        assert "ResidentShape::host" not in src          # no host CSR rebuild on the per-call path
    # exact invariants travel with the hash, and structure words do not share the data words' mixing step
    assert "pub nnz: [u64; 3]" in csr and "let structure = |" in csr and "let data = |" in csr


def _glue(name):
    return open(os.path.join(RUST, "kogarashi-amd", "src", name)).read()


def test_groth16_shape_is_bound_to_the_cached_parameters_entry():
    """a proof against a resident CRS does not hash (or clone) the constraint matrices: first sight, debug builds and
    KOGARASHI_AMD_VERIFY_SHAPE only"""
    src = _glue("groth16.rs")
    body = src.split("fn shape_for")[1].split("\n    }\n")[0]
    assert "if shape.is_none() || verify_shape_every_proof()" in body
    assert body.index("mats()") > body.index("if shape.is_none()")          # the closure is called inside the branch only
    assert "cfg!(debug_assertions)" in src and "KOGARASHI_AMD_VERIFY_SHAPE" in src
    assert "pub fn prove_cs_with(" in src


def test_contexts_are_locked_one_by_one():
    """no process-wide guard is held across a backend call: contexts() hands out a reference-counted list, every use locks the
    context it needs (lock / lock_any / lock_all in index order)"""
    lib = _glue("lib.rs")
    assert "pub struct Contexts(Arc<Vec<CtxSlot>>)" in lib and "pub struct CtxSlot(Mutex<Context>)" in lib
    for fn in ("pub fn lock(&self, i: usize)", "pub fn lock_any(&self)", "pub fn lock_all(&self)"):
        assert fn in lib, fn
    assert "MutexGuard<'static" not in lib
    for f in os.listdir(os.path.join(RUST, "kogarashi-amd", "src")):
        src = _glue(f)
        assert not re.search(r"\bctxs\[\d+\]", src), f             # no direct indexing: a context is reached through its lock
        if "contexts()" in src and f != "lib.rs":
            assert re.search(r"ctxs\.lock(_any|_all)?\(", src), f


def test_msm_bases_are_marshalled_once_per_slice():
    lib = _glue("lib.rs")
    body = lib.split("fn msm_typed")[1].split("\n}\n")[0]
    assert "MSM_BASES" in body and "digest_for(bases, mode)" in body and "sys::kg_msm_host_scalars(" in lib and "kg_bases_register" in lib
    assert "kg_memcpy_h2d" not in body                                 # no whole-slice scalar upload in front of the MSM any more
    assert body.index("if !cache.contains_key(&id)") < body.index("Resident::upload(&ctx, bases, print)")
    assert "MSM_CACHE_SLOTS" in lib and "min_by_key" in body           # bounded: least recently used slice is dropped


def test_msm_cache_validation_is_content_based_and_residency_can_be_opted_out():
    """ADVICE r4 / VERDICT r5: the address-keyed cache is re-validated on every call with a digest over every coordinate word and flag of
    EVERY point by default (exact: msm_curve_addition borrows a slice and promises no immutability, groth16/src/msm.rs:6); the 256-point
    sample is an opt-in (KOGARASHI_AMD_MSM_RESIDENT=sampled), residency can be switched off (=0: kg_msm_host per call), and an explicit
    handle exists (register_bases -> ResidentMsmBases::msm)"""
    lib = _glue("lib.rs")
    assert "KOGARASHI_AMD_MSM_RESIDENT" in lib and "ResidencyMode::Off" in lib and "ResidencyMode::Hash" in lib
    mode = lib.split("fn residency_mode()")[1].split("\n}\n")[0]
    assert '_ => ResidencyMode::Hash' in mode and 'Some("sampled") | Some("probe") => ResidencyMode::Probe' in mode      # exact unless asked otherwise
    assert "let step = if mode == ResidencyMode::Hash { 1 }" in lib                                                     # Hash = every point
    dg = lib.split("fn digest<C: GpuCurve>")[1].split("\n}\n")[0]
    assert "put_xy(&mut words)" in dg and "is_identity()" in dg and "take(n - 1" in dg
    assert "const PROBE_POINTS: usize = 256;" in lib and "probe3" not in lib
    assert "pub fn register_bases<C: GpuCurve>(bases: &[C]) -> Option<ResidentMsmBases>" in lib
    body = lib.split("fn msm_typed")[1].split("\n}\n")[0]
    assert "mode == ResidencyMode::Off" in body and body.index("ResidencyMode::Off") < body.index("MSM_BASES.lock()")
    # a value of the cache is dropped only while GPU 0's lock is held (DeviceBuf::drop calls kg_free on that context)
    assert body.index("ctxs.lock(0)") < body.index("MSM_BASES.lock()")


def test_setup_and_folds_are_wired():
    """ZkSnark::setup reaches kg_groth16_setup_bn254 through the glue and a patch to zksnark.rs that keeps the rng
    order; RelaxedR1csWitness::fold reaches kg_field_vec_axpy"""
    g16, nova = _glue("groth16.rs"), _glue("nova.rs")
    assert "sys::kg_groth16_setup_bn254(" in g16 and "pub fn setup(" in g16 and "pub struct SetupOutput" in g16
    assert "sys::kg_field_vec_axpy(" in nova and "pub fn axpy<" in nova
    z = open(os.path.join(RUST, "patches", "groth16_zksnark.diff")).read()
    assert "kogarashi_amd::groth16::setup(&ma, &mb, &mc" in z and "&[alpha, beta, gamma, delta, tau]" in z
    w = open(os.path.join(RUST, "patches", "nova_witness.diff")).read()
    assert w.count("kogarashi_amd::nova::axpy(") == 2


def test_one_gpu_per_proof_is_the_default():
    """the task split of one proof over several GPUs is opt-in (KOGARASHI_AMD_SHARD_GPUS) and locks only the contexts it
    uses"""
    src = _glue("groth16.rs")
    assert "KOGARASHI_AMD_SHARD_GPUS" in src and "fn shard_gpus() -> usize" in src and "_ => 1," in src
    assert "lock_all" not in src and src.count("ctxs.lock_set(") >= 3
    assert "pub fn lock_set(&self, idx: &[usize])" in _glue("lib.rs")


def test_sharded_proof_is_wrapped():
    src = _glue("groth16.rs")
    assert "sys::kg_groth16_prove_sharded(" in src and "fn owners(n_ctx: usize) -> [usize; 3]" in src
    assert "[0, 1 % n_ctx, 2 % n_ctx]" in src                          # the split kg_groth16_prove_sharded itself uses (groth16.hip)
    hip = open(os.path.join(ROOT, "kogarashi_amd", "csrc", "groth16.hip")).read()
    assert "const int owner[3] = {0, 1 % n_ctx, 2 % n_ctx};" in hip
    patch = open(os.path.join(RUST, "patches", "zkstd_matrix_csr.diff")).read()
    assert "pub fn for_each_entry" in patch and "pub fn rows" in patch
