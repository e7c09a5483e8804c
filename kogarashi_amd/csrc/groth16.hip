// groth16.hip -- fixed-base multiples and the Groth16 prover orchestration on the device.
//
// kg_groth16_prove_bn254 replaces groth16/src/prover.rs:20-99 (after synthesis): 6 NTTs (idft + coset_dft of the
// A, B, C evaluation vectors, :36-41), the point-wise h = a*b - c and the division by Z on the coset (:43-46,
// fused into one kernel here), one coset iNTT (:47), eight MSMs against the resident CRS (:51-65) and the proof
// assembly (:71-98).  The polynomial vectors never leave HBM between steps; only the eight MSM results and the
// three proof points touch the host, which also runs the six short scalar multiplications of the assembly
// (sequential double-and-add, as in the reference).
#include "common.h"
#include "msm_internal.h"
#include <chrono>
#include <cstdlib>
#include "host_fp.h"
#include <future>
#include <vector>
#include <type_traits>

using namespace kg;

namespace {

// ---- generator multiples --------------------------------------------------------------------------
template <class F> struct Gen;
template <> struct Gen<Fq> {                    // bn254/src/params.rs:8-9: (1, 2)
  static __device__ Affine<Fq> g() {
    uint32_t two[8] = {2, 0, 0, 0, 0, 0, 0, 0};
    return {Fq::one(), from_int<FqParams>(two)};
  }
};
template <> struct Gen<Fr> {                    // grumpkin/src/params.rs:4-10: (1, sqrt(-16))
  static __device__ Affine<Fr> g() {
    // GENERATOR_Y in the reference's Montgomery form, converted on the fly
    const uint32_t y[8] = {0x448c41d8u, 0x11b2dff1u, 0x21c77dc3u, 0x23d3446fu, 0x35dfafbbu, 0xaa7b8cf4u, 0x9dc25d68u, 0x14b34cf6u};
    return {Fr::one(), from_ref<FrParams>(y)};
  }
};
template <> struct Gen<Fq2> {                   // bn254/src/params.rs:15-42 (canonical integers)
  static __device__ Affine<Fq2> g() {
    const uint32_t x0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
    const uint32_t x1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
    const uint32_t y0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
    const uint32_t y1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};
    return {{from_int<FqParams>(x0), from_int<FqParams>(x1)}, {from_int<FqParams>(y0), from_int<FqParams>(y1)}};
  }
};

template <class P>
__device__ __forceinline__ void put_ref(const Fp<P>& a, uint64_t* dst) {
  uint32_t w[8];
  to_ref(a, w);
  store_words(dst, 0, w);
}
template <class F>
__device__ __forceinline__ void put_ref(const Fp2<F>& a, uint64_t* dst) { put_ref(a.c0, dst); put_ref(a.c1, dst + 4); }

// one lane per scalar: MSB-first double-and-add on the generator, then one Fermat inversion to affine
template <class F, class SP, int E64>
__global__ void __launch_bounds__(64) k_fixed_base_mul(const uint64_t* __restrict__ k, size_t n, uint64_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8], e[8];
  load_words(k, i, w);
  ref_to_int<SP>(w, e);
  const Affine<F> g = Gen<F>::g();
  XYZZ<F> acc = XYZZ<F>::identity();
  for (int b = 253; b >= 0; --b) {
    acc = double_xyzz(acc);
    if ((e[b >> 5] >> (b & 31)) & 1) acc = add_mixed(acc, g);
  }
  Affine<F> a;
  uint64_t* dst = out_xy + i * 2 * E64;
  if (to_affine(acc, a)) {
    put_ref(a.x, dst);
    put_ref(a.y, dst + E64);
    out_inf[i] = 0;
  } else {                                   // (0, 1, inf): macros/curve/weierstrass/group.rs:22-26
    put_ref(F::zero(), dst);
    put_ref(F::one(), dst + E64);
    out_inf[i] = 1;
  }
}

// h[i] = (a[i] * b[i] - c[i]) * zinv   (poly.rs:168-195 + fft.rs:150-154 in one pass; data stays in ABI form)
__global__ void __launch_bounds__(256) k_qap_combine(uint64_t* __restrict__ a, const uint64_t* __restrict__ b, const uint64_t* __restrict__ c,
                                                     size_t n, Words8 zinv) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[8], wb[8], wc[8], wo[8];
  load_words(a, i, wa); load_words(b, i, wb); load_words(c, i, wc);
  Fr x = from_ref<FrParams>(wa), y = from_ref<FrParams>(wb), z = from_ref<FrParams>(wc);
  Fr t = norm(sub<4, 1>(mul(x, y), z));
  to_ref(mul(t, from_ref<FrParams>(zinv.w)), wo);
  store_words(a, i, wo);
}

// ---- host curve helpers (assembly of the proof, prover.rs:71-98) ---------------------------------------
template <class HF, int E>
Affine<HF> h_load_aff(const uint64_t* p);
template <>
Affine<HostFq> h_load_aff<HostFq, 4>(const uint64_t* p) { return {HostFq::from_words(p), HostFq::from_words(p + 4)}; }
template <>
Affine<HostFq2> h_load_aff<HostFq2, 8>(const uint64_t* p) {
  return {{HostFq::from_words(p), HostFq::from_words(p + 4)}, {HostFq::from_words(p + 8), HostFq::from_words(p + 12)}};
}
// ABI projective (x, y, z in {0, 1}) -> XYZZ
template <class HF>
XYZZ<HF> h_from_abi(const Affine<HF>& a, bool inf) { return inf ? XYZZ<HF>::identity() : from_affine(a); }

template <class HF>
XYZZ<HF> h_scalar_mul(const XYZZ<HF>& p, const uint64_t k[4]) {
  XYZZ<HF> acc = XYZZ<HF>::identity();
  for (int b = 255; b >= 0; --b) {
    acc = double_xyzz(acc);
    if ((k[b >> 6] >> (b & 63)) & 1) acc = add_xyzz(acc, p);
  }
  return acc;
}
// k1 * p + k2 * q in one double-and-add pass (Straus)
template <class HF>
XYZZ<HF> h_scalar_mul2(const XYZZ<HF>& p, const uint64_t k1[4], const XYZZ<HF>& q, const uint64_t k2[4]) {
  const XYZZ<HF> pq = add_xyzz(p, q);
  XYZZ<HF> acc = XYZZ<HF>::identity();
  for (int b = 255; b >= 0; --b) {
    acc = double_xyzz(acc);
    const int b1 = (int)((k1[b >> 6] >> (b & 63)) & 1), b2 = (int)((k2[b >> 6] >> (b & 63)) & 1);
    if (b1 && b2) acc = add_xyzz(acc, pq);
    else if (b1) acc = add_xyzz(acc, p);
    else if (b2) acc = add_xyzz(acc, q);
  }
  return acc;
}
void h_store(const HostFq& a, uint64_t* w) { a.to_words(w); }
void h_store(const HostFq2& a, uint64_t* w) { a.c0.to_words(w); a.c1.to_words(w + 4); }
template <class HF, int E>
void h_store_affine(const XYZZ<HF>& p, uint64_t* xy, uint8_t* inf) {
  Affine<HF> a;
  if (to_affine(p, a)) { h_store(a.x, xy); h_store(a.y, xy + E); *inf = 0; }
  else { h_store(HF::zero(), xy); h_store(HF::one(), xy + E); *inf = 1; }
}

// ---- windowed fixed-base multiples (zksnark.rs:57,168-187: 5 m generator multiples per setup) -----------------------
// table[w][d - 1] = (d * 2^(8w)) * G for w < 32, d = 1..255, affine in the MSM's resident form (18 / 36 words per point):
// a multiple is then 32 mixed additions and no doubling.  The table is built once per (context, curve) with the
// double-and-add kernel above; 8 160 points = 587 KB (G1) stay in L2.
constexpr int FB_WIN = 32, FB_ENT = 255;

template <class F> struct FbIO;                     // resident words of an affine point -> Affine<F>
template <class P> struct FbIO<Fp<P>> {
  static constexpr int PW = 18;
  static __device__ __forceinline__ Affine<Fp<P>> load(const uint32_t* w) {
    Affine<Fp<P>> a;
#pragma unroll
    for (int k = 0; k < 9; ++k) { a.x.l[k] = w[k]; a.y.l[k] = w[9 + k]; }
    return a;
  }
};
template <class G> struct FbIO<Fp2<G>> {
  static constexpr int PW = 36;
  static __device__ __forceinline__ Affine<Fp2<G>> load(const uint32_t* w) {
    Affine<Fp2<G>> a;
#pragma unroll
    for (int k = 0; k < 9; ++k) { a.x.c0.l[k] = w[k]; a.x.c1.l[k] = w[9 + k]; a.y.c0.l[k] = w[18 + k]; a.y.c1.l[k] = w[27 + k]; }
    return a;
  }
};
// resident words <- ABI affine (x | y): the same conversion k_prep_bases does, one lane per point
template <class F, int E64>
__global__ void __launch_bounds__(64) k_fb_pack(const uint64_t* __restrict__ xy, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int PW = FbIO<F>::PW;
  for (int e = 0; e < 2 * E64 / 4; ++e) {            // field elements of the point: x, y (Fq2: x.c0, x.c1, y.c0, y.c1)
    uint32_t w[8];
    load_words(xy + i * 2 * E64 + 4 * e, 0, w);
    const auto v = from_ref<typename std::conditional<std::is_same<F, Fr>::value, FrParams, FqParams>::type>(w);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[i * PW + 9 * e + k] = v.l[k];
  }
}

// pass 1: one lane per scalar, 32 table additions -> XYZZ, raw internal form, array of structures (4 coordinates)
template <class F, class SP>
__global__ void __launch_bounds__(64) k_fb_window(const uint64_t* __restrict__ k, size_t n, const uint32_t* __restrict__ table,
                                                  uint32_t* __restrict__ tmp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8], e[8];
  load_words(k, i, w);
  ref_to_int<SP>(w, e);
  constexpr int PW = FbIO<F>::PW;
  XYZZ<F> acc = XYZZ<F>::identity();
  for (int win = 0; win < FB_WIN; ++win) {
    const uint32_t d = (e[win >> 2] >> (8 * (win & 3))) & 0xffu;
    if (d) acc = add_mixed(acc, FbIO<F>::load(table + ((size_t)win * FB_ENT + (d - 1)) * PW));
  }
  PointIO<F>::store(tmp, n, i, acc);                  // structure of arrays over the n scalars
}

// pass 2: XYZZ -> affine ABI words with ONE inversion per lane for FB_BATCH consecutive points (Montgomery's trick)
constexpr int FB_BATCH = 8;
template <class F, int E64>
__global__ void __launch_bounds__(64) k_fb_affine(const uint32_t* __restrict__ tmp, size_t n, uint64_t* __restrict__ out_xy,
                                                  uint8_t* __restrict__ out_inf) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t first = t * FB_BATCH;
  if (first >= n) return;
  const int cnt = (int)(n - first < (size_t)FB_BATCH ? n - first : (size_t)FB_BATCH);
  constexpr int E = RawIO<F>::NW;
  F pre[FB_BATCH];                                    // prefix products of the non-zero ZZZ
  F run = F::one();
#pragma unroll
  for (int j = 0; j < FB_BATCH; ++j) {
    if (j < cnt) {
      const F zzz = RawIO<F>::load(tmp + (size_t)3 * E * n, n, first + j);
      if (!is_zero_2p(zzz)) run = mul(run, zzz);     // an identity (k = 0) contributes nothing
    }
    pre[j] = run;
  }
  F inv_run = inv(run);                               // (prod of the batch's ZZZ)^-1
#pragma unroll
  for (int j = FB_BATCH - 1; j >= 0; --j) {
    if (j >= cnt) continue;
    const size_t i = first + j;
    const F zzz = RawIO<F>::load(tmp + (size_t)3 * E * n, n, i);
    uint64_t* dst = out_xy + i * 2 * E64;
    if (is_zero_2p(zzz)) {                            // (0, 1, inf): macros/curve/weierstrass/group.rs:22-26
      put_ref(F::zero(), dst);
      put_ref(F::one(), dst + E64);
      out_inf[i] = 1;
      continue;
    }
    const F zi = mul(inv_run, j ? pre[j - 1] : F::one());      // ZZZ_i^-1
    inv_run = mul(inv_run, zzz);
    const F x = RawIO<F>::load(tmp, n, i), y = RawIO<F>::load(tmp + (size_t)E * n, n, i), zz = RawIO<F>::load(tmp + (size_t)2 * E * n, n, i);
    const F zzi = mul(mul(zi, zi), sqr(zz));          // ZZ^-1 = ZZZ^-2 * ZZ^2   (ZZ^3 = ZZZ^2)
    put_ref(mul(x, zzi), dst);
    put_ref(mul(y, zi), dst + E64);
    out_inf[i] = 0;
  }
}

template <class F, class SP, int E64>
int fixed_base_t(kg_ctx* ctx, int curve, const uint64_t* d_k, size_t n, uint64_t* d_out_xy, uint8_t* d_out_inf) {
  hipStream_t st = ctx->stream;
  constexpr int PW = FbIO<F>::PW;
  const size_t tn = (size_t)FB_WIN * FB_ENT;
  if (n < 512 && !ctx->fb_table[curve]) {             // few scalars and no table yet (vk elements of a host that never makes a CRS, tests): plain
                                                      // double-and-add, a 255-step chain per lane (2 ms in G1, 5.5 ms in G2 whatever n is); also builds the
                                                      // table.  Once the table exists (kg_groth16_setup_bn254 has made its long vectors first) the few go
                                                      // through it as well: 32 additions and an inversion, ~0.5 ms
    hipLaunchKernelGGL((k_fixed_base_mul<F, SP, E64>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_k, n, d_out_xy, d_out_inf);
    KG_HIP(ctx, hipGetLastError());
    return KG_OK;
  }
  if (!ctx->fb_table[curve]) {
    // table scalars d * 2^(8w) as canonical integers -> the scalar field's Montgomery form -> generator multiples -> resident form
    std::vector<uint64_t> hk(tn * 4, 0);
    for (int w = 0; w < FB_WIN; ++w)
      for (int d = 1; d <= FB_ENT; ++d) hk[((size_t)w * FB_ENT + (d - 1)) * 4 + (size_t)(w / 8)] = (uint64_t)d << (8 * (w % 8));
    // (w = 31: d * 2^248 may exceed the group order; it is used as a plain integer multiple, which is what the table needs --
    // the double-and-add below takes the canonical 254 bits of k mod p only, so reduce on the host instead)
    uint64_t *d_tk = nullptr, *d_txy = nullptr; uint8_t* d_tinf = nullptr; uint32_t* table = nullptr;
    auto release = [&](bool keep_table) {             // every exit below frees what was allocated before it
      hipFree(d_tk); hipFree(d_txy); hipFree(d_tinf);
      if (!keep_table) hipFree(table);
    };
    hipError_t e = dev_alloc(ctx, (void**)&d_tk, tn * 32);
    if (e == hipSuccess) e = dev_alloc(ctx, (void**)&d_txy, tn * 2 * E64 * 8);
    if (e == hipSuccess) e = dev_alloc(ctx, (void**)&d_tinf, tn);
    if (e == hipSuccess) e = dev_alloc(ctx, (void**)&table, tn * PW * 4);
    if (e != hipSuccess) { release(false); return set_err(ctx, KG_ERR_OOM, "fixed-base table allocation", e); }
    e = hipMemcpyAsync(d_tk, hk.data(), tn * 32, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { release(false); return set_err(ctx, KG_ERR_HIP, "fixed-base table upload", e); }
    int rc = kg_field_vec_op(ctx, curve == KG_GRUMPKIN ? KG_FQ : KG_FR, KG_OP_TO_MONT, d_tk, nullptr, d_tk, tn);
    if (rc == KG_OK) {
      hipLaunchKernelGGL((k_fixed_base_mul<F, SP, E64>), dim3((unsigned)((tn + 63) / 64)), dim3(64), 0, st, d_tk, tn, d_txy, d_tinf);
      hipLaunchKernelGGL((k_fb_pack<F, E64>), dim3((unsigned)((tn + 63) / 64)), dim3(64), 0, st, d_txy, tn, table);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = set_err(ctx, KG_ERR_HIP, "fixed-base table construction");
    }
    release(rc == KG_OK);
    if (rc != KG_OK) return rc;
    ctx->fb_table[curve] = table;
  }
  constexpr int NW = PointIO<F>::NW;
  if (n * NW * 4 > ctx->fb_tmp_bytes) {
    if (ctx->fb_tmp) { sync_all(ctx); hipFree(ctx->fb_tmp); ctx->fb_tmp = nullptr; ctx->fb_tmp_bytes = 0; }
    KG_HIP(ctx, dev_alloc(ctx, &ctx->fb_tmp, n * NW * 4));
    ctx->fb_tmp_bytes = n * NW * 4;
  }
  uint32_t* tmp = (uint32_t*)ctx->fb_tmp;
  hipLaunchKernelGGL((k_fb_window<F, SP>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_k, n, ctx->fb_table[curve], tmp);
  const size_t lanes = (n + FB_BATCH - 1) / FB_BATCH;
  hipLaunchKernelGGL((k_fb_affine<F, E64>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, tmp, n, d_out_xy, d_out_inf);
  KG_HIP(ctx, hipGetLastError());
  return KG_OK;
}

}  // namespace

extern "C" {

int kg_fixed_base_mul(kg_ctx* ctx, int curve, const uint64_t* d_k, size_t n, uint64_t* d_out_xy, uint8_t* d_out_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {              // (the table construction holds a std::vector)
  if (!ctx || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  if (!d_k || !d_out_xy || !d_out_inf) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  if (curve == KG_G1) return fixed_base_t<Fq, FrParams, 4>(ctx, curve, d_k, n, d_out_xy, d_out_inf);
  if (curve == KG_GRUMPKIN) return fixed_base_t<Fr, FqParams, 4>(ctx, curve, d_k, n, d_out_xy, d_out_inf);
  return fixed_base_t<Fq2, FrParams, 8>(ctx, curve, d_k, n, d_out_xy, d_out_inf);
  });
}

}  // extern "C"

namespace {
// The blinding terms of prover.rs:75-77 -- five scalar multiplications of CRS points by r, s and r s that depend on no MSM result: three
// 255-step chains on worker threads (the G2 one three times as long as a G1 one), started when the proof is ENQUEUED (until round 5 by the
// assembly, i.e. behind the ~0.1 ms of launches of a short proof, whose critical path they were: 0.25 ms of Fq2 doublings).
struct Blinding {
  XYZZ<HostFq> g_a, rs_delta, sa_rb;         // r delta1 + alpha (:75), r s delta1 and s alpha + r beta1 (:77)
  XYZZ<HostFq2> g_b;                         // s delta2 + beta2 (:76)
  HostFr rk, sk;                             // r, s out of Montgomery form
  std::future<int> t[3];
  bool started = false;
  void wait() { for (std::future<int>& f : t) if (f.valid()) f.wait(); }
  ~Blinding() { wait(); }                    // the tasks write into this object
};
void start_blinding(WorkerPool* workers, const kg_groth16_crs& vk, const uint64_t* rr, const uint64_t* ss, Blinding* bl);
// One proof in flight: everything the host side needs between "all device work enqueued" and "proof read".
struct ProofJob {
  Blinding blind;
  uint64_t q_p[12], l_p[12], ai[12], b1i[12], b2i[24];
  std::future<int> f_q, f_l, f_a, f_b1, f_b2, assembly;
  uint64_t proof[32];
  uint8_t inf[3];
  uint64_t rr[4], ss[4];     // the blinding scalars, copied when the proof is enqueued
  int rc0 = 0;               // status of the enqueue, reported by the assembly after it has joined every host finish
  bool active = false;
  ~ProofJob() {              // a proof still in flight when its context goes: its tasks write into this object
    for (std::future<int>* f : {&assembly, &f_q, &f_l, &f_a, &f_b1, &f_b2}) if (f->valid()) f->wait();
  }
};
struct ProofJobs { ProofJob j[2]; };   // job 0: kg_groth16_prove_bn254 and ticket 0 (result slots 6..10); job 1: ticket 1 (slots 11..15)
ProofJob* job_of(kg_ctx* ctx, int i) {
  if (!ctx->prover_jobs) ctx->prover_jobs = std::make_shared<ProofJobs>();
  return &static_cast<ProofJobs*>(ctx->prover_jobs.get())->j[i];
}

// Enqueues the whole proof on the device and starts the host-side assembly on a worker thread; returns once h's MSM is
// on the queue (the scalar sorts read two words back, so this call spans most of the proof's device time).
// mats != nullptr: the constraint matrices (CSR over z = x || w) instead of the three evaluation vectors -- cs.evaluate()
// (zkstd/src/r1cs.rs:137-142) then runs on the device as the first step of each transform chain.
// roles: which of the proof's independent parts this context runs (kg_groth16_prove_sharded spreads them over several
// contexts; a single-context proof runs all three).  defer_assembly: leave the host finishes in the job's futures and do not
// start the assembly -- the sharded entry assembles from several jobs.
enum { ROLE_G2 = 1, ROLE_G1W = 2, ROLE_H = 4, ROLE_ALL = 7 };
static bool g16_h_early() {                               // KG_G16_H_EARLY=0: the blocking proof in the order of the pipelined ones (experiments)
  return tuning().g16_h_early != 0;
}
static bool g16_h_first() {                               // KG_G16_H_EARLY=2: h's point-wise step and coset_idft in front of the G2 accumulation (experiments)
  return tuning().g16_h_early == 2;
}
static bool g16_h_early_pipelined() {                     // KG_G16_H_EARLY_PIPE=0: proofs in flight keep h's chain last (the order up to round 3)
  return tuning().g16_h_early_pipe != 0;
}
int assemble_proof(WorkerPool* workers, const kg_groth16_crs& vk, const uint64_t* rr, const uint64_t* ss, int rc0, ProofJob* j_g2, ProofJob* j_g1w, ProofJob* j_h,
                   uint64_t* proof, uint8_t* inf, Blinding* early = nullptr);
int prove_enqueue(kg_ctx* ctx, const kg_groth16_crs* crs, const uint64_t* d_a_eval, const uint64_t* d_b_eval,
                  const uint64_t* d_c_eval, const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r,
                  const uint64_t* s, ProofJob* job, int slot_base, const kg_csr* const* mats = nullptr, int roles = ROLE_ALL,
                  bool defer_assembly = false, bool h_early = false, bool alone_front = false) {
  if (!ctx || !crs || !r || !s) return KG_ERR_BAD_ARG;
  const bool do_g2 = (roles & ROLE_G2) != 0, do_g1w = (roles & ROLE_G1W) != 0, do_h = (roles & ROLE_H) != 0;
  const bool need_z = do_g2 || do_g1w || mats != nullptr;      // z = x || w feeds the witness MSMs and cs.evaluate()
  if (need_z && !d_x) return KG_ERR_BAD_ARG;
  if (!mats && do_h && (!d_a_eval || !d_b_eval || !d_c_eval)) return KG_ERR_BAD_ARG;
  if (mats)
    for (int v = 0; v < 3; ++v)
      if (!mats[v] || !mats[v]->d_row_ptr || !mats[v]->d_col || !mats[v]->d_val) return KG_ERR_BAD_ARG;
  const size_t m = crs->m, l = crs->l, m_l_1 = crs->m_l_1;
  if (m < 1 || l < 1 || (need_z && m_l_1 && !d_w)) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  if (roles == ROLE_ALL && !defer_assembly) {             // the blinding terms first: three host chains that need nothing from the device
    job->blind.wait();
    job->blind.started = false;
    if (ctx->tune.g16_blind_early && !(crs->delta_g1_inf || crs->delta_g2_inf)) start_blinding(&pool(ctx), *crs, r, s, &job->blind);
  }
  uint32_t k = 0;
  size_t n = 1;
  while (n < m) { n <<= 1; ++k; }                       // cs.m().next_power_of_two() (prover.rs:28-29)
  if (k < 1) { k = 1; n = 2; }
  if (k > 28) return KG_ERR_BAD_ARG;
  hipStream_t st = ctx->stream;
  // polynomial buffers of this ticket: a, b, c (n each), z = x || w, three transform scratch vectors.  One set per ticket,
  // so the next proof's transforms and witness sort start while this proof is still accumulating.
  const int tk = slot_base >= 10 ? 1 : 0;
  KG_TRY(ensure_ws3(ctx, tk, (6 * n + l + m_l_1) * 32));
  uint64_t* A = (uint64_t*)ctx->ws3[tk];
  uint64_t* B = A + 4 * n;
  uint64_t* C = B + 4 * n;
  uint64_t* Z = C + 4 * n;                                // z = x || w
  uint64_t* TMP = Z + 4 * (l + m_l_1);                    // 3 n elements of transform scratch
  // Queue layout.  The four MSMs against z = x || w need nothing from the transforms, and the transforms are short,
  // latency-bound launches (2^18 points = 128..512 workgroups): the witness sort goes out first (it gates the accumulations),
  // the three idft -> coset_dft chains (prover.rs:36-41) run on the two reduction queues beside it and in front of the witness MSMs
  // of the main queue, and h's MSM -- the only consumer of the transforms -- goes last.
  // The order since round 4 (KG_G16_H_EARLY = 1, the default): h's point-wise step and coset_idft run BETWEEN the G2 accumulation and
  // the fused G1 accumulation, h's sort beside the latter, h's accumulation right behind it.  (Round 2 measured the same idea with the
  // 160-VGPR reductions of the time and dropped it: the reductions of a, b_g1 and l lost the gap they ran in, 3.28 -> 3.40 ms
  // blocking; with reductions that fit beside an accumulation it wins -- EXPERIMENTS.md Part I section 11.  KG_G16_H_EARLY = 0 is the old
  // order, = 2 puts h's chain in front of the G2 accumulation.)
  if (!ctx->side_stream) KG_TRY(make_side_stream(ctx));
  if (!ctx->ev_fork) {
    KG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) KG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
  }
  if (do_h) {
    KG_TRY(ntt_prepare(ctx, k, 0));
    KG_TRY(ntt_prepare(ctx, k, 1));
  }
  const size_t hn = (m - 1) < n ? (m - 1) : n;
  const size_t nz = l + m_l_1;
  hipStream_t sq;                                         // z is assembled on the scalar queue (its only reader is the sort)
  KG_TRY(scalar_queue(ctx, &sq));
  // stream semantics: the inputs may still be in flight on the main queue (and the twiddle tables are built there on first
  // use); with complete inputs everything starts at once -- this ticket's buffers are free since its previous proof was collected
  const bool fork = !ctx->inputs_complete || ctx->tw_fresh;
  ctx->tw_fresh = false;
  if (fork) {
    KG_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
    KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->ev_fork, 0));
  }
  // The eight MSMs of prover.rs:51-65 as five: a_inputs + a_aux (:58-59,80) is one MSM of a[..] against
  // z = x || w, likewise b_g1 (:61-62,85) and b_g2 (:64-65,86) -- the sums are what the proof uses.  The four MSMs
  // that meet the witness share ONE scalar-side sort (l's bases start l entries into z); it is enqueued before the
  // transform chains and its two result words are awaited after them.
  if (need_z) KG_HIP(ctx, hipMemcpyAsync(Z, d_x, l * 32, hipMemcpyDeviceToDevice, sq));
  if (need_z && m_l_1) KG_HIP(ctx, hipMemcpyAsync(Z + 4 * l, d_w, m_l_1 * 32, hipMemcpyDeviceToDevice, sq));
  if (mats && do_h) {
    if (m >= ((size_t)1 << 32)) return set_err(ctx, KG_ERR_BAD_ARG, "more than 2^32 constraints");
    KG_TRY(ensure_ws_vec(ctx, 3 * (m + 16) * 4));
    KG_HIP(ctx, hipEventRecord(ctx->ev_order, sq));       // z is complete: the matrix-vector products of the chains wait for it
  }
  MsmSorted Sz;
  // window tables on all four vectors that meet z (kg_bases_precompute): one merged sort, one set of buckets for all windows
  const bool tz = (!do_g2 || has_window_table(ctx, KG_G2, crs->d_b_g2, crs->d_b_g2_inf, nz, nz)) &&
                  (!do_g1w || (has_window_table(ctx, KG_G1, crs->d_a, crs->d_a_inf, nz, nz) && has_window_table(ctx, KG_G1, crs->d_b_g1, crs->d_b_g1_inf, nz, nz) &&
                               (!m_l_1 || has_window_table(ctx, KG_G1, crs->d_l, crs->d_l_inf, m_l_1, nz))));
  // Short proofs (every MSM within the short-input kernel's reach, msm_small.hip; no window tables at these lengths): no sort, no
  // read-back -- each of the five MSMs is one launch (two from 1537 pairs) on a queue of its own behind z, h's behind its transform chain
  int sc2 = 0, sr2 = 0, sc1 = 0, sr1 = 0, scl = 0, srl = 0, sch = 0, srh = 0;
  // A blocking proof gains at every length the kernel takes (2^12 constraints 0.80 ms against 1.38, 2^14 1.33 against 1.49), and so do proofs
  // in flight: proof i + 1 starts beside proof i (inputs complete at the call: 2^12 0.57 ms per proof against 1.17, 2^14 1.11 against 1.26; ordered
  // behind the caller's queue, with h's chain on a queue of the library's own -- see hq below: 2^12 0.69 against 1.18, 2^14 1.19 against 1.37)
  const size_t small_cap = (size_t)ctx->tune.small_max;
  // The halved scalars (GLV, msm_digits.h) shorten the host chains and lengthen the kernels: a single blocking MSM gains at every length, but a
  // proof's five launches run at once and the G1 ones are bound by their kernels -- 2^10 constraints 0.575 -> 0.64 ms with them, 2^12 0.76 -> 0.88;
  // up to 256 witness entries the chains are the critical path (2^4 .. 2^7 constraints: 0.44 -> 0.35 ms).  b_g2 keeps them (its Fq2 host chain is a
  // proof's long pole: 0.19 -> 0.10 ms) up to KG_G16_G2_GLV_MAX = 1100 entries, in fewer and longer workgroups than a lone MSM would take (r = 2: 64
  // workgroups instead of 256 -- the other four launches need CUs too): 2^8 constraints 0.47 -> 0.41 ms, 2^9 0.51 -> 0.44, 2^10 0.57 -> 0.53.  Beyond,
  // such a workgroup's list outgrows LDS (2^12 constraints: 0.78 -> 1.03 ms) and the lone MSM's shape fills the chip (0.78 -> 0.86): off
  // (h's MSM, the last launch of a proof, with halved scalars: level at 2^8 .. 2^12 -- measured, not kept)
  struct GlvGuard { kg_ctx* c; unsigned old; ~GlvGuard() { c->small_glv_off = old; } } glv_guard{ctx, ctx->small_glv_off};
  const bool proof_glv = nz <= 256 && hn <= 256;
  if (!proof_glv) ctx->small_glv_off |= (1u << KG_G1) | (nz > (size_t)ctx->tune.g16_g2_glv_max ? (1u << KG_G2) : 0u);
  const bool small = !tz && nz <= small_cap && hn <= small_cap && (!do_g2 || msm_small_plan(ctx, KG_G2, nz, &sc2, &sr2)) && (!do_g1w || msm_small_plan(ctx, KG_G1, nz, &sc1, &sr1)) &&
                     (!do_g1w || !m_l_1 || msm_small_plan(ctx, KG_G1, m_l_1, &scl, &srl)) && (!(do_h && hn) || msm_small_plan(ctx, KG_G1, hn, &sch, &srh));
  if (small && (do_g2 || do_g1w) && !(mats && do_h)) KG_HIP(ctx, hipEventRecord(ctx->ev_order, sq));      // z is complete
  if ((do_g2 || do_g1w) && !small) {
    ctx->sort_alone = alone_front;                        // a blocking proof: the witness sort has the chip (the transforms beside it are few workgroups)
    const int rs = msm_sort(ctx, KG_FR, Z, nz, &Sz, true, tz ? merged_window(ctx, nz) : 0, 2, false);
    ctx->sort_alone = false;
    KG_TRY(rs);
  }
  int rc = KG_OK;
  auto hip_rc = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == KG_OK) rc = set_err(ctx, e == hipErrorOutOfMemory ? KG_ERR_OOM : KG_ERR_HIP, what, e);
  };
  uint64_t *q_p = job->q_p, *l_p = job->l_p, *ai = job->ai, *b1i = job->b1i, *b2i = job->b2i;
  std::future<int>&f_q = job->f_q, &f_l = job->f_l, &f_a = job->f_a, &f_b1 = job->f_b1, &f_b2 = job->f_b2;
  auto finish_async = [&](int curve, int slot, uint64_t* out) {
    return pool(ctx).submit([ctx, curve, slot, out] { return msm_finish(ctx, curve, slot, out); });
  };
  // result slots: consecutive MSMs alternate between the two reduction queues (slot parity), each with run space of
  // its own (slot mod 8); measured against giving G2's long reduction a queue of its own: 3.67 vs 3.84 ms per proof
  const int SL[5] = {slot_base + 1, slot_base + 2, slot_base + 3, slot_base + 4, slot_base + 5};
  // (from the first finish_async on host finishes may be running on worker threads: no early return below -- failures travel through rc)
  if (small && do_g2) {
    // a short proof's long pole, issued first: b_g2's one-launch MSM (Fq2 on the device) on the scalar queue right behind z, its host
    // chain (255 doublings in Fq2, ~170 us -- three G1 chains) on a worker thread while this thread is still enqueuing the transforms
    if (!proof_glv && msm_small_glv(ctx, KG_G2, nz) && sc2 == 4 && sr2 < 2) sr2 = 2;
    rc = msm_small_enqueue(ctx, sq, KG_G2, crs->d_b_g2, crs->d_b_g2_inf, Z, nz, SL[0], sc2, sr2);
    if (rc == KG_OK) f_b2 = finish_async(KG_G2, SL[0], b2i);
  }
  const uint64_t* src[3] = {d_a_eval, d_b_eval, d_c_eval};
  uint64_t* dst[3] = {A, B, C};
  // the chains share the two reduction queues (queues of their own: 3.31 ms per proof against 2.83 -- transforms and halving
  // levels do not fit the chip together with an accumulation anyway, and in one queue they do not fight each other for it)
  hipStream_t lanes[3] = {ctx->side_stream, ctx->side2_stream, ctx->side_stream};
  for (int v = 0; v < 3 && rc == KG_OK && do_h; ++v) {    // prepare_fft zero padding, then idft + coset_dft
    hipStream_t sv = lanes[v];
    uint64_t* tmp = TMP + (size_t)v * 4 * n;
    if (fork) hip_rc(hipStreamWaitEvent(sv, ctx->ev_fork, 0), "hipStreamWaitEvent(fork)");
    if (mats) {                                           // row v of cs.evaluate(): M_v z, z as the scalar queue assembled it
      hip_rc(hipStreamWaitEvent(sv, ctx->ev_order, 0), "hipStreamWaitEvent(z)");
      if (rc == KG_OK) rc = r1cs_prod_enqueue(ctx, sv, KG_FR, mats[v]->d_row_ptr, mats[v]->d_col, mats[v]->d_val, m, Z, dst[v],
                                              (uint32_t*)ctx->ws_vec + (size_t)v * (m + 16));
    } else
      hip_rc(hipMemcpyAsync(dst[v], src[v], m * 32, hipMemcpyDeviceToDevice, sv), "hipMemcpyAsync(evaluations)");
    if (n > m) hip_rc(hipMemsetAsync(dst[v] + 4 * m, 0, (n - m) * 32, sv), "hipMemsetAsync(padding)");
    if (rc == KG_OK) rc = ntt_enqueue(ctx, sv, tmp, dst[v], k, 1, 0);
    if (rc == KG_OK) rc = ntt_enqueue(ctx, sv, tmp, dst[v], k, 0, 1);
    hip_rc(hipEventRecord(ctx->ev_join[v], sv), "hipEventRecord(join)");
  }
  // q keeps its trailing zeros: zero scalars are skipped by the MSM, which is what Coefficients::new's trimming plus zip
  // achieves in the reference (poly.rs:61-63, msm.rs:25).  Each MSM's 255-step host finish runs on a worker thread while the
  // device continues with the next one.
  // Order of the main queue (h_early, the default since round 4): G2 accumulation, h's point-wise step and coset_idft, the fused G1
  // accumulation with h's sort beside it, h's accumulation directly behind.  Up to round 3 h's whole chain went LAST (the point-wise
  // step, coset_idft, h's sort, h's MSM: ~1.4 ms in a row behind the G1 accumulation, on a chip that only the reductions used)
  // because a service queue that happened to share the main queue's compute pipe started its work ~0.7 ms late and the reductions
  // did not fit beside an accumulation; with the queues placed (capi.cpp place_queues) and the reduction kernels at <= 96 VGPRs
  // neither holds.  Measured: blocking proof 3.34 -> 3.18 ms, two in flight 2.91 -> 2.86 ms (same box, alternating runs).
  MsmSorted Sq;
  bool h_sorted = false;
  // The queue of h's point-wise step, coset_idft and MSM: the main queue -- except for a SHORT proof in flight ordered behind the caller's queue
  // (no kg_ctx_set_inputs_complete): with h's chain on the main queue the next proof's fork event would sit behind it and the device would run
  // the proofs one after the other (2^10 constraints: 0.75 ms per proof against 0.59 blocking; now 0.48); on a queue of the library's own, behind the
  // fork event like every other part of the proof, the caller's queue carries nothing of ours and proof i + 1 starts beside proof i.
  hipStream_t hq = st;
  if (small && fork && !alone_front && do_h) {
    if (!ctx->acc_stream[0]) hip_rc(create_stream(ctx, &ctx->acc_stream[0], false), "queue creation");
    if (ctx->acc_stream[0]) { hq = ctx->acc_stream[0]; hip_rc(hipStreamWaitEvent(hq, ctx->ev_fork, 0), "hipStreamWaitEvent(fork)"); }
  }
  auto h_front = [&]() {                                  // h = (a o b - c) / Z on the coset, back to coefficients (prover.rs:43-47)
    for (int v = 1; v < 3; ++v) hip_rc(hipStreamWaitEvent(hq, ctx->ev_join[v], 0), "hipStreamWaitEvent(join)");   // chain 0 shares chain 2's queue, in front of it
    HostFr seven = HostFr::one();                         // (7^n - 1)^-1 on the host: n = 2^k squarings of 7
    {
      HostFr one = HostFr::one(), acc = HostFr::zero();
      for (int i = 0; i < 7; ++i) acc = add(acc, one);
      seven = acc;
    }
    HostFr z = seven;
    for (uint32_t i = 0; i < k; ++i) z = sqr(z);
    z = inv(sub<4, 1>(z, HostFr::one()));                 // z_on_coset().invert() (fft.rs:141-151)
    Words8 zw;
    for (int i = 0; i < 4; ++i) { zw.w[2 * i] = (uint32_t)z.v[i]; zw.w[2 * i + 1] = (uint32_t)(z.v[i] >> 32); }
    hipLaunchKernelGGL(k_qap_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hq, A, B, C, n, zw);
    hip_rc(hipGetLastError(), "k_qap_combine launch");
    if (rc == KG_OK) rc = ntt_enqueue(ctx, hq, TMP, A, k, 1, 1);             // coset_idft (prover.rs:47)
  };
  auto h_sort = [&](bool wait) {                           // h's coefficients come off the main queue
    hip_rc(hipEventRecord(ctx->ev_order, st), "hipEventRecord(order)");
    hip_rc(hipStreamWaitEvent(sq, ctx->ev_order, 0), "hipStreamWaitEvent(order)");
    const bool th = has_window_table(ctx, KG_G1, crs->d_h, crs->d_h_inf, hn, hn);
    if (rc == KG_OK) rc = msm_sort(ctx, KG_FR, A, hn, &Sq, true, th ? merged_window(ctx, hn) : 0, 1, wait, 1);
    h_sorted = rc == KG_OK;
  };
  const bool early = h_early && do_h && hn && (do_g2 || do_g1w) && !small;
  bool h_out = false;
  if (small && do_h && nz > (size_t)ctx->tune.g16_small_h_first) {
    // The order in which this thread issues a short proof's ~35 launches (5 us each) decides what starts when.  h's chain -- transforms,
    // point-wise step, coset_idft, h's MSM -- is the longest chain of dependent launches: from 2^13 constraints all of it goes out before the
    // three G1 witness MSMs, which need nothing but z (2^14 constraints: 1.11 -> 1.05 ms blocking).  Shorter proofs keep the witness MSMs
    // first: their host finishes feed the assembly's s A + r B1 chain, which is the critical path there (2^4: 0.34 -> 0.37 ms with h first).
    h_front();
    if (rc == KG_OK && hn) {
      rc = msm_small_enqueue(ctx, hq, KG_G1, crs->d_h, crs->d_h_inf, A, hn, SL[4], sch, srh);
      if (rc == KG_OK) f_q = finish_async(KG_G1, SL[4], q_p);
    } else if (!hn) msm_identity(KG_G1, q_p);
    h_out = true;
  }
  if (small && (do_g2 || do_g1w)) {
    // (b_g2 went out in front of the transform chains); a, b_g1 and l on queues of their own
    if (do_g1w) {
      struct { const uint64_t* b; const uint8_t* inf; const uint64_t* sc; size_t n; int slot, c, r; uint64_t* out; std::future<int>* f; } q3[3] = {
          {crs->d_a, crs->d_a_inf, Z, nz, SL[1], sc1, sr1, ai, &f_a}, {crs->d_b_g1, crs->d_b_g1_inf, Z, nz, SL[2], sc1, sr1, b1i, &f_b1},
          {crs->d_l, crs->d_l_inf, Z + 4 * l, m_l_1, SL[3], scl, srl, l_p, &f_l}};
      for (int k = 0; k < 3 && rc == KG_OK; ++k) {
        if (q3[k].n == 0) { msm_identity(KG_G1, q3[k].out); continue; }
        hipStream_t qk = ctx->acc_stream[1 + k];
        if (!qk) { hip_rc(create_stream(ctx, &ctx->acc_stream[1 + k], false), "queue creation"); qk = ctx->acc_stream[1 + k]; }
        if (rc != KG_OK) break;
        hip_rc(hipStreamWaitEvent(qk, ctx->ev_order, 0), "hipStreamWaitEvent(z)");
        if (rc == KG_OK) rc = msm_small_enqueue(ctx, qk, KG_G1, q3[k].b, q3[k].inf, q3[k].sc, q3[k].n, q3[k].slot, q3[k].c, q3[k].r);
        if (rc == KG_OK) *q3[k].f = finish_async(KG_G1, q3[k].slot, q3[k].out);
      }
    } else msm_identity(KG_G1, l_p);
  } else
  if (do_g2 || do_g1w) {
    const int rw = msm_sort_wait(ctx, &Sz);                // always: the next sort may not start before this read-back
    if (rc == KG_OK) rc = rw;
    // G2 first: its host finish (Fq2 arithmetic, ~3x a G1 finish) and its slow reduction then overlap the G1 accumulations
    if (early && g16_h_first() && rc == KG_OK) { h_front(); h_sort(false); }
    if (rc == KG_OK && do_g2) rc = msm_run(ctx, Sz, KG_G2, crs->d_b_g2, crs->d_b_g2_inf, nz, 0, SL[0]);
    if (rc == KG_OK && do_g2) f_b2 = finish_async(KG_G2, SL[0], b2i);
    if (early && !g16_h_first() && rc == KG_OK) { h_front(); h_sort(false); }
    // the three G1 queries against z (a, b_g1, l) are accumulated by ONE launch: 13 000 waves instead of three
    // one-round launches of 4 352 (see k_acc_tasks)
    if (rc == KG_OK && do_g1w) {
      MsmRunJob jobs3[3] = {{crs->d_a, crs->d_a_inf, nz, 0u, SL[1]}, {crs->d_b_g1, crs->d_b_g1_inf, nz, 0u, SL[2]},
                            {crs->d_l, crs->d_l_inf, m_l_1, (uint32_t)l, SL[3]}};
      rc = msm_run_multi(ctx, Sz, KG_G1, jobs3, m_l_1 ? 3 : 2);
    }
    if (rc == KG_OK && do_g1w) f_a = finish_async(KG_G1, SL[1], ai);
    if (rc == KG_OK && do_g1w) f_b1 = finish_async(KG_G1, SL[2], b1i);
    if (rc == KG_OK && do_g1w && m_l_1) f_l = finish_async(KG_G1, SL[3], l_p);
    else msm_identity(KG_G1, l_p);
  }
  // From here on host finishes may already be running on worker threads: no early return -- every failure travels
  // through rc into the assembly task below, which joins all of them before it reports.
  // (Measured and dropped: h's chain on a service queue under the witness accumulations -- the transform's workgroups only
  // get onto a CU once an accumulation has drained, and then queue behind the reductions: 3.3 -> 3.85 ms per proof.)
  if (h_out) {
  } else {
  if (do_h && !early) h_front();
  if (rc == KG_OK && hn && do_h && small) {               // h's coefficients come off the main queue: its MSM right behind them
    rc = msm_small_enqueue(ctx, hq, KG_G1, crs->d_h, crs->d_h_inf, A, hn, SL[4], sch, srh);
    if (rc == KG_OK) f_q = finish_async(KG_G1, SL[4], q_p);
  } else
  if (rc == KG_OK && hn && do_h) {
    if (!h_sorted) h_sort(true);
    else { const int rw = msm_sort_wait(ctx, &Sq); if (rc == KG_OK) rc = rw; }
    // a blocking proof: h's reduction follows its accumulation on the main queue (nothing else is coming there) -- the two reduction
    // queues may still hold the witness MSMs' reductions (a 0/1-heavy witness: hot-bucket trees; h's reduction waited 0.7 ms for a queue)
    const bool h_inline = ctx->tune.g16_h_inline != 0;
    if (alone_front && h_inline) Sq.reduce_inline = true;
    if (alone_front) Sq.tail_alone = true;                // a blocking proof: h's reduction is the last thing on the device
    if (rc == KG_OK) rc = msm_run(ctx, Sq, KG_G1, crs->d_h, crs->d_h_inf, hn, 0, SL[4]);
    if (rc == KG_OK) f_q = finish_async(KG_G1, SL[4], q_p);
  } else msm_identity(KG_G1, q_p);
  }
  // Host side of the proof on a worker thread: first the parts of the assembly (prover.rs:75-77) that depend on no MSM
  // result, then A, B and C up to h's term as the witness MSMs' host finishes arrive, then h's term.
  for (int i = 0; i < 4; ++i) { job->rr[i] = r[i]; job->ss[i] = s[i]; }
  job->rc0 = rc;
  job->active = true;
  if (defer_assembly) return KG_OK;
  const kg_groth16_crs vk = *crs;                       // the host-resident part (alpha, beta, delta) is read by value
  WorkerPool* wp = &pool(ctx);
  job->assembly = wp->submit([job, vk, wp]() -> int { return assemble_proof(wp, vk, job->rr, job->ss, job->rc0, job, job, job, job->proof, job->inf, &job->blind); });
  return KG_OK;
}

void start_blinding(WorkerPool* workers, const kg_groth16_crs& vk, const uint64_t* rr, const uint64_t* ss, Blinding* bl) {
  bl->wait();                                            // (tasks of a proof that failed while it was enqueued)
  const HostFr rm = HostFr::from_words(rr), sm = HostFr::from_words(ss), raw_one{{1, 0, 0, 0}};
  const HostFr rk = mul(rm, raw_one), sk = mul(sm, raw_one), rsk = mul(mul(rm, sm), raw_one);   // out of Montgomery form
  bl->rk = rk; bl->sk = sk;
  bl->started = true;
  const XYZZ<HostFq> alpha = from_affine(h_load_aff<HostFq, 4>(vk.alpha_g1)), beta1 = from_affine(h_load_aff<HostFq, 4>(vk.beta_g1)),
                     delta1 = from_affine(h_load_aff<HostFq, 4>(vk.delta_g1));
  const XYZZ<HostFq2> beta2 = from_affine(h_load_aff<HostFq2, 8>(vk.beta_g2)), delta2 = from_affine(h_load_aff<HostFq2, 8>(vk.delta_g2));
  auto chain_b = [bl, delta2, beta2, sk]() -> int { bl->g_b = add_xyzz(h_scalar_mul(delta2, sk.v), beta2); return KG_OK; };              // :76
  auto chain_a = [bl, delta1, alpha, rk, rsk]() -> int {                                                                                  // :75, first term of :77
    bl->g_a = add_xyzz(h_scalar_mul(delta1, rk.v), alpha);
    bl->rs_delta = h_scalar_mul(delta1, rsk.v);
    return KG_OK;
  };
  auto chain_c = [bl, alpha, beta1, sk, rk]() -> int { bl->sa_rb = h_scalar_mul2(alpha, sk.v, beta1, rk.v); return KG_OK; };             // s alpha + r beta1 (:77)
  // the longest first; a chain no thread is to be had for runs here, in a row
  int on_pool = 0;
  if (workers) {
    try { bl->t[0] = workers->submit(chain_b); ++on_pool; bl->t[1] = workers->submit(chain_a); ++on_pool; bl->t[2] = workers->submit(chain_c); ++on_pool; } catch (...) {}
  }
  if (on_pool < 1) chain_b();
  if (on_pool < 2) chain_a();
  if (on_pool < 3) chain_c();
}

// prover.rs:75-92 on the host.  The five MSM sums may come from up to three jobs (one per context of a sharded proof):
// j_g2 holds b2i, j_g1w holds ai / b1i / l_p, j_h holds q_p, each with the futures of its host finishes.  early: the blinding terms a
// single-context proof started when it was enqueued; a sharded proof starts them here.
int assemble_proof(WorkerPool* workers, const kg_groth16_crs& vk, const uint64_t* rr, const uint64_t* ss, int rc0, ProofJob* j_g2, ProofJob* j_g1w, ProofJob* j_h,
                   uint64_t* proof, uint8_t* inf, Blinding* early) {
  int rc = rc0;
  uint64_t *q_p = j_h->q_p, *l_p = j_g1w->l_p, *ai = j_g1w->ai, *b1i = j_g1w->b1i, *b2i = j_g2->b2i;
  const bool bad_delta = vk.delta_g1_inf || vk.delta_g2_inf;
  Blinding local;
  Blinding* bl = early && early->started ? early : &local;
  if (rc == KG_OK && !bad_delta && !bl->started) start_blinding(workers, vk, rr, ss, bl);
  XYZZ<HostFq> g_a = XYZZ<HostFq>::identity(), g_c = g_a;
  XYZZ<HostFq2> g_b = XYZZ<HostFq2>::identity();
  auto join = [&](std::future<int>& f) { if (f.valid()) { int r2 = f.get(); if (rc == KG_OK) rc = r2; } };
  auto g1pt = [](const uint64_t* xyz) {
    bool pinf = !(xyz[8] | xyz[9] | xyz[10] | xyz[11]);
    return h_from_abi<HostFq>(h_load_aff<HostFq, 4>(xyz), pinf);
  };
  auto g2pt = [](const uint64_t* xyz) {
    bool pinf = true;
    for (int i = 16; i < 24; ++i) pinf = pinf && xyz[i] == 0;
    return h_from_abi<HostFq2>(h_load_aff<HostFq2, 8>(xyz), pinf);
  };
  // everything that does not need h's MSM is assembled while the device is still working on it -- s A + r B1 (another 255-step chain) as soon
  // as the two G1 sums are there, under the G2 MSM's host finish (three times a G1 one: the last of the four to arrive)
  join(j_g1w->f_a); join(j_g1w->f_b1);
  host_trace("proof: a, b1 in");
  XYZZ<HostFq> sa_rb1 = XYZZ<HostFq>::identity();
  const bool go = rc == KG_OK && !bad_delta;
  if (go) sa_rb1 = h_scalar_mul2(g1pt(ai), bl->sk.v, g1pt(b1i), bl->rk.v);                               // s * a_answer + r * b1_answer (:83,90), under the blinding chains
  host_trace("proof: sA + rB1");
  bl->wait();
  host_trace("proof: chains done");
  if (go) {
    g_a = add_xyzz(bl->g_a, g1pt(ai));                                                                   // :81
    h_store_affine<HostFq, 4>(g_a, proof, inf);
    g_b = bl->g_b;
    g_c = add_xyzz(bl->rs_delta, bl->sa_rb);                                                             // :77
  }
  join(j_g1w->f_l); join(j_g2->f_b2);
  host_trace("proof: l, b2 in");
  if (rc == KG_OK && !bad_delta) {
    g_b = add_xyzz(g_b, g2pt(b2i));                                                                      // :88
    g_c = add_xyzz(g_c, sa_rb1);
    g_c = add_xyzz(g_c, g1pt(l_p));                                                                      // :92 (l part)
    h_store_affine<HostFq2, 8>(g_b, proof + 8, inf + 1);
  }
  host_trace("proof: B stored");
  join(j_h->f_q);
  host_trace("proof: h in");
  if (rc != KG_OK) return rc;
  if (bad_delta) return KG_ERR_CRS;                     // prover.rs:67-69 (the message is set by prove_collect, on the caller's thread)
  g_c = add_xyzz(g_c, g1pt(q_p));                                                                        // :92 (h part)
  h_store_affine<HostFq, 4>(g_c, proof + 24, inf + 2);
  host_trace("proof: assembled");
  return KG_OK;
}

int prove_collect(kg_ctx* ctx, ProofJob* job, uint64_t* proof_out, uint8_t* proof_inf) {
  if (!job->active) return KG_ERR_BAD_ARG;
  job->active = false;
  const int rc = job->assembly.get();
  if (rc == KG_ERR_CRS) return set_err(ctx, KG_ERR_CRS, "delta is the identity");
  if (rc != KG_OK) return rc;
  for (int i = 0; i < 32; ++i) proof_out[i] = job->proof[i];
  for (int i = 0; i < 3; ++i) proof_inf[i] = job->inf[i];
  return KG_OK;
}
}  // namespace

extern "C" {

int kg_groth16_prove_bn254(kg_ctx* ctx, const kg_groth16_crs* crs, const uint64_t* d_a_eval, const uint64_t* d_b_eval,
                           const uint64_t* d_c_eval, const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r,
                           const uint64_t* s, uint64_t* proof_out, uint8_t* proof_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !proof_out || !proof_inf) return KG_ERR_BAD_ARG;
  ProofJob* job = job_of(ctx, 0);
  if (job->active) return KG_ERR_BAD_ARG;                // a proof begun with ticket 0 has not been collected
  host_trace("proof: enter");
  KG_TRY(prove_enqueue(ctx, crs, d_a_eval, d_b_eval, d_c_eval, d_x, d_w, r, s, job, 5, nullptr, ROLE_ALL, false, g16_h_early(), true));
  host_trace("proof: enqueued");
  return prove_collect(ctx, job, proof_out, proof_inf);
  });
}

// Two proofs in flight (tickets 0 and 1): begin(i + 1) may be called before end(i), so that the next proof's transforms
// and sorts start while the previous proof's last reduction, host finish and assembly are still running.  crs and the
// device inputs must stay valid until the matching end.
int kg_groth16_prove_begin(kg_ctx* ctx, const kg_groth16_crs* crs, const uint64_t* d_a_eval, const uint64_t* d_b_eval,
                           const uint64_t* d_c_eval, const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r,
                           const uint64_t* s, int ticket) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || ticket < 0 || ticket > 1 || job_of(ctx, ticket)->active) return KG_ERR_BAD_ARG;
  return prove_enqueue(ctx, crs, d_a_eval, d_b_eval, d_c_eval, d_x, d_w, r, s, job_of(ctx, ticket), 5 + 5 * ticket, nullptr, ROLE_ALL, false, g16_h_early_pipelined());
  });
}
// The same with cs.evaluate() on the device: the constraint matrices (resident CSR) instead of the evaluation vectors
int kg_groth16_prove_r1cs_bn254(kg_ctx* ctx, const kg_groth16_crs* crs, const kg_csr* a, const kg_csr* b, const kg_csr* c,
                                const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r, const uint64_t* s, uint64_t* proof_out,
                                uint8_t* proof_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !proof_out || !proof_inf) return KG_ERR_BAD_ARG;
  ProofJob* job = job_of(ctx, 0);
  if (job->active) return KG_ERR_BAD_ARG;
  const kg_csr* mats[3] = {a, b, c};
  KG_TRY(prove_enqueue(ctx, crs, nullptr, nullptr, nullptr, d_x, d_w, r, s, job, 5, mats, ROLE_ALL, false, g16_h_early(), true));
  return prove_collect(ctx, job, proof_out, proof_inf);
  });
}
int kg_groth16_prove_r1cs_begin(kg_ctx* ctx, const kg_groth16_crs* crs, const kg_csr* a, const kg_csr* b, const kg_csr* c,
                                const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r, const uint64_t* s, int ticket) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || ticket < 0 || ticket > 1 || job_of(ctx, ticket)->active) return KG_ERR_BAD_ARG;
  const kg_csr* mats[3] = {a, b, c};
  return prove_enqueue(ctx, crs, nullptr, nullptr, nullptr, d_x, d_w, r, s, job_of(ctx, ticket), 5 + 5 * ticket, mats, ROLE_ALL, false, g16_h_early_pipelined());
  });
}
int kg_groth16_prove_end(kg_ctx* ctx, int ticket, uint64_t* proof_out, uint8_t* proof_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || ticket < 0 || ticket > 1 || !proof_out || !proof_inf) return KG_ERR_BAD_ARG;
  return prove_collect(ctx, job_of(ctx, ticket), proof_out, proof_inf);
  });
}


// One proof over several contexts, task-parallel (the MSMs of prover.rs:51-65 are independent of each other and
// of the transforms until the assembly): role G2 (the b_g2 query against z -- the long pole), role G1W (the a, b_g1 and l
// queries against z, one fused accumulation) and role H (three idft -> coset_dft chains, h = (a o b - c) / Z, coset_idft,
// h's MSM) go to contexts 0, 1 % n and 2 % n.  Every context enqueues its part from a host thread of its own (a sort reads
// two words back); the five sums meet on the host, where the assembly of prover.rs:75-92 runs once.
int kg_groth16_prove_sharded(kg_ctx* const* ctxs, int n_ctx, const kg_groth16_crs* const* crs, const uint64_t* const* d_a_eval,
                             const uint64_t* const* d_b_eval, const uint64_t* const* d_c_eval, const uint64_t* const* d_x,
                             const uint64_t* const* d_w, const uint64_t* r, const uint64_t* s, uint64_t* proof_out, uint8_t* proof_inf) {
  return kg::kg_guarded((ctxs && n_ctx > 0 ? ctxs[0] : nullptr), [&]() -> int {
  if (!ctxs || n_ctx < 1 || n_ctx > 8 || !crs || !d_x || !d_w || !r || !s || !proof_out || !proof_inf) return KG_ERR_BAD_ARG;   // d_x, d_w: the arrays; entries of contexts that do not read z may be NULL
  for (int i = 0; i < n_ctx; ++i)
    if (!ctxs[i] || !crs[i]) return KG_ERR_BAD_ARG;
  for (int i = 0; i < n_ctx; ++i)
    for (int j = 0; j < i; ++j)
      if (ctxs[i] == ctxs[j]) return set_err(ctxs[0], KG_ERR_BAD_ARG, "a context is listed twice");
  const int owner[3] = {0, 1 % n_ctx, 2 % n_ctx};         // of ROLE_G2, ROLE_G1W, ROLE_H
  int roles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  roles[owner[0]] |= ROLE_G2; roles[owner[1]] |= ROLE_G1W; roles[owner[2]] |= ROLE_H;
  for (int i = 0; i < n_ctx; ++i) {
    if (!roles[i]) continue;
    if (crs[i]->m != crs[0]->m || crs[i]->l != crs[0]->l || crs[i]->m_l_1 != crs[0]->m_l_1) return set_err(ctxs[0], KG_ERR_BAD_ARG, "the contexts' CRS copies differ in shape");
    if (job_of(ctxs[i], 0)->active) return set_err(ctxs[i], KG_ERR_BAD_ARG, "a proof is in flight on this context");
    if ((roles[i] & ROLE_H) && (!d_a_eval || !d_b_eval || !d_c_eval || !d_a_eval[i] || !d_b_eval[i] || !d_c_eval[i])) return KG_ERR_BAD_ARG;
  }
  // one context per device (the multi-GPU case): each has its device to itself for this blocking call -- its first sort is shaped for an
  // idle device and h's reduction follows h's accumulation, as in kg_groth16_prove_bn254; contexts that share a device do not
  bool own_device = true;
  for (int i = 0; i < n_ctx; ++i)
    for (int j = 0; j < i; ++j)
      if (roles[i] && roles[j] && ctxs[i]->device == ctxs[j]->device) own_device = false;
  std::vector<std::future<int>> enq;
  std::vector<int> who;
  for (int i = 0; i < n_ctx; ++i) {
    if (!roles[i]) continue;
    who.push_back(i);
    enq.push_back(pool(ctxs[0]).submit([=]() -> int {
      return prove_enqueue(ctxs[i], crs[i], d_a_eval ? d_a_eval[i] : nullptr, d_b_eval ? d_b_eval[i] : nullptr, d_c_eval ? d_c_eval[i] : nullptr,
                           d_x[i], d_w[i], r, s, job_of(ctxs[i], 0), 5, nullptr, roles[i], true, false, own_device);
    }));
  }
  int rc = KG_OK;
  for (size_t t = 0; t < enq.size(); ++t) {
    const int r2 = enq[t].get();
    if (rc == KG_OK) rc = r2;
  }
  ProofJob* jg2 = job_of(ctxs[owner[0]], 0);
  ProofJob* jg1 = job_of(ctxs[owner[1]], 0);
  ProofJob* jh = job_of(ctxs[owner[2]], 0);
  for (int i : who) {                                     // a failed enqueue may have left host finishes running on the others: the
    ProofJob* j = job_of(ctxs[i], 0);                     // assembly joins every future before it reports
    if (rc == KG_OK && j->rc0 != KG_OK) rc = j->rc0;
  }
  uint64_t proof[32];
  uint8_t inf[3] = {0, 0, 0};
  rc = assemble_proof(&pool(ctxs[0]), *crs[0], r, s, rc, jg2, jg1, jh, proof, inf);
  for (int i : who) {
    ProofJob* j = job_of(ctxs[i], 0);
    for (std::future<int>* f : {&j->f_q, &j->f_l, &j->f_a, &j->f_b1, &j->f_b2})
      if (f->valid()) f->get();
    j->active = false;
  }
  if (rc == KG_ERR_CRS) return set_err(ctxs[0], KG_ERR_CRS, "delta is the identity");
  if (rc != KG_OK) return rc;
  for (int i = 0; i < 32; ++i) proof_out[i] = proof[i];
  for (int i = 0; i < 3; ++i) proof_inf[i] = inf[i];
  return KG_OK;
  });
}

}  // extern "C"
