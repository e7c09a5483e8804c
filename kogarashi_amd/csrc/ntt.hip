// ntt.hip -- radix-2 Fr NTT on gfx950, natural order in and out.
//
// Replaces groth16/src/fft.rs: Fft::new (:27-89: the reference re-materialises n/2 + n/2 + n + n table
// entries per proof), prepare_fft (:157-162, bit-reversal swaps), classic_fft_arithmetic (:166-192, recursive
// DIT) and butterfly_arithmetic (:195-218).  dft(v)[i] = sum_j v[j] w^(ij) with w = ROOT_OF_UNITY^(2^(28-k)).
//
// Decomposition n = n1*n2*n3 (ntt_tile.h: ntt_plan): one step (n <= 2^11), two (n <= 2^21: factors up to 2^11, two HBM
// round trips) or three; no separate bit-reversal pass exists and every step is one HBM round trip.  A step works on
// tiles of 1024-4096 elements that live in LDS as nine 29-bit limb planes (36-144 KiB); its kernel is instantiated per
// (DFT size, tile width), so index arithmetic is constant shifts, the first register pass is fused with the load and the
// last with the store (ntt_tile.h).  Data never leaves the caller's Montgomery domain: the transform is linear and
// twiddles are multiplied in as internal-form constants (fp29.h), so there is no domain conversion, only a limb
// re-packing at load/store; intermediate arrays hold values below 2p (not canonical), the last step canonicalises.
// Algorithmic HBM bytes: 64 B/element; this design moves 64 B/element per step.
#include "common.h"
#include "ntt_core.h"
#include "ntt_tile.h"
#include <type_traits>

using namespace kg;

struct kg_tw_cache {
  uint32_t log_n;
  int inverse;
  uint32_t lo_bits;
  uint32_t* small = nullptr;   // w_2048^e, e < 1024, Shoup form [e][18]
  uint32_t* lo = nullptr;      // w_n^e, e < 2^lo_bits        [e][9]
  uint32_t* hi = nullptr;      // w_n^(e << lo_bits)          [e][9]
  uint32_t* cos_lo = nullptr;  // g^(+-e) (g = 7), e < 2^lo_bits, for the coset shift [* n^-1 when inverse]
  uint32_t* cos_hi = nullptr;  // g^(+-(e << lo_bits))
  uint32_t* zinv = nullptr;    // (7^n - 1)^-1, one entry (fft.rs:141-154)
  // inter-step twiddles read instead of composed: direct[s][r * inner + c] = w_n^(r * c * mult) for step s (0: A, 1: B),
  // Montgomery form, 36 B per entry.  Step A's table has n entries and exists up to direct_a_max_log() (36 MB at 2^20,
  // 151 MB at 2^22, per direction); step B's (three-step plans) has n2 * n3 <= 2^18 entries.  Without a table the step
  // composes the twiddle from lo / hi: one more product per element.
  uint32_t* direct[2] = {nullptr, nullptr};
};

namespace {

// Largest transform whose step-A inter-step twiddles are read from a direct table of n entries (36 B each: 36 MB at 2^20,
// 151 MB at 2^22, per direction) instead of being composed from the two-level tables (one more product per element).
// Measured (sustained clocks): 93 against 100 us at 2^20 (two steps); 395-399 against 400-410 us for the three-step plan of 2^22 on
// one box, level on another -- the table stays the default there for those 1-2 %.
// KG_NTT_DIRECT_MAX_LOG lowers it for hosts that would rather keep the memory (0 = never).
uint32_t direct_a_max_log() {
  const int x = tuning().ntt_direct_max_log;
  return (uint32_t)(x < 0 ? 0 : (x > 22 ? 22 : x));
}

__device__ __forceinline__ Fr ld_tw(const uint32_t* __restrict__ tab, size_t e) { return NttIO<Fr>::table(tab, e); }
__device__ __forceinline__ void st_tw(uint32_t* tab, size_t e, const Fr& a) {
#pragma unroll
  for (int k = 0; k < 9; ++k) tab[e * 9 + k] = a.l[k];
}
__device__ Fr root_of(uint32_t log, int inverse) { return ntt_root_of<Fr>(log, inverse); }
__device__ Fr pow_u64(Fr base, uint64_t e) { return ntt_pow<Fr>(base, e); }

// kind 0: in-tile table (root of order 2^11, e < 1024); 1: lo; 2: hi; 3: coset lo; 4: coset hi
__global__ void __launch_bounds__(64) k_build_table(int kind, uint32_t log_n, int inverse, uint32_t lo_bits, uint32_t count, uint32_t* __restrict__ tab) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  Fr v;
  if (kind == 0) {                         // Shoup-form constant: the plain canonical root power and its quotient (fp29.h mulc)
    Fr raw_one = Fr::zero();
    raw_one.l[0] = 1;
    const FpConst<FrParams> c = make_const(reduce_2p(mul(pow_u64(root_of(NTT_TW_LOG, inverse), e), raw_one)));
#pragma unroll
    for (int k = 0; k < 9; ++k) { tab[e * 18 + k] = c.w[k]; tab[e * 18 + 9 + k] = c.q[k]; }
    return;
  }
  else if (kind == 1) v = pow_u64(root_of(log_n, inverse), e);
  else if (kind == 2) v = pow_u64(root_of(log_n, inverse), (uint64_t)e << lo_bits);
  else {
    Fr g = Fr::from_const(inverse ? FrParams::GEN7_INV : FrParams::GEN7);   // fft.rs:56,64
    v = pow_u64(g, kind == 3 ? (uint64_t)e : ((uint64_t)e << lo_bits));
    if (kind == 3 && inverse) {            // fold n^-1 (fft.rs:86,104) into the low table (always multiplied in)
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      w[0] = log_n < 32 ? (1u << log_n) : 0;
      Fr nn = from_int<FrParams>(w);
      v = mul(v, inv(nn));
    }
  }
  st_tw(tab, e, v);
}

// direct inter-step table: entry (r, c) = w_n^(r * c * mult), r < 2^log_m, c < 2^log_inner
__global__ void __launch_bounds__(64) k_build_direct(uint32_t log_n, int inverse, uint32_t log_m, uint32_t log_inner, uint64_t mult, uint32_t* __restrict__ tab) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ((size_t)1 << (log_m + log_inner))) return;
  const uint64_t r = e >> log_inner, c = e & (((uint64_t)1 << log_inner) - 1);
  st_tw(tab, e, pow_u64(root_of(log_n, inverse), r * c * mult));
}

#ifndef KG_NTT_WAVES
#define KG_NTT_WAVES 4
#endif
// One tile per workgroup; see ntt_tile.h.  NT = tile / 4 threads (one radix-4 group per lane and pass).
template <int LOG_M, int LOG_TC, bool ROW>
__global__ void __launch_bounds__((NttTile<Fr, LOG_M, LOG_TC, ROW>::NT)) __attribute__((amdgpu_waves_per_eu(KG_NTT_WAVES))) k_ntt_tile(NttStepArgs A) {   // <= 128 VGPRs: LDS admits four waves per SIMD
  KG_SERVICE_PRIO();
  using T = NttTile<Fr, LOG_M, LOG_TC, ROW>;
  extern __shared__ uint32_t lds[];
  uint32_t tile = blockIdx.x;
  if (A.tile_shift) tile = ((tile & 7u) << A.tile_shift) | (tile >> 3);     // workgroup i runs on XCD i mod 8: neighbouring tiles share an L2
  const T t{A, tile};
  const LdsPlanes<T::ELEMS> st{lds};
  t.first(threadIdx.x, st);
  if constexpr (!T::SINGLE) {
#ifdef KG_NTT_EXP_NOBAR      // phase-off experiment (tools/dbg): no workgroup barriers (results are wrong, timing only)
    const auto full = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
#else
    const auto full = [] { __syncthreads(); };
#endif
    // between two wave-private passes: the wave's own LDS writes are ordered before its reads by the in-order LDS queue; the
    // fence keeps the compiler from moving them across the boundary
    const auto wave = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    t.after_first(full, wave);
    t.template mids<T::G0>(threadIdx.x, st, full, wave);
    t.last(threadIdx.x, st);
  }
}

// data[i] *= c   (c: one internal-form constant in device memory)
__global__ void __launch_bounds__(256) k_scale_const(uint64_t* __restrict__ data, size_t n, const uint32_t* __restrict__ c) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words(data, i, w);
  Fr v = mul(limbs_from_words<FrParams>(w), ld_tw(c, 0));
  words_from_limbs(reduce_2p(v), w);
  store_words(data, i, w);
}
// (7^n - 1)^-1 for n = 2^log_n
__global__ void k_build_zinv(uint32_t log_n, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Fr g = Fr::from_const(FrParams::GEN7);
  for (uint32_t i = 0; i < log_n; ++i) g = sqr(g);
  st_tw(out, 0, inv(norm(sub<4, 1>(g, Fr::one()))));
}

int plan_tile_env() {        // KG_NTT_TILE=10..12: log2 of the tile size (experiments)
  return tuning().ntt_tile;
}
int plan_steps_env() {       // KG_NTT_STEPS=3: three-step plans from 2^18 up; 2: two steps up to 2^22 (experiments; default: ntt_plan's automatic rule)
  return tuning().ntt_steps;
}

void free_tables(kg_tw_cache* t) {
  hipFree(t->small); hipFree(t->lo); hipFree(t->hi); hipFree(t->cos_lo); hipFree(t->cos_hi); hipFree(t->zinv);
  hipFree(t->direct[0]); hipFree(t->direct[1]);
  delete t;
}

int get_tables(kg_ctx* ctx, uint32_t log_n, int inverse, kg_tw_cache** out) {
  for (kg_tw_cache* t : ctx->tw)
    if (t->log_n == log_n && t->inverse == inverse) { *out = t; return KG_OK; }
  kg_tw_cache* t = new kg_tw_cache();
  t->log_n = log_n; t->inverse = inverse;
  t->lo_bits = (log_n + 1) / 2;
  const uint32_t n_lo = 1u << t->lo_bits, n_hi = 1u << (log_n - t->lo_bits);
  NttStepDesc d[3];
  const int nsteps = ntt_plan(log_n, plan_steps_env(), d, plan_tile_env());
  // direct inter-step tables: step A w_n^(i1 * c), c < n / n1 (n entries; kept while it is worth its memory: 36 MB at 2^20,
  // 151 MB at 2^22), step B of a three-step plan w_n^(n1 * i2 * j3) (n2 * n3 entries)
  const size_t cnt_a = (nsteps >= 2 && log_n <= direct_a_max_log()) ? (size_t)1 << log_n : 0;
  const size_t cnt_b = nsteps == 3 ? (size_t)1 << (d[1].log_m + d[2].log_m) : 0;
  auto alloc = [&](uint32_t** p, size_t entries) { return entries == 0 || dev_alloc(ctx, (void**)p, entries * 36) == hipSuccess; };
  if (!alloc(&t->small, 2u << (NTT_TW_LOG - 1)) || !alloc(&t->lo, n_lo) || !alloc(&t->hi, n_hi) || !alloc(&t->cos_lo, n_lo) ||
      !alloc(&t->cos_hi, n_hi) || !alloc(&t->zinv, 1)) {
    (void)hipGetLastError();
    free_tables(t);                                    // releases whatever was allocated before the failure
    return set_err(ctx, KG_ERR_OOM, "twiddle table allocation");
  }
  // The direct inter-step tables are an optimisation (36 B per entry: 151 MB per direction at 2^22, kept for the life of the
  // context -- twice that for a forward + inverse pair, per context): when one cannot be allocated the step composes its twiddle
  // from the lo / hi tables instead (one more product per element), it does not fail the transform.
  size_t cnt_a_eff = cnt_a, cnt_b_eff = cnt_b;
  if (!alloc(&t->direct[0], cnt_a)) { (void)hipGetLastError(); t->direct[0] = nullptr; cnt_a_eff = 0; }
  if (!alloc(&t->direct[1], cnt_b)) { (void)hipGetLastError(); t->direct[1] = nullptr; cnt_b_eff = 0; }
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_build_table, dim3((1u << (NTT_TW_LOG - 1)) / 64), dim3(64), 0, st, 0, log_n, inverse, t->lo_bits, 1u << (NTT_TW_LOG - 1), t->small);
  hipLaunchKernelGGL(k_build_table, dim3((n_lo + 63) / 64), dim3(64), 0, st, 1, log_n, inverse, t->lo_bits, n_lo, t->lo);
  hipLaunchKernelGGL(k_build_table, dim3((n_hi + 63) / 64), dim3(64), 0, st, 2, log_n, inverse, t->lo_bits, n_hi, t->hi);
  hipLaunchKernelGGL(k_build_table, dim3((n_lo + 63) / 64), dim3(64), 0, st, 3, log_n, inverse, t->lo_bits, n_lo, t->cos_lo);
  hipLaunchKernelGGL(k_build_table, dim3((n_hi + 63) / 64), dim3(64), 0, st, 4, log_n, inverse, t->lo_bits, n_hi, t->cos_hi);
  hipLaunchKernelGGL(k_build_zinv, dim3(1), dim3(64), 0, st, log_n, t->zinv);
  if (cnt_a_eff)
    hipLaunchKernelGGL(k_build_direct, dim3((unsigned)((cnt_a + 63) / 64)), dim3(64), 0, st, log_n, inverse, (uint32_t)d[0].log_m, log_n - d[0].log_m, (uint64_t)1, t->direct[0]);
  if (cnt_b_eff)
    hipLaunchKernelGGL(k_build_direct, dim3((unsigned)((cnt_b + 63) / 64)), dim3(64), 0, st, log_n, inverse, (uint32_t)d[1].log_m, (uint32_t)d[2].log_m, (uint64_t)1 << d[0].log_m, t->direct[1]);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    free_tables(t);
    return set_err(ctx, KG_ERR_HIP, "twiddle table kernels", le);
  }
  ctx->tw_fresh = true;
  ctx->tw.push_back(t);
  *out = t;
  return KG_OK;
}

// ---- dispatch: one kernel instantiation per tile shape ntt_plan can ask for ------------------------------------
template <int LOG_M, int LOG_TC, bool ROW>
int launch_tile(kg_ctx* ctx, hipStream_t st, const NttStepArgs& a, uint32_t ntiles) {
  using T = NttTile<Fr, LOG_M, LOG_TC, ROW>;
  const size_t lds_bytes = T::SINGLE ? 0 : (size_t)T::ELEMS * 36;
  if (lds_bytes > 48 * 1024)               // more than the default dynamic LDS limit needs the attribute (per device: set at every launch)
    KG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_tile<LOG_M, LOG_TC, ROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((k_ntt_tile<LOG_M, LOG_TC, ROW>), dim3(ntiles), dim3(T::NT), lds_bytes, st, a);
  return KG_OK;
}
int launch_step(kg_ctx* ctx, hipStream_t st, const NttStepDesc& d, const NttStepArgs& a, uint32_t ntiles) {
  const int key = d.log_m * 16 + d.log_tc;
  if (!d.row) {
    switch (key) {
#define X(m, tc) case (m) * 16 + (tc): return launch_tile<m, tc, false>(ctx, st, a, ntiles);
      KG_NTT_SHAPES(X) KG_NTT_SHAPES_COL_ONLY(X)
#undef X
    }
  } else {
    switch (key) {
#define X(m, tc) case (m) * 16 + (tc): return launch_tile<m, tc, true>(ctx, st, a, ntiles);
      KG_NTT_SHAPES(X) KG_NTT_SHAPES_ROW_ONLY(X)
#undef X
    }
  }
  return set_err(ctx, KG_ERR_UNSUPPORTED, "ntt tile shape not instantiated");
}

}  // namespace

namespace kg {
void tw_cache_free(kg_ctx* c) {
  for (kg_tw_cache* t : c->tw) free_tables(t);
  c->tw.clear();
}
}  // namespace kg

extern "C" {

}  // extern "C"

namespace kg {
// Build (or find) the twiddle tables of a transform size on the context's main stream.
int ntt_prepare(kg_ctx* ctx, uint32_t log_n, int inverse) {
  kg_tw_cache* T;
  return get_tables(ctx, log_n, inverse ? 1 : 0, &T);
}

// Enqueue one transform on `st`; tmp: scratch of n elements private to this call (unused when log_n <= 11).
int ntt_enqueue(kg_ctx* ctx, hipStream_t st, uint64_t* tmp, uint64_t* d_data, uint32_t log_n, int inverse, int coset) {
  inverse = inverse ? 1 : 0;
  kg_tw_cache* T;
  KG_TRY(get_tables(ctx, log_n, inverse, &T));
  NttStepDesc d[3];
  const int nsteps = ntt_plan(log_n, plan_steps_env(), d, plan_tile_env());
  if (nsteps > 1 && !tmp) return set_err(ctx, KG_ERR_BAD_ARG, "ntt scratch missing");

  const NttTables tabs{T->small, T->lo, T->hi, T->cos_lo, T->cos_hi, T->direct[0], T->direct[1], T->lo_bits};
  PhaseScope ph(ctx, "ntt", st);
  for (int i = 0; i < nsteps; ++i) {
    NttStepArgs a;
    const uint32_t ntiles = ntt_step_args(log_n, nsteps, d, i, tabs, d_data, tmp, inverse, coset, a);
    KG_TRY(launch_step(ctx, st, d[i], a, ntiles));
  }
  ph.end();
  KG_HIP(ctx, hipGetLastError());
  return KG_OK;
}
}  // namespace kg

extern "C" {

int kg_ntt_bn254_fr(kg_ctx* ctx, uint64_t* d_data, uint32_t log_n, int inverse, int coset) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !d_data || log_n < 1 || log_n > 28) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t* tmp = nullptr;
  if (log_n > (uint32_t)NTT_MAX_LOG_M) {
    KG_TRY(ensure_ws2(ctx, ((size_t)1 << log_n) * 32));
    tmp = (uint64_t*)ctx->ws2;
  }
  return kg::ntt_enqueue(ctx, ctx->stream, tmp, d_data, log_n, inverse, coset);
  });
}

int kg_ntt_plan(uint32_t log_n, uint32_t* log_m, uint32_t* log_tile) {
  if (log_n < 1 || log_n > 28 || !log_m || !log_tile) return 0;
  NttStepDesc d[3];
  const int s = ntt_plan(log_n, plan_steps_env(), d, plan_tile_env());
  for (int i = 0; i < 3; ++i) {
    log_m[i] = i < s ? (uint32_t)d[i].log_m : 0u;
    log_tile[i] = i < s ? (uint32_t)(d[i].log_m + d[i].log_tc) : 0u;
  }
  return s;
}

int kg_fr_divide_by_z_on_coset(kg_ctx* ctx, uint64_t* d_data, uint32_t log_n) {
  // fft.rs:150-154: every evaluation * (7^n - 1)^-1
  if (!ctx || !d_data || log_n < 1 || log_n > 28) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  kg_tw_cache* T;
  KG_TRY(get_tables(ctx, log_n, 0, &T));
  const size_t n = (size_t)1 << log_n;
  hipLaunchKernelGGL(k_scale_const, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_data, n, T->zinv);
  KG_HIP(ctx, hipGetLastError());
  return KG_OK;
}

}  // extern "C"
