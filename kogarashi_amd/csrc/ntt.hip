// ntt.hip -- radix-2 Fr NTT on gfx950, natural order in and out.
//
// Replaces groth16/src/fft.rs: Fft::new (:27-89: the reference re-materialises n/2 + n/2 + n + n table
// entries per proof), prepare_fft (:157-162, bit-reversal swaps), classic_fft_arithmetic (:166-192, recursive
// DIT) and butterfly_arithmetic (:195-218).  dft(v)[i] = sum_j v[j] w^(ij) with w = ROOT_OF_UNITY^(2^(28-k)).
//
// Decomposition n = n1*n2*n3 (each factor <= 2^8..2^10; Cooley-Tukey index map j = j1*n2*n3 + j2*n3 + j3,
// i = i1 + n1*i2 + n1*n2*i3), so no separate bit-reversal pass exists and every step is one HBM round trip:
//   step A  n1-point DFTs down columns (stride n2*n3), 8 adjacent columns per workgroup, then * w_n^(i1*c)
//   step B  n2-point DFTs at stride n3 inside each i1 slab, in place, then * w_n^(n1*i2*j3)
//   step C  n3-point DFTs along contiguous rows, written transposed (8 adjacent i1 = 256 B per row of the output)
// A tile (<= 2048 elements = 72 KiB of 29-bit limbs, structure-of-arrays) lives in LDS for all log2(m) butterfly
// stages; two workgroups share a CU's 160 KiB.  Data never leaves the caller's Montgomery domain: the
// transform is linear and twiddles are multiplied in as internal-form constants (fp29.h), so there is no
// domain conversion, only a limb re-packing at load/store.
// Algorithmic HBM bytes: 64 B/element; this design moves 64 B/element per step.
#include "common.h"
#include "ntt_core.h"
#include <type_traits>

using namespace kg;

struct kg_tw_cache {
  uint32_t log_n;
  int inverse;
  uint32_t lo_bits;
  uint32_t* small = nullptr;   // w_1024^e, e < 512          [e][9]
  uint32_t* lo = nullptr;      // w_n^e, e < 2^lo_bits        [e][9]
  uint32_t* hi = nullptr;      // w_n^(e << lo_bits)          [e][9]
  uint32_t* cos_lo = nullptr;  // g^(+-e) (g = 7), e < 2^lo_bits, for the coset shift [* n^-1 when inverse]
  uint32_t* cos_hi = nullptr;  // g^(+-(e << lo_bits))
  uint32_t* zinv = nullptr;    // (7^n - 1)^-1, one entry (fft.rs:141-154)
  // inter-step twiddles read instead of generated: direct[s][r * inner + c] = w_n^(r * c * mult) for step s (0: A, 1: B).
  // Step B's table is n2 * n3 <= 2^16 entries; step A's is n entries and is kept only while it stays cache-resident
  // is worth its memory (log_n <= DIRECT_A_MAX_LOG = 22: 36 MB at 2^20, 151 MB at 2^22).  Saves the running-product update (one of the
  // two products per element): 2^18 59 -> 52 us, 2^20 164 -> 145 us, 2^22 557 -> 548 us.
  uint32_t* direct[2] = {nullptr, nullptr};
};

namespace {

constexpr int TILE = 2048;            // elements per workgroup tile
constexpr int NT = 512;               // threads per workgroup: one radix-4 group per lane on a 2048-element tile (92-96 VGPRs)
constexpr int SMALL_LOG = 10;         // largest in-LDS DFT: 2^10
constexpr uint32_t DIRECT_A_MAX_LOG = 22;

__device__ __forceinline__ Fr ld_tw(const uint32_t* __restrict__ tab, size_t e) {
  Fr r;
  const uint32_t* p = tab + e * 9;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = p[k];
  return r;
}
__device__ __forceinline__ void st_tw(uint32_t* tab, size_t e, const Fr& a) {
#pragma unroll
  for (int k = 0; k < 9; ++k) tab[e * 9 + k] = a.l[k];
}

// base = ROOT_OF_UNITY^(+-1) squared (28 - log) times  => primitive 2^log-th root (fft.rs:34,44)
__device__ Fr root_of(uint32_t log, int inverse) {
  Fr g = Fr::from_const(inverse ? FrParams::ROOT_OF_UNITY_INV : FrParams::ROOT_OF_UNITY);
  for (uint32_t i = log; i < 28; ++i) g = sqr(g);
  return g;
}
__device__ Fr pow_u64(Fr base, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = mul(r, base);
    base = sqr(base);
    e >>= 1;
  }
  return r;
}

// kind 0: small (root of order 2^10, e < 512); 1: lo; 2: hi; 3: coset lo; 4: coset hi
__global__ void __launch_bounds__(64) k_build_table(int kind, uint32_t log_n, int inverse, uint32_t lo_bits, uint32_t count, uint32_t* __restrict__ tab) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  Fr v;
  if (kind == 0) v = pow_u64(root_of(SMALL_LOG, inverse), e);
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

struct StepArgs {
  const uint32_t* tw_direct;   // col flavour: inter-step twiddle table [r][c] (nullptr: generate w_n^(r * cexp) by a running product)
  const uint64_t* in;
  uint64_t* out;
  uint32_t log_m;        // DFT size of this step
  uint32_t log_tc;       // tile columns
  uint64_t inner;        // col flavour: contiguous run length (elements)
  uint64_t mult;         // col flavour: twiddle exponent multiplier; 0 = no twiddle
  uint64_t n1;           // row flavour: i1 extent;  n2 = G / n1
  uint64_t G;            // number of independent DFTs (= n / m)
  uint32_t log_inner, log_n1, log_G;   // inner, n1 and G are powers of two: index arithmetic is shifts and masks (a 64-bit
                                       // division costs ~100 instructions, and there were five per element and step)
  uint32_t lo_bits;
  const uint32_t* tw_small;
  const uint32_t* tw_lo;
  const uint32_t* tw_hi;
  uint64_t scale_mode;   // 0 none; 1: multiply input element j by cos table at exponent j (coset dft, fft.rs:109-116)
                         // 2: multiply output element i by cos table at exponent i (coset idft / n^-1, fft.rs:104,119-127)
  const uint32_t* cos_lo;
  const uint32_t* cos_hi;
};

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t bits) { return __brev(v) >> (32 - bits); }

__device__ __forceinline__ Fr two_level(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t lo_bits, uint64_t e) {
  const uint64_t el = e & ((1ull << lo_bits) - 1), eh = e >> lo_bits;
  Fr a = ld_tw(lo, el);
  if (eh == 0) return a;
  return mul(a, ld_tw(hi, eh));
}

// LDS tile, structure-of-arrays: limb k of tile element e at lds[k * TILE + e]
template <bool ROW>
__device__ __forceinline__ uint32_t tile_index(uint32_t r, uint32_t col, uint32_t log_m, uint32_t log_tc) {
  // col flavour: columns fastest (global loads run along columns); row flavour: rows fastest, odd pitch.
  // The row index is XOR-swizzled (bits 0-4 ^= bits 3-7): whichever 5 index bits vary across a 32-lane group in the
  // three register passes (bits 3-7, then 0-2 and 6-7, then 0-4), the 32 lanes land on 32 different banks.
  r ^= (r >> 3) & 31u;
  if (ROW) return col * ((1u << log_m) + 1u) + r;
  return (r << log_tc) + col;
}
constexpr int LDS_TILE_WORDS = 9 * (TILE + 16);
constexpr int LDS_WORDS = LDS_TILE_WORDS + 9 * 128;      // tile + twiddle table w_m^e, e < 128

__device__ __forceinline__ Fr lds_load(const uint32_t* lds, uint32_t e) {
  Fr r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = lds[k * (TILE + 16) + e];
  return r;
}
__device__ __forceinline__ void lds_store(uint32_t* lds, uint32_t e, const Fr& a) {
#pragma unroll
  for (int k = 0; k < 9; ++k) lds[k * (TILE + 16) + e] = a.l[k];
}

// twiddle source of one register pass: w_{2^s}^j = w_m^(j * m / 2^s), from the LDS copy (m <= 256) or the global table
struct TwSrc {
  const uint32_t* twl;
  const uint32_t* tw_small;
  bool in_lds;
  uint32_t log_m, s0, r_low;
  __device__ __forceinline__ Fr operator()(int t, int k0) const {
    const uint32_t s = s0 + (uint32_t)t;
    const uint32_t j = (((uint32_t)k0 & ((1u << (t - 1)) - 1)) << s0) | r_low;     // r mod 2^(s-1)
    if (in_lds) {
      Fr r;
      uint32_t e = j << (log_m - s);
      e ^= e >> 4;                                            // same swizzle as the fill: strided exponents spread over banks
#pragma unroll
      for (int k = 0; k < 9; ++k) r.l[k] = twl[k * 128 + e];
      return r;
    }
    return ld_tw(tw_small, (size_t)j << (SMALL_LOG - s));
  }
};

// G consecutive radix-2 DIT stages (butterfly_arithmetic, fft.rs:195-218) with the 2^G elements of a lane in
// registers: the lane owns the elements that differ in index bits [s0, s0+G), so a 2^8-point tile makes 3 LDS round
// trips and 3 barriers instead of 8.
template <bool ROW, int G>
__device__ __forceinline__ void radix_pass(uint32_t* lds, const uint32_t* twl, bool tw_in_lds, const StepArgs& A, uint32_t s0) {
  const uint32_t m = 1u << A.log_m, tc = 1u << A.log_tc;
  const uint32_t groups = (m << A.log_tc) >> G;
  for (uint32_t q = threadIdx.x; q < groups; q += NT) {
    uint32_t col, rest;
    if (ROW) { rest = q & ((m >> G) - 1); col = q >> (A.log_m - G); }
    else { col = q & (tc - 1); rest = q >> A.log_tc; }
    const uint32_t r_low = rest & ((1u << s0) - 1), r_high = rest >> s0;
    const uint32_t base_r = (r_high << (s0 + G)) | r_low;
    Fr x[1 << G];
#pragma unroll
    for (int k = 0; k < (1 << G); ++k) x[k] = lds_load(lds, tile_index<ROW>(base_r | ((uint32_t)k << s0), col, A.log_m, A.log_tc));
    TwSrc tw{twl, A.tw_small, tw_in_lds, A.log_m, s0, r_low};
    dit_network<G>(x, s0 == 0, tw);
#pragma unroll
    for (int k = 0; k < (1 << G); ++k) lds_store(lds, tile_index<ROW>(base_r | ((uint32_t)k << s0), col, A.log_m, A.log_tc), norm(x[k]));
  }
}

template <bool ROW>
__global__ void __launch_bounds__(NT, 2) k_ntt_step(StepArgs A) {
  KG_SERVICE_PRIO();
  extern __shared__ uint32_t lds[];
  const uint32_t m = 1u << A.log_m, tc = 1u << A.log_tc;
  const uint64_t g0 = (uint64_t)blockIdx.x << A.log_tc;
  const uint32_t tile_elems = m << A.log_tc;

  // ---- load (limb re-packing only), bit-reversed row placement for the DIT stages -------------------
  for (uint32_t idx = threadIdx.x; idx < tile_elems; idx += NT) {
    uint32_t r, col;
    uint64_t addr;
    if (ROW) {
      r = idx & (m - 1); col = idx >> A.log_m;
      const uint64_t g = g0 + col, i1 = g & (A.n1 - 1), i2 = g >> A.log_n1;
      addr = (((i1 << (A.log_G - A.log_n1)) + i2) << A.log_m) + r;
    } else {
      col = idx & (tc - 1); r = idx >> A.log_tc;
      const uint64_t g = g0 + col;
      addr = ((g >> A.log_inner) << (A.log_m + A.log_inner)) + ((uint64_t)r << A.log_inner) + (g & (A.inner - 1));
    }
    uint32_t w[8];
    load_words(A.in, addr, w);
    Fr v = limbs_from_words<FrParams>(w);
    if (A.scale_mode == 1) v = mul(v, two_level(A.cos_lo, A.cos_hi, A.lo_bits, addr));
    lds_store(lds, tile_index<ROW>(bitrev(r, A.log_m), col, A.log_m, A.log_tc), v);
  }
  __syncthreads();

  // ---- log2(m) radix-2 DIT stages (butterfly_arithmetic, fft.rs:195-218), two at a time in registers: one lane
  // owns the 4 elements that differ in index bits [s0, s0+2), so a tile makes 4 LDS round trips and 4 barriers instead
  // of 8; twiddles w_m^e (e < m/2) sit in LDS behind the tile when m <= 256
  uint32_t* twl = lds + LDS_TILE_WORDS;
  const bool tw_in_lds = A.log_m <= 8;
  if (tw_in_lds) {
    const uint32_t cnt = m >> 1;
    for (uint32_t e = threadIdx.x; e < cnt; e += NT) {
      Fr w = ld_tw(A.tw_small, (size_t)e << (SMALL_LOG - A.log_m));
#pragma unroll
      for (int k = 0; k < 9; ++k) twl[k * 128 + (e ^ (e >> 4))] = w.l[k];
    }
    __syncthreads();
  }
  for (uint32_t s0 = 0; s0 < A.log_m;) {
    const uint32_t left = A.log_m - s0;
    // radix-4 register passes (8 -> 2,2,2,2; 7 -> 2,2,2,1): with radix-8 passes only 256 lanes of a 2048-element tile have
    // work and the kernel needs 210 VGPRs (two waves per SIMD); radix-4 keeps 512 lanes busy at four waves per SIMD
    const uint32_t g = left >= 2 ? 2 : left;
    if (g == 2) radix_pass<ROW, 2>(lds, twl, tw_in_lds, A, s0);
    else radix_pass<ROW, 1>(lds, twl, tw_in_lds, A, s0);
    __syncthreads();
    s0 += g;
  }

  // ---- store: inter-step twiddle (or a plain value reduction), canonicalise, re-pack ------------------
  Fr tw_cur = Fr::one(), tw_step = Fr::one();
  for (uint32_t idx = threadIdx.x; idx < tile_elems; idx += NT) {
    uint32_t r, col;
    uint64_t addr;
    Fr v;
    if (ROW) {
      col = idx & (tc - 1); r = idx >> A.log_tc;           // adjacent lanes -> adjacent i1 -> contiguous output
      addr = (g0 + col) + ((uint64_t)r << A.log_G);
      v = lds_load(lds, tile_index<ROW>(r, col, A.log_m, A.log_tc));
      if (A.scale_mode == 2) v = mul(v, two_level(A.cos_lo, A.cos_hi, A.lo_bits, addr));
      else v = vred(v);
    } else {
      col = idx & (tc - 1); r = idx >> A.log_tc;
      const uint64_t g = g0 + col;
      addr = ((g >> A.log_inner) << (A.log_m + A.log_inner)) + ((uint64_t)r << A.log_inner) + (g & (A.inner - 1));
      v = lds_load(lds, tile_index<ROW>(r, col, A.log_m, A.log_tc));
      if (A.tw_direct) {
        v = mul(v, ld_tw(A.tw_direct, ((size_t)r << A.log_inner) + (size_t)(g & (A.inner - 1))));
      } else if (A.mult) {
        // w_n^(r * cexp): this lane keeps its column and walks r in steps of NT / tc, so the twiddle advances by a
        // fixed ratio -- one product per element instead of the two of a table lookup
        if (idx == threadIdx.x) {
          const uint64_t cexp = (g & (A.inner - 1)) * A.mult;
          tw_cur = two_level(A.tw_lo, A.tw_hi, A.lo_bits, (uint64_t)r * cexp);
          tw_step = two_level(A.tw_lo, A.tw_hi, A.lo_bits, (uint64_t)(NT >> A.log_tc) * cexp);
        } else {
          tw_cur = mul(tw_cur, tw_step);
        }
        v = mul(v, tw_cur);
      } else v = vred(v);
    }
    uint32_t w[8];
    words_from_limbs(reduce_2p(v), w);
    store_words(A.out, addr, w);
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

// log_n = k1 + k2 + k3: at most three steps of <= 8 bits (<= 10 for log_n > 24)
void factor_steps(uint32_t log_n, uint32_t& k1, uint32_t& k2, uint32_t& k3) {
  k1 = k2 = k3 = 0;
  const uint32_t cap = log_n > 24 ? SMALL_LOG : 8;
  if (log_n <= cap) k3 = log_n;
  else if (log_n <= 2 * cap) { k1 = (log_n + 1) / 2; k3 = log_n - k1; }
  else { k1 = (log_n + 2) / 3; k2 = (log_n - k1 + 1) / 2; k3 = log_n - k1 - k2; }
}

int get_tables(kg_ctx* ctx, uint32_t log_n, int inverse, kg_tw_cache** out) {
  for (kg_tw_cache* t : ctx->tw)
    if (t->log_n == log_n && t->inverse == inverse) { *out = t; return KG_OK; }
  kg_tw_cache* t = new kg_tw_cache();
  t->log_n = log_n; t->inverse = inverse;
  t->lo_bits = (log_n + 1) / 2;
  const uint32_t n_lo = 1u << t->lo_bits, n_hi = 1u << (log_n - t->lo_bits);
  auto alloc = [&](uint32_t** p, size_t entries) { return hipMalloc((void**)p, entries * 36); };
  if (alloc(&t->small, 512) != hipSuccess || alloc(&t->lo, n_lo) != hipSuccess || alloc(&t->hi, n_hi) != hipSuccess ||
      alloc(&t->cos_lo, n_lo) != hipSuccess || alloc(&t->cos_hi, n_hi) != hipSuccess || alloc(&t->zinv, 1) != hipSuccess) {
    delete t;
    return set_err(ctx, KG_ERR_OOM, "twiddle table allocation");
  }
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_build_table, dim3(8), dim3(64), 0, st, 0, log_n, inverse, t->lo_bits, 512u, t->small);
  hipLaunchKernelGGL(k_build_table, dim3((n_lo + 63) / 64), dim3(64), 0, st, 1, log_n, inverse, t->lo_bits, n_lo, t->lo);
  hipLaunchKernelGGL(k_build_table, dim3((n_hi + 63) / 64), dim3(64), 0, st, 2, log_n, inverse, t->lo_bits, n_hi, t->hi);
  hipLaunchKernelGGL(k_build_table, dim3((n_lo + 63) / 64), dim3(64), 0, st, 3, log_n, inverse, t->lo_bits, n_lo, t->cos_lo);
  hipLaunchKernelGGL(k_build_table, dim3((n_hi + 63) / 64), dim3(64), 0, st, 4, log_n, inverse, t->lo_bits, n_hi, t->cos_hi);
  hipLaunchKernelGGL(k_build_zinv, dim3(1), dim3(64), 0, st, log_n, t->zinv);
  {
    uint32_t k1, k2, k3;
    factor_steps(log_n, k1, k2, k3);
    if (k1 && log_n <= DIRECT_A_MAX_LOG) {                        // step A: w_n^(i1 * c), c < n / n1
      const size_t cnt = (size_t)1 << log_n;
      if (alloc(&t->direct[0], cnt) != hipSuccess) { delete t; return set_err(ctx, KG_ERR_OOM, "twiddle table allocation"); }
      hipLaunchKernelGGL(k_build_direct, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st, log_n, inverse, k1, log_n - k1, (uint64_t)1, t->direct[0]);
    }
    if (k2) {                                                     // step B: w_n^(n1 * i2 * j3)
      const size_t cnt = (size_t)1 << (k2 + k3);
      if (alloc(&t->direct[1], cnt) != hipSuccess) { delete t; return set_err(ctx, KG_ERR_OOM, "twiddle table allocation"); }
      hipLaunchKernelGGL(k_build_direct, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st, log_n, inverse, k2, k3, (uint64_t)1 << k1, t->direct[1]);
    }
  }
  KG_HIP(ctx, hipGetLastError());
  ctx->tw_fresh = true;
  ctx->tw.push_back(t);
  *out = t;
  return KG_OK;
}

}  // namespace

namespace kg {
void tw_cache_free(kg_ctx* c) {
  for (kg_tw_cache* t : c->tw) {
    hipFree(t->small); hipFree(t->lo); hipFree(t->hi); hipFree(t->cos_lo); hipFree(t->cos_hi); hipFree(t->zinv);
    if (t->direct[0]) hipFree(t->direct[0]);
    if (t->direct[1]) hipFree(t->direct[1]);
    delete t;
  }
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

// Enqueue one transform on `st`; tmp: scratch of n elements private to this call (unused when log_n <= 8).
int ntt_enqueue(kg_ctx* ctx, hipStream_t st, uint64_t* tmp, uint64_t* d_data, uint32_t log_n, int inverse, int coset) {
  inverse = inverse ? 1 : 0;
  kg_tw_cache* T;
  KG_TRY(get_tables(ctx, log_n, inverse, &T));
  const uint64_t n = 1ull << log_n;
  const size_t lds_bytes = (size_t)LDS_WORDS * 4;
  KG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_step<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  KG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));

  uint32_t k1, k2, k3;
  factor_steps(log_n, k1, k2, k3);
  const uint64_t n1 = 1ull << k1, n2 = 1ull << k2, n3 = 1ull << k3;

  StepArgs base{};
  base.lo_bits = T->lo_bits; base.tw_small = T->small; base.tw_lo = T->lo; base.tw_hi = T->hi;
  base.cos_lo = T->cos_lo; base.cos_hi = T->cos_hi;
  auto tile_cols = [&](uint32_t log_m, uint64_t limit) {
    uint32_t l = 11 - log_m;                 // TILE = 2^11 elements
    const uint32_t cap = log_n >= 17 ? 4u : 3u;   // 16 columns (512 B runs) once there are enough tiles to fill the chip; measured 2^18: 68 -> 59 us
    if (l > cap) l = cap;
    while ((1ull << l) > limit) --l;
    return l;
  };
  const bool pre_scale = coset && !inverse;          // coset_dft: * 7^j before the transform
  const bool post_scale = inverse != 0;              // idft: * n^-1 ; coset_idft: * n^-1 * 7^-i
  // plain idft (no coset) scales by the constant n^-1 in a separate pass below.

  if (k1 && !tmp) return set_err(ctx, KG_ERR_BAD_ARG, "ntt scratch missing");
  PhaseScope ph(ctx, "ntt", st);
  if (k1) {
    // step A: data -> tmp
    StepArgs a = base;
    a.in = d_data; a.out = tmp; a.log_m = k1; a.inner = n >> k1; a.mult = 1; a.G = n >> k1; a.n1 = 1; a.log_inner = log_n - k1; a.log_G = log_n - k1; a.log_n1 = 0;
    a.log_tc = tile_cols(k1, a.inner);
    a.scale_mode = pre_scale ? 1 : 0;
    a.tw_direct = T->direct[0];
    hipLaunchKernelGGL(k_ntt_step<false>, dim3((unsigned)(a.G >> a.log_tc)), dim3(NT), lds_bytes, st, a);
    if (k2) {
      // step B: tmp in place
      StepArgs b = base;
      b.in = tmp; b.out = tmp; b.log_m = k2; b.inner = n3; b.mult = n1; b.G = n >> k2; b.n1 = 1; b.log_inner = k3; b.log_G = log_n - k2; b.log_n1 = 0;
      b.log_tc = tile_cols(k2, b.inner);
      b.tw_direct = T->direct[1];
      hipLaunchKernelGGL(k_ntt_step<false>, dim3((unsigned)(b.G >> b.log_tc)), dim3(NT), lds_bytes, st, b);
    }
    // step C: tmp -> data (transposed write)
    StepArgs c = base;
    c.in = tmp; c.out = d_data; c.log_m = k3; c.G = n >> k3; c.n1 = n1; c.log_G = log_n - k3; c.log_n1 = k1; c.log_inner = 0;
    c.log_tc = tile_cols(k3, n1);
    c.scale_mode = (post_scale && coset) ? 2 : 0;
    hipLaunchKernelGGL(k_ntt_step<true>, dim3((unsigned)(c.G >> c.log_tc)), dim3(NT), lds_bytes, st, c);
  } else {
    StepArgs c = base;
    c.in = d_data; c.out = d_data; c.log_m = k3; c.G = 1; c.n1 = 1; c.log_tc = 0; c.log_G = 0; c.log_n1 = 0; c.log_inner = 0;
    c.scale_mode = pre_scale ? 1 : ((post_scale && coset) ? 2 : 0);
    hipLaunchKernelGGL(k_ntt_step<true>, dim3(1), dim3(NT), lds_bytes, st, c);
  }
  ph.end();
  KG_HIP(ctx, hipGetLastError());
  if (post_scale && !coset) {
    // plain idft: * n^-1 (fft.rs:104); cos_lo[0] of the inverse tables is 7^0 * n^-1
    hipLaunchKernelGGL(k_scale_const, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_data, (size_t)n, T->cos_lo);
    KG_HIP(ctx, hipGetLastError());
  }
  return KG_OK;
}
}  // namespace kg

extern "C" {

int kg_ntt_bn254_fr(kg_ctx* ctx, uint64_t* d_data, uint32_t log_n, int inverse, int coset) {
  if (!ctx || !d_data || log_n < 1 || log_n > 28) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t* tmp = nullptr;
  if (log_n > 8) {
    KG_TRY(ensure_ws2(ctx, ((size_t)1 << log_n) * 32));
    tmp = (uint64_t*)ctx->ws2;
  }
  return kg::ntt_enqueue(ctx, ctx->stream, tmp, d_data, log_n, inverse, coset);
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
