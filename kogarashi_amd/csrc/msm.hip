// msm.hip -- Pippenger bucket MSM for BN254 G1 / G2 and Grumpkin on gfx950.
//
// Replaces groth16/src/msm.rs:6-48 (msm_curve_addition: unsigned c-bit windows, one rayon task per window,
// serial bucket fill, summation by parts, c*i doublings per window) and, behind kg_commit, the naive
// scalar-mul fold of nova/src/pedersen.rs:15-20.  Output parity is on the AFFINE sum, so the
// device pipeline is its own design:
//
//   prep      scalars -> canonical integers k (one Montgomery product) biased by H = sum_w 2^(wc+c-1), so
//             every window's SIGNED digit is a plain bit-field of k+H (halves the bucket count);
//             bases -> internal Montgomery form as 9-limb coordinates (72 B per G1 point), identity flag in a spare bit
//   count     one workgroup per (scalar chunk, window): the window's whole histogram (2^(c-1) counters,
//   scan      up to 128 KiB) lives in LDS -- a single-pass counting sort with a 15-bit digit
//   scatter   -> per-bucket lists of (point index | sign)
//   accumulate one lane per bucket: XYZZ += +-P over its list (madd, 8M+2S), bases gathered through L2/MALL
//   reduce    sum_b b*B_b by log2(B) halving levels (pair sums + odd-index sums = bit planes of b); depth
//             c-1 point additions instead of the reference's 2*2^c-long serial chain
//   finish    the c*W bit-plane sums go to the host, which runs the 255-step double-and-add (host_fp.h)
//
// Algorithmic HBM bytes: 96 B/pair (G1, Grumpkin), 160 B/pair (G2): SURVEY.md 8d.
#include "common.h"
#include "host_fp.h"
#include "fp2s.h"
#include "msm_internal.h"
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace kg;

namespace {

struct G1Cfg { using F = Fq; using KF = Fq; using SP = FrParams; using HF = HostFq; static constexpr int E64 = 4; static constexpr int ID = KG_G1; };
struct GkCfg { using F = Fr; using KF = Fr; using SP = FqParams; using HF = HostFr; static constexpr int E64 = 4; static constexpr int ID = KG_GRUMPKIN; };
struct G2Cfg { using F = Fq2; using KF = Fp2S<Fq>; using SP = FrParams; using HF = HostFq2; static constexpr int E64 = 8; static constexpr int ID = KG_G2; };

constexpr uint32_t INF_BIT = 0x80000000u;   // bit 255 of the packed x coordinate marks an identity base

// ---------------------------------------------------------------------------------------------------
// prep
// ---------------------------------------------------------------------------------------------------
// kt: structure-of-arrays, word j of scalar i at kt[j * n + i]
template <class SP>
__global__ void __launch_bounds__(256) k_prep_scalars(const uint64_t* __restrict__ scalars, size_t n, Words8 H,
                                                      uint32_t* __restrict__ kt) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8], k[8];
  load_words(scalars, i, w);
  ref_to_int<SP>(w, k);
  uint64_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint64_t s = (uint64_t)k[j] + H.w[j] + cy;
    kt[(size_t)j * n + i] = (uint32_t)s;
    cy = s >> 32;
  }
}

// The same conversion for the two-pass sort, which also wants the first pass's histogram: a workgroup converts PREP_CH
// consecutive scalars, peels all W digits off each k + H while the words are still in registers (a 256-bit funnel
// shift by c per window), counts bucket groups in LDS ([W][G] counters) and adds its counters to the (window, chunk,
// group) table that k_group_scan turns into offsets -- no second read of kt for counting.
constexpr int PREP_CH = 4096;
constexpr int PREP_NT = 256;      // one wave per SIMD at <= 64 VGPRs: fits beside a resident accumulation (4 x 112 VGPRs per SIMD)
template <class SP>
__global__ void __launch_bounds__(PREP_NT) k_prep_scalars_count(const uint64_t* __restrict__ scalars, size_t n, Words8 H, uint32_t* __restrict__ kt,
                                                             int c, int W, int shift, int G, int nch, size_t chunk_len, uint32_t* __restrict__ cnt, int per_wg) {
  KG_SERVICE_PRIO();
  extern __shared__ uint32_t hist[];                 // [W][G]
  for (int t = threadIdx.x; t < W * G; t += blockDim.x) hist[t] = 0;
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * per_wg;      // per_wg divides PREP_CH, which divides chunk_len: a workgroup stays inside one chunk
  const uint32_t cmask = (1u << c) - 1u, half = 1u << (c - 1);
  for (int r = 0; r < per_wg / PREP_NT; ++r) {
    const size_t i = lo + (size_t)r * PREP_NT + threadIdx.x;
    if (i >= n) break;
    uint32_t w[8], k[8];
    load_words(scalars, i, w);
    ref_to_int<SP>(w, k);
    uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t s = (uint64_t)k[j] + H.w[j] + cy;
      k[j] = (uint32_t)s;
      kt[(size_t)j * n + i] = k[j];
      cy = s >> 32;
    }
    for (int wd = 0; wd < W; ++wd) {
      uint32_t m;
      if (wd == W - 1) m = k[0] & 0x1ffffu;           // unsigned top window (window_digit)
      else {
        const int32_t d = (int32_t)(k[0] & cmask) - (int32_t)half;
        m = (uint32_t)(d < 0 ? -d : d);
      }
      if (m) atomicAdd(&hist[wd * G + ((m - 1) >> shift)], 1u);
#pragma unroll
      for (int j = 0; j < 7; ++j) k[j] = (k[j] >> c) | (k[j + 1] << (32 - c));
      k[7] >>= c;
    }
  }
  __syncthreads();
  const int ch = (int)(lo / chunk_len);
  for (int t = threadIdx.x; t < W * G; t += blockDim.x) {
    const uint32_t v = hist[t];
    if (v) atomicAdd(&cnt[((size_t)(t / G) * nch + ch) * G + (t % G)], v);
  }
}

// Limb form of a resident base (KG_FMT64_MIN_LOG=30; the default is the 64-byte form below): the 9 x 29-bit limbs of each
// coordinate as they are (internal Montgomery form, < 2p), i.e. 72 bytes per G1 / Grumpkin point and 144 per G2 point.  The gather pays 12.5 % more bytes
// -- it is not what bounds the accumulation -- and the ~50 shift / mask instructions per addition that re-spread 8
// words over 9 limbs disappear.  The identity flag rides in bit 31 of the first coordinate's top limb (< 2^23).
template <class F> struct FieldOf;
template <class Q> struct FieldOf<Fp<Q>> { using P = Q; };
template <class G> struct FieldOf<Fp2<G>> { using P = typename G::Params; };
template <class F> struct BaseIO;
template <class P> struct BaseIO<Fp<P>> {
  static constexpr int W = 8;    // u32 words of an element in the ABI (= u64 words of a point)
  static constexpr int PE = 9;   // u32 words of a resident element
  static __device__ __forceinline__ void convert(const uint64_t* src, uint32_t* dst) {   // ABI -> resident
    uint32_t w[8];
    load_words(src, 0, w);
    const Fp<P> v = from_ref<P>(w);                  // normalised limbs, < 2p
#pragma unroll
    for (int j = 0; j < 9; ++j) dst[j] = v.l[j];
  }
  static __device__ __forceinline__ Fp<P> from_words(const uint32_t* w) {
    Fp<P> r;
#pragma unroll
    for (int j = 0; j < 9; ++j) r.l[j] = w[j];
    return r;
  }
  // 64-byte form of the same point (the default, see resident_fmt64): the limbs of a value < 2p < 2^255 re-packed into 8
  // words per coordinate, identity flag in bit 255 of x.  A 72-byte point always straddles two 64-byte sectors, a 64-byte
  // one is exactly one; the price is ~50 shift / mask instructions per addition to spread the words over the limbs again.
  static constexpr int PK = 8;   // u32 words of a packed element
  static __device__ __forceinline__ void pack(const Fp<P>& v, uint32_t* dst) { words_from_limbs(v, dst); }
  static __device__ __forceinline__ bool load_point64(const uint32_t* src, Fp<P>& x, Fp<P>& y) {
    uint32_t w[16];
    const uint4* p = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint4 v = p[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
    const bool inf = (w[7] & INF_BIT) != 0;
    w[7] &= ~INF_BIT;
    x = limbs_from_words<P>(w); y = limbs_from_words<P>(w + 8);
    return inf;
  }
  // the same from the sixteen words themselves (k_acc_tasks_q reads them out of LDS)
  static __device__ __forceinline__ bool point64_from_words(uint32_t (&w)[16], Fp<P>& x, Fp<P>& y) {
    const bool inf = (w[7] & INF_BIT) != 0;
    w[7] &= ~INF_BIT;
    x = limbs_from_words<P>(w); y = limbs_from_words<P>(w + 8);
    return inf;
  }
  // whole point (x | y, 18 words at an 8-byte aligned address); returns the identity flag
  static __device__ __forceinline__ bool load_point(const uint32_t* src, Fp<P>& x, Fp<P>& y) {
    uint32_t w[18];
    const uint2* p = reinterpret_cast<const uint2*>(src);
#pragma unroll
    for (int j = 0; j < 9; ++j) { const uint2 v = p[j]; w[2 * j] = v.x; w[2 * j + 1] = v.y; }
    const bool inf = (w[8] & INF_BIT) != 0;
    w[8] &= ~INF_BIT;
    x = from_words(w); y = from_words(w + 9);
    return inf;
  }
};
template <class F> struct BaseIO<Fp2<F>> {
  static constexpr int W = 16;
  static constexpr int PE = 18;
  static __device__ __forceinline__ void convert(const uint64_t* src, uint32_t* dst) {
    BaseIO<F>::convert(src, dst);
    BaseIO<F>::convert(src + 4, dst + 9);
  }
  static constexpr int PK = 16;
  static __device__ __forceinline__ void pack(const Fp2<F>& v, uint32_t* dst) { BaseIO<F>::pack(v.c0, dst); BaseIO<F>::pack(v.c1, dst + 8); }
  static __device__ __forceinline__ bool load_point64(const uint32_t* src, Fp2<F>& x, Fp2<F>& y) {
    uint32_t w[32];
    const uint4* p = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint4 v = p[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
    const bool inf = (w[7] & INF_BIT) != 0;
    w[7] &= ~INF_BIT;
    using P = typename F::Params;
    x = {limbs_from_words<P>(w), limbs_from_words<P>(w + 8)};
    y = {limbs_from_words<P>(w + 16), limbs_from_words<P>(w + 24)};
    return inf;
  }
  // whole point (x.c0 | x.c1 | y.c0 | y.c1, 36 words at a 16-byte aligned address)
  static __device__ __forceinline__ bool load_point(const uint32_t* src, Fp2<F>& x, Fp2<F>& y) {
    uint32_t w[36];
    const uint4* p = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int j = 0; j < 9; ++j) { const uint4 v = p[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
    const bool inf = (w[8] & INF_BIT) != 0;
    w[8] &= ~INF_BIT;
    x = {BaseIO<F>::from_words(w), BaseIO<F>::from_words(w + 9)};
    y = {BaseIO<F>::from_words(w + 18), BaseIO<F>::from_words(w + 27)};
    return inf;
  }
};

}  // namespace
namespace kg {
template <class F> struct RawIO<Fp2S<F>> {
  static constexpr int NW = 18;
  static __device__ __forceinline__ Fp2S<F> load(const uint32_t* base, size_t stride, size_t i) {
    return {RawIO<F>::load(base + (size_t)(9 * Fp2S<F>::half()) * stride, stride, i)};
  }
  static __device__ __forceinline__ void store(uint32_t* base, size_t stride, size_t i, const Fp2S<F>& a) {
    RawIO<F>::store(base + (size_t)(9 * Fp2S<F>::half()) * stride, stride, i, a.v);
  }
};
}  // namespace kg
namespace {

// A resident G2 point read by a lane pair (fp2s.h): the even lane takes c0 of x and y, the odd lane c1 -- the same bytes a
// single lane reads through BaseIO<Fp2<G>>, half each.  The identity flag lives in the even lane's x and is shared by a
// lane exchange (both lanes of the pair always execute it).
template <class G> struct BaseIO<Fp2S<G>> {
  static constexpr int W = 16, PE = 18, PK = 16;
  using P = typename G::Params;
  static __device__ __forceinline__ bool load_point64(const uint32_t* src, Fp2S<G>& x, Fp2S<G>& y) {
    const int h = Fp2S<G>::half();
    const uint4* p = reinterpret_cast<const uint4*>(src);
    uint32_t w[16];
    const uint4 a0 = p[2 * h], a1 = p[2 * h + 1], b0 = p[4 + 2 * h], b1 = p[5 + 2 * h];
    w[0] = a0.x; w[1] = a0.y; w[2] = a0.z; w[3] = a0.w; w[4] = a1.x; w[5] = a1.y; w[6] = a1.z; w[7] = a1.w;
    w[8] = b0.x; w[9] = b0.y; w[10] = b0.z; w[11] = b0.w; w[12] = b1.x; w[13] = b1.y; w[14] = b1.z; w[15] = b1.w;
    int flag = (h == 0 && (w[7] & INF_BIT) != 0) ? 1 : 0;
    flag |= __shfl_xor(flag, 1);
    if (h == 0) w[7] &= ~INF_BIT;
    x.v = limbs_from_words<P>(w); y.v = limbs_from_words<P>(w + 8);
    return flag != 0;
  }
  static __device__ __forceinline__ bool load_point(const uint32_t* src, Fp2S<G>& x, Fp2S<G>& y) {
    const int h = Fp2S<G>::half();
    uint32_t wx[9], wy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { wx[k] = src[9 * h + k]; wy[k] = src[18 + 9 * h + k]; }
    int flag = (h == 0 && (wx[8] & INF_BIT) != 0) ? 1 : 0;
    flag |= __shfl_xor(flag, 1);
    if (h == 0) wx[8] &= ~INF_BIT;
    x.v = BaseIO<G>::from_words(wx); y.v = BaseIO<G>::from_words(wy);
    return flag != 0;
  }
};

// bases: ABI affine (x | y) -> resident form (limbs of x | limbs of y), 2*PE words per point; identity flag -> INF_BIT
// a resident point from its limbs: 2 * PE words (fmt64 = 0) or the 64-byte form, 2 * PK words (fmt64 = 1)
template <class F>
__device__ __forceinline__ void store_resident(uint32_t* out, size_t i, uint32_t* buf /* 2 * PE limbs words, flag applied */, int fmt64) {
  constexpr int PE = BaseIO<F>::PE, PK = BaseIO<F>::PK;
  if (fmt64) {
    const bool inf = (buf[8] & INF_BIT) != 0;
    buf[8] &= ~INF_BIT;
    uint32_t pk[2 * PK];
#pragma unroll
    for (int e = 0; e < PE / 9; ++e) {                       // PE / 9 base-field elements per coordinate
      Fp<typename FieldOf<F>::P> vx, vy;
#pragma unroll
      for (int k = 0; k < 9; ++k) { vx.l[k] = buf[9 * e + k]; vy.l[k] = buf[PE + 9 * e + k]; }
      words_from_limbs(vx, pk + 8 * e);
      words_from_limbs(vy, pk + PK + 8 * e);
    }
    if (inf) pk[7] |= INF_BIT;
    uint4* dst = reinterpret_cast<uint4*>(out + i * 2 * PK);
#pragma unroll
    for (int j = 0; j < PK / 2; ++j) dst[j] = make_uint4(pk[4 * j], pk[4 * j + 1], pk[4 * j + 2], pk[4 * j + 3]);
    return;
  }
  uint2* dst = reinterpret_cast<uint2*>(out + i * 2 * PE);
#pragma unroll
  for (int j = 0; j < PE; ++j) dst[j] = make_uint2(buf[2 * j], buf[2 * j + 1]);
}
template <class F, bool P64>
__global__ void __launch_bounds__(256) k_prep_bases(const uint64_t* __restrict__ bases, const uint8_t* __restrict__ inf, size_t n,
                                                    uint32_t* __restrict__ out) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int W = BaseIO<F>::W, PE = BaseIO<F>::PE;
  uint32_t buf[2 * PE];
  BaseIO<F>::convert(bases + (size_t)i * W, buf);            // W u32 words == W/2 u64 words per element
  BaseIO<F>::convert(bases + (size_t)i * W + W / 2, buf + PE);
  if (inf && inf[i]) buf[8] |= INF_BIT;
  store_resident<F>(out, i, buf, P64 ? 1 : 0);          // compile-time: the 72-byte form keeps its 32 VGPRs (it runs beside accumulations)
}

template <class F>
static void launch_prep_bases(hipStream_t st, const uint64_t* bases, const uint8_t* inf, size_t n, uint32_t* out, bool fmt64) {
  const dim3 grid((unsigned)((n + 255) / 256));
  if (fmt64) hipLaunchKernelGGL((k_prep_bases<F, true>), grid, dim3(256), 0, st, bases, inf, n, out);
  else hipLaunchKernelGGL((k_prep_bases<F, false>), grid, dim3(256), 0, st, bases, inf, n, out);
}

// Window tables (kg_bases_precompute): next[i] = 2^c * prev[i], both in resident form.  c doublings in XYZZ and one inversion
// per point -- a one-off per registered array (2^20 G1 points x 14 windows: ~50 ms), so no batching of the inversions.
template <class P> __device__ __forceinline__ void put_limbs(const Fp<P>& a, uint32_t* w) {
#pragma unroll
  for (int j = 0; j < 9; ++j) w[j] = a.l[j];
}
template <class F> __device__ __forceinline__ void put_limbs(const Fp2<F>& a, uint32_t* w) { put_limbs(a.c0, w); put_limbs(a.c1, w + 9); }
template <class F>
__global__ void __launch_bounds__(64) k_table_next(const uint32_t* __restrict__ prev, size_t n, int c, uint32_t* __restrict__ next, int fmt64) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int PE = BaseIO<F>::PE, PK = BaseIO<F>::PK;
  Affine<F> a;
  bool inf = fmt64 ? BaseIO<F>::load_point64(prev + i * 2 * PK, a.x, a.y) : BaseIO<F>::load_point(prev + i * 2 * PE, a.x, a.y);
  uint32_t buf[2 * PE];
#pragma unroll
  for (int j = 0; j < 2 * PE; ++j) buf[j] = 0;
  if (!inf) {
    XYZZ<F> p = double_affine(a);
    for (int k = 1; k < c; ++k) p = double_xyzz(p);
    Affine<F> o;
    inf = !to_affine(p, o);                          // a point of 2-power order (never in the prime-order groups) would end here
    if (!inf) { put_limbs(reduce(o.x), buf); put_limbs(reduce(o.y), buf + PE); }
  }
  if (inf) buf[8] |= INF_BIT;
  store_resident<F>(next, i, buf, fmt64);
}

}  // namespace
namespace kg {
void prep_bases_enqueue(int curve, hipStream_t st, const uint64_t* d_bases, const uint8_t* d_inf, size_t n, uint32_t* out, bool fmt64) {
  if (curve == KG_G1) launch_prep_bases<Fq>(st, d_bases, d_inf, n, out, fmt64);
  else if (curve == KG_GRUMPKIN) launch_prep_bases<Fr>(st, d_bases, d_inf, n, out, fmt64);
  else launch_prep_bases<Fq2>(st, d_bases, d_inf, n, out, fmt64);
}
void table_next_enqueue(int curve, hipStream_t st, const uint32_t* prev, size_t n, int c, uint32_t* next, bool fmt64) {
  const dim3 grid((unsigned)((n + 63) / 64));
  if (curve == KG_G1) hipLaunchKernelGGL(k_table_next<Fq>, grid, dim3(64), 0, st, prev, n, c, next, fmt64 ? 1 : 0);
  else if (curve == KG_GRUMPKIN) hipLaunchKernelGGL(k_table_next<Fr>, grid, dim3(64), 0, st, prev, n, c, next, fmt64 ? 1 : 0);
  else hipLaunchKernelGGL(k_table_next<Fq2>, grid, dim3(64), 0, st, prev, n, c, next, fmt64 ? 1 : 0);
}
}  // namespace kg
namespace {

// ---------------------------------------------------------------------------------------------------
// signed window digit of the biased scalar: returns bucket id + 1 (0 = skip) and the sign
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t window_digit(const uint32_t* __restrict__ kt, size_t n, size_t i, int w, int c, int W, bool& negative) {
  const int o = w * c;
  const int j = o >> 5, sh = o & 31;
  uint64_t v = kt[(size_t)j * n + i];
  // second plane only when the digit straddles a word (never for c = 16: half of the loads of the sort)
  if (j + 1 < 8 && sh + (w == W - 1 ? 17 : c) > 32) v |= (uint64_t)kt[(size_t)(j + 1) * n + i] << 32;
  uint32_t e = (uint32_t)(v >> sh);
  if (w == W - 1) {            // top window: unsigned remainder (no bias term was added for it)
    negative = false;
    return e & 0x1ffffu;
  }
  e &= (1u << c) - 1u;
  const int32_t d = (int32_t)e - (int32_t)(1u << (c - 1));
  negative = d < 0;
  return (uint32_t)(d < 0 ? -d : d);
}

// ---------------------------------------------------------------------------------------------------
// counting sort by (window, bucket): histogram of one (chunk, window) in LDS
// ---------------------------------------------------------------------------------------------------
//   shift = 0: one bin per bucket (single-pass sort);  shift = FINE_BITS: one bin per group of 2^shift buckets
__global__ void __launch_bounds__(1024) k_count(const uint32_t* __restrict__ kt, size_t n, int c, int W, size_t chunk_len,
                                                int shift, uint32_t* __restrict__ cnt) {
  KG_SERVICE_PRIO();
  extern __shared__ uint32_t hist[];
  const int B = (1 << (c - 1)) >> shift;
  // workgroups are dealt round-robin over the 8 XCDs by linear id: with the window in blockIdx.x (W = 16 or 17) all
  // chunks of a window land on the same XCD, so its L2 sees every write to that window's region of the sorted lists
  const int w = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;
  for (int b = threadIdx.x; b < B; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)ch * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    bool neg;
    uint32_t m = window_digit(kt, n, i, w, c, W, neg);
    if (m) atomicAdd(&hist[(m - 1) >> shift], 1u);
  }
  __syncthreads();
  uint32_t* dst = cnt + ((size_t)w * nch + ch) * B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) dst[b] = hist[b];
}

// zero fill with the service priority (the runtime's own fill kernel runs at the default priority and crawls beside an
// accumulation); words: number of 32-bit words, a multiple of 4, 16-byte aligned
__global__ void __launch_bounds__(256) k_zero(uint4* __restrict__ p, size_t quads) {
  KG_SERVICE_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}
static inline void zero_fill(hipStream_t st, void* p, size_t bytes) {
  const size_t quads = (bytes + 15) / 16;              // carved regions are padded to 256 bytes
  unsigned blocks = (unsigned)((quads + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks) hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(p), quads);
}

// per (window, bucket): exclusive prefix over chunks (in place) and the bucket's total
__global__ void __launch_bounds__(256) k_scan_chunks(uint32_t* __restrict__ cnt, int W, int nch, int B, uint32_t* __restrict__ bsize) {
  KG_SERVICE_PRIO();
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)W * B) return;
  const int w = (int)(t / B), b = (int)(t % B);
  uint32_t run = 0;
  for (int ch = 0; ch < nch; ++ch) {
    uint32_t* p = cnt + ((size_t)w * nch + ch) * B + b;
    uint32_t v = *p;
    *p = run;
    run += v;
  }
  bsize[t] = run;
}

// exclusive scan of one row of B counters by one 1024-thread workgroup: every lane owns a contiguous run (read as
// 16-byte vectors when the run allows), runs are combined with wave shuffles and one LDS hop
// (any block size that is a multiple of 64, up to 1024)
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* sh, uint32_t& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    uint32_t x = lane < nw ? sh[lane] : 0, xi = x;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { uint32_t o = __shfl_up(xi, d); if (lane >= d) xi += o; }
    if (lane < 16) sh[16 + lane] = xi - x;
    if (lane == 15) sh[32] = xi;
  }
  __syncthreads();
  total = sh[32];
  return inc - v + sh[16 + wv];
}
__device__ __forceinline__ void scan_row(const uint32_t* __restrict__ in, int B, uint32_t* __restrict__ out, uint32_t* __restrict__ row_total, int w) {
  __shared__ uint32_t sh[40];
  const int T = (int)blockDim.x;
  const int per = (B + T - 1) / T;
  const int lo = threadIdx.x * per, hi = lo + per < B ? lo + per : B;
  const uint32_t* src = in + (size_t)w * B;
  uint32_t* dst = out + (size_t)w * B;
  uint32_t s = 0;
  if ((per & 3) == 0 && hi - lo == per) {
    for (int b = lo; b < hi; b += 4) { uint4 q = *reinterpret_cast<const uint4*>(src + b); s += q.x + q.y + q.z + q.w; }
  } else {
    for (int b = lo; b < hi; ++b) s += src[b];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan_1024(s, sh, total);
  if ((per & 3) == 0 && hi - lo == per) {
    for (int b = lo; b < hi; b += 4) {
      uint4 q = *reinterpret_cast<const uint4*>(src + b), o;
      o.x = run; o.y = run + q.x; o.z = o.y + q.y; o.w = o.z + q.z;
      run = o.w + q.w;
      *reinterpret_cast<uint4*>(dst + b) = o;
    }
  } else {
    for (int b = lo; b < hi; ++b) { uint32_t v = src[b]; dst[b] = run; run += v; }
  }
  if (threadIdx.x == 0 && row_total) row_total[w] = total;
}
// shift = 0: final entries (index | sign << 31) in bucket order.  shift = FINE_BITS: first pass of the two-pass sort --
// entries land in their bucket GROUP and carry the bucket's low bits (index | fine << 24 | sign << 31).  A workgroup then
// has only B >> shift open output runs, so the L2 sees every line completed before it is evicted (the one-pass
// scatter pays a read-modify-write per 4-byte store once W * n * 4 B outgrows the L2: tools/ubench/scatter_rate.hip).
__global__ void __launch_bounds__(1024) k_scatter(const uint32_t* __restrict__ kt, size_t n, int c, int W, size_t chunk_len,
                                                  int shift, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bstart,
                                                  uint32_t* __restrict__ sorted) {
  KG_SERVICE_PRIO();
  extern __shared__ uint32_t off[];
  const int B = (1 << (c - 1)) >> shift;
  const uint32_t fine_mask = (1u << shift) - 1u;
  const int w = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;
  const uint32_t* src = cnt + ((size_t)w * nch + ch) * B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) off[b] = src[b] + bstart[(size_t)w * B + b];
  __syncthreads();
  const size_t lo = (size_t)ch * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
  uint32_t* dst = sorted + (size_t)w * n;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    bool neg;
    uint32_t m = window_digit(kt, n, i, w, c, W, neg);
    if (m) {
      uint32_t pos = atomicAdd(&off[(m - 1) >> shift], 1u);
      dst[pos] = (uint32_t)i | (((m - 1) & fine_mask) << 24) | (neg ? 0x80000000u : 0u);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// second pass of the two-pass sort.  A bucket group's entries are contiguous after the first pass; they are cut into
// segments of at most SEG entries, one workgroup per segment (so a 0/1-heavy witness, whose entries pile into one
// group, still spreads over the chip).  k_fine_local histograms a segment by the bucket's low bits and reserves the
// segment's place inside each bucket (atomicAdd on the bucket size returns it); k_fine_scatter sorts the segment in
// LDS and copies it out run by run, so consecutive lanes write consecutive addresses.
// ---------------------------------------------------------------------------------------------------
constexpr int FINE_BITS = 7, SEG = 8192;
// Entries between the two passes: the point index (24 bits; the window rides above it in a merged sort), the bucket's low FB bits
// and the sign.  FB = 7 fits four bytes and serves windows up to c = 18 (2^17 buckets = 1024 groups of 128); wider windows
// (c = 19, 20: the 2^23..2^24-pair commitments) take FB = 9 -- 1024 groups of 512 buckets -- in eight-byte entries.
template <int FB> struct Ent;
template <> struct Ent<7> {
  using T = uint32_t;
  static __device__ __forceinline__ T make(uint32_t idx_tag, uint32_t fine, bool neg) { return idx_tag | (fine << 24) | (neg ? 0x80000000u : 0u); }
  static __device__ __forceinline__ uint32_t fine(T e) { return (e >> 24) & 127u; }
  static __device__ __forceinline__ uint32_t out(T e) { return e & 0x80ffffffu; }
};
template <> struct Ent<9> {
  using T = uint64_t;
  static __device__ __forceinline__ T make(uint32_t idx_tag, uint32_t fine, bool neg) { return ((uint64_t)fine << 32) | idx_tag | (neg ? 0x80000000u : 0u); }
  static __device__ __forceinline__ uint32_t fine(T e) { return (uint32_t)(e >> 32); }
  static __device__ __forceinline__ uint32_t out(T e) { return (uint32_t)e; }
};
static inline int fine_bits_for(int c) { return c >= 19 ? 9 : FINE_BITS; }
// Segment length of the second pass.  A group of a uniform input holds n / G entries (G <= 1024 groups per window): 4096 at 2^22
// with FB = 7, but 16384 at 2^24 with FB = 9 -- the wide windows' segments are 20480 entries (80 KiB of LDS), so that their groups
// stay single segments too (the one-read, no-atomics path of k_fine_local).
template <int FB> struct SegLen { static constexpr uint32_t V = FB == 9 ? 20480u : (uint32_t)SEG; };
static inline uint32_t seg_len_for(int fb) { return fb == 9 ? SegLen<9>::V : SegLen<7>::V; }
constexpr uint32_t MULTI_SEG = 0x80000000u;          // flag in a window's segment total (segbase[G]): a group of several segments exists

// First pass of the two-pass sort, staged through LDS: a workgroup walks its (scalar chunk, window) pair in tiles of
// GS_TILE entries, ranks a tile's entries inside their bucket group with LDS atomics, lays the tile out group by group
// in LDS and copies it out, so that consecutive lanes write consecutive addresses of a group's run (a 4-byte store per
// lane to a random line is what bounds the unstaged k_scatter: tools/ubench/scatter_rate.hip).
// 256 threads and at most 64 VGPRs: one wave per SIMD that fits in the registers a resident accumulation leaves free, so the
// sort of the next MSM runs beside it (see KG_SERVICE_PRIO).
// (Since round 4 the default is k_group_scatter_big below; this kernel -- entries held in registers across the tile's barriers -- stays
// reachable with KG_GS_TILE=0 and is parity-tested: tests/test_gpu_parity.py::test_first_sort_pass_variants_give_the_oracles_point.)
constexpr int GS_NT = 256, GS_TILE = 1024, GS_MAXG = 1024;
// buffer addressing (descriptor + scalar plane offset + one 32-bit lane offset) instead of 64-bit flat addresses: the digit planes
// and the intermediate runs stay far below the 4 GiB a descriptor spans (two-pass sort: n <= 2^24)
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc soa_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffffu, 0x00020000);
}
// window_digit through a descriptor of kt: i4 = 4 * scalar index, n4 = 4 * n (bytes per word plane); w is uniform over the workgroup
__device__ __forceinline__ uint32_t window_digit_buf(BufRsrc kt, uint32_t n4, uint32_t i4, int w, int c, int W, bool& negative) {
  const int o = w * c;
  const int j = o >> 5, sh = o & 31;
  uint64_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(kt, i4, (uint32_t)j * n4, 0);
  if (j + 1 < 8 && sh + (w == W - 1 ? 17 : c) > 32) v |= (uint64_t)(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(kt, i4, (uint32_t)(j + 1) * n4, 0) << 32;
  uint32_t e = (uint32_t)(v >> sh);
  if (w == W - 1) {            // top window: unsigned remainder (no bias term was added for it)
    negative = false;
    return e & 0x1ffffu;
  }
  e &= (1u << c) - 1u;
  const int32_t d = (int32_t)e - (int32_t)(1u << (c - 1));
  negative = d < 0;
  return (uint32_t)(d < 0 ? -d : d);
}
template <int FB> struct EntLoad;
template <> struct EntLoad<7> { static __device__ __forceinline__ uint32_t ld(BufRsrc r, uint32_t idx) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, idx * 4u, 0, 0); } };
template <> struct EntLoad<9> {
  static __device__ __forceinline__ uint64_t ld(BufRsrc r, uint32_t idx) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, idx * 8u, 0, 0);
    return ((uint64_t)v.y << 32) | v.x;
  }
};
template <int FB> struct EntStore;
template <> struct EntStore<7> { static __device__ __forceinline__ void st(BufRsrc r, uint32_t idx, uint32_t e) { __builtin_amdgcn_raw_buffer_store_b32(e, r, idx * 4u, 0, 0); } };
template <> struct EntStore<9> {
  static __device__ __forceinline__ void st(BufRsrc r, uint32_t idx, uint64_t e) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 v; v.x = (uint32_t)e; v.y = (uint32_t)(e >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, idx * 8u, 0, 0);
  }
};
#ifndef KG_GS_ATTR
#define KG_GS_ATTR
#endif
template <int FB>
__global__ void __launch_bounds__(GS_NT) KG_GS_ATTR k_group_scatter(const uint32_t* __restrict__ kt, size_t n, int c, int W, size_t chunk_len, int G,
                                                         const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ gstart,
                                                         typename Ent<FB>::T* __restrict__ tmp, const uint32_t* __restrict__ woff, int mshift, int w0) {
  KG_SERVICE_PRIO();
  using E = typename Ent<FB>::T;
  constexpr uint32_t FINE = 1u << FB;
  __shared__ uint32_t cursor[GS_MAXG], hist[GS_MAXG], lstart[GS_MAXG], sh[40];    // hist doubles as the tile's address delta
  __shared__ E stage[GS_TILE];
  __shared__ uint16_t sg[GS_TILE];
  const int w = (int)blockIdx.x + w0, ch = blockIdx.y, nch = gridDim.y, tid = threadIdx.x;   // w0: first window of the group being sorted (all tables are indexed by the absolute window)
  const int per = (G + GS_NT - 1) / GS_NT;            // groups a lane owns in the scans (consecutive; <= 4)
  // merged sort (woff != nullptr): all windows share one run per group -- gstart is the merged table, woff[w][g] the entries
  // of the windows in front of w inside the group's run, and the window number rides in the entry above the scalar index
  // (not unrolled: four iterations' worth of 64-bit addresses in flight made this prologue, not the tile loop, set the kernel's
  // register count -- 64, one wave per SIMD beside an accumulation; at <= 48 two workgroups per CU fit there)
  if (woff) {
#pragma unroll 1
    for (int g = tid; g < G; g += GS_NT) cursor[g] = cnt[((size_t)w * nch + ch) * G + g] + woff[(size_t)w * G + g] + gstart[g];
  } else {
#pragma unroll 1
    for (int g = tid; g < G; g += GS_NT) cursor[g] = cnt[((size_t)w * nch + ch) * G + g] + gstart[(size_t)w * G + g];
  }
  const uint32_t lo = (uint32_t)((size_t)ch * chunk_len), hi = (size_t)lo + chunk_len < n ? lo + (uint32_t)chunk_len : (uint32_t)n;     // n < 2^31
  const BufRsrc rkt = soa_rsrc(kt), rdst = soa_rsrc(woff ? tmp : tmp + (size_t)w * n);
  const uint32_t n4 = (uint32_t)n * 4u;
  const uint32_t wtag = woff ? (uint32_t)w << mshift : 0u;
  for (uint32_t tile = lo; tile < hi; tile += GS_TILE) {
    for (int g = tid; g < G; g += GS_NT) hist[g] = 0;
    __syncthreads();
    E rec[GS_TILE / GS_NT];
    uint32_t key[GS_TILE / GS_NT];                              // key = group << 16 | rank inside the group (tile-local; < 2048)
#pragma unroll
    for (int r = 0; r < GS_TILE / GS_NT; ++r) {
      const uint32_t i = tile + (uint32_t)r * GS_NT + (uint32_t)tid;
      key[r] = 0xffffffffu;
      if (i < hi) {
        bool neg;
        const uint32_t m = window_digit_buf(rkt, n4, i * 4u, w, c, W, neg);
        if (m) {
          const uint32_t g = (m - 1) >> FB;
          rec[r] = Ent<FB>::make(i | wtag, (m - 1) & (FINE - 1), neg);
          key[r] = (g << 16) | atomicAdd(&hist[g], 1u);
        }
      }
    }
    __syncthreads();
    uint32_t v[4], vsum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int g = tid * per + j; v[j] = (j < per && g < G) ? hist[g] : 0u; vsum += v[j]; }
    uint32_t total;
    uint32_t ex = block_exclusive_scan_1024(vsum, sh, total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid * per + j;
      if (j < per && g < G) {
        lstart[g] = ex;
        hist[g] = cursor[g] - ex;                     // destination = position in the tile + this
        cursor[g] += v[j];
        ex += v[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GS_TILE / GS_NT; ++r) {
      if (key[r] != 0xffffffffu) {
        const uint32_t g = key[r] >> 16, p = lstart[g] + (key[r] & 0xffffu);
        stage[p] = rec[r];
        sg[p] = (uint16_t)g;
      }
    }
    __syncthreads();
    for (uint32_t p = tid; p < total; p += GS_NT) EntStore<FB>::st(rdst, p + hist[sg[p]], stage[p]);
    __syncthreads();
  }
}

// The same pass on tiles of TILE > GS_TILE entries.  With G groups a tile of GS_TILE entries leaves GS_TILE / G entries per run and
// tile -- ONE eight-byte entry at c = 20 (1024 groups), four four-byte ones at c = 16 -- so every store instruction touches its own
// sector, and a workgroup pays the load and the store latency once per 1024 entries.  Here a tile is TILE / G entries per run
// (64 bytes at TILE = 8192, c = 20) and TILE / GS_NT loads per lane are in flight at once; a lane cannot hold that many entries
// in registers, so the tile is walked twice -- count, then place (the digit planes of a tile are a few KiB: the second read is an
// L2 hit) -- and an entry's slot inside its run is handed out by the second walk's LDS atomic (the order inside a run is free: the
// fine pass re-sorts it).  stage / sg are dynamic LDS (gs_big_lds).
template <int FB, int TILE> constexpr size_t gs_big_lds() { return (size_t)TILE * (sizeof(typename Ent<FB>::T) + 2); }
template <int FB, int TILE, int NT>
__global__ void __launch_bounds__(NT) k_group_scatter_big(const uint32_t* __restrict__ kt, size_t n, int c, int W, size_t chunk_len, int G,
                                                             const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ gstart,
                                                             typename Ent<FB>::T* __restrict__ tmp, const uint32_t* __restrict__ woff, int mshift, int w0) {
  KG_SERVICE_PRIO();
  using E = typename Ent<FB>::T;
  constexpr uint32_t FINE = 1u << FB;
  constexpr int R = TILE / NT;
  __shared__ uint32_t cursor[GS_MAXG], delta[GS_MAXG], fill[GS_MAXG], sh[40];
  extern __shared__ __align__(8) unsigned char gs_dyn[];
  E* stage = reinterpret_cast<E*>(gs_dyn);
  uint16_t* sg = reinterpret_cast<uint16_t*>(gs_dyn + sizeof(E) * TILE);
  const int w = (int)blockIdx.x + w0, ch = blockIdx.y, nch = gridDim.y, tid = threadIdx.x;
  const int per = (G + NT - 1) / NT;
  if (woff) {
#pragma unroll 1
    for (int g = tid; g < G; g += NT) cursor[g] = cnt[((size_t)w * nch + ch) * G + g] + woff[(size_t)w * G + g] + gstart[g];
  } else {
#pragma unroll 1
    for (int g = tid; g < G; g += NT) cursor[g] = cnt[((size_t)w * nch + ch) * G + g] + gstart[(size_t)w * G + g];
  }
  const uint32_t lo = (uint32_t)((size_t)ch * chunk_len), hi = (size_t)lo + chunk_len < n ? lo + (uint32_t)chunk_len : (uint32_t)n;
  const BufRsrc rkt = soa_rsrc(kt), rdst = soa_rsrc(woff ? tmp : tmp + (size_t)w * n);
  const uint32_t n4 = (uint32_t)n * 4u;
  const uint32_t wtag = woff ? (uint32_t)w << mshift : 0u;
  for (uint32_t tile = lo; tile < hi; tile += TILE) {
    for (int g = tid; g < G; g += NT) fill[g] = 0;
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < R; ++r) {
      const uint32_t i = tile + (uint32_t)r * NT + (uint32_t)tid;
      if (i < hi) {
        bool neg;
        const uint32_t m = window_digit_buf(rkt, n4, i * 4u, w, c, W, neg);
        if (m) atomicAdd(&fill[(m - 1) >> FB], 1u);
      }
    }
    __syncthreads();
    uint32_t v[4], vsum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int g = tid * per + j; v[j] = (j < per && g < G) ? fill[g] : 0u; vsum += v[j]; }
    uint32_t total;
    uint32_t ex = block_exclusive_scan_1024(vsum, sh, total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid * per + j;
      if (j < per && g < G) {
        fill[g] = ex;                                 // the run's first slot in the tile: the second walk's cursor
        delta[g] = cursor[g] - ex;                    // destination = slot in the tile + this
        cursor[g] += v[j];
        ex += v[j];
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < R; ++r) {
      const uint32_t i = tile + (uint32_t)r * NT + (uint32_t)tid;
      if (i < hi) {
        bool neg;
        const uint32_t m = window_digit_buf(rkt, n4, i * 4u, w, c, W, neg);
        if (m) {
          const uint32_t g = (m - 1) >> FB;
          const uint32_t p = atomicAdd(&fill[g], 1u);
          stage[p] = Ent<FB>::make(i | wtag, (m - 1) & (FINE - 1), neg);
          sg[p] = (uint16_t)g;
        }
      }
    }
    __syncthreads();
    for (uint32_t p = tid; p < total; p += NT) EntStore<FB>::st(rdst, p + delta[sg[p]], stage[p]);
    // no barrier here: the next tile's first walk touches only fill[], and its barriers order everything else
  }
}

template <int FB, int TILE, int NT>
static hipError_t launch_gs_big(dim3 grid, hipStream_t st, const uint32_t* kt, size_t n, int c, int W, size_t chunk_len, int G, const uint32_t* cnt,
                                const uint32_t* gstart, typename Ent<FB>::T* tmp, const uint32_t* woff, int mshift, int w0) {
  constexpr size_t lds = gs_big_lds<FB, TILE>();
  auto* kern = &k_group_scatter_big<FB, TILE, NT>;
  if (lds + 4 * (3 * GS_MAXG + 40) > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, kt, n, c, W, chunk_len, G, cnt, gstart, tmp, woff, mshift, w0);
  return hipSuccess;
}

template <int FB>
static hipError_t launch_gs_big_any(int tile, int nt, dim3 grid, hipStream_t st, const uint32_t* kt, size_t n, int c, int W, size_t chunk_len, int G,
                                    const uint32_t* cnt, const uint32_t* gstart, typename Ent<FB>::T* tmp, const uint32_t* woff, int mshift, int w0) {
#define KG_GS_CASE(TILE, NT) \
  if (tile == TILE && nt == NT) return launch_gs_big<FB, TILE, NT>(grid, st, kt, n, c, W, chunk_len, G, cnt, gstart, tmp, woff, mshift, w0);
  KG_GS_CASE(4096, 256) KG_GS_CASE(4096, 512) KG_GS_CASE(4096, 1024)
  KG_GS_CASE(8192, 256) KG_GS_CASE(8192, 512) KG_GS_CASE(8192, 1024)
#undef KG_GS_CASE
  return hipErrorInvalidValue;
}

// One workgroup per window, one lane per bucket group: exclusive prefix of the group's counters over the chunks (in
// place), group sizes and starts, the segment table, and the window's bucket sizes zeroed for k_fine_local.
// (workgroup 0 also clears the `zwords` words at `zero`: the task decomposition's counters and length histogram)
__global__ void __launch_bounds__(1024) k_group_scan(uint32_t* __restrict__ cnt, int nch, int G, int B, uint32_t* __restrict__ gsize,
                                                       uint32_t* __restrict__ gstart, uint32_t* __restrict__ segbase, uint32_t* __restrict__ bsize,
                                                       uint32_t* __restrict__ zero, int zwords, uint32_t seg) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t sh[40];
  const int w = blockIdx.x, tid = threadIdx.x;
  const int NT = (int)blockDim.x;                     // 256, or 512 from 512 groups on (wide windows: 1024 groups and 2 MiB of bucket sizes to clear per window; 512 threads of 48 VGPRs still fit beside an accumulation)
  if (w == 0) for (int t = tid; t < zwords; t += NT) zero[t] = 0;
  const int per = (G + NT - 1) / NT;                  // consecutive groups per lane (<= 4)
  uint32_t run[4], ns[4], rsum = 0, nsum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = tid * per + j;
    run[j] = 0;
    if (j < per && g < G) {
      // eight chunks' counters are requested before the first running sum goes back (one load, one dependent store per chunk
      // was 26 us of pure latency at 256 chunks)
      uint32_t* col = cnt + (size_t)w * nch * G + g;
      int ch = 0;
      for (; ch + 8 <= nch; ch += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = col[(size_t)(ch + k) * G];
#pragma unroll
        for (int k = 0; k < 8; ++k) { col[(size_t)(ch + k) * G] = run[j]; run[j] += v[k]; }
      }
      for (; ch < nch; ++ch) {
        uint32_t* p = col + (size_t)ch * G;
        const uint32_t v = *p;
        *p = run[j];
        run[j] += v;
      }
      gsize[(size_t)w * G + g] = run[j];
    }
    ns[j] = (run[j] + seg - 1) / seg;
    rsum += run[j]; nsum += ns[j];
  }
  uint32_t total;
  uint32_t st = block_exclusive_scan_1024(rsum, sh, total);
  uint32_t ex = block_exclusive_scan_1024(nsum, sh, total);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = tid * per + j;
    if (j < per && g < G) {
      gstart[(size_t)w * G + g] = st;
      segbase[(size_t)w * (G + 1) + g] = ex;
      st += run[j]; ex += ns[j];
    }
  }
  // bit 31 of the segment total: some group of this window is more than one segment (the two-kernel fine path has work)
  const int any_multi = __syncthreads_or((ns[0] > 1) | (ns[1] > 1) | (ns[2] > 1) | (ns[3] > 1));
  if (tid == 0) segbase[(size_t)w * (G + 1) + G] = total | (any_multi ? MULTI_SEG : 0u);
  uint4* z = reinterpret_cast<uint4*>(bsize + (size_t)w * B);            // B is a multiple of 4 here (c >= 12)
  for (int b = threadIdx.x; b < B / 4; b += blockDim.x) z[b] = make_uint4(0, 0, 0, 0);
}

// Merged sort: the per-window group sizes of k_group_scan -> one run per group over all windows.  woff[w][g] = entries of
// group g in the windows in front of w; gsize_m / gstart_m / segbase_m = the W' = 1 tables the fine pass works from.
__global__ void __launch_bounds__(GS_NT) k_merge_groups(const uint32_t* __restrict__ gsize, int W, int G, uint32_t* __restrict__ woff,
                                                        uint32_t* __restrict__ gsize_m, uint32_t* __restrict__ gstart_m,
                                                        uint32_t* __restrict__ segbase_m) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t sh[40];
  const int tid = threadIdx.x;
  const int per = (G + GS_NT - 1) / GS_NT;            // consecutive groups per lane (<= 4)
  uint32_t run[4], ns[4], rsum = 0, nsum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = tid * per + j;
    run[j] = 0;
    if (j < per && g < G) {
      for (int w = 0; w < W; ++w) {
        woff[(size_t)w * G + g] = run[j];
        run[j] += gsize[(size_t)w * G + g];
      }
      gsize_m[g] = run[j];
    }
    ns[j] = (run[j] + SEG - 1) / SEG;
    rsum += run[j]; nsum += ns[j];
  }
  uint32_t total;
  uint32_t st = block_exclusive_scan_1024(rsum, sh, total);
  uint32_t ex = block_exclusive_scan_1024(nsum, sh, total);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = tid * per + j;
    if (j < per && g < G) {
      gstart_m[g] = st;
      segbase_m[g] = ex;
      st += run[j]; ex += ns[j];
    }
  }
  const int any_multi = __syncthreads_or((ns[0] > 1) | (ns[1] > 1) | (ns[2] > 1) | (ns[3] > 1));
  if (tid == 0) segbase_m[G] = total | (any_multi ? MULTI_SEG : 0u);
}

struct SegRange { int g; uint32_t lo, hi; };
// which group / entry range does segment s of window w cover?  (sb: the window's segbase row in LDS)
__device__ __forceinline__ bool seg_locate(const uint32_t* sb, int G, uint32_t s, const uint32_t* __restrict__ gstart,
                                           const uint32_t* __restrict__ gsize, int w, SegRange& r, uint32_t seg = SEG) {
  if (s >= (sb[G] & ~MULTI_SEG)) return false;
  int lo = 0, hi = G;                                // largest g with sb[g] <= s (empty groups repeat the value: take the last)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sb[mid] <= s) lo = mid; else hi = mid; }
  r.g = lo;
  const uint32_t st = gstart[(size_t)w * G + lo], sz = gsize[(size_t)w * G + lo];
  r.lo = st + (s - sb[lo]) * seg;
  r.hi = r.lo + seg < st + sz ? r.lo + seg : st + sz;
  return true;
}

// Second pass, segment-local form: a bucket group that fits ONE segment (<= SEG entries -- every group of a uniform input:
// 4096 entries at 2^20, c = 16) is histogrammed, ordered and written by a single workgroup in a single read of the
// intermediate.  The group's run in `sorted` is contiguous and its buckets follow each other inside it, so bucket starts need no
// global prefix: bstart = gstart + the local exclusive prefix (what k_bucket_rows computes from the sizes written here), no
// atomics on the bucket sizes, and the copy out is one coalesced stream.  Groups of several segments (skewed witnesses, every
// group of a merged sort) are only counted here (place reserved per segment and bucket); k_fine_scatter places them.
// exclusive prefix of FINE (<= 512) counters, one per thread of a 512-thread workgroup (threads >= FINE pass 0)
__device__ __forceinline__ uint32_t fine_exclusive(uint32_t v, uint32_t* wsum8) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o2 = __shfl_up(inc, d); if (lane >= d) inc += o2; }
  if (lane == 63) wsum8[wv] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (int k = 0; k < wv; ++k) before += wsum8[k];
  return inc - v + before;
}
#ifndef KG_FL_UNROLL
#define KG_FL_UNROLL 4      // entries a lane requests before it touches LDS (1 / 4 / 8: 2^24 commitment 19.18 / 19.00 / 19.00 ms, blocking 2^20 1.634 / 1.597 / 1.594)
#endif
template <int FB>
__global__ void __launch_bounds__(512) k_fine_local(const typename Ent<FB>::T* __restrict__ tmp, size_t n, int G, int B, int maxseg, const uint32_t* __restrict__ gstart,
                                                    const uint32_t* __restrict__ gsize, const uint32_t* __restrict__ segbase,
                                                    uint32_t* __restrict__ bsize, uint32_t* __restrict__ segcnt, uint32_t* __restrict__ segoff,
                                                    uint32_t* __restrict__ sorted) {
  KG_SERVICE_PRIO();
  using E = typename Ent<FB>::T;
  constexpr uint32_t FINE = 1u << FB, SEGN = SegLen<FB>::V;
  extern __shared__ uint32_t fl_lds[];                 // sb[1025 (+3)] | hist[FINE] | cursor[FINE] | wsum8[8] | stage[SEGN]
  uint32_t* const sb = fl_lds;
  uint32_t* const hist = sb + 1028;
  uint32_t* const cursor = hist + FINE;
  uint32_t* const wsum8 = cursor + FINE;
  uint32_t* const stage = wsum8 + 8;
  const int w = blockIdx.x;
  const uint32_t s = blockIdx.y;
  for (int g = threadIdx.x; g <= G; g += blockDim.x) sb[g] = segbase[(size_t)w * (G + 1) + g];
  if (threadIdx.x < FINE) hist[threadIdx.x] = 0;
  __syncthreads();
  SegRange r;
  if (!seg_locate(sb, G, s, gstart, gsize, w, r, SEGN)) return;
  const E* src = tmp + (size_t)w * n;
  if (gsize[(size_t)w * G + r.g] > SEGN) {
    // a segment of a larger group: histogram it and reserve its place inside each bucket (the atomicAdd on the bucket size returns
    // the segment's offset there); k_fine_scatter places the entries once the bucket starts are known
    uint32_t i = r.lo + threadIdx.x;
    for (; i + 3u * blockDim.x < r.hi; i += 4u * blockDim.x) {
      E e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = src[i + (uint32_t)k * blockDim.x];
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&hist[Ent<FB>::fine(e[k])], 1u);
    }
    for (; i < r.hi; i += blockDim.x) atomicAdd(&hist[Ent<FB>::fine(src[i])], 1u);
    __syncthreads();
    if (threadIdx.x < FINE) {
      const uint32_t cnt = hist[threadIdx.x];
      const size_t o = ((size_t)w * maxseg + s) * FINE + threadIdx.x;
      segcnt[o] = cnt;
      segoff[o] = cnt ? atomicAdd(&bsize[(size_t)w * B + (size_t)r.g * FINE + threadIdx.x], cnt) : 0u;
    }
    return;
  }
  // FB = 7: the segment's entries stay in registers between the histogram and the placement (16 per lane).  FB = 9: 40 eight-byte
  // entries per lane would not (the kernel has to fit beside an accumulation): the segment is read twice, the second time out of
  // the cache the first read filled (160 KiB per workgroup).  Buffer addressing (one 32-bit offset per access) and loops that are not
  // unrolled further than they must keep the kernel at two workgroups per CU beside an accumulation.
  constexpr int PER = (int)(SEGN / 512);
#ifdef KG_FINE_KEEP
  constexpr bool KEEP = FB == 7;
#else
  constexpr bool KEEP = false;      // measured: see EXPERIMENTS.md (round 4, sort kernels beside an accumulation)
#endif
  const BufRsrc rsrc = soa_rsrc(src + r.lo), rdst = soa_rsrc(sorted + (size_t)w * n + r.lo);      // r.lo = the group's start: its only segment
  const uint32_t len = r.hi - r.lo;
  E rec[KEEP ? PER : 1];
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = threadIdx.x + (uint32_t)k * 512u;
      if (i < len) {
        rec[k] = EntLoad<FB>::ld(rsrc, i);
        atomicAdd(&hist[Ent<FB>::fine(rec[k])], 1u);
      }
    }
  } else {
    // KG_FL_UNROLL entries requested before the first LDS atomic (left to the compiler the loads stay one at a time: 14 VGPRs)
    uint32_t i = threadIdx.x;
    for (; i + (KG_FL_UNROLL - 1) * 512u < len; i += KG_FL_UNROLL * 512u) {
      E e[KG_FL_UNROLL];
#pragma unroll
      for (int k = 0; k < KG_FL_UNROLL; ++k) e[k] = EntLoad<FB>::ld(rsrc, i + (uint32_t)k * 512u);
#pragma unroll
      for (int k = 0; k < KG_FL_UNROLL; ++k) atomicAdd(&hist[Ent<FB>::fine(e[k])], 1u);
    }
    for (; i < len; i += 512u) atomicAdd(&hist[Ent<FB>::fine(EntLoad<FB>::ld(rsrc, i))], 1u);
  }
  __syncthreads();
  uint32_t cnt = 0;
  if (threadIdx.x < FINE) {                                      // sizes out
    cnt = hist[threadIdx.x];
    bsize[(size_t)w * B + (size_t)r.g * FINE + threadIdx.x] = cnt;
  }
  const uint32_t ex = fine_exclusive(cnt, wsum8);               // exclusive prefix of the FINE counters
  if (threadIdx.x < FINE) cursor[threadIdx.x] = ex;
  __syncthreads();
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = threadIdx.x + (uint32_t)k * 512u;
      if (i < len) stage[atomicAdd(&cursor[Ent<FB>::fine(rec[k])], 1u)] = Ent<FB>::out(rec[k]);
    }
  } else {
    uint32_t i = threadIdx.x;
    for (; i + (KG_FL_UNROLL - 1) * 512u < len; i += KG_FL_UNROLL * 512u) {
      E e[KG_FL_UNROLL];
#pragma unroll
      for (int k = 0; k < KG_FL_UNROLL; ++k) e[k] = EntLoad<FB>::ld(rsrc, i + (uint32_t)k * 512u);
#pragma unroll
      for (int k = 0; k < KG_FL_UNROLL; ++k) stage[atomicAdd(&cursor[Ent<FB>::fine(e[k])], 1u)] = Ent<FB>::out(e[k]);
    }
    for (; i < len; i += 512u) {
      const E e = EntLoad<FB>::ld(rsrc, i);
      stage[atomicAdd(&cursor[Ent<FB>::fine(e)], 1u)] = Ent<FB>::out(e);
    }
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t p = threadIdx.x; p < len; p += blockDim.x) __builtin_amdgcn_raw_buffer_store_b32(stage[p], rdst, p * 4u, 0, 0);
}
template <int FB> static constexpr size_t fine_local_lds() { return (size_t)(1028 + 2 * (1u << FB) + 8 + SegLen<FB>::V) * 4; }

template <int FB> static constexpr size_t fine_scatter_lds() { return (size_t)(1028 + 3 * (1u << FB) + 8 + SegLen<FB>::V) * 4 + (size_t)SegLen<FB>::V * 2; }
constexpr int FS_ROWS = 64;                          // workgroups per window: a workgroup walks the window's segments in steps of gridDim.y
template <int FB>
__global__ void __launch_bounds__(512) k_fine_scatter(const typename Ent<FB>::T* __restrict__ tmp, size_t n, int G, int B, int maxseg,
                                                      const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gsize,
                                                      const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ bstart,
                                                      const uint32_t* __restrict__ segcnt, const uint32_t* __restrict__ segoff,
                                                      uint32_t* __restrict__ sorted, int Wg, int extra_w) {
  KG_SERVICE_PRIO();
  using E = typename Ent<FB>::T;
  constexpr uint32_t FINE = 1u << FB, SEGN = SegLen<FB>::V;
  extern __shared__ uint32_t fs_lds[];                 // sb[1028] | lstart | cursor | gbase [FINE each] | wsum8[8] | stage[SEGN] | sfine[SEGN] (16 bit)
  uint32_t* const sb = fs_lds;
  uint32_t* const lstart = sb + 1028;
  uint32_t* const cursor = lstart + FINE;
  uint32_t* const gbase = cursor + FINE;
  uint32_t* const wsum8 = gbase + FINE;
  uint32_t* const stage = wsum8 + 8;
  uint16_t* const sfine = reinterpret_cast<uint16_t*>(stage + SEGN);
  // columns >= Wg of the grid are further rows of window `extra_w` (the unsigned top window: its ~2^12 digit values fill a tenth of
  // the groups, every one of them several segments even on uniform scalars -- 64 workgroups walked them at 2^20 in 36 us, at 2^24 in 340)
  const int w = (int)blockIdx.x < Wg ? (int)blockIdx.x : extra_w;
  const uint32_t row0 = (int)blockIdx.x < Wg ? blockIdx.y : ((uint32_t)blockIdx.x - (uint32_t)Wg + 1u) * gridDim.y + blockIdx.y;
  const uint32_t rows = (extra_w >= 0 && w == extra_w) ? ((uint32_t)gridDim.x - (uint32_t)Wg + 1u) * gridDim.y : gridDim.y;
  // every group of this window is one segment (any uniform input): k_fine_local did it all.  (A launch of one workgroup per
  // segment that only returned cost 36 us per 2^20-pair sort: hence the few rows and the loop.)
  if (!(segbase[(size_t)w * (G + 1) + G] & MULTI_SEG)) return;
  for (int g = threadIdx.x; g <= G; g += blockDim.x) sb[g] = segbase[(size_t)w * (G + 1) + g];
  __syncthreads();
  const uint32_t nseg = sb[G] & ~MULTI_SEG;
  for (uint32_t s = row0; s < nseg; s += rows) {
    SegRange r;
    seg_locate(sb, G, s, gstart, gsize, w, r, SEGN);
    if (gsize[(size_t)w * G + r.g] <= SEGN) continue;              // done by k_fine_local (uniform over the workgroup)
    __syncthreads();                                 // the previous segment's stage / tables are no longer read
    uint32_t cnt = 0;
    if (threadIdx.x < FINE) {
      const size_t o = ((size_t)w * maxseg + s) * FINE + threadIdx.x;
      cnt = segcnt[o];
      gbase[threadIdx.x] = bstart[(size_t)w * B + (size_t)r.g * FINE + threadIdx.x] + segoff[o];
    }
    const uint32_t ex = fine_exclusive(cnt, wsum8);   // exclusive prefix of the segment's FINE counters
    if (threadIdx.x < FINE) {
      lstart[threadIdx.x] = ex;
      cursor[threadIdx.x] = ex;
    }
    __syncthreads();
    const E* src = tmp + (size_t)w * n;
    uint32_t i = r.lo + threadIdx.x;
    for (; i + 3u * blockDim.x < r.hi; i += 4u * blockDim.x) {      // four entries in flight per lane (see KG_FL_UNROLL)
      E rec[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) rec[k] = src[i + (uint32_t)k * blockDim.x];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t f = Ent<FB>::fine(rec[k]), pos = atomicAdd(&cursor[f], 1u);
        stage[pos] = Ent<FB>::out(rec[k]);
        sfine[pos] = (uint16_t)f;
      }
    }
    for (; i < r.hi; i += blockDim.x) {
      const E rec = src[i];
      const uint32_t f = Ent<FB>::fine(rec), pos = atomicAdd(&cursor[f], 1u);
      stage[pos] = Ent<FB>::out(rec);
      sfine[pos] = (uint16_t)f;
    }
    __syncthreads();
    uint32_t* dst = sorted + (size_t)w * n;
    const uint32_t len = r.hi - r.lo;
    for (uint32_t p = threadIdx.x; p < len; p += blockDim.x) {
      const uint32_t f = sfine[p];
      dst[gbase[f] + (p - lstart[f])] = stage[p];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// bucket accumulation, load balanced.  A bucket's list is cut into tasks of at most T entries; one lane per
// task.  Real witnesses are heavily skewed (0/1 scalars put ~n points in one bucket; the top window has few
// buckets), so partial sums of one bucket are re-summed in further rounds (T2 partials per lane) until every
// bucket owns a single point.  With uniform scalars every bucket is one task and no extra round runs.
// ---------------------------------------------------------------------------------------------------
struct Level {            // one round's task bookkeeping, all device pointers
  const uint32_t* cnt;    // [W*B] tasks of each bucket in this round
  const uint32_t* rel;    // [W*B] exclusive prefix of cnt inside the window
  const uint32_t* base;   // [W+1] first task of each window; base[W] = total
};

// tasks per bucket for item counts `in` and a segment length T; block-reduced maximum of `in`
__global__ void __launch_bounds__(1024) k_task_count(const uint32_t* __restrict__ in, size_t total, uint32_t T,
                                                     uint32_t* __restrict__ ntask, uint32_t* __restrict__ maxv) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t red[16];
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = 0;
  if (t < total) {
    v = in[t];
    ntask[t] = (v + T - 1) / T;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = 0;
    for (unsigned i = 0; i < (blockDim.x >> 6); ++i) m = red[i] > m ? red[i] : m;
    if (m) atomicMax(maxv, m);
  }
}

// per window: exclusive prefix of `in` -> rel, window total -> row_total[w]
__global__ void __launch_bounds__(1024) k_scan_rows(const uint32_t* __restrict__ in, int B, uint32_t* __restrict__ rel,
                                                    uint32_t* __restrict__ row_total) {
  KG_SERVICE_PRIO();
  scan_row(in, B, rel, row_total, blockIdx.x);
}
// base[w] = sum_{w' < w} row_total[w'], base[W] = grand total; info[0] = grand total, info[1] = *maxv
__global__ void k_row_bases(const uint32_t* __restrict__ row_total, int W, uint32_t* __restrict__ base, const uint32_t* __restrict__ maxv,
                            uint32_t* __restrict__ info) {
  KG_SERVICE_PRIO();
  if (threadIdx.x || blockIdx.x) return;
  uint32_t run = 0;
  for (int w = 0; w < W; ++w) { base[w] = run; run += row_total[w]; }
  base[W] = run;
  info[0] = run;
  info[1] = maxv ? *maxv : 0;
}

// Partial sums are written by lanes in length order but indexed by task id, so they use an array-of-structures
// layout (NW contiguous words per point, 16-byte vector accesses): a scattered point costs whole cache lines,
// not one sector per limb.
template <class F>
struct PointAoS {
  static constexpr int NW = PointIO<F>::NW;
  static __device__ __forceinline__ void store(uint32_t* base, size_t i, const XYZZ<F>& p) {
    uint32_t w[NW];
    pack(p, w);
    uint4* dst = reinterpret_cast<uint4*>(base + i * NW);
#pragma unroll
    for (int j = 0; j < NW / 4; ++j) dst[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
  }
  static __device__ __forceinline__ XYZZ<F> load(const uint32_t* base, size_t i) {
    uint32_t w[NW];
    const uint4* src = reinterpret_cast<const uint4*>(base + i * NW);
#pragma unroll
    for (int j = 0; j < NW / 4; ++j) { uint4 v = src[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
    return unpack(w);
  }
  template <class P> static __device__ __forceinline__ void put(const Fp<P>& a, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.l[k];
  }
  template <class G> static __device__ __forceinline__ void put(const Fp2<G>& a, uint32_t* w) { put(a.c0, w); put(a.c1, w + 9); }
  template <class P> static __device__ __forceinline__ void get(Fp<P>& a, const uint32_t* w) {
#pragma unroll
    for (int k = 0; k < 9; ++k) a.l[k] = w[k];
  }
  template <class G> static __device__ __forceinline__ void get(Fp2<G>& a, const uint32_t* w) { get(a.c0, w); get(a.c1, w + 9); }
  static __device__ __forceinline__ void pack(const XYZZ<F>& p, uint32_t* w) {
    constexpr int E = RawIO<F>::NW;
    put(p.x, w); put(p.y, w + E); put(p.zz, w + 2 * E); put(p.zzz, w + 3 * E);
  }
  static __device__ __forceinline__ XYZZ<F> unpack(const uint32_t* w) {
    constexpr int E = RawIO<F>::NW;
    XYZZ<F> p;
    get(p.x, w); get(p.y, w + E); get(p.zz, w + 2 * E); get(p.zzz, w + 3 * E);
    return p;
  }
};

// lane-pair Fq2: a lane's four coordinates are contiguous (36 words, nine 16-byte vectors) at half() * 36 inside the point
template <class G>
struct PointAoS<Fp2S<G>> {
  using F = Fp2S<G>;
  static constexpr int NW = PointIO<F>::NW;          // 72
  static __device__ __forceinline__ void store(uint32_t* base, size_t i, const XYZZ<F>& p) {
    uint32_t w[36];
#pragma unroll
    for (int k = 0; k < 9; ++k) { w[k] = p.x.v.l[k]; w[9 + k] = p.y.v.l[k]; w[18 + k] = p.zz.v.l[k]; w[27 + k] = p.zzz.v.l[k]; }
    uint4* dst = reinterpret_cast<uint4*>(base + i * NW + 36 * F::half());
#pragma unroll
    for (int j = 0; j < 9; ++j) dst[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
  }
  static __device__ __forceinline__ XYZZ<F> load(const uint32_t* base, size_t i) {
    uint32_t w[36];
    const uint4* src = reinterpret_cast<const uint4*>(base + i * NW + 36 * F::half());
#pragma unroll
    for (int j = 0; j < 9; ++j) { uint4 v = src[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
    XYZZ<F> p;
#pragma unroll
    for (int k = 0; k < 9; ++k) { p.x.v.l[k] = w[k]; p.y.v.l[k] = w[9 + k]; p.zz.v.l[k] = w[18 + k]; p.zzz.v.l[k] = w[27 + k]; }
    return p;
  }
};

// ---- order tasks by length (longest first) so the 64 lanes of a wave run equally long loops ---------------
// key = min(length, 255); bins are laid out in DESCENDING key order.  One lane per bucket: a bucket contributes
// ntask-1 full tasks (length T) and one remainder.
constexpr int LEN_BINS = 256;
// a bucket may reach the gather with up to GATHER_SUM_MAX partial sums (k_gather_sum adds them lane by lane); a bucket with
// more is "hot" (a 0/1-heavy witness piles half of window 0 into one bucket): the sort lists such buckets and k_hot_sum folds each
// one's partial sums with a workgroup-wide tree before the gather
constexpr uint32_t GATHER_SUM_MAX = 32;
constexpr uint32_t HOT_MAX = 4096;                   // hot buckets one k_hot_sum launch takes (more: the lane-by-lane rounds)
__device__ __forceinline__ uint32_t task_len(uint32_t T, uint32_t T_top, int w, int top_w) { return w == top_w ? T_top : T; }
__device__ __forceinline__ uint32_t len_key(uint32_t len) { return len > 255u ? 255u : len; }
// Task length of ONE bucket: a hot bucket (more than GATHER_SUM_MAX tasks of the window's length T) is cut four times finer.  Its
// partial sums meet in k_hot_sum's trees whatever their number, and its tasks are what a launch waits for: on a 0/1-heavy witness the
// hot buckets' T = 80-entry chains (80 x 14 us) were the whole accumulation of a window group whose other buckets hold three entries.
// Every kernel that derives tasks from a bucket size uses bucket_task_len / bucket_tasks (T >= 32: kg::msm_sort_begin clamps it).
// The task length the kernels receive carries the cut in its top two bits (hot_shift: T >> shift for hot buckets; 0 = none, the merged
// sort of window tables -- one launch of 15 n entries hides its hot chains, and four times the partial sums cost it 5-8 %).
constexpr uint32_t T_MASK = 0x3fffffffu;
__device__ __forceinline__ uint32_t t_plain(uint32_t T) { return T & T_MASK; }
__device__ __forceinline__ uint32_t bucket_task_len(uint32_t T, uint32_t v) {
  const uint32_t t = T & T_MASK, sh = T >> 30;
  return v > GATHER_SUM_MAX * t ? t >> sh : t;
}
__device__ __forceinline__ uint32_t bucket_tasks(uint32_t T, uint32_t v) { const uint32_t Tb = bucket_task_len(T, v); return (v + Tb - 1) / Tb; }

__global__ void __launch_bounds__(1024) k_len_scatter(const uint32_t* __restrict__ bsize, const uint32_t* __restrict__ ntask,
                                                      const uint32_t* __restrict__ rel, const uint32_t* __restrict__ base, size_t total, int B,
                                                      uint32_t T0, uint32_t* __restrict__ cursor, uint32_t* __restrict__ task_bkt,
                                                      uint32_t* __restrict__ task_id, uint32_t T_top, int top_w) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t h[LEN_BINS], start[LEN_BINS], fill[LEN_BINS];
  __shared__ uint32_t big_n, big_t[64], big_first[64], big_cnt[64], big_pos[64];
  if (threadIdx.x < LEN_BINS) { h[threadIdx.x] = 0; fill[threadIdx.x] = 0; }
  if (threadIdx.x == 0) big_n = 0;
  __syncthreads();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t T = task_len(T0, T_top, (int)(t / B), top_w);
  uint32_t nt = 0, rem = 0, Tb = t_plain(T);
  if (t < total) {
    nt = ntask[t];
    if (nt) {
      Tb = bucket_task_len(T, bsize[t]);
      rem = bsize[t] - (nt - 1) * Tb;
      atomicAdd(&h[len_key(rem)], 1u);
      if (nt > 1) atomicAdd(&h[len_key(Tb)], nt - 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < LEN_BINS && h[threadIdx.x]) start[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  // a bucket cut into many tasks (0/1-heavy scalars: one bucket of a 2^24-pair witness holds 8 M entries = 10^5 tasks) hands its
  // full-length tasks to the whole workgroup -- written by its own lane they were 5.4 ms of a 20 ms commitment
  constexpr uint32_t BIG = 16, BIG_CAP = 64;
  if (nt) {
    const uint32_t first = base[t / B] + rel[t];
    const uint32_t kf = len_key(Tb);
    uint32_t slot = BIG_CAP;
    if (nt - 1 > BIG) slot = atomicAdd(&big_n, 1u);
    if (slot < BIG_CAP) {
      big_t[slot] = (uint32_t)t; big_first[slot] = first; big_cnt[slot] = nt - 1;
      big_pos[slot] = start[kf] + atomicAdd(&fill[kf], nt - 1);
    } else {
      for (uint32_t sgm = 0; sgm + 1 < nt; ++sgm) {
        const uint32_t pos = start[kf] + atomicAdd(&fill[kf], 1u);
        task_bkt[pos] = (uint32_t)t;
        task_id[pos] = first + sgm;
      }
    }
    const uint32_t kr = len_key(rem);
    const uint32_t pos = start[kr] + atomicAdd(&fill[kr], 1u);
    task_bkt[pos] = (uint32_t)t;
    task_id[pos] = first + nt - 1;
  }
  __syncthreads();
  const uint32_t nb = big_n < BIG_CAP ? big_n : BIG_CAP;
  for (uint32_t b = 0; b < nb; ++b) {
    const uint32_t bt = big_t[b], bf = big_first[b], bc = big_cnt[b], bp = big_pos[b];
    for (uint32_t i = threadIdx.x; i < bc; i += blockDim.x) {
      task_bkt[bp + i] = bt;
      task_id[bp + i] = bf + i;
    }
  }
}

// One workgroup per window over the bucket sizes: bucket starts (exclusive prefix), tasks per bucket and their prefix,
// the window's task total, the largest bucket, and the histogram of task lengths -- everything the task decomposition
// needs from one read of the sizes.
constexpr int BR_NT = 256;
__global__ void __launch_bounds__(BR_NT) k_bucket_rows(const uint32_t* __restrict__ bsize, int B, uint32_t T0, uint32_t* __restrict__ bstart,
                                                      uint32_t* __restrict__ ntask, uint32_t* __restrict__ rel, uint32_t* __restrict__ row_total,
                                                      uint32_t* __restrict__ maxv, uint32_t* __restrict__ ghist, uint32_t T_top, int top_w,
                                                      uint32_t* __restrict__ hot_list, uint32_t hot_cap) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t sh[40], h[LEN_BINS], red[16];
  const int w = blockIdx.x;
  const uint32_t T = task_len(T0, T_top, w, top_w);
  if (threadIdx.x < LEN_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  const int per = (B + BR_NT - 1) / BR_NT;
  const int lo = threadIdx.x * per, hi = lo + per < B ? lo + per : B;
  const uint32_t* src = bsize + (size_t)w * B;
  const bool vec = (per & 3) == 0 && hi - lo == per;       // every lane owns whole 16-byte groups
  uint32_t ssum = 0, tsum = 0, mx = 0, full = 0, mt = 0;     // full: tasks of the full length T (one shared bin: counted per lane, added once); mt: most tasks of a bucket
  auto tally = [&](uint32_t v) {
    if (v) {
      const uint32_t Tb = bucket_task_len(T, v), nt = (v + Tb - 1) / Tb;
      ssum += v; tsum += nt; mx = v > mx ? v : mx; mt = nt > mt ? nt : mt;
      atomicAdd(&h[len_key(v - (nt - 1) * Tb)], 1u);
      if (Tb == t_plain(T)) full += nt - 1;
      else atomicAdd(&h[len_key(Tb)], nt - 1);           // a hot bucket's finer tasks: a bin of their own
    }
  };
  if (vec) {
    for (int b = lo; b < hi; b += 4) { const uint4 q = *reinterpret_cast<const uint4*>(src + b); tally(q.x); tally(q.y); tally(q.z); tally(q.w); }
  } else {
    for (int b = lo; b < hi; ++b) tally(src[b]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) full += __shfl_xor(full, d);
  if ((threadIdx.x & 63) == 0 && full) atomicAdd(&h[len_key(t_plain(T))], full);
  uint32_t total_s, total_t;
  uint32_t run_s = block_exclusive_scan_1024(ssum, sh, total_s);
  uint32_t run_t = block_exclusive_scan_1024(tsum, sh, total_t);
  uint32_t* o_start = bstart + (size_t)w * B;
  uint32_t* o_nt = ntask + (size_t)w * B;
  uint32_t* o_rel = rel + (size_t)w * B;
  if (vec) {
    for (int b = lo; b < hi; b += 4) {
      const uint4 q = *reinterpret_cast<const uint4*>(src + b);
      uint4 st, nt, rl;
      nt.x = bucket_tasks(T, q.x); nt.y = bucket_tasks(T, q.y); nt.z = bucket_tasks(T, q.z); nt.w = bucket_tasks(T, q.w);
      st.x = run_s; st.y = st.x + q.x; st.z = st.y + q.y; st.w = st.z + q.z; run_s = st.w + q.w;
      rl.x = run_t; rl.y = rl.x + nt.x; rl.z = rl.y + nt.y; rl.w = rl.z + nt.z; run_t = rl.w + nt.w;
      *reinterpret_cast<uint4*>(o_start + b) = st;
      *reinterpret_cast<uint4*>(o_nt + b) = nt;
      *reinterpret_cast<uint4*>(o_rel + b) = rl;
    }
  } else {
    for (int b = lo; b < hi; ++b) {
      const uint32_t v = src[b], nt = bucket_tasks(T, v);
      o_start[b] = run_s; o_nt[b] = nt; o_rel[b] = run_t;
      run_s += v; run_t += nt;
    }
  }
  if (mx > GATHER_SUM_MAX * t_plain(T))                 // rare: list this lane's hot buckets (maxv + 1 counts them)
    for (int b = lo; b < hi; ++b)
      if (src[b] > GATHER_SUM_MAX * t_plain(T)) { const uint32_t pos = atomicAdd(maxv + 1, 1u); if (pos < hot_cap) hot_list[pos] = (uint32_t)(w * B + b); }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(mx, d); mx = o > mx ? o : mx; o = __shfl_xor(mt, d); mt = o > mt ? o : mt; }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; if (mt) atomicMax(maxv + 2, mt); }      // maxv[2]: most tasks any bucket has
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = 0;
    for (int i = 0; i < BR_NT / 64; ++i) m = red[i] > m ? red[i] : m;
    if (m) atomicMax(maxv, m);
    row_total[w] = total_t;
  }
  if (threadIdx.x < LEN_BINS && h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
}
// The same for long rows (B >= 8192: the c = 15..17 windows, and the single 2^16-bucket row of a merged sort), cut into
// `nsplit` parts of B / nsplit buckets with a workgroup each: k_bucket_part sums a part (entries, tasks, largest bucket,
// length histogram), k_bucket_fill adds the parts in front of its own and writes starts / task counts / task prefixes.
// One workgroup per row took 80 us for 65536 buckets; sixteen parts take two launches of ~10 us.
__global__ void __launch_bounds__(BR_NT) k_bucket_part(const uint32_t* __restrict__ bsize, int B, uint32_t T0, int nsplit, uint32_t* __restrict__ part,
                                                      uint32_t* __restrict__ maxv, uint32_t* __restrict__ ghist, uint32_t T_top, int top_w,
                                                      uint32_t* __restrict__ hot_list, uint32_t hot_cap) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t sh[40], h[LEN_BINS], red[16];
  const int w = blockIdx.x, k = blockIdx.y, len = B / nsplit;
  const uint32_t T = task_len(T0, T_top, w, top_w);
  if (threadIdx.x < LEN_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  const int per = len / BR_NT;                           // len is a multiple of 4 * BR_NT (B >= 8192, nsplit <= B / 4096)
  const uint32_t* src = bsize + (size_t)w * B + (size_t)k * len + (size_t)threadIdx.x * per;
  uint32_t ssum = 0, tsum = 0, mx = 0, full = 0, mt = 0;
  auto tally = [&](uint32_t v) {
    if (v) {
      const uint32_t Tb = bucket_task_len(T, v), nt = (v + Tb - 1) / Tb;
      ssum += v; tsum += nt; mx = v > mx ? v : mx; mt = nt > mt ? nt : mt;
      atomicAdd(&h[len_key(v - (nt - 1) * Tb)], 1u);
      if (Tb == t_plain(T)) full += nt - 1;
      else atomicAdd(&h[len_key(Tb)], nt - 1);           // a hot bucket's finer tasks: a bin of their own
    }
  };
  for (int b = 0; b < per; b += 4) { const uint4 q = *reinterpret_cast<const uint4*>(src + b); tally(q.x); tally(q.y); tally(q.z); tally(q.w); }
  if (mx > GATHER_SUM_MAX * t_plain(T))                 // rare: list this lane's hot buckets (maxv + 1 counts them)
    for (int b = 0; b < per; ++b)
      if (src[b] > GATHER_SUM_MAX * t_plain(T)) {
        const uint32_t pos = atomicAdd(maxv + 1, 1u);
        if (pos < hot_cap) hot_list[pos] = (uint32_t)((size_t)w * B + (size_t)k * len + (size_t)threadIdx.x * per + b);
      }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { full += __shfl_xor(full, d); uint32_t o = __shfl_xor(mx, d); mx = o > mx ? o : mx; o = __shfl_xor(mt, d); mt = o > mt ? o : mt; }
  if ((threadIdx.x & 63) == 0) { if (full) atomicAdd(&h[len_key(t_plain(T))], full); red[threadIdx.x >> 6] = mx; if (mt) atomicMax(maxv + 2, mt); }
  uint32_t total_s, total_t;
  block_exclusive_scan_1024(ssum, sh, total_s);
  block_exclusive_scan_1024(tsum, sh, total_t);
  if (threadIdx.x == 0) {
    uint32_t m = 0;
    for (int i = 0; i < BR_NT / 64; ++i) m = red[i] > m ? red[i] : m;
    if (m) atomicMax(maxv, m);
    part[((size_t)w * nsplit + k) * 2] = total_s;
    part[((size_t)w * nsplit + k) * 2 + 1] = total_t;
  }
  if (threadIdx.x < LEN_BINS && h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(BR_NT) k_bucket_fill(const uint32_t* __restrict__ bsize, int B, uint32_t T0, int nsplit, const uint32_t* __restrict__ part,
                                                      uint32_t* __restrict__ bstart, uint32_t* __restrict__ ntask, uint32_t* __restrict__ rel,
                                                      uint32_t* __restrict__ row_total, uint32_t T_top, int top_w) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t sh[40];
  const int w = blockIdx.x, k = blockIdx.y, len = B / nsplit;
  const uint32_t T = task_len(T0, T_top, w, top_w);
  uint32_t base_s = 0, base_t = 0;
  for (int j = 0; j < k; ++j) { base_s += part[((size_t)w * nsplit + j) * 2]; base_t += part[((size_t)w * nsplit + j) * 2 + 1]; }
  const int per = len / BR_NT;
  const size_t off = (size_t)w * B + (size_t)k * len + (size_t)threadIdx.x * per;
  const uint32_t* src = bsize + off;
  uint32_t ssum = 0, tsum = 0;
  for (int b = 0; b < per; b += 4) {
    const uint4 q = *reinterpret_cast<const uint4*>(src + b);
    ssum += q.x + q.y + q.z + q.w;
    tsum += bucket_tasks(T, q.x) + bucket_tasks(T, q.y) + bucket_tasks(T, q.z) + bucket_tasks(T, q.w);
  }
  uint32_t total_s, total_t;
  uint32_t run_s = base_s + block_exclusive_scan_1024(ssum, sh, total_s);
  uint32_t run_t = base_t + block_exclusive_scan_1024(tsum, sh, total_t);
  for (int b = 0; b < per; b += 4) {
    const uint4 q = *reinterpret_cast<const uint4*>(src + b);
    uint4 st, nt, rl;
    nt.x = bucket_tasks(T, q.x); nt.y = bucket_tasks(T, q.y); nt.z = bucket_tasks(T, q.z); nt.w = bucket_tasks(T, q.w);
    st.x = run_s; st.y = st.x + q.x; st.z = st.y + q.y; st.w = st.z + q.z; run_s = st.w + q.w;
    rl.x = run_t; rl.y = rl.x + nt.x; rl.z = rl.y + nt.y; rl.w = rl.z + nt.z; run_t = rl.w + nt.w;
    *reinterpret_cast<uint4*>(bstart + off + b) = st;
    *reinterpret_cast<uint4*>(ntask + off + b) = nt;
    *reinterpret_cast<uint4*>(rel + off + b) = rl;
  }
  if (k == nsplit - 1 && threadIdx.x == 0) row_total[w] = base_t + total_t;
}
// k_row_bases + k_len_scan in one launch (one wave): window task bases and the descending-length cursors
// host_info: the two result words go straight into pinned host memory (a copy kernel at the default wave priority crawled
// beside a resident accumulation: 4 us alone, 69 us there)
__global__ void __launch_bounds__(64) k_task_bases(const uint32_t* __restrict__ row_total, int W, uint32_t* __restrict__ base,
                                                   const uint32_t* __restrict__ maxv, uint32_t* __restrict__ info,
                                                   const uint32_t* __restrict__ ghist, uint32_t* __restrict__ cursor, uint32_t* __restrict__ host_info) {
  KG_SERVICE_PRIO();
  const int lane = threadIdx.x;
  if (lane == 0) {
    uint32_t run = 0;
    for (int w = 0; w < W; ++w) { base[w] = run; run += row_total[w]; }
    base[W] = run;
    const uint32_t mx = *maxv, hot = maxv[1];        // largest bucket; buckets with more than GATHER_SUM_MAX tasks
    info[0] = run;
    info[1] = mx;
    host_info[0] = run;
    host_info[1] = mx;
    host_info[2] = hot;
    host_info[3] = maxv[2];                          // most tasks any bucket has (exact: the top window's tasks are longer)
  }
  static_assert(LEN_BINS == 256, "four bins per lane");
  const uint32_t h0 = ghist[4 * lane], h1 = ghist[4 * lane + 1], h2 = ghist[4 * lane + 2], h3 = ghist[4 * lane + 3];
  uint32_t inc = h0 + h1 + h2 + h3;                 // inclusive suffix sum over lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_down(inc, d); if (lane + d < 64) inc += o; }
  const uint32_t above = inc - (h0 + h1 + h2 + h3); // tasks with a key in a higher lane's bins
  cursor[4 * lane + 3] = above;
  cursor[4 * lane + 2] = above + h3;
  cursor[4 * lane + 1] = above + h3 + h2;
  cursor[4 * lane] = above + h3 + h2 + h1;
}

// task id -> (window, bucket, segment index)
__device__ __forceinline__ void locate_task(const Level& L, int W, int B, uint32_t t, int& w, int& b, uint32_t& seg) {
  int lo = 0, hi = W;                        // last w with base[w] <= t
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (L.base[mid] <= t) lo = mid; else hi = mid; }
  w = lo;
  const uint32_t x = t - L.base[w];
  const uint32_t* rel = L.rel + (size_t)w * B;
  lo = 0; hi = B;                            // last b with rel[b] <= x  (that bucket is never empty)
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (rel[mid] <= x) lo = mid; else hi = mid; }
  b = lo;
  seg = x - rel[b];
}

// round 1: lists of (base index | sign) -> partial XYZZ per task; lane p runs the p-th longest task.
// Up to MAX_FUSED base arrays that share one scalar sort (the prover's a, b_g1 and l queries against z) are accumulated
// by ONE launch, waves dealt to the arrays in turn: with a single array a 2^18-pair MSM has 4352 waves for 4096 resident
// wave slots -- one round, no refill, and a tail at one wave per SIMD (69 % of the four-wave issue rate); three arrays
// make 3.2 rounds (measured: 10.2 -> 13 G additions/s).
constexpr int MAX_FUSED = 3;
struct AccSets {
  int nsets;
  const uint32_t* pb[MAX_FUSED];      // packed bases of each array
  uint32_t idx_off[MAX_FUSED];        // scalars in front of the array (shared sort, z = x || w)
  uint32_t* partial[MAX_FUSED];       // partial sums, one per task
  uint32_t tab_n[MAX_FUSED];          // merged sort: points per window of the array's table (pb = the table)
  uint8_t fmt64[MAX_FUSED];           // the array is in the 64-byte resident form (BaseIO::load_point64)
};
// G2: the compiler lands on 256 VGPRs + 1 AGPR = one wave per SIMD; asking for two waves costs a few spilled registers
// and buys the second wave (the issue rate of this code at one wave per SIMD is ~69 % of its rate at four)
// where a task's partial sum goes: the one-lane array-of-structures layout, also when a lane pair computed it (the later
// rounds and the gather read that layout either way)
template <class F> struct AccStore {
  static __device__ __forceinline__ void store(uint32_t* base, size_t i, const XYZZ<F>& p) { PointAoS<F>::store(base, i, p); }
};
template <class G> struct AccStore<Fp2S<G>> {
  static __device__ __forceinline__ void store(uint32_t* base, size_t i, const XYZZ<Fp2S<G>>& p) {
    uint32_t* dst = base + i * 72 + 9 * Fp2S<G>::half();          // PointAoS<Fp2<G>>: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
#pragma unroll
    for (int k = 0; k < 9; ++k) { dst[k] = p.x.v.l[k]; dst[18 + k] = p.y.v.l[k]; dst[36 + k] = p.zz.v.l[k]; dst[54 + k] = p.zzz.v.l[k]; }
  }
};
#ifdef KG_EXP_ACC_WAVES      // experiment (tools/dbg/build_variants.sh): base-field accumulation capped for this many waves per SIMD
template <class F> struct AccWaves { static constexpr int MIN = KG_EXP_ACC_WAVES; };
#else
template <class F> struct AccWaves { static constexpr int MIN = 1; };
#endif
template <class G> struct AccWaves<Fp2<G>> { static constexpr int MIN = 2; };
template <class F>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AccWaves<F>::MIN))) k_acc_tasks(AccSets A, const uint32_t* __restrict__ sorted,
                                                  const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ bsize, Level L,
                                                  const uint32_t* __restrict__ task_bkt, const uint32_t* __restrict__ task_id,
                                                  size_t n, int W, int B, uint32_t T0, size_t pstride, int mshift, uint32_t T_top, int top_w) {
  const int set = A.nsets > 1 ? (int)(blockIdx.x % (unsigned)A.nsets) : 0;
  const uint32_t p = ((A.nsets > 1 ? blockIdx.x / (unsigned)A.nsets : blockIdx.x) * blockDim.x + threadIdx.x) / Lanes<F>::N;   // Fq2: a lane pair per task
  if (p >= L.base[W]) return;
  const uint32_t* __restrict__ pbases = A.pb[0];
  uint32_t idx_off = A.idx_off[0], tab_n = A.tab_n[0];
  bool fmt64 = A.fmt64[0] != 0;
  uint32_t* __restrict__ partial = A.partial[0];
#pragma unroll
  for (int k = 1; k < MAX_FUSED; ++k)
    if (set == k) { pbases = A.pb[k]; idx_off = A.idx_off[k]; partial = A.partial[k]; tab_n = A.tab_n[k]; fmt64 = A.fmt64[k] != 0; }
  const size_t bi = task_bkt[p];
  const uint32_t t = task_id[p];
  const int w = (int)(bi / B);
  const uint32_t seg = t - L.base[w] - L.rel[bi];
  const uint32_t len_all = bsize[bi];
  const uint32_t T = bucket_task_len(task_len(T0, T_top, w, top_w), len_all);
  const uint32_t lo = seg * T, hi = lo + T < len_all ? lo + T : len_all;
  const uint32_t* list = sorted + (size_t)w * n + bstart[bi];
  constexpr int PW = 2 * BaseIO<F>::PE;
  XYZZ<F> acc = XYZZ<F>::identity();
  for (uint32_t j = lo; j < hi; ++j) {
    const uint32_t e = list[j];
    uint32_t idx = e & 0x7fffffffu;
    size_t row = 0;                                      // merged sort: the entry names (window, scalar); the table row of the window
    if (mshift) { row = (size_t)(idx >> mshift) * tab_n; idx &= (1u << mshift) - 1u; }
    if (idx < idx_off) continue;                         // scalars in front of this base array (shared sort, z = x || w)
    Affine<F> a;
#ifdef KG_EXP_ACC_CACHED      // timing experiment (wrong sums): every base comes out of a 4 MiB region -- what the accumulation would run at if its gathers never left the cache
    const size_t at = (row + (idx - idx_off)) & 0xffffu;
#else
    const size_t at = row + (idx - idx_off);
#endif
    if (fmt64 ? BaseIO<F>::load_point64(pbases + at * (2 * BaseIO<F>::PK), a.x, a.y) : BaseIO<F>::load_point(pbases + at * PW, a.x, a.y))
      continue;                                          // identity base (msm.rs:58-64 adds it as a no-op)
    acc = add_mixed_signed(acc, a, (e & 0x80000000u) != 0);
  }
  AccStore<F>::store(partial, t, acc);
}

// Experiment (KG_ACC_PREFETCH=1; off by default -- measured level, see EXPERIMENTS.md Part I section 10): the accumulation with the
// NEXT base on its way while the current addition runs, without a register for it.  gfx950's global_load_lds_dwordx4 writes 16 bytes
// per lane straight into LDS (address = M0 base + 16 x lane); request k of lane L fetches piece L & 3 of the point of lane
// 16 k + (L >> 2), so a quad covers one 64-byte point with one contiguous request (16 lines per instruction instead of 64, every line
// requested once instead of by four instructions) and the point of lane l lands contiguously at LDS slot 64 (l >> 4) + 4 (l & 15).
// Per entry: read the point out of LDS, request the next one (its list entry was requested an iteration earlier), add.  The loop is
// wave-uniform (every lane fetches for its quad's owners until the longest task of the wave is done); a lane adds while its own task
// lasts.  Base-field curves in the 64-byte resident form only.  Why it was tried: a 2^24-pair array is 1 GiB of bases -- no cache holds
// it -- and with every gather forced into a 4 MiB region (-DKG_EXP_ACC_CACHED) the 2^24 accumulation takes 13.97 instead of 15.16 ms.
template <class F>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AccWaves<F>::MIN))) k_acc_tasks_q(AccSets A, const uint32_t* __restrict__ sorted,
                                                  const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ bsize, Level L,
                                                  const uint32_t* __restrict__ task_bkt, const uint32_t* __restrict__ task_id,
                                                  size_t n, int W, int B, uint32_t T0, size_t pstride, int mshift, uint32_t T_top, int top_w) {
  __shared__ uint4 pf[4 * 64];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr uint32_t NONE = 0xffffffffu;
  const int lane = threadIdx.x;
  const int set = A.nsets > 1 ? (int)(blockIdx.x % (unsigned)A.nsets) : 0;
  const uint32_t p = (A.nsets > 1 ? blockIdx.x / (unsigned)A.nsets : blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = p < L.base[W];
  const uint32_t* __restrict__ pbases = A.pb[0];
  uint32_t idx_off = A.idx_off[0], tab_n = A.tab_n[0];
  uint32_t* __restrict__ partial = A.partial[0];
#pragma unroll
  for (int k = 1; k < MAX_FUSED; ++k)
    if (set == k) { pbases = A.pb[k]; idx_off = A.idx_off[k]; partial = A.partial[k]; tab_n = A.tab_n[k]; }
  uint32_t t = 0, len = 0;
  const uint32_t* list = sorted;
  if (live) {
    const size_t bi = task_bkt[p];
    t = task_id[p];
    const int w = (int)(bi / B);
    const uint32_t seg = t - L.base[w] - L.rel[bi];
    const uint32_t len_all = bsize[bi];
    const uint32_t T = bucket_task_len(task_len(T0, T_top, w, top_w), len_all);
    const uint32_t lo = seg * T, hi = lo + T < len_all ? lo + T : len_all;
    list = sorted + (size_t)w * n + bstart[bi] + lo;
    len = hi - lo;
  }
  uint32_t mx = len;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d); mx = o > mx ? o : mx; }
  // entry -> 32-bit point index in the array (window-table row included); NONE: no entry, or a scalar in front of this array
  auto locate = [&](uint32_t e) -> uint32_t {
    if (e == NONE) return NONE;
    uint32_t idx = e & 0x7fffffffu, row = 0;
    if (mshift) { row = (idx >> mshift) * tab_n; idx &= (1u << mshift) - 1u; }
    return idx >= idx_off ? row + (idx - idx_off) : NONE;
  };
  auto request = [&](uint32_t e) {                       // all 64 lanes, each for the owners of its quad's points
    const uint32_t at = locate(e);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t ak = (uint32_t)__shfl((int)at, 16 * k + (lane >> 2));
      if (ak != NONE) {
        const uint4* src = reinterpret_cast<const uint4*>(pbases + (size_t)ak * (2 * BaseIO<F>::PK)) + (lane & 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(pf + k * 64), 16, 0, 0);
      }
    }
  };
  XYZZ<F> acc = XYZZ<F>::identity();
  uint32_t e1 = len > 0 ? list[0] : NONE, e2 = len > 1 ? list[1] : NONE;
  request(e1);
  const int slot = 64 * (lane >> 4) + 4 * (lane & 15);
  for (uint32_t j = 0; j < mx; ++j) {
    uint32_t wd[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint4 v = pf[slot + k]; wd[4 * k] = v.x; wd[4 * k + 1] = v.y; wd[4 * k + 2] = v.z; wd[4 * k + 3] = v.w; }
    const uint32_t e = e1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads above are done before the next requests overwrite the buffer
    e1 = e2;
    request(e1);
    e2 = j + 2 < len ? list[j + 2] : NONE;
    if (locate(e) == NONE) continue;
    Affine<F> a;
    if (BaseIO<F>::point64_from_words(wd, a.x, a.y)) continue;        // identity base
    acc = add_mixed_signed(acc, a, (e & 0x80000000u) != 0);
  }
  if (live) AccStore<F>::store(partial, t, acc);
}

// which launches take k_acc_tasks_q (KG_ACC_PREFETCH=1): base-field arrays in the 64-byte form with at least 2^KG_ACC_PREFETCH_LOG
// bases; 0 (default): none
template <class F> struct PfField { using T = Fq; static constexpr bool ok = false; };      // Fq2: never launched (the alias only keeps the launch expression well-formed)
template <class P> struct PfField<Fp<P>> { using T = Fp<P>; static constexpr bool ok = true; };
template <class F>
static int acc_prefetch(const AccSets& A, int njobs, size_t nbases) {
  if (!PfField<F>::ok) return 0;
  const int mode = tuning().acc_prefetch, from_log = tuning().acc_prefetch_log;
  if (mode == 0 || nbases < ((size_t)1 << from_log)) return 0;
  for (int k = 0; k < njobs; ++k)
    if (!A.fmt64[k]) return 0;
  return mode;
}

// final: dense bucket array for the halving reduction
template <class F>
__global__ void __launch_bounds__(256) k_gather_buckets(const uint32_t* __restrict__ pin, size_t in_stride, Level L, int W, int B,
                                                        uint32_t* __restrict__ buckets) {
  KG_REDUCE_PRIO();
  const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / Lanes<F>::N;
  const size_t total = (size_t)W * B;
  if (t >= total) return;
  const int w = (int)(t / B);
  XYZZ<F> p = XYZZ<F>::identity();
  if (L.cnt[t]) p = PointAoS<F>::load(pin, (size_t)L.base[w] + L.rel[t]);
  PointIO<F>::store(buckets, total, t, p);
}

// Buckets cut into a few tasks (every bucket of a merged sort: W n / B entries in tasks of T): the dense bucket array straight
// from the partial sums, a lane (lane pair for G2) per bucket adding its <= GATHER_SUM_MAX partial sums -- instead of a
// partial-sum round (task count, row scan, bases, k_sum_tasks) followed by the gather: six launches and ~130 us less on the
// reduction queue per MSM.  KF = F, or the lane-pair form of Fq2 reading the one-lane layout k_acc_tasks<Fq2> wrote.
template <class F, class KF> struct PartialIO {
  static __device__ __forceinline__ XYZZ<KF> load(const uint32_t* base, size_t i) { return PointAoS<F>::load(base, i); }
};
template <class G> struct PartialIO<Fp2<G>, Fp2S<G>> {      // PointAoS<Fp2<G>>: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1, nine words each
  static __device__ __forceinline__ XYZZ<Fp2S<G>> load(const uint32_t* base, size_t i) {
    const uint32_t* src = base + i * 72 + 9 * Fp2S<G>::half();
    XYZZ<Fp2S<G>> p;
#pragma unroll
    for (int k = 0; k < 9; ++k) { p.x.v.l[k] = src[k]; p.y.v.l[k] = src[18 + k]; p.zz.v.l[k] = src[36 + k]; p.zzz.v.l[k] = src[54 + k]; }
    return p;
  }
};
// ---------------------------------------------------------------------------------------------------
// bucket reduction by halving.  Arrays per window at level s: A (pair sums so far) and T_0..T_{s-1}
// (odd-index sums), each of length 2*n_out; the level emits A', the halved T's and a new T_s = odd items of A.
// After log2(B) levels every array has length 1: T_l = sum of buckets whose (index) bit l is set, A = sum of
// all buckets; sum_b (b+1)*B_b = A + sum_l 2^l T_l.
// Layout: point (window w, array a, item i) at index (w * narr + a) * len + i of a PointIO buffer.
// ---------------------------------------------------------------------------------------------------
// One point of a PointIO buffer (structure of arrays), coordinates read / written on demand (add_xyzz_stream).  Buffer
// addressing: the descriptor and the limb plane's offset are wave-uniform (SGPRs), the item's byte offset is ONE 32-bit VGPR
// per point -- flat loads cost a 64-bit address pair per limb plane (72 planes: the compiler kept ~70 VGPRs of addresses
// live).  The buffers stay far below the 4 GiB a descriptor spans (72 planes x 15 x 2^17 items x 4 B = 566 MB for G2, c = 18).
template <class F> struct SoaLimbs;
template <class P> struct SoaLimbs<Fp<P>> {
  static __device__ __forceinline__ Fp<P> load(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off) {     // stride4: bytes per plane
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, (plane + (uint32_t)k) * stride4, 0);
    return r;
  }
  static __device__ __forceinline__ void store(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off, const Fp<P>& a) {
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b32(a.l[k], rs, off, (plane + (uint32_t)k) * stride4, 0);
  }
};
template <class G> struct SoaLimbs<Fp2S<G>> {       // the pair's halves: c0 planes, then c1 planes (RawIO<Fp2S>); the half goes into the lane offset
  static __device__ __forceinline__ Fp2S<G> load(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off) {
    return {SoaLimbs<G>::load(rs, plane, stride4, off + 9u * (uint32_t)Fp2S<G>::half() * stride4)};
  }
  static __device__ __forceinline__ void store(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off, const Fp2S<G>& a) {
    SoaLimbs<G>::store(rs, plane, stride4, off + 9u * (uint32_t)Fp2S<G>::half() * stride4, a.v);
  }
};
template <class F> struct SoaSrc {
  BufRsrc rs; uint32_t stride4, off;                 // off = item * 4
  static constexpr uint32_t E = RawIO<F>::NW;
  __device__ __forceinline__ F x() const { return SoaLimbs<F>::load(rs, 0, stride4, off); }
  __device__ __forceinline__ F y() const { return SoaLimbs<F>::load(rs, E, stride4, off); }
  __device__ __forceinline__ F zz() const { return SoaLimbs<F>::load(rs, 2 * E, stride4, off); }
  __device__ __forceinline__ F zzz() const { return SoaLimbs<F>::load(rs, 3 * E, stride4, off); }
};
template <class F> struct SoaDst {
  BufRsrc rs; uint32_t stride4, off;
  static constexpr uint32_t E = RawIO<F>::NW;
  __device__ __forceinline__ void x(const F& v) { SoaLimbs<F>::store(rs, 0, stride4, off, v); }
  __device__ __forceinline__ void y(const F& v) { SoaLimbs<F>::store(rs, E, stride4, off, v); }
  __device__ __forceinline__ void zz(const F& v) { SoaLimbs<F>::store(rs, 2 * E, stride4, off, v); }
  __device__ __forceinline__ void zzz(const F& v) { SoaLimbs<F>::store(rs, 3 * E, stride4, off, v); }
};
template <class F> struct HalveWaves { static constexpr int MIN = 5; };
template <class G> struct HalveWaves<Fp2S<G>> { static constexpr int MIN = 4; };     // 110 VGPRs as it comes (was 170); nothing fits beside a G2 accumulation anyway
#ifndef KG_HALVE_ATTR
#define KG_HALVE_ATTR __attribute__((amdgpu_waves_per_eu(HalveWaves<F>::MIN)))
#endif
// Five waves per SIMD = 96 VGPRs: what four resident accumulation waves (4 x 104) leave free, so a halving level runs beside
// an accumulation instead of waiting for its tail.  With both operands loaded up front the kernel took 160.
template <class F>
__global__ void __launch_bounds__(64) KG_HALVE_ATTR k_halve(const uint32_t* in, size_t in_stride, uint32_t* out, size_t out_stride,
                                              int W, int narr_in, uint32_t n_out) {
  KG_REDUCE_PRIO();
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<F>::N;       // < 2^27 items (W * B <= 15 * 2^17)
  const uint32_t per_w = (uint32_t)narr_in * n_out;
  if (t >= per_w * (uint32_t)W) return;
  const uint32_t w = t / per_w;
  const uint32_t r = t % per_w;
  const uint32_t a = r / n_out;
  const uint32_t i = r % n_out;
  const uint32_t narr_out = (uint32_t)narr_in + 1;
  const uint32_t src = (w * narr_in + a) * (2 * n_out) + 2 * i;
  const BufRsrc rin = soa_rsrc(in), rout = soa_rsrc(out);
  const uint32_t is4 = (uint32_t)in_stride * 4u, os4 = (uint32_t)out_stride * 4u;
  const SoaSrc<F> p0{rin, is4, src * 4u}, p1{rin, is4, src * 4u + 4u};
  SoaDst<F> sum{rout, os4, ((w * narr_out + a) * n_out + i) * 4u};
  add_xyzz_stream<F>(p0, p1, sum);
  if (a == 0) {
    SoaDst<F> odd{rout, os4, ((w * narr_out + narr_in) * n_out + i) * 4u};
    copy_xyzz_stream<F>(p1, odd);
  }
}

// The dense bucket array and the first halving level in one launch (buckets that own a single partial sum: every bucket of a
// uniform input): lane i of window w reads the partial sums of buckets 2i and 2i + 1 where the accumulation left them
// (array-of-structures, by task id) and writes level 1 -- the pair sum and the odd item -- instead of k_gather_buckets
// writing W * B points that k_halve reads back: one launch and 2 x 75 MB of traffic less per 2^20-pair MSM.
template <class F> struct AosSrc;                   // a partial sum in the PointAoS layout the accumulation writes, read coordinate by coordinate
template <class P> struct AosSrc<Fp<P>> {
  BufRsrc rs; uint32_t off; bool valid;             // off: byte offset of the point; !valid: an empty bucket (the identity)
  __device__ __forceinline__ Fp<P> get(int coord) const {
    Fp<P> r = Fp<P>::zero();
    if (valid) {
#pragma unroll
      for (int k = 0; k < 9; ++k) r.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off + (uint32_t)(coord * 9 + k) * 4u, 0, 0);
    }
    return r;
  }
  __device__ __forceinline__ Fp<P> x() const { return get(0); }
  __device__ __forceinline__ Fp<P> y() const { return get(1); }
  __device__ __forceinline__ Fp<P> zz() const { return get(2); }
  __device__ __forceinline__ Fp<P> zzz() const { return get(3); }
};
template <class G> struct AosSrc<Fp2S<G>> {         // PointAoS<Fp2<G>>: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1, nine words each; a lane reads its half
  BufRsrc rs; uint32_t off; bool valid;
  __device__ __forceinline__ Fp2S<G> get(int coord) const {
    Fp2S<G> r = Fp2S<G>::zero();
    if (valid) {
      const uint32_t o = off + (uint32_t)(coord * 18 + 9 * Fp2S<G>::half()) * 4u;
#pragma unroll
      for (int k = 0; k < 9; ++k) r.v.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, o + (uint32_t)k * 4u, 0, 0);
    }
    return r;
  }
  __device__ __forceinline__ Fp2S<G> x() const { return get(0); }
  __device__ __forceinline__ Fp2S<G> y() const { return get(1); }
  __device__ __forceinline__ Fp2S<G> zz() const { return get(2); }
  __device__ __forceinline__ Fp2S<G> zzz() const { return get(3); }
};
template <class KF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_gather_halve(const uint32_t* pin, Level L, int W, int B, uint32_t* out, size_t out_stride) {
  KG_REDUCE_PRIO();
  constexpr uint32_t NWB = (uint32_t)PointIO<KF>::NW * 4u;       // bytes per partial sum (PointIO<Fp2S>::NW counts both halves of the lane pair)
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<KF>::N;
  const uint32_t n_out = (uint32_t)B / 2;
  if (t >= n_out * (uint32_t)W) return;
  const uint32_t w = t / n_out, i = t % n_out;
  const uint32_t b0 = w * (uint32_t)B + 2 * i;
  const uint32_t first = L.base[w];
  const uint2 cnt = *reinterpret_cast<const uint2*>(L.cnt + b0), rel = *reinterpret_cast<const uint2*>(L.rel + b0);
  const BufRsrc rin = soa_rsrc(pin), rout = soa_rsrc(out);
  const AosSrc<KF> p0{rin, (first + rel.x) * NWB, cnt.x != 0}, p1{rin, (first + rel.y) * NWB, cnt.y != 0};
  const uint32_t os4 = (uint32_t)out_stride * 4u;
  SoaDst<KF> sum{rout, os4, ((w * 2) * n_out + i) * 4u}, odd{rout, os4, ((w * 2 + 1) * n_out + i) * 4u};
  add_xyzz_stream<KF>(p0, p1, sum);
  copy_xyzz_stream<KF>(p1, odd);
}

// Buckets cut into a few tasks (every bucket of a merged sort; skewed inputs): the dense bucket array straight from the partial
// sums, a lane (lane pair for G2) per bucket.  The running sum lives in the bucket's own slot of the output array and every
// addition streams both operands (add_xyzz_stream is safe in place on its first operand): ~90 VGPRs instead of 141, so the
// gather of a proof's merged MSMs runs beside the next accumulation.
template <class F, class KF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_gather_sum(const uint32_t* pin, Level L, int W, int B, uint32_t* buckets) {
  KG_REDUCE_PRIO();
  constexpr uint32_t NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<KF>::N;
  const uint32_t total = (uint32_t)W * (uint32_t)B;
  if (t >= total) return;
  const uint32_t w = t / (uint32_t)B;
  uint32_t cnt = L.cnt[t];
  if (cnt > GATHER_SUM_MAX) cnt = 1;                   // a hot bucket: k_hot_sum left its sum in the first partial's place
  const BufRsrc rin = soa_rsrc(pin), rout = soa_rsrc(buckets);
  const uint32_t first = cnt ? L.base[w] + L.rel[t] : 0u;
  SoaDst<KF> dst{rout, total * 4u, t * 4u};
  const SoaSrc<KF> cur{rout, total * 4u, t * 4u};
  copy_xyzz_stream<KF>(AosSrc<KF>{rin, first * NWB, cnt != 0}, dst);
  for (uint32_t j = 1; j < cnt; ++j) {
    KG_STREAM_FENCE();
    add_xyzz_stream<KF>(cur, AosSrc<KF>{rin, (first + j) * NWB, true}, dst);
  }
}

// round r > 1: partial sums of the previous round (grouped by bucket through Lin) -> fewer partial sums.  Like k_gather_sum the
// running sum lives in its output slot (array of structures here) and both operands are streamed.
template <class F> struct AosDst;
template <class P> struct AosDst<Fp<P>> {
  BufRsrc rs; uint32_t off;
  __device__ __forceinline__ void put(int coord, const Fp<P>& v) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b32(v.l[k], rs, off + (uint32_t)(coord * 9 + k) * 4u, 0, 0);
  }
  __device__ __forceinline__ void x(const Fp<P>& v) const { put(0, v); }
  __device__ __forceinline__ void y(const Fp<P>& v) const { put(1, v); }
  __device__ __forceinline__ void zz(const Fp<P>& v) const { put(2, v); }
  __device__ __forceinline__ void zzz(const Fp<P>& v) const { put(3, v); }
};
template <class G> struct AosDst<Fp2S<G>> {
  BufRsrc rs; uint32_t off;
  __device__ __forceinline__ void put(int coord, const Fp2S<G>& v) const {
    const uint32_t o = off + (uint32_t)(coord * 18 + 9 * Fp2S<G>::half()) * 4u;
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b32(v.v.l[k], rs, o + (uint32_t)k * 4u, 0, 0);
  }
  __device__ __forceinline__ void x(const Fp2S<G>& v) const { put(0, v); }
  __device__ __forceinline__ void y(const Fp2S<G>& v) const { put(1, v); }
  __device__ __forceinline__ void zz(const Fp2S<G>& v) const { put(2, v); }
  __device__ __forceinline__ void zzz(const Fp2S<G>& v) const { put(3, v); }
};
template <class KF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_sum_tasks(const uint32_t* pin, Level Lin, Level L, int W, int B, uint32_t T2,
                                                  uint32_t* pout) {
  KG_REDUCE_PRIO();
  constexpr uint32_t NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<KF>::N;
  if (t >= L.base[W]) return;
  int w, b;
  uint32_t seg;
  locate_task(L, W, B, t, w, b, seg);
  const size_t bi = (size_t)w * B + b;
  const uint32_t len_all = Lin.cnt[bi];
  const uint32_t lo = seg * T2, hi = lo + T2 < len_all ? lo + T2 : len_all;
  const uint32_t first = Lin.base[w] + Lin.rel[bi];
  const BufRsrc rin = soa_rsrc(pin), rout = soa_rsrc(pout);
  const AosDst<KF> dst{rout, t * NWB};
  const AosSrc<KF> cur{rout, t * NWB, true};
  copy_xyzz_stream<KF>(AosSrc<KF>{rin, (first + lo) * NWB, true}, dst);
  for (uint32_t j = lo + 1; j < hi; ++j) {
    KG_STREAM_FENCE();
    add_xyzz_stream<KF>(cur, AosSrc<KF>{rin, (first + j) * NWB, true}, dst);
  }
}

// raw internal XYZZ -> ABI words (x | y | zz | zzz), array-of-structures for the D2H copy
template <class P>
__device__ __forceinline__ void export_el(const Fp<P>& a, uint64_t* dst) {
  uint32_t w[8];
  to_ref(a, w);
  store_words(dst, 0, w);
}
template <class F>
__device__ __forceinline__ void export_el(const Fp2<F>& a, uint64_t* dst) { export_el(a.c0, dst); export_el(a.c1, dst + 4); }
template <class F>
__device__ __forceinline__ void export_el(const Fp2S<F>& a, uint64_t* dst) { export_el(a.v, dst + 4 * Fp2S<F>::half()); }
// ---- fused tail of the reduction --------------------------------------------------------------------------------
// Once the arrays are short (L <= TailCfg::L items) the remaining log2(L) levels run inside ONE launch: a workgroup per
// (window, array) keeps its items in LDS and walks the levels with barriers instead of kernel launches, then converts its
// results to the ABI form itself (the export).  Array 0 (the pair sums A) also spawns the new odd-index arrays T_s, T_s+1, ...
// and sums them on the lanes the halving frees: step t reads t live arrays of 2 * (L >> t) items and writes t + 1 arrays of
// L >> t items (array k at item k * (L >> t); the odd items of A become array t), i.e. t * (L >> t) pairs <= L / 2 lanes.
// Two LDS images used in turn (odd steps write A: L items, even steps write B: 3 L / 4 items), so a step's operands are read
// coordinate by coordinate while other lanes already write (add_xyzz_stream: the kernel fits the 96 VGPRs a resident
// accumulation leaves free; with both operands in registers and one image shrinking in place it took 159) and a step needs one
// barrier, not two.
// LDS image: structure of arrays, word k of lane-item q at img[k * stride + q] (consecutive lanes, consecutive banks); an Fq2
// item is two lane-items (the pair's halves).
template <class P> __device__ __forceinline__ uint32_t (&tail_limbs(Fp<P>& a))[9] { return a.l; }
template <class G> __device__ __forceinline__ uint32_t (&tail_limbs(Fp2S<G>& a))[9] { return a.v.l; }
template <class P> __device__ __forceinline__ const uint32_t (&tail_limbs(const Fp<P>& a))[9] { return a.l; }
template <class G> __device__ __forceinline__ const uint32_t (&tail_limbs(const Fp2S<G>& a))[9] { return a.v.l; }
template <class F> struct LdsPt {                   // one lane-item of an LDS image, coordinates read / written on demand
  uint32_t* img; uint32_t stride, li;
  __device__ __forceinline__ F get(int coord) const {
    F r;
#pragma unroll
    for (int k = 0; k < 9; ++k) tail_limbs(r)[k] = img[(uint32_t)(coord * 9 + k) * stride + li];
    return r;
  }
  __device__ __forceinline__ void put(int coord, const F& v) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) img[(uint32_t)(coord * 9 + k) * stride + li] = tail_limbs(v)[k];
  }
  __device__ __forceinline__ F x() const { return get(0); }
  __device__ __forceinline__ F y() const { return get(1); }
  __device__ __forceinline__ F zz() const { return get(2); }
  __device__ __forceinline__ F zzz() const { return get(3); }
  __device__ __forceinline__ void x(const F& v) const { put(0, v); }
  __device__ __forceinline__ void y(const F& v) const { put(1, v); }
  __device__ __forceinline__ void zz(const F& v) const { put(2, v); }
  __device__ __forceinline__ void zzz(const F& v) const { put(3, v); }
};
template <class F> struct TailCfg { static constexpr int L = 256; };          // items per array the fused tail takes over at
template <class G> struct TailCfg<Fp2S<G>> { static constexpr int L = 128; }; // (63 KiB of LDS either way: images of L and 3 L / 4 items)
static inline size_t tail_lds_bytes(uint32_t L, int lpt) { const uint32_t b = 3 * L / 4 ? 3 * L / 4 : 1; return (size_t)36 * (L + b) * lpt * 4; }

// LT: the array length as a compile-time constant (the usual case, TailCfg<F>::L: every LDS access is then base register +
// immediate offset), or 0 for the run-time length of a small window (B < TailCfg::L)
template <class F, int E64, int LT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HalveWaves<F>::MIN))) k_reduce_tail(const uint32_t* in, size_t in_stride, int narr_in, uint32_t Lrt, int c,
                                                     uint64_t* __restrict__ out) {
  KG_REDUCE_PRIO();
  extern __shared__ uint32_t lds[];
  constexpr uint32_t LPT = Lanes<F>::N;
  const uint32_t L = LT ? (uint32_t)LT : Lrt;
  const uint32_t capA = L * LPT, capB = (3 * L / 4 ? 3 * L / 4 : 1) * LPT;       // lane-items per image
  const uint32_t S = capA + capB;                    // the images are interleaved: word k of image A's item q at lds[k * S + q], of image B's at lds[k * S + capA + q]
  uint32_t* const imgA = lds;
  uint32_t* const imgB = lds + capA;
  const uint32_t w = blockIdx.x / (uint32_t)narr_in, a = blockIdx.x % (uint32_t)narr_in;
  const uint32_t task = threadIdx.x / LPT, half = threadIdx.x % LPT;           // LPT consecutive lanes form a task
  const bool spawns = a == 0;
  int steps = 0;
  while ((1u << steps) < L) ++steps;
  const BufRsrc rin = soa_rsrc(in);
  const uint32_t is4 = (uint32_t)in_stride * 4u;
  const uint32_t src0 = (w * (uint32_t)narr_in + a) * L;
  for (int t = 1; t <= steps; ++t) {
    const uint32_t per = L >> t;                       // pairs per live array in this step
    const uint32_t live = spawns ? (uint32_t)t : 1u;
    const bool odd_step = (t & 1) != 0;
    uint32_t* const oimg = odd_step ? imgA : imgB;
    uint32_t* const iimg = odd_step ? imgB : imgA;
    const uint32_t ocap = S, icap = S;
    if (task < live * per) {
      const uint32_t k = task / per, q = task % per;
      const LdsPt<F> sum{oimg, ocap, (k * per + q) * LPT + half};
      const LdsPt<F> spawn{oimg, ocap, ((uint32_t)t * per + q) * LPT + half};    // the odd items of A become array t
      if (t == 1) {
        const SoaSrc<F> p0{rin, is4, (src0 + 2 * q) * 4u}, p1{rin, is4, (src0 + 2 * q + 1) * 4u};
        add_xyzz_stream<F>(p0, p1, sum);
        if (spawns) copy_xyzz_stream<F>(p1, spawn);
      } else {
        const LdsPt<F> p0{iimg, icap, (k * 2 * per + 2 * q) * LPT + half}, p1{iimg, icap, (k * 2 * per + 2 * q + 1) * LPT + half};
        add_xyzz_stream<F>(p0, p1, sum);
        if (spawns && k == 0) copy_xyzz_stream<F>(p1, spawn);
      }
    }
    __syncthreads();                                   // this step's image is complete; the other one is free to be overwritten
  }
  // results: one item per array -- A (or this workgroup's T array) is item 0, the array spawned at step j item j
  const uint32_t nres = spawns ? (uint32_t)steps + 1u : 1u;
  if (task < nres) {
    const bool in_a = (steps & 1) != 0;
    const LdsPt<F> p{in_a ? imgA : imgB, S, task * LPT + half};
    const int arr = task == 0 ? (int)a : narr_in - 1 + (int)task;        // 0 = A, 1 + l = T_l
    uint64_t* dst = out + ((size_t)w * c + arr) * 4 * E64;
    export_el(p.x(), dst);
    export_el(p.y(), dst + E64);
    export_el(p.zz(), dst + 2 * E64);
    export_el(p.zzz(), dst + 3 * E64);
  }
}

// Hot buckets (more than GATHER_SUM_MAX partial sums: the sort lists them).  Stage 1: HOT_SPLIT workgroups per bucket each fold a
// contiguous share of its partial sums -- every task-lane sums a strided part into an LDS slot, then a tree over the slots -- into a
// scratch point; stage 2: one wave per bucket folds the HOT_SPLIT scratch points and leaves the total where the bucket's first partial
// sum was.  A 0/1-heavy witness against window tables puts 8 192 partial sums into one bucket: 4 + 7 + 4 additions deep instead of the
// 64 + 7 of a single workgroup (1.3 ms of G2 additions); the lane-by-lane rounds this replaced (k_task_count / k_scan_rows / k_row_bases /
// k_sum_tasks, sixteen partial sums per lane and round) took two rounds of four launches, 0.5-0.8 ms per MSM.
// The number of shares follows the fullest bucket (hot_split: ~512 partial sums per share, 16 .. one per task-lane of the folding
// workgroup): the 4 x 10^5 partial sums of a 2^24-pair witness' bucket of ones were 100 additions per lane in sixteen shares (0.97 ms).
constexpr uint32_t HOT_SPLIT = 16;                   // the fewest shares
template <class KF> constexpr uint32_t hot_split_max() { return 256u / Lanes<KF>::N; }
template <class KF>
static uint32_t hot_split(uint32_t max_tasks) {
  uint32_t sp = (max_tasks + 511u) / 512u;
  if (sp < HOT_SPLIT) sp = HOT_SPLIT;
  if (sp > hot_split_max<KF>()) sp = hot_split_max<KF>();
  return sp;
}
template <class KF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_hot_sum(const uint32_t* part, Level L, int B, const uint32_t* __restrict__ hot_list,
                                                                                                     uint32_t* scratch, uint32_t HOT_SPLIT) {
  KG_REDUCE_PRIO();
  extern __shared__ uint32_t lds[];                   // 36 words x 256 lane-items
  constexpr uint32_t LPT = Lanes<KF>::N, NT = 256 / LPT, NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = hot_list[blockIdx.x];
  const uint32_t w = t / (uint32_t)B, cnt = L.cnt[t], first = L.base[w] + L.rel[t];
  const uint32_t share = (cnt + HOT_SPLIT - 1) / HOT_SPLIT;
  const uint32_t lo = blockIdx.y * share, hi = lo + share < cnt ? lo + share : cnt;       // this workgroup's partial sums
  const uint32_t task = threadIdx.x / LPT, half = threadIdx.x % LPT;
  const BufRsrc rp = soa_rsrc(part), rs = soa_rsrc(scratch);
  const LdsPt<KF> mine{lds, 256u, threadIdx.x};
  if (lo + task < hi) {
    copy_xyzz_stream<KF>(AosSrc<KF>{rp, (first + lo + task) * NWB, true}, mine);
    for (uint32_t j = lo + task + NT; j < hi; j += NT) {
      KG_STREAM_FENCE();
      add_xyzz_stream<KF>(mine, AosSrc<KF>{rp, (first + j) * NWB, true}, mine);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 36; ++k) lds[(uint32_t)k * 256u + threadIdx.x] = 0u;      // the identity
  }
  __syncthreads();
  for (uint32_t h = NT / 2; h >= 1; h >>= 1) {
    if (task < h) {
      const LdsPt<KF> other{lds, 256u, (task + h) * LPT + half};
      add_xyzz_stream<KF>(mine, other, mine);
    }
    __syncthreads();
  }
  if (task == 0) {
    const AosDst<KF> dst{rs, (blockIdx.x * HOT_SPLIT + blockIdx.y) * NWB};
    copy_xyzz_stream<KF>(mine, dst);
  }
}
template <class KF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_hot_fold(uint32_t* part, Level L, int B, const uint32_t* __restrict__ hot_list,
                                                                                                      const uint32_t* scratch, uint32_t HOT_SPLIT) {
  KG_REDUCE_PRIO();
  extern __shared__ uint32_t lds[];                   // 36 words x 256 lane-items
  constexpr uint32_t LPT = Lanes<KF>::N, NT = 256 / LPT, NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = hot_list[blockIdx.x];
  const uint32_t w = t / (uint32_t)B, first = L.base[w] + L.rel[t];
  const uint32_t task = threadIdx.x / LPT, half = threadIdx.x % LPT;
  const BufRsrc rp = soa_rsrc(part), rs = soa_rsrc(scratch);
  const LdsPt<KF> mine{lds, 256u, threadIdx.x};
  if (task < HOT_SPLIT) copy_xyzz_stream<KF>(AosSrc<KF>{rs, (blockIdx.x * HOT_SPLIT + task) * NWB, true}, mine);
  else {
#pragma unroll
    for (int k = 0; k < 36; ++k) lds[(uint32_t)k * 256u + threadIdx.x] = 0u;
  }
  __syncthreads();
  uint32_t top = NT / 2;
  while (top >= HOT_SPLIT && top > 1) top >>= 1;       // the first level that has a partner with data: skip the levels of identities
  for (uint32_t h = top; h >= 1; h >>= 1) {
    if (task < h) {
      const LdsPt<KF> other{lds, 256u, (task + h) * LPT + half};
      add_xyzz_stream<KF>(mine, other, mine);
    }
    __syncthreads();
  }
  if (task == 0) {
    const AosDst<KF> dst{rp, first * NWB};
    copy_xyzz_stream<KF>(mine, dst);
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// debugging aid (KG_TRACE_HOST=1): host-side timestamps of the pipeline's calls
}  // namespace
namespace kg {
void host_trace(const char* what) {
  if (!tuning().trace_host) return;
  static const auto t0 = std::chrono::steady_clock::now();
  fprintf(stderr, "[host] %-18s %10.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
}

int pick_window(size_t n, int forced) {
  const int wide_from = tuning().wide_window;
  if (forced) return forced;
  int lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) ++lg;      // floor(log2 n)
  // The top window holds the 254 - (W-1)c leftover bits; c = 15 / 16 leave it 14 bits (as many buckets as the
  // signed windows use), while c = 12..14 would leave 2..7 bits, i.e. a handful of buckets holding ~n points each.
  // c = 17 (W = 15, a 16-bit top window) needs the two-pass sort, i.e. n <= 2^24; measured faster from 2^21 up
  // (2^22: 6.6 vs 7.0 ms, 2^24: 25.8 vs 35.2 ms), slower at 2^20 where its bucket reduction doubles
  // c = 20 (13 windows, 2^19 buckets each, 32 entries per bucket at 2^24 -- the density of the 2^20 / c = 16 optimum), UNSLICED and
  // pipelined by window groups (kg_msm): 13 instead of 15 additions per pair and one bucket reduction; needs the nine-bit fine field
  // of the two-pass sort (eight-byte intermediate entries, 20480-entry segments).  Measured at 2^24 (MI355X, alternating runs on one
  // box): 21.35 ms in groups of 3,3,3,4 windows against 22.0 ms for c = 17 in four index slices (four equal groups: 21.8); at 2^23 the
  // slices win (11.3 against 11.75 ms).  The accumulation drops from 19.3 to 16.0 ms of launch time, but each group's sort still runs
  // 2-4x slower beside an accumulation than alone (latency-bound kernels at one or two waves per SIMD) and bounds the pipeline -- before
  // the sort kernels were slimmed to two workgroups per CU beside an accumulation the wide window lost (21.3 against 21.1).
  // KG_WIDE_WINDOW=0 keeps the slices, =23 widens from 2^23.
  if (wide_from > 0 && lg >= wide_from && n <= ((size_t)1 << 24)) return 20;
  if (lg >= 21 && n <= ((size_t)1 << 24)) return 17;
  if (lg >= 19) return 16;
  if (lg >= 14) return 15;
  int c = lg - 3;
  if (c < 2) c = 2;
  if (c > 10) c = 10;
  return c;
}

// Which resident form an array of n bases gets.  The 64-byte point is the default at every size: its ~50 re-spreading
// instructions per addition cost 1-2 % of the accumulation when it runs alone, but in the pipeline -- where the next sort and
// the previous reductions compete for the memory system -- halving the sectors per gather wins (2^20: 1.452 -> 1.421 ms per
// step; 2^22 blocking 6.40 -> 6.19 ms; the PMC traffic of a launch halves).  KG_FMT64_MIN_LOG=30 brings the 72-byte form back
// (experiments, and the cross-format test).
bool resident_fmt64(size_t n) { return n >= ((size_t)1 << tuning().fmt64_min_log); }
bool table_fmt64() { return tuning().table64 != 0; }      // tables never fit the cache: 2^20 1.36 -> 1.30 ms per step, Groth16 2.94 -> 2.87
}  // namespace kg
namespace {

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace

namespace kg {

// Window width of the merged form: one set of 2^(c-1) buckets for all windows, so the bucket reduction and the host finish
// shrink by the window count and a wider window costs nothing extra -- c = 17 (15 windows) from 2^17 scalars.  Offered where
// the two-pass sort runs and (window << ceil(log2 n)) | index fits an entry's 24-bit field.
int merged_window(const kg_ctx* ctx, size_t n) {
  if (n < ((size_t)1 << 16) || n > ((size_t)1 << 20)) return 0;
  int c = n >= ((size_t)1 << 17) ? 17 : 16;
  if (ctx && ctx->msm_window >= 15 && ctx->msm_window <= 18) c = ctx->msm_window;
  int s = 0;
  while (((size_t)1 << s) < n) ++s;
  const int W = (255 + c - 1) / c;
  if (((size_t)W << s) > ((size_t)1 << 24)) return 0;
  return c;
}

// Window groups of a blocking MSM (see msm_grouped): offered where the two-pass sort runs.  KG_MSM_GROUPS = 0 / 1 switches
// them off, = k asks for k equal groups, = "a,b,c" names the groups' window counts from the top window down (experiments).
int msm_group_plan(const kg_ctx* ctx, size_t n, int* gw) {
  if (n < ((size_t)1 << 16) || n > ((size_t)1 << 24)) return 0;
  const int c = pick_window(n, ctx ? ctx->msm_window : 0);
  if (c - 1 < FINE_BITS + 4) return 0;
  const int W = (255 + c - 1) / c;
  // measured (MI355X, blocking kg_msm, two accumulation queues): two groups give 2^17 0.70 -> 0.68 ms, 2^18 0.88 -> 0.82, 2^19 1.195 -> 1.116,
  // 2^20 1.87 -> 1.79, 2^21 3.16 -> 3.01, 2^22 6.14 -> 5.97; three or four groups pay more launches and more sort beside the
  // accumulations than their shorter reduction tail returns (2^20: 1.98 / 2.03 ms)
  int NG = n >= ((size_t)1 << 22) ? 3 : (n >= ((size_t)1 << 17) ? 2 : 0);       // 2^22 (round 4, slimmed sort kernels): 5.97 / 5.79 / 5.83 ms in 2 / 3 / 4 groups
  if (c >= 19) NG = 4;                                 // the unsliced 2^23..2^24-pair commitments: the sort of 13-14 windows is 4 ms, hidden group by group
  if (ctx && ctx->msm_groups) NG = ctx->msm_groups;   // kg_msm_set_groups
  const kg_tuning& tn = ctx ? ctx->tune : tuning();
  if (!tn.msm_groups_list.empty() || tn.msm_groups >= 0) {
    if (!tn.msm_groups_list.empty()) {
      int k = 0, sum = 0;
      const char* p = tn.msm_groups_list.c_str();
      while (*p && k < kg_ctx::MAX_GROUPS) {
        const int v = atoi(p);
        if (v < 1) return 0;
        gw[k++] = v; sum += v;
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
      if (sum == W && !*p) return k;                 // a list that does not fit this window count falls through to the default
    } else NG = tn.msm_groups;
  }
  if (NG > kg_ctx::MAX_GROUPS) NG = kg_ctx::MAX_GROUPS;
  if (NG > W) NG = W;
  if (NG < 2) return 0;
  // equal groups, the remainder to the top ones -- except the wide windows, where the LAST group takes it (3,3,3,4 of 13: the top
  // group's sort is the only one nothing hides; 21.35 against 21.8 ms at 2^24)
  for (int g = 0; g < NG; ++g) gw[g] = W / NG + (c >= 19 ? (g >= NG - W % NG ? 1 : 0) : (g < W % NG ? 1 : 0));
  return NG;
}

int msm_sort_begin(kg_ctx* ctx, int scalar_field, const uint64_t* d_scalars, size_t n, MsmSortPlan* P, bool ordered, int merged_c, int lane_mult,
                   int ngroups, const int* gw, bool on_main) {
  if (n == 0 || n >= ((size_t)1 << 31)) return set_err(ctx, KG_ERR_BAD_ARG, "msm length must be in [1, 2^31)");
  host_trace("sort: enter");
  KG_HIP(ctx, hipSetDevice(ctx->device));
  const bool merged = merged_c != 0;
  int c = merged ? merged_c : pick_window(n, ctx->msm_window);
  if (!merged && c > 16 && !(n >= ((size_t)1 << 16) && n <= ((size_t)1 << 24))) c = 16;   // one-pass histogram: 2^(c-1) LDS counters
  const int W = (255 + c - 1) / c;                  // windows of the scalars
  const int B = 1 << (c - 1);
  int mshift = 0;
  if (merged) {
    while (((size_t)1 << mshift) < n) ++mshift;
    if (c < 15 || c > 18 || n < ((size_t)1 << 16) || ((size_t)W << mshift) > ((size_t)1 << 24))
      return set_err(ctx, KG_ERR_BAD_ARG, "merged sort not offered for this length / window");
  }
  const int Wb = merged ? 1 : W;                    // windows of the BUCKET space: the merged form keeps one set for all digits
  const size_t nv = merged ? (size_t)W * n : n;     // entries that can meet one bucket window
  int nch = (int)((n + 16383) / 16384);
  const int nch_cap = ctx->tune.sort_nch >= 1 && ctx->tune.sort_nch <= 1024 ? ctx->tune.sort_nch : 64;
  if (nch > nch_cap) nch = nch_cap;               // (window, chunk) workgroups of the first sort pass: 1024 of them at 2^20
  if (nch < 1) nch = 1;
  size_t chunk_len = (n + nch - 1) / nch;
  // Task length.  A task is one lane's sequential chain of additions, so the accumulation can never be shorter than T
  // additions' latency (14 us each at four waves per SIMD, ~23 us for G2) however little work there is -- a 0/1-heavy witness
  // (a third of a uniform input's additions) took LONGER than a uniform one with T = 4 n / B + 32: 1.47 against 1.04 ms for the
  // prover's G2 query.  T = 2 n / B + 16 still leaves a uniform input one task per bucket (a bucket holds n / B entries on
  // average, Poisson: 2 n / B + 16 is 8 sigma out at n / B = 16 and 8.5 at 32); the unsigned top window holds twice the load per
  // bucket and gets twice the length (T_top).  Hot buckets pay for the shorter tasks with more partial sums: k_hot_sum.
  uint32_t T = (uint32_t)(2 * (n / B) + 16);
  if (ctx->tune.msm_t >= 4 && ctx->tune.msm_t <= 4096) T = (uint32_t)ctx->tune.msm_t;      // experiments
  if (T < 32) T = 32;
  if (T > 2048) T = 2048;
  if (merged) {
    // a bucket holds ~W n / B entries (60 at 2^18, 240 at 2^20): cut so that the accumulation has about one resident round of
    // lanes (4096 waves); every task beyond the first of a bucket costs one partial-sum addition afterwards
    const size_t lanes = (size_t)(lane_mult < 1 ? 1 : lane_mult);
    size_t t = nv * lanes / ((size_t)4096 * 64);
    T = 16;
    while (T < 128 && 2 * (size_t)T <= t) T *= 2;
    if (ctx->tune.merged_t >= 4 && ctx->tune.merged_t <= 4096) T = (uint32_t)ctx->tune.merged_t;
  }
  // two passes (bucket group, then bucket inside the group) once the sorted lists outgrow the L2; entries carry the
  // bucket's low FINE_BITS between the passes, which leaves 24 bits for the index
  const bool two_pass = merged || (c - 1 >= FINE_BITS + 4 && n >= ((size_t)1 << 16) && n <= ((size_t)1 << 24));
  const bool alone_ok = ctx->tune.sort_alone != 0;   // 0 (experiments): every sort shaped for a busy device
  const bool alone = alone_ok && (ctx->sort_alone || ngroups > 1);      // the (first group's) sort runs on an otherwise idle device
  if (c >= 19 && (!two_pass || merged)) return set_err(ctx, KG_ERR_BAD_ARG, "windows of 19 and 20 bits need the two-pass sort (2^16 .. 2^24 scalars)");
  const int fb = fine_bits_for(c);                  // low bucket bits an entry carries between the passes
  const uint32_t FINE = 1u << fb;
  const int G = two_pass ? B >> fb : 0;             // bucket groups per window (<= 1024)
  const uint32_t seg = seg_len_for(fb);
  const int maxseg = two_pass ? G + (int)((nv + seg - 1) / seg) : 0;
  if (two_pass) chunk_len = (chunk_len + PREP_CH - 1) / PREP_CH * PREP_CH;   // k_prep_scalars_count: one chunk per workgroup
  // window groups: gw[0] windows from the top, then gw[1], ... (the host's double-and-add chain consumes them in that order)
  if (ngroups < 1 || ngroups > kg_ctx::MAX_GROUPS) return set_err(ctx, KG_ERR_BAD_ARG, "bad number of window groups");
  if (ngroups > 1 && (!two_pass || merged || !gw)) return set_err(ctx, KG_ERR_BAD_ARG, "window groups need the two-pass, unmerged sort");
  MsmSortPlan& Q = *P;
  Q = MsmSortPlan();
  {
    int top = Wb, sum = 0;
    for (int g = 0; g < ngroups; ++g) {
      const int wg = ngroups == 1 ? Wb : gw[g];
      if (wg < 1) return set_err(ctx, KG_ERR_BAD_ARG, "empty window group");
      top -= wg; sum += wg;
      Q.gw0[g] = top; Q.gW[g] = wg;
    }
    if (sum != Wb) return set_err(ctx, KG_ERR_BAD_ARG, "window groups do not add up to the window count");
  }
  Carver cv;
  Q.o_kt = cv.take(n * 32); Q.o_cnt = cv.take((size_t)W * nch * (two_pass ? G : B) * 4); Q.o_bsize = cv.take((size_t)W * B * 4);
  Q.o_bstart = cv.take((size_t)Wb * B * 4);
  Q.o_tmp = cv.take(two_pass ? (size_t)W * n * (fb == 9 ? 8 : 4) : 0); Q.o_gsize = cv.take((size_t)W * G * 4); Q.o_gstart = cv.take((size_t)W * G * 4);
  Q.o_segbase = cv.take((size_t)W * (G + 1) * 4); Q.o_segcnt = cv.take((size_t)Wb * maxseg * FINE * 4); Q.o_segoff = cv.take((size_t)Wb * maxseg * FINE * 4);
  Q.o_sorted = cv.take((size_t)W * n * 4); Q.o_lcnt = cv.take((size_t)Wb * B * 4); Q.o_lrel = cv.take((size_t)Wb * B * 4);
  Q.o_rowtot = cv.take((size_t)W * 4);
  Q.o_woff = cv.take(merged ? (size_t)W * G * 4 : 0); Q.o_gsize_m = cv.take(merged ? (size_t)G * 4 : 0); Q.o_gstart_m = cv.take(merged ? (size_t)G * 4 : 0);
  Q.o_segbase_m = cv.take(merged ? (size_t)(G + 1) * 4 : 0);
  Q.o_bpart = cv.take((size_t)W * (B / 4096 + 32) * 2 * 4);      // k_bucket_part: (entries, tasks) of each part of each row
  for (int g = 0; g < ngroups; ++g) {                     // what the task decomposition keeps per group
    Q.part_cap[g] = (size_t)Q.gW[g] * (4 * ((nv + T - 1) / T)) + (size_t)Q.gW[g] * B;     // upper bound on round-1 tasks (hot buckets: tasks of T / 4, bucket_task_len)
    Q.o_lbase[g] = cv.take((size_t)(Q.gW[g] + 1) * 4);
    Q.o_misc[g] = cv.take(64); Q.o_lenh[g] = cv.take(2 * LEN_BINS * 4);             // adjacent: one zero fill covers both
    Q.o_tbkt[g] = cv.take(Q.part_cap[g] * 4); Q.o_tid[g] = cv.take(Q.part_cap[g] * 4);
    Q.o_hot[g] = cv.take((size_t)HOT_MAX * 4);
  }
  // The scalar side runs on a queue of its own and alternates between two spaces: while MSM i accumulates (main queue,
  // reading set i & 1), MSM i+1 is sorted into the other set.  Ordering: the scalar queue waits for `after` (the producer
  // of d_scalars), or -- stream semantics -- for everything enqueued on the main queue so far, unless the context's inputs
  // are declared complete (kg_ctx_set_inputs_complete); and for the last reader of the set it is about to overwrite.
  const int set = (int)(ctx->sort_seq++ & 1u);
  KG_TRY(ensure_ws_sort(ctx, set, cv.off));
  KG_TRY(ensure_pinned(ctx, 4096));
  KG_TRY(make_sort_stream(ctx));
  char* ws = (char*)ctx->ws_sort[set];
  // on_main: the conversion runs on the main queue (a blocking call whose first window group is sorted and accumulated there:
  // no cross-queue hand-over in front of the first accumulation); the scalar queue is put behind it by the caller
  hipStream_t st = on_main ? ctx->stream : ctx->sort_stream;
  if (!on_main && !ordered && !ctx->inputs_complete) {
    KG_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_order, 0));
  }
  for (int j = 0; j < ctx->ws_idle_n[set]; ++j) KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_ws_idle[set][j], 0));
  ctx->ws_idle_n[set] = 0;
  Q.n = n; Q.chunk_len = chunk_len; Q.nv = nv; Q.c = c; Q.W = W; Q.B = B; Q.Wb = Wb; Q.G = G; Q.nch = nch; Q.maxseg = maxseg; Q.mshift = mshift;
  Q.set = set; Q.ngroups = ngroups; Q.merged = merged; Q.two_pass = two_pass; Q.T = T; Q.ws = ws;
  Q.alone = alone;
  Q.T_top = merged ? T : 2 * T;
  Q.fb = fb;
  uint32_t* kt = (uint32_t*)(ws + Q.o_kt);
  uint32_t* cnt = (uint32_t*)(ws + Q.o_cnt);

  Words8 H;                                          // bias H = sum_{w < W-1} 2^(w*c + c - 1)
  for (int j = 0; j < 8; ++j) H.w[j] = 0;
  for (int w = 0; w < W - 1; ++w) {
    int bit = w * c + c - 1;
    H.w[bit >> 5] |= 1u << (bit & 31);
  }
  {
    PhaseScope ph(ctx, "prep_scalars", st);
    if (two_pass) {
      const size_t hl = (size_t)W * G * 4;
      zero_fill(st, cnt, (size_t)W * nch * G * 4);
      // PREP_CH scalars per workgroup is 256 workgroups at 2^20 -- one wave per SIMD, which is all that fits beside an accumulation
      // anyway; the first conversion of a blocking MSM has the chip to itself and takes a quarter of that per workgroup (four times
      // the flushes of the [W][G] counters: only where those are few, i.e. not the wide windows)
      const int per_wg = (alone && n < ((size_t)1 << 22) && fb == FINE_BITS) ? PREP_CH / 4 : PREP_CH;
      const dim3 grid((unsigned)((n + per_wg - 1) / per_wg));
      if (scalar_field == KG_FR) {
        if (hl > 48 * 1024) KG_HIP(ctx, hipFuncSetAttribute((const void*)k_prep_scalars_count<FrParams>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
        hipLaunchKernelGGL(k_prep_scalars_count<FrParams>, grid, dim3(PREP_NT), hl, st, d_scalars, n, H, kt, c, W, fb, G, nch, chunk_len, cnt, per_wg);
      } else {
        if (hl > 48 * 1024) KG_HIP(ctx, hipFuncSetAttribute((const void*)k_prep_scalars_count<FqParams>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
        hipLaunchKernelGGL(k_prep_scalars_count<FqParams>, grid, dim3(PREP_NT), hl, st, d_scalars, n, H, kt, c, W, fb, G, nch, chunk_len, cnt, per_wg);
      }
    } else if (scalar_field == KG_FR) hipLaunchKernelGGL(k_prep_scalars<FrParams>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, H, kt);
    else hipLaunchKernelGGL(k_prep_scalars<FqParams>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, H, kt);
    ph.end();
  }
  KG_HIP(ctx, hipGetLastError());
  return KG_OK;
}

// Sort of one window group: windows [w0, w0 + Wg) of the plan (all of them for the one-pass and the merged sort).  The
// per-window tables are window-major, so a group's view is a pointer offset; only k_group_scatter needs the absolute window
// (the digit's position in the scalar).
int msm_sort_group(kg_ctx* ctx, const MsmSortPlan& Q, int g, MsmSorted* S, bool on_main) {
  if (g < 0 || g >= Q.ngroups) return set_err(ctx, KG_ERR_BAD_ARG, "bad window group");
  KG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = Q.n, nv = Q.nv;
  const int c = Q.c, W = Q.W, B = Q.B, G = Q.G, nch = Q.nch, maxseg = Q.maxseg;
  const bool merged = Q.merged, two_pass = Q.two_pass;
  const int w0 = Q.gw0[g], Wg = Q.gW[g];              // bucket-space windows of the group (merged: the single set)
  const int sw0 = merged ? 0 : w0, sWg = merged ? W : Wg;      // scalar windows the group's first pass covers
  // hot buckets are cut 2^shift times finer (bucket_task_len; the shift travels in the top bits of the task length): 2 by default, 0 for
  // a merged sort.  KG_HOT_SHIFT: experiments.  Measured with shift 2 (MI355X): witness-like 2^20 MSM 1.17 -> 0.67 ms per step, proof from a
  // 0/1-heavy witness 2.48 -> 2.16 ms; with window tables (merged) 1.87 -> 2.03, hence 0 there.
  const int hot_shift_env = ctx->tune.hot_shift;
  const uint32_t hot_shift = hot_shift_env >= 0 && hot_shift_env <= 2 ? (uint32_t)hot_shift_env : (Q.merged ? 0u : 2u);
  const uint32_t T = Q.T | (hot_shift << 30);
  char* ws = Q.ws;
  hipStream_t st = on_main ? ctx->stream : ctx->sort_stream;
  const size_t npts = (size_t)Wg * B;
  uint32_t* kt = (uint32_t*)(ws + Q.o_kt);
  uint32_t* cnt = (uint32_t*)(ws + Q.o_cnt);
  uint32_t* rowtot = (uint32_t*)(ws + Q.o_rowtot) + w0;
  uint32_t* misc = (uint32_t*)(ws + Q.o_misc[g]);
  uint32_t* lenh = (uint32_t*)(ws + Q.o_lenh[g]);
  S->set = Q.set; S->ready = ctx->ev_sorted[Q.set][g];
  S->n = n; S->c = c; S->W = Wg; S->B = B; S->T = T; S->npts = npts; S->part_cap = Q.part_cap[g];
  const int gi = Q.info_base + g;                     // read-back words / event of this sort
  if (gi >= kg_ctx::MAX_GROUPS) return set_err(ctx, KG_ERR_BAD_ARG, "bad read-back index");
  S->merged_shift = Q.mshift; S->windows = W; S->w0 = w0; S->group = gi; S->acc_stream = nullptr; S->sorted_on = st;
  S->sorted = (uint32_t*)(ws + Q.o_sorted) + (size_t)w0 * n; S->bsize = (uint32_t*)(ws + Q.o_bsize) + (size_t)w0 * B;
  S->bstart = (uint32_t*)(ws + Q.o_bstart) + (size_t)w0 * B;
  S->lcnt = (uint32_t*)(ws + Q.o_lcnt) + (size_t)w0 * B; S->lrel = (uint32_t*)(ws + Q.o_lrel) + (size_t)w0 * B; S->lbase = (uint32_t*)(ws + Q.o_lbase[g]);
  S->task_bkt = (uint32_t*)(ws + Q.o_tbkt[g]); S->task_id = (uint32_t*)(ws + Q.o_tid[g]);
  S->hot_list = (uint32_t*)(ws + Q.o_hot[g]);
  S->T_top = Q.T_top | (hot_shift << 30);
  S->top_w = (!merged && w0 + Wg == W) ? Wg - 1 : -1;       // the unsigned top window, if this group holds it
  const uint32_t T_top = S->T_top;
  const int top_w = S->top_w;
  {
    PhaseScope ph(ctx, "sort", st);
    const size_t lds = (size_t)(two_pass ? G : B) * 4;
    if (lds > 48 * 1024) {      // the whole-window histogram needs more than the default dynamic LDS limit
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int fb = Q.fb;
    const uint32_t FINE = 1u << fb;
    uint32_t* tmp = (uint32_t*)(ws + Q.o_tmp);                // uint32_t entries (FB = 7) or uint64_t (FB = 9)
    uint64_t* tmp8 = (uint64_t*)(ws + Q.o_tmp);
    uint32_t* gsize = (uint32_t*)(ws + Q.o_gsize);
    uint32_t* gstart = (uint32_t*)(ws + Q.o_gstart);
    uint32_t* segbase = (uint32_t*)(ws + Q.o_segbase);
    uint32_t* segcnt = (uint32_t*)(ws + Q.o_segcnt) + (size_t)w0 * maxseg * FINE;
    uint32_t* segoff = (uint32_t*)(ws + Q.o_segoff) + (size_t)w0 * maxseg * FINE;
    const size_t zbytes = (Q.o_lenh[g] - Q.o_misc[g]) + 2 * LEN_BINS * 4;     // misc and the length histogram: cleared by k_group_scan, or
    if (!two_pass) zero_fill(st, misc, zbytes);
    uint32_t* woff = merged ? (uint32_t*)(ws + Q.o_woff) : nullptr;
    // the tables the fine pass and the task decomposition read: per window (the group's rows), or the merged single set
    const uint32_t* f_gstart = merged ? (uint32_t*)(ws + Q.o_gstart_m) : gstart + (size_t)w0 * G;
    const uint32_t* f_gsize = merged ? (uint32_t*)(ws + Q.o_gsize_m) : gsize + (size_t)w0 * G;
    const uint32_t* f_segbase = merged ? (uint32_t*)(ws + Q.o_segbase_m) : segbase + (size_t)w0 * (G + 1);
    const uint32_t* f_tmp = merged ? tmp : tmp + (size_t)w0 * n;
    const uint64_t* f_tmp8 = tmp8 + (size_t)w0 * n;
    // first pass on big tiles (k_group_scatter_big): 8192 entries with the wide windows' 1024 groups, 4096 otherwise; 1024 threads
    // where nothing else runs (the first window group of a blocking MSM), 256 beside an accumulation.  Measured (MI355X, round 4):
    // 2^24-pair commitment 20.32 -> 19.00 ms, blocking 2^20 1.746 -> 1.685 ms, the four-deep 2^20 step 1.32 -> 1.295 ms; 512 threads
    // beside the accumulation shorten the sorts (12.9 -> 9.8 ms summed at 2^24) and lengthen the accumulations by as much.
    const int gs_tile_env = ctx->tune.gs_tile;     // experiments: 0 = the 1024-entry tiles of k_group_scatter
    const int gs_nt_set = ctx->tune.gs_nt;
    // wide windows: 512 -- level on uniform scalars (2^24: 19.05 / 19.0 ms), and a witness-like 2^24-pair vector, whose accumulations are
    // short and whose sorts therefore run mostly alone, 10.65 -> 8.9 ms; below, 512 costs the four-deep 2^20 step 1 %
    const int gs_nt_env = gs_nt_set ? gs_nt_set : (fb == 9 ? 512 : 256);
    const int gs_nt0_env = ctx->tune.gs_nt0;
    const int gs_tile = gs_tile_env >= 0 ? gs_tile_env : (fb == 9 ? 8192 : 4096);
    const int gs_nt = (Q.alone && g == 0) ? gs_nt0_env : gs_nt_env;
    if (two_pass) {
      hipLaunchKernelGGL(k_group_scan, dim3(sWg), dim3(G > 256 ? 512 : GS_NT), 0, st, cnt + (size_t)sw0 * nch * G, nch, G, B, gsize + (size_t)sw0 * G, gstart + (size_t)sw0 * G,
                         segbase + (size_t)sw0 * (G + 1), (uint32_t*)(ws + Q.o_bsize) + (size_t)sw0 * B, misc, (int)(zbytes / 4), seg_len_for(fb));
      if (merged)
        hipLaunchKernelGGL(k_merge_groups, dim3(1), dim3(GS_NT), 0, st, gsize, W, G, woff, (uint32_t*)(ws + Q.o_gsize_m), (uint32_t*)(ws + Q.o_gstart_m),
                           (uint32_t*)(ws + Q.o_segbase_m));
      if (fb == 9) {
        if (gs_tile) KG_HIP(ctx, launch_gs_big_any<9>(gs_tile, gs_nt, dim3(sWg, nch), st, kt, n, c, W, Q.chunk_len, G, cnt, gstart, tmp8, woff, Q.mshift, sw0));
        else
          hipLaunchKernelGGL(k_group_scatter<9>, dim3(sWg, nch), dim3(GS_NT), 0, st, kt, n, c, W, Q.chunk_len, G, cnt, gstart, tmp8, woff, Q.mshift, sw0);
        KG_HIP(ctx, hipFuncSetAttribute((const void*)k_fine_local<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_local_lds<9>()));
        hipLaunchKernelGGL(k_fine_local<9>, dim3(Wg, maxseg), dim3(512), fine_local_lds<9>(), st, f_tmp8, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bsize, segcnt, segoff, S->sorted);
      } else {
        if (gs_tile) KG_HIP(ctx, launch_gs_big_any<7>(gs_tile, gs_nt, dim3(sWg, nch), st, kt, n, c, W, Q.chunk_len, G, cnt, merged ? f_gstart : gstart, tmp, woff, Q.mshift, sw0));
        else
          hipLaunchKernelGGL(k_group_scatter<7>, dim3(sWg, nch), dim3(GS_NT), 0, st, kt, n, c, W, Q.chunk_len, G, cnt, merged ? f_gstart : gstart, tmp, woff, Q.mshift, sw0);
        hipLaunchKernelGGL(k_fine_local<7>, dim3(Wg, maxseg), dim3(512), fine_local_lds<7>(), st, f_tmp, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bsize, segcnt, segoff, S->sorted);
      }
    } else {
      hipLaunchKernelGGL(k_count, dim3(W, nch), dim3(1024), lds, st, kt, n, c, W, Q.chunk_len, 0, cnt);
      hipLaunchKernelGGL(k_scan_chunks, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, cnt, W, nch, B, S->bsize);
    }
    // task decomposition (needs only the bucket sizes); its two result words travel to the host while the
    // scatter below still runs, so the read-back does not stall the queue
    const unsigned g1024 = (unsigned)((npts + 1023) / 1024);
    if (B >= 8192) {
      const int nsplit = B / 4096;                    // <= 16 parts per row
      uint32_t* bpart = (uint32_t*)(ws + Q.o_bpart) + (size_t)w0 * nsplit * 2;
      hipLaunchKernelGGL(k_bucket_part, dim3(Wg, nsplit), dim3(BR_NT), 0, st, S->bsize, B, T, nsplit, bpart, misc, lenh, T_top, top_w, S->hot_list, HOT_MAX);
      hipLaunchKernelGGL(k_bucket_fill, dim3(Wg, nsplit), dim3(BR_NT), 0, st, S->bsize, B, T, nsplit, bpart, S->bstart, S->lcnt, S->lrel, rowtot, T_top, top_w);
    } else
      hipLaunchKernelGGL(k_bucket_rows, dim3(Wg), dim3(BR_NT), 0, st, S->bsize, B, T, S->bstart, S->lcnt, S->lrel, rowtot, misc, lenh, T_top, top_w, S->hot_list, HOT_MAX);
    hipLaunchKernelGGL(k_task_bases, dim3(1), dim3(64), 0, st, rowtot, Wg, S->lbase, misc, misc + 4, lenh, lenh + LEN_BINS, (uint32_t*)ctx->h_pinned_dev + 4 * gi);
    KG_HIP(ctx, hipEventRecord(ctx->ev_info[gi], st));
    hipLaunchKernelGGL(k_len_scatter, dim3(g1024), dim3(1024), 0, st, S->bsize, S->lcnt, S->lrel, S->lbase, npts, B, T, lenh + LEN_BINS, S->task_bkt, S->task_id, T_top, top_w);
    // rows of segment walkers per window; the top window (top_w >= 0: this group holds it, the sort is not merged) gets three more sets
    const int fs_rows = maxseg < FS_ROWS ? maxseg : FS_ROWS;
    const int fs_extra = top_w >= 0 ? 3 : 0;
    if (two_pass && fb == 9) {
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_fine_scatter<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_scatter_lds<9>()));
      hipLaunchKernelGGL(k_fine_scatter<9>, dim3(Wg + fs_extra, fs_rows), dim3(512), fine_scatter_lds<9>(), st, f_tmp8, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bstart, segcnt, segoff, S->sorted, Wg, fs_extra ? top_w : -1);
    } else if (two_pass) {
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_fine_scatter<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_scatter_lds<7>()));
      hipLaunchKernelGGL(k_fine_scatter<7>, dim3(Wg + fs_extra, fs_rows), dim3(512), fine_scatter_lds<7>(), st, f_tmp, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bstart, segcnt, segoff, S->sorted, Wg, fs_extra ? top_w : -1);
    } else
      hipLaunchKernelGGL(k_scatter, dim3(W, nch), dim3(1024), lds, st, kt, n, c, W, Q.chunk_len, 0, cnt, S->bstart, S->sorted);
    ph.end();
    KG_HIP(ctx, hipGetLastError());
    KG_HIP(ctx, hipEventRecord(S->ready, st));
    host_trace("sort: enqueued");
  }
  (void)nv;
  return KG_OK;
}

int msm_sort(kg_ctx* ctx, int scalar_field, const uint64_t* d_scalars, size_t n, MsmSorted* S, bool ordered, int merged_c, int lane_mult, bool wait_info, int info_idx) {
  MsmSortPlan Q;
  KG_TRY(msm_sort_begin(ctx, scalar_field, d_scalars, n, &Q, ordered, merged_c, lane_mult));
  Q.info_base = info_idx;
  KG_TRY(msm_sort_group(ctx, Q, 0, S));
  return wait_info ? msm_sort_wait(ctx, S) : KG_OK;
}

// The task count and the largest bucket of a sort (two words read back through pinned memory, one pair per window group):
// the caller may put other work on the queues between msm_sort(..., wait_info = false) and this, but no other sort with the
// same group index.
int msm_sort_wait(kg_ctx* ctx, MsmSorted* S) {
  KG_HIP(ctx, hipEventSynchronize(ctx->ev_info[S->group]));
  host_trace("sort: info back");
  const uint32_t* h_info = (const uint32_t*)ctx->h_pinned + 4 * S->group;
  S->ntasks = h_info[0];
  S->max_cnt = h_info[3];                              // most tasks any bucket has
  S->nhot = h_info[2];                                 // buckets with more than GATHER_SUM_MAX tasks (listed up to HOT_MAX)
  if (S->ntasks > S->part_cap) return set_err(ctx, KG_ERR_HIP, "task count exceeds its bound");
  return KG_OK;
}

// Base-side half: accumulate + reduce against up to MAX_FUSED base arrays that share the scalar sort S, export, and start
// the copy of each result into its host slot.  One accumulation launch serves all arrays (see k_acc_tasks); everything
// after it runs per array, its reduction on one of the two side queues.
struct RunJob { const uint64_t* d_bases; const uint8_t* d_inf; size_t nbases; uint32_t idx_off; int slot; bool bases_complete; const uint32_t* packed; bool packed64; };

template <class Cfg>
int msm_run_multi_t(kg_ctx* ctx, const MsmSorted& S, const RunJob* jobs, int njobs) {
  using F = typename Cfg::F;
  using KF = typename Cfg::KF;                      // field type of the reduction kernels (Fq2: a lane pair per task, fp2s.h)
  constexpr unsigned LPT = Lanes<KF>::N;            // lanes per task
  constexpr int PW = 2 * BaseIO<F>::PE;             // resident words per base
  constexpr int NW = PointIO<F>::NW;                // raw words per XYZZ point
  if (njobs < 1 || njobs > MAX_FUSED) return set_err(ctx, KG_ERR_BAD_ARG, "bad number of fused base arrays");
  const int W = S.W, B = S.B, c = S.c;
  const size_t npts = S.npts, part_cap = S.part_cap, nexp = (size_t)W * c;
  const size_t exp_bytes = nexp * 4 * Cfg::E64 * 8;
  hipStream_t st = S.acc_stream ? S.acc_stream : ctx->stream, sq;      // a window group may accumulate on a queue of its own
  KG_TRY(scalar_queue(ctx, &sq));
  bool ordered_bases = false, converted = false, shared_pb = false;
  if (!ctx->side_stream) KG_TRY(make_side_stream(ctx));
  if (st != ctx->stream && !ctx->inputs_complete) {      // stream semantics: the group's queue follows what the main queue holds so far
    KG_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_order, 0));
  }
  struct Lay { size_t o_pb, o_lc[2], o_lr[2], o_lb[2], o_part[2], o_pbuf[2], o_rowtot, o_misc, o_hot; char* ws; const uint32_t* pb; int set; };
  Lay lay[MAX_FUSED];
  AccSets A;
  A.nsets = njobs;
  for (int k = 0; k < MAX_FUSED; ++k) { A.pb[k] = nullptr; A.idx_off[k] = 0; A.partial[k] = nullptr; A.tab_n[k] = 0; A.fmt64[k] = 0; }
  for (int k = 0; k < njobs; ++k) {
    const RunJob& J = jobs[k];
    Lay& Y = lay[k];
    Carver cv;
    const uint32_t* reg_pb = nullptr;               // bases inside a registered array are already in packed internal form
    if (J.packed) {                                 // or the caller converted them (kg_msm: once for all window groups)
      if (S.merged_shift) return set_err(ctx, KG_ERR_BAD_ARG, "merged sort against caller-converted bases");
      reg_pb = J.packed; A.fmt64[k] = J.packed64 ? 1 : 0; shared_pb = true;
    }
    for (const auto& r : ctx->registered) {
      if (reg_pb) break;
      if (r.curve != Cfg::ID || J.d_bases < r.base) continue;
      const size_t off64 = (size_t)(J.d_bases - r.base);
      if (off64 % (size_t)BaseIO<F>::W != 0 || off64 / BaseIO<F>::W + J.nbases > r.n) continue;
      // the identity flags were baked in at registration: the resident copy serves the call only when the call's flag
      // array is the registered one (same offset), or both are absent; any other combination converts per call
      const size_t off = off64 / BaseIO<F>::W;
      if (J.d_inf != (r.inf ? r.inf + off : nullptr)) continue;
      if (S.merged_shift) {                         // merged sort: the whole array through its window table
        if (off != 0 || J.nbases != r.n || !r.table || r.table_c != c || r.table_W != S.windows) continue;
        reg_pb = r.table;
        A.tab_n[k] = (uint32_t)r.n;
        A.fmt64[k] = r.table64 ? 1 : 0;
      } else {
        reg_pb = r.packed + off * (r.fmt64 ? 2 * BaseIO<F>::PK : PW);
        A.fmt64[k] = r.fmt64 ? 1 : 0;
      }
      break;
    }
    if (S.merged_shift && !reg_pb) return set_err(ctx, KG_ERR_BAD_ARG, "merged sort against bases without a matching window table");
    const bool conv64 = !reg_pb && resident_fmt64(J.nbases);          // per-call conversion: the same rule as registration
    if (conv64) A.fmt64[k] = 1;
    Y.o_pb = cv.take(reg_pb ? 256 : J.nbases * (conv64 ? 2 * BaseIO<F>::PK : PW) * 4);
    for (int i = 0; i < 2; ++i) {
      Y.o_lc[i] = cv.take(npts * 4); Y.o_lr[i] = cv.take(npts * 4); Y.o_lb[i] = cv.take((size_t)(W + 1) * 4);
      Y.o_part[i] = cv.take(part_cap * NW * 4); Y.o_pbuf[i] = cv.take(npts * NW * 4);
    }
    Y.o_rowtot = cv.take((size_t)W * 4); Y.o_misc = cv.take(64);
    Y.o_hot = cv.take(S.nhot ? (size_t)(S.nhot < HOT_MAX ? S.nhot : HOT_MAX) * hot_split<KF>(S.max_cnt) * NW * 4 : 0);      // k_hot_sum's shares
    Y.set = J.slot % kg_ctx::RUN_SETS;              // run space per set: the reductions of the previous MSMs may still read the other sets
    for (int k2 = 0; k2 < k; ++k2)
      if (lay[k2].set == Y.set) return set_err(ctx, KG_ERR_BAD_ARG, "fused MSMs need result slots in different run-space sets");
    KG_TRY(ensure_ws_run(ctx, Y.set, cv.off));
    KG_TRY(ensure_slot(ctx, J.slot, exp_bytes));
    if (!ctx->ev_acc[Y.set]) KG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_acc[Y.set], hipEventDisableTiming));
    Y.ws = (char*)ctx->ws_run[Y.set];
    Y.pb = reg_pb ? reg_pb : (uint32_t*)(Y.ws + Y.o_pb);
    // this buffer set was last used by an earlier slot: its side-stream work must be over before we overwrite it
    for (int s2 = 0; s2 < kg_ctx::NSLOTS; ++s2)
      if (s2 % kg_ctx::RUN_SETS == Y.set && ctx->slots[s2].done && ctx->slots[s2].busy) {
        KG_HIP(ctx, hipStreamWaitEvent(st, ctx->slots[s2].done, 0));
        if (!reg_pb) KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->slots[s2].done, 0));
        ctx->slots[s2].busy = false;
      }
    if (!reg_pb) {
      // per-call conversion of the bases, on the scalar queue: it runs beside the previous MSM's accumulation instead
      // of between two accumulations on the main queue
      if (!ctx->inputs_complete && !ordered_bases && !J.bases_complete) {   // stream semantics: the bases may still be in flight on the main queue
        KG_HIP(ctx, hipEventRecord(ctx->ev_order, st));
        KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->ev_order, 0));
        ordered_bases = true;
      }
      PhaseScope ph(ctx, "prep_bases", sq);
      launch_prep_bases<F>(sq, J.d_bases, J.d_inf, J.nbases, (uint32_t*)(Y.ws + Y.o_pb), conv64);
      ph.end();
      converted = true;
    }
    A.pb[k] = Y.pb; A.idx_off[k] = J.idx_off; A.partial[k] = (uint32_t*)(Y.ws + Y.o_part[0]);
  }
  const Level L0{S.lcnt, S.lrel, S.lbase};
  if (converted) {                                                       // later on the scalar queue than the sort: covers both
    KG_HIP(ctx, hipEventRecord(ctx->ev_bases, sq));
    KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_bases, 0));
  } else if (S.ready && S.sorted_on != st) KG_HIP(ctx, hipStreamWaitEvent(st, S.ready, 0));   // the scalar queue's sort of this set
  if (shared_pb) KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_pb, 0));     // the caller's conversion of the bases
  if (S.ntasks) {
    PhaseScope ph(ctx, "accumulate", st);
    const bool pair_acc = ctx->tune.g2_pair_acc != 0;
    if (LPT > 1 && pair_acc)          // experiment: G2 accumulation on lane pairs, ~150 VGPRs instead of 256
      hipLaunchKernelGGL(k_acc_tasks<KF>, dim3((unsigned)(((size_t)S.ntasks * LPT + 63) / 64) * (unsigned)njobs), dim3(64), 0, st, A, S.sorted, S.bstart, S.bsize, L0,
                         S.task_bkt, S.task_id, S.n, W, B, S.T, part_cap, S.merged_shift, S.T_top, S.top_w);
    else if (acc_prefetch<F>(A, njobs, jobs[0].nbases))
      hipLaunchKernelGGL(k_acc_tasks_q<typename PfField<F>::T>, dim3(((S.ntasks + 63) / 64) * (unsigned)njobs), dim3(64), 0, st, A, S.sorted, S.bstart, S.bsize, L0, S.task_bkt,
                         S.task_id, S.n, W, B, S.T, part_cap, S.merged_shift, S.T_top, S.top_w);
    else
    hipLaunchKernelGGL(k_acc_tasks<F>, dim3(((S.ntasks + 63) / 64) * (unsigned)njobs), dim3(64), 0, st, A, S.sorted, S.bstart, S.bsize, L0, S.task_bkt, S.task_id,
                       S.n, W, B, S.T, part_cap, S.merged_shift, S.T_top, S.top_w);
    ph.end();
  }
  for (int k = 0; k < njobs; ++k) {
    const RunJob& J = jobs[k];
    const Lay& Y = lay[k];
    const int slot = J.slot, set = Y.set;
    char* ws = Y.ws;
    const size_t* o_lc = Y.o_lc; const size_t* o_lr = Y.o_lr; const size_t* o_lb = Y.o_lb;
    uint32_t* part[2] = {(uint32_t*)(ws + Y.o_part[0]), (uint32_t*)(ws + Y.o_part[1])};
    uint32_t* pbuf[2] = {(uint32_t*)(ws + Y.o_pbuf[0]), (uint32_t*)(ws + Y.o_pbuf[1])};
    uint32_t* rowtot = (uint32_t*)(ws + Y.o_rowtot);
    uint32_t* misc = (uint32_t*)(ws + Y.o_misc);
    kg_ctx::Slot& sl = ctx->slots[slot];
    // two reduction queues, by slot parity: a long reduction (G2: ~4x a G1 one) does not hold up the next MSM's
    hipStream_t side = S.reduce_inline ? st : ((slot & 1) ? ctx->side2_stream : ctx->side_stream);     // reduce_inline: behind the accumulation on its own queue (the last window group: no cross-queue hand-over on the critical path)     // reduce_inline: behind the accumulation on its own queue (the last window group: no cross-queue hand-over on the critical path)
    // Everything after the accumulation runs on a reduction queue, so that the main queue goes from one accumulation straight
    // to the next: the partial-sum rounds, the dense bucket array (gather) and the c-1 latency-bound halving levels.
    if (side != st) {
      KG_HIP(ctx, hipEventRecord(ctx->ev_acc[set], st));
      KG_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_acc[set], 0));
    }
    Level L = L0;
    int pcur = 0;
    bool fused_first = false;
    {
      // buckets cut into several tasks (skewed inputs; every bucket of a merged sort): re-sum a bucket's partial sums until
      // it owns one point
      PhaseScope ph(ctx, "partial_sums", side);
      uint32_t max_cnt = S.max_cnt;
      int lv = -1;                                     // -1: level arrays of S; 0/1: local ping-pong
      const unsigned g1024 = (unsigned)((npts + 1023) / 1024);
      const bool hot_ok = ctx->tune.hot_sum != 0;
      if (max_cnt > GATHER_SUM_MAX && hot_ok && S.nhot >= 1 && S.nhot <= HOT_MAX && (size_t)part_cap * NW * 4 < ((size_t)1 << 32)) {
        // the few buckets with more partial sums than the gather takes: one workgroup-wide tree each
        PhaseScope ph2(ctx, "hot_sum", side);
        uint32_t* hot_scratch = (uint32_t*)(ws + Y.o_hot);
        const uint32_t split = hot_split<KF>(S.max_cnt);
        hipLaunchKernelGGL(k_hot_sum<KF>, dim3(S.nhot, split), dim3(256), 36 * 256 * 4, side, part[pcur], L, B, S.hot_list, hot_scratch, split);
        hipLaunchKernelGGL(k_hot_fold<KF>, dim3(S.nhot), dim3(256), 36 * 256 * 4, side, part[pcur], L, B, S.hot_list, hot_scratch, split);
        ph2.end();
        max_cnt = GATHER_SUM_MAX;
      }
      while (max_cnt > GATHER_SUM_MAX) {               // (at most GATHER_SUM_MAX partial sums per bucket are left to the gather below)
        const int nx = lv < 0 ? 0 : (lv ^ 1);
        PhaseScope pr(ctx, "partial_round", side);         // one per extra round: its count is what a skewed input costs (bench.py msm_skewed)
        uint32_t* ncnt = (uint32_t*)(ws + o_lc[nx]);
        uint32_t* nrel = (uint32_t*)(ws + o_lr[nx]);
        uint32_t* nbase = (uint32_t*)(ws + o_lb[nx]);
        KG_HIP(ctx, hipMemsetAsync(misc, 0, 64, side));
        hipLaunchKernelGGL(k_task_count, dim3(g1024), dim3(1024), 0, side, L.cnt, npts, S.T2, ncnt, misc + 8);
        hipLaunchKernelGGL(k_scan_rows, dim3(W), dim3(1024), 0, side, ncnt, B, nrel, rowtot);
        hipLaunchKernelGGL(k_row_bases, dim3(1), dim3(64), 0, side, rowtot, W, nbase, (const uint32_t*)nullptr, misc + 4);
        Level Lout{ncnt, nrel, nbase};
        // the task count of this round is bounded by the previous one; threads beyond base[W] exit
        const uint32_t bound = lv < 0 ? S.ntasks : (uint32_t)part_cap;
        hipLaunchKernelGGL(k_sum_tasks<KF>, dim3((unsigned)(((size_t)bound * LPT + 63) / 64)), dim3(64), 0, side, part[pcur], L, Lout, W, B, S.T2, part[pcur ^ 1]);
        pr.end();
        pcur ^= 1;
        L = Lout;
        lv = nx;
        max_cnt = (max_cnt + S.T2 - 1) / S.T2;
      }
      ph.end();
      PhaseScope pg(ctx, "gather", side);
      const bool fuse_ok = ctx->tune.gather_fuse != 0;
      fused_first = fuse_ok && max_cnt <= 1 && (uint32_t)B > (uint32_t)TailCfg<KF>::L && (size_t)part_cap * NW * 4 < ((size_t)1 << 32);
      if (max_cnt > 1)
        hipLaunchKernelGGL((k_gather_sum<F, KF>), dim3((unsigned)((npts * LPT + 63) / 64)), dim3(64), 0, side, part[pcur], L, W, B, pbuf[0]);
      else if (fused_first)        // the dense bucket array is never written: level 1 straight from the partial sums
        hipLaunchKernelGGL(k_gather_halve<KF>, dim3((unsigned)((npts / 2 * LPT + 63) / 64)), dim3(64), 0, side, part[pcur], L, W, B, pbuf[0], (size_t)W * 2 * (B / 2));
      else
        hipLaunchKernelGGL(k_gather_buckets<F>, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, side, part[pcur], part_cap, L, W, B, pbuf[0]);
      pg.end();
    }
    if (ctx->ws_idle_n[S.set] < kg_ctx::IDLE_EVS) {      // the gather is the last reader of the scalar-side set (level tables of S)
      KG_HIP(ctx, hipEventRecord(ctx->ev_ws_idle[S.set][ctx->ws_idle_n[S.set]], side));
      ctx->ws_idle_n[S.set] += 1;
    } else KG_HIP(ctx, hipStreamSynchronize(side));       // more readers than events: wait here instead (never in practice)
    int cur = 0;
    {
      PhaseScope ph(ctx, "reduce", side);
      size_t in_stride = npts;                     // level 0 reads the bucket array: stride = W*B items
      int narr = 1;
      uint32_t len = (uint32_t)B;                  // items per array
      if (fused_first) { narr = 2; len = (uint32_t)B / 2; in_stride = (size_t)W * 2 * len; }     // k_gather_halve wrote level 1
      constexpr uint32_t TL = (uint32_t)TailCfg<KF>::L;
      for (; len > TL; len /= 2) {                 // the wide levels: one launch each
        const uint32_t n_out = len / 2;
        const size_t tasks = (size_t)W * narr * n_out;
        const size_t out_stride = (size_t)W * (narr + 1) * n_out;
        hipLaunchKernelGGL(k_halve<KF>, dim3((unsigned)((tasks * LPT + 63) / 64)), dim3(64), 0, side, pbuf[cur], in_stride, pbuf[cur ^ 1], out_stride,
                           W, narr, n_out);
        cur ^= 1;
        in_stride = out_stride;
        ++narr;
      }
      // the remaining log2(len) levels and the export in one launch (W * narr workgroups)
      const size_t tail_lds = tail_lds_bytes(len, (int)LPT);
      const unsigned tail_threads = (unsigned)(len / 2 * LPT) < 64u ? 64u : (unsigned)(len / 2 * LPT);
      // the sums go straight into the slot's pinned host buffer (its device view): no copy kernel behind the tail
      if (len == TL) {
        KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_reduce_tail<KF, Cfg::E64, (int)TL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
        hipLaunchKernelGGL((k_reduce_tail<KF, Cfg::E64, (int)TL>), dim3((unsigned)(W * narr)), dim3(tail_threads), tail_lds, side, pbuf[cur], in_stride, narr, len, c, (uint64_t*)sl.host_dev);
      } else {
        KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_reduce_tail<KF, Cfg::E64, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
        hipLaunchKernelGGL((k_reduce_tail<KF, Cfg::E64, 0>), dim3((unsigned)(W * narr)), dim3(tail_threads), tail_lds, side, pbuf[cur], in_stride, narr, len, c, (uint64_t*)sl.host_dev);
      }
      ph.end();
    }
    KG_HIP(ctx, hipGetLastError());
    KG_HIP(ctx, hipEventRecord(sl.done, side));
    sl.W = W; sl.c = c; sl.w0 = S.w0; sl.busy = true;
  }
  host_trace("run: enqueued");
  return KG_OK;
}

bool has_window_table(const kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t nbases, size_t msm_len) {
  const int c = merged_window(ctx, msm_len);
  if (!c) return false;
  for (const auto& r : ctx->registered)
    if (r.base == d_bases && r.curve == curve && r.n == nbases && r.inf == d_inf && r.table && r.table_c == c) return true;
  return false;
}

int scalar_queue(kg_ctx* ctx, hipStream_t* out) {
  KG_TRY(make_sort_stream(ctx));
  *out = ctx->sort_stream;
  return KG_OK;
}

int msm_run_multi(kg_ctx* ctx, const MsmSorted& S, int curve, const MsmRunJob* jobs, int njobs) {
  RunJob rj[MAX_FUSED];
  if (njobs < 1 || njobs > MAX_FUSED) return KG_ERR_BAD_ARG;
  for (int k = 0; k < njobs; ++k)
    rj[k] = RunJob{jobs[k].d_bases, jobs[k].d_inf, jobs[k].nbases, jobs[k].idx_off, jobs[k].slot, jobs[k].bases_complete, jobs[k].packed, jobs[k].packed64};
  switch (curve) {
    case KG_G1: return msm_run_multi_t<G1Cfg>(ctx, S, rj, njobs);
    case KG_GRUMPKIN: return msm_run_multi_t<GkCfg>(ctx, S, rj, njobs);
    case KG_G2: return msm_run_multi_t<G2Cfg>(ctx, S, rj, njobs);
    default: return KG_ERR_BAD_ARG;
  }
}
int msm_run(kg_ctx* ctx, const MsmSorted& S, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t nbases, uint32_t idx_off, int slot) {
  const MsmRunJob j{d_bases, d_inf, nbases, idx_off, slot, false, nullptr, false};
  return msm_run_multi(ctx, S, curve, &j, 1);
}

}  // namespace kg


