// msm_bases.h -- resident form of the bases (72-byte limb points or 64-byte packed points, identity flag in a spare bit), their
// conversion from the ABI form (k_prep_bases) and the window tables of registered arrays (k_table_next).
#pragma once
#include "msm_common.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

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
}  // namespace msm
template <class F> struct RawIO<Fp2S<F>> {
  static constexpr int NW = 18;
  static __device__ __forceinline__ Fp2S<F> load(const uint32_t* base, size_t stride, size_t i) {
    return {RawIO<F>::load(base + (size_t)(9 * Fp2S<F>::half()) * stride, stride, i)};
  }
  static __device__ __forceinline__ void store(uint32_t* base, size_t stride, size_t i, const Fp2S<F>& a) {
    RawIO<F>::store(base + (size_t)(9 * Fp2S<F>::half()) * stride, stride, i, a.v);
  }
};
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

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
}  // namespace msm
}  // namespace kg
