// msm_small_kernels.h -- the MSM of SHORT inputs (n <= 2^15 pairs; G2 20480) as ONE kernel shape: one launch up to 1536 pairs, a second one that
// adds the bucket ranges of split windows beyond, a third in front that converts the scalars once from 2049 pairs.  The reference's own tests and
// its bench live at these lengths (groth16/src/msm.rs:118-135: 32 pairs; bn254/benches: 2^10; groth16/src/lib.rs:29-77: a handful of constraints).
//
// The long-input pipeline (msm_sort.hip / msm_run.hip) is a chain of 15-20 dependent launches with one read-back in the middle: 0.24 ms
// for 16 pairs, 0.44 ms for 2^10 .. 2^13.  Short inputs are latency, so here a workgroup owns a (window, bucket range) and does everything
// for it out of LDS:
//
//   digits      every lane converts scalars to integers (one Montgomery product) and cuts THIS window's signed digit (msm_digits.h);
//               entries of the workgroup's bucket range are counted in an LDS histogram              [replaces get_at, msm.rs:75-91]
//               glv: every scalar first becomes two 127-bit halves k1 + k2 lambda (glv_decompose_with) -- two list entries against P and
//               (beta x, y), half the windows; KT form: the (half-)scalars come as word planes from k_small_prep, two loads per digit
//   sort        exclusive scan, scatter of (index | sign) into an LDS list ordered by bucket
//   tasks       the bucket lists are cut into <= 256 tasks of at most T entries (T from the entry count: a skewed input -- every scalar
//               equal -- still fills the lanes)
//   accumulate  one lane per task: XYZZ += +-P (madd-2008-s), bases read in the ABI form and converted on the fly, the next base on its way
//               while the current addition runs                                                       [bucket fill, msm.rs:25-35]
//   merge       partial sums of one bucket are added by a tree in LDS (depth log2 of the most tasks a bucket has)
//   merge, halve, combine run on QUADS of lanes (coop_add.h: the 14 products of an addition in four steps of one product, 4.5 us per level)
//   halve       sum_b (b + 1) B_b by halving levels (pair sums + odd items = bit planes of b), depth log2(buckets)
//                                                                                                       [summation by parts, msm.rs:37-45]
//   combine     window sum S_w = A + sum_l 2^l T_l: lane l doubles its plane l times, a tree adds the planes -- the host's chain then has
//               ONE addition per window instead of c (255 doublings + W additions: ~65 us instead of ~130)
//   export      S_w -> ABI words in the slot's pinned host buffer
//
// One workgroup per window up to 2^10 pairs (grid = W, everything above in one kernel).  Longer inputs split a window's buckets over NB
// workgroups (grid = W x NB; every workgroup scans all digits and keeps its range): each leaves its r + 1 local planes in global memory
// and k_msm_small_combine (grid = W) adds them across the ranges -- the planes of the high bucket bits are sums of the ranges' totals --
// and runs the combine.  Results are the same group elements as the long pipeline's (affine parity: SURVEY.md 8c).
#pragma once
#include "msm_acc_kernels.h"
#include "msm_digits.h"
#include "coop_add.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

constexpr int SM_NT = 256;            // threads per workgroup: one wave per SIMD of its CU
constexpr uint32_t SM_TASKS = 256;    // tasks per workgroup (one per lane)
constexpr uint32_t SM_SKIP = 0xffffu;
constexpr uint32_t SM_MAX_N = 1u << 13;      // longest input whose workgroups convert the scalars themselves (LDS: a digit code and a list entry per scalar)
constexpr int SM_MAX_R = 7;           // at most 128 buckets per workgroup

struct SmallArgs {
  const uint64_t* bases;      // ABI affine points (x | y), Montgomery R = 2^256
  const uint8_t* inf;         // identity flags or nullptr
  const uint64_t* scalars;    // ABI scalars
  uint32_t n;                 // scalars the windows are cut from: the input's, or -- glv -- two half-length ones per input scalar (entry v: input v >> 1)
  uint32_t nr;                // pairs of the input
  int glv;                    // 1: every scalar as k1 + k2 lambda (msm_digits.h glv_decompose_with), sub-scalar v & 1 = 1 meets (beta x, y)
  int npl;                    // word planes of the KT form: 8, or 4 (glv: 128-bit sub-scalars)
  const uint8_t* meta;        // glv, KT form: per sub-scalar its sign (bit 0) and the base's identity flag (bit 1), written by k_small_prep
  int c, W, r, NB;            // window width, windows, log2(buckets per workgroup), workgroups per window (NB << r == 2^(c-1))
  Words8 H;                   // digit bias (msm_digits.h)
  uint64_t* out;              // NB == 1: W window sums, 4 * E64 words each (x | y | zz | zzz, ABI form) -- the slot's pinned buffer
  uint32_t* planes;           // NB > 1: [W][NB][r + 1] plane points, raw internal form, NW words each
  const uint32_t* kt;         // KT form (longer inputs): the scalars as biased integers, word j of scalar i at kt[j * n + i] (k_small_prep)
  uint16_t* spill;            // KT form: W * n list entries of overflow space (a workgroup whose entries outgrow its LDS list takes a run of it)
  uint32_t* spill_cursor;     // KT form: W counters, zeroed by k_small_prep
#ifdef KG_EXPERIMENTS
  uint64_t* stamps;           // KG_SMALL_STAMPS=1 (A/B builds): wall-clock ticks (10 ns) of workgroup (0, 0) at its phase boundaries
#endif
};
#ifdef KG_EXPERIMENTS
#define KG_SM_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[k] = wall_clock64(); } while (0)
#else
#define KG_SM_STAMP(k) ((void)0)
#endif

template <class F> struct SmIO;
template <class P> struct SmIO<Fp<P>> {
  static constexpr int E64 = 4;
  // raw ABI words of point i (prefetched one entry ahead), and their conversion
  struct Raw { uint32_t w[16]; };
  static __device__ __forceinline__ Raw fetch(const uint64_t* bases, uint32_t i) {
    Raw r;
    const uint4* p = reinterpret_cast<const uint4*>(bases) + 4 * (size_t)i;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint4 v = p[j]; r.w[4 * j] = v.x; r.w[4 * j + 1] = v.y; r.w[4 * j + 2] = v.z; r.w[4 * j + 3] = v.w; }
    return r;
  }
  static __device__ __forceinline__ Affine<Fp<P>> convert(const Raw& r) { return {from_ref<P>(r.w), from_ref<P>(r.w + 8)}; }
};
template <class G> struct SmIO<Fp2<G>> {
  static constexpr int E64 = 8;
  using P = typename G::Params;
  struct Raw { uint32_t w[32]; };
  static __device__ __forceinline__ Raw fetch(const uint64_t* bases, uint32_t i) {
    Raw r;
    const uint4* p = reinterpret_cast<const uint4*>(bases) + 8 * (size_t)i;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint4 v = p[j]; r.w[4 * j] = v.x; r.w[4 * j + 1] = v.y; r.w[4 * j + 2] = v.z; r.w[4 * j + 3] = v.w; }
    return r;
  }
  static __device__ __forceinline__ Affine<Fp2<G>> convert(const Raw& r) {
    return {{from_ref<P>(r.w), from_ref<P>(r.w + 8)}, {from_ref<P>(r.w + 16), from_ref<P>(r.w + 24)}};
  }
};

// raw internal XYZZ -> ABI words (x | y | zz | zzz)
template <class P>
__device__ __forceinline__ void sm_export_el(const Fp<P>& a, uint64_t* dst) {
  uint32_t w[8];
  to_ref(a, w);
  store_words(dst, 0, w);
}
template <class G>
__device__ __forceinline__ void sm_export_el(const Fp2<G>& a, uint64_t* dst) { sm_export_el(a.c0, dst); sm_export_el(a.c1, dst + 4); }
template <class F>
__device__ __forceinline__ void sm_export(const XYZZ<F>& p, uint64_t* dst) {
  constexpr int E = SmIO<F>::E64;
  sm_export_el(p.x, dst); sm_export_el(p.y, dst + E); sm_export_el(p.zz, dst + 2 * E); sm_export_el(p.zzz, dst + 3 * E);
}

constexpr uint32_t SM_LIST_CAP = 4096;   // KT form: list entries a workgroup holds in LDS (more -- a skewed input -- spill to global memory)
constexpr uint32_t SM_MAX_N_KT = 1u << 15;   // KT form: 15-bit index field
constexpr uint32_t SM_MAX_N_G2 = 20480;      // G2: beyond, the long pipeline is as fast (profiles/r06_small_shapes_g2.txt)

// The longer short inputs (from 2049 pairs): ONE conversion of the scalars for all workgroups -- k + H as eight word planes -- instead of one
// per workgroup (n / 256 Montgomery products per lane in each of the W * NB workgroups: 32 us at 2^13 pairs); also clears the spill cursors.
template <class SP>
__global__ void __launch_bounds__(256) k_small_prep(const uint64_t* __restrict__ scalars, uint32_t nr, Words8 H, uint32_t* __restrict__ kt,
                                                    uint32_t* __restrict__ spill_cursor, int W, int glv, const uint8_t* __restrict__ inf, uint8_t* __restrict__ meta) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (uint32_t)W) spill_cursor[i] = 0;
  if (i >= nr) return;
  uint32_t w[8], k[8];
  load_words(scalars, i, w);
  ref_to_int<SP>(w, k);
  if (glv) {                                                // two sub-scalars of 127 bits: |k_e| + H as four word planes, sign and identity flag aside
    uint32_t ks[2][4];
    bool ng[2];
    glv_decompose_with<GlvLattice<SP>>(k, ks[0], ng[0], ks[1], ng[1]);
    const size_t nv = (size_t)2 * nr;
    const uint32_t fl = (inf && inf[i]) ? 2u : 0u;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      uint64_t cy = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t s = (uint64_t)ks[e][j] + H.w[j] + cy;
        kt[(size_t)j * nv + 2 * i + e] = (uint32_t)s;
        cy = s >> 32;
      }
      meta[2 * i + e] = (uint8_t)(fl | (ng[e] ? 1u : 0u));
    }
    return;
  }
  uint64_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t s = (uint64_t)k[j] + H.w[j] + cy;
    kt[(size_t)j * nr + i] = (uint32_t)s;
    cy = s >> 32;
  }
}
// the two words of the planes a window's digit lives in, for SM_UN scalars of a lane at once: the loads of a batch are all in flight
// before the first digit is cut (one at a time their latency -- 0.5-1 us out of L2 / HBM -- is the whole loop: 64 rounds of it at 2^14 pairs)
constexpr int SM_UN = 8;
struct SmWords { uint32_t lo[SM_UN], hi[SM_UN]; };
__device__ __forceinline__ SmWords sm_fetch_kt(const uint32_t* __restrict__ kt, uint32_t n, uint32_t i0, int w, int c, int W, int npl) {
  const int o = w * c, j = o >> 5, sh = o & 31;
  const bool two = j + 1 < npl && (sh + c > 32 || w == W - 1);
  SmWords v;
#pragma unroll
  for (int u = 0; u < SM_UN; ++u) {
    const uint32_t i = i0 + (uint32_t)u * SM_NT;
    v.lo[u] = i < n ? kt[(size_t)j * n + i] : 0u;
    v.hi[u] = (two && i < n) ? kt[(size_t)(j + 1) * n + i] : 0u;
  }
  return v;
}
__device__ __forceinline__ uint32_t sm_digit_words(uint32_t lo, uint32_t hi, int w, int c, int W, bool& negative) {
  const int sh = (w * c) & 31;
  const uint32_t e = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
  if (w == W - 1) { negative = false; return e; }
  const int32_t d = (int32_t)(e & ((1u << c) - 1u)) - (int32_t)(1u << (c - 1));
  negative = d < 0;
  return (uint32_t)(d < 0 ? -d : d);
}
// LDS words of the cooperative additions' temporaries and flags (coop_add.h) for a 256-thread workgroup
template <class F> constexpr uint32_t sm_coop_words() { return coop_lds_words<F>(SM_NT / 4); }      // (none since the steps keep their values in registers)
constexpr int SM_CLASSES = 9;          // a bucket owns 2^k task slots, k = 0 .. 8

// LDS bytes of k_msm_small for n scalars and 2^r buckets per workgroup
template <class F>
static size_t small_lds_bytes(uint32_t n, int r, bool kt = false) {
  const size_t R = (size_t)1 << r, cap = SM_TASKS + R + 1;
  const size_t n_pad = (n + 1) & ~(size_t)1;
  const size_t lists = kt ? (size_t)SM_LIST_CAP : 2 * n_pad;        // KT form: no digit codes, a capped list
  return ((size_t)PointIO<F>::NW * cap + sm_coop_words<F>() + 4 * R + 2 + 80) * 4 + (SM_TASKS + lists) * 2 + 64;
}

// S = P_0 + sum_{l >= 0} 2^l P_{1 + l} over `np` plane points held as items 0 .. np-1 of an LDS image: quad 1 + l doubles its plane l
// times (cooperative doubling: three steps each), then a tree of cooperative additions over the planes; the sum ends in item 0.  All
// threads of the workgroup call it.
template <class F>
__device__ __forceinline__ void sm_combine_planes(uint32_t* img, uint32_t cap, uint32_t* tmp, uint32_t* flg, uint32_t np) {
  const uint32_t qd = threadIdx.x >> 2;
  coop_dbl_level<F>(img, cap, tmp, flg, qd < np ? qd : 0u, (qd >= 2 && qd < np) ? qd - 1 : 0u, np > 2 ? np - 2 : 0u);
  for (uint32_t s = 1; s < np; s <<= 1) {
    const uint32_t t = qd * 2 * s;
    const bool on = t + s < np;
    coop_add_level<F>(img, cap, tmp, flg, on, on ? t : 0u, on ? t + s : 0u, on ? t : 0u);
  }
}

// KT form: bit 1 = the entry's base is the identity (skip), bit 0 = the sub-scalar is negative (glv)
__device__ __forceinline__ uint32_t sm_flags(const SmallArgs& a, uint32_t i) {
  if (a.glv) return a.meta[i];
  return (a.inf && a.inf[i]) ? 2u : 0u;
}
// (beta x, y): the endomorphism's image of an affine point
template <class P, class SP>
__device__ __forceinline__ void sm_endo(Affine<Fp<P>>& pt) { pt.x = mul(pt.x, Fp<P>::from_const(GlvBeta<P, SP>::BETA)); }
template <class G, class SP>
__device__ __forceinline__ void sm_endo(Affine<Fp2<G>>& pt) {
  const G beta = G::from_const(GlvBetaG2::BETA);
  pt.x = {mul(pt.x.c0, beta), mul(pt.x.c1, beta)};
}
template <class F, class SP> struct SmEndo;
template <class P, class SP> struct SmEndo<Fp<P>, SP> { static __device__ __forceinline__ void apply(Affine<Fp<P>>& pt) { sm_endo<P, SP>(pt); } };
template <class G, class SP> struct SmEndo<Fp2<G>, SP> { static __device__ __forceinline__ void apply(Affine<Fp2<G>>& pt) { sm_endo<G, SP>(pt); } };

template <class F, class SP, bool KT>
__global__ void __launch_bounds__(SM_NT) k_msm_small(SmallArgs a) {
  extern __shared__ uint32_t lds[];
  constexpr int NW = PointIO<F>::NW;
  const uint32_t tid = threadIdx.x, qd = tid >> 2;
  const int c = a.c, W = a.W, r = a.r;
  const uint32_t R = 1u << r;
  // A split window's buckets are dealt to its NB workgroups INTERLEAVED: bucket b (0-based) belongs to workgroup b mod NB, local index b / NB.
  // The top window holds fewer bits than the others (values 0 .. 83 of 128 buckets for eight-bit windows over 127-bit halves): in contiguous
  // ranges its first workgroups would carry 1.5 times the entries of any other workgroup of the launch, and the rest none.
  const uint32_t w = blockIdx.x, jb = blockIdx.y, nbm = (uint32_t)a.NB - 1u, nbs = (uint32_t)(c - 1 - r);      // NB = 2^nbs
  const uint32_t n = a.n, n_pad = (n + 1u) & ~1u;
  const uint32_t CAP = SM_TASKS + R + 1, IDENT = SM_TASKS + R;      // the last item stays the identity (an empty bucket's sum)
  uint32_t* const pts = lds;                                // NW planes x CAP items: items [0, SM_TASKS) task sums / image Y, [SM_TASKS, SM_TASKS + R) image X
  uint32_t* const ctmp = pts + (size_t)NW * CAP;            // temporaries of the cooperative additions
  uint32_t* const cflg = ctmp + (sm_coop_words<F>() ? COOP_TMP_SLOTS * CoopEl<F>::E * (SM_NT / 4) : 0u);
  uint32_t* const hist = ctmp + sm_coop_words<F>();                  // R: entries per bucket, then the scatter cursors
  uint32_t* const boff = hist + R;                          // R + 1: first list position of each bucket
  uint32_t* const tfirst = boff + R + 1;                    // R: first task slot of each bucket (IDENT for an empty one)
  uint32_t* const tsize = tfirst + R;                       // R + 1: task slots of each bucket (0 or a power of two)
  uint32_t* const misc = tsize + R + 1;                     // 80: scan scratch [0, 40), slot classes: count [40, 49), base [50, 60)
  uint16_t* const task_b = reinterpret_cast<uint16_t*>(misc + 80);    // SM_TASKS: bucket of each task slot
  uint16_t* const dig = task_b + SM_TASKS;                  // n: (local bucket index) | sign << 15, or SM_SKIP (not in the KT form: the digit is cut again)
  uint16_t* sorted = KT ? dig : dig + n_pad;                // (index | sign << 15) ordered by bucket: n entries; KT form: SM_LIST_CAP, or a run of the spill space

  KG_SM_STAMP(0);
  // ---- digits of this window; histogram of the workgroup's bucket range
  for (uint32_t t = tid; t < R; t += SM_NT) hist[t] = 0;
  if (tid < 20) misc[40 + tid] = 0;
  for (uint32_t k = tid; k < (uint32_t)NW; k += SM_NT) pts[(size_t)k * CAP + IDENT] = 0u;
  __syncthreads();
  if constexpr (KT) {
    for (uint32_t i0 = tid; i0 < n; i0 += SM_UN * SM_NT) {
      const SmWords v = sm_fetch_kt(a.kt, n, i0, (int)w, c, W, a.npl);
      uint32_t fl[SM_UN];
#pragma unroll
      for (int u = 0; u < SM_UN; ++u) { const uint32_t i = i0 + (uint32_t)u * SM_NT; fl[u] = i < n ? sm_flags(a, i) : 0u; }
#pragma unroll
      for (int u = 0; u < SM_UN; ++u) {
        const uint32_t i = i0 + (uint32_t)u * SM_NT;
        if (i >= n) break;
        bool neg;
        const uint32_t m = (fl[u] & 2u) ? 0u : sm_digit_words(v.lo[u], v.hi[u], (int)w, c, W, neg);
        if (m && ((m - 1) & nbm) == jb) atomicAdd(&hist[(m - 1) >> nbs], 1u);
      }
    }
  } else if (a.glv) {
    for (uint32_t i = tid; i < a.nr; i += SM_NT) {          // two sub-scalars per scalar: entries 2 i and 2 i + 1
      uint32_t ks[2][8];
      bool ng[2];
      {
        uint32_t sw[8], k[8];
        load_words(a.scalars, i, sw);
        ref_to_int<SP>(sw, k);
        glv_decompose_with<GlvLattice<SP>>(k, ks[0], ng[0], ks[1], ng[1]);
      }
      const bool ident = a.inf && a.inf[i];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        uint64_t cy = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t s = (uint64_t)ks[e][j] + a.H.w[j] + cy;
          ks[e][j] = (uint32_t)s;
          cy = s >> 32;
        }
#pragma unroll
        for (int j = 4; j < 8; ++j) ks[e][j] = 0;
        bool neg;
        uint32_t m = small_window_digit(ks[e], (int)w, c, W, neg);
        if (ident) m = 0;
        uint32_t code = SM_SKIP;
        if (m && ((m - 1) & nbm) == jb) {
          code = ((m - 1) >> nbs) | ((neg != ng[e]) ? 0x8000u : 0u);
          atomicAdd(&hist[(m - 1) >> nbs], 1u);
        }
        if constexpr (!KT) dig[2 * i + e] = (uint16_t)code;
      }
    }
  } else
  for (uint32_t i = tid; i < n; i += SM_NT) {
    bool neg;
    uint32_t m;
    {
      uint32_t sw[8], k[8];
      load_words(a.scalars, i, sw);
      ref_to_int<SP>(sw, k);
      uint64_t cy = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint64_t s = (uint64_t)k[j] + a.H.w[j] + cy;
        k[j] = (uint32_t)s;
        cy = s >> 32;
      }
      m = small_window_digit(k, (int)w, c, W, neg);
    }
    if (a.inf && a.inf[i]) m = 0;                          // identity base (msm.rs adds it as a no-op)
    uint32_t code = SM_SKIP;
    if (m && ((m - 1) & nbm) == jb) {
      code = ((m - 1) >> nbs) | (neg ? 0x8000u : 0u);
      atomicAdd(&hist[(m - 1) >> nbs], 1u);
    }
    if constexpr (!KT) dig[i] = (uint16_t)code;
  }
  __syncthreads();
  KG_SM_STAMP(1);
  // ---- bucket offsets; task slots.  A bucket of cnt entries is cut into 2^k tasks (the least power of two of ceil(cnt / T)), and the
  // buckets take their slots in DESCENDING order of k: every bucket's first slot is then a multiple of its size, and the buckets with at
  // least 2s slots are a prefix [0, P_s) of the slot space -- level s of the merge tree is the regular pair list (t, t + s), t = 0, 2s, 4s ..
  // below P_s, without a table.  T = 2 total / (slots - R) keeps the slot count within SM_TASKS (2^k < 2 cnt / T for k >= 1).
  const uint32_t cnt = tid < R ? hist[tid] : 0u;
  uint32_t total = 0;
  const uint32_t off = block_exclusive_scan_1024(cnt, misc, total);
  // T: the shortest task length whose power-of-two slot counts fit -- from total / (slots - R) (no rounding loss) upwards by a quarter per
  // try; 2 total / (slots - R) always fits (2^k < 2 cnt / T for k >= 1), so at most four tries
  uint32_t T = (total + (SM_TASKS - R) - 1) / (SM_TASKS - R);
  if (T < 1) T = 1;
  const uint32_t T_safe = (2 * total + (SM_TASKS - R) - 1) / (SM_TASKS - R);
  uint32_t kcls = 0;
  for (;;) {
    kcls = 0;
    while (((cnt + T - 1) / T) > (1u << kcls)) ++kcls;      // ceil(cnt / T) <= 2^kcls
    if (T >= T_safe) break;
    uint32_t slots = 0;
    (void)block_exclusive_scan_1024((tid < R && cnt) ? 1u << kcls : 0u, misc, slots);
    __syncthreads();
    if (slots <= SM_TASKS) break;
    T = T + (T + 3) / 4;
    if (T > T_safe) T = T_safe;
  }
  uint32_t rank = 0;
  if (tid < R && cnt) rank = atomicAdd(&misc[40 + kcls], 1u);
  __syncthreads();
  if (tid == 0) {                                           // class bases: slots of all larger classes; misc[50 + j] = P for level s = 2^j
    uint32_t run = 0;
    for (int k = SM_CLASSES - 1; k >= 0; --k) { misc[50 + k] = run; run += misc[40 + k] << k; }
    misc[59] = run;                                         // slots in use
  }
  __syncthreads();
  const uint32_t nslots = misc[59];
  if (tid < R) {
    const uint32_t tf = cnt ? misc[50 + kcls] + (rank << kcls) : IDENT;
    boff[tid] = off;
    hist[tid] = off;                                        // scatter cursor
    tfirst[tid] = tf;
    tsize[tid] = cnt ? 1u << kcls : 0u;
    if (cnt) for (uint32_t k = 0; k < (1u << kcls); ++k) task_b[tf + k] = (uint16_t)tid;
  }
  if (tid == 0) {
    boff[R] = total;
    if (KT && total > SM_LIST_CAP) misc[60] = atomicAdd(&a.spill_cursor[w], total);      // a skewed input: this workgroup's list lives in the window's spill run
  }
  __syncthreads();
  if (KT && total > SM_LIST_CAP) sorted = a.spill + (size_t)w * n + misc[60];             // (a window's workgroups hold at most n entries together)
  // ---- scatter
  if constexpr (KT) {                                       // the digits again (two word loads each): no digit codes are kept for 2^15 scalars
    for (uint32_t i0 = tid; i0 < n; i0 += SM_UN * SM_NT) {
      const SmWords v = sm_fetch_kt(a.kt, n, i0, (int)w, c, W, a.npl);
      uint32_t fl[SM_UN];
#pragma unroll
      for (int u = 0; u < SM_UN; ++u) { const uint32_t i = i0 + (uint32_t)u * SM_NT; fl[u] = i < n ? sm_flags(a, i) : 0u; }
#pragma unroll
      for (int u = 0; u < SM_UN; ++u) {
        const uint32_t i = i0 + (uint32_t)u * SM_NT;
        if (i >= n) break;
        bool neg;
        const uint32_t m = (fl[u] & 2u) ? 0u : sm_digit_words(v.lo[u], v.hi[u], (int)w, c, W, neg);
        if (m && ((m - 1) & nbm) == jb) {
          const uint32_t pos = atomicAdd(&hist[(m - 1) >> nbs], 1u);
          sorted[pos] = (uint16_t)(i | ((neg != ((fl[u] & 1u) != 0)) ? 0x8000u : 0u));     // the digit's sign times the sub-scalar's
        }
      }
    }
  } else
  for (uint32_t i = tid; i < n; i += SM_NT) {
    const uint32_t code = dig[i];
    if (code != SM_SKIP) {
      const uint32_t pos = atomicAdd(&hist[code & 0x7fffu], 1u);
      sorted[pos] = (uint16_t)(i | (code & 0x8000u));
    }
  }
  if (KT && total > SM_LIST_CAP) __threadfence_block();     // the spilled list is read back by this workgroup
  __syncthreads();
  KG_SM_STAMP(2);
  // ---- accumulate: one lane per task slot (slot i of a bucket with 2^k slots takes entries [i * per, (i + 1) * per), per = ceil(cnt / 2^k))
  if (tid < nslots) {
    const uint32_t b = task_b[tid];
    const uint32_t cb = boff[b + 1] - boff[b], per = (cb + tsize[b] - 1) / tsize[b];
    const uint32_t lo = boff[b] + (tid - tfirst[b]) * per;
    uint32_t hi = lo + per;
    if (hi > boff[b + 1]) hi = boff[b + 1];
    XYZZ<F> acc = XYZZ<F>::identity();
    if (lo < hi) {
      const uint32_t g = (uint32_t)a.glv;                   // glv: entry v is sub-scalar v & 1 of pair v >> 1
      uint32_t e = sorted[lo];
      typename SmIO<F>::Raw raw = SmIO<F>::fetch(a.bases, (e & 0x7fffu) >> g);
      for (uint32_t j = lo; j < hi; ++j) {
        const bool neg = (e & 0x8000u) != 0;
        const bool endo = (e & g) != 0;
        Affine<F> pt = SmIO<F>::convert(raw);
        if (j + 1 < hi) {                                   // the next base travels while this addition runs
          e = sorted[j + 1];
          raw = SmIO<F>::fetch(a.bases, (e & 0x7fffu) >> g);
        }
        if (endo) SmEndo<F, SP>::apply(pt);                 // k2's half: lambda P = (beta x, y)
        acc = add_mixed_signed(acc, pt, neg);
      }
    }
    PointIO<F>::store(pts, CAP, tid, acc);
  }
  __syncthreads();
  KG_SM_STAMP(3);
  // ---- partial sums of one bucket: cooperative additions (coop_add.h: a quad of lanes per addition), level s = 2^j over the slot prefix
  // misc[50 + j], in place on the left slot, 64 quads per round
  for (int j = 0; j < SM_CLASSES - 1; ++j) {
    const uint32_t P = misc[50 + j], s = 1u << j;
    if (P == 0) break;                                      // no bucket has more than 2^j slots (the prefixes shrink with j)
    const uint32_t pairs = P >> (j + 1);
    for (uint32_t base = 0; base < pairs; base += SM_NT / 4) {
      const uint32_t pi = base + qd;
      const bool on = pi < pairs;
      const uint32_t t = on ? pi << (j + 1) : 0u;
      coop_add_level<F>(pts, CAP, ctmp, cflg, on, t, t + (on ? s : 0u), t);
    }
  }
  KG_SM_STAMP(4);
  // ---- halving levels over the R bucket sums (bucket i: its first slot, or the identity item).  Step t reads t arrays of 2 * (R >> t)
  // items and writes t + 1 arrays of R >> t items: array k at item k * (R >> t), the odd items of array 0 become array t.  Images
  // alternate: odd steps write X (items SM_TASKS ..), even steps write Y (items 0 ..: the task sums are dead after step 1).  A quad per
  // pair; t * (R >> t) <= R / 2 <= 64 pairs: one round.
  for (int t = 1; t <= r; ++t) {
    const uint32_t per = R >> t;
    const bool odd_step = (t & 1) != 0;
    const uint32_t obase = odd_step ? SM_TASKS : 0u, ibase = odd_step ? 0u : SM_TASKS;
    const bool on = qd < (uint32_t)t * per;
    uint32_t i0 = 0, i1 = 0, io = 0;
    if (on) {
      const uint32_t k = qd / per, q = qd % per;
      if (t == 1) { i0 = tfirst[2 * q]; i1 = tfirst[2 * q + 1]; }
      else { i0 = ibase + k * 2 * per + 2 * q; i1 = i0 + 1; }
      io = obase + k * per + q;
      if (k == 0) {                                         // the odd item of array 0 spawns array t: lane j of the quad copies coordinate j
        const CoopQuad<F> cq{pts, CAP, ctmp, SM_NT / 4, qd, cflg, i1, i1, obase + (uint32_t)t * per + q, (int)(tid & 3u)};
        cq.st(cq.coord(cq.io, (uint32_t)cq.lane), cq.ld(cq.coord(i1, (uint32_t)cq.lane)));
      }
    }
    coop_add_level<F>(pts, CAP, ctmp, cflg, on, i0, i1, io);
  }
  KG_SM_STAMP(5);
  // the r + 1 planes: item 0 = A (all buckets), item 1 + l = T_l (buckets whose local index has bit l set)
  const uint32_t fin = (r & 1) ? SM_TASKS : 0u;
  const uint32_t np = (uint32_t)r + 1u;
  uint32_t* const my_planes = a.planes + (size_t)(w * (uint32_t)a.NB + jb) * (SM_MAX_R + 1) * NW;
  if (r == 0) {                                             // one bucket per workgroup: its sum is the only plane
    if (tid == 0) {
      const XYZZ<F> p = PointIO<F>::load(pts, CAP, tfirst[0]);
      if (a.NB > 1) PointAoS<F>::store(my_planes, 0, p);
      else sm_export(p, a.out + (size_t)w * 4 * SmIO<F>::E64);
    }
    return;
  }
  if (a.NB > 1) {                                           // a bucket range of a split window: the planes go to global memory
    if (tid < np) PointAoS<F>::store(my_planes + (size_t)tid * NW, 0, PointIO<F>::load(pts, CAP, fin + tid));
    KG_SM_STAMP(6);                                         // (a split window: "combine" is the planes' store, the second launch is not in the stamps)
    KG_SM_STAMP(7);
    return;
  }
  sm_combine_planes<F>(pts + fin, CAP, ctmp, cflg, np);
  KG_SM_STAMP(6);
  if (tid == 0) sm_export(PointIO<F>::load(pts, CAP, fin), a.out + (size_t)w * 4 * SmIO<F>::E64);
  KG_SM_STAMP(7);
}

// Second launch of a split window (its 2^(c-1) buckets dealt to NB workgroups, bucket b to workgroup b mod NB): the plane of bucket bit
// h < log2 NB is the sum of the totals A_j of the workgroups whose index j has bit h set, the plane of bit log2 NB + l the sum of the
// workgroups' local planes l, the total the sum of all A_j.  LDS: c planes x NB items; a tree of cooperative additions along the ranges, then the combine of the fused kernel.
template <class F>
__global__ void __launch_bounds__(SM_NT) k_msm_small_combine(SmallArgs a) {
  extern __shared__ uint32_t lds[];
  constexpr int NW = PointIO<F>::NW;
  const uint32_t tid = threadIdx.x, w = blockIdx.x, qd = tid >> 2;
  const int c = a.c, r = a.r;
  const uint32_t NB = (uint32_t)a.NB, np = (uint32_t)c;     // planes of the whole window: total, T_0 .. T_{c-2}
  const uint32_t CAP = np * NB;
  uint32_t* const ctmp = lds + (size_t)NW * (CAP + np);
  uint32_t* const cflg = ctmp + (sm_coop_words<F>() ? COOP_TMP_SLOTS * CoopEl<F>::E * (SM_NT / 4) : 0u);
  // item (p, j) at p * NB + j
  for (uint32_t it = tid; it < CAP; it += SM_NT) {
    const uint32_t p = it / NB, j = it % NB;
    XYZZ<F> v = XYZZ<F>::identity();
    const uint32_t* src = a.planes + (size_t)(w * NB + j) * (SM_MAX_R + 1) * NW;
    // bucket b = local * NB + j (interleaved): bit h < log2 NB of b is bit h of the workgroup's index j -- plane 1 + h is the sum of the totals
    // of the workgroups whose j has it set; bit log2 NB + l is the local plane l
    const uint32_t nbs = (uint32_t)c - 1u - (uint32_t)r;
    if (p == 0) v = PointAoS<F>::load(src, 0);                                                    // the total
    else if (p <= nbs) { if ((j >> (p - 1)) & 1u) v = PointAoS<F>::load(src, 0); }                // low bucket bit h = p - 1: the workgroup's total
    else v = PointAoS<F>::load(src + (size_t)(p - nbs) * NW, 0);                                  // local plane l = p - nbs - 1: item 1 + l
    PointIO<F>::store(lds, CAP, it, v);
  }
  __syncthreads();
  for (uint32_t s = 1; s < NB; s <<= 1) {
    const uint32_t per_plane = NB / (2 * s), pairs = np * per_plane;
    for (uint32_t base = 0; base < pairs; base += SM_NT / 4) {
      const uint32_t pi = base + qd;
      const bool on = pi < pairs;
      const uint32_t t = on ? (pi / per_plane) * NB + (pi % per_plane) * 2 * s : 0u;
      coop_add_level<F>(lds, CAP, ctmp, cflg, on, t, t + (on ? s : 0u), t);
    }
  }
  // gather the planes' sums (item p * NB) into items 0 .. np-1 of a second image behind the first
  uint32_t* const img = lds + (size_t)NW * CAP;
  if (tid < np) PointIO<F>::store(img, np, tid, PointIO<F>::load(lds, CAP, tid * NB));
  __syncthreads();
  sm_combine_planes<F>(img, np, ctmp, cflg, np);
  if (tid == 0) sm_export(PointIO<F>::load(img, np, 0), a.out + (size_t)w * 4 * SmIO<F>::E64);
}
template <class F>
static size_t small_combine_lds_bytes(int c, int NB) { return ((size_t)PointIO<F>::NW * ((size_t)c * NB + c) + sm_coop_words<F>()) * 4; }

}  // namespace
}  // namespace msm
}  // namespace kg
