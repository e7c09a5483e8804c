// msm_small_kernels.h -- the MSM of SHORT inputs (n <= 2^13 pairs) in ONE launch (two from 2^11): the reference's own tests and its bench
// live at these lengths (groth16/src/msm.rs:118-135: 32 pairs; bn254/benches: 2^10; groth16/src/lib.rs:29-77: a handful of constraints).
//
// The long-input pipeline (msm_sort.hip / msm_run.hip) is a chain of 15-20 dependent launches with one read-back in the middle: 0.24 ms
// for 16 pairs, 0.44 ms for 2^10 .. 2^13.  Short inputs are latency, so here a workgroup owns a (window, bucket range) and does everything
// for it out of LDS:
//
//   digits      every lane converts scalars to integers (one Montgomery product) and cuts THIS window's signed digit (msm_digits.h);
//               entries of the workgroup's bucket range are counted in an LDS histogram              [replaces get_at, msm.rs:75-91]
//   sort        exclusive scan, scatter of (index | sign) into an LDS list ordered by bucket
//   tasks       the bucket lists are cut into <= 256 tasks of at most T entries (T from the entry count: a skewed input -- every scalar
//               equal -- still fills the lanes)
//   accumulate  one lane per task: XYZZ += +-P (madd-2008-s), bases read in the ABI form and converted on the fly, the next base on its way
//               while the current addition runs                                                       [bucket fill, msm.rs:25-35]
//   merge       partial sums of one bucket are added by a tree in LDS (depth log2 of the most tasks a bucket has)
//   halve       sum_b (b + 1) B_b by halving levels (pair sums + odd items = bit planes of b), depth log2(buckets), operands in REGISTERS
//               (the workgroup is alone on its CU: one wave per SIMD, 4.8 us per dependent addition)   [summation by parts, msm.rs:37-45]
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

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

constexpr int SM_NT = 256;            // threads per workgroup: one wave per SIMD of its CU
constexpr uint32_t SM_TASKS = 256;    // tasks per workgroup (one per lane)
constexpr uint32_t SM_SKIP = 0xffffu;
constexpr uint32_t SM_MAX_N = 1u << 13;      // index field of a list entry: 13 bits (+ sign in bit 15)
constexpr int SM_MAX_R = 7;           // at most 128 buckets per workgroup

struct SmallArgs {
  const uint64_t* bases;      // ABI affine points (x | y), Montgomery R = 2^256
  const uint8_t* inf;         // identity flags or nullptr
  const uint64_t* scalars;    // ABI scalars
  uint32_t n;
  int c, W, r, NB;            // window width, windows, log2(buckets per workgroup), workgroups per window (NB << r == 2^(c-1))
  Words8 H;                   // digit bias (msm_digits.h)
  uint64_t* out;              // NB == 1: W window sums, 4 * E64 words each (x | y | zz | zzz, ABI form) -- the slot's pinned buffer
  uint32_t* planes;           // NB > 1: [W][NB][r + 1] plane points, raw internal form, NW words each
};

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

// LDS bytes of k_msm_small for n scalars and 2^r buckets per workgroup
template <class F>
static size_t small_lds_bytes(uint32_t n, int r) {
  const size_t R = (size_t)1 << r, cap = SM_TASKS + R;
  const size_t n_pad = (n + 1) & ~(size_t)1;
  return ((size_t)PointIO<F>::NW * cap + R + 2 * (R + 1) + 48) * 4 + (SM_TASKS + 2 * n_pad) * 2 + 64;
}

// S = P_0 + sum_{l >= 0} 2^l P_{1 + l} over `np` plane points held as items 0 .. np-1 of an LDS image (structure of arrays, stride cap):
// lane 1 + l doubles its plane l times, then a tree over the np lanes; the sum ends in item 0.  All lanes of the workgroup call it.
template <class F>
__device__ __forceinline__ void sm_combine_planes(uint32_t* img, uint32_t cap, uint32_t np) {
  const uint32_t t = threadIdx.x;
  if (t >= 2 && t < np) {
    XYZZ<F> p = PointIO<F>::load(img, cap, t);
    for (uint32_t k = 1; k < t; ++k) p = double_xyzz(p);
    PointIO<F>::store(img, cap, t, p);
  }
  for (uint32_t s = 1; s < np; s <<= 1) {
    __syncthreads();
    if ((t & (2 * s - 1)) == 0 && t + s < np) {
      const XYZZ<F> p = PointIO<F>::load(img, cap, t), q = PointIO<F>::load(img, cap, t + s);
      PointIO<F>::store(img, cap, t, add_xyzz(p, q));
    }
  }
  __syncthreads();
}

template <class F, class SP>
__global__ void __launch_bounds__(SM_NT) k_msm_small(SmallArgs a) {
  extern __shared__ uint32_t lds[];
  constexpr int NW = PointIO<F>::NW;
  const uint32_t tid = threadIdx.x;
  const int c = a.c, W = a.W, r = a.r;
  const uint32_t R = 1u << r;
  const uint32_t w = blockIdx.x, jb = blockIdx.y, b0 = jb << r;
  const uint32_t n = a.n, n_pad = (n + 1u) & ~1u;
  const uint32_t CAP = SM_TASKS + R;
  uint32_t* const pts = lds;                                // NW planes x CAP items: items [0, SM_TASKS) task partial sums / image Y, [SM_TASKS, CAP) image X
  uint32_t* const hist = pts + (size_t)NW * CAP;            // R: entries per bucket, then the scatter cursors
  uint32_t* const boff = hist + R;                          // R + 1: first list position of each bucket
  uint32_t* const tfirst = boff + R + 1;                    // R + 1: first task of each bucket
  uint32_t* const misc = tfirst + R + 1;                    // 48: scan scratch [0, 40), most tasks of a bucket [40]
  uint16_t* const task_b = reinterpret_cast<uint16_t*>(misc + 48);    // SM_TASKS: bucket of each task
  uint16_t* const dig = task_b + SM_TASKS;                  // n: (bucket - b0) | sign << 15, or SM_SKIP
  uint16_t* const sorted = dig + n_pad;                     // n: (index | sign << 15) ordered by bucket

  // ---- digits of this window; histogram of the workgroup's bucket range
  for (uint32_t t = tid; t < R; t += SM_NT) hist[t] = 0;
  if (tid == 0) misc[40] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += SM_NT) {
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
    bool neg;
    uint32_t m = small_window_digit(k, (int)w, c, W, neg);
    if (a.inf && a.inf[i]) m = 0;                          // identity base (msm.rs adds it as a no-op)
    uint32_t code = SM_SKIP;
    if (m && ((m - 1) >> r) == jb) {
      code = (m - 1 - b0) | (neg ? 0x8000u : 0u);
      atomicAdd(&hist[m - 1 - b0], 1u);
    }
    dig[i] = (uint16_t)code;
  }
  __syncthreads();
  // ---- bucket offsets, task length, tasks per bucket
  const uint32_t cnt = tid < R ? hist[tid] : 0u;
  uint32_t total = 0;
  const uint32_t off = block_exclusive_scan_1024(cnt, misc, total);
  __syncthreads();                                          // the scan scratch is reused below
  uint32_t T = (total + (SM_TASKS - R) - 1) / (SM_TASKS - R);
  if (T < 1) T = 1;
  const uint32_t tc = (cnt + T - 1) / T;
  uint32_t ntasks = 0;
  const uint32_t tf = block_exclusive_scan_1024(tc, misc, ntasks);
  if (tid < R) {
    boff[tid] = off;
    hist[tid] = off;                                        // scatter cursor
    tfirst[tid] = tf;
    for (uint32_t k = 0; k < tc; ++k) task_b[tf + k] = (uint16_t)tid;
    if (tc > 1) atomicMax(&misc[40], tc);
  }
  if (tid == 0) { boff[R] = total; tfirst[R] = ntasks; }
  __syncthreads();
  // ---- scatter
  for (uint32_t i = tid; i < n; i += SM_NT) {
    const uint32_t code = dig[i];
    if (code != SM_SKIP) {
      const uint32_t pos = atomicAdd(&hist[code & 0x7fffu], 1u);
      sorted[pos] = (uint16_t)(i | (code & 0x8000u));
    }
  }
  __syncthreads();
  // ---- accumulate: one lane per task
  if (tid < ntasks) {
    const uint32_t b = task_b[tid];
    const uint32_t lo = boff[b] + (tid - tfirst[b]) * T;
    uint32_t hi = lo + T;
    if (hi > boff[b + 1]) hi = boff[b + 1];
    XYZZ<F> acc = XYZZ<F>::identity();
    uint32_t e = sorted[lo];
    typename SmIO<F>::Raw raw = SmIO<F>::fetch(a.bases, e & 0x7fffu);
    for (uint32_t j = lo; j < hi; ++j) {
      const bool neg = (e & 0x8000u) != 0;
      const Affine<F> pt = SmIO<F>::convert(raw);
      if (j + 1 < hi) {                                     // the next base travels while this addition runs
        e = sorted[j + 1];
        raw = SmIO<F>::fetch(a.bases, e & 0x7fffu);
      }
      acc = add_mixed_signed(acc, pt, neg);
    }
    PointIO<F>::store(pts, CAP, tid, acc);
  }
  // ---- partial sums of one bucket: tree over its tasks (in place: the writer reads a task nobody writes in that step)
  const uint32_t max_tc = misc[40];
  for (uint32_t s = 1; s < max_tc; s <<= 1) {
    __syncthreads();
    if (tid < ntasks) {
      const uint32_t b = task_b[tid], rel = tid - tfirst[b], tcb = tfirst[b + 1] - tfirst[b];
      if ((rel & (2 * s - 1)) == 0 && rel + s < tcb) {
        const XYZZ<F> p = PointIO<F>::load(pts, CAP, tid), q = PointIO<F>::load(pts, CAP, tid + s);
        PointIO<F>::store(pts, CAP, tid, add_xyzz(p, q));
      }
    }
  }
  __syncthreads();
  // ---- halving levels over the R bucket sums (bucket i: the first task of bucket i, or the identity).  Step t reads t arrays of
  // 2 * (R >> t) items and writes t + 1 arrays of R >> t items: array k at item k * (R >> t), the odd items of array 0 become array t.
  // Images alternate: odd steps write X (items SM_TASKS ..), even steps write Y (items 0 ..: the task sums are dead after step 1).
  for (int t = 1; t <= r; ++t) {
    const uint32_t per = R >> t;
    const bool odd_step = (t & 1) != 0;
    const uint32_t obase = odd_step ? SM_TASKS : 0u, ibase = odd_step ? 0u : SM_TASKS;
    if (tid < (uint32_t)t * per) {
      const uint32_t k = tid / per, q = tid % per;
      XYZZ<F> p0, p1;
      if (t == 1) {
        const uint32_t i0 = 2 * q, i1 = 2 * q + 1;
        p0 = tfirst[i0 + 1] > tfirst[i0] ? PointIO<F>::load(pts, CAP, tfirst[i0]) : XYZZ<F>::identity();
        p1 = tfirst[i1 + 1] > tfirst[i1] ? PointIO<F>::load(pts, CAP, tfirst[i1]) : XYZZ<F>::identity();
      } else {
        p0 = PointIO<F>::load(pts, CAP, ibase + k * 2 * per + 2 * q);
        p1 = PointIO<F>::load(pts, CAP, ibase + k * 2 * per + 2 * q + 1);
      }
      PointIO<F>::store(pts, CAP, obase + k * per + q, add_xyzz(p0, p1));       // (a step reads one image and writes the other: one barrier per step)
      if (k == 0) PointIO<F>::store(pts, CAP, obase + (uint32_t)t * per + q, p1);
    }
    __syncthreads();
  }
  // the r + 1 planes: item 0 = A (all buckets), item 1 + l = T_l (buckets whose local index has bit l set)
  const uint32_t fin = (r & 1) ? SM_TASKS : 0u;
  const uint32_t np = (uint32_t)r + 1u;
  uint32_t* const my_planes = a.planes + (size_t)(w * (uint32_t)a.NB + jb) * (SM_MAX_R + 1) * NW;
  if (r == 0) {                                             // one bucket per workgroup: its sum is the only plane
    if (tid == 0) {
      const XYZZ<F> p = tfirst[1] > tfirst[0] ? PointIO<F>::load(pts, CAP, tfirst[0]) : XYZZ<F>::identity();
      if (a.NB > 1) PointAoS<F>::store(my_planes, 0, p);
      else sm_export(p, a.out + (size_t)w * 4 * SmIO<F>::E64);
    }
    return;
  }
  if (a.NB > 1) {                                           // a bucket range of a split window: the planes go to global memory
    if (tid < np) PointAoS<F>::store(my_planes + (size_t)tid * NW, 0, PointIO<F>::load(pts, CAP, fin + tid));
    return;
  }
  sm_combine_planes<F>(pts + fin, CAP, np);
  if (tid == 0) sm_export(PointIO<F>::load(pts, CAP, fin), a.out + (size_t)w * 4 * SmIO<F>::E64);
}

// Second launch of a split window (NB > 1 bucket ranges of 2^r buckets): plane l < r is the sum of the ranges' local planes, plane
// r + h the sum of the totals A_j of the ranges whose index has bit h set (the high bits of the bucket number), the total the sum of
// all A_j.  LDS: c planes x NB items; a tree along the ranges, then the combine of the fused kernel.
template <class F>
__global__ void __launch_bounds__(SM_NT) k_msm_small_combine(SmallArgs a) {
  extern __shared__ uint32_t lds[];
  constexpr int NW = PointIO<F>::NW;
  const uint32_t tid = threadIdx.x, w = blockIdx.x;
  const int c = a.c, r = a.r;
  const uint32_t NB = (uint32_t)a.NB, np = (uint32_t)c;     // planes of the whole window: total, T_0 .. T_{c-2}
  const uint32_t CAP = np * NB;
  // item (p, j) at p * NB + j
  for (uint32_t it = tid; it < CAP; it += SM_NT) {
    const uint32_t p = it / NB, j = it % NB;
    XYZZ<F> v = XYZZ<F>::identity();
    const uint32_t* src = a.planes + (size_t)(w * NB + j) * (SM_MAX_R + 1) * NW;
    if (p <= (uint32_t)r) v = PointAoS<F>::load(src + (size_t)p * NW, 0);                       // total (p = 0) and the local planes
    else if ((j >> (p - 1 - (uint32_t)r)) & 1u) v = PointAoS<F>::load(src, 0);                  // high bucket bit h = p - 1 - r: the range's total
    PointIO<F>::store(lds, CAP, it, v);
  }
  for (uint32_t s = 1; s < NB; s <<= 1) {
    __syncthreads();
    for (uint32_t it = tid; it < CAP; it += SM_NT) {
      const uint32_t j = it % NB;
      if ((j & (2 * s - 1)) == 0 && j + s < NB) {
        const XYZZ<F> p = PointIO<F>::load(lds, CAP, it), q = PointIO<F>::load(lds, CAP, it + s);
        PointIO<F>::store(lds, CAP, it, add_xyzz(p, q));
      }
    }
  }
  __syncthreads();
  // gather the planes' sums (item p * NB) into items 0 .. np-1 of a second image behind the first
  uint32_t* const img = lds + (size_t)NW * CAP;
  if (tid < np) PointIO<F>::store(img, np, tid, PointIO<F>::load(lds, CAP, tid * NB));
  __syncthreads();
  sm_combine_planes<F>(img, np, np);
  if (tid == 0) sm_export(PointIO<F>::load(img, np, 0), a.out + (size_t)w * 4 * SmIO<F>::E64);
}
template <class F>
static size_t small_combine_lds_bytes(int c, int NB) { return (size_t)PointIO<F>::NW * ((size_t)c * NB + c) * 4; }

}  // namespace
}  // namespace msm
}  // namespace kg
