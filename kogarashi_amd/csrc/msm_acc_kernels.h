// msm_acc_kernels.h -- the bucket accumulation: one lane per task, XYZZ += +-P over the task's list (madd-2008-s, 8M + 2S), bases gathered
// through L2 / Infinity Cache; partial sums in an array-of-structures layout.
#pragma once
#include "msm_bases.h"
#include "msm_level_kernels.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

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

#ifdef KG_EXPERIMENTS
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
#endif      // KG_EXPERIMENTS

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

}  // namespace
}  // namespace msm
}  // namespace kg
