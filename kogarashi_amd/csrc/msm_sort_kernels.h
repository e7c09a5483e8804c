// msm_sort_kernels.h -- scalars -> biased integers -> signed window digits -> per-bucket lists of (index | sign): the counting sort by
// (window, bucket), one pass for short inputs, two passes (bucket group, then bucket inside the group) once the lists outgrow the L2.
#pragma once
#include "msm_level_kernels.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

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
#ifdef KG_EXPERIMENTS      // the round-3 first pass (entries held in registers): lost to k_group_scatter_big, kept for A/B builds only (KG_GS_TILE=0)
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
#endif      // KG_EXPERIMENTS

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


}  // namespace
}  // namespace msm
}  // namespace kg
