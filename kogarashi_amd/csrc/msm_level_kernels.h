// msm_level_kernels.h -- the bookkeeping kernels of one round of tasks (tasks per bucket, their prefix inside the window, the windows'
// bases): run by the sort's task decomposition and again by every extra partial-sum round of a skewed input (msm_run.hip).
#pragma once
#include "msm_common.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

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

// ---------------------------------------------------------------------------------------------------
// bucket accumulation, load balanced.  A bucket's list is cut into tasks of at most T entries; one lane per
// task.  Real witnesses are heavily skewed (0/1 scalars put ~n points in one bucket; the top window has few
// buckets), so partial sums of one bucket are re-summed in further rounds (T2 partials per lane) until every
// bucket owns a single point.  With uniform scalars every bucket is one task and no extra round runs.
// ---------------------------------------------------------------------------------------------------

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

}  // namespace
}  // namespace msm
}  // namespace kg
