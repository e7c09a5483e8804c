// msm_task_kernels.h -- bucket lists -> tasks: a bucket's list is cut into tasks of at most T entries (hot buckets finer), the tasks are
// ordered longest first so that the 64 lanes of a wave run equally long loops, and four words (task count, largest bucket, hot buckets,
// most tasks per bucket) go straight to pinned host memory.
#pragma once
#include "msm_level_kernels.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

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

}  // namespace
}  // namespace msm
}  // namespace kg
