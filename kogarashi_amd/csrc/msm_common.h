// msm_common.h -- what every translation unit of the MSM shares on the device side: the curve configurations, the task bookkeeping of a
// round (Level) and the rules that derive tasks from bucket sizes (used by the sort's task decomposition AND by the accumulation).
// The units: msm_sort.hip (scalars -> sorted digit lists -> tasks; msm_sort_kernels.h, msm_task_kernels.h), msm_run.hip (bases,
// accumulation, bucket reduction; msm_bases.h, msm_acc_kernels.h, msm_reduce_kernels.h), msm_host.cpp (the C ABI's orchestration and
// the host finish).  Replaces groth16/src/msm.rs:6-48.
#pragma once
#include "common.h"
#include "host_fp.h"
#include "fp2s.h"
#include "msm_internal.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

struct G1Cfg { using F = Fq; using KF = Fq; using SP = FrParams; using HF = HostFq; static constexpr int E64 = 4; static constexpr int ID = KG_G1; };
struct GkCfg { using F = Fr; using KF = Fr; using SP = FqParams; using HF = HostFr; static constexpr int E64 = 4; static constexpr int ID = KG_GRUMPKIN; };
struct G2Cfg { using F = Fq2; using KF = Fp2S<Fq>; using SP = FrParams; using HF = HostFq2; static constexpr int E64 = 8; static constexpr int ID = KG_G2; };

constexpr uint32_t INF_BIT = 0x80000000u;   // bit 255 of the packed x coordinate marks an identity base

struct Level {            // one round's task bookkeeping, all device pointers
  const uint32_t* cnt;    // [W*B] tasks of each bucket in this round
  const uint32_t* rel;    // [W*B] exclusive prefix of cnt inside the window
  const uint32_t* base;   // [W+1] first task of each window; base[W] = total
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


// buffer addressing (descriptor + scalar plane offset + one 32-bit lane offset) instead of 64-bit flat addresses: the digit planes
// and the intermediate runs stay far below the 4 GiB a descriptor spans (two-pass sort: n <= 2^24; the reduction buffers: msm_reduce_kernels.h)
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc soa_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffffu, 0x00020000);
}

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace
}  // namespace msm
}  // namespace kg
