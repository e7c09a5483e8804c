// common.h -- context, error plumbing and device-side layout helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <atomic>
#include <future>
#include <cstdio>
#include <string>
#include <memory>
#include <map>
#include <unordered_map>
#include <vector>
#include <thread>
#include <new>
#include "../../include/kogarashi_amd.h"
#include "curve.h"
#include "tuning.h"
#include "worker_pool.h"

struct kg_tw_cache;   // ntt.hip

struct kg_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;          // stream every launch goes to
  std::string last_error;
  kg_tuning tune;                        // the knob table (tuning.h): the process-wide values at creation
  int msm_window = 0;                    // 0 = auto
  bool sort_alone = false;               // hint for the next msm_sort / msm_sort_begin: nothing else is on the device (a blocking call's first sort) -- set and cleared by the caller
  int msm_groups = 0;                    // window groups of a blocking MSM: 0 = auto, 1 = none, 2..MAX_GROUPS (kg_msm_set_groups)
  // grow-only scratch
  // MSM scalar-side space (sorted digit lists, task tables), two sets used in turn: the sort of MSM i+1 runs on the scalar
  // queue while the accumulation of MSM i still reads the other set
  void* ws_sort[2] = {nullptr, nullptr};
  size_t ws_sort_bytes[2] = {0, 0};
  unsigned sort_seq = 0;
  bool queues_placed = false, sort_events = false;   // capi.cpp place_queues: the service queues were created (and dealt over the compute pipes)
  int placement = 0;                     // 0: not probed; 1 + j: the probe's picture (candidate j shares the main queue's pipe); -1: no clear picture, creation order
  hipStream_t sort_stream = nullptr;     // scalar-side queue (prep_scalars, sort, task bookkeeping); == stream when no overlap is possible
  // A blocking MSM pipelines against itself by WINDOW GROUPS (msm.hip, kg_msm): the scalars are converted once, then the
  // windows are sorted, accumulated and reduced group by group, top windows first -- group g+1 is sorted under group g's
  // accumulation, group g is reduced under group g+1's accumulation, and the host's double-and-add chain starts with the top
  // group's sums while the lower groups are still on the device.  No extra work (index slices add reduction tails).
  static constexpr int MAX_GROUPS = 4;
  hipEvent_t ev_sorted[2][MAX_GROUPS] = {};         // sort of the set's window group complete (recorded on the scalar queue)
  hipStream_t acc_stream[MAX_GROUPS] = {};          // accumulation queues of window groups 1.. (group 0: the main queue), so that the
                                                    // groups' launches share the chip instead of draining one after the other
  void* ws_pb = nullptr;                            // per-call resident form of the bases, shared by the window groups of one MSM
  size_t ws_pb_bytes = 0;
  hipEvent_t ev_pb = nullptr;                       // its conversion complete
  hipEvent_t ev_prep = nullptr;                     // scalar conversion on the main queue complete (the scalar queue's sorts follow it)
  static constexpr int IDLE_EVS = 8;                // readers of one set whose completion the next sort into it waits for
  hipEvent_t ev_ws_idle[2][IDLE_EVS] = {};          // a bucket gather that read the set's level tables is complete (reduction queues)
  int ws_idle_n[2] = {0, 0};
  hipEvent_t ev_bases = nullptr;                    // per-call base conversion on the scalar queue complete
  hipEvent_t ev_order = nullptr;         // stream-order hand-over main -> scalar queue
  unsigned small_glv_off = 0;            // bit per curve: no halved scalars (GLV) for that curve's one-launch MSMs -- set by the prover and by kg_msm_begin around their launches, which are bound by the kernels and not by the host chains GLV halves
  bool inputs_complete = false;          // kg_ctx_set_inputs_complete: MSM inputs are complete when the call is made

  // kg_malloc / kg_free keep released blocks for the next request of the same size class (capi.cpp): the first DMA into a FRESH
  // hipMalloc allocation runs at 1-5 GB/s (its pages are mapped on first touch: 15-28 ms per 32 MiB, measured), a copy into a block
  // that has been used before at 56 GB/s -- and every host of the boundary allocates per call (DeviceBuf::new in the Rust glue,
  // DeviceBuffer in the C++ mirror, Context.upload in Python)
  std::multimap<size_t, void*> pool_free;            // size class -> released blocks
  size_t pool_cached = 0;                            // bytes sitting in pool_free
  void* ws_vec = nullptr;                // kg_r1cs_prod: work list of long rows (grow-only)
  size_t ws_vec_bytes = 0;
  void* ws2 = nullptr;                   // NTT ping-pong buffer
  size_t ws2_bytes = 0;
  void* ws3[2] = {nullptr, nullptr};     // prover polynomial buffers (a, b, c, z, transform scratch), one set per proof ticket
  size_t ws3_bytes[2] = {0, 0};
  static constexpr int RUN_SETS = 8;     // a slow reduction (G2) may overlap all the later accumulations of a proof
  void* ws_run[RUN_SETS] = {};           // MSM base-side scratch (packed bases, partial sums, halving buffers), one set per slot mod RUN_SETS
  size_t ws_run_bytes[RUN_SETS] = {};
  hipStream_t side_stream = nullptr;     // bucket reduction of MSM i overlaps the accumulation of MSM i+1
  hipStream_t side2_stream = nullptr;    // second reduction queue (odd slots): a slow G2 reduction does not hold up the next MSM's
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_acc[RUN_SETS] = {};
  hipEvent_t ev_info[MAX_GROUPS] = {};   // marks the task-count read-back of msm_sort (per window group)
  struct Slot { void* host = nullptr; void* host_dev = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; int W = 0, c = 0, w0 = 0; bool busy = false;
                bool combined = false; };   // w0: first window of the group the slot holds; combined: ONE point per window (the short-input MSM adds a window's bit planes on the device), otherwise c bit-plane sums per window
  static constexpr int NSLOTS = 24;      // 0 kg_msm, 1..4 kg_msm_begin tickets, 6..10 prover job 0, 11..15 prover job 1, 16..23 slices of kg_msm_host / kg_msm_host_scalars, 16..19 window groups / index slices of kg_msm (blocking calls: never at the same time)
  Slot slots[NSLOTS];
  void* ws_small[NSLOTS] = {};           // short-input MSM (msm_small.hip): plane points of a window split over several workgroups, per result slot (grow-only)
  size_t ws_small_bytes[NSLOTS] = {};
  std::unique_ptr<kg::WorkerPool> workers;   // host worker threads (worker_pool.h), started on demand, alive until the context is destroyed
  std::shared_ptr<void> prover_jobs;     // groth16.hip: proofs in flight (kg_groth16_prove_begin / _end)
  size_t ticket_n[4] = {0, 0, 0, 0};
  // kg_msm_begin starts the ticket's host finish (wait for the reduction, 255-step double-and-add, inversion) on a worker
  // thread, so the calling thread goes straight on to enqueue the next MSM; kg_msm_end joins it
  std::future<int> ticket_fut[4];
  uint64_t ticket_out[4][24] = {};
  // table: optional window multiples of the array (kg_bases_precompute): table[w][i] = 2^(table_c * w) * base[i] in resident
  // form, w < table_W -- an MSM against them needs one set of buckets for all windows (merged sort, msm.hip)
  struct Registered { const uint64_t* base; const uint8_t* inf; size_t n; int curve; uint32_t* packed; uint32_t* table = nullptr; int table_c = 0, table_W = 0; bool fmt64 = false, table64 = false; };   // fmt64 / table64: 64-byte points (msm.hip, BaseIO::load_point64)
  std::vector<Registered> registered;    // bases converted once by kg_bases_register     // lengths of the MSMs begun with kg_msm_begin                         // pinned result slots: MSMs in flight whose host finish is pending
  // kg_msm_host: cached device copies of the caller's host arrays (grow-only) and the upload queue
  void* up_buf[3] = {nullptr, nullptr, nullptr};     // bases, scalars, identity flags
  size_t up_bytes[3] = {0, 0, 0};
  hipStream_t up_stream = nullptr;
  static constexpr int UP_SLICES = 8;    // index slices of a host-array MSM (kg_msm_host, kg_msm_host_scalars): result slots 16 .. 23
  hipEvent_t ev_up_s[UP_SLICES] = {}, ev_up_b[UP_SLICES] = {};
  void* fb_tmp = nullptr;                // XYZZ results between the two passes of kg_fixed_base_mul (grow-only)
  size_t fb_tmp_bytes = 0;
  uint32_t* fb_table[3] = {nullptr, nullptr, nullptr};   // kg_fixed_base_mul: 32 x 255 generator multiples d * 2^(8w) * G per curve, resident form
  void* h_pinned = nullptr;              // small pinned staging buffer for results
  void* h_pinned_dev = nullptr;          // its device view (kernels write the sort's result words straight into it)
  size_t h_pinned_bytes = 0;
  std::vector<kg_tw_cache*> tw;          // per-(log_n, inverse) twiddle tables
  bool tw_fresh = false;                 // tables were built on the main queue since the last fork (transform lanes must wait for them)
  // profiling
  bool prof = false;
  struct Phase { const char* name; hipEvent_t e0, e1; };
  std::vector<Phase> phases;
  std::vector<hipEvent_t> event_pool;
  size_t event_next = 0;
  std::atomic<long long> host_finish_us{0};   // summed wall time of the host finishes (double-and-add + inversion) since the last
  std::atomic<int> host_finish_calls{0};      // reset; atomics: a prover's finishes run on up to five worker threads at once
};

namespace kg {

// the context's worker pool (created on first use)
inline WorkerPool& pool(kg_ctx* c) {
  if (!c->workers) c->workers.reset(new WorkerPool(c->tune.pool_max_threads));
  return *c->workers;
}
inline int set_err(kg_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    c->last_error = what;
    if (e != hipSuccess) { c->last_error += ": "; c->last_error += hipGetErrorString(e); }
  }
  if (e != hipSuccess) (void)hipGetLastError();     // the runtime's sticky copy: a later launch check must not report this failure again
  return code;
}
// Host-side C++ failures -- std::bad_alloc, a worker thread that cannot be started (std::system_error) -- never cross the C ABI as exceptions
// (the callers are Rust with panic = "abort", C and ctypes): entry points that allocate host containers or start threads run their body
// through kg_guarded and report a status.  JoinGuard: a std::thread that is joined when its scope unwinds.
template <class Fn>
inline int kg_guarded(kg_ctx* c, Fn&& fn) noexcept {
  try { return fn(); }
  catch (const std::bad_alloc&) { return set_err(c, KG_ERR_OOM, "host allocation failed"); }
  catch (...) { return set_err(c, KG_ERR_HIP, "host runtime failure (a worker thread could not be started?)"); }
}
struct JoinGuard {
  std::thread& t;
  ~JoinGuard() { if (t.joinable()) t.join(); }
};
#define KG_HIP(ctx, call)                                                        \
  do {                                                                           \
    hipError_t e__ = (call);                                                     \
    if (e__ != hipSuccess) return kg::set_err(ctx, e__ == hipErrorOutOfMemory ? KG_ERR_OOM : KG_ERR_HIP, #call, e__); \
  } while (0)
#define KG_TRY(call)            \
  do {                          \
    int s__ = (call);           \
    if (s__ != KG_OK) return s__; \
  } while (0)

// hipMalloc for the library's own work spaces: when the device refuses, the blocks kg_free has kept are released and the request is
// repeated once (the pool must never be what makes a call fail)
hipError_t dev_alloc(kg_ctx* c, void** p, size_t bytes);
void pool_trim(kg_ctx* c);               // releases every kept block
int ensure_ws_sort(kg_ctx* c, int set, size_t bytes);
int make_sort_stream(kg_ctx* c);
hipError_t create_stream(kg_ctx* c, hipStream_t* out, bool service);
int ensure_ws2(kg_ctx* c, size_t bytes);
int ensure_ws_vec(kg_ctx* c, size_t bytes);
int ensure_ws3(kg_ctx* c, int which, size_t bytes);
int ensure_ws_run(kg_ctx* c, int which, size_t bytes);
int ensure_slot(kg_ctx* c, int slot, size_t bytes);
int ensure_pinned(kg_ctx* c, size_t bytes);
int make_side_stream(kg_ctx* c);
int slice_table_window(const kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t cnt);
void sync_all(kg_ctx* c);

// RAII-free phase timer: PhaseScope p(ctx, "name"); ... p.end();
struct PhaseScope {
  kg_ctx* c; int idx = -1; hipStream_t s;
  PhaseScope(kg_ctx* ctx, const char* name, hipStream_t stream = nullptr);
  void end();
};
void prof_reset(kg_ctx* c);

// Wave priority of the "service" kernels (sorts, bookkeeping, reductions, transforms, vector ops).  VALU issue on a SIMD is
// arbitrated by priority, then age: next to a resident, VALU-saturating accumulation (whose waves are older) a young wave at
// the default priority gets only the leftover issue slots and runs ~10x slower (tools/ubench/coexec.hip).  With priority 3
// the service waves -- mostly waiting on memory or LDS -- issue when they are ready and cost the accumulation ~2 %.
// The bucket-reduction kernels (partial sums, gathers, halving levels, tail) take priority 1: above the accumulation, below
// the sorts -- a sort gates the next accumulation while a reduction's result is only needed at the end (measured on one box:
// Groth16 2.86 -> 2.77 ms per proof, MSM 2^20 1.393 -> 1.373 ms per step; priority 2 is level for the prover and 1.5 % worse
// for the MSM).
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef KG_PRIO_LEVEL
#define KG_PRIO_LEVEL 3
#endif
#define KG_SERVICE_PRIO() __builtin_amdgcn_s_setprio(KG_PRIO_LEVEL)
#ifndef KG_REDUCE_LEVEL
#define KG_REDUCE_LEVEL 1
#endif
#define KG_REDUCE_PRIO() __builtin_amdgcn_s_setprio(KG_REDUCE_LEVEL)
#else
#define KG_SERVICE_PRIO() ((void)0)
#define KG_REDUCE_PRIO() ((void)0)
#endif

// ---- device-side layouts ------------------------------------------------------------------------
// A field element in the ABI: 8 x u32 words (4 x u64 LE).  Loaded/stored as two 16-byte vectors.
struct alignas(16) Words8 { uint32_t w[8]; };

__device__ __forceinline__ void load_words(const uint64_t* base, size_t elem, uint32_t w[8]) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * elem;
  uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store_words(uint64_t* base, size_t elem, const uint32_t w[8]) {
  uint4* p = reinterpret_cast<uint4*>(base) + 2 * elem;
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// Raw 9-limb storage of internal values in scratch memory, structure-of-arrays: word k of element i lives at
// base[k * stride + i], so a wave touches 64 consecutive dwords per limb (fully coalesced).
template <class F> struct RawIO;
template <class P> struct RawIO<Fp<P>> {
  static constexpr int NW = 9;
  static __device__ __forceinline__ Fp<P> load(const uint32_t* base, size_t stride, size_t i) {
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = base[(size_t)k * stride + i];
    return r;
  }
  static __device__ __forceinline__ void store(uint32_t* base, size_t stride, size_t i, const Fp<P>& a) {
#pragma unroll
    for (int k = 0; k < 9; ++k) base[(size_t)k * stride + i] = a.l[k];
  }
};
template <class F> struct RawIO<Fp2<F>> {
  static constexpr int NW = 18;
  static __device__ __forceinline__ Fp2<F> load(const uint32_t* base, size_t stride, size_t i) {
    return {RawIO<F>::load(base, stride, i), RawIO<F>::load(base + 9 * stride, stride, i)};
  }
  static __device__ __forceinline__ void store(uint32_t* base, size_t stride, size_t i, const Fp2<F>& a) {
    RawIO<F>::store(base, stride, i, a.c0);
    RawIO<F>::store(base + 9 * stride, stride, i, a.c1);
  }
};
template <class F>
struct PointIO {
  static constexpr int NW = 4 * RawIO<F>::NW;
  static __device__ __forceinline__ XYZZ<F> load(const uint32_t* base, size_t stride, size_t i) {
    constexpr int E = RawIO<F>::NW;
    return {RawIO<F>::load(base, stride, i), RawIO<F>::load(base + (size_t)E * stride, stride, i),
            RawIO<F>::load(base + (size_t)2 * E * stride, stride, i), RawIO<F>::load(base + (size_t)3 * E * stride, stride, i)};
  }
  static __device__ __forceinline__ void store(uint32_t* base, size_t stride, size_t i, const XYZZ<F>& p) {
    constexpr int E = RawIO<F>::NW;
    RawIO<F>::store(base, stride, i, p.x);
    RawIO<F>::store(base + (size_t)E * stride, stride, i, p.y);
    RawIO<F>::store(base + (size_t)2 * E * stride, stride, i, p.zz);
    RawIO<F>::store(base + (size_t)3 * E * stride, stride, i, p.zzz);
  }
};

// ABI <-> internal for whole base-field elements (Fq, Fr: 8 words; Fq2: 16 words)
template <class F> struct RefIO;
template <class P> struct RefIO<Fp<P>> {
  static constexpr int W64 = 4;
  static __device__ __forceinline__ Fp<P> load(const uint64_t* p) {
    uint32_t w[8];
    load_words(p, 0, w);
    return from_ref<P>(w);
  }
};
template <class F> struct RefIO<Fp2<F>> {
  static constexpr int W64 = 8;
  static __device__ __forceinline__ Fp2<F> load(const uint64_t* p) { return {RefIO<F>::load(p), RefIO<F>::load(p + 4)}; }
};


// msm.hip: scalar-side / base-side / host halves of an MSM (used by kg_msm and by the Groth16 prover).
// MsmSorted is the scalar-side state: digits sorted into per-bucket lists plus the task decomposition.  It depends
// only on the scalars, so several base arrays (the CRS vectors a, b_g1, b_g2, l of a Groth16 proof all meet the same
// witness) share one.  It lives in one of the two scalar-side spaces of the context until the sort after next.
struct MsmSorted {
  size_t n = 0;
  int c = 0, W = 0, B = 0;
  uint32_t T = 0, T2 = 16, T_top = 0, nhot = 0;   // T_top: task length of the unsigned top window (window top_w of this object, -1: none); nhot: hot buckets
  int top_w = -1;
  uint32_t* hot_list = nullptr;                   // buckets with more than GATHER_SUM_MAX tasks (index w * B + b)
  size_t npts = 0, part_cap = 0;
  uint32_t ntasks = 0, max_cnt = 0;
  uint32_t *sorted = nullptr, *bsize = nullptr, *bstart = nullptr, *lcnt = nullptr, *lrel = nullptr, *lbase = nullptr;
  uint32_t *task_bkt = nullptr, *task_id = nullptr;
  int set = 0;                // which scalar-side space it lives in
  hipEvent_t ready = nullptr; // recorded on the scalar queue after the last sort kernel
  // window group (kg_msm): this object covers windows [w0, w0 + W) of an MSM with `windows_total` windows; its pointers are
  // offset to the group, so the base side treats it like an MSM of W windows.  group = index of its read-back words / events.
  int w0 = 0, group = 0;
  hipStream_t acc_stream = nullptr;   // queue of the group's accumulation (nullptr: the context's main queue)
  bool reduce_inline = false;         // the bucket reduction follows the accumulation on the same queue
  bool tail_alone = false;            // nothing runs beside this reduction's tail (a blocking call's last reduction): the lane-cooperative tail kernel
  hipStream_t sorted_on = nullptr;    // queue the sort was enqueued on
  // merged sort (bases with window tables): the digits of all windows share ONE set of B buckets; W = 1 above, windows = the
  // real window count, an entry's index field is (window << merged_shift) | scalar index
  int merged_shift = 0, windows = 0;
};
// vec.hip: CSR matrix-vector product on a given queue; scratch = (m + 16) words (kg_ctx::ws_vec holds three such regions)
int r1cs_prod_enqueue(kg_ctx* c, hipStream_t st, int field, const uint64_t* row_ptr, const uint64_t* col, const uint64_t* val, size_t m,
                      const uint64_t* z, uint64_t* out, uint32_t* scratch);
// ntt.hip
int ntt_prepare(kg_ctx* ctx, uint32_t log_n, int inverse);
int ntt_enqueue(kg_ctx* ctx, hipStream_t st, uint64_t* tmp, uint64_t* d_data, uint32_t log_n, int inverse, int coset);
// ordered: the caller has already put the scalar queue (ctx->sort_stream, see scalar_queue()) behind the producer of
// d_scalars; otherwise msm_sort orders it after everything enqueued on the main queue so far (stream semantics), or not
// at all when the context's inputs are declared complete (kg_ctx_set_inputs_complete)
// merged_c: 0, or the window width of the bases' tables (merged_window(n)): sort all windows into one set of buckets;
// lane_mult: how many accumulation lanes each task of this sort will occupy (fused arrays, G2's lane pairs) -- sizes the tasks
// wait_info = false: return once the sort is enqueued; msm_sort_wait(ctx, S) must follow before S is used (and before another sort)
// info_idx: which pair of read-back words / event the sort uses (< kg_ctx::MAX_GROUPS): two sorts whose read-backs are both
// pending (the prover's witness sort and h's) need different ones
int msm_sort(kg_ctx* ctx, int scalar_field, const uint64_t* d_scalars, size_t n, MsmSorted* S, bool ordered = false, int merged_c = 0, int lane_mult = 1,
             bool wait_info = true, int info_idx = 0);
int msm_sort_wait(kg_ctx* ctx, MsmSorted* S);
// The same sort in two steps, for window groups: msm_sort_begin converts the scalars and lays the space out for `ngroups`
// groups of gw[0], gw[1], ... windows counted from the TOP window down (ngroups = 1: all windows; groups need the two-pass
// sort, see msm_group_plan); msm_sort_group enqueues group g's sort (any order, each once) and msm_sort_wait reads its two
// result words back.
struct MsmSortPlan {
  size_t n = 0, chunk_len = 0, nv = 0;
  int c = 0, W = 0, B = 0, Wb = 0, G = 0, nch = 0, maxseg = 0, mshift = 0, set = 0, ngroups = 1, info_base = 0, fb = 7;
  bool alone = false;                    // the first group's sort has the chip to itself (kg_ctx::sort_alone, or a call split into window groups)
  bool merged = false, two_pass = false;
  uint32_t T = 0, T_top = 0;
  int gw0[kg_ctx::MAX_GROUPS] = {}, gW[kg_ctx::MAX_GROUPS] = {};
  char* ws = nullptr;
  size_t o_kt = 0, o_cnt = 0, o_bsize = 0, o_bstart = 0, o_tmp = 0, o_gsize = 0, o_gstart = 0, o_segbase = 0, o_segcnt = 0, o_segoff = 0, o_sorted = 0,
         o_lcnt = 0, o_lrel = 0, o_rowtot = 0, o_bpart = 0, o_woff = 0, o_gsize_m = 0, o_gstart_m = 0, o_segbase_m = 0;
  size_t o_lbase[kg_ctx::MAX_GROUPS] = {}, o_misc[kg_ctx::MAX_GROUPS] = {}, o_lenh[kg_ctx::MAX_GROUPS] = {}, o_tbkt[kg_ctx::MAX_GROUPS] = {},
         o_tid[kg_ctx::MAX_GROUPS] = {}, o_hot[kg_ctx::MAX_GROUPS] = {}, part_cap[kg_ctx::MAX_GROUPS] = {};
};
int msm_sort_begin(kg_ctx* ctx, int scalar_field, const uint64_t* d_scalars, size_t n, MsmSortPlan* P, bool ordered, int merged_c, int lane_mult,
                   int ngroups = 1, const int* gw = nullptr, bool on_main = false);
int msm_sort_group(kg_ctx* ctx, const MsmSortPlan& P, int g, MsmSorted* S, bool on_main = false);
// window groups a blocking n-pair MSM is cut into (0 = not offered: one-pass sort, merged sort); gw[g] = windows of group g, top first
int msm_group_plan(const kg_ctx* ctx, size_t n, int* gw);
int merged_window(const kg_ctx* ctx, size_t n);       // window width of the merged form for an n-scalar MSM, 0 = not offered at this length
// does this registered array carry a window table that serves an n-scalar merged MSM?  (d_inf must be the registered flag array)
bool has_window_table(const kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t nbases, size_t msm_len);
int scalar_queue(kg_ctx* ctx, hipStream_t* out);     // the scalar-side queue, created on first use
// several base arrays against one scalar sort, accumulated by one launch (at most 3; result slots in distinct run-space sets)
// bases_complete: the base array is complete in device memory when the call is made (no ordering against the main queue)
// packed: the caller holds the array's resident form already (kg_msm converts once for all its window groups); the job's
// queue must be ordered behind its producer by the caller
struct MsmRunJob { const uint64_t* d_bases; const uint8_t* d_inf; size_t nbases; uint32_t idx_off; int slot; bool bases_complete = false;
                   const uint32_t* packed = nullptr; bool packed64 = false; };
int msm_run_multi(kg_ctx* ctx, const MsmSorted& S, int curve, const MsmRunJob* jobs, int njobs);
int msm_run(kg_ctx* ctx, const MsmSorted& S, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t nbases, uint32_t idx_off, int slot);
int msm_finish(kg_ctx* ctx, int curve, int slot, uint64_t* out_xyz);
// msm_small.hip: the short-input MSM (one or two launches, no sort, no read-back).  msm_small_plan: does it take an n-pair MSM of this
// context, and with which window width c and bucket range 2^r per workgroup; msm_small_enqueue: the launches on queue st, the window
// sums into the slot (msm_finish is the host half, as for the long pipeline)
bool msm_small_plan(const kg_ctx* ctx, int curve, size_t n, int* c, int* r);
bool msm_small_glv(const kg_ctx* ctx, int curve, size_t n);   // does an n-pair short MSM split its scalars into two 127-bit halves (msm_digits.h)?
bool msm_small_kt(const kg_ctx* ctx, size_t n);      // does an n-pair short MSM convert its scalars once, by a launch of its own (the KT form)?
int msm_small_enqueue(kg_ctx* ctx, hipStream_t st, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, int slot,
                      int c, int r);
// the same over the slots of an MSM's window groups, top group first: one double-and-add chain, each slot awaited when the
// chain reaches its windows
int msm_finish_groups(kg_ctx* ctx, int curve, const int* slots, int nslots, uint64_t* out_xyz);
void msm_identity(int curve, uint64_t* out_xyz);

}  // namespace kg
