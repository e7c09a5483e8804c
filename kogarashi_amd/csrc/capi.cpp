// capi.cpp -- context management and memory plumbing of the C ABI (include/kogarashi_amd.h).
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace kg;

namespace kg {
// Every queue of the context drained: before a work space is released, and around profiling resets.  (The sort space is
// read on the main queue only today; draining all of them keeps a future side-queue reader safe.)
void sync_all(kg_ctx* c) {
  hipStreamSynchronize(c->stream);
  if (c->own_stream && c->own_stream != c->stream) hipStreamSynchronize(c->own_stream);
  for (hipStream_t s : {c->sort_stream, c->side_stream, c->side2_stream, c->up_stream})
    if (s) hipStreamSynchronize(s);
  for (hipStream_t s : c->acc_stream)
    if (s) hipStreamSynchronize(s);
}
// ---- the block pool behind kg_malloc / kg_free ---------------------------------------------------------------------------------
static size_t pool_class(size_t bytes) {
  if (bytes <= 4096) return 4096;
  if (bytes <= ((size_t)1 << 20)) { size_t c = 8192; while (c < bytes) c <<= 1; return c; }      // powers of two up to 1 MiB
  return (bytes + (((size_t)1 << 20) - 1)) & ~(((size_t)1 << 20) - 1);                          // whole MiB above
}
// Process-wide bookkeeping of the pool: the live contexts (a device that runs out of memory gets the kept blocks of EVERY context on it
// back, not only the caller's) and the blocks kg_malloc has handed out, by address -> (owner, size class): kg_free through another
// context than the allocating one still finds the block (it is released, not kept), and a stale entry cannot outlive its address.
static std::mutex g_pool_mu;
static std::vector<kg_ctx*> g_ctxs;
static std::unordered_map<void*, std::pair<kg_ctx*, size_t>> g_live;
static void pool_trim_locked(kg_ctx* c) {                  // g_pool_mu held
  for (auto& kv : c->pool_free) hipFree(kv.second);
  c->pool_free.clear();
  c->pool_cached = 0;
}
void pool_trim(kg_ctx* c) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  pool_trim_locked(c);
}
// the kept blocks of every context of the device (a kept block is idle by construction: its context was drained when it was kept)
static size_t pool_trim_device(int device) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  size_t freed = 0;
  for (kg_ctx* o : g_ctxs)
    if (o->device == device) { freed += o->pool_cached; pool_trim_locked(o); }
  return freed;
}
hipError_t dev_alloc(kg_ctx* c, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    sync_all(c);
    if (pool_trim_device(c->device)) e = hipMalloc(p, bytes);
  }
  return e;
}
int ensure_ws_sort(kg_ctx* c, int set, size_t bytes) {
  if (bytes <= c->ws_sort_bytes[set]) return KG_OK;
  if (c->ws_sort[set]) { sync_all(c); hipFree(c->ws_sort[set]); c->ws_sort[set] = nullptr; c->ws_sort_bytes[set] = 0; }
  size_t want = bytes + bytes / 8;
  hipError_t e = dev_alloc(c, &c->ws_sort[set], want);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "workspace allocation", e);
  c->ws_sort_bytes[set] = want;
  return KG_OK;
}
// The context's queues (main, scalar, two reduction queues, an upload queue for kg_msm_host) should each own a hardware queue:
// the runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with whatever else the process
// creates), and two streams on one hardware queue run in submission order -- measured on the prover: 2.83 ms per proof with a
// queue each against 3.28 ms at the default.  The variable is read when the HIP runtime initialises.  The library never
// touches the environment on its own: a host that wants the setting calls kg_init() (or exports the variable) before
// anything initialises HIP; kg_hw_queue_setting() reports what the process has.
// Queues of the context.  (CU-masked queues -- hipExtStreamCreateWithCUMask, a compute / service partition -- and queue
// priorities were measured and dropped: tools/ubench/cumask_probe.hip; what makes concurrent queues
// work is the wave priority of the service kernels, KG_SERVICE_PRIO.)
// KG_STREAM_PAD=n0,n1,...: experiment -- n_i never-used streams are created in front of the context's i-th queue (creation
// order: main, scalar, reduction 1, reduction 2, ...), which shifts the queues over the hardware queues / pipes
static int stream_pad(int idx) {
  const char* e = tuning().stream_pad.empty() ? nullptr : tuning().stream_pad.c_str();
  if (!e) return 0;
  for (int i = 0; i < idx && e; ++i) { e = strchr(e, ','); if (e) ++e; }
  return e ? atoi(e) : 0;
}
hipError_t create_stream(kg_ctx* c, hipStream_t* out, bool service) {
  (void)service;
  static std::atomic<int> created{0};
  const int idx = created++;
  for (int i = stream_pad(idx); i > 0; --i) { hipStream_t dummy; if (hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking) == hipSuccess) { /* leaked on purpose: experiment only */ } }
  (void)c;
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
// ---- queue placement ---------------------------------------------------------------------------------------------------
// Measured (round 4, tools/dbg/pad_sweep.sh): the runtime deals a process's streams over the hardware queues in creation order,
// and hardware queue k is served by compute pipe k mod 4.  A pipe that is launching the workgroups of a big dispatch -- an
// accumulation of several resident rounds keeps it busy until its LAST round is placed -- serves no other queue meanwhile, so a
// queue that shares the main queue's pipe starts its sort or reduction ~0.7 ms late.  Which of the library's queues share a
// pipe used to be an accident of how many streams the host had created first: with one never-used stream in front of the scalar
// queue the 2^20 MSM step went 1.37 -> 1.50 ms and the Groth16 proof 2.93 -> 3.18 ms (two in flight), with three 1.65 / 3.25.
// So the service queues are PLACED: eight candidate streams are created, one probe finds the candidates that share the main
// queue's pipe (a kernel of many short workgroups on the main queue, a one-wave kernel behind an event on every candidate: a
// candidate on the same pipe finishes with the long dispatch, the others at once), and the scalar queue and the two reduction
// queues take one candidate from each of the three other pipes; the second accumulation queue of a blocking MSM's window
// groups takes the main queue's pipe on purpose (its launch then follows the first group's last placed workgroup).
// KG_QUEUE_PLACEMENT=0: creation order as before (experiments).
__global__ void __launch_bounds__(64) k_probe_busy(unsigned long long ticks) {
  extern __shared__ uint32_t probe_pad[];
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (ticks == ~0ull) probe_pad[threadIdx.x] = 1;      // keeps the LDS allocation (64 KiB: two workgroups per CU, wave slots stay free)
}
__global__ void __launch_bounds__(64) k_probe_nop(uint32_t* p) { if (p) *p = 1; }

static int place_queues(kg_ctx* c) {
  if (c->queues_placed) return KG_OK;
  constexpr int NC = 8;
  hipStream_t cand[NC] = {};
  hipError_t e = hipSuccess;
  for (int j = 0; j < NC && e == hipSuccess; ++j) e = create_stream(c, &cand[j], true);
  if (e != hipSuccess) {                                // nothing is kept: the candidates made so far are released, a later call tries again
    for (int j = 0; j < NC; ++j) if (cand[j]) hipStreamDestroy(cand[j]);
    return set_err(c, KG_ERR_HIP, "queue creation", e);
  }
  int cls[NC];
  for (int j = 0; j < NC; ++j) cls[j] = -1;
  const bool enabled = c->tune.queue_placement != 0;
  bool ok = enabled;
  if (ok) {
    hipEvent_t ev_s = nullptr, ev_a = nullptr, ev_b[NC] = {};
    ok = hipEventCreate(&ev_s) == hipSuccess && hipEventCreate(&ev_a) == hipSuccess;
    for (int j = 0; j < NC && ok; ++j) ok = hipEventCreate(&ev_b[j]) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)k_probe_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess;
    bool shares[NC] = {};
    int j0 = -1;
    // the first pass warms the queues up, the second one counts; a pass whose picture is not the expected one (the host thread lost the
    // CPU between the launches: the pods' cores are shared) is repeated, three more times at most (~150 us each)
    for (int rep = 0; rep < 5 && ok && j0 < 0; ++rep) {
      ok = hipEventRecord(ev_s, c->stream) == hipSuccess;
      hipLaunchKernelGGL(k_probe_busy, dim3(6144), dim3(64), 64 * 1024, c->stream, 1000ull);      // 12 rounds of 512 resident workgroups x 10 us
      ok = ok && hipEventRecord(ev_a, c->stream) == hipSuccess;
      for (int j = 0; j < NC && ok; ++j) {
        ok = hipStreamWaitEvent(cand[j], ev_s, 0) == hipSuccess;
        hipLaunchKernelGGL(k_probe_nop, dim3(1), dim3(64), 0, cand[j], (uint32_t*)nullptr);
        ok = ok && hipEventRecord(ev_b[j], cand[j]) == hipSuccess;
      }
      ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
      for (int j = 0; j < NC && ok; ++j) ok = hipStreamSynchronize(cand[j]) == hipSuccess;
      float ta = 0.f;
      ok = ok && hipEventElapsedTime(&ta, ev_s, ev_a) == hipSuccess;
      for (int j = 0; j < NC && ok; ++j) {
        float tb = 0.f;
        ok = hipEventElapsedTime(&tb, ev_s, ev_b[j]) == hipSuccess;
        shares[j] = tb > 0.5f * ta;
      }
      if (rep == 0 || !ok) continue;
      // the expected picture: exactly the candidates j, j + 4 share the main queue's pipe
      for (int j = 0; j < 4; ++j) if (shares[j]) { j0 = j; break; }
      for (int j = 0; j < NC && j0 >= 0; ++j) if (shares[j] != ((j - j0) % 4 == 0)) j0 = -1;
    }
    (void)hipGetLastError();
    if (ev_s) hipEventDestroy(ev_s);
    if (ev_a) hipEventDestroy(ev_a);
    for (int j = 0; j < NC; ++j) if (ev_b[j]) hipEventDestroy(ev_b[j]);
    ok = ok && j0 >= 0;
    if (ok) for (int j = 0; j < NC; ++j) cls[j] = ((j - j0) % 4 + 4) % 4;      // 0: the main queue's pipe
    c->placement = ok ? 1 + j0 : -1;
  } else c->placement = 0;
  auto take = [&](int want) -> hipStream_t {                // first unused candidate of the class (any class when the probe gave no picture)
    for (int j = 0; j < NC; ++j)
      if (cand[j] && (cls[j] == want || !ok)) { hipStream_t t = cand[j]; cand[j] = nullptr; return t; }
    for (int j = 0; j < NC; ++j) if (cand[j]) { hipStream_t t = cand[j]; cand[j] = nullptr; return t; }
    return nullptr;
  };
  c->sort_stream = take(1);
  c->side_stream = take(2);
  c->side2_stream = take(3);
  c->acc_stream[1] = take(0);
  c->up_stream = take(3);                                 // kg_msm_host's upload queue: copies only
  for (int j = 0; j < NC; ++j) if (cand[j]) hipStreamDestroy(cand[j]);
  c->queues_placed = true;                                // only now: every queue the context uses exists
  return KG_OK;
}
int make_sort_stream(kg_ctx* c) {
  if (c->sort_events) return KG_OK;
  KG_TRY(place_queues(c));
  hipError_t e = c->sort_stream ? hipSuccess : create_stream(c, &c->sort_stream, true);
  if (e != hipSuccess) return set_err(c, KG_ERR_HIP, "scalar-queue creation", e);
  for (int i = 0; i < 2; ++i) {
    for (int g = 0; g < kg_ctx::MAX_GROUPS; ++g)
      if ((e = hipEventCreateWithFlags(&c->ev_sorted[i][g], hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
    for (int j = 0; j < kg_ctx::IDLE_EVS; ++j)
      if ((e = hipEventCreateWithFlags(&c->ev_ws_idle[i][j], hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
  }
  if ((e = hipEventCreateWithFlags(&c->ev_bases, hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
  if ((e = hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
  if ((e = hipEventCreateWithFlags(&c->ev_pb, hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
  if ((e = hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
  for (int g = 0; g < kg_ctx::MAX_GROUPS; ++g)
    if ((e = hipEventCreateWithFlags(&c->ev_info[g], hipEventDisableTiming)) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation", e);
  c->sort_events = true;
  return KG_OK;
}
int ensure_ws_vec(kg_ctx* c, size_t bytes) {
  if (bytes <= c->ws_vec_bytes) return KG_OK;
  if (c->ws_vec) { sync_all(c); hipFree(c->ws_vec); c->ws_vec = nullptr; c->ws_vec_bytes = 0; }
  hipError_t e = dev_alloc(c, &c->ws_vec, bytes);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "workspace allocation", e);
  c->ws_vec_bytes = bytes;
  return KG_OK;
}
int ensure_ws2(kg_ctx* c, size_t bytes) {
  if (bytes <= c->ws2_bytes) return KG_OK;
  if (c->ws2) { sync_all(c); hipFree(c->ws2); c->ws2 = nullptr; c->ws2_bytes = 0; }
  hipError_t e = dev_alloc(c, &c->ws2, bytes);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "ntt buffer allocation", e);
  c->ws2_bytes = bytes;
  return KG_OK;
}
int ensure_ws3(kg_ctx* c, int which, size_t bytes) {
  if (bytes <= c->ws3_bytes[which]) return KG_OK;
  if (c->ws3[which]) { sync_all(c); hipFree(c->ws3[which]); c->ws3[which] = nullptr; c->ws3_bytes[which] = 0; }
  hipError_t e = dev_alloc(c, &c->ws3[which], bytes);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "prover buffer allocation", e);
  c->ws3_bytes[which] = bytes;
  return KG_OK;
}
int ensure_ws_run(kg_ctx* c, int which, size_t bytes) {
  if (bytes <= c->ws_run_bytes[which]) return KG_OK;
  if (c->ws_run[which]) {
    sync_all(c);
    hipFree(c->ws_run[which]);
    c->ws_run[which] = nullptr; c->ws_run_bytes[which] = 0;
  }
  size_t want = bytes + bytes / 8;
  hipError_t e = dev_alloc(c, &c->ws_run[which], want);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "msm run-space allocation", e);
  c->ws_run_bytes[which] = want;
  return KG_OK;
}
int ensure_slot(kg_ctx* c, int slot, size_t bytes) {
  if (slot < 0 || slot >= kg_ctx::NSLOTS) return set_err(c, KG_ERR_BAD_ARG, "bad result slot");
  kg_ctx::Slot& s = c->slots[slot];
  if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return set_err(c, KG_ERR_HIP, "event creation");
  if (bytes <= s.bytes) return KG_OK;
  if (s.host) hipHostFree(s.host);
  s.host = nullptr; s.host_dev = nullptr; s.bytes = 0;
  hipError_t e = hipHostMalloc(&s.host, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "pinned slot allocation", e);
  if ((e = hipHostGetDevicePointer(&s.host_dev, s.host, 0)) != hipSuccess) return set_err(c, KG_ERR_HIP, "pinned slot device view", e);
  s.bytes = bytes;
  return KG_OK;
}
// The side stream carries the latency-bound bucket reductions of MSM i under the sort / accumulation of MSM i+1.
// (A lowest-priority stream was measured and made no difference: the two queues do not compete for issue slots.)
int make_side_stream(kg_ctx* c) {
  // (stream priorities were measured, high and low, for the prover and the MSM pipeline: no gain either way)
  if (c->side_stream && c->side2_stream) return KG_OK;
  KG_TRY(place_queues(c));
  hipError_t e = c->side_stream ? hipSuccess : create_stream(c, &c->side_stream, true);
  if (e != hipSuccess) return set_err(c, KG_ERR_HIP, "side stream creation", e);
  e = c->side2_stream ? hipSuccess : create_stream(c, &c->side2_stream, true);
  if (e != hipSuccess) return set_err(c, KG_ERR_HIP, "side stream creation", e);
  return KG_OK;
}
int ensure_pinned(kg_ctx* c, size_t bytes) {
  if (bytes <= c->h_pinned_bytes) return KG_OK;
  if (c->h_pinned) hipHostFree(c->h_pinned);
  c->h_pinned = nullptr; c->h_pinned_bytes = 0;
  hipError_t e = hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return set_err(c, KG_ERR_OOM, "pinned staging allocation", e);
  if ((e = hipHostGetDevicePointer(&c->h_pinned_dev, c->h_pinned, 0)) != hipSuccess) return set_err(c, KG_ERR_HIP, "pinned staging device view", e);
  c->h_pinned_bytes = bytes;
  return KG_OK;
}

static hipEvent_t next_event(kg_ctx* c) {
  if (c->event_next == c->event_pool.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    c->event_pool.push_back(e);
  }
  return c->event_pool[c->event_next++];
}
void prof_reset(kg_ctx* c) { c->phases.clear(); c->event_next = 0; c->host_finish_us = 0; c->host_finish_calls = 0; }
PhaseScope::PhaseScope(kg_ctx* ctx, const char* name, hipStream_t stream) : c(ctx), s(stream ? stream : ctx->stream) {
  if (!c->prof) return;
  kg_ctx::Phase p{name, next_event(c), next_event(c)};
  hipEventRecord(p.e0, s);
  idx = (int)c->phases.size();
  c->phases.push_back(p);
}
void PhaseScope::end() {
  if (idx >= 0) hipEventRecord(c->phases[idx].e1, s);
  idx = -1;
}
void tw_cache_free(kg_ctx* c);   // ntt.hip
}  // namespace kg

extern "C" {

int kg_init(void) {
  // setenv is not thread-safe against concurrent getenv: call this from the host's start-up path, before threads exist
  if (getenv("GPU_MAX_HW_QUEUES")) return 0;
  return setenv("GPU_MAX_HW_QUEUES", "16", 0) == 0 ? 1 : 0;
}
int kg_hw_queue_setting(void) {
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  return e ? atoi(e) : 0;
}
int kg_version(void) { return 6; }      // the round the ABI was last extended in (6: kg_msm_set_small; 5: kg_msm_host_scalars, kg_commit_host_scalars, kg_tuning_describe; 4: kg_msm_set_groups; 3: kg_init, kg_hw_queue_setting, kg_groth16_prove_sharded)

int kg_experiments_built(void) {
#ifdef KG_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

int kg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* kg_strerror(int s) {
  switch (s) {
    case KG_OK: return "ok";
    case KG_ERR_NO_DEVICE: return "no HIP device";
    case KG_ERR_BAD_ARG: return "bad argument";
    case KG_ERR_OOM: return "out of device memory";
    case KG_ERR_HIP: return "HIP runtime error";
    case KG_ERR_UNSUPPORTED: return "unsupported";
    case KG_ERR_CRS: return "CRS delta is the identity (ProverSubVersionCrsAttack)";
    case KG_ERR_INVERSION: return "a toxic scalar has no inverse (ProverInversionFailed)";
    default: return "unknown status";
  }
}

int kg_ctx_create(int device, kg_ctx** out) {
  return kg::kg_guarded((kg_ctx*)nullptr, [&]() -> int {
  if (!out) return KG_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return KG_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return KG_ERR_BAD_ARG;
  if (hipSetDevice(device) != hipSuccess) return KG_ERR_NO_DEVICE;
  kg_ctx* c = new kg_ctx();
  c->device = device;
  c->tune = tuning();
  if (create_stream(c, &c->own_stream, false) != hipSuccess) { delete c; return KG_ERR_HIP; }
  c->stream = c->own_stream;
  { std::lock_guard<std::mutex> lk(g_pool_mu); g_ctxs.push_back(c); }
  *out = c;
  return KG_OK;
  });
}

void kg_ctx_destroy(kg_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  c->prover_jobs.reset();                             // joins the worker threads of proofs still in flight
  for (auto& f : c->ticket_fut) if (f.valid()) f.wait();      // and of MSM tickets begun and never ended
  c->workers.reset();                                 // the worker threads are idle now: joined here, before what they work on is released
  hipStreamSynchronize(c->stream);
  tw_cache_free(c);
  for (int i = 0; i < 2; ++i) {
    if (c->ws_sort[i]) hipFree(c->ws_sort[i]);
    for (int g = 0; g < kg_ctx::MAX_GROUPS; ++g) if (c->ev_sorted[i][g]) hipEventDestroy(c->ev_sorted[i][g]);
    for (int j = 0; j < kg_ctx::IDLE_EVS; ++j) if (c->ev_ws_idle[i][j]) hipEventDestroy(c->ev_ws_idle[i][j]);
  }
  if (c->ev_bases) hipEventDestroy(c->ev_bases);
  if (c->ev_order) hipEventDestroy(c->ev_order);
  if (c->sort_stream) { hipStreamSynchronize(c->sort_stream); hipStreamDestroy(c->sort_stream); }
  if (c->ws2) hipFree(c->ws2);
  if (c->ws_vec) hipFree(c->ws_vec);
  for (int i = 0; i < 2; ++i) if (c->ws3[i]) hipFree(c->ws3[i]);
  if (c->side_stream) hipStreamSynchronize(c->side_stream);
  for (int i = 0; i < kg_ctx::RUN_SETS; ++i) { if (c->ws_run[i]) hipFree(c->ws_run[i]); if (c->ev_acc[i]) hipEventDestroy(c->ev_acc[i]); }
  for (int i = 0; i < kg_ctx::NSLOTS; ++i) if (c->ws_small[i]) hipFree(c->ws_small[i]);
  if (c->side_stream) hipStreamDestroy(c->side_stream);
  if (c->side2_stream) { hipStreamSynchronize(c->side2_stream); hipStreamDestroy(c->side2_stream); }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  for (int i = 0; i < 3; ++i) if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
  for (int g = 0; g < kg_ctx::MAX_GROUPS; ++g) {
    if (c->ev_info[g]) hipEventDestroy(c->ev_info[g]);
    if (c->acc_stream[g]) { hipStreamSynchronize(c->acc_stream[g]); hipStreamDestroy(c->acc_stream[g]); }
  }
  if (c->ev_pb) hipEventDestroy(c->ev_pb);
  if (c->ev_prep) hipEventDestroy(c->ev_prep);
  {                                                      // (blocks the host still holds stay allocated, as before the pool: they are the host's -- released through any other context, or with the process)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    pool_trim_locked(c);
    for (auto it = g_ctxs.begin(); it != g_ctxs.end(); ++it) if (*it == c) { g_ctxs.erase(it); break; }
    for (auto& kv : g_live) if (kv.second.first == c) kv.second.first = nullptr;
  }
  if (c->ws_pb) hipFree(c->ws_pb);
  for (auto& sl : c->slots) { if (sl.host) hipHostFree(sl.host); if (sl.done) hipEventDestroy(sl.done); }
  if (c->h_pinned) hipHostFree(c->h_pinned);
  for (void* b : c->up_buf) if (b) hipFree(b);
  for (uint32_t* t : c->fb_table) if (t) hipFree(t);
  if (c->fb_tmp) hipFree(c->fb_tmp);
  if (c->up_stream) { hipStreamSynchronize(c->up_stream); hipStreamDestroy(c->up_stream); }
  for (int i = 0; i < kg_ctx::UP_SLICES; ++i) { if (c->ev_up_s[i]) hipEventDestroy(c->ev_up_s[i]); if (c->ev_up_b[i]) hipEventDestroy(c->ev_up_b[i]); }
  for (auto& r : c->registered) { hipFree(r.packed); if (r.table) hipFree(r.table); }
  for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}

const char* kg_last_error(kg_ctx* c) { return c ? c->last_error.c_str() : "null context"; }

int kg_ctx_set_stream(kg_ctx* c, void* s) {
  if (!c) return KG_ERR_BAD_ARG;
  hipStream_t next = s ? (hipStream_t)s : c->own_stream;
  if (next != c->stream) {
    // work queued on the outgoing stream may still use the context's work spaces: the new stream has no ordering
    // against it, so drain it (a stream switch is a set-up step, not a hot-path call)
    KG_HIP(c, hipSetDevice(c->device));
    KG_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = next;
  }
  return KG_OK;
}
int kg_ctx_sync(kg_ctx* c) {
  if (!c) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  KG_HIP(c, hipStreamSynchronize(c->stream));
  for (hipStream_t s : {c->sort_stream, c->side_stream, c->side2_stream})
    if (s) KG_HIP(c, hipStreamSynchronize(s));
  for (hipStream_t s : c->acc_stream)
    if (s) KG_HIP(c, hipStreamSynchronize(s));
  return KG_OK;
}
int kg_ctx_queue_placement2(kg_ctx* c, int* out_placement) {
  if (!c || !out_placement) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  KG_TRY(make_sort_stream(c));
  // kg_ctx::placement: 0 probe off, -1 no clear picture, 1 + j probed -> the ABI's non-negative codes
  *out_placement = c->placement == 0 ? 0 : (c->placement < 0 ? 1 : 1 + c->placement);
  return KG_OK;
}
int kg_ctx_queue_placement(kg_ctx* c) {                 // the version-4 entry, version-4 meaning: the placement is the return value
  if (!c) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  KG_TRY(make_sort_stream(c));
  return c->placement;
}
int kg_ctx_set_inputs_complete(kg_ctx* c, int on) {
  if (!c) return KG_ERR_BAD_ARG;
  c->inputs_complete = on != 0;
  return KG_OK;
}
int kg_malloc(kg_ctx* c, size_t bytes, void** p) {
  return kg::kg_guarded(c, [&]() -> int {
  if (!c || !p) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  const size_t cls = pool_class(bytes ? bytes : 1);
  if (cls < bytes) return set_err(c, KG_ERR_OOM, "kg_malloc: size overflow");
  *p = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = c->pool_free.find(cls);
    if (it != c->pool_free.end()) {                        // a block of this class that has been used before: its pages are mapped
      *p = it->second;
      c->pool_free.erase(it);
      c->pool_cached -= cls;
    }
  }
  if (!*p) {
    const hipError_t e = dev_alloc(c, p, cls);
    if (e != hipSuccess) { *p = nullptr; return set_err(c, e == hipErrorOutOfMemory ? KG_ERR_OOM : KG_ERR_HIP, "kg_malloc", e); }
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_live[*p] = {c, cls};                                   // (an entry left behind by a block released outside kg_free is overwritten here)
  return KG_OK;
  });
}
int kg_ctx_trim(kg_ctx* c) {
  if (!c) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  pool_trim(c);
  return KG_OK;
}
int kg_mem_info(kg_ctx* c, size_t* free_bytes, size_t* total_bytes) {
  if (!c) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  size_t f = 0, t = 0;
  KG_HIP(c, hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f + c->pool_cached;        // kept blocks are released on demand: they count as free
  if (total_bytes) *total_bytes = t;
  return KG_OK;
}
int kg_free(kg_ctx* c, void* p) {
  return kg::kg_guarded(c, [&]() -> int {
  if (!c) return KG_ERR_BAD_ARG;
  if (!p) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  kg_bases_unregister(c, (const uint64_t*)p);     // a freed array must never be served from its registration
  kg_ctx* owner = nullptr;
  size_t cls = 0;
  bool ours = false;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) { owner = it->second.first; cls = it->second.second; ours = true; g_live.erase(it); }
  }
  // not from kg_malloc (a foreign HIP allocation), or handed out by ANOTHER context (kept blocks belong to the pool of the context that
  // drained its queues for them): released
  if (!ours || owner != c) { KG_HIP(c, hipFree(p)); return KG_OK; }
  const size_t cap = (size_t)(c->tune.pool_mb > 0 ? c->tune.pool_mb : 0) << 20;
  if (c->pool_cached + cls <= cap) {
    // hipFree waits for the device; a kept block must be just as safe to hand out again: nothing of this context may still use it
    sync_all(c);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    c->pool_free.emplace(cls, p);
    c->pool_cached += cls;
    return KG_OK;
  }
  KG_HIP(c, hipFree(p));
  return KG_OK;
  });
}
int kg_memcpy_h2d(kg_ctx* c, void* d, const void* h, size_t bytes) {
  if (!c || (bytes && (!d || !h))) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  KG_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  KG_HIP(c, hipStreamSynchronize(c->stream));
  return KG_OK;
}
int kg_memcpy_d2h(kg_ctx* c, void* h, const void* d, size_t bytes) {
  if (!c || (bytes && (!d || !h))) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  KG_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  KG_HIP(c, hipStreamSynchronize(c->stream));
  return KG_OK;
}
int kg_memcpy_d2d(kg_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (bytes && (!dst || !src))) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  KG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return KG_OK;
}
int kg_msm_set_window(kg_ctx* c, int w) {
  if (!c || w < 0 || w > 20) return KG_ERR_BAD_ARG;   // 17 .. 20 need the two-pass sort (2^16 <= n <= 2^24)
  c->msm_window = w;
  return KG_OK;
}
int kg_msm_set_groups(kg_ctx* c, int groups) {
  if (!c || groups < 0 || groups > kg_ctx::MAX_GROUPS) return KG_ERR_BAD_ARG;
  c->msm_groups = groups;
  return KG_OK;
}
int kg_ctx_worker_threads(kg_ctx* c, int* started) {
  if (!c || !started) return KG_ERR_BAD_ARG;
  *started = c->workers ? c->workers->threads_started() : 0;
  return KG_OK;
}
int kg_msm_set_small(kg_ctx* c, int max_pairs, int cw, int r) {
  if (!c || max_pairs < -2 || max_pairs == -1 || max_pairs > 32768 || cw < 0 || cw == 1 || cw > 10 || r < -1 || r > 7) return KG_ERR_BAD_ARG;
  if (max_pairs != -2) c->tune.small_max = max_pairs;
  c->tune.small_c = cw;
  c->tune.small_r = r;
  return KG_OK;
}
int kg_profile_enable(kg_ctx* c, int on) {
  return kg::kg_guarded(c, [&]() -> int {
  if (!c) return KG_ERR_BAD_ARG;
  hipSetDevice(c->device);
  sync_all(c);                   // events still pending on any queue must not be re-recorded by the reset below
  c->prof = on != 0;
  prof_reset(c);                 // phases accumulate from here until the next enable / disable
  return KG_OK;
  });
}
int kg_profile_last(kg_ctx* c, const char** names, float* ms, int cap) {
  return kg_profile_summary(c, names, ms, nullptr, cap);
}
int kg_profile_summary(kg_ctx* c, const char** names, float* total_ms, int* counts, int cap) {
  return kg::kg_guarded(c, [&]() -> int {
  if (!c) return KG_ERR_BAD_ARG;
  hipSetDevice(c->device);
  sync_all(c);
  if (c->tune.profile_timeline && !c->phases.empty()) {     // debugging aid: phase start / end relative to the first phase
    for (auto& p : c->phases) {
      float a = 0, b = 0;
      if (hipEventElapsedTime(&a, c->phases[0].e0, p.e0) == hipSuccess && hipEventElapsedTime(&b, c->phases[0].e0, p.e1) == hipSuccess)
        fprintf(stderr, "[timeline] %-14s %9.1f -> %9.1f us\n", p.name, a * 1e3f, b * 1e3f);
    }
  }
  int n = 0;
  std::vector<const char*> nm;
  std::vector<float> tot;
  std::vector<int> cnt;
  for (auto& p : c->phases) {
    float t = 0;
    if (hipEventElapsedTime(&t, p.e0, p.e1) != hipSuccess) continue;
    size_t k = 0;
    while (k < nm.size() && nm[k] != p.name) ++k;
    if (k == nm.size()) { nm.push_back(p.name); tot.push_back(0.f); cnt.push_back(0); }
    tot[k] += t;
    cnt[k] += 1;
  }
  if (c->host_finish_calls) { nm.push_back("host_finish"); tot.push_back((float)c->host_finish_us.load() * 1e-3f); cnt.push_back(c->host_finish_calls.load()); }
  for (size_t k = 0; k < nm.size() && n < cap; ++k, ++n) {
    if (names) names[n] = nm[k];
    if (total_ms) total_ms[n] = tot[k];
    if (counts) counts[n] = cnt[k];
  }
  return n;
  });
}

}  // extern "C"
