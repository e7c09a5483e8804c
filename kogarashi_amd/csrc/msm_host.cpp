// msm_host.cpp -- the host side of the MSM entries of the C ABI: the blocking / ticketed / sliced / grouped orchestration of
// kg_msm, kg_msm_begin/_end, kg_msm_host, kg_commit, base registration, and the host finish (255-step double-and-add over the
// bit-plane sums, host_fp.h).  No kernels live here: everything on the device is enqueued through the kg:: functions of
// msm.hip (msm_sort*, msm_run*, prep_bases_enqueue, table_next_enqueue).
//
// Replaces the call shapes of groth16/src/msm.rs:6-48 (msm_curve_addition) and nova/src/pedersen.rs:15-20 (commit).
#include "common.h"
#include "host_fp.h"
#include "msm_internal.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace kg;

namespace {

struct G1Cfg { using HF = HostFq; static constexpr int E64 = 4; static constexpr int ID = KG_G1; };
struct GkCfg { using HF = HostFr; static constexpr int E64 = 4; static constexpr int ID = KG_GRUMPKIN; };
struct G2Cfg { using HF = HostFq2; static constexpr int E64 = 8; static constexpr int ID = KG_G2; };

template <class HF> struct HostIO;
template <class P> struct HostIO<HostFp<P>> {
  static HostFp<P> load(const uint64_t* w) { return HostFp<P>::from_words(w); }
  static void store(const HostFp<P>& a, uint64_t* w) { a.to_words(w); }
};
template <class F> struct HostIO<Fp2<F>> {
  static Fp2<F> load(const uint64_t* w) { return {HostIO<F>::load(w), HostIO<F>::load(w + 4)}; }
  static void store(const Fp2<F>& a, uint64_t* w) { HostIO<F>::store(a.c0, w); HostIO<F>::store(a.c1, w + 4); }
};

template <class Cfg>
XYZZ<typename Cfg::HF> host_load_point(const uint64_t* p) {
  using HF = typename Cfg::HF;
  constexpr int E = Cfg::E64;
  return {HostIO<HF>::load(p), HostIO<HF>::load(p + E), HostIO<HF>::load(p + 2 * E), HostIO<HF>::load(p + 3 * E)};
}

template <class Cfg>
void store_projective(const XYZZ<typename Cfg::HF>& p, uint64_t* out_xyz) {
  using HF = typename Cfg::HF;
  constexpr int E = Cfg::E64;
  Affine<HF> a;
  if (!to_affine(p, a)) {                        // (0, 1, 0): macros/curve/weierstrass/group.rs:106-110
    HostIO<HF>::store(HF::zero(), out_xyz);
    HostIO<HF>::store(HF::one(), out_xyz + E);
    HostIO<HF>::store(HF::zero(), out_xyz + 2 * E);
    return;
  }
  HostIO<HF>::store(a.x, out_xyz);
  HostIO<HF>::store(a.y, out_xyz + E);
  HostIO<HF>::store(HF::one(), out_xyz + 2 * E);
}

}  // namespace

namespace kg {

// Host half: wait for the slot's copy, then the 255-step double-and-add over the c*W bit-plane sums.
// Window w contributes 2^(w*c) * (A_w + sum_l 2^l T_{w,l}); array 0 = A, array 1 + l = T_l.
// An MSM cut into window groups holds one slot per group, top windows first: the chain runs through the groups in that
// order and waits for a slot only when it reaches the slot's windows, so the top of the chain is computed while the lower
// groups are still on the device.
template <class Cfg>
int msm_finish_t(kg_ctx* ctx, const int* slots, int nslots, uint64_t* out_xyz) {
  using HF = typename Cfg::HF;
  hipSetDevice(ctx->device);
  host_trace("finish: enter");
  constexpr int PE = 4 * Cfg::E64;
  XYZZ<HF> acc = XYZZ<HF>::identity();
  long long busy_us = 0;
  for (int s = 0; s < nslots; ++s) {
    kg_ctx::Slot& sl = ctx->slots[slots[s]];
    if (hipEventSynchronize(sl.done) != hipSuccess) return KG_ERR_HIP;
    host_trace("finish: slot ready");
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t* hp = (const uint64_t*)sl.host;
    const int W = sl.W, c = sl.c, w0 = sl.w0;
    if (sl.combined) {                                     // one point per window (msm_small.hip): 2^c * acc + S_w
      for (int w = W - 1; w >= 0; --w) {
        for (int l = 0; l < c; ++l) acc = double_xyzz(acc);
        acc = add_xyzz(acc, host_load_point<Cfg>(hp + (size_t)w * PE));
      }
    } else
    for (int bit = (w0 + W) * c - 1; bit >= w0 * c; --bit) {
      acc = double_xyzz(acc);
      const int w = bit / c - w0, l = bit % c;
      if (c > 1 && l < c - 1) acc = add_xyzz(acc, host_load_point<Cfg>(hp + ((size_t)w * c + 1 + l) * PE));
      if (l == 0) acc = add_xyzz(acc, host_load_point<Cfg>(hp + ((size_t)w * c) * PE));
    }
    if (s == nslots - 1) store_projective<Cfg>(acc, out_xyz);
    busy_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  host_trace("finish: done");
  if (ctx->prof) {
    ctx->host_finish_us += busy_us;
    ctx->host_finish_calls += 1;
  }
  return KG_OK;
}

int msm_finish_groups(kg_ctx* ctx, int curve, const int* slots, int nslots, uint64_t* out_xyz) {
  if (nslots < 1) return KG_ERR_BAD_ARG;
  switch (curve) {
    case KG_G1: return msm_finish_t<G1Cfg>(ctx, slots, nslots, out_xyz);
    case KG_GRUMPKIN: return msm_finish_t<GkCfg>(ctx, slots, nslots, out_xyz);
    case KG_G2: return msm_finish_t<G2Cfg>(ctx, slots, nslots, out_xyz);
    default: return KG_ERR_BAD_ARG;
  }
}
int msm_finish(kg_ctx* ctx, int curve, int slot, uint64_t* out_xyz) { return msm_finish_groups(ctx, curve, &slot, 1, out_xyz); }
void msm_identity(int curve, uint64_t* out_xyz) {
  if (curve == KG_G2) store_projective<G2Cfg>(XYZZ<HostFq2>::identity(), out_xyz);
  else if (curve == KG_GRUMPKIN) store_projective<GkCfg>(XYZZ<HostFr>::identity(), out_xyz);
  else store_projective<G1Cfg>(XYZZ<HostFq>::identity(), out_xyz);
}

}  // namespace kg

namespace {

template <class Cfg>
int sum_affine_impl(const uint64_t* pts, const uint8_t* inf, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
  using HF = typename Cfg::HF;
  constexpr int E = Cfg::E64;
  XYZZ<HF> acc = XYZZ<HF>::identity();
  for (size_t i = 0; i < count; ++i) {
    if (inf && inf[i]) continue;
    Affine<HF> a{HostIO<HF>::load(pts + i * 2 * E), HostIO<HF>::load(pts + i * 2 * E + E)};
    acc = add_mixed(acc, a);
  }
  uint64_t xyz[3 * 8];
  store_projective<Cfg>(acc, xyz);
  std::memcpy(out_xy, xyz, 2 * E * 8);
  *out_inf = is_identity(acc) ? 1 : 0;
  return KG_OK;
}

}  // namespace

extern "C" {

// Sum of the slices' results (projective (x, y, 1) / (0, 1, 0) each) -> the same form
static int sum_slices(kg_ctx* ctx, int curve, const uint64_t (*part)[24], int K, uint64_t* out_xyz) {
  const int E = curve == KG_G2 ? 8 : 4;
  if (K == 1) { std::memcpy(out_xyz, part[0], (size_t)3 * E * 8); return KG_OK; }
  uint64_t pts[kg_ctx::UP_SLICES * 16];
  uint8_t pinf[kg_ctx::UP_SLICES];
  for (int j = 0; j < K; ++j) {
    bool z0 = true;
    for (int k = 0; k < E; ++k) z0 = z0 && part[j][2 * E + k] == 0;
    pinf[j] = z0 ? 1 : 0;
    std::memcpy(pts + (size_t)j * 2 * E, part[j], (size_t)2 * E * 8);
  }
  uint64_t xy[16];
  uint8_t inf = 0;
  KG_TRY(kg_points_sum_affine(ctx, curve, pts, pinf, (size_t)K, xy, &inf));
  kg::msm_identity(curve, out_xyz);                        // (0, 1, 0); y doubles as the field's one
  if (!inf) {
    for (int k = 0; k < E; ++k) out_xyz[2 * E + k] = out_xyz[E + k];
    std::memcpy(out_xyz, xy, (size_t)2 * E * 8);
  }
  return KG_OK;
}

// A large blocking MSM as a pipeline over index slices: slice j+1 is sorted (scalar queue) while slice j accumulates, and
// the slices' reductions and host finishes run under the later accumulations; the slices' sums are added on the host.
// Unsliced, the sort (4.0 ms of 25.8 at 2^24) sits in front of the accumulation.  The slices keep the
// window width of the whole (c = 17), so the number of additions does not change; the extra bucket reductions are hidden.
static int msm_sliced(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, uint64_t* out_xyz) {
  constexpr int K = 4;
  const size_t pw = curve == KG_G2 ? 16 : 8;               // u64 words per base
  size_t lo[K + 1];
  for (int j = 0; j <= K; ++j) lo[j] = n / K * j + (j == K ? n % K : 0);
  const int saved_window = ctx->msm_window;
  struct Restore { kg_ctx* c; int w; ~Restore() { c->msm_window = w; } } restore_window{ctx, saved_window};      // also when a submit below throws
  if (!saved_window) { const int cw = pick_window(n, 0), cs = pick_window(n / K, 0); ctx->msm_window = cw > cs ? cw : cs; }
  uint64_t part[K][24];
  std::future<int> fin[K];
  kg::WaitAll fin_done{fin, K};                            // the finishes write into part[]
  int rc = KG_OK;
  const int sfield = curve == KG_GRUMPKIN ? KG_FQ : KG_FR;
  for (int j = 0; j < K && rc == KG_OK; ++j) {
    const size_t a = lo[j], cnt = lo[j + 1] - lo[j];
    kg::MsmSorted S;
    rc = kg::msm_sort(ctx, sfield, d_scalars + 4 * a, cnt, &S, j > 0);      // slice 0 orders the scalar queue behind the inputs' producer
    if (rc != KG_OK) break;
    rc = kg::msm_run(ctx, S, curve, d_bases + a * pw, d_inf ? d_inf + a : nullptr, cnt, 0, 16 + j);
    if (rc != KG_OK) break;
    uint64_t* out = part[j];
    fin[j] = kg::pool(ctx).submit([ctx, curve, j, out] { return kg::msm_finish(ctx, curve, 16 + j, out); });
  }
  ctx->msm_window = saved_window;
  for (int j = 0; j < K; ++j)
    if (fin[j].valid()) { const int r2 = fin[j].get(); if (rc == KG_OK) rc = r2; }
  if (rc != KG_OK) { kg_ctx_sync(ctx); return rc; }
  return sum_slices(ctx, curve, part, K, out_xyz);
}

// A blocking MSM pipelined against itself by window groups (kg_ctx::MAX_GROUPS): the windows of one MSM are independent
// until the host's double-and-add, so after ONE conversion of the scalars (k_prep_scalars_count peels all digits) the top
// group is sorted, then accumulated while the next group is sorted, reduced while the next group accumulates, and its sums
// feed the top of the host chain while the lower groups are still on the device.  Exactly the work of the unsplit call
// (an index slice would add a bucket reduction and a host chain per slice); the groups' accumulations go to queues of
// their own so that a group's tail and the next group's head share the chip.
static int msm_grouped(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, const int* gw, int NG,
                       uint64_t* out_xyz) {
  KG_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->side_stream) KG_TRY(make_side_stream(ctx));
  KG_TRY(make_sort_stream(ctx));
  for (int g = 1; g < NG && g < ctx->tune.group_accq; ++g)      // only the queues the groups rotate over (two: acc_stream[1] comes placed from place_queues)
    if (!ctx->acc_stream[g]) KG_HIP(ctx, create_stream(ctx, &ctx->acc_stream[g], false));
  kg::MsmSortPlan Q;
  // KG_GROUP_MAIN_FIRST=1 (experiment): conversion and the first group's sort on the main queue, in front of its accumulation -- no
  // cross-queue hand-over there, but the second group's sort then runs BESIDE the first one's, the two accumulations start 50 us
  // apart, share the chip from the start and neither finishes early: 1.93-2.04 ms against 1.77-1.85 with both sorts in turn on the
  // scalar queue (2^20, same box, alternating runs; unsplit 1.85-1.90)
  const int main_first = ctx->tune.group_main_first;
  KG_TRY(kg::msm_sort_begin(ctx, curve == KG_GRUMPKIN ? KG_FQ : KG_FR, d_scalars, n, &Q, false, 0, 1, NG, gw, main_first != 0));
  if (main_first == 1) {                                   // the scalar queue (the later groups' sorts) follows the conversion
    KG_HIP(ctx, hipEventRecord(ctx->ev_prep, ctx->stream));
    KG_HIP(ctx, hipStreamWaitEvent(ctx->sort_stream, ctx->ev_prep, 0));
  }
  // the bases: a registered array is resident already; otherwise ONE conversion for all groups, on a reduction queue (idle
  // at this point) beside the scalar conversion
  const size_t pw64 = curve == KG_G2 ? 16 : 8;             // u64 words per ABI point
  bool resident = false;
  for (const auto& r : ctx->registered) {
    if (r.curve != curve || d_bases < r.base) continue;
    const size_t off64 = (size_t)(d_bases - r.base);
    if (off64 % pw64 != 0 || off64 / pw64 + n > r.n) continue;
    if (d_inf != (r.inf ? r.inf + off64 / pw64 : nullptr)) continue;
    resident = true;
    break;
  }
  const bool f64 = resident_fmt64(n);
  if (!resident) {
    const size_t bytes = n * (f64 ? (curve == KG_G2 ? 128 : 64) : (curve == KG_G2 ? 144 : 72));
    if (bytes > ctx->ws_pb_bytes) {
      if (ctx->ws_pb) { sync_all(ctx); hipFree(ctx->ws_pb); ctx->ws_pb = nullptr; ctx->ws_pb_bytes = 0; }
      const hipError_t e = dev_alloc(ctx, &ctx->ws_pb, bytes + bytes / 8);
      if (e != hipSuccess) return set_err(ctx, KG_ERR_OOM, "resident-bases allocation", e);
      ctx->ws_pb_bytes = bytes + bytes / 8;
    }
    hipStream_t cq = ctx->side_stream;
    if (!ctx->inputs_complete) {                          // stream semantics: the bases may still be in flight on the main queue
      KG_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
      KG_HIP(ctx, hipStreamWaitEvent(cq, ctx->ev_order, 0));
    }
    PhaseScope ph(ctx, "prep_bases", cq);
    prep_bases_enqueue(curve, cq, d_bases, d_inf, n, (uint32_t*)ctx->ws_pb, f64);
    ph.end();
    KG_HIP(ctx, hipGetLastError());
    KG_HIP(ctx, hipEventRecord(ctx->ev_pb, cq));
  }
  kg::MsmSorted S[kg_ctx::MAX_GROUPS];
  int slots[kg_ctx::MAX_GROUPS];
  int rc = kg::msm_sort_group(ctx, Q, 0, &S[0], main_first != 0);
  // KG_GROUP_MAIN_FIRST=2: as 1, but the scalar queue waits for the first group's SORT -- the sorts still run in turn, and the
  // cross-queue hand-over (20-60 us) sits in front of the second sort, which has the whole first accumulation to hide in, instead of
  // in front of the first accumulation
  if (rc == KG_OK && main_first == 2) KG_HIP(ctx, hipStreamWaitEvent(ctx->sort_stream, S[0].ready, 0));
  if (rc == KG_OK && NG > 1) rc = kg::msm_sort_group(ctx, Q, 1, &S[1]);
  int launched = 0;
  for (int g = 0; g < NG && rc == KG_OK; ++g) {
    rc = kg::msm_sort_wait(ctx, &S[g]);
    if (rc != KG_OK) break;
    // accumulation queues the groups rotate over: 2 (the main queue and one more).  On ONE queue a group's launch waits for the
    // previous group's last wave -- and the top window's tasks are twice as long as the others (unsigned digits: half the buckets),
    // so the chip idles behind them: 2^20 in two groups 2.09 ms on one queue, 1.80 on two; a queue per group (four) is no better
    // than two, and more than ~4 busy hardware queues start to delay each other's hand-overs
    const int accq = ctx->tune.group_accq;
    S[g].acc_stream = (accq > 1 && g % accq) ? ctx->acc_stream[g % accq] : nullptr;
    const int rinl = ctx->tune.group_reduce_inline;
    S[g].reduce_inline = rinl && g == NG - 1;
    S[g].tail_alone = g == NG - 1;                          // the last group's tail has the chip
    const int one_side = ctx->tune.group_one_side;
    slots[g] = one_side ? 16 + 2 * g : 16 + g;
    kg::MsmRunJob job{d_bases, d_inf, n, 0u, slots[g], false, resident ? nullptr : (const uint32_t*)ctx->ws_pb, f64};
    rc = kg::msm_run_multi(ctx, S[g], curve, &job, 1);
    if (rc != KG_OK) break;
    ++launched;
    if (g + 2 < NG) rc = kg::msm_sort_group(ctx, Q, g + 2, &S[g + 2]);
  }
  if (rc != KG_OK) { sync_all(ctx); return rc; }
  (void)launched;
  return kg::msm_finish_groups(ctx, curve, slots, NG, out_xyz);
}

static int msm_blocking(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, uint64_t* out_xyz);

int kg_msm(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, uint64_t* out_xyz) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !out_xyz || (n && (!d_bases || !d_scalars)) || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  if (n == 0) { kg::msm_identity(curve, out_xyz); return KG_OK; }
  return msm_blocking(ctx, curve, d_bases, d_inf, d_scalars, n, out_xyz);
  });
}

// one blocking MSM over device arrays (kg_msm; the unsliced kg_msm_host_scalars behind its upload)
static int msm_blocking(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, uint64_t* out_xyz) {
  int sc = 0, sr = 0;
  if (kg::msm_small_plan(ctx, curve, n, &sc, &sr)) {      // short inputs: one launch on the main queue (stream-ordered behind the inputs' producer), then the host chain
    KG_TRY(kg::msm_small_enqueue(ctx, ctx->stream, curve, d_bases, d_inf, d_scalars, n, 0, sc, sr));
    return kg::msm_finish(ctx, curve, 0, out_xyz);
  }
  const bool sliced_ok = ctx->tune.msm_sliced != 0;     // 0 (experiments): window groups instead of index slices
  if (n >= ((size_t)1 << 23) && sliced_ok && pick_window(n, ctx->msm_window) < 19) return msm_sliced(ctx, curve, d_bases, d_inf, d_scalars, n, out_xyz);   // measured: 2^24 25.8 -> 23.0 ms, 2^23 13.1 -> 12.1; below that the slices' own tails cost more than the hidden sort
  kg::MsmSorted S;
  // window tables serve a BLOCKING call up to 2^18 pairs only: the merged sort is one piece, so nothing of it hides under an accumulation,
  // while the plain form runs in window groups (msm_grouped).  Measured (MI355X, registered bases, ms without / with tables): 2^16 0.54 / 0.42,
  // 2^17 0.63 / 0.50, 2^18 0.73 / 0.67, 2^19 1.00 / 1.05, 2^20 1.54 / 1.73 (profiles/r05_blocking_tables.txt); calls in flight
  // (kg_msm_begin) keep the tables at every size -- there the next call's sort is hidden either way
  const bool tables_ok = n <= ((size_t)1 << ctx->tune.blocking_tables_log);
  const int mc = tables_ok && kg::has_window_table(ctx, curve, d_bases, d_inf, n, n) ? kg::merged_window(ctx, n) : 0;
  if (!mc) {
    int gw[kg_ctx::MAX_GROUPS];
    const int NG = kg::msm_group_plan(ctx, n, gw);
    if (NG > 1) return msm_grouped(ctx, curve, d_bases, d_inf, d_scalars, n, gw, NG, out_xyz);
  }
  ctx->sort_alone = true;                                 // a blocking call: its sort is all the device has to do
  const int rs = kg::msm_sort(ctx, curve == KG_GRUMPKIN ? KG_FQ : KG_FR, d_scalars, n, &S, false, mc);
  ctx->sort_alone = false;
  KG_TRY(rs);
  S.tail_alone = true;                                    // a blocking call: nothing runs beside its reduction
  S.reduce_inline = ctx->tune.blocking_reduce_inline != 0;    // ... and nothing follows its accumulation: the reduction stays on the accumulation's queue (no cross-queue hand-over)
  KG_TRY(kg::msm_run(ctx, S, curve, d_bases, d_inf, n, 0, 0));
  return kg::msm_finish(ctx, curve, 0, out_xyz);
}

int kg_msm_pick_window(size_t n) { return pick_window(n ? n : 1, 0); }
int kg_msm_table_window(size_t msm_len) { return kg::merged_window(nullptr, msm_len); }

int kg_bases_register(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t n) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;                         // nothing to convert (e.g. an empty CRS vector)
  if (!d_bases) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  kg_bases_unregister(ctx, d_bases);
  const bool fmt64 = resident_fmt64(n);
  const size_t pw = fmt64 ? (curve == KG_G2 ? 32 : 16) : (curve == KG_G2 ? 36 : 18);
  uint32_t* packed = nullptr;
  KG_HIP(ctx, dev_alloc(ctx, (void**)&packed, n * pw * 4));
  prep_bases_enqueue(curve, ctx->stream, d_bases, d_inf, n, packed, fmt64);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) {
    hipFree(packed);
    return set_err(ctx, KG_ERR_HIP, "k_prep_bases launch", e);
  }
  kg_ctx::Registered reg{d_bases, d_inf, n, curve, packed};
  reg.fmt64 = fmt64;
  ctx->registered.push_back(reg);
  return KG_OK;
  });
}

int kg_bases_unregister(kg_ctx* ctx, const uint64_t* d_bases) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx) return KG_ERR_BAD_ARG;
  for (size_t i = 0; i < ctx->registered.size(); ++i) {
    if (ctx->registered[i].base == d_bases) {
      kg_ctx_sync(ctx);
      hipFree(ctx->registered[i].packed);
      if (ctx->registered[i].table) hipFree(ctx->registered[i].table);
      ctx->registered.erase(ctx->registered.begin() + i);
      return KG_OK;
    }
  }
  return KG_OK;
  });
}

int kg_bases_precompute(kg_ctx* ctx, const uint64_t* d_bases, size_t msm_len) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !d_bases) return KG_ERR_BAD_ARG;
  KG_HIP(ctx, hipSetDevice(ctx->device));
  kg_ctx::Registered* r = nullptr;
  for (auto& e : ctx->registered) if (e.base == d_bases) r = &e;
  if (!r) return set_err(ctx, KG_ERR_BAD_ARG, "kg_bases_precompute: the array is not registered");
  if (msm_len == 0) msm_len = r->n;
  if (msm_len < r->n) return set_err(ctx, KG_ERR_BAD_ARG, "kg_bases_precompute: msm_len is shorter than the array");
  const int c = kg::merged_window(ctx, msm_len);
  if (!c) return set_err(ctx, KG_ERR_BAD_ARG, "kg_bases_precompute: window tables are offered for MSMs of 2^16 .. 2^20 scalars");
  if (r->table && r->table_c == c) return KG_OK;
  if (r->table) { kg_ctx_sync(ctx); hipFree(r->table); r->table = nullptr; r->table_c = r->table_W = 0; }
  const int W = (255 + c - 1) / c;
  const bool t64 = table_fmt64();
  const size_t pw = t64 ? (r->curve == KG_G2 ? 32 : 16) : (r->curve == KG_G2 ? 36 : 18), row = r->n * pw;
  uint32_t* table = nullptr;
  if (hipError_t e = dev_alloc(ctx, (void**)&table, (size_t)W * row * 4); e != hipSuccess) {
    (void)hipGetLastError();
    return set_err(ctx, KG_ERR_OOM, "window table allocation", e);
  }
  hipStream_t st = ctx->stream;
  // row 0 from the caller's array (the resident copy may be in the other form), then one launch per further window
  prep_bases_enqueue(r->curve, st, r->base, r->inf, r->n, table, t64);
  hipError_t e = hipGetLastError();
  for (int w = 1; w < W && e == hipSuccess; ++w) {
    table_next_enqueue(r->curve, st, table + (size_t)(w - 1) * row, r->n, c, table + (size_t)w * row, t64);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { hipFree(table); return set_err(ctx, KG_ERR_HIP, "window table build", e); }
  r->table = table; r->table_c = c; r->table_W = W; r->table64 = t64;
  return KG_OK;
  });
}

int kg_msm_begin(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, int ticket) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || (n && (!d_bases || !d_scalars)) || curve < 0 || curve > KG_G2 || ticket < 0 || ticket > 3) return KG_ERR_BAD_ARG;
  ctx->ticket_n[ticket] = n;
  if (n == 0) return KG_OK;
  int sc = 0, sr = 0;
  // short inputs: one launch, the host chain on a worker thread.  Only up to 8192 pairs (G2: 4096) for calls in flight: the kernel's grid
  // fills the chip at these lengths and consecutive calls run one after the other, where the long pipeline's phases overlap across the
  // tickets (per call, four in flight, short kernel / long pipeline: 2^10 0.09 / 0.15 ms, 2^12 0.14 / 0.17, 2^13 0.172 / 0.174, 2^14 0.21 / 0.18;
  // G2 2^12 0.32 / 0.33, 2^13 0.44 / 0.37 -- profiles/r06_small_ab.txt)
  // (calls in flight are bound by their kernels, which the halved scalars of msm_digits.h lengthen for the base-field curves -- 2^10 pairs 0.091 -> 0.114 ms
  // per call, their host chains run on worker threads beside each other; G2: level)
  struct GlvGuard { kg_ctx* c; unsigned old; ~GlvGuard() { c->small_glv_off = old; } } glv_guard{ctx, ctx->small_glv_off};
  ctx->small_glv_off |= (1u << KG_G1) | (1u << KG_GRUMPKIN);
  if (n <= (size_t)(curve == KG_G2 ? ctx->tune.small_max_flight / 2 : ctx->tune.small_max_flight) && kg::msm_small_plan(ctx, curve, n, &sc, &sr)) {
    if (ctx->ticket_fut[ticket].valid()) ctx->ticket_fut[ticket].wait();
    KG_TRY(kg::msm_small_enqueue(ctx, ctx->stream, curve, d_bases, d_inf, d_scalars, n, 1 + ticket, sc, sr));
    uint64_t* out = ctx->ticket_out[ticket];
    ctx->ticket_fut[ticket] = kg::pool(ctx).submit([ctx, curve, ticket, out] { return kg::msm_finish(ctx, curve, 1 + ticket, out); });
    return KG_OK;
  }
  kg::MsmSorted S;
  const int mc = kg::has_window_table(ctx, curve, d_bases, d_inf, n, n) ? kg::merged_window(ctx, n) : 0;
  KG_TRY(kg::msm_sort(ctx, curve == KG_GRUMPKIN ? KG_FQ : KG_FR, d_scalars, n, &S, false, mc));
  // KG_PIPE_ACCQ=2 (experiment, round 4): MSMs in flight alternate between two accumulation queues on the same compute pipe, so that the
  // next launch's first waves fill the tail in which this one drains.  Measured level to slightly worse (1.347 -> 1.358 ms per step,
  // three alternating runs): a two-round launch with longest-first tasks has little tail to fill.  Off.
  const int pipe_accq = ctx->tune.pipe_accq;
  if (pipe_accq > 1 && (ticket & 1) && ctx->acc_stream[1]) S.acc_stream = ctx->acc_stream[1];
  if (ctx->ticket_fut[ticket].valid()) ctx->ticket_fut[ticket].wait();      // a ticket begun twice without its end: drop the older result
  KG_TRY(kg::msm_run(ctx, S, curve, d_bases, d_inf, n, 0, 1 + ticket));      // slots 1..4 (slot 0: kg_msm; 6..15: the prover's two jobs)
  uint64_t* out = ctx->ticket_out[ticket];
  ctx->ticket_fut[ticket] = kg::pool(ctx).submit([ctx, curve, ticket, out] { return kg::msm_finish(ctx, curve, 1 + ticket, out); });
  return KG_OK;
  });
}

int kg_msm_end(kg_ctx* ctx, int curve, int ticket, uint64_t* out_xyz) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !out_xyz || curve < 0 || curve > KG_G2 || ticket < 0 || ticket > 3) return KG_ERR_BAD_ARG;
  if (ctx->ticket_n[ticket] == 0) { kg::msm_identity(curve, out_xyz); return KG_OK; }
  if (!ctx->ticket_fut[ticket].valid()) return kg::set_err(ctx, KG_ERR_BAD_ARG, "kg_msm_end without a matching kg_msm_begin");
  const int rc = ctx->ticket_fut[ticket].get();
  if (rc != KG_OK) return rc;
  std::memcpy(out_xyz, ctx->ticket_out[ticket], (curve == KG_G2 ? 24 : 12) * 8);
  return KG_OK;
  });
}

// Host arrays in, one point out: the call shape of the reference's slices (msm_curve_addition(&[C], &[C::Scalar])).
// The index range is cut into slices that travel through a pipeline: an uploader thread copies slice j's scalars (then, for
// kg_msm_host, its bases) into cached device buffers (upload queue); the scalar queue sorts slice j as soon as its scalars
// have landed, converts its bases when they have, and the main queue accumulates it while slice j+1 is still on the bus.
// The slices' partial sums are added on the host.
//   kg_msm_host          both arrays from host memory: PCIe (96 B per G1 pair) is the floor, 1.8 ms per 2^20 pairs.
//   kg_msm_host_scalars  the bases are a device array (a CRS vector, a commitment key: registered, so nothing is converted per
//                        call) and only the scalars travel -- 32 B per pair.  This is what the Rust call sites do per call:
//                        the bases of groth16/src/msm.rs:6 / nova/src/pedersen.rs:15-20 are fixed, the scalars are new.
// Only the FIRST slice's upload is exposed (everything downstream needs its scalars), so it is the short one:
// msm_host_plan cuts the range into one half-share and K - 1 full shares.
static int grow_device(kg_ctx* ctx, int which, size_t bytes) {
  if (bytes <= ctx->up_bytes[which]) return KG_OK;
  if (ctx->up_buf[which]) { sync_all(ctx); hipFree(ctx->up_buf[which]); ctx->up_buf[which] = nullptr; ctx->up_bytes[which] = 0; }
  const hipError_t e = dev_alloc(ctx, &ctx->up_buf[which], bytes + bytes / 8);
  if (e != hipSuccess) { (void)hipGetLastError(); return set_err(ctx, KG_ERR_OOM, "upload buffer allocation", e); }
  ctx->up_bytes[which] = bytes + bytes / 8;
  return KG_OK;
}

// Slice boundaries lo[0 .. K] of an n-pair host-array MSM; returns K.  scalars_only: the bases are resident.
//   both arrays : 4 equal slices from 2^20 pairs, 2 from 2^18 (the bus is the floor).  Measured, K = 1 / 2 / 4 (round 5, ms): 2^16 0.67 / 0.84 / 1.16,
//                 2^17 0.93 / 1.01 / 1.24, 2^18 1.28 / 1.22 / 1.54, 2^19 1.99 / 1.78 / 1.86, 2^20 3.62 / 2.95 / 2.80
//   scalars only: the first slice is 1 / first_div of an equal share (its upload is the one nothing hides), the others equal;
//                 K grows with n so that a slice stays a well-filled MSM (>= 2^17 pairs) and the exposed upload stays short
static int msm_host_plan(const kg_tuning& tune, size_t n, bool scalars_only, size_t* lo) {
  int K = n >= ((size_t)1 << 20) ? 4 : (n >= ((size_t)1 << 18) ? 2 : 1);
  int first_div = 1;
  if (scalars_only) {
    // measured (MI355X, registered G1 bases, pageable scalars, ms over the resident blocking kg_msm; tools/dbg/host_scalars.py):
    //   2^20: K = 2 +0.13, 3 +0.23, 4 +0.32      2^21: K = 2 +0.67, 3 +0.63, 4 +0.71      2^22: K = 2 +1.23, 3 +0.93, 4 +0.92
    //   2^23: K = 6 +0.92, 8 +1.13               2^24: K = 4 +3.65, 6 +2.86, 8 +2.82 (1.15x: the slices run c = 17, 15 windows against 13)
    // first_div 2 beats 1, 3 and 4 at every size (2^20, K = 2: +0.13 / +0.14 / +0.18 / +0.12 within noise of each other above 1)
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    //   2^16 .. 2^19, K = 1 / 2 (round 5, profiles/r05_host_small.txt): 0.61 / 0.77, 0.73 / 0.89, 0.97 / 1.06, 1.44 / 1.41
    K = lg >= 24 ? 8 : (lg == 23 ? 6 : (lg == 22 ? 4 : (lg == 21 ? 3 : (lg >= 19 ? 2 : 1))));
    first_div = K > 1 ? 2 : 1;
  }
  if (tune.host_slices >= 1 && tune.host_slices <= kg_ctx::UP_SLICES) K = tune.host_slices;
  if (tune.host_first_div >= 1 && tune.host_first_div <= 16) first_div = tune.host_first_div;
  if ((size_t)K > n) K = (int)n;
  if (K <= 1) { lo[0] = 0; lo[1] = n; return 1; }
  // shares: 1 for the first slice, first_div for each of the others
  const size_t units = 1 + (size_t)(K - 1) * first_div;
  size_t first = n / units;
  if (first < 1) first = 1;
  first &= ~(size_t)255;                                   // keeps the later slices' device addresses 8 KiB-aligned
  if (first == 0) first = n / units ? n / units : 1;
  lo[0] = 0; lo[1] = first;
  const size_t rest = n - first;
  for (int j = 2; j <= K; ++j) lo[j] = first + rest / (K - 1) * (j - 1) + (j == K ? rest % (K - 1) : 0);
  return K;
}

static int msm_host_impl(kg_ctx* ctx, int curve, const uint64_t* bases, const uint8_t* inf, bool bases_on_device, const uint64_t* h_scalars, size_t n,
                         uint64_t* out_xyz) {
  KG_HIP(ctx, hipSetDevice(ctx->device));
  host_trace("host: enter");
  const size_t pb = curve == KG_G2 ? 128 : 64;             // bytes per base
  if (!bases_on_device) {
    KG_TRY(grow_device(ctx, 0, n * pb));
    if (inf) KG_TRY(grow_device(ctx, 2, n));
  }
  KG_TRY(grow_device(ctx, 1, n * 32));
  if (!ctx->ev_up_s[0]) {
    KG_TRY(make_sort_stream(ctx));                         // places the context's queues, the upload queue among them
    if (!ctx->up_stream) KG_HIP(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
    for (int i = 0; i < kg_ctx::UP_SLICES; ++i) {
      KG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_up_s[i], hipEventDisableTiming));
      KG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_up_b[i], hipEventDisableTiming));
    }
  }
  hipStream_t sq;
  KG_TRY(kg::scalar_queue(ctx, &sq));
  if (bases_on_device && !ctx->inputs_complete) {          // stream semantics: a device array may still be in flight on the main queue
    KG_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->ev_order, 0));
  }
  const char* d_b = bases_on_device ? (const char*)bases : (const char*)ctx->up_buf[0];
  uint64_t* d_s = (uint64_t*)ctx->up_buf[1];
  const uint8_t* d_i = bases_on_device ? inf : (inf ? (const uint8_t*)ctx->up_buf[2] : nullptr);
  int sc = 0, sr = 0;
  if (kg::msm_small_plan(ctx, curve, n, &sc, &sr)) {
    // short inputs: the arrays go up on the main queue (a few KB: the copies are staged and return at once), the one-launch MSM behind them
    KG_HIP(ctx, hipMemcpyAsync(d_s, h_scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (!bases_on_device) {
      KG_HIP(ctx, hipMemcpyAsync(ctx->up_buf[0], bases, n * pb, hipMemcpyHostToDevice, ctx->stream));
      if (inf) KG_HIP(ctx, hipMemcpyAsync(ctx->up_buf[2], inf, n, hipMemcpyHostToDevice, ctx->stream));
    }
    KG_TRY(kg::msm_small_enqueue(ctx, ctx->stream, curve, (const uint64_t*)d_b, d_i, d_s, n, 0, sc, sr));
    return kg::msm_finish(ctx, curve, 0, out_xyz);
  }
  size_t lo[kg_ctx::UP_SLICES + 1];
  const int K = msm_host_plan(ctx->tune, n, bases_on_device, lo);
  // the previous call's readers of the cached buffers are done (every call joins its slices before it returns)
  std::atomic<int> up_s{0}, up_b{0}, up_rc{(int)hipSuccess};
  std::mutex up_mu;                                        // the enqueuing thread sleeps on up_cv until the uploader has issued the slice's copies
  std::condition_variable up_cv;
  auto upload_scalars = [&](int j) {
    const size_t a = lo[j], cnt = lo[j + 1] - lo[j];
    hipError_t e = hipMemcpyAsync(d_s + 4 * a, h_scalars + 4 * a, cnt * 32, hipMemcpyHostToDevice, ctx->up_stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_up_s[j], ctx->up_stream);
    if (e != hipSuccess) up_rc = (int)e;
    { std::lock_guard<std::mutex> lk(up_mu); up_s = j + 1; }
    up_cv.notify_all();
    host_trace("upload: scalars");
  };
  auto upload_bases = [&](int j) {
    if (!bases_on_device) {
      const size_t a = lo[j], cnt = lo[j + 1] - lo[j];
      hipError_t e = hipMemcpyAsync((char*)ctx->up_buf[0] + a * pb, (const char*)bases + a * pb, cnt * pb, hipMemcpyHostToDevice, ctx->up_stream);
      if (e == hipSuccess && inf) e = hipMemcpyAsync((uint8_t*)ctx->up_buf[2] + a, inf + a, cnt, hipMemcpyHostToDevice, ctx->up_stream);
      if (e == hipSuccess) e = hipEventRecord(ctx->ev_up_b[j], ctx->up_stream);
      if (e != hipSuccess) up_rc = (int)e;
      host_trace("upload: bases");
    }
    { std::lock_guard<std::mutex> lk(up_mu); up_b = j + 1; }
    up_cv.notify_all();
  };
  // The first slice's scalars go up from the calling thread (nothing can start before them: a thread start in front of that copy is
  // 30-40 us on the critical path); everything else from an uploader thread that feeds the pipeline while this thread enqueues
  // sorts and accumulations (a copy from pageable memory occupies its thread for the duration of the copy).
  upload_scalars(0);
  if (K == 1 && bases_on_device) {
    // unsliced (up to 2^18 pairs the whole upload is shorter than what a second slice costs): the blocking kg_msm behind the copy --
    // window groups, or the window table of registered bases
    if (up_rc != (int)hipSuccess) return set_err(ctx, KG_ERR_HIP, "host-to-device upload", (hipError_t)up_rc.load());
    KG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_up_s[0], 0));
    KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->ev_up_s[0], 0));
    return msm_blocking(ctx, curve, bases, inf, d_s, n, out_xyz);
  }
  // (everything that can fail and return comes before the uploader thread exists)
  if (K > 1 && !ctx->acc_stream[1]) { if (create_stream(ctx, &ctx->acc_stream[1], false) != hipSuccess) return set_err(ctx, KG_ERR_HIP, "queue creation"); }
  std::future<int> uploader = kg::pool(ctx).submit([&]() -> int {
    hipSetDevice(ctx->device);
    upload_bases(0);
    for (int j = 1; j < K; ++j) { upload_scalars(j); upload_bases(j); }
    return KG_OK;
  });
  struct WaitGuard { std::future<int>& f; ~WaitGuard() { if (f.valid()) f.wait(); } } uploader_done{uploader};   // the task refers to this frame: also when a later submit throws
  uint64_t part[kg_ctx::UP_SLICES][24];
  std::future<int> fin[kg_ctx::UP_SLICES];
  kg::WaitAll fin_done{fin, kg_ctx::UP_SLICES};            // the finishes write into part[]
  int rc = KG_OK;
  const int sfield = curve == KG_GRUMPKIN ? KG_FQ : KG_FR;
  const bool was_alone = ctx->sort_alone;
  for (int j = 0; j < K && rc == KG_OK; ++j) {
    const size_t a = lo[j], cnt = lo[j + 1] - lo[j];
    { std::unique_lock<std::mutex> lk(up_mu); up_cv.wait(lk, [&] { return up_s.load() > j; }); }
    if (up_rc != (int)hipSuccess) { rc = set_err(ctx, KG_ERR_HIP, "host-to-device upload", (hipError_t)up_rc.load()); break; }
    if (hipStreamWaitEvent(sq, ctx->ev_up_s[j], 0) != hipSuccess) { rc = KG_ERR_HIP; break; }
    kg::MsmSorted S;
    ctx->sort_alone = j == 0;                              // the first slice's sort has the device to itself
    // registered bases with window tables: a slice reads the table's rows from its offset (15 additions per pair instead of 16 and one small
    // bucket set per slice; every slice's sort is one piece anyway)
    const int mc = (bases_on_device && ctx->tune.host_slice_tables) ? kg::slice_table_window(ctx, curve, (const uint64_t*)(d_b + a * pb), d_i ? d_i + a : nullptr, cnt) : 0;
    rc = kg::msm_sort(ctx, sfield, d_s + 4 * a, cnt, &S, true, mc);
    ctx->sort_alone = was_alone;
    if (rc != KG_OK) break;
    if (bases_on_device) {
      // consecutive slices accumulate on two queues (like the window groups of a blocking kg_msm): the next launch's first waves fill the
      // chip while this one's last waves drain; the last slice's reduction follows its accumulation on the same queue
      if (ctx->tune.host_accq > 1 && (j & 1)) S.acc_stream = ctx->acc_stream[1];
      S.reduce_inline = ctx->tune.group_reduce_inline && K > 1 && j == K - 1;
      S.tail_alone = j == K - 1;
    }
    if (!bases_on_device) {
      { std::unique_lock<std::mutex> lk(up_mu); up_cv.wait(lk, [&] { return up_b.load() > j; }); }
      if (up_rc != (int)hipSuccess) { rc = set_err(ctx, KG_ERR_HIP, "host-to-device upload", (hipError_t)up_rc.load()); break; }
      if (hipStreamWaitEvent(sq, ctx->ev_up_b[j], 0) != hipSuccess) { rc = KG_ERR_HIP; break; }
    }
    const kg::MsmRunJob job{(const uint64_t*)(d_b + a * pb), d_i ? d_i + a : nullptr, cnt, 0u, 16 + j, true};
    rc = kg::msm_run_multi(ctx, S, curve, &job, 1);
    if (rc != KG_OK) break;
    uint64_t* out = part[j];
    fin[j] = kg::pool(ctx).submit([ctx, curve, j, out] { return kg::msm_finish(ctx, curve, 16 + j, out); });
  }
  if (uploader.valid()) uploader.wait();
  hipStreamSynchronize(ctx->up_stream);
  host_trace("host: uploads synced");
  for (int j = 0; j < K; ++j)
    if (fin[j].valid()) { const int r2 = fin[j].get(); if (rc == KG_OK) rc = r2; }
  if (rc != KG_OK) { kg_ctx_sync(ctx); return rc; }
  return sum_slices(ctx, curve, part, K, out_xyz);
}

int kg_msm_host_slices(size_t n, int scalars_only, size_t* lo) {
  if (!lo) return KG_ERR_BAD_ARG;
  if (n == 0) { lo[0] = 0; return 0; }
  return msm_host_plan(tuning(), n, scalars_only != 0, lo);
}

int kg_msm_host(kg_ctx* ctx, int curve, const uint64_t* h_bases, const uint8_t* h_inf, const uint64_t* h_scalars, size_t n, uint64_t* out_xyz) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !out_xyz || (n && (!h_bases || !h_scalars)) || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  if (n == 0) return kg_msm(ctx, curve, nullptr, nullptr, nullptr, 0, out_xyz);
  return msm_host_impl(ctx, curve, h_bases, h_inf, false, h_scalars, n, out_xyz);
  });
}

int kg_msm_host_scalars(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* h_scalars, size_t n, uint64_t* out_xyz) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !out_xyz || (n && (!d_bases || !h_scalars)) || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  if (n == 0) return kg_msm(ctx, curve, nullptr, nullptr, nullptr, 0, out_xyz);
  return msm_host_impl(ctx, curve, d_bases, d_inf, true, h_scalars, n, out_xyz);
  });
}

static void xyz_to_commit(int curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf) {
  const int E = curve == KG_G2 ? 8 : 4;
  std::memcpy(out_xy, xyz, 2 * E * 8);
  bool z0 = true;
  for (int i = 0; i < E; ++i) z0 = z0 && xyz[2 * E + i] == 0;
  *out_inf = z0 ? 1 : 0;
}

int kg_commit_host_scalars(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* h_scalars, size_t n,
                           uint64_t* out_xy, uint8_t* out_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !out_xy || !out_inf || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  uint64_t xyz[24];
  KG_TRY(kg_msm_host_scalars(ctx, curve, d_bases, d_inf, h_scalars, n, xyz));
  xyz_to_commit(curve, xyz, out_xy, out_inf);
  return KG_OK;
  });
}

int kg_commit(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n,
              uint64_t* out_xy, uint8_t* out_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !out_xy || !out_inf || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  uint64_t xyz[24];
  KG_TRY(kg_msm(ctx, curve, d_bases, d_inf, d_scalars, n, xyz));
  xyz_to_commit(curve, xyz, out_xy, out_inf);
  return KG_OK;
  });
}

int kg_points_sum_affine(kg_ctx* ctx, int curve, const uint64_t* pts, const uint8_t* inf, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
  (void)ctx;
  if (!out_xy || !out_inf || (count && !pts)) return KG_ERR_BAD_ARG;
  switch (curve) {
    case KG_G1: return sum_affine_impl<G1Cfg>(pts, inf, count, out_xy, out_inf);
    case KG_GRUMPKIN: return sum_affine_impl<GkCfg>(pts, inf, count, out_xy, out_inf);
    case KG_G2: return sum_affine_impl<G2Cfg>(pts, inf, count, out_xy, out_inf);
    default: return KG_ERR_BAD_ARG;
  }
}

}  // extern "C"
