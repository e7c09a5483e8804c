// msm_run.hip -- the base side of an MSM: conversion of the bases, accumulation against one sorted scalar vector (up to three base
// arrays per launch), partial-sum rounds, gather and bucket reduction, export into the result slot.
// (kernels: msm_bases.h, msm_acc_kernels.h, msm_reduce_kernels.h)
#include "msm_reduce_kernels.h"
#include <cstdlib>
#include <cstring>

using namespace kg;
using namespace kg::msm;

namespace kg {
void prep_bases_enqueue(int curve, hipStream_t st, const uint64_t* d_bases, const uint8_t* d_inf, size_t n, uint32_t* out, bool fmt64) {
  if (curve == KG_G1) launch_prep_bases<Fq>(st, d_bases, d_inf, n, out, fmt64);
  else if (curve == KG_GRUMPKIN) launch_prep_bases<Fr>(st, d_bases, d_inf, n, out, fmt64);
  else launch_prep_bases<Fq2>(st, d_bases, d_inf, n, out, fmt64);
}
void table_next_enqueue(int curve, hipStream_t st, const uint32_t* prev, size_t n, int c, uint32_t* next, bool fmt64) {
  const dim3 grid((unsigned)((n + 63) / 64));
  if (curve == KG_G1) hipLaunchKernelGGL(k_table_next<Fq>, grid, dim3(64), 0, st, prev, n, c, next, fmt64 ? 1 : 0);
  else if (curve == KG_GRUMPKIN) hipLaunchKernelGGL(k_table_next<Fr>, grid, dim3(64), 0, st, prev, n, c, next, fmt64 ? 1 : 0);
  else hipLaunchKernelGGL(k_table_next<Fq2>, grid, dim3(64), 0, st, prev, n, c, next, fmt64 ? 1 : 0);
}
}  // namespace kg

namespace kg {

// Base-side half: accumulate + reduce against up to MAX_FUSED base arrays that share the scalar sort S, export, and start
// the copy of each result into its host slot.  One accumulation launch serves all arrays (see k_acc_tasks); everything
// after it runs per array, its reduction on one of the two side queues.
struct RunJob { const uint64_t* d_bases; const uint8_t* d_inf; size_t nbases; uint32_t idx_off; int slot; bool bases_complete; const uint32_t* packed; bool packed64; };

template <class Cfg>
int msm_run_multi_t(kg_ctx* ctx, const MsmSorted& S, const RunJob* jobs, int njobs) {
  using F = typename Cfg::F;
  using KF = typename Cfg::KF;                      // field type of the reduction kernels (Fq2: a lane pair per task, fp2s.h)
  constexpr unsigned LPT = Lanes<KF>::N;            // lanes per task
  constexpr int PW = 2 * BaseIO<F>::PE;             // resident words per base
  constexpr int NW = PointIO<F>::NW;                // raw words per XYZZ point
  if (njobs < 1 || njobs > MAX_FUSED) return set_err(ctx, KG_ERR_BAD_ARG, "bad number of fused base arrays");
  const int W = S.W, B = S.B, c = S.c;
  const size_t npts = S.npts, part_cap = S.part_cap, nexp = (size_t)W * c;
  const size_t exp_bytes = nexp * 4 * Cfg::E64 * 8;
  hipStream_t st = S.acc_stream ? S.acc_stream : ctx->stream, sq;      // a window group may accumulate on a queue of its own
  KG_TRY(scalar_queue(ctx, &sq));
  bool ordered_bases = false, converted = false, shared_pb = false;
  if (!ctx->side_stream) KG_TRY(make_side_stream(ctx));
  if (st != ctx->stream && !ctx->inputs_complete) {      // stream semantics: the group's queue follows what the main queue holds so far
    KG_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_order, 0));
  }
  struct Lay { size_t o_pb, o_lc[2], o_lr[2], o_lb[2], o_part[2], o_pbuf[2], o_rowtot, o_misc, o_hot; char* ws; const uint32_t* pb; int set; };
  Lay lay[MAX_FUSED];
  AccSets A;
  A.nsets = njobs;
  for (int k = 0; k < MAX_FUSED; ++k) { A.pb[k] = nullptr; A.idx_off[k] = 0; A.partial[k] = nullptr; A.tab_n[k] = 0; A.fmt64[k] = 0; }
  for (int k = 0; k < njobs; ++k) {
    const RunJob& J = jobs[k];
    Lay& Y = lay[k];
    Carver cv;
    const uint32_t* reg_pb = nullptr;               // bases inside a registered array are already in packed internal form
    if (J.packed) {                                 // or the caller converted them (kg_msm: once for all window groups)
      if (S.merged_shift) return set_err(ctx, KG_ERR_BAD_ARG, "merged sort against caller-converted bases");
      reg_pb = J.packed; A.fmt64[k] = J.packed64 ? 1 : 0; shared_pb = true;
    }
    for (const auto& r : ctx->registered) {
      if (reg_pb) break;
      if (r.curve != Cfg::ID || J.d_bases < r.base) continue;
      const size_t off64 = (size_t)(J.d_bases - r.base);
      if (off64 % (size_t)BaseIO<F>::W != 0 || off64 / BaseIO<F>::W + J.nbases > r.n) continue;
      // the identity flags were baked in at registration: the resident copy serves the call only when the call's flag
      // array is the registered one (same offset), or both are absent; any other combination converts per call
      const size_t off = off64 / BaseIO<F>::W;
      if (J.d_inf != (r.inf ? r.inf + off : nullptr)) continue;
      if (S.merged_shift) {                         // merged sort: the whole array through its window table
        // (an index slice of the array reads the same rows from its offset: row w of point i is table[w * r.n + i])
        if (!r.table || r.table_c != c || r.table_W != S.windows) continue;
        reg_pb = r.table + off * (r.table64 ? 2 * BaseIO<F>::PK : PW);
        A.tab_n[k] = (uint32_t)r.n;
        A.fmt64[k] = r.table64 ? 1 : 0;
      } else {
        reg_pb = r.packed + off * (r.fmt64 ? 2 * BaseIO<F>::PK : PW);
        A.fmt64[k] = r.fmt64 ? 1 : 0;
      }
      break;
    }
    if (S.merged_shift && !reg_pb) return set_err(ctx, KG_ERR_BAD_ARG, "merged sort against bases without a matching window table");
    const bool conv64 = !reg_pb && resident_fmt64(J.nbases);          // per-call conversion: the same rule as registration
    if (conv64) A.fmt64[k] = 1;
    Y.o_pb = cv.take(reg_pb ? 256 : J.nbases * (conv64 ? 2 * BaseIO<F>::PK : PW) * 4);
    for (int i = 0; i < 2; ++i) {
      Y.o_lc[i] = cv.take(npts * 4); Y.o_lr[i] = cv.take(npts * 4); Y.o_lb[i] = cv.take((size_t)(W + 1) * 4);
      Y.o_part[i] = cv.take(part_cap * NW * 4); Y.o_pbuf[i] = cv.take(npts * NW * 4);
    }
    Y.o_rowtot = cv.take((size_t)W * 4); Y.o_misc = cv.take(64);
    Y.o_hot = cv.take(S.nhot ? (size_t)(S.nhot < HOT_MAX ? S.nhot : HOT_MAX) * hot_split<KF>(S.max_cnt) * NW * 4 : 0);      // k_hot_sum's shares
    Y.set = J.slot % kg_ctx::RUN_SETS;              // run space per set: the reductions of the previous MSMs may still read the other sets
    for (int k2 = 0; k2 < k; ++k2)
      if (lay[k2].set == Y.set) return set_err(ctx, KG_ERR_BAD_ARG, "fused MSMs need result slots in different run-space sets");
    KG_TRY(ensure_ws_run(ctx, Y.set, cv.off));
    KG_TRY(ensure_slot(ctx, J.slot, exp_bytes));
    if (!ctx->ev_acc[Y.set]) KG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_acc[Y.set], hipEventDisableTiming));
    Y.ws = (char*)ctx->ws_run[Y.set];
    Y.pb = reg_pb ? reg_pb : (uint32_t*)(Y.ws + Y.o_pb);
    // this buffer set was last used by an earlier slot: its side-stream work must be over before we overwrite it
    for (int s2 = 0; s2 < kg_ctx::NSLOTS; ++s2)
      if (s2 % kg_ctx::RUN_SETS == Y.set && ctx->slots[s2].done && ctx->slots[s2].busy) {
        KG_HIP(ctx, hipStreamWaitEvent(st, ctx->slots[s2].done, 0));
        if (!reg_pb) KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->slots[s2].done, 0));
        ctx->slots[s2].busy = false;
      }
    if (!reg_pb) {
      // per-call conversion of the bases, on the scalar queue: it runs beside the previous MSM's accumulation instead
      // of between two accumulations on the main queue
      if (!ctx->inputs_complete && !ordered_bases && !J.bases_complete) {   // stream semantics: the bases may still be in flight on the main queue
        KG_HIP(ctx, hipEventRecord(ctx->ev_order, st));
        KG_HIP(ctx, hipStreamWaitEvent(sq, ctx->ev_order, 0));
        ordered_bases = true;
      }
      PhaseScope ph(ctx, "prep_bases", sq);
      launch_prep_bases<F>(sq, J.d_bases, J.d_inf, J.nbases, (uint32_t*)(Y.ws + Y.o_pb), conv64);
      ph.end();
      converted = true;
    }
    A.pb[k] = Y.pb; A.idx_off[k] = J.idx_off; A.partial[k] = (uint32_t*)(Y.ws + Y.o_part[0]);
  }
  const Level L0{S.lcnt, S.lrel, S.lbase};
  if (converted) {                                                       // later on the scalar queue than the sort: covers both
    KG_HIP(ctx, hipEventRecord(ctx->ev_bases, sq));
    KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_bases, 0));
  } else if (S.ready && S.sorted_on != st) KG_HIP(ctx, hipStreamWaitEvent(st, S.ready, 0));   // the scalar queue's sort of this set
  if (shared_pb) KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_pb, 0));     // the caller's conversion of the bases
  if (S.ntasks) {
    PhaseScope ph(ctx, "accumulate", st);
#ifdef KG_EXPERIMENTS      // two accumulation variants that measured level (EXPERIMENTS.md): compiled into A/B builds only
    if (LPT > 1 && ctx->tune.g2_pair_acc)      // KG_G2_PAIR_ACC=1: G2 accumulation on lane pairs, ~150 VGPRs instead of 250
      hipLaunchKernelGGL(k_acc_tasks<KF>, dim3((unsigned)(((size_t)S.ntasks * LPT + 63) / 64) * (unsigned)njobs), dim3(64), 0, st, A, S.sorted, S.bstart, S.bsize, L0,
                         S.task_bkt, S.task_id, S.n, W, B, S.T, part_cap, S.merged_shift, S.T_top, S.top_w);
    else if (acc_prefetch<F>(A, njobs, jobs[0].nbases))      // KG_ACC_PREFETCH=1: the next base prefetched into LDS
      hipLaunchKernelGGL(k_acc_tasks_q<typename PfField<F>::T>, dim3(((S.ntasks + 63) / 64) * (unsigned)njobs), dim3(64), 0, st, A, S.sorted, S.bstart, S.bsize, L0, S.task_bkt,
                         S.task_id, S.n, W, B, S.T, part_cap, S.merged_shift, S.T_top, S.top_w);
    else
#endif
    hipLaunchKernelGGL(k_acc_tasks<F>, dim3(((S.ntasks + 63) / 64) * (unsigned)njobs), dim3(64), 0, st, A, S.sorted, S.bstart, S.bsize, L0, S.task_bkt, S.task_id,
                       S.n, W, B, S.T, part_cap, S.merged_shift, S.T_top, S.top_w);
    ph.end();
  }
  for (int k = 0; k < njobs; ++k) {
    const RunJob& J = jobs[k];
    const Lay& Y = lay[k];
    const int slot = J.slot, set = Y.set;
    char* ws = Y.ws;
    const size_t* o_lc = Y.o_lc; const size_t* o_lr = Y.o_lr; const size_t* o_lb = Y.o_lb;
    uint32_t* part[2] = {(uint32_t*)(ws + Y.o_part[0]), (uint32_t*)(ws + Y.o_part[1])};
    uint32_t* pbuf[2] = {(uint32_t*)(ws + Y.o_pbuf[0]), (uint32_t*)(ws + Y.o_pbuf[1])};
    uint32_t* rowtot = (uint32_t*)(ws + Y.o_rowtot);
    uint32_t* misc = (uint32_t*)(ws + Y.o_misc);
    kg_ctx::Slot& sl = ctx->slots[slot];
    // two reduction queues, by slot parity: a long reduction (G2: ~4x a G1 one) does not hold up the next MSM's
    hipStream_t side = S.reduce_inline ? st : ((slot & 1) ? ctx->side2_stream : ctx->side_stream);     // reduce_inline: behind the accumulation on its own queue (a blocking call's last or only window group: no cross-queue hand-over on the critical path)
    // Everything after the accumulation runs on a reduction queue, so that the main queue goes from one accumulation straight
    // to the next: the partial-sum rounds, the dense bucket array (gather) and the c-1 latency-bound halving levels.
    if (side != st) {
      KG_HIP(ctx, hipEventRecord(ctx->ev_acc[set], st));
      KG_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_acc[set], 0));
    }
    Level L = L0;
    int pcur = 0;
    bool fused_first = false;
    {
      // buckets cut into several tasks (skewed inputs; every bucket of a merged sort): re-sum a bucket's partial sums until
      // it owns one point
      PhaseScope ph(ctx, "partial_sums", side);
      uint32_t max_cnt = S.max_cnt;
      int lv = -1;                                     // -1: level arrays of S; 0/1: local ping-pong
      const unsigned g1024 = (unsigned)((npts + 1023) / 1024);
      const bool hot_ok = ctx->tune.hot_sum != 0;
      if (max_cnt > GATHER_SUM_MAX && hot_ok && S.nhot >= 1 && S.nhot <= HOT_MAX && (size_t)part_cap * NW * 4 < ((size_t)1 << 32)) {
        // the few buckets with more partial sums than the gather takes: one workgroup-wide tree each
        PhaseScope ph2(ctx, "hot_sum", side);
        uint32_t* hot_scratch = (uint32_t*)(ws + Y.o_hot);
        const uint32_t split = hot_split<KF>(S.max_cnt);
        hipLaunchKernelGGL(k_hot_sum<KF>, dim3(S.nhot, split), dim3(256), 36 * 256 * 4, side, part[pcur], L, B, S.hot_list, hot_scratch, split);
        hipLaunchKernelGGL(k_hot_fold<KF>, dim3(S.nhot), dim3(256), 36 * 256 * 4, side, part[pcur], L, B, S.hot_list, hot_scratch, split);
        ph2.end();
        max_cnt = GATHER_SUM_MAX;
      }
      while (max_cnt > GATHER_SUM_MAX) {               // (at most GATHER_SUM_MAX partial sums per bucket are left to the gather below)
        const int nx = lv < 0 ? 0 : (lv ^ 1);
        PhaseScope pr(ctx, "partial_round", side);         // one per extra round: its count is what a skewed input costs (bench.py msm_skewed)
        uint32_t* ncnt = (uint32_t*)(ws + o_lc[nx]);
        uint32_t* nrel = (uint32_t*)(ws + o_lr[nx]);
        uint32_t* nbase = (uint32_t*)(ws + o_lb[nx]);
        KG_HIP(ctx, hipMemsetAsync(misc, 0, 64, side));
        hipLaunchKernelGGL(k_task_count, dim3(g1024), dim3(1024), 0, side, L.cnt, npts, S.T2, ncnt, misc + 8);
        hipLaunchKernelGGL(k_scan_rows, dim3(W), dim3(1024), 0, side, ncnt, B, nrel, rowtot);
        hipLaunchKernelGGL(k_row_bases, dim3(1), dim3(64), 0, side, rowtot, W, nbase, (const uint32_t*)nullptr, misc + 4);
        Level Lout{ncnt, nrel, nbase};
        // the task count of this round is bounded by the previous one; threads beyond base[W] exit
        const uint32_t bound = lv < 0 ? S.ntasks : (uint32_t)part_cap;
        hipLaunchKernelGGL(k_sum_tasks<KF>, dim3((unsigned)(((size_t)bound * LPT + 63) / 64)), dim3(64), 0, side, part[pcur], L, Lout, W, B, S.T2, part[pcur ^ 1]);
        pr.end();
        pcur ^= 1;
        L = Lout;
        lv = nx;
        max_cnt = (max_cnt + S.T2 - 1) / S.T2;
      }
      ph.end();
      PhaseScope pg(ctx, "gather", side);
      const bool fuse_ok = ctx->tune.gather_fuse != 0;
      fused_first = fuse_ok && max_cnt <= 1 && (uint32_t)B > (uint32_t)TailCfg<KF>::L && (size_t)part_cap * NW * 4 < ((size_t)1 << 32);
      if (max_cnt > 1)
        hipLaunchKernelGGL((k_gather_sum<F, KF>), dim3((unsigned)((npts * LPT + 63) / 64)), dim3(64), 0, side, part[pcur], L, W, B, pbuf[0]);
      else if (fused_first)        // the dense bucket array is never written: level 1 straight from the partial sums
        hipLaunchKernelGGL(k_gather_halve<KF>, dim3((unsigned)((npts / 2 * LPT + 63) / 64)), dim3(64), 0, side, part[pcur], L, W, B, pbuf[0], (size_t)W * 2 * (B / 2));
      else
        hipLaunchKernelGGL(k_gather_buckets<F>, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, side, part[pcur], part_cap, L, W, B, pbuf[0]);
      pg.end();
    }
    if (ctx->ws_idle_n[S.set] < kg_ctx::IDLE_EVS) {      // the gather is the last reader of the scalar-side set (level tables of S)
      KG_HIP(ctx, hipEventRecord(ctx->ev_ws_idle[S.set][ctx->ws_idle_n[S.set]], side));
      ctx->ws_idle_n[S.set] += 1;
    } else KG_HIP(ctx, hipStreamSynchronize(side));       // more readers than events: wait here instead (never in practice)
    int cur = 0;
    {
      PhaseScope ph(ctx, "reduce", side);
      size_t in_stride = npts;                     // level 0 reads the bucket array: stride = W*B items
      int narr = 1;
      uint32_t len = (uint32_t)B;                  // items per array
      if (fused_first) { narr = 2; len = (uint32_t)B / 2; in_stride = (size_t)W * 2 * len; }     // k_gather_halve wrote level 1
      constexpr uint32_t TL = (uint32_t)TailCfg<KF>::L;
      for (; len > TL; len /= 2) {                 // the wide levels: one launch each
        const uint32_t n_out = len / 2;
        const size_t tasks = (size_t)W * narr * n_out;
        const size_t out_stride = (size_t)W * (narr + 1) * n_out;
        hipLaunchKernelGGL(k_halve<KF>, dim3((unsigned)((tasks * LPT + 63) / 64)), dim3(64), 0, side, pbuf[cur], in_stride, pbuf[cur ^ 1], out_stride,
                           W, narr, n_out);
        cur ^= 1;
        in_stride = out_stride;
        ++narr;
      }
      // the remaining log2(len) levels and the export in one launch (W * narr workgroups)
      const size_t tail_lds = tail_lds_bytes(len, (int)LPT);
      const unsigned tail_threads = (unsigned)(len / 2 * LPT) < 64u ? 64u : (unsigned)(len / 2 * LPT);
      // the sums go straight into the slot's pinned host buffer (its device view): no copy kernel behind the tail
      if (S.tail_alone && ctx->tune.coop_tail && len >= 2) {        // nothing beside it: a quad of lanes per addition (k_reduce_tail_coop)
        const size_t lds = tail_coop_lds_bytes<F>(len);
        static std::atomic<uint64_t> attr_devs{0};      // once per device
        const uint64_t bit = (uint64_t)1 << (ctx->device & 63);
        if (!(attr_devs.load() & bit)) { KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_reduce_tail_coop<F, Cfg::E64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_devs |= bit; }
        hipLaunchKernelGGL((k_reduce_tail_coop<F, Cfg::E64>), dim3((unsigned)(W * narr)), dim3(TailCoopNT<F>::NT), lds, side, pbuf[cur], in_stride, narr, len, c, (uint64_t*)sl.host_dev);
      } else
      if (len == TL) {
        KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_reduce_tail<KF, Cfg::E64, (int)TL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
        hipLaunchKernelGGL((k_reduce_tail<KF, Cfg::E64, (int)TL>), dim3((unsigned)(W * narr)), dim3(tail_threads), tail_lds, side, pbuf[cur], in_stride, narr, len, c, (uint64_t*)sl.host_dev);
      } else {                                           // a small window (len <= TL / 2): the images keep the stride of the longest such array
        if (len > TL / 2) return set_err(ctx, KG_ERR_UNSUPPORTED, "reduction tail: array length");     // (B is a power of two: never)
        const size_t lds0 = tail_lds_bytes(TL / 2, (int)LPT);
        KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_reduce_tail<KF, Cfg::E64, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
        hipLaunchKernelGGL((k_reduce_tail<KF, Cfg::E64, 0>), dim3((unsigned)(W * narr)), dim3(tail_threads), lds0, side, pbuf[cur], in_stride, narr, len, c, (uint64_t*)sl.host_dev);
      }
      ph.end();
    }
    KG_HIP(ctx, hipGetLastError());
    KG_HIP(ctx, hipEventRecord(sl.done, side));
    sl.W = W; sl.c = c; sl.w0 = S.w0; sl.busy = true; sl.combined = false;
  }
  host_trace("run: enqueued");
  return KG_OK;
}

bool has_window_table(const kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t nbases, size_t msm_len) {
  const int c = merged_window(ctx, msm_len);
  if (!c) return false;
  for (const auto& r : ctx->registered)
    if (r.base == d_bases && r.curve == curve && r.n == nbases && r.inf == d_inf && r.table && r.table_c == c) return true;
  return false;
}

// window width of the table that covers [d_bases, d_bases + cnt) inside a registered array, if a merged sort of cnt scalars can use it
// (an index slice of a host-scalar call); 0: none
int slice_table_window(const kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t cnt) {
  const size_t pw64 = curve == KG_G2 ? 16 : 8;
  if (cnt < ((size_t)1 << 16)) return 0;
  int s = 0;
  while (((size_t)1 << s) < cnt) ++s;
  for (const auto& r : ctx->registered) {
    if (r.curve != curve || !r.table || d_bases < r.base) continue;
    const size_t off64 = (size_t)(d_bases - r.base);
    if (off64 % pw64 != 0 || off64 / pw64 + cnt > r.n) continue;
    if (d_inf != (r.inf ? r.inf + off64 / pw64 : nullptr)) continue;
    if (((size_t)r.table_W << s) > ((size_t)1 << 24)) continue;       // (window, index) must fit the entry's 24-bit field
    return r.table_c;
  }
  return 0;
}

int scalar_queue(kg_ctx* ctx, hipStream_t* out) {
  KG_TRY(make_sort_stream(ctx));
  *out = ctx->sort_stream;
  return KG_OK;
}

int msm_run_multi(kg_ctx* ctx, const MsmSorted& S, int curve, const MsmRunJob* jobs, int njobs) {
  RunJob rj[MAX_FUSED];
  if (njobs < 1 || njobs > MAX_FUSED) return KG_ERR_BAD_ARG;
  for (int k = 0; k < njobs; ++k)
    rj[k] = RunJob{jobs[k].d_bases, jobs[k].d_inf, jobs[k].nbases, jobs[k].idx_off, jobs[k].slot, jobs[k].bases_complete, jobs[k].packed, jobs[k].packed64};
  switch (curve) {
    case KG_G1: return msm_run_multi_t<G1Cfg>(ctx, S, rj, njobs);
    case KG_GRUMPKIN: return msm_run_multi_t<GkCfg>(ctx, S, rj, njobs);
    case KG_G2: return msm_run_multi_t<G2Cfg>(ctx, S, rj, njobs);
    default: return KG_ERR_BAD_ARG;
  }
}
int msm_run(kg_ctx* ctx, const MsmSorted& S, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t nbases, uint32_t idx_off, int slot) {
  const MsmRunJob j{d_bases, d_inf, nbases, idx_off, slot, false, nullptr, false};
  return msm_run_multi(ctx, S, curve, &j, 1);
}

}  // namespace kg
