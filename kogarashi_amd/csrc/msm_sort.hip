// msm_sort.hip -- the scalar side of an MSM: window rule, sort plan, and the enqueue of conversion, sort and task decomposition.
// (kernels: msm_sort_kernels.h, msm_task_kernels.h)
//
// The MSM as a whole -- Pippenger bucket MSM for BN254 G1 / G2 and Grumpkin on gfx950.
//
// Replaces groth16/src/msm.rs:6-48 (msm_curve_addition: unsigned c-bit windows, one rayon task per window,
// serial bucket fill, summation by parts, c*i doublings per window) and, behind kg_commit, the naive
// scalar-mul fold of nova/src/pedersen.rs:15-20.  Output parity is on the AFFINE sum, so the
// device pipeline is its own design:
//
//   prep      scalars -> canonical integers k (one Montgomery product) biased by H = sum_w 2^(wc+c-1), so
//             every window's SIGNED digit is a plain bit-field of k+H (halves the bucket count);
//             bases -> internal Montgomery form as 9-limb coordinates (72 B per G1 point), identity flag in a spare bit
//   count     one workgroup per (scalar chunk, window): the window's whole histogram (2^(c-1) counters,
//   scan      up to 128 KiB) lives in LDS -- a single-pass counting sort with a 15-bit digit
//   scatter   -> per-bucket lists of (point index | sign)
//   accumulate one lane per bucket: XYZZ += +-P over its list (madd, 8M+2S), bases gathered through L2/MALL
//   reduce    sum_b b*B_b by log2(B) halving levels (pair sums + odd-index sums = bit planes of b); depth
//             c-1 point additions instead of the reference's 2*2^c-long serial chain
//   finish    the c*W bit-plane sums go to the host, which runs the 255-step double-and-add (host_fp.h)
//
// Algorithmic HBM bytes: 96 B/pair (G1, Grumpkin), 160 B/pair (G2): SURVEY.md 8d.
#include "msm_sort_kernels.h"
#include "msm_task_kernels.h"
#include <chrono>
#include <cstdlib>
#include <cstring>

using namespace kg;
using namespace kg::msm;

namespace kg {
void host_trace(const char* what) {
  if (!tuning().trace_host) return;
  static const auto t0 = std::chrono::steady_clock::now();
  fprintf(stderr, "[host] %-18s %10.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
}

int pick_window(size_t n, int forced) {
  const int wide_from = tuning().wide_window;
  if (forced) return forced;
  int lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) ++lg;      // floor(log2 n)
  // The top window holds the 254 - (W-1)c leftover bits; c = 15 / 16 leave it 14 bits (as many buckets as the
  // signed windows use), while c = 12..14 would leave 2..7 bits, i.e. a handful of buckets holding ~n points each.
  // c = 17 (W = 15, a 16-bit top window) needs the two-pass sort, i.e. n <= 2^24; measured faster from 2^21 up
  // (2^22: 6.6 vs 7.0 ms, 2^24: 25.8 vs 35.2 ms), slower at 2^20 where its bucket reduction doubles
  // c = 20 (13 windows, 2^19 buckets each, 32 entries per bucket at 2^24 -- the density of the 2^20 / c = 16 optimum), UNSLICED and
  // pipelined by window groups (kg_msm): 13 instead of 15 additions per pair and one bucket reduction; needs the nine-bit fine field
  // of the two-pass sort (eight-byte intermediate entries, 20480-entry segments).  Measured at 2^24 (MI355X, alternating runs on one
  // box): 21.35 ms in groups of 3,3,3,4 windows against 22.0 ms for c = 17 in four index slices (four equal groups: 21.8); at 2^23 the
  // slices win (11.3 against 11.75 ms).  The accumulation drops from 19.3 to 16.0 ms of launch time, but each group's sort still runs
  // 2-4x slower beside an accumulation than alone (latency-bound kernels at one or two waves per SIMD) and bounds the pipeline -- before
  // the sort kernels were slimmed to two workgroups per CU beside an accumulation the wide window lost (21.3 against 21.1).
  // Round 5, after round 4's later sort work (big tiles, 512-thread first pass): 2^23 10.32 ms in four window groups against 10.7 in index
  // slices (three groups 10.55, c = 19 10.5; profiles/r05_wide23.txt) -- the wide window is the default from 2^23 pairs.  At 2^22 c = 19 loses
  // (5.9 against 5.4 ms), at 2^24 c = 19 loses to 20 (19.9 against 18.7).  KG_WIDE_WINDOW=0 keeps the slices, =24 widens from 2^24 only.
  if (wide_from > 0 && lg >= wide_from && n <= ((size_t)1 << 24)) return 20;
  if (lg >= 21 && n <= ((size_t)1 << 24)) return 17;
  if (lg >= 19) return 16;
  // Short inputs are latency, not work: what counts is the depth of the chains (a task's additions, log2 B reduction levels), and above all
  // that the unsigned TOP window is not a handful of buckets holding ~n / 8 points each -- which it is for every width that leaves 255 mod c
  // small (c = 4, 6, 7, 9, 11, 12, 14).  Measured (MI355X, blocking kg_msm, ms; profiles/r05_small_windows.txt), c = 5 / 8 / 13 / 15 against
  // the former rule c = lg - 3: 2^6 0.26 / 0.27 / 0.33 / 0.38 (0.32); 2^8 0.38 / 0.31 / 0.38 / 0.41 (0.38); 2^9 0.47 / 0.34 / 0.38 / 0.43 (0.75);
  // 2^10 0.71 / 0.41 / 0.42 / 0.44 (0.78); 2^11 1.18 / 0.54 / 0.48 / 0.43 (0.54); 2^12 2.13 / 0.77 / 0.62 / 0.43 (0.83); 2^13 3.92 / 1.26 / 0.67 / 0.45 (0.85)
  // (2^10: level between 8 and 15 when blocking, 0.19 against 0.16 ms per step with four calls in flight; 1536 pairs 0.49 against 0.43)
  if (lg >= 10) return 15;
  return lg >= 7 ? 8 : 5;
}

// Which resident form an array of n bases gets.  The 64-byte point is the default at every size: its ~50 re-spreading
// instructions per addition cost 1-2 % of the accumulation when it runs alone, but in the pipeline -- where the next sort and
// the previous reductions compete for the memory system -- halving the sectors per gather wins (2^20: 1.452 -> 1.421 ms per
// step; 2^22 blocking 6.40 -> 6.19 ms; the PMC traffic of a launch halves).  KG_FMT64_MIN_LOG=30 brings the 72-byte form back
// (experiments, and the cross-format test).
bool resident_fmt64(size_t n) { return n >= ((size_t)1 << tuning().fmt64_min_log); }
bool table_fmt64() { return tuning().table64 != 0; }      // tables never fit the cache: 2^20 1.36 -> 1.30 ms per step, Groth16 2.94 -> 2.87
}  // namespace kg

namespace kg {

// Window width of the merged form: one set of 2^(c-1) buckets for all windows, so the bucket reduction and the host finish
// shrink by the window count and a wider window costs nothing extra -- c = 17 (15 windows) from 2^17 scalars.  Offered where
// the two-pass sort runs and (window << ceil(log2 n)) | index fits an entry's 24-bit field.
int merged_window(const kg_ctx* ctx, size_t n) {
  if (n < ((size_t)1 << 16) || n > ((size_t)1 << 20)) return 0;
  int c = n >= ((size_t)1 << 17) ? 17 : 16;
  if (ctx && ctx->msm_window >= 15 && ctx->msm_window <= 18) c = ctx->msm_window;
  int s = 0;
  while (((size_t)1 << s) < n) ++s;
  const int W = (255 + c - 1) / c;
  if (((size_t)W << s) > ((size_t)1 << 24)) return 0;
  return c;
}

// Window groups of a blocking MSM (see msm_grouped): offered where the two-pass sort runs.  KG_MSM_GROUPS = 0 / 1 switches
// them off, = k asks for k equal groups, = "a,b,c" names the groups' window counts from the top window down (experiments).
int msm_group_plan(const kg_ctx* ctx, size_t n, int* gw) {
  if (n < ((size_t)1 << 16) || n > ((size_t)1 << 24)) return 0;
  const int c = pick_window(n, ctx ? ctx->msm_window : 0);
  if (c - 1 < FINE_BITS + 4) return 0;
  const int W = (255 + c - 1) / c;
  // measured (MI355X, blocking kg_msm, two accumulation queues): two groups give 2^17 0.70 -> 0.68 ms, 2^18 0.88 -> 0.82, 2^19 1.195 -> 1.116,
  // 2^20 1.87 -> 1.79, 2^21 3.16 -> 3.01, 2^22 6.14 -> 5.97; three or four groups pay more launches and more sort beside the
  // accumulations than their shorter reduction tail returns (2^20: 1.98 / 2.03 ms)
  int NG = n >= ((size_t)1 << 22) ? 3 : (n >= ((size_t)1 << 17) ? 2 : 0);       // 2^22 (round 4, slimmed sort kernels): 5.97 / 5.79 / 5.83 ms in 2 / 3 / 4 groups
  if (c >= 19) NG = 4;                                 // the unsliced 2^23..2^24-pair commitments: the sort of 13-14 windows is 4 ms, hidden group by group
  if (ctx && ctx->msm_groups) NG = ctx->msm_groups;   // kg_msm_set_groups
  const kg_tuning& tn = ctx ? ctx->tune : tuning();
  if (!tn.msm_groups_list.empty() || tn.msm_groups >= 0) {
    if (!tn.msm_groups_list.empty()) {
      int k = 0, sum = 0;
      const char* p = tn.msm_groups_list.c_str();
      while (*p && k < kg_ctx::MAX_GROUPS) {
        const int v = atoi(p);
        if (v < 1) return 0;
        gw[k++] = v; sum += v;
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
      if (sum == W && !*p) return k;                 // a list that does not fit this window count falls through to the default
    } else NG = tn.msm_groups;
  }
  if (NG > kg_ctx::MAX_GROUPS) NG = kg_ctx::MAX_GROUPS;
  if (NG > W) NG = W;
  if (NG < 2) return 0;
  // equal groups, the remainder to the top ones -- except the wide windows, where the LAST group takes it (3,3,3,4 of 13: the top
  // group's sort is the only one nothing hides; 21.35 against 21.8 ms at 2^24)
  for (int g = 0; g < NG; ++g) gw[g] = W / NG + (c >= 19 ? (g >= NG - W % NG ? 1 : 0) : (g < W % NG ? 1 : 0));
  return NG;
}

int msm_sort_begin(kg_ctx* ctx, int scalar_field, const uint64_t* d_scalars, size_t n, MsmSortPlan* P, bool ordered, int merged_c, int lane_mult,
                   int ngroups, const int* gw, bool on_main) {
  if (n == 0 || n >= ((size_t)1 << 31)) return set_err(ctx, KG_ERR_BAD_ARG, "msm length must be in [1, 2^31)");
  host_trace("sort: enter");
  KG_HIP(ctx, hipSetDevice(ctx->device));
  const bool merged = merged_c != 0;
  int c = merged ? merged_c : pick_window(n, ctx->msm_window);
  if (!merged && c > 16 && !(n >= ((size_t)1 << 16) && n <= ((size_t)1 << 24))) c = 16;   // one-pass histogram: 2^(c-1) LDS counters
  const int W = (255 + c - 1) / c;                  // windows of the scalars
  const int B = 1 << (c - 1);
  int mshift = 0;
  if (merged) {
    while (((size_t)1 << mshift) < n) ++mshift;
    if (c < 15 || c > 18 || n < ((size_t)1 << 16) || ((size_t)W << mshift) > ((size_t)1 << 24))
      return set_err(ctx, KG_ERR_BAD_ARG, "merged sort not offered for this length / window");
  }
  const int Wb = merged ? 1 : W;                    // windows of the BUCKET space: the merged form keeps one set for all digits
  const size_t nv = merged ? (size_t)W * n : n;     // entries that can meet one bucket window
  int nch = (int)((n + 16383) / 16384);
  const int nch_cap = ctx->tune.sort_nch >= 1 && ctx->tune.sort_nch <= 1024 ? ctx->tune.sort_nch : 64;
  if (nch > nch_cap) nch = nch_cap;               // (window, chunk) workgroups of the first sort pass: 1024 of them at 2^20
  if (nch < 1) nch = 1;
  size_t chunk_len = (n + nch - 1) / nch;
  // Task length.  A task is one lane's sequential chain of additions, so the accumulation can never be shorter than T
  // additions' latency (14 us each at four waves per SIMD, ~23 us for G2) however little work there is -- a 0/1-heavy witness
  // (a third of a uniform input's additions) took LONGER than a uniform one with T = 4 n / B + 32: 1.47 against 1.04 ms for the
  // prover's G2 query.  T = 2 n / B + 16 still leaves a uniform input one task per bucket (a bucket holds n / B entries on
  // average, Poisson: 2 n / B + 16 is 8 sigma out at n / B = 16 and 8.5 at 32); the unsigned top window holds twice the load per
  // bucket and gets twice the length (T_top).  Hot buckets pay for the shorter tasks with more partial sums: k_hot_sum.
  uint32_t T = (uint32_t)(2 * (n / B) + 16);
  if (ctx->tune.msm_t >= 4 && ctx->tune.msm_t <= 4096) T = (uint32_t)ctx->tune.msm_t;      // experiments
  if (T < 32) T = 32;
  if (T > 2048) T = 2048;
  if (merged) {
    // a bucket holds ~W n / B entries (60 at 2^18, 240 at 2^20): cut so that the accumulation has about one resident round of
    // lanes (4096 waves); every task beyond the first of a bucket costs one partial-sum addition afterwards
    const size_t lanes = (size_t)(lane_mult < 1 ? 1 : lane_mult);
    size_t t = nv * lanes / ((size_t)4096 * 64);
    T = 16;
    while (T < 128 && 2 * (size_t)T <= t) T *= 2;
    if (ctx->tune.merged_t >= 4 && ctx->tune.merged_t <= 4096) T = (uint32_t)ctx->tune.merged_t;
  }
  // two passes (bucket group, then bucket inside the group) once the sorted lists outgrow the L2; entries carry the
  // bucket's low FINE_BITS between the passes, which leaves 24 bits for the index
  const bool two_pass = merged || (c - 1 >= FINE_BITS + 4 && n >= ((size_t)1 << 16) && n <= ((size_t)1 << 24));
  const bool alone_ok = ctx->tune.sort_alone != 0;   // 0 (experiments): every sort shaped for a busy device
  const bool alone = alone_ok && (ctx->sort_alone || ngroups > 1);      // the (first group's) sort runs on an otherwise idle device
  if (c >= 19 && (!two_pass || merged)) return set_err(ctx, KG_ERR_BAD_ARG, "windows of 19 and 20 bits need the two-pass sort (2^16 .. 2^24 scalars)");
  const int fb = fine_bits_for(c);                  // low bucket bits an entry carries between the passes
  const uint32_t FINE = 1u << fb;
  const int G = two_pass ? B >> fb : 0;             // bucket groups per window (<= 1024)
  const uint32_t seg = seg_len_for(fb);
  const int maxseg = two_pass ? G + (int)((nv + seg - 1) / seg) : 0;
  if (two_pass) chunk_len = (chunk_len + PREP_CH - 1) / PREP_CH * PREP_CH;   // k_prep_scalars_count: one chunk per workgroup
  // window groups: gw[0] windows from the top, then gw[1], ... (the host's double-and-add chain consumes them in that order)
  if (ngroups < 1 || ngroups > kg_ctx::MAX_GROUPS) return set_err(ctx, KG_ERR_BAD_ARG, "bad number of window groups");
  if (ngroups > 1 && (!two_pass || merged || !gw)) return set_err(ctx, KG_ERR_BAD_ARG, "window groups need the two-pass, unmerged sort");
  MsmSortPlan& Q = *P;
  Q = MsmSortPlan();
  {
    int top = Wb, sum = 0;
    for (int g = 0; g < ngroups; ++g) {
      const int wg = ngroups == 1 ? Wb : gw[g];
      if (wg < 1) return set_err(ctx, KG_ERR_BAD_ARG, "empty window group");
      top -= wg; sum += wg;
      Q.gw0[g] = top; Q.gW[g] = wg;
    }
    if (sum != Wb) return set_err(ctx, KG_ERR_BAD_ARG, "window groups do not add up to the window count");
  }
  Carver cv;
  Q.o_kt = cv.take(n * 32); Q.o_cnt = cv.take((size_t)W * nch * (two_pass ? G : B) * 4); Q.o_bsize = cv.take((size_t)W * B * 4);
  Q.o_bstart = cv.take((size_t)Wb * B * 4);
  Q.o_tmp = cv.take(two_pass ? (size_t)W * n * (fb == 9 ? 8 : 4) : 0); Q.o_gsize = cv.take((size_t)W * G * 4); Q.o_gstart = cv.take((size_t)W * G * 4);
  Q.o_segbase = cv.take((size_t)W * (G + 1) * 4); Q.o_segcnt = cv.take((size_t)Wb * maxseg * FINE * 4); Q.o_segoff = cv.take((size_t)Wb * maxseg * FINE * 4);
  Q.o_sorted = cv.take((size_t)W * n * 4); Q.o_lcnt = cv.take((size_t)Wb * B * 4); Q.o_lrel = cv.take((size_t)Wb * B * 4);
  Q.o_rowtot = cv.take((size_t)W * 4);
  Q.o_woff = cv.take(merged ? (size_t)W * G * 4 : 0); Q.o_gsize_m = cv.take(merged ? (size_t)G * 4 : 0); Q.o_gstart_m = cv.take(merged ? (size_t)G * 4 : 0);
  Q.o_segbase_m = cv.take(merged ? (size_t)(G + 1) * 4 : 0);
  Q.o_bpart = cv.take((size_t)W * (B / 4096 + 32) * 2 * 4);      // k_bucket_part: (entries, tasks) of each part of each row
  for (int g = 0; g < ngroups; ++g) {                     // what the task decomposition keeps per group
    Q.part_cap[g] = (size_t)Q.gW[g] * (4 * ((nv + T - 1) / T)) + (size_t)Q.gW[g] * B;     // upper bound on round-1 tasks (hot buckets: tasks of T / 4, bucket_task_len)
    Q.o_lbase[g] = cv.take((size_t)(Q.gW[g] + 1) * 4);
    Q.o_misc[g] = cv.take(64); Q.o_lenh[g] = cv.take(2 * LEN_BINS * 4);             // adjacent: one zero fill covers both
    Q.o_tbkt[g] = cv.take(Q.part_cap[g] * 4); Q.o_tid[g] = cv.take(Q.part_cap[g] * 4);
    Q.o_hot[g] = cv.take((size_t)HOT_MAX * 4);
  }
  // The scalar side runs on a queue of its own and alternates between two spaces: while MSM i accumulates (main queue,
  // reading set i & 1), MSM i+1 is sorted into the other set.  Ordering: the scalar queue waits for `after` (the producer
  // of d_scalars), or -- stream semantics -- for everything enqueued on the main queue so far, unless the context's inputs
  // are declared complete (kg_ctx_set_inputs_complete); and for the last reader of the set it is about to overwrite.
  const int set = (int)(ctx->sort_seq++ & 1u);
  KG_TRY(ensure_ws_sort(ctx, set, cv.off));
  KG_TRY(ensure_pinned(ctx, 4096));
  KG_TRY(make_sort_stream(ctx));
  char* ws = (char*)ctx->ws_sort[set];
  // on_main: the conversion runs on the main queue (a blocking call whose first window group is sorted and accumulated there:
  // no cross-queue hand-over in front of the first accumulation); the scalar queue is put behind it by the caller
  hipStream_t st = on_main ? ctx->stream : ctx->sort_stream;
  if (!on_main && !ordered && !ctx->inputs_complete) {
    KG_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_order, 0));
  }
  for (int j = 0; j < ctx->ws_idle_n[set]; ++j) KG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_ws_idle[set][j], 0));
  ctx->ws_idle_n[set] = 0;
  Q.n = n; Q.chunk_len = chunk_len; Q.nv = nv; Q.c = c; Q.W = W; Q.B = B; Q.Wb = Wb; Q.G = G; Q.nch = nch; Q.maxseg = maxseg; Q.mshift = mshift;
  Q.set = set; Q.ngroups = ngroups; Q.merged = merged; Q.two_pass = two_pass; Q.T = T; Q.ws = ws;
  Q.alone = alone;
  Q.T_top = merged ? T : 2 * T;
  Q.fb = fb;
  uint32_t* kt = (uint32_t*)(ws + Q.o_kt);
  uint32_t* cnt = (uint32_t*)(ws + Q.o_cnt);

  Words8 H;                                          // bias H = sum_{w < W-1} 2^(w*c + c - 1)
  for (int j = 0; j < 8; ++j) H.w[j] = 0;
  for (int w = 0; w < W - 1; ++w) {
    int bit = w * c + c - 1;
    H.w[bit >> 5] |= 1u << (bit & 31);
  }
  {
    PhaseScope ph(ctx, "prep_scalars", st);
    if (two_pass) {
      const size_t hl = (size_t)W * G * 4;
      zero_fill(st, cnt, (size_t)W * nch * G * 4);
      // PREP_CH scalars per workgroup is 256 workgroups at 2^20 -- one wave per SIMD, which is all that fits beside an accumulation
      // anyway; the first conversion of a blocking MSM has the chip to itself and takes a quarter of that per workgroup (four times
      // the flushes of the [W][G] counters: only where those are few, i.e. not the wide windows)
      const int per_wg = (alone && n < ((size_t)1 << 22) && fb == FINE_BITS) ? PREP_CH / 4 : PREP_CH;
      const dim3 grid((unsigned)((n + per_wg - 1) / per_wg));
      if (scalar_field == KG_FR) {
        if (hl > 48 * 1024) KG_HIP(ctx, hipFuncSetAttribute((const void*)k_prep_scalars_count<FrParams>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
        hipLaunchKernelGGL(k_prep_scalars_count<FrParams>, grid, dim3(PREP_NT), hl, st, d_scalars, n, H, kt, c, W, fb, G, nch, chunk_len, cnt, per_wg);
      } else {
        if (hl > 48 * 1024) KG_HIP(ctx, hipFuncSetAttribute((const void*)k_prep_scalars_count<FqParams>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
        hipLaunchKernelGGL(k_prep_scalars_count<FqParams>, grid, dim3(PREP_NT), hl, st, d_scalars, n, H, kt, c, W, fb, G, nch, chunk_len, cnt, per_wg);
      }
    } else if (scalar_field == KG_FR) hipLaunchKernelGGL(k_prep_scalars<FrParams>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, H, kt);
    else hipLaunchKernelGGL(k_prep_scalars<FqParams>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, H, kt);
    ph.end();
  }
  KG_HIP(ctx, hipGetLastError());
  return KG_OK;
}

// Sort of one window group: windows [w0, w0 + Wg) of the plan (all of them for the one-pass and the merged sort).  The
// per-window tables are window-major, so a group's view is a pointer offset; only k_group_scatter needs the absolute window
// (the digit's position in the scalar).
int msm_sort_group(kg_ctx* ctx, const MsmSortPlan& Q, int g, MsmSorted* S, bool on_main) {
  if (g < 0 || g >= Q.ngroups) return set_err(ctx, KG_ERR_BAD_ARG, "bad window group");
  KG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = Q.n, nv = Q.nv;
  const int c = Q.c, W = Q.W, B = Q.B, G = Q.G, nch = Q.nch, maxseg = Q.maxseg;
  const bool merged = Q.merged, two_pass = Q.two_pass;
  const int w0 = Q.gw0[g], Wg = Q.gW[g];              // bucket-space windows of the group (merged: the single set)
  const int sw0 = merged ? 0 : w0, sWg = merged ? W : Wg;      // scalar windows the group's first pass covers
  // hot buckets are cut 2^shift times finer (bucket_task_len; the shift travels in the top bits of the task length): 2 by default, 0 for
  // a merged sort.  KG_HOT_SHIFT: experiments.  Measured with shift 2 (MI355X): witness-like 2^20 MSM 1.17 -> 0.67 ms per step, proof from a
  // 0/1-heavy witness 2.48 -> 2.16 ms; with window tables (merged) 1.87 -> 2.03, hence 0 there.
  const int hot_shift_env = ctx->tune.hot_shift;
  const uint32_t hot_shift = hot_shift_env >= 0 && hot_shift_env <= 2 ? (uint32_t)hot_shift_env : (Q.merged ? 0u : 2u);
  const uint32_t T = Q.T | (hot_shift << 30);
  char* ws = Q.ws;
  hipStream_t st = on_main ? ctx->stream : ctx->sort_stream;
  const size_t npts = (size_t)Wg * B;
  uint32_t* kt = (uint32_t*)(ws + Q.o_kt);
  uint32_t* cnt = (uint32_t*)(ws + Q.o_cnt);
  uint32_t* rowtot = (uint32_t*)(ws + Q.o_rowtot) + w0;
  uint32_t* misc = (uint32_t*)(ws + Q.o_misc[g]);
  uint32_t* lenh = (uint32_t*)(ws + Q.o_lenh[g]);
  S->set = Q.set; S->ready = ctx->ev_sorted[Q.set][g];
  S->n = n; S->c = c; S->W = Wg; S->B = B; S->T = T; S->npts = npts; S->part_cap = Q.part_cap[g];
  const int gi = Q.info_base + g;                     // read-back words / event of this sort
  if (gi >= kg_ctx::MAX_GROUPS) return set_err(ctx, KG_ERR_BAD_ARG, "bad read-back index");
  S->merged_shift = Q.mshift; S->windows = W; S->w0 = w0; S->group = gi; S->acc_stream = nullptr; S->sorted_on = st;
  S->sorted = (uint32_t*)(ws + Q.o_sorted) + (size_t)w0 * n; S->bsize = (uint32_t*)(ws + Q.o_bsize) + (size_t)w0 * B;
  S->bstart = (uint32_t*)(ws + Q.o_bstart) + (size_t)w0 * B;
  S->lcnt = (uint32_t*)(ws + Q.o_lcnt) + (size_t)w0 * B; S->lrel = (uint32_t*)(ws + Q.o_lrel) + (size_t)w0 * B; S->lbase = (uint32_t*)(ws + Q.o_lbase[g]);
  S->task_bkt = (uint32_t*)(ws + Q.o_tbkt[g]); S->task_id = (uint32_t*)(ws + Q.o_tid[g]);
  S->hot_list = (uint32_t*)(ws + Q.o_hot[g]);
  S->T_top = Q.T_top | (hot_shift << 30);
  S->top_w = (!merged && w0 + Wg == W) ? Wg - 1 : -1;       // the unsigned top window, if this group holds it
  const uint32_t T_top = S->T_top;
  const int top_w = S->top_w;
  {
    PhaseScope ph(ctx, "sort", st);
    const size_t lds = (size_t)(two_pass ? G : B) * 4;
    if (lds > 48 * 1024) {      // the whole-window histogram needs more than the default dynamic LDS limit
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int fb = Q.fb;
    const uint32_t FINE = 1u << fb;
    uint32_t* tmp = (uint32_t*)(ws + Q.o_tmp);                // uint32_t entries (FB = 7) or uint64_t (FB = 9)
    uint64_t* tmp8 = (uint64_t*)(ws + Q.o_tmp);
    uint32_t* gsize = (uint32_t*)(ws + Q.o_gsize);
    uint32_t* gstart = (uint32_t*)(ws + Q.o_gstart);
    uint32_t* segbase = (uint32_t*)(ws + Q.o_segbase);
    uint32_t* segcnt = (uint32_t*)(ws + Q.o_segcnt) + (size_t)w0 * maxseg * FINE;
    uint32_t* segoff = (uint32_t*)(ws + Q.o_segoff) + (size_t)w0 * maxseg * FINE;
    const size_t zbytes = (Q.o_lenh[g] - Q.o_misc[g]) + 2 * LEN_BINS * 4;     // misc and the length histogram: cleared by k_group_scan, or
    if (!two_pass) zero_fill(st, misc, zbytes);
    uint32_t* woff = merged ? (uint32_t*)(ws + Q.o_woff) : nullptr;
    // the tables the fine pass and the task decomposition read: per window (the group's rows), or the merged single set
    const uint32_t* f_gstart = merged ? (uint32_t*)(ws + Q.o_gstart_m) : gstart + (size_t)w0 * G;
    const uint32_t* f_gsize = merged ? (uint32_t*)(ws + Q.o_gsize_m) : gsize + (size_t)w0 * G;
    const uint32_t* f_segbase = merged ? (uint32_t*)(ws + Q.o_segbase_m) : segbase + (size_t)w0 * (G + 1);
    const uint32_t* f_tmp = merged ? tmp : tmp + (size_t)w0 * n;
    const uint64_t* f_tmp8 = tmp8 + (size_t)w0 * n;
    // first pass on big tiles (k_group_scatter_big): 8192 entries with the wide windows' 1024 groups, 4096 otherwise; 1024 threads
    // where nothing else runs (the first window group of a blocking MSM), 256 beside an accumulation.  Measured (MI355X, round 4):
    // 2^24-pair commitment 20.32 -> 19.00 ms, blocking 2^20 1.746 -> 1.685 ms, the four-deep 2^20 step 1.32 -> 1.295 ms; 512 threads
    // beside the accumulation shorten the sorts (12.9 -> 9.8 ms summed at 2^24) and lengthen the accumulations by as much.
    const int gs_tile_env = ctx->tune.gs_tile;     // experiments: 0 = the 1024-entry tiles of k_group_scatter
    const int gs_nt_set = ctx->tune.gs_nt;
    // wide windows: 512 -- level on uniform scalars (2^24: 19.05 / 19.0 ms), and a witness-like 2^24-pair vector, whose accumulations are
    // short and whose sorts therefore run mostly alone, 10.65 -> 8.9 ms; below, 512 costs the four-deep 2^20 step 1 %
    const int gs_nt_env = gs_nt_set ? gs_nt_set : (fb == 9 ? 512 : 256);
    const int gs_nt0_env = ctx->tune.gs_nt0;
#ifdef KG_EXPERIMENTS
    const int gs_tile = gs_tile_env >= 0 ? gs_tile_env : (fb == 9 ? 8192 : 4096);
#else
    const int gs_tile = gs_tile_env > 0 ? gs_tile_env : (fb == 9 ? 8192 : 4096);      // (KG_GS_TILE=0, the 1024-entry kernel, exists in -DKG_EXPERIMENTS builds only)
#endif
    const int gs_nt = (Q.alone && g == 0) ? gs_nt0_env : gs_nt_env;
    if (two_pass) {
      hipLaunchKernelGGL(k_group_scan, dim3(sWg), dim3(G > 256 ? 512 : GS_NT), 0, st, cnt + (size_t)sw0 * nch * G, nch, G, B, gsize + (size_t)sw0 * G, gstart + (size_t)sw0 * G,
                         segbase + (size_t)sw0 * (G + 1), (uint32_t*)(ws + Q.o_bsize) + (size_t)sw0 * B, misc, (int)(zbytes / 4), seg_len_for(fb));
      if (merged)
        hipLaunchKernelGGL(k_merge_groups, dim3(1), dim3(GS_NT), 0, st, gsize, W, G, woff, (uint32_t*)(ws + Q.o_gsize_m), (uint32_t*)(ws + Q.o_gstart_m),
                           (uint32_t*)(ws + Q.o_segbase_m));
      if (fb == 9) {
#ifdef KG_EXPERIMENTS
        if (!gs_tile) hipLaunchKernelGGL(k_group_scatter<9>, dim3(sWg, nch), dim3(GS_NT), 0, st, kt, n, c, W, Q.chunk_len, G, cnt, gstart, tmp8, woff, Q.mshift, sw0);
        else
#endif
        KG_HIP(ctx, launch_gs_big_any<9>(gs_tile, gs_nt, dim3(sWg, nch), st, kt, n, c, W, Q.chunk_len, G, cnt, gstart, tmp8, woff, Q.mshift, sw0));
        KG_HIP(ctx, hipFuncSetAttribute((const void*)k_fine_local<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_local_lds<9>()));
        hipLaunchKernelGGL(k_fine_local<9>, dim3(Wg, maxseg), dim3(512), fine_local_lds<9>(), st, f_tmp8, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bsize, segcnt, segoff, S->sorted);
      } else {
#ifdef KG_EXPERIMENTS
        if (!gs_tile) hipLaunchKernelGGL(k_group_scatter<7>, dim3(sWg, nch), dim3(GS_NT), 0, st, kt, n, c, W, Q.chunk_len, G, cnt, merged ? f_gstart : gstart, tmp, woff, Q.mshift, sw0);
        else
#endif
        KG_HIP(ctx, launch_gs_big_any<7>(gs_tile, gs_nt, dim3(sWg, nch), st, kt, n, c, W, Q.chunk_len, G, cnt, merged ? f_gstart : gstart, tmp, woff, Q.mshift, sw0));
        hipLaunchKernelGGL(k_fine_local<7>, dim3(Wg, maxseg), dim3(512), fine_local_lds<7>(), st, f_tmp, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bsize, segcnt, segoff, S->sorted);
      }
    } else {
      hipLaunchKernelGGL(k_count, dim3(W, nch), dim3(1024), lds, st, kt, n, c, W, Q.chunk_len, 0, cnt);
      hipLaunchKernelGGL(k_scan_chunks, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, cnt, W, nch, B, S->bsize);
    }
    // task decomposition (needs only the bucket sizes); its two result words travel to the host while the
    // scatter below still runs, so the read-back does not stall the queue
    const unsigned g1024 = (unsigned)((npts + 1023) / 1024);
    if (B >= 8192) {
      const int nsplit = B / 4096;                    // <= 16 parts per row
      uint32_t* bpart = (uint32_t*)(ws + Q.o_bpart) + (size_t)w0 * nsplit * 2;
      hipLaunchKernelGGL(k_bucket_part, dim3(Wg, nsplit), dim3(BR_NT), 0, st, S->bsize, B, T, nsplit, bpart, misc, lenh, T_top, top_w, S->hot_list, HOT_MAX);
      hipLaunchKernelGGL(k_bucket_fill, dim3(Wg, nsplit), dim3(BR_NT), 0, st, S->bsize, B, T, nsplit, bpart, S->bstart, S->lcnt, S->lrel, rowtot, T_top, top_w);
    } else
      hipLaunchKernelGGL(k_bucket_rows, dim3(Wg), dim3(BR_NT), 0, st, S->bsize, B, T, S->bstart, S->lcnt, S->lrel, rowtot, misc, lenh, T_top, top_w, S->hot_list, HOT_MAX);
    hipLaunchKernelGGL(k_task_bases, dim3(1), dim3(64), 0, st, rowtot, Wg, S->lbase, misc, misc + 4, lenh, lenh + LEN_BINS, (uint32_t*)ctx->h_pinned_dev + 4 * gi);
    KG_HIP(ctx, hipEventRecord(ctx->ev_info[gi], st));
    hipLaunchKernelGGL(k_len_scatter, dim3(g1024), dim3(1024), 0, st, S->bsize, S->lcnt, S->lrel, S->lbase, npts, B, T, lenh + LEN_BINS, S->task_bkt, S->task_id, T_top, top_w);
    // rows of segment walkers per window; the top window (top_w >= 0: this group holds it, the sort is not merged) gets three more sets
    const int fs_rows = maxseg < FS_ROWS ? maxseg : FS_ROWS;
    const int fs_extra = top_w >= 0 ? 3 : 0;
    if (two_pass && fb == 9) {
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_fine_scatter<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_scatter_lds<9>()));
      hipLaunchKernelGGL(k_fine_scatter<9>, dim3(Wg + fs_extra, fs_rows), dim3(512), fine_scatter_lds<9>(), st, f_tmp8, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bstart, segcnt, segoff, S->sorted, Wg, fs_extra ? top_w : -1);
    } else if (two_pass) {
      KG_HIP(ctx, hipFuncSetAttribute((const void*)k_fine_scatter<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_scatter_lds<7>()));
      hipLaunchKernelGGL(k_fine_scatter<7>, dim3(Wg + fs_extra, fs_rows), dim3(512), fine_scatter_lds<7>(), st, f_tmp, n, G, B, maxseg, f_gstart, f_gsize, f_segbase, S->bstart, segcnt, segoff, S->sorted, Wg, fs_extra ? top_w : -1);
    } else
      hipLaunchKernelGGL(k_scatter, dim3(W, nch), dim3(1024), lds, st, kt, n, c, W, Q.chunk_len, 0, cnt, S->bstart, S->sorted);
    ph.end();
    KG_HIP(ctx, hipGetLastError());
    KG_HIP(ctx, hipEventRecord(S->ready, st));
    host_trace("sort: enqueued");
  }
  (void)nv;
  return KG_OK;
}

int msm_sort(kg_ctx* ctx, int scalar_field, const uint64_t* d_scalars, size_t n, MsmSorted* S, bool ordered, int merged_c, int lane_mult, bool wait_info, int info_idx) {
  MsmSortPlan Q;
  KG_TRY(msm_sort_begin(ctx, scalar_field, d_scalars, n, &Q, ordered, merged_c, lane_mult));
  Q.info_base = info_idx;
  KG_TRY(msm_sort_group(ctx, Q, 0, S));
  return wait_info ? msm_sort_wait(ctx, S) : KG_OK;
}

// The task count and the largest bucket of a sort (two words read back through pinned memory, one pair per window group):
// the caller may put other work on the queues between msm_sort(..., wait_info = false) and this, but no other sort with the
// same group index.
int msm_sort_wait(kg_ctx* ctx, MsmSorted* S) {
  KG_HIP(ctx, hipEventSynchronize(ctx->ev_info[S->group]));
  host_trace("sort: info back");
  const uint32_t* h_info = (const uint32_t*)ctx->h_pinned + 4 * S->group;
  S->ntasks = h_info[0];
  S->max_cnt = h_info[3];                              // most tasks any bucket has
  S->nhot = h_info[2];                                 // buckets with more than GATHER_SUM_MAX tasks (listed up to HOT_MAX)
  if (S->ntasks > S->part_cap) return set_err(ctx, KG_ERR_HIP, "task count exceeds its bound");
  return KG_OK;
}

}  // namespace kg
