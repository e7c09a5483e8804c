// msm_small.hip -- enqueue of the short-input MSM (msm_small_kernels.h): the plan (window width and bucket ranges by length) and the one or
// two launches that leave the W window sums in the result slot's pinned buffer.  Replaces, for n <= 2^12 pairs, the launch chain of
// msm_sort.hip + msm_run.hip behind kg_msm / kg_msm_begin / kg_commit / the prover (groth16/src/msm.rs:6-48 at the lengths of its own tests).
#include "msm_small_kernels.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

using namespace kg;
using namespace kg::msm;

namespace kg {

// Does the short-input path take an n-pair MSM, and with which shape?  c: window width; r: log2 of the buckets a workgroup owns
// (2^(c-1-r) workgroups per window).  A forced window (kg_msm_set_window) keeps the long pipeline: the parity tests walk its widths.
bool msm_small_plan(const kg_ctx* ctx, int curve, size_t n, int* c_out, int* r_out) {
  const kg_tuning& tn = ctx ? ctx->tune : tuning();
  if (tn.small_max <= 0 || n == 0 || n > (size_t)tn.small_max || n > SM_MAX_N_KT) return false;
  if (ctx && ctx->msm_window) return false;
  const bool kt = msm_small_kt(ctx, n);
  // Measured on MI355X (profiles/r06_small_shapes*.txt: blocking kg_msm over every width and range).  With the tree levels run by lane quads
  // (coop_add.h: ~4.5 us per level) the widths 2 .. 5 are within a few per cent of each other at every length: a narrow window is a short
  // reduction on the device and more additions on the host (255 doublings + one addition per window).  Up to 1536 pairs the two-bit
  // window -- 128 workgroups of two buckets, ONE launch; beyond, a window's sixteen buckets go to four or eight workgroups (their CUs
  // are idle otherwise) and a second launch adds the ranges' planes.  G2 (three times the arithmetic per addition on the device, the
  // host chain in Fq2): wider windows pay earlier.
  int c, r;
  const bool glv = msm_small_glv(ctx, curve, n);
  if (glv) {
    // two 127-bit halves per scalar (msm_digits.h): widths that divide 128 -- the top window of any other width holds one or two bits, i.e. two
    // or four buckets with a quarter of all entries each (c = 5 at 4096 pairs: 0.38 ms against 0.19).  profiles/r06_small_shapes*_glv.txt
    if (curve == KG_G2) {
      if (n <= 48) { c = 2; r = 1; }
      else if (n <= 160) { c = 2; r = 0; }
      else if (n <= 2048) { c = 4; r = 1; }
      else { c = 8; r = 3; }
    } else {
      if (n <= 384) { c = 2; r = 1; }
      else if (n <= 2048) { c = 4; r = 1; }
      else { c = 8; r = 3; }
    }
  } else
  if (curve == KG_G2) {
    if (n > (size_t)SM_MAX_N_G2) return false;       // (2^15 G2 pairs: 1.18 ms here, 1.09 ms through the long pipeline)
    if (n <= 32) { c = 3; r = 2; }
    else if (n <= 384) { c = 4; r = 3; }
    else { c = 5; r = 2; }
  } else {
    if (n <= 1536) { c = 2; r = 1; }
    else if (n <= 6144) { c = 5; r = 2; }
    else if (n <= 28672) { c = 8; r = 4; }           // 32 windows x 8 workgroups of sixteen buckets (interleaved): fewer passes over the converted scalars
    else { c = 8; r = 3; }
  }
  if (tn.small_c >= 2 && tn.small_c <= 10) { c = tn.small_c; r = c - 1 < SM_MAX_R ? c - 1 : SM_MAX_R; }
  if (tn.small_r >= 0 && tn.small_r <= SM_MAX_R) r = tn.small_r;
  if (r > c - 1) r = c - 1;
  if (c - 1 - r > 5) r = c - 1 - 5;                  // at most 32 workgroups per window
  if (r > SM_MAX_R) return false;
  const uint32_t nv = glv ? 2 * (uint32_t)n : (uint32_t)n;        // list entries per window
  auto lds = [&](int rr) { return curve == KG_G2 ? small_lds_bytes<Fq2>(nv, rr, kt) : small_lds_bytes<Fq>(nv, rr, kt); };
  while (r > 0 && lds(r) > 160 * 1024) --r;          // G2 points are twice the words: smaller bucket ranges, more workgroups per window
  if (lds(r) > 160 * 1024 || c - 1 - r > 5) return false;
  *c_out = c; *r_out = r;
  return true;
}

// GLV (msm_digits.h): every scalar as two 127-bit halves k1 + k2 lambda against P and (beta x, y) -- half the windows on the device, half the
// doublings of the host chain.  The list entries hold a 15-bit index of the halves: up to 16384 pairs.
bool msm_small_glv(const kg_ctx* ctx, int curve, size_t n) {
  const kg_tuning& tn = ctx ? ctx->tune : tuning();
  if (tn.small_glv == 0 || 2 * n > SM_MAX_N_KT || (ctx && ((ctx->small_glv_off >> curve) & 1u))) return false;
  // where it pays (blocking kg_msm, same box): G1 / Grumpkin up to 6144 pairs (16 pairs 0.114 -> 0.095 ms, 2^12 0.198 -> 0.186, 6144 0.217 -> 0.209;
  // 2^13 0.224 -> 0.232), G2 -- whose host chain is Fq2 and whose tree levels cost three times a G1 level -- at every length the entries'
  // index field holds (32 pairs 0.32 -> 0.25 ms, 2^13 0.64 -> 0.52, 2^14 0.86 -> 0.62)
  return tn.small_glv == 2 || curve == KG_G2 || n <= 6144;
}

// From which length the scalars are converted once by a launch of their own (the KT form of the kernel) instead of by every workgroup
bool msm_small_kt(const kg_ctx* ctx, size_t n) {
  const kg_tuning& tn = ctx ? ctx->tune : tuning();
  return n > (size_t)(tn.small_kt_from > 0 ? tn.small_kt_from : 0) || n > SM_MAX_N;
}

template <class F, class SP>
static int small_launch(kg_ctx* ctx, hipStream_t st, const SmallArgs& a, size_t lds, size_t lds2) {
  static std::atomic<uint64_t> attr_devs{0};          // once per DEVICE and instance (the attribute belongs to the function on a device): the whole 160 KiB of LDS
  const uint64_t bit = (uint64_t)1 << (ctx->device & 63);
  if (!(attr_devs.load() & bit)) {
    KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_msm_small<F, SP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_msm_small<F, SP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    KG_HIP(ctx, hipFuncSetAttribute((const void*)(k_msm_small_combine<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_devs |= bit;
  }
  if (a.kt) {
    hipLaunchKernelGGL((k_small_prep<SP>), dim3((a.nr + 255) / 256), dim3(256), 0, st, a.scalars, a.nr, a.H, const_cast<uint32_t*>(a.kt), a.spill_cursor, a.W, a.glv, a.inf,
                       const_cast<uint8_t*>(a.meta));
    hipLaunchKernelGGL((k_msm_small<F, SP, true>), dim3((unsigned)a.W, (unsigned)a.NB), dim3(SM_NT), lds, st, a);
  } else
  hipLaunchKernelGGL((k_msm_small<F, SP, false>), dim3((unsigned)a.W, (unsigned)a.NB), dim3(SM_NT), lds, st, a);
  if (a.NB > 1) hipLaunchKernelGGL((k_msm_small_combine<F>), dim3((unsigned)a.W), dim3(SM_NT), lds2, st, a);
  KG_HIP(ctx, hipGetLastError());
  return KG_OK;
}

// The whole MSM on queue `st` (ordered behind whatever produced the inputs there); the W window sums land in the slot's pinned buffer
// and the slot's event is recorded -- msm_finish(ctx, curve, slot, out) is the host half, as for the long pipeline.
int msm_small_enqueue(kg_ctx* ctx, hipStream_t st, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, int slot,
                      int c, int r) {
  if (slot < 0 || slot >= kg_ctx::NSLOTS) return set_err(ctx, KG_ERR_BAD_ARG, "bad result slot");
  KG_HIP(ctx, hipSetDevice(ctx->device));
  const bool glv = msm_small_glv(ctx, curve, n);
  const int W = ((glv ? 128 : 255) + c - 1) / c, NB = 1 << (c - 1 - r), E64 = curve == KG_G2 ? 8 : 4;     // glv: |k1|, |k2| < 2^127
  const size_t nv = glv ? 2 * n : n;
  const int NW = curve == KG_G2 ? PointIO<Fq2>::NW : PointIO<Fq>::NW;
  KG_TRY(ensure_slot(ctx, slot, (size_t)W * 4 * E64 * 8));
  kg_ctx::Slot& sl = ctx->slots[slot];
  SmallArgs a;
  a.bases = d_bases; a.inf = d_inf; a.scalars = d_scalars; a.n = (uint32_t)nv; a.nr = (uint32_t)n; a.glv = glv ? 1 : 0; a.npl = glv ? 4 : 8; a.meta = nullptr;
  a.c = c; a.W = W; a.r = r; a.NB = NB;
  uint32_t H[8];
  small_bias(c, W, H);
  for (int j = 0; j < 8; ++j) a.H.w[j] = H[j];
  a.out = (uint64_t*)sl.host_dev;
  a.planes = nullptr;
  a.kt = nullptr; a.spill = nullptr; a.spill_cursor = nullptr;
  const bool kt = msm_small_kt(ctx, n);
#ifdef KG_EXPERIMENTS
  static uint64_t* stamps = nullptr;                  // KG_SMALL_STAMPS=1: the previous call's phase boundaries (workgroup (0, 0)) on stderr
  const bool want_stamps = ctx->tune.small_stamps != 0;
  a.stamps = nullptr;
  if (want_stamps) {
    if (!stamps) { if (hipHostMalloc((void**)&stamps, 64 * 8, hipHostMallocDefault) != hipSuccess) stamps = nullptr; else memset(stamps, 0, 64 * 8); }
    if (stamps && stamps[7]) {
      fprintf(stderr, "[small] digits %.1f  scan+scatter %.1f  accumulate %.1f  merge %.1f  halve %.1f  (planes) %.1f  combine %.1f  export %.1f us\n",
              (stamps[1] - stamps[0]) * 0.01, (stamps[2] - stamps[1]) * 0.01, (stamps[3] - stamps[2]) * 0.01, (stamps[4] - stamps[3]) * 0.01,
              (stamps[5] - stamps[4]) * 0.01, 0.0, (stamps[6] - stamps[5]) * 0.01, (stamps[7] - stamps[6]) * 0.01);
      memset(stamps, 0, 64 * 8);
    }
    a.stamps = stamps;
  }
#endif
  size_t lds2 = 0;
  if (NB > 1 || kt) {
    // scratch of the slot: plane points of split windows | the scalars' word planes | spill cursors | spill space of the lists
    const size_t b_planes = ((size_t)W * NB * (SM_MAX_R + 1) * NW * 4 + 255) & ~(size_t)255;
    const size_t b_kt = kt ? (nv * 4 * (glv ? 4 : 8) + 255) & ~(size_t)255 : 0, b_cur = kt ? 512 : 0, b_spill = kt ? ((size_t)W * nv * 2 + 255) & ~(size_t)255 : 0;
    const size_t b_meta = (kt && glv) ? (nv + 255) & ~(size_t)255 : 0;
    const size_t bytes = b_planes + b_kt + b_cur + b_spill + b_meta;
    if (bytes > ctx->ws_small_bytes[slot]) {
      if (ctx->ws_small[slot]) { sync_all(ctx); (void)hipFree(ctx->ws_small[slot]); ctx->ws_small[slot] = nullptr; ctx->ws_small_bytes[slot] = 0; }
      const hipError_t e = dev_alloc(ctx, &ctx->ws_small[slot], bytes);
      if (e != hipSuccess) return set_err(ctx, KG_ERR_OOM, "short-input plane buffer", e);
      ctx->ws_small_bytes[slot] = bytes;
    }
    char* ws = (char*)ctx->ws_small[slot];
    a.planes = (uint32_t*)ws;
    if (kt) { a.kt = (const uint32_t*)(ws + b_planes); a.spill_cursor = (uint32_t*)(ws + b_planes + b_kt); a.spill = (uint16_t*)(ws + b_planes + b_kt + b_cur); }
    if (kt && glv) a.meta = (const uint8_t*)(ws + b_planes + b_kt + b_cur + b_spill);
    if (NB > 1) lds2 = curve == KG_G2 ? small_combine_lds_bytes<Fq2>(c, NB) : small_combine_lds_bytes<Fq>(c, NB);
  }
  PhaseScope ph(ctx, "small_msm", st);
  int rc;
  if (curve == KG_G1) rc = small_launch<Fq, FrParams>(ctx, st, a, small_lds_bytes<Fq>(a.n, r, kt), lds2);
  else if (curve == KG_GRUMPKIN) rc = small_launch<Fr, FqParams>(ctx, st, a, small_lds_bytes<Fr>(a.n, r, kt), lds2);
  else rc = small_launch<Fq2, FrParams>(ctx, st, a, small_lds_bytes<Fq2>(a.n, r, kt), lds2);
  ph.end();
  KG_TRY(rc);
  KG_HIP(ctx, hipEventRecord(sl.done, st));
  sl.W = W; sl.c = c; sl.w0 = 0; sl.combined = true; sl.busy = false;
  return KG_OK;
}

}  // namespace kg
