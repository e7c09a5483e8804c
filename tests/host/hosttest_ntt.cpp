// hosttest_ntt.cpp -- runs the NTT kernel's per-thread bodies (csrc/ntt_tile.h) on the host, thread by thread and pass by
// pass, with the plain field type and with the bound-tracking FrC: whole transforms (the same plans, step arguments and
// table formulas the device uses) are compared with the oracle by tests/test_host_ntt.py, and FrC proves that no column,
// limb or value bound of fp29.h can overflow in any pass.  Also counts LDS bank conflicts of every pass.  Test
// infrastructure only.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../../kogarashi_amd/csrc/fp29.h"
#include "../../kogarashi_amd/csrc/fp29_checked.h"
#include "../../kogarashi_amd/csrc/ntt_tile.h"

namespace kg {
template <class P> struct NttIO<FpChecked<P>> {
  static FpChecked<P> raw(const uint32_t w[8]) { return FpChecked<P>::wrap(limbs_from_words<P>(w), 5.4); }       // any 256-bit value
  static FpChecked<P> table(const uint32_t* tab, size_t e) { return FpChecked<P>::wrap(NttIO<Fp<P>>::table(tab, e), 2.0); }
  static FpConst<P> twc(const uint32_t* tab, size_t e) { return NttIO<Fp<P>>::twc(tab, e); }
  static void words(const FpChecked<P>& a, uint32_t w[8]) {
    a.check_actual();
    if (!(a.lb <= (double)M29 && a.kb <= 5.4)) BoundFail::fail("store of a value that is not a normalised 256-bit integer", a.lb, a.kb);
    words_from_limbs(a.v, w);
  }
};
}  // namespace kg
using namespace kg;

// tile store of the emulation: one F per LDS word, plus the bank-conflict bookkeeping of the current wave instruction
static long g_lds_cycles = 0, g_lds_ideal = 0;
static long g_private_passes = 0, g_barriers = 0;
template <class F>
struct HostStore {
  F* arr;
  mutable std::vector<uint32_t>* trace;     // LDS words touched by the current thread, in program order
  int block;                                // >= 0: this pass is wave-private -- every access must stay in the wave's block (word >> 8 = wave id)
  void check(uint32_t w) const {
    if (block >= 0 && (int)(w >> 8) != block) { std::fprintf(stderr, "wave-private pass touches LDS word %u outside block %d\n", w, block); std::abort(); }
  }
  void store(uint32_t w, const F& a) const { check(w); arr[w] = a; trace->push_back(w); }
  template <class G> G load(uint32_t w) const { check(w); trace->push_back(w); return arr[w]; }
};
// accesses[t] = words thread t touched in one pass (same count for every active thread): instruction i of a wave is the
// i-th access of its 64 lanes; ds_read_b32 / ds_write_b32 cost per 32-lane half = max distinct words on one bank (mod 32)
static void count_conflicts(const std::vector<std::vector<uint32_t>>& acc) {
  const size_t nt = acc.size();
  for (size_t w0 = 0; w0 < nt; w0 += 32) {
    size_t ninstr = 0;
    for (size_t t = w0; t < w0 + 32 && t < nt; ++t) ninstr = acc[t].size() > ninstr ? acc[t].size() : ninstr;
    for (size_t i = 0; i < ninstr; ++i) {
      std::vector<uint32_t> bank[32];
      bool any = false;
      for (size_t t = w0; t < w0 + 32 && t < nt; ++t) {
        if (i >= acc[t].size()) continue;
        any = true;
        auto& b = bank[acc[t][i] & 31u];
        bool dup = false;
        for (uint32_t x : b) dup |= x == acc[t][i];
        if (!dup) b.push_back(acc[t][i]);
      }
      if (!any) continue;
      size_t worst = 1;
      for (auto& b : bank) worst = b.size() > worst ? b.size() : worst;
      g_lds_cycles += (long)worst;
      g_lds_ideal += 1;
    }
  }
}

template <class F, int LOG_M, int LOG_TC, bool ROW>
static void run_tiles(const NttStepArgs& A, uint32_t ntiles) {
  using T = NttTile<F, LOG_M, LOG_TC, ROW>;
  std::vector<F> lds((size_t)T::ELEMS * 2);          // tile_phys never leaves [0, ELEMS)
  for (uint32_t blk = 0; blk < ntiles; ++blk) {
    uint32_t tile = blk;
    if (A.tile_shift) tile = ((tile & 7u) << A.tile_shift) | (tile >> 3);
    const T t{A, tile};
    std::vector<std::vector<uint32_t>> acc(T::NT);
    auto pass = [&](bool wave_private, auto body) {
      for (auto& a : acc) a.clear();
      for (uint32_t tid = 0; tid < (uint32_t)T::NT; ++tid) {
        HostStore<F> st{lds.data(), &acc[tid], wave_private ? (int)(tid >> 6) : -1};
        body(tid, st);
      }
      if (blk == 0) { count_conflicts(acc); g_private_passes += wave_private; }
    };
    // the kernel's synchronisation plan, replayed: a wave-private pass may only touch its wave's block; a boundary between two
    // wave-private passes has no workgroup barrier (the emulation runs pass by pass, so what it checks is the block discipline
    // that makes the missing barrier safe)
    pass(T::FIRST_PRIVATE, [&](uint32_t tid, const HostStore<F>& st) { t.first(tid, st); });
    if constexpr (!T::SINGLE) {
      t.after_first([&] { if (blk == 0) ++g_barriers; }, [] {});
      auto mids = [&](auto self, auto s0c) -> void {
        constexpr int S0 = decltype(s0c)::value;
        if constexpr (S0 < T::S_LAST) {
          pass(T::template mid_private<S0>(), [&](uint32_t tid, const HostStore<F>& st) { t.template mid<S0>(tid, st); });
          if constexpr (!(S0 + 2 < T::S_LAST && T::template mid_private<S0>() && T::template mid_private<S0 + 2>())) { if (blk == 0) ++g_barriers; }
          self(self, std::integral_constant<int, S0 + 2>{});
        }
      };
      mids(mids, std::integral_constant<int, T::G0>{});
      pass(false, [&](uint32_t tid, const HostStore<F>& st) { t.last(tid, st); });
    }
  }
}

#ifdef KG_NTT_HOST_FEW      // sanitizer build (tests/test_sanitizers.py): a few shapes keep the compile short
#define HT_SHAPES(X) X(6, 4) X(7, 3)
#define HT_COL_ONLY(X)
#define HT_ROW_ONLY(X) X(5, 0) X(11, 0)
#else                       // every shape that exists as a kernel (ntt_tile.h)
#define HT_SHAPES(X) KG_NTT_SHAPES(X)
#define HT_COL_ONLY(X) KG_NTT_SHAPES_COL_ONLY(X)
#define HT_ROW_ONLY(X) KG_NTT_SHAPES_ROW_ONLY(X)
#endif
template <class F>
static int run_step(const NttStepDesc& d, const NttStepArgs& a, uint32_t ntiles) {
  const int key = d.log_m * 16 + d.log_tc;
  if (!d.row) {
    switch (key) {
#define X(m, tc) case (m) * 16 + (tc): run_tiles<F, m, tc, false>(a, ntiles); return 0;
      HT_SHAPES(X) HT_COL_ONLY(X)
#undef X
    }
  } else {
    switch (key) {
#define X(m, tc) case (m) * 16 + (tc): run_tiles<F, m, tc, true>(a, ntiles); return 0;
      HT_SHAPES(X) HT_ROW_ONLY(X)
#undef X
    }
  }
  return -1;
}

static void st_tw(std::vector<uint32_t>& tab, size_t e, const Fr& a) { for (int k = 0; k < 9; ++k) tab[e * 9 + k] = a.l[k]; }

extern "C" {
// data: n = 2^log_n elements in the ABI form, transformed in place like kg_ntt_bn254_fr.  steps: 0 automatic, 3 forces
// three steps (ntt_plan).  checked: run with FrC.  Returns 0, or -1 for a shape the dispatch does not know.
int ht_ntt_tile(int checked, uint32_t log_n, int steps, int tile, int inverse, int coset, uint64_t* data, long* lds_cycles, long* lds_ideal) {
  const size_t n = (size_t)1 << log_n;
  NttStepDesc d[3];
  const int nsteps = ntt_plan(log_n, steps, d, tile);
  // tables: the formulas of k_build_table / k_build_direct / (ntt.hip)
  const uint32_t lo_bits = (log_n + 1) / 2, n_lo = 1u << lo_bits, n_hi = 1u << (log_n - lo_bits);
  std::vector<uint32_t> small(18u << (NTT_TW_LOG - 1)), lo(9u * n_lo), hi(9u * n_hi), cos_lo(9u * n_lo), cos_hi(9u * n_hi), d0, d1;
  auto fill_pow = [&](std::vector<uint32_t>& tab, size_t cnt, Fr start, Fr ratio) {
    Fr v = start;
    for (size_t e = 0; e < cnt; ++e) { st_tw(tab, e, mul(v, Fr::one())); v = mul(v, ratio); }     // mul by one: normalised, < 2p like the device entries
  };
  const Fr wn = ntt_root_of<Fr>(log_n, inverse);
  {                                             // in-tile twiddles as Shoup-form constants, like k_build_table kind 0
    Fr raw_one = Fr::zero();
    raw_one.l[0] = 1;
    Fr v = Fr::one();
    const Fr ratio = ntt_root_of<Fr>(NTT_TW_LOG, inverse);
    for (size_t e = 0; e < ((size_t)1 << (NTT_TW_LOG - 1)); ++e) {
      const FpConst<FrParams> c = make_const(reduce_2p(mul(v, raw_one)));
      for (int k = 0; k < 9; ++k) { small[e * 18 + k] = c.w[k]; small[e * 18 + 9 + k] = c.q[k]; }
      v = mul(v, ratio);
    }
  }
  fill_pow(lo, n_lo, Fr::one(), wn);
  fill_pow(hi, n_hi, Fr::one(), ntt_pow<Fr>(wn, (uint64_t)1 << lo_bits));
  const Fr g = Fr::from_const(inverse ? FrParams::GEN7_INV : FrParams::GEN7);
  Fr c0 = Fr::one();
  if (inverse) {
    uint32_t w[8] = {log_n < 32 ? (1u << log_n) : 0u, 0, 0, 0, 0, 0, 0, 0};
    c0 = inv(from_int<FrParams>(w));
  }
  fill_pow(cos_lo, n_lo, c0, g);
  fill_pow(cos_hi, n_hi, Fr::one(), ntt_pow<Fr>(g, (uint64_t)1 << lo_bits));
  auto fill_direct = [&](std::vector<uint32_t>& tab, uint32_t log_m, uint32_t log_inner, uint64_t mult) {
    const size_t inner = (size_t)1 << log_inner;
    tab.resize(9u * ((size_t)1 << (log_m + log_inner)));
    for (size_t r = 0; r < ((size_t)1 << log_m); ++r) {
      const Fr ratio = ntt_pow<Fr>(wn, r * mult);
      Fr v = Fr::one();
      for (size_t c = 0; c < inner; ++c) { st_tw(tab, r * inner + c, mul(v, Fr::one())); v = mul(v, ratio); }
    }
  };
  if (nsteps >= 2 && log_n <= 22) fill_direct(d0, (uint32_t)d[0].log_m, log_n - (uint32_t)d[0].log_m, 1);
  if (nsteps == 3) fill_direct(d1, (uint32_t)d[1].log_m, (uint32_t)d[2].log_m, (uint64_t)1 << d[0].log_m);
  const NttTables tabs{small.data(), lo.data(), hi.data(), cos_lo.data(), cos_hi.data(), d0.empty() ? nullptr : d0.data(),
                       d1.empty() ? nullptr : d1.data(), lo_bits};
  std::vector<uint64_t> tmp(4 * n);
  g_lds_cycles = g_lds_ideal = 0;
  g_private_passes = g_barriers = 0;
  for (int i = 0; i < nsteps; ++i) {
    NttStepArgs a;
    const uint32_t ntiles = ntt_step_args(log_n, nsteps, d, i, tabs, data, tmp.data(), inverse, coset, a);
    const int rc = checked ? run_step<FrC>(d[i], a, ntiles) : run_step<Fr>(d[i], a, ntiles);
    if (rc) return rc;
  }
  if (lds_cycles) *lds_cycles = g_lds_cycles;
  if (lds_ideal) *lds_ideal = g_lds_ideal;
  return 0;
}
int ht_ntt(int checked, uint32_t log_n, int steps, int inverse, int coset, uint64_t* data, long* lds_cycles, long* lds_ideal) {
  return ht_ntt_tile(checked, log_n, steps, 0, inverse, coset, data, lds_cycles, lds_ideal);
}
// wave-private passes and workgroup barriers of the first tile of every step of the last ht_ntt call
void ht_ntt_sync_counts(long* private_passes, long* barriers) { *private_passes = g_private_passes; *barriers = g_barriers; }
int ht_ntt_plan(uint32_t log_n, int steps, int tile, int* out9) {
  NttStepDesc d[3];
  const int c = ntt_plan(log_n, steps, d, tile);
  for (int i = 0; i < c; ++i) { out9[3 * i] = d[i].log_m; out9[3 * i + 1] = d[i].log_tc; out9[3 * i + 2] = d[i].row; }
  return c;
}
}
