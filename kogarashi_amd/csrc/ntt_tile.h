// ntt_tile.h -- one step of the Fr NTT on a tile held in LDS, with every size a compile-time constant.
//
// Replaces the arithmetic of groth16/src/fft.rs:166-218 (classic_fft_arithmetic / butterfly_arithmetic) for one
// factor m = 2^LOG_M of the transform length: a tile is 2^LOG_TC adjacent DFTs of m points each.  The radix-2 DIT
// stages run two at a time in registers (ntt_core.h).  The first register pass is fused with the global load and the
// last one with the global store, so a tile of m points makes ceil(LOG_M / 2) - 1 LDS round trips; between them the
// tile lives in LDS as nine 29-bit limb planes.  All index arithmetic is shifts and masks by constants.
//
// Every function here is a per-thread body, generic over the field type F and over the tile store: the kernel
// (ntt.hip) instantiates it with Fr and LDS planes, tests/host/hosttest_ntt.cpp runs the same code thread by thread on the
// host with the bound-checking FrC and checks whole transforms against the oracle.
#pragma once
#include <cstddef>
#include "fp29.h"
#include "ntt_core.h"

namespace kg {

constexpr int NTT_TW_LOG = 11;   // in-tile twiddle table: w_2048^e, e < 1024, as Shoup-form constants {w, floor(w 2^261 / p)}: [e][18] words
constexpr int NTT_MAX_LOG_M = 11;

struct NttStepArgs {
  const uint64_t* in;
  uint64_t* out;
  const uint32_t* tw_m;        // w_2048^e (e < 1024), direction of the transform
  const uint32_t* tw_direct;   // column steps: inter-step twiddles [row][inner column], or nullptr
  const uint32_t* tw_lo;       // two-level w_n^e = lo[e & mask] * hi[e >> lo_bits] (inter-step fallback above the direct tables)
  const uint32_t* tw_hi;
  const uint32_t* cos_lo;      // coset shift g^(+-e) [* n^-1 when inverse], two-level like w_n
  const uint32_t* cos_hi;
  uint64_t mult;               // column steps: exponent multiplier of the fallback twiddle w_n^(row * column * mult); 0 = none
  uint32_t lo_bits;
  uint32_t log_inner;          // column steps: log2 of the contiguous run (elements) between two rows of a DFT
  uint32_t log_G;              // row steps: log2 of the number of DFTs (= output stride)
  uint32_t log_n1;             // row steps: DFT g = i1 + n1 * i2 reads row i1 * (G / n1) + i2
  uint32_t scale_mode;         // 0 none; 1: input element j * cos(j) (coset_dft, fft.rs:109-116); 2: output element i * cos(i)
                               // (coset_idft, fft.rs:119-127); 3: output * cos(0) (= n^-1: idft, fft.rs:100-106)
  uint32_t tile_shift;         // XCD-aware tile order: tile = (block % 8) << tile_shift | block / 8 when tile_shift != 0
};

// ---- compile-time bit maps: thread bit j carries logical bit map[j] -------------------------------------------
struct BitMap {
  int n;
  int map[16];
};
// logical tile index idx = (p << LOG_TC) | col (p: position in the DIT working order, col: which DFT of the tile); LDS word
// = idx with its low five bits XORed with the higher five-bit groups.  ds_read_b32 / ds_write_b32 bank = word mod 32 per
// 32-lane half: a pass is conflict-free when the five idx bits its lanes vary sit in five different residue classes mod 5.
template <int B>
KG_HD uint32_t tile_phys(uint32_t idx) {
  if constexpr (B <= 5) return idx;
  else if constexpr (B <= 10) return idx ^ ((idx >> 5) & 31u);
  else return idx ^ ((idx >> 5) & 31u) ^ ((idx >> 10) & 31u);
}
// register pass over idx bits [F0, F0 + FW): the other B - FW bits come from the thread id
template <int B, int F0, int FW>
constexpr BitMap mid_map() {
  BitMap m{B - FW, {}};
  bool used[16] = {};
  for (int b = F0; b < F0 + FW; ++b) used[b] = true;
  int j = 0;
  if (B >= 10) {
    for (int r = 0; r < 5; ++r) {                  // lane bits: the lowest free idx bit of each residue class
      int b = r;
      while (used[b]) b += 5;
      m.map[j++] = b;
      used[b] = true;
    }
  }
  for (int b = 0; b < B; ++b)
    if (!used[b]) { m.map[j++] = b; used[b] = true; }
  return m;
}
// first pass of a column step: source index s = (r_low << LOG_TC) | col; the lanes keep the columns (global runs) and take
// the r_low bits whose LDS destination bits (bit-reversed: r_low bit b -> idx bit B - 1 - b) fill the remaining residues
template <int B, int LOG_TC, int RL>
constexpr BitMap first_col_map() {
  BitMap m{LOG_TC + RL, {}};
  bool used[16] = {};
  int j = 0;
  for (int b = 0; b < LOG_TC; ++b) { m.map[j++] = b; used[b] = true; }
  if (LOG_TC < 5 && B >= 10) {
    const int want = 5 - LOG_TC;
    int b0 = B % 5;
    if (b0 + want > RL) b0 = RL - want > 0 ? RL - want : 0;
    for (int b = b0; b < b0 + want && b < RL; ++b) { m.map[j++] = LOG_TC + b; used[LOG_TC + b] = true; }
  }
  for (int b = 0; b < LOG_TC + RL; ++b)
    if (!used[b]) { m.map[j++] = b; used[b] = true; }
  return m;
}
constexpr int run_len(const BitMap& m, int j) {
  int len = 1;
  while (j + len < m.n && m.map[j + len] == m.map[j] + len) ++len;
  return len;
}
template <class MapFn, int J>
KG_HD uint32_t scatter_bits(uint32_t t) {
  constexpr BitMap M = MapFn::get();
  if constexpr (J >= M.n) return 0u;
  else {
    constexpr int len = run_len(M, J);
    constexpr uint32_t mask = (1u << len) - 1u;
    return (((t >> J) & mask) << M.map[J]) | scatter_bits<MapFn, J + len>(t);
  }
}
template <int B, int F0, int FW> struct MidMapFn { static constexpr BitMap get() { return mid_map<B, F0, FW>(); } };
template <int B, int LOG_TC, int RL> struct FirstColMapFn { static constexpr BitMap get() { return first_col_map<B, LOG_TC, RL>(); } };

// ---- wave-private passes ------------------------------------------------------------------------------------
// A wave (64 lanes x 4 elements) owns the 256 tile elements whose idx shares its high bits ("block" = idx >> 8 = wave id).
// A register pass whose field lies inside the low eight idx bits touches only the wave's own block, and so does the first
// pass of a column step when the wave takes the rows that land in its block: between two such passes no workgroup barrier is
// needed (one wave's LDS operations execute in order), only where a pass exchanges data between blocks.  That removes 3 of
// the 5 barriers of a 2^11-point tile and 3 of the 4 of a 2^9-point one, and lets the waves of a workgroup drift apart.
// pick the lane bits (thread bits 0..5) among `cand` idx bits so that the first five fall into different residue classes
// mod 5 where possible (bank rule above); thread bits 6.. are the block
constexpr void lanes_by_residue(BitMap& m, int& j, const bool (&cand)[16]) {
  bool taken[16] = {};
  bool res[5] = {};
  for (int round = 0; round < 2; ++round)
    for (int b = 0; b < 16; ++b) {
      if (!cand[b] || taken[b]) continue;
      if (round == 0 && (res[b % 5] || j >= 5)) continue;
      m.map[j++] = b;
      taken[b] = true;
      res[b % 5] = true;
    }
}
template <int B, int F0>
constexpr BitMap mid_private_map() {                  // field [F0, F0 + 2) inside idx bits [0, 8)
  BitMap m{B - 2, {}};
  bool cand[16] = {};
  for (int b = 0; b < 8; ++b) cand[b] = !(b >= F0 && b < F0 + 2);
  int j = 0;
  lanes_by_residue(m, j, cand);
  for (int b = 8; b < B; ++b) m.map[j++] = b;          // wave id = block
  return m;
}
// first pass of a column step, wave-private: thread bits -> source bits s = (r_low << LOG_TC) | col.  The block a row lands
// in is brev(low NB bits of r_low) (bit-reversed placement), so the wave id supplies those bits reversed; the lanes (and the
// iteration bit of a radix-2 first pass) supply the columns and the remaining r_low bits, ordered by the residues of the idx
// bits they end up on (r_low bit NB + i -> idx bit 7 - i).
template <int B, int LOG_TC, int RL, int G0>
constexpr BitMap first_col_private_map() {
  constexpr int NB = B - 8;
  BitMap m{LOG_TC + RL, {}};
  int j = 0;
  // lanes: destination idx bits available = col bits (idx 0..LOG_TC-1) and idx bits 7 - i for r_low bit NB + i; the field bits
  // [LOG_TC, LOG_TC + G0) are the register index
  bool cand[16] = {};
  for (int b = 0; b < LOG_TC; ++b) cand[b] = true;
  for (int i = 0; i < RL - NB; ++i) cand[7 - i] = true;
  BitMap dst{0, {}};
  int jd = 0;
  lanes_by_residue(dst, jd, cand);
  // the columns must stay the fastest lane bits (global runs): put them first, then the rest in residue order
  for (int b = 0; b < LOG_TC; ++b) m.map[j++] = b;
  for (int t = 0; t < jd; ++t) {
    const int d = dst.map[t];
    if (d < LOG_TC) continue;
    if (j == 6) break;                                 // six lane bits
    m.map[j++] = LOG_TC + NB + (7 - d);
  }
  // wave bits 6..6+NB-1: r_low bit NB - 1 - i for wave bit i
  for (int i = 0; i < NB; ++i) m.map[j++] = LOG_TC + (NB - 1 - i);
  // what is left (the iteration bit of a radix-2 first pass): the remaining r_low bits
  bool used[16] = {};
  for (int t = 0; t < j; ++t) used[m.map[t]] = true;
  for (int b = 0; b < LOG_TC + RL; ++b)
    if (!used[b]) m.map[j++] = b;
  return m;
}
template <int B, int F0> struct MidPrivateMapFn { static constexpr BitMap get() { return mid_private_map<B, F0>(); } };
template <int B, int LOG_TC, int RL, int G0> struct FirstColPrivateMapFn { static constexpr BitMap get() { return first_col_private_map<B, LOG_TC, RL, G0>(); } };

template <int BITS>
KG_HD uint32_t brev_bits(uint32_t v) {
  if constexpr (BITS == 0) return 0u;
  else {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - BITS);
#else
    uint32_t r = 0;
    for (int i = 0; i < BITS; ++i) r |= ((v >> i) & 1u) << (BITS - 1 - i);
    return r;
#endif
  }
}

// ---- memory -----------------------------------------------------------------------------------------------
KG_HD void ntt_ld_words(const uint64_t* base, size_t elem, uint32_t w[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * elem;
  const uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
#else
  const uint32_t* p = reinterpret_cast<const uint32_t*>(base) + 8 * elem;
  for (int i = 0; i < 8; ++i) w[i] = p[i];
#endif
}
KG_HD void ntt_st_words(uint64_t* base, size_t elem, const uint32_t w[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint4* p = reinterpret_cast<uint4*>(base) + 2 * elem;
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
#else
  uint32_t* p = reinterpret_cast<uint32_t*>(base) + 8 * elem;
  for (int i = 0; i < 8; ++i) p[i] = w[i];
#endif
}
// field-type adapters (the bound-checking type supplies its own in tests/host/hosttest_ntt.cpp)
template <class F> struct NttIO;
template <class P> struct NttIO<Fp<P>> {
  static KG_HD Fp<P> raw(const uint32_t w[8]) { return limbs_from_words<P>(w); }          // any 256-bit value, no domain change
  static KG_HD Fp<P> table(const uint32_t* tab, size_t e) {                                // table entry: normalised, < 2p
    Fp<P> r;
    const uint32_t* p = tab + e * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = p[k];
    return r;
  }
  static KG_HD void words(const Fp<P>& a, uint32_t w[8]) { words_from_limbs(a, w); }
  static KG_HD FpConst<P> twc(const uint32_t* tab, size_t e) {                             // in-tile twiddle: {w, floor(w 2^261 / p)}
    FpConst<P> c;
    const uint32_t* p = tab + e * 18;
#pragma unroll
    for (int k = 0; k < 9; ++k) { c.w[k] = p[k]; c.q[k] = p[9 + k]; }
    return c;
  }
};
template <class F>
KG_HD F ntt_two_level(const uint32_t* lo, const uint32_t* hi, uint32_t lo_bits, uint64_t e) {
  const uint64_t el = e & ((1ull << lo_bits) - 1), eh = e >> lo_bits;
  F a = NttIO<F>::table(lo, el);
  if (eh == 0) return a;
  return mul(a, NttIO<F>::table(hi, eh));
}

// the kernel's tile store: limbs (2j, 2j+1) of LDS word w as one 8-byte pair at plane j (j < 4), limb 8 in a fifth plane of
// 4-byte words: four ds_read_b64 / ds_write_b64 and one b32 per element instead of nine b32 (a b64 wave access moves 512 B in
// the two LDS cycles a b32 one needs for 256 B).  Banks: a pair at word w sits on banks 2w, 2w + 1 mod 64 -- the 32 lanes of a
// half are conflict-free exactly when their w differ mod 32, the same rule as for b32 (tile_phys).
template <int ELEMS>
struct LdsPlanes {
  uint32_t* base;
  template <class P>
  KG_HD void store(uint32_t w, const Fp<P>& a) const {
#ifdef KG_NTT_EXP_NOLDS     // phase-off experiment: one word per element instead of nine
    base[w] = a.l[0] ^ a.l[1] ^ a.l[2] ^ a.l[3] ^ a.l[4] ^ a.l[5] ^ a.l[6] ^ a.l[7] ^ a.l[8];
    return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    uint2* b2 = reinterpret_cast<uint2*>(base);
#pragma unroll
    for (int j = 0; j < 4; ++j) b2[j * ELEMS + w] = make_uint2(a.l[2 * j], a.l[2 * j + 1]);
    base[8 * ELEMS + w] = a.l[8];
#else
    for (int k = 0; k < 9; ++k) base[k * ELEMS + w] = a.l[k];
#endif
  }
  template <class F>
  KG_HD F load(uint32_t w) const {
    F r;
#ifdef KG_NTT_EXP_NOLDS
    const uint32_t v = base[w];
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = (v >> k) & M29;
    return r;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const uint2* b2 = reinterpret_cast<const uint2*>(base);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint2 v = b2[j * ELEMS + w]; r.l[2 * j] = v.x; r.l[2 * j + 1] = v.y; }
    r.l[8] = base[8 * ELEMS + w];
#else
    for (int k = 0; k < 9; ++k) r.l[k] = base[k * ELEMS + w];
#endif
    return r;
  }
};

// ---- one tile ------------------------------------------------------------------------------------------------
// ROW = false ("column step"): the DFT runs over rows that lie `inner` elements apart; the tile is 2^LOG_TC adjacent
//   columns; results go back to the same places (of `out`), multiplied by the inter-step twiddle; not canonical (< 2p).
// ROW = true ("row step", always the last): each DFT is a contiguous row of m elements; output element i of DFT g goes to
//   out[g + i * G] (adjacent DFTs -> adjacent addresses), canonical.
template <class F, int LOG_M, int LOG_TC, bool ROW>
struct NttTile {
  static constexpr int B = LOG_M + LOG_TC;
  static constexpr int ELEMS = 1 << B;
  static constexpr int NT = ELEMS / 4 >= 64 ? ELEMS / 4 : 64;
  static constexpr int G0 = (LOG_M & 1) ? 1 : 2;            // stages of the first pass (fused with the load)
  static constexpr int RL = LOG_M - G0;                     // bits of r_low in the first pass
  static constexpr bool SINGLE = LOG_M <= 2;                // one pass: load -> butterflies -> store, no LDS
  static constexpr int S_LAST = LOG_M - 2;                  // first stage index of the last pass (fused with the store)
  static constexpr int M = 1 << LOG_M, TC = 1 << LOG_TC;
  // wave-private passes (see above): tiles of at least four waves, lanes and blocks as thread bits 0..5 / 6..
#ifdef KG_NTT_NO_WAVE_PRIVATE
  static constexpr bool WAVE_OK = false;
#else
  static constexpr bool WAVE_OK = B >= 10 && NT == ELEMS / 4;
#endif
  static constexpr bool FIRST_PRIVATE = WAVE_OK && !ROW;       // a row step loads contiguous rows: its first pass scatters over the blocks
  template <int S0> static constexpr bool mid_private() { return WAVE_OK && LOG_TC + S0 + 2 <= 8; }

  const NttStepArgs& A;
  uint32_t tile;
#ifdef KG_NTT_EXP_NOMEM        // phase-off experiment: every workgroup works on one of tiles 0..63 (8 MB: the data stays in the L2 / Infinity Cache)
#define KG_NTT_TILE_ID (tile & 63u)
#else
#define KG_NTT_TILE_ID tile
#endif

  // twiddles of a register pass over stages s0+1 .. s0+G: w_{2^s}^j, j = r mod 2^(s-1)
  struct Tw {
    const uint32_t* tw_m;
    uint32_t s0, r_low;
    KG_HD F mul(const F& x, int t, int k0) const {
      const uint32_t s = s0 + (uint32_t)t;
      const uint32_t j = (((uint32_t)k0 & ((1u << (t - 1)) - 1u)) << s0) | r_low;
#ifdef KG_NTT_EXP_NOTW      // phase-off experiment (tools/dbg): every twiddle is the same table entry
      return mulc(x, NttIO<F>::twc(tw_m, (size_t)(j & 0u) + 5));
#endif
      return mulc(x, NttIO<F>::twc(tw_m, (size_t)j << (NTT_TW_LOG - s)));
    }
  };

  // ---- addresses ----
  KG_HD uint64_t col_base() const {         // column step: address of (row 0, column 0 of the tile)
    const uint64_t g0 = (uint64_t)KG_NTT_TILE_ID << LOG_TC;
    return ((g0 >> A.log_inner) << (LOG_M + A.log_inner)) + (g0 & ((1ull << A.log_inner) - 1));
  }
  KG_HD uint64_t row_in(uint32_t col) const {   // row step: first element of DFT g0 + col
    const uint64_t g = ((uint64_t)KG_NTT_TILE_ID << LOG_TC) + col;
    const uint64_t i1 = g & ((1ull << A.log_n1) - 1), i2 = g >> A.log_n1;
    return ((i1 << (A.log_G - A.log_n1)) + i2) << LOG_M;
  }

  KG_HD F load_elem(uint64_t addr) const {
    uint32_t w[8];
    ntt_ld_words(A.in, addr, w);
    F v = NttIO<F>::raw(w);
    if (A.scale_mode == 1) v = mul(v, ntt_two_level<F>(A.cos_lo, A.cos_hi, A.lo_bits, addr));
    return v;
  }
  // x: lazy output of the last stage; i: output row; col: DFT of the tile
  KG_HD void store_elem(const F& x, uint32_t i, uint32_t col) const {
    uint32_t w[8];
    if constexpr (ROW) {
      const uint64_t addr = (((uint64_t)KG_NTT_TILE_ID << LOG_TC) + col) + ((uint64_t)i << A.log_G);
      F v;
      if (A.scale_mode == 2) v = mul(x, ntt_two_level<F>(A.cos_lo, A.cos_hi, A.lo_bits, addr));
      else if (A.scale_mode == 3) v = mul(x, NttIO<F>::table(A.cos_lo, 0));
      else v = vred(norm(x));
      NttIO<F>::words(reduce_2p(v), w);
      ntt_st_words(A.out, addr, w);
    } else {
      const uint64_t off = ((uint64_t)i << A.log_inner) + ((((uint64_t)KG_NTT_TILE_ID << LOG_TC) & ((1ull << A.log_inner) - 1)) + col);
      const uint64_t slab = ((((uint64_t)KG_NTT_TILE_ID << LOG_TC) >> A.log_inner) << (LOG_M + A.log_inner));
      F v;
#ifdef KG_NTT_EXP_NOMATH
      if (A.tw_direct) v = add(x, NttIO<F>::table(A.tw_direct, off));
      else
#endif
      if (A.tw_direct) v = mul(x, NttIO<F>::table(A.tw_direct, off));
      else if (A.mult) v = mul(x, ntt_two_level<F>(A.tw_lo, A.tw_hi, A.lo_bits, (uint64_t)i * ((off & ((1ull << A.log_inner) - 1)) * A.mult)));
      else v = vred(norm(x));
      NttIO<F>::words(v, w);                      // < 2p < 2^255: the next step reads it as a raw 256-bit value
      ntt_st_words(A.out, slab + off, w);
    }
  }

  // ---- first pass: global load, stages 1..G0 (all twiddles trivial but w_4), tile -> store ----
  template <class Store>
  KG_HD void first(uint32_t tid, const Store& st) const {
    constexpr int groups = ELEMS >> G0;
    constexpr int iters = (groups + NT - 1) / NT;
#pragma unroll
    for (int it = 0; it < iters; ++it) {
      const uint32_t q = tid + (uint32_t)it * NT;
      if (groups < NT && q >= (uint32_t)groups) break;
      uint32_t col, r_low;
      if constexpr (ROW) {
        r_low = q & ((1u << RL) - 1u);
        col = q >> RL;
      } else if constexpr (FIRST_PRIVATE) {
        const uint32_t s = scatter_bits<FirstColPrivateMapFn<B, LOG_TC, RL, G0>, 0>(q);
        col = s & (TC - 1u);
        r_low = s >> LOG_TC;
      } else {
        const uint32_t s = scatter_bits<FirstColMapFn<B, LOG_TC, RL>, 0>(q);
        col = s & (TC - 1u);
        r_low = s >> LOG_TC;
      }
      F x[1 << G0];
      const uint64_t base = ROW ? row_in(col) + r_low : col_base() + col + ((uint64_t)r_low << A.log_inner);
#pragma unroll
      for (int k = 0; k < (1 << G0); ++k) {
        const uint32_t j = brev_bits<G0>((uint32_t)k);                    // position bit k <-> row bit (reversed)
        const uint64_t addr = ROW ? base + ((uint64_t)j << RL) : base + (((uint64_t)j << RL) << A.log_inner);
        x[k] = load_elem(addr);
      }
      Tw tw{A.tw_m, 0u, 0u};
      dit_network<G0>(x, true, tw);
      if constexpr (SINGLE) {
#pragma unroll
        for (int k = 0; k < (1 << G0); ++k) store_elem(x[k], (uint32_t)k, col);
      } else {
        const uint32_t p_high = brev_bits<RL>(r_low);
#pragma unroll
        for (int k = 0; k < (1 << G0); ++k)
          st.store(tile_phys<B>((((p_high << G0) | (uint32_t)k) << LOG_TC) | col), norm(x[k]));
      }
    }
  }

  // ---- middle pass over stages S0+1, S0+2: LDS -> LDS, in place ----
  template <int S0, class Store>
  KG_HD void mid(uint32_t tid, const Store& st) const {
    constexpr int F0 = LOG_TC + S0;
    if (ELEMS / 4 < NT && tid >= (uint32_t)(ELEMS / 4)) return;
    uint32_t others;
    if constexpr (mid_private<S0>()) others = scatter_bits<MidPrivateMapFn<B, F0>, 0>(tid);
    else others = scatter_bits<MidMapFn<B, F0, 2>, 0>(tid);
    const uint32_t r_low = (others >> LOG_TC) & ((1u << S0) - 1u);
    const uint32_t w0 = tile_phys<B>(others);
    F x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = st.template load<F>(w0 ^ tile_phys<B>((uint32_t)k << F0));
    Tw tw{A.tw_m, (uint32_t)S0, r_low};
    dit_network<2>(x, false, tw);
    // the never-multiplied path gains 3p per stage: a 2^11-point tile would pass the 42p the final value reduction
    // admits, so its middle pass stores value-reduced elements (45 instructions per element, once)
    constexpr bool REDUCE_HERE = LOG_M >= 11 && S0 == 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) st.store(w0 ^ tile_phys<B>((uint32_t)k << F0), REDUCE_HERE ? vred(norm(x[k])) : norm(x[k]));
  }

  // ---- last pass over stages LOG_M-1, LOG_M: LDS -> inter-step twiddle / scaling -> global store ----
  template <class Store>
  KG_HD void last(uint32_t tid, const Store& st) const {
    constexpr int F0 = LOG_TC + S_LAST;
    if (ELEMS / 4 < NT && tid >= (uint32_t)(ELEMS / 4)) return;
    const uint32_t col = tid & (TC - 1u), p_low = tid >> LOG_TC;
    const uint32_t w0 = tile_phys<B>((p_low << LOG_TC) | col);
    F x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = st.template load<F>(w0 ^ tile_phys<B>((uint32_t)k << F0));
    Tw tw{A.tw_m, (uint32_t)S_LAST, p_low};
    dit_network<2>(x, false, tw);
#pragma unroll
    for (int k = 0; k < 4; ++k) store_elem(x[k], p_low + ((uint32_t)k << S_LAST), col);
  }

  // the passes between first() and last(), in order.  full(): workgroup barrier; wave(): only this wave's LDS operations need
  // ordering (both neighbours of the boundary are wave-private passes)
  template <class Full, class Wave>
  KG_HD void after_first(const Full& full, const Wave& wave) const {
    if constexpr (G0 < S_LAST && FIRST_PRIVATE && mid_private<G0>()) wave();
    else full();
  }
  template <int S0, class Store, class Full, class Wave>
  KG_HD void mids(uint32_t tid, const Store& st, const Full& full, const Wave& wave) const {
    if constexpr (S0 < S_LAST) {
      mid<S0>(tid, st);
      if constexpr (S0 + 2 < S_LAST && mid_private<S0>() && mid_private<S0 + 2>()) wave();
      else full();
      mids<S0 + 2>(tid, st, full, wave);
    }
  }
};

}  // namespace kg

// ---- table entries (shared by the device table kernels and the host test) ----------------------------------------
namespace kg {
// ROOT_OF_UNITY^(+-1) squared (28 - log) times: a primitive 2^log-th root (fft.rs:34,44)
template <class F>
KG_HD F ntt_root_of(uint32_t log, int inverse) {
  using P = typename F::Params;
  F g = F::from_const(inverse ? P::ROOT_OF_UNITY_INV : P::ROOT_OF_UNITY);
  for (uint32_t i = log; i < 28; ++i) g = sqr(g);
  return g;
}
template <class F>
KG_HD F ntt_pow(F base, uint64_t e) {
  F r = F::one();
  while (e) {
    if (e & 1) r = mul(r, base);
    base = sqr(base);
    e >>= 1;
  }
  return r;
}

// ---- plan: which tile shapes transform 2^log_n elements --------------------------------------------------------
// n = n1 * n2 * n3 with the Cooley-Tukey index map j = j1*n2*n3 + j2*n3 + j3 -> i = i1 + n1*i2 + n1*n2*i3 (natural order in
// and out, no bit-reversal pass):
//   step A (column)  n1-point DFTs over j1, * w_n^(i1 * (j2*n3 + j3)),          data -> scratch
//   step B (column)  n2-point DFTs over j2 inside each i1 slab, * w_n^(n1*i2*j3), scratch in place   [three-step plans only]
//   step C (row)     n3-point DFTs over j3 along contiguous rows, written transposed, scratch -> data
// Two steps (n2 = 1) up to 2^21 (2^22 on request): two HBM round trips, factors up to 2^11 (a 4096-element tile is 144 KiB of LDS).
struct NttStepDesc { int log_m, log_tc; bool row; };
// The tile shapes (LOG_M, LOG_TC) that exist as kernels: every shape the automatic plans use, the three-step plans' and the
// 2048-element tiles of 2^10- / 2^11-point factors; row steps also run the single-step transforms (LOG_TC = 0).
#define KG_NTT_SHAPES(X) X(6, 4) X(7, 3) X(7, 4) X(8, 2) X(8, 3) X(9, 1) X(9, 2) X(10, 1) X(11, 1)
#define KG_NTT_SHAPES_COL_ONLY(X) X(10, 0) X(11, 0)
#define KG_NTT_SHAPES_ROW_ONLY(X) X(1, 0) X(2, 0) X(3, 0) X(4, 0) X(5, 0) X(6, 0) X(7, 0) X(8, 0) X(9, 0) X(10, 0) X(11, 0)
inline bool ntt_shape_exists(int log_m, int log_tc, bool row) {
#define X(m, tc) if (log_m == (m) && log_tc == (tc)) return true;
  KG_NTT_SHAPES(X)
  if (row) { KG_NTT_SHAPES_ROW_ONLY(X) } else { KG_NTT_SHAPES_COL_ONLY(X) }
#undef X
  return false;
}
inline int ntt_tile_log(int log_m, int want, bool forced = false) {   // tile = 2^(log_m + log_tc) elements: 1024, 2048 or 4096
  const int least = forced ? log_m : log_m + 1;        // automatic plans keep at least two adjacent DFTs per tile (64-byte runs)
  return want < least ? least : want;
}
// a forced tile size falls back to the automatic one for a factor whose shape does not exist as a kernel
inline NttStepDesc ntt_step_desc(int log_m, int want_auto, int tile, bool row) {
  if (tile) {
    const int t = ntt_tile_log(log_m, tile, true);
    if (ntt_shape_exists(log_m, t - log_m, row)) return {log_m, t - log_m, row};
  }
  return {log_m, ntt_tile_log(log_m, want_auto) - log_m, row};
}
// steps: 0 = automatic (one step up to 2^11, two up to 2^21, three above); 2 forces two steps up to 2^22, 3 three from 2^18 up.
// tile: 0 = automatic, else log2 of the tile size wanted (10..12)
// Measured (tools/dbg/ntt_plans.py, round 3): at 2^22 two steps of 2^11 points -- 4096-element tiles, one 16-wave workgroup per
// CU, six register passes per step of which the first is a lone radix-2 stage -- take 508-522 us, three steps on 1024-element
// tiles (four workgroups per CU) 462-481 us although they move the data once more and multiply once more per element; at
// 2^21 two steps still win (213-221 against 231-240 us).
inline int ntt_plan(uint32_t log_n, int steps, NttStepDesc out[3], int tile = 0) {
  const int k = (int)log_n;
  if (k <= NTT_MAX_LOG_M) { out[0] = {k, 0, true}; return 1; }
  const bool three = k > 2 * NTT_MAX_LOG_M || (steps == 3 && k >= 18) || (steps != 2 && k >= 22);
  if (!three) {
    const int k1 = (k + 1) / 2, k3 = k - k1;       // (two odd halves stay: 18 = 10 + 8 saves a pass but leaves step A 128 workgroups: 48 against 40 us)
    const int want = k <= 19 ? 10 : (k <= 21 ? 11 : 12);
    out[0] = ntt_step_desc(k1, want, tile, false);
    out[1] = ntt_step_desc(k3, want, tile, true);
    return 2;
  }
  // a factor of 2^m costs ceil(m / 2) register passes (an odd m starts with a lone radix-2 stage): balanced factors, then two
  // odd ones trade a stage so that both become even -- 22 = 8 + 8 + 6 (11 passes) instead of 8 + 7 + 7 (12)
  int k1 = (k + 2) / 3, k2 = (k - k1 + 1) / 2, k3 = k - k1 - k2;
  if ((k2 & 1) && (k3 & 1) && k2 + 1 <= NTT_MAX_LOG_M) { ++k2; --k3; }
  else if ((k1 & 1) && (k2 & 1) && k1 + 1 <= NTT_MAX_LOG_M) { ++k1; --k2; }
  else if ((k1 & 1) && (k3 & 1) && k1 + 1 <= NTT_MAX_LOG_M) { ++k1; --k3; }
  const int want = k <= 24 ? 10 : 11;               // 1024-element tiles while the factors allow them: 2^22 478 / 494, 2^24 1783 / 1834 us
  out[0] = ntt_step_desc(k1, want, tile, false);
  out[1] = ntt_step_desc(k2, want, tile, false);
  out[2] = ntt_step_desc(k3, want, tile, true);
  return 3;
}
}  // namespace kg

namespace kg {
// the arguments of step i of a plan; returns the number of tiles (= workgroups)
struct NttTables { const uint32_t *small, *lo, *hi, *cos_lo, *cos_hi, *direct0, *direct1; uint32_t lo_bits; };
inline uint32_t ntt_step_args(uint32_t log_n, int nsteps, const NttStepDesc* d, int i, const NttTables& T, uint64_t* data, uint64_t* tmp,
                              int inverse, int coset, NttStepArgs& a) {
  a = NttStepArgs{};
  a.tw_m = T.small; a.tw_lo = T.lo; a.tw_hi = T.hi; a.cos_lo = T.cos_lo; a.cos_hi = T.cos_hi; a.lo_bits = T.lo_bits;
  const uint32_t pre = (coset && !inverse) ? 1u : 0u;                     // coset_dft: * 7^j before the transform
  const uint32_t post = inverse ? (coset ? 2u : 3u) : 0u;                 // coset_idft: * n^-1 * 7^-i ; idft: * n^-1 (cos_lo[0] of the inverse tables)
  const uint32_t lt = log_n - (uint32_t)(d[i].log_m + d[i].log_tc);      // log2(number of tiles)
  a.tile_shift = lt >= 3 ? lt - 3 : 0;
  if (nsteps == 1) {
    a.in = data; a.out = data; a.scale_mode = pre ? 1u : post;
    return 1u;
  }
  const uint32_t k1 = (uint32_t)d[0].log_m, k3 = (uint32_t)d[nsteps - 1].log_m;
  if (i == 0) {                                    // step A: data -> tmp
    a.in = data; a.out = tmp; a.log_inner = log_n - k1; a.mult = 1; a.tw_direct = T.direct0; a.scale_mode = pre;
  } else if (i < nsteps - 1) {                     // step B: tmp in place
    a.in = tmp; a.out = tmp; a.log_inner = k3; a.mult = 1ull << k1; a.tw_direct = T.direct1;
  } else {                                         // step C: tmp -> data, transposed write
    a.in = tmp; a.out = data; a.log_G = log_n - k3; a.log_n1 = k1; a.scale_mode = post;
  }
  return 1u << lt;
}
}  // namespace kg
