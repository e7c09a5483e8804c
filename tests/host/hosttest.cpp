// hosttest.cpp -- compiles the DEVICE arithmetic templates (fp29.h, curve.h) for the host with g++ so the
// CPU test-suite can check them against the oracle without a GPU, and runs every formula once more with the
// bound-tracking FpChecked shadow type (aborts on any possible overflow).  Test infrastructure only.
#include <cstring>
#include <vector>
#include "../../kogarashi_amd/csrc/fp29.h"
#include "../../kogarashi_amd/csrc/fp29_checked.h"
#include "../../kogarashi_amd/csrc/curve.h"
#include "../../kogarashi_amd/csrc/ntt_core.h"
#include "../../kogarashi_amd/csrc/vecops.h"
#include "../../kogarashi_amd/csrc/msm_digits.h"
#include "../../kogarashi_amd/csrc/coop_add.h"
#include "../../kogarashi_amd/csrc/host_fp.h"

using namespace kg;

template <class F> struct Conv;   // reference-form words <-> field type (plain or checked)
template <class P> struct Conv<Fp<P>> {
  static Fp<P> in(const uint32_t* w) { return from_ref<P>(w); }
  static void out(const Fp<P>& a, uint32_t* w) { to_ref(a, w); }
  static constexpr int W = 8;
};
template <class P> struct Conv<FpChecked<P>> {
  static FpChecked<P> in(const uint32_t* w) {
    FpChecked<P> raw = FpChecked<P>::wrap(limbs_from_words<P>(w), 5.4);   // any 256-bit input
    return mul(raw, FpChecked<P>::from_const(P::C_FROM_REF));
  }
  static void out(const FpChecked<P>& a, uint32_t* w) {
    words_from_limbs(reduce_2p(mul(a, FpChecked<P>::from_const(P::C_TO_REF))).v, w);
  }
  static constexpr int W = 8;
};
template <class F> struct Conv<Fp2<F>> {
  static Fp2<F> in(const uint32_t* w) { return {Conv<F>::in(w), Conv<F>::in(w + 8)}; }
  static void out(const Fp2<F>& a, uint32_t* w) { Conv<F>::out(a.c0, w); Conv<F>::out(a.c1, w + 8); }
  static constexpr int W = 16;
};

template <class F> static F xsub(const F& a, const F& b) { return norm(sub<4, 1>(a, b)); }

// op: 0 mul, 1 sqr, 2 add, 3 sub, 4 neg, 5 dbl, 6 inv, 7 (a*b+a*a) via lazy chain, 8 identity round trip, 9 inv by binary GCD
template <class F>
static void field_ops(int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
  const int W = Conv<F>::W;
  for (size_t i = 0; i < n; ++i) {
    F x = Conv<F>::in(a + W * i), y = Conv<F>::in(b + W * i), r = x;
    switch (op) {
      case 0: r = mul(x, y); break;
      case 1: r = sqr(x); break;
      case 2: r = norm(add(x, y)); break;
      case 3: r = xsub(x, y); break;
      case 4: r = xsub(F::zero(), x); break;
      case 5: r = norm(dbl(x)); break;
      case 6: r = inv(x); break;
      case 7: r = mul(norm(add(mul(x, y), sqr(x))), F::one()); break;
      case 8: r = x; break;
      case 9: r = inv_fast(x); break;          // binary-GCD inversion (fp_inv.h)
    }
    Conv<F>::out(r, o + W * i);
  }
}

// The HOST field type of the product (host_fp.h: 4 x 64-bit Montgomery words, the ABI form itself; the MSM's host finish and the
// prover's assembly run on it).  field: 0 Fr, 1 Fq; op: 0 mul, 1 sqr, 2 add, 3 sub, 4 inv, 5 dbl
template <class HF>
static void hostfp_ops(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const HF x = HF::from_words(a + 4 * i), y = HF::from_words(b + 4 * i);
    HF r = x;
    switch (op) {
      case 0: r = mul(x, y); break;
      case 1: r = sqr(x); break;
      case 2: r = add(x, y); break;
      case 3: r = sub<4, 1>(x, y); break;
      case 4: r = inv(x); break;
      case 5: r = dbl(x); break;
    }
    r.to_words(o + 4 * i);
  }
}
extern "C" {
// field: 0 Fr, 1 Fq, 2 Fq2 ; checked: run with the bound-tracking shadow type
void ht_field_ops(int field, int checked, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
  if (field == 0) { if (checked) field_ops<FrC>(op, a, b, o, n); else field_ops<Fr>(op, a, b, o, n); }
  else if (field == 1) { if (checked) field_ops<FqC>(op, a, b, o, n); else field_ops<Fq>(op, a, b, o, n); }
  else { if (checked) field_ops<Fq2C>(op, a, b, o, n); else field_ops<Fq2>(op, a, b, o, n); }
}
void ht_hostfp_ops(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  if (field == 0) hostfp_ops<kg::HostFr>(op, a, b, o, n); else hostfp_ops<kg::HostFq>(op, a, b, o, n);
}
void ht_ref_to_int(int field, const uint32_t* a, uint32_t* o, size_t n) {
  for (size_t i = 0; i < n; ++i) { if (field == 0) ref_to_int<FrParams>(a + 8 * i, o + 8 * i); else ref_to_int<FqParams>(a + 8 * i, o + 8 * i); }
}
void ht_int_to_ref(int field, const uint32_t* a, uint32_t* o, size_t n) {
  for (size_t i = 0; i < n; ++i) { if (field == 0) int_to_ref<FrParams>(a + 8 * i, o + 8 * i); else int_to_ref<FqParams>(a + 8 * i, o + 8 * i); }
}
}

// ---- curve ----------------------------------------------------------------------------------------
template <class F>
static Affine<F> ld_aff(const uint32_t* xy) { return {Conv<F>::in(xy), Conv<F>::in(xy + Conv<F>::W)}; }
template <class F>
static int st_aff(const XYZZ<F>& p, uint32_t* xy) {
  Affine<F> a;
  if (!to_affine(p, a)) { std::memset(xy, 0, 8 * Conv<F>::W); return 1; }
  Conv<F>::out(a.x, xy); Conv<F>::out(a.y, xy + Conv<F>::W);
  return 0;
}
// a point in memory read and written coordinate by coordinate: the device's SoaSrc / SoaDst / AosSrc accessors over one slot
template <class F> struct MemPt {
  XYZZ<F>* s;
  F x() const { return s->x; }
  F y() const { return s->y; }
  F zz() const { return s->zz; }
  F zzz() const { return s->zzz; }
  void x(const F& v) const { s->x = v; }
  void y(const F& v) const { s->y = v; }
  void zz(const F& v) const { s->zz = v; }
  void zzz(const F& v) const { s->zzz = v; }
};
// The lane-cooperative addition / doubling of coop_add.h on the host: the step functions run lane by lane, step by step, over an "image"
// that holds field elements of type F (the device's image holds their limbs; the bound-tracking type keeps its bounds this way).
template <class F> struct HostQuad {
  using Field = F;
  struct Ref { int kind; uint32_t a, b; };           // 0: coordinate b of item a; 1: temporary a
  std::vector<XYZZ<F>>* img; std::vector<F>* tmp; uint32_t* flg;
  uint32_t ia, ib, io; int lane;
  Ref coord(uint32_t item, uint32_t cd) const { return {0, item, cd}; }
  Ref t(uint32_t s_) const { return {1, s_, 0}; }
  F ld(const Ref& r) const {
    if (r.kind == 1) return (*tmp)[r.a];
    const XYZZ<F>& p = (*img)[r.a];
    return r.b == 0 ? p.x : (r.b == 1 ? p.y : (r.b == 2 ? p.zz : p.zzz));
  }
  void st(const Ref& r, const F& v) const {
    if (r.kind == 1) { (*tmp)[r.a] = v; return; }
    XYZZ<F>& p = (*img)[r.a];
    (r.b == 0 ? p.x : (r.b == 1 ? p.y : (r.b == 2 ? p.zz : p.zzz))) = v;
  }
  uint32_t flag() const { return *flg; }
  void set_flag(uint32_t v) const { *flg = v; }
  static F pick(bool first, const F& a, const F& b) { return first ? a : b; }
};
template <class F>
static void host_coop_add(std::vector<XYZZ<F>>& img, uint32_t ia, uint32_t ib, uint32_t io) {
  std::vector<F> tmp(COOP_TMP_SLOTS, F::zero());
  uint32_t flag = 0;
  auto quad = [&](int lane) { return HostQuad<F>{&img, &tmp, &flag, ia, ib, io, lane}; };
  const int mode = coop_add_mode(quad(0));
  if (mode == COOP_ADD) for (int l = 0; l < 4; ++l) coop_add_s1(quad(l));
  if (mode == COOP_ADD) for (int l = 3; l >= 0; --l) coop_add_s2(quad(l));       // (lane order inside a step must not matter)
  for (int l = 0; l < 4; ++l) coop_add_s3(quad(l), mode);
  for (int l = 3; l >= 0; --l) coop_add_s4(quad(l), mode);
}
template <class F>
static void host_coop_dbl(std::vector<XYZZ<F>>& img, uint32_t it) {
  std::vector<F> tmp(COOP_TMP_SLOTS, F::zero());
  uint32_t flag = 0;
  auto quad = [&](int lane) { return HostQuad<F>{&img, &tmp, &flag, it, it, it, lane}; };
  if (!coop_dbl_active(quad(0))) return;
  for (int l = 0; l < 4; ++l) coop_dbl_s1(quad(l));
  for (int l = 3; l >= 0; --l) coop_dbl_s2(quad(l));
  for (int l = 0; l < 4; ++l) coop_dbl_s3(quad(l));
}
// The register form of the same steps (coop_a1 .. a4, coop_d1 .. d3: what the device runs): the four lanes' values between the steps are
// arrays here, the exchange schedule of coop_add.h's comment is the indexing below (the device's DPP patterns in coop_add_level).
template <class F>
static void host_coop_add_regs(std::vector<XYZZ<F>>& img, uint32_t ia, uint32_t ib, uint32_t io) {
  std::vector<F> tmp(COOP_TMP_SLOTS, F::zero());
  uint32_t flag = 0;
  auto quad = [&](int lane) { return HostQuad<F>{&img, &tmp, &flag, ia, ib, io, lane}; };
  const int mode = coop_add_mode(quad(0));
  if (mode == COOP_TAKE_A || mode == COOP_TAKE_B) { for (int l = 0; l < 4; ++l) coop_add_s3(quad(l), mode); return; }
  if (mode != COOP_ADD) return;
  F r1[4], d[4], r2[4], r3[4];
  for (int l = 0; l < 4; ++l) r1[l] = coop_a1(quad(l));
  const int pa[4] = {1, 3, 2, 3}, pb[4] = {0, 2, 2, 3};
  for (int l = 3; l >= 0; --l) coop_a2(quad(l), r1[pa[l]], r1[pb[l]], d[l], r2[l]);
  const F pp = r2[0], rr = d[1];
  if (is_zero_2p(pp)) {
    if (is_zero(rr)) coop_set_point(quad(0), io, double_xyzz(coop_point(quad(0), ia)));
    else coop_set_point(quad(0), io, XYZZ<F>::identity());
    return;
  }
  for (int l = 0; l < 4; ++l) r3[l] = coop_a3(quad(l), d[l], r2[l], r1[0], pp);
  const int pe2[4] = {1, 3, 2, 3};
  for (int l = 3; l >= 0; --l) coop_a4(quad(l), r3[l], r3[1], l == 0 ? rr : r3[0], r2[pe2[l]], r1[2]);
}
template <class F>
static void host_coop_dbl_regs(std::vector<XYZZ<F>>& img, uint32_t it) {
  std::vector<F> tmp(COOP_TMP_SLOTS, F::zero());
  uint32_t flag = 0;
  auto quad = [&](int lane) { return HostQuad<F>{&img, &tmp, &flag, it, it, it, lane}; };
  if (!coop_dbl_active(quad(0))) return;
  F r1[4], r2[4];
  for (int l = 0; l < 4; ++l) r1[l] = coop_d1(quad(l));
  const int pv[4] = {0, 0, 0, 1};
  // (the device's lanes read the item before any lane of the quad writes it: step 2's reads of all lanes come before lane 2's store in
  // program order; here the loop order does the same -- lane 2 last)
  const int order2[4] = {0, 1, 3, 2};
  for (int k = 0; k < 4; ++k) { const int l = order2[k]; r2[l] = coop_d2(quad(l), r1[pv[l]]); }
  const int pe0[4] = {3, 0, 2, 3};
  for (int l = 1; l >= 0; --l) coop_d3(quad(l), r2[l], r2[pe0[l]], r2[1], r1[1]);
}
// mode 0: left fold with add_mixed; 1: pairwise tree with add_xyzz; 2: fold of add_xyzz(from_affine);
// 3: sum_i 2*P_i via double_affine + add_xyzz; 4: double_xyzz applied `n` times to point 0; 5: fold, negated;
// 7: fold of add_mixed_signed(acc, -P_i, negate = true), i.e. the sign folded back: equals mode 0;
// 8: the device pipeline's shape -- chunks of three folded with add_mixed_signed (their X is not value-reduced), the
//    partial sums combined by an add_xyzz tree
// 9: fold of add_xyzz_stream with the running sum as first operand AND output (in place); 10: double_xyzz_stream in place, n times
template <class F>
static int curve_sum(int mode, const uint32_t* pts, const uint8_t* inf, size_t n, uint32_t* out_xy) {
  const int W2 = 2 * Conv<F>::W;
  XYZZ<F> acc = XYZZ<F>::identity();
  if (mode == 0 || mode == 5) {
    for (size_t i = 0; i < n; ++i) if (!inf || !inf[i]) acc = add_mixed(acc, ld_aff<F>(pts + W2 * i));
    if (mode == 5) acc = neg_xyzz(acc);
  } else if (mode == 7) {
    for (size_t i = 0; i < n; ++i) if (!inf || !inf[i]) {
      Affine<F> a = ld_aff<F>(pts + W2 * i);
      acc = (i & 1) ? add_mixed_signed(acc, neg_affine(a), true) : add_mixed_signed(acc, a, false);
    }
  } else if (mode == 8) {
    std::vector<XYZZ<F>> v;
    XYZZ<F> part = XYZZ<F>::identity();
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
      if (!(inf && inf[i])) {
        Affine<F> a = ld_aff<F>(pts + W2 * i);
        part = (i & 1) ? add_mixed_signed(part, neg_affine(a), true) : add_mixed_signed(part, a, false);
      }
      if (++cnt == 3 || i + 1 == n) { v.push_back(part); part = XYZZ<F>::identity(); cnt = 0; }
    }
    while (v.size() > 1) {
      std::vector<XYZZ<F>> w;
      for (size_t i = 0; i + 1 < v.size(); i += 2) w.push_back(add_xyzz(v[i], v[i + 1]));
      if (v.size() & 1) w.push_back(v.back());
      v.swap(w);
    }
    if (!v.empty()) acc = v[0];
  } else if (mode == 1) {
    std::vector<XYZZ<F>> v;
    for (size_t i = 0; i < n; ++i) v.push_back((inf && inf[i]) ? XYZZ<F>::identity() : from_affine(ld_aff<F>(pts + W2 * i)));
    while (v.size() > 1) {
      std::vector<XYZZ<F>> w;
      for (size_t i = 0; i + 1 < v.size(); i += 2) w.push_back(add_xyzz(v[i], v[i + 1]));
      if (v.size() & 1) w.push_back(v.back());
      v.swap(w);
    }
    if (!v.empty()) acc = v[0];
  } else if (mode == 2) {
    for (size_t i = 0; i < n; ++i) if (!inf || !inf[i]) acc = add_xyzz(acc, from_affine(ld_aff<F>(pts + W2 * i)));
  } else if (mode == 3) {
    for (size_t i = 0; i < n; ++i) if (!inf || !inf[i]) acc = add_xyzz(double_affine(ld_aff<F>(pts + W2 * i)), acc);
  } else if (mode == 4) {
    acc = from_affine(ld_aff<F>(pts));
    for (size_t i = 0; i < n; ++i) acc = double_xyzz(acc);
  } else if (mode == 9) {
    // the streaming formulas IN PLACE on their first operand, as the device runs them (k_gather_sum, k_hot_sum, k_hot_fold: the
    // running sum lives in its output slot): out aliases p, every coordinate of p is read before the same coordinate is written
    for (size_t i = 0; i < n; ++i) if (!inf || !inf[i]) {
      XYZZ<F> q = from_affine(ld_aff<F>(pts + W2 * i));
      MemPt<F> P{&acc}, Q{&q}, O{&acc};
      add_xyzz_stream<F>(P, Q, O);
    }
  } else if (mode == 11) {
    // the short-input kernel's shape (msm_small_kernels.h): task sums by add_mixed_signed (X not value-reduced), plane i = task sum i
    // doubled i times as it is (a plane of a two-bucket range is a COPY of a task sum), the planes added by an add_xyzz tree:
    // sum_i 2^i (P_{2i} - P_{2i+1})
    std::vector<XYZZ<F>> v;
    for (size_t i = 0; i + 1 < n; i += 2) {
      XYZZ<F> part = XYZZ<F>::identity();
      if (!(inf && inf[i])) part = add_mixed_signed(part, ld_aff<F>(pts + W2 * i), false);
      if (!(inf && inf[i + 1])) part = add_mixed_signed(part, ld_aff<F>(pts + W2 * (i + 1)), true);
      for (size_t k = 0; k < i / 2; ++k) part = double_xyzz(part);
      v.push_back(part);
    }
    while (v.size() > 1) {
      std::vector<XYZZ<F>> w;
      for (size_t i = 0; i + 1 < v.size(); i += 2) w.push_back(add_xyzz(v[i], v[i + 1]));
      if (v.size() & 1) w.push_back(v.back());
      v.swap(w);
    }
    if (!v.empty()) acc = v[0];
  } else if (mode == 14 || mode == 15) {                   // 12 / 13 through the register form of the steps (what the device runs)
    std::vector<XYZZ<F>> v;
    XYZZ<F> part = XYZZ<F>::identity();
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
      if (!(inf && inf[i])) {
        Affine<F> a = ld_aff<F>(pts + W2 * i);
        part = (i & 1) ? add_mixed_signed(part, neg_affine(a), true) : add_mixed_signed(part, a, false);
      }
      if (++cnt == 3 || i + 1 == n) { v.push_back(part); part = XYZZ<F>::identity(); cnt = 0; }
    }
    for (size_t s_ = 1; s_ < v.size(); s_ <<= 1)
      for (size_t i = 0; i + s_ < v.size(); i += 2 * s_) host_coop_add_regs(v, (uint32_t)i, (uint32_t)(i + s_), (uint32_t)i);
    if (!v.empty()) {
      if (mode == 15) { host_coop_dbl_regs(v, 0); host_coop_dbl_regs(v, 0); }
      acc = v[0];
    }
  } else if (mode == 12 || mode == 13) {
    // coop_add.h: 12 = chunks of three folded with add_mixed_signed (X not value-reduced), then a pairwise tree of COOPERATIVE additions
    // in place on the left operand (the kernels' merge / halving / combine levels); 13 = the same tree, then the sum doubled twice by the
    // cooperative doubling: 4 * sum
    std::vector<XYZZ<F>> v;
    XYZZ<F> part = XYZZ<F>::identity();
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
      if (!(inf && inf[i])) {
        Affine<F> a = ld_aff<F>(pts + W2 * i);
        part = (i & 1) ? add_mixed_signed(part, neg_affine(a), true) : add_mixed_signed(part, a, false);
      }
      if (++cnt == 3 || i + 1 == n) { v.push_back(part); part = XYZZ<F>::identity(); cnt = 0; }
    }
    for (size_t s_ = 1; s_ < v.size(); s_ <<= 1)
      for (size_t i = 0; i + s_ < v.size(); i += 2 * s_) host_coop_add(v, (uint32_t)i, (uint32_t)(i + s_), (uint32_t)i);
    if (!v.empty()) {
      if (mode == 13) { host_coop_dbl(v, 0); host_coop_dbl(v, 0); }
      acc = v[0];
    }
  } else if (mode == 10) {                                  // and the doubling on its own, in place
    acc = from_affine(ld_aff<F>(pts));
    for (size_t i = 0; i < n; ++i) { MemPt<F> P{&acc}, O{&acc}; double_xyzz_stream<F>(P, O); }
  }
  return st_aff(acc, out_xy);
}
extern "C" int ht_curve_sum(int curve, int checked, int mode, const uint32_t* pts, const uint8_t* inf, size_t n, uint32_t* out_xy) {
  // curve: 0 G1 (Fq), 1 Grumpkin (Fr), 2 G2 (Fq2)
  if (curve == 0) return checked ? curve_sum<FqC>(mode, pts, inf, n, out_xy) : curve_sum<Fq>(mode, pts, inf, n, out_xy);
  if (curve == 1) return checked ? curve_sum<FrC>(mode, pts, inf, n, out_xy) : curve_sum<Fr>(mode, pts, inf, n, out_xy);
  return checked ? curve_sum<Fq2C>(mode, pts, inf, n, out_xy) : curve_sum<Fq2>(mode, pts, inf, n, out_xy);
}

// signed window digits of canonical integers k (msm_digits.h): digits[i * 128 + w], w < W = ceil(255 / c) <= 128; returns W
// GLV decomposition (msm_digits.h): k (canonical, 8 words each) -> |k1|, |k2| (4 words each) and their signs; field: 0 Fr, 1 Fq
extern "C" void ht_glv_decompose(int field, const uint32_t* k, size_t n, uint32_t* k1, uint32_t* k2, uint8_t* neg) {
  for (size_t i = 0; i < n; ++i) {
    bool n1, n2;
    if (field == 0) kg::glv_decompose_with<kg::GlvLattice<kg::FrParams>>(k + 8 * i, k1 + 4 * i, n1, k2 + 4 * i, n2);
    else kg::glv_decompose_with<kg::GlvLattice<kg::FqParams>>(k + 8 * i, k1 + 4 * i, n1, k2 + 4 * i, n2);
    neg[2 * i] = n1; neg[2 * i + 1] = n2;
  }
}
extern "C" int ht_small_digits(const uint32_t* k, size_t n, int c, int32_t* digits) {
  const int W = (255 + c - 1) / c;
  uint32_t H[8];
  small_bias(c, W, H);
  for (size_t i = 0; i < n; ++i) {
    uint32_t kb[8];
    uint64_t cy = 0;
    for (int j = 0; j < 8; ++j) { const uint64_t s_ = (uint64_t)k[8 * i + j] + H[j] + cy; kb[j] = (uint32_t)s_; cy = s_ >> 32; }
    for (int w = 0; w < W; ++w) {
      bool neg;
      const uint32_t m = small_window_digit(kb, w, c, W, neg);
      digits[i * 128 + w] = neg ? -(int32_t)m : (int32_t)m;
    }
  }
  return W;
}

// The whole digit path of a scalar with halved scalars (glv): decomposition, |k_e| + H over W = ceil(128 / c) windows, signed digits with the
// half's sign folded in.  digits: [n][2][64] (window w of half e); returns W.
extern "C" int ht_glv_digits(int field, const uint32_t* k, size_t n, int c, int32_t* digits) {
  const int W = (128 + c - 1) / c;
  uint32_t H[8];
  small_bias(c, W, H);
  for (size_t i = 0; i < n; ++i) {
    uint32_t ks[2][8];
    bool ng[2];
    if (field == 0) kg::glv_decompose_with<kg::GlvLattice<kg::FrParams>>(k + 8 * i, ks[0], ng[0], ks[1], ng[1]);
    else kg::glv_decompose_with<kg::GlvLattice<kg::FqParams>>(k + 8 * i, ks[0], ng[0], ks[1], ng[1]);
    for (int e = 0; e < 2; ++e) {
      uint64_t cy = 0;
      for (int j = 0; j < 4; ++j) { const uint64_t s_ = (uint64_t)ks[e][j] + H[j] + cy; ks[e][j] = (uint32_t)s_; cy = s_ >> 32; }
      for (int j = 4; j < 8; ++j) ks[e][j] = 0;
      for (int w = 0; w < W; ++w) {
        bool neg;
        const uint32_t m = small_window_digit(ks[e], w, c, W, neg);
        digits[(i * 2 + e) * 64 + w] = (neg != ng[e]) ? -(int32_t)m : (int32_t)m;
      }
    }
  }
  return W;
}

// ---- NTT butterfly network (ntt_core.h) ---------------------------------------------------------------
// data: 8 raw elements (any 256-bit value, as loaded by the kernel without a domain change), tw: 7 twiddles in the
// ABI's Montgomery form (stage 1: tw[0]; stage 2: tw[1..2]; stage 3: tw[3..6]); `rounds` networks are chained with a
// norm in between, like the kernel's passes.  Output: elements reduced to canonical form.
template <class F> struct RawIn;
template <class P> struct RawIn<Fp<P>> { static Fp<P> in(const uint32_t* w) { return limbs_from_words<P>(w); } };
template <class P> struct RawIn<FpChecked<P>> { static FpChecked<P> in(const uint32_t* w) { return FpChecked<P>::wrap(limbs_from_words<P>(w), 5.4); } };
template <class P> static void raw_out(const Fp<P>& a, uint32_t* w) { words_from_limbs(reduce_2p(vred(a)), w); }
template <class P> static void raw_out(const FpChecked<P>& a, uint32_t* w) { words_from_limbs(reduce_2p(vred(a)).v, w); }

template <class F>
static void ntt_network(const uint32_t* data, const uint32_t* tw, int rounds, int trivial_first, uint32_t* out) {
  F x[8], w[7];
  for (int k = 0; k < 8; ++k) x[k] = RawIn<F>::in(data + 8 * k);
  for (int k = 0; k < 7; ++k) w[k] = Conv<F>::in(tw + 8 * k);
  for (int r = 0; r < rounds; ++r) {
    struct MontTw {                                  // twiddles in Montgomery form here; the kernel's are Shoup-form constants (hosttest_ntt.cpp)
      const F* w;
      F mul(const F& v, int t, int k0) const { return kg::mul(v, t == 1 ? w[0] : (t == 2 ? w[1 + (k0 & 1)] : w[3 + (k0 & 3)])); }
    } tw{w};
    dit_network<3>(x, trivial_first && r == 0, tw);
    for (int k = 0; k < 8; ++k) x[k] = norm(x[k]);
  }
  for (int k = 0; k < 8; ++k) raw_out(x[k], out + 8 * k);
}
extern "C" void ht_ntt_network(int checked, const uint32_t* data, const uint32_t* tw, int rounds, int trivial_first, uint32_t* out) {
  if (checked) ntt_network<FrC>(data, tw, rounds, trivial_first, out); else ntt_network<Fr>(data, tw, rounds, trivial_first, out);
}

// ---- vector kernels (vecops.h) ------------------------------------------------------------------------
// Nova cross term of one row from dense "rows": az1 = sum_k va[k] * z1[k] etc. over cnt terms (the kernel's row_dot with
// every lane's terms folded in sequence, then merged pairwise like the shuffle tree), then cross_term_row.
// va, vb, vc, z1, z2: cnt elements each in the ABI form; u1, u2: one element each.  out: canonical ABI form.
template <class F>
static void cross_term(const uint32_t* va, const uint32_t* vb, const uint32_t* vc, const uint32_t* z1, const uint32_t* z2, size_t cnt,
                       const uint32_t* u1, const uint32_t* u2, uint32_t* out) {
  using P = typename F::Params;
  auto dot = [&](const uint32_t* v, const uint32_t* z) {
    F lane[4] = {F::zero(), F::zero(), F::zero(), F::zero()};
    for (size_t k = 0; k < cnt; ++k) lane[k & 3] = dot_step(lane[k & 3], RawIn<F>::in(z + 8 * k), RawIn<F>::in(v + 8 * k));
    return dot_merge(dot_merge(lane[0], lane[1]), dot_merge(lane[2], lane[3]));
  };
  const F ku = F::from_const(P::C_XT_U);                 // u * 2^256 -> u * 2^266 (the kernel gets it from ten host-side doublings)
  const F r = cross_term_row(dot(va, z1), dot(va, z2), dot(vb, z1), dot(vb, z2), dot(vc, z1), dot(vc, z2), mul(RawIn<F>::in(u1), ku),
                             mul(RawIn<F>::in(u2), ku), F::from_const(P::C_XT_HAD));
  raw_out(r, out);
}
extern "C" void ht_cross_term(int field, int checked, const uint32_t* va, const uint32_t* vb, const uint32_t* vc, const uint32_t* z1,
                              const uint32_t* z2, size_t cnt, const uint32_t* u1, const uint32_t* u2, uint32_t* out) {
  if (field == 0) { if (checked) cross_term<FrC>(va, vb, vc, z1, z2, cnt, u1, u2, out); else cross_term<Fr>(va, vb, vc, z1, z2, cnt, u1, u2, out); }
  else { if (checked) cross_term<FqC>(va, vb, vc, z1, z2, cnt, u1, u2, out); else cross_term<Fq>(va, vb, vc, z1, z2, cnt, u1, u2, out); }
}
