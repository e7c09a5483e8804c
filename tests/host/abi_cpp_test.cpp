// abi_cpp_test.cpp -- the C++ mirror of the reference's call sites (include/kogarashi_amd.hpp) against the oracle's C
// restatement, on the GPU box: msm_curve_addition (G1 and Grumpkin, identity bases and zero scalars mixed in), Fft (all five
// operations, ragged input), PedersenCommitment (against the reference's naive fold), Prover::create_proof on the chain
// circuit (CRS from the oracle's setup) and the ProverSubVersionCrsAttack error, SparseMatrix::prod and Nova's cross term on both
// scalar fields, the commitment key spread over two contexts.  Test infrastructure: built and run by
// tests/test_gpu_cpp_host.py; links libkogarashi_amd.so and liboracle.so.
#include <cstdio>
#include <cstring>
#include "../../include/kogarashi_amd.hpp"

typedef uint64_t u64;
extern "C" {
void kgo_gen_scalars(int fd, u64 seed, size_t start, size_t n, u64* out);
void kgo_gen_bases(int curve, u64 seed, size_t start, size_t n, u64* out);
void kgo_f_consts(int fd, u64* out17);
void kgo_f_add(int fd, const u64* a, const u64* b, u64* o);
void kgo_f_mul(int fd, const u64* a, const u64* b, u64* o);
void g1_kgo_msm(const u64* bases, const unsigned char* inf, const u64* scalars, size_t n, u64* out, int threads);
void gk_kgo_msm(const u64* bases, const unsigned char* inf, const u64* scalars, size_t n, u64* out, int threads);
void g1_kgo_to_affine(const u64* p, u64* xy, unsigned char* inf);
void gk_kgo_to_affine(const u64* p, u64* xy, unsigned char* inf);
void g1_kgo_commit_naive(const u64* bases, const unsigned char* inf, const u64* scalars, size_t n, u64* out_xy, unsigned char* out_inf);
struct kgo_fft;
kgo_fft* kgo_fft_new(int k);
void kgo_fft_free(kgo_fft* f);
void kgo_fft_dft(const kgo_fft* f, u64* data, int threads);
void kgo_fft_idft(const kgo_fft* f, u64* data, int threads);
void kgo_fft_coset_dft(const kgo_fft* f, u64* data, int threads);
void kgo_fft_coset_idft(const kgo_fft* f, u64* data, int threads);
void kgo_fft_divide_by_z_on_coset(const kgo_fft* f, u64* data);
int kgo_groth16_setup_scalars(const u64* a_rp, const u64* a_col, const u64* a_val, const u64* b_rp, const u64* b_col, const u64* b_val,
                              const u64* c_rp, const u64* c_col, const u64* c_val, size_t m, size_t l, size_t m_l_1, const u64* toxic,
                              u64* h_s, u64* l_s, u64* a_s, u64* b_s, u64* ic_s);
void kgo_fixed_base_mul(int curve, const u64* k, size_t n, u64* xy, unsigned char* inf, int threads);
void kgo_r1cs_evaluate(const u64* row_ptr, const u64* col, const u64* val, size_t m, const u64* z, u64* out);
void kgo_matrix_prod(int fd, const u64* row_ptr, const u64* col, const u64* val, size_t m, const u64* z, u64* out);
void kgo_nova_cross_term(int fd, const u64* a_rp, const u64* a_col, const u64* a_val, const u64* b_rp, const u64* b_col, const u64* b_val,
                         const u64* c_rp, const u64* c_col, const u64* c_val, size_t m, const u64* z1, const u64* z2, const u64* u1, const u64* u2, u64* out);
int kgo_groth16_prove(const u64* a_ev, const u64* b_ev, const u64* c_ev, size_t m, const u64* x, size_t l, const u64* w, size_t m_l_1,
                      const u64* h, const unsigned char* h_inf, const u64* lq, const unsigned char* l_inf, const u64* a, const unsigned char* a_inf,
                      const u64* bg1, const unsigned char* bg1_inf, const u64* bg2, const unsigned char* bg2_inf, const u64* vk_g1, const u64* vk_g2,
                      int delta_is_identity, const u64* r, const u64* s, u64* proof_out, unsigned char* proof_inf, int threads);
}

using namespace kogarashi;
static const u64 SEED = 0x4B6F676172617368ull;
static int failures = 0;
#define CHECK(cond, what)                                                 \
  do {                                                                    \
    if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } \
  } while (0)

static std::vector<Fe> scalars(int fd, u64 seed, size_t n) {
  std::vector<Fe> v(n);
  kgo_gen_scalars(fd, seed, 0, n, reinterpret_cast<u64*>(v.data()));
  return v;
}
static std::vector<G1Affine> points(const u64* xy, const unsigned char* inf, size_t n) {
  std::vector<G1Affine> p(n);
  for (size_t i = 0; i < n; ++i) {
    std::memcpy(p[i].x.data(), xy + 8 * i, 32);
    std::memcpy(p[i].y.data(), xy + 8 * i + 4, 32);
    p[i].is_infinity = inf && inf[i];
  }
  return p;
}

static void test_msm(const Context& ctx, int curve) {
  const size_t n = 5000;
  std::vector<u64> xy(8 * n);
  kgo_gen_bases(curve, SEED + 11 + curve, 0, n, xy.data());
  std::vector<unsigned char> inf(n, 0);
  inf[3] = inf[n - 1] = 1;
  std::vector<Fe> k = scalars(curve == KG_G1 ? 0 : 1, SEED + 12, n + 7);      // longer than the bases: the reference zips
  k[5] = Fe{0, 0, 0, 0};
  const G1Projective got = msm_curve_addition(ctx, points(xy.data(), inf.data(), n), k, curve);
  u64 proj[12], want_xy[8];
  unsigned char want_inf = 0;
  (curve == KG_G1 ? g1_kgo_msm : gk_kgo_msm)(xy.data(), inf.data(), reinterpret_cast<const u64*>(k.data()), n, proj, 4);
  (curve == KG_G1 ? g1_kgo_to_affine : gk_kgo_to_affine)(proj, want_xy, &want_inf);
  u64 consts[17];
  kgo_f_consts(curve == KG_G1 ? 1 : 0, consts);          // base field of the curve: R mod p is the ABI's z = 1
  CHECK(!want_inf && !std::memcmp(got.x.data(), want_xy, 32) && !std::memcmp(got.y.data(), want_xy + 4, 32), "msm_curve_addition affine sum");
  CHECK(!std::memcmp(got.z.data(), consts + 5, 32), "msm_curve_addition z = 1");
  // empty input: the additive identity (0, 1, 0)
  const G1Projective id = msm_curve_addition(ctx, {}, {}, curve);
  const Fe zero{0, 0, 0, 0};
  CHECK(id.x == zero && id.z == zero && !std::memcmp(id.y.data(), consts + 5, 32), "msm_curve_addition of nothing");
  // the same bases resident (marshalled and registered once), the scalars a host vector per call
  const ResidentBases rb(ctx, points(xy.data(), inf.data(), n), curve);
  for (int rep = 0; rep < 2; ++rep) {
    const G1Projective r2 = rb.msm(k);
    CHECK(r2.x == got.x && r2.y == got.y && r2.z == got.z, "ResidentBases::msm");
  }
}

static void test_fft(const Context& ctx, int k) {
  const size_t n = (size_t)1 << k;
  const std::vector<Fe> v = scalars(0, SEED + 20 + k, n - 3);                 // ragged: zero padded like prepare_fft
  kgo_fft* fo = kgo_fft_new(k);
  Fft f(ctx, k);
  struct { const char* name; std::vector<Fe> (Fft::*fn)(const std::vector<Fe>&) const; void (*ref)(const kgo_fft*, u64*, int); } ops[] = {
      {"dft", &Fft::dft, kgo_fft_dft}, {"idft", &Fft::idft, kgo_fft_idft}, {"coset_dft", &Fft::coset_dft, kgo_fft_coset_dft}, {"coset_idft", &Fft::coset_idft, kgo_fft_coset_idft}};
  for (auto& op : ops) {
    std::vector<Fe> want(n, Fe{0, 0, 0, 0});
    std::copy(v.begin(), v.end(), want.begin());
    op.ref(fo, reinterpret_cast<u64*>(want.data()), 4);
    CHECK((f.*op.fn)(v) == want, op.name);
  }
  std::vector<Fe> want(n, Fe{0, 0, 0, 0});
  std::copy(v.begin(), v.end(), want.begin());
  kgo_fft_divide_by_z_on_coset(fo, reinterpret_cast<u64*>(want.data()));
  CHECK(f.divide_by_z_on_coset(v) == want, "divide_by_z_on_coset");
  CHECK(f.idft(f.dft(v)) == [&] { std::vector<Fe> p(n, Fe{0, 0, 0, 0}); std::copy(v.begin(), v.end(), p.begin()); return p; }(), "idft(dft(v)) = v");
  kgo_fft_free(fo);
}

static void test_commit(const Context& ctx) {
  const size_t n = 700;
  std::vector<u64> xy(8 * n);
  kgo_gen_bases(0, SEED + 31, 0, n, xy.data());
  const std::vector<Fe> m = scalars(0, SEED + 32, n - 10);
  PedersenCommitment pc(ctx, points(xy.data(), nullptr, n));
  const G1Affine got = pc.commit(m);
  u64 want[8];
  unsigned char winf = 0;
  g1_kgo_commit_naive(xy.data(), nullptr, reinterpret_cast<const u64*>(m.data()), n - 10, want, &winf);
  CHECK(got.is_infinity == (winf != 0) && !std::memcmp(got.x.data(), want, 32) && !std::memcmp(got.y.data(), want + 4, 32), "PedersenCommitment::commit");
  const G1Affine again = pc.commit(m);
  CHECK(again.x == got.x && again.y == got.y, "commit twice");
}

static void test_prover(const Context& ctx) {
  // the chain circuit t_{i+1} = t_i (t_i + 1): x = [1, t_0], w = t_1..t_m; constraint i: A = t_i, B = t_i + 1, C = t_{i+1}
  const size_t m = 64, l = 2, m_l_1 = m;
  u64 cf[17];
  kgo_f_consts(0, cf);
  Fe one;
  std::memcpy(one.data(), cf + 5, 32);
  std::vector<Fe> t(m + 1);
  t[0] = scalars(0, SEED + 41, 1)[0];
  for (size_t i = 0; i < m; ++i) {
    Fe s1;
    kgo_f_add(0, t[i].data(), one.data(), s1.data());
    kgo_f_mul(0, t[i].data(), s1.data(), t[i + 1].data());
  }
  std::vector<Fe> x{one, t[0]}, w(t.begin() + 1, t.end());
  auto wire = [](size_t i) -> u64 { return i == 0 ? 1 : 2 + i - 1; };
  std::vector<u64> a_rp(m + 1), a_col(m), b_rp(m + 1), b_col(2 * m), c_rp(m + 1), c_col(m);
  std::vector<Fe> a_val(m, one), b_val(2 * m, one), c_val(m, one);
  for (size_t i = 0; i <= m; ++i) { a_rp[i] = i; b_rp[i] = 2 * i; c_rp[i] = i; }
  for (size_t i = 0; i < m; ++i) { a_col[i] = wire(i); b_col[2 * i] = wire(i); b_col[2 * i + 1] = 0; c_col[i] = wire(i + 1); }
  const std::vector<Fe> toxic = scalars(0, SEED + 42, 5);
  const size_t nv = l + m_l_1;
  std::vector<Fe> hs(m - 1), ls(m_l_1), as(nv), bs(nv), ics(l);
  auto U = [](std::vector<Fe>& v) { return reinterpret_cast<u64*>(v.data()); };
  auto UC = [](const std::vector<Fe>& v) { return reinterpret_cast<const u64*>(v.data()); };
  CHECK(kgo_groth16_setup_scalars(a_rp.data(), a_col.data(), UC(a_val), b_rp.data(), b_col.data(), UC(b_val), c_rp.data(), c_col.data(), UC(c_val), m, l,
                                  m_l_1, UC(toxic), U(hs), U(ls), U(as), U(bs), U(ics)) == 0, "oracle setup");
  auto g1 = [&](const std::vector<Fe>& k, std::vector<u64>& xy, std::vector<unsigned char>& inf) {
    xy.assign(8 * k.size(), 0); inf.assign(k.size(), 0);
    kgo_fixed_base_mul(0, UC(k), k.size(), xy.data(), inf.data(), 4);
  };
  std::vector<u64> h_xy, l_xy, a_xy, b1_xy, b2_xy(16 * nv), vk1(24), vk2(48);
  std::vector<unsigned char> h_inf, l_inf, a_inf, b1_inf, b2_inf(nv), vi(3);
  g1(hs, h_xy, h_inf); g1(ls, l_xy, l_inf); g1(as, a_xy, a_inf); g1(bs, b1_xy, b1_inf);
  kgo_fixed_base_mul(2, UC(bs), nv, b2_xy.data(), b2_inf.data(), 4);
  const std::vector<Fe> k1{toxic[0], toxic[1], toxic[3]}, k2{toxic[1], toxic[3], toxic[2]};
  kgo_fixed_base_mul(0, UC(k1), 3, vk1.data(), vi.data(), 1);
  kgo_fixed_base_mul(2, UC(k2), 3, vk2.data(), vi.data(), 1);
  const std::vector<Fe> z = [&] { std::vector<Fe> v(x); v.insert(v.end(), w.begin(), w.end()); return v; }();
  std::vector<Fe> ae(m), be(m), ce(m);
  kgo_r1cs_evaluate(a_rp.data(), a_col.data(), UC(a_val), m, UC(z), U(ae));
  kgo_r1cs_evaluate(b_rp.data(), b_col.data(), UC(b_val), m, UC(z), U(be));
  kgo_r1cs_evaluate(c_rp.data(), c_col.data(), UC(c_val), m, UC(z), U(ce));
  const std::vector<Fe> rs = scalars(0, SEED + 43, 2);
  u64 want[32];
  unsigned char winf[3];
  CHECK(kgo_groth16_prove(UC(ae), UC(be), UC(ce), m, UC(x), l, UC(w), m_l_1, h_xy.data(), h_inf.data(), l_xy.data(), l_inf.data(), a_xy.data(), a_inf.data(),
                          b1_xy.data(), b1_inf.data(), b2_xy.data(), b2_inf.data(), vk1.data(), vk2.data(), 0, rs[0].data(), rs[1].data(), want, winf, 4) == 0,
        "oracle create_proof");
  Parameters P;
  P.h = points(h_xy.data(), h_inf.data(), m - 1); P.l = points(l_xy.data(), l_inf.data(), m_l_1);
  P.a = points(a_xy.data(), a_inf.data(), nv); P.b_g1 = points(b1_xy.data(), b1_inf.data(), nv);
  P.b_g2.resize(nv);
  for (size_t i = 0; i < nv; ++i) {
    std::memcpy(P.b_g2[i].x0.data(), &b2_xy[16 * i], 32); std::memcpy(P.b_g2[i].x1.data(), &b2_xy[16 * i + 4], 32);
    std::memcpy(P.b_g2[i].y0.data(), &b2_xy[16 * i + 8], 32); std::memcpy(P.b_g2[i].y1.data(), &b2_xy[16 * i + 12], 32);
    P.b_g2[i].is_infinity = b2_inf[i];
  }
  P.alpha_g1 = points(vk1.data(), nullptr, 3)[0]; P.beta_g1 = points(vk1.data(), nullptr, 3)[1]; P.delta_g1 = points(vk1.data(), nullptr, 3)[2];
  auto g2at = [&](size_t i) { G2Affine q; std::memcpy(q.x0.data(), &vk2[16 * i], 32); std::memcpy(q.x1.data(), &vk2[16 * i + 4], 32);
                              std::memcpy(q.y0.data(), &vk2[16 * i + 8], 32); std::memcpy(q.y1.data(), &vk2[16 * i + 12], 32); return q; };
  P.beta_g2 = g2at(0); P.delta_g2 = g2at(1);
  {
    Prover prover(ctx, P, m, l, m_l_1);
    const Proof pr = prover.create_proof(ae, be, ce, x, w, rs[0], rs[1]);
    CHECK(!std::memcmp(pr.a.x.data(), want, 32) && !std::memcmp(pr.a.y.data(), want + 4, 32) && pr.a.is_infinity == (winf[0] != 0), "proof.a");
    CHECK(!std::memcmp(pr.b.x0.data(), want + 8, 32) && !std::memcmp(pr.b.x1.data(), want + 12, 32) && !std::memcmp(pr.b.y0.data(), want + 16, 32) &&
          !std::memcmp(pr.b.y1.data(), want + 20, 32), "proof.b");
    CHECK(!std::memcmp(pr.c.x.data(), want + 24, 32) && !std::memcmp(pr.c.y.data(), want + 28, 32), "proof.c");
  }
  {
    // ZkSnark::setup behind the boundary: the same Parameters and VerifyingKey from the matrices and the five toxic scalars
    SparseMatrix A{a_rp, a_col, a_val}, B{b_rp, b_col, b_val}, C{c_rp, c_col, c_val};
    const auto made = setup(ctx, A, B, C, l, m_l_1, std::array<Fe, 5>{toxic[0], toxic[1], toxic[2], toxic[3], toxic[4]});
    const Parameters& Q = made.first;
    auto same1 = [](const std::vector<G1Affine>& u, const std::vector<G1Affine>& v) {
      if (u.size() != v.size()) return false;
      for (size_t i = 0; i < u.size(); ++i) if (u[i].x != v[i].x || u[i].y != v[i].y || u[i].is_infinity != v[i].is_infinity) return false;
      return true;
    };
    CHECK(same1(Q.h, P.h) && same1(Q.l, P.l) && same1(Q.a, P.a) && same1(Q.b_g1, P.b_g1), "setup: G1 vectors of Parameters");
    bool g2ok = Q.b_g2.size() == P.b_g2.size();
    for (size_t i = 0; g2ok && i < nv; ++i)
      g2ok = Q.b_g2[i].x0 == P.b_g2[i].x0 && Q.b_g2[i].x1 == P.b_g2[i].x1 && Q.b_g2[i].y0 == P.b_g2[i].y0 && Q.b_g2[i].y1 == P.b_g2[i].y1 &&
             Q.b_g2[i].is_infinity == P.b_g2[i].is_infinity;
    CHECK(g2ok, "setup: b_g2");
    CHECK(Q.alpha_g1.x == P.alpha_g1.x && Q.beta_g1.y == P.beta_g1.y && Q.delta_g1.x == P.delta_g1.x && Q.beta_g2.x1 == P.beta_g2.x1 &&
          Q.delta_g2.y0 == P.delta_g2.y0, "setup: vk points of Parameters");
    const G2Affine gamma = g2at(2);
    CHECK(made.second.gamma_g2.x0 == gamma.x0 && made.second.gamma_g2.y1 == gamma.y1, "setup: gamma_g2");
    std::vector<u64> ic_xy;
    std::vector<unsigned char> ic_inf;
    g1(ics, ic_xy, ic_inf);
    CHECK(same1(made.second.ic, points(ic_xy.data(), ic_inf.data(), l)), "setup: vk.ic");
    bool inv_threw = false;
    try {
      setup(ctx, A, B, C, l, m_l_1, std::array<Fe, 5>{toxic[0], toxic[1], Fe{0, 0, 0, 0}, toxic[3], toxic[4]});
    } catch (const ProverInversionFailed&) { inv_threw = true; }
    CHECK(inv_threw, "Error::ProverInversionFailed");
  }
  P.delta_g1.is_infinity = true;                       // prover.rs:67-69
  bool threw = false;
  try {
    Prover(ctx, P, m, l, m_l_1).create_proof(ae, be, ce, x, w, rs[0], rs[1]);
  } catch (const ProverSubVersionCrsAttack&) { threw = true; }
  CHECK(threw, "Error::ProverSubVersionCrsAttack");
}

// a random sparse system: rows of 0..5 entries (one row empty, one long), both scalar fields
static void test_nova(const Context& ctx, int field) {
  const size_t m = 1500, nz = 300;
  unsigned long long st = 0x9e3779b97f4a7c15ull + field;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  SparseMatrix M[3];
  for (int k = 0; k < 3; ++k) {
    M[k].row_ptr.push_back(0);
    for (size_t i = 0; i < m; ++i) {
      size_t cnt = rnd() % 6;
      if (i == 7) cnt = 0;
      if (i == 9) cnt = 70;
      for (size_t e = 0; e < cnt; ++e) M[k].col.push_back(rnd() % nz);
      M[k].row_ptr.push_back(M[k].col.size());
    }
    M[k].val = scalars(field, SEED + 50 + k, M[k].col.size());
  }
  const std::vector<Fe> z1 = scalars(field, SEED + 60, nz), z2 = scalars(field, SEED + 61, nz), us = scalars(field, SEED + 62, 2);
  R1csShape shape(ctx, M[0], M[1], M[2], field);
  auto UC = [](const std::vector<Fe>& v) { return reinterpret_cast<const u64*>(v.data()); };
  std::vector<Fe> want(m);
  kgo_matrix_prod(field, M[1].row_ptr.data(), M[1].col.data(), UC(M[1].val), m, UC(z1), reinterpret_cast<u64*>(want.data()));
  CHECK(shape.prod(1, z1) == want, "SparseMatrix::prod");
  kgo_nova_cross_term(field, M[0].row_ptr.data(), M[0].col.data(), UC(M[0].val), M[1].row_ptr.data(), M[1].col.data(), UC(M[1].val), M[2].row_ptr.data(),
                      M[2].col.data(), UC(M[2].val), m, UC(z1), UC(z2), us[0].data(), us[1].data(), reinterpret_cast<u64*>(want.data()));
  CHECK(shape.compute_cross_term(z1, z2, us[0], us[1]) == want, "Prover::compute_cross_term");
  bool threw = false;
  try { shape.prod(0, std::vector<Fe>(nz - 200)); } catch (const std::out_of_range&) { threw = true; }
  CHECK(threw, "short z is refused");
  // RelaxedR1csWitness::fold (witness.rs:56-70): W1 + r W2 element by element
  std::vector<Fe> folded(nz);
  for (size_t i = 0; i < nz; ++i) {
    Fe t;
    kgo_f_mul(field, us[0].data(), z2[i].data(), t.data());
    kgo_f_add(field, z1[i].data(), t.data(), folded[i].data());
  }
  CHECK(fold(ctx, z1, us[0], z2, field) == folded, "RelaxedR1csWitness::fold");
}

// the commitment key over two contexts (sharing device 0 on a one-GPU box) against the single-context commitment
static void test_sharded_commit(const Context& ctx) {
  const size_t n = 9001;
  std::vector<u64> xy(8 * n);
  kgo_gen_bases(0, SEED + 71, 0, n, xy.data());
  const std::vector<G1Affine> g = points(xy.data(), nullptr, n);
  const std::vector<Fe> m = scalars(0, SEED + 72, n);
  Context second(0);
  ShardedPedersenCommitment key({&ctx, &second}, g);
  CHECK(key.len() == n, "sharded key length");
  const G1Affine a = key.commit(m), b = PedersenCommitment(ctx, g).commit(m);
  CHECK(a.x == b.x && a.y == b.y && a.is_infinity == b.is_infinity, "sharded commit = single-context commit");
}

int main() {
  kg_init();
  try {
    Context ctx(0);
    test_msm(ctx, KG_G1);
    test_msm(ctx, KG_GRUMPKIN);
    test_fft(ctx, 5);
    test_fft(ctx, 13);
    test_commit(ctx);
    test_prover(ctx);
    test_nova(ctx, KG_FR);
    test_nova(ctx, KG_FQ);
    test_sharded_commit(ctx);
  } catch (const std::exception& e) {
    std::printf("FAIL exception: %s\n", e.what());
    return 2;
  }
  std::printf(failures ? "FAILED: %d checks\n" : "ok: msm_curve_addition, Fft, PedersenCommitment, Prover, R1csShape, ShardedPedersenCommitment match the oracle\n", failures);
  return failures ? 1 : 0;
}
