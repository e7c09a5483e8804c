// kogarashi_amd.hpp -- C++ host-side mirror of the reference's call sites for the hot path, header-only, over the C ABI of
// kogarashi_amd.h.  The reference is compiled code (Rust) and its toolchain is absent from the build image, so this is the
// compiled-language counterpart of rust/kogarashi-amd: the same names, argument meaning and error behaviour as
//
//   msm_curve_addition(bases, coeffs)                                    groth16/src/msm.rs:6-48
//   Fft(k).dft / idft / coset_dft / coset_idft / divide_by_z_on_coset    groth16/src/fft.rs:27-154
//   PedersenCommitment(g).commit(m)                                      nova/src/pedersen.rs:6-20
//   Prover(params).create_proof(a, b, c, x, w, r, s)                     groth16/src/prover.rs:14-99
//   R1csShape(a, b, c).prod / compute_cross_term(z1, z2, u1, u2)         zkstd/src/matrix.rs:36-48, nova/src/prover.rs:53-90
//   ShardedPedersenCommitment(ctxs, g).commit(m)                         nova/src/pedersen.rs:6-20 over every GPU of the node
//
// with the reference's in-memory data: a field element is 4 x uint64 Montgomery limbs (Fr / Fq), an affine point x | y plus an
// identity flag.  Nothing here computes: every call goes to libkogarashi_amd.so; without the library or a device the
// constructors throw (there is no CPU path).  tests/host/abi_cpp_test.cpp drives it against the oracle on the GPU box.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "kogarashi_amd.h"

namespace kogarashi {

using Fe = std::array<uint64_t, 4>;        // Fr or Fq, Montgomery form (bn254/src/fr.rs:71, fq.rs:48)

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& what) : std::runtime_error(what + ": " + kg_strerror(st)), status(st) {}
};
// groth16::Error::ProverSubVersionCrsAttack (groth16/src/error.rs:2-8, prover.rs:67-69)
struct ProverSubVersionCrsAttack : Error {
  ProverSubVersionCrsAttack() : Error(KG_ERR_CRS, "create_proof") {}
};

// affine points as the reference holds them (bn254/src/g1.rs:18-22, g2.rs:16-20): coordinates + is_infinity
struct G1Affine { Fe x, y; bool is_infinity = false; };
struct G2Affine { Fe x0, x1, y0, y1; bool is_infinity = false; };
// msm_curve_addition's result: the projective sum normalised to z = 1, or (0, 1, 0)
struct G1Projective { Fe x, y, z; };

class Context {
 public:
  explicit Context(int device = 0) {
    const int rc = kg_ctx_create(device, &ctx_);
    if (rc != KG_OK) throw Error(rc, "kg_ctx_create (libkogarashi_amd needs a visible MI355X; there is no CPU path)");
  }
  ~Context() { if (ctx_) kg_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  kg_ctx* raw() const { return ctx_; }
  void check(int rc, const char* what) const {
    if (rc == KG_ERR_CRS) throw ProverSubVersionCrsAttack();
    if (rc != KG_OK) throw Error(rc, std::string(what) + " [" + kg_last_error(ctx_) + "]");
  }

 private:
  kg_ctx* ctx_ = nullptr;
};

// device allocation owned by a context
class DeviceBuffer {
 public:
  DeviceBuffer(const Context& c, size_t bytes) : c_(c), bytes_(bytes) { c.check(kg_malloc(c.raw(), bytes ? bytes : 1, &p_), "kg_malloc"); }
  DeviceBuffer(const Context& c, const void* host, size_t bytes) : DeviceBuffer(c, bytes) {
    if (bytes) c.check(kg_memcpy_h2d(c.raw(), p_, host, bytes), "kg_memcpy_h2d");
  }
  ~DeviceBuffer() { if (p_) kg_free(c_.raw(), p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  template <class T> T* as() const { return static_cast<T*>(p_); }
  void download(void* host) const { if (bytes_) c_.check(kg_memcpy_d2h(c_.raw(), host, p_, bytes_), "kg_memcpy_d2h"); }
  size_t bytes() const { return bytes_; }

 private:
  const Context& c_;
  void* p_ = nullptr;
  size_t bytes_;
};

namespace detail {
inline void marshal(const std::vector<G1Affine>& pts, size_t n, std::vector<uint64_t>& xy, std::vector<uint8_t>& inf) {
  xy.resize(8 * n); inf.resize(n);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) { xy[8 * i + k] = pts[i].x[k]; xy[8 * i + 4 + k] = pts[i].y[k]; }
    inf[i] = pts[i].is_infinity;
  }
}
inline void marshal(const std::vector<G2Affine>& pts, size_t n, std::vector<uint64_t>& xy, std::vector<uint8_t>& inf) {
  xy.resize(16 * n); inf.resize(n);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) {
      xy[16 * i + k] = pts[i].x0[k]; xy[16 * i + 4 + k] = pts[i].x1[k];
      xy[16 * i + 8 + k] = pts[i].y0[k]; xy[16 * i + 12 + k] = pts[i].y1[k];
    }
    inf[i] = pts[i].is_infinity;
  }
}
inline G1Affine g1_from(const uint64_t* w, bool inf) {
  G1Affine p;
  for (int k = 0; k < 4; ++k) { p.x[k] = w[k]; p.y[k] = w[4 + k]; }
  p.is_infinity = inf;
  return p;
}
inline G2Affine g2_from(const uint64_t* w, bool inf) {
  G2Affine p;
  for (int k = 0; k < 4; ++k) { p.x0[k] = w[k]; p.x1[k] = w[4 + k]; p.y0[k] = w[8 + k]; p.y1[k] = w[12 + k]; }
  p.is_infinity = inf;
  return p;
}
}  // namespace detail

// groth16/src/msm.rs:6 msm_curve_addition(bases, coeffs): zips (min of the lengths), identity bases are no-ops.
// curve: KG_G1 (Fr scalars) or KG_GRUMPKIN (Fq scalars); host arrays in, one point out (kg_msm_host).
inline G1Projective msm_curve_addition(const Context& c, const std::vector<G1Affine>& bases, const std::vector<Fe>& coeffs, int curve = KG_G1) {
  const size_t n = bases.size() < coeffs.size() ? bases.size() : coeffs.size();
  std::vector<uint64_t> xy;
  std::vector<uint8_t> inf;
  detail::marshal(bases, n, xy, inf);
  uint64_t out[12];
  c.check(kg_msm_host(c.raw(), curve, xy.data(), inf.data(), reinterpret_cast<const uint64_t*>(coeffs.data()), n, out), "kg_msm_host");
  G1Projective r;
  for (int k = 0; k < 4; ++k) { r.x[k] = out[k]; r.y[k] = out[4 + k]; r.z[k] = out[8 + k]; }
  return r;
}

// The same call against bases that stay on the device: what a Rust host does behind `msm_curve_addition` for a CRS vector or any
// other slice it meets again (rust/kogarashi-amd/src/lib.rs: register_bases / the address-keyed cache) -- marshal, upload and
// kg_bases_register ONCE, then every call moves only its scalars (kg_msm_host_scalars: index slices uploaded under the accumulations).
class ResidentBases {
 public:
  ResidentBases(const Context& c, const std::vector<G1Affine>& bases, int curve = KG_G1) : c_(c), curve_(curve), n_(bases.size()) {
    std::vector<uint64_t> xy;
    std::vector<uint8_t> inf;
    detail::marshal(bases, n_, xy, inf);
    bool any = false;
    for (uint8_t f : inf) any = any || f;
    g_.reset(new DeviceBuffer(c, xy.data(), xy.size() * 8));
    if (any) inf_.reset(new DeviceBuffer(c, inf.data(), inf.size()));
    if (n_) c.check(kg_bases_register(c.raw(), curve, g_->as<uint64_t>(), any ? inf_->as<uint8_t>() : nullptr, n_), "kg_bases_register");
  }
  ~ResidentBases() { if (n_) kg_bases_unregister(c_.raw(), g_->as<uint64_t>()); }
  size_t size() const { return n_; }
  G1Projective msm(const std::vector<Fe>& coeffs) const {
    const size_t n = coeffs.size() < n_ ? coeffs.size() : n_;
    uint64_t out[12];
    c_.check(kg_msm_host_scalars(c_.raw(), curve_, g_->as<uint64_t>(), inf_ ? inf_->as<uint8_t>() : nullptr,
                                 reinterpret_cast<const uint64_t*>(coeffs.data()), n, out), "kg_msm_host_scalars");
    G1Projective r;
    for (int k = 0; k < 4; ++k) { r.x[k] = out[k]; r.y[k] = out[4 + k]; r.z[k] = out[8 + k]; }
    return r;
  }

 private:
  const Context& c_;
  int curve_;
  size_t n_;
  std::unique_ptr<DeviceBuffer> g_, inf_;
};

// nova/src/relaxed_r1cs/witness.rs:56-70 (W = W1 + r W2, E = E1 + r T) and instance.rs:81-101 (x = x1 + r x2): a[i] + s * b[i] on the
// scalar field of either curve (field: KG_FR / KG_FQ); two vectors up, one down
inline std::vector<Fe> fold(const Context& c, const std::vector<Fe>& a, const Fe& s, const std::vector<Fe>& b, int field = KG_FR) {
  if (a.size() != b.size()) throw std::invalid_argument("fold: DenseVectors of different lengths");
  DeviceBuffer da(c, a.data(), a.size() * 32), db(c, b.data(), b.size() * 32);
  c.check(kg_field_vec_axpy(c.raw(), field, da.as<uint64_t>(), s.data(), db.as<uint64_t>(), da.as<uint64_t>(), a.size()), "kg_field_vec_axpy");
  std::vector<Fe> out(a.size());
  da.download(out.data());
  return out;
}

// groth16/src/fft.rs:27-154 Fft<Fr>: 2^k-point transforms, natural order; shorter inputs are zero padded (prepare_fft :157-162)
class Fft {
 public:
  Fft(const Context& c, uint32_t k) : c_(c), k_(k), n_((size_t)1 << k) {
    if (k < 1 || k > 28) throw std::invalid_argument("Fft: 1 <= k <= 28 (S = 28, bn254/src/fr.rs:53)");
  }
  size_t size() const { return n_; }
  std::vector<Fe> dft(const std::vector<Fe>& v) const { return run(v, 0, 0); }
  std::vector<Fe> idft(const std::vector<Fe>& v) const { return run(v, 1, 0); }
  std::vector<Fe> coset_dft(const std::vector<Fe>& v) const { return run(v, 0, 1); }
  std::vector<Fe> coset_idft(const std::vector<Fe>& v) const { return run(v, 1, 1); }
  // the same transforms in place on a vector of exactly 2^k elements -- the shape of the reference's own methods (they consume and
  // return the vector, fft.rs:92-127) and of the Rust glue's fft::transform: no padded copy, no result vector, the host cost is the
  // two bus trips
  void transform_in_place(std::vector<Fe>& v, int inverse, int coset) const {
    if (v.size() != n_) throw std::invalid_argument("Fft::transform_in_place: the vector must hold 2^k elements");
    DeviceBuffer d(c_, v.data(), n_ * 32);
    c_.check(kg_ntt_bn254_fr(c_.raw(), d.as<uint64_t>(), k_, inverse, coset), "kg_ntt_bn254_fr");
    d.download(v.data());
  }
  std::vector<Fe> divide_by_z_on_coset(const std::vector<Fe>& v) const {
    std::vector<Fe> buf = padded(v);
    DeviceBuffer d(c_, buf.data(), n_ * 32);
    c_.check(kg_fr_divide_by_z_on_coset(c_.raw(), d.as<uint64_t>(), k_), "kg_fr_divide_by_z_on_coset");
    d.download(buf.data());
    return buf;
  }

 private:
  std::vector<Fe> padded(const std::vector<Fe>& v) const {
    std::vector<Fe> buf(n_, Fe{0, 0, 0, 0});
    for (size_t i = 0; i < v.size() && i < n_; ++i) buf[i] = v[i];
    return buf;
  }
  std::vector<Fe> run(const std::vector<Fe>& v, int inverse, int coset) const {
    std::vector<Fe> buf = padded(v);
    DeviceBuffer d(c_, buf.data(), n_ * 32);
    c_.check(kg_ntt_bn254_fr(c_.raw(), d.as<uint64_t>(), k_, inverse, coset), "kg_ntt_bn254_fr");
    d.download(buf.data());
    return buf;
  }
  const Context& c_;
  uint32_t k_;
  size_t n_;
};

// nova/src/pedersen.rs:6-20 PedersenCommitment<C> { g }: the generators go to the device once and stay resident in the
// MSM's internal form (kg_bases_register); commit(m) = affine(sum_i m[i] * g[i]) over min(len) pairs.
class PedersenCommitment {
 public:
  PedersenCommitment(const Context& c, const std::vector<G1Affine>& g, int curve = KG_G1) : c_(c), curve_(curve), n_(g.size()) {
    std::vector<uint64_t> xy;
    std::vector<uint8_t> inf;
    detail::marshal(g, n_, xy, inf);
    bool any = false;
    for (uint8_t f : inf) any = any || f;
    g_.reset(new DeviceBuffer(c, xy.data(), xy.size() * 8));
    if (any) inf_.reset(new DeviceBuffer(c, inf.data(), inf.size()));
    c.check(kg_bases_register(c.raw(), curve, g_->as<uint64_t>(), any ? inf_->as<uint8_t>() : nullptr, n_), "kg_bases_register");
  }
  ~PedersenCommitment() { kg_bases_unregister(c_.raw(), g_->as<uint64_t>()); }
  G1Affine commit(const std::vector<Fe>& m) const {
    const size_t n = m.size() < n_ ? m.size() : n_;
    uint64_t xy[8];
    uint8_t inf = 0;
    // the key is resident, m is the caller's slice: uploaded in index slices under the accumulations (no whole-vector copy in front)
    c_.check(kg_commit_host_scalars(c_.raw(), curve_, g_->as<uint64_t>(), inf_ ? inf_->as<uint8_t>() : nullptr,
                                    reinterpret_cast<const uint64_t*>(m.data()), n, xy, &inf), "kg_commit_host_scalars");
    return detail::g1_from(xy, inf != 0);
  }

 private:
  const Context& c_;
  int curve_;
  size_t n_;
  std::unique_ptr<DeviceBuffer> g_, inf_;
};

// groth16/src/params.rs:6-28 Parameters (the prover's part) and groth16/src/proof.rs Proof { a, b, c }
struct Parameters {
  std::vector<G1Affine> h, l, a, b_g1;
  std::vector<G2Affine> b_g2;
  G1Affine alpha_g1, beta_g1, delta_g1;
  G2Affine beta_g2, delta_g2;
};
struct Proof { G1Affine a; G2Affine b; G1Affine c; };

// groth16/src/verifier.rs:20-46 VerifyingKey, as ZkSnark::setup fills it
struct VerifyingKey {
  G1Affine alpha_g1, beta_g1, delta_g1;
  G2Affine beta_g2, gamma_g2, delta_g2;
  std::vector<G1Affine> ic;
};
// groth16::Error::ProverInversionFailed (zksnark.rs:37-38): gamma or delta has no inverse
struct ProverInversionFailed : Error {
  ProverInversionFailed() : Error(KG_ERR_INVERSION, "setup") {}
};

// groth16/src/zksnark.rs:17-127 ZkSnark::setup after circuit synthesis: (Parameters, VerifyingKey) of the circuit whose constraint
// matrices a, b, c are given as CSR over z = x || w (SparseMatrix below), from the five toxic scalars alpha, beta, gamma, delta, tau --
// the reference draws them from its rng in that order (zksnark.rs:28-32) and so does the caller here.  One call:
// kg_groth16_setup_bn254 computes everything on the device.
struct SparseMatrix;
inline std::pair<struct Parameters, VerifyingKey> setup(const Context& c, const SparseMatrix& a, const SparseMatrix& b, const SparseMatrix& cm, size_t l,
                                                        size_t m_l_1, const std::array<Fe, 5>& toxic);

// groth16/src/prover.rs:14-99 Prover { params }: the CRS is uploaded and registered once; create_proof takes the synthesised
// constraint system's evaluations (cs.evaluate()), x = cs.x(), w = cs.w() and the blinding scalars the reference draws from its
// rng (prover.rs:71-72).  Throws ProverSubVersionCrsAttack like prover.rs:67-69.
class Prover {
 public:
  Prover(const Context& c, const Parameters& p, size_t m, size_t l, size_t m_l_1) : c_(c) {
    crs_ = kg_groth16_crs{};
    crs_.m = m; crs_.l = l; crs_.m_l_1 = m_l_1;
    upload(p.h, KG_G1, crs_.d_h, crs_.d_h_inf);
    upload(p.l, KG_G1, crs_.d_l, crs_.d_l_inf);
    upload(p.a, KG_G1, crs_.d_a, crs_.d_a_inf);
    upload(p.b_g1, KG_G1, crs_.d_b_g1, crs_.d_b_g1_inf);
    upload(p.b_g2, KG_G2, crs_.d_b_g2, crs_.d_b_g2_inf);
    for (int k = 0; k < 4; ++k) {
      crs_.alpha_g1[k] = p.alpha_g1.x[k]; crs_.alpha_g1[4 + k] = p.alpha_g1.y[k];
      crs_.beta_g1[k] = p.beta_g1.x[k]; crs_.beta_g1[4 + k] = p.beta_g1.y[k];
      crs_.delta_g1[k] = p.delta_g1.x[k]; crs_.delta_g1[4 + k] = p.delta_g1.y[k];
      crs_.beta_g2[k] = p.beta_g2.x0[k]; crs_.beta_g2[4 + k] = p.beta_g2.x1[k]; crs_.beta_g2[8 + k] = p.beta_g2.y0[k]; crs_.beta_g2[12 + k] = p.beta_g2.y1[k];
      crs_.delta_g2[k] = p.delta_g2.x0[k]; crs_.delta_g2[4 + k] = p.delta_g2.x1[k]; crs_.delta_g2[8 + k] = p.delta_g2.y0[k]; crs_.delta_g2[12 + k] = p.delta_g2.y1[k];
    }
    crs_.delta_g1_inf = p.delta_g1.is_infinity;
    crs_.delta_g2_inf = p.delta_g2.is_infinity;
  }
  ~Prover() {
    for (const uint64_t* p : {crs_.d_h, crs_.d_l, crs_.d_a, crs_.d_b_g1, crs_.d_b_g2})
      if (p) kg_bases_unregister(c_.raw(), p);
  }
  Proof create_proof(const std::vector<Fe>& a_eval, const std::vector<Fe>& b_eval, const std::vector<Fe>& c_eval, const std::vector<Fe>& x,
                     const std::vector<Fe>& w, const Fe& r, const Fe& s) const {
    DeviceBuffer da(c_, a_eval.data(), a_eval.size() * 32), db(c_, b_eval.data(), b_eval.size() * 32), dc(c_, c_eval.data(), c_eval.size() * 32);
    DeviceBuffer dx(c_, x.data(), x.size() * 32), dw(c_, w.data(), w.size() * 32);
    uint64_t out[32];
    uint8_t inf[3];
    c_.check(kg_groth16_prove_bn254(c_.raw(), &crs_, da.as<uint64_t>(), db.as<uint64_t>(), dc.as<uint64_t>(), dx.as<uint64_t>(), dw.as<uint64_t>(),
                                    r.data(), s.data(), out, inf), "kg_groth16_prove_bn254");
    return {detail::g1_from(out, inf[0] != 0), detail::g2_from(out + 8, inf[1] != 0), detail::g1_from(out + 24, inf[2] != 0)};
  }

 private:
  template <class Pt>
  void upload(const std::vector<Pt>& pts, int curve, const uint64_t*& d_xy, const uint8_t*& d_inf) {
    std::vector<uint64_t> xy;
    std::vector<uint8_t> inf;
    detail::marshal(pts, pts.size(), xy, inf);
    bool any = false;
    for (uint8_t f : inf) any = any || f;
    keep_.emplace_back(new DeviceBuffer(c_, xy.data(), xy.size() * 8));
    d_xy = keep_.back()->template as<uint64_t>();
    d_inf = nullptr;
    if (any) {
      keep_.emplace_back(new DeviceBuffer(c_, inf.data(), inf.size()));
      d_inf = keep_.back()->template as<uint8_t>();
    }
    if (!pts.empty()) c_.check(kg_bases_register(c_.raw(), curve, d_xy, d_inf, pts.size()), "kg_bases_register");
  }
  const Context& c_;
  kg_groth16_crs crs_;
  std::vector<std::unique_ptr<DeviceBuffer>> keep_;
};

// zkstd::matrix::SparseMatrix as compressed sparse rows over z = (u | x | w): Wire::Instance(i) -> column i, Wire::Witness(i) ->
// column i + l (the index rule of SparseMatrix::prod, matrix.rs:36-48), resolved by whoever builds the CSR
struct SparseMatrix {
  std::vector<uint64_t> row_ptr, col;
  std::vector<Fe> val;
  size_t rows() const { return row_ptr.empty() ? 0 : row_ptr.size() - 1; }
};

inline std::pair<Parameters, VerifyingKey> setup(const Context& c, const SparseMatrix& a, const SparseMatrix& b, const SparseMatrix& cm, size_t l, size_t m_l_1,
                                                 const std::array<Fe, 5>& toxic) {
  const size_t m = a.rows(), nv = l + m_l_1;
  if (m == 0 || b.rows() != m || cm.rows() != m) throw std::invalid_argument("setup: three matrices with one row per constraint");
  const SparseMatrix* mats[3] = {&a, &b, &cm};
  std::unique_ptr<DeviceBuffer> buf[9];
  kg_csr csr[3];
  const uint64_t pad64 = 0;
  const Fe padfe{0, 0, 0, 0};
  for (int k = 0; k < 3; ++k) {
    const SparseMatrix& M = *mats[k];
    if (M.col.size() != M.val.size() || M.row_ptr.size() != m + 1 || M.row_ptr.back() != M.col.size()) throw std::invalid_argument("setup: malformed CSR");
    for (uint64_t cidx : M.col) if (cidx >= nv) throw std::out_of_range("setup: a column index past z = x || w");
    buf[3 * k].reset(new DeviceBuffer(c, M.row_ptr.data(), M.row_ptr.size() * 8));
    buf[3 * k + 1].reset(new DeviceBuffer(c, M.col.empty() ? &pad64 : M.col.data(), (M.col.empty() ? 1 : M.col.size()) * 8));
    buf[3 * k + 2].reset(new DeviceBuffer(c, M.val.empty() ? &padfe : M.val.data(), (M.val.empty() ? 1 : M.val.size()) * 32));
    csr[k] = kg_csr{buf[3 * k]->as<uint64_t>(), buf[3 * k + 1]->as<uint64_t>(), buf[3 * k + 2]->as<uint64_t>()};
  }
  auto pts = [&](size_t n, size_t words) { return std::unique_ptr<DeviceBuffer>(new DeviceBuffer(c, (n ? n : 1) * words * 8)); };
  auto flg = [&](size_t n) { return std::unique_ptr<DeviceBuffer>(new DeviceBuffer(c, n ? n : 1)); };
  auto h = pts(m - 1, 8), lq = pts(m_l_1, 8), qa = pts(nv, 8), qb1 = pts(nv, 8), qb2 = pts(nv, 16), ic = pts(l, 8);
  auto hi = flg(m - 1), li = flg(m_l_1), ai = flg(nv), b1i = flg(nv), b2i = flg(nv), ici = flg(l);
  kg_groth16_crs crs{};
  crs.d_h = h->as<uint64_t>(); crs.d_h_inf = hi->as<uint8_t>(); crs.d_l = lq->as<uint64_t>(); crs.d_l_inf = li->as<uint8_t>();
  crs.d_a = qa->as<uint64_t>(); crs.d_a_inf = ai->as<uint8_t>(); crs.d_b_g1 = qb1->as<uint64_t>(); crs.d_b_g1_inf = b1i->as<uint8_t>();
  crs.d_b_g2 = qb2->as<uint64_t>(); crs.d_b_g2_inf = b2i->as<uint8_t>();
  uint64_t gamma_g2[16];
  uint8_t vinf[6];
  const int rc = kg_groth16_setup_bn254(c.raw(), &csr[0], &csr[1], &csr[2], m, l, m_l_1, reinterpret_cast<const uint64_t*>(toxic.data()), &crs,
                                        ic->as<uint64_t>(), ici->as<uint8_t>(), gamma_g2, vinf);
  if (rc == KG_ERR_INVERSION) throw ProverInversionFailed();
  c.check(rc, "kg_groth16_setup_bn254");
  auto g1s = [&](const DeviceBuffer& xy, const DeviceBuffer& inf, size_t n) {
    std::vector<uint64_t> w((n ? n : 1) * 8);
    std::vector<uint8_t> f(n ? n : 1);
    xy.download(w.data()); inf.download(f.data());
    std::vector<G1Affine> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = detail::g1_from(&w[8 * i], f[i] != 0);
    return v;
  };
  Parameters P;
  P.h = g1s(*h, *hi, m - 1); P.l = g1s(*lq, *li, m_l_1); P.a = g1s(*qa, *ai, nv); P.b_g1 = g1s(*qb1, *b1i, nv);
  {
    std::vector<uint64_t> w((nv ? nv : 1) * 16);
    std::vector<uint8_t> f(nv ? nv : 1);
    qb2->download(w.data()); b2i->download(f.data());
    P.b_g2.resize(nv);
    for (size_t i = 0; i < nv; ++i) P.b_g2[i] = detail::g2_from(&w[16 * i], f[i] != 0);
  }
  P.alpha_g1 = detail::g1_from(crs.alpha_g1, vinf[0] != 0); P.beta_g1 = detail::g1_from(crs.beta_g1, vinf[1] != 0);
  P.delta_g1 = detail::g1_from(crs.delta_g1, vinf[2] != 0);
  P.beta_g2 = detail::g2_from(crs.beta_g2, vinf[3] != 0); P.delta_g2 = detail::g2_from(crs.delta_g2, vinf[5] != 0);
  VerifyingKey vk{P.alpha_g1, P.beta_g1, P.delta_g1, P.beta_g2, detail::g2_from(gamma_g2, vinf[4] != 0), P.delta_g2, g1s(*ic, *ici, l)};
  return {std::move(P), std::move(vk)};
}

// nova::R1csShape's three matrices, resident on the device: prod (SparseMatrix::prod) and Prover::compute_cross_term
// (nova/src/prover.rs:53-90: T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1, one fused kernel).  field: KG_FR for the bn254 driver,
// KG_FQ for the Grumpkin driver (nova/src/driver.rs:9-42).
class R1csShape {
 public:
  R1csShape(const Context& c, const SparseMatrix& a, const SparseMatrix& b, const SparseMatrix& cm, int field = KG_FR) : c_(c), field_(field), m_(a.rows()) {
    if (b.rows() != m_ || cm.rows() != m_) throw std::invalid_argument("R1csShape: the three matrices have one row per constraint");
    const SparseMatrix* mats[3] = {&a, &b, &cm};
    for (int k = 0; k < 3; ++k) {
      const uint64_t pad64 = 0;
      const Fe padfe{0, 0, 0, 0};
      const SparseMatrix& M = *mats[k];
      if (M.col.size() != M.val.size() || M.row_ptr.size() != m_ + 1 || M.row_ptr.back() != M.col.size()) throw std::invalid_argument("R1csShape: malformed CSR");
      for (uint64_t cidx : M.col) max_col_ = cidx > max_col_ ? cidx : max_col_;
      buf_[3 * k].reset(new DeviceBuffer(c, M.row_ptr.data(), M.row_ptr.size() * 8));
      buf_[3 * k + 1].reset(new DeviceBuffer(c, M.col.empty() ? &pad64 : M.col.data(), (M.col.empty() ? 1 : M.col.size()) * 8));
      buf_[3 * k + 2].reset(new DeviceBuffer(c, M.val.empty() ? &padfe : M.val.data(), (M.val.empty() ? 1 : M.val.size()) * 32));
      csr_[k] = kg_csr{buf_[3 * k]->as<uint64_t>(), buf_[3 * k + 1]->as<uint64_t>(), buf_[3 * k + 2]->as<uint64_t>()};
    }
  }
  size_t m() const { return m_; }
  // SparseMatrix::prod of matrix `which` (0 = A, 1 = B, 2 = C) with z
  std::vector<Fe> prod(int which, const std::vector<Fe>& z) const {
    covers(z);
    DeviceBuffer dz(c_, z.data(), z.size() * 32), out(c_, m_ * 32);
    c_.check(kg_r1cs_prod(c_.raw(), field_, csr_[which].d_row_ptr, csr_[which].d_col, csr_[which].d_val, m_, dz.as<uint64_t>(), out.as<uint64_t>()), "kg_r1cs_prod");
    std::vector<Fe> r(m_);
    out.download(r.data());
    return r;
  }
  std::vector<Fe> compute_cross_term(const std::vector<Fe>& z1, const std::vector<Fe>& z2, const Fe& u1, const Fe& u2) const {
    covers(z1);
    covers(z2);
    DeviceBuffer d1(c_, z1.data(), z1.size() * 32), d2(c_, z2.data(), z2.size() * 32), out(c_, m_ * 32);
    c_.check(kg_nova_cross_term(c_.raw(), field_, &csr_[0], &csr_[1], &csr_[2], m_, d1.as<uint64_t>(), d2.as<uint64_t>(), u1.data(), u2.data(), out.as<uint64_t>()),
             "kg_nova_cross_term");
    std::vector<Fe> t(m_);
    out.download(t.data());
    return t;
  }

 private:
  void covers(const std::vector<Fe>& z) const {          // the reference would panic on an index past z; the kernels do not check
    if (m_ && z.size() <= max_col_) throw std::out_of_range("R1csShape: z is shorter than the largest column index");
  }
  const Context& c_;
  int field_;
  size_t m_;
  uint64_t max_col_ = 0;
  std::unique_ptr<DeviceBuffer> buf_[9];
  kg_csr csr_[3];
};

// nova/src/pedersen.rs:6-20 with the key g cut over several contexts (one per GPU of the node): slice i is uploaded to and
// registered on context i once (kg_sharded_key_create); commit uploads each device's slice of m, every device runs the whole
// pipeline on its slice and the affine partial sums are added on the host (index-range sharding).
class ShardedPedersenCommitment {
 public:
  ShardedPedersenCommitment(const std::vector<const Context*>& ctxs, const std::vector<G1Affine>& g, int curve = KG_G1) : first_(*ctxs.at(0)) {
    std::vector<uint64_t> xy;
    std::vector<uint8_t> inf;
    detail::marshal(g, g.size(), xy, inf);
    bool any = false;
    for (uint8_t f : inf) any = any || f;
    std::vector<kg_ctx*> raw;
    for (const Context* c : ctxs) raw.push_back(c->raw());
    first_.check(kg_sharded_key_create(raw.data(), (int)raw.size(), curve, xy.data(), any ? inf.data() : nullptr, g.size(), &key_), "kg_sharded_key_create");
  }
  ~ShardedPedersenCommitment() { if (key_) kg_sharded_key_destroy(key_); }
  ShardedPedersenCommitment(const ShardedPedersenCommitment&) = delete;
  ShardedPedersenCommitment& operator=(const ShardedPedersenCommitment&) = delete;
  size_t len() const { return kg_sharded_key_len(key_); }
  G1Affine commit(const std::vector<Fe>& m) const {
    uint64_t xy[8];
    uint8_t inf = 0;
    first_.check(kg_sharded_key_commit(key_, reinterpret_cast<const uint64_t*>(m.data()), m.size(), xy, &inf), "kg_sharded_key_commit");
    return detail::g1_from(xy, inf != 0);
  }

 private:
  const Context& first_;
  kg_sharded_key* key_ = nullptr;
};

}  // namespace kogarashi
