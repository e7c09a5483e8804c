// host_cost.cpp -- what the reference's call sites cost through a COMPILED host above the C ABI (the C++ mirror, include/kogarashi_amd.hpp;
// the Rust glue does the same steps and cannot be compiled in this image): marshalling of repr(Rust)-like point structs, the first-sight
// costs (upload, registration), and the per-call costs with the scalars in pageable host memory.  Feeds the table in INTEGRATION.md.
//   g++ -O2 -std=c++17 -o host_cost tools/host/host_cost.cpp -Lkogarashi_amd -lkogarashi_amd -Wl,-rpath,$PWD/kogarashi_amd && ./host_cost
#include <chrono>
#include <cstdio>
#include <cstring>
#include "../../include/kogarashi_amd.hpp"

using namespace kogarashi;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
template <class Fn> static double per_call(int reps, Fn fn) {
  fn(); fn();
  const auto t0 = Clock::now();
  for (int i = 0; i < reps; ++i) fn();
  return ms_since(t0) / reps;
}
static const uint64_t SEED = 0x4B6F676172617368ull;

static std::vector<Fe> gen_scalars(const Context& c, int field, uint64_t seed, size_t n) {
  DeviceBuffer d(c, n * 32);
  c.check(kg_gen_scalars(c.raw(), field, seed, 0, n, d.as<uint64_t>()), "kg_gen_scalars");
  std::vector<Fe> v(n);
  d.download(v.data());
  return v;
}
static std::vector<G1Affine> gen_points(const Context& c, uint64_t seed, size_t n) {
  DeviceBuffer d(c, n * 64);
  c.check(kg_gen_bases(c.raw(), KG_G1, seed, 0, n, d.as<uint64_t>()), "kg_gen_bases");
  std::vector<uint64_t> w(8 * n);
  d.download(w.data());
  std::vector<G1Affine> p(n);
  for (size_t i = 0; i < n; ++i) p[i] = detail::g1_from(&w[8 * i], false);
  return p;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20;
  kg_init();
  Context ctx(0);
  const size_t n = (size_t)1 << lg;
  std::printf("host-side costs through include/kogarashi_amd.hpp (g++ -O2), one MI355X, pageable host memory\n");
  const std::vector<G1Affine> bases = gen_points(ctx, SEED + 1, n);
  const std::vector<Fe> k = gen_scalars(ctx, KG_FR, SEED + 2, n);
  {   // 1. marshalling: point structs -> x | y words + flag bytes (what `marshal` does in the Rust glue through get_x() / get_y())
    std::vector<uint64_t> xy; std::vector<uint8_t> inf;
    const double t = per_call(5, [&] { detail::marshal(bases, n, xy, inf); });
    std::printf("marshal 2^%d G1 points                      %8.3f ms\n", lg, t);
  }
  {   // 2. msm_curve_addition, both arrays from the host on every call (kg_msm_host; marshal included)
    const double t = per_call(5, [&] { msm_curve_addition(ctx, bases, k); });
    std::printf("msm_curve_addition 2^%d, host arrays        %8.3f ms per call (marshal + 96 B per pair over PCIe)\n", lg, t);
  }
  {   // 3. the same bases resident: first sight, then per call
    auto t0 = Clock::now();
    ResidentBases rb(ctx, bases);
    kg_ctx_sync(ctx.raw());
    std::printf("ResidentBases: marshal + upload + register   %8.3f ms once\n", ms_since(t0));
    const double t = per_call(20, [&] { rb.msm(k); });
    std::printf("ResidentBases::msm 2^%d                      %8.3f ms per call (32 B per pair over PCIe inside the call)\n", lg, t);
  }
  {   // 4. Pedersen commitment of 2^(lg+1) scalars against a resident key (the slice one rank of the 8-GPU configuration commits at lg = 20)
    const size_t n2 = 2 * n;
    const std::vector<G1Affine> g = gen_points(ctx, SEED + 3, n2);
    const std::vector<Fe> m = gen_scalars(ctx, KG_FR, SEED + 4, n2);
    auto t0 = Clock::now();
    PedersenCommitment pc(ctx, g);
    kg_ctx_sync(ctx.raw());
    std::printf("PedersenCommitment::new 2^%d                 %8.3f ms once\n", lg + 1, ms_since(t0));
    const double t = per_call(10, [&] { pc.commit(m); });
    std::printf("PedersenCommitment::commit 2^%d              %8.3f ms per call\n", lg + 1, t);
  }
  {   // 5. one transform with the vector crossing the bus both ways
    const uint32_t kk = (uint32_t)lg + 2;
    const std::vector<Fe> v = gen_scalars(ctx, KG_FR, SEED + 5, (size_t)1 << kk);
    Fft f(ctx, kk);
    const double t = per_call(5, [&] { f.dft(v); });
    std::printf("Fft::dft 2^%u, vector up and down            %8.3f ms per call (64 B per element over PCIe; a padded copy and a fresh result vector on the host)\n", kk, t);
    std::vector<Fe> u(v);
    const double t2 = per_call(5, [&] { f.transform_in_place(u, 0, 0); });
    std::printf("Fft::transform_in_place 2^%u                 %8.3f ms per call (the two bus trips around a ~0.4 ms transform)\n", kk, t2);
  }
  {   // 6. witness fold
    const std::vector<Fe> a = gen_scalars(ctx, KG_FR, SEED + 6, n), b = gen_scalars(ctx, KG_FR, SEED + 7, n);
    const double t = per_call(5, [&] { fold(ctx, a, k[0], b); });
    std::printf("fold (W1 + r W2) 2^%d                        %8.3f ms per call (96 B per element over PCIe)\n", lg, t);
  }
  {   // 7. ZkSnark::setup and one proof on the chain circuit t_{i+1} = t_i (t_i + 1) with m = 2^(lg - 4) constraints
    const size_t m = n >> 4, l = 2, m_l_1 = m;
    Fe one;
    {
      DeviceBuffer d(ctx, 32);
      const uint64_t zero[4] = {0, 0, 0, 0};
      // one in Montgomery form = 0 + 1 (canonical) -> TO_MONT of the integer 1
      const uint64_t i1[4] = {1, 0, 0, 0};
      DeviceBuffer di(ctx, i1, 32);
      ctx.check(kg_field_vec_op(ctx.raw(), KG_FR, KG_OP_TO_MONT, di.as<uint64_t>(), nullptr, d.as<uint64_t>(), 1), "to_mont");
      d.download(one.data());
      (void)zero;
    }
    auto wire = [](size_t i) -> uint64_t { return i == 0 ? 1 : 2 + i - 1; };
    SparseMatrix A, B, C;
    A.row_ptr.resize(m + 1); B.row_ptr.resize(m + 1); C.row_ptr.resize(m + 1);
    A.col.resize(m); B.col.resize(2 * m); C.col.resize(m);
    A.val.assign(m, one); B.val.assign(2 * m, one); C.val.assign(m, one);
    for (size_t i = 0; i <= m; ++i) { A.row_ptr[i] = i; B.row_ptr[i] = 2 * i; C.row_ptr[i] = i; }
    for (size_t i = 0; i < m; ++i) { A.col[i] = wire(i); B.col[2 * i] = wire(i); B.col[2 * i + 1] = 0; C.col[i] = wire(i + 1); }
    const std::vector<Fe> tox = gen_scalars(ctx, KG_FR, SEED + 8, 5);
    const std::array<Fe, 5> toxic{tox[0], tox[1], tox[2], tox[3], tox[4]};
    setup(ctx, A, B, C, l, m_l_1, toxic);                       // first call: builds the context's generator tables
    auto t0 = Clock::now();
    auto made = setup(ctx, A, B, C, l, m_l_1, toxic);
    std::printf("ZkSnark::setup, m = 2^%d                     %8.3f ms (matrices up, CRS on the device, Parameters down and unmarshalled)\n", lg - 4, ms_since(t0));
    // a satisfying witness is not needed for timing: any z gives a proof of the same cost
    const std::vector<Fe> x = gen_scalars(ctx, KG_FR, SEED + 9, l), w = gen_scalars(ctx, KG_FR, SEED + 10, m_l_1);
    std::vector<Fe> z(x); z.insert(z.end(), w.begin(), w.end());
    R1csShape shape(ctx, A, B, C);
    const std::vector<Fe> ae = shape.prod(0, z), be = shape.prod(1, z), ce = shape.prod(2, z);
    t0 = Clock::now();
    Prover prover(ctx, made.first, m, l, m_l_1);
    kg_ctx_sync(ctx.raw());
    std::printf("Prover::new (CRS marshal + upload + register) %7.3f ms once\n", ms_since(t0));
    const double t = per_call(10, [&] { prover.create_proof(ae, be, ce, x, w, tox[0], tox[1]); });
    std::printf("Prover::create_proof, m = 2^%d               %8.3f ms per call (five host vectors up, proof down)\n", lg - 4, t);
  }
  return 0;
}
