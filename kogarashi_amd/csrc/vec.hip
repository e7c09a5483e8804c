// vec.hip -- element-wise field kernels and the deterministic synthetic-input generators.
//
// kg_field_vec_op replaces, one GPU thread per element, the reference's limb functions
// zkstd/src/arithmetic/limbs/bits_256/normal.rs:4-31,34-53,56-80,83-121,124-166,170-184,256-270 and the
// point-wise polynomial ops groth16/src/poly.rs:168-195.  HBM-bound for add/sub/mul (96 B/element).
#include "common.h"
#include "vecops.h"
#include "host_fp.h"

using namespace kg;

namespace {

template <class P>
__global__ void __launch_bounds__(256) k_vec_op(int op, const uint64_t* a, const uint64_t* b,
                                                uint64_t* out, size_t n) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[8], wb[8], wo[8];
  load_words(a, i, wa);
  const bool binary = (op == KG_OP_ADD || op == KG_OP_SUB || op == KG_OP_MUL);
  if (binary) load_words(b, i, wb);
  using F = Fp<P>;
  // The linear ops never leave the caller's Montgomery domain (a*R + b*R = (a+b)*R): re-pack, lazy add / fat subtract,
  // one value reduction, canonicalise.  A product needs one extra constant product: mont'(aR, bR) = ab*R^2/2^261, and
  // mont'(., 2^522/R) brings it back to ab*R.
  const F A = limbs_from_words<P>(wa);
  switch (op) {
    case KG_OP_ADD: words_from_limbs(reduce_2p(vred(norm(add(A, limbs_from_words<P>(wb))))), wo); break;
    case KG_OP_SUB: words_from_limbs(reduce_2p(vred(norm(sub<8, 1>(A, limbs_from_words<P>(wb))))), wo); break;
    case KG_OP_MUL: words_from_limbs(reduce_2p(mul(mul(A, limbs_from_words<P>(wb)), F::from_const(P::C_FROM_REF))), wo); break;
    case KG_OP_SQUARE: words_from_limbs(reduce_2p(mul(sqr(A), F::from_const(P::C_FROM_REF))), wo); break;
    case KG_OP_NEG: words_from_limbs(reduce_2p(vred(norm(sub<8, 1>(F::zero(), A)))), wo); break;
    case KG_OP_DOUBLE: words_from_limbs(reduce_2p(vred(norm(dbl(A)))), wo); break;
    case KG_OP_INVERT: to_ref(inv_fast(from_ref<P>(wa)), wo); break;
    case KG_OP_FROM_MONT: ref_to_int<P>(wa, wo); break;
    case KG_OP_TO_MONT: int_to_ref<P>(wa, wo); break;
    default: return;
  }
  store_words(out, i, wo);
}

template <class P>
__global__ void __launch_bounds__(256) k_vec_scale(const uint64_t* a, Words8 s, uint64_t* out, size_t n) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[8], wo[8];
  load_words(a, i, wa);
  // data stays in the caller's Montgomery domain: multiply by the constant in internal form
  Fp<P> sc = from_ref<P>(s.w);
  words_from_limbs(reduce_2p(mul(limbs_from_words<P>(wa), sc)), wo);
  store_words(out, i, wo);
}

// out[i] = start * base^i: each lane raises base to its own index (<= 64 squarings), no serial scan
template <class P>
__global__ void __launch_bounds__(256) k_powers(Words8 start, Words8 base, uint64_t* __restrict__ out, size_t n) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp<P> b = from_ref<P>(base.w), r = from_ref<P>(start.w);
  for (size_t e = i; e; e >>= 1) {
    if (e & 1) r = mul(r, b);
    b = sqr(b);
  }
  uint32_t wo[8];
  to_ref(r, wo);
  store_words(out, i, wo);
}

// out = a + s * b  (Nova fold, witness.rs:56-70): one product, lazy sum, one canonicalisation per element
template <class P>
__global__ void __launch_bounds__(256) k_vec_axpy(const uint64_t* a, Words8 s, const uint64_t* b,
                                                  uint64_t* out, size_t n) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[8], wb[8], wo[8];
  load_words(a, i, wa);
  load_words(b, i, wb);
  // raw(b) * internal(s) stays in the ABI's Montgomery domain, like raw(a)
  Fp<P> t = mul(limbs_from_words<P>(wb), from_ref<P>(s.w));
  words_from_limbs(reduce_2p(vred(norm(add(limbs_from_words<P>(wa), t)))), wo);
  store_words(out, i, wo);
}

// CSR sparse matrix-vector product (zkstd/src/matrix/row.rs:43-51, matrix.rs:36-48), G lanes per row: lanes stride over the
// row's entries, partial sums meet through log2(G) shuffle steps.  G = 1 (a lane per row) for constraint rows, G = 64 (a
// wave per row) for the long rows of kg_r1cs_prod's work list.
template <class P>
__device__ __forceinline__ Fp<P> shfl_xor_f(const Fp<P>& a, int mask) {
  Fp<P> r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = __shfl_xor(a.l[k], mask);
  return r;
}
// sum_e val[e] * z[col[e]] over the row, for one or two z vectors (every lane of the group holds the result).  Both factors are
// used as they lie in memory (x * 2^256), so a term is ONE Montgomery product -- z * val * 2^251 -- and the sums live in that
// "raw product" domain; the caller's next multiplication carries the constant that leaves it (C_FROM_REF for a plain
// matrix-vector product, the folded constants of cross_term_row).  Converting each coefficient first cost a second product per entry.
template <class P, int G, int NZ>
__device__ __forceinline__ void row_dot(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col, const uint64_t* __restrict__ val,
                                        size_t row, int lane, const uint64_t* __restrict__ z1, const uint64_t* __restrict__ z2, Fp<P> (&sum)[NZ]) {
#pragma unroll
  for (int q = 0; q < NZ; ++q) sum[q] = Fp<P>::zero();
  for (uint64_t e = row_ptr[row] + lane; e < row_ptr[row + 1]; e += G) {
    uint32_t wv[8], wz[8];
    load_words(val, e, wv);
    const Fp<P> v = limbs_from_words<P>(wv);
    const uint64_t c = col[e];
    load_words(z1, c, wz);
    sum[0] = dot_step(sum[0], limbs_from_words<P>(wz), v);
    if (NZ > 1) {
      load_words(z2, c, wz);
      sum[NZ - 1] = dot_step(sum[NZ - 1], limbs_from_words<P>(wz), v);
    }
  }
  if (G > 1) {
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1)
#pragma unroll
      for (int q = 0; q < NZ; ++q) sum[q] = dot_merge(sum[q], shfl_xor_f(sum[q], d));
  }
}

// Matrix-vector product in two launches.  Constraint rows hold a handful of terms, so a lane takes a whole row (a wave per
// row kept 63 of 64 lanes idle: 0.45 ms per 2^18-row product); a row of more than SHORT_ROW terms -- the transposed systems
// of the setup have one per heavily used wire, the constant-one wire touching every constraint -- is put on a work list
// instead and summed by a whole wave in the second launch.
constexpr uint32_t SHORT_ROW = 48;
constexpr uint32_t HUGE_ROW = 4096;      // beyond: a 1024-lane workgroup per row (the constant-one wire's row of a transposed system: an entry per constraint)
template <class P>
__global__ void __launch_bounds__(256) k_r1cs_rows_short(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col,
                                                         const uint64_t* __restrict__ val, size_t m, const uint64_t* __restrict__ z,
                                                         uint64_t* __restrict__ out, uint32_t* __restrict__ long_rows, uint32_t* __restrict__ long_count) {
  KG_SERVICE_PRIO();
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  const uint64_t len = row_ptr[row + 1] - row_ptr[row];
  if (len > SHORT_ROW) {      // work lists: long rows from the front of the list (a wave each), huge rows from its back (a workgroup each)
    if (len > HUGE_ROW) long_rows[m - 1 - atomicAdd(long_count + 1, 1u)] = (uint32_t)row;
    else long_rows[atomicAdd(long_count, 1u)] = (uint32_t)row;
    return;
  }
  Fp<P> sum[1];
  row_dot<P, 1, 1>(row_ptr, col, val, row, 0, z, z, sum);
  uint32_t wo[8];
  words_from_limbs(reduce_2p(mul(sum[0], Fp<P>::from_const(P::C_FROM_REF))), wo);        // raw product domain -> the ABI's
  store_words(out, row, wo);
}
template <class P>
__global__ void __launch_bounds__(256) k_r1cs_rows_long(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col,
                                                        const uint64_t* __restrict__ val, const uint64_t* __restrict__ z, uint64_t* __restrict__ out,
                                                        const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ long_count) {
  KG_SERVICE_PRIO();
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6), wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const uint32_t total = *long_count;
  for (uint32_t i = wave; i < total; i += nwaves) {      // wave-uniform trip count: the shuffles inside row_dot see full waves
    const size_t row = long_rows[i];
    Fp<P> sum[1];
    row_dot<P, 64, 1>(row_ptr, col, val, row, lane, z, z, sum);
    if (lane == 0) {
      uint32_t wo[8];
      words_from_limbs(reduce_2p(mul(sum[0], Fp<P>::from_const(P::C_FROM_REF))), wo);
      store_words(out, row, wo);
    }
  }
}

// A HUGE row (more than HUGE_ROW entries) per WORKGROUP: 1024 lanes stride over its entries; wave sums by shuffles, the sixteen wave sums
// through LDS.  Up to round 4 a wave took every long row: the transposed systems of the setup have a row per wire, and the constant-one wire's
// row holds an entry per constraint -- 2^18 entries on 64 lanes took 4.8 ms of a 29 ms setup.  Rows of 49 .. 4096 entries (a range check's bit
// sum: 254) keep a wave each: a circuit may hold tens of thousands of them, and 1024 waves take them in parallel.
constexpr int LONG_NT = 1024;
template <class P>
__global__ void __launch_bounds__(LONG_NT) k_r1cs_rows_huge(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col,
                                                            const uint64_t* __restrict__ val, const uint64_t* __restrict__ z, uint64_t* __restrict__ out,
                                                            const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ long_count, size_t m) {
  KG_SERVICE_PRIO();
  __shared__ uint32_t part[LONG_NT / 64][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t total = long_count[1];
  for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {      // workgroup-uniform trip count
    const size_t row = long_rows[m - 1 - i];
    Fp<P> sum = Fp<P>::zero();
    for (uint64_t e = row_ptr[row] + threadIdx.x; e < row_ptr[row + 1]; e += LONG_NT) {
      uint32_t wv[8], wz[8];
      load_words(val, e, wv);
      load_words(z, col[e], wz);
      sum = dot_step(sum, limbs_from_words<P>(wz), limbs_from_words<P>(wv));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum = dot_merge(sum, shfl_xor_f(sum, d));
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) part[wave][k] = sum.l[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      Fp<P> tot = sum;                                          // wave 0's own sum
      for (int w = 1; w < LONG_NT / 64; ++w) {
        Fp<P> o;
#pragma unroll
        for (int k = 0; k < 9; ++k) o.l[k] = part[w][k];
        tot = dot_merge(tot, o);
      }
      uint32_t wo[8];
      words_from_limbs(reduce_2p(mul(tot, Fp<P>::from_const(P::C_FROM_REF))), wo);
      store_words(out, row, wo);
    }
    __syncthreads();
  }
}

// Nova's cross term (nova/src/prover.rs:53-90): T = AZ1 o BZ2 + AZ2 o BZ1 - u1 * CZ2 - u2 * CZ1, one kernel: every matrix row
// is read once for both z vectors, the six matrix-vector products never touch memory.  ONE lane per constraint row: R1CS
// rows hold a handful of terms (1.3 per matrix in the chain circuit), so a lane group per row leaves most lanes idle in the
// entry loops and pays shuffle-tree merges for nothing -- 8 lanes per row ran at 230 GB/s of algorithmic traffic
// (1.17 ms at 2^20 rows, instruction-bound), this form is bound by its ~12 products per row.  A long row only delays its own wave.
// u1s, u2s: u * 2^266 as limbs (host-side), the factor form cross_term_row wants next to raw row sums.
struct CsrView { const uint64_t* row_ptr; const uint64_t* col; const uint64_t* val; };
struct Limbs9 { uint32_t l[9]; };
template <class P>
__global__ void __launch_bounds__(256) k_nova_cross_term(CsrView A, CsrView B, CsrView C, size_t m, const uint64_t* __restrict__ z1,
                                                         const uint64_t* __restrict__ z2, Limbs9 u1s, Limbs9 u2s, uint64_t* __restrict__ out,
                                                         uint32_t* __restrict__ long_rows, uint32_t* __restrict__ long_count) {
  KG_SERVICE_PRIO();
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  // a constraint with a long linear combination (a range check's bit sum: 254 terms) would hold its whole wave for ~0.5 us per term and
  // z vector: such rows go to a work list and get a wave each (k_nova_cross_term_long)
  if (A.row_ptr[row + 1] - A.row_ptr[row] > SHORT_ROW || B.row_ptr[row + 1] - B.row_ptr[row] > SHORT_ROW || C.row_ptr[row + 1] - C.row_ptr[row] > SHORT_ROW) {
    long_rows[atomicAdd(long_count, 1u)] = (uint32_t)row;
    return;
  }
  Fp<P> az[2], bz[2], cz[2];
  row_dot<P, 1, 2>(A.row_ptr, A.col, A.val, row, 0, z1, z2, az);
  row_dot<P, 1, 2>(B.row_ptr, B.col, B.val, row, 0, z1, z2, bz);
  row_dot<P, 1, 2>(C.row_ptr, C.col, C.val, row, 0, z1, z2, cz);
  const Fp<P> r = cross_term_row(az[0], az[1], bz[0], bz[1], cz[0], cz[1], Fp<P>::from_const(u1s.l), Fp<P>::from_const(u2s.l),
                                 Fp<P>::from_const(P::C_XT_HAD));
  uint32_t wo[8];
  words_from_limbs(reduce_2p(r), wo);
  store_words(out, row, wo);
}

template <class P>
__global__ void __launch_bounds__(256) k_nova_cross_term_long(CsrView A, CsrView B, CsrView C, const uint64_t* __restrict__ z1, const uint64_t* __restrict__ z2,
                                                              Limbs9 u1s, Limbs9 u2s, uint64_t* __restrict__ out, const uint32_t* __restrict__ long_rows,
                                                              const uint32_t* __restrict__ long_count) {
  KG_SERVICE_PRIO();
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6), wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const uint32_t total = *long_count;
  for (uint32_t i = wave; i < total; i += nwaves) {      // wave-uniform trip count: the shuffles inside row_dot see full waves
    const size_t row = long_rows[i];
    Fp<P> az[2], bz[2], cz[2];
    row_dot<P, 64, 2>(A.row_ptr, A.col, A.val, row, lane, z1, z2, az);
    row_dot<P, 64, 2>(B.row_ptr, B.col, B.val, row, lane, z1, z2, bz);
    row_dot<P, 64, 2>(C.row_ptr, C.col, C.val, row, lane, z1, z2, cz);
    if (lane == 0) {
      const Fp<P> r = cross_term_row(az[0], az[1], bz[0], bz[1], cz[0], cz[1], Fp<P>::from_const(u1s.l), Fp<P>::from_const(u2s.l),
                                     Fp<P>::from_const(P::C_XT_HAD));
      uint32_t wo[8];
      words_from_limbs(reduce_2p(r), wo);
      store_words(out, row, wo);
    }
  }
}

// ---- splitmix64 streams (oracle/pyoracle.py stream_at, oracle/kg_oracle.c stream_words) ---------------
__device__ __forceinline__ uint64_t splitmix_next(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ULL;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ void stream_words(uint64_t seed, uint64_t index, uint32_t lo[8], uint32_t hi[8]) {
  uint64_t s = seed + 8 * index * 0x9E3779B97F4A7C15ULL;
#pragma unroll
  for (int i = 0; i < 4; ++i) { uint64_t v = splitmix_next(s); lo[2 * i] = (uint32_t)v; lo[2 * i + 1] = (uint32_t)(v >> 32); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { uint64_t v = splitmix_next(s); hi[2 * i] = (uint32_t)v; hi[2 * i + 1] = (uint32_t)(v >> 32); }
}

// uniform scalar by the reference's wide reduction (represent.rs:18-28,80-103): (lo + hi*2^256) mod p
template <class P>
__global__ void __launch_bounds__(256) k_gen_scalars(uint64_t seed, size_t start, size_t n, uint64_t* __restrict__ out) {
  KG_SERVICE_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo[8], hi[8], wo[8];
  stream_words(seed, start + i, lo, hi);
  Fp<P> a = mul(limbs_from_words<P>(lo), Fp<P>::from_const(P::C_INT_TO_MONT));
  Fp<P> b = mul(limbs_from_words<P>(hi), Fp<P>::from_const(P::C_INT_HI_TO_MONT));
  to_ref(norm(add(a, b)), wo);
  store_words(out, i, wo);
}

template <class P>
__device__ __forceinline__ bool feq(const Fp<P>& a, const Fp<P>& b) { return is_zero(norm(sub<4, 1>(a, b))); }

// square root in Fq (q = 3 mod 4): a^((q+1)/4), accepted iff it squares back (bn254/src/fq.rs:121-127)
__device__ bool sqrt_field(const Fq& a, Fq& y) {
  y = pow_words(a, FqParams::E_SQRT);
  return feq(sqr(y), a);
}
// Tonelli-Shanks in Fr (bn254/src/fr.rs:165-208), returning the root with the smaller canonical integer
__device__ bool sqrt_field(const Fr& a, Fr& y) {
  using P = FrParams;
  if (is_zero(a)) { y = Fr::zero(); return true; }
  if (!feq(pow_words(a, P::E_P_MINUS_1_HALF), Fr::one())) return false;
  Fr c = Fr::from_const(P::ROOT_OF_UNITY);
  Fr tt = pow_words(a, P::E_TS_T);
  Fr r = pow_words(a, P::E_TS_T1H);
  int m = P::TS_S;
  while (!feq(tt, Fr::one())) {
    int i = 0;
    Fr x = tt;
    while (!feq(x, Fr::one())) { x = sqr(x); ++i; }
    Fr b = c;
    for (int j = 0; j < m - i - 1; ++j) b = sqr(b);
    m = i;
    c = sqr(b);
    tt = mul(tt, c);
    r = mul(r, b);
  }
  Fr nr = vred(norm(sub<4, 1>(Fr::zero(), r)));
  // compare canonical integers: mont(x*2^261, 1) = x
  Fr one_raw = Fr::zero();
  one_raw.l[0] = 1;
  Fr ri = reduce_2p(mul(r, one_raw));
  Fr ni = reduce_2p(mul(nr, one_raw));
  bool n_smaller = false;
  for (int k = 8; k >= 0; --k) {
    if (ri.l[k] != ni.l[k]) { n_smaller = ni.l[k] < ri.l[k]; break; }
  }
  y = n_smaller ? nr : r;
  return true;
}

template <class F> struct CurveB;
template <> struct CurveB<Fq> { static __device__ Fq b() { return Fq::from_const(FqParams::G1_B); } };
template <> struct CurveB<Fr> { static __device__ Fr b() { return Fr::from_const(FrParams::GRUMPKIN_B); } };

// valid curve point by try-and-increment on a seeded x (oracle/pyoracle.py base_at)
template <class F>
__global__ void __launch_bounds__(64) k_gen_bases(uint64_t seed, size_t start, size_t n, uint64_t* __restrict__ out) {
  using P = typename F::Params;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo[8], hi[8];
  stream_words(seed, start + i, lo, hi);
  F x = mul(limbs_from_words<P>(lo), F::from_const(P::C_INT_TO_MONT));
  F y;
  const F b = CurveB<F>::b();
  for (;;) {
    F rhs = norm(add(mul(sqr(x), x), b));
    if (sqrt_field(rhs, y) && !is_zero(y)) break;
    x = vred(norm(add(x, F::one())));
  }
  if (hi[0] & 1) y = vred(norm(sub<4, 1>(F::zero(), y)));
  uint32_t wx[8], wy[8];
  to_ref(x, wx);
  to_ref(y, wy);
  store_words(out, 2 * i, wx);
  store_words(out, 2 * i + 1, wy);
}

}  // namespace

extern "C" {

int kg_field_vec_op(kg_ctx* c, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || op < 0 || op > KG_OP_TO_MONT || (field != KG_FR && field != KG_FQ)) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  const bool binary = (op == KG_OP_ADD || op == KG_OP_SUB || op == KG_OP_MUL);
  if (!a || !out || (binary && !b)) return KG_ERR_BAD_ARG;
  dim3 grid((unsigned)((n + 255) / 256));
  if (field == KG_FR) hipLaunchKernelGGL(k_vec_op<FrParams>, grid, dim3(256), 0, c->stream, op, a, b, out, n);
  else hipLaunchKernelGGL(k_vec_op<FqParams>, grid, dim3(256), 0, c->stream, op, a, b, out, n);
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

int kg_field_vec_scale(kg_ctx* c, int field, const uint64_t* a, const uint64_t* h_s, uint64_t* out, size_t n) {
  if (!c || !h_s || (field != KG_FR && field != KG_FQ)) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  if (!a || !out) return KG_ERR_BAD_ARG;
  Words8 s;
  for (int i = 0; i < 4; ++i) { s.w[2 * i] = (uint32_t)h_s[i]; s.w[2 * i + 1] = (uint32_t)(h_s[i] >> 32); }
  dim3 grid((unsigned)((n + 255) / 256));
  if (field == KG_FR) hipLaunchKernelGGL(k_vec_scale<FrParams>, grid, dim3(256), 0, c->stream, a, s, out, n);
  else hipLaunchKernelGGL(k_vec_scale<FqParams>, grid, dim3(256), 0, c->stream, a, s, out, n);
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

int kg_field_powers(kg_ctx* c, int field, const uint64_t* h_start, const uint64_t* h_base, uint64_t* out, size_t n) {
  if (!c || !h_start || !h_base || (field != KG_FR && field != KG_FQ)) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  if (!out) return KG_ERR_BAD_ARG;
  Words8 s, b;
  for (int i = 0; i < 4; ++i) {
    s.w[2 * i] = (uint32_t)h_start[i]; s.w[2 * i + 1] = (uint32_t)(h_start[i] >> 32);
    b.w[2 * i] = (uint32_t)h_base[i]; b.w[2 * i + 1] = (uint32_t)(h_base[i] >> 32);
  }
  dim3 grid((unsigned)((n + 255) / 256));
  if (field == KG_FR) hipLaunchKernelGGL(k_powers<FrParams>, grid, dim3(256), 0, c->stream, s, b, out, n);
  else hipLaunchKernelGGL(k_powers<FqParams>, grid, dim3(256), 0, c->stream, s, b, out, n);
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

int kg_field_vec_axpy(kg_ctx* c, int field, const uint64_t* a, const uint64_t* h_s, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || !h_s || (field != KG_FR && field != KG_FQ)) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  if (!a || !b || !out) return KG_ERR_BAD_ARG;
  Words8 s;
  for (int i = 0; i < 4; ++i) { s.w[2 * i] = (uint32_t)h_s[i]; s.w[2 * i + 1] = (uint32_t)(h_s[i] >> 32); }
  dim3 grid((unsigned)((n + 255) / 256));
  if (field == KG_FR) hipLaunchKernelGGL(k_vec_axpy<FrParams>, grid, dim3(256), 0, c->stream, a, s, b, out, n);
  else hipLaunchKernelGGL(k_vec_axpy<FqParams>, grid, dim3(256), 0, c->stream, a, s, b, out, n);
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

}  // extern "C"

namespace kg {
// The product on a queue of the caller's choice; scratch: (m + 16) words for the work list of long rows.
int r1cs_prod_enqueue(kg_ctx* c, hipStream_t st, int field, const uint64_t* row_ptr, const uint64_t* col, const uint64_t* val, size_t m,
                      const uint64_t* z, uint64_t* out, uint32_t* scratch) {
  uint32_t* count = scratch;
  uint32_t* list = scratch + 16;
  KG_HIP(c, hipMemsetAsync(count, 0, 8, st));
  const dim3 grid((unsigned)((m + 255) / 256));
  if (field == KG_FR) {
    hipLaunchKernelGGL(k_r1cs_rows_short<FrParams>, grid, dim3(256), 0, st, row_ptr, col, val, m, z, out, list, count);
    hipLaunchKernelGGL(k_r1cs_rows_long<FrParams>, dim3(256), dim3(256), 0, st, row_ptr, col, val, z, out, list, count);
    hipLaunchKernelGGL(k_r1cs_rows_huge<FrParams>, dim3(64), dim3(LONG_NT), 0, st, row_ptr, col, val, z, out, list, count, m);
  } else {
    hipLaunchKernelGGL(k_r1cs_rows_short<FqParams>, grid, dim3(256), 0, st, row_ptr, col, val, m, z, out, list, count);
    hipLaunchKernelGGL(k_r1cs_rows_long<FqParams>, dim3(256), dim3(256), 0, st, row_ptr, col, val, z, out, list, count);
    hipLaunchKernelGGL(k_r1cs_rows_huge<FqParams>, dim3(64), dim3(LONG_NT), 0, st, row_ptr, col, val, z, out, list, count, m);
  }
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}
}  // namespace kg

extern "C" {

int kg_r1cs_prod(kg_ctx* c, int field, const uint64_t* row_ptr, const uint64_t* col, const uint64_t* val, size_t m, const uint64_t* z, uint64_t* out) {
  if (!c || (field != KG_FR && field != KG_FQ)) return KG_ERR_BAD_ARG;
  if (m == 0) return KG_OK;
  if (!row_ptr || !col || !val || !z || !out) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  if (m >= ((size_t)1 << 32)) return set_err(c, KG_ERR_BAD_ARG, "more than 2^32 rows");
  KG_TRY(ensure_ws_vec(c, 3 * (m + 16) * 4));             // work lists of long rows (three: the prover runs three products at once)
  return kg::r1cs_prod_enqueue(c, c->stream, field, row_ptr, col, val, m, z, out, (uint32_t*)c->ws_vec);
}

int kg_r1cs_evaluate(kg_ctx* c, const uint64_t* row_ptr, const uint64_t* col, const uint64_t* val, size_t m, const uint64_t* z, uint64_t* out) {
  return kg_r1cs_prod(c, KG_FR, row_ptr, col, val, m, z, out);
}

int kg_nova_cross_term(kg_ctx* c, int field, const kg_csr* a, const kg_csr* b, const kg_csr* cm, size_t m, const uint64_t* d_z1,
                       const uint64_t* d_z2, const uint64_t* h_u1, const uint64_t* h_u2, uint64_t* d_out) {
  if (!c || (field != KG_FR && field != KG_FQ) || !a || !b || !cm || !h_u1 || !h_u2) return KG_ERR_BAD_ARG;
  if (m == 0) return KG_OK;
  if (!d_z1 || !d_z2 || !d_out) return KG_ERR_BAD_ARG;
  for (const kg_csr* x : {a, b, cm})
    if (!x->d_row_ptr || !x->d_col || !x->d_val) return KG_ERR_BAD_ARG;
  KG_HIP(c, hipSetDevice(c->device));
  // u * 2^266 mod p as limbs: ten modular doublings of the ABI form u * 2^256
  auto scaled = [&](const uint64_t* h_u, bool fr) {
    Limbs9 out;
    uint64_t w64[4];
    if (fr) { HostFr x = HostFr::from_words(h_u); for (int i = 0; i < 10; ++i) x = dbl(x); for (int i = 0; i < 4; ++i) w64[i] = x.v[i]; }
    else { HostFq x = HostFq::from_words(h_u); for (int i = 0; i < 10; ++i) x = dbl(x); for (int i = 0; i < 4; ++i) w64[i] = x.v[i]; }
    uint32_t w[8];
    for (int i = 0; i < 4; ++i) { w[2 * i] = (uint32_t)w64[i]; w[2 * i + 1] = (uint32_t)(w64[i] >> 32); }
    const Fp<FrParams> lim = limbs_from_words<FrParams>(w);        // the spreading is the same for both fields
    for (int k = 0; k < 9; ++k) out.l[k] = lim.l[k];
    return out;
  };
  const Limbs9 u1 = scaled(h_u1, field == KG_FR), u2 = scaled(h_u2, field == KG_FR);
  const CsrView A{a->d_row_ptr, a->d_col, a->d_val}, B{b->d_row_ptr, b->d_col, b->d_val}, C{cm->d_row_ptr, cm->d_col, cm->d_val};
  if (m >= ((size_t)1 << 32)) return set_err(c, KG_ERR_BAD_ARG, "more than 2^32 rows");
  KG_TRY(ensure_ws_vec(c, 3 * (m + 16) * 4));             // the work list of long rows (same space as kg_r1cs_prod's)
  uint32_t* count = (uint32_t*)c->ws_vec;
  uint32_t* list = count + 16;
  KG_HIP(c, hipMemsetAsync(count, 0, 8, c->stream));
  const dim3 grid((unsigned)((m + 255) / 256));
  if (field == KG_FR) {
    hipLaunchKernelGGL(k_nova_cross_term<FrParams>, grid, dim3(256), 0, c->stream, A, B, C, m, d_z1, d_z2, u1, u2, d_out, list, count);
    hipLaunchKernelGGL(k_nova_cross_term_long<FrParams>, dim3(256), dim3(256), 0, c->stream, A, B, C, d_z1, d_z2, u1, u2, d_out, list, count);
  } else {
    hipLaunchKernelGGL(k_nova_cross_term<FqParams>, grid, dim3(256), 0, c->stream, A, B, C, m, d_z1, d_z2, u1, u2, d_out, list, count);
    hipLaunchKernelGGL(k_nova_cross_term_long<FqParams>, dim3(256), dim3(256), 0, c->stream, A, B, C, d_z1, d_z2, u1, u2, d_out, list, count);
  }
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

int kg_gen_scalars(kg_ctx* c, int field, uint64_t seed, size_t start, size_t n, uint64_t* out) {
  if (!c || (field != KG_FR && field != KG_FQ)) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  if (!out) return KG_ERR_BAD_ARG;
  dim3 grid((unsigned)((n + 255) / 256));
  if (field == KG_FR) hipLaunchKernelGGL(k_gen_scalars<FrParams>, grid, dim3(256), 0, c->stream, seed, start, n, out);
  else hipLaunchKernelGGL(k_gen_scalars<FqParams>, grid, dim3(256), 0, c->stream, seed, start, n, out);
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

int kg_gen_bases(kg_ctx* c, int curve, uint64_t seed, size_t start, size_t n, uint64_t* out) {
  if (!c || (curve != KG_G1 && curve != KG_GRUMPKIN)) return KG_ERR_BAD_ARG;
  if (n == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  if (!out) return KG_ERR_BAD_ARG;
  dim3 grid((unsigned)((n + 63) / 64));
  if (curve == KG_G1) hipLaunchKernelGGL(k_gen_bases<Fq>, grid, dim3(64), 0, c->stream, seed, start, n, out);
  else hipLaunchKernelGGL(k_gen_bases<Fr>, grid, dim3(64), 0, c->stream, seed, start, n, out);
  KG_HIP(c, hipGetLastError());
  return KG_OK;
}

}  // extern "C"
