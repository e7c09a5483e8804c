// vec.hip -- element-wise field kernels and the deterministic synthetic-input generators.
//
// kg_field_vec_op replaces, one GPU thread per element, the reference's limb functions
// zkstd/src/arithmetic/limbs/bits_256/normal.rs:4-31,34-53,56-80,83-121,124-166,170-184,256-270 and the
// point-wise polynomial ops groth16/src/poly.rs:168-195.  HBM-bound for add/sub/mul (96 B/element).
#include "common.h"

using namespace kg;

namespace {

template <class P>
__global__ void __launch_bounds__(256) k_vec_op(int op, const uint64_t* a, const uint64_t* b,
                                                uint64_t* out, size_t n) {
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
    case KG_OP_INVERT: to_ref(inv(from_ref<P>(wa)), wo); break;
    case KG_OP_FROM_MONT: ref_to_int<P>(wa, wo); break;
    case KG_OP_TO_MONT: int_to_ref<P>(wa, wo); break;
    default: return;
  }
  store_words(out, i, wo);
}

template <class P>
__global__ void __launch_bounds__(256) k_vec_scale(const uint64_t* a, Words8 s, uint64_t* out, size_t n) {
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

// CSR sparse matrix-vector product over Fr (zkstd/src/matrix/row.rs:43-51), one WAVE per row: lanes stride over the
// row's entries, partial sums meet through 6 shuffle steps.  R1CS rows are short, but the transposed system used by the
// setup has a column (the constant-one wire) touching every constraint, so a row may hold millions of entries.
__device__ __forceinline__ Fr shfl_xor_fr(const Fr& a, int mask) {
  Fr r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = __shfl_xor(a.l[k], mask);
  return r;
}
__global__ void __launch_bounds__(256) k_r1cs_evaluate(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col,
                                                       const uint64_t* __restrict__ val, size_t m, const uint64_t* __restrict__ z,
                                                       uint64_t* __restrict__ out) {
  const size_t row = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= m) return;
  Fr sum = Fr::zero();
  for (uint64_t e = row_ptr[row] + lane; e < row_ptr[row + 1]; e += 64) {
    uint32_t wv[8], wz[8];
    load_words(val, e, wv);
    load_words(z, col[e], wz);
    // raw(z) * internal(val) keeps the product in the ABI's Montgomery domain
    Fr t = mul(limbs_from_words<FrParams>(wz), from_ref<FrParams>(wv));
    sum = vred(norm(add(sum, t)));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum = vred(norm(add(sum, shfl_xor_fr(sum, d))));
  if (lane == 0) {
    uint32_t wo[8];
    words_from_limbs(reduce_2p(sum), wo);
    store_words(out, row, wo);
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

int kg_r1cs_evaluate(kg_ctx* c, const uint64_t* row_ptr, const uint64_t* col, const uint64_t* val, size_t m, const uint64_t* z, uint64_t* out) {
  if (!c) return KG_ERR_BAD_ARG;
  if (m == 0) return KG_OK;
  KG_HIP(c, hipSetDevice(c->device));
  if (!row_ptr || !col || !val || !z || !out) return KG_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_r1cs_evaluate, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, c->stream, row_ptr, col, val, m, z, out);
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
