// setup.hip -- ZkSnark::setup behind the C ABI: kg_groth16_setup_bn254.
//
// Replaces groth16/src/zksnark.rs:17-127 (the CRS of a circuit from the five toxic scalars) and its helpers eval (:131-187) and
// eval_at_tau (:190-194) after circuit synthesis.  Everything that scales with the circuit runs on the device, composed from
// the library's own primitives:
//
//   powers of tau            kg_field_powers                      zksnark.rs:44-49
//   h scalars                kg_field_vec_scale by (tau^n-1)/delta   :51-58
//   Lagrange coefficients    kg_ntt_bn254_fr (idft)               :61
//   u_i(tau), v_i(tau), w_i(tau)   the constraint matrices TRANSPOSED on the device (counting sort by column: the reference's
//                            SparseMatrix::x_and_w, zkstd/src/matrix.rs:17-29) and multiplied with the coefficient vector
//                            (the CSR product of vec.hip)         :190-194
//   (beta u + alpha v + w) / gamma | delta   kg_field_vec_axpy x 2, kg_field_vec_scale x 2   :180-187
//   g * scalar               kg_fixed_base_mul (windowed generator tables)   :57, :168-187, :104-112
//
// Field additions are exact, so the order in which a column's entries are summed does not matter: the transposition scatters
// with atomics and the results are bit-identical to the reference's in-order fold.  The host touches five scalars (two
// inversions, tau^n by k squarings).
#include "common.h"
#include "host_fp.h"
#include <cstring>
#include <vector>

using namespace kg;

namespace {

// counter[key] += 1 for every active lane, ONE atomic per distinct key of the wave; returns the lane's own slot (the counter's value
// before the wave's addition + the lane's rank among its peers).  Constraint systems have wires that nearly every row uses (the constant
// one: every second entry of B in the chain circuit) -- 2^18 atomics on one address took 3 ms per matrix, the aggregated form 0.1.
__device__ __forceinline__ uint32_t wave_add_by_key(uint32_t* counters, uint32_t key, bool active) {
  const int lane = (int)(threadIdx.x & 63);
  unsigned long long todo = __ballot(active);
  uint32_t pos = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k = (uint32_t)__shfl((int)key, leader);
    const bool mine = active && key == k;
    const unsigned long long peers = __ballot(mine);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&counters[k], (uint32_t)__popcll(peers));
    base = (uint32_t)__shfl((int)base, leader);
    if (mine) pos = base + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    todo &= ~peers;
  }
  return pos;
}

// entries per column (+1: the scan below turns them into row pointers of the transpose)
__global__ void __launch_bounds__(256) k_col_count(const uint64_t* __restrict__ col, size_t nnz, uint32_t* __restrict__ cnt) {
  KG_SERVICE_PRIO();
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = e < nnz;
  wave_add_by_key(cnt, active ? (uint32_t)col[e] + 1u : 0u, active);
}

// exclusive scan of cnt[0 .. len) by ONE workgroup (len = variables + 1: a few hundred thousand; setup is not a hot path):
// t_ptr (u64 row pointers of the transpose) and cursor (u32 write positions) both receive it
__global__ void __launch_bounds__(1024) k_scan_cols(const uint32_t* __restrict__ cnt, size_t len, uint64_t* __restrict__ t_ptr, uint32_t* __restrict__ cursor) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t base = 0; base < len; base += 1024) {
    const size_t i = base + threadIdx.x;
    const uint32_t v = i < len ? cnt[i] : 0u;
    uint32_t incl = v;                                    // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; ++w) off += wave_sum[w];
    // cnt[0] = 0 and cnt[j + 1] = entries of column j: the INCLUSIVE scan at i is the row pointer of column i
    if (i < len) { t_ptr[i] = off + incl; cursor[i] = off + incl; }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + incl;
    __syncthreads();
  }
}

// entry e of the matrix (row r, column j, value v) -> entry of row j of the transpose: (column r, value v)
__global__ void __launch_bounds__(256) k_transpose_fill(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col, const uint64_t* __restrict__ val,
                                                        size_t m, size_t nnz, uint32_t* __restrict__ cursor, uint64_t* __restrict__ t_col, uint64_t* __restrict__ t_val) {
  KG_SERVICE_PRIO();
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = e < nnz;                            // (no early return: the wave's lanes meet in wave_add_by_key)
  size_t lo = 0, hi = m;                                  // the row of entry e: largest r with row_ptr[r] <= e
  while (active && hi - lo > 1) {
    const size_t mid = (lo + hi) >> 1;
    if (row_ptr[mid] <= e) lo = mid; else hi = mid;
  }
  const uint32_t pos = wave_add_by_key(cursor, active ? (uint32_t)col[e] : 0u, active);
  if (!active) return;
  t_col[pos] = lo;
  const uint4* src = reinterpret_cast<const uint4*>(val) + 2 * e;
  uint4* dst = reinterpret_cast<uint4*>(t_val) + 2 * (size_t)pos;
  dst[0] = src[0]; dst[1] = src[1];
}

// The matrices come from the host as they are: a column index beyond the variables or a row pointer that runs backwards would send the
// transposition's counters and cursors out of their arrays.  One pass over each matrix raises a flag; the call then fails with
// KG_ERR_BAD_ARG instead of writing out of bounds (the hosts' mirrors validate on their side; a C caller need not have).
__global__ void __launch_bounds__(256) k_csr_check(const uint64_t* __restrict__ row_ptr, const uint64_t* __restrict__ col, size_t m, size_t nnz, size_t nv,
                                                   uint32_t* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i == 0 && row_ptr[0] != 0) bad = true;
  if (i < m && row_ptr[i] > row_ptr[i + 1]) bad = true;
  if (i < nnz && col[i] >= nv) bad = true;
  if (bad) atomicOr(flag, 1u);
}

struct Carve {
  size_t off = 0;
  size_t take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace

extern "C" {

int kg_groth16_setup_bn254(kg_ctx* ctx, const kg_csr* a, const kg_csr* b, const kg_csr* c, size_t m, size_t l, size_t m_l_1,
                           const uint64_t* h_toxic, kg_groth16_crs* crs, uint64_t* d_ic, uint8_t* d_ic_inf, uint64_t* out_gamma_g2,
                           uint8_t* out_vk_inf) {
  return kg::kg_guarded(ctx, [&]() -> int {
  if (!ctx || !a || !b || !c || !h_toxic || !crs || !out_gamma_g2 || !out_vk_inf) return KG_ERR_BAD_ARG;
  const size_t nv = l + m_l_1;
  if (m < 1 || m >= ((size_t)1 << 28)) return set_err(ctx, KG_ERR_BAD_ARG, "kg_groth16_setup_bn254: 1 <= m < 2^28 constraints (Fr has two-adicity 28)");
  if (nv >= ((size_t)1 << 32)) return set_err(ctx, KG_ERR_BAD_ARG, "kg_groth16_setup_bn254: more than 2^32 variables");
  const kg_csr* mats[3] = {a, b, c};
  for (const kg_csr* x : mats)
    if (!x->d_row_ptr || !x->d_col || !x->d_val) return KG_ERR_BAD_ARG;
  if ((m > 1 && (!crs->d_h || !crs->d_h_inf)) || (m_l_1 && (!crs->d_l || !crs->d_l_inf)) || (l && (!d_ic || !d_ic_inf)) ||
      (nv && (!crs->d_a || !crs->d_a_inf || !crs->d_b_g1 || !crs->d_b_g1_inf || !crs->d_b_g2 || !crs->d_b_g2_inf)))
    return set_err(ctx, KG_ERR_BAD_ARG, "kg_groth16_setup_bn254: an output array is missing");
  KG_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t k = 1;                                         // Fft::new(k) asserts k >= 1 (fft.rs:28): a one-constraint circuit takes n = 2
  while (((size_t)1 << k) < m) ++k;
  const size_t n = (size_t)1 << k;

  // ---- the five scalars on the host (zksnark.rs:28-38, :51-53) ---------------------------------------------------------------
  const HostFr alpha = HostFr::from_words(h_toxic), beta = HostFr::from_words(h_toxic + 4), gamma = HostFr::from_words(h_toxic + 8),
               delta = HostFr::from_words(h_toxic + 12), tau = HostFr::from_words(h_toxic + 16);
  if (is_zero(gamma) || is_zero(delta)) return set_err(ctx, KG_ERR_INVERSION, "kg_groth16_setup_bn254: gamma or delta is zero (Error::ProverInversionFailed)");
  const HostFr gamma_inv = inv(gamma), delta_inv = inv(delta), one = HostFr::one();
  HostFr tn = tau;
  for (uint32_t i = 0; i < k; ++i) tn = sqr(tn);
  const HostFr coeff = mul(sub<2, 2>(tn, one), delta_inv);      // fft.z(&tau) * delta^-1 = (tau^n - 1) / delta
  uint64_t w_one[4], w_tau[4], w_coeff[4], w_alpha[4], w_beta[4], w_ginv[4], w_dinv[4];
  one.to_words(w_one); tau.to_words(w_tau); coeff.to_words(w_coeff); alpha.to_words(w_alpha); beta.to_words(w_beta);
  gamma_inv.to_words(w_ginv); delta_inv.to_words(w_dinv);

  // ---- sizes of the three matrices (one 8-byte read-back each: setup is not a hot path) ---------------------------------------
  size_t nnz[3], nnz_max = 0;
  for (int j = 0; j < 3; ++j) {
    uint64_t v = 0;
    KG_HIP(ctx, hipMemcpyAsync(&v, mats[j]->d_row_ptr + m, 8, hipMemcpyDeviceToHost, ctx->stream));
    KG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (v >= ((uint64_t)1 << 32)) return set_err(ctx, KG_ERR_BAD_ARG, "kg_groth16_setup_bn254: more than 2^32 matrix entries");
    nnz[j] = (size_t)v;
    if (nnz[j] > nnz_max) nnz_max = nnz[j];
  }

  // ---- scratch: one allocation, released when the call returns ----------------------------------------------------------------
  Carve cv;
  const size_t o_pw = cv.take(n * 32), o_hs = cv.take((m > 1 ? m - 1 : 1) * 32), o_ev = cv.take(3 * (nv + 1) * 32), o_ext = cv.take((nv + 1) * 32),
               o_ics = cv.take((l + 1) * 32), o_ls = cv.take((m_l_1 + 1) * 32), o_cnt = cv.take((nv + 2) * 4), o_cur = cv.take((nv + 2) * 4),
               o_tptr = cv.take((nv + 2) * 8), o_tcol = cv.take((nnz_max + 1) * 8), o_tval = cv.take((nnz_max + 1) * 32),
               o_vks = cv.take(6 * 32), o_vk1 = cv.take(3 * 64), o_vk2 = cv.take(3 * 128), o_vki = cv.take(64);
  char* ws = nullptr;
  if (hipError_t e = dev_alloc(ctx, (void**)&ws, cv.off); e != hipSuccess) return set_err(ctx, KG_ERR_OOM, "setup scratch allocation", e);
  struct Free { kg_ctx* c; char* p; ~Free() { kg_ctx_sync(c); hipFree(p); } } guard{ctx, ws};
  KG_TRY(ensure_ws_vec(ctx, 3 * (nv + 16) * 4));
  hipStream_t st = ctx->stream;
  uint64_t* pw = (uint64_t*)(ws + o_pw);
  uint64_t* h_s = (uint64_t*)(ws + o_hs);
  uint64_t* ev[3] = {(uint64_t*)(ws + o_ev), (uint64_t*)(ws + o_ev) + (nv + 1) * 4, (uint64_t*)(ws + o_ev) + 2 * (nv + 1) * 4};
  uint64_t* ext = (uint64_t*)(ws + o_ext);
  uint64_t* ic_s = (uint64_t*)(ws + o_ics);
  uint64_t* l_s = (uint64_t*)(ws + o_ls);
  uint32_t* cnt = (uint32_t*)(ws + o_cnt);
  uint32_t* cursor = (uint32_t*)(ws + o_cur);
  uint64_t* t_ptr = (uint64_t*)(ws + o_tptr);
  uint64_t* t_col = (uint64_t*)(ws + o_tcol);
  uint64_t* t_val = (uint64_t*)(ws + o_tval);

  // ---- the matrices' contents: columns inside the variables, row pointers monotone from zero ---------------------------------------
  {
    uint32_t* flag = cnt;                                 // (the counters are cleared again below)
    KG_HIP(ctx, hipMemsetAsync(flag, 0, 4, st));
    for (int j = 0; j < 3; ++j) {
      const size_t span = nnz[j] > m ? nnz[j] : m;
      hipLaunchKernelGGL(k_csr_check, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, mats[j]->d_row_ptr, mats[j]->d_col, m, nnz[j], nv, flag);
    }
    KG_HIP(ctx, hipGetLastError());
    uint32_t h_flag = 0;
    KG_HIP(ctx, hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st));
    KG_HIP(ctx, hipStreamSynchronize(st));
    if (h_flag) return set_err(ctx, KG_ERR_BAD_ARG, "kg_groth16_setup_bn254: a matrix has a column index beyond l + m_l_1 or row pointers that do not ascend from zero");
  }

  // ---- powers of tau, h's scalars, Lagrange coefficients ------------------------------------------------------------------------
  KG_TRY(kg_field_powers(ctx, KG_FR, w_one, w_tau, pw, m));
  if (n > m) KG_HIP(ctx, hipMemsetAsync(pw + 4 * m, 0, (n - m) * 32, st));
  if (m > 1) KG_TRY(kg_field_vec_scale(ctx, KG_FR, pw, w_coeff, h_s, m - 1));
  KG_TRY(kg_ntt_bn254_fr(ctx, pw, k, 1, 0));

  // ---- u_i(tau), v_i(tau), w_i(tau): transposed products ------------------------------------------------------------------------
  for (int j = 0; j < 3 && nv; ++j) {
    KG_HIP(ctx, hipMemsetAsync(cnt, 0, (nv + 2) * 4, st));
    if (nnz[j]) hipLaunchKernelGGL(k_col_count, dim3((unsigned)((nnz[j] + 255) / 256)), dim3(256), 0, st, mats[j]->d_col, nnz[j], cnt);
    hipLaunchKernelGGL(k_scan_cols, dim3(1), dim3(1024), 0, st, cnt, nv + 1, t_ptr, cursor);
    if (nnz[j])
      hipLaunchKernelGGL(k_transpose_fill, dim3((unsigned)((nnz[j] + 255) / 256)), dim3(256), 0, st, mats[j]->d_row_ptr, mats[j]->d_col, mats[j]->d_val, m, nnz[j],
                         cursor, t_col, t_val);
    KG_HIP(ctx, hipGetLastError());
    KG_TRY(r1cs_prod_enqueue(ctx, st, KG_FR, t_ptr, t_col, t_val, nv, pw, ev[j], (uint32_t*)ctx->ws_vec));
  }
  if (nv) {
    // (beta * at + alpha * bt + ct) * inv, inv = 1/gamma for the l instance wires and 1/delta for the witness wires (zksnark.rs:180-187)
    KG_TRY(kg_field_vec_axpy(ctx, KG_FR, ev[2], w_beta, ev[0], ext, nv));
    KG_TRY(kg_field_vec_axpy(ctx, KG_FR, ext, w_alpha, ev[1], ext, nv));
    if (l) KG_TRY(kg_field_vec_scale(ctx, KG_FR, ext, w_ginv, ic_s, l));
    if (m_l_1) KG_TRY(kg_field_vec_scale(ctx, KG_FR, ext + 4 * l, w_dinv, l_s, m_l_1));
  }

  // ---- g * scalar (zksnark.rs:57, :168-187): an all-zero polynomial gives the identity, as the reference's untouched ADDITIVE_IDENTITY ----
  if (m > 1) KG_TRY(kg_fixed_base_mul(ctx, KG_G1, h_s, m - 1, const_cast<uint64_t*>(crs->d_h), const_cast<uint8_t*>(crs->d_h_inf)));
  if (m_l_1) KG_TRY(kg_fixed_base_mul(ctx, KG_G1, l_s, m_l_1, const_cast<uint64_t*>(crs->d_l), const_cast<uint8_t*>(crs->d_l_inf)));
  if (nv) {
    KG_TRY(kg_fixed_base_mul(ctx, KG_G1, ev[0], nv, const_cast<uint64_t*>(crs->d_a), const_cast<uint8_t*>(crs->d_a_inf)));
    KG_TRY(kg_fixed_base_mul(ctx, KG_G1, ev[1], nv, const_cast<uint64_t*>(crs->d_b_g1), const_cast<uint8_t*>(crs->d_b_g1_inf)));
    KG_TRY(kg_fixed_base_mul(ctx, KG_G2, ev[1], nv, const_cast<uint64_t*>(crs->d_b_g2), const_cast<uint8_t*>(crs->d_b_g2_inf)));
  }
  if (l) KG_TRY(kg_fixed_base_mul(ctx, KG_G1, ic_s, l, d_ic, d_ic_inf));

  // ---- the verifying key's six generator multiples (zksnark.rs:104-112) ---------------------------------------------------------
  uint64_t vks[6 * 4];                                    // G1: alpha, beta, delta; G2: beta, gamma, delta
  std::memcpy(vks, h_toxic, 32); std::memcpy(vks + 4, h_toxic + 4, 32); std::memcpy(vks + 8, h_toxic + 12, 32);
  std::memcpy(vks + 12, h_toxic + 4, 32); std::memcpy(vks + 16, h_toxic + 8, 32); std::memcpy(vks + 20, h_toxic + 12, 32);
  uint64_t* d_vks = (uint64_t*)(ws + o_vks);
  uint64_t* d_vk1 = (uint64_t*)(ws + o_vk1);
  uint64_t* d_vk2 = (uint64_t*)(ws + o_vk2);
  uint8_t* d_vki = (uint8_t*)(ws + o_vki);
  KG_HIP(ctx, hipMemcpyAsync(d_vks, vks, sizeof vks, hipMemcpyHostToDevice, st));
  KG_HIP(ctx, hipStreamSynchronize(st));                  // vks is a stack buffer
  KG_TRY(kg_fixed_base_mul(ctx, KG_G1, d_vks, 3, d_vk1, d_vki));
  KG_TRY(kg_fixed_base_mul(ctx, KG_G2, d_vks + 12, 3, d_vk2, d_vki + 3));
  uint64_t g1[3 * 8], g2[3 * 16];
  uint8_t vinf[6];
  KG_HIP(ctx, hipMemcpyAsync(g1, d_vk1, sizeof g1, hipMemcpyDeviceToHost, st));
  KG_HIP(ctx, hipMemcpyAsync(g2, d_vk2, sizeof g2, hipMemcpyDeviceToHost, st));
  KG_HIP(ctx, hipMemcpyAsync(vinf, d_vki, 6, hipMemcpyDeviceToHost, st));
  KG_TRY(kg_ctx_sync(ctx));                               // every output array is complete when the call returns
  crs->m = m; crs->l = l; crs->m_l_1 = m_l_1;
  std::memcpy(crs->alpha_g1, g1, 64); std::memcpy(crs->beta_g1, g1 + 8, 64); std::memcpy(crs->delta_g1, g1 + 16, 64);
  std::memcpy(crs->beta_g2, g2, 128); std::memcpy(out_gamma_g2, g2 + 16, 128); std::memcpy(crs->delta_g2, g2 + 32, 128);
  crs->delta_g1_inf = vinf[2]; crs->delta_g2_inf = vinf[5];
  out_vk_inf[0] = vinf[0]; out_vk_inf[1] = vinf[1]; out_vk_inf[2] = vinf[2]; out_vk_inf[3] = vinf[3]; out_vk_inf[4] = vinf[4]; out_vk_inf[5] = vinf[5];
  return KG_OK;
  });
}

}  // extern "C"
