// msm_reduce_kernels.h -- from partial sums to the c * W bit-plane sums the host finishes: gathers (fused with the first halving level
// where every bucket is one task), halving levels, the LDS tail, and the workgroup-wide trees of hot buckets.
#pragma once
#include "msm_acc_kernels.h"
#include "coop_add.h"

namespace kg {
namespace msm {
namespace {        // internal linkage: the kernels of a header exist once per translation unit that includes it

// ---------------------------------------------------------------------------------------------------
// bucket reduction by halving.  Arrays per window at level s: A (pair sums so far) and T_0..T_{s-1}
// (odd-index sums), each of length 2*n_out; the level emits A', the halved T's and a new T_s = odd items of A.
// After log2(B) levels every array has length 1: T_l = sum of buckets whose (index) bit l is set, A = sum of
// all buckets; sum_b (b+1)*B_b = A + sum_l 2^l T_l.
// Layout: point (window w, array a, item i) at index (w * narr + a) * len + i of a PointIO buffer.
// ---------------------------------------------------------------------------------------------------
// One point of a PointIO buffer (structure of arrays), coordinates read / written on demand (add_xyzz_stream).  Buffer
// addressing: the descriptor and the limb plane's offset are wave-uniform (SGPRs), the item's byte offset is ONE 32-bit VGPR
// per point -- flat loads cost a 64-bit address pair per limb plane (72 planes: the compiler kept ~70 VGPRs of addresses
// live).  The buffers stay far below the 4 GiB a descriptor spans (72 planes x 15 x 2^17 items x 4 B = 566 MB for G2, c = 18).
template <class F> struct SoaLimbs;
template <class P> struct SoaLimbs<Fp<P>> {
  static __device__ __forceinline__ Fp<P> load(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off) {     // stride4: bytes per plane
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, (plane + (uint32_t)k) * stride4, 0);
    return r;
  }
  static __device__ __forceinline__ void store(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off, const Fp<P>& a) {
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b32(a.l[k], rs, off, (plane + (uint32_t)k) * stride4, 0);
  }
};
template <class G> struct SoaLimbs<Fp2S<G>> {       // the pair's halves: c0 planes, then c1 planes (RawIO<Fp2S>); the half goes into the lane offset
  static __device__ __forceinline__ Fp2S<G> load(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off) {
    return {SoaLimbs<G>::load(rs, plane, stride4, off + 9u * (uint32_t)Fp2S<G>::half() * stride4)};
  }
  static __device__ __forceinline__ void store(BufRsrc rs, uint32_t plane, uint32_t stride4, uint32_t off, const Fp2S<G>& a) {
    SoaLimbs<G>::store(rs, plane, stride4, off + 9u * (uint32_t)Fp2S<G>::half() * stride4, a.v);
  }
};
template <class F> struct SoaSrc {
  BufRsrc rs; uint32_t stride4, off;                 // off = item * 4
  static constexpr uint32_t E = RawIO<F>::NW;
  __device__ __forceinline__ F x() const { return SoaLimbs<F>::load(rs, 0, stride4, off); }
  __device__ __forceinline__ F y() const { return SoaLimbs<F>::load(rs, E, stride4, off); }
  __device__ __forceinline__ F zz() const { return SoaLimbs<F>::load(rs, 2 * E, stride4, off); }
  __device__ __forceinline__ F zzz() const { return SoaLimbs<F>::load(rs, 3 * E, stride4, off); }
};
template <class F> struct SoaDst {
  BufRsrc rs; uint32_t stride4, off;
  static constexpr uint32_t E = RawIO<F>::NW;
  __device__ __forceinline__ void x(const F& v) { SoaLimbs<F>::store(rs, 0, stride4, off, v); }
  __device__ __forceinline__ void y(const F& v) { SoaLimbs<F>::store(rs, E, stride4, off, v); }
  __device__ __forceinline__ void zz(const F& v) { SoaLimbs<F>::store(rs, 2 * E, stride4, off, v); }
  __device__ __forceinline__ void zzz(const F& v) { SoaLimbs<F>::store(rs, 3 * E, stride4, off, v); }
};
template <class F> struct HalveWaves { static constexpr int MIN = 5; };
template <class G> struct HalveWaves<Fp2S<G>> { static constexpr int MIN = 4; };     // 110 VGPRs as it comes (was 170); nothing fits beside a G2 accumulation anyway
#ifndef KG_HALVE_ATTR
#define KG_HALVE_ATTR __attribute__((amdgpu_waves_per_eu(HalveWaves<F>::MIN)))
#endif
// Five waves per SIMD = 96 VGPRs: what four resident accumulation waves (4 x 104) leave free, so a halving level runs beside
// an accumulation instead of waiting for its tail.  With both operands loaded up front the kernel took 160.
template <class F>
__global__ void __launch_bounds__(64) KG_HALVE_ATTR k_halve(const uint32_t* in, size_t in_stride, uint32_t* out, size_t out_stride,
                                              int W, int narr_in, uint32_t n_out) {
  KG_REDUCE_PRIO();
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<F>::N;       // < 2^27 items (W * B <= 15 * 2^17)
  const uint32_t per_w = (uint32_t)narr_in * n_out;
  if (t >= per_w * (uint32_t)W) return;
  const uint32_t w = t / per_w;
  const uint32_t r = t % per_w;
  const uint32_t a = r / n_out;
  const uint32_t i = r % n_out;
  const uint32_t narr_out = (uint32_t)narr_in + 1;
  const uint32_t src = (w * narr_in + a) * (2 * n_out) + 2 * i;
  const BufRsrc rin = soa_rsrc(in), rout = soa_rsrc(out);
  const uint32_t is4 = (uint32_t)in_stride * 4u, os4 = (uint32_t)out_stride * 4u;
  const SoaSrc<F> p0{rin, is4, src * 4u}, p1{rin, is4, src * 4u + 4u};
  SoaDst<F> sum{rout, os4, ((w * narr_out + a) * n_out + i) * 4u};
  add_xyzz_stream<F>(p0, p1, sum);
  if (a == 0) {
    SoaDst<F> odd{rout, os4, ((w * narr_out + narr_in) * n_out + i) * 4u};
    copy_xyzz_stream<F>(p1, odd);
  }
}

// The dense bucket array and the first halving level in one launch (buckets that own a single partial sum: every bucket of a
// uniform input): lane i of window w reads the partial sums of buckets 2i and 2i + 1 where the accumulation left them
// (array-of-structures, by task id) and writes level 1 -- the pair sum and the odd item -- instead of k_gather_buckets
// writing W * B points that k_halve reads back: one launch and 2 x 75 MB of traffic less per 2^20-pair MSM.
template <class F> struct AosSrc;                   // a partial sum in the PointAoS layout the accumulation writes, read coordinate by coordinate
template <class P> struct AosSrc<Fp<P>> {
  BufRsrc rs; uint32_t off; bool valid;             // off: byte offset of the point; !valid: an empty bucket (the identity)
  __device__ __forceinline__ Fp<P> get(int coord) const {
    Fp<P> r = Fp<P>::zero();
    if (valid) {
#pragma unroll
      for (int k = 0; k < 9; ++k) r.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off + (uint32_t)(coord * 9 + k) * 4u, 0, 0);
    }
    return r;
  }
  __device__ __forceinline__ Fp<P> x() const { return get(0); }
  __device__ __forceinline__ Fp<P> y() const { return get(1); }
  __device__ __forceinline__ Fp<P> zz() const { return get(2); }
  __device__ __forceinline__ Fp<P> zzz() const { return get(3); }
};
template <class G> struct AosSrc<Fp2S<G>> {         // PointAoS<Fp2<G>>: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1, nine words each; a lane reads its half
  BufRsrc rs; uint32_t off; bool valid;
  __device__ __forceinline__ Fp2S<G> get(int coord) const {
    Fp2S<G> r = Fp2S<G>::zero();
    if (valid) {
      const uint32_t o = off + (uint32_t)(coord * 18 + 9 * Fp2S<G>::half()) * 4u;
#pragma unroll
      for (int k = 0; k < 9; ++k) r.v.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, o + (uint32_t)k * 4u, 0, 0);
    }
    return r;
  }
  __device__ __forceinline__ Fp2S<G> x() const { return get(0); }
  __device__ __forceinline__ Fp2S<G> y() const { return get(1); }
  __device__ __forceinline__ Fp2S<G> zz() const { return get(2); }
  __device__ __forceinline__ Fp2S<G> zzz() const { return get(3); }
};
template <class KF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_gather_halve(const uint32_t* pin, Level L, int W, int B, uint32_t* out, size_t out_stride) {
  KG_REDUCE_PRIO();
  constexpr uint32_t NWB = (uint32_t)PointIO<KF>::NW * 4u;       // bytes per partial sum (PointIO<Fp2S>::NW counts both halves of the lane pair)
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<KF>::N;
  const uint32_t n_out = (uint32_t)B / 2;
  if (t >= n_out * (uint32_t)W) return;
  const uint32_t w = t / n_out, i = t % n_out;
  const uint32_t b0 = w * (uint32_t)B + 2 * i;
  const uint32_t first = L.base[w];
  const uint2 cnt = *reinterpret_cast<const uint2*>(L.cnt + b0), rel = *reinterpret_cast<const uint2*>(L.rel + b0);
  const BufRsrc rin = soa_rsrc(pin), rout = soa_rsrc(out);
  const AosSrc<KF> p0{rin, (first + rel.x) * NWB, cnt.x != 0}, p1{rin, (first + rel.y) * NWB, cnt.y != 0};
  const uint32_t os4 = (uint32_t)out_stride * 4u;
  SoaDst<KF> sum{rout, os4, ((w * 2) * n_out + i) * 4u}, odd{rout, os4, ((w * 2 + 1) * n_out + i) * 4u};
  add_xyzz_stream<KF>(p0, p1, sum);
  copy_xyzz_stream<KF>(p1, odd);
}

// Buckets cut into a few tasks (every bucket of a merged sort; skewed inputs): the dense bucket array straight from the partial
// sums, a lane (lane pair for G2) per bucket.  The running sum lives in the bucket's own slot of the output array and every
// addition streams both operands (add_xyzz_stream is safe in place on its first operand): ~90 VGPRs instead of 141, so the
// gather of a proof's merged MSMs runs beside the next accumulation.
template <class F, class KF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_gather_sum(const uint32_t* pin, Level L, int W, int B, uint32_t* buckets) {
  KG_REDUCE_PRIO();
  constexpr uint32_t NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<KF>::N;
  const uint32_t total = (uint32_t)W * (uint32_t)B;
  if (t >= total) return;
  const uint32_t w = t / (uint32_t)B;
  uint32_t cnt = L.cnt[t];
  if (cnt > GATHER_SUM_MAX) cnt = 1;                   // a hot bucket: k_hot_sum left its sum in the first partial's place
  const BufRsrc rin = soa_rsrc(pin), rout = soa_rsrc(buckets);
  const uint32_t first = cnt ? L.base[w] + L.rel[t] : 0u;
  SoaDst<KF> dst{rout, total * 4u, t * 4u};
  const SoaSrc<KF> cur{rout, total * 4u, t * 4u};
  copy_xyzz_stream<KF>(AosSrc<KF>{rin, first * NWB, cnt != 0}, dst);
  for (uint32_t j = 1; j < cnt; ++j) {
    KG_STREAM_FENCE();
    add_xyzz_stream<KF>(cur, AosSrc<KF>{rin, (first + j) * NWB, true}, dst);
  }
}

// round r > 1: partial sums of the previous round (grouped by bucket through Lin) -> fewer partial sums.  Like k_gather_sum the
// running sum lives in its output slot (array of structures here) and both operands are streamed.
template <class F> struct AosDst;
template <class P> struct AosDst<Fp<P>> {
  BufRsrc rs; uint32_t off;
  __device__ __forceinline__ void put(int coord, const Fp<P>& v) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b32(v.l[k], rs, off + (uint32_t)(coord * 9 + k) * 4u, 0, 0);
  }
  __device__ __forceinline__ void x(const Fp<P>& v) const { put(0, v); }
  __device__ __forceinline__ void y(const Fp<P>& v) const { put(1, v); }
  __device__ __forceinline__ void zz(const Fp<P>& v) const { put(2, v); }
  __device__ __forceinline__ void zzz(const Fp<P>& v) const { put(3, v); }
};
template <class G> struct AosDst<Fp2S<G>> {
  BufRsrc rs; uint32_t off;
  __device__ __forceinline__ void put(int coord, const Fp2S<G>& v) const {
    const uint32_t o = off + (uint32_t)(coord * 18 + 9 * Fp2S<G>::half()) * 4u;
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b32(v.v.l[k], rs, o + (uint32_t)k * 4u, 0, 0);
  }
  __device__ __forceinline__ void x(const Fp2S<G>& v) const { put(0, v); }
  __device__ __forceinline__ void y(const Fp2S<G>& v) const { put(1, v); }
  __device__ __forceinline__ void zz(const Fp2S<G>& v) const { put(2, v); }
  __device__ __forceinline__ void zzz(const Fp2S<G>& v) const { put(3, v); }
};
template <class KF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_sum_tasks(const uint32_t* pin, Level Lin, Level L, int W, int B, uint32_t T2,
                                                  uint32_t* pout) {
  KG_REDUCE_PRIO();
  constexpr uint32_t NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<KF>::N;
  if (t >= L.base[W]) return;
  int w, b;
  uint32_t seg;
  locate_task(L, W, B, t, w, b, seg);
  const size_t bi = (size_t)w * B + b;
  const uint32_t len_all = Lin.cnt[bi];
  const uint32_t lo = seg * T2, hi = lo + T2 < len_all ? lo + T2 : len_all;
  const uint32_t first = Lin.base[w] + Lin.rel[bi];
  const BufRsrc rin = soa_rsrc(pin), rout = soa_rsrc(pout);
  const AosDst<KF> dst{rout, t * NWB};
  const AosSrc<KF> cur{rout, t * NWB, true};
  // byte offsets walked instead of indices multiplied, the loop kept rolled: with `(first + j) * NWB` inside an unrolled loop the kernel
  // spilled 44 B per lane under its 96-VGPR cap (G2: 72 B, now 40)
  uint32_t off = (first + lo) * NWB;
  const uint32_t end = (first + hi) * NWB;
  copy_xyzz_stream<KF>(AosSrc<KF>{rin, off, true}, dst);
#pragma clang loop unroll(disable)
  for (off += NWB; off < end; off += NWB) {
    KG_STREAM_FENCE();
    add_xyzz_stream<KF>(cur, AosSrc<KF>{rin, off, true}, dst);
  }
}

// raw internal XYZZ -> ABI words (x | y | zz | zzz), array-of-structures for the D2H copy
template <class P>
__device__ __forceinline__ void export_el(const Fp<P>& a, uint64_t* dst) {
  uint32_t w[8];
  to_ref(a, w);
  store_words(dst, 0, w);
}
template <class F>
__device__ __forceinline__ void export_el(const Fp2<F>& a, uint64_t* dst) { export_el(a.c0, dst); export_el(a.c1, dst + 4); }
template <class F>
__device__ __forceinline__ void export_el(const Fp2S<F>& a, uint64_t* dst) { export_el(a.v, dst + 4 * Fp2S<F>::half()); }
// ---- fused tail of the reduction --------------------------------------------------------------------------------
// Once the arrays are short (L <= TailCfg::L items) the remaining log2(L) levels run inside ONE launch: a workgroup per
// (window, array) keeps its items in LDS and walks the levels with barriers instead of kernel launches, then converts its
// results to the ABI form itself (the export).  Array 0 (the pair sums A) also spawns the new odd-index arrays T_s, T_s+1, ...
// and sums them on the lanes the halving frees: step t reads t live arrays of 2 * (L >> t) items and writes t + 1 arrays of
// L >> t items (array k at item k * (L >> t); the odd items of A become array t), i.e. t * (L >> t) pairs <= L / 2 lanes.
// Two LDS images used in turn (odd steps write A: L items, even steps write B: 3 L / 4 items), so a step's operands are read
// coordinate by coordinate while other lanes already write (add_xyzz_stream: the kernel fits the 96 VGPRs a resident
// accumulation leaves free; with both operands in registers and one image shrinking in place it took 159) and a step needs one
// barrier, not two.
// LDS image: structure of arrays, word k of lane-item q at img[k * stride + q] (consecutive lanes, consecutive banks); an Fq2
// item is two lane-items (the pair's halves).
template <class P> __device__ __forceinline__ uint32_t (&tail_limbs(Fp<P>& a))[9] { return a.l; }
template <class G> __device__ __forceinline__ uint32_t (&tail_limbs(Fp2S<G>& a))[9] { return a.v.l; }
template <class P> __device__ __forceinline__ const uint32_t (&tail_limbs(const Fp<P>& a))[9] { return a.l; }
template <class G> __device__ __forceinline__ const uint32_t (&tail_limbs(const Fp2S<G>& a))[9] { return a.v.l; }
template <class F> struct LdsPt {                   // one lane-item of an LDS image, coordinates read / written on demand
  uint32_t* img; uint32_t stride, li;
  __device__ __forceinline__ F get(int coord) const {
    F r;
#pragma unroll
    for (int k = 0; k < 9; ++k) tail_limbs(r)[k] = img[(uint32_t)(coord * 9 + k) * stride + li];
    return r;
  }
  __device__ __forceinline__ void put(int coord, const F& v) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) img[(uint32_t)(coord * 9 + k) * stride + li] = tail_limbs(v)[k];
  }
  __device__ __forceinline__ F x() const { return get(0); }
  __device__ __forceinline__ F y() const { return get(1); }
  __device__ __forceinline__ F zz() const { return get(2); }
  __device__ __forceinline__ F zzz() const { return get(3); }
  __device__ __forceinline__ void x(const F& v) const { put(0, v); }
  __device__ __forceinline__ void y(const F& v) const { put(1, v); }
  __device__ __forceinline__ void zz(const F& v) const { put(2, v); }
  __device__ __forceinline__ void zzz(const F& v) const { put(3, v); }
};
#ifndef KG_TAIL_L
#define KG_TAIL_L 256      // 512 (126 KiB of LDS, one halving launch fewer) measured level: EXPERIMENTS.md Part 0 section 3
#endif
template <class F> struct TailCfg { static constexpr int L = KG_TAIL_L; };    // items per array the fused tail takes over at
template <class G> struct TailCfg<Fp2S<G>> { static constexpr int L = 128; }; // (63 KiB of LDS either way: images of L and 3 L / 4 items)
static inline size_t tail_lds_bytes(uint32_t L, int lpt) { const uint32_t b = 3 * L / 4 ? 3 * L / 4 : 1; return (size_t)36 * (L + b) * lpt * 4; }

// LT: the array length as a compile-time constant (the usual case, TailCfg<F>::L: every LDS access is then base register +
// immediate offset), or 0 for the run-time length of a small window (B < TailCfg::L)
template <class F, int E64, int LT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HalveWaves<F>::MIN))) k_reduce_tail(const uint32_t* in, size_t in_stride, int narr_in, uint32_t Lrt, int c,
                                                     uint64_t* __restrict__ out) {
  KG_REDUCE_PRIO();
  extern __shared__ uint32_t lds[];
  constexpr uint32_t LPT = Lanes<F>::N;
  const uint32_t L = LT ? (uint32_t)LT : Lrt;
  const uint32_t capA = L * LPT;                     // lane-items of image A
  // the images are interleaved: word k of image A's item q at lds[k * S + q], of image B's at lds[k * S + capA + q].  The stride S is a
  // compile-time constant in both forms -- for a run-time length (a small window: L <= TailCfg::L / 2) that of the longest such array:
  // 36 word offsets per point are then immediates instead of registers (the run-time stride cost 60 B of scratch per lane, G2 224 B)
  constexpr uint32_t LS = LT ? (uint32_t)LT : (uint32_t)TailCfg<F>::L / 2;
  constexpr uint32_t S = (LS + (3 * LS / 4 ? 3 * LS / 4 : 1)) * LPT;
  uint32_t* const imgA = lds;
  uint32_t* const imgB = lds + capA;
  const uint32_t w = blockIdx.x / (uint32_t)narr_in, a = blockIdx.x % (uint32_t)narr_in;
  const uint32_t task = threadIdx.x / LPT, half = threadIdx.x % LPT;           // LPT consecutive lanes form a task
  const bool spawns = a == 0;
  int steps = 0;
  while ((1u << steps) < L) ++steps;
  const BufRsrc rin = soa_rsrc(in);
  const uint32_t is4 = (uint32_t)in_stride * 4u;
  const uint32_t src0 = (w * (uint32_t)narr_in + a) * L;
  for (int t = 1; t <= steps; ++t) {
    const uint32_t per = L >> t;                       // pairs per live array in this step
    const uint32_t live = spawns ? (uint32_t)t : 1u;
    const bool odd_step = (t & 1) != 0;
    uint32_t* const oimg = odd_step ? imgA : imgB;
    uint32_t* const iimg = odd_step ? imgB : imgA;
    const uint32_t ocap = S, icap = S;
    if (task < live * per) {
      const uint32_t k = task / per, q = task % per;
      const LdsPt<F> sum{oimg, ocap, (k * per + q) * LPT + half};
      const LdsPt<F> spawn{oimg, ocap, ((uint32_t)t * per + q) * LPT + half};    // the odd items of A become array t
      if (t == 1) {
        const SoaSrc<F> p0{rin, is4, (src0 + 2 * q) * 4u}, p1{rin, is4, (src0 + 2 * q + 1) * 4u};
        add_xyzz_stream<F>(p0, p1, sum);
        if (spawns) copy_xyzz_stream<F>(p1, spawn);
      } else {
        const LdsPt<F> p0{iimg, icap, (k * 2 * per + 2 * q) * LPT + half}, p1{iimg, icap, (k * 2 * per + 2 * q + 1) * LPT + half};
        add_xyzz_stream<F>(p0, p1, sum);
        if (spawns && k == 0) copy_xyzz_stream<F>(p1, spawn);
      }
    }
    __syncthreads();                                   // this step's image is complete; the other one is free to be overwritten
  }
  // results: one item per array -- A (or this workgroup's T array) is item 0, the array spawned at step j item j
  const uint32_t nres = spawns ? (uint32_t)steps + 1u : 1u;
  if (task < nres) {
    const bool in_a = (steps & 1) != 0;
    const LdsPt<F> p{in_a ? imgA : imgB, S, task * LPT + half};
    const int arr = task == 0 ? (int)a : narr_in - 1 + (int)task;        // 0 = A, 1 + l = T_l
    uint64_t* dst = out + ((size_t)w * c + arr) * 4 * E64;
    export_el(p.x(), dst);
    export_el(p.y(), dst + E64);
    export_el(p.zz(), dst + 2 * E64);
    export_el(p.zzz(), dst + 3 * E64);
  }
}

// The same tail for a reduction that has the chip to itself (a blocking call's last window group, an unsliced short call: nothing runs
// beside it, so the 96-VGPR budget of k_reduce_tail buys nothing): every addition of a level is computed by a QUAD of lanes (coop_add.h:
// four steps of one product instead of fourteen products in a row), 4.5 us per level instead of 12.  The workgroup first copies its
// array into LDS (items [0, L)), then walks the same steps as above -- odd steps write image X (items L ..), even steps image Y (items
// 0 ..: the input is dead after step 1) -- with 64 quads per round.  F is the one-lane field type (Fq2 for G2: the buffers' layout is the
// lane-pair type's, c0 planes then c1 planes per coordinate).
// NT threads per workgroup: 512 for the base-field curves (128 quads: every level of a 256-item array is one round), 256 for G2 (LDS)
template <class F> struct TailCoopNT { static constexpr int NT = 512; };
template <class G> struct TailCoopNT<Fp2<G>> { static constexpr int NT = 256; };
template <class F>
static size_t tail_coop_lds_bytes(uint32_t L) { return ((size_t)PointIO<F>::NW * 2 * L + coop_lds_words<F>(TailCoopNT<F>::NT / 4)) * 4; }
template <class F, int E64>
__global__ void __launch_bounds__(TailCoopNT<F>::NT) k_reduce_tail_coop(const uint32_t* in, size_t in_stride, int narr_in, uint32_t L, int c, uint64_t* __restrict__ out) {
  extern __shared__ uint32_t lds[];
  constexpr uint32_t NW = (uint32_t)PointIO<F>::NW;
  constexpr uint32_t NT = (uint32_t)TailCoopNT<F>::NT, NQ = NT / 4;
  const uint32_t CAP = 2 * L, tid = threadIdx.x, qd = tid >> 2;
  uint32_t* const img = lds;
  uint32_t* const ctmp = img + (size_t)NW * CAP;
  uint32_t* const cflg = ctmp + COOP_TMP_SLOTS * CoopEl<F>::E * NQ;
  const uint32_t w = blockIdx.x / (uint32_t)narr_in, a = blockIdx.x % (uint32_t)narr_in;
  const bool spawns = a == 0;
  int steps = 0;
  while ((1u << steps) < L) ++steps;
  const uint32_t src0 = (w * (uint32_t)narr_in + a) * L;
  for (uint32_t k = 0; k < NW; ++k)
    for (uint32_t it = tid; it < L; it += NT) img[(size_t)k * CAP + it] = in[(size_t)k * in_stride + src0 + it];
  __syncthreads();
  for (int t = 1; t <= steps; ++t) {
    const uint32_t per = L >> t;
    const uint32_t live = spawns ? (uint32_t)t : 1u, pairs = live * per;
    const bool odd_step = (t & 1) != 0;
    const uint32_t obase = odd_step ? L : 0u, ibase = odd_step ? 0u : L;
    for (uint32_t base = 0; base < pairs; base += NQ) {
      const uint32_t pi = base + qd;
      const bool on = pi < pairs;
      uint32_t i0 = 0, io = 0;
      if (on) {
        const uint32_t k = pi / per, q = pi % per;
        i0 = ibase + k * 2 * per + 2 * q;
        io = obase + k * per + q;
        if (spawns && k == 0) {                          // the odd item of array 0 spawns array t: lane j of the quad copies coordinate j
          const CoopQuad<F> cq{img, CAP, ctmp, NQ, qd, cflg, i0 + 1, i0 + 1, obase + (uint32_t)t * per + q, (int)(tid & 3u)};
          cq.st(cq.coord(cq.io, (uint32_t)cq.lane), cq.ld(cq.coord(i0 + 1, (uint32_t)cq.lane)));
        }
      }
      coop_add_level<F>(img, CAP, ctmp, cflg, on, i0, i0 + 1, io);
    }
  }
  const uint32_t nres = spawns ? (uint32_t)steps + 1u : 1u;
  if (tid < nres) {
    const uint32_t fin = (steps & 1) ? L : 0u;
    const XYZZ<F> p = PointIO<F>::load(img, CAP, fin + tid);
    const int arr = tid == 0 ? (int)a : narr_in - 1 + (int)tid;          // 0 = A, 1 + l = T_l
    uint64_t* dst = out + ((size_t)w * c + arr) * 4 * E64;
    export_el(p.x, dst);
    export_el(p.y, dst + E64);
    export_el(p.zz, dst + 2 * E64);
    export_el(p.zzz, dst + 3 * E64);
  }
}

// Hot buckets (more than GATHER_SUM_MAX partial sums: the sort lists them).  Stage 1: HOT_SPLIT workgroups per bucket each fold a
// contiguous share of its partial sums -- every task-lane sums a strided part into an LDS slot, then a tree over the slots -- into a
// scratch point; stage 2: one wave per bucket folds the HOT_SPLIT scratch points and leaves the total where the bucket's first partial
// sum was.  A 0/1-heavy witness against window tables puts 8 192 partial sums into one bucket: 4 + 7 + 4 additions deep instead of the
// 64 + 7 of a single workgroup (1.3 ms of G2 additions); the lane-by-lane rounds this replaced (k_task_count / k_scan_rows / k_row_bases /
// k_sum_tasks, sixteen partial sums per lane and round) took two rounds of four launches, 0.5-0.8 ms per MSM.
// The number of shares follows the fullest bucket (hot_split: ~512 partial sums per share, 16 .. one per task-lane of the folding
// workgroup): the 4 x 10^5 partial sums of a 2^24-pair witness' bucket of ones were 100 additions per lane in sixteen shares (0.97 ms).
constexpr uint32_t HOT_SPLIT = 16;                   // the fewest shares
template <class KF> constexpr uint32_t hot_split_max() { return 256u / Lanes<KF>::N; }
template <class KF>
static uint32_t hot_split(uint32_t max_tasks) {
  uint32_t sp = (max_tasks + 511u) / 512u;
  if (sp < HOT_SPLIT) sp = HOT_SPLIT;
  if (sp > hot_split_max<KF>()) sp = hot_split_max<KF>();
  return sp;
}
template <class KF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_hot_sum(const uint32_t* part, Level L, int B, const uint32_t* __restrict__ hot_list,
                                                                                                     uint32_t* scratch, uint32_t HOT_SPLIT) {
  KG_REDUCE_PRIO();
  extern __shared__ uint32_t lds[];                   // 36 words x 256 lane-items
  constexpr uint32_t LPT = Lanes<KF>::N, NT = 256 / LPT, NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = hot_list[blockIdx.x];
  const uint32_t w = t / (uint32_t)B, cnt = L.cnt[t], first = L.base[w] + L.rel[t];
  const uint32_t share = (cnt + HOT_SPLIT - 1) / HOT_SPLIT;
  const uint32_t lo = blockIdx.y * share, hi = lo + share < cnt ? lo + share : cnt;       // this workgroup's partial sums
  const uint32_t task = threadIdx.x / LPT, half = threadIdx.x % LPT;
  const BufRsrc rp = soa_rsrc(part), rs = soa_rsrc(scratch);
  const LdsPt<KF> mine{lds, 256u, threadIdx.x};
  if (lo + task < hi) {
    copy_xyzz_stream<KF>(AosSrc<KF>{rp, (first + lo + task) * NWB, true}, mine);
    for (uint32_t j = lo + task + NT; j < hi; j += NT) {
      KG_STREAM_FENCE();
      add_xyzz_stream<KF>(mine, AosSrc<KF>{rp, (first + j) * NWB, true}, mine);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 36; ++k) lds[(uint32_t)k * 256u + threadIdx.x] = 0u;      // the identity
  }
  __syncthreads();
  for (uint32_t h = NT / 2; h >= 1; h >>= 1) {
    if (task < h) {
      const LdsPt<KF> other{lds, 256u, (task + h) * LPT + half};
      add_xyzz_stream<KF>(mine, other, mine);
    }
    __syncthreads();
  }
  if (task == 0) {
    const AosDst<KF> dst{rs, (blockIdx.x * HOT_SPLIT + blockIdx.y) * NWB};
    copy_xyzz_stream<KF>(mine, dst);
  }
}
template <class KF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HalveWaves<KF>::MIN))) k_hot_fold(uint32_t* part, Level L, int B, const uint32_t* __restrict__ hot_list,
                                                                                                      const uint32_t* scratch, uint32_t HOT_SPLIT) {
  KG_REDUCE_PRIO();
  extern __shared__ uint32_t lds[];                   // 36 words x 256 lane-items
  constexpr uint32_t LPT = Lanes<KF>::N, NT = 256 / LPT, NWB = (uint32_t)PointIO<KF>::NW * 4u;
  const uint32_t t = hot_list[blockIdx.x];
  const uint32_t w = t / (uint32_t)B, first = L.base[w] + L.rel[t];
  const uint32_t task = threadIdx.x / LPT, half = threadIdx.x % LPT;
  const BufRsrc rp = soa_rsrc(part), rs = soa_rsrc(scratch);
  const LdsPt<KF> mine{lds, 256u, threadIdx.x};
  if (task < HOT_SPLIT) copy_xyzz_stream<KF>(AosSrc<KF>{rs, (blockIdx.x * HOT_SPLIT + task) * NWB, true}, mine);
  else {
#pragma unroll
    for (int k = 0; k < 36; ++k) lds[(uint32_t)k * 256u + threadIdx.x] = 0u;
  }
  __syncthreads();
  uint32_t top = NT / 2;
  while (top >= HOT_SPLIT && top > 1) top >>= 1;       // the first level that has a partner with data: skip the levels of identities
  for (uint32_t h = top; h >= 1; h >>= 1) {
    if (task < h) {
      const LdsPt<KF> other{lds, 256u, (task + h) * LPT + half};
      add_xyzz_stream<KF>(mine, other, mine);
    }
    __syncthreads();
  }
  if (task == 0) {
    const AosDst<KF> dst{rp, first * NWB};
    copy_xyzz_stream<KF>(mine, dst);
  }
}


}  // namespace
}  // namespace msm
}  // namespace kg
