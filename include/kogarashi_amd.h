/*
 * kogarashi_amd.h -- C ABI of the MI355X (gfx950) proving backend for Kogarashi's MSM + NTT hot path.
 *
 * The reference has no FFI boundary: the path sits behind crate-internal generic Rust
 * functions.  Each entry point below names the reference function it replaces; INTEGRATION.md shows the
 * Rust `extern "C"` shim a maintainer adds behind a cargo feature.
 *
 * Data formats (identical to the reference's in-memory representation):
 *   field element  : 4 x uint64 little-endian limbs, Montgomery form (x * 2^256 mod p), fully reduced
 *                    (bn254/src/fr.rs:71, bn254/src/fq.rs:48)
 *   G1 / Grumpkin affine base : 8 x uint64 = x | y, plus one uint8 infinity flag in a separate array
 *                    (bn254/src/g1.rs:18-22 is repr(Rust): the shim marshals x, y, is_infinity explicitly)
 *   G2 affine base : 16 x uint64 = x.c0 | x.c1 | y.c0 | y.c1 (+ flag)         (bn254/src/g2.rs:16-20)
 *   point outputs  : projective (x, y, z) in the reference's homogeneous coordinates, normalised so that
 *                    z = 1 (or (0, 1, 0) for the identity) -- i.e. to_extended(to_affine(result)),
 *                    macros/curve/weierstrass.rs:33-66.  The reference's own (X:Y:Z) triple depends on its
 *                    summation order; equality there is by cross-multiplication (group.rs:89-97).
 *
 * Pointers named d_* are DEVICE pointers (from kg_malloc, or any HIP allocation of the same device, e.g. a
 * torch tensor's data_ptr()); pointers named h_* / out_* are host pointers.
 * All functions return KG_OK (0) or a negative kg_status; none aborts, and no C++ exception leaves the library (a host allocation
 * that fails is KG_ERR_OOM, a worker thread that cannot be started KG_ERR_HIP; the context stays usable).  Calls on one kg_ctx are
 * serialised by the caller; distinct contexts may be used from distinct threads.
 */
#ifndef KOGARASHI_AMD_H
#define KOGARASHI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  KG_OK = 0,
  KG_ERR_NO_DEVICE = -1,   /* no gfx950 device visible / HIP runtime failed to initialise */
  KG_ERR_BAD_ARG = -2,     /* null pointer, bad length, log_n out of range, unknown enum */
  KG_ERR_OOM = -3,         /* device allocation failed */
  KG_ERR_HIP = -4,         /* a HIP call or kernel launch failed (see kg_last_error) */
  KG_ERR_UNSUPPORTED = -5,
  KG_ERR_CRS = -6,         /* delta is the identity: Error::ProverSubVersionCrsAttack (groth16/src/prover.rs:67-69) */
  KG_ERR_INVERSION = -7    /* a toxic scalar has no inverse: Error::ProverInversionFailed (groth16/src/zksnark.rs:37-38) */
} kg_status;

/* field / curve selectors */
enum { KG_FR = 0, KG_FQ = 1 };                       /* bn254 scalar field r, base field q */
enum { KG_G1 = 0, KG_GRUMPKIN = 1, KG_G2 = 2 };      /* curve ids */

typedef struct kg_ctx kg_ctx;

int kg_version(void);                    /* 6: kg_msm_set_small, kg_ctx_worker_threads, kg_ctx_trim, kg_ctx_queue_placement2 (kg_ctx_queue_placement is the version-4 entry again); 5: kg_msm_host_scalars, kg_commit_host_scalars, kg_tuning_describe, kg_mem_info, kg_groth16_setup_bn254, kg_experiments_built, kg_msm_host_slices; 4: kg_msm_set_groups, kg_ctx_queue_placement; 3: kg_init, kg_hw_queue_setting, kg_groth16_prove_sharded, kg_ntt_plan; only additions since 1 */
/* Optional process-level setup; call it (or export the variable yourself) BEFORE anything in the process initialises the
 * HIP runtime -- before the first kg_device_count / kg_ctx_create and before any other HIP user -- and before the host
 * starts threads (it calls setenv).  Sets GPU_MAX_HW_QUEUES=16 unless the variable is already set, so that each of a
 * context's five queues owns a hardware queue (the runtime's default of 4 is shared by every stream of the process; the
 * Groth16 prover measured 2.83 ms per proof with a queue each against 3.28 ms at the default; MSM throughput does not
 * depend on it).  Returns 1 if it set the variable, 0 if it was set already.  The library never modifies the environment
 * otherwise; everything works without this call. */
int kg_init(void);
/* The process's GPU_MAX_HW_QUEUES as an integer, 0 when unset (the runtime default of 4 applies). */
int kg_hw_queue_setting(void);
/* 1 when the library was compiled with -DKG_EXPERIMENTS (`python -m kogarashi_amd.build --experiments`: libkogarashi_amd_exp.so), which
 * adds three kernels that lost their A/B runs -- the round-3 first sort pass (KG_GS_TILE=0), the LDS-prefetching accumulation
 * (KG_ACC_PREFETCH=1) and the lane-pair G2 accumulation (KG_G2_PAIR_ACC=1) -- for further measurements; 0 for the product build, where
 * those knobs are ignored. */
int kg_experiments_built(void);
int kg_device_count(void);
const char* kg_strerror(int status);

/* One context per (process, GPU): owns a stream, twiddle caches and MSM work space. */
int kg_ctx_create(int device, kg_ctx** out);
void kg_ctx_destroy(kg_ctx* ctx);
const char* kg_last_error(kg_ctx* ctx);
/* Launch on a caller-owned HIP stream (e.g. torch's current stream) instead of the context's own. */
int kg_ctx_set_stream(kg_ctx* ctx, void* hip_stream);
int kg_ctx_sync(kg_ctx* ctx);
/* Stream semantics of the MSM entry points: by default an MSM's inputs may still be in flight on the context's stream
 * (e.g. produced by kg_field_vec_op just before); its scalar-side pipeline (digit extraction, bucket sort) then starts
 * behind everything enqueued so far.  on = 1 declares that device inputs handed to kg_msm / kg_msm_begin / kg_commit are
 * COMPLETE when the call is made (uploaded with kg_memcpy_h2d, or the caller synchronised): the scalar side of MSM i+1 then
 * runs on its own queue UNDER the accumulation of MSM i instead of behind it. */
int kg_ctx_set_inputs_complete(kg_ctx* ctx, int on);
/* Diagnostics: how the context's service queues were placed over the GPU's compute pipes.  The runtime deals a process's streams over
 * hardware queues in creation order and hardware queue k is served by pipe k mod 4; a queue that shares the main queue's pipe starts its
 * work only when an accumulation's last round of workgroups is placed (~0.7 ms late), so the context creates eight candidate streams on
 * first use, probes which of them share the main queue's pipe, and puts the scalar queue and the two reduction queues on the three
 * other pipes.  kg_ctx_queue_placement2 (version 6): *out_placement: 0 = the probe is switched off (KG_QUEUE_PLACEMENT=0), 1 = the probe
 * gave no clear picture (the queues are then taken in creation order, as up to version 3), 2 + j = probed, candidates j and j + 4 share
 * the main queue's pipe; returns KG_OK or a negative kg_status like every other entry; creates the queues if they do not exist yet.
 * kg_ctx_queue_placement is the version-4 entry with its version-4 meaning, kept so that a host built against that header keeps working
 * (version 5 had changed its signature in place): the placement itself is the return value -- 0 = probe off, 1 + j = probed, -1 = no
 * clear picture (which collides with KG_ERR_NO_DEVICE: why the out-parameter form exists); other negative values are kg_status codes. */
int kg_ctx_queue_placement(kg_ctx* ctx);
int kg_ctx_queue_placement2(kg_ctx* ctx, int* out_placement);

/* device memory plumbing so that non-HIP hosts (Rust shim, ctypes) never link the HIP runtime */
int kg_malloc(kg_ctx* ctx, size_t bytes, void** d_ptr);
int kg_free(kg_ctx* ctx, void* d_ptr);
/* kg_free keeps released blocks (up to KG_POOL_MB = 1024 MiB per context) for the next kg_malloc of the same size class: the first DMA
 * into a fresh allocation pays for mapping its pages (15-28 ms per 32 MiB).  Kept blocks are invisible to other allocators of the
 * device (hipMemGetInfo, the torch caching allocator, other processes see them as used) until they are given back: by kg_ctx_trim, when
 * a work-space allocation of ANY context of this process on the device fails (every context's kept blocks are released and the request
 * repeated), and at kg_ctx_destroy.  kg_free should be given the context that allocated the block; through another context the block is
 * released instead of kept.  kg_mem_info's free figure counts the context's kept bytes as free (they are, to this library) and so exceeds
 * hipMemGetInfo's by that amount. */
int kg_ctx_trim(kg_ctx* ctx);
/* Free and total memory of the context's device (hipMemGetInfo): a host sizes its resident CRS / keys with it.  Work spaces the
 * library keeps are grow-only and counted as used; an allocation the device refuses makes the call return KG_ERR_OOM with the
 * context intact (the call can be repeated once memory has been released). */
int kg_mem_info(kg_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int kg_memcpy_h2d(kg_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int kg_memcpy_d2h(kg_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int kg_memcpy_d2d(kg_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* ---- field vectors -------------------------------------------------------------------------------
 * Element-wise ops on n field elements (device).  Replaces the Montgomery limb functions
 * zkstd/src/arithmetic/limbs/bits_256/normal.rs:4-31 (add), 34-53 (sub), 56-80 (double), 83-121 (mul),
 * 124-166 (square), 170-184 (neg), 256-270 (invert) and, for Fr vectors, the point-wise polynomial ops
 * groth16/src/poly.rs:168-195.  `out` may alias an input. */
typedef enum { KG_OP_ADD = 0, KG_OP_SUB = 1, KG_OP_MUL = 2, KG_OP_SQUARE = 3, KG_OP_NEG = 4, KG_OP_DOUBLE = 5,
               KG_OP_INVERT = 6, KG_OP_FROM_MONT = 7, KG_OP_TO_MONT = 8 } kg_field_op;
int kg_field_vec_op(kg_ctx* ctx, int field, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n);
/* out[i] = a[i] + s * b[i]  (s: one element, HOST pointer): the vector folds of Nova's NIFS,
 * nova/src/relaxed_r1cs/witness.rs:56-70 (W = W1 + r W2, E = E1 + r T + r^2 E2) and instance.rs:81-101 (x = x1 + r x2);
 * Fr for the bn254 driver, Fq for the Grumpkin driver (nova/src/driver.rs:9-42). */
int kg_field_vec_axpy(kg_ctx* ctx, int field, const uint64_t* d_a, const uint64_t* h_s, const uint64_t* d_b, uint64_t* d_out, size_t n);
/* out[i] = start * base^i, i < n  (start, base: one element each, HOST pointers): the `scan(one, *= g)` tables of
 * fft.rs:35-41,56-70 and the powers of tau of the trusted setup, groth16/src/zksnark.rs:44-49. */
int kg_field_powers(kg_ctx* ctx, int field, const uint64_t* h_start, const uint64_t* h_base, uint64_t* d_out, size_t n);
/* out[i] = a[i] * s  (s: one element, HOST pointer); fft.rs:104,150-154 */
int kg_field_vec_scale(kg_ctx* ctx, int field, const uint64_t* d_a, const uint64_t* h_s, uint64_t* d_out, size_t n);

/* ---- NTT -------------------------------------------------------------------------------------------
 * groth16/src/fft.rs: Fft::<Fr>::dft (:92-97), idft (:100-106), coset_dft (:109-116), coset_idft (:119-127),
 * divide_by_z_on_coset (:150-154).  d_data holds n = 2^log_n elements, natural order in and out, in
 * place; 1 <= log_n <= 28 (S = 28, bn254/src/fr.rs:53).  The caller zero-pads (prepare_fft :157-162). */
int kg_ntt_bn254_fr(kg_ctx* ctx, uint64_t* d_data, uint32_t log_n, int inverse, int coset);
int kg_fr_divide_by_z_on_coset(kg_ctx* ctx, uint64_t* d_data, uint32_t log_n);
/* How a transform of 2^log_n elements is decomposed (no device needed; reporting only: bench.py prices the roofline with it):
 * returns the number of steps s in 1..3 (one HBM round trip each, 0 for log_n outside 1..28); log_m[i] = log2 of step i's DFT
 * length (they sum to log_n), log_tile[i] = log2 of the elements one workgroup holds in LDS.  KG_NTT_STEPS / KG_NTT_TILE apply. */
int kg_ntt_plan(uint32_t log_n, uint32_t* log_m, uint32_t* log_tile);   /* three entries each */

/* ---- MSM -------------------------------------------------------------------------------------------
 * groth16/src/msm.rs:6-48 msm_curve_addition(bases, coeffs): sum_i coeffs[i] * bases[i] over n pairs
 * (n = min(len), resolved by the caller, msm.rs:25).  d_inf may be NULL (no identity bases).
 * Scalars: Fr for KG_G1 / KG_G2, Fq for KG_GRUMPKIN (nova/src/driver.rs:9-42).
 * out_xyz: HOST, 12 (G1, Grumpkin) or 24 (G2) uint64.  n == 0 yields the identity. */
int kg_msm(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars,
           size_t n, uint64_t* out_xyz);
/* Split form for callers with several MSMs to run (a prover issues five per proof): kg_msm_begin enqueues the whole
 * device pipeline of one MSM and returns; kg_msm_end waits for it and runs the short host finish.  Up to 4 MSMs may be
 * in flight (ticket = 0..3, reusable after its kg_msm_end); results are identical to kg_msm.  While MSM i+1 sorts and
 * accumulates, MSM i's bucket reduction (side stream) and host finish overlap it. */
int kg_msm_begin(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars,
                 size_t n, int ticket);
int kg_msm_end(kg_ctx* ctx, int curve, int ticket, uint64_t* out_xyz);
/* Same with HOST inputs: the call shape of the Rust slices.  The arrays are cut into up to four index slices that are
 * uploaded (into device buffers the context keeps), sorted and accumulated as a pipeline -- slice j accumulates while
 * slice j + 1 is still on the bus -- and the slices' sums are added on the host. */
int kg_msm_host(kg_ctx* ctx, int curve, const uint64_t* h_bases, const uint8_t* h_inf, const uint64_t* h_scalars,
                size_t n, uint64_t* out_xyz);
/* The call shape of the reference's call sites: the bases are FIXED per circuit / per commitment key (a CRS vector of
 * groth16/src/params.rs:6-28, the generators of nova/src/pedersen.rs:6-13) and live on the device -- register them once with
 * kg_bases_register -- while the scalars are a fresh HOST slice per call (groth16/src/msm.rs:6 `coeffs`, pedersen.rs:15 `m`).
 * d_bases / d_inf: device (any whole-point offset into a registered array is served from its resident copy); h_scalars: HOST,
 * pageable or pinned.  From 2^19 pairs the scalars are uploaded in index slices, each sorted and accumulated while the next one is
 * still on the bus; only the first, short slice's upload is exposed.  Shorter calls run the blocking kg_msm behind one copy (the
 * whole upload is shorter than what a second slice costs).  Result identical to kg_msm on the uploaded scalars. */
int kg_msm_host_scalars(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* h_scalars,
                        size_t n, uint64_t* out_xyz);
/* How a host-array MSM of n pairs is cut into index slices (no device needed; reporting and tests): lo[0 .. K] are the slice boundaries
 * (lo[0] = 0, lo[K] = n, 9 entries at most), K is returned (0 for n = 0).  scalars_only = 1: kg_msm_host_scalars / kg_commit_host_scalars
 * (K = 1 below 2^19 pairs, 2 up to 2^20, 3 / 4 / 6 / 8 at 2^21 / 22 / 23 / 24; the first slice is half a share: its upload is the one
 * nothing hides), 0: kg_msm_host (1 / 2 / 4 equal slices: below 2^18, below 2^20, from there).  KG_HOST_SLICES / KG_HOST_FIRST_DIV apply. */
int kg_msm_host_slices(size_t n, int scalars_only, size_t* lo);
/* nova/src/pedersen.rs:15-20 PedersenCommitment::commit: affine(sum_i m[i] * g[i]).
 * out_xy: HOST, 8 or 16 uint64; *out_inf = 1 for the identity (then out_xy = (0, 1)). */
int kg_commit(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* d_scalars,
              size_t n, uint64_t* out_xy, uint8_t* out_inf);
/* kg_commit with the scalars in HOST memory (see kg_msm_host_scalars): what nova/src/pedersen.rs:15-20 does per call against
 * a resident key. */
int kg_commit_host_scalars(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, const uint64_t* h_scalars,
                           size_t n, uint64_t* out_xy, uint8_t* out_inf);
/* Per-GPU partial for the sharded commit: the un-normalised device result (raw window sums) is reduced on
 * the host to ONE affine partial; ranks exchange these (RCCL all_gather of 17/33 words) and add them with
 * kg_points_sum_affine.  See DESIGN.md "Multi-GPU". */
int kg_points_sum_affine(kg_ctx* ctx, int curve, const uint64_t* h_points_xy, const uint8_t* h_inf, size_t count,
                         uint64_t* out_xy, uint8_t* out_inf);
/* Resident bases (CRS vectors, commitment keys): registering an array converts it ONCE to the device's internal form
 * (the reference re-reads and re-converts its bases on every call).  Afterwards every entry point that is handed a
 * pointer inside a registered array (kg_msm, kg_msm_begin, kg_commit, kg_groth16_prove_bn254; any whole-point offset,
 * e.g. params.a[cs.l()..]) skips the per-call conversion.  The caller must not modify a registered array or its
 * flag array; the resident copy is used when a call passes the registered flag array at the same offset (or NULL
 * where NULL was registered) -- any other d_inf is honoured by converting per call.  kg_bases_unregister(d_bases)
 * releases it. */
int kg_bases_register(kg_ctx* ctx, int curve, const uint64_t* d_bases, const uint8_t* d_inf, size_t n);
int kg_bases_unregister(kg_ctx* ctx, const uint64_t* d_bases);
/* Window tables for a registered array (fixed bases: a CRS vector, a commitment key): stores 2^(c*w) * base[i] for every
 * window w next to the resident copy -- ceil(255 / c) x the array, 64 B per G1 / Grumpkin point and window, 128 B per G2
 * point.  An MSM over the WHOLE array (kg_msm, kg_msm_begin, kg_commit, and kg_groth16_prove_bn254 when all five CRS
 * vectors carry tables) then sorts the digits of all windows into ONE set of 2^(c-1) buckets: the bucket reduction and the
 * host finish shrink by the window count, and the window can be one bit wider (c = 17: 15 additions per scalar instead
 * of 16-17).  Results are bit-identical.  msm_len = the length of the MSMs the array will meet (0 = the array's own
 * length; the Groth16 vector l meets the witness z = x || w, so its msm_len is |z|); offered for 2^16 <= msm_len <= 2^20,
 * KG_ERR_BAD_ARG otherwise.  The reference has no counterpart (groth16/src/msm.rs re-reads affine bases per call). */
int kg_bases_precompute(kg_ctx* ctx, const uint64_t* d_bases, size_t msm_len);
/* Window width c of the tables kg_bases_precompute builds for MSMs of msm_len scalars, 0 where none are offered: a table holds
 * ceil(255 / c) rows of the array (64 B per G1 / Grumpkin point, 128 B per G2 point).  Pure function: no device, no context. */
int kg_msm_table_window(size_t msm_len);
/* Tuning knob: window width c, 0 (automatic) .. 20; KG_ERR_BAD_ARG otherwise.  Widths above 16 need the two-pass sort (2^16 .. 2^24
 * scalars; shorter or longer MSMs fall back to 16); 19 and 20 are the wide windows of the 2^23 .. 2^24-pair commitments. */
int kg_msm_set_window(kg_ctx* ctx, int c);
/* Tuning knob: window groups of a blocking kg_msm / kg_commit (groth16/src/msm.rs:6-48 and nova/src/pedersen.rs:15-20 are blocking
 * calls).  A blocking MSM of 2^17 .. 2^24 pairs pipelines against itself (2^23: in index slices instead): the scalars are converted once, then the windows are sorted,
 * accumulated and reduced in groups, top windows first, the next group's sort under this group's accumulation, this group's reduction
 * under the next one's accumulation, and the host's double-and-add chain starts on the top group's sums.  groups: 0 = automatic (two for 2^17 .. 2^21 pairs, three from 2^22, four for the 19- and 20-bit windows of 2^23 .. 2^24 pairs),
 * 1 = none (one accumulation launch per MSM), 2 .. 4.  Results are bit-identical for every setting. */
int kg_msm_set_groups(kg_ctx* ctx, int groups);
/* Tuning knob: the short-input MSM (csrc/msm_small.hip).  Blocking MSMs of up to max_pairs pairs (default and most: 32768, G2: 20480; 0 = never;
 * kg_msm_begin: KG_SMALL_MAX_FLIGHT, 8192) run as
 * ONE launch -- a workgroup per window: digits, an LDS counting sort, bucket accumulation, the bucket reduction and the window sum in
 * LDS -- two launches from 1537 pairs (a window's buckets spread over several workgroups), three from 2049 (the scalars are converted
 * once, into word planes, for all workgroups); the host finishes with one addition per window.  Where it pays (G1 / Grumpkin up to 6144
 * pairs, G2 up to 16384; KG_SMALL_GLV) every scalar is split by the curve's endomorphism into two 127-bit halves: half the windows, half the
 * host chain.  These are the lengths of the reference's own tests and bench (groth16/src/msm.rs:118-135: 32 pairs; bn254/benches: 2^10).
 * c: window width, 0 = by length, 2 .. 10; r: log2 of the buckets one workgroup owns, -1 = by length, 0 .. 7.  -2 for max_pairs keeps the
 * current value.  Results are bit-identical for every setting; a forced window (kg_msm_set_window) selects the long pipeline. */
int kg_msm_set_small(kg_ctx* ctx, int max_pairs, int c, int r);
/* Host worker threads the context has started so far (host finishes of tickets and slices, proof assemblies, the uploader of a
 * host-scalar call).  They are started on demand, kept, and reused: after the first calls of a kind the count stays where it is (no
 * thread creation per call).  KG_POOL_MAX_THREADS caps it (64); a call that would need one more fails with KG_ERR_HIP, never aborts. */
int kg_ctx_worker_threads(kg_ctx* ctx, int* started);
/* The automatic rule: window width c for n pairs (W = ceil(255 / c) signed windows of 2^(c-1) buckets; the reference's
 * rule is groth16/src/msm.rs:7-14).  Pure function: needs no device and no context. */
int kg_msm_pick_window(size_t n);

/* ---- several devices, one process ------------------------------------------------------------------
 * MSM / commitment over n_ctx contexts (normally one per GPU of the node; several contexts on one GPU also work): the
 * index range is cut into contiguous slices, context i holds slice i (kg_shard_range gives the cut), one host thread per
 * context runs the single-device pipeline on its slice and the n_ctx affine partial sums are added on the host -- the
 * exchange is 72 B (G1) per device, so no collective library is involved.  d_bases[i] / d_inf[i] / d_scalars[i] are
 * DEVICE pointers of context i's device, n_local[i] pairs each (d_inf or d_inf[i] may be NULL).  Results are identical
 * to kg_commit / kg_msm over the concatenated arrays.  The multi-process form of the same sharding (one rank per GPU,
 * RCCL all_gather of the partials) is kogarashi_amd/dist.py. */
int kg_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi);
int kg_commit_sharded(kg_ctx* const* ctxs, int n_ctx, int curve, const uint64_t* const* d_bases, const uint8_t* const* d_inf,
                      const uint64_t* const* d_scalars, const size_t* n_local, uint64_t* out_xy, uint8_t* out_inf);
int kg_msm_sharded(kg_ctx* const* ctxs, int n_ctx, int curve, const uint64_t* const* d_bases, const uint8_t* const* d_inf,
                   const uint64_t* const* d_scalars, const size_t* n_local, uint64_t* out_xyz);
/* nova/src/pedersen.rs:6-20 PedersenCommitment<C> { g } spread over the devices: _create (= new, given the generators)
 * uploads slice i of the HOST arrays h_bases / h_inf (may be NULL) to context i and registers it (kg_bases_register);
 * _commit takes the HOST scalar vector m (n elements; zip semantics: min(n, key length) pairs), uploads each device's
 * slice and returns affine(sum_i m[i] * g[i]) like kg_commit.  The contexts must outlive the key. */
typedef struct kg_sharded_key kg_sharded_key;
int kg_sharded_key_create(kg_ctx* const* ctxs, int n_ctx, int curve, const uint64_t* h_bases, const uint8_t* h_inf, size_t n,
                          kg_sharded_key** out);
void kg_sharded_key_destroy(kg_sharded_key* key);
size_t kg_sharded_key_len(const kg_sharded_key* key);
int kg_sharded_key_commit(kg_sharded_key* key, const uint64_t* h_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf);

/* ---- fixed-base multiples ------------------------------------------------------------------------
 * out[i] = affine(generator * k[i]): the `(g * scalar).into()` of the CRS construction (groth16/src/zksnark.rs:57,
 * 168-187), of VerifyingKey (zksnark.rs:104-112) and of Group::random (macros/curve/weierstrass/group.rs:39-41,
 * used by PedersenCommitment::new, nova/src/pedersen.rs:10-13).  d_k: n scalars of the curve's scalar field;
 * d_out_xy: n x (8 | 16) words; d_out_inf: n flags (k = 0 gives the identity, stored as (0, 1) + flag 1). */
int kg_fixed_base_mul(kg_ctx* ctx, int curve, const uint64_t* d_k, size_t n, uint64_t* d_out_xy, uint8_t* d_out_inf);

/* ---- Groth16 prover --------------------------------------------------------------------------------
 * groth16/src/prover.rs:20-99 Prover::create_proof after circuit synthesis.  The CRS (groth16/src/params.rs:6-28)
 * is resident on the device; vk elements needed for the proof assembly are host affine points. */
typedef struct {
  size_t m, l, m_l_1;                               /* cs.m(), cs.l(), cs.m_l_1() (zkstd/src/r1cs.rs:30-40) */
  const uint64_t* d_h;    const uint8_t* d_h_inf;    /* m - 1 G1            */
  const uint64_t* d_l;    const uint8_t* d_l_inf;    /* m_l_1 G1            */
  const uint64_t* d_a;    const uint8_t* d_a_inf;    /* l + m_l_1 G1        */
  const uint64_t* d_b_g1; const uint8_t* d_b_g1_inf; /* l + m_l_1 G1        */
  const uint64_t* d_b_g2; const uint8_t* d_b_g2_inf; /* l + m_l_1 G2        */
  uint64_t alpha_g1[8], beta_g1[8], delta_g1[8];    /* vk, affine x | y    */
  uint64_t beta_g2[16], delta_g2[16];
  uint8_t delta_g1_inf, delta_g2_inf;
} kg_groth16_crs;
/* d_a_eval / d_b_eval / d_c_eval: cs.evaluate() (m elements each); d_x = cs.x() (l), d_w = cs.w() (m_l_1);
 * r, s: HOST, the prover's blinding scalars (the reference draws them from its RNG, prover.rs:71-72).
 * proof_out: HOST, A (8) | B (16) | C (8) words, affine; proof_inf[3] identity flags. */
int kg_groth16_prove_bn254(kg_ctx* ctx, const kg_groth16_crs* crs, const uint64_t* d_a_eval, const uint64_t* d_b_eval,
                           const uint64_t* d_c_eval, const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r,
                           const uint64_t* s, uint64_t* proof_out, uint8_t* proof_inf);
/* Two proofs in flight (tickets 0 and 1), for provers that make proofs back to back: _begin enqueues the whole proof and
 * starts its host-side assembly on worker threads; _end waits for it and writes the proof (same layout and status codes
 * as kg_groth16_prove_bn254).  _begin(i + 1) before _end(i) lets the next proof's transforms and sorts run while the
 * previous proof's last reduction, host finish and assembly complete.  crs and the device inputs must stay valid
 * until the matching _end.  Every _begin must be followed by its _end (also after an error: _end returns the status).
 * Result slots of kg_msm, of the kg_msm_begin tickets and of the two proof tickets are disjoint, so these calls may be
 * interleaved on one context. */
int kg_groth16_prove_begin(kg_ctx* ctx, const kg_groth16_crs* crs, const uint64_t* d_a_eval, const uint64_t* d_b_eval,
                           const uint64_t* d_c_eval, const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r,
                           const uint64_t* s, int ticket);
int kg_groth16_prove_end(kg_ctx* ctx, int ticket, uint64_t* proof_out, uint8_t* proof_inf);

/* One proof over n_ctx contexts (normally one per GPU of the node; several contexts on one GPU also work), task-parallel
 * -- the MSMs of prover.rs:51-65 are independent until the assembly, the G2 query is the long pole.  Context 0
 * runs the b_g2 query against z = x || w, context 1 % n the three G1 queries against z (a, b_g1, l), context 2 % n the
 * transforms, h = (a o b - c) / Z and h's MSM; the five sums meet on the host (5 x 72..144 B), where prover.rs:75-92 runs.
 * crs[i] and the d_*[i] inputs are context i's OWN copies (device pointers of its device).  A context needs only what its
 * part reads: d_b_g2 (+ x, w) for the first, d_a / d_b_g1 / d_l (+ x, w) for the second, d_h and the three evaluation
 * vectors for the third (which reads neither x nor w); everything else may be NULL.  m, l, m_l_1 must agree; alpha, beta, delta
 * are taken from crs[0].  Contexts beyond the third are left idle.  The proof is bit-identical to
 * kg_groth16_prove_bn254's.  No proof may be in flight (ticket 0) on the contexts used. */
int kg_groth16_prove_sharded(kg_ctx* const* ctxs, int n_ctx, const kg_groth16_crs* const* crs, const uint64_t* const* d_a_eval,
                             const uint64_t* const* d_b_eval, const uint64_t* const* d_c_eval, const uint64_t* const* d_x,
                             const uint64_t* const* d_w, const uint64_t* r, const uint64_t* s, uint64_t* proof_out, uint8_t* proof_inf);

/* a sparse matrix in CSR form: row_ptr (rows + 1 entries), col and val (4 words per entry); all device pointers */
typedef struct { const uint64_t* d_row_ptr; const uint64_t* d_col; const uint64_t* d_val; } kg_csr;

/* ---- Groth16 setup ---------------------------------------------------------------------------------
 * groth16/src/zksnark.rs:17-127 ZkSnark::setup after circuit synthesis: the CRS of a circuit from the five toxic scalars.
 * a, b, c: the constraint matrices as CSR over z = x || w (m rows; the same arrays kg_groth16_prove_r1cs_bn254 takes; all device
 * pointers).  h_toxic: HOST, 5 x 4 words alpha | beta | gamma | delta | tau (the reference draws them from its rng in that order,
 * zksnark.rs:28-32 -- the shim draws them the same way and passes them in).
 * crs: on entry its d_* pointers name caller-allocated DEVICE arrays (kg_malloc) of the lengths the struct documents -- h: m - 1,
 * l: m_l_1, a / b_g1 / b_g2: l + m_l_1 points, one flag byte per point; on return they hold Parameters { h, l, a, b_g1, b_g2 }
 * (groth16/src/params.rs:6-28), and m, l, m_l_1, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2 and the two delta flags are
 * filled in: the struct is ready for kg_groth16_prove_bn254 (register its arrays to keep them in the MSM's internal form).
 * d_ic / d_ic_inf: DEVICE, l points -- vk.ic; out_gamma_g2: HOST, 16 words; out_vk_inf: HOST, 6 flags in the order alpha_g1,
 * beta_g1, delta_g1, beta_g2, gamma_g2, delta_g2 (a zero toxic scalar gives an identity).
 * The powers of tau, the idft, the matrices' transposition (zkstd/src/matrix.rs:17-29 x_and_w), the three transposed products
 * (eval_at_tau, zksnark.rs:190-194), the linear combinations and every generator multiple run on the device; the host computes
 * two inversions and tau^n.  KG_ERR_INVERSION when gamma or delta is zero.  Blocking: all outputs are complete on return. */
int kg_groth16_setup_bn254(kg_ctx* ctx, const kg_csr* a, const kg_csr* b, const kg_csr* c, size_t m, size_t l, size_t m_l_1,
                           const uint64_t* h_toxic, kg_groth16_crs* crs, uint64_t* d_ic, uint8_t* d_ic_inf, uint64_t* out_gamma_g2,
                           uint8_t* out_vk_inf);

/* ---- R1CS evaluation -------------------------------------------------------------------------------
 * zkstd/src/matrix.rs:31-33 SparseMatrix::evaluate_with_z (row.rs:43-51): out[i] = sum_e val[e] * z[col[e]] over the
 * entries row_ptr[i] <= e < row_ptr[i+1] of a CSR matrix with m rows; z = x || w (instance wires first).  The step
 * right before the prover's NTTs (cs.evaluate(), zkstd/src/r1cs.rs:137-142).  All pointers are device pointers. */
int kg_r1cs_evaluate(kg_ctx* ctx, const uint64_t* d_row_ptr, const uint64_t* d_col, const uint64_t* d_val, size_t m,
                     const uint64_t* d_z, uint64_t* d_out);
/* zkstd/src/matrix.rs:36-48 SparseMatrix::prod over either scalar field (Fr: Bn254Driver, Fq: GrumpkinDriver,
 * nova/src/driver.rs:9-42): the same CSR product with z = (u | x | w) -- prod's wire arithmetic (Instance(i) -> z[i],
 * Witness(i) -> z[i + l]) is resolved when the CSR columns are written.  kg_r1cs_evaluate is the Fr case. */
int kg_r1cs_prod(kg_ctx* ctx, int field, const uint64_t* d_row_ptr, const uint64_t* d_col, const uint64_t* d_val, size_t m,
                 const uint64_t* d_z, uint64_t* d_out);

/* ---- Nova cross term -------------------------------------------------------------------------------
 * nova/src/prover.rs:53-90 Prover::compute_cross_term: T = AZ1 o BZ2 + AZ2 o BZ1 - u1 * CZ2 - u2 * CZ1 (m elements), the
 * vector Prover::prove commits to next (commit_t = ck.commit(&t), prover.rs:35: kg_commit on d_out).  a, b, c: the shape's
 * matrices as CSR over z = (u | x | w); d_z1 / d_z2: the two z vectors (relaxed instance-witness pair and the fresh one);
 * h_u1, h_u2: HOST, one element each (the reference passes instance1.u and one).  One fused kernel: each matrix row is
 * read once, the six products stay in registers. */
int kg_nova_cross_term(kg_ctx* ctx, int field, const kg_csr* a, const kg_csr* b, const kg_csr* c, size_t m, const uint64_t* d_z1,
                       const uint64_t* d_z2, const uint64_t* h_u1, const uint64_t* h_u2, uint64_t* d_out);
/* create_proof with cs.evaluate() (zkstd/src/r1cs.rs:137-142, prover.rs:33) on the device as well: the constraint matrices
 * a, b, c (CSR over z = x || w, m = crs->m rows; resident, uploaded once per circuit) take the place of the three
 * evaluation vectors -- each transform chain starts with its matrix-vector product.  Otherwise as kg_groth16_prove_bn254 /
 * kg_groth16_prove_begin (the proof is collected with kg_groth16_prove_end). */
int kg_groth16_prove_r1cs_bn254(kg_ctx* ctx, const kg_groth16_crs* crs, const kg_csr* a, const kg_csr* b, const kg_csr* c,
                                const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r, const uint64_t* s, uint64_t* proof_out,
                                uint8_t* proof_inf);
int kg_groth16_prove_r1cs_begin(kg_ctx* ctx, const kg_groth16_crs* crs, const kg_csr* a, const kg_csr* b, const kg_csr* c,
                                const uint64_t* d_x, const uint64_t* d_w, const uint64_t* r, const uint64_t* s, int ticket);

/* ---- deterministic synthetic inputs (identical streams in oracle/) -------------------- */
int kg_gen_scalars(kg_ctx* ctx, int field, uint64_t seed, size_t start, size_t n, uint64_t* d_out);
int kg_gen_bases(kg_ctx* ctx, int curve, uint64_t seed, size_t start, size_t n, uint64_t* d_out); /* G1, Grumpkin */

/* ---- tuning knobs ---------------------------------------------------------------------------------------------------
 * Every environment variable the library reads (experiment switches, scheduling parameters; results are bit-identical for every
 * setting) lives in ONE table (kogarashi_amd/csrc/tuning.h), parsed once per process.  kg_tuning_describe(-1, ...) returns the
 * number of rows; for 0 <= index < rows it writes the variable's name, its description, the default and the value this process
 * runs with (any out-pointer may be NULL).  README.md's table is generated from it (tools/gen_knob_table.py). */
int kg_tuning_describe(int index, const char** env, const char** doc, int* dflt, int* value);

/* ---- timing: when enabled, the library brackets its device phases with HIP events on its stream ------ */
int kg_profile_enable(kg_ctx* ctx, int on);       /* (re)starts the accumulation */
/* per phase name: summed milliseconds and number of occurrences since kg_profile_enable; returns the number of
 * distinct names written (<= cap); names are static strings.  Synchronises the context's streams. */
int kg_profile_summary(kg_ctx* ctx, const char** names, float* total_ms, int* counts, int cap);
int kg_profile_last(kg_ctx* ctx, const char** names, float* ms, int cap);   /* = summary without the counts */

#ifdef __cplusplus
}
#endif
#endif /* KOGARASHI_AMD_H */
