// tuning.h -- every environment knob of the library, in ONE table, parsed ONCE.
//
// The knobs are experiment switches and scheduling parameters (results are bit-identical for every setting: tests/ run the
// parity suite over the ones that change a code path).  The process-wide values are read from the environment the first time
// kg::tuning() is called; every kg_ctx takes a copy at creation (kg_ctx::tune), which is what the code consults where it has a
// context.  kg_tuning_describe (C ABI) walks the table: README.md's knob table is generated from it
// (tools/gen_knob_table.py), so the documentation cannot drift from the code.
//
// -1 means "automatic" wherever the default column says so.
#pragma once
#include <string>

// X(field, "ENV_NAME", default, "what it does")
#define KG_TUNING_TABLE(X)                                                                                                                   \
  /* ---- MSM: windows, groups, slices ------------------------------------------------------------------------------------------------ */ \
  X(wide_window, "KG_WIDE_WINDOW", 23, "log2 of the length from which a blocking MSM takes the 20-bit window (13 windows, unsliced, window groups); 0 = never (index slices with c = 17)") \
  X(msm_groups, "KG_MSM_GROUPS", -1, "window groups of a blocking MSM: -1 automatic, 0/1 none, k = k equal groups; a list \"a,b,c\" names the groups' window counts from the top window down") \
  X(msm_sliced, "KG_MSM_SLICED", 1, "0 = a blocking MSM of 2^23 pairs or more runs in window groups instead of four index slices") \
  X(msm_t, "KG_MSM_T", -1, "task length of the accumulation (entries per lane): -1 = 2 n / B + 16") \
  X(merged_t, "KG_MERGED_T", -1, "task length of a merged (window-table) accumulation: -1 = one resident round of lanes") \
  X(hot_shift, "KG_HOT_SHIFT", -1, "hot buckets are cut 2^shift times finer: -1 = 2 (0 for a merged sort)") \
  X(hot_sum, "KG_HOT_SUM", 1, "0 = hot buckets go through the generic partial-sum rounds instead of the workgroup-wide trees") \
  X(gather_fuse, "KG_GATHER_FUSE", 1, "0 = the bucket gather writes the dense bucket array instead of fusing the first halving level") \
  X(fmt64_min_log, "KG_FMT64_MIN_LOG", 0, "log2 of the array length from which resident bases take the 64-byte point form (30 = always the 72-byte limb form)") \
  X(table64, "KG_TABLE64", 1, "0 = window tables in the 72-byte limb form") \
  X(pipe_accq, "KG_PIPE_ACCQ", 1, "2 = MSM tickets alternate between two accumulation queues (measured level; off)") \
  X(coop_tail, "KG_COOP_TAIL", 1, "0 = the last reduction of a blocking MSM runs the streamed tail kernel (96 VGPRs) instead of the lane-cooperative one") \
  X(small_max, "KG_SMALL_MAX", 32768, "longest MSM (pairs) that runs as the short-input kernel (msm_small.hip): 0 = never, at most 32768 (G2: 20480)") \
  X(small_max_flight, "KG_SMALL_MAX_FLIGHT", 8192, "longest MSM begun with kg_msm_begin that runs as the short-input kernel (a call in flight shares the chip with its neighbours, and the kernel's grid fills it at these lengths); G2: half of it") \
  X(small_glv, "KG_SMALL_GLV", 1, "short-input MSMs of up to 16384 pairs split every scalar into two 127-bit halves k1 + k2 lambda against P and (beta x, y): half the windows, half the host chain (0 = off, 1 = where it pays: G1 / Grumpkin up to 6144 pairs, G2 up to 16384, 2 = wherever the index field allows)") \
  X(small_kt_from, "KG_SMALL_KT_FROM", 2048, "short-input MSMs LONGER than this convert their scalars once, by a launch of their own, instead of in every workgroup (0 = always)") \
  X(small_c, "KG_SMALL_C", 0, "window width of the short-input kernel: 0 = by length, 2..10") \
  X(small_r, "KG_SMALL_R", -1, "log2 of the buckets a workgroup of the short-input kernel owns: -1 = by length, 0..7 (2^(c-1-r) workgroups per window)") \
  /* ---- MSM: sort shaping ----------------------------------------------------------------------------------------------------------- */ \
  X(sort_nch, "KG_SORT_NCH", 64, "cap on the scalar chunks (workgroups per window) of the first sort pass, 1..1024") \
  X(sort_alone, "KG_SORT_ALONE", 1, "0 = every sort is shaped for a busy device (a blocking call's first sort normally takes the whole chip)") \
  X(gs_tile, "KG_GS_TILE", -1, "entries per tile of the first sort pass: -1 = 4096 (8192 for the wide windows), 0 = the 1024-entry kernel (needs -DKG_EXPERIMENTS)") \
  X(gs_nt, "KG_GS_NT", 0, "threads per workgroup of the first sort pass beside an accumulation: 0 = 256 (512 for the wide windows)") \
  X(gs_nt0, "KG_GS_NT0", 1024, "threads per workgroup of the first sort pass on an idle device") \
  /* ---- MSM: window groups of a blocking call --------------------------------------------------------------------------------------- */ \
  X(group_main_first, "KG_GROUP_MAIN_FIRST", 0, "1 / 2 = conversion and first group's sort on the main queue (measured slower; off)") \
  X(group_accq, "KG_GROUP_ACCQ", 2, "accumulation queues the window groups rotate over") \
  X(group_reduce_inline, "KG_GROUP_REDUCE_INLINE", 1, "the last group's bucket reduction follows its accumulation on the same queue") \
  X(blocking_reduce_inline, "KG_BLOCKING_REDUCE_INLINE", 1, "an unsplit blocking MSM (8193 .. 2^17 pairs, window tables) runs its bucket reduction behind the accumulation on the main queue instead of handing it to a reduction queue") \
  X(group_one_side, "KG_GROUP_ONE_SIDE", 0, "1 = all groups reduce on one side queue") \
  /* ---- host-scalar entries (kg_msm_host_scalars, kg_sharded_key_commit) ------------------------------------------------------------ */ \
  X(host_slices, "KG_HOST_SLICES", 0, "index slices a host-scalar MSM is uploaded and run in: 0 = automatic (by length), 1..8") \
  X(host_first_div, "KG_HOST_FIRST_DIV", 0, "the first slice (whose upload nothing hides) is 1/div of an equal share: 0 = automatic") \
  X(blocking_tables_log, "KG_BLOCKING_TABLES_LOG", 18, "longest blocking kg_msm (log2 pairs) that goes through the window tables of registered bases; longer ones run in window groups") \
  X(host_slice_tables, "KG_HOST_SLICE_TABLES", 1, "index slices of a host-scalar MSM go through the window tables of registered bases when they have them") \
  X(host_accq, "KG_HOST_ACCQ", 2, "accumulation queues the slices of a host-scalar MSM alternate over (1 = the main queue only)") \
  /* ---- experiments compiled only with -DKG_EXPERIMENTS ----------------------------------------------------------------------------- */ \
  X(small_stamps, "KG_SMALL_STAMPS", 0, "experiment: 1 = the short-input kernel stamps its phase boundaries; the previous call's phase times go to stderr (tools/dbg/small_stamps.py)") \
  X(acc_prefetch, "KG_ACC_PREFETCH", 0, "experiment: 1 = accumulation with the next base prefetched into LDS (k_acc_tasks_q; measured level)") \
  X(acc_prefetch_log, "KG_ACC_PREFETCH_LOG", 0, "experiment: log2 of the array length from which KG_ACC_PREFETCH applies") \
  X(g2_pair_acc, "KG_G2_PAIR_ACC", 0, "experiment: 1 = G2 accumulation on lane pairs (~150 VGPRs instead of 250; measured level)") \
  /* ---- Groth16 prover -------------------------------------------------------------------------------------------------------------- */ \
  X(g16_h_early, "KG_G16_H_EARLY", 1, "order of h's transform chain in a proof: 1 = before the fused G1 accumulation, 0 = last (the order up to round 3), 2 = in front of the G2 accumulation") \
  X(g16_h_early_pipe, "KG_G16_H_EARLY_PIPE", 1, "0 = the early-h order only for blocking proofs") \
  X(g16_g2_glv_max, "KG_G16_G2_GLV_MAX", 1100, "a proof's b_g2 MSM keeps the halved scalars (KG_SMALL_GLV) up to this many witness entries") \
  X(g16_blind_early, "KG_G16_BLIND_EARLY", 1, "a proof's blinding chains (prover.rs:75-77, three 255-step host chains) start when the proof is enqueued; 0 = when its assembly starts") \
  X(g16_small_h_first, "KG_G16_SMALL_H_FIRST", 4096, "a short proof (one-launch MSMs) with more witness entries than this issues h's whole chain before the three G1 witness MSMs") \
  X(g16_h_inline, "KG_G16_H_INLINE", 1, "a blocking proof runs h's reduction behind h's accumulation on the main queue") \
  /* ---- NTT ------------------------------------------------------------------------------------------------------------------------- */ \
  X(ntt_direct_max_log, "KG_NTT_DIRECT_MAX_LOG", 22, "largest log2 of a direct inter-step twiddle table, 0..22 (beyond it, and when the allocation fails: composed twiddles)") \
  X(ntt_steps, "KG_NTT_STEPS", 0, "force the number of steps (HBM round trips) of a transform, 1..3; 0 = by size") \
  X(ntt_tile, "KG_NTT_TILE", 0, "force log2 of the elements a workgroup holds in LDS; 0 = by size") \
  /* ---- device memory ---------------------------------------------------------------------------------------------------------------- */ \
  X(pool_mb, "KG_POOL_MB", 1024, "MiB of released kg_malloc blocks a context keeps for the next request of the same size (0 = none: every kg_free is a hipFree): the per-call upload buffers of the hosts (a 2^24-scalar vector is 512 MiB); kept blocks are given back when ANY context of the device runs out, and by kg_ctx_trim") \
  X(pool_max_threads, "KG_POOL_MAX_THREADS", 64, "most host worker threads a context starts (host finishes, proof assemblies, uploaders; started on demand, kept until the context is destroyed); a task that needs one more fails with a status -- 0 makes every such call fail (tests)") \
  /* ---- queues, diagnostics --------------------------------------------------------------------------------------------------------- */ \
  X(queue_placement, "KG_QUEUE_PLACEMENT", 1, "0 = the context's queues in creation order instead of probed over the compute pipes") \
  X(trace_host, "KG_TRACE_HOST", 0, "1 = host-side timestamps of the MSM pipeline's calls on stderr") \
  X(profile_timeline, "KG_PROFILE_TIMELINE", 0, "1 = kg_profile_summary prints every phase's start / end on stderr")

struct kg_tuning {
#define KG_X(field, env, def, doc) int field = def;
  KG_TUNING_TABLE(KG_X)
#undef KG_X
  std::string msm_groups_list;      // KG_MSM_GROUPS given as "a,b,c"
  std::string stream_pad;           // KG_STREAM_PAD=n0,n1,...: never-used streams in front of the context's i-th queue (experiment)
};

namespace kg {
const kg_tuning& tuning();          // the process-wide values: the table's defaults overridden by the environment, read once
}
