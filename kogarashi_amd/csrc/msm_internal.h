// msm_internal.h -- what the MSM's translation units share beyond common.h: msm.hip (kernels and their enqueue functions) and
// msm_host.cpp (the C ABI's orchestration and the host finish).
#pragma once
#include "common.h"

namespace kg {

// automatic window width c for n pairs (forced != 0: that width); the reference's rule is groth16/src/msm.rs:7-14
int pick_window(size_t n, int forced);
// resident form of a base array of n points / of window tables: true = 64-byte points (see msm.hip BaseIO::load_point64)
bool resident_fmt64(size_t n);
bool table_fmt64();
// bases: ABI affine (x | y) + flags -> resident form, on queue st (k_prep_bases)
void prep_bases_enqueue(int curve, hipStream_t st, const uint64_t* d_bases, const uint8_t* d_inf, size_t n, uint32_t* out, bool fmt64);
// window tables: next[i] = 2^c * prev[i], both resident (k_table_next)
void table_next_enqueue(int curve, hipStream_t st, const uint32_t* prev, size_t n, int c, uint32_t* next, bool fmt64);
// debugging aid (KG_TRACE_HOST=1): host-side timestamps of the pipeline's calls
void host_trace(const char* what);

}  // namespace kg
