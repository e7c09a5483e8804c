// scatter_rate.hip -- cost of 4-byte scattered stores on gfx950 as a function of the footprint the
// scattered addresses fall in.  N = 2^24 stores per launch, 256 workgroups x 1024 threads (the shape of k_scatter).
//   footprint F (bytes): element i of workgroup g is stored at  region(g) + hash(i) mod (F/4)   where region(g) is
//   private to the workgroup (F <= 256 KiB), or shared by the workgroups of one XCD (g mod 8), or global.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: coalesced; 1: random inside a per-workgroup region of `span` words; 2: random inside a per-XCD region; 3: global
__global__ void __launch_bounds__(1024) k_store(uint32_t* out, size_t per_block, int mode, uint32_t span, size_t total_words) {
  const size_t g = blockIdx.x;
  const size_t lo = g * per_block;
  for (size_t k = threadIdx.x; k < per_block; k += blockDim.x) {
    const size_t i = lo + k;
    size_t pos;
    if (mode == 0) pos = i;
    else if (mode == 1) pos = g * (size_t)span + mix((uint32_t)i) % span;
    else if (mode == 2) pos = (g & 7) * (size_t)span + mix((uint32_t)i) % span;
    else pos = mix((uint32_t)i) % total_words;
    out[pos] = (uint32_t)i;
  }
}
// as k_scatter: position from an LDS cursor (atomic with return), bins = span
__global__ void __launch_bounds__(1024) k_store_lds(uint32_t* out, size_t per_block, uint32_t bins, uint32_t run, size_t region_words) {
  extern __shared__ uint32_t cur[];
  const size_t g = blockIdx.x;
  for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) cur[b] = b * run;
  __syncthreads();
  const size_t lo = g * per_block;
  uint32_t* dst = out + (g & 7) * region_words;
  for (size_t k = threadIdx.x; k < per_block; k += blockDim.x) {
    const size_t i = lo + k;
    uint32_t b = mix((uint32_t)i) % bins;
    uint32_t pos = atomicAdd(&cur[b], 1u);
    dst[(pos + (g >> 3) * (run / 32)) % region_words] = (uint32_t)i;
  }
}

int main() {
  const size_t N = 1u << 24;
  const int blocks = 256;
  const size_t per_block = N / blocks;
  uint32_t* out;
  hipMalloc(&out, N * 4 * 2);
  hipMemset(out, 0, N * 4 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f us   %6.1f G stores/s\n", name, ms * 100.f, N / (ms / 10 * 1e-3) / 1e9);
  };
  run("coalesced", [&] { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(1024), 0, 0, out, per_block, 0, 0u, N); });
  for (uint32_t kb : {16u, 64u, 256u}) {
    char nm[96]; snprintf(nm, sizeof nm, "random inside a per-workgroup region of %u KiB", kb);
    run(nm, [&] { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(1024), 0, 0, out, per_block, 1, kb * 256u, N); });
  }
  for (uint32_t mb : {1u, 2u, 4u, 8u}) {
    char nm[96]; snprintf(nm, sizeof nm, "random inside a per-XCD region of %u MiB", mb);
    run(nm, [&] { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(1024), 0, 0, out, per_block, 2, mb * 262144u, N); });
  }
  run("random over 64 MiB", [&] { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(1024), 0, 0, out, per_block, 3, 0u, N); });
  hipFuncSetAttribute((const void*)k_store_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  run("LDS cursors, 32768 bins (k_scatter shape), 8 MiB/XCD", [&] { hipLaunchKernelGGL(k_store_lds, dim3(blocks), dim3(1024), 128 * 1024, 0, out, per_block, 32768u, 64u, (size_t)2097152); });
  run("LDS cursors, 256 bins x 8 KiB runs, 8 MiB/XCD", [&] { hipLaunchKernelGGL(k_store_lds, dim3(blocks), dim3(1024), 1024, 0, out, per_block, 256u, 8192u, (size_t)2097152); });
  run("LDS cursors, 128 bins x 128 B runs (16 KiB/block)", [&] { hipLaunchKernelGGL(k_store_lds, dim3(blocks), dim3(1024), 512, 0, out, per_block, 128u, 32u, (size_t)2097152); });
  return 0;
}
