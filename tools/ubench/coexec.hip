// coexec.hip -- can a small-footprint kernel run BESIDE a chip-filling VALU kernel launched earlier on another stream?
// A: many single-wave workgroups, VREGS live registers each (register-limited occupancy), pure VALU.
// B: 256- or 1024-thread workgroups with few registers, memory-bound (strided read-modify-write), on a second stream.
// Prints B's duration alone and when enqueued right after A.
//   hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <functional>
#include <chrono>
#include <cstdlib>

template <int VREGS>
__global__ void __launch_bounds__(64) k_valu(uint32_t* out, int iters) {
  uint32_t r[VREGS];
#pragma unroll
  for (int i = 0; i < VREGS; ++i) r[i] = threadIdx.x * 2654435761u + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < VREGS; ++i) r[i] = r[i] * 1664525u + (uint32_t)(i * 7919 + it);
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < VREGS; ++i) s ^= r[i];
  if (s == 0x12345678u) out[0] = s;
}

template <int NT>
__global__ void __launch_bounds__(NT) k_mem(uint32_t* buf, size_t n, int rounds) {
  __shared__ uint32_t h[256];
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  __syncthreads();
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
      const uint32_t v = buf[i];
      atomicAdd(&h[v & 255], 1u);
      buf[i] = v + 1;
    }
  __syncthreads();
  if (threadIdx.x < 256 && h[threadIdx.x] == 0xffffffffu) buf[0] = 0;
}

// B2: register-heavier (BREGS live values) + dynamic LDS, like the scalar-preparation kernel
template <int NT, int BREGS>
__global__ void __launch_bounds__(NT) k_mem2(uint32_t* buf, size_t n, int rounds) {
  extern __shared__ uint32_t hd[];
  if (rounds < 0) { __builtin_amdgcn_s_setprio(3); rounds = -rounds; }
  for (int t = threadIdx.x; t < 4096; t += NT) hd[t] = 0;
  __syncthreads();
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
      uint32_t v[BREGS];
      v[0] = buf[i];
#pragma unroll
      for (int k = 1; k < BREGS; ++k) v[k] = v[k - 1] * 1664525u + 1013904223u;
      uint32_t x = 0;
#pragma unroll
      for (int k = 0; k < BREGS; ++k) x += v[k] * v[(k * 7 + 3) % BREGS];
      atomicAdd(&hd[x & 4095], 1u);
      buf[i] = x;
    }
  __syncthreads();
  if (hd[threadIdx.x] == 0xffffffffu) buf[0] = 0;
}

static float timed(hipStream_t st, const std::function<void()>& f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  f();
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int VREGS, int NT>
static void scenario(const char* label, hipStream_t sa, hipStream_t sb, uint32_t* out, uint32_t* buf, size_t n) {
  const int a_blocks = 64 * 1024, a_iters = 120;
  const int b_blocks = 1024 * 256 / NT;
  auto launchA = [&] { hipLaunchKernelGGL(k_valu<VREGS>, dim3(a_blocks), dim3(64), 0, sa, out, a_iters); };
  auto launchB = [&] { hipLaunchKernelGGL(k_mem<NT>, dim3(b_blocks), dim3(NT), 0, sb, buf, n, 2); };
  launchA(); launchB(); hipDeviceSynchronize();
  const float a_alone = timed(sa, launchA);
  hipDeviceSynchronize();
  const float b_alone = timed(sb, launchB);
  hipDeviceSynchronize();
  // A first, then B on the other stream while A runs
  hipEvent_t a0, a1, b0, b1;
  hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
  hipEventRecord(a0, sa); launchA(); hipEventRecord(a1, sa);
  hipEventRecord(b0, sb); launchB(); hipEventRecord(b1, sb);
  hipDeviceSynchronize();
  float a_co, b_co, b_start, b_end;
  hipEventElapsedTime(&a_co, a0, a1); hipEventElapsedTime(&b_co, b0, b1);
  hipEventElapsedTime(&b_start, a0, b0); hipEventElapsedTime(&b_end, a0, b1);
  printf("%-34s A alone %.3f  B alone %.3f | together: A %.3f  B %.3f (B ran %.3f -> %.3f after A's start)\n", label, a_alone, b_alone, a_co, b_co, b_start, b_end);
}

static int g_a_blocks = 64 * 1024, g_a_iters = 120;
template <int VREGS, int NT, int BREGS>
static void scenario2(const char* label, hipStream_t sa, hipStream_t sb, uint32_t* out, uint32_t* buf, size_t n, int delay_us) {
  const int a_blocks = g_a_blocks, a_iters = g_a_iters;
  const int b_blocks = 1024 * 256 / NT;
  hipFuncAttributes fa, fb;
  hipFuncGetAttributes(&fa, (const void*)k_valu<VREGS>);
  hipFuncGetAttributes(&fb, (const void*)k_mem2<NT, BREGS>);
  auto launchA = [&] { hipLaunchKernelGGL(k_valu<VREGS>, dim3(a_blocks), dim3(64), 0, sa, out, a_iters); };
  const int b_rounds = getenv("B_PRIO") ? -2 : 2;
  auto launchB = [&] { hipLaunchKernelGGL((k_mem2<NT, BREGS>), dim3(b_blocks), dim3(NT), 16384, sb, buf, n, b_rounds); };
  launchA(); launchB(); hipDeviceSynchronize();
  const float a_alone = timed(sa, launchA);
  hipDeviceSynchronize();
  const float b_alone = timed(sb, launchB);
  hipDeviceSynchronize();
  hipEvent_t a0, a1, b0, b1;
  hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
  hipEventRecord(a0, sa); launchA(); hipEventRecord(a1, sa);
  if (delay_us) { hipStreamQuery(sa); auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < delay_us) {} }
  hipEvent_t bm;
  hipEventCreate(&bm);
  hipEventRecord(b0, sb);
  if (getenv("WITH_MEMSET")) hipMemsetAsync(out + 64, 0, 262144, sb);
  hipEventRecord(bm, sb);
  launchB(); hipEventRecord(b1, sb);
  hipDeviceSynchronize();
  float a_co, b_co, b_start, b_end, m_ms;
  hipEventElapsedTime(&a_co, a0, a1); hipEventElapsedTime(&b_co, b0, b1);
  hipEventElapsedTime(&b_start, a0, b0); hipEventElapsedTime(&b_end, a0, b1); hipEventElapsedTime(&m_ms, b0, bm);
  printf("[memset %.3f] ", m_ms);
  printf("%-30s A %d vgpr, B %d vgpr | A alone %.3f  B alone %.3f | together: A %.3f  B %.3f (B ran %.3f -> %.3f)\n", label, fa.numRegs, fb.numRegs, a_alone, b_alone,
         a_co, b_co, b_start, b_end);
}

int main() {
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  uint32_t *out, *buf;
  const size_t n = 16u << 20;
  hipMalloc(&out, 1 << 20); hipMalloc(&buf, n * 4);
  hipMemset(buf, 0, n * 4);
  scenario<104, 256>("A 110 regs | B 256 threads", sa, sb, out, buf, n);
  scenario<104, 1024>("A 110 regs | B 1024 threads", sa, sb, out, buf, n);
  scenario<56, 256>("A ~60 regs | B 256 threads", sa, sb, out, buf, n);
  scenario<56, 1024>("A ~60 regs | B 1024 threads", sa, sb, out, buf, n);
  scenario<24, 256>("A ~28 regs | B 256 threads", sa, sb, out, buf, n);
  scenario<24, 1024>("A ~28 regs | B 1024 threads", sa, sb, out, buf, n);
  scenario2<104, 256, 24>("B2 256thr 16KB lds", sa, sb, out, buf, n, 0);
  scenario2<104, 256, 24>("B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<104, 1024, 24>("B2 1024thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<104, 256, 8>("B2 256thr few regs, +300us", sa, sb, out, buf, n, 300);
  scenario2<104, 1024, 8>("B2 1024thr few regs, +300us", sa, sb, out, buf, n, 300);
  scenario2<96, 256, 24>("A 96: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<112, 256, 24>("A 112: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<50, 256, 24>("A 50: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<52, 256, 24>("A 52: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<54, 256, 24>("A 54: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<56, 256, 24>("A 56: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  scenario2<54, 1024, 8>("A 54: B2 1024thr few, +300us", sa, sb, out, buf, n, 300);
  scenario2<60, 256, 24>("A 60: B2 256thr, +300us", sa, sb, out, buf, n, 300);
  // one generation of long-lived A waves: nothing retires while B wants in
  g_a_blocks = 4096; g_a_iters = 2000;
  scenario2<56, 256, 24>("1gen A56(112v,4/simd) B 256thr", sa, sb, out, buf, n, 300);
  scenario2<56, 256, 8>("1gen A56 B 256thr few regs", sa, sb, out, buf, n, 300);
  g_a_blocks = 2048;
  scenario2<104, 256, 24>("1gen A104(208v,2/simd) B 256thr", sa, sb, out, buf, n, 300);
  g_a_blocks = 3072;
  scenario2<56, 256, 24>("A56 3/simd only, B 256thr", sa, sb, out, buf, n, 300);
  g_a_blocks = 8192; g_a_iters = 1000;
  scenario2<24, 256, 24>("1gen A24(48v) 8/simd, B 256thr", sa, sb, out, buf, n, 300);
  g_a_blocks = 64 * 1024; g_a_iters = 120;
  hipStream_t hi;
  int lo_p, hi_p;
  hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
  hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, hi_p);
  scenario<104, 256>("A 110 regs | B 256 thr, high prio", sa, hi, out, buf, n);
  scenario<104, 1024>("A 110 regs | B 1024 thr, high prio", sa, hi, out, buf, n);
  return 0;
}
