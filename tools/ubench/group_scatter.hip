// What bounds the first pass of the two-pass bucket sort (msm.hip k_group_scatter / k_group_scatter_big) on inputs that no longer
// fit the L2 (2^24 scalars: 64 MiB per digit plane, 128 MiB of eight-byte entries per window)?  The kernel below is the same
// walk -- digits from the word planes, LDS histogram, scan, LDS placement, run-ordered stores -- with its parts switchable:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o group_scatter group_scatter.hip && ./group_scatter [log_n] [c]
// Digits are uniform random words, run cursors are spaced by the expected run length (with slack), so the access pattern is the
// sort's; the output is not checked (a rate probe).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int MAXG = 1024;
__device__ __forceinline__ uint32_t digit(const uint32_t* __restrict__ kt, size_t n, uint32_t i, int w, int c, bool& neg) {
  const int o = w * c, j = o >> 5, sh = o & 31;
  uint64_t v = kt[(size_t)j * n + i];
  if (j + 1 < 8 && sh + c > 32) v |= (uint64_t)kt[(size_t)(j + 1) * n + i] << 32;
  uint32_t e = (uint32_t)(v >> sh) & ((1u << c) - 1u);
  const int32_t d = (int32_t)e - (int32_t)(1u << (c - 1));
  neg = d < 0;
  return (uint32_t)(d < 0 ? -d : d);
}
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* sh, uint32_t& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  uint32_t inc = v;
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    uint32_t x = lane < nw ? sh[lane] : 0, xi = x;
    for (int d = 1; d < 16; d <<= 1) { uint32_t o = __shfl_up(xi, d); if (lane >= d) xi += o; }
    if (lane < 16) sh[16 + lane] = xi - x;
    if (lane == 15) sh[32] = xi;
  }
  __syncthreads();
  total = sh[32];
  return inc - v + sh[16 + wv];
}

// MODE bit 0: global stores; bit 1: digits from memory (else a hash of the index); bit 2: the LDS staging (else entries are stored
// straight from the second walk to cursor positions -- the unstaged scatter)
template <int NT, int TILE, int MODE, class E>
__global__ void __launch_bounds__(NT) k_gs(const uint32_t* __restrict__ kt, size_t n, int c, int FB, size_t chunk_len, int G, uint32_t run_len,
                                           uint32_t run_chunk, E* __restrict__ tmp, int w0) {
  constexpr int R = TILE / NT;
  __shared__ uint32_t cursor[MAXG], delta[MAXG], fill[MAXG], sh[40];
  extern __shared__ __align__(8) unsigned char dyn[];
  E* stage = reinterpret_cast<E*>(dyn);
  uint16_t* sg = reinterpret_cast<uint16_t*>(dyn + sizeof(E) * TILE);
  const int w = (int)blockIdx.x + w0, ch = blockIdx.y, tid = threadIdx.x;
  const int per = (G + NT - 1) / NT;
  for (int g = tid; g < G; g += NT) cursor[g] = (uint32_t)g * run_len + (uint32_t)ch * run_chunk;
  const uint32_t lo = (uint32_t)((size_t)ch * chunk_len), hi = (size_t)lo + chunk_len < n ? lo + (uint32_t)chunk_len : (uint32_t)n;
  E* dst = tmp + (size_t)blockIdx.x * (size_t)G * run_len;
  auto dig = [&](uint32_t i, bool& neg) -> uint32_t {
    if (MODE & 2) return digit(kt, n, i, w, c, neg);
    uint32_t h = (i + (uint32_t)w * 0x9e3779b9u) * 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    neg = h >> 31;
    return h & ((1u << (c - 1)) - 1u);
  };
  for (uint32_t tile = lo; tile < hi; tile += TILE) {
    for (int g = tid; g < G; g += NT) fill[g] = 0;
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < R; ++r) {
      const uint32_t i = tile + (uint32_t)r * NT + (uint32_t)tid;
      if (i < hi) { bool neg; const uint32_t m = dig(i, neg); if (m) atomicAdd(&fill[(m - 1) >> FB], 1u); }
    }
    __syncthreads();
    uint32_t v[4], vsum = 0;
    for (int j = 0; j < 4; ++j) { const int g = tid * per + j; v[j] = (j < per && g < G) ? fill[g] : 0u; vsum += v[j]; }
    uint32_t total;
    uint32_t ex = block_scan(vsum, sh, total);
    for (int j = 0; j < 4; ++j) {
      const int g = tid * per + j;
      if (j < per && g < G) { fill[g] = (MODE & 4) ? ex : cursor[g]; delta[g] = cursor[g] - ex; cursor[g] += v[j]; ex += v[j]; }
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < R; ++r) {
      const uint32_t i = tile + (uint32_t)r * NT + (uint32_t)tid;
      if (i < hi) {
        bool neg;
        const uint32_t m = dig(i, neg);
        if (m) {
          const uint32_t g = (m - 1) >> FB;
          const uint32_t p = atomicAdd(&fill[g], 1u);
          const E e = (E)(((uint64_t)((m - 1) & ((1u << FB) - 1u)) << (sizeof(E) == 8 ? 32 : 24)) | i | (neg ? 0x80000000u : 0u));
          if (MODE & 4) { stage[p] = e; sg[p] = (uint16_t)g; }
          else if (MODE & 1) dst[p] = e;
        }
      }
    }
    if (MODE & 4) {
      __syncthreads();
      if (MODE & 1) for (uint32_t p = tid; p < total; p += NT) dst[p + delta[sg[p]]] = stage[p];
    }
  }
}

template <int NT, int TILE, int MODE, class E>
static void run(const char* name, const uint32_t* kt, size_t n, int c, int FB, int nw, int nch, E* tmp) {
  const int G = 1 << (c - 1 - FB);
  const size_t chunk_len = (n + nch - 1) / nch;
  const uint32_t run_len = (uint32_t)((n / G) * 9 / 8 + 64), run_chunk = run_len / nch;
  const size_t lds = (size_t)TILE * (sizeof(E) + 2);
  auto* kern = &k_gs<NT, TILE, MODE, E>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(nw, nch), dim3(NT), lds, 0, kt, n, c, FB, chunk_len, G, run_len, run_chunk, tmp, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  printf("%-58s NT %4d tile %5d nch %4d lds %3zu KiB: %8.1f us  %6.1f G entries/s\n", name, NT, TILE, nch, (lds + 12448) >> 10, best * 1e3, nw * (double)n / best / 1e6);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 24, c = argc > 2 ? atoi(argv[2]) : 20, nw = 3;
  const int FB = c >= 19 ? 9 : 7;
  const size_t n = (size_t)1 << lg;
  const int G = 1 << (c - 1 - FB);
  uint32_t* kt; CK(hipMalloc(&kt, n * 8 * 4));
  {
    std::vector<uint32_t> h(n * 8);
    uint64_t s = 0x4B6F676172617368ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s >> 16); }
    CK(hipMemcpy(kt, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  const size_t run_len = (n / G) * 9 / 8 + 64;
  void* tmp; CK(hipMalloc(&tmp, (size_t)nw * G * run_len * 8 + (1 << 20)));
  printf("n = 2^%d, c = %d (FB %d, %d groups), %d windows per launch, %zu-byte entries\n", lg, c, FB, G, nw, FB == 9 ? (size_t)8 : (size_t)4);
#define RUNS(E)                                                                                            \
  run<256, 1024, 7, E>("full, tile 1024", kt, n, c, FB, nw, 256, (E*)tmp);                                   \
  run<256, 8192, 7, E>("full, tile 8192", kt, n, c, FB, nw, 256, (E*)tmp);                                   \
  run<256, 8192, 6, E>("no global stores", kt, n, c, FB, nw, 256, (E*)tmp);                                  \
  run<256, 8192, 5, E>("digits from a hash (no loads)", kt, n, c, FB, nw, 256, (E*)tmp);                     \
  run<256, 8192, 4, E>("LDS work only (no loads, no stores)", kt, n, c, FB, nw, 256, (E*)tmp);               \
  run<256, 8192, 3, E>("unstaged: stores straight from the second walk", kt, n, c, FB, nw, 256, (E*)tmp);    \
  run<256, 4096, 7, E>("full, tile 4096", kt, n, c, FB, nw, 256, (E*)tmp);                                   \
  run<256, 4096, 7, E>("full, tile 4096, 1024 chunks", kt, n, c, FB, nw, 1024, (E*)tmp);                      \
  run<256, 8192, 7, E>("full, tile 8192, 512 chunks", kt, n, c, FB, nw, 512, (E*)tmp);                        \
  run<512, 8192, 7, E>("full, 512 threads", kt, n, c, FB, nw, 256, (E*)tmp);                                 \
  run<1024, 8192, 7, E>("full, 1024 threads", kt, n, c, FB, nw, 256, (E*)tmp);                               \
  run<1024, 8192, 6, E>("1024 threads, no global stores", kt, n, c, FB, nw, 256, (E*)tmp);
  if (FB == 9) { RUNS(uint64_t) } else { RUNS(uint32_t) }
  return 0;
}
