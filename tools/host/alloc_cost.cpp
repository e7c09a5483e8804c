// alloc_cost.cpp -- where the per-call milliseconds of a host that allocates per call go: kg_malloc / kg_free, the two copies, a fresh
// host vector (page faults).   g++ -O2 -std=c++17 -o alloc_cost tools/host/alloc_cost.cpp -Lkogarashi_amd -lkogarashi_amd -Wl,-rpath,$PWD/kogarashi_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/kogarashi_amd.h"
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
int main() {
  kg_init();
  kg_ctx* c = nullptr;
  if (kg_ctx_create(0, &c) != KG_OK) return 1;
  for (size_t mb : {1, 32, 128, 512}) {
    const size_t bytes = mb << 20;
    std::vector<char> h(bytes, 1), h2(bytes, 2);
    void* d = nullptr;
    double t_alloc = 0, t_free = 0, t_up = 0, t_down = 0, t_vec = 0;
    const int reps = 6;
    for (int i = 0; i < reps; ++i) {
      auto t0 = Clock::now(); kg_malloc(c, bytes, &d); t_alloc += ms_since(t0);
      t0 = Clock::now(); kg_memcpy_h2d(c, d, h.data(), bytes); t_up += ms_since(t0);
      t0 = Clock::now(); kg_memcpy_d2h(c, h2.data(), d, bytes); t_down += ms_since(t0);
      t0 = Clock::now(); kg_free(c, d); t_free += ms_since(t0);
      t0 = Clock::now(); { std::vector<char> fresh(bytes); fresh[bytes / 2] = 1; } t_vec += ms_since(t0);
    }
    {   // the same copies into ONE device buffer that stays allocated: is the slow part the fresh device memory?
      kg_malloc(c, bytes, &d);
      double up2 = 0, down2 = 0;
      kg_memcpy_h2d(c, d, h.data(), bytes);
      for (int i = 0; i < reps; ++i) {
        auto t0 = Clock::now(); kg_memcpy_h2d(c, d, h.data(), bytes); up2 += ms_since(t0);
        t0 = Clock::now(); kg_memcpy_d2h(c, h2.data(), d, bytes); down2 += ms_since(t0);
      }
      kg_free(c, d);
      std::printf("%4zu MiB, device buffer kept: h2d %.3f ms (%.1f GB/s)  d2h %.3f ms (%.1f GB/s)\n", mb, up2 / reps, bytes / (up2 / reps) / 1e6, down2 / reps,
                  bytes / (down2 / reps) / 1e6);
    }
    std::printf("%4zu MiB: kg_malloc %.3f ms  kg_free %.3f ms  h2d %.3f ms (%.1f GB/s)  d2h %.3f ms (%.1f GB/s)  fresh zeroed host vector %.3f ms\n", mb, t_alloc / reps,
                t_free / reps, t_up / reps, bytes / (t_up / reps) / 1e6, t_down / reps, bytes / (t_down / reps) / 1e6, t_vec / reps);
  }
  kg_ctx_destroy(c);
  return 0;
}
