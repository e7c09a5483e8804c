// cumask_probe.hip -- which physical CUs does a HIP CU-masked stream run on?  (hipExtStreamCreateWithCUMask)
// For a list of masks, launches a grid of long-running blocks on the masked stream and lets every block record
// (XCC id, SE id, SH id, CU id) from the hardware id registers; prints the set of CUs per XCC that saw work.
//   hipcc --offload-arch=gfx950 -O3 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <cstdlib>

__global__ void k_probe(uint32_t* seen, int spin) {
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    atomicOr(&seen[(xcc & 0xf) * 8 + se], 1u << (sh * 16 + cu));
  }
  // keep the CU busy so that the grid spreads over every CU the mask allows
  volatile float x = 1.0f;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (x == 12345.f) seen[0] = 0;
}

static void run(const char* label, const std::vector<uint32_t>& mask, uint32_t* d_seen) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", label); return; }
  hipMemsetAsync(d_seen, 0, 16 * 8 * 4, st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(k_probe, dim3(8192), dim3(256), 0, st, d_seen, 20000);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  uint32_t h[16 * 8];
  hipMemcpy(h, d_seen, sizeof h, hipMemcpyDeviceToHost);
  int total = 0;
  std::string per;
  for (int x = 0; x < 16; ++x) {
    int cnt = 0;
    for (int se = 0; se < 8; ++se) cnt += __builtin_popcount(h[x * 8 + se]);
    if (cnt) { char b[32]; snprintf(b, sizeof b, " x%d:%d", x, cnt); per += b; }
    total += cnt;
  }
  printf("%-28s CUs used %3d  %.2f ms |%s", label, total, ms, per.c_str());
  if (getenv("PROBE_X0")) { printf("  x0:"); for (int se = 0; se < 8; ++se) if (h[se]) printf(" se%d=%08x", se, h[se]); }
  printf("\n");
  hipStreamDestroy(st);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("device CUs: %d\n", ncu);
  uint32_t* d_seen;
  hipMalloc(&d_seen, 16 * 8 * 4);
  const size_t words = (size_t)(ncu + 31) / 32;
  auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0); for (int i = 0; i < ncu; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  run("all", mk([](int) { return true; }), d_seen);
  run("first 32", mk([](int i) { return i < 32; }), d_seen);
  run("first 64", mk([](int i) { return i < 64; }), d_seen);
  run("first 224", mk([](int i) { return i < 224; }), d_seen);
  run("last 32", mk([&](int i) { return i >= ncu - 32; }), d_seen);
  run("every 8th (i%8==0)", mk([](int i) { return i % 8 == 0; }), d_seen);
  run("not every 8th", mk([](int i) { return i % 8 != 0; }), d_seen);
  run("every 16th", mk([](int i) { return i % 16 == 0; }), d_seen);
  run("not every 16th", mk([](int i) { return i % 16 != 0; }), d_seen);
  run("i%8<7 (7 of 8)", mk([](int i) { return i % 8 < 7; }), d_seen);
  run("i/8 even", mk([](int i) { return (i / 8) % 2 == 0; }), d_seen);
  run("bit 0 only", mk([](int i) { return i == 0; }), d_seen);
  run("bit 1 only", mk([](int i) { return i == 1; }), d_seen);
  run("bit 8 only", mk([](int i) { return i == 8; }), d_seen);
  run("bits 0..7", mk([](int i) { return i < 8; }), d_seen);
  run("bits 0..15", mk([](int i) { return i < 16; }), d_seen);
  run("i%32<28", mk([](int i) { return i % 32 < 28; }), d_seen);
  run("i%32>=28", mk([](int i) { return i % 32 >= 28; }), d_seen);
  if (getenv("PROBE_X0")) {
    for (int sl = 0; sl < 32; ++sl) { char b[32]; snprintf(b, sizeof b, "xcd0 slot %d", sl); run(b, mk([&](int i) { return i == 8 * sl; }), d_seen); }
  }
  // balanced candidates: drop k slots per XCD
  run("slots 0..27 of each", mk([](int i) { return i / 8 < 28; }), d_seen);
  run("slots != 7 mod 8", mk([](int i) { return (i / 8) % 8 != 7; }), d_seen);
  run("slots != 3 mod 4", mk([](int i) { return (i / 8) % 4 != 3; }), d_seen);
  run("slots even", mk([](int i) { return (i / 8) % 2 == 0; }), d_seen);
  run("slots < 16", mk([](int i) { return (i / 8) < 16; }), d_seen);
  // one word only (the API may take fewer words than CUs)
  { std::vector<uint32_t> m(1, 0xffffffffu); run("1 word, all ones", m, d_seen); }
  { std::vector<uint32_t> m(2, 0xffffffffu); run("2 words, all ones", m, d_seen); }
  return 0;
}
