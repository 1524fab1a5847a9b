// Round 5 follow-up of two_streams.hip: does a CHAIN of dependent short kernels on one stream run beside a long kernel of
// another stream when (a) the number of hardware queues is raised (run this binary under GPU_MAX_HW_QUEUES=8 etc.),
// (b) the two streams own disjoint CU masks (hipExtStreamCreateWithCUMask), (c) the chain's launches carry
// hipExtAnyOrderLaunch (diagnostic only: no barrier bit, i.e. is the barrier packet what serialises the queues?).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/two_streams2.hip -o tools/probes/build_two_streams2
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_spin(float* out, int iters, float seed) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = seed + threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.25f, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.125f, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.0625f, a3, 0, 0, 0);
  }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const char* q = getenv("GPU_MAX_HW_QUEUES");
  printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(unset)");
  float* d;
  CK(hipMalloc(&d, 8192 * sizeof(float)));
  auto T = [&](auto fn) -> double { fn(); (void)hipDeviceSynchronize(); double t0 = now(); for (int r = 0; r < 20; r++) fn(); (void)hipDeviceSynchronize(); return (now() - t0) / 20 * 1e6; };
  auto chain = [&](hipStream_t s, int wgs) { for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, d + 2048, 250, 0.3f); };
  auto lng = [&](hipStream_t s, int wgs) { hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, d + 1024, 5000, 0.4f); };
  // (a) plain streams, many of them
  std::vector<hipStream_t> ss(16);
  for (auto& s : ss) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double c0 = T([&] { chain(ss[0], 64); return 0; });
  double l0 = T([&] { lng(ss[1], 128); return 0; });
  printf("chain alone %.1f us, long alone %.1f us\n", c0, l0);
  for (int i = 1; i < 16; i++) {
    double t = T([&] { lng(ss[i], 128); chain(ss[0], 64); return 0; });
    printf("  plain: chain on stream 0 beside long on stream %2d: %.1f us\n", i, t);
  }
  // (b) CU-masked streams: the chain on the CUs of XCDs 0-3, the long kernel on XCDs 4-7, and an interleaved split
  {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> lo(words, 0), hi(words, 0), ev(words, 0), od(words, 0);
    for (int c = 0; c < ncu; c++) {
      (c < ncu / 2 ? lo : hi)[c / 32] |= 1u << (c % 32);
      (c % 2 == 0 ? ev : od)[c / 32] |= 1u << (c % 32);
    }
    hipStream_t m0, m1, m2, m3;
    CK(hipExtStreamCreateWithCUMask(&m0, words, lo.data()));
    CK(hipExtStreamCreateWithCUMask(&m1, words, hi.data()));
    CK(hipExtStreamCreateWithCUMask(&m2, words, ev.data()));
    CK(hipExtStreamCreateWithCUMask(&m3, words, od.data()));
    double a = T([&] { chain(m0, 64); return 0; });
    double b = T([&] { lng(m1, 128); return 0; });
    double c = T([&] { lng(m1, 128); chain(m0, 64); return 0; });
    printf("CU masks (%d CUs) lower / upper half: chain alone %.1f, long alone %.1f, together %.1f us\n", ncu, a, b, c);
    a = T([&] { chain(m2, 64); return 0; });
    b = T([&] { lng(m3, 128); return 0; });
    c = T([&] { lng(m3, 128); chain(m2, 64); return 0; });
    printf("CU masks even / odd CUs: chain alone %.1f, long alone %.1f, together %.1f us\n", a, b, c);
    c = T([&] { lng(m1, 128); chain(ss[0], 64); return 0; });
    printf("long on a masked stream, chain on a plain stream: %.1f us\n", c);
  }
  // (c) the chain without barrier bits (NOT a correct chain: diagnostic of what serialises)
  {
    double t0 = T([&] { for (int k = 0; k < 20; k++) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, ss[0], nullptr, nullptr, hipExtAnyOrderLaunch, d + 2048, 250, 0.3f); return 0; });
    double t1 = T([&] { lng(ss[1], 128);
                        for (int k = 0; k < 20; k++) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, ss[0], nullptr, nullptr, hipExtAnyOrderLaunch, d + 2048, 250, 0.3f); return 0; });
    printf("any-order launches: 20 short kernels alone %.1f us; beside the long kernel %.1f us\n", t0, t1);
  }
  // (d) two chains, and a chain beside a chain of 4 medium kernels, on plain streams far apart
  {
    double t = T([&] { for (int k = 0; k < 20; k++) { hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, ss[0], d + 2048, 250, 0.3f);
                                                      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, ss[5], d + 1024, 250, 0.4f); } return 0; });
    printf("two chains on streams 0 and 5: %.1f us\n", t);
  }
  return 0;
}
