// Do two HIP streams run their kernels at the same time on this box?  (The precondition of running the weight gradients
// beside the backward chain.)  Kernel: `wgs` workgroups of 256 threads, each holding `lds` bytes of LDS (150 000: a CU of its
// own), spinning on independent MFMAs for `iters` x 4 instructions per wave.
//   A  one stream, two launches back to back            -> 2 T if a launch takes T
//   B  two streams, one launch each                     -> T if they overlap
//   C  stream 1: a CAPTURED GRAPH of 20 short kernels (64 workgroups each); stream 2: one long launch -> max if they overlap
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/two_streams.hip -o /tmp/two_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_spin(float* out, int iters, float seed) {
  extern __shared__ float lds[];
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = seed + threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.25f, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.125f, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.0625f, a3, 0, 0, 0);
  }
  if (threadIdx.x == 0 && lds) out[blockIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float* d;
  CK(hipMalloc(&d, 4096 * sizeof(float)));
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int lds = 150000;
  auto T = [&](auto fn) -> double { fn(); (void)hipDeviceSynchronize(); double t0 = now(); for (int r = 0; r < 20; r++) fn(); (void)hipDeviceSynchronize(); return (now() - t0) / 20 * 1e6; };
  for (int wgs : {64, 128}) {
    const int iters = 5000;
    double a = T([&] { hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), lds, s1, d, iters, 0.3f); hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), lds, s1, d, iters, 0.4f); return 0; });
    double b = T([&] { hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), lds, s1, d, iters, 0.3f); hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), lds, s2, d + 1024, iters, 0.4f); return 0; });
    printf("%3d workgroups x %d iters: one stream, two launches %.1f us; two streams, one launch each %.1f us\n", wgs, iters, a, b);
  }
  // C: a captured graph of 20 short kernels on s1, one long kernel on s2
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
  for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f);
  CK(hipStreamEndCapture(s1, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  double c0 = T([&] { (void)hipGraphLaunch(ge, s1); return 0; });
  double c1 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s2, d + 1024, 5000, 0.4f); return 0; });
  double c2 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s2, d + 1024, 5000, 0.4f); (void)hipGraphLaunch(ge, s1); return 0; });
  double c3 = T([&] { (void)hipGraphLaunch(ge, s1); hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s2, d + 1024, 5000, 0.4f); return 0; });
  printf("graph of 20 short kernels alone %.1f us; long kernel alone %.1f us; long kernel first then graph %.1f us; graph first then long kernel %.1f us\n", c0, c1, c2, c3);
  // D: the same 20 short kernels launched one by one (no graph) on s1 beside the long kernel on s2; and the host's launch rate
  double d0 = T([&] { for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f); return 0; });
  double d1 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s2, d + 1024, 5000, 0.4f);
                      for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f); return 0; });
  printf("20 short kernels launched one by one alone %.1f us; beside the long kernel on the other stream %.1f us\n", d0, d1);
  double t0 = now();
  for (int k = 0; k < 2000; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, d + 3000, 1, 0.3f);
  double issue = (now() - t0) / 2000 * 1e6;
  (void)hipDeviceSynchronize();
  double total = (now() - t0) / 2000 * 1e6;
  printf("2000 trivial launches on one stream: %.2f us of host time per launch, %.2f us per launch until the last one finished\n", issue, total);
  // D2: two chains of short kernels, one per stream; a chain beside a long kernel that holds no LDS; the chain FIRST
  double f0 = T([&] { for (int k = 0; k < 20; k++) { hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f);
                                                     hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s2, d + 1024, 250, 0.4f); } return 0; });
  double f1 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, s2, d + 1024, 5000, 0.4f);
                      for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f); return 0; });
  double f2 = T([&] { for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f);
                      hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s2, d + 1024, 5000, 0.4f); return 0; });
  double f3 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s2, d + 1024, 5000, 0.4f);
                      for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 1250, 0.3f); return 0; });
  printf("two chains of 20 short kernels, one per stream: %.1f us; chain beside a long kernel WITHOUT LDS %.1f us; chain first, long kernel second %.1f us; "
         "4 kernels of 75 us beside the long kernel %.1f us\n", f0, f1, f2, f3);
  // D3: the same with the second stream at another PRIORITY (another hardware queue / pipe?)
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t s3;
    CK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, hi));
    double g1 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s3, d + 1024, 5000, 0.4f);
                        for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f); return 0; });
    hipStream_t s4;
    CK(hipStreamCreateWithPriority(&s4, hipStreamNonBlocking, lo));
    double g2 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, s4, d + 1024, 5000, 0.4f);
                        for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s3, d + 2048, 250, 0.3f); return 0; });
    printf("priority range %d..%d: chain on a default stream beside a long kernel on a priority-%d stream %.1f us; chain on the priority-%d stream beside a "
           "long kernel on a priority-%d stream %.1f us\n", lo, hi, hi, g1, hi, lo, g2);
    // many streams: does some pair land on different pipes?
    hipStream_t ss[8];
    for (int i = 0; i < 8; i++) CK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
    for (int i = 1; i < 8; i++) {
      double g3 = T([&] { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), lds, ss[i], d + 1024, 5000, 0.4f);
                          for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, ss[0], d + 2048, 250, 0.3f); return 0; });
      printf("  chain on stream 0 beside a long kernel on stream %d: %.1f us\n", i, g3);
    }
  }
  // E: events between the streams (what a fork / join costs outside a graph)
  hipEvent_t ev1, ev2;
  CK(hipEventCreateWithFlags(&ev1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  double e0 = T([&] { for (int k = 0; k < 20; k++) { hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f);
                        if (k % 5 == 4) { (void)hipEventRecord(ev1, s1); (void)hipStreamWaitEvent(s2, ev1, 0);
                                          hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), lds, s2, d + 1024, 1000, 0.4f); } }
                      (void)hipEventRecord(ev2, s2); (void)hipStreamWaitEvent(s1, ev2, 0);
                      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, d + 2048, 250, 0.3f); return 0; });
  printf("20 short kernels on s1 with 4 forks to s2 (64 workgroups x 1000 iters = 55 us each) and one join + a last kernel: %.1f us\n", e0);
  return 0;
}
