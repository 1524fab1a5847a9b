// Launch-floor probe: kernel duration (rocprofv3 --kernel-trace) and back-to-back event time of
// kernels that do (almost) nothing, by grid size, LDS size and bytes written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(float* p) {
  extern __shared__ float sm[];
  if (p == nullptr && threadIdx.x == 999999) sm[0] = 1.0f;
}
__global__ void k_write(float* p, long n) {
  extern __shared__ float sm[];
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i + 3 < n; i += (long)gridDim.x * blockDim.x * 4) *reinterpret_cast<float4*>(p + i) = make_float4(1, 2, 3, 4);
}
__global__ void k_read(const float* p, float* out, long n) {
  extern __shared__ float sm[];
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  float4 a = make_float4(0, 0, 0, 0);
  for (; i + 3 < n; i += (long)gridDim.x * blockDim.x * 4) {
    float4 v = *reinterpret_cast<const float4*>(p + i);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (a.x + a.y + a.z + a.w == 12345.f) out[0] = a.x;
}

template <class F>
static void timeit(const char* name, F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; i++) f();
  hipEventRecord(e0);
  for (int i = 0; i < 50; i++) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-48s %7.2f us per launch (back to back)\n", name, ms * 1e3 / 50);
}

int main() {
  float* buf; hipMalloc(&buf, 64 << 20);
  float* out; hipMalloc(&out, 4096);
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_write, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_read, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  timeit("empty 1x64 lds0", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, buf); });
  timeit("empty 256x256 lds0", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, buf); });
  timeit("empty 256x512 lds0", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, buf); });
  timeit("empty 256x256 lds147k", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 147 * 1024, 0, buf); });
  timeit("empty 256x512 lds70k", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 70 * 1024, 0, buf); });
  timeit("empty 2048x256 lds0", [&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, 0, buf); });
  for (long mb : {1, 8, 32}) {
    char nm[64];
    snprintf(nm, 64, "write %ld MB 1024x256", mb);
    timeit(nm, [&] { hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, 0, buf, mb * 262144); });
    snprintf(nm, 64, "read %ld MB 1024x256", mb);
    timeit(nm, [&] { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, 0, buf, out, mb * 262144); });
    snprintf(nm, 64, "write+read pair %ld MB", mb);
    timeit(nm, [&] {
      hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, 0, buf, mb * 262144);
      hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, 0, buf, out, mb * 262144);
    });
  }
  // the same empty kernel 100 times: back to back on a stream vs one replayed hipGraph (stream capture)
  {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, st);
      for (int r = 0; r < 10; r++)
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, buf);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %7.2f us per kernel\n", "stream, 100 empty 256x256 kernels", ms * 1e3 / 1000);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, buf);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 3; r++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int r = 0; r < 10; r++) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %7.2f us per kernel\n", "hipGraph replay, 100 empty 256x256 kernels", ms * 1e3 / 1000);
  }
  return 0;
}
