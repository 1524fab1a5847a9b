// Does the hardware queue a stream lands on change the cost of a dependent chain?  Eight streams; on each a captured graph of 200
// dependent kernels (256 x 256 threads, each writes a little and reads what the previous one wrote) is replayed 50 times; prints
// us per kernel per stream.  Run it several times: the spread ACROSS streams of one process against the spread across processes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

__global__ void k_step(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 1.0001f + 1.0f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// usage: queue_probe [K kernels per graph = 200] [streams = 8]   (K = 1, 3, 6, 58 ...: what ONE graph launch costs beside its kernels)
int main(int argc, char** argv) {
  const int n = 256 * 256;
  const int K = argc > 1 ? atoi(argv[1]) : 200, NS = argc > 2 ? atoi(argv[2]) : 8;
  float *a, *b;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  std::vector<hipStream_t> st(NS);
  for (int s = 0; s < NS; s++) hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
  for (int round = 0; round < 2; round++)
    for (int s = 0; s < NS; s++) {
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st[s], hipStreamCaptureModeRelaxed);
      for (int k = 0; k < K; k++) {
        hipLaunchKernelGGL(k_step, dim3(256), dim3(256), 0, st[s], (k & 1) ? b : a, (k & 1) ? a : b, n);
      }
      hipStreamEndCapture(st[s], &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int i = 0; i < 5; i++) hipGraphLaunch(ge, st[s]);
      hipStreamSynchronize(st[s]);
      double t0 = now();
      const int reps = K >= 100 ? 50 : 2000;
      for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st[s]);
      hipStreamSynchronize(st[s]);
      double t1 = now();
      printf("round %d stream %d: %.3f us per dependent kernel, %.3f us per replay of the %d-kernel graph\n", round, s,
             (t1 - t0) / ((double)reps * K) * 1e6, (t1 - t0) / reps * 1e6, K);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  return 0;
}
