// Probe: can a wave feed v_mfma_f32_32x32x2_f32 with its B operand straight from global memory
// (L1/L2-resident activations, one 256-byte wave load per MFMA, nine taps per channel pair read ahead)
// and its A operand from LDS, one wave per SIMD?  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float* __restrict__ x, float* out, unsigned long long* cyc,
                                                 int HW, int tiles_per_wave) {
  __shared__ float Wl[C * 9 * 32];
  for (int i = threadIdx.x; i < C * 9 * 32; i += blockDim.x) Wl[i] = (float)(i & 15) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int gw = blockIdx.x * WAVES + wave;
  float tot = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int tt = 0; tt < tiles_per_wave; tt++) {
    const int tile = gw * tiles_per_wave + tt;
    const float* xb = x + (size_t)(tile / 8) * C * HW + (tile % 8) * 32 + l31 + half * HW;   // image = 8 tiles of 32 px
    int off[9];
#pragma unroll
    for (int t = 0; t < 9; t++) off[t] = ((t / 3) - 1) * 16 + (t % 3) - 1 + 17;   // stay inside the buffer
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float b[2][9];
#pragma unroll
    for (int t = 0; t < 9; t++) b[0][t] = xb[off[t]];
#pragma unroll 2
    for (int kp = 0; kp < C / 2; kp++) {
      if (kp + 1 < C / 2) {
#pragma unroll
        for (int t = 0; t < 9; t++) b[(kp + 1) & 1][t] = xb[(size_t)(kp + 1) * 2 * HW + off[t]];
      }
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const float a = Wl[((2 * kp + half) * 9 + t) * 32 + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[kp & 1][t], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) tot += acc[r];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
  if (lane == 0) cyc[gw] = t1 - t0;
}

template <int C, int WAVES>
void run(const char* name, int tiles_per_wave) {
  const int HW = 256 + 64, blocks = 256;
  const int images = blocks * WAVES * tiles_per_wave / 8 + 2;
  float* x; hipMalloc(&x, (size_t)images * C * HW * 4); hipMemset(x, 0, (size_t)images * C * HW * 4);
  float* out; hipMalloc(&out, (size_t)blocks * 64 * WAVES * 4);
  unsigned long long* cyc; hipMalloc(&cyc, (size_t)blocks * WAVES * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<C, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, x, out, cyc, HW, tiles_per_wave);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<C, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, x, out, cyc, HW, tiles_per_wave);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (double)tiles_per_wave * (C / 2) * 9;
  printf("%-34s C %3d, %d waves/WG, %d tiles/wave: wave0 %8llu cycles = %.1f per own MFMA (%.1f per SIMD slot); kernel %.1f us\n",
         name, C, WAVES, tiles_per_wave, h, (double)h / mf, (double)h / mf / (WAVES / 4), ms * 1e3);
  hipFree(x); hipFree(out); hipFree(cyc);
}

int main() {
  run<32, 4>("global B, LDS A", 2);
  run<32, 8>("global B, LDS A", 1);
  run<64, 4>("global B, LDS A", 2);
  run<64, 8>("global B, LDS A", 1);
  run<128, 4>("global B, LDS A", 1);
  return 0;
}
