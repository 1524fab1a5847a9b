// Probe: issue rate of v_mfma_f32_32x32x2_f32 with one or two waves per SIMD, each wave on ONE
// accumulator chain (the forward conv loop) or on several, with and without LDS operand reads.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_pair_rate mfma_pair_rate.hip && ./mfma_pair_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, bool SEL>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float sm[64 * 64];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) sm[i] = (float)(i & 7);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++)
    for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  float av = (float)lane, bv = 1.0f;
  const bool ok = (lane & 3) != 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      float a = av, b = bv;
      if (LDS) {
        a = sm[((it * 8 + u) & 31) * 64 + lane];
        b = sm[((it * 8 + u + 32) & 63) * 64 + (lane ^ 1)];
      }
      if (SEL) b = ok ? b : 0.0f;
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < NACC; a++)
    for (int r = 0; r < 16; r++) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, bool LDS, bool SEL>
void run(const char* name, int threads, int blocks) {
  float* out; unsigned long long* cyc;
  const int iters = 200;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDS, SEL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDS, SEL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h[0]) * (threads / 64), hipMemcpyDeviceToHost);
  const double mfma_per_wave = (double)iters * 8 * NACC;
  const double per_simd = mfma_per_wave * (threads / 256);          // waves per SIMD x MFMAs per wave
  printf("%-44s threads %4d: wave0 %8llu ticks -> %.1f ticks per MFMA on its SIMD; kernel %.1f us -> %.1f ns per SIMD-MFMA\n",
         name, threads, h[0], (double)h[0] / per_simd, ms * 1e3, ms * 1e6 / per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int blocks = 256;
  run<1, false, false>("1 chain, registers", 256, blocks);
  run<1, false, false>("1 chain, registers", 512, blocks);
  run<4, false, false>("4 chains, registers", 256, blocks);
  run<4, false, false>("4 chains, registers", 512, blocks);
  run<1, true, false>("1 chain, 2 ds_read_b32 per MFMA", 256, blocks);
  run<1, true, false>("1 chain, 2 ds_read_b32 per MFMA", 512, blocks);
  run<1, true, true>("1 chain, 2 ds_read_b32 + select per MFMA", 512, blocks);
  run<1, false, true>("1 chain, select per MFMA", 512, blocks);
  return 0;
}
