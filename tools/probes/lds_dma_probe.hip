// Probe: semantics of __builtin_amdgcn_global_load_lds on gfx950 as used by the loader waves:
//  (1) per-lane source address, LDS destination = wave-uniform base + lane*4
//  (2) exec-masked lanes do not write their LDS word
//  (3) data issued before a raw s_barrier is visible to another wave after the issuing wave's
//      s_waitcnt vmcnt(0) + the barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

__global__ void probe(const float* src, const int* off, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = -1.0f;
  __syncthreads();
  if (wave == 1) {
    // two slices of 64 floats; lanes whose offset is negative are masked off
    for (int s = 0; s < 2; s++) {
      int o = off[s * 64 + lane];
      if (o >= 0)
        __builtin_amdgcn_global_load_lds((glb_ptr)(src + o), (lds_ptr)(lds + 100 + s * 64), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (wave == 0)
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = 1000.0f + i;
  std::vector<int> off(128);
  for (int i = 0; i < 128; i++) off[i] = (i % 5 == 0) ? -1 : (i * 7) % 1000;
  float *d, *o; int* doff;
  hipMalloc(&d, 4096); hipMalloc(&o, 2048); hipMalloc(&doff, 512);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipMemcpy(doff, off.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(128), 4096, 0, d, doff, o);
  std::vector<float> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; i++) {
    float want = -1.0f;
    if (i >= 100 && i < 228) { int k = i - 100; want = off[k] >= 0 ? h[off[k]] : -1.0f; }
    if (r[i] != want) { if (bad < 10) printf("mismatch at %d: got %f want %f\n", i, r[i], want); bad++; }
  }
  printf("lds_dma_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
