// Probe: issue cost and sustained rate of global_load_lds (4 B and 16 B per lane) vs plain
// global_load + ds_write staging, one wave per SIMD, on data resident in L2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* src, unsigned long long* out, int n_instr, int stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = src + (size_t)blockIdx.x * 65536 + wave * 16384;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    for (int i = 0; i < n_instr; i++)
      __builtin_amdgcn_global_load_lds((glb_ptr)(base + (i % 64) * stride + lane), (lds_ptr)(lds + wave * 4096 + (i % 16) * 64), 4, 0, 0);
  } else if (MODE == 1) {
    for (int i = 0; i < n_instr; i++)
      __builtin_amdgcn_global_load_lds((glb_ptr)(base + (i % 16) * 256 + lane * 4), (lds_ptr)(lds + wave * 4096 + (i % 16) * 256), 16, 0, 0);
  } else {
    for (int i = 0; i < n_instr; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = base[((i + u) % 64) * stride + lane];
#pragma unroll
      for (int u = 0; u < 8; u++) lds[wave * 4096 + ((i + u) % 16) * 64 + lane] = v[u];
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

int main() {
  const int blocks = 256, n = 64;
  float* d; unsigned long long* o;
  (void)hipMalloc(&d, (size_t)blocks * 65536 * 4 + 65536); (void)hipMemset(d, 0, (size_t)blocks * 65536 * 4);
  (void)hipMalloc(&o, blocks * 4 * 2 * 8);
  std::vector<unsigned long long> h(blocks * 8);
  const char* names[3] = {"global_load_lds dword ", "global_load_lds dwordx4", "load+ds_write (dword)  "};
  for (int rep = 0; rep < 2; rep++)
    for (int m = 0; m < 3; m++) {
      if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 65536, 0, d, o, n, 64);
      if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 65536, 0, d, o, n, 64);
      if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 65536, 0, d, o, n, 64);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), o, blocks * 64, hipMemcpyDeviceToHost);
      double a = 0, b = 0;
      for (int i = 0; i < blocks * 4; i++) { a += h[2 * i]; b += h[2 * i + 1]; }
      if (rep) printf("%s: issue %.0f cycles/instr, issue+land %.0f cycles/instr (%d instr/wave, 4 waves/CU, 256 CUs)\n",
                      names[m], a / (blocks * 4) / n, b / (blocks * 4) / n, n);
    }
  return 0;
}
