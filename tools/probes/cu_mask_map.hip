// Where do the workgroups of a CU-masked stream land?  (hipExtStreamCreateWithCUMask bit i -> which XCC / SE / CU)
// Every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID; the host prints a histogram per mask.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_map.hip -o tools/probes/build_cu_mask_map
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#include <string>
__global__ __launch_bounds__(256) void k_where(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw + (a == 12345.f); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int ncu = 256, words = 8, wgs = 2048;
  uint32_t* d; CK(hipMalloc(&d, wgs * 2 * sizeof(uint32_t)));
  std::vector<uint32_t> h(wgs * 2);
  auto run = [&](const char* name, std::vector<uint32_t> mask) -> int {
    hipStream_t s;
    if (mask.empty()) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
    hipLaunchKernelGGL(k_where, dim3(wgs), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<int, std::map<int, int>> per_xcc;    // xcc -> (se, cu) -> count
    for (int i = 0; i < wgs; i++) {
      int xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
      int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_xcc[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    printf("%s:\n", name);
    for (auto& x : per_xcc) {
      int n = 0; for (auto& c : x.second) n += c.second;
      printf("  xcc %d: %4d workgroups on %2zu CUs:", x.first, n, x.second.size());
      for (auto& c : x.second) printf(" se%d.cu%d", c.first >> 8, c.first & 0xf);
      printf("\n");
    }
    return 0;
  };
  std::vector<uint32_t> lo(words, 0), hi(words, 0), ev(words, 0), q0(words, 0), x8(words, 0);
  for (int c = 0; c < ncu; c++) {
    (c < ncu / 2 ? lo : hi)[c / 32] |= 1u << (c % 32);
    if (c % 2 == 0) ev[c / 32] |= 1u << (c % 32);
    if (c < 64) q0[c / 32] |= 1u << (c % 32);
    if (c % 8 == 0) x8[c / 32] |= 1u << (c % 32);
  }
  run("no mask", {});
  run("bits 0..127", lo);
  run("bits 128..255", hi);
  run("even bits", ev);
  run("bits 0..63", q0);
  run("bits = 0 mod 8", x8);
  return 0;
}
