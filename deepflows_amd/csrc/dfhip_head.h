// The cross-image sums of the classifier head's backward pass (dfhip_head_ce_forward in dfhip_fused.hip computes everything
// of the head that is per image -- pooling, Linear, cross-entropy, d loss / d logits, the input gradient -- in ONE launch
// of the forward pass): dW (C, P) = pooled^T . dlogits (tensor.py:716) and db (P) = column sums of dlogits
// (tensor.py:462-483).  They are a launch of their own (k_head_grads) or -- inside a weight-gradient group, whose slab-sum
// launch ends the backward sweep anyway -- extra workgroups of k_wgrad_group_sum (dfhip_wgrad.hip): the same body, the same
// bits, no launch.
#pragma once
#include "dfhip_internal.h"

namespace dfhip {

struct HeadGrads {
  const float* dl;       // (N, Pn) d loss / d logits
  const float* pooled;   // (N, C)
  float* dw;             // (C, Pn) or nullptr
  float* db;             // (Pn) or nullptr
  const float* rowloss;  // (N) row losses dfhip_head_ce_forward left behind, and
  float* loss;           // (1) their mean (dfhip_softmax_ce_forward's order of additions), or nullptr
  float inv_rows;
  int N, C, Pn;
};
constexpr int kHeadP = 16;                         // compile-time bound of Pn
constexpr int kHeadLdsFloats = 256 * kHeadP + 3 * 64 * kHeadP;
__host__ __device__ inline int head_grads_blocks(const HeadGrads& h) { return (h.C + 63) / 64; }

// host side (dfhip_wgrad.hip): hand the sums to the slab-sum launch of the weight-gradient group that is open; false = no
// group is open (launch them)
bool head_grads_defer(const HeadGrads& h);
// dfhip_fused.hip: the stand-alone launch
int head_grads_launch(const HeadGrads& h);

#if defined(__HIPCC__)
// (dfhip_fused.hip's wave_sum, instruction for instruction: the loss below has k_ce_small's bits)
__device__ __forceinline__ float head_wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, true));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true));   // row_ror:1
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

// loss[0] = mean of the row losses in k_ce_small's order (dfhip_fused.hip): its thread r owns the rows r, r + 1024, ...; a
// wave is summed by wave_sum, the sixteen waves in order.  256 threads: wave w plays k_ce_small's waves 4 w .. 4 w + 3.
// part: 16 floats of LDS.
__device__ __forceinline__ void loss_finish_body(const float* __restrict__ rowloss, float* __restrict__ loss, int N,
                                                 float inv_rows, float* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int W16 = 4 * wave + k;
    float tot = 0.0f;
    for (int r = 64 * W16 + lane; r < N; r += 1024) tot += rowloss[r];
    tot = head_wave_sum(tot);
    if (lane == 0) part[W16] = tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s2 = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < 16; w2++) s2 += part[w2];
    loss[0] = s2 * inv_rows;
  }
}

// 256 threads; block `blk` owns the 64 channels blk * 64 + lane; its four waves take every fourth image (a row of pooled is
// read coalesced over the channels, 16 rows in flight per wave), the rows of dlogits wait in LDS; the waves meet through LDS
// in the order 0, 1, 2, 3.  Block 0 also forms db (thread j: the images in order).  A fixed order of additions: the same bits
// in every run, inside a group or not.  lds: kHeadLdsFloats floats.
__device__ __forceinline__ void head_grads_body(const HeadGrads& h, int blk, float* lds) {
  constexpr int P = kHeadP;
  if (blk == head_grads_blocks(h) - 1 && h.loss) { // (the block with the fewest channels also finishes the loss)
    loss_finish_body(h.rowloss, h.loss, h.N, h.inv_rows, lds);
    __syncthreads();
  }
  float* sdy = lds;                                // [256][P]
  float* shw = lds + 256 * P;                      // [3][64][P]
  // (the wave index as a scalar: the row offsets of the 64 images below are then scalar arithmetic, not 64 vector registers
  // of loop-invariant addresses -- the first build held 416 registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blk * 64 + lane, cc = min(c, h.C - 1);
  float acc[P];
#pragma unroll
  for (int j = 0; j < P; j++) acc[j] = 0.0f;
  float dbv = 0.0f;
  for (int n0 = 0; n0 < h.N; n0 += 256) {
    const int nb = min(256, h.N - n0);
    if (n0 > 0) __syncthreads();
    float stg[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
      const float v = h.dl[(int64_t)(n0 + min(tid, nb - 1)) * h.Pn + min(j, h.Pn - 1)];
      stg[j] = (tid < nb && j < h.Pn) ? v : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < P; j++) sdy[tid * P + j] = stg[j];
    if (nb == 256) {
      // a full chunk: the wave's 64 images in four rounds of 16 rows of pooled, one round ahead.  The round loop is a real
      // loop over a walking pointer (unrolled, the 64 row addresses became loop invariants held in registers: 416 of them in
      // the first build), and only two images' LDS reads are in flight: inside k_wgrad_group_sum this body must stay below
      // the slab sum's own registers, or the whole launch drops to one wave per SIMD
      const float* rp = h.pooled + (int64_t)(n0 + wave) * h.C + cc;
      const int64_t step = (int64_t)4 * h.C;
      float pa[16], pb[16];
#pragma unroll
      for (int u = 0; u < 16; u++) pa[u] = rp[u * step];
      __syncthreads();
#pragma unroll 1
      for (int r = 0; r < 4; r++) {
        rp += 16 * step;
        if (r < 3) {
#pragma unroll
          for (int u = 0; u < 16; u++) pb[u] = rp[u * step];
        }
        const float4* g = reinterpret_cast<const float4*>(sdy + (wave + 64 * r) * P);
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
          for (int q = 0; q < P / 4; q++) {
            const float4 gv = g[(4 * u) * (P / 4) + q];
            acc[4 * q] = fmaf(pa[u], gv.x, acc[4 * q]); acc[4 * q + 1] = fmaf(pa[u], gv.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(pa[u], gv.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(pa[u], gv.w, acc[4 * q + 3]);
          }
          if (u & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) pa[u] = pb[u];
      }
    } else {
      __syncthreads();
      for (int n = wave; n < nb; n += 4) {             // a ragged chunk: image by image
        const float pv = h.pooled[(int64_t)(n0 + n) * h.C + cc];
#pragma unroll
        for (int j = 0; j < P; j++) acc[j] = fmaf(pv, sdy[n * P + j], acc[j]);
      }
    }
    if (blk == 0 && h.db && tid < h.Pn) {
      for (int n = 0; n < nb; n++) dbv += sdy[n * P + tid];
    }
  }
  if (wave >= 1) {
#pragma unroll
    for (int j = 0; j < P; j++) shw[((wave - 1) * 64 + lane) * P + j] = acc[j];
  }
  __syncthreads();
  if (wave == 0 && c < h.C && h.dw) {
#pragma unroll
    for (int j = 0; j < P; j++) {
      float r = acc[j];
#pragma unroll
      for (int w2 = 0; w2 < 3; w2++) r += shw[(w2 * 64 + lane) * P + j];
      if (j < h.Pn) h.dw[(int64_t)c * h.Pn + j] = r;
    }
  }
  if (blk == 0 && h.db && tid < h.Pn) h.db[tid] = dbv;
}
#endif

}  // namespace dfhip
