// The last step of the one-pass stem backward (dfhip_stem.hip: k_stem_bwd<.., 2> leaves one slab of totals per workgroup): the
// totals over the slabs in a fixed order and dW, dgamma, dbeta, dbias from them.  It is a launch of its own
// (k_stem_combine) or -- inside a weight-gradient group, whose slab-sum launch runs right behind the stem's backward pass
// anyway -- extra workgroups of k_wgrad_group_sum (dfhip_wgrad.hip): the same body, the same bits.
#pragma once
#include "dfhip_internal.h"

namespace dfhip {

struct StemCombine {
  const float* slabs;
  const float* gamma;
  const float* var;
  float* dw;
  float* dbias;
  float* dgamma;
  float* dbeta;
  int O, KK, splits;
  float m, eps;
  int plain;             // 1: the two-pass form's last step (5x5 stems) -- slabs [splits][O][KK + 1] of finished gradients, column KK
                         // the bias gradient: plain sums in a fixed order (k_stem_slab_sum's), blocks of 32 elements
};

// host side (dfhip_wgrad.hip): hand the combine to the next grouped slab sum; false = no group is collecting (launch it)
bool stem_combine_defer(const StemCombine& c, int blocks);
// dfhip_stem.hip: the stand-alone launch
int stem_combine_launch(const StemCombine& c, int blocks);

#if defined(__HIPCC__)
// One-pass backward: totals over the workgroups' slabs (fixed order) and the weight gradient from them.  A block of 256
// threads owns eight consecutive elements (o, k) of dW; a thread sums every 32nd slab of the six totals its element needs
// (A[o][k], B[o][k], the column sum S[k] and the three per-channel sums), all loads of a round in flight together; the 32
// slab groups meet through the lanes and LDS.
__device__ __forceinline__ void stem_slab_sum_body(const float* __restrict__ slabs, float* __restrict__ dw,
                                                   float* __restrict__ db, int n, int splits, int KK, int blk) {
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int i = min(blk * 32 + e, n - 1);
  float t = 0.0f;
  for (int s0 = sg; s0 < splits; s0 += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = slabs[(int64_t)min(s0 + 8 * u, splits - 1) * n + i];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (s0 + 8 * u < splits) t += v[u];
  }
  part[sg][e] = t;
  __syncthreads();
  if (threadIdx.x < 32 && blk * 32 + e < n) {
    float r = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) r += part[k][e];
    const int ch = i / (KK + 1), col = i - ch * (KK + 1);
    if (col < KK) dw[ch * KK + col] = r;
    else if (db) db[ch] = r;
  }
}

__device__ __forceinline__ void stem_combine_body(const StemCombine& c, int blk) {
  if (c.plain) {                                   // (uniform per launch)
    stem_slab_sum_body(c.slabs, c.dw, c.dbias, c.O * (c.KK + 1), c.splits, c.KK, blk);
    return;
  }
  __shared__ float part[4][6][8];
  const int e = threadIdx.x & 7, sg = threadIdx.x >> 3;
  const int n = c.O * c.KK, SL = 2 * n + 3 * c.O + c.KK;
  const int i = min(blk * 8 + e, n - 1);
  const int o = i / c.KK, k = i - o * c.KK;
  const int idx[6] = {i, n + i, 2 * n + 3 * c.O + k, 2 * n + 3 * o, 2 * n + 3 * o + 1, 2 * n + 3 * o + 2};
  float t[6] = {0, 0, 0, 0, 0, 0};
  for (int s0 = sg; s0 < c.splits; s0 += 256) {
    float v[8][6];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const float* slab = c.slabs + (int64_t)min(s0 + 32 * u, c.splits - 1) * SL;
#pragma unroll
      for (int q = 0; q < 6; q++) v[u][q] = slab[idx[q]];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (s0 + 32 * u < c.splits) {
#pragma unroll
        for (int q = 0; q < 6; q++) t[q] += v[u][q];
      }
  }
#pragma unroll
  for (int q = 0; q < 6; q++) {
    t[q] += __shfl_xor(t[q], 8);
    t[q] += __shfl_xor(t[q], 16);
    t[q] += __shfl_xor(t[q], 32);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 8) {
#pragma unroll
    for (int q = 0; q < 6; q++) part[wave][q][e] = t[q];
  }
  __syncthreads();
  if (threadIdx.x < 8 && blk * 8 + e < n) {
    float r[6];
#pragma unroll
    for (int q = 0; q < 6; q++) r[q] = (part[0][q][e] + part[1][q][e]) + (part[2][q][e] + part[3][q][e]);
    const float A = r[0], B = r[1], S = r[2], s1 = r[3], s2 = r[4], sxh = r[5];
    const float gw = c.gamma ? c.gamma[o] : 1.0f;
    const float rsd = 1.0f / sqrtf(c.var[o] + c.eps);
    const float k1 = s1 / c.m, k2 = s2 / c.m;
    c.dw[i] = (A - k1 * S - k2 * B) * gw * rsd;
    if (k == 0) {
      if (c.dgamma) c.dgamma[o] = s2;
      if (c.dbeta) c.dbeta[o] = s1;
      if (c.dbias) c.dbias[o] = (s1 - k1 * c.m - k2 * sxh) * gw * rsd;      // sum of d loss / d z (zero up to rounding)
    }
  }
}
#endif

}  // namespace dfhip
