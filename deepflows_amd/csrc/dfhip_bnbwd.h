// BatchNorm2d backward without a launch of its own (internal; the C ABI side is dfhip_bn_bwd_in / dfhip_bn_stats_out
// in include/dfhip.h).
//
// Reference being replaced: the autograd chain of nn/modules/batchnorm.py:30-55 (mean, biased variance, eps 1e-5),
// which for y = gamma (z - mu) r + beta, r = 1 / sqrt(var + eps), m = N*H*W elements per channel, gives
//     dz = gamma r (dy - sum(dy) / m - xhat sum(dy xhat) / m),   xhat = (z - mu) r
//        = A dy + B (z - mu) + Cc          A = gamma r,  B = -A r k2,  Cc = -A k1,
//                                          k1 = sum(dy) / m,  k2 = r sum(dy (z - mu)) / m
//     dgamma = r sum(dy (z - mu)),  dbeta = sum(dy).
// The two sums need all of dy, so a stand-alone backward kernel is a statistics pass and an apply pass behind a
// launch boundary (5-7 us each way for 0.1-25 MB at batch 256: ramp + two dependent round trips).  Here
//   * the kernel that PRODUCES dy (the input-gradient kernel of the next layer) leaves per-channel partial rows
//     (sum dy, sum dy (z - mu)) behind from its epilogue, one row per workgroup -- one extra read stream of z at the
//     positions it stores; no atomics, nothing a workgroup waits for at its END (an in-launch combination of the rows by
//     the last workgroup to arrive was built first: its write-through stores, counter round trip and re-read cost every
//     workgroup 3-5 us, 5-15 us per launch, more than the stand-alone BatchNorm kernel it replaces);
//   * the kernels that CONSUME dz (the input gradient and the weight gradient of the convolution whose output z is): every
//     workgroup of the input-gradient launch adds the rows up at its START while its operands are in flight
//     (BnTableBuilder) and forms dz on its operand loads; its first workgroup writes dgamma / dbeta and the per-channel
//     table the weight gradient reads.
// HBM bytes are the same as with the stand-alone kernel (z is read once by the producer and once by each consumer,
// dz is never written); what disappears is the launch.
#pragma once
#include "dfhip_internal.h"

namespace dfhip {

constexpr int kBnRowsMax = 512;      // rows a consumer adds up per channel (one per producer workgroup)

// the descriptors of the C ABI (include/dfhip.h) are what the kernels take
typedef ::dfhip_bn_bwd_in BnBwdIn;
typedef ::dfhip_bn_stats_out BnStatsOut;
static_assert(kBnRowsMax == DFHIP_BN_ROWS_MAX, "header and kernels agree on the row count");

// overwrite-mode input gradient family (dfhip_dgrad_ow.hip).  Return DFHIP_OK after a launch, -1 when no kernel of
// the family takes the shape (nothing launched), -2 when one does but not with the BatchNorm descriptors asked for.
// dry_run: only the test.
int run_dgrad_ow(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H, int W, int O, int k,
                 int stride, int pad, const BnBwdIn* bi, BnStatsOut* so, bool dry_run = false);
int run_dgrad_ow_pair(const float* dy, const float* w, const float* dy2, const float* w2, const float* add, float* dx,
                      int N, int C, int H, int W, int O, const BnBwdIn* bi, const BnBwdIn* bi2, BnStatsOut* so,
                      bool dry_run = false);

// argument check of a dfhip_bn_bwd_in at the C entry points (dfhip_conv.hip, dfhip_wgrad.hip)
inline int check_bn_in(const char* fn, const BnBwdIn* b) {
  if (!b) return DFHIP_OK;
  DF_CHECK_ARG(b->z && b->rows && b->mean && b->var, "%s: dfhip_bn_bwd_in with a null array", fn);
  DF_CHECK_ARG(b->nrows >= 1 && b->nrows <= kBnRowsMax && b->K >= 2 && b->K <= 3 && b->kidx >= 1 && b->kidx < b->K,
               "%s: dfhip_bn_bwd_in rows %d (1..%d), K %d, kidx %d", fn, b->nrows, kBnRowsMax, b->K, b->kidx);
  return DFHIP_OK;
}

}  // namespace dfhip

#if defined(__HIPCC__)
namespace dfhip {

// (A, B, mean, Cc) of a channel from its complete sums s1 = sum dy, s2 = sum dy (z - mean); sdyx = dgamma
__device__ __forceinline__ float4 bn_bwd_coeffs(const BnBwdIn& b, float mean, float var, float w, float s1, float s2,
                                                float& sdyx) {
  const float rsd = 1.0f / sqrtf(var + b.eps);
  sdyx = s2 * rsd;
  const float k1 = s1 * b.inv_m, k2 = sdyx * b.inv_m;
  const float A = w * rsd;
  return make_float4(A, -A * rsd * k2, mean, -A * k1);
}

__device__ __forceinline__ float bn_bwd_point(float dy, float z, const float4 t) {
  return fmaf(t.y, z - t.z, fmaf(t.x, dy, t.w));
}

// The per-channel table of a pending BatchNorm backward, built by ALL threads of a workgroup (NT of them) from the rows
// its producer left behind.  The rows are [nrows][O][K] floats, a row is O*K/4 <= NT 16-byte pieces; thread (g, c) adds
// the pieces c of rows g, g + NG, ... -- issue() requests the first 16 of them (and the channel constants), the caller
// then requests its operands, finish() adds them up (further rows, if any, in rounds of 16) --, the NG partial sums of a
// column are added through LDS in the order of g, and thread o turns (sum dy, sum dy (z - mean)) into (A, B, mean, Cc).
// A fixed order of additions: the same bits in every workgroup and every run.  tab: O float4 in LDS; part:
// kBnPartFloats floats of LDS scratch.  `first`: this workgroup also leaves the table and the layer's parameter
// gradients behind.  finish() contains workgroup barriers (every thread calls it) and ends with its threads' LDS
// stores: the caller's next barrier publishes tab.
// What this costs a 7.5-8.5 us input-gradient launch of the 16 x 16 / 8 x 8 stages (tools/bn_bwd_bench.py, same box):
// +1.0-1.3 us for the second operand stream and the transform, +1.7-1.9 us for the table (every workgroup pulls the 64 KB
// of rows through its L2, two LDS rounds).  Built and measured instead, both dropped: (a) the rows combined inside the
// PRODUCING launch by the last workgroup of a group to arrive (write-through stores, arrival counter, re-read): every
// workgroup waits 3-5 us at its end, +5-15 us per launch; (b) 16 group sums published by the first workgroups of the
// CONSUMING launch as 8-byte {value, tag} granules that every workgroup polls behind its operand loads (correct over 150
// repetitions from cleared tags, a self-computing fall-back instead of a progress assumption): the publish is a
// write-through store behind an L2 read, visible after 2.5-3 us -- +2.3-2.7 us, no better than everybody reading all rows.
// LDS scratch of a builder of NT threads: a 16-byte partial per thread, then the complete sums of the (up to 768) columns
constexpr int bn_part_floats(int NT) { return 4 * NT + 768; }
constexpr int kBnPartFloats = bn_part_floats(256);
template <int NT>
struct BnTableBuilder {
  static constexpr int OPT = (256 + NT - 1) / NT;      // channels per thread
  float4 v[16];
  float cm[OPT], cv[OPT], cw[OPT];                     // the thread's channels: mean, var, gamma
  int w4, NG, g, c;
  __device__ __forceinline__ void issue(const BnBwdIn& b, int O) {
    if (!b.rows) return;
    // (the channel constants with the first requests, not behind the caller's operands)
#pragma unroll
    for (int i = 0; i < OPT; i++) {
      const int o = min((int)threadIdx.x + i * NT, O - 1);
      cm[i] = b.mean[o]; cv[i] = b.var[o]; cw[i] = b.gamma ? b.gamma[o] : 1.0f;
    }
    w4 = (O * b.K) >> 2;
    NG = NT / w4;
    g = (int)threadIdx.x / w4;
    c = (int)threadIdx.x - g * w4;
    const float4* src = reinterpret_cast<const float4*>(b.rows) + c;
    const int gg = min(g, NG - 1);                     // (threads beyond the last group re-read it; their sums are not used)
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = src[(int64_t)min(gg + j * NG, b.nrows - 1) * w4];
  }
  __device__ __forceinline__ void finish(const BnBwdIn& b, int O, float4* tab, float* part, bool first) {
    const int tid = threadIdx.x;
    if (!b.rows) {                                     // (no pending backward on this operand: dz = dy)
      for (int o = tid; o < O; o += NT) tab[o] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
      return;
    }
    float* tot = part + 4 * NT;                        // the complete sums of the (up to 768) columns
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int gg = min(g, NG - 1);
    const float4* src = reinterpret_cast<const float4*>(b.rows) + c;
    for (int r0 = gg; r0 < b.nrows; r0 += 16 * NG) {
      if (r0 != gg) {
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = src[(int64_t)min(r0 + j * NG, b.nrows - 1) * w4];
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const float on = r0 + j * NG < b.nrows ? 1.0f : 0.0f;      // (a row past the end re-read the last one: times 0)
        acc.x = fmaf(v[j].x, on, acc.x); acc.y = fmaf(v[j].y, on, acc.y);
        acc.z = fmaf(v[j].z, on, acc.z); acc.w = fmaf(v[j].w, on, acc.w);
      }
    }
    if (g < NG) reinterpret_cast<float4*>(part)[g * w4 + c] = acc;
    __syncthreads();
    for (int col = tid; col < 4 * w4; col += NT) {
      float t = 0.0f;
      for (int k = 0; k < NG; k++) t += part[k * 4 * w4 + col];
      tot[col] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OPT; i++) {
      const int o = tid + i * NT;
      if (o < O) {
        float sdyx;
        const float s1 = tot[o * b.K];
        const float4 t = bn_bwd_coeffs(b, cm[i], cv[i], cw[i], s1, tot[o * b.K + b.kidx], sdyx);
        tab[o] = t;
        if (first) {
          if (b.table) reinterpret_cast<float4*>(b.table)[o] = t;
          if (b.dgamma) b.dgamma[o] = sdyx;
          if (b.dbeta) b.dbeta[o] = s1;
        }
      }
    }
  }
};

// sum of NV per-lane values over the 32 lanes of each half-wave, one value per (lane pair, index): after the call
// v[0] of lane l holds the sum of index  perm(l) = 8 b4 + 4 b3 + 2 b2 + b1  (b_i = bit i of l & 31) over its half.
// 16 + 8 + 4 + 2 + 1 exchanges instead of 5 x 16.
__device__ __forceinline__ void half_reduce16(float (&v)[16], int lane) {
#define DF_HALVE(CNT, BIT)                                                   \
  _Pragma("unroll") for (int i = 0; i < CNT; i++) {                          \
    const bool up = (lane & BIT) != 0;                                       \
    const float keep = up ? v[i + CNT] : v[i], send = up ? v[i] : v[i + CNT]; \
    v[i] = keep + __shfl_xor(send, BIT, 64);                                 \
  }
  DF_HALVE(8, 16) DF_HALVE(4, 8) DF_HALVE(2, 4) DF_HALVE(1, 2)
#undef DF_HALVE
  v[0] += __shfl_xor(v[0], 1, 64);
}

}  // namespace dfhip
#endif
