// Fused HBM-bound hot-path ops: BatchNorm2d train/eval/backward, MaxPool2d, ReLU,
// softmax-cross-entropy, device-side un-broadcast, axis reduce, gradient accumulate,
// multi-tensor Adam / SGD, bucket pack/unpack.
//
// In the reference each of these is a chain of 5-160 primitive launches issued from Python
// (BatchNorm2d 115 fwd + 164 bwd device calls and five full-activation
// host round trips; Adam 17 kernels per parameter).  Here each is one or two launches that
// move the algorithmic bytes once: 16-byte accesses, wave64 shuffle reductions, Chan's
// parallel mean/M2 combination for the batch statistics (one pass over HBM, no E[x^2]-E[x]^2
// cancellation).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <algorithm>

#include "dfhip_internal.h"
#include "dfhip_head.h"

using namespace dfhip;

namespace {

constexpr int kThreads = 256;

inline int grid_for(int64_t work_items, int per_thread = 1) {
  int64_t blocks = df_ceil_div(work_items, (int64_t)kThreads * per_thread);
  int64_t cap = (int64_t)num_cus() * 8;
  return (int)std::max<int64_t>(1, std::min(blocks, cap));
}

// sum over the 64 lanes, the same value in every lane: four DPP rotations inside the rows of 16 lanes, then the four
// row sums through scalar registers (six dependent ds_bpermute of a shuffle tree are ~600 cycles on the critical
// path of every reduction kernel; this is ~80)
__device__ __forceinline__ float wave_sum(float v) {
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
__device__ __forceinline__ float wave_max(float v) {      // (the same value in every lane)
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true)));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// block-wide sum, result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 4 floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// Combination of convolution partials (count, mean, centred second moment per row) WITHOUT a chain of divisions: every
// row is moved to one shift s (row 0's mean, a sample of the channel: the distances are of the size of its spread),
//   sum (y - s) = n (mu - s),   sum (y - s)^2 = m2 + n (mu - s)^2,
// the three sums are plain additions (butterflies inside a wave, the four waves through LDS in a fixed order: the same
// value in every thread of every block of the channel), and mean = s + A1 / n, M2 = A2 - A1^2 / n.  Chan's pairwise
// update in a shuffle tree was two IEEE divisions per level, nine levels: ~1 us on the critical path of every
// BatchNorm forward kernel.
// a per-channel parameter (gamma / beta; nullptr = `dflt`) REQUESTED where this is called: at the top of a kernel, in
// front of its barriers -- read where it is used (behind the statistics' barriers) it is a global round trip of its
// own on the critical path of a 5 us kernel.  `any` = any valid array of C floats (the load is unconditional).
__device__ __forceinline__ float chan_param(const float* p, const float* any, int c, float dflt) {
  const float v = (p ? p : any)[c];
  return p ? v : dflt;
}
struct RowSums { float n, a1, a2; };
__device__ __forceinline__ void rows_add(RowSums& t, float nb, float mub, float m2b, float s) {
  const float d = nb > 0.0f ? mub - s : 0.0f;
  t.n += nb;
  t.a1 = fmaf(nb, d, t.a1);
  t.a2 += fmaf(nb * d, d, m2b);
}
// sw: 12 floats of LDS not in use by anyone else between the two barriers inside
__device__ __forceinline__ void rows_finish(RowSums t, float s, float (*sw)[3], float& mean, float& var) {
  t.n = wave_sum(t.n); t.a1 = wave_sum(t.a1); t.a2 = wave_sum(t.a2);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sw[threadIdx.x >> 6][0] = t.n; sw[threadIdx.x >> 6][1] = t.a1; sw[threadIdx.x >> 6][2] = t.a2; }
  __syncthreads();
  const float n = (sw[0][0] + sw[1][0]) + (sw[2][0] + sw[3][0]);
  const float a1 = (sw[0][1] + sw[1][1]) + (sw[2][1] + sw[3][1]);
  const float a2 = (sw[0][2] + sw[1][2]) + (sw[2][2] + sw[3][2]);
  const float mu_s = n > 0.0f ? a1 / n : 0.0f;
  mean = s + mu_s;
  var = n > 0.0f ? fmaxf(a2 - a1 * mu_s, 0.0f) / n : 0.0f;      // biased variance (batchnorm.py:41-42)
}

// ---------------------------------------------------------------------------------------
// BatchNorm2d.  x is (N, C, HW) with HW contiguous.  Block (c, s) owns images
// [s*ipb, (s+1)*ipb) of channel c.
// ---------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void for_each_chunk(const float* __restrict__ base, int64_t img_stride,
                                               int n_lo, int n_hi, int HW, F f) {
  // visits every element of images [n_lo, n_hi) of one channel; f(ptr, count<=4)
  if ((HW & 3) == 0) {
    int hw4 = HW >> 2;
    int total = (n_hi - n_lo) * hw4;
    for (int e = threadIdx.x; e < total; e += kThreads) {
      int img = e / hw4, h4 = e - img * hw4;
      f(base + (int64_t)(n_lo + img) * img_stride + 4 * h4, 4);
    }
  } else {
    int total = (n_hi - n_lo) * HW;
    for (int e = threadIdx.x; e < total; e += kThreads) {
      int img = e / HW, h = e - img * HW;
      f(base + (int64_t)(n_lo + img) * img_stride + h, 1);
    }
  }
}

// partial[c][s] = (count, mean, M2) of the block's slice
__global__ __launch_bounds__(kThreads) void k_bn_stats(const float* __restrict__ x,
                                                       float* __restrict__ partial, int N, int C,
                                                       int HW, int S, int ipb) {
  __shared__ float sh[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const float* base = x + (int64_t)c * HW;
  const int64_t img_stride = (int64_t)C * HW;
  float cnt = (float)max(0, n_hi - n_lo) * (float)HW;
  float sum = 0.0f;
  for_each_chunk(base, img_stride, n_lo, n_hi, HW, [&](const float* p, int w) {
    if (w == 4) {
      float4 v = *reinterpret_cast<const float4*>(p);
      sum += (v.x + v.y) + (v.z + v.w);
    } else {
      sum += p[0];
    }
  });
  sum = block_sum(sum, sh);
  float mean = cnt > 0 ? sum / cnt : 0.0f;
  float m2 = 0.0f;
  for_each_chunk(base, img_stride, n_lo, n_hi, HW, [&](const float* p, int w) {
    if (w == 4) {
      float4 v = *reinterpret_cast<const float4*>(p);
      float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
      m2 += (a * a + b * b) + (cc * cc + d * d);
    } else {
      float a = p[0] - mean;
      m2 += a * a;
    }
  });
  m2 = block_sum(m2, sh);
  if (threadIdx.x == 0) {
    float* o = partial + ((int64_t)c * S + s) * 3;
    o[0] = cnt;
    o[1] = mean;
    o[2] = m2;
  }
}

__device__ __forceinline__ void bn_combine(const float* __restrict__ partial, int c, int S,
                                           float& mean, float& var) {
  float n = 0.0f, m = 0.0f, m2 = 0.0f;
  for (int s = 0; s < S; s++) {
    const float* p = partial + ((int64_t)c * S + s) * 3;
    float nb = p[0], mb = p[1], m2b = p[2];
    if (nb > 0.0f) {
      float nn = n + nb;
      float delta = mb - m;
      m = m + delta * (nb / nn);
      m2 = m2 + m2b + delta * delta * (n * nb / nn);
      n = nn;
    }
  }
  mean = m;
  var = n > 0.0f ? m2 / n : 0.0f;  // biased variance (batchnorm.py:41-42)
}

__global__ __launch_bounds__(kThreads) void k_bn_apply(const float* __restrict__ x,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       float* __restrict__ y,
                                                       const float* __restrict__ partial,
                                                       float* __restrict__ save_mean,
                                                       float* __restrict__ save_var,
                                                       float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, int N, int C,
                                                       int HW, int S, int ipb, float eps,
                                                       float momentum) {
  const int c = blockIdx.x, s = blockIdx.y;
  float mean, var;
  bn_combine(partial, c, S, mean, var);
  if (s == 0 && threadIdx.x == 0) {
    if (save_mean) save_mean[c] = mean;
    if (save_var) save_var[c] = var;
    // running = running*(1-momentum) + batch*momentum (batchnorm.py:44-46)
    if (running_mean) running_mean[c] = running_mean[c] * (1.0f - momentum) + mean * momentum;
    if (running_var) running_var[c] = running_var[c] * (1.0f - momentum) + var * momentum;
  }
  const float sd = sqrtf(var + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const float w = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int64_t img_stride = (int64_t)C * HW;
  const float* base = x + (int64_t)c * HW;
  float* ybase = y + (int64_t)c * HW;
  for_each_chunk(base, img_stride, n_lo, n_hi, HW, [&](const float* p, int wdt) {
    float* q = ybase + (p - base);
    if (wdt == 4) {
      float4 v = *reinterpret_cast<const float4*>(p), r;
      r.x = (v.x - mean) * rsd * w + b;
      r.y = (v.y - mean) * rsd * w + b;
      r.z = (v.z - mean) * rsd * w + b;
      r.w = (v.w - mean) * rsd * w + b;
      *reinterpret_cast<float4*>(q) = r;
    } else {
      q[0] = (p[0] - mean) * rsd * w + b;
    }
  });
}

__global__ __launch_bounds__(kThreads) void k_bn_eval(const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ y,
                                                      const float* __restrict__ rmean,
                                                      const float* __restrict__ rvar, int N, int C,
                                                      int HW, int ipb, float eps) {
  const int c = blockIdx.x, s = blockIdx.y;
  const float mean = rmean[c], sd = sqrtf(rvar[c] + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const float w = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int64_t img_stride = (int64_t)C * HW;
  const float* base = x + (int64_t)c * HW;
  float* ybase = y + (int64_t)c * HW;
  for_each_chunk(base, img_stride, n_lo, n_hi, HW, [&](const float* p, int wdt) {
    float* q = ybase + (p - base);
    for (int i = 0; i < wdt; i++) q[i] = (p[i] - mean) * rsd * w + b;
  });
}

// partial[c][s] = (sum dy, sum dy*xhat)
__global__ __launch_bounds__(kThreads) void k_bn_bwd_stats(const float* __restrict__ dy,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_var,
                                                           float* __restrict__ partial, int N, int C,
                                                           int HW, int S, int ipb, float eps) {
  __shared__ float sh[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const float mean = save_mean[c], sd = sqrtf(save_var[c] + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int64_t img_stride = (int64_t)C * HW;
  const float* xb = x + (int64_t)c * HW;
  const float* gb = dy + (int64_t)c * HW;
  float s1 = 0.0f, s2 = 0.0f;
  for_each_chunk(xb, img_stride, n_lo, n_hi, HW, [&](const float* p, int wdt) {
    const float* gq = gb + (p - xb);
    if (wdt == 4) {
      float4 v = *reinterpret_cast<const float4*>(p);
      float4 gv = *reinterpret_cast<const float4*>(gq);
      s1 += (gv.x + gv.y) + (gv.z + gv.w);
      s2 += (gv.x * ((v.x - mean) * rsd) + gv.y * ((v.y - mean) * rsd)) +
            (gv.z * ((v.z - mean) * rsd) + gv.w * ((v.w - mean) * rsd));
    } else {
      s1 += gq[0];
      s2 += gq[0] * ((p[0] - mean) * rsd);
    }
  });
  s1 = block_sum(s1, sh);
  s2 = block_sum(s2, sh);
  if (threadIdx.x == 0) {
    float* o = partial + ((int64_t)c * S + s) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_var,
    const float* __restrict__ partial, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int N, int C, int HW, int S, int ipb, float eps) {
  const int c = blockIdx.x, s = blockIdx.y;
  // (the channel's constants are requested in front of the slices' sums, not behind their loop)
  const float mean = save_mean[c], var_c = save_var[c];
  const float w = chan_param(gamma, save_mean, c, 1.0f);
  float sdy = 0.0f, sdyx = 0.0f;
  for (int i = 0; i < S; i++) {
    sdy += partial[((int64_t)c * S + i) * 2];
    sdyx += partial[((int64_t)c * S + i) * 2 + 1];
  }
  if (s == 0 && threadIdx.x == 0) {
    if (dgamma) dgamma[c] = sdyx;
    if (dbeta) dbeta[c] = sdy;
  }
  if (!dx) return;
  const float sd = sqrtf(var_c + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const float m = (float)N * (float)HW;
  const float k1 = sdy / m, k2 = sdyx / m;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int64_t img_stride = (int64_t)C * HW;
  const float* xb = x + (int64_t)c * HW;
  const float* gb = dy + (int64_t)c * HW;
  float* db = dx + (int64_t)c * HW;
  for_each_chunk(xb, img_stride, n_lo, n_hi, HW, [&](const float* p, int wdt) {
    const float* gq = gb + (p - xb);
    float* dq = db + (p - xb);
    if (wdt == 4) {
      float4 v = *reinterpret_cast<const float4*>(p);
      float4 gv = *reinterpret_cast<const float4*>(gq), r;
      r.x = (gv.x - k1 - ((v.x - mean) * rsd) * k2) * w * rsd;
      r.y = (gv.y - k1 - ((v.y - mean) * rsd) * k2) * w * rsd;
      r.z = (gv.z - k1 - ((v.z - mean) * rsd) * k2) * w * rsd;
      r.w = (gv.w - k1 - ((v.w - mean) * rsd) * k2) * w * rsd;
      *reinterpret_cast<float4*>(dq) = r;
    } else {
      dq[0] = (gq[0] - k1 - ((p[0] - mean) * rsd) * k2) * w * rsd;
    }
  });
}

// ---- BatchNorm forward from the partials a convolution left behind ------------------------------
// partials[rows][C][3] = (count, mean, centred second moment) per convolution workgroup.  Every
// block of channel c combines the rows with Chan's update (a tree over 256 threads), then
// normalises its slice of images; y = (x - mean) / sqrt(var + eps) * gamma + beta (+ residual).
__global__ __launch_bounds__(kThreads) void k_bn_apply_partials(
    const float* __restrict__ x, const float* __restrict__ partials, int rows, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ residual, float* __restrict__ y,
    float* __restrict__ save_mean, float* __restrict__ save_var, float* __restrict__ running_mean,
    float* __restrict__ running_var, int N, int C, int HW, int ipb, float eps, float momentum) {
  __shared__ float sw[4][3];
  const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  // The thread's first partial row and its (up to two) 16-byte pieces of x / residual are REQUESTED before anything
  // is computed -- the row first, so that the combination below starts as soon as the rows are back (L2) while
  // the pieces of x are still on their way from HBM; the kernel is one round of memory latency long instead of two.
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int64_t img_stride = (int64_t)C * HW;
  const float* base = x + (int64_t)c * HW;
  float* ybase = y + (int64_t)c * HW;
  const float* rbase = residual ? residual + (int64_t)c * HW : nullptr;
  const int hw4 = HW >> 2;
  const bool fast = (HW & 3) == 0 && (n_hi - n_lo) * hw4 <= 2 * kThreads;
  float p0, p1, p2;
  const float shift = partials[(int64_t)c * 3 + 1];          // row 0's mean
  const float w = chan_param(gamma, partials, c, 1.0f), b = chan_param(beta, partials, c, 0.0f);
  {
    const float* p = partials + ((int64_t)min(tid, rows - 1) * C + c) * 3;      // (clamped: no load under a condition)
    p0 = p[0]; p1 = p[1]; p2 = p[2];
    if (tid >= rows) { p0 = 0.0f; p2 = 0.0f; }
  }
  __builtin_amdgcn_sched_barrier(0);
  int64_t off[2] = {0, 0};
  bool have[2] = {false, false};
  float4 xv[2], rv[2];
  if (fast) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int e = tid + u * kThreads;
      have[u] = e < (n_hi - n_lo) * hw4;
      const int ee = have[u] ? e : 0;
      const int img = ee / hw4, h4 = ee - img * hw4;
      off[u] = (int64_t)(n_lo + img) * img_stride + 4 * h4;
      xv[u] = *reinterpret_cast<const float4*>(base + off[u]);
      rv[u] = rbase ? *reinterpret_cast<const float4*>(rbase + off[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  RowSums t = {0.0f, 0.0f, 0.0f};
  rows_add(t, p0, p1, p2, shift);
  for (int r = tid + kThreads; r < rows; r += kThreads) {
    const float* p = partials + ((int64_t)r * C + c) * 3;
    rows_add(t, p[0], p[1], p[2], shift);
  }
  float mean, var;
  rows_finish(t, shift, sw, mean, var);
  if (s == 0 && tid == 0) {
    if (save_mean) save_mean[c] = mean;
    if (save_var) save_var[c] = var;
    if (running_mean) running_mean[c] = running_mean[c] * (1.0f - momentum) + mean * momentum;
    if (running_var) running_var[c] = running_var[c] * (1.0f - momentum) + var * momentum;
  }
  const float sd = sqrtf(var + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  if (fast) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!have[u]) continue;
      const float4 v = xv[u];
      float4 r;
      r.x = (v.x - mean) * rsd * w + b;
      r.y = (v.y - mean) * rsd * w + b;
      r.z = (v.z - mean) * rsd * w + b;
      r.w = (v.w - mean) * rsd * w + b;
      if (rbase) { r.x += rv[u].x; r.y += rv[u].y; r.z += rv[u].z; r.w += rv[u].w; }
      *reinterpret_cast<float4*>(ybase + off[u]) = r;
    }
    return;
  }
  for_each_chunk(base, img_stride, n_lo, n_hi, HW, [&](const float* p, int wdt) {
    const int64_t off = p - base;
    if (wdt == 4) {
      float4 v = *reinterpret_cast<const float4*>(p), r;
      r.x = (v.x - mean) * rsd * w + b;
      r.y = (v.y - mean) * rsd * w + b;
      r.z = (v.z - mean) * rsd * w + b;
      r.w = (v.w - mean) * rsd * w + b;
      if (rbase) {
        float4 t = *reinterpret_cast<const float4*>(rbase + off);
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      *reinterpret_cast<float4*>(ybase + off) = r;
    } else {
      float r = (p[0] - mean) * rsd * w + b;
      if (rbase) r += rbase[off];
      ybase[off] = r;
    }
  });
}

// y = bnA(xA) + bnB(xB): the tail of a residual block whose shortcut is itself convolution +
// BatchNorm (batch statistics of both from their convolutions' partials).  One launch instead of
// two, and bnB's output is never written and read back.
struct BnFwdSet {
  const float* x; const float* partials; int rows;
  const float* gamma; const float* beta;
  float* save_mean; float* save_var; float* running_mean; float* running_var;
  float eps, momentum;
};

__device__ __forceinline__ void bn_partials_stats(const BnFwdSet& q, int c, int C, bool writer, float (*sw)[3],
                                                  float& mean, float& rsd, float& w, float& b) {
  const int tid = threadIdx.x;
  const float shift = q.partials[(int64_t)c * 3 + 1];
  RowSums t = {0.0f, 0.0f, 0.0f};
  for (int r = tid; r < q.rows; r += kThreads) {
    const float* p = q.partials + ((int64_t)r * C + c) * 3;
    rows_add(t, p[0], p[1], p[2], shift);
  }
  float var;
  rows_finish(t, shift, sw, mean, var);
  if (writer) {
    if (q.save_mean) q.save_mean[c] = mean;
    if (q.save_var) q.save_var[c] = var;
    if (q.running_mean) q.running_mean[c] = q.running_mean[c] * (1.0f - q.momentum) + mean * q.momentum;
    if (q.running_var) q.running_var[c] = q.running_var[c] * (1.0f - q.momentum) + var * q.momentum;
  }
  rsd = 1.0f / sqrtf(var + q.eps);
  w = q.gamma ? q.gamma[c] : 1.0f;
  b = q.beta ? q.beta[c] : 0.0f;
}

__global__ __launch_bounds__(kThreads) void k_bn_apply_partials_pair(BnFwdSet A, BnFwdSet B, float* __restrict__ y,
                                                                     int N, int C, int HW, int ipb) {
  __shared__ float sw[2][4][3];
  const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const bool writer = s == 0 && tid == 0;
  // everything the block needs from memory is requested up front -- its first partial row of BOTH sets, then its
  // (up to two) 16-byte pieces of both tensors -- and the two sets are combined side by side with one barrier: the
  // kernel is one round of memory latency long (it was three: set A, set B, then the data)
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int64_t img_stride = (int64_t)C * HW;
  const float* base = A.x + (int64_t)c * HW;
  const float* bbase = B.x + (int64_t)c * HW;
  float* ybase = y + (int64_t)c * HW;
  const int hw4 = HW >> 2;
  const bool fast = (HW & 3) == 0 && (n_hi - n_lo) * hw4 <= 2 * kThreads;
  float pa[3] = {0.0f, 0.0f, 0.0f}, pb[3] = {0.0f, 0.0f, 0.0f};
  {
    const float* qa = A.partials + ((int64_t)min(tid, A.rows - 1) * C + c) * 3;
    const float* qb = B.partials + ((int64_t)min(tid, B.rows - 1) * C + c) * 3;
    const float a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
    if (tid < A.rows) { pa[0] = a0; pa[1] = a1; pa[2] = a2; }
    if (tid < B.rows) { pb[0] = b0; pb[1] = b1; pb[2] = b2; }
  }
  const float shift_a = A.partials[(int64_t)c * 3 + 1], shift_b = B.partials[(int64_t)c * 3 + 1];   // row 0's means
  const float wa = chan_param(A.gamma, A.partials, c, 1.0f), ba = chan_param(A.beta, A.partials, c, 0.0f);
  const float wb = chan_param(B.gamma, B.partials, c, 1.0f), bb = chan_param(B.beta, B.partials, c, 0.0f);
  __builtin_amdgcn_sched_barrier(0);
  int64_t off[2] = {0, 0};
  bool have[2] = {false, false};
  float4 xv[2], uv[2];
  if (fast) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int e = tid + u * kThreads;
      have[u] = e < (n_hi - n_lo) * hw4;
      const int ee = have[u] ? e : 0;
      const int img = ee / hw4, h4 = ee - img * hw4;
      off[u] = (int64_t)(n_lo + img) * img_stride + 4 * h4;
      xv[u] = *reinterpret_cast<const float4*>(base + off[u]);
      uv[u] = *reinterpret_cast<const float4*>(bbase + off[u]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  RowSums ta = {0.0f, 0.0f, 0.0f}, tb = {0.0f, 0.0f, 0.0f};
  rows_add(ta, pa[0], pa[1], pa[2], shift_a);
  rows_add(tb, pb[0], pb[1], pb[2], shift_b);
  for (int r = tid + kThreads; r < A.rows; r += kThreads) {
    const float* q = A.partials + ((int64_t)r * C + c) * 3;
    rows_add(ta, q[0], q[1], q[2], shift_a);
  }
  for (int r = tid + kThreads; r < B.rows; r += kThreads) {
    const float* q = B.partials + ((int64_t)r * C + c) * 3;
    rows_add(tb, q[0], q[1], q[2], shift_b);
  }
  // the two sets side by side: butterflies, one barrier pair, the four waves in a fixed order
  ta.n = wave_sum(ta.n); ta.a1 = wave_sum(ta.a1); ta.a2 = wave_sum(ta.a2);
  tb.n = wave_sum(tb.n); tb.a1 = wave_sum(tb.a1); tb.a2 = wave_sum(tb.a2);
  if ((tid & 63) == 0) {
    sw[0][tid >> 6][0] = ta.n; sw[0][tid >> 6][1] = ta.a1; sw[0][tid >> 6][2] = ta.a2;
    sw[1][tid >> 6][0] = tb.n; sw[1][tid >> 6][1] = tb.a1; sw[1][tid >> 6][2] = tb.a2;
  }
  __syncthreads();
  float ma, mb, vara, varb;
  {
    const float n = (sw[0][0][0] + sw[0][1][0]) + (sw[0][2][0] + sw[0][3][0]);
    const float a1 = (sw[0][0][1] + sw[0][1][1]) + (sw[0][2][1] + sw[0][3][1]);
    const float a2 = (sw[0][0][2] + sw[0][1][2]) + (sw[0][2][2] + sw[0][3][2]);
    const float mu_s = n > 0.0f ? a1 / n : 0.0f;
    ma = shift_a + mu_s;
    vara = n > 0.0f ? fmaxf(a2 - a1 * mu_s, 0.0f) / n : 0.0f;      // biased (batchnorm.py:41-42)
  }
  {
    const float n = (sw[1][0][0] + sw[1][1][0]) + (sw[1][2][0] + sw[1][3][0]);
    const float a1 = (sw[1][0][1] + sw[1][1][1]) + (sw[1][2][1] + sw[1][3][1]);
    const float a2 = (sw[1][0][2] + sw[1][1][2]) + (sw[1][2][2] + sw[1][3][2]);
    const float mu_s = n > 0.0f ? a1 / n : 0.0f;
    mb = shift_b + mu_s;
    varb = n > 0.0f ? fmaxf(a2 - a1 * mu_s, 0.0f) / n : 0.0f;
  }
  if (writer) {
    if (A.save_mean) A.save_mean[c] = ma;
    if (A.save_var) A.save_var[c] = vara;
    if (A.running_mean) A.running_mean[c] = A.running_mean[c] * (1.0f - A.momentum) + ma * A.momentum;
    if (A.running_var) A.running_var[c] = A.running_var[c] * (1.0f - A.momentum) + vara * A.momentum;
    if (B.save_mean) B.save_mean[c] = mb;
    if (B.save_var) B.save_var[c] = varb;
    if (B.running_mean) B.running_mean[c] = B.running_mean[c] * (1.0f - B.momentum) + mb * B.momentum;
    if (B.running_var) B.running_var[c] = B.running_var[c] * (1.0f - B.momentum) + varb * B.momentum;
  }
  const float ra = 1.0f / sqrtf(vara + A.eps), rb = 1.0f / sqrtf(varb + B.eps);
  if (fast) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!have[u]) continue;
      const float4 v = xv[u], q = uv[u];
      float4 r;
      r.x = ((v.x - ma) * ra * wa + ba) + ((q.x - mb) * rb * wb + bb);
      r.y = ((v.y - ma) * ra * wa + ba) + ((q.y - mb) * rb * wb + bb);
      r.z = ((v.z - ma) * ra * wa + ba) + ((q.z - mb) * rb * wb + bb);
      r.w = ((v.w - ma) * ra * wa + ba) + ((q.w - mb) * rb * wb + bb);
      *reinterpret_cast<float4*>(ybase + off[u]) = r;
    }
    return;
  }
  for_each_chunk(base, img_stride, n_lo, n_hi, HW, [&](const float* p, int wdt) {
    const int64_t off2 = p - base;
    if (wdt == 4) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      const float4 u = *reinterpret_cast<const float4*>(bbase + off2);
      float4 r;
      r.x = ((v.x - ma) * ra * wa + ba) + ((u.x - mb) * rb * wb + bb);
      r.y = ((v.y - ma) * ra * wa + ba) + ((u.y - mb) * rb * wb + bb);
      r.z = ((v.z - ma) * ra * wa + ba) + ((u.z - mb) * rb * wb + bb);
      r.w = ((v.w - ma) * ra * wa + ba) + ((u.w - mb) * rb * wb + bb);
      *reinterpret_cast<float4*>(ybase + off2) = r;
    } else {
      ybase[off2] = ((p[0] - ma) * ra * wa + ba) + ((bbase[off2] - mb) * rb * wb + bb);
    }
  });
}

// The two kernels above on 2 x 2 planes (HW == 4, ResNet layer4): a workgroup owns EIGHT neighbouring channels -- one 128-byte
// line per image -- and a slice of the images; thread (image, channel) requests 16-byte pieces that are contiguous over the
// eight channels (with a workgroup per channel every lane touched a line of its own, and three quarters of the threads had
// nothing to do).  The partial rows of the eight channels are combined by all threads (row = thread / 8 + 32 i), lane
// exchanges + LDS in a fixed order.  PAIR: y = bnA(xA) + bnB(xB); otherwise y = bn(x) (+ residual).
struct Hw4Stats { float mean, rsd, w, b; };
__device__ __forceinline__ Hw4Stats bn_hw4_stats(const BnFwdSet& q, int c, int C, bool writer, float (*sw)[8][3]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = tid & 7;
  const float shift = q.partials[(int64_t)c * 3 + 1];                 // row 0's mean
  const float w = chan_param(q.gamma, q.partials, c, 1.0f), b = chan_param(q.beta, q.partials, c, 0.0f);
  RowSums t = {0.0f, 0.0f, 0.0f};
  {
    const int r = tid >> 3;                                             // (clamped: no load under a condition)
    const float* p = q.partials + ((int64_t)min(r, q.rows - 1) * C + c) * 3;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    if (r < q.rows) rows_add(t, p0, p1, p2, shift);
  }
  for (int r = (tid >> 3) + 32; r < q.rows; r += 32) {
    const float* p = q.partials + ((int64_t)r * C + c) * 3;
    rows_add(t, p[0], p[1], p[2], shift);
  }
#pragma unroll
  for (int m = 8; m < 64; m *= 2) { t.n += __shfl_xor(t.n, m, 64); t.a1 += __shfl_xor(t.a1, m, 64); t.a2 += __shfl_xor(t.a2, m, 64); }
  if (lane < 8) { sw[wave][lane][0] = t.n; sw[wave][lane][1] = t.a1; sw[wave][lane][2] = t.a2; }
  __syncthreads();
  const float n = (sw[0][j][0] + sw[1][j][0]) + (sw[2][j][0] + sw[3][j][0]);
  const float a1 = (sw[0][j][1] + sw[1][j][1]) + (sw[2][j][1] + sw[3][j][1]);
  const float a2 = (sw[0][j][2] + sw[1][j][2]) + (sw[2][j][2] + sw[3][j][2]);
  const float mu_s = n > 0.0f ? a1 / n : 0.0f;
  const float mean = shift + mu_s;
  const float var = n > 0.0f ? fmaxf(a2 - a1 * mu_s, 0.0f) / n : 0.0f;  // biased variance (batchnorm.py:41-42)
  if (writer) {
    if (q.save_mean) q.save_mean[c] = mean;
    if (q.save_var) q.save_var[c] = var;
    if (q.running_mean) q.running_mean[c] = q.running_mean[c] * (1.0f - q.momentum) + mean * q.momentum;
    if (q.running_var) q.running_var[c] = q.running_var[c] * (1.0f - q.momentum) + var * q.momentum;
  }
  return {mean, 1.0f / sqrtf(var + q.eps), w, b};
}

template <bool PAIR>
__global__ __launch_bounds__(kThreads) void k_bn_apply_partials_hw4(BnFwdSet A, BnFwdSet B, const float* __restrict__ residual,
                                                                    float* __restrict__ y, int N, int C, int ipb) {
  __shared__ float swa[4][8][3], swb[PAIR ? 4 : 1][8][3];
  const int tid = threadIdx.x, j = tid & 7, c = blockIdx.x * 8 + j, s = blockIdx.y;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  // the thread's (up to two) pieces are requested before the rows are added up
  float4 xv[2], uv[2];
  int64_t off[2];
  bool have[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int img = n_lo + (tid >> 3) + 32 * u;
    have[u] = img < n_hi;
    off[u] = ((int64_t)(have[u] ? img : n_lo) * C + c) * 4;
    xv[u] = *reinterpret_cast<const float4*>(A.x + off[u]);
    if constexpr (PAIR) uv[u] = *reinterpret_cast<const float4*>(B.x + off[u]);
    else uv[u] = residual ? *reinterpret_cast<const float4*>(residual + off[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_sched_barrier(0);
  const bool writer = s == 0 && tid < 8;
  const Hw4Stats a = bn_hw4_stats(A, c, C, writer, swa);
  Hw4Stats b = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (PAIR) b = bn_hw4_stats(B, c, C, writer, swb);
#pragma unroll
  for (int u = 0; u < 2; u++) {
    if (!have[u]) continue;
    const float4 v = xv[u], q = uv[u];
    float4 r;
    if constexpr (PAIR) {
      r.x = ((v.x - a.mean) * a.rsd * a.w + a.b) + ((q.x - b.mean) * b.rsd * b.w + b.b);
      r.y = ((v.y - a.mean) * a.rsd * a.w + a.b) + ((q.y - b.mean) * b.rsd * b.w + b.b);
      r.z = ((v.z - a.mean) * a.rsd * a.w + a.b) + ((q.z - b.mean) * b.rsd * b.w + b.b);
      r.w = ((v.w - a.mean) * a.rsd * a.w + a.b) + ((q.w - b.mean) * b.rsd * b.w + b.b);
    } else {
      r.x = (v.x - a.mean) * a.rsd * a.w + a.b;
      r.y = (v.y - a.mean) * a.rsd * a.w + a.b;
      r.z = (v.z - a.mean) * a.rsd * a.w + a.b;
      r.w = (v.w - a.mean) * a.rsd * a.w + a.b;
      if (residual) { r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
    }
    *reinterpret_cast<float4*>(y + off[u]) = r;
  }
}

// ---- input pipeline on the device (test/ResNet_CIFAR10_cuda.py:129-148, 180-184) -----------------
// augment_batch: reflect-pad by `pad`, crop at (ys, xs), horizontal flip, optional erase rectangle,
// clip to [-1, 1] — pure data movement, bit-exact.  The per-image decisions (drawn on the host in
// the reference's RNG order) arrive as `dec[n] = {ys, xs, flip, ey, ex, eh, ew, 0}`.
__device__ __forceinline__ int reflect_index(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

__global__ __launch_bounds__(kThreads) void k_augment_batch(const float* __restrict__ in, float* __restrict__ out,
                                                            const int* __restrict__ dec, int N, int C, int H, int W,
                                                            int pad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)N * C * H * W, step = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += step) {
    const int x = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C), n = (int)(t / C);
    const int* d = dec + (int64_t)n * 8;
    const int xc = d[2] ? W - 1 - x : x;                  // column of the cropped image before the flip
    const int sy = reflect_index(y + d[0] - pad, H), sx = reflect_index(xc + d[1] - pad, W);
    float v = in[(((int64_t)n * C + c) * H + sy) * W + sx];
    if (d[5] > 0 && y >= d[3] && y < d[3] + d[5] && x >= d[4] && x < d[4] + d[6]) v = 0.0f;
    out[i] = fminf(fmaxf(v, -1.0f), 1.0f);
  }
}

// labels -> one-hot * (1 - eps) + eps / K   (the scripts' label smoothing; eps = 0: plain one-hot)
__global__ __launch_bounds__(kThreads) void k_onehot_smooth(const int* __restrict__ labels, float* __restrict__ out,
                                                            int N, int K, float on, float off) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * K) out[i] = (i % K == labels[i / K]) ? on : off;
}

// ---- BatchNorm2d -> ReLU -> MaxPool2d(2) as one forward kernel and one backward pair --------------
// (the stem of the CIFAR models: the largest activation of the network goes through three
// element-wise passes forward and three backward in the reference).  Forward reads the convolution
// output once and writes only the pooled tensor; backward recomputes bn / relu from the convolution
// output with the SAME rounded operations (no fma contraction), so the reference's equality tests
//   max-pool: every element equal to the window maximum gets the gradient (tensor.py:779-791)
//   relu:     gradient passes where output == input                        (tensor.py:872-877)
// see the values the forward produced.  Block (c, s) owns images [s*ipb, (s+1)*ipb) of channel c;
// a thread item is a 2 x 8 input patch = 4 pooled outputs.  W % 8 == 0, H % 2 == 0.
__device__ __forceinline__ float max_nan(float m, float v) { return (m != m) ? m : ((v != v) ? v : (v > m ? v : m)); }

// (mean, biased variance) of channel c from convolution partials[rows][C][3] = (count, mean, centred second moment):
// a row per thread, Chan's update by a shuffle tree inside the wave, then the four waves in order -- the same
// sequence in every block (as k_bn_apply_partials); sw: 12 floats of LDS
__device__ __forceinline__ void bn_combine_rows(const float* __restrict__ partials, int rows, int C, int c,
                                                float (*sw)[3], float& mean, float& var) {
  const int tid = threadIdx.x;
  const float shift = partials[(int64_t)c * 3 + 1];          // row 0's mean
  RowSums t = {0.0f, 0.0f, 0.0f};
  for (int r = tid; r < rows; r += kThreads) {
    const float* p = partials + ((int64_t)r * C + c) * 3;
    rows_add(t, p[0], p[1], p[2], shift);
  }
  rows_finish(t, shift, sw, mean, var);
}

// S > 0: partial = this library's [C][S][3] statistics pass;  S < 0: partial = convolution partials [-S rows][C][3]
__global__ __launch_bounds__(kThreads) void k_bn_relu_pool_fwd(
    const float* __restrict__ x, const float* __restrict__ partial, int S, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ save_mean,
    float* __restrict__ save_var, float* __restrict__ running_mean, float* __restrict__ running_var, int N,
    int C, int H, int W, int ipb, float eps, float momentum) {
  __shared__ float sw[4][3];
  const int c = blockIdx.x, s = blockIdx.y;
  const float w = chan_param(gamma, partial, c, 1.0f), b = chan_param(beta, partial, c, 0.0f);   // (requested up front)
  float mean, var;
  if (S < 0) bn_combine_rows(partial, -S, C, c, sw, mean, var);
  else bn_combine(partial, c, S, mean, var);
  if (s == 0 && threadIdx.x == 0) {
    if (save_mean) save_mean[c] = mean;
    if (save_var) save_var[c] = var;
    if (running_mean) running_mean[c] = running_mean[c] * (1.0f - momentum) + mean * momentum;
    if (running_var) running_var[c] = running_var[c] * (1.0f - momentum) + var * momentum;
  }
  const float sd = sqrtf(var + eps);
  const float rsd = 1.0f / sd;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int oh = H >> 1, ow = W >> 1, ow4 = ow >> 2;
  const int items = (n_hi - n_lo) * oh * ow4;
  for (int e = threadIdx.x; e < items; e += kThreads) {
    const int q = e % ow4;
    const int t = e / ow4;
    const int oy = t % oh, img = n_lo + t / oh;
    const float* p = x + (((int64_t)img * C + c) * H + 2 * oy) * W + 8 * q;
    const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p + W), b1 = *reinterpret_cast<const float4*>(p + W + 4);
    const float r0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float r1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float m = relu_point(bn_point(r0[2 * j], mean, rsd, w, b));
      m = max_nan(m, relu_point(bn_point(r0[2 * j + 1], mean, rsd, w, b)));
      m = max_nan(m, relu_point(bn_point(r1[2 * j], mean, rsd, w, b)));
      m = max_nan(m, relu_point(bn_point(r1[2 * j + 1], mean, rsd, w, b)));
      o[j] = m;
    }
    *reinterpret_cast<float4*>(y + (((int64_t)img * C + c) * oh + oy) * ow + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// The stem without its full-resolution tensor (dfhip_stem.hip): `sel` holds, per 2 x 2 window of the convolution
// output, the extreme that survives bn + relu + max (the maximum for gamma >= 0, the minimum for gamma < 0: both maps
// are monotone, so relu(bn(sel)) IS the pooled value, bit for bit); the statistics come from the convolution's
// partials as in k_bn_relu_pool_fwd.  `sel` has CS >= C channel planes per image.  Block (c, s) owns images [s * ipb, (s + 1) * ipb) of channel c; P = pooled
// pixels per plane (a multiple of 4).
__global__ __launch_bounds__(kThreads) void k_sel_bn_relu_fwd(
    const float* __restrict__ sel, int CS, const float* __restrict__ partial, int rows, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_var,
    float* __restrict__ running_mean, float* __restrict__ running_var, int N, int C, int P, int ipb, float eps,
    float momentum) {
  __shared__ float sw[4][3];
  const int c = blockIdx.x, s = blockIdx.y;
  const float w = chan_param(gamma, partial, c, 1.0f), b = chan_param(beta, partial, c, 0.0f);   // (requested up front)
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int p4 = P >> 2, items = (n_hi - n_lo) * p4;
  // the first vector of every thread is requested before the rows are combined
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((int)threadIdx.x < items) {
    const int img = n_lo + (int)threadIdx.x / p4, q = (int)threadIdx.x % p4;
    v0 = *reinterpret_cast<const float4*>(sel + ((int64_t)img * CS + c) * P + 4 * q);
  }
  float mean, var;
  bn_combine_rows(partial, rows, C, c, sw, mean, var);
  if (s == 0 && threadIdx.x == 0) {
    if (save_mean) save_mean[c] = mean;
    if (save_var) save_var[c] = var;
    if (running_mean) running_mean[c] = running_mean[c] * (1.0f - momentum) + mean * momentum;
    if (running_var) running_var[c] = running_var[c] * (1.0f - momentum) + var * momentum;
  }
  const float rsd = 1.0f / sqrtf(var + eps);
  for (int e = threadIdx.x; e < items; e += kThreads) {
    const int img = n_lo + e / p4, q = e % p4;
    const int64_t at = ((int64_t)img * C + c) * P + 4 * q;
    const float4 v = e == (int)threadIdx.x ? v0 : *reinterpret_cast<const float4*>(sel + ((int64_t)img * CS + c) * P + 4 * q);
    float4 r;
    r.x = relu_point(bn_point(v.x, mean, rsd, w, b));
    r.y = relu_point(bn_point(v.y, mean, rsd, w, b));
    r.z = relu_point(bn_point(v.z, mean, rsd, w, b));
    r.w = relu_point(bn_point(v.w, mean, rsd, w, b));
    *reinterpret_cast<float4*>(y + at) = r;
  }
}

// gradient that reaches the BatchNorm output at one input position

template <bool APPLY>
__global__ __launch_bounds__(kThreads) void k_bn_relu_pool_bwd(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
    const float* __restrict__ save_var, float* __restrict__ partial, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int H, int W, int S, int ipb, float eps) {
  __shared__ float sh[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const float mean = save_mean[c], sd = sqrtf(save_var[c] + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const float w = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  float k1 = 0.0f, k2 = 0.0f;
  if (APPLY) {
    float sdy = 0.0f, sdyx = 0.0f;
    for (int i = 0; i < S; i++) {
      sdy += partial[((int64_t)c * S + i) * 2];
      sdyx += partial[((int64_t)c * S + i) * 2 + 1];
    }
    if (s == 0 && threadIdx.x == 0) {
      if (dgamma) dgamma[c] = sdyx;
      if (dbeta) dbeta[c] = sdy;
    }
    if (!dx) return;
    const float m = (float)N * (float)H * (float)W;
    k1 = sdy / m;
    k2 = sdyx / m;
  }
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const int oh = H >> 1, ow = W >> 1, ow4 = ow >> 2;
  const int items = (n_hi - n_lo) * oh * ow4;
  float s1 = 0.0f, s2 = 0.0f;
  for (int e = threadIdx.x; e < items; e += kThreads) {
    const int q = e % ow4;
    const int t = e / ow4;
    const int oy = t % oh, img = n_lo + t / oh;
    const int64_t at = (((int64_t)img * C + c) * H + 2 * oy) * W + 8 * q;
    const float* p = x + at;
    const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p + W), b1 = *reinterpret_cast<const float4*>(p + W + 4);
    const int64_t ao = (((int64_t)img * C + c) * oh + oy) * ow + 4 * q;
    const float4 yv = *reinterpret_cast<const float4*>(y + ao), gv = *reinterpret_cast<const float4*>(g + ao);
    const float r0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float r1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float d0[8], d1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float dy0 = stem_dy(r0[j], yy[j >> 1], gg[j >> 1], mean, rsd, w, b);
      const float dy1 = stem_dy(r1[j], yy[j >> 1], gg[j >> 1], mean, rsd, w, b);
      const float xh0 = (r0[j] - mean) * rsd, xh1 = (r1[j] - mean) * rsd;
      if (APPLY) {
        d0[j] = (dy0 - k1 - xh0 * k2) * w * rsd;
        d1[j] = (dy1 - k1 - xh1 * k2) * w * rsd;
      } else {
        s1 += dy0 + dy1;
        s2 += dy0 * xh0 + dy1 * xh1;
      }
    }
    if (APPLY) {
      float* q0 = dx + at;
      *reinterpret_cast<float4*>(q0) = make_float4(d0[0], d0[1], d0[2], d0[3]);
      *reinterpret_cast<float4*>(q0 + 4) = make_float4(d0[4], d0[5], d0[6], d0[7]);
      *reinterpret_cast<float4*>(q0 + W) = make_float4(d1[0], d1[1], d1[2], d1[3]);
      *reinterpret_cast<float4*>(q0 + W + 4) = make_float4(d1[4], d1[5], d1[6], d1[7]);
    }
  }
  if (!APPLY) {
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
      float* o = partial + ((int64_t)c * S + s) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
}

// ---- single-launch BatchNorm for channels of at most 16 384 elements (N*HW) ------------------
// One 512-thread workgroup owns a channel and keeps its elements in registers (<= 8 float4 per
// thread): x is read from HBM once (8 B per element instead of 12), mean and the centred second
// moment are two block reductions over registers, no partials buffer, no second launch.
constexpr int kBnVec = 8;

__global__ __launch_bounds__(512) void k_bn_fused_fwd(const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ y,
                                                      float* __restrict__ save_mean,
                                                      float* __restrict__ save_var,
                                                      float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, int N, int C,
                                                      int HW, float eps, float momentum) {
  __shared__ float sh[8];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int hw4 = HW >> 2, nvec = N * hw4;
  const int64_t img_stride = (int64_t)C * HW;
  const float* base = x + (int64_t)c * HW;
  float4 v[kBnVec];
  int64_t off[kBnVec];
  float s = 0.0f;
#pragma unroll
  for (int u = 0; u < kBnVec; u++) {
    int e = tid + 512 * u;
    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    off[u] = -1;
    if (e < nvec) {
      int n = e / hw4, h4 = e - n * hw4;
      off[u] = (int64_t)n * img_stride + 4 * h4;
      v[u] = *reinterpret_cast<const float4*>(base + off[u]);
      s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
  }
  auto block_sum512 = [&](float t) {
    t = wave_sum(t);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = t;
    __syncthreads();
    return ((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]));
  };
  const float m = (float)N * (float)HW;
  const float mean = block_sum512(s) / m;
  float q = 0.0f;
#pragma unroll
  for (int u = 0; u < kBnVec; u++)
    if (off[u] >= 0) {
      float a = v[u].x - mean, b = v[u].y - mean, cc = v[u].z - mean, d = v[u].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  const float var = block_sum512(q) / m;
  if (tid == 0) {
    if (save_mean) save_mean[c] = mean;
    if (save_var) save_var[c] = var;
    if (running_mean) running_mean[c] = running_mean[c] * (1.0f - momentum) + mean * momentum;
    if (running_var) running_var[c] = running_var[c] * (1.0f - momentum) + var * momentum;
  }
  const float sd = sqrtf(var + eps);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const float w = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  float* ybase = y + (int64_t)c * HW;
#pragma unroll
  for (int u = 0; u < kBnVec; u++)
    if (off[u] >= 0) {
      float4 r;
      r.x = (v[u].x - mean) * rsd * w + b;
      r.y = (v[u].y - mean) * rsd * w + b;
      r.z = (v[u].z - mean) * rsd * w + b;
      r.w = (v[u].w - mean) * rsd * w + b;
      *reinterpret_cast<float4*>(ybase + off[u]) = r;
    }
}

__device__ __forceinline__ void bn_fused_bwd_body(int c, const float* __restrict__ dy,
                                                      const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_var,
                                                      float* __restrict__ dx, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, int N, int C, int HW,
                                                      float eps) {
  __shared__ float sh[16];
  const int tid = threadIdx.x;
  const int hw4 = HW >> 2, nvec = N * hw4;
  const int64_t img_stride = (int64_t)C * HW;
  const float mean = save_mean[c], sd = sqrtf(save_var[c] + eps);
  const float w = chan_param(gamma, save_mean, c, 1.0f);
  const float rsd = 1.0f / sd;   // one division per thread: (x - mean) / sd becomes a multiply
  const float* xb = x + (int64_t)c * HW;
  const float* gb = dy + (int64_t)c * HW;
  float4 xh[kBnVec], gv[kBnVec];
  int64_t off[kBnVec];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int u = 0; u < kBnVec; u++) {
    int e = tid + 512 * u;
    off[u] = -1;
    if (e < nvec) {
      int n = e / hw4, h4 = e - n * hw4;
      off[u] = (int64_t)n * img_stride + 4 * h4;
      float4 xv = *reinterpret_cast<const float4*>(xb + off[u]);
      gv[u] = *reinterpret_cast<const float4*>(gb + off[u]);
      xh[u].x = (xv.x - mean) * rsd; xh[u].y = (xv.y - mean) * rsd;
      xh[u].z = (xv.z - mean) * rsd; xh[u].w = (xv.w - mean) * rsd;
      s1 += (gv[u].x + gv[u].y) + (gv[u].z + gv[u].w);
      s2 += (gv[u].x * xh[u].x + gv[u].y * xh[u].y) + (gv[u].z * xh[u].z + gv[u].w * xh[u].w);
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) { sh[tid >> 6] = s1; sh[8 + (tid >> 6)] = s2; }
  __syncthreads();
  float sdy = 0.0f, sdyx = 0.0f;
#pragma unroll
  for (int w8 = 0; w8 < 8; w8++) { sdy += sh[w8]; sdyx += sh[8 + w8]; }
  if (tid == 0) {
    if (dgamma) dgamma[c] = sdyx;
    if (dbeta) dbeta[c] = sdy;
  }
  if (!dx) return;
  const float m = (float)N * (float)HW;
  const float k1 = sdy / m, k2 = sdyx / m;
  float* db = dx + (int64_t)c * HW;
#pragma unroll
  for (int u = 0; u < kBnVec; u++)
    if (off[u] >= 0) {
      float4 r;
      r.x = (gv[u].x - k1 - xh[u].x * k2) * w * rsd;
      r.y = (gv[u].y - k1 - xh[u].y * k2) * w * rsd;
      r.z = (gv[u].z - k1 - xh[u].z * k2) * w * rsd;
      r.w = (gv[u].w - k1 - xh[u].w * k2) * w * rsd;
      *reinterpret_cast<float4*>(db + off[u]) = r;
    }
}

__global__ __launch_bounds__(512) void k_bn_fused_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_var, float* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                      int C, int HW, float eps) {
  bn_fused_bwd_body(blockIdx.x, dy, x, gamma, save_mean, save_var, dx, dgamma, dbeta, N, C, HW, eps);
}

// the two BatchNorms that feed one residual sum receive the same dy: one launch, blockIdx.y picks the set
struct BnBwdSet {
  const float* x; const float* gamma; const float* save_mean; const float* save_var;
  float* dx; float* dgamma; float* dbeta; float eps;
};
__global__ __launch_bounds__(512) void k_bn_fused_bwd_pair(const float* __restrict__ dy, BnBwdSet A, BnBwdSet B, int N,
                                                           int C, int HW) {
  const BnBwdSet& q = blockIdx.y ? B : A;
  bn_fused_bwd_body(blockIdx.x, dy, q.x, q.gamma, q.save_mean, q.save_var, q.dx, q.dgamma, q.dbeta, N, C, HW, q.eps);
}

// 2 x 2 planes (HW == 4, ResNet layer4): a channel's plane is 16 bytes, so with a workgroup per channel every lane touches a
// cache line of its own (64 lines per load instruction, 8 channels' workgroups pulling the same lines).  Here a workgroup
// owns EIGHT neighbouring channels -- 128 bytes = one line per image --, thread (image, channel) requests 16-byte pieces
// that are contiguous over the eight channels, the per-channel sums meet through three lane exchanges and LDS (a fixed
// order), and the pass that applies them reuses the registers.  N <= 512 * kBn4 / 8 images.  (The same form for the 4 x 4
// planes of layer3 -- four channels = two lines per image -- was built and measured: 4.3 -> 4.4 us, a channel's 64-byte
// pieces already coalesce well enough; dropped.)
constexpr int kBn4 = 8;                             // 16-byte pieces per thread
__device__ __forceinline__ void bn_fused_bwd_hw4_body(int cg, const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_var, float* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C,
                                                      float eps) {
  __shared__ float sh[8][8][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = tid & 7, c = cg * 8 + j;            // (512 % 8 == 0: a thread keeps its channel over its pieces)
  const float mean = save_mean[c], rsd = 1.0f / sqrtf(save_var[c] + eps);
  const float w = chan_param(gamma, save_mean, c, 1.0f);
  float4 xh[kBn4], gv[kBn4];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int u = 0; u < kBn4; u++) {
    const int img = (tid >> 3) + 64 * u;
    const int64_t off = ((int64_t)min(img, N - 1) * C + c) * 4;
    const float4 xv = *reinterpret_cast<const float4*>(x + off);
    gv[u] = *reinterpret_cast<const float4*>(dy + off);
    if (img >= N) gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    xh[u].x = (xv.x - mean) * rsd; xh[u].y = (xv.y - mean) * rsd;
    xh[u].z = (xv.z - mean) * rsd; xh[u].w = (xv.w - mean) * rsd;
    s1 += (gv[u].x + gv[u].y) + (gv[u].z + gv[u].w);
    s2 += (gv[u].x * xh[u].x + gv[u].y * xh[u].y) + (gv[u].z * xh[u].z + gv[u].w * xh[u].w);
  }
#pragma unroll
  for (int m = 8; m < 64; m *= 2) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (lane < 8) { sh[wave][lane][0] = s1; sh[wave][lane][1] = s2; }
  __syncthreads();
  float sdy = 0.0f, sdyx = 0.0f;
#pragma unroll
  for (int w8 = 0; w8 < 8; w8++) { sdy += sh[w8][j][0]; sdyx += sh[w8][j][1]; }
  if (tid < 8) {
    if (dgamma) dgamma[c] = sdyx;
    if (dbeta) dbeta[c] = sdy;
  }
  if (!dx) return;
  const float m = (float)N * 4.0f;
  const float k1 = sdy / m, k2 = sdyx / m;
#pragma unroll
  for (int u = 0; u < kBn4; u++) {
    const int img = (tid >> 3) + 64 * u;
    if (img < N) {
      float4 r;
      r.x = (gv[u].x - k1 - xh[u].x * k2) * w * rsd;
      r.y = (gv[u].y - k1 - xh[u].y * k2) * w * rsd;
      r.z = (gv[u].z - k1 - xh[u].z * k2) * w * rsd;
      r.w = (gv[u].w - k1 - xh[u].w * k2) * w * rsd;
      *reinterpret_cast<float4*>(dx + ((int64_t)img * C + c) * 4) = r;
    }
  }
}
__global__ __launch_bounds__(512) void k_bn_fused_bwd_hw4(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_var, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C,
                                                          float eps) {
  bn_fused_bwd_hw4_body(blockIdx.x, dy, x, gamma, save_mean, save_var, dx, dgamma, dbeta, N, C, eps);
}
__global__ __launch_bounds__(512) void k_bn_fused_bwd_pair_hw4(const float* __restrict__ dy, BnBwdSet A, BnBwdSet B, int N, int C) {
  const BnBwdSet& q = blockIdx.y ? B : A;
  bn_fused_bwd_hw4_body(blockIdx.x, dy, q.x, q.gamma, q.save_mean, q.save_var, q.dx, q.dgamma, q.dbeta, N, C, q.eps);
}

inline bool bn_hw4_ok(int N, int C, int HW) {
  return HW == 4 && C % 8 == 0 && N <= 64 * kBn4;
}

inline bool bn_fused_ok(int N, int C, int HW) {
  return (HW & 3) == 0 && (int64_t)N * HW <= (int64_t)512 * 4 * kBnVec && C >= 32;
}

void bn_split(int N, int C, int& S, int& ipb) {
  int want = (int)df_ceil_div((int64_t)num_cus() * 4, C);
  S = std::max(1, std::min(want, N));
  ipb = (int)df_ceil_div(N, S);
  S = (int)df_ceil_div(N, ipb);
}

// ---------------------------------------------------------------------------------------
// MaxPool2d (kernel == stride), ReLU
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_maxpool_fwd(const float* __restrict__ x,
                                                          float* __restrict__ y, int64_t n_out,
                                                          int H, int W, int oh, int ow, int k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_out; i += step) {
    int ox = (int)(i % ow);
    int64_t t = i / ow;
    int oy = (int)(t % oh);
    int64_t nc = t / oh;
    const float* p = x + (nc * H + (int64_t)oy * k) * W + (int64_t)ox * k;
    float m = p[0];
    for (int a = 0; a < k; a++)
      for (int b = 0; b < k; b++) {
        float v = p[a * W + b];
        m = (m != m) ? m : ((v != v) ? v : (v > m ? v : m));
      }
    y[i] = m;
  }
}

__global__ __launch_bounds__(kThreads) void k_maxpool_bwd(const float* __restrict__ x,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ dx, int64_t n_in,
                                                          int H, int W, int oh, int ow, int k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_in; i += step) {
    int w = (int)(i % W);
    int64_t t = i / W;
    int h = (int)(t % H);
    int64_t nc = t / H;
    int oy = h / k, ox = w / k;
    float r = 0.0f;
    if (oy < oh && ox < ow) {
      int64_t o = (nc * oh + oy) * ow + ox;
      r = (y[o] == x[i] ? 1.0f : 0.0f) * dy[o];  // every tied maximum gets the gradient
    }
    dx[i] = r;
  }
}

// Dropout in training mode (nn/modules/dropout.py:31: `x * mask / (1 - p)`): forward BWD = false: y = (x * mask) / s; backward
// BWD = true: dx = (g / s) * mask -- the two nodes' arithmetic (a product, a division by the scalar: backend_tensor.py:551-586
// and their grad_fn's, tensor.py:640-676) in the same order, one launch instead of two each way.
template <bool BWD>
__global__ __launch_bounds__(kThreads) void k_dropout(const float* __restrict__ x, const float* __restrict__ m,
                                                      float* __restrict__ y, int64_t n, float s) {
  const int64_t n4 = n >> 2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  auto f = [s](float a, float k) { return BWD ? __fmul_rn(__fdiv_rn(a, s), k) : __fdiv_rn(__fmul_rn(a, k), s); };
  for (int64_t i = tid; i < n4; i += step) {
    const float4 v = reinterpret_cast<const float4*>(x)[i], k = reinterpret_cast<const float4*>(m)[i];
    float4 r;
    r.x = f(v.x, k.x); r.y = f(v.y, k.y); r.z = f(v.z, k.z); r.w = f(v.w, k.w);
    reinterpret_cast<float4*>(y)[i] = r;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) y[i] = f(x[i], m[i]);
}

__global__ __launch_bounds__(kThreads) void k_relu_fwd(const float* __restrict__ x,
                                                       float* __restrict__ y, int64_t n) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  auto f = [](float a) { return (a != a) ? a : (a > 0.0f ? a : 0.0f); };
  for (int64_t i = tid; i < n4; i += step) {
    float4 v = x4[i], r;
    r.x = f(v.x); r.y = f(v.y); r.z = f(v.z); r.w = f(v.w);
    y4[i] = r;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) y[i] = f(x[i]);
}

__global__ __launch_bounds__(kThreads) void k_relu_bwd(const float* __restrict__ x,
                                                       const float* __restrict__ y,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ dx, int64_t n) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* d4 = reinterpret_cast<float4*>(dx);
  auto f = [](float a, float b, float gg) { return (b == a ? 1.0f : 0.0f) * gg; };
  for (int64_t i = tid; i < n4; i += step) {
    float4 a = x4[i], b = y4[i], gg = g4[i], r;
    r.x = f(a.x, b.x, gg.x); r.y = f(a.y, b.y, gg.y); r.z = f(a.z, b.z, gg.z); r.w = f(a.w, b.w, gg.w);
    d4[i] = r;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) dx[i] = f(x[i], y[i], g[i]);
}

// ---------------------------------------------------------------------------------------
// softmax cross-entropy on dense targets, one wave per row
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_ce_rows(const float* __restrict__ logits,
                                                      const float* __restrict__ target,
                                                      float* __restrict__ row_loss, int rows,
                                                      int cols) {
  int lane = threadIdx.x & 63;
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < rows; r += nwaves) {
    const float* x = logits + (int64_t)r * cols;
    const float* t = target + (int64_t)r * cols;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float se = 0.0f;
    for (int j = lane; j < cols; j += 64) se += expf(x[j] - m);
    se = wave_sum(se);
    float lse = logf(se);
    float acc = 0.0f;
    for (int j = lane; j < cols; j += 64) acc += -((x[j] - m) - lse) * t[j];
    acc = wave_sum(acc);
    if (lane == 0) row_loss[r] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void k_ce_finish(const float* __restrict__ row_loss,
                                                        float* __restrict__ loss, int rows,
                                                        float inv_rows) {
  __shared__ float sh[4];
  float s = 0.0f;
  for (int i = threadIdx.x; i < rows; i += kThreads) s += row_loss[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss[0] = s * inv_rows;
}

// rows <= 4096, cols <= 32 (a classifier head): rows and the mean in ONE launch of one workgroup.
// A THREAD owns a row (its few logits live in registers, one round of loads), the row losses are
// summed by a fixed-shape tree in LDS.
// GRAD: also d loss / d logits for an upstream gradient of exactly 1 (what loss.backward() seeds, tensor.py:428),
// with k_ce_bwd's arithmetic: the backward pass of the usual training step then needs no launch for this node.
// CM: compile-time bound of cols (16 or 32).  The row losses are summed by a fixed-shape tree: shuffles inside a
// wave, then the sixteen waves in order.
template <bool GRAD, int CM>
__global__ __launch_bounds__(1024) void k_ce_small(const float* __restrict__ logits,
                                                   const float* __restrict__ target,
                                                   float* __restrict__ loss, float* __restrict__ dlogits, int rows,
                                                   int cols, float inv_rows) {
  __shared__ float part[16];
  float tot = 0.0f;
  for (int r = threadIdx.x; r < rows; r += 1024) {
    const float* x = logits + (int64_t)r * cols;
    const float* t = target + (int64_t)r * cols;
    float xv[CM], tv[CM];
#pragma unroll
    for (int j = 0; j < CM; j++) {
      xv[j] = j < cols ? x[j] : -INFINITY;
      tv[j] = j < cols ? t[j] : 0.0f;
    }
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < CM; j++) m = fmaxf(m, xv[j]);
    float ex[CM], se = 0.0f;
#pragma unroll
    for (int j = 0; j < CM; j++) {
      ex[j] = expf(xv[j] - m);                 // (exp(-inf) = 0 for the columns past the end)
      if (j < cols) se += ex[j];
    }
    const float lse = logf(se);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < CM; j++)
      if (j < cols) acc += -((xv[j] - m) - lse) * tv[j];
    tot += acc;
    if constexpr (GRAD) {
      const float gs = 1.0f * inv_rows;
      float st = 0.0f;
#pragma unroll
      for (int j = 0; j < CM; j++)
        if (j < cols) st += tv[j];
      // g_u[j] = gs * (softmax_j * sum(t) - t_j); the max() node routes -sum_j g_u to the ties (as k_ce_bwd)
      float gu[CM], gsum = 0.0f;
#pragma unroll
      for (int j = 0; j < CM; j++) {
        gu[j] = gs * tv[j] * -1.0f + ex[j] * ((gs * st) / se);
        if (j < cols) gsum += gu[j];
      }
      float* d = dlogits + (int64_t)r * cols;
#pragma unroll
      for (int j = 0; j < CM; j++)
        if (j < cols) d[j] = gu[j] + (xv[j] == m ? -gsum : 0.0f);
    }
  }
  tot = wave_sum(tot);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s2 = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; w++) s2 += part[w];
    loss[0] = s2 * inv_rows;
  }
}

__global__ __launch_bounds__(kThreads) void k_ce_bwd(const float* __restrict__ logits,
                                                     const float* __restrict__ target,
                                                     const float* __restrict__ gloss,
                                                     float* __restrict__ dlogits, int rows, int cols,
                                                     float inv_rows) {
  int lane = threadIdx.x & 63;
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float gs = gloss[0] * inv_rows;
  for (int r = wave; r < rows; r += nwaves) {
    const float* x = logits + (int64_t)r * cols;
    const float* t = target + (int64_t)r * cols;
    float* d = dlogits + (int64_t)r * cols;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float se = 0.0f, st = 0.0f;
    for (int j = lane; j < cols; j += 64) {
      se += expf(x[j] - m);
      st += t[j];
    }
    se = wave_sum(se);
    st = wave_sum(st);
    // g_u[j] = gs * (softmax_j * sum(t) - t_j); the max() node routes -sum_j g_u to the ties
    float gsum = 0.0f;
    for (int j = lane; j < cols; j += 64) gsum += gs * t[j] * -1.0f + expf(x[j] - m) * ((gs * st) / se);
    gsum = wave_sum(gsum);
    for (int j = lane; j < cols; j += 64) {
      float gu = gs * t[j] * -1.0f + expf(x[j] - m) * ((gs * st) / se);
      d[j] = gu + (x[j] == m ? -gsum : 0.0f);
    }
  }
}

// ---------------------------------------------------------------------------------------
// un-broadcast, axis reduce, axpy
// ---------------------------------------------------------------------------------------
struct RShape {
  int nd;
  int64_t src[DFHIP_MAX_DIMS];  // source extents
  int keep[DFHIP_MAX_DIMS];     // 1 if the axis survives in dst
};

// one block per destination element
__global__ __launch_bounds__(kThreads) void k_reduce_to_shape(const float* __restrict__ src,
                                                              float* __restrict__ dst, RShape rs,
                                                              int64_t n_dst, int64_t n_red) {
  __shared__ float sh[4];
  for (int64_t d = blockIdx.x; d < n_dst; d += gridDim.x) {
    // destination coordinates over the kept axes
    int64_t base = 0, rem = d, stride = 1;
    int64_t strides[DFHIP_MAX_DIMS];
    for (int a = rs.nd - 1; a >= 0; a--) {
      strides[a] = stride;
      stride *= rs.src[a];
    }
    for (int a = rs.nd - 1; a >= 0; a--)
      if (rs.keep[a]) {
        int64_t c = rem % rs.src[a];
        rem /= rs.src[a];
        base += c * strides[a];
      }
    float s = 0.0f;
    for (int64_t r = threadIdx.x; r < n_red; r += kThreads) {
      int64_t off = base, rr = r;
      for (int a = rs.nd - 1; a >= 0; a--)
        if (!rs.keep[a]) {
          int64_t c = rr % rs.src[a];
          rr /= rs.src[a];
          off += c * strides[a];
        }
      s += src[off];
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) dst[d] = s;
    __syncthreads();
  }
}

// (N, C, H, W) -> (1, C, 1, 1): the bias gradient of a convolution (conv.py:104-108 adds the bias
// by broadcasting, so its gradient is un-broadcast by the engine).  Block (c, s) sums its images of
// channel c with 16-byte loads; a second tiny launch adds the S partials in order.
__global__ __launch_bounds__(kThreads) void k_channel_sum(const float* __restrict__ x, float* __restrict__ partial,
                                                          int N, int C, int HW, int S, int ipb) {
  __shared__ float sh[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const int n_lo = s * ipb, n_hi = min(N, n_lo + ipb);
  const float* base = x + (int64_t)c * HW;
  float sum = 0.0f;
  for_each_chunk(base, (int64_t)C * HW, n_lo, n_hi, HW, [&](const float* p, int w) {
    if (w == 4) {
      float4 v = *reinterpret_cast<const float4*>(p);
      sum += (v.x + v.y) + (v.z + v.w);
    } else {
      sum += p[0];
    }
  });
  sum = block_sum(sum, sh);
  if (threadIdx.x == 0) partial[(int64_t)c * S + s] = sum;
}

__global__ __launch_bounds__(kThreads) void k_channel_sum_finish(const float* __restrict__ partial,
                                                                 float* __restrict__ dst, int C, int S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.0f;
  for (int i = 0; i < S; i++) s += partial[(int64_t)c * S + i];
  dst[c] = s;
}

// dst[o][i] = scale * reduce_r src[o][r][i]
template <bool IS_MAX>
__global__ __launch_bounds__(kThreads) void k_reduce_axis(const float* __restrict__ src,
                                                          float* __restrict__ dst, int64_t outer,
                                                          int64_t red, int64_t inner, float scale) {
  int64_t n = outer * inner;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int64_t o = i / inner, in = i - o * inner;
    const float* p = src + o * red * inner + in;
    float v = p[0];
    for (int64_t r = 1; r < red; r++) {
      float u = p[r * inner];
      if (IS_MAX) v = (v != v) ? v : ((u != u) ? u : (u > v ? u : v));
      else v += u;
    }
    dst[i] = v * scale;
  }
}

__global__ __launch_bounds__(kThreads) void k_broadcast_axis(const float* __restrict__ src,
                                                             float* __restrict__ dst, int64_t outer,
                                                             int64_t red, int64_t inner, float scale) {
  int64_t n = outer * red * inner;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int64_t in = i % inner;
    int64_t o = i / (red * inner);
    dst[i] = src[o * inner + in] * scale;
  }
}

__global__ __launch_bounds__(kThreads) void k_axpy(float* __restrict__ dst,
                                                   const float* __restrict__ src, float alpha,
                                                   int64_t n) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  float4* d4 = reinterpret_cast<float4*>(dst);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (int64_t i = tid; i < n4; i += step) {
    float4 a = d4[i], b = s4[i];
    a.x += alpha * b.x; a.y += alpha * b.y; a.z += alpha * b.z; a.w += alpha * b.w;
    d4[i] = a;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) dst[i] += alpha * src[i];
}

// ---------------------------------------------------------------------------------------
// multi-tensor optimizers: up to kBatch tensors per launch, descriptors passed by value in
// the kernel argument block (captured by value in a hipGraph, no device-side table upload)
// ---------------------------------------------------------------------------------------
constexpr int kBatch = 64;   // 64 x 44 B of descriptors: inside the 4 KB kernel-argument block
constexpr int kChunk = 4096;  // elements per block

struct TensorBatch {
  float* p[kBatch];
  const float* g[kBatch];
  float* s1[kBatch];
  float* s2[kBatch];
  int64_t size[kBatch];
  int blk_start[kBatch + 1];
  int n;
};

struct AdamHyper {
  float lr, b1, b2, omb1, omb2, bc1, bc2, eps, wd, gscale;
};

// the tensor a block belongs to: the last t with blk_start[t] <= blk (blk_start is ascending; six dependent scalar loads
// instead of up to 62 -- the big tensors are the last ones of a ResNet's list)
__device__ __forceinline__ int find_tensor(const TensorBatch& tb, int blk) {
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (blk >= tb.blk_start[mid]) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// one element of the update — optim/adam.py:42-60, same operation order
__device__ __forceinline__ void adam_element(float& p, float gi, float& m, float& v, const AdamHyper& h) {
  if (h.gscale != 1.0f) gi = gi * h.gscale;
  const float pi = p;
  if (h.wd > 0.0f) gi = gi + pi * h.wd;
  const float mi = m * h.b1 + gi * h.omb1;
  const float vi = v * h.b2 + (gi * gi) * h.omb2;
  m = mi;
  v = vi;
  const float mh = mi / h.bc1;
  const float vh = vi / h.bc2;
  const float upd = mh / (sqrtf(vh) + h.eps) * h.lr;
  p = pi + upd * -1.0f;
}

// a block owns kChunk consecutive elements of one tensor; 16-byte accesses when the four arrays
// are 16-byte aligned (pool blocks and bucket views are, except behind an odd-sized tensor)
__device__ __forceinline__ void adam_block(const TensorBatch& tb, const AdamHyper& h) {
  const int t = find_tensor(tb, blockIdx.x);
  const int64_t lo = (int64_t)(blockIdx.x - tb.blk_start[t]) * kChunk;
  const int64_t hi = min(tb.size[t], lo + kChunk);
  float* p = tb.p[t];
  const float* g = tb.g[t];
  float* m = tb.s1[t];
  float* v = tb.s2[t];
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  int64_t i = lo + threadIdx.x;
  const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
  if (vec && hi4 > lo) {
    // the block's (up to) four 16-byte pieces per thread and array are requested TOGETHER, in front of a compiler barrier
    // (round 6): written as a loop, the stores of one trip keep the loads of the next behind them (p, m, v may alias as far as
    // the compiler knows) -- four dependent round trips per block; and instruction selection sinks plain loads to their uses
    constexpr int U = kChunk / (4 * kThreads);
    float4 pv[U], mv[U], vv[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t jj = min(lo + 4 * ((int64_t)threadIdx.x + (int64_t)u * kThreads), hi4 - 4);   // (clamped: no load under a condition)
      pv[u] = *reinterpret_cast<const float4*>(p + jj); mv[u] = *reinterpret_cast<const float4*>(m + jj);
      vv[u] = *reinterpret_cast<const float4*>(v + jj); gv[u] = *reinterpret_cast<const float4*>(g + jj);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = lo + 4 * ((int64_t)threadIdx.x + (int64_t)u * kThreads);
      if (j < hi4) {
        adam_element(pv[u].x, gv[u].x, mv[u].x, vv[u].x, h);
        adam_element(pv[u].y, gv[u].y, mv[u].y, vv[u].y, h);
        adam_element(pv[u].z, gv[u].z, mv[u].z, vv[u].z, h);
        adam_element(pv[u].w, gv[u].w, mv[u].w, vv[u].w, h);
        *reinterpret_cast<float4*>(p + j) = pv[u];
        *reinterpret_cast<float4*>(m + j) = mv[u];
        *reinterpret_cast<float4*>(v + j) = vv[u];
      }
    }
    i = hi4 + threadIdx.x;                         // the (< 4 element) tail
  }
  for (; i < hi; i += kThreads) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_element(pi, g[i], mi, vi, h);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

__global__ __launch_bounds__(kThreads) void k_adam(TensorBatch tb, AdamHyper h) { adam_block(tb, h); }

// the same update with the hyper-parameters read from device memory: a captured training step
// (hipGraph) replays this launch unchanged while dfhip_adam_hyper refreshes the scalars (bias
// corrections change every step) with a one-thread launch in front of the replay
// hyper_dev: int idx, int done, int cap, ... then from float 16 on `cap` entries of 16 floats (an AdamHyper each): the
// scalars of steps first, first + 1, ...  (dfhip_adam_hyper_table).  Every block reads entry idx; with `advance`
// the LAST block to finish moves idx on -- every block has read idx before it counts itself done, so none can see
// the new value -- and the next replay of the captured step finds the next step's bias corrections without any
// launch in between (the one-thread launch that used to write them cost 4.9 us per step).
__global__ __launch_bounds__(kThreads) void k_adam_dev(TensorBatch tb, float* __restrict__ hyper_dev, int advance) {
  int* ctl = reinterpret_cast<int*>(hyper_dev);
  const int idx = ctl[0], cap = ctl[2];
  const AdamHyper h = *reinterpret_cast<const AdamHyper*>(hyper_dev + 16 + 16 * min(idx, cap - 1));
  adam_block(tb, h);
  if (!advance) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = atomicAdd(&ctl[1], 1);
    if (old == (int)gridDim.x - 1) {
      ctl[1] = 0;
      ctl[0] = idx + 1;
    }
  }
}

__global__ void k_set_adam_hyper(float* hyper_dev, AdamHyper h) {
  int* ctl = reinterpret_cast<int*>(hyper_dev);
  ctl[0] = 0; ctl[1] = 0; ctl[2] = 1;
  *reinterpret_cast<AdamHyper*>(hyper_dev + 16) = h;
}

__device__ __forceinline__ void sgd_block(const TensorBatch& tb, const SgdHyper& h);

__global__ __launch_bounds__(kThreads) void k_sgd(TensorBatch tb, SgdHyper h) { sgd_block(tb, h); }

// the same update with the scalars read from device memory (dfhip_sgd_hyper writes them): inside a captured training
// step the launch is replayed unchanged while a learning-rate schedule keeps moving the scalars from the host
__global__ __launch_bounds__(kThreads) void k_sgd_dev(TensorBatch tb, const float* __restrict__ hyper_dev) {
  const SgdHyper h = *reinterpret_cast<const SgdHyper*>(hyper_dev);
  sgd_block(tb, h);
}

__global__ void k_set_sgd_hyper(float* hyper_dev, SgdHyper h) { *reinterpret_cast<SgdHyper*>(hyper_dev) = h; }

__device__ __forceinline__ void sgd_block(const TensorBatch& tb, const SgdHyper& h) {
  int t = find_tensor(tb, blockIdx.x);
  int64_t lo = (int64_t)(blockIdx.x - tb.blk_start[t]) * kChunk;
  int64_t hi = min(tb.size[t], lo + kChunk);
  float* p = tb.p[t];
  const float* g = tb.g[t];
  float* mb = tb.s1[t];
  // (a block's 16 elements per thread are requested together: a plain loop was 16 dependent round trips -- 6.9 us for the
  // perceptron's 0.2 M parameters, most of it latency)
  constexpr int U = kChunk / kThreads;
  float gv[U], pv[U], mv[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t i = min(lo + threadIdx.x + (int64_t)u * kThreads, hi - 1);
    gv[u] = g[i];
    pv[u] = p[i];
    mv[u] = h.use_momentum ? mb[i] : 0.0f;
  }
  asm volatile("" ::: "memory");                   // (a compiler barrier: the selection DAG sinks plain loads to their uses)
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t i = lo + threadIdx.x + (int64_t)u * kThreads;
    if (i >= hi) break;
    sgd_elem(p, mb, i, gv[u], pv[u], mv[u], h);       // optim/sgd.py:16-24 (dfhip_head.h)
  }
}

// dir 0: flat <- tensors, 1: tensors <- flat
__global__ __launch_bounds__(kThreads) void k_pack(TensorBatch tb, float* __restrict__ flat,
                                                   int64_t flat_base, int dir) {
  int t = find_tensor(tb, blockIdx.x);
  int64_t lo = (int64_t)(blockIdx.x - tb.blk_start[t]) * kChunk;
  int64_t hi = min(tb.size[t], lo + kChunk);
  // s1[t] reused to carry nothing; the flat offset of tensor t is kept in s2 as an integer
  int64_t off = (int64_t)(intptr_t)tb.s2[t];
  float* f = flat + flat_base + off;
  if (dir == 0) {
    const float* s = tb.g[t];
    for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) f[i] = s[i];
  } else {
    float* d = tb.p[t];
    for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) d[i] = f[i];
  }
}

int fill_batch(TensorBatch& tb, int n, int first, float* const* p, const float* const* g,
               float* const* s1, float* const* s2, const int64_t* sizes) {
  tb.n = n;
  int blk = 0;
  for (int i = 0; i < n; i++) {
    tb.p[i] = p ? p[first + i] : nullptr;
    tb.g[i] = g ? g[first + i] : nullptr;
    tb.s1[i] = s1 ? s1[first + i] : nullptr;
    tb.s2[i] = s2 ? s2[first + i] : nullptr;
    tb.size[i] = sizes[first + i];
    tb.blk_start[i] = blk;
    blk += (int)df_ceil_div(std::max<int64_t>(sizes[first + i], 1), kChunk);
  }
  tb.blk_start[n] = blk;
  return blk;
}

}  // namespace


// ---- classifier head: y = (scale * sum_hw x) . W + b, the two mean(axis=2) calls and the Linear of
// test/ResNet_CIFAR10_cuda.py:96-99 as one launch each way.  A workgroup is an image; a thread walks channels.
template <int P>
__device__ __forceinline__ void head_block_sum(float (&acc)[P], float (*sh)[P]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < P; j++) {
    float v = acc[j];
    v = wave_sum(v);
    if (lane == 0) sh[wave][j] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < P; j++) acc[j] = sh[0][j] + sh[1][j] + sh[2][j] + sh[3][j];
}

template <int P>
__global__ __launch_bounds__(256) void k_pool_linear_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ pooled,
                                                        float* __restrict__ y, int C, int HW, int Pn, float scale) {
  __shared__ float sh[4][P];
  const int n = blockIdx.x;
  float acc[P];
#pragma unroll
  for (int j = 0; j < P; j++) acc[j] = 0.0f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* xp = x + ((int64_t)n * C + c) * HW;
    float s = 0.0f;
    if ((HW & 3) == 0) {
      for (int i = 0; i < HW; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(xp + i);
        s += v.x; s += v.y; s += v.z; s += v.w;
      }
    } else {
      for (int i = 0; i < HW; i++) s += xp[i];
    }
    const float pv = s * scale;
    pooled[(int64_t)n * C + c] = pv;
    // (no `if (j < Pn)` around the multiply-adds: on the uniform Pn each becomes a branch; the columns past Pn
    // re-read the last one and are never stored)
    const float* wr = w + (int64_t)c * Pn;
#pragma unroll
    for (int j = 0; j < P; j++) acc[j] = fmaf(pv, wr[min(j, Pn - 1)], acc[j]);
  }
  head_block_sum<P>(acc, sh);
  if ((int)threadIdx.x < Pn) {
    float r = 0.0f;
#pragma unroll
    for (int j = 0; j < P; j++)
      if (j == (int)threadIdx.x) r = acc[j];
    y[(int64_t)n * Pn + threadIdx.x] = r + (b ? b[threadIdx.x] : 0.0f);
  }
}

// 1024 threads.  Blocks [0, ceil(N / 4)): dx[n][c][:] = gscale * sum_j dy[n][j] w[c][j], four images per block;
// blocks [.., + ceil(C / 64)): dw[c][j] = sum_n pooled[n][c] dy[n][j] for 64 channels (a lane is a channel: coalesced
// rows of pooled; 256 images at a time: their dy rows go to LDS padded to P columns, a wave requests the pooled
// values of its 16 images -- every sixteenth -- in one burst, the sixteen waves are summed through LDS in a fixed
// order);  last block: db[j] = sum_n dy[n][j].  No `if (j < Pn)` around inner-loop arithmetic: on the uniform Pn each
// becomes a branch, and a load under a condition a round trip of its own (the first version took 37 us).
template <int P>
__global__ __launch_bounds__(1024) void k_pool_linear_bwd(const float* __restrict__ dy, const float* __restrict__ pooled,
                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                         float* __restrict__ dw, float* __restrict__ db, int N, int C,
                                                         int HW, int Pn, float gscale) {
  __shared__ float shw[8][64][P + 1];
  __shared__ float sdy[256 * P];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ndx = (N + 3) / 4, ndw = (C + 63) / 64;
  if (b < ndx) {
    const int n = b * 4 + (tid >> 8), t8 = tid & 255;
    if (!dx || n >= N) return;
    float g[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
      const float v = dy[(int64_t)n * Pn + min(j, Pn - 1)];
      g[j] = j < Pn ? v : 0.0f;
    }
    for (int c = t8; c < C; c += 256) {
      const float* wr = w + (int64_t)c * Pn;
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < P; j++) s = fmaf(g[j], wr[min(j, Pn - 1)], s);
      s *= gscale;
      float* o = dx + ((int64_t)n * C + c) * HW;
      if ((HW & 3) == 0) {
        const float4 v = make_float4(s, s, s, s);
        for (int i = 0; i < HW; i += 4) *reinterpret_cast<float4*>(o + i) = v;
      } else {
        for (int i = 0; i < HW; i++) o[i] = s;
      }
    }
    return;
  }
  if (b < ndx + ndw) {
    if (!dw) return;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = (b - ndx) * 64 + lane;
    const int cc = c < C ? c : C - 1;
    float acc[P];
#pragma unroll
    for (int j = 0; j < P; j++) acc[j] = 0.0f;
    for (int n0 = 0; n0 < N; n0 += 256) {
      const int nb = min(256, N - n0);
      if (n0 > 0) __syncthreads();
      float stg[P];
      if (tid < 256) {
#pragma unroll
        for (int j = 0; j < P; j++) {
          const float v = dy[(int64_t)(n0 + min(tid, nb - 1)) * Pn + min(j, Pn - 1)];
          stg[j] = (tid < nb && j < Pn) ? v : 0.0f;
        }
      }
      float pv[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int n = wave + 16 * u;
        const float v = pooled[(int64_t)(n0 + min(n, nb - 1)) * C + cc];
        pv[u] = n < nb ? v : 0.0f;
      }
      if (tid < 256) {
#pragma unroll
        for (int j = 0; j < P; j++) sdy[tid * P + j] = stg[j];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const float* g = sdy + (wave + 16 * u) * P;
#pragma unroll
        for (int j = 0; j < P; j++) acc[j] = fmaf(pv[u], g[j], acc[j]);
      }
    }
    if (wave >= 8) {
#pragma unroll
      for (int j = 0; j < P; j++) shw[wave - 8][lane][j] = acc[j];
    }
    __syncthreads();
    if (wave < 8) {
#pragma unroll
      for (int j = 0; j < P; j++) acc[j] += shw[wave][lane][j];
    }
    __syncthreads();
    if (wave >= 1 && wave < 8) {
#pragma unroll
      for (int j = 0; j < P; j++) shw[wave][lane][j] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && c < C) {
#pragma unroll
      for (int j = 0; j < P; j++) {
        float r = acc[j];
#pragma unroll
        for (int w2 = 1; w2 < 8; w2++) r += shw[w2][lane][j];
        if (j < Pn) dw[(int64_t)c * Pn + j] = r;
      }
    }
    return;
  }
  if (!db) return;
  float acc[P];
#pragma unroll
  for (int j = 0; j < P; j++) acc[j] = 0.0f;
  for (int n = tid; n < N; n += 1024) {
    const float* g = dy + (int64_t)n * Pn;
#pragma unroll
    for (int j = 0; j < P; j++) acc[j] += g[min(j, Pn - 1)];
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < P; j++) {
    acc[j] = wave_sum(acc[j]);
    if (lane == 0) sdy[wave * P + j] = acc[j];
  }
  __syncthreads();
  if (tid < Pn) {
    float r = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < 16; w2++) r += sdy[w2 * P + tid];
    db[tid] = r;
  }
}

// ---- classifier head AND its loss in one launch (training step): y = fc(mean(mean(x, 2), 2)), loss = CE(y, target) -- what
// test/ResNet_CIFAR10_cuda.py:96-99 + :186 (`criterion(outputs, targets)`) compute with three nodes -- and, because every
// quantity of the backward pass that belongs to ONE image depends on that image's logits only, d loss / d logits (for the
// upstream gradient of exactly 1 that loss.backward() seeds, tensor.py:428) and the input gradient dx as well.  A workgroup
// is an image, a thread walks CPT channels (their pooled values and rows of W stay in registers between the forward and the
// backward half).  The arithmetic is that of k_pool_linear_fwd, k_ce_small<true> and k_pool_linear_bwd expression by
// expression.  The loss is the mean of the row losses: the last workgroup to arrive (agent-scope counter) adds them up in
// the order k_ce_small uses; what is left of the backward pass -- dW, db: sums over the images -- is head_grads_body
// (dfhip_head.h).  Rounds 1-5: three launches (4.7 + 5.1 + 7.7 us at batch 256) for 2 MB of traffic.
// IPB images per workgroup (256 threads each).
// DROP (round 6, the CNN workload's tail -- test/CNN_CIFAR10_cuda.py:104-107 `x = self.drop(x); x = self.fc(x)` + the
// criterion): the rows are two-dimensional (HW == 1) and what the Linear reads is Dropout's x * mask / keep
// (nn/modules/dropout.py:31, k_dropout's two roundings); `pooled` is then Dropout's output and dx the gradient at Dropout's
// INPUT (k_dropout<true>'s (g / keep) * mask).  All of a thread's operands are requested before the first is used.
template <int P, int CPT, int IPB, bool DROP = false>
__global__ __launch_bounds__(256 * IPB) void k_head_ce(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, const float* __restrict__ target,
                                                      float* __restrict__ pooled, float* __restrict__ logits,
                                                      float* __restrict__ dlogits, float* __restrict__ dx,
                                                      float* rowloss, float* __restrict__ loss, int N, int C, int HW, int Pn,
                                                      float scale, float gscale, float inv_rows,
                                                      const float* __restrict__ mask, float keep) {
  __shared__ float sh[IPB][4][P];
  __shared__ float slog[IPB][P];
  const int img = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int n = min((int)blockIdx.x * IPB + img, N - 1);        // (an image past the end re-does the last one; nothing of it is stored)
  const bool live = (int)blockIdx.x * IPB + img < N;
  float acc[P], pv[CPT], wr[DROP ? 1 : CPT][P];      // (DROP: the weights stay in LDS)
#pragma unroll
  for (int j = 0; j < P; j++) acc[j] = 0.0f;
  float tv[P];                                     // the target row: requested with the first loads
#pragma unroll
  for (int j = 0; j < P; j++) {
    const float v = target[(int64_t)n * Pn + min(j, Pn - 1)];
    tv[j] = j < Pn ? v : 0.0f;
  }
  float mk[DROP ? CPT : 1];
  if constexpr (DROP) {
    // The weights (C x Pn: 80 KB for the CNN workload's 2 048 x 10) go through LDS: a thread's channels are 256 apart and a
    // channel's Pn weights 4 Pn bytes from the next channel's, so loading them per channel is 16 CPT requests of 64 scattered
    // 4-byte words each (measured: 14.4 us for the launch); the flat copy is C Pn / 1 024 16-byte requests per thread.
    extern __shared__ __attribute__((aligned(16))) float head_wl[];
    float xr[CPT];
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const int cc = min(tid + 256 * q, C - 1);
      xr[q] = x[(int64_t)n * C + cc];
      mk[q] = mask[(int64_t)n * C + cc];
    }
    {
      constexpr int R4 = 20;                       // 16-byte requests of a thread per round
      const int total = C * Pn, total4 = (((uintptr_t)w & 15) == 0) ? total >> 2 : 0;
      for (int base = 0; base < total4; base += 256 * R4) {
        typedef float f32x4w __attribute__((ext_vector_type(4)));   // (HIP's float4 struct is not promoted to registers here)
        f32x4w wv4[R4];
#pragma unroll
        for (int u = 0; u < R4; u++) wv4[u] = reinterpret_cast<const f32x4w*>(w)[min(base + tid + 256 * u, total4 - 1)];
        asm volatile("" ::: "memory");             // (a compiler barrier: plain loads are sunk to their first use otherwise)
#pragma unroll
        for (int u = 0; u < R4; u++)
          if (base + tid + 256 * u < total4) reinterpret_cast<f32x4w*>(head_wl)[base + tid + 256 * u] = wv4[u];
      }
      for (int i = (total4 << 2) + tid; i < total; i += 256) head_wl[i] = w[i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const int c = tid + 256 * q, cc = min(c, C - 1);
      pv[q] = c < C ? __fdiv_rn(__fmul_rn(xr[q], mk[q]), keep) : 0.0f;
      if (c < C && live) pooled[(int64_t)n * C + c] = pv[q];
#pragma unroll
      for (int j = 0; j < P; j++) acc[j] = fmaf(pv[q], head_wl[cc * Pn + min(j, Pn - 1)], acc[j]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const int c = tid + 256 * q, cc = min(c, C - 1);
      const float* xp = x + ((int64_t)n * C + cc) * HW;
      float s = 0.0f;
      if ((HW & 3) == 0) {
        for (int i = 0; i < HW; i += 4) {
          const float4 v = *reinterpret_cast<const float4*>(xp + i);
          s += v.x; s += v.y; s += v.z; s += v.w;
        }
      } else {
        for (int i = 0; i < HW; i++) s += xp[i];
      }
      pv[q] = c < C ? s * scale : 0.0f;
      if (c < C && live) pooled[(int64_t)n * C + c] = pv[q];
      const float* wp = w + (int64_t)cc * Pn;
#pragma unroll
      for (int j = 0; j < P; j++) wr[q][j] = wp[min(j, Pn - 1)];
#pragma unroll
      for (int j = 0; j < P; j++) acc[j] = fmaf(pv[q], wr[q][j], acc[j]);
    }
  }
  {                                                // head_block_sum per image (its four waves)
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < P; j++) {
      float v = acc[j];
      v = wave_sum(v);
      if (lane == 0) sh[img][wave][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < P; j++) acc[j] = sh[img][0][j] + sh[img][1][j] + sh[img][2][j] + sh[img][3][j];
  }
  if (tid < P) {
    float r = 0.0f;
#pragma unroll
    for (int j = 0; j < P; j++)
      if (j == tid) r = acc[j];
    const float y = r + ((b && tid < Pn) ? b[tid] : 0.0f);
    slog[img][tid] = tid < Pn ? y : -INFINITY;
    if (tid < Pn && live) logits[(int64_t)n * Pn + tid] = y;
  }
  __syncthreads();
  // cross-entropy of the row and d loss / d logits, by every thread (k_ce_small<true>'s expressions)
  float xv[P];
#pragma unroll
  for (int j = 0; j < P; j++) xv[j] = slog[img][j];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < P; j++) m = fmaxf(m, xv[j]);
  float ex[P], se = 0.0f;
#pragma unroll
  for (int j = 0; j < P; j++) {
    ex[j] = expf(xv[j] - m);
    if (j < Pn) se += ex[j];
  }
  const float lse = logf(se);
  float row = 0.0f;
#pragma unroll
  for (int j = 0; j < P; j++)
    if (j < Pn) row += -((xv[j] - m) - lse) * tv[j];
  const float gs = 1.0f * inv_rows;
  float st = 0.0f;
#pragma unroll
  for (int j = 0; j < P; j++)
    if (j < Pn) st += tv[j];
  float gu[P], gsum = 0.0f;
#pragma unroll
  for (int j = 0; j < P; j++) {
    gu[j] = gs * tv[j] * -1.0f + ex[j] * ((gs * st) / se);
    if (j < Pn) gsum += gu[j];
  }
  float d[P];
#pragma unroll
  for (int j = 0; j < P; j++) d[j] = j < Pn ? gu[j] + (xv[j] == m ? -gsum : 0.0f) : 0.0f;
  if (tid < Pn && live) {
    float r = 0.0f;
#pragma unroll
    for (int j = 0; j < P; j++)
      if (j == tid) r = d[j];
    dlogits[(int64_t)n * Pn + tid] = r;
  }
  // the input gradient: dx[n][c][:] = gscale * sum_j d[j] w[c][j] (k_pool_linear_bwd's expressions)
  if (dx) {
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const int c = tid + 256 * q;
      float s = 0.0f;
      if constexpr (DROP) {
        extern __shared__ __attribute__((aligned(16))) float head_wl[];
        const int cc = min(c, C - 1);
#pragma unroll
        for (int j = 0; j < P; j++) s = fmaf(d[j], head_wl[cc * Pn + min(j, Pn - 1)], s);
        s = __fmul_rn(__fdiv_rn(s * gscale, keep), mk[q]);
      } else {
#pragma unroll
        for (int j = 0; j < P; j++) s = fmaf(d[j], wr[q][j], s);
        s *= gscale;
      }
      if (c < C && live) {
        float* o = dx + ((int64_t)n * C + c) * HW;
        if ((HW & 3) == 0) {
          const float4 v = make_float4(s, s, s, s);
          for (int i = 0; i < HW; i += 4) *reinterpret_cast<float4*>(o + i) = v;
        } else {
          for (int i = 0; i < HW; i++) o[i] = s;
        }
      }
    }
  }
  // The loss is the mean of the row losses: a sum over the workgroups.  Built and measured first (round 6): the last
  // workgroup to arrive adds them up -- a written-through store of the row loss, then a ticket (agent-scope atomics): two
  // dependent round trips to the memory side at the end of EVERY workgroup, 5.5 -> 10.3 us (12.3 us with a release fence,
  // which writes the XCD's whole L2 back; 13.1 us with four images per workgroup and a quarter of the tickets).  So the rows
  // are left as they are and the sum is part of whatever reads them next: head_grads_body in the backward pass (no launch
  // inside a weight-gradient group), dfhip_loss_finish when the loss is read before that (hip_device.DeferredHipArray).
  if (tid == 0 && live) rowloss[n] = row;
}

__global__ __launch_bounds__(256) void k_loss_finish(const float* __restrict__ rowloss, float* __restrict__ loss, int N,
                                                    float inv_rows) {
  __shared__ float part[16];
  loss_finish_body(rowloss, loss, N, inv_rows, part);
}

__global__ __launch_bounds__(256) void k_head_grads(const HeadGrads h) {
  __shared__ __attribute__((aligned(16))) float lds[kHeadLdsFloats];
  head_grads_body<false>(h, (int)blockIdx.x, lds);
}

int stem_bwd_stats(const float* z, const float* y, const float* g, const float* gamma, const float* beta,
                   const float* mean, const float* var, int N, int C, int H, int W, float eps, const float** partial,
                   int* S_out) {
  int S, ipb;
  bn_split(N, C, S, ipb);
  float* part = workspace_small((size_t)C * S * 2 * sizeof(float));
  if (!part) return DFHIP_OUT_OF_MEMORY;
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH((k_bn_relu_pool_bwd<false>), grid, dim3(kThreads), 0, stream(), z, y, g, gamma, beta, mean, var, part,
            (float*)nullptr, (float*)nullptr, (float*)nullptr, N, C, H, W, S, ipb, eps);
  DF_LAUNCH_CHECK();
  *partial = part;
  *S_out = S;
  return DFHIP_OK;
}

int stem_pool_apply(const float* sel, int sel_planes, const float* partials, int rows, const float* gamma, const float* beta, float* y,
                    float* save_mean, float* save_var, float* running_mean, float* running_var, int N, int C, int P,
                    float eps, float momentum) {
  if (rows < 1 || (P & 3)) {
    set_error("stem_pool_apply: no partial rows / pooled plane not a multiple of 4");
    return DFHIP_INVALID_ARGUMENT;
  }
  int S, ipb;
  bn_split(N, C, S, ipb);
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH(k_sel_bn_relu_fwd, grid, dim3(kThreads), 0, stream(), sel, sel_planes, partials, rows, gamma, beta, y, save_mean, save_var,
            running_mean, running_var, N, C, P, ipb, eps, momentum);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

namespace dfhip {
int head_grads_launch(const HeadGrads& h) {
  ProfScope prof("head_grads", 2.0 * h.N * (double)h.C * h.Pn, 4.0 * ((double)h.N * h.C + (double)h.N * h.Pn + (double)h.C * h.Pn));
  DF_LAUNCH(k_head_grads, dim3((unsigned)head_grads_blocks(h)), dim3(256), 0, stream(), h);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}
}  // namespace dfhip

extern "C" {

int dfhip_bn2d_forward_train(const float* x, const float* gamma, const float* beta, float* y,
                             float* save_mean, float* save_var, float* running_mean,
                             float* running_var, int N, int C, int HW, float eps, float momentum) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bn2d_forward: non-positive dimension");
  DF_CHECK_ARG(x && y, "bn2d_forward: null array");
  double n = (double)N * C * HW;
  ProfScope prof("bn2d_fwd", 0.0, 12.0 * n);
  if (bn_fused_ok(N, C, HW)) {
    DF_LAUNCH(k_bn_fused_fwd, dim3((unsigned)C), dim3(512), 0, stream(), x, gamma, beta, y, save_mean,
                       save_var, running_mean, running_var, N, C, HW, eps, momentum);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  int S, ipb;
  bn_split(N, C, S, ipb);
  float* partial = workspace_small((size_t)C * S * 3 * sizeof(float));
  if (!partial) return DFHIP_OUT_OF_MEMORY;
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH(k_bn_stats, grid, dim3(kThreads), 0, stream(), x, partial, N, C, HW, S, ipb);
  DF_LAUNCH_CHECK();
  DF_LAUNCH(k_bn_apply, grid, dim3(kThreads), 0, stream(), x, gamma, beta, y, partial,
                     save_mean, save_var, running_mean, running_var, N, C, HW, S, ipb, eps, momentum);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_bn2d_forward_partials(const float* x, const float* partials, int rows, const float* gamma,
                                const float* beta, const float* residual, float* y, float* save_mean,
                                float* save_var, float* running_mean, float* running_var, int N, int C,
                                int HW, float eps, float momentum) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0 && rows > 0, "bn2d_forward_partials: non-positive dimension");
  DF_CHECK_ARG(x && y && partials, "bn2d_forward_partials: null array");
  double n = (double)N * C * HW;
  ProfScope prof("bn2d_fwd", 0.0, (residual ? 12.0 : 8.0) * n);
  if (HW == 4 && C % 8 == 0) {
    // 2 x 2 planes: eight channels (a line per image) x 64 images per workgroup
    const int ipb4 = 64, S4 = (int)df_ceil_div(N, ipb4);
    BnFwdSet A = {x, partials, rows, gamma, beta, save_mean, save_var, running_mean, running_var, eps, momentum};
    DF_LAUNCH((k_bn_apply_partials_hw4<false>), dim3((unsigned)(C / 8), (unsigned)S4), dim3(kThreads), 0, stream(), A, A, residual,
              y, N, C, ipb4);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  int S, ipb;
  bn_split(N, C, S, ipb);
  DF_LAUNCH(k_bn_apply_partials, dim3((unsigned)C, (unsigned)S), dim3(kThreads), 0, stream(), x, partials,
                     rows, gamma, beta, residual, y, save_mean, save_var, running_mean, running_var, N, C, HW, ipb,
                     eps, momentum);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_bn2d_forward_eval(const float* x, const float* gamma, const float* beta, float* y,
                            const float* running_mean, const float* running_var, int N, int C,
                            int HW, float eps) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bn2d_eval: non-positive dimension");
  DF_CHECK_ARG(x && y && running_mean && running_var, "bn2d_eval: null array");
  int S, ipb;
  bn_split(N, C, S, ipb);
  ProfScope prof("bn2d_eval", 0.0, 8.0 * (double)N * C * HW);
  DF_LAUNCH(k_bn_eval, dim3((unsigned)C, (unsigned)S), dim3(kThreads), 0, stream(), x, gamma,
                     beta, y, running_mean, running_var, N, C, HW, ipb, eps);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_bn2d_backward(const float* dy, const float* x, const float* gamma, const float* save_mean,
                        const float* save_var, float* dx, float* dgamma, float* dbeta, int N, int C,
                        int HW, float eps) {
  DF_INIT_KEEP();
  DF_TOUCH(dy);
  DF_TOUCH(x);
  DF_TOUCH(gamma);
  DF_TOUCH(save_mean);
  DF_TOUCH(save_var);
  DF_TOUCH(dx);
  DF_TOUCH(dgamma);
  DF_TOUCH(dbeta);
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bn2d_backward: non-positive dimension");
  DF_CHECK_ARG(dy && x && save_mean && save_var, "bn2d_backward: null array");
  ProfScope prof("bn2d_bwd", 0.0, 20.0 * (double)N * C * HW);
  if (bn_hw4_ok(N, C, HW)) {
    DF_LAUNCH(k_bn_fused_bwd_hw4, dim3((unsigned)(C / 8)), dim3(512), 0, stream(), dy, x, gamma, save_mean, save_var, dx,
              dgamma, dbeta, N, C, eps);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  if (bn_fused_ok(N, C, HW)) {
    DF_LAUNCH(k_bn_fused_bwd, dim3((unsigned)C), dim3(512), 0, stream(), dy, x, gamma, save_mean,
                       save_var, dx, dgamma, dbeta, N, C, HW, eps);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  int S, ipb;
  bn_split(N, C, S, ipb);
  float* partial = workspace_small((size_t)C * S * 2 * sizeof(float));
  if (!partial) return DFHIP_OUT_OF_MEMORY;
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH(k_bn_bwd_stats, grid, dim3(kThreads), 0, stream(), dy, x, save_mean, save_var,
                     partial, N, C, HW, S, ipb, eps);
  DF_LAUNCH_CHECK();
  DF_LAUNCH(k_bn_bwd_apply, grid, dim3(kThreads), 0, stream(), dy, x, gamma, save_mean,
                     save_var, partial, dx, dgamma, dbeta, N, C, HW, S, ipb, eps);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// y = bnA(xA) + bnB(xB), both in training mode from their convolutions' partials (dfhip_conv2d_forward_stats)
int dfhip_bn2d_forward_partials_pair(const float* xa, const float* partials_a, int rows_a, const float* gamma_a,
                                     const float* beta_a, float* save_mean_a, float* save_var_a,
                                     float* running_mean_a, float* running_var_a, float eps_a, float momentum_a,
                                     const float* xb, const float* partials_b, int rows_b, const float* gamma_b,
                                     const float* beta_b, float* save_mean_b, float* save_var_b,
                                     float* running_mean_b, float* running_var_b, float eps_b, float momentum_b,
                                     float* y, int N, int C, int HW) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0 && rows_a > 0 && rows_b > 0, "bn2d_forward_partials_pair: non-positive dimension");
  DF_CHECK_ARG(xa && xb && y && partials_a && partials_b, "bn2d_forward_partials_pair: null array");
  ProfScope prof("bn2d_fwd", 0.0, 12.0 * (double)N * C * HW);
  int S, ipb;
  bn_split(N, C, S, ipb);
  BnFwdSet A = {xa, partials_a, rows_a, gamma_a, beta_a, save_mean_a, save_var_a, running_mean_a, running_var_a, eps_a,
                momentum_a};
  BnFwdSet B = {xb, partials_b, rows_b, gamma_b, beta_b, save_mean_b, save_var_b, running_mean_b, running_var_b, eps_b,
                momentum_b};
  if (HW == 4 && C % 8 == 0) {
    const int ipb4 = 64, S4 = (int)df_ceil_div(N, ipb4);
    DF_LAUNCH((k_bn_apply_partials_hw4<true>), dim3((unsigned)(C / 8), (unsigned)S4), dim3(kThreads), 0, stream(), A, B,
              (const float*)nullptr, y, N, C, ipb4);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  DF_LAUNCH(k_bn_apply_partials_pair, dim3((unsigned)C, (unsigned)S), dim3(kThreads), 0, stream(), A, B, y, N,
                     C, HW, ipb);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// backward of y = bnA(xA) + bnB(xB): both receive dy
int dfhip_bn2d_backward_pair(const float* dy, const float* xa, const float* gamma_a, const float* save_mean_a,
                             const float* save_var_a, float* dxa, float* dgamma_a, float* dbeta_a, float eps_a,
                             const float* xb, const float* gamma_b, const float* save_mean_b,
                             const float* save_var_b, float* dxb, float* dgamma_b, float* dbeta_b, float eps_b,
                             int N, int C, int HW) {
  DF_INIT_KEEP();
  DF_TOUCH(dy); DF_TOUCH(xa); DF_TOUCH(xb); DF_TOUCH(gamma_a); DF_TOUCH(gamma_b);
  DF_TOUCH(dxa); DF_TOUCH(dxb); DF_TOUCH(dgamma_a); DF_TOUCH(dgamma_b); DF_TOUCH(dbeta_a); DF_TOUCH(dbeta_b);
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bn2d_backward_pair: non-positive dimension");
  DF_CHECK_ARG(dy && xa && xb && save_mean_a && save_var_a && save_mean_b && save_var_b, "bn2d_backward_pair: null array");
  if (!bn_fused_ok(N, C, HW)) {
    int st = dfhip_bn2d_backward(dy, xa, gamma_a, save_mean_a, save_var_a, dxa, dgamma_a, dbeta_a, N, C, HW, eps_a);
    if (st != DFHIP_OK) return st;
    return dfhip_bn2d_backward(dy, xb, gamma_b, save_mean_b, save_var_b, dxb, dgamma_b, dbeta_b, N, C, HW, eps_b);
  }
  ProfScope prof("bn2d_bwd", 0.0, 36.0 * (double)N * C * HW);
  BnBwdSet A = {xa, gamma_a, save_mean_a, save_var_a, dxa, dgamma_a, dbeta_a, eps_a};
  BnBwdSet B = {xb, gamma_b, save_mean_b, save_var_b, dxb, dgamma_b, dbeta_b, eps_b};
  if (bn_hw4_ok(N, C, HW)) {
    DF_LAUNCH(k_bn_fused_bwd_pair_hw4, dim3((unsigned)(C / 8), 2), dim3(512), 0, stream(), dy, A, B, N, C);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  DF_LAUNCH(k_bn_fused_bwd_pair, dim3((unsigned)C, 2), dim3(512), 0, stream(), dy, A, B, N, C, HW);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_bn_relu_maxpool2_forward(const float* x, const float* gamma, const float* beta, float* y,
                                   float* save_mean, float* save_var, float* running_mean,
                                   float* running_var, int N, int C, int H, int W, float eps, float momentum) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 7) == 0,
               "bn_relu_maxpool2_forward: needs H even and W a multiple of 8");
  DF_CHECK_ARG(x && y && save_mean && save_var, "bn_relu_maxpool2_forward: null array");
  const int HW = H * W;
  ProfScope prof("bn_relu_pool_fwd", 0.0, (12.0 + 1.0) * (double)N * C * HW);
  int S, ipb;
  bn_split(N, C, S, ipb);
  float* partial = workspace_small((size_t)C * S * 3 * sizeof(float));
  if (!partial) return DFHIP_OUT_OF_MEMORY;
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH(k_bn_stats, grid, dim3(kThreads), 0, stream(), x, partial, N, C, HW, S, ipb);
  DF_LAUNCH_CHECK();
  DF_LAUNCH(k_bn_relu_pool_fwd, grid, dim3(kThreads), 0, stream(), x, partial, S, gamma, beta, y, save_mean,
                     save_var, running_mean, running_var, N, C, H, W, ipb, eps, momentum);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// dfhip_bn_relu_maxpool2_forward with the batch statistics taken from the partials the convolution that produced x
// left behind (dfhip_conv2d_forward_stats): no statistics pass over x
int dfhip_bn_relu_maxpool2_forward_partials(const float* x, const float* partials, int rows, const float* gamma,
                                            const float* beta, float* y, float* save_mean, float* save_var,
                                            float* running_mean, float* running_var, int N, int C, int H, int W,
                                            float eps, float momentum) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 7) == 0 && rows > 0,
               "bn_relu_maxpool2_forward_partials: needs H even, W a multiple of 8, rows > 0");
  DF_CHECK_ARG(x && y && partials && save_mean && save_var, "bn_relu_maxpool2_forward_partials: null array");
  ProfScope prof("bn_relu_pool_fwd", 0.0, (4.0 + 1.0) * (double)N * C * H * W);
  int S, ipb;
  bn_split(N, C, S, ipb);
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH(k_bn_relu_pool_fwd, grid, dim3(kThreads), 0, stream(), x, partials, -rows, gamma, beta, y, save_mean,
            save_var, running_mean, running_var, N, C, H, W, ipb, eps, momentum);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_bn_relu_maxpool2_backward(const float* x, const float* y, const float* g, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_var,
                                    float* dx, float* dgamma, float* dbeta, int N, int C, int H, int W,
                                    float eps) {
  DF_INIT_KEEP();
  DF_TOUCH(x); DF_TOUCH(y); DF_TOUCH(g); DF_TOUCH(gamma); DF_TOUCH(beta); DF_TOUCH(save_mean); DF_TOUCH(save_var);
  DF_TOUCH(dx); DF_TOUCH(dgamma); DF_TOUCH(dbeta);
  DF_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 7) == 0,
               "bn_relu_maxpool2_backward: needs H even and W a multiple of 8");
  DF_CHECK_ARG(x && y && g && save_mean && save_var, "bn_relu_maxpool2_backward: null array");
  ProfScope prof("bn_relu_pool_bwd", 0.0, (12.0 + 4.0) * (double)N * C * H * W);
  int S, ipb;
  bn_split(N, C, S, ipb);
  float* partial = workspace_small((size_t)C * S * 2 * sizeof(float));
  if (!partial) return DFHIP_OUT_OF_MEMORY;
  dim3 grid((unsigned)C, (unsigned)S);
  DF_LAUNCH((k_bn_relu_pool_bwd<false>), grid, dim3(kThreads), 0, stream(), x, y, g, gamma, beta, save_mean,
                     save_var, partial, dx, dgamma, dbeta, N, C, H, W, S, ipb, eps);
  DF_LAUNCH_CHECK();
  DF_LAUNCH((k_bn_relu_pool_bwd<true>), grid, dim3(kThreads), 0, stream(), x, y, g, gamma, beta, save_mean,
                     save_var, partial, dx, dgamma, dbeta, N, C, H, W, S, ipb, eps);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_augment_batch(const float* in, float* out, const int* decisions, int N, int C, int H, int W, int pad) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && H > 1 && W > 1 && pad >= 0 && pad < H && pad < W, "augment_batch: bad geometry");
  DF_CHECK_ARG(in && out && decisions, "augment_batch: null array");
  const int64_t n = (int64_t)N * C * H * W;
  ProfScope prof("augment_batch", 0.0, 8.0 * (double)n);
  DF_LAUNCH(k_augment_batch, dim3(grid_for(n)), dim3(kThreads), 0, stream(), in, out, decisions, N, C, H, W,
                     pad);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_onehot_smooth(const int* labels, float* out, int N, int K, double eps) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && K > 0 && labels && out, "onehot_smooth: bad arguments");
  // (1 - eps) + eps / K and eps / K in the reference's float32 arithmetic: onehot * (1 - eps) + eps / K
  const float scale = (float)(1.0 - eps), add = (float)(eps / K);
  const float on = 1.0f * scale + add, off = 0.0f * scale + add;
  DF_LAUNCH(k_onehot_smooth, dim3((unsigned)df_ceil_div((int64_t)N * K, kThreads)), dim3(kThreads), 0, stream(),
                     labels, out, N, K, on, off);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_maxpool2d_forward(const float* x, float* y, int NC, int H, int W, int k) {
  DF_INIT();
  DF_CHECK_ARG(NC > 0 && H > 0 && W > 0 && k >= 1 && k <= H && k <= W, "maxpool2d: bad geometry");
  DF_CHECK_ARG(x && y, "maxpool2d: null array");
  int oh = (H - k) / k + 1, ow = (W - k) / k + 1;
  int64_t n_out = (int64_t)NC * oh * ow;
  ProfScope prof("maxpool2d_fwd", 0.0, 4.0 * ((double)NC * H * W + (double)n_out));
  DF_LAUNCH(k_maxpool_fwd, dim3(grid_for(n_out)), dim3(kThreads), 0, stream(), x, y, n_out, H,
                     W, oh, ow, k);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_maxpool2d_backward(const float* x, const float* y, const float* dy, float* dx, int NC,
                             int H, int W, int k) {
  DF_INIT_KEEP();
  DF_TOUCH(x);
  DF_TOUCH(y);
  DF_TOUCH(dy);
  DF_TOUCH(dx);
  DF_CHECK_ARG(NC > 0 && H > 0 && W > 0 && k >= 1 && k <= H && k <= W, "maxpool2d: bad geometry");
  DF_CHECK_ARG(x && y && dy && dx, "maxpool2d_backward: null array");
  int oh = (H - k) / k + 1, ow = (W - k) / k + 1;
  int64_t n_in = (int64_t)NC * H * W;
  ProfScope prof("maxpool2d_bwd", 0.0, 8.0 * (double)n_in + 8.0 * (double)NC * oh * ow);
  DF_LAUNCH(k_maxpool_bwd, dim3(grid_for(n_in)), dim3(kThreads), 0, stream(), x, y, dy, dx,
                     n_in, H, W, oh, ow, k);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_dropout_apply(const float* x, const float* mask, float* y, int64_t n, float keep, int backward) {
  DF_INIT_KEEP();
  DF_TOUCH(x); DF_TOUCH(mask); DF_TOUCH(y);
  DF_CHECK_ARG(n >= 0 && keep > 0.0f, "dropout_apply: bad arguments");
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(x && mask && y, "dropout_apply: null array");
  ProfScope prof(backward ? "dropout_bwd" : "dropout_fwd", 0.0, 12.0 * (double)n);
  if (backward) DF_LAUNCH(k_dropout<true>, dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), x, mask, y, n, keep);
  else DF_LAUNCH(k_dropout<false>, dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), x, mask, y, n, keep);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_relu_forward(const float* x, float* y, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0, "relu: negative size");
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(x && y, "relu: null array");
  ProfScope prof("relu_fwd", 0.0, 8.0 * (double)n);
  DF_LAUNCH(k_relu_fwd, dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), x, y, n);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_relu_backward(const float* x, const float* y, const float* g, float* dx, int64_t n) {
  DF_INIT_KEEP();
  DF_TOUCH(x);
  DF_TOUCH(y);
  DF_TOUCH(g);
  DF_TOUCH(dx);
  DF_CHECK_ARG(n >= 0, "relu_backward: negative size");
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(x && y && g && dx, "relu_backward: null array");
  ProfScope prof("relu_bwd", 0.0, 16.0 * (double)n);
  DF_LAUNCH(k_relu_bwd, dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), x, y, g, dx, n);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_softmax_ce_forward(const float* logits, const float* target, float* loss, int rows,
                             int cols) {
  DF_INIT();
  DF_CHECK_ARG(rows > 0 && cols > 0, "softmax_ce: non-positive dimension");
  DF_CHECK_ARG(logits && target && loss, "softmax_ce: null array");
  ProfScope prof("softmax_ce_fwd", 0.0, 8.0 * (double)rows * cols);
  if (rows <= 4096 && cols <= 32) {
    if (cols <= 16)
      DF_LAUNCH((k_ce_small<false, 16>), dim3(1), dim3(1024), 0, stream(), logits, target, loss, (float*)nullptr, rows,
                cols, (float)(1.0 / (double)rows));
    else
      DF_LAUNCH((k_ce_small<false, 32>), dim3(1), dim3(1024), 0, stream(), logits, target, loss, (float*)nullptr, rows,
                cols, (float)(1.0 / (double)rows));
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  float* row_loss = workspace_small((size_t)rows * sizeof(float));
  if (!row_loss) return DFHIP_OUT_OF_MEMORY;
  DF_LAUNCH(k_ce_rows, dim3(grid_for((int64_t)rows * 64)), dim3(kThreads), 0, stream(), logits,
                     target, row_loss, rows, cols);
  DF_LAUNCH_CHECK();
  float inv = (float)(1.0 / (double)rows);
  DF_LAUNCH(k_ce_finish, dim3(1), dim3(kThreads), 0, stream(), row_loss, loss, rows, inv);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// loss as dfhip_softmax_ce_forward, plus dlogits = d loss / d logits for an upstream gradient of 1 (the seed of
// loss.backward(), tensor.py:428): *done = 0 (dlogits untouched) for shapes outside the one-workgroup kernel
int dfhip_softmax_ce_forward_grad(const float* logits, const float* target, float* loss, float* dlogits, int rows,
                                  int cols, int* done) {
  DF_INIT();
  DF_CHECK_ARG(rows > 0 && cols > 0, "softmax_ce: non-positive dimension");
  DF_CHECK_ARG(logits && target && loss && dlogits && done, "softmax_ce_forward_grad: null array");
  *done = 0;
  if (!(rows <= 4096 && cols <= 32)) return dfhip_softmax_ce_forward(logits, target, loss, rows, cols);
  ProfScope prof("softmax_ce_fwd", 0.0, 12.0 * (double)rows * cols);
  if (cols <= 16)
    DF_LAUNCH((k_ce_small<true, 16>), dim3(1), dim3(1024), 0, stream(), logits, target, loss, dlogits, rows, cols,
              (float)(1.0 / (double)rows));
  else
    DF_LAUNCH((k_ce_small<true, 32>), dim3(1), dim3(1024), 0, stream(), logits, target, loss, dlogits, rows, cols,
              (float)(1.0 / (double)rows));
  DF_LAUNCH_CHECK();
  *done = 1;
  return DFHIP_OK;
}

// y (N, P) = (scale * sum over the HW pixels of x (N, C, HW)) . w (C, P) + b (P or NULL); pooled (N, C) is kept for
// the backward pass.  P <= 16.  Replaces tensor.mean(axis=2) twice (tensor.py:752-769 via backend_tensor.py:659-662,
// `scale` carries their divide-by-total-count rule) followed by nn/modules/linear.py:47-51.
int dfhip_pool_linear_forward(const float* x, const float* w, const float* b, float* pooled, float* y, int N, int C,
                              int HW, int P, float scale) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0 && P > 0 && P <= 16, "pool_linear_forward: bad dimensions (P <= 16)");
  DF_CHECK_ARG(x && w && pooled && y, "pool_linear_forward: null array");
  ProfScope prof("pool_linear_fwd", 2.0 * N * (double)C * P, 4.0 * ((double)N * C * HW + (double)N * C + (double)C * P));
  DF_LAUNCH(k_pool_linear_fwd<16>, dim3((unsigned)N), dim3(256), 0, stream(), x, w, b, pooled, y, C, HW, P, scale);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// gradients of the above from dy (N, P): dx (N, C, HW) = gscale * (dy . w^T) broadcast over the pixels, dw (C, P) =
// pooled^T . dy, db (P) = column sums of dy; any of the three may be NULL.  One launch.
int dfhip_pool_linear_backward(const float* dy, const float* pooled, const float* w, float* dx, float* dw, float* db,
                               int N, int C, int HW, int P, float gscale) {
  DF_INIT();
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0 && P > 0 && P <= 16, "pool_linear_backward: bad dimensions (P <= 16)");
  DF_CHECK_ARG(dy && pooled && w, "pool_linear_backward: null array");
  ProfScope prof("pool_linear_bwd", 4.0 * N * (double)C * P, 4.0 * ((double)N * C * HW + 2.0 * N * C + 2.0 * C * P));
  const unsigned grid = (unsigned)((N + 3) / 4 + (C + 63) / 64 + 1);
  DF_LAUNCH(k_pool_linear_bwd<16>, dim3(grid), dim3(1024), 0, stream(), dy, pooled, w, dx, dw, db, N, C, HW, P, gscale);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_head_ce_forward(const float* x, const float* w, const float* b, const float* target, float* pooled,
                          float* logits, float* loss, float* dlogits, float* dx, float* rowloss, int N, int C, int HW,
                          int P, float scale, float gscale, int* done) {
  DF_INIT();
  DF_CHECK_ARG(done, "head_ce_forward: null done");
  *done = 0;
  DF_CHECK_ARG(N > 0 && C > 0 && HW > 0 && P > 0, "head_ce_forward: bad dimensions");
  DF_CHECK_ARG(x && w && target && pooled && logits && loss && dlogits && rowloss, "head_ce_forward: null array");
  if (!(P <= 16 && C <= 512 && N <= 4096)) return DFHIP_OK;       // (the caller launches the three nodes)
  ProfScope prof("head_ce_fwd_bwd", 4.0 * N * (double)C * P,
                 4.0 * ((dx ? 2.0 : 1.0) * N * (double)C * HW + (double)N * C + (double)C * P + 3.0 * N * P));
  const float inv_rows = 1.0f / (float)N;
  constexpr int IPB = 1;
  const dim3 grid((unsigned)((N + IPB - 1) / IPB));
  if (C <= 256)
    DF_LAUNCH((k_head_ce<16, 1, IPB>), grid, dim3(256 * IPB), 0, stream(), x, w, b, target, pooled, logits, dlogits, dx,
              rowloss, loss, N, C, HW, P, scale, gscale, inv_rows, (const float*)nullptr, 1.0f);
  else
    DF_LAUNCH((k_head_ce<16, 2, IPB>), grid, dim3(256 * IPB), 0, stream(), x, w, b, target, pooled, logits, dlogits, dx,
              rowloss, loss, N, C, HW, P, scale, gscale, inv_rows, (const float*)nullptr, 1.0f);
  DF_LAUNCH_CHECK();
  *done = 1;
  return DFHIP_OK;
}

int dfhip_droplin_ce_forward(const float* x, const float* mask, float keep, const float* w, const float* b, const float* target,
                             float* xd, float* logits, float* loss, float* dlogits, float* dx, float* rowloss, int N, int C,
                             int P, int* done) {
  DF_INIT();
  DF_CHECK_ARG(done, "droplin_ce_forward: null done");
  *done = 0;
  DF_CHECK_ARG(N > 0 && C > 0 && P > 0 && keep > 0.0f, "droplin_ce_forward: bad dimensions");
  DF_CHECK_ARG(x && mask && w && target && xd && logits && loss && dlogits && rowloss, "droplin_ce_forward: null array");
  if (!(P <= 16 && C <= 2048 && N <= 4096)) return DFHIP_OK;      // (the caller launches the nodes one by one)
  ProfScope prof("droplin_ce_fwd_bwd", 4.0 * N * (double)C * P,
                 4.0 * ((dx ? 4.0 : 3.0) * N * (double)C + (double)C * P + 3.0 * N * P));
  const float inv_rows = 1.0f / (float)N;
  const dim3 grid((unsigned)N);
  const size_t lds = (size_t)C * P * sizeof(float);
#define DF_DROPLIN(CPT)                                                                                                    \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set) {                                                                                                       \
      DF_HIP(hipFuncSetAttribute((const void*)k_head_ce<16, CPT, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                 2048 * 16 * (int)sizeof(float)));                                                         \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    DF_LAUNCH((k_head_ce<16, CPT, 1, true>), grid, dim3(256), lds, stream(), x, w, b, target, xd, logits, dlogits, dx,     \
              rowloss, loss, N, C, 1, P, 1.0f, 1.0f, inv_rows, mask, keep);                                                 \
  } while (0)
  if (C <= 256) DF_DROPLIN(1);
  else if (C <= 512) DF_DROPLIN(2);
  else if (C <= 1024) DF_DROPLIN(4);
  else DF_DROPLIN(8);
#undef DF_DROPLIN
  DF_LAUNCH_CHECK();
  *done = 1;
  return DFHIP_OK;
}

int dfhip_loss_finish(const float* rowloss, float* loss, int N) {
  DF_INIT();
  DF_CHECK_ARG(rowloss && loss && N > 0, "loss_finish: bad arguments");
  ProfScope prof("loss_finish", 0.0, 4.0 * N);
  DF_LAUNCH(k_loss_finish, dim3(1), dim3(256), 0, stream(), rowloss, loss, N, 1.0f / (float)N);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_head_grads(const float* dlogits, const float* pooled, float* dw, float* db, const float* rowloss, float* loss,
                     int N, int C, int P) {
  DF_INIT_KEEP();
  DF_TOUCH(dw); DF_TOUCH(db); DF_TOUCH(loss);
  DF_CHECK_ARG(N > 0 && C > 0 && P > 0 && P <= dfhip::kHeadP, "head_grads: bad dimensions (P <= 16)");
  DF_CHECK_ARG(dlogits && pooled && (!loss || rowloss), "head_grads: null array");
  if (!dw && !db && !loss) return DFHIP_OK;
  dfhip::HeadGrads h = {dlogits, pooled, dw, db, rowloss, loss, 1.0f / (float)N, N, C, P};
  if (dfhip::head_grads_defer(h)) return DFHIP_OK;      // extra workgroups of the open group's slab-sum launch
  return dfhip::head_grads_launch(h);
}

int dfhip_softmax_ce_backward(const float* logits, const float* target, const float* gloss,
                              float* dlogits, int rows, int cols) {
  DF_INIT();
  DF_CHECK_ARG(rows > 0 && cols > 0, "softmax_ce_backward: non-positive dimension");
  DF_CHECK_ARG(logits && target && gloss && dlogits, "softmax_ce_backward: null array");
  ProfScope prof("softmax_ce_bwd", 0.0, 12.0 * (double)rows * cols);
  float inv = (float)(1.0 / (double)rows);
  DF_LAUNCH(k_ce_bwd, dim3(grid_for((int64_t)rows * 64)), dim3(kThreads), 0, stream(), logits,
                     target, gloss, dlogits, rows, cols, inv);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_reduce_to_shape(const float* src, float* dst, int src_ndim, const int64_t* src_shape,
                          int dst_ndim, const int64_t* dst_shape) {
  // (a convolution's bias gradient in the middle of a backward sweep: what the sweep has deferred -- queued weight gradients,
  // the head's sums waiting for the group's slab-sum launch -- stays deferred unless it touches these arrays)
  DF_INIT_KEEP();
  DF_TOUCH(src);
  DF_TOUCH(dst);
  DF_CHECK_ARG(src_ndim >= 1 && src_ndim <= DFHIP_MAX_DIMS && dst_ndim >= 0 && dst_ndim <= src_ndim,
               "reduce_to_shape: bad ranks %d -> %d", src_ndim, dst_ndim);
  DF_CHECK_ARG(src && dst, "reduce_to_shape: null array");
  RShape rs;
  rs.nd = src_ndim;
  int64_t n_dst = 1, n_red = 1;
  int lead = src_ndim - dst_ndim;
  for (int a = 0; a < src_ndim; a++) {
    rs.src[a] = src_shape[a];
    int64_t d = a < lead ? 1 : dst_shape[a - lead];
    DF_CHECK_ARG(d == src_shape[a] || d == 1, "reduce_to_shape: axis %d: %lld does not broadcast to %lld",
                 a, (long long)d, (long long)src_shape[a]);
    rs.keep[a] = (a >= lead && d == src_shape[a] && d != 1) ? 1 : 0;
    if (rs.keep[a]) n_dst *= src_shape[a];
    else n_red *= src_shape[a];
  }
  if (n_dst * n_red == 0) return DFHIP_OK;
  const bool channel_sum = src_ndim == 4 && rs.keep[0] == 0 && rs.keep[1] == 1 && rs.keep[2] == 0 && rs.keep[3] == 0 &&
                           n_red >= 4096 && src_shape[1] < (1 << 20);
  // (a queued layer's bias gradient: workgroups of the sweep's slab-sum launch, no launch here)
  if (channel_sum && bias_sum_defer(src, dst, (int)src_shape[0], (int)src_shape[1], (int)(src_shape[2] * src_shape[3])))
    return DFHIP_OK;
  ProfScope prof("reduce_to_shape", 0.0, 4.0 * (double)(n_dst * n_red + n_dst));
  if (channel_sum) {
    const int N = (int)src_shape[0], C = (int)src_shape[1], HW = (int)(src_shape[2] * src_shape[3]);
    int S, ipb;
    bn_split(N, C, S, ipb);
    float* partial = workspace_small((size_t)C * S * sizeof(float));
    if (!partial) return DFHIP_OUT_OF_MEMORY;
    DF_TOUCH(partial);
    DF_LAUNCH(k_channel_sum, dim3((unsigned)C, (unsigned)S), dim3(kThreads), 0, stream(), src, partial, N, C,
                       HW, S, ipb);
    DF_LAUNCH_CHECK();
    DF_LAUNCH(k_channel_sum_finish, dim3((unsigned)df_ceil_div(C, kThreads)), dim3(kThreads), 0, stream(),
                       partial, dst, C, S);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  int blocks = (int)std::min<int64_t>(n_dst, (int64_t)num_cus() * 8);
  DF_LAUNCH(k_reduce_to_shape, dim3(blocks), dim3(kThreads), 0, stream(), src, dst, rs, n_dst,
                     n_red);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_reduce_axis(const float* src, float* dst, int64_t outer, int64_t red, int64_t inner, int op,
                      float scale) {
  DF_INIT();
  DF_CHECK_ARG(outer >= 0 && red > 0 && inner >= 0, "reduce_axis: bad extents");
  DF_CHECK_ARG(op == 0 || op == 1, "reduce_axis: op must be 0 (sum) or 1 (max)");
  int64_t n = outer * inner;
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(src && dst, "reduce_axis: null array");
  ProfScope prof("reduce_axis", 0.0, 4.0 * (double)(n * red + n));
  if (op == 0)
    DF_LAUNCH((k_reduce_axis<false>), dim3(grid_for(n)), dim3(kThreads), 0, stream(), src, dst,
                       outer, red, inner, scale);
  else
    DF_LAUNCH((k_reduce_axis<true>), dim3(grid_for(n)), dim3(kThreads), 0, stream(), src, dst,
                       outer, red, inner, scale);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_broadcast_axis(const float* src, float* dst, int64_t outer, int64_t red, int64_t inner,
                         float scale) {
  DF_INIT();
  DF_CHECK_ARG(outer >= 0 && red > 0 && inner >= 0, "broadcast_axis: bad extents");
  int64_t n = outer * red * inner;
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(src && dst, "broadcast_axis: null array");
  ProfScope prof("broadcast_axis", 0.0, 4.0 * (double)(n + outer * inner));
  DF_LAUNCH(k_broadcast_axis, dim3(grid_for(n)), dim3(kThreads), 0, stream(), src, dst, outer,
                     red, inner, scale);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_axpy(float* dst, const float* src, float alpha, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0, "axpy: negative size");
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(dst && src, "axpy: null array");
  ProfScope prof("axpy", 0.0, 12.0 * (double)n);
  DF_LAUNCH(k_axpy, dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), dst, src, alpha, n);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_adam_step(int n_tensors, float* const* params, const float* const* grads,
                    float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int step,
                    double grad_scale) {
  DF_INIT();
  DF_CHECK_ARG(n_tensors >= 0 && step >= 1, "adam_step: bad arguments");
  if (n_tensors == 0) return DFHIP_OK;
  DF_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes, "adam_step: null table");
  // scalars are rounded to fp32 exactly where the reference's NumPy device rounds them
  // (np.float32(python_float) in backend_tensor.py:109-125)
  AdamHyper h;
  h.lr = (float)lr;
  h.b1 = (float)beta1;
  h.b2 = (float)beta2;
  h.omb1 = (float)(1.0 - beta1);
  h.omb2 = (float)(1.0 - beta2);
  h.bc1 = (float)(1.0 - pow(beta1, (double)step));
  h.bc2 = (float)(1.0 - pow(beta2, (double)step));
  h.eps = (float)eps;
  h.wd = (float)weight_decay;
  h.gscale = (float)grad_scale;
  double total = 0;
  for (int i = 0; i < n_tensors; i++) total += (double)sizes[i];
  ProfScope prof("adam_step", 0.0, 28.0 * total);
  for (int first = 0; first < n_tensors; first += kBatch) {
    int n = std::min(kBatch, n_tensors - first);
    TensorBatch tb;
    int blocks = fill_batch(tb, n, first, params, grads, exp_avg, exp_avg_sq, sizes);
    DF_LAUNCH(k_adam, dim3(blocks), dim3(kThreads), 0, stream(), tb, h);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

static AdamHyper make_adam_hyper(double lr, double beta1, double beta2, double eps, double weight_decay,
                                 int step, double grad_scale) {
  // scalars are rounded to fp32 exactly where the reference's NumPy device rounds them
  AdamHyper h;
  h.lr = (float)lr;
  h.b1 = (float)beta1;
  h.b2 = (float)beta2;
  h.omb1 = (float)(1.0 - beta1);
  h.omb2 = (float)(1.0 - beta2);
  h.bc1 = (float)(1.0 - pow(beta1, (double)step));
  h.bc2 = (float)(1.0 - pow(beta2, (double)step));
  h.eps = (float)eps;
  h.wd = (float)weight_decay;
  h.gscale = (float)grad_scale;
  return h;
}

int dfhip_adam_hyper(float* hyper_dev, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int step, double grad_scale) {
  DF_INIT();
  DF_CHECK_ARG(hyper_dev && step >= 1, "adam_hyper: bad arguments");
  AdamHyper h = make_adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  DF_LAUNCH(k_set_adam_hyper, dim3(1), dim3(1), 0, stream(), hyper_dev, h);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_adam_hyper_table(float* hyper_dev, int cap, double lr, double beta1, double beta2, double eps,
                           double weight_decay, int first_step, double grad_scale) {
  DF_INIT();
  DF_CHECK_ARG(hyper_dev && cap >= 1 && first_step >= 1, "adam_hyper_table: bad arguments");
  if (capturing()) {
    set_error("adam_hyper_table: a host-to-device copy cannot be part of a captured step");
    return DFHIP_INVALID_ARGUMENT;
  }
  std::vector<float> host((size_t)16 + 16 * (size_t)cap, 0.0f);
  int* ctl = reinterpret_cast<int*>(host.data());
  ctl[0] = 0; ctl[1] = 0; ctl[2] = cap;
  for (int i = 0; i < cap; i++) {
    const AdamHyper h = make_adam_hyper(lr, beta1, beta2, eps, weight_decay, first_step + i, grad_scale);
    memcpy(host.data() + 16 + 16 * (size_t)i, &h, sizeof(h));
  }
  // rare (once per `cap` steps, or when a scalar changes): the stream is drained, then a plain copy
  DF_HIP(hipStreamSynchronize(stream()));
  DF_HIP(hipMemcpy(hyper_dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return DFHIP_OK;
}

int dfhip_adam_hyper_index(const float* hyper_dev, int* index, int* cap) {
  DF_INIT();
  DF_CHECK_ARG(hyper_dev && index && cap, "adam_hyper_index: null argument");
  if (capturing()) {
    set_error("adam_hyper_index: a device-to-host copy cannot be part of a captured step");
    return DFHIP_INVALID_ARGUMENT;
  }
  int ctl[4] = {0, 0, 0, 0};
  DF_HIP(hipStreamSynchronize(stream()));
  DF_HIP(hipMemcpy(ctl, hyper_dev, sizeof(ctl), hipMemcpyDeviceToHost));
  *index = ctl[0];
  *cap = ctl[2];
  return DFHIP_OK;
}

int dfhip_adam_step_dev(int n_tensors, float* const* params, const float* const* grads,
                        float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                        const float* hyper_dev) {
  DF_INIT();
  DF_CHECK_ARG(n_tensors >= 0 && hyper_dev, "adam_step_dev: bad arguments");
  if (n_tensors == 0) return DFHIP_OK;
  DF_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes, "adam_step_dev: null table");
  double total = 0;
  for (int i = 0; i < n_tensors; i++) total += (double)sizes[i];
  ProfScope prof("adam_step", 0.0, 28.0 * total);
  for (int first = 0; first < n_tensors; first += kBatch) {
    int n = std::min(kBatch, n_tensors - first);
    TensorBatch tb;
    int blocks = fill_batch(tb, n, first, params, grads, exp_avg, exp_avg_sq, sizes);
    DF_LAUNCH(k_adam_dev, dim3(blocks), dim3(kThreads), 0, stream(), tb, const_cast<float*>(hyper_dev),
              first + kBatch >= n_tensors ? 1 : 0);          // the last launch of the step moves the table on
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

static SgdHyper make_sgd_hyper(double lr, double momentum, double weight_decay, int nesterov, double grad_scale) {
  SgdHyper h;
  h.lr = (float)lr;
  h.mu = (float)momentum;
  h.wd = (float)weight_decay;
  h.gscale = (float)grad_scale;
  h.nesterov = nesterov;
  h.use_momentum = momentum > 0.0;
  return h;
}

int dfhip_sgd_hyper(float* hyper_dev, double lr, double momentum, double weight_decay, int nesterov,
                    double grad_scale) {
  DF_INIT();
  DF_CHECK_ARG(hyper_dev, "sgd_hyper: null array");
  if (capturing()) {
    set_error("sgd_hyper: the scalars are written in front of a captured step, not inside it");
    return DFHIP_INVALID_ARGUMENT;
  }
  const SgdHyper h = make_sgd_hyper(lr, momentum, weight_decay, nesterov, grad_scale);
  DF_LAUNCH(k_set_sgd_hyper, dim3(1), dim3(1), 0, stream(), hyper_dev, h);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_sgd_step_dev(int n_tensors, float* const* params, const float* const* grads,
                       float* const* momentum_buf, const int64_t* sizes, const float* hyper_dev) {
  DF_INIT_KEEP();
  if (hyper_dev) {        // (a perceptron's deferred backward launch takes the update over exactly its gradients along)
    const int took = mlp_bwd_take_sgd(n_tensors, params, grads, momentum_buf, sizes, nullptr, hyper_dev);
    if (took) return took < 0 ? DFHIP_RUNTIME_ERROR : DFHIP_OK;
  }
  DF_INIT();
  DF_CHECK_ARG(n_tensors >= 0 && hyper_dev, "sgd_step_dev: bad arguments");
  if (n_tensors == 0) return DFHIP_OK;
  DF_CHECK_ARG(params && grads && sizes, "sgd_step_dev: null table");
  double total = 0;
  for (int i = 0; i < n_tensors; i++) total += (double)sizes[i];
  ProfScope prof("sgd_step", 0.0, (momentum_buf ? 20.0 : 12.0) * total);
  for (int first = 0; first < n_tensors; first += kBatch) {
    int n = std::min(kBatch, n_tensors - first);
    TensorBatch tb;
    int blocks = fill_batch(tb, n, first, params, grads, momentum_buf, nullptr, sizes);
    DF_LAUNCH(k_sgd_dev, dim3(blocks), dim3(kThreads), 0, stream(), tb, hyper_dev);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

int dfhip_sgd_step(int n_tensors, float* const* params, const float* const* grads,
                   float* const* momentum_buf, const int64_t* sizes, double lr, double momentum,
                   double weight_decay, int nesterov, double grad_scale) {
  DF_INIT_KEEP();
  const SgdHyper h = make_sgd_hyper(lr, momentum, weight_decay, nesterov, grad_scale);
  if (momentum <= 0.0 || momentum_buf) {
    const int took = mlp_bwd_take_sgd(n_tensors, params, grads, momentum_buf, sizes, &h, nullptr);
    if (took) return took < 0 ? DFHIP_RUNTIME_ERROR : DFHIP_OK;
  }
  DF_INIT();
  DF_CHECK_ARG(n_tensors >= 0, "sgd_step: bad arguments");
  if (n_tensors == 0) return DFHIP_OK;
  DF_CHECK_ARG(params && grads && sizes, "sgd_step: null table");
  DF_CHECK_ARG(momentum <= 0.0 || momentum_buf, "sgd_step: momentum needs buffers");
  double total = 0;
  for (int i = 0; i < n_tensors; i++) total += (double)sizes[i];
  ProfScope prof("sgd_step", 0.0, (h.use_momentum ? 20.0 : 12.0) * total);
  for (int first = 0; first < n_tensors; first += kBatch) {
    int n = std::min(kBatch, n_tensors - first);
    TensorBatch tb;
    int blocks = fill_batch(tb, n, first, params, grads, h.use_momentum ? momentum_buf : nullptr,
                            nullptr, sizes);
    DF_LAUNCH(k_sgd, dim3(blocks), dim3(kThreads), 0, stream(), tb, h);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

static int pack_impl(int n_tensors, float* const* dsts, const float* const* srcs,
                     const int64_t* sizes, float* flat, int dir) {
  DF_INIT();
  DF_CHECK_ARG(n_tensors >= 0, "pack: bad arguments");
  if (n_tensors == 0) return DFHIP_OK;
  DF_CHECK_ARG(flat && sizes, "pack: null table");
  double total = 0;
  for (int i = 0; i < n_tensors; i++) total += (double)sizes[i];
  ProfScope prof(dir == 0 ? "bucket_pack" : "bucket_unpack", 0.0, 8.0 * total);
  int64_t base = 0;
  for (int first = 0; first < n_tensors; first += kBatch) {
    int n = std::min(kBatch, n_tensors - first);
    TensorBatch tb;
    int blocks = fill_batch(tb, n, first, dsts, srcs, nullptr, nullptr, sizes);
    int64_t off = 0;
    for (int i = 0; i < n; i++) {
      tb.s2[i] = (float*)(intptr_t)off;  // element offset inside this batch's span
      off += sizes[first + i];
    }
    DF_LAUNCH(k_pack, dim3(blocks), dim3(kThreads), 0, stream(), tb, flat, base, dir);
    DF_LAUNCH_CHECK();
    base += off;
  }
  return DFHIP_OK;
}

int dfhip_pack(int n_tensors, const float* const* srcs, const int64_t* sizes, float* flat) {
  return pack_impl(n_tensors, nullptr, srcs, sizes, flat, 0);
}

int dfhip_unpack(int n_tensors, float* const* dsts, const int64_t* sizes, const float* flat) {
  return pack_impl(n_tensors, dsts, nullptr, sizes, const_cast<float*>(flat), 1);
}

}  // extern "C"
