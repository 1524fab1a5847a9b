// Two-layer perceptron training step, one launch each way: logits = relu(x . W1 + b1) . W2 + b2, loss = CE(logits, target)
// -- test/MLP_MNIST_cuda.py:37-59 (fc1 -> F.relu -> fc2, 784 -> 256 -> 10 at batch 128; BASELINE.json config 2) -- and the
// whole backward pass of it.
//
// Reference being replaced: nn/functional.py:8-12 (linear), :15-16 (relu = tensor.maximum, tensor.py:872-877: the gradient
// passes where y == x, i.e. also at 0), :104-115 (cross_entropy), tensor.py:699-716 (matmul gradients), :462-483 (the bias
// un-broadcast).  Rounds 1-5 ran it as six launches of 4-12 us each (47 us per step): every launch sits on the ~4 us floor of
// a dependent dispatch, the GEMMs are 0.05 GFLOP.
//
// Forward (k_mlp2_fwd): the ROWS of the batch are independent through fc1 -> relu -> fc2 -> CE -> d loss / d logits ->
// d loss / d (fc1 output), so a workgroup owns FOUR rows and all of the hidden layer: v_mfma_f32_4x4x1 with its sixteen
// 4 x 4 blocks laid side by side is a (4 rows) x (64 hidden units) product per instruction whose B operand -- W1[k][64 w ..
// 64 w + 63] -- is one coalesced 256-byte load per k.  Every workgroup streams all of W1 (803 KB out of L2) -- that, not
// the arithmetic, is what the launch takes --, so the K range is split over four wave groups (sixteen waves: more loads in
// flight) that meet through LDS.  What crosses workgroups is left for the backward launch: the mean of the row losses.
// Backward (k_mlp2_bwd): dW1 = x^T . dz1 has the BATCH as its reduction index, which a workgroup holds completely (16 input
// features x all hidden units per workgroup, v_mfma_f32_16x16x4): no partial sums across workgroups, no slabs; one more
// workgroup forms dW2, db1, db2, another the loss.
#include <stdlib.h>

#include <vector>

#include "dfhip_internal.h"
#include "dfhip_head.h"

using namespace dfhip;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kP = 16;        // compile-time bound of the class count
constexpr int kKS = 4;        // wave groups the K range of fc1 is split over

// NW: hidden units / 64 (waves per K group); R: k per round of the fc1 loop
template <int NW, int R>
__global__ __launch_bounds__(64 * NW * kKS) void k_mlp2_fwd(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ target, float* __restrict__ zpre, float* __restrict__ h,
    float* __restrict__ logits,
    float* __restrict__ dlogits, float* __restrict__ dz1, float* __restrict__ rowloss, int M, int K, int Pn, float inv_rows) {
  constexpr int HD = 64 * NW, P = kP, NWAVES = NW * kKS;
  __shared__ float red[kKS - 1][4][HD];            // fc1 partial sums of K groups 1 .. 3
  __shared__ float hs[4][HD];                      // relu(fc1) of the four rows
  __shared__ float sw2[HD][P];                     // W2, padded to P columns (zeros)
  __shared__ float plog[NWAVES][4][P];             // fc2 partial sums per wave
  __shared__ float sdl[4][P];                      // d loss / d logits of the four rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / NW, wc = wave - ks * NW;   // K group, column group
  const int r0 = (int)blockIdx.x * 4;
  const int col = 64 * wc + lane;
  // W2 into LDS (off the critical path: requested first, committed behind the fc1 loop)
  constexpr int W2V = (HD * P + 64 * NWAVES - 1) / (64 * NWAVES);
  float w2v[W2V];
#pragma unroll
  for (int i = 0; i < W2V; i++) {
    const int e = tid + 64 * NWAVES * i, c = e / P, j = e - c * P;
    const float v = w2[(int64_t)min(c, HD - 1) * Pn + min(j, Pn - 1)];
    w2v[i] = v * ((c < HD && j < Pn) ? 1.0f : 0.0f);
  }
  // the four rows of x into LDS once (every wave multiplies them; from global memory each of the sixteen waves re-read them:
  // 3 136 more load instructions per workgroup through the address unit)
  extern __shared__ __attribute__((aligned(16))) float xs[];       // [4][K]
  {
    const int k4 = K / 4;
    for (int e = tid; e < 4 * k4; e += 64 * NWAVES) {
      const int r = e / k4, q = e - r * k4;
      reinterpret_cast<float4*>(xs)[e] = *reinterpret_cast<const float4*>(x + (int64_t)min(r0 + r, M - 1) * K + 4 * q);
    }
  }
  __syncthreads();
  // fc1: this wave's K range, 16 k per round, one round ahead in a second register set.  The W1 operand is
  // (scalar row address)[lane]: no vector instruction per load; the x operand is a ds_read_b128 at a constant lane offset
  const int kq = K / kKS;                          // (K % 16 == 0: a multiple of 4)
  const int kb = ks * kq;
  const float* xl = xs + (lane & 3) * K + kb;
  const float* wb = w1 + (int64_t)kb * HD + 64 * wc;           // (uniform)
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = {0.0f, 0.0f, 0.0f, 0.0f};
  // rounds of R k (R = 28 when the range is a multiple of it -- 784 / 4 = 7 x 28 --, else 32 with the last round clamped:
  // a k past the end re-reads the last row of W1 and multiplies it by 0)
  const int rounds = (kq + R - 1) / R;
  float4 xa[R / 4], xn[R / 4];
  float wa[R], wn[R];
  auto request = [&](float4 (&xv)[R / 4], float (&wv)[R], int k0) {
#pragma unroll
    for (int i = 0; i < R; i++) wv[i] = (wb + (int64_t)min(k0 + i, kq - 1) * HD)[lane];
#pragma unroll
    for (int q = 0; q < R / 4; q++) {
      const int k = k0 + 4 * q;                      // (kq % 4 == 0: a float4 is inside the range or outside it)
      const float4 v = *reinterpret_cast<const float4*>(xl + min(k, kq - 4));
      const float on = k < kq ? 1.0f : 0.0f;
      xv[q] = make_float4(v.x * on, v.y * on, v.z * on, v.w * on);
    }
  };
  auto multiply = [&](const float4 (&xv)[R / 4], const float (&wv)[R]) {
#pragma unroll
    for (int q = 0; q < R / 4; q++) {              // four independent accumulator chains
      acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[q].x, wv[4 * q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[q].y, wv[4 * q + 1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[q].z, wv[4 * q + 2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[q].w, wv[4 * q + 3], acc[3], 0, 0, 0);
    }
  };
  request(xa, wa, 0);
  int it = 0;
#pragma unroll 1
  for (; it + 2 <= rounds; it += 2) {              // (two rounds per iteration: the sets swap roles, no register copies)
    request(xn, wn, R * (it + 1));
    asm volatile("" ::: "memory");                 // (a compiler barrier: the selection DAG sinks plain loads to their uses)
    __builtin_amdgcn_sched_barrier(0);             // (the next round's loads are requested before this round's MFMAs wait)
    multiply(xa, wa);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 2 < rounds) request(xa, wa, R * (it + 2));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    multiply(xn, wn);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (it < rounds) multiply(xa, wa);
  f32x4 z;                                         // rows r0 .. r0 + 3 of hidden unit `col`, this K group's share
#pragma unroll
  for (int r = 0; r < 4; r++) z[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
#pragma unroll
  for (int i = 0; i < W2V; i++) {
    const int e = tid + 64 * NWAVES * i;
    if (e < HD * P) (&sw2[0][0])[e] = w2v[i];
  }
  if (ks > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) red[ks - 1][r][col] = z[r];
  }
  __syncthreads();
  float hv[4];
  bool pass[4];
  float w2r[P];
  if (ks == 0) {
    const float bias = b1 ? b1[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float s = z[r];
#pragma unroll
      for (int g = 0; g < kKS - 1; g++) s += red[g][r][col];
      s += bias;
      const float y = relu_point(s);               // maximum(z, 0): NaN stays NaN (np.maximum)
      hv[r] = y;
      pass[r] = y == s;                            // tensor.py:872-877: the gradient passes where y == x (also at 0)
      hs[r][col] = y;
      if (r0 + r < M) {
        h[(int64_t)(r0 + r) * HD + col] = y;
        if (zpre) zpre[(int64_t)(r0 + r) * HD + col] = s;
      }
    }
#pragma unroll
    for (int j = 0; j < P; j++) w2r[j] = sw2[col][j];
  }
  __syncthreads();
  // fc2: wave w adds up the hidden units [cb, cb + HD / NWAVES) for (row, class) = (lane / 16, lane % 16)
  {
    constexpr int CPW = HD / NWAVES;
    const int r = lane >> 4, j = lane & 15, cb = wave * CPW;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CPW; c++) s = fmaf(hs[r][cb + c], sw2[cb + c][j], s);
    plog[wave][r][j] = s;
  }
  __syncthreads();
  if (wave == 0) {
    // one wave: lane (row, class).  Cross-entropy of the four rows with the expressions of k_ce_small<true>
    // (dfhip_fused.hip) -- the sums over the classes run over the sixteen lanes of a row
    const int r = lane >> 4, j = lane & 15;
    const int row = min(r0 + r, M - 1);
    float y = 0.0f;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) y += plog[w][r][j];
    y += (b2 && j < Pn) ? b2[j] : 0.0f;
    const float xv = j < Pn ? y : -INFINITY;
    const float tv = j < Pn ? target[(int64_t)row * Pn + j] : 0.0f;
    auto row_max = [](float v) {
      v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
      v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 1, 64));
      return v;
    };
    auto row_sum = [](float v) {
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      return v;
    };
    const float m = row_max(xv);
    const float ex = expf(xv - m);                 // (exp(-inf) = 0 for the columns past the end)
    const float se = row_sum(ex);
    const float lse = logf(se);
    const float nll = j < Pn ? -((xv - m) - lse) * tv : 0.0f;
    const float rl = row_sum(nll);
    const float gs = 1.0f * inv_rows;
    const float st = row_sum(tv);
    const float gu = j < Pn ? gs * tv * -1.0f + ex * ((gs * st) / se) : 0.0f;
    const float gsum = row_sum(gu);
    const float d = j < Pn ? gu + (xv == m ? -gsum : 0.0f) : 0.0f;
    sdl[r][j] = d;
    if (r0 + r < M && j < Pn) {
      logits[(int64_t)row * Pn + j] = y;
      dlogits[(int64_t)row * Pn + j] = d;
    }
    if (r0 + r < M && j == 0) rowloss[row] = rl;
  }
  __syncthreads();
  // d loss / d (fc1 output) = pass ? sum_j d[r][j] W2[col][j] : 0
  if (ks == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < P; j++) s = fmaf(sdl[r][j], w2r[j], s);
      if (r0 + r < M) dz1[(int64_t)(r0 + r) * HD + col] = pass[r] ? s : 0.0f;
    }
  }
}

// ---- the forward pass as TWO launches (the default from 64 rows on): every workgroup of the one-launch kernel above streams
// all of W1 (803 KB for 784 -> 256) through its CU -- 17.8 us, whatever the depth of its prefetch --, and nothing but a
// split of the hidden units or of K over workgroups shortens that, which puts fc2 (all hidden units of a row) behind a
// launch boundary.  k_mlp_fc1: z = x . W1 + b1, a 16 x 16 tile per workgroup (128 workgroups at 128 x 256), its four waves a
// quarter of K each, every operand requested up front; k_mlp2_tail: relu, fc2, cross-entropy, d loss / d logits,
// d loss / d z for four rows per workgroup (the second half of the kernel above).
__global__ __launch_bounds__(256) void k_mlp_fc1(const float* __restrict__ x, const float* __restrict__ w1,
                                                const float* __restrict__ b1, float* __restrict__ z, int M, int K, int HD) {
  __shared__ f32x4 red[3][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int r0 = (int)blockIdx.x * 16, c0 = (int)blockIdx.y * 16;
  const int kq = K / 4, kb = wave * kq;            // (K % 16 == 0: kq is a multiple of 4)
  const float* xr = x + (int64_t)min(r0 + i, M - 1) * K + kb + 4 * kk;        // A[i][k]: 4 consecutive k per 16-k chunk
  const float* wp = w1 + (int64_t)(kb + 4 * kk) * HD + c0 + i;                // B[k][j = i]
  constexpr int CH = 13;                            // 16-k chunks in flight (K <= 4 * 16 * CH + tail: 784 -> 12.25)
  const int full = kq / 16, tail = (kq - 16 * full) / 4;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int t0 = 0; t0 < full; t0 += CH) {
    float4 xv[CH];
    float wv[CH][4];
#pragma unroll
    for (int t = 0; t < CH; t++) {
      const int tc = min(t0 + t, full - 1);         // (a chunk past the end re-reads the last one: times 0)
      xv[t] = *reinterpret_cast<const float4*>(xr + 16 * tc);
#pragma unroll
      for (int e = 0; e < 4; e++) wv[t][e] = wp[(int64_t)(16 * tc + e) * HD];
    }
    asm volatile("" ::: "memory");                 // (a compiler barrier: the selection DAG sinks plain loads to their uses)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < CH; t++) {
      const float on = t0 + t < full ? 1.0f : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].x * on, wv[t][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].y * on, wv[t][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].z * on, wv[t][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].w * on, wv[t][3], acc, 0, 0, 0);
    }
  }
  for (int s2 = 0; s2 < tail; s2++) {              // the rest of the range, 4 k at a time: lane (i, kk) takes k = kk
    const int k = kb + 16 * full + 4 * s2 + kk;
    const float a = x[(int64_t)min(r0 + i, M - 1) * K + k];
    const float b = w1[(int64_t)k * HD + c0 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  if (wave > 0) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave == 0) {                                  // lane holds D[4 kk + v][i]
    const float bias = b1 ? b1[c0 + i] : 0.0f;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float s3 = ((acc[v] + red[0][lane][v]) + (red[1][lane][v] + red[2][lane][v])) + bias;
      if (r0 + 4 * kk + v < M) z[(int64_t)(r0 + 4 * kk + v) * HD + c0 + i] = s3;
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_mlp2_tail(
    const float* __restrict__ z, const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ target,
    float* __restrict__ h, float* __restrict__ logits, float* __restrict__ dlogits, float* __restrict__ dz1,
    float* __restrict__ rowloss, int M, int Pn, float inv_rows) {
  constexpr int HD = 64 * NW, P = kP, NWAVES = NW;
  __shared__ float hs[4][HD];
  __shared__ float sw2[HD][P];
  __shared__ float plog[NWAVES][4][P];
  __shared__ float sdl[4][P];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = (int)blockIdx.x * 4;
  const int col = tid;
  constexpr int W2V = (HD * P + 64 * NWAVES - 1) / (64 * NWAVES);
  float w2v[W2V];
#pragma unroll
  for (int i = 0; i < W2V; i++) {
    const int e = tid + 64 * NWAVES * i, c = e / P, j = e - c * P;
    const float v = w2[(int64_t)min(c, HD - 1) * Pn + min(j, Pn - 1)];
    w2v[i] = v * ((c < HD && j < Pn) ? 1.0f : 0.0f);
  }
  float zv[4];
#pragma unroll
  for (int r = 0; r < 4; r++) zv[r] = z[(int64_t)min(r0 + r, M - 1) * HD + col];
#pragma unroll
  for (int i = 0; i < W2V; i++) {
    const int e = tid + 64 * NWAVES * i;
    if (e < HD * P) (&sw2[0][0])[e] = w2v[i];
  }
  bool pass[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float y = relu_point(zv[r]);             // maximum(z, 0): NaN stays NaN (np.maximum)
    pass[r] = y == zv[r];                          // tensor.py:872-877: the gradient passes where y == x (also at 0)
    hs[r][col] = y;
    if (r0 + r < M) h[(int64_t)(r0 + r) * HD + col] = y;
  }
  __syncthreads();
  float w2r[P];
#pragma unroll
  for (int j = 0; j < P; j++) w2r[j] = sw2[col][j];
  {
    constexpr int CPW = HD / NWAVES;               // 64 hidden units per wave, for (row, class) = (lane / 16, lane % 16)
    const int r = lane >> 4, j = lane & 15, cb = wave * CPW;
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};        // (four chains; added in a fixed order)
#pragma unroll
    for (int c = 0; c < CPW; c++) s4[c & 3] = fmaf(hs[r][cb + c], sw2[cb + c][j], s4[c & 3]);
    plog[wave][r][j] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  __syncthreads();
  if (wave == 0) {
    const int r = lane >> 4, j = lane & 15;
    const int row = min(r0 + r, M - 1);
    float y = 0.0f;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) y += plog[w][r][j];
    y += (b2 && j < Pn) ? b2[j] : 0.0f;
    const float xv = j < Pn ? y : -INFINITY;
    const float tv = j < Pn ? target[(int64_t)row * Pn + j] : 0.0f;
    auto row_max = [](float v) {
      v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
      v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 1, 64));
      return v;
    };
    auto row_sum = [](float v) {
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      return v;
    };
    const float m = row_max(xv);
    const float ex = expf(xv - m);
    const float se = row_sum(ex);
    const float lse = logf(se);
    const float nll = j < Pn ? -((xv - m) - lse) * tv : 0.0f;
    const float rl = row_sum(nll);
    const float gs = 1.0f * inv_rows;
    const float st = row_sum(tv);
    const float gu = j < Pn ? gs * tv * -1.0f + ex * ((gs * st) / se) : 0.0f;
    const float gsum = row_sum(gu);
    const float d = j < Pn ? gu + (xv == m ? -gsum : 0.0f) : 0.0f;
    sdl[r][j] = d;
    if (r0 + r < M && j < Pn) {
      logits[(int64_t)row * Pn + j] = y;
      dlogits[(int64_t)row * Pn + j] = d;
    }
    if (r0 + r < M && j == 0) rowloss[row] = rl;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < P; j++) s = fmaf(sdl[r][j], w2r[j], s);
    if (r0 + r < M) dz1[(int64_t)(r0 + r) * HD + col] = pass[r] ? s : 0.0f;
  }
}

struct MlpBwd {
  const float* x;        // (M, K)
  const float* h;        // (M, HD)
  const float* dz1;      // (M, HD)
  const float* dl;       // (M, Pn)
  float* dw1;            // (K, HD)
  float* db1;            // (HD) or nullptr
  float* dw2;            // (HD, Pn)
  float* db2;            // (Pn) or nullptr
  const float* rowloss;  // (M), with
  float* loss;           // (1) or nullptr: the mean of the row losses
  int M, K, Pn;
  float inv_rows;
};

// blocks [0, (K / 16) * HALVES): dW1 rows [16 b, 16 b + 16) x one half of the hidden units (all of them for 64), every operand
// of the round requested up front; the blocks of the first 16 rows also form db1 from their dz1 operand.  Then HD / 64 blocks
// of head_grads_body (dfhip_head.h): dW2 = h^T . dlogits, db2 and the mean of the row losses.
// SGD (the optimizer's call arrived while this launch was still deferred -- see mlp_bwd_take_sgd below): every gradient
// element is complete in the thread that stores it, so optim/sgd.py:16-24 is applied to its parameter right there (sgd_elem,
// dfhip_head.h: the multi-tensor launch's expressions and bits) and the step has no optimizer launch.
template <int NW, bool SGD>
__global__ __launch_bounds__(256) void k_mlp2_bwd(const MlpBwd a, const HeadGrads hg, const SgdSites sg) {
  constexpr int HD = 64 * NW, NT = NW >= 2 ? NW / 2 : 1, HALVES = NW / NT;      // NT 16-wide column tiles per wave (four waves)
  __shared__ __attribute__((aligned(16))) float lds[kHeadLdsFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb1 = (a.K / 16) * HALVES;
  const int blk = blockIdx.x;
  SgdHyper sh = {};
  if constexpr (SGD) sh = sg.hyper_dev ? *reinterpret_cast<const SgdHyper*>(sg.hyper_dev) : sg.h;
  if (blk >= nb1) {
    if constexpr (SGD) head_grads_body<false>(hg, blk - nb1, lds, &sg, &sh);
    else head_grads_body<false>(hg, blk - nb1, lds);
    return;
  }
  const int b = blk / HALVES, hh = blk - b * HALVES;
  // D (16 input features x 16 hidden units) += A (16 x 4 batch rows) . B (4 x 16): lane (i = lane % 16, kk = lane / 16)
  const int i = lane & 15, kk = lane >> 4;
  const int c0 = (hh * 4 + wave) * NT * 16;        // this wave's first hidden unit
  const float* ap = a.x + (int64_t)b * 16 + i;
  const float* bp = a.dz1 + c0 + i;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = {0.0f, 0.0f, 0.0f, 0.0f};
  float colsum[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) colsum[t] = 0.0f;
  const int steps = (a.M + 3) / 4;
  constexpr int U = 32;                            // batch steps per round: 128 rows, every load of the round in flight
#pragma unroll 1
  for (int s0 = 0; s0 < steps; s0 += U) {
    float av[U], bv[U][NT];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int nc = min(4 * (s0 + u) + kk, a.M - 1);          // (a row past the end re-reads the last one: times 0 below)
      av[u] = ap[(int64_t)nc * a.K];
#pragma unroll
      for (int t = 0; t < NT; t++) bv[u][t] = bp[(int64_t)nc * HD + 16 * t];
    }
    asm volatile("" ::: "memory");                 // (a compiler barrier: the selection DAG sinks plain loads to their uses)
    __builtin_amdgcn_sched_barrier(0);             // (every load of the round is requested before the first MFMA waits)
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float on = 4 * (s0 + u) + kk < a.M ? 1.0f : 0.0f;
      const float am = av[u] * on;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const float bm = bv[u][t] * on;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc[t], 0, 0, 0);
        colsum[t] += bm;
      }
    }
  }
  // lane holds D[4 kk + v][i] of each tile
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int64_t at = (int64_t)(b * 16 + 4 * kk + v) * HD + c0 + 16 * t + i;
      a.dw1[at] = acc[t][v];
      if constexpr (SGD)
        if (sg.p[0]) sgd_elem_at(sg.p[0], sg.mb[0], at, acc[t][v], sh);
    }
  if (b == 0 && a.db1) {                           // db1 = column sums of dz1: the four row groups of a lane's column meet
#pragma unroll
    for (int t = 0; t < NT; t++) {
      float sct = colsum[t];
      sct += __shfl_xor(sct, 16, 64);
      sct += __shfl_xor(sct, 32, 64);
      if (kk == 0) {
        a.db1[c0 + 16 * t + i] = sct;
        if constexpr (SGD)
          if (sg.p[1]) sgd_elem_at(sg.p[1], sg.mb[1], c0 + 16 * t + i, sct, sh);
      }
    }
  }
}

struct MlpPending { bool active = false; MlpBwd a; int HD = 0; std::vector<float*> frees; } g_mlp_pending;

// arrays of the pending launch that were freed while it waited go back to the pool behind it
int release_deferred_frees() {
  int st = DFHIP_OK;
  for (float* p : g_mlp_pending.frees) {
    const int s = dfhip_free(p);
    if (st == DFHIP_OK) st = s;
  }
  g_mlp_pending.frees.clear();
  return st;
}

int mlp_bwd_launch(const MlpBwd& a, int HD, const SgdSites* sg) {
  const int M = a.M, K = a.K, P = a.Pn;
  ProfScope prof(sg ? "mlp2_ce_bwd_sgd" : "mlp2_ce_bwd", 2.0 * M * ((double)K * HD + (double)HD * P),
                 4.0 * ((double)M * K + (sg ? 3.0 : 1.0) * ((double)K * HD + (double)HD * P) + 2.0 * M * HD + (double)M * P));
  const HeadGrads hg = {a.dl, a.h, a.dw2, a.db2, a.rowloss, a.loss, a.inv_rows, M, HD, P};
  const int halves = HD >= 128 ? 2 : 1;
  const dim3 grid((unsigned)((K / 16) * halves + head_grads_blocks(hg)));
  const SgdSites none = {};
#define DF_BWD(NWV)                                                                                       \
  do {                                                                                                    \
    if (sg) DF_LAUNCH((k_mlp2_bwd<NWV, true>), grid, dim3(256), 0, stream(), a, hg, *sg);                 \
    else DF_LAUNCH((k_mlp2_bwd<NWV, false>), grid, dim3(256), 0, stream(), a, hg, none);                  \
  } while (0)
  if (HD == 64) DF_BWD(1);
  else if (HD == 128) DF_BWD(2);
  else DF_BWD(4);
#undef DF_BWD
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

}  // namespace

extern "C" {

int dfhip_mlp2_ce_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* target, float* zpre, float* h, float* logits, float* dlogits, float* dz1,
                          float* rowloss, int M, int K, int HD, int P, int* done) {
  DF_INIT();
  DF_CHECK_ARG(done, "mlp2_ce_forward: null done");
  *done = 0;
  DF_CHECK_ARG(M > 0 && K > 0 && HD > 0 && P > 0, "mlp2_ce_forward: bad dimensions");
  DF_CHECK_ARG(x && w1 && w2 && target && h && logits && dlogits && dz1 && rowloss, "mlp2_ce_forward: null array");
  if (!(K % 16 == 0 && K <= 4096 && (HD == 64 || HD == 128 || HD == 256) && P <= kP && M <= 4096)) return DFHIP_OK;
  const float inv_rows = 1.0f / (float)M;
  const dim3 grid((unsigned)((M + 3) / 4));
  static int one = -1;                               // DFHIP_MLP_ONE=1: the one-launch forward kernel at every size
  if (one < 0) one = getenv("DFHIP_MLP_ONE") ? atoi(getenv("DFHIP_MLP_ONE")) : 0;
  if (M >= 64 && !one) {
    // two launches: z = x . W1 + b1 on 16 x 16 tiles, then everything that belongs to a row (zpre is the buffer between them)
    float* zb = zpre ? zpre : dz1;                   // (without a caller's array the pre-activations pass through dz1's)
    {
      ProfScope prof("mlp_fc1_fwd", 2.0 * M * (double)K * HD, 4.0 * ((double)M * K + (double)K * HD + (double)M * HD));
      DF_LAUNCH(k_mlp_fc1, dim3((unsigned)((M + 15) / 16), (unsigned)(HD / 16)), dim3(256), 0, stream(), x, w1, b1, zb, M, K, HD);
      DF_LAUNCH_CHECK();
    }
    ProfScope prof("mlp2_ce_fwd", 4.0 * M * (double)HD * P, 4.0 * (4.0 * M * HD + (double)HD * P + 3.0 * M * P));
    if (HD == 64) DF_LAUNCH(k_mlp2_tail<1>, grid, dim3(64), 0, stream(), zb, w2, b2, target, h, logits, dlogits, dz1, rowloss, M, P, inv_rows);
    else if (HD == 128) DF_LAUNCH(k_mlp2_tail<2>, grid, dim3(128), 0, stream(), zb, w2, b2, target, h, logits, dlogits, dz1, rowloss, M, P, inv_rows);
    else DF_LAUNCH(k_mlp2_tail<4>, grid, dim3(256), 0, stream(), zb, w2, b2, target, h, logits, dlogits, dz1, rowloss, M, P, inv_rows);
    DF_LAUNCH_CHECK();
    *done = 1;
    return DFHIP_OK;
  }
  ProfScope prof("mlp2_ce_fwd", 2.0 * M * ((double)K * HD + 2.0 * HD * P),
                 4.0 * ((double)M * K + (double)K * HD + (double)HD * P + 2.0 * M * HD + 3.0 * M * P));
#define DF_FWD(NWV, RV)                                                                                                 \
  DF_LAUNCH((k_mlp2_fwd<NWV, RV>), grid, dim3(64 * NWV * kKS), (size_t)4 * K * sizeof(float), stream(), x, w1, b1, w2, b2, \
            target, zpre, h, logits, dlogits, dz1, rowloss, M, K, P, inv_rows)
  // (16 k per round.  Rounds of 20 and 24 -- deeper prefetch, up to the 128 registers a 1 024-thread workgroup allows -- were
  // measured in round 6: 17.8 / 17.5 / 18.5 us; rounds of 28 spill: 26.9 us.  The launch is bound by its W1 stream, not by
  // the depth of the prefetch.)
  if (HD == 64) DF_FWD(1, 16);
  else if (HD == 128) DF_FWD(2, 16);
  else DF_FWD(4, 16);
#undef DF_FWD
  DF_LAUNCH_CHECK();
  *done = 1;
  return DFHIP_OK;
}

int dfhip_mlp2_ce_backward(const float* x, const float* h, const float* dz1, const float* dlogits, float* dw1, float* db1,
                           float* dw2, float* db2, const float* rowloss, float* loss, int M, int K, int HD, int P) {
  DF_INIT();
  DF_CHECK_ARG(M > 0 && K > 0 && K % 16 == 0 && (HD == 64 || HD == 128 || HD == 256) && P > 0 && P <= kP,
               "mlp2_ce_backward: bad dimensions (K %% 16 == 0, HD 64 / 128 / 256, P <= 16)");
  DF_CHECK_ARG(x && h && dz1 && dlogits && dw1 && dw2 && (!loss || rowloss), "mlp2_ce_backward: null array");
  // Not launched yet: when the next thing that happens to these gradients is the optimizer's SGD step over exactly them, ONE
  // launch forms the gradients and applies the step (dfhip_sgd_step / _dev ask mlp_bwd_take_sgd); anything else that touches
  // one of the arrays launches the backward pass as it is (DF_INIT / DF_TOUCH -> flush_deferred -> mlp_bwd_flush).
  g_mlp_pending.a = {x, h, dz1, dlogits, dw1, db1, dw2, db2, rowloss, loss, M, K, P, 1.0f / (float)M};
  g_mlp_pending.HD = HD;
  g_mlp_pending.active = true;
  return DFHIP_OK;
}

}  // extern "C"

namespace dfhip {

bool mlp_bwd_pending_touches(const void* p) {
  if (!g_mlp_pending.active || !p) return false;
  const MlpBwd& a = g_mlp_pending.a;
  const int HD = g_mlp_pending.HD;
  const float* q = (const float*)p;
  auto in = [&](const float* lo, int64_t n) { return lo && q >= lo && q < lo + n; };
  return in(a.dw1, (int64_t)a.K * HD) || in(a.db1, HD) || in(a.dw2, (int64_t)HD * a.Pn) || in(a.db2, a.Pn) || in(a.loss, 1) ||
         in(a.x, (int64_t)a.M * a.K) || in(a.h, (int64_t)a.M * HD) || in(a.dz1, (int64_t)a.M * HD) ||
         in(a.dl, (int64_t)a.M * a.Pn) || in(a.rowloss, a.M);
}

int mlp_bwd_flush() {
  if (!g_mlp_pending.active) return DFHIP_OK;
  g_mlp_pending.active = false;
  const int st = mlp_bwd_launch(g_mlp_pending.a, g_mlp_pending.HD, nullptr);
  const int st2 = release_deferred_frees();
  return st != DFHIP_OK ? st : st2;
}

// dfhip_free of an array the pending launch reads or writes (Tensor.backward frees the tape -- the launch's operands -- before
// the optimizer is called): the block stays allocated until the launch is on the stream
bool mlp_bwd_defer_free(float* p) {
  if (!mlp_bwd_pending_touches(p)) return false;
  g_mlp_pending.frees.push_back(p);
  return true;
}

// dfhip_sgd_step / dfhip_sgd_step_dev, before anything else: 1 = the deferred backward launch took this update along (the
// call's gradients are exactly the pending launch's), 0 = not that case (the caller proceeds; its DF_INIT launches the pending
// backward pass first), < 0 an error
int mlp_bwd_take_sgd(int n, float* const* params, const float* const* grads, float* const* mbuf, const int64_t* sizes,
                     const SgdHyper* h, const float* hyper_dev) {
  if (!g_mlp_pending.active || !params || !grads || !sizes || n < 2 || n > 4) return 0;
  if (h && h->use_momentum && !mbuf) return 0;
  const MlpBwd& a = g_mlp_pending.a;
  const int HD = g_mlp_pending.HD;
  const float* slot_g[4] = {a.dw1, a.db1, a.dw2, a.db2};
  const int64_t slot_n[4] = {(int64_t)a.K * HD, HD, (int64_t)HD * a.Pn, a.Pn};
  int want = 0;
  for (int k = 0; k < 4; k++) want += slot_g[k] ? 1 : 0;
  if (n != want) return 0;
  SgdSites sg = {};
  for (int i = 0; i < n; i++) {
    int k = 0;
    while (k < 4 && !(slot_g[k] && grads[i] == slot_g[k] && sizes[i] == slot_n[k] && !sg.p[k])) k++;
    if (k == 4 || !params[i]) return 0;
    // (a parameter that is one of the launch's own inputs or outputs would be read and written in one launch)
    if (mlp_bwd_pending_touches(params[i]) || (mbuf && mbuf[i] && mlp_bwd_pending_touches(mbuf[i]))) return 0;
    sg.p[k] = params[i];
    sg.mb[k] = mbuf ? mbuf[i] : nullptr;
  }
  if (h) sg.h = *h;
  sg.hyper_dev = hyper_dev;
  g_mlp_pending.active = false;
  const int st = mlp_bwd_launch(a, HD, &sg);
  const int st2 = release_deferred_frees();
  if (st == DFHIP_OK && st2 == DFHIP_OK) return 1;
  return -1;
}

}  // namespace dfhip
