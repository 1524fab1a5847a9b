// The cross-image sums of the classifier head's backward pass (dfhip_head_ce_forward in dfhip_fused.hip computes everything
// of the head that is per image -- pooling, Linear, cross-entropy, d loss / d logits, the input gradient -- in ONE launch
// of the forward pass): dW (C, P) = pooled^T . dlogits (tensor.py:716) and db (P) = column sums of dlogits
// (tensor.py:462-483).  They are a launch of their own (k_head_grads) or -- inside a weight-gradient group, whose slab-sum
// launch ends the backward sweep anyway -- extra workgroups of k_wgrad_group_sum (dfhip_wgrad.hip): the same body, the same
// bits, no launch.
#pragma once
#include "dfhip_internal.h"

namespace dfhip {

// optim/sgd.py:16-24 on ONE gradient element (the multi-tensor launch k_sgd and the launches that apply the update to a
// gradient the moment it is complete -- dfhip_mlp.hip -- share the expressions, hence the bits)
struct SgdHyper {
  float lr, mu, wd, gscale;
  int nesterov, use_momentum;
};
// where the parameters (and momentum buffers) of the gradients a launch produces live, for such a launch
struct SgdSites {
  float* p[4];           // parameter of gradient slot 0 .. 3 (dfhip_mlp.hip: dW1, db1, dW2, db2); nullptr = no update
  float* mb[4];          // its momentum buffer (use_momentum)
  SgdHyper h;            // the scalars, or
  const float* hyper_dev;   // where they live on the device (a captured step: dfhip_sgd_hyper keeps them current)
};

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
constexpr int kHeadLdsFloats = 256 * kHeadP;
__host__ __device__ inline int head_grads_blocks(const HeadGrads& h) { return (h.C + 63) / 64; }

// dfhip_mlp.hip: the perceptron's backward launch stays deferred until something touches its arrays; an SGD step over exactly
// its gradients is taken into the launch (1), see there
bool mlp_bwd_pending_touches(const void* p);
int mlp_bwd_flush();
bool mlp_bwd_defer_free(float* p);
int mlp_bwd_take_sgd(int n, float* const* params, const float* const* grads, float* const* mbuf, const int64_t* sizes,
                     const SgdHyper* h, const float* hyper_dev);

// host side (dfhip_wgrad.hip): hand the sums to the slab-sum launch of the weight-gradient group that is open; false = no
// group is open (launch them)
bool head_grads_defer(const HeadGrads& h);
// dfhip_fused.hip: the stand-alone launch
int head_grads_launch(const HeadGrads& h);

#if defined(__HIPCC__)
__device__ __forceinline__ void sgd_elem(float* __restrict__ p, float* __restrict__ mb, int64_t i, float gi, float pi, float mi,
                                         const SgdHyper& h) {
  if (h.gscale != 1.0f) gi = gi * h.gscale;
  gi = gi + pi * h.wd;
  float upd = gi;
  if (h.use_momentum) {
    const float vi = mi * h.mu + gi;
    mb[i] = vi;
    upd = h.nesterov ? gi + h.mu * vi : vi;
  }
  p[i] = pi + (h.lr * upd) * -1.0f;
}
// (the element's parameter and momentum read here: for a launch that has just produced the gradient)
__device__ __forceinline__ void sgd_elem_at(float* __restrict__ p, float* __restrict__ mb, int64_t i, float gi, const SgdHyper& h) {
  sgd_elem(p, mb, i, gi, p[i], h.use_momentum ? mb[i] : 0.0f, h);
}

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
  // (the four rows of a thread's first pass are requested together: one round trip, not four)
  float tot[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = 64 * (4 * wave + k) + lane;
    tot[k] = rowloss[min(r, N - 1)] * (r < N ? 1.0f : 0.0f);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int W16 = 4 * wave + k;
    for (int r = 64 * W16 + lane + 1024; r < N; r += 1024) tot[k] += rowloss[r];
    const float t = head_wave_sum(tot[k]);
    if (lane == 0) part[W16] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s2 = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < 16; w2++) s2 += part[w2];
    loss[0] = s2 * inv_rows;
  }
}

// 256 threads; block `blk` owns the 64 channels [64 blk, 64 blk + 64), a wave sixteen of them: dW tile (16 channels x 16
// classes) += A (16 channels x 4 images) . B (4 images x 16 classes) with v_mfma_f32_16x16x4 -- A is pooled^T (lane: channel
// lane % 16 of image 4 s + lane / 16; a chunk's 64 operands are requested up front), B the rows of dlogits out of LDS (256
// images at a time, zero rows past the end of a ragged chunk).  A wave holds ALL images of its channels: no sum across waves
// or workgroups.  (Round 6, first form: a thread per channel and 16 multiply-adds per image out of LDS -- 1 024 dependent
// fmas per thread, 3.8 us of a 8.6 us launch.)  Block 0 also forms db (thread j: the images in four chains), the last block
// the loss.  A fixed order of additions: the same bits in every run, inside a group or not.  lds: kHeadLdsFloats floats.
// LEAN (inside k_wgrad_group_sum: the body must stay below the slab sum's own registers): 32 operands in flight, not 64.
// sg / sh (dfhip_mlp.hip's backward launch with the optimizer step in it): slots 2, 3 of sg are dW's and db's parameters.
template <bool LEAN>
__device__ __forceinline__ void head_grads_body(const HeadGrads& h, int blk, float* lds, const SgdSites* sg = nullptr,
                                                const SgdHyper* sh = nullptr) {
  typedef float f32x4h __attribute__((ext_vector_type(4)));
  constexpr int P = kHeadP;
  if (blk == head_grads_blocks(h) - 1 && h.loss) { // (the block with the fewest channels also finishes the loss)
    loss_finish_body(h.rowloss, h.loss, h.N, h.inv_rows, lds);
    __syncthreads();
  }
  float* sdy = lds;                                // [256][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int c0 = blk * 64 + wave * 16;
  const int cc = min(c0 + i, h.C - 1);
  f32x4h acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float dbv = 0.0f;
  for (int n0 = 0; n0 < h.N; n0 += 256) {
    const int nb = min(256, h.N - n0);
    if (n0 > 0) __syncthreads();
    float stg[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
      // (times 0 / 1, not a select: under a condition the compiler sinks each load into a branch of its own -- sixteen
      // dependent round trips)
      const float v = h.dl[(int64_t)(n0 + min(tid, nb - 1)) * h.Pn + min(j, h.Pn - 1)];
      stg[j] = v * ((tid < nb && j < h.Pn) ? 1.0f : 0.0f);
    }
    const float* base = h.pooled + (int64_t)n0 * h.C + cc;
    constexpr int RS = LEAN ? 32 : 64;             // operands per round
    float av[RS];
#pragma unroll
    for (int s2 = 0; s2 < RS; s2++) av[s2] = base[(int64_t)min(4 * s2 + kk, nb - 1) * h.C];
    asm volatile("" ::: "memory");                 // (a compiler barrier: the selection DAG sinks plain loads to their uses)
#pragma unroll
    for (int j = 0; j < P; j++) sdy[tid * P + j] = stg[j];
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < 64 / RS; r++) {
      if (r > 0) {
#pragma unroll
        for (int s2 = 0; s2 < RS; s2++) av[s2] = base[(int64_t)min(4 * (RS * r + s2) + kk, nb - 1) * h.C];
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int s2 = 0; s2 < RS; s2++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], sdy[(4 * (RS * r + s2) + kk) * P + i], acc, 0, 0, 0);
    }
    if (blk == 0 && h.db && tid < h.Pn) {
      // (sixteen LDS reads in flight, four chains: one read per trip of a plain loop was 256 dependent round trips)
      float c4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int n = 0; n < 256; n += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = sdy[(n + u) * P + tid];       // (rows past nb hold zeros)
#pragma unroll
        for (int u = 0; u < 16; u++) c4[u & 3] += v[u];
      }
      dbv += (c4[0] + c4[1]) + (c4[2] + c4[3]);
    }
  }
  // lane holds dW[c0 + 4 kk + v][i]
  if (h.dw && i < h.Pn) {
#pragma unroll
    for (int v = 0; v < 4; v++)
      if (c0 + 4 * kk + v < h.C) {
        const int64_t at = (int64_t)(c0 + 4 * kk + v) * h.Pn + i;
        h.dw[at] = acc[v];
        if (sg && sg->p[2]) sgd_elem_at(sg->p[2], sg->mb[2], at, acc[v], *sh);
      }
  }
  if (blk == 0 && h.db && tid < h.Pn) {
    h.db[tid] = dbv;
    if (sg && sg->p[3]) sgd_elem_at(sg->p[3], sg->mb[3], tid, dbv, *sh);
  }
}
#endif

}  // namespace dfhip
