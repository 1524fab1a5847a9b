// Conv2d forward / input-gradient / weight-gradient as implicit GEMMs on the CDNA4 fp32
// matrix cores (v_mfma_f32_32x32x2_f32), NCHW in and out.
//
// Reference algorithm being replaced (all Python over the device primitives):
//   forward  nn/functional.py:316-344  pad -> im2col (k*k strided gather/scatter pairs,
//            :263-283) -> permute/compact -> matmul(M=N*oh*ow, K=C*k*k, P=O) -> NCHW view
//   dgrad    tensor.py:708 (dC . B^T) -> __im2col2d.grad_fn (functional.py:285-294, the
//            overlapping windows are ASSIGNED in (i,j) row-major order: last writer wins)
//            -> __pad2d.grad_fn (:307-313)
//   wgrad    tensor.py:716 (A^T . dC)
// The col matrix (k*k times the input) is never materialised here: a workgroup stages the
// raw input patch of its pixel group (with halo, zero padding and, for the strided input
// gradient, the zero dilation) into LDS once per channel chunk, and every im2col element the
// matrix cores consume is read from that patch as  patch_base(pixel) + tap_offset(k).
//
// Orientation: D[i][j] += A[i][kk] * B[kk][j] with the OUTPUT CHANNEL on i and the PIXEL on j,
// so the 32x32 accumulator has the pixel on the lane and an epilogue store writes 32
// consecutive pixels (128 B) per half-wave straight into the NCHW plane.
//
// "overwrite" input gradient (reference-exact): for every input pixel only the LAST tap
// (i*, j*) that the reference's k*k scatter loop writes survives.  That is the accumulate
// GEMM with the pixel operand masked to the lanes whose selected tap equals the tap of the
// current k index: same kernel, one compare+select per operand read, results identical to
// the reference because every masked product is an exact +0.
#include <stdlib.h>

#include <algorithm>

#include "dfhip_internal.h"

using namespace dfhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTP = 128;  // destination pixels per workgroup (4 waves x one 32-pixel tile)

struct Geo {
  // source tensor (what is gathered): x for forward/wgrad, dy for dgrad
  const float* src;
  const float* wgt;
  const float* bias;
  float* dst;
  int N, CS, HS, WS;  // source dims
  int CD, HD, WD;     // destination dims
  int k;
  int so;  // patch step per destination pixel (forward: stride, dgrad: 1)
  int sd;  // source dilation inside the patch (forward: 1, dgrad: stride)
  int u0;  // patch row/col pr maps to source coordinate numerator r0*so + u0 + pr
  int BI, BR, BC, gr, gc;  // pixel group shape and group counts along rows / cols
  int PR, PC, PE;          // patch rows, cols, elements per channel (BI*PR*PC)
  int CC;                  // source channels per LDS chunk
  int cs_per_split;
  int mode;                // 0 forward, 1 dgrad accumulate, 2 dgrad overwrite
  int wC;                  // C of the (O,C,k,k) weight tensor
  int pad, stride, oh, ow; // conv geometry (for the overwrite tap selection)
  int64_t slab_stride;
  int dbg;                 // debug: 1 skip staging, 2 skip compute, 4 skip epilogue, 8 skip goff
};

__device__ __forceinline__ void decode_pixel(const Geo& g, int q, int& img, int& r, int& col) {
  int per_img = g.BR * g.BC;
  img = q / per_img;
  int rem = q - img * per_img;
  r = rem / g.BC;
  col = rem - r * g.BC;
}

// last valid tap along one axis for padded coordinate hp (reference loop order: ascending i,
// assignment => the largest valid i wins); -1 if no window covers the pixel
__device__ __forceinline__ int last_tap(int hp, int k, int s, int out_extent) {
  int best = -1;
  for (int i = 0; i < k; i++) {
    int u = hp - i;
    if (u >= 0 && (u % s) == 0 && (u / s) < out_extent) best = i;
  }
  return best;
}

// ---------------------------------------------------------------------------------------
// forward and input-gradient
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_igemm(Geo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k2 = g.k * g.k;
  const int KKP = (g.CC * k2 + 1) & ~1;
  constexpr int WLD = 33;
  float* Xs = smem;                          // [CC][PE]
  float* Ws = Xs + (size_t)g.CC * g.PE;      // [KKP][WLD]
  int* goff = (int*)(Ws + (size_t)KKP * WLD);  // [PE]
  int* koff = goff + g.PE;                   // [KKP]
  int* tapid = koff + KKP;                   // [KKP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;

  int gx = blockIdx.x;
  const int gcx = gx % g.gc; gx /= g.gc;
  const int grx = gx % g.gr; gx /= g.gr;
  const int n0 = gx * g.BI, r0 = grx * g.BR, c0 = gcx * g.BC;
  const int d0 = blockIdx.y * 32;
  const int cs_beg = blockIdx.z * g.cs_per_split;
  const int cs_end = min(g.CS, cs_beg + g.cs_per_split);
  const int64_t src_plane = (int64_t)g.HS * g.WS;

  // gather table: patch element -> source offset (relative to image n0, channel 0 plane)
  for (int j = tid; j < g.PE; j += 256) {
    int per_img = g.PR * g.PC;
    int img = j / per_img;
    int rem = j - img * per_img;
    int pr = rem / g.PC, pc = rem - pr * g.PC;
    int u = r0 * g.so + g.u0 + pr, v = c0 * g.so + g.u0 + pc;
    bool ok = (n0 + img < g.N) && u >= 0 && v >= 0 && (u % g.sd) == 0 && (v % g.sd) == 0;
    int su = u / g.sd, sv = v / g.sd;
    ok = ok && su < g.HS && sv < g.WS;
    goff[j] = ok ? (int)((int64_t)img * g.CS * src_plane + (int64_t)su * g.WS + sv) : -1;
  }
  for (int kk = tid; kk < KKP; kk += 256) {
    int c = kk / k2, t = kk - c * k2;
    int i = t / g.k, j = t - i * g.k;
    int off = 0;
    if (c < g.CC) off = g.mode == 0 ? c * g.PE + i * g.PC + j
                                    : c * g.PE + (g.k - 1 - i) * g.PC + (g.k - 1 - j);
    koff[kk] = off;
    tapid[kk] = t;
  }

  // this wave's 32-pixel tile
  int q = wave * 32 + l31, img, r, col;
  decode_pixel(g, q, img, r, col);
  const bool pvalid = q < g.BI * g.BR * g.BC && n0 + img < g.N && r0 + r < g.HD && c0 + col < g.WD;
  const int abase = pvalid ? img * g.PR * g.PC + r * g.so * g.PC + col * g.so : 0;
  int tapsel = -2;
  if (g.mode == 2 && pvalid) {
    int ti = last_tap(r0 + r + g.pad, g.k, g.stride, g.oh);
    int tj = last_tap(c0 + col + g.pad, g.k, g.stride, g.ow);
    tapsel = (ti < 0 || tj < 0) ? -1 : ti * g.k + tj;
  }

  f32x16 acc;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) acc[rr] = 0.0f;

  for (int cb = cs_beg; cb < cs_end; cb += g.CC) {
    const int ccn = min(g.CC, cs_end - cb);
    const int kkn = ccn * k2;
    __syncthreads();  // tables ready / previous chunk consumed
    // stage the patch: wave w takes channels w, w+4, ...
    for (int c = wave; c < ccn; c += 4) {
      const float* base = g.src + ((int64_t)n0 * g.CS + cb + c) * src_plane;
      float* xs = Xs + (size_t)c * g.PE;
#pragma unroll 4
      for (int j = lane; j < g.PE; j += 64) {
        int go = goff[j];
        xs[j] = go >= 0 ? base[go] : 0.0f;
      }
    }
    // stage the weights as Ws[kk][d]
    if (g.mode == 0) {
      // W[(d0+d)][cb + c][t]: kk = c*k2 + t is contiguous in memory for a fixed d
      for (int d = tid >> 5; d < 32; d += 8) {
        const bool dok = d0 + d < g.CD;
        const float* wrow = g.wgt + ((int64_t)(d0 + d) * g.wC + cb) * k2;
        for (int kk = tid & 31; kk < KKP; kk += 32)
          Ws[kk * WLD + d] = (dok && kk < kkn) ? wrow[kk] : 0.0f;
      }
    } else {
      // W[cb + c][d0 + d][t]
      const int d = tid & 31;
      const bool dok = d0 + d < g.CD;
#pragma unroll 4
      for (int kk = tid >> 5; kk < KKP; kk += 8) {
        int c = kk / k2, t = kk - c * k2;
        float v = 0.0f;
        if (kk < kkn && dok) v = g.wgt[((int64_t)(cb + c) * g.wC + d0 + d) * k2 + t];
        Ws[kk * WLD + d] = v;
      }
    }
    __syncthreads();

    const int npairs = kkn >> 1;
    if (g.mode == 2) {
#pragma unroll 4
      for (int p = 0; p < npairs; p++) {
        int kk = 2 * p + half;
        float a = Ws[kk * WLD + l31];
        float b = Xs[abase + koff[kk]];
        b = (tapid[kk] == tapsel) ? b : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int p = 0; p < npairs; p++) {
        int kk = 2 * p + half;
        float a = Ws[kk * WLD + l31];
        float b = Xs[abase + koff[kk]];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    if (kkn & 1) {  // odd K (e.g. 3 channels x 3x3): last k index alone in the lower half
      int kk = kkn - 1;
      float a = half ? 0.0f : Ws[kk * WLD + l31];
      float b = half ? 0.0f : Xs[abase + koff[kk]];
      if (g.mode == 2) b = (tapid[kk] == tapsel) ? b : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }

  // epilogue: accumulator register rr of lane (half, l31) is D[channel][pixel] with
  // channel = (rr&3) + 8*(rr>>2) + 4*half, pixel = l31
  if (pvalid) {
    const int64_t dplane = (int64_t)g.HD * g.WD;
    float* out = g.dst + (int64_t)blockIdx.z * g.slab_stride +
                 (int64_t)(n0 + img) * g.CD * dplane + (int64_t)(r0 + r) * g.WD + (c0 + col);
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
      int d = d0 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      if (d < g.CD) {
        float v = acc[rr];
        if (g.bias) v += g.bias[d];
        out[(int64_t)d * dplane] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// forward / input-gradient, producer-consumer form (source channels % 8 == 0):
//   * 512-thread workgroup: waves 0-3 run the matrix cores (one 32-pixel x 32-channel tile
//     each), waves 4-7 stage the NEXT channel chunk (patch + weights) into the other LDS buffer
//     while they do, one barrier per chunk.  The MFMA pipe of a SIMD is fed by its compute
//     wave, its loader wave only issues VMEM / LDS / address arithmetic;
//   * K is ordered (tap, channel): inside a tap the operand addresses advance by constants
//     (2*PE and 2*WLD floats per k pair), so the inner loop is ds_read + MFMA only, unrolled
//     by four k pairs with the next group's reads issued ahead of the current MFMAs;
//   * overwrite mode: a tap that no lane of the wave selects contributes exact zeros and is
//     skipped (wave-uniform branch) - for a 3x3 stride-1 layer that leaves one tap in nine
//     for interior tiles.
// ---------------------------------------------------------------------------------------
template <int KS, int MODE>
__global__ __launch_bounds__(512, 4) void k_conv_igemm_ws(Geo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int k = KS, k2 = KS * KS;
  constexpr int WLD = 33;
  const int CCP = g.CC;                 // channels per chunk, multiple of 8
  const int CCQ = CCP + 1;              // weight rows per tap (+1: spreads the staging writes over banks)
  const int xs_floats = CCP * g.PE + 64;          // +64: the last prefetch group reads past the chunk
  const int ws_floats = k2 * CCQ * WLD + 8 * WLD;
  float* Xs0 = smem;
  float* Ws0 = Xs0 + 2 * xs_floats;
  int* goff = (int*)(Ws0 + 2 * ws_floats);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const bool loader = wave >= 4;

  int gx = blockIdx.x;
  const int gcx = gx % g.gc; gx /= g.gc;
  const int grx = gx % g.gr; gx /= g.gr;
  const int n0 = gx * g.BI, r0 = grx * g.BR, c0 = gcx * g.BC;
  const int d0 = blockIdx.y * 32;
  const int cs_beg = blockIdx.z * g.cs_per_split;
  const int cs_end = min(g.CS, cs_beg + g.cs_per_split);
  const int nchunks = (cs_end - cs_beg) / CCP;
  const int64_t src_plane = (int64_t)g.HS * g.WS;

  for (int j = tid; j < ((g.dbg & 8) ? 0 : g.PE); j += 512) {
    int per_img = g.PR * g.PC;
    int img = j / per_img;
    int rem = j - img * per_img;
    int pr = rem / g.PC, pc = rem - pr * g.PC;
    int u = r0 * g.so + g.u0 + pr, v = c0 * g.so + g.u0 + pc;
    bool ok = (n0 + img < g.N) && u >= 0 && v >= 0 && (u % g.sd) == 0 && (v % g.sd) == 0;
    int su = u / g.sd, sv = v / g.sd;
    ok = ok && su < g.HS && sv < g.WS;
    goff[j] = ok ? (int)((int64_t)img * g.CS * src_plane + (int64_t)su * g.WS + sv) : -1;
  }

  // compute waves: this wave's 32-pixel tile
  int q = (wave & 3) * 32 + l31, img, r, col;
  decode_pixel(g, q, img, r, col);
  const bool pvalid = q < g.BI * g.BR * g.BC && n0 + img < g.N && r0 + r < g.HD && c0 + col < g.WD;
  const int abase = (pvalid ? img * g.PR * g.PC + r * g.so * g.PC + col * g.so : 0) + half * g.PE;
  int tapsel = -2;
  if (MODE == 2 && pvalid) {
    int ti = last_tap(r0 + r + g.pad, k, g.stride, g.oh);
    int tj = last_tap(c0 + col + g.pad, k, g.stride, g.ow);
    tapsel = (ti < 0 || tj < 0) ? -1 : ti * k + tj;
  }

  auto stage = [&](int chunk, int buf, int nthreads, int t0) {
    // called by `nthreads` threads whose local index is t0 (all 512 in the prologue, the 256
    // loader threads in the main loop)
    const int cb = cs_beg + chunk * CCP;
    float* Xs = Xs0 + buf * xs_floats;
    float* Ws = Ws0 + buf * ws_floats;
    const int nw = nthreads >> 6, w0 = t0 >> 6, ln = t0 & 63;
    // Patch: the gather offsets of 4 consecutive 64-element slices are read once and used for
    // a group of 4 channels: 16 independent global loads in flight per thread, no branches
    // (out-of-image elements load offset 0 and are zeroed by a select).
    for (int cg = w0 * 4; cg < CCP; cg += nw * 4) {
      const float* base = g.src + ((int64_t)n0 * g.CS + cb + cg) * src_plane;
      float* xs = Xs + (size_t)cg * g.PE;
      for (int j0 = ln; j0 < g.PE; j0 += 256) {
        int go[4];
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          int j = j0 + 64 * u;
          go[u] = j < g.PE ? goff[j] : -1;
        }
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
          for (int u = 0; u < 4; u++) v[cc][u] = base[(int64_t)cc * src_plane + (go[u] >= 0 ? go[u] : 0)];
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            int j = j0 + 64 * u;
            if (j < g.PE) xs[cc * g.PE + j] = go[u] >= 0 ? v[cc][u] : 0.0f;
          }
      }
    }
    // Weights: 32*CCP*k2 elements per chunk, flat index f -> (row, e) with the lanes running
    // along the contiguous global dimension; all of a thread's loads are issued before the
    // first LDS store.
    {
      const int run = MODE == 0 ? CCP * k2 : 32 * k2;      // contiguous floats per row
      const int nrows = MODE == 0 ? 32 : CCP;               // rows: d (forward) or c (dgrad)
      const int total = run * nrows;
      const int dmax = g.CD - d0;
      constexpr int U = 6;
      for (int f0 = t0; f0 < total; f0 += nthreads * U) {
        float v[U];
        int row[U], e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          int f = f0 + nthreads * u;
          f = f < total ? f : 0;
          row[u] = f / run;
          e[u] = f - row[u] * run;
          int64_t src;
          bool ok;
          if (MODE == 0) {
            ok = row[u] < dmax;
            src = ((int64_t)(d0 + (ok ? row[u] : 0)) * g.wC + cb) * k2 + e[u];
          } else {
            ok = e[u] / k2 < dmax;
            src = ((int64_t)(cb + row[u]) * g.wC + d0) * k2 + (ok ? e[u] : 0);
          }
          v[u] = g.wgt[src];
          if (!ok) v[u] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (f0 + nthreads * u < total) {
            int c, t, d;
            if (MODE == 0) { d = row[u]; c = e[u] / k2; t = e[u] - c * k2; }
            else { c = row[u]; d = e[u] / k2; t = e[u] - d * k2; }
            Ws[(t * CCQ + c) * WLD + d] = v[u];
          }
        }
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) acc[rr] = 0.0f;

  __syncthreads();                      // goff ready
  if (nchunks > 0 && !(g.dbg & 1)) stage(0, 0, 512, tid);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ch++) {
    const int buf = ch & 1;
    if (loader) {
      if (ch + 1 < nchunks && !(g.dbg & 1)) stage(ch + 1, buf ^ 1, 256, tid - 256);
    } else if (!(g.dbg & 2)) {
      const float* Xs = Xs0 + buf * xs_floats + abase;
      const float* Ws = Ws0 + buf * ws_floats + half * WLD + l31;
      const int np = CCP >> 1;          // k pairs per tap, multiple of 4
#pragma unroll
      for (int t = 0; t < k2; t++) {
        if (MODE == 2) {
          if (__ballot(tapsel == t) == 0ull) continue;
        }
        const int i = t / k, j = t - i * k;
        const int toff = MODE == 0 ? i * g.PC + j : (k - 1 - i) * g.PC + (k - 1 - j);
        const float* bp = Xs + toff;
        const float* ap = Ws + t * CCQ * WLD;
        const int bstep = 2 * g.PE, astep = 2 * WLD;
        float a0 = ap[0], a1 = ap[astep], a2 = ap[2 * astep], a3 = ap[3 * astep];
        float b0 = bp[0], b1 = bp[bstep], b2 = bp[2 * bstep], b3 = bp[3 * bstep];
        for (int p = 0; p < np; p += 4) {
          ap += 4 * astep;
          bp += 4 * bstep;
          float na0 = 0.f, na1 = 0.f, na2 = 0.f, na3 = 0.f, nb0 = 0.f, nb1 = 0.f, nb2 = 0.f, nb3 = 0.f;
          if (p + 4 < np) {
            na0 = ap[0]; na1 = ap[astep]; na2 = ap[2 * astep]; na3 = ap[3 * astep];
            nb0 = bp[0]; nb1 = bp[bstep]; nb2 = bp[2 * bstep]; nb3 = bp[3 * bstep];
          }
          if (MODE == 2) {
            const bool sel = tapsel == t;
            b0 = sel ? b0 : 0.0f; b1 = sel ? b1 : 0.0f; b2 = sel ? b2 : 0.0f; b3 = sel ? b3 : 0.0f;
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);
          a0 = na0; a1 = na1; a2 = na2; a3 = na3;
          b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
        }
      }
    }
    __syncthreads();
  }

  if (!loader && pvalid && !(g.dbg & 4)) {
    const int64_t dplane = (int64_t)g.HD * g.WD;
    float* out = g.dst + (int64_t)blockIdx.z * g.slab_stride +
                 (int64_t)(n0 + img) * g.CD * dplane + (int64_t)(r0 + r) * g.WD + (c0 + col);
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
      int d = d0 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      if (d < g.CD) {
        float v = acc[rr];
        if (g.bias) v += g.bias[d];
        out[(int64_t)d * dplane] = v;
      }
    }
  }
}

// out[i] = sum_s slabs[s][i] (+ bias[(i / plane) % CD])
__global__ __launch_bounds__(256) void k_conv_slab_sum(const float* __restrict__ slabs,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ out, int64_t n,
                                                       int splits, int64_t plane, int CD) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    float v = slabs[i];
    for (int s = 1; s < splits; s++) v += slabs[(int64_t)s * n + i];
    if (bias) v += bias[(i / plane) % CD];
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------
// weight gradient: D[o][kk] += dY[o][pixel] * patch[pixel][kk]; the four waves split the
// pixels of a group, every wave keeps all OT x KT accumulator tiles, a block walks several
// pixel groups and the wave partials are summed through LDS at the end.
// ---------------------------------------------------------------------------------------
struct WGeo {
  const float* x;
  const float* dy;
  float* dst;  // slabs [gridDim.x][O][C*k*k]
  int N, C, H, W, O, oh, ow;
  int k, stride, pad;
  int BI, BR, BC, gr, gc, ngroups;
  int PR, PC, PE;
  int CH;      // max channels a block's kk range can touch
  int lds_floats;  // size of the staging region (also used for the cross-wave reduction)
};

template <int OT, int KT>
__global__ __launch_bounds__(256) void k_conv_wgrad(WGeo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k2 = g.k * g.k;
  const int KW = g.C * k2;
  const int BMP = g.BI * g.BR * g.BC;
  const int BMPP = (BMP + 7) & ~7;   // pixels padded so every wave gets whole pairs
  const int DLD = BMPP + 1;          // odd row stride: lanes run along o
  float* Xs = smem;                              // [CH][PE]
  float* dYs = Xs + (size_t)g.CH * g.PE;         // [OT*32][DLD]
  int* goff = (int*)(dYs + (size_t)OT * 32 * DLD);  // [PE]
  int* pbase = goff + g.PE;                      // [BMPP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int kk0 = blockIdx.y * (KT * 32);
  const int o0 = blockIdx.z * (OT * 32);
  const int cA = kk0 / k2;
  const int cB = min(g.C, (min(KW, kk0 + KT * 32) - 1) / k2 + 1);
  const int chn = cB - cA;
  const int64_t xplane = (int64_t)g.H * g.W, yplane = (int64_t)g.oh * g.ow;

  int koffL[KT];
#pragma unroll
  for (int t = 0; t < KT; t++) {
    int kk = kk0 + t * 32 + l31;
    int off = 0;
    if (kk < KW) {
      int c = kk / k2, tt = kk - c * k2;
      int i = tt / g.k, j = tt - i * g.k;
      off = (c - cA) * g.PE + i * g.PC + j;
    }
    koffL[t] = off;
  }

  f32x16 acc[OT][KT];
#pragma unroll
  for (int a = 0; a < OT; a++)
#pragma unroll
    for (int b = 0; b < KT; b++)
#pragma unroll
      for (int rr = 0; rr < 16; rr++) acc[a][b][rr] = 0.0f;

  const int pairs_per_wave = BMPP / 8;  // BMPP/2 pairs over 4 waves

  for (int grp = blockIdx.x; grp < g.ngroups; grp += gridDim.x) {
    int gx = grp;
    const int gcx = gx % g.gc; gx /= g.gc;
    const int grx = gx % g.gr; gx /= g.gr;
    const int n0 = gx * g.BI, r0 = grx * g.BR, c0 = gcx * g.BC;
    __syncthreads();  // previous group consumed
    for (int j = tid; j < g.PE; j += 256) {
      int per_img = g.PR * g.PC;
      int img = j / per_img;
      int rem = j - img * per_img;
      int pr = rem / g.PC, pc = rem - pr * g.PC;
      int u = r0 * g.stride - g.pad + pr, v = c0 * g.stride - g.pad + pc;
      bool ok = (n0 + img < g.N) && u >= 0 && v >= 0 && u < g.H && v < g.W;
      goff[j] = ok ? (int)((int64_t)img * g.C * xplane + (int64_t)u * g.W + v) : -1;
    }
    // pixel -> patch base, and the dY tile (zero for pixels outside the tensor)
    for (int q = tid; q < BMPP; q += 256) {
      int img, r, col;
      int per_img = g.BR * g.BC;
      img = q / per_img;
      int rem = q - img * per_img;
      r = rem / g.BC;
      col = rem - r * g.BC;
      bool ok = q < BMP && n0 + img < g.N && r0 + r < g.oh && c0 + col < g.ow;
      pbase[q] = ok ? img * g.PR * g.PC + r * g.stride * g.PC + col * g.stride : 0;
      int64_t yo = ok ? (int64_t)(n0 + img) * g.O * yplane + (int64_t)(r0 + r) * g.ow + (c0 + col) : -1;
      for (int o = 0; o < OT * 32; o++) {
        float v = 0.0f;
        if (ok && o0 + o < g.O) v = g.dy[yo + (int64_t)(o0 + o) * yplane];
        dYs[o * DLD + q] = v;
      }
    }
    __syncthreads();
    for (int c = wave; c < chn; c += 4) {
      const float* base = g.x + ((int64_t)n0 * g.C + cA + c) * xplane;
      float* xs = Xs + (size_t)c * g.PE;
#pragma unroll 4
      for (int j = lane; j < g.PE; j += 64) {
        int go = goff[j];
        xs[j] = go >= 0 ? base[go] : 0.0f;
      }
    }
    __syncthreads();

    const int qbeg = wave * pairs_per_wave * 2;
#pragma unroll 2
    for (int p = 0; p < pairs_per_wave; p++) {
      int q = qbeg + 2 * p + half;
      int pb = pbase[q];
      float a[OT], b[KT];
#pragma unroll
      for (int t = 0; t < OT; t++) a[t] = dYs[(t * 32 + l31) * DLD + q];
#pragma unroll
      for (int t = 0; t < KT; t++) b[t] = Xs[pb + koffL[t]];
#pragma unroll
      for (int i = 0; i < OT; i++)
#pragma unroll
        for (int j = 0; j < KT; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // cross-wave reduction through LDS, four tiles per round; wave w finishes tile 4*round + w
  float* red = smem;  // [4 tiles][4 waves][16][64]
  float* out = g.dst + (int64_t)blockIdx.x * g.O * KW;
  constexpr int NTILES = OT * KT;
#pragma unroll
  for (int round = 0; round < (NTILES + 3) / 4; round++) {
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
      int tile = round * 4 + tt;
      if (tile < NTILES) {
        int ti = tile / KT, tj = tile % KT;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          float v = 0.0f;
          // static indexing of the accumulator array
#pragma unroll
          for (int a = 0; a < OT; a++)
#pragma unroll
            for (int b = 0; b < KT; b++)
              if (a == ti && b == tj) v = acc[a][b][rr];
          red[((tt * 4 + wave) * 16 + rr) * 64 + lane] = v;
        }
      }
    }
    __syncthreads();
    int tile = round * 4 + wave;
    if (tile < NTILES) {
      int ti = tile / KT, tj = tile % KT;
      int kk = kk0 + tj * 32 + l31;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        float v = red[((wave * 4 + 0) * 16 + rr) * 64 + lane];
        v += red[((wave * 4 + 1) * 16 + rr) * 64 + lane];
        v += red[((wave * 4 + 2) * 16 + rr) * 64 + lane];
        v += red[((wave * 4 + 3) * 16 + rr) * 64 + lane];
        int o = o0 + ti * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        if (o < g.O && kk < KW) out[(int64_t)o * KW + kk] = v;
      }
    }
  }
}

// out[i] = sum_s slabs[s][i], summed in a fixed order.  A block owns 256 consecutive elements
// (64 lanes x float4); its 16 waves take the slabs s = w, w+16, ... with 8 loads in flight,
// and the 16 partials are combined through LDS.
__global__ __launch_bounds__(1024) void k_flat_slab_sum(const float* __restrict__ slabs,
                                                        float* __restrict__ out, int64_t n,
                                                        int splits) {
  __shared__ float4 part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((n & 3) == 0 && i + 3 < n) {
    for (int s0 = w; s0 < splits; s0 += 16 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int s = s0 + 16 * u;
        v[u] = *reinterpret_cast<const float4*>(slabs + (int64_t)(s < splits ? s : 0) * n + i);
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (s0 + 16 * u < splits) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  } else {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < 4; e++)
      if (i + e < n)
        for (int s = w; s < splits; s += 16) t[e] += slabs[(int64_t)s * n + i + e];
    acc = make_float4(t[0], t[1], t[2], t[3]);
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    float4 r = part[0][lane];
    for (int k = 1; k < 16; k++) {
      float4 q = part[k][lane];
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    float t[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 4; e++)
      if (i + e < n) out[i + e] = t[e];
  }
}

void plan_groups(int N, int HD, int WD, int tp, int& BI, int& BR, int& BC, int& gi, int& gr, int& gc) {
  int PI = HD * WD;
  if (PI <= tp) {
    BI = std::max(1, tp / PI);
    BI = std::min(BI, N);
    BR = HD;
    BC = WD;
  } else if (WD <= tp) {
    BI = 1;
    BC = WD;
    BR = std::max(1, tp / WD);
  } else {
    BI = 1;
    BR = 1;
    BC = tp;
  }
  gi = (int)df_ceil_div(N, BI);
  gr = (int)df_ceil_div(HD, BR);
  gc = (int)df_ceil_div(WD, BC);
}

int check_conv_args(const char* name, int N, int C, int H, int W, int O, int k, int stride, int pad,
                    int& oh, int& ow) {
  DF_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && O > 0, "%s: non-positive dimension", name);
  DF_CHECK_ARG(k >= 1 && stride >= 1 && pad >= 0, "%s: bad kernel/stride/padding", name);
  DF_CHECK_ARG(H + 2 * pad >= k && W + 2 * pad >= k, "%s: kernel larger than padded input", name);
  oh = (H + 2 * pad - k) / stride + 1;
  ow = (W + 2 * pad - k) / stride + 1;
  DF_CHECK_ARG((int64_t)N * std::max(C, O) * std::max(H, oh) * std::max(W, ow) < ((int64_t)1 << 31),
               "%s: tensor too large for 32-bit offsets", name);
  return DFHIP_OK;
}

constexpr size_t kLdsBudget = 64 * 1024;  // keeps two workgroups resident per CU

template <int KS, int MODE>
int launch_igemm_ws(const Geo& g, dim3 grid, size_t lds) {
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_igemm_ws<KS, MODE>,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  hipLaunchKernelGGL((k_conv_igemm_ws<KS, MODE>), grid, dim3(512), lds, stream(), g);
  return DFHIP_OK;
}

int finish_slabs(const Geo& g, int splits, float* final_dst, const float* bias, int64_t out_elems) {
  if (splits > 1) {
    int nb = (int)std::min<int64_t>(df_ceil_div(out_elems, 256), (int64_t)num_cus() * 8);
    hipLaunchKernelGGL(k_conv_slab_sum, dim3(nb), dim3(256), 0, stream(), g.dst, bias, final_dst,
                       out_elems, splits, (int64_t)g.HD * g.WD, g.CD);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

// producer-consumer kernel; returns -1 when the shape is not eligible
int run_igemm_ws(Geo g, int dest_channels, float* final_dst, const float* bias) {
  if (g.CS % 8 != 0 || !(g.k == 1 || g.k == 3 || g.k == 5)) return -1;
  {
    const char* e = getenv("DFHIP_CONV_DBG");
    g.dbg = e ? atoi(e) : 0;
  }
  const int k2 = g.k * g.k;
  int gi;
  plan_groups(g.N, g.HD, g.WD, kTP, g.BI, g.BR, g.BC, gi, g.gr, g.gc);
  g.PR = (g.BR - 1) * g.so + g.k;
  g.PC = (g.BC - 1) * g.so + g.k;
  g.PE = g.BI * g.PR * g.PC;
  auto lds_for = [&](int ccp) {
    size_t xs = (size_t)ccp * g.PE + 64, ws = (size_t)k2 * (ccp + 1) * 33 + 8 * 33;
    return (2 * (xs + ws) + (size_t)g.PE) * 4;
  };
  int ccp = 0;
  for (int c : {32, 16, 8})
    if (g.CS % c == 0 && lds_for(c) <= 76 * 1024) { ccp = c; break; }
  if (!ccp) {
    if (lds_for(8) > 150 * 1024) return -1;
    ccp = 8;
  }
  // prefer at least two chunks per block so staging overlaps the matrix cores
  if (ccp > 8 && g.CS / ccp < 2) ccp /= 2;
  g.CC = ccp;
  const int pixel_blocks = gi * g.gr * g.gc;
  const int dtiles = (int)df_ceil_div(dest_channels, 32);
  const int chunks = g.CS / ccp;
  int64_t blocks = (int64_t)pixel_blocks * dtiles;
  int splits = 1;
  int64_t target = (int64_t)num_cus() * 2;
  if (blocks < target) splits = (int)std::min<int64_t>(chunks, df_ceil_div(target, blocks));
  while (chunks % splits) splits--;      // equal, whole chunks per split
  g.cs_per_split = (chunks / splits) * ccp;
  const int64_t out_elems = (int64_t)g.N * g.CD * g.HD * g.WD;
  g.slab_stride = 0;
  g.dst = final_dst;
  g.bias = bias;
  if (splits > 1) {
    float* ws = workspace((size_t)out_elems * splits * sizeof(float));
    if (!ws) return DFHIP_OUT_OF_MEMORY;
    g.dst = ws;
    g.slab_stride = out_elems;
    g.bias = nullptr;
  }
  size_t lds = lds_for(ccp);
  dim3 grid((unsigned)pixel_blocks, (unsigned)dtiles, (unsigned)splits);
  int st = DFHIP_OK;
#define DF_WS(KSV)                                                        \
  (g.mode == 0 ? launch_igemm_ws<KSV, 0>(g, grid, lds)                    \
               : (g.mode == 1 ? launch_igemm_ws<KSV, 1>(g, grid, lds)     \
                              : launch_igemm_ws<KSV, 2>(g, grid, lds)))
  if (g.k == 1) st = DF_WS(1);
  else if (g.k == 3) st = DF_WS(3);
  else st = DF_WS(5);
#undef DF_WS
  if (st != DFHIP_OK) return st;
  DF_LAUNCH_CHECK();
  return finish_slabs(g, splits, final_dst, bias, out_elems);
}

int run_igemm(Geo g, int dest_channels, float* final_dst, const float* bias) {
  {
    int st = run_igemm_ws(g, dest_channels, final_dst, bias);
    if (st >= 0) return st;
  }
  const int k2 = g.k * g.k;
  int gi;
  plan_groups(g.N, g.HD, g.WD, kTP, g.BI, g.BR, g.BC, gi, g.gr, g.gc);
  g.PR = (g.BR - 1) * g.so + g.k;
  g.PC = (g.BC - 1) * g.so + g.k;
  g.PE = g.BI * g.PR * g.PC;
  // channels per LDS chunk from the budget: CC*(PE + k2*33) floats + tables
  size_t fixed = (size_t)g.PE * 4 + 64;
  int cc = (int)((kLdsBudget - fixed) / (4 * ((size_t)g.PE + (size_t)k2 * 33 + 2 * (size_t)k2)));
  cc = std::max(1, std::min(cc, g.CS));
  if (cc > 1 && ((cc * k2) & 1)) cc -= 1;  // keep chunk K even so only the very last pair can be odd
  g.CC = cc;
  const int pixel_blocks = gi * g.gr * g.gc;
  const int dtiles = (int)df_ceil_div(dest_channels, 32);
  // split the source channels until the grid covers the chip about twice
  int chunks = (int)df_ceil_div(g.CS, g.CC);
  int64_t blocks = (int64_t)pixel_blocks * dtiles;
  int splits = 1;
  int64_t target = (int64_t)num_cus() * 2;
  if (blocks < target) splits = (int)std::min<int64_t>(chunks, df_ceil_div(target, blocks));
  int chunks_per_split = (int)df_ceil_div(chunks, splits);
  g.cs_per_split = chunks_per_split * g.CC;
  splits = (int)df_ceil_div(g.CS, g.cs_per_split);
  const int64_t out_elems = (int64_t)g.N * g.CD * g.HD * g.WD;
  g.slab_stride = 0;
  g.dst = final_dst;
  g.bias = bias;
  if (splits > 1) {
    float* ws = workspace((size_t)out_elems * splits * sizeof(float));
    if (!ws) return DFHIP_OUT_OF_MEMORY;
    g.dst = ws;
    g.slab_stride = out_elems;
    g.bias = nullptr;
  }
  const int KKP = (g.CC * k2 + 1) & ~1;
  size_t lds = ((size_t)g.CC * g.PE + (size_t)KKP * 33) * 4 + ((size_t)g.PE + 2 * (size_t)KKP) * 4;
  DF_CHECK_ARG(lds <= 160 * 1024, "conv2d: receptive field too large for the LDS-staged kernel (%zu B)", lds);
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_igemm, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  dim3 grid((unsigned)pixel_blocks, (unsigned)dtiles, (unsigned)splits);
  hipLaunchKernelGGL(k_conv_igemm, grid, dim3(256), lds, stream(), g);
  DF_LAUNCH_CHECK();
  return finish_slabs(g, splits, final_dst, bias, out_elems);
}

template <int OT, int KT>
int launch_wgrad(WGeo g, float* dw) {
  const int k2 = g.k * g.k;
  const int KW = g.C * k2;
  const int BMP = g.BI * g.BR * g.BC;
  const int BMPP = (BMP + 7) & ~7;
  g.CH = std::min(g.C, (KT * 32 + k2 - 2) / k2 + 1);
  size_t stage = ((size_t)g.CH * g.PE + (size_t)OT * 32 * (BMPP + 1)) * 4 + ((size_t)g.PE + BMPP) * 4;
  size_t red = (size_t)4 * 4 * 16 * 64 * 4;
  size_t lds = std::max(stage, red);
  DF_CHECK_ARG(lds <= 160 * 1024, "conv2d wgrad: tile does not fit LDS (%zu B)", lds);
  int ktiles = (int)df_ceil_div(KW, KT * 32), otiles = (int)df_ceil_div(g.O, OT * 32);
  int64_t target = (int64_t)num_cus() * 2;
  int gx = (int)std::max<int64_t>(1, std::min<int64_t>(g.ngroups, df_ceil_div(target, (int64_t)ktiles * otiles)));
  const int64_t n = (int64_t)g.O * KW;
  float* slabs = dw;
  if (gx > 1) {
    slabs = workspace((size_t)n * gx * sizeof(float));
    if (!slabs) return DFHIP_OUT_OF_MEMORY;
  }
  g.dst = slabs;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_wgrad<OT, KT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  dim3 grid((unsigned)gx, (unsigned)ktiles, (unsigned)otiles);
  hipLaunchKernelGGL((k_conv_wgrad<OT, KT>), grid, dim3(256), lds, stream(), g);
  DF_LAUNCH_CHECK();
  if (gx > 1) {
    hipLaunchKernelGGL(k_flat_slab_sum, dim3((unsigned)df_ceil_div(n, 256)), dim3(1024), 0, stream(), slabs,
                       dw, n, gx);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

}  // namespace

extern "C" {

int dfhip_conv2d_forward(const float* x, const float* w, const float* bias, float* y, int N, int C,
                         int H, int W, int O, int k, int stride, int pad) {
  DF_INIT();
  int oh, ow;
  int s = check_conv_args("conv2d_forward", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(x && w && y, "conv2d_forward: null array");
  double flops = 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + (double)N * O * oh * ow);
  ProfScope prof("conv2d_fwd_igemm", flops, bytes);
  Geo g = {};
  g.src = x; g.wgt = w;
  g.N = N; g.CS = C; g.HS = H; g.WS = W;
  g.CD = O; g.HD = oh; g.WD = ow;
  g.k = k; g.so = stride; g.sd = 1; g.u0 = -pad;
  g.mode = 0; g.wC = C; g.pad = pad; g.stride = stride; g.oh = oh; g.ow = ow;
  return run_igemm(g, O, y, bias);
}

int dfhip_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int C, int H, int W,
                       int O, int k, int stride, int pad, int mode) {
  DF_INIT();
  int oh, ow;
  int s = check_conv_args("conv2d_dgrad", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null array");
  DF_CHECK_ARG(mode == 0 || mode == 1, "conv2d_dgrad: mode must be 0 (overwrite) or 1 (accumulate)");
  double flops = 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + (double)N * O * oh * ow);
  ProfScope prof(mode == 0 ? "conv2d_dgrad_overwrite_igemm" : "conv2d_dgrad_igemm", flops, bytes);
  Geo g = {};
  g.src = dy; g.wgt = w;
  g.N = N; g.CS = O; g.HS = oh; g.WS = ow;
  g.CD = C; g.HD = H; g.WD = W;
  g.k = k; g.so = 1; g.sd = stride; g.u0 = pad - (k - 1);
  g.mode = mode == 0 ? 2 : 1; g.wC = C; g.pad = pad; g.stride = stride; g.oh = oh; g.ow = ow;
  return run_igemm(g, C, dx, nullptr);
}

int dfhip_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W,
                       int O, int k, int stride, int pad) {
  DF_INIT();
  int oh, ow;
  int s = check_conv_args("conv2d_wgrad", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(x && dy && dw, "conv2d_wgrad: null array");
  double flops = 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + (double)N * O * oh * ow);
  ProfScope prof("conv2d_wgrad_igemm", flops, bytes);
  WGeo g = {};
  g.x = x; g.dy = dy;
  g.N = N; g.C = C; g.H = H; g.W = W; g.O = O; g.oh = oh; g.ow = ow;
  g.k = k; g.stride = stride; g.pad = pad;
  const int KW = C * k * k;
  const int k2 = k * k;
  const int ktiles32 = (int)df_ceil_div(KW, 32);
  int OT, KT;
  if (O <= 32) {
    OT = 1;
    KT = ktiles32 >= 7 ? 9 : (ktiles32 >= 3 ? 4 : 1);
  } else {
    OT = 2;
    KT = ktiles32 >= 3 ? 4 : 1;
  }
  // shrink the pixel group until patch + dY tile fit the LDS
  const int CH = std::min(C, (KT * 32 + k2 - 2) / k2 + 1);
  for (int tp = 256; tp >= 32; tp >>= 1) {
    int gi;
    plan_groups(N, oh, ow, tp, g.BI, g.BR, g.BC, gi, g.gr, g.gc);
    g.ngroups = gi * g.gr * g.gc;
    g.PR = (g.BR - 1) * stride + k;
    g.PC = (g.BC - 1) * stride + k;
    g.PE = g.BI * g.PR * g.PC;
    int BMPP = (g.BI * g.BR * g.BC + 7) & ~7;
    size_t stage = ((size_t)CH * g.PE + (size_t)OT * 32 * (BMPP + 1)) * 4 + ((size_t)g.PE + BMPP) * 4;
    if (stage <= 144 * 1024) break;
  }
  if (OT == 1 && KT == 9) return launch_wgrad<1, 9>(g, dw);
  if (OT == 1 && KT == 4) return launch_wgrad<1, 4>(g, dw);
  if (OT == 1) return launch_wgrad<1, 1>(g, dw);
  if (KT == 4) return launch_wgrad<2, 4>(g, dw);
  return launch_wgrad<2, 1>(g, dw);
}

}  // extern "C"
