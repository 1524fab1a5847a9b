// Conv2d weight gradient (tensor.py:716: A^T . dC of the reference's im2col GEMM) on the CDNA4 fp32 matrix cores: the
// staged kernels (k_conv_wgrad, _v3, the 3-channel stem form), the tap-major kernel, the row-unit kernel with its
// operands straight from global memory, the GROUPED launch that runs every queued layer of a backward sweep as one
// kernel (k_wgrad_group + k_wgrad_group_sum: dfhip_wgrad_group_begin / _end), the deferred slab sum, and the forms that
// take a pending BatchNorm backward or an affine operand on their loads (dfhip_bnbwd.h).  Split out of dfhip_conv.hip in
// round 4 (forward and input gradients stay there); plan_groups / check_conv_args / magic_u32 are shared through
// dfhip_internal.h.
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "dfhip_internal.h"
#include "dfhip_bnbwd.h"
#include "dfhip_stem_combine.h"
#include "dfhip_head.h"

using namespace dfhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// n / d for n*d < 2^32 with magic = ceil(2^32 / d); magic == 0 encodes d == 1
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic) {
  return magic ? __umulhi(n, magic) : n;
}

// the 3-channel stem's LDS tile (the geometry of k_conv_rows3, dfhip_conv.hip) and its diagnostic stamps (-DDFHIP_STEM_STAMPS)
#ifdef DFHIP_STEM_STAMPS
__device__ unsigned long long g_stem_stamps[1024 * 8 * 8];
#define STEM_STAMP(i)                                                                                   \
  do {                                                                                                  \
    if (lane == 0) g_stem_stamps[(blockIdx.x * 8 + wave) * 8 + (i)] =                                    \
        (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();                     \
  } while (0)
#else
#define STEM_STAMP(i) do { } while (0)
#endif
constexpr int kStemPitch = 40, kStemRows = 34;     // LDS tile of a channel: 34 rows of 40 floats, image column c at 4 + c


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
  // v2 (producer-consumer) only: exact division by multiplication, magic = ceil(2^32 / d)
  unsigned m_prpc, m_pc, m_brbc, m_bc, m_br;
  unsigned long long* stamps;   // diagnostics (DFHIP_CONV_STAMP)
  StemBwd st;                   // k_conv_wgrad_v3<.., STEM>: dy is the convolution OUTPUT z, dY is rebuilt while staging
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

// ---------------------------------------------------------------------------------------
// weight gradient, main kernel.  Same decomposition as the generic kernel above (every wave
// keeps all OT x KT accumulator tiles and reduces its share of the pixel rows of a group;
// partials are summed through LDS at the end), with the staging scheme of k_conv_igemm_v4:
//   * all 8 waves run the matrix cores (two per SIMD); a wave takes every 8th pixel row;
//   * the x planes and the dY tile are staged AS THEY ARE with 16-byte loads (scalar group base
//     + a lane offset computed once per block): x as [channel][image][rows x W] without halo,
//     dY as [o][pixels]; group g+1 is loaded into registers while group g is in the matrix
//     cores and stored to the other LDS buffer before the barrier;
//   * padding is resolved on the READ side: a lane of the B operand owns one (channel, tap), so
//     per pixel row / pixel pair two compares decide whether its element exists; a missing one
//     reads a neighbouring in-range address and is zeroed by a select.
// ---------------------------------------------------------------------------------------
// STEM: the convolution is followed by BatchNorm -> ReLU -> MaxPool 2x2 (the stem of the CIFAR models) and g.dy is
// its OUTPUT z: d loss / d z of a float4 is rebuilt from z, the pooled output and its gradient (two floats each)
// and the channel's constants while the vector waits in registers between its load and its LDS commit -- the
// 33.5 MB gradient tensor of the separate "apply" launch is never written or read.
template <int OT, int KT, int DV, bool STEM = false>
__global__ __launch_bounds__(512) void k_conv_wgrad_v3(WGeo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float stc[STEM ? OT * 32 : 1][6];     // (mean, 1/sd, gamma, beta, k1, k2) of the block's output channels
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int k2 = g.k * g.k;
  const int KW = g.C * k2;
  const int BRBC = g.BR * g.BC;
  const int BMP = g.BI * BRBC;
  const int DLD = BMP + 4;                    // dY row stride: 16-byte aligned rows, 4-way read conflict
  const int PEp = g.PE;                       // floats per channel in LDS: BI * RL
  const int RL = PEp / g.BI;
  const int RLr = RL / g.W;
  const int guard = (g.k * g.W + g.k + 3) & ~3;
  const int dy_floats = OT * 32 * DLD;
  const int buf_floats = dy_floats + guard + g.CH * PEp + guard;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  [[maybe_unused]] int stamp_i = 0;
#ifdef DFHIP_STAMPS
#define WG_STAMP()                                                                              \
  do {                                                                                          \
    if (g.stamps && tid == 0 && stamp_i < 32)                                                   \
      g.stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + stamp_i++] = \
          __builtin_readcyclecounter();                                                         \
  } while (0)
#else
#define WG_STAMP() do { } while (0)
#endif
  WG_STAMP();
  const int kk0 = blockIdx.y * (KT * 32);
  const int o0 = blockIdx.z * (OT * 32);
  const int cA = kk0 / k2;
  const int cB = min(g.C, (min(KW, kk0 + KT * 32) - 1) / k2 + 1);
  const int chn = cB - cA;
  const int orows = min(OT * 32, g.O - o0);
  const int64_t xplane = (int64_t)g.H * g.W, yplane = (int64_t)g.oh * g.ow;

  // B-operand lane descriptors: this lane's (channel, tap) of every kk tile
  int koffL[KT], il[KT], jl[KT];
#pragma unroll
  for (int t = 0; t < KT; t++) {
    int kk = kk0 + t * 32 + l31;
    koffL[t] = 0; il[t] = -(1 << 20); jl[t] = -(1 << 20);      // no such kk: never valid
    if (kk < KW) {
      int c = kk / k2, tt = kk - c * k2;
      int i = tt / g.k, j = tt - i * g.k;
      il[t] = i - g.pad;
      jl[t] = j - g.pad;
      koffL[t] = (c - cA) * PEp + il[t] * g.W + jl[t];
    }
  }

  f32x16 acc[OT][KT];
#pragma unroll
  for (int a = 0; a < OT; a++)
#pragma unroll
    for (int b = 0; b < KT; b++)
#pragma unroll
      for (int rr = 0; rr < 16; rr++) acc[a][b][rr] = 0.0f;

  // zero both buffers once: dY rows of missing output channels, guards
  for (int e = tid; e < 2 * buf_floats; e += 512) smem[e] = 0.0f;

  // staging descriptors (group-invariant up to a scalar base)
  const int RL4 = RL >> 2, BRBC4 = BRBC >> 2;
  const int pvecs = chn * g.BI * RL4, dvecs = orows * g.BI * BRBC4;
  unsigned p_src[2], d_src[DV];
  int d_lx[DV];
  int d_pool[DV], d_row[DV], d_o[DV];          // STEM: pooled-plane offset (group-invariant part), row in the band, channel
#pragma unroll
  for (int u = 0; u < 2; u++) {
    int v = tid + 512 * u;
    v = v < pvecs ? v : 0;
    int c = v / (g.BI * RL4);
    int rem = v - c * (g.BI * RL4);
    int im = rem / RL4, q4 = rem - im * RL4;
    p_src[u] = (unsigned)(((int64_t)im * g.C + c) * xplane + 4 * q4);
  }
#pragma unroll
  for (int u = 0; u < DV; u++) {
    int v = tid + 512 * u;
    const bool live = v < dvecs;
    v = live ? v : 0;
    int o = v / (g.BI * BRBC4);
    int rem = v - o * (g.BI * BRBC4);
    int im = rem / BRBC4, q4 = rem - im * BRBC4;
    d_src[u] = (unsigned)(((int64_t)im * g.O + o) * yplane + 4 * q4);
    d_lx[u] = live ? o * DLD + im * BRBC + 4 * q4 : -1;
    if constexpr (STEM) {
      const int f = 4 * q4, rl = f / g.ow, col = f - rl * g.ow;
      d_row[u] = rl;
      d_o[u] = o;
      d_pool[u] = (int)(((int64_t)im * g.O + o) * (yplane >> 2)) + (col >> 1);
    }
  }
  if constexpr (STEM) {
    // the constants of this block's output channels: batch statistics, affine pair, and the two means of the
    // BatchNorm backward formula from the statistics pass's per-slice sums
    if (tid < orows) {
      const int o = o0 + tid;
      float sdy = 0.0f, sdyx = 0.0f;
      for (int i = 0; i < g.st.S; i++) {
        sdy += g.st.partial[((int64_t)o * g.st.S + i) * 2];
        sdyx += g.st.partial[((int64_t)o * g.st.S + i) * 2 + 1];
      }
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (g.st.dgamma) g.st.dgamma[o] = sdyx;
        if (g.st.dbeta) g.st.dbeta[o] = sdy;
      }
      const float m = (float)g.N * (float)g.oh * (float)g.ow;
      stc[tid][0] = g.st.mean[o];
      stc[tid][1] = 1.0f / sqrtf(g.st.var[o] + g.st.eps);
      stc[tid][2] = g.st.gamma ? g.st.gamma[o] : 1.0f;
      stc[tid][3] = g.st.beta ? g.st.beta[o] : 0.0f;
      stc[tid][4] = sdy / m;
      stc[tid][5] = sdyx / m;
    }
  }

  const int my_groups = (g.ngroups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  f32x4 pv0, pv1, dv[DV];
  f32x2 yv[STEM ? DV : 1], gv[STEM ? DV : 1];
  int su_lo_cur = 0, r0_cur = 0, su_lo_nxt = 0, r0_nxt = 0;
#define WG_ISSUE(gi_)                                                                            \
  do {                                                                                           \
    int grp_ = blockIdx.x + (gi_) * gridDim.x;                                                   \
    int gimg_ = grp_ / g.gr;                                                                     \
    int n0_ = gimg_ * g.BI;                                                                      \
    r0_nxt = (grp_ - gimg_ * g.gr) * g.BR;                                                       \
    su_lo_nxt = min(max(0, r0_nxt * g.stride - g.pad), g.H - RLr);                               \
    const float* xb_ = g.x + ((int64_t)n0_ * g.C + cA) * xplane + (int64_t)su_lo_nxt * g.W;      \
    const float* yb_ = g.dy + ((int64_t)n0_ * g.O + o0) * yplane + (int64_t)r0_nxt * g.ow;       \
    _Pragma("unroll") for (int u_ = 0; u_ < DV; u_++) dv[u_] = *reinterpret_cast<const f32x4*>(yb_ + d_src[u_]); \
    if constexpr (STEM) {                                                                        \
      const int64_t pb_ = ((int64_t)n0_ * g.O + o0) * (yplane >> 2);                             \
      _Pragma("unroll") for (int u_ = 0; u_ < DV; u_++) {                                        \
        const int64_t at_ = pb_ + d_pool[u_] + (int64_t)((r0_nxt + d_row[u_]) >> 1) * (g.ow >> 1); \
        yv[u_] = *reinterpret_cast<const f32x2*>(g.st.y + at_);                                  \
        gv[u_] = *reinterpret_cast<const f32x2*>(g.st.g + at_);                                  \
      }                                                                                          \
    }                                                                                            \
    pv0 = *reinterpret_cast<const f32x4*>(xb_ + p_src[0]);                                       \
    pv1 = *reinterpret_cast<const f32x4*>(xb_ + p_src[1]);                                       \
  } while (0)
#define WG_COMMIT(buf)                                                                           \
  do {                                                                                           \
    float* Ds_ = smem + (buf) * buf_floats;                                                      \
    float* Xs_ = Ds_ + dy_floats + guard;                                                        \
    _Pragma("unroll") for (int u_ = 0; u_ < DV; u_++)                                            \
      if (d_lx[u_] >= 0) {                                                                       \
        f32x4 t_ = dv[u_];                                                                       \
        if constexpr (STEM) {                                                                    \
          const float* c_ = stc[d_o[u_]];                                                        \
          const float mean_ = c_[0], rsd_ = c_[1], w_ = c_[2], b_ = c_[3], k1_ = c_[4], k2_ = c_[5]; \
          _Pragma("unroll") for (int e_ = 0; e_ < 4; e_++) {                                     \
            const float z_ = dv[u_][e_];                                                         \
            const float dy_ = stem_dy(z_, yv[u_][e_ >> 1], gv[u_][e_ >> 1], mean_, rsd_, w_, b_); \
            const float xh_ = (z_ - mean_) * rsd_;                                               \
            t_[e_] = (dy_ - k1_ - xh_ * k2_) * w_ * rsd_;                                        \
          }                                                                                      \
        }                                                                                        \
        *reinterpret_cast<f32x4*>(Ds_ + d_lx[u_]) = t_;                                          \
      }                                                                                          \
    if (tid < pvecs) *reinterpret_cast<f32x4*>(Xs_ + 4 * tid) = pv0;                             \
    if (tid + 512 < pvecs) *reinterpret_cast<f32x4*>(Xs_ + 4 * (tid + 512)) = pv1;               \
  } while (0)

  WG_STAMP();
  __syncthreads();                             // zero fill done
  WG_STAMP();
  if (my_groups > 0) {
    WG_ISSUE(0);
    WG_COMMIT(0);
  }
  su_lo_cur = su_lo_nxt; r0_cur = r0_nxt;
  __syncthreads();
  WG_STAMP();

  const int R = g.BI * g.BR;          // pixel rows of a group
  const int ppr = g.BC >> 1;          // pixel pairs per row
  for (int gi_ = 0; gi_ < my_groups; gi_++) {
    const int buf = gi_ & 1;
    if (gi_ + 1 < my_groups) WG_ISSUE(gi_ + 1);
    {
      const float* Ds = smem + buf * buf_floats + l31 * DLD + half;
      const float* Xs = smem + buf * buf_floats + dy_floats + guard;
      for (int ri = wave; ri < R; ri += 8) {
        unsigned img = fast_div(ri, g.m_br);
        int r = ri - (int)img * g.BR;
        const int srow = (r0_cur + r) * g.stride;                  // source row of tap i = pad
        const float* xr = Xs + (int)img * RL + (srow - su_lo_cur) * g.W + half * g.stride;
        const float* dr = Ds + ri * g.BC;
        bool rok[KT];
#pragma unroll
        for (int t = 0; t < KT; t++) rok[t] = (unsigned)(srow + il[t]) < (unsigned)g.H;
#pragma unroll 2
        for (int p = 0; p < ppr; p++) {
          const int scol = (2 * p + half) * g.stride;              // source column of tap j = pad
          float a[OT], b[KT];
#pragma unroll
          for (int t = 0; t < OT; t++) a[t] = dr[t * 32 * DLD + 2 * p];
#pragma unroll
          for (int t = 0; t < KT; t++) {
            float v = xr[koffL[t] + 2 * p * g.stride];
            b[t] = (rok[t] && (unsigned)(scol + jl[t]) < (unsigned)g.W) ? v : 0.0f;
          }
#pragma unroll
          for (int i = 0; i < OT; i++)
#pragma unroll
            for (int j = 0; j < KT; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    WG_STAMP();
    if (gi_ + 1 < my_groups) WG_COMMIT(buf ^ 1);
    su_lo_cur = su_lo_nxt; r0_cur = r0_nxt;
    __syncthreads();
    WG_STAMP();
  }
#undef WG_ISSUE
#undef WG_COMMIT

  // cross-wave reduction through LDS, two tiles per round: every wave stores its partial of both
  // tiles, then wave w sums registers 4*(w&3).. of tile w>>2 over the eight partials
  float* red = smem;  // [2 tiles][8 waves][16][64]
  float* out = g.dst + (int64_t)blockIdx.x * g.O * KW;
  constexpr int NTILES = OT * KT;
#pragma unroll
  for (int round = 0; round < (NTILES + 1) / 2; round++) {
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      int tile = round * 2 + tt;
      if (tile < NTILES) {
        int ti = tile / KT, tj = tile % KT;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          float v = 0.0f;
#pragma unroll
          for (int a = 0; a < OT; a++)
#pragma unroll
            for (int b = 0; b < KT; b++)
              if (a == ti && b == tj) v = acc[a][b][rr];
          red[((tt * 8 + wave) * 16 + rr) * 64 + lane] = v;
        }
      }
    }
    __syncthreads();
    const int tt = wave >> 2;
    const int tile = round * 2 + tt;
    if (tile < NTILES) {
      int ti = tile / KT, tj = tile % KT;
      int kk = kk0 + tj * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int rr = (wave & 3) * 4 + q;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) v += red[((tt * 8 + w) * 16 + rr) * 64 + lane];
        int o = o0 + ti * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        if (o < g.O && kk < KW) out[(int64_t)o * KW + kk] = v;
      }
    }
  }
  WG_STAMP();
#undef WG_STAMP
}

// ---------------------------------------------------------------------------------------
// weight gradient of the few-channel 3x3 / stride 1 / pad 1 stem on 32-wide images (K = C * 9 <= 32, O <= 32):
// one 32 x 32 accumulator tile (output channel x (channel, tap)) per wave, the reduction index is the pixel.
// A lane of the A operand is an OUTPUT CHANNEL: it loads four consecutive pixels of its own channel plane of dY
// with one 16-byte load (the two half-waves take the two halves of an 8-pixel step), so dY never passes through
// LDS and -- STEM -- the channel's constants of the BatchNorm / ReLU / MaxPool backward are six registers of the
// lane; the image (12 KB with its zero padding, the tile of the forward kernel) is the only thing staged, and a
// (channel, tap) lane of the B operand reads its pixel from it.  A workgroup walks whole images, a wave every
// eighth step of an image with the loads of four steps in flight.  The generic kernel staged 256 pixels of dY per
// group and barrier, sixteen MFMAs per wave between two global round trips: 27.6 us per launch.
// ---------------------------------------------------------------------------------------
template <bool STEM>
__global__ __launch_bounds__(512) void k_wgrad_stem3(WGeo g) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 [[maybe_unused]] __attribute__((ext_vector_type(2)));
  constexpr int CH = kStemRows * kStemPitch;
  constexpr int CMAX = 3, NQ = CMAX * kStemRows * 8, QR = (NQ + 511) / 512;
  constexpr int DEPTH = 2;
  constexpr int ZP = 68, YP = 20;                    // LDS pitches of a channel's 64 pixels / 16 pooled pixels (bank spread)
  constexpr int WSLOT = 32 * ZP + 2 * 32 * YP;       // a wave's private transposition area (floats)
  extern __shared__ __attribute__((aligned(16))) float smem_w3[];
  float* Xl = smem_w3;                               // [CMAX][CH]
  float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(smem_w3 + CMAX * CH);   // [8][16][64]: over the waves' areas
  static_assert(8 * WSLOT >= 8 * 16 * 64, "the reduction fits the transposition areas");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Zs = smem_w3 + CMAX * CH + wave * WSLOT;
  float* Ys = Zs + 32 * ZP;
  float* Gs = Ys + 32 * YP;
  const int half = lane >> 5, l31 = lane & 31;
  const int C = g.C, H = g.H, O = g.O, KK = C * 9;
  const int nq = C * kStemRows * 8;
  STEM_STAMP(6); STEM_STAMP(0);
  const int o = min(l31, O - 1);                     // (a lane past the last channel repeats it: its rows are not stored)
  float mean = 0.f, rsd = 1.f, gw = 1.f, gb = 0.f, k1 = 0.f, k2 = 0.f;
  // STEM: the slices' sums of the statistics pass and the channel's constants are REQUESTED here and combined after
  // the first tile is in place (one request per iteration was a round trip per slice, 32 slices; two batches in
  // front of everything else still cost 7.5 us before the first load of the image left)
  constexpr int SMAX = 32;
  f32x4 prv = {0.f, 0.f, 0.f, 0.f};                  // 16 bytes of the [O][S][2] sums per thread, coalesced; transposed via LDS
  const int npr4 = (O * g.st.S * 2) >> 2;            // (S is even or O * S * 2 a multiple of 4: see the launcher)
  float c_mean = 0.f, c_var = 1.f, c_gamma = 1.f, c_beta = 0.f;
  if constexpr (STEM) {
    prv = *reinterpret_cast<const f32x4*>(g.st.partial + 4 * min(tid, npr4 - 1));
    c_mean = g.st.mean[o];
    c_var = g.st.var[o];
    if (g.st.gamma) c_gamma = g.st.gamma[o];
    if (g.st.beta) c_beta = g.st.beta[o];
  }
  // the (channel, tap) of this lane's column of the tile; a lane past K reads tap 0 (its columns are not stored)
  const int ke = l31 < KK ? l31 : 0;
  const int koff = (ke / 9) * CH + ((ke % 9) / 3 + half) * kStemPitch + (ke % 3) + 3;
  if (tid < CMAX * kStemRows * 2) {                  // the columns either side of the image
    const int t = tid >> 1;
    Xl[(t / kStemRows) * CH + (t % kStemRows) * kStemPitch + ((tid & 1) ? 36 : 3)] = 0.0f;
  }
  f32x16 acc;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) acc[rr] = 0.0f;
  STEM_STAMP(1);
  // a step is a PAIR of image rows (one per half-wave): a lane takes the whole 128-byte row of its channel plane at
  // once, and both rows of a pair share one row of the pooled planes.  (With 16 bytes per lane and step every line
  // was touched eight times from four waves and came from L2 most of those times: 10.5 / 26.5 us.)
  const int steps = H >> 1;                          // a wave takes steps wave, wave + 8, ...
  const int nsw = (steps - wave + 7) >> 3;
  for (int n = blockIdx.x; n < g.N; n += gridDim.x) {
    // a step's dY rows (32 channels x 64 pixels) and pooled rows (32 x 16) are requested COALESCED -- a wave load covers
    // whole lines: four channels x 256 bytes -- and turned into the lane-per-channel order through the wave's own LDS
    // area when the step is consumed.  (A lane loading its own channel's line 16 bytes at a time made every line eight
    // requests at the address unit: 12 000 line requests per CU and 11.5 us before the first MFMA.)
    const float* zn = g.dy + (int64_t)n * O * H * 32 + (lane & 15) * 4;
    const float* yn = nullptr;
    const float* gn = nullptr;
    if constexpr (STEM) {
      yn = g.st.y + (int64_t)n * O * (H >> 1) * 16 + (lane & 3) * 4;
      gn = g.st.g + (int64_t)n * O * (H >> 1) * 16 + (lane & 3) * 4;
    }
    f32x4 rz[DEPTH][8];
    f32x4 ry[STEM ? DEPTH : 1][2], rg[STEM ? DEPTH : 1][2];
    auto request = [&](int k, int slot) {
      const int sidx = min(wave + 8 * k, steps - 1);
#pragma unroll
      for (int q = 0; q < 8; q++)
        rz[slot][q] = *reinterpret_cast<const f32x4*>(zn + (int64_t)min(4 * q + (lane >> 4), O - 1) * H * 32 + sidx * 64);
      if constexpr (STEM) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int64_t at = (int64_t)min(16 * q + (lane >> 2), O - 1) * (H >> 1) * 16 + sidx * 16;
#ifdef DFHIP_STEM3_NOYG
          ry[slot][q] = (f32x4){1.f, 1.f, 1.f, 1.f}; rg[slot][q] = (f32x4){(float)at, 1.f, 1.f, 1.f};
#else
          ry[slot][q] = *reinterpret_cast<const f32x4*>(yn + at);
          rg[slot][q] = *reinterpret_cast<const f32x4*>(gn + at);
#endif
        }
      }
    };
    // the image tile: requests, then the first steps' operands, then the commit
    f32x4 tv[QR];
    {
      const float* xn = g.x + (int64_t)n * C * H * 32;
#pragma unroll
      for (int q0 = 0; q0 < QR; q0++) {
        const int q = min(tid + 512 * q0, nq - 1);
        const int pc = q & 7, t = q >> 3, lr = t % kStemRows, c = t / kStemRows;
        const int row = min(max(lr - 1, 0), H - 1);
        tv[q0] = *reinterpret_cast<const f32x4*>(xn + ((int64_t)c * H + row) * 32 + pc * 4);
      }
    }
    // every wave is done with the previous image's tile -- and has ISSUED its tile requests: they sit in front of
    // all the step requests at the address unit (a tile request queued behind seven other waves' steps arrived with
    // the last of the data, and the barrier below held the first MFMA until then); likewise the first steps of all
    // waves go in front of the second ones
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DEPTH; k++) {
      request(k, k);
      if (k + 1 < DEPTH) __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int q0 = 0; q0 < QR; q0++) {
      const int q = tid + 512 * q0;
      const int pc = q & 7, t = q >> 3, lr = t % kStemRows, c = t / kStemRows;
      const bool in = lr >= 1 && lr <= H;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      if (q < nq) *reinterpret_cast<f32x4*>(&Xl[c * CH + lr * kStemPitch + 4 + pc * 4]) = in ? tv[q0] : zero;
    }
    if constexpr (STEM) {
      if (n == (int)blockIdx.x && tid < npr4) {
        // float 4 tid + e of [O][S][2] -> row (4 tid + e) / (2 S) at pitch 2 S + 2 (bank spread of the reads below)
        float* pl = smem_w3 + CMAX * CH + 8 * WSLOT;       // [O][2 S + 2], its own area
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int f = 4 * tid + e, row = f / (2 * g.st.S);
          pl[f + 2 * row] = prv[e];
        }
      }
    }
    __syncthreads();
    if constexpr (STEM) {
      if (n == (int)blockIdx.x) {
        const float* pl = smem_w3 + CMAX * CH + 8 * WSLOT + o * (2 * g.st.S + 2);
        float sdy = 0.0f, sdyx = 0.0f;
#pragma unroll
        for (int u = 0; u < SMAX; u++)
          if (u < g.st.S) { sdy += pl[2 * u]; sdyx += pl[2 * u + 1]; }
        if (blockIdx.x == 0 && tid < O) {
          if (g.st.dgamma) g.st.dgamma[tid] = sdyx;
          if (g.st.dbeta) g.st.dbeta[tid] = sdy;
        }
        const float m = (float)g.N * (float)H * 32.0f;
        mean = c_mean;
        rsd = 1.0f / sqrtf(c_var + g.st.eps);
        gw = c_gamma;
        gb = c_beta;
        k1 = sdy / m;
        k2 = sdyx / m;
      }
    }
    STEM_STAMP(2);
    for (int kb = 0; kb < nsw; kb += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
        const int k = kb + u;
        if (k < nsw) {
          const int sidx = wave + 8 * k;
          const float* xr = Xl + koff + (2 * sidx) * kStemPitch;
          // through the wave's area: channel (4 q + lane / 16) x 4 pixels in, this lane's channel row out
          f32x4 zv[8], yv[STEM ? 4 : 1], gv[STEM ? 4 : 1];
#pragma unroll
          for (int q = 0; q < 8; q++)
            *reinterpret_cast<f32x4*>(Zs + (4 * q + (lane >> 4)) * ZP + (lane & 15) * 4) = rz[u][q];
          if constexpr (STEM) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
              *reinterpret_cast<f32x4*>(Ys + (16 * q + (lane >> 2)) * YP + (lane & 3) * 4) = ry[u][q];
              *reinterpret_cast<f32x4*>(Gs + (16 * q + (lane >> 2)) * YP + (lane & 3) * 4) = rg[u][q];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int q = 0; q < 8; q++) zv[q] = *reinterpret_cast<const f32x4*>(Zs + l31 * ZP + half * 32 + 4 * q);
          if constexpr (STEM) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              yv[q] = *reinterpret_cast<const f32x4*>(Ys + l31 * YP + 4 * q);
              gv[q] = *reinterpret_cast<const f32x4*>(Gs + l31 * YP + 4 * q);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int e = 0; e < 32; e++) {
            float t = zv[e >> 2][e & 3];
            if constexpr (STEM) {
              const float z_ = t;
              const float dy_ = stem_dy(z_, yv[e >> 3][(e >> 1) & 3], gv[e >> 3][(e >> 1) & 3], mean, rsd, gw, gb);
              const float xh_ = (z_ - mean) * rsd;
              t = (dy_ - k1 - xh_ * k2) * gw * rsd;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t, xr[e], acc, 0, 0, 0);
          }
        }
        request(k + DEPTH, u);                     // (no load under a condition: past the end the last step again)
      }
    }
  }
  STEM_STAMP(3);
  // the eight waves' tiles through LDS (over the transposition areas: every wave is done with its own); slab
  // [blockIdx.x][O][K]
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 16; rr++) red[wave][rr][lane] = acc[rr];
  __syncthreads();
  float* out = g.dst + (int64_t)blockIdx.x * O * KK;
#pragma unroll
  for (int h2 = 0; h2 < 2; h2++) {
    const int idx = tid + 512 * h2, rr = idx >> 6, ln = idx & 63;
    float v = 0.0f;
#pragma unroll
    for (int w8 = 0; w8 < 8; w8++) v += red[w8][rr][ln];
    const int oo = (rr & 3) + 8 * (rr >> 2) + 4 * (ln >> 5), kk = ln & 31;
    if (oo < O && kk < KK) out[(int64_t)oo * KK + kk] = v;
  }
  STEM_STAMP(4); STEM_STAMP(5); STEM_STAMP(7);
}

// ---------------------------------------------------------------------------------------
// weight gradient, tap-major kernel (3x3 and 1x1 layers with >= 16 input channels).
//
// dW[o][c][i][j] = sum over (n, r, s) of dY[n][o][r][s] * X[n][c][r*stride+i-pad][s*stride+j-pad].
// An accumulator tile is 32 output channels x 32 input channels of ONE tap, so a wave keeps
// k*k tiles (9 x 16 registers) and the reduction index of an MFMA step is a pixel: whether a tap
// falls into the padding is then the same for all 64 lanes and such steps are SKIPPED by a scalar
// branch instead of being multiplied by zero (2x2 images: 56 % of the dense steps, 4x4: 31 %).
// The two k slots of the 32x32x2 MFMA are the two images of an image pair at the same pixel, so
// both halves of the wave agree on validity as well.  The dY operand of a step is read once and
// used by all k*k taps.
//
//   grid = (pixel slabs, C/32, O/32); a block walks its groups (BI images x BR output rows),
//   staging X[32 channels] and dY[32 channels] of group g+1 into the other LDS buffer while group
//   g is in the matrix cores; the eight waves split the steps of a group; at the end they are
//   summed through LDS (float4 reduce-scatter, four taps per round) and written as one slab in
//   accumulator order, which k_wgrad_tap_sum adds over the pixel slabs and scatters to OIHW.
//
// LDS image: [channel][image pair][pixel][2 images], channel stride == 2 (mod 64) floats: the 64
// lanes of an operand read (32 channels x 2 images) hit 64 different banks.
// ---------------------------------------------------------------------------------------
struct TGeo {
  const float* x;
  const float* dy;
  float* slabs;
  int N, C, H, W, O, oh, ow;
  int k, stride, pad;
  int BI, BR, gr, ngroups;   // group = BI images x BR output rows (whole rows)
  int RLr;                   // source rows staged per group
  int XST, DST;              // channel strides in LDS (floats)
  int ksg;                   // MFMA steps per group = BI/2 * BR * ow
  unsigned m_brow, m_ow;     // magic dividers: BR*ow, ow
  unsigned m_xc, m_xp;       // staging: P*RL4, RL4
  unsigned m_dc, m_dp;       // staging: P*BRBC4, BRBC4
  unsigned long long* stamps;
};

template <int KS, int XV, int DV, bool DENSE>
__global__ __launch_bounds__(512) void k_conv_wgrad_tap(TGeo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int T = KS * KS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  [[maybe_unused]] int stamp_i = 0;
#ifdef DFHIP_STAMPS
#define TG_STAMP()                                                                              \
  do {                                                                                          \
    if (g.stamps && tid == 0 && stamp_i < 16)                                                   \
      g.stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + stamp_i++] = \
          __builtin_readcyclecounter();                                                         \
  } while (0)
#else
#define TG_STAMP() do { } while (0)
#endif
  TG_STAMP();
  const int c0 = blockIdx.y * 32, o0 = blockIdx.z * 32;
  const int cn = min(32, g.C - c0), on = min(32, g.O - o0);
  const int P = g.BI >> 1;
  const int RL = g.RLr * g.W;                 // staged floats per (image, channel)
  const int BRBC = g.BR * g.ow;
  const int RL4 = RL >> 2, BRBC4 = BRBC >> 2;
  const int buf_floats = 32 * (g.DST + g.XST);
  const int64_t xplane = (int64_t)g.H * g.W, yplane = (int64_t)g.oh * g.ow;

  // ---- staging descriptors: item = (channel, image pair, 4 pixels); group-invariant ----
  const int xitems = cn * P * RL4, ditems = on * P * BRBC4;
  unsigned x_src[XV], d_src[DV];
  int x_lx[XV], d_lx[DV];
#pragma unroll
  for (int u = 0; u < XV; u++) {
    int v = tid + 512 * u;
    const bool live = v < xitems;
    v = live ? v : 0;
    int c = (int)fast_div((unsigned)v, g.m_xc);
    int rem = v - c * (P * RL4);
    int pp = (int)fast_div((unsigned)rem, g.m_xp), q4 = rem - pp * RL4;
    x_src[u] = (unsigned)(((int64_t)(2 * pp) * g.C + c) * xplane + 4 * q4);
    x_lx[u] = live ? 32 * g.DST + c * g.XST + (pp * RL + 4 * q4) * 2 : -1;
  }
#pragma unroll
  for (int u = 0; u < DV; u++) {
    int v = tid + 512 * u;
    const bool live = v < ditems;
    v = live ? v : 0;
    int o = (int)fast_div((unsigned)v, g.m_dc);
    int rem = v - o * (P * BRBC4);
    int pp = (int)fast_div((unsigned)rem, g.m_dp), q4 = rem - pp * BRBC4;
    d_src[u] = (unsigned)(((int64_t)(2 * pp) * g.O + o) * yplane + 4 * q4);
    d_lx[u] = live ? o * g.DST + (pp * BRBC + 4 * q4) * 2 : -1;
  }
  const int64_t x_img = (int64_t)g.C * xplane, d_img = (int64_t)g.O * yplane;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int rr = 0; rr < 16; rr++) acc[t][rr] = 0.0f;

  const int my_groups = (g.ngroups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  f32x4 xa[XV], xb[XV], da[DV], db[DV];
  int su_lo_cur = 0, r0_cur = 0, su_lo_nxt = 0, r0_nxt = 0;
#define TG_ISSUE(gi_)                                                                            \
  do {                                                                                           \
    int grp_ = blockIdx.x + (gi_) * gridDim.x;                                                   \
    int gimg_ = grp_ / g.gr;                                                                     \
    int n0_ = gimg_ * g.BI;                                                                      \
    r0_nxt = (grp_ - gimg_ * g.gr) * g.BR;                                                       \
    su_lo_nxt = min(max(0, r0_nxt * g.stride - g.pad), g.H - g.RLr);                             \
    const float* xb_ = g.x + ((int64_t)n0_ * g.C + c0) * xplane + (int64_t)su_lo_nxt * g.W;      \
    const float* yb_ = g.dy + ((int64_t)n0_ * g.O + o0) * yplane + (int64_t)r0_nxt * g.ow;       \
    _Pragma("unroll") for (int u_ = 0; u_ < DV; u_++) {                                          \
      da[u_] = *reinterpret_cast<const f32x4*>(yb_ + d_src[u_]);                                 \
      db[u_] = *reinterpret_cast<const f32x4*>(yb_ + d_src[u_] + d_img);                         \
    }                                                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < XV; u_++) {                                          \
      xa[u_] = *reinterpret_cast<const f32x4*>(xb_ + x_src[u_]);                                 \
      xb[u_] = *reinterpret_cast<const f32x4*>(xb_ + x_src[u_] + x_img);                         \
    }                                                                                            \
  } while (0)
#define TG_PUT(base_, lx_, va_, vb_)                                                             \
  do {                                                                                           \
    if ((lx_) >= 0) {                                                                            \
      f32x2* q_ = reinterpret_cast<f32x2*>((base_) + (lx_));                                     \
      q_[0] = (f32x2){(va_).x, (vb_).x};                                                         \
      q_[1] = (f32x2){(va_).y, (vb_).y};                                                         \
      q_[2] = (f32x2){(va_).z, (vb_).z};                                                         \
      q_[3] = (f32x2){(va_).w, (vb_).w};                                                         \
    }                                                                                            \
  } while (0)
#define TG_COMMIT(buf)                                                                           \
  do {                                                                                           \
    float* B_ = smem + (buf) * buf_floats;                                                       \
    _Pragma("unroll") for (int u_ = 0; u_ < DV; u_++) TG_PUT(B_, d_lx[u_], da[u_], db[u_]);      \
    _Pragma("unroll") for (int u_ = 0; u_ < XV; u_++) TG_PUT(B_, x_lx[u_], xa[u_], xb[u_]);      \
  } while (0)

  TG_STAMP();
  if (my_groups > 0) {
    TG_ISSUE(0);
    TG_COMMIT(0);
  }
  su_lo_cur = su_lo_nxt; r0_cur = r0_nxt;
  __syncthreads();
  TG_STAMP();

  // this wave's share of the steps of a group
  const int e_lo = (wave * g.ksg) >> 3, e_hi = ((wave + 1) * g.ksg) >> 3;
  const int lane_d = l31 * g.DST + half;
  const int lane_x = 32 * g.DST + l31 * g.XST + half;
  const int W2 = 2 * g.W;

  for (int gi_ = 0; gi_ < my_groups; gi_++) {
    const int buf = gi_ & 1;
    if (gi_ + 1 < my_groups) TG_ISSUE(gi_ + 1);
    {
      const float* Dl = smem + buf * buf_floats + lane_d;
      const float* Xl = smem + buf * buf_floats + lane_x;
      // Two operand sets (this step / next step) used in turn: while the MFMAs of a step issue,
      // the scalar decode and the LDS reads of the next step are placed BETWEEN them in program
      // order (scheduling barriers keep them there) — the two waves of a SIMD run in lockstep
      // through the matrix pipe, so whatever sits before the first or after the last MFMA of
      // a step is time in which the pipe idles for both.
      float a_0, b_0[T], a_1, b_1[T];
      unsigned m_0, m_1;
      int e_n = e_lo;
      int pr_n = (int)fast_div((unsigned)e_lo, g.m_brow);
      int r_n, s_n;
      {
        const int rem_ = e_lo - pr_n * BRBC;
        r_n = (int)fast_div((unsigned)rem_, g.m_ow);
        s_n = rem_ - r_n * g.ow;
      }
      int xoff_n;
#define TG_DECODE(MOUT)                                                                          \
  do {                                                                                           \
    const int rs_ = (r0_cur + r_n) * g.stride - g.pad, cs_ = s_n * g.stride - g.pad;             \
    unsigned rowb_ = 0, colb_ = 0;                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < KS; i_++) {                                          \
      rowb_ |= ((unsigned)(rs_ + i_) < (unsigned)g.H ? 1u : 0u) << i_;                           \
      colb_ |= ((unsigned)(cs_ + i_) < (unsigned)g.W ? 1u : 0u) << i_;                           \
    }                                                                                            \
    unsigned mm_ = 0;                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < KS; i_++)                                            \
      mm_ |= ((rowb_ >> i_) & 1u) ? (colb_ << (i_ * KS)) : 0u;                                   \
    MOUT = mm_;                                                                                  \
    xoff_n = 2 * (pr_n * RL + (rs_ - su_lo_cur) * g.W + cs_);                                    \
  } while (0)
#define TG_ADVANCE()                                                                             \
  do {                                                                                           \
    if (e_n + 1 < e_hi) {                                                                        \
      e_n++; s_n++;                                                                              \
      if (s_n == g.ow) { s_n = 0; r_n++; if (r_n == g.BR) { r_n = 0; pr_n++; } }                 \
    }                                                                                            \
  } while (0)
#define TG_READ_ROW(i_, BN)                                                                      \
  do {                                                                                           \
    const float* xq_ = Xl + xoff_n + (i_) * W2;                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < KS; j_++) BN[(i_) * KS + j_] = xq_[2 * j_];          \
  } while (0)
#define TG_MFMA(t_, AC, BC, MC)                                                                  \
  do {                                                                                           \
    if (DENSE) {                                                                                 \
      const float bz_ = ((MC >> (t_)) & 1u) ? BC[t_] : 0.0f;                                     \
      acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32(AC, bz_, acc[t_], 0, 0, 0);                 \
    } else if ((MC >> (t_)) & 1u) {                                                              \
      acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32(AC, BC[t_], acc[t_], 0, 0, 0);              \
    }                                                                                            \
  } while (0)
#define TG_SB() __builtin_amdgcn_sched_barrier(0)
#define TG_STEP(AC, BC, MC, AN, BN, MN)                                                          \
  do {                                                                                           \
    TG_MFMA(0, AC, BC, MC); TG_SB();                                                             \
    TG_ADVANCE(); TG_DECODE(MN); TG_SB();                                                        \
    if (T > 1) { TG_MFMA(1 % T, AC, BC, MC); TG_SB(); }                                          \
    AN = Dl[2 * e_n]; TG_READ_ROW(0, BN); TG_SB();                                               \
    if (T > 1) {                                                                                 \
      TG_MFMA(2 % T, AC, BC, MC); TG_SB();                                                       \
      TG_READ_ROW(1 % KS, BN); TG_SB();                                                          \
      TG_MFMA(3 % T, AC, BC, MC); TG_SB();                                                       \
      TG_READ_ROW(2 % KS, BN); TG_SB();                                                          \
      TG_MFMA(4 % T, AC, BC, MC); TG_MFMA(5 % T, AC, BC, MC); TG_MFMA(6 % T, AC, BC, MC);        \
      TG_MFMA(7 % T, AC, BC, MC); TG_MFMA(8 % T, AC, BC, MC);                                    \
    }                                                                                            \
  } while (0)
#ifdef DFHIP_STAMPS
      if (g.stamps && lane == 0 && gi_ == 0)
        g.stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + 24 + wave] =
            __builtin_readcyclecounter();
#endif
      if (e_lo < e_hi) {
        TG_DECODE(m_0);
        a_0 = Dl[2 * e_n];
#pragma unroll
        for (int i = 0; i < KS; i++) TG_READ_ROW(i, b_0);
        for (int e = e_lo; e < e_hi; e += 2) {      // host: an even number of steps per wave
          TG_STEP(a_0, b_0, m_0, a_1, b_1, m_1);
          TG_SB();
          TG_STEP(a_1, b_1, m_1, a_0, b_0, m_0);
          TG_SB();
        }
      }
#ifdef DFHIP_STAMPS
      if (g.stamps && lane == 0 && gi_ == 0)
        g.stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + 16 + wave] =
            __builtin_readcyclecounter();
#endif
#undef TG_DECODE
#undef TG_ADVANCE
#undef TG_READ_ROW
#undef TG_MFMA
#undef TG_SB
#undef TG_STEP
    }
    TG_STAMP();
    if (gi_ + 1 < my_groups) TG_COMMIT(buf ^ 1);
    su_lo_cur = su_lo_nxt; r0_cur = r0_nxt;
    __syncthreads();
    TG_STAMP();
  }
#undef TG_ISSUE
#undef TG_PUT
#undef TG_COMMIT

  // reduce-scatter of the eight wave partials through LDS, four taps per round:
  // red[tt][wave][q][lane] (float4 = accumulator registers 4q..4q+3); wave w sums (tt, q) = 2w, 2w+1
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  float* slab = g.slabs + (((int64_t)blockIdx.x * gridDim.z + blockIdx.z) * gridDim.y + blockIdx.y) * (T * 1024);
#pragma unroll
  for (int round = 0; round < (T + 3) / 4; round++) {
    if (round > 0) __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
      if (round * 4 + tt < T) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const f32x16& A_ = acc[(round * 4 + tt) < T ? (round * 4 + tt) : 0];
          red[((tt * 8 + wave) * 4 + q) * 64 + lane] = (f32x4){A_[4 * q], A_[4 * q + 1], A_[4 * q + 2], A_[4 * q + 3]};
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
      const int combo = 2 * wave + cc;
      const int tt = combo >> 2, q = combo & 3;
      const int t = round * 4 + tt;
      if (t < T) {
        f32x4 v = red[((tt * 8 + 0) * 4 + q) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 8; w++) {
          f32x4 u = red[((tt * 8 + w) * 4 + q) * 64 + lane];
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        *reinterpret_cast<f32x4*>(slab + (int64_t)t * 1024 + (q * 64 + lane) * 4) = v;
      }
    }
  }
  TG_STAMP();
#undef TG_STAMP
}

// dW[o][c][t] = sum over pixel slabs of slab[ps][ot][ct][t][q][lane][comp]; accumulator register
// rr = 4q + comp of lane (half, col) is row comp + 8q + 4*half, column col of the 32 x 32 tile.
// A (virtual) block of 256 threads takes 256 / SL float4 items and SL slices of the slabs; slices
// are combined through LDS (part: 256 float4).
struct SlabSum {
  const float* slabs;
  float* dw;
  int tstride, toff;            // taps per (o, c) record of dw and the record's first tap this sum writes (T, 0 but for a tap row of a 5x5 kernel)
  int ps, SL, OT, CT, T, O, C;
  int nvb;                      // virtual blocks: ceil(OT*CT*T*256 / (256 / SL)); 0 = nothing to do
};

// load phase: the (at most eight) slabs this thread adds for virtual block vb (host: ps <= 8 * SL)
__device__ __forceinline__ void slab_sum_load(const SlabSum& a, int vb, float4 (&v)[8]) {
  const int ipb = 256 / a.SL;                     // items per block
  const int it = threadIdx.x % ipb, sl = threadIdx.x / ipb;
  const int64_t item = (int64_t)vb * ipb + it;
  const int64_t nitems = (int64_t)a.OT * a.CT * a.T * 256;
  const float4* src = reinterpret_cast<const float4*>(a.slabs) + item;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    int s = sl + a.SL * u;
    v[u] = (item < nitems && s < a.ps) ? src[(int64_t)s * nitems] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// finish phase: add, combine the slices through LDS, scatter to OIHW
__device__ __forceinline__ void slab_sum_finish(const SlabSum& a, int vb, const float4 (&v)[8], float4* part) {
  const int ipb = 256 / a.SL;
  const int it = threadIdx.x % ipb, sl = threadIdx.x / ipb;
  const int64_t item = (int64_t)vb * ipb + it;
  const int64_t nitems = (int64_t)a.OT * a.CT * a.T * 256;
  float4 acc = v[0];
#pragma unroll
  for (int u = 1; u < 8; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && item < nitems) {
    float4 r = part[it];
    for (int k = 1; k < a.SL; k++) {
      float4 q = part[k * ipb + it];
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    const int lane = (int)(item & 63), q = (int)((item >> 6) & 3);
    int64_t tile = item >> 8;
    const int t = (int)(tile % a.T);
    tile /= a.T;
    const int ct = (int)(tile % a.CT), ot = (int)(tile / a.CT);
    const int c = ct * 32 + (lane & 31);
    const int o = ot * 32 + 8 * q + 4 * (lane >> 5);
    if (c < a.C) {
      float vals[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (o + e < a.O) a.dw[((int64_t)(o + e) * a.C + c) * a.T + t] = vals[e];
    }
  }
  __syncthreads();                                // part is reused by the caller's next virtual block
}

__device__ __forceinline__ void slab_sum_vblock(const SlabSum& a, int vb, float4* part) {
  float4 v[8];
  slab_sum_load(a, vb, v);
  slab_sum_finish(a, vb, v, part);
}

__global__ __launch_bounds__(256) void k_wgrad_tap_sum(SlabSum a) {
  __shared__ float4 part[256];
  slab_sum_vblock(a, blockIdx.x, part);
}

// ---------------------------------------------------------------------------------------
// weight gradient, row-unit kernel: the tap-major decomposition above with the image width fixed
// at compile time.  The unit of work of a wave is one output row of one image pair: it reads the
// dY row (OW values per lane) and the k source rows (W values per lane each) with 16-byte LDS
// reads and then issues, straight-line and from registers only,
//     acc[i][j] += dY[s] (x) X[i][s*STRIDE + j - PAD]     for every (s, j) inside the image,
// so padding columns cost nothing (the combinations do not exist in the code), padding rows are
// one scalar branch per source row, and there is no per-step address or mask arithmetic left.
// LDS image: [channel][image][row][W] as in global memory (staging is a plain 16-byte copy);
// the two k slots of the MFMA are images 2p and 2p+1.  Channel stride == 4 (mod 64) floats
// (16-byte reads of 16 lanes cover the 64 banks), == 2 (mod 64) when rows are two floats wide.
// ---------------------------------------------------------------------------------------
struct RGeo {
  const float* x;
  const float* dy;
  float* slabs;
  int N, C, H, O, oh;
  int BI, BR, gr, ngroups;   // group = BI images x BR output rows
  int RLr;                   // source rows staged per group
  int XST, DST;              // channel strides in LDS (floats)
  int units;                 // BI/2 * BR
  unsigned m_br;             // magic divider: BR
  unsigned m_xc, m_xi;       // staging x: BI*RL4, RL4
  unsigned m_dc, m_di;       // staging dY: BI*B4, B4
  SlabSum pend;              // slab sum of the previous weight gradient, done by this launch's blocks
  unsigned long long* stamps;
  int single;                // grouped launches: one staging buffer instead of two
};

template <int NV>
__device__ __forceinline__ void lds_read_row(const float* p, float (&v)[NV]) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NV % 4 == 0) {
#pragma unroll
    for (int q = 0; q < NV / 4; q++) {
      f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < NV / 2; q++) {
      f32x2 t = *reinterpret_cast<const f32x2*>(p + 2 * q);
      v[2 * q] = t.x; v[2 * q + 1] = t.y;
    }
  }
}

// One wave per SIMD (256 threads): a single wave issues independent MFMAs back to back at the
// pipe rate, the operands of the next unit are read while the current one is in the matrix cores,
// and the final cross-wave sum moves half the bytes of an 8-wave block (one LDS round).
constexpr int kRowsWaves = 4;

template <int KS, int PAD, int STRIDE, int W, int OW, int UR, bool GROUPED>
__device__ __forceinline__ void wgrad_rows_body(const RGeo& g, float* smem, const int bx, const int by, const int bz,
                                                const int gdx, const int gdy, const int gdz) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int T = KS * KS;
  constexpr int NW = kRowsWaves, NT = 64 * NW;
  constexpr int SVX = 10, SVD = GROUPED ? 8 : 4;   // X / dY vectors per thread in flight across a group
  constexpr int NXR = (UR - 1) * STRIDE + KS;  // source rows under the UR output rows of a unit
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  [[maybe_unused]] int stamp_i = 0;
#ifdef DFHIP_STAMPS
#define RG_STAMP()                                                                              \
  do {                                                                                          \
    if (!GROUPED && g.stamps && tid == 0 && stamp_i < 16)                                       \
      g.stamps[(((size_t)bz * gdy + by) * gdx + bx) * 32 + stamp_i++] =                         \
          __builtin_readcyclecounter();                                                         \
  } while (0)
#else
#define RG_STAMP() do { } while (0)
#endif
  RG_STAMP();
  const int c0 = by * 32, o0 = bz * 32;
  const int cn = min(32, g.C - c0), on = min(32, g.O - o0);
  const int RL = g.RLr * W;                   // staged floats per (image, channel)
  const int BRBC = g.BR * OW;
  const int RL4 = RL >> 2, B4 = BRBC >> 2;
  const int xbase = 32 * g.DST;               // X image follows the dY image
  const int buf_floats = 32 * (g.DST + g.XST);
  const int64_t xplane = (int64_t)g.H * W, yplane = (int64_t)g.oh * OW;
  const int xitems = cn * g.BI * RL4, ditems = on * g.BI * B4;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int rr = 0; rr < 16; rr++) acc[t][rr] = 0.0f;

  // ---- staging: 16-byte copies, item v -> (channel, image, float4 of the contiguous run) ----
  // A thread without an item (v >= items) takes item 0 again: same source, same LDS destination,
  // same value as its owner writes.  Every load then has an UNCONDITIONAL consumer -- with the store
  // under a branch the compiler's wait-count pass leaves the load "pending" on the skipped path and
  // makes the first LDS read of the next group wait for the prefetch that was just issued.
  auto x_item = [&](int v, unsigned& src, int& lx) {
    int c = (int)fast_div((unsigned)v, g.m_xc);
    int rem = v - c * (g.BI * RL4);
    int im = (int)fast_div((unsigned)rem, g.m_xi), q4 = rem - im * RL4;
    src = (unsigned)(((int64_t)im * g.C + c) * xplane + 4 * q4);
    lx = xbase + c * g.XST + im * RL + 4 * q4;
  };
  auto d_item = [&](int v, unsigned& src, int& lx) {
    int o = (int)fast_div((unsigned)v, g.m_dc);
    int rem = v - o * (g.BI * B4);
    int im = (int)fast_div((unsigned)rem, g.m_di), q4 = rem - im * B4;
    src = (unsigned)(((int64_t)im * g.O + o) * yplane + 4 * q4);
    lx = o * g.DST + im * BRBC + 4 * q4;
  };
  auto lds_put = [&](float* base, int lx, f32x4 v) {
    if (((g.XST | g.DST) & 3) == 0) {
      *reinterpret_cast<f32x4*>(base + lx) = v;
    } else {
      f32x2* q = reinterpret_cast<f32x2*>(base + lx);
      q[0] = (f32x2){v.x, v.y};
      q[1] = (f32x2){v.z, v.w};
    }
  };
  int su_lo_cur = 0, r0_cur = 0, su_lo_nxt = 0, r0_nxt = 0;
  const float* xg_nxt = nullptr;
  const float* yg_nxt = nullptr;
  auto group_base = [&](int gi) {
    int grp = bx + gi * gdx;
    int gimg = grp / g.gr;
    int n0 = gimg * g.BI;
    r0_nxt = (grp - gimg * g.gr) * g.BR;
    su_lo_nxt = min(max(0, r0_nxt * STRIDE - PAD), g.H - g.RLr);
    xg_nxt = g.x + ((int64_t)n0 * g.C + c0) * xplane + (int64_t)su_lo_nxt * W;
    yg_nxt = g.dy + ((int64_t)n0 * g.O + o0) * yplane + (int64_t)r0_nxt * OW;
  };
  // descriptors of the vectors this thread keeps in flight across a group
  unsigned xs_src[SVX], ds_src[SVD];
  int xs_lx[SVX], ds_lx[SVD];
#pragma unroll
  for (int u = 0; u < SVX; u++) {
    int v = tid + NT * u;
    x_item(v < xitems ? v : 0, xs_src[u], xs_lx[u]);
  }
#pragma unroll
  for (int u = 0; u < SVD; u++) {
    int v = tid + NT * u;
    d_item(v < ditems ? v : 0, ds_src[u], ds_lx[u]);
  }
  // everything beyond SV vectors per thread is copied synchronously, six X and four dY vectors
  // per thread in flight per pass (host: groups after the first fit the SV vectors)
  auto stage_rest = [&](float* base, int firstx, int firstd) {
    int vx = tid + NT * firstx, vd = tid + NT * firstd;
    while (vx < xitems || vd < ditems) {
      f32x4 tx[6], td[4];
      int lx[6], ld[4];
#pragma unroll
      for (int u = 0; u < 6; u++) {
        int v = vx + NT * u;
        unsigned src;
        x_item(v < xitems ? v : 0, src, lx[u]);
        tx[u] = *reinterpret_cast<const f32x4*>(xg_nxt + src);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int v = vd + NT * u;
        unsigned src;
        d_item(v < ditems ? v : 0, src, ld[u]);
        td[u] = *reinterpret_cast<const f32x4*>(yg_nxt + src);
      }
#pragma unroll
      for (int u = 0; u < 6; u++) lds_put(base, lx[u], tx[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) lds_put(base, ld[u], td[u]);
      vx += NT * 6;
      vd += NT * 4;
    }
  };

  const int my_groups = (g.ngroups - bx + gdx - 1) / gdx;
  RG_STAMP();
  f32x4 xsv[SVX], dsv[SVD];
  if (my_groups > 0) {
    group_base(0);
#pragma unroll
    for (int u = 0; u < SVD; u++) dsv[u] = *reinterpret_cast<const f32x4*>(yg_nxt + ds_src[u]);
#pragma unroll
    for (int u = 0; u < SVX; u++) xsv[u] = *reinterpret_cast<const f32x4*>(xg_nxt + xs_src[u]);
  }
  // This block's share of the previous layer's slab sum: the slabs of its first two virtual blocks
  // are loaded now and added after the main loop (their latency hides behind it); any further
  // ones are done on the spot.
  constexpr int PVB = 2;
  float4 pv[PVB][8];
  const int nblk = gdx * gdy * gdz;
  const int me = (bz * gdy + by) * gdx + bx;
  if (!GROUPED && g.pend.nvb > 0) {
#pragma unroll
    for (int j = 0; j < PVB; j++)
      if (me + j * nblk < g.pend.nvb) slab_sum_load(g.pend, me + j * nblk, pv[j]);
    for (int vb = me + PVB * nblk; vb < g.pend.nvb; vb += nblk)
      slab_sum_vblock(g.pend, vb, reinterpret_cast<float4*>(smem));
  }
  if (my_groups > 0) {
#pragma unroll
    for (int u = 0; u < SVD; u++) lds_put(smem, ds_lx[u], dsv[u]);
#pragma unroll
    for (int u = 0; u < SVX; u++) lds_put(smem, xs_lx[u], xsv[u]);
    if constexpr (!GROUPED) stage_rest(smem, SVX, SVD);   // grouped plans always fit the SV vectors
  }
  su_lo_cur = su_lo_nxt; r0_cur = r0_nxt;
  __syncthreads();
  RG_STAMP();

  const int lane_d = l31 * g.DST + half * BRBC;
  const int lane_x = xbase + l31 * g.XST + half * RL;

  // grouped launches may run on ONE staging buffer (two workgroups share a CU's LDS): the commit of
  // the next group then waits for a barrier after the MFMAs instead of going to the other buffer
  const bool single = GROUPED && g.single;
  for (int gi_ = 0; gi_ < my_groups; gi_++) {
    const int buf = single ? 0 : (gi_ & 1);
    const bool more = gi_ + 1 < my_groups;
    bool committed = false;
    if (more) {
      group_base(gi_ + 1);
#pragma unroll
      for (int u = 0; u < SVD; u++) dsv[u] = *reinterpret_cast<const f32x4*>(yg_nxt + ds_src[u]);
#pragma unroll
      for (int u = 0; u < SVX; u++) xsv[u] = *reinterpret_cast<const f32x4*>(xg_nxt + xs_src[u]);
    }
    {
      const float* Dl = smem + buf * buf_floats + lane_d;
      const float* Xl = smem + buf * buf_floats + lane_x;
      // a source row (of the NXR under a unit) is read only if some (output row, tap row) uses it
      float a_n[UR][OW], b_n[NXR][W];
      int rs_n = 0;
      committed = false;
#define RG_LOAD(u_)                                                                              \
  do {                                                                                           \
    const int uu_ = (u_);                                                                        \
    const int pr_ = (int)fast_div((unsigned)uu_, g.m_br);                                        \
    const int r_ = (uu_ - pr_ * (g.BR / UR)) * UR;                                               \
    rs_n = (r0_cur + r_) * STRIDE - PAD;         /* source row of (output row 0, tap row 0) */   \
    _Pragma("unroll") for (int ur_ = 0; ur_ < UR; ur_++)                                         \
      lds_read_row<OW>(Dl + 2 * pr_ * BRBC + (r_ + ur_) * OW, a_n[ur_]);                         \
    _Pragma("unroll") for (int xr_ = 0; xr_ < NXR; xr_++)                                        \
      if (row_used(xr_)) lds_read_row<W>(Xl + 2 * pr_ * RL + (rs_n + xr_ - su_lo_cur) * W, b_n[xr_]); \
  } while (0)
      auto row_used = [](int xr) {
        for (int ur = 0; ur < UR; ur++)
          for (int i = 0; i < KS; i++)
            if (ur * STRIDE + i == xr) return true;
        return false;
      };
      if (wave < g.units) {
        RG_LOAD(wave);
        for (int u = wave; u < g.units; u += NW) {
          float a[UR][OW], b[NXR][W];
#pragma unroll
          for (int ur = 0; ur < UR; ur++)
#pragma unroll
            for (int s = 0; s < OW; s++) a[ur][s] = a_n[ur][s];
#pragma unroll
          for (int xr = 0; xr < NXR; xr++)
#pragma unroll
            for (int c = 0; c < W; c++) b[xr][c] = b_n[xr][c];
          const int rs = rs_n;
          RG_LOAD(min(u + NW, g.units - 1));       // next unit's operands land during the MFMAs below
#pragma unroll
          for (int ur = 0; ur < UR; ur++) {
#pragma unroll
            for (int i = 0; i < KS; i++) {
              if ((unsigned)(rs + ur * STRIDE + i) < (unsigned)g.H) {
#pragma unroll
                for (int s = 0; s < OW; s++)
#pragma unroll
                  for (int j = 0; j < KS; j++) {
                    const int col = s * STRIDE + j - PAD;
                    if (col >= 0 && col < W)
                      acc[i * KS + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                          a[ur][s], b[ur * STRIDE + i][col >= 0 && col < W ? col : 0], acc[i * KS + j], 0, 0, 0);
                  }
              }
              // grouped launches: the prefetched group goes to the other LDS buffer before the last
              // tap row of the wave's last unit (the loads have landed by then, the LDS writes run
              // under that row's MFMAs) instead of after it
              if (GROUPED && KS == 3 && ur == UR - 1 && i == KS - 2 && more && !single && u + NW >= g.units) {
                float* B_ = smem + (buf ^ 1) * buf_floats;
#pragma unroll
                for (int v = 0; v < SVD; v++) lds_put(B_, ds_lx[v], dsv[v]);
#pragma unroll
                for (int v = 0; v < SVX; v++) lds_put(B_, xs_lx[v], xsv[v]);
                committed = true;
              }
            }
          }
        }
      }
#undef RG_LOAD
    }
    RG_STAMP();
    if (single) __syncthreads();                   // every wave is done reading the only buffer
    if (more && !committed) {
      float* B_ = smem + (single ? 0 : (buf ^ 1)) * buf_floats;
#pragma unroll
      for (int u = 0; u < SVD; u++) lds_put(B_, ds_lx[u], dsv[u]);
#pragma unroll
      for (int u = 0; u < SVX; u++) lds_put(B_, xs_lx[u], xsv[u]);
      if constexpr (!GROUPED) stage_rest(B_, SVX, SVD);
    }
    su_lo_cur = su_lo_nxt; r0_cur = r0_nxt;
    __syncthreads();
    RG_STAMP();
  }

  if (!GROUPED && g.pend.nvb > 0) {
#pragma unroll
    for (int j = 0; j < PVB; j++)
      if (me + j * nblk < g.pend.nvb) slab_sum_finish(g.pend, me + j * nblk, pv[j], reinterpret_cast<float4*>(smem));
  }

  // cross-wave sum through LDS: red[t][wave][q][lane] (float4 = accumulator registers 4q..4q+3); wave w
  // then sums the (tap, q = w) combinations.  One round for the stand-alone kernel; RR taps per
  // round for grouped launches (the buffer is 16 KB per tap and two workgroups share the LDS).
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  float* slab = g.slabs + (((int64_t)bx * gdz + bz) * gdy + by) * (T * 1024);
  constexpr int RR = GROUPED ? (T >= 3 ? 3 : 1) : T;
#pragma unroll
  for (int t0 = 0; t0 < T; t0 += RR) {
    if (t0 > 0) __syncthreads();
#pragma unroll
    for (int t = t0; t < t0 + RR && t < T; t++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        red[(((t - t0) * NW + wave) * 4 + q) * 64 + lane] =
            (f32x4){acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
    __syncthreads();
#pragma unroll
    for (int t = t0; t < t0 + RR && t < T; t++) {
      const int q = wave;
      f32x4 v = red[(((t - t0) * NW + 0) * 4 + q) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; w++) {
        f32x4 u = red[(((t - t0) * NW + w) * 4 + q) * 64 + lane];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      *reinterpret_cast<f32x4*>(slab + (int64_t)t * 1024 + (q * 64 + lane) * 4) = v;
    }
  }
  RG_STAMP();
#undef RG_STAMP
}



template <int KS, int PAD, int STRIDE, int W, int OW, int UR>
__global__ __launch_bounds__(64 * kRowsWaves) void k_conv_wgrad_rows(RGeo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad_rows_body<KS, PAD, STRIDE, W, OW, UR, false>(g, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y,
                                                    gridDim.z);
}

// ---------------------------------------------------------------------------------------
// grouped weight gradient: ONE launch for the weight gradients of many layers (they depend only on
// tensors the backward sweep has already produced, so the front end may queue them and flush at the
// end of backward()).  A layer's (pixel slab, channel tile, output tile) workgroups are the row-unit
// kernel above; the launch concatenates the layers' workgroups.  What this buys over one launch per
// layer: no launch floor per layer, and with two workgroups resident per CU (LDS <= 80 KB, <= 256
// registers) one workgroup's first loads / cross-wave sum / slab stores hide behind the other's
// MFMAs -- neighbours are at different phases because they start at different times.  Each layer is
// cut into fewer, longer workgroups than a stand-alone launch would need to fill the chip, which
// also divides the slab traffic.  k_wgrad_group_sum then sums every layer's slabs in one launch.
// ---------------------------------------------------------------------------------------
constexpr int kMaxGroup = 21;        // (the launch's argument block stays below 4 KB)
struct GDesc {
  const float* x;
  const float* dy;
  float* slabs;
  int N, C, H, O, oh;
  int BI, BR, gr, ngroups, RLr, XST, DST, units;
  unsigned m_br, m_xc, m_xi, m_dc, m_di;
  int single, kind, ps, CT, OT;
  int wg_end;                // workgroups of layers 0..this one (exclusive end of this layer's range)
  int tap_row0;              // kind 13 (one tap row of a 5x5 kernel per entry): the tap row
  int direct;                // operands straight from global memory (units = units per block, m_br = magic(oh / UR))
  const float* dy2;          // direct, 3x3 / stride 2 only, or nullptr: dY of the block's 1x1 shortcut, whose weight
  float* slabs2;             //   gradient this layer's workgroups compute too (group_flush pairs the two queue entries)
  const float* aff;          // direct only, or nullptr: C x (scale, shift), the source operand is x * scale + shift
                             // (the BatchNorm a convolution applied to its input on load: dfhip_conv2d_forward_bn_input)
  // direct only, or nullptr: dy is the gradient of a BatchNorm OUTPUT whose backward pass has not been applied
  // (dfhip_bnbwd.h): the operand is A dy + B (z - mean) + Cc with z = bnz and (A, B, mean, Cc) = bntab[o]; bnz2 / bntab2: the
  // same for dy2
  const float* bnz;
  const float4* bntab;
  const float* bnz2;
  const float4* bntab2;
#ifdef DFHIP_STAMPS
  unsigned long long* dbg;   // diagnostic builds: per-workgroup cycles (load issue, MFMA section) of wave 0
#endif
};
struct WGroup {
  int n;
  unsigned long long* ticks;   // diagnostic builds (-DDFHIP_STAMPS): cycles per workgroup
  GDesc d[kMaxGroup];
};
// A convolution's bias gradient (nn/modules/conv.py:104-108 adds a (1, O, 1, 1) bias: its gradient is dY summed over images
// and pixels, tensor.py:462-483) as workgroups of the sweep's slab-sum launch: a workgroup per channel adds the channel's N
// planes up in a fixed order.  Stand-alone it is two launches on the ~4 us floor (k_channel_sum + _finish: 8 us per layer).
constexpr int kBiasSums = 4;
struct BiasSums {
  const float* src[kBiasSums];   // dY (N, C, HW)
  float* dst[kBiasSums];         // (C)
  int N[kBiasSums], C[kBiasSums], lq[kBiasSums];     // lq: log2(HW / 4)
  int blk_end[kBiasSums];        // running channel count
  int n;
};

struct SGroup {
  int n;
  int vb_end[kMaxGroup];
  SlabSum s[kMaxGroup];
  int bias_blocks;              // after the stem's and the head's workgroups: bias gradients, a workgroup per channel
  BiasSums bias;
  int stem_blocks;              // the first workgroups of the launch: the stem's one-pass backward, last step (dfhip_stem_combine.h)
  StemCombine stem;
  int head_blocks;              // then the classifier head's sums over the images (dfhip_head.h)
  HeadGrads head;
};

// Row-unit weight gradient with the operands straight from global memory (grouped launches, channel
// counts that are multiples of 32).  A lane's operand rows -- OW floats of dY, W floats of each source
// row -- are contiguous 8-64 byte runs of its own (image, channel) plane: it reads them with 16-byte
// global loads one unit ahead (a unit is 144 MFMAs = 4 us of matrix-core time, far more than the
// latency), so the LDS staging of a group, its commit and its barrier do not exist: what remains per
// unit is the MFMA stream.  Adjacent rows are taken by the four waves at the same time (their halo
// rows meet in L1).  LDS is only the cross-wave sum at the end.
template <int NV>
__device__ __forceinline__ void gl_read_row(const float* __restrict__ p, float (&v)[NV]) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NV % 4 == 0) {
#pragma unroll
    for (int q = 0; q < NV / 4; q++) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < NV / 2; q++) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(p + 2 * q);
      v[2 * q] = t.x; v[2 * q + 1] = t.y;
    }
  }
}

// The operands of a unit: NP image pairs x (UR rows of dY, the NXR source rows under them).  Two such sets exist and the
// unit loop is unrolled twice over them (MFMAs on one while the loads of the next unit land in the other): with ONE working
// set and a register copy per unit the copy sat behind `s_waitcnt vmcnt(0)` -- every unit began by waiting for the load
// issued nine MFMAs before its end -- and cost 64 moves of its own.
template <int KS, int STRIDE, int UR>
constexpr bool wg_row_used(int xr) {
  for (int ur = 0; ur < UR; ur++)
    for (int i = 0; i < KS; i++)
      if (ur * STRIDE + i == xr) return true;
  return false;
}
template <int KS, int STRIDE, int UR>
constexpr int wg_used_rows() {
  int n = 0;
  for (int xr = 0; xr < (UR - 1) * STRIDE + KS; xr++) n += wg_row_used<KS, STRIDE, UR>(xr) ? 1 : 0;
  return n;
}
template <int KS, int STRIDE, int UR>
constexpr int wg_nth_used_row(int n) {
  for (int xr = 0; xr < (UR - 1) * STRIDE + KS; xr++)
    if (wg_row_used<KS, STRIDE, UR>(xr) && n-- == 0) return xr;
  return 0;
}
template <int NP, int UR, int OW, int NXR, int W, bool PAIR, bool BN>
struct WgOps {
  float a[NP][UR][OW];
  float a2[PAIR ? NP : 1][PAIR ? UR : 1][PAIR ? OW : 1];    // PAIR: dY of the block's 1x1 shortcut (see wgrad_rows_direct)
  float b[NP][NXR][W];
  float az[BN ? NP : 1][BN ? UR : 1][BN ? OW : 1];          // BN: the BatchNorm input under dY (and under the shortcut's dY)
  float az2[(BN && PAIR) ? NP : 1][(BN && PAIR) ? UR : 1][(BN && PAIR) ? OW : 1];
};
struct WgUnit {
  int rs;            // source row of (output row 0, tap row 0)
  unsigned ybase;    // byte offsets of the unit's first pair / row in dY and (first pair) in x
  unsigned xbase;
};

// PAIR (3x3 / stride 2 only): the weight gradient of the block's 1x1 / stride-2 shortcut rides along.  The shortcut reads the
// pixels of the centre tap, so its gradient is one more accumulator fed by the same source operand and the shortcut's own dY
// (d.dy2, same shape as dY): the 1x1 layers stop being launches of their own kind -- units of 2-8 MFMAs per image pair, bound
// by their loads (5.5 % of the grouped launch's cycles for 0.7 % of its MFMAs).
// BN: dY (and the shortcut's dY) are gradients of BatchNorm outputs whose backward pass is applied here, on the operand: a lane
// is an output channel, so the channel's (A, B, mean, Cc) are four registers; the BatchNorm input z arrives through pieces of
// its own (same offsets as dY) and a row of dY becomes A dy + B (z - mean) + Cc when the MFMAs first reach it.
// BF (opt-in, dfhip_set_conv_matmul_dtype(1); outside the 1e-4 parity contract): the contraction on v_mfma_f32_32x32x8_bf16
// -- fp32 accumulators, the operands rounded to bf16 (nearest even) as they are packed.  The fp32 body multiplies one pixel
// of the image pair per instruction (K = 2: the two images of the lane halves); four consecutive pixels of a row are the
// bf16 instruction's K = 8, and in its operand layout lane half h supplies k = 4 h .. 4 h + 3 -- exactly the four values that
// lane already holds for those four pixels.  So nothing about the loads or the registers changes: four fp32 operands are
// packed into two registers and ONE instruction of 32 cycles replaces four of 64.  A column outside the image is a zero
// operand (the fp32 body skips the instruction).
// KR (round 6; default KS: the whole kernel): tap ROWS this call computes.  A 5x5 kernel's 25 accumulator tiles do not fit a wave's
// registers, so its weight gradient is queued as five entries of one tap row each (KR = 1, d.tap_row0 = 0 .. 4): five
// accumulators, per unit one row of dY and ONE source row (the row the tap row reads), the same MFMAs-per-load balance as the 3x3
// body; the slab sum writes each entry's five taps into its row of the (O, C, 5, 5) tensor.
template <int KS, int PAD, int STRIDE, int W, int OW, int UR, int NP, bool PAIR, bool BN, bool BF = false, int KR = KS>
__device__ __forceinline__ void wgrad_rows_direct(const GDesc& d, float* smem, const int bx, const int by,
                                                  const int bz) {
  static_assert(KR == KS || (KR == 1 && !PAIR), "a call computes the whole kernel or one tap row");
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  static_assert(!PAIR || (KS == 3 && STRIDE == 2 && PAD == 1), "the shortcut rides on the 3x3 / stride-2 body");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int T = KR * KS, NW = kRowsWaves;
  constexpr int NXR = (UR - 1) * STRIDE + KR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = d.H, oh = d.oh;
  const int row0 = KR == KS ? 0 : d.tap_row0;      // (a one-row call: the tap row it computes)
  const int upr = oh / UR;                         // units per group of NP image pairs
  const int rot = (upr & (upr - 1)) == 0 ? upr - 1 : 0;
  const int rsh = upr == 2 ? 1 : 0;
  const int U = (d.N / (2 * NP)) * upr;
  const int u_lo = bx * d.units, u_hi = min(U, u_lo + d.units);
  const int xplane = H * W, yplane = oh * OW;
  // A lane is a channel plane: its offset is a loop constant (the buffer load's vector offset), everything that changes
  // from piece to piece is the same for the whole wave (the scalar offset) -- no vector instruction per load
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2res =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PAIR ? d.dy2 : d.dy), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t zres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BN && d.bnz) ? d.bnz : d.dy), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t z2res =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BN && PAIR && d.bnz2) ? d.bnz2 : (PAIR ? d.dy2 : d.dy)), 0, 0x7fffffff,
                                        0x00020000);
  // (a descriptor without a table -- one of the two gradients of a PAIR is a plain one -- is the identity; its z pieces
  // re-read dY)
  float4 bt = make_float4(1.0f, 0.0f, 0.0f, 0.0f), bt2 = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
  if constexpr (BN) {
    if (d.bntab) bt = d.bntab[bz * 32 + l31];
    if constexpr (PAIR) { if (d.bntab2) bt2 = d.bntab2[bz * 32 + l31]; }
  }
  const int xlane = ((half * d.C + by * 32 + l31) * xplane) * 4;
  const int ylane = ((half * d.O + bz * 32 + l31) * yplane) * 4;
  const unsigned xpair = (unsigned)(2 * d.C * xplane) * 4u, ypair = (unsigned)(2 * d.O * yplane) * 4u;

  f32x16 acc[T], acc2;
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int rr = 0; rr < 16; rr++) acc[t][rr] = 0.0f;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) acc2[rr] = 0.0f;

#ifdef DFHIP_STAMPS
  unsigned long long dbg_issue = 0, dbg_mfma = 0;
  const unsigned long long dbg_t0 = __builtin_readcyclecounter();
  unsigned long long dbg_first = 0, dbg_last = 0;
#endif
  // The operand rows of a unit are 8- or 16-byte vectors ("pieces"), in the order the MFMAs want them: per pair UR * OW/VA of
  // dY, then W/VB per used source row.  A piece is one wave load that touches up to 64 different cache lines (a lane per
  // channel plane), which the address unit takes ~64-110 cycles to accept: they are spread evenly over the MFMAs of the unit
  // before.  Every load is unconditional (a row outside the image is read from the nearest row inside and never multiplied;
  // the last unit of a wave is loaded twice): the wait counts are then exact -- an MFMA waits for its own operand, not for
  // the youngest load.
  constexpr int VA = OW % 4 == 0 ? 4 : 2, VB = W % 4 == 0 ? 4 : 2;
  constexpr int PA = UR * (OW / VA), PBR = W / VB;
  constexpr int NUSED = wg_used_rows<KR, STRIDE, UR>();
  constexpr int PA2 = PAIR ? 2 * PA : PA;                   // (PAIR: the shortcut's dY rows follow the convolution's)
  constexpr int PAZ = BN ? PA2 : 0;                         // (BN: then the BatchNorm inputs under both)
  constexpr int PPP = PA2 + PAZ + NUSED * PBR;              // pieces per pair
  constexpr int NPIECE = NP * PPP;
  typedef WgOps<NP, UR, OW, NXR, W, PAIR, BN> Ops;
  auto piece = [&](Ops& o, const WgUnit& un, int k) {      // k is a compile-time constant at every call
    const int p = k / PPP, kk = k % PPP;
    if (kk < PA) {
      const int ur = kk / (OW / VA), q = kk % (OW / VA);
      const unsigned so = un.ybase + (unsigned)p * ypair + (unsigned)(ur * OW * 4);
      if constexpr (VA == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, ylane + 16 * q, so, 0));
        o.a[p][ur][4 * q] = t.x; o.a[p][ur][4 * q + 1] = t.y; o.a[p][ur][4 * q + 2] = t.z; o.a[p][ur][4 * q + 3] = t.w;
      } else {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yres, ylane + 8 * q, so, 0));
        o.a[p][ur][2 * q] = t.x; o.a[p][ur][2 * q + 1] = t.y;
      }
    } else if (PAIR && kk < PA2) {
      const int k2 = kk - PA, ur = k2 / (OW / VA), q = k2 % (OW / VA);
      const unsigned so = un.ybase + (unsigned)p * ypair + (unsigned)(ur * OW * 4);
      if constexpr (VA == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y2res, ylane + 16 * q, so, 0));
        o.a2[p][ur][4 * q] = t.x; o.a2[p][ur][4 * q + 1] = t.y; o.a2[p][ur][4 * q + 2] = t.z; o.a2[p][ur][4 * q + 3] = t.w;
      } else {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(y2res, ylane + 8 * q, so, 0));
        o.a2[p][ur][2 * q] = t.x; o.a2[p][ur][2 * q + 1] = t.y;
      }
    } else if (BN && kk < PA2 + PA) {
      const int k2 = kk - PA2, ur = k2 / (OW / VA), q = k2 % (OW / VA);
      const unsigned so = un.ybase + (unsigned)p * ypair + (unsigned)(ur * OW * 4);
      if constexpr (VA == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zres, ylane + 16 * q, so, 0));
        o.az[p][ur][4 * q] = t.x; o.az[p][ur][4 * q + 1] = t.y; o.az[p][ur][4 * q + 2] = t.z; o.az[p][ur][4 * q + 3] = t.w;
      } else {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(zres, ylane + 8 * q, so, 0));
        o.az[p][ur][2 * q] = t.x; o.az[p][ur][2 * q + 1] = t.y;
      }
    } else if (BN && PAIR && kk < PA2 + PAZ) {
      const int k2 = kk - PA2 - PA, ur = k2 / (OW / VA), q = k2 % (OW / VA);
      const unsigned so = un.ybase + (unsigned)p * ypair + (unsigned)(ur * OW * 4);
      if constexpr (VA == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z2res, ylane + 16 * q, so, 0));
        o.az2[p][ur][4 * q] = t.x; o.az2[p][ur][4 * q + 1] = t.y; o.az2[p][ur][4 * q + 2] = t.z; o.az2[p][ur][4 * q + 3] = t.w;
      } else {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(z2res, ylane + 8 * q, so, 0));
        o.az2[p][ur][2 * q] = t.x; o.az2[p][ur][2 * q + 1] = t.y;
      }
    } else {
      // the kk-th used source row
      const int xr = wg_nth_used_row<KR, STRIDE, UR>((kk - PA2 - PAZ) / PBR);
      const int q = (kk - PA2 - PAZ) % PBR;
      const int row = min(max(un.rs + xr, 0), H - 1);
      const unsigned so = un.xbase + (unsigned)p * xpair + (unsigned)(row * W * 4);
      if constexpr (VB == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, xlane + 16 * q, so, 0));
        o.b[p][xr][4 * q] = t.x; o.b[p][xr][4 * q + 1] = t.y; o.b[p][xr][4 * q + 2] = t.z; o.b[p][xr][4 * q + 3] = t.w;
      } else {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xres, xlane + 8 * q, so, 0));
        o.b[p][xr][2 * q] = t.x; o.b[p][xr][2 * q + 1] = t.y;
      }
    }
  };
  auto unit_of = [&](int uu) {                               // rows / offsets of unit uu
    const int pg = (int)fast_div((unsigned)uu, d.m_br);
    // a wave takes every fourth unit: without the rotation by the group index it would meet the same rows of every image --
    // the waves with the image's first and last rows (a tap row less) idled 8 % (16 x 16 planes) to 16 % (8 x 8) of the
    // workgroup at the cross-wave sum
    int ri = uu - pg * upr + ((pg >> rsh) & rot);
    ri -= ri >= upr ? upr : 0;
    const int r = ri * UR;
    WgUnit un;
    un.rs = r * STRIDE - PAD + row0;
    un.ybase = (unsigned)pg * (unsigned)NP * ypair + (unsigned)(r * OW * 4);
    un.xbase = (unsigned)pg * (unsigned)NP * xpair;
    return un;
  };
  // affine source operand: a lane is a channel plane, so (scale, shift) are two registers of the lane; applied to a row in
  // place when the MFMAs first reach it (rows outside the image are never multiplied, columns outside it do not exist in
  // the code)
  const bool affine = d.aff != nullptr;
  float aff_sc = 1.0f, aff_sh = 0.0f;
  if (affine) { aff_sc = d.aff[2 * (by * 32 + l31)]; aff_sh = d.aff[2 * (by * 32 + l31) + 1]; }
  constexpr int NIT = NP * UR * KR * OW;                     // (pair, ur, i, s) iterations of a unit, <= KS MFMAs each
  // MFMAs of the unit in `c` (rows from cu.rs), the pieces of unit `nu` into `n` on the way
  auto compute = [&](Ops& c, const WgUnit& cu, Ops& n, const WgUnit& nu) {
#ifdef DFHIP_STAMPS
    const unsigned long long tq1 = __builtin_readcyclecounter();
    if (!dbg_first) dbg_first = tq1;
#endif
#pragma unroll
    for (int p = 0; p < NP; p++) {
#pragma unroll
      for (int ur = 0; ur < UR; ur++) {
#pragma unroll
        for (int i = 0; i < KR; i++) {
          // (a scalar branch per tap row even where the unit is the whole plane and the answer is known at compile time: as one
          // basic block the unit of a 2 x 2 plane took 50 600 cycles per workgroup against 44 700 -- same box, -DDFHIP_STAMPS)
          const bool rowok = (unsigned)(cu.rs + ur * STRIDE + i) < (unsigned)H;
          if constexpr (BN) {
            if (i == 0) {                                    // first use of output row ur of pair p
#pragma unroll
              for (int s = 0; s < OW; s++) c.a[p][ur][s] = bn_bwd_point(c.a[p][ur][s], c.az[p][ur][s], bt);
              if constexpr (PAIR) {
#pragma unroll
                for (int s = 0; s < OW; s++) c.a2[p][ur][s] = bn_bwd_point(c.a2[p][ur][s], c.az2[p][ur][s], bt2);
              }
            }
          }
          if (affine && (ur == 0 || i + STRIDE >= KR)) {     // first use of source row ur * STRIDE + i
#pragma unroll
            for (int cc = 0; cc < W; cc++) c.b[p][ur * STRIDE + i][cc] = fmaf(c.b[p][ur * STRIDE + i][cc], aff_sc, aff_sh);
          }
          if constexpr (BF && OW % 4 == 0) {
#pragma unroll
            for (int s4 = 0; s4 < OW; s4 += 4) {
              const int it0 = ((p * UR + ur) * KR + i) * OW + s4;
#pragma unroll
              for (int k = (it0 * NPIECE + NIT - 1) / NIT; k < ((it0 + 4) * NPIECE + NIT - 1) / NIT; k++) piece(n, nu, k);
              __builtin_amdgcn_sched_barrier(0);
              if (rowok) {
                const bf16x4 pa = {(__bf16)c.a[p][ur][s4], (__bf16)c.a[p][ur][s4 + 1], (__bf16)c.a[p][ur][s4 + 2],
                                   (__bf16)c.a[p][ur][s4 + 3]};
#pragma unroll
                for (int j = 0; j < KS; j++) {
                  float bv[4];
#pragma unroll
                  for (int e = 0; e < 4; e++) {
                    const int col = (s4 + e) * STRIDE + j - PAD;
                    bv[e] = (col >= 0 && col < W) ? c.b[p][ur * STRIDE + i][col >= 0 && col < W ? col : 0] : 0.0f;
                  }
                  const bf16x4 pb = {(__bf16)bv[0], (__bf16)bv[1], (__bf16)bv[2], (__bf16)bv[3]};
                  acc[i * KS + j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(pa, pb, acc[i * KS + j], 0, 0, 0);
                }
                if constexpr (PAIR) {                        // the centre tap's pixels (row 2 (r + ur), columns 2 s: always inside)
                  if (i == 1) {
                    const bf16x4 pa2 = {(__bf16)c.a2[p][ur][s4], (__bf16)c.a2[p][ur][s4 + 1], (__bf16)c.a2[p][ur][s4 + 2],
                                        (__bf16)c.a2[p][ur][s4 + 3]};
                    const bf16x4 pb2 = {(__bf16)c.b[p][ur * STRIDE + 1][2 * s4], (__bf16)c.b[p][ur * STRIDE + 1][2 * s4 + 2],
                                        (__bf16)c.b[p][ur * STRIDE + 1][2 * s4 + 4], (__bf16)c.b[p][ur * STRIDE + 1][2 * s4 + 6]};
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(pa2, pb2, acc2, 0, 0, 0);
                  }
                }
              }
            }
          } else
#pragma unroll
          for (int s = 0; s < OW; s++) {
            const int it = ((p * UR + ur) * KR + i) * OW + s;
            // pieces k with floor(k * NIT / NPIECE) == it
#pragma unroll
            for (int k = (it * NPIECE + NIT - 1) / NIT; k < ((it + 1) * NPIECE + NIT - 1) / NIT; k++) piece(n, nu, k);
            __builtin_amdgcn_sched_barrier(0);
            if (rowok) {
#pragma unroll
              for (int j = 0; j < KS; j++) {
                const int col = s * STRIDE + j - PAD;
                if (col >= 0 && col < W)
                  acc[i * KS + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                      c.a[p][ur][s], c.b[p][ur * STRIDE + i][col >= 0 && col < W ? col : 0], acc[i * KS + j], 0, 0, 0);
              }
              if constexpr (PAIR) {                          // the centre tap's pixel (row 2 (r + ur), column 2 s: always inside)
                if (i == 1)
                  acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a2[p][ur][s], c.b[p][ur * STRIDE + 1][2 * s], acc2, 0, 0, 0);
              }
            }
          }
        }
      }
    }
#ifdef DFHIP_STAMPS
    const unsigned long long tq2 = __builtin_readcyclecounter();
    dbg_mfma += tq2 - tq1;
    dbg_last = tq2;
#endif
  };
  {
    Ops s0, s1;
    int u = u_lo + wave;
    if (u < u_hi) {
      WgUnit u0 = unit_of(u), u1;
#pragma unroll
      for (int k = 0; k < NPIECE; k++) piece(s0, u0, k);
      while (true) {
        u1 = unit_of(min(u + NW, u_hi - 1));
        compute(s0, u0, s1, u1);
        u += NW;
        if (u >= u_hi) break;
        u0 = unit_of(min(u + NW, u_hi - 1));
        compute(s1, u1, s0, u0);
        u += NW;
        if (u >= u_hi) break;
      }
    }
  }
#ifdef DFHIP_STAMPS
  // (prologue: entry .. first MFMA section; between: first .. last section minus the sections themselves)
  if (d.dbg && tid == 0) { d.dbg[2 * blockIdx.x] = ((dbg_first - dbg_t0) << 32) | ((dbg_last - dbg_first) & 0xffffffffull); d.dbg[2 * blockIdx.x + 1] = dbg_mfma; }
#endif
  // cross-wave sum through LDS, RR taps per round (as the staged body of a grouped launch)
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  float* slab = d.slabs + (((int64_t)bx * d.OT + bz) * d.CT + by) * (T * 1024);
  constexpr int RR = T >= 3 ? 3 : 1;
#pragma unroll
  for (int t0 = 0; t0 < T; t0 += RR) {
    if (t0 > 0) __syncthreads();
#pragma unroll
    for (int t = t0; t < t0 + RR && t < T; t++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        red[(((t - t0) * NW + wave) * 4 + q) * 64 + lane] =
            (f32x4){acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
    __syncthreads();
#pragma unroll
    for (int t = t0; t < t0 + RR && t < T; t++) {
      const int q = wave;
      f32x4 v = red[(((t - t0) * NW + 0) * 4 + q) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; w++) {
        const f32x4 uu = red[(((t - t0) * NW + w) * 4 + q) * 64 + lane];
        v.x += uu.x; v.y += uu.y; v.z += uu.z; v.w += uu.w;
      }
      *reinterpret_cast<f32x4*>(slab + (int64_t)t * 1024 + (q * 64 + lane) * 4) = v;
    }
  }
  if constexpr (PAIR) {                                      // the shortcut's tile: one more round
    float* slab2 = d.slabs2 + (((int64_t)bx * d.OT + bz) * d.CT + by) * 1024;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++)
      red[(wave * 4 + q) * 64 + lane] = (f32x4){acc2[4 * q], acc2[4 * q + 1], acc2[4 * q + 2], acc2[4 * q + 3]};
    __syncthreads();
    f32x4 v = red[(0 * 4 + wave) * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; w++) {
      const f32x4 uu = red[(w * 4 + wave) * 64 + lane];
      v.x += uu.x; v.y += uu.y; v.z += uu.z; v.w += uu.w;
    }
    *reinterpret_cast<f32x4*>(slab2 + (wave * 64 + lane) * 4) = v;
  }
}

template <int KS, int PAD, int STRIDE, int W, int OW, int UR, int NP, bool BF = false>
__device__ __forceinline__ void wgrad_direct_any(const GDesc& d, float* smem, const int bx, const int by, const int bz) {
  const bool bn = d.bnz != nullptr || d.bnz2 != nullptr;
  if constexpr (KS == 3 && STRIDE == 2) {
    if (d.dy2) {
      if (bn) wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, true, true, BF>(d, smem, bx, by, bz);
      else wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, true, false, BF>(d, smem, bx, by, bz);
      return;
    }
  }
  if constexpr (W <= 16) {       // (the BatchNorm operand exists for the planes of the CIFAR models' residual stages)
    if (bn) { wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, false, true, BF>(d, smem, bx, by, bz); return; }
  }
  wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, false, false, BF>(d, smem, bx, by, bz);
}

// (id, kernel, pad, stride, W, OW, output rows per unit, image pairs per unit of the direct body: units of few MFMAs are
// bound by the latency of their loads and by the hand-over between units, so small planes take several pairs)
#define DF_ROWS_KINDS(X) \
  X(0, 3, 1, 1, 16, 16, 1, 1) X(1, 3, 1, 1, 8, 8, 2, 1) X(2, 3, 1, 1, 4, 4, 4, 1) X(3, 3, 1, 1, 2, 2, 2, 4)      \
  X(4, 3, 1, 2, 16, 8, 1, 1) X(5, 3, 1, 2, 8, 4, 2, 1) X(6, 3, 1, 2, 4, 2, 2, 2)                              \
  X(7, 1, 0, 2, 16, 8, 1, 4) X(8, 1, 0, 2, 8, 4, 1, 8) X(9, 1, 0, 2, 4, 2, 1, 8)                              \
  X(10, 3, 1, 1, 32, 32, 1, 1) X(11, 3, 1, 2, 32, 16, 1, 1) X(12, 1, 0, 2, 32, 16, 1, 2)

// (one wave per SIMD, like the stand-alone kernel: the row-unit body needs ~470 registers, and built
// for two resident workgroups -- 256 registers -- it spills 120-210 of them)
template <bool BF>
__device__ __forceinline__ void wgrad_group_body(const WGroup& a, float* smem) {
  const int b = blockIdx.x;
#ifdef DFHIP_STAMPS
  const unsigned long long t_begin = __builtin_readcyclecounter();
  const unsigned long long r_begin = wall_clock64();          // (constant 100 MHz: the clock the cycles ran at, CU occupancy)
#endif
  int li = 0;
  while (li + 1 < a.n && b >= a.d[li].wg_end) li++;
  const GDesc& d = a.d[li];
  const int local = b - (li ? a.d[li - 1].wg_end : 0);
  const int bx = local % d.ps, rest = local / d.ps;
  const int by = rest % d.CT, bz = rest / d.CT;
  RGeo g;
  g.x = d.x; g.dy = d.dy; g.slabs = d.slabs;
  g.N = d.N; g.C = d.C; g.H = d.H; g.O = d.O; g.oh = d.oh;
  g.BI = d.BI; g.BR = d.BR; g.gr = d.gr; g.ngroups = d.ngroups; g.RLr = d.RLr;
  g.XST = d.XST; g.DST = d.DST; g.units = d.units;
  g.m_br = d.m_br; g.m_xc = d.m_xc; g.m_xi = d.m_xi; g.m_dc = d.m_dc; g.m_di = d.m_di;
  g.pend.nvb = 0;
  g.stamps = nullptr;
  g.single = d.single;
  switch (d.kind) {
    // (BF: the bf16 contraction exists for the direct body on rows of 4 and more output pixels, W <= 16 -- the CIFAR
    // models' planes; everything else in a bf16 launch runs the fp32 body)
#define DF_CASE(ID, KSV, PADV, SV, WV, OWV, URV, NPV)                                              \
  case ID:                                                                                         \
    if (d.direct) wgrad_direct_any<KSV, PADV, SV, WV, OWV, URV, NPV, (BF && OWV % 4 == 0 && WV <= 16)>(d, smem, bx, by, bz); \
    else wgrad_rows_body<KSV, PADV, SV, WV, OWV, URV, true>(g, smem, bx, by, bz, d.ps, d.CT, d.OT); \
    break;
    DF_ROWS_KINDS(DF_CASE)
#undef DF_CASE
    case 13:      // one tap row of a 5x5 / stride 1 / pad 2 kernel on 16 x 16 planes (queue_wgrad_k5)
      wgrad_rows_direct<5, 2, 1, 16, 16, 1, 1, false, false, BF, 1>(d, smem, bx, by, bz);
      break;
    default: break;
  }
#ifdef DFHIP_STAMPS
  if (a.ticks && threadIdx.x == 0) {
    a.ticks[b] = __builtin_readcyclecounter() - t_begin;
    a.ticks[3 * (size_t)gridDim.x + b] = r_begin;
    a.ticks[4 * (size_t)gridDim.x + b] = wall_clock64();
  }
#endif
}

__global__ __launch_bounds__(64 * kRowsWaves) void k_wgrad_group(WGroup a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad_group_body<false>(a, smem);
}

// the same launch with the bf16 contraction (dfhip_set_conv_matmul_dtype(1)): a kernel of its own, so that the fp32
// launch's code and registers are exactly what they were
__global__ __launch_bounds__(64 * kRowsWaves) void k_wgrad_group_bf16(WGroup a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad_group_body<true>(a, smem);
}

// Slab sum of a grouped launch, tap-major per thread: a thread owns one (tile, q, lane) position --
// four output channels x one source channel -- and ALL taps of it, so that what it stores is T
// consecutive floats per output channel and a wave stores 32 x T x 4 contiguous bytes (the per-tap
// version above leaves 4-byte stores 4*T bytes apart: PMC 7x write amplification).  SL slices of
// the slabs are summed in a fixed order and combined through LDS.
// (TT = the number of taps as a compile-time constant: with `if (t < T)` on the run-time T every tap's load sat in a
// basic block of its own and the 36 loads of a round became dependent round trips -- 16 us for 24 MB)
template <int TT>
__device__ __forceinline__ void wgrad_group_sum_body(const SlabSum& d, const int vb, float4* part) {
  const int ipb = 256 / d.SL;
  const int it = threadIdx.x % ipb, sl = threadIdx.x / ipb;
  const int item = vb * ipb + it;                         // (tile, q, lane)
  const int nitem = d.OT * d.CT * 256;
  const int64_t slab_f4 = (int64_t)nitem * TT;            // float4s per slab
  float4 acc[TT];
#pragma unroll
  for (int t = 0; t < TT; t++) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (item < nitem) {
    const float4* src = reinterpret_cast<const float4*>(d.slabs) + (int64_t)(item >> 8) * TT * 256 + (item & 255);
    for (int s0 = sl; s0 < d.ps; s0 += 4 * d.SL) {       // four slabs' loads in flight
      float4 v[4][TT];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int s = s0 + u * d.SL;
        const float4* p = src + (int64_t)(s < d.ps ? s : s0) * slab_f4;
#pragma unroll
        for (int t = 0; t < TT; t++) v[u][t] = p[t * 256];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float on = s0 + u * d.SL < d.ps ? 1.0f : 0.0f;     // (a slab past the end re-read slab s0: times 0)
#pragma unroll
        for (int t = 0; t < TT; t++) {
          acc[t].x = fmaf(v[u][t].x, on, acc[t].x); acc[t].y = fmaf(v[u][t].y, on, acc[t].y);
          acc[t].z = fmaf(v[u][t].z, on, acc[t].z); acc[t].w = fmaf(v[u][t].w, on, acc[t].w);
        }
      }
    }
  }
  if (d.SL > 1) {
#pragma unroll
    for (int t = 0; t < TT; t++) part[t * 256 + threadIdx.x] = acc[t];
    __syncthreads();
  }
  if (sl == 0 && item < nitem) {
    for (int k = 1; k < d.SL; k++)
#pragma unroll
      for (int t = 0; t < TT; t++) {
        const float4 v = part[t * 256 + k * ipb + it];
        acc[t].x += v.x; acc[t].y += v.y; acc[t].z += v.z; acc[t].w += v.w;
      }
    const int lane = item & 63, q = (item >> 6) & 3, tile = item >> 8;
    const int ct = tile % d.CT, ot = tile / d.CT;
    const int c = ct * 32 + (lane & 31);
    const int o = ot * 32 + 8 * q + 4 * (lane >> 5);
    if (c < d.C) {
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (o + e < d.O) {
          float* dst = d.dw + ((int64_t)(o + e) * d.C + c) * d.tstride + d.toff;
#pragma unroll
          for (int t = 0; t < TT; t++)
            dst[t] = e == 0 ? acc[t].x : (e == 1 ? acc[t].y : (e == 2 ? acc[t].z : acc[t].w));
        }
    }
  }
}

// 256 threads, one channel: 16-byte loads, sixteen in flight per thread; lanes, then waves, added up in a fixed order
__device__ __forceinline__ void bias_sum_body(const BiasSums& b, int blk, float* lds) {
  typedef float f32x4b __attribute__((ext_vector_type(4)));
  int i = 0;
  while (i + 1 < b.n && blk >= b.blk_end[i]) i++;
  const int c = blk - (i ? b.blk_end[i - 1] : 0);
  const int lq = b.lq[i], q = 1 << lq, C = b.C[i];
  const int F = b.N[i] << lq;                        // 16-byte pieces of the channel
  const f32x4b* src = reinterpret_cast<const f32x4b*>(b.src[i]);
  f32x4b acc = {0.0f, 0.0f, 0.0f, 0.0f};
  constexpr int U = 16;
  for (int f0 = (int)threadIdx.x; f0 < F; f0 += 256 * U) {
    f32x4b v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int f = min(f0 + 256 * u, F - 1);
      v[u] = src[(((int64_t)(f >> lq) * C + c) << lq) + (f & (q - 1))];
    }
    asm volatile("" ::: "memory");                 // (a compiler barrier: the selection DAG sinks plain loads to their uses)
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float on = f0 + 256 * u < F ? 1.0f : 0.0f;
      acc += v[u] * on;
    }
  }
  float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  t = head_wave_sum(t);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) lds[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) b.dst[i][c] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(256) void k_wgrad_group_sum(SGroup a) {
  __shared__ float4 part[256 * 9];
  // (the stem's workgroups are a chain of dependent round trips over its 256 slabs: dispatched first, they run beside the
  // streaming ones instead of behind them)
  if ((int)blockIdx.x < a.stem_blocks) {             // (uniform per workgroup)
    stem_combine_body(a.stem, (int)blockIdx.x);
    return;
  }
  if ((int)blockIdx.x < a.stem_blocks + a.head_blocks) {
    static_assert(sizeof(part) >= kHeadLdsFloats * sizeof(float), "the head's sums fit the slab sum's LDS");
    head_grads_body<true>(a.head, (int)blockIdx.x - a.stem_blocks, reinterpret_cast<float*>(part));
    return;
  }
  if ((int)blockIdx.x < a.stem_blocks + a.head_blocks + a.bias_blocks) {
    bias_sum_body(a.bias, (int)blockIdx.x - a.stem_blocks - a.head_blocks, reinterpret_cast<float*>(part));
    return;
  }
  const int b = (int)blockIdx.x - a.stem_blocks - a.head_blocks - a.bias_blocks;
  int li = 0, hi = a.n - 1;                          // the first layer with b < vb_end[li] (ascending: a binary search)
  while (li < hi) {
    const int mid = (li + hi) >> 1;
    if (b >= a.vb_end[mid]) li = mid + 1; else hi = mid;
  }
  const SlabSum& d = a.s[li];
  const int vb = b - (li ? a.vb_end[li - 1] : 0);
  if (d.T == 9) wgrad_group_sum_body<9>(d, vb, part);
  else if (d.T == 5) wgrad_group_sum_body<5>(d, vb, part);
  else wgrad_group_sum_body<1>(d, vb, part);
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
  DF_LAUNCH((k_conv_wgrad<OT, KT>), grid, dim3(256), lds, stream(), g);
  DF_LAUNCH_CHECK();
  if (gx > 1) {
    DF_LAUNCH(k_flat_slab_sum, dim3((unsigned)df_ceil_div(n, 256)), dim3(1024), 0, stream(), slabs,
                       dw, n, gx);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

template <int OT, int KT, int DV, bool STEM = false>
int launch_wgrad_v3(WGeo g, float* dw, size_t lds, int gx, int ktg, int otg) {
  const int64_t n = (int64_t)g.O * g.C * g.k * g.k;
  float* slabs = dw;
  if (gx > 1) {
    slabs = workspace((size_t)n * gx * sizeof(float));
    if (!slabs) return DFHIP_OUT_OF_MEMORY;
  }
  g.dst = slabs;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_wgrad_v3<OT, KT, DV, STEM>,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  dim3 grid((unsigned)gx, (unsigned)ktg, (unsigned)otg);
  static int stamp_mode = -1;
  if (stamp_mode < 0) stamp_mode = getenv("DFHIP_CONV_STAMP") ? atoi(getenv("DFHIP_CONV_STAMP")) : 0;
  const size_t nb = (size_t)gx * ktg * otg;
  g.stamps = nullptr;
  if (stamp_mode) {
    g.stamps = (unsigned long long*)workspace2(nb * 32 * 8);
    (void)hipMemsetAsync(g.stamps, 0, nb * 32 * 8, stream());
  }
  DF_LAUNCH((k_conv_wgrad_v3<OT, KT, DV, STEM>), grid, dim3(512), lds, stream(), g);
  DF_LAUNCH_CHECK();
  if (stamp_mode) {
    std::vector<unsigned long long> h(nb * 32);
    (void)hipStreamSynchronize(stream());
    (void)hipMemcpy(h.data(), g.stamps, nb * 32 * 8, hipMemcpyDeviceToHost);
    double seg[32] = {0};
    int nseg = 0;
    for (size_t b = 0; b < nb; b++) {
      const unsigned long long* r = &h[b * 32];
      int n = 0;
      while (n < 32 && r[n]) n++;
      for (int i = 1; i < n; i++) seg[i] += (double)(r[i] - r[i - 1]);
      nseg = std::max(nseg, n);
    }
    fprintf(stderr, "[stampw] OT %d KT %d grid %d x %d x %d groups %d (BI %d BR %d) lds %zu; mean segments:", OT, KT, gx,
            ktg, otg, g.ngroups, g.BI, g.BR, lds);
    for (int i = 1; i < nseg; i++) fprintf(stderr, " %.0f", seg[i] / nb);
    fprintf(stderr, "\n");
  }
  if (gx > 1) {
    DF_LAUNCH(k_flat_slab_sum, dim3((unsigned)df_ceil_div(n, 256)), dim3(1024), 0, stream(), slabs,
                       dw, n, gx);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

// the few-channel stem on 32-wide images (k_wgrad_stem3); -1 when the shape is not one it takes
int run_wgrad_stem3(WGeo g, float* dw, bool stem) {
  if (g.k != 3 || g.stride != 1 || g.pad != 1 || g.W != 32 || g.ow != 32 || g.oh != g.H) return -1;
  if (g.C < 1 || g.C > 3 || g.O < 1 || g.O > 32 || g.H > 32 || (g.H & 1)) return -1;
  if (stem && (g.st.S < 1 || g.st.S > 32 || (g.O * g.st.S) % 2 != 0 || g.O * g.st.S * 2 > 4 * 512)) return -1;
  const int64_t n = (int64_t)g.O * g.C * 9;
  const int gx = (int)std::min<int64_t>(g.N, num_cus());
  float* slabs = dw;
  if (gx > 1) {
    slabs = workspace((size_t)n * gx * sizeof(float));
    if (!slabs) return DFHIP_OUT_OF_MEMORY;
  }
  g.dst = slabs;
  const size_t lds = (size_t)(3 * kStemRows * kStemPitch + 8 * (32 * 68 + 2 * 32 * 20) + 32 * 66) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DF_HIP(hipFuncSetAttribute((const void*)k_wgrad_stem3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DF_HIP(hipFuncSetAttribute((const void*)k_wgrad_stem3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  if (stem) DF_LAUNCH(k_wgrad_stem3<true>, dim3((unsigned)gx), dim3(512), lds, stream(), g);
  else DF_LAUNCH(k_wgrad_stem3<false>, dim3((unsigned)gx), dim3(512), lds, stream(), g);
  DF_LAUNCH_CHECK();
#ifdef DFHIP_STEM_STAMPS
  {
    static int calls = 0;
    if (++calls % 10 == 0) {
      std::vector<unsigned long long> h((size_t)gx * 8 * 8);
      (void)hipStreamSynchronize(stream());
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stem_stamps), h.size() * sizeof(unsigned long long));
      double seg[6] = {0};
      unsigned long long r0 = ~0ull, r1 = 0;
      const size_t nw = (size_t)gx * 8;
      for (size_t b = 0; b < nw; b++) {
        const unsigned long long* r = &h[b * 8];
        for (int k = 1; k <= 4; k++) seg[k] += (double)(r[k] - r[k - 1]);
        r0 = std::min(r0, r[6]); r1 = std::max(r1, r[7]);
      }
      fprintf(stderr, "[stem3-stamp] grid %d stem %d: span %.2f us; mean cycles per wave: constants %.0f, tile %.0f, steps %.0f, "
              "reduce %.0f\n", gx, (int)stem, (double)(r1 - r0) / 100.0, seg[1] / nw, seg[2] / nw, seg[3] / nw, seg[4] / nw);
    }
  }
#endif
  if (gx > 1) {
    DF_LAUNCH(k_flat_slab_sum, dim3((unsigned)df_ceil_div(n, 256)), dim3(1024), 0, stream(), slabs, dw, n, gx);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

// returns -1 when the shape is not eligible for the main weight-gradient kernel
int run_wgrad_v3(WGeo g, float* dw, bool stem = false) {
  {
    const int st3 = run_wgrad_stem3(g, dw, stem);
    if (st3 >= 0) return st3;
  }
  const int k2 = g.k * g.k;
  const int KW = g.C * k2;
  if (g.ow & 1) return -1;
  if ((g.H * g.W) % 4 != 0 || (g.oh * g.ow) % 4 != 0) return -1;   // 16-byte aligned channel planes
  const int OT = g.O <= 32 ? 1 : 2;
  // (column tiles per workgroup: with 64 output channels and a 5x5 kernel -- 800 columns -- five tiles per workgroup
  // halve the dY operand reads per MFMA)
  const int KT = (g.k == 1 || KW <= 32) ? 1 : (g.O <= 32 ? 5 : ((g.k == 5 && KW >= 320) ? 5 : 2));
  const int CH = std::min(g.C, (KT * 32 + k2 - 2) / k2 + 1);
  int gi = 1;
  size_t lds = 0;
  bool ok = false;
  for (int tp : {256, 128, 64, 32}) {
    plan_groups(g.N, g.oh, g.ow, tp, g.BI, g.BR, g.BC, gi, g.gr, g.gc);
    if (g.BC != g.ow || g.gc != 1 || g.N % g.BI != 0 || g.oh % g.BR != 0) continue;
    if ((g.BI * g.BR) % 8) continue;               // rows deal evenly to the eight waves
    int RLr = g.gr == 1 ? g.H : std::min(g.H, (g.BR - 1) * g.stride + g.k);
    int RL = RLr * g.W, BRBC = g.BR * g.BC;
    if (RL % 4 || BRBC % 4) continue;
    if (g.gr > 1 && (g.W % 4 || g.ow % 4)) continue;
    g.PE = g.BI * RL;
    int BMP = g.BI * BRBC;
    int guard = (g.k * g.W + g.k + 3) & ~3;
    size_t buf = (size_t)OT * 32 * (BMP + 4) + 2 * guard + (size_t)CH * g.PE;
    size_t red = (size_t)2 * 8 * 16 * 64;
    lds = std::max(2 * buf, red) * 4;
    int pvecs = CH * g.PE / 4, dvecs = OT * 32 * BMP / 4;
    if (lds <= 150 * 1024 && pvecs <= 1024 && dvecs <= 512 * 4) { ok = true; break; }
  }
  if (!ok) return -1;
  g.CH = CH;
  g.ngroups = gi * g.gr;
  g.m_br = magic_u32(g.BR);
  const int ktg = (int)df_ceil_div(KW, KT * 32), otg = (int)df_ceil_div(g.O, OT * 32);
  int gx = (int)std::max<int64_t>(1, std::min<int64_t>(g.ngroups, (int64_t)num_cus() / ((int64_t)ktg * otg)));
  if (stem) {
    // (the stem form: one tile of output channels, whole-row bands so that a float4 never straddles a pooling row)
    if (OT != 1 || (g.ow & 3) || (g.oh & 1)) return -1;
    if (KT == 5) return launch_wgrad_v3<1, 5, 4, true>(g, dw, lds, gx, ktg, otg);
    return launch_wgrad_v3<1, 1, 4, true>(g, dw, lds, gx, ktg, otg);
  }
  if (OT == 1 && KT == 5) return launch_wgrad_v3<1, 5, 4>(g, dw, lds, gx, ktg, otg);
  if (OT == 1) return launch_wgrad_v3<1, 1, 4>(g, dw, lds, gx, ktg, otg);
  if (KT == 5) return launch_wgrad_v3<2, 5, 4>(g, dw, lds, gx, ktg, otg);
  if (KT == 2) return launch_wgrad_v3<2, 2, 4>(g, dw, lds, gx, ktg, otg);
  return launch_wgrad_v3<2, 1, 4>(g, dw, lds, gx, ktg, otg);
}

// ---- deferred slab sum (see dfhip_internal.h) ----
struct DeferredSum {
  SlabSum a;
  bool active = false;
  int slot = -1;
};
DeferredSum g_def;

SlabSum make_slab_sum(const float* slabs, float* dw, int ps, int OT, int CT, int T, int O, int C) {
  SlabSum a;
  a.slabs = slabs; a.dw = dw; a.ps = ps; a.OT = OT; a.CT = CT; a.T = T; a.O = O; a.C = C;
  a.tstride = T; a.toff = 0;
  a.SL = 1;                                       // slices: at most eight slabs per thread
  while (a.SL < 256 && a.SL * 8 < ps) a.SL *= 2;
  a.nvb = (int)df_ceil_div((int64_t)OT * CT * T * 256, 256 / a.SL);
  return a;
}

int launch_slab_sum(const SlabSum& a) {
  DF_LAUNCH(k_wgrad_tap_sum, dim3((unsigned)a.nvb), dim3(256), 0, stream(), a);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// ---- row-unit weight gradient: plan + launch ----
template <int KS, int PAD, int STRIDE, int W, int OW, int UR>
int launch_wgrad_rows(RGeo g, float* dw, size_t lds, int ps, int CT, int OT) {
  const int T = KS * KS;
  // the slabs go to the scratch slot the pending sum does not read
  const int slot = g_def.active ? 1 - g_def.slot : 0;
  g.slabs = workspace_slot(slot, (size_t)ps * OT * CT * T * 1024 * sizeof(float));
  if (!g.slabs) return DFHIP_OUT_OF_MEMORY;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_wgrad_rows<KS, PAD, STRIDE, W, OW, UR>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  dim3 grid((unsigned)ps, (unsigned)CT, (unsigned)OT);
  g.stamps = nullptr;
  g.pend = SlabSum{};
  if (g_def.active) {                             // this launch's blocks do the previous layer's sum
    g.pend = g_def.a;
    g_def.active = false;
    g_def.slot = -1;
  }
  static int stamp_mode = -1;
  if (stamp_mode < 0) stamp_mode = getenv("DFHIP_CONV_STAMP") ? atoi(getenv("DFHIP_CONV_STAMP")) : 0;
  const size_t nb = (size_t)ps * CT * OT;
  if (stamp_mode) {
    g.stamps = (unsigned long long*)workspace2(nb * 32 * 8);
    (void)hipMemsetAsync(g.stamps, 0, nb * 32 * 8, stream());
  }
  DF_LAUNCH((k_conv_wgrad_rows<KS, PAD, STRIDE, W, OW, UR>), grid, dim3(64 * kRowsWaves), lds, stream(), g);
  DF_LAUNCH_CHECK();
  if (stamp_mode) {
    std::vector<unsigned long long> h(nb * 32);
    (void)hipStreamSynchronize(stream());
    (void)hipMemcpy(h.data(), g.stamps, nb * 32 * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
    double seg[32] = {0};
    int nseg = 0;
    for (size_t b = 0; b < nb; b++) {
      const unsigned long long* r = &h[b * 32];
      int n = 0;
      while (n < 32 && r[n]) n++;
      if (!n) continue;
      tmin = std::min(tmin, r[0]); smax = std::max(smax, r[0]); tmax = std::max(tmax, r[n - 1]);
      for (int i = 1; i < n; i++) seg[i] += (double)(r[i] - r[i - 1]);
      nseg = std::max(nseg, n);
    }
    fprintf(stderr, "[stampr] k %d W %d UR %d grid %d x %d x %d groups %d units %d (BI %d BR %d) lds %zu pend %d; span %lld last start +%lld; mean segments:",
            KS, W, UR, ps, CT, OT, g.ngroups, g.units, g.BI, g.BR, lds, g.pend.nvb, (long long)(tmax - tmin), (long long)(smax - tmin));
    for (int i = 1; i < nseg; i++) fprintf(stderr, " %.0f", seg[i] / nb);
    fprintf(stderr, "\n");
  }
  SlabSum mine = make_slab_sum(g.slabs, dw, ps, OT, CT, T, g.O, g.C);
  g_def.a = mine;
  g_def.active = true;
  g_def.slot = slot;
  return DFHIP_OK;
}

// LDS channel stride: the smallest value >= n that is r (mod 64)
static int bank_stride_r(int n, int r) { return ((n - r + 63) / 64) * 64 + r; }

// ---- grouped weight gradients (dfhip_wgrad_group_begin / _end): queue + one launch ----
struct GQueued {
  GDesc d;
  float* dw;
  int T;
  size_t lds;
  double flops, bytes;
  double work;                   // estimated cycles of one of its workgroups (set by group_flush)
  int rides_on;                  // -1, or the queue index of the 3x3 / stride-2 layer whose workgroups compute this 1x1 shortcut too
  int tstride, toff;             // (kind 13) the slab sum's record stride and first tap in dw; 0, 0 = T, 0
  int W, ow, k, stride, pad;     // the rest of the call's arguments (a small group is re-planned as stand-alone launches)
  int U;                         // (direct) units of one (channel tile, output tile): ps = ceil(U / units) -- group_flush may re-cut
};
std::vector<GQueued> g_queue;
bool g_group_on = false;
int g_matmul_dtype = 0;                           // dfhip_set_conv_matmul_dtype: 0 fp32, 1 bf16 operands (grouped weight gradients)
struct StemPending { bool active = false; StemCombine c; int blocks = 0; } g_stem_pending;
struct HeadPending { bool active = false; HeadGrads h; } g_head_pending;
BiasSums g_bias_pending = {};                     // n > 0: bias gradients waiting for the group's slab-sum launch
int run_wgrad_rows(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int O, int oh, int ow,
                   int k, int stride, int pad, const float* aff = nullptr, const float* bnz = nullptr,
                   const float* bntab = nullptr);
int bn_bwd_materialise(const float* dy, const float* z, const float* tab, float* dz, int N, int O, int HW);
int wgrad_affine_materialised(const float* x, const float* aff, const float* dy, float* dw, int N, int C, int H, int W,
                              int O, int oh, int ow, int k, int stride, int pad);

// Side-stream flushes (DFHIP_WGRAD_SIDE=n): while a backward sweep is running, every n queued weight gradients are
// launched on a second stream that forks off the library stream at that point -- the matrix-core-bound grouped
// launch then runs BESIDE the rest of the sweep (input gradients and BatchNorm backward: latency / HBM bound) instead
// of after it.  dfhip_wgrad_group_end (and any flush) joins the side stream back.  Inside a capture the fork / join
// become graph edges.
hipStream_t g_side = nullptr;
hipEvent_t g_side_fork = nullptr;
struct SidePending { hipEvent_t done; float* slabs; std::vector<std::pair<const float*, const float*>> dws; };
std::vector<SidePending> g_side_pending;
std::vector<hipEvent_t> g_side_events;            // recycled join events
hipStream_t g_group_stream = nullptr;             // stream the group kernels go to (nullptr: the library stream)
float* g_group_slabs = nullptr;                   // slab area supplied by the side flush (nullptr: scratch slot 0)

// Estimated cycles of one workgroup of a grouped layer (operands from global memory): the launch dispatches the longest
// workgroups first.  Cycles of a wave per unit were measured with the in-kernel counters (-DDFHIP_STAMPS, DFHIP_CONV_STAMP=1;
// batch 256): the MFMA issue slots of the unit plus what its operand loads and hand-over cost -- units of few MFMAs (1x1
// kernels, 2 x 2 planes) are bound by the latency of their loads, not by their MFMA count; ~14 000 cycles around the MFMA
// stream (first operands, cross-wave sum, slab stores).  (Cutting every layer to a common cycle target instead of 32
// workgroups per layer was measured in round 3: the shorter tail came back as prologues / epilogues of the extra workgroups.)
static double replan_direct(const GQueued& e) {
  static const double kUnitCycles[14] = {9260, 8720, 7640, 5560, 4780, 4680, 4490, 9400, 7200, 2840, 19000, 9800, 9400, 4500};
  const double unit_cycles = e.d.kind >= 0 && e.d.kind < 14 ? kUnitCycles[e.d.kind] : 5000.0;
  return e.d.units / 4 * unit_cycles + 14000.0;
}
static double unit_cycles_of(const GQueued& e) { GQueued t = e; t.d.units = 4; return replan_direct(t) - 14000.0; }

// A group that holds a 5x5 kernel's tap rows (queue_wgrad_k5 cut them for 7/8 of the chip as if they were alone) AND other
// layers, all of it about one round of workgroups: every workgroup starts at once and the launch lasts as long as the longest
// one -- so the layers are re-cut in proportion to their matrix cycles, to end together.  (The CNN workload's conv3 + conv2:
// 32 workgroups of 285 k cycles beside 220 of 120 k, a launch of 130 us, after the cut 64 x 140 k + 190 x 140 k.)  Sweeps of many
// layers (ResNet-18: 2.4 rounds, planned per layer) are left as queued.
static void recut_one_round(std::vector<GQueued>& q) {
  bool rows5 = false, others = false;
  int64_t wgs = 0;
  double total = 0.0;
  for (const auto& e : q) {
    if (!e.d.direct || e.U <= 0 || e.d.aff || (e.k == 1 && e.stride == 2)) return;
    (e.d.kind == 13 ? rows5 : others) = true;
    wgs += (int64_t)e.d.ps * e.d.CT * e.d.OT;
    total += (double)e.d.CT * e.d.OT * (e.U / 4.0) * unit_cycles_of(e);
  }
  if (!rows5 || !others || wgs > (int64_t)num_cus() * 5 / 4) return;
  for (auto& e : q) {
    const double share = (double)e.d.CT * e.d.OT * (e.U / 4.0) * unit_cycles_of(e) / total;
    int ps = (int)(share * num_cus() / (e.d.CT * e.d.OT) + 0.5);
    ps = std::max(1, std::min(ps, e.U / 4));
    int upb = (int)df_ceil_div(e.U, ps);
    upb = std::max(4, (upb + 3) & ~3);
    e.d.units = upb;
    e.d.ps = (int)df_ceil_div(e.U, upb);
  }
  // (rounding may have gone past one round: the workgroups left over would run a second one by themselves -- take slabs from
  // the layers whose workgroups are shortest until everything is resident at once)
  for (;;) {
    int64_t n = 0;
    for (const auto& e : q) n += (int64_t)e.d.ps * e.d.CT * e.d.OT;
    if (n <= num_cus()) break;
    GQueued* pick = nullptr;
    for (auto& e : q)
      if (e.d.ps > 1 && (!pick || replan_direct(e) < replan_direct(*pick))) pick = &e;
    if (!pick) break;
    int upb = (int)df_ceil_div(pick->U, pick->d.ps - 1);
    upb = std::max(4, (upb + 3) & ~3);
    pick->d.units = upb;
    pick->d.ps = (int)df_ceil_div(pick->U, upb);
  }
}

int group_flush() {
  if (g_queue.empty()) return DFHIP_OK;
  std::vector<GQueued> q;
  q.swap(g_queue);                                  // (re-entrancy: scratch() may ask for a flush)
  // A grouped layer is cut into 8x fewer, longer workgroups than a stand-alone launch (the other layers of the
  // sweep fill the chip).  A sweep that queued too little for that (the CNN workload: one eligible layer) would
  // leave most CUs idle: its layers are re-planned and launched stand-alone instead.  (A quarter of the chip: the
  // two half sweeps of the overlapped data-parallel step -- layer4 + classifier, then the rest, ~200 workgroups
  // each -- stay grouped, so their sums are those of the single sweep bit for bit.)
  if (!g_group_stream) {
    int64_t wgs_total = 0;
    for (auto& e : q) wgs_total += (int64_t)e.d.ps * e.d.CT * e.d.OT;
    if (wgs_total < num_cus() / 4) {
      const bool was = g_group_on;
      g_group_on = false;
      int st = DFHIP_OK;
      for (auto& e : q) {
        if (e.d.kind == 13 && e.d.tap_row0 > 0) continue;      // (a 5x5 kernel's five entries: the general path computes all rows)
        const float* edy = e.d.dy;
        float* tmp = nullptr;
        if (e.d.bnz) {                              // the pending BatchNorm backward written out once
          st = dfhip_malloc((int64_t)e.d.N * e.d.O * e.d.oh * e.ow, &tmp);
          if (st != DFHIP_OK) break;
          st = bn_bwd_materialise(e.d.dy, e.d.bnz, reinterpret_cast<const float*>(e.d.bntab), tmp, e.d.N, e.d.O, e.d.oh * e.ow);
          if (st != DFHIP_OK) { (void)dfhip_free(tmp); break; }
          edy = tmp;
        }
        if (e.d.aff)
          st = wgrad_affine_materialised(e.d.x, e.d.aff, edy, e.dw, e.d.N, e.d.C, e.d.H, e.W, e.d.O, e.d.oh, e.ow,
                                         e.k, e.stride, e.pad);
        else
          st = run_wgrad_rows(e.d.x, edy, e.dw, e.d.N, e.d.C, e.d.H, e.W, e.d.O, e.d.oh, e.ow, e.k, e.stride, e.pad);
        if (st < 0)                                 // (a shape the stand-alone row kernel does not take: the general path)
          st = dfhip_conv2d_wgrad(e.d.x, edy, e.dw, e.d.N, e.d.C, e.d.H, e.W, e.d.O, e.k, e.stride, e.pad);
        if (tmp) (void)dfhip_free(tmp);             // stream-ordered pool: later work on the stream may reuse it
        if (st != DFHIP_OK) break;
      }
      g_group_on = was;
      return st < 0 ? DFHIP_RUNTIME_ERROR : st;
    }
  }
  if (!g_group_stream) recut_one_round(q);
  for (auto& e : q) {
    e.work = e.d.direct ? replan_direct(e) : (double)e.d.N * e.d.oh * e.d.oh * e.T / e.d.ps * 16.0;
    e.rides_on = -1;
    e.d.dy2 = nullptr; e.d.slabs2 = nullptr;
    e.d.bnz2 = nullptr; e.d.bntab2 = nullptr;
  }
  // longest workgroups first (they are dispatched in grid order): the launch ends on short ones
  std::stable_sort(q.begin(), q.end(), [](const GQueued& a, const GQueued& b) { return a.work > b.work; });
  // a block's 1x1 / stride-2 shortcut reads the same tensor as its 3x3 / stride-2 convolution: the shortcut's weight gradient
  // rides on that layer's workgroups (wgrad_rows_direct<PAIR>) -- it keeps its slabs (one per workgroup of the host layer) and
  // its entry in the slab sum, and has no workgroups of its own
  if (q.size() <= (size_t)kMaxGroup) {
    for (size_t b = 0; b < q.size(); b++) {
      GQueued& sc = q[b];
      if (!(sc.d.direct && sc.k == 1 && sc.stride == 2 && !sc.d.aff)) continue;
      for (size_t a = 0; a < q.size(); a++) {
        GQueued& cv = q[a];
        if (cv.d.direct && cv.k == 3 && cv.stride == 2 && cv.pad == 1 && !cv.d.aff && !cv.d.dy2 && cv.d.x == sc.d.x &&
            cv.d.N == sc.d.N && cv.d.C == sc.d.C && cv.d.H == sc.d.H && cv.W == sc.W && cv.d.O == sc.d.O &&
            cv.d.oh == sc.d.oh && cv.ow == sc.ow) {
          sc.rides_on = (int)a;
          sc.d.ps = cv.d.ps;
          cv.d.dy2 = sc.d.dy;
          cv.d.bnz2 = sc.d.bnz; cv.d.bntab2 = sc.d.bntab;
          break;
        }
      }
    }
  }
  size_t total = 0, lds = 0;
  for (auto& e : q) {
    total += (size_t)e.d.ps * e.d.OT * e.d.CT * e.T * 1024 * sizeof(float);
    lds = std::max(lds, e.lds);
  }
  hipStream_t gst = g_group_stream ? g_group_stream : stream();
  float* base = nullptr;
  if (g_group_stream) {
    void* blk = nullptr;
    if (dfhip_malloc((int64_t)((total + 3) / 4), (float**)&blk) != DFHIP_OK) return DFHIP_OUT_OF_MEMORY;
    base = g_group_slabs = (float*)blk;
  } else {
    base = workspace_slot(0, total);
  }
  if (!base) return DFHIP_OUT_OF_MEMORY;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_wgrad_group, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    DF_HIP(hipFuncSetAttribute((const void*)k_wgrad_group_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  for (size_t i0 = 0; i0 < q.size(); i0 += kMaxGroup) {
    const int n = (int)std::min<size_t>(kMaxGroup, q.size() - i0);
    WGroup wg = {};
    SGroup sg = {};
    sg.n = n;
    int wgs = 0, vbs = 0, nl = 0;
    double flops = 0, bytes = 0, sbytes = 0;
    for (int i = 0; i < n; i++) {                             // slab areas first: a host layer needs its rider's
      GQueued& e = q[i0 + i];
      e.d.slabs = base;
      base += (size_t)e.d.ps * e.d.OT * e.d.CT * e.T * 1024;
    }
    for (int i = 0; i < n; i++) {
      GQueued& e = q[i0 + i];
      if (e.rides_on >= 0) q[e.rides_on].d.slabs2 = e.d.slabs;
    }
    for (int i = 0; i < n; i++) {
      GQueued& e = q[i0 + i];
      if (e.rides_on < 0) {
        wgs += e.d.ps * e.d.CT * e.d.OT;
        e.d.wg_end = wgs;
        wg.d[nl++] = e.d;
      }
      sg.s[i] = make_slab_sum(e.d.slabs, e.dw, e.d.ps, e.d.OT, e.d.CT, e.T, e.d.O, e.d.C);
      if (e.tstride) { sg.s[i].tstride = e.tstride; sg.s[i].toff = e.toff; }
      // tap-major items: (tile, q, lane) positions, 256 / SL of them per block
      sg.s[i].SL = 1;
      while (sg.s[i].SL < 8 && sg.s[i].SL * 2 <= e.d.ps) sg.s[i].SL *= 2;
      sg.s[i].nvb = (int)df_ceil_div((int64_t)e.d.OT * e.d.CT * 256, 256 / sg.s[i].SL);
      vbs += sg.s[i].nvb;
      sg.vb_end[i] = vbs;
      flops += e.flops; bytes += e.bytes;
      sbytes += (double)(e.d.ps + 1) * e.d.OT * e.d.CT * e.T * 4096.0;
    }
    wg.n = nl;
    static int stamp_mode = -1;
    if (stamp_mode < 0) stamp_mode = getenv("DFHIP_CONV_STAMP") ? atoi(getenv("DFHIP_CONV_STAMP")) : 0;
    hipEvent_t se0 = nullptr, se1 = nullptr;
    if (stamp_mode) {
      wg.ticks = (unsigned long long*)workspace2((size_t)wgs * 8 * 5);
      (void)hipMemsetAsync(wg.ticks, 0, (size_t)wgs * 8 * 5, gst);
#ifdef DFHIP_STAMPS
      for (int i = 0; i < nl; i++) wg.d[i].dbg = wg.ticks + wgs;
#endif
      (void)hipEventCreate(&se0); (void)hipEventCreate(&se1);
      (void)hipEventRecord(se0, gst);
    }
    {
      ProfScope prof(g_matmul_dtype == 1 ? "conv2d_wgrad_grouped_bf16" : "conv2d_wgrad_grouped", flops, bytes);
      if (g_matmul_dtype == 1) DF_LAUNCH(k_wgrad_group_bf16, dim3((unsigned)wgs), dim3(64 * kRowsWaves), lds, gst, wg);
      else DF_LAUNCH(k_wgrad_group, dim3((unsigned)wgs), dim3(64 * kRowsWaves), lds, gst, wg);
      DF_LAUNCH_CHECK();
    }
    if (stamp_mode) {      // cycles of a workgroup against the launch's wall time: the clock under this kernel
      (void)hipEventRecord(se1, gst);
      (void)hipEventSynchronize(se1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, se0, se1);
      std::vector<unsigned long long> h(wgs), hd(2 * (size_t)wgs);
      (void)hipMemcpy(h.data(), wg.ticks, (size_t)wgs * 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(hd.data(), wg.ticks + wgs, (size_t)wgs * 16, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; double sum = 0, si = 0, sm = 0;
      for (auto v : h) { mx = std::max(mx, v); sum += (double)v; }
      {
        std::vector<unsigned long long> rt(2 * (size_t)wgs);
        (void)hipMemcpy(rt.data(), wg.ticks + 3 * (size_t)wgs, (size_t)wgs * 16, hipMemcpyDeviceToHost);
        unsigned long long r0 = ~0ull, r1 = 0;
        double res = 0, ghz_lo = 1e9, ghz_hi = 0;
        for (int b = 0; b < wgs; b++) {
          r0 = std::min(r0, rt[b]); r1 = std::max(r1, rt[wgs + b]);
          const double dt = (double)(rt[wgs + b] - rt[b]) * 10.0;          // ns
          res += dt;
          if (dt > 0) { const double g = (double)h[b] / dt; ghz_lo = std::min(ghz_lo, g); ghz_hi = std::max(ghz_hi, g); }
        }
        const double span = (double)(r1 - r0) * 10.0;
        fprintf(stderr, "[stampg]   first start .. last end %.1f us; workgroups resident %.1f CU-us = %.3f of %d CUs x span; clock of a "
                "workgroup %.2f-%.2f GHz (mean %.2f)\n", span * 1e-3, res * 1e-3, res / (span * num_cus()), num_cus(), ghz_lo, ghz_hi,
                sum / res);
        // how many workgroups are resident over time (20 slices of the span)
        fprintf(stderr, "[stampg]   resident workgroups per twentieth of the span:");
        for (int sl = 0; sl < 20; sl++) {
          const double t0 = (double)r0 + (double)(r1 - r0) * sl / 20.0, t1 = (double)r0 + (double)(r1 - r0) * (sl + 1) / 20.0;
          double occ = 0;
          for (int b = 0; b < wgs; b++) {
            const double a0 = std::max(t0, (double)rt[b]), a1 = std::min(t1, (double)rt[wgs + b]);
            if (a1 > a0) occ += (a1 - a0) / (t1 - t0);
          }
          fprintf(stderr, " %.0f", occ);
        }
        fprintf(stderr, "\n");
      }
      for (int i = 0; i < wgs; i++) { si += (double)hd[2 * i]; sm += (double)hd[2 * i + 1]; }
      fprintf(stderr, "[stampg]   wave 0 of a workgroup: load issue %.0f, MFMA sections %.0f cycles (mean)\n", si / wgs, sm / wgs);
      for (int i = 0, b0 = 0; i < nl; i++) {                  // per layer: where the workgroups' cycles go
        const GDesc& dd = wg.d[i];
        double tc = 0, tm = 0;
        unsigned long long lmx = 0;
        double tp = 0, tspan = 0;
        for (int b = b0; b < dd.wg_end; b++) {
          tc += (double)h[b]; tm += (double)hd[2 * b + 1]; lmx = std::max(lmx, h[b]);
          tp += (double)(hd[2 * b] >> 32); tspan += (double)(hd[2 * b] & 0xffffffffull);
        }
        const int cnt = dd.wg_end - b0;
        fprintf(stderr, "[stampg]   layer %2d kind %d direct %d C %3d O %3d H %2d: %4d workgroups (ps %d), cycles mean %.0f max %llu, "
                "MFMA sections %.0f, prologue %.0f, first..last section %.0f, after %.0f; share of all cycles %.3f\n", i, dd.kind, dd.direct,
                dd.C, dd.O, dd.H, cnt, dd.ps, tc / cnt, lmx, tm / cnt, tp / cnt, tspan / cnt, (tc - tp - tspan) / cnt, tc / sum);
        b0 = dd.wg_end;
      }
      fprintf(stderr, "[stampg] %d workgroups, launch %.1f us; cycles per workgroup: mean %.0f max %llu -> >= %.2f GHz if the longest one spans the launch\n",
              wgs, ms * 1e3, sum / wgs, mx, (double)mx / (ms * 1e3) * 1e-3);
      (void)hipEventDestroy(se0); (void)hipEventDestroy(se1);
    }
    {
      // the stem's one-pass backward left its last step for this launch (on the library stream only: its totals were written there)
      if (g_stem_pending.active && !g_group_stream && i0 + kMaxGroup >= q.size()) {
        sg.stem = g_stem_pending.c;
        sg.stem_blocks = g_stem_pending.blocks;
        g_stem_pending.active = false;
        sbytes += 4.0 * (double)sg.stem.splits * (2.0 * sg.stem.O * sg.stem.KK + 3.0 * sg.stem.O + sg.stem.KK);
      }
      if (g_head_pending.active && !g_group_stream && i0 + kMaxGroup >= q.size()) {
        sg.head = g_head_pending.h;
        sg.head_blocks = head_grads_blocks(sg.head);
        g_head_pending.active = false;
        sbytes += 4.0 * ((double)sg.head.N * sg.head.C + (double)sg.head.N * sg.head.Pn + (double)sg.head.C * sg.head.Pn);
      }
      if (g_bias_pending.n > 0 && !g_group_stream && i0 + kMaxGroup >= q.size()) {
        sg.bias = g_bias_pending;
        sg.bias_blocks = sg.bias.blk_end[sg.bias.n - 1];
        g_bias_pending.n = 0;
        for (int bi = 0; bi < sg.bias.n; bi++)
          sbytes += 4.0 * ((double)sg.bias.N[bi] * sg.bias.C[bi] * (4 << sg.bias.lq[bi]) + sg.bias.C[bi]);
      }
      ProfScope prof("wgrad_slab_sum", 0.0, sbytes);
      DF_LAUNCH(k_wgrad_group_sum, dim3((unsigned)(vbs + sg.stem_blocks + sg.head_blocks + sg.bias_blocks)), dim3(256), 0, gst, sg);
      DF_LAUNCH_CHECK();
    }
  }
  return DFHIP_OK;
}

int side_flush() {
  if (g_queue.empty()) return DFHIP_OK;
  if (!g_side) {
    if (overlap_cus()) {
      int st = make_masked_stream(&g_side, overlap_cus(), 256 - overlap_cus());
      if (st != DFHIP_OK) return st;
    } else {
      DF_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
    }
    DF_HIP(hipEventCreateWithFlags(&g_side_fork, hipEventDisableTiming));
  }
  hipEvent_t done = nullptr;
  if (!g_side_events.empty()) { done = g_side_events.back(); g_side_events.pop_back(); }
  else DF_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  std::vector<std::pair<const float*, const float*>> dws;
  for (const auto& e : g_queue) dws.push_back({e.dw, e.dw + (int64_t)e.d.O * e.d.C * e.T});
  DF_HIP(hipEventRecord(g_side_fork, stream()));
  DF_HIP(hipStreamWaitEvent(g_side, g_side_fork, 0));
  g_group_stream = g_side;
  g_group_slabs = nullptr;
  const int st = group_flush();
  g_group_stream = nullptr;
  if (st != DFHIP_OK) return st;
  DF_HIP(hipEventRecord(done, g_side));
  g_side_pending.push_back({done, g_group_slabs, dws});
  g_group_slabs = nullptr;
  return DFHIP_OK;
}

int side_join() {
  for (auto& p : g_side_pending) {
    DF_HIP(hipStreamWaitEvent(stream(), p.done, 0));
    if (p.slabs) (void)dfhip_free(p.slabs);          // stream-ordered pool: later users run behind the join
    g_side_events.push_back(p.done);
  }
  g_side_pending.clear();
  return DFHIP_OK;
}

// returns -1 when the shape is not one the row-unit kernel is built for
// 5x5 / stride 1 / pad 2 on 16 x 16 planes inside a group (the CNN workload's conv2, test/CNN_CIFAR10_cuda.py:66): five queue
// entries, one tap row each (wgrad_rows_direct<.., KR = 1>), cut so that the five together are about one workgroup per CU
static int queue_wgrad_k5(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int O, int oh, int ow) {
  const int CT = C / 32, OT = O / 32;
  const int upr = oh, U = (N / 2) * upr;                       // UR = 1, NP = 1
  const int want = std::max(1, (num_cus() * 7 / 8) / (5 * CT * OT));
  int upb = (int)df_ceil_div(U, want);
  upb = std::max(4, (upb + 3) & ~3);
  for (int i0 = 0; i0 < 5; i0++) {
    GQueued e = {};
    GDesc& d = e.d;
    d.x = x; d.dy = dy; d.slabs = nullptr;
    d.N = N; d.C = C; d.H = H; d.O = O; d.oh = oh;
    d.kind = 13; d.direct = 1; d.CT = CT; d.OT = OT; d.tap_row0 = i0;
    d.units = upb;
    d.ps = (int)df_ceil_div(U, upb);
    e.U = U;
    d.m_br = magic_u32((unsigned)upr);
    e.dw = dw; e.T = 5; e.tstride = 25; e.toff = 5 * i0;
    e.lds = (size_t)3 * kRowsWaves * 4 * 64 * 16;
    e.flops = 2.0 * N * oh * ow * (double)C * 5 * O;
    e.bytes = 4.0 * ((double)N * C * H * W + (double)O * C * 25 + (double)N * O * oh * ow) / 5.0;
    e.W = W; e.ow = ow; e.k = 5; e.stride = 1; e.pad = 2;
    g_queue.push_back(e);
  }
  return DFHIP_OK;
}

int run_wgrad_rows(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int O, int oh, int ow,
                   int k, int stride, int pad, const float* aff, const float* bnz, const float* bntab) {
  if (g_group_on && !g_group_stream && k == 5 && stride == 1 && pad == 2 && H == 16 && W == 16 && oh == 16 && ow == 16 &&
      C % 32 == 0 && O % 32 == 0 && N % 2 == 0 && !aff && !bnz && g_queue.size() + 5 <= (size_t)kMaxGroup)
    return queue_wgrad_k5(x, dy, dw, N, C, H, W, O, oh, ow);
  if (aff && !g_group_on) return -1;        // the affine operand exists in the grouped direct body only
  if (bnz && (!g_group_on || W > 16)) return -1;   // so does the BatchNorm-backward operand (planes up to 16 wide)
  if (C < 16 || (N & 1)) return -1;
  const bool k3 = k == 3 && pad == 1, k1 = k == 1 && pad == 0 && stride == 2;
  if (!k3 && !k1) return -1;
  if (!(W == 32 || W == 16 || W == 8 || W == 4 || (W == 2 && stride == 1))) return -1;
  if (W == 32 && !g_group_on) return -1;     // (32-wide planes: the grouped body only -- stand-alone it spills 50-70 registers)
  if (ow != (W + 2 * pad - k) / stride + 1 || (stride == 2 && ow * 2 != W) || stride > 2) return -1;
  if ((H * W) % 4 != 0 || (oh * ow) % 4 != 0) return -1;
  const bool grouped = g_group_on;
  const int CT = (int)df_ceil_div(C, 32), OT = (int)df_ceil_div(O, 32);
  // pixel slabs per (channel tile, output tile): a stand-alone launch needs one workgroup per CU;
  // a grouped launch has the other layers to fill the chip with and takes longer workgroups
  // (8: re-measured with the round-3 body -- 4 / 6 / 8 / 10 / 12 / 16 / 24: 131 / 138 / 127 / 140 / 135 / 138 / 240 us for the R18 sweep)
  static int gdiv = getenv("DFHIP_WGRAD_GROUP_DIV") ? atoi(getenv("DFHIP_WGRAD_GROUP_DIV")) : 8;
  const int ps0 = std::max(1, num_cus() / (CT * OT) / (grouped ? std::max(1, gdiv) : 1));
  constexpr int gpb1 = 2;   // groups per block aimed at
  constexpr int gpbg = 2;
  const int gpb = grouped ? gpbg : gpb1;
  // output rows per unit (must match the instantiations)
  const int UR = k == 1 ? 1 : (stride == 1 ? (W >= 16 ? 1 : (W == 8 ? 2 : (W == 4 ? 4 : 2))) : (W >= 16 ? 1 : 2));
  const double per_block = (double)(N / 2) * (oh / UR) / ps0;
  const double target = std::max((double)kRowsWaves, per_block / std::max(1, gpb));
  constexpr int gcap = 160;
  const size_t lds_cap = grouped ? (size_t)gcap * 1024 : 160 * 1024;
  RGeo g = {};
  double best = 1e30;
  size_t best_lds = 0;
  for (int BR = oh; BR >= 1; BR--) {
    if (oh % BR || BR % UR) continue;
    const int gr = oh / BR;
    const int RLr = gr == 1 ? H : std::min(H, (BR - 1) * stride + k);
    const int RL = RLr * W, BRBC = BR * ow;
    if (RL % 4 || BRBC % 4) continue;
    if (gr > 1 && (W % 4)) continue;
    for (int BI = 2; BI <= N && BI <= 128; BI *= 2) {
      if (N % BI) break;
      const int units = BI / 2 * (BR / UR);
      if (units % kRowsWaves) continue;
      const int xv = (int)df_ceil_div(32 * BI * (RL / 4), 64 * kRowsWaves), dv = (int)df_ceil_div(32 * BI * (BRBC / 4), 64 * kRowsWaves);
      if ((xv > 10 || dv > (grouped ? 8 : 4)) && (grouped || (N / BI) * gr > ps0)) continue;  // later groups must fit the in-flight vectors
      const int XST = bank_stride_r(BI * RL, W >= 4 ? 4 : 2), DST = bank_stride_r(BI * BRBC, ow >= 4 ? 4 : 2);
      const size_t stage = (size_t)32 * (XST + DST) * 4;
      const size_t red = grouped ? (size_t)(k == 3 ? 3 : 1) * kRowsWaves * 4 * 64 * 16
                                 : (size_t)k * k * kRowsWaves * 4 * 64 * 16;
      int single = 0;
      size_t lds = std::max(2 * stage, red);
      if (grouped && lds > lds_cap) {
        single = 1;
        lds = std::max(stage, red);
      }
      if (lds > lds_cap) break;
      double score = std::fabs(std::log((double)units / target)) - 1e-3 * BR + (single ? 0.05 : 0.0);
      if (score < best) {
        best = score; best_lds = lds;
        g.BI = BI; g.BR = BR; g.gr = gr; g.RLr = RLr; g.XST = XST; g.DST = DST; g.units = units;
        g.single = single;
      }
    }
  }
  // (no staging plan -- 32-wide planes exceed the vectors a thread keeps in flight -- is fine for a layer the grouped launch
  // takes with its operands straight from global memory)
  const bool staged_plan = best < 1e29;
  if (!staged_plan && !(grouped && C % 32 == 0 && O % 32 == 0)) return -1;
  if (!staged_plan) { g.BI = 2; g.BR = UR; g.gr = oh / UR; g.RLr = std::min(H, (UR - 1) * stride + k); g.units = 4; }
  g.x = x; g.dy = dy;
  g.N = N; g.C = C; g.H = H; g.O = O; g.oh = oh;
  g.ngroups = (N / g.BI) * g.gr;
  const int RL4 = g.RLr * W / 4, B4 = g.BR * ow / 4;
  g.m_br = magic_u32((unsigned)(g.BR / UR));
  g.m_xc = magic_u32((unsigned)(g.BI * RL4));
  g.m_xi = magic_u32((unsigned)RL4);
  g.m_dc = magic_u32((unsigned)(g.BI * B4));
  g.m_di = magic_u32((unsigned)B4);
  const int ps = std::max(1, std::min(g.ngroups, ps0));
  if (grouped) {
    int kind = -1, NPk = 1;
#define DF_KIND(ID, KSV, PADV, SV, WV, OWV, URV, NPV) \
  if (k == KSV && stride == SV && W == WV && UR == URV) { kind = ID; NPk = NPV; }
    DF_ROWS_KINDS(DF_KIND)
#undef DF_KIND
    if (kind < 0) return -1;
    GQueued e = {};
    GDesc& d = e.d;
    d.x = x; d.dy = dy; d.slabs = nullptr;
    d.N = N; d.C = C; d.H = H; d.O = O; d.oh = oh;
    d.BI = g.BI; d.BR = g.BR; d.gr = g.gr; d.ngroups = g.ngroups; d.RLr = g.RLr;
    d.XST = g.XST; d.DST = g.DST; d.units = g.units;
    d.m_br = g.m_br; d.m_xc = g.m_xc; d.m_xi = g.m_xi; d.m_dc = g.m_dc; d.m_di = g.m_di;
    d.single = g.single; d.kind = kind; d.ps = ps; d.CT = CT; d.OT = OT;
    e.dw = dw; e.T = k * k; e.lds = best_lds;
    const bool direct = C % 32 == 0 && O % 32 == 0 && (N / 2) % NPk == 0;
    if (!direct && !staged_plan) return -1;
    if ((aff || bnz) && !direct) return -1;
    d.aff = aff;
    d.bnz = bnz; d.bntab = reinterpret_cast<const float4*>(bntab);
    if (direct) {
      // operands straight from global memory: a block takes a contiguous run of units (a multiple of 4)
      const int upr = oh / UR, U = (N / (2 * NPk)) * upr;
      int upb = (int)df_ceil_div(U, ps0);
      upb = std::max(4, (upb + 3) & ~3);
      d.direct = 1;
      d.units = upb;
      d.ps = (int)df_ceil_div(U, upb);
      e.U = U;
      d.m_br = magic_u32((unsigned)upr);
      e.lds = (size_t)(k == 3 ? 3 : 1) * kRowsWaves * 4 * 64 * 16;
    }
    e.flops = 2.0 * N * oh * ow * (double)C * k * k * O;
    e.bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + (double)N * O * oh * ow);
    e.W = W; e.ow = ow; e.k = k; e.stride = stride; e.pad = pad;
    g_queue.push_back(e);
    static int side_every = getenv("DFHIP_WGRAD_SIDE") ? atoi(getenv("DFHIP_WGRAD_SIDE")) : 0;
    if (side_every > 0 && (int)g_queue.size() >= side_every && !profiling()) return side_flush();
    return DFHIP_OK;
  }
#define DF_ROWS(ID, KSV, PADV, SV, WV, OWV, URV, NPV) \
  if (k == KSV && stride == SV && W == WV && UR == URV)  \
    return launch_wgrad_rows<KSV, PADV, SV, WV, OWV, URV>(g, dw, best_lds, ps, CT, OT);
  DF_ROWS_KINDS(DF_ROWS)
#undef DF_ROWS
  return -1;
}

// y[n][c][:] = x[n][c][:] * aff[c][0] + aff[c][1] (16-byte pieces; HW % 4 == 0)
__global__ __launch_bounds__(256) void k_affine_nchw(const float* __restrict__ x, const float* __restrict__ aff,
                                                    float* __restrict__ y, int C, int hw4, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i / hw4) % C);
  const float sc = aff[2 * c], sh = aff[2 * c + 1];
  float4 v = reinterpret_cast<const float4*>(x)[i];
  v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
  reinterpret_cast<float4*>(y)[i] = v;
}

// dz[n][o][:] = A dy + B (z - mean) + Cc with (A, B, mean, Cc) = tab[o]: a pending BatchNorm backward (dfhip_bnbwd.h) written
// out, for the consumers that have no operand path for it (16-byte pieces; HW % 4 == 0)
__global__ __launch_bounds__(256) void k_bn_bwd_apply_table(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float4* __restrict__ tab, float* __restrict__ dz, int O,
                                                           int hw4, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 t = tab[(int)((i / hw4) % O)];
  const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(z)[i];
  float4 r;
  r.x = bn_bwd_point(g.x, v.x, t); r.y = bn_bwd_point(g.y, v.y, t);
  r.z = bn_bwd_point(g.z, v.z, t); r.w = bn_bwd_point(g.w, v.w, t);
  reinterpret_cast<float4*>(dz)[i] = r;
}

int bn_bwd_materialise(const float* dy, const float* z, const float* tab, float* dz, int N, int O, int HW) {
  if (HW % 4 != 0) {
    set_error("bn2d_backward_apply_table: H * W must be a multiple of 4");
    return DFHIP_INVALID_ARGUMENT;
  }
  const int64_t n = (int64_t)N * O * HW;
  ProfScope prof("bn2d_bwd", 0.0, 12.0 * (double)n);
  DF_LAUNCH(k_bn_bwd_apply_table, dim3((unsigned)df_ceil_div(n / 4, 256)), dim3(256), 0, stream(), dy, z,
            reinterpret_cast<const float4*>(tab), dz, O, HW / 4, n / 4);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// the table (and dgamma / dbeta) of a pending BatchNorm backward from the rows its producer left behind, as a launch of
// its own: for a convolution that needs no input gradient (the input-gradient kernels build the table themselves)
__global__ __launch_bounds__(256) void k_bn_bwd_table(const BnBwdIn b, int O) {
  __shared__ float4 tab[256];
  __shared__ __attribute__((aligned(16))) float part[kBnPartFloats];
  BnTableBuilder<256> tb;
  tb.issue(b, O);
  tb.finish(b, O, tab, part, true);
}

// fall-back of dfhip_conv2d_wgrad_affine: the operand is written out once and the weight gradient runs NOW on the
// ordinary (ungrouped) paths -- a queued call would read the temporary after it has gone back to the pool
int wgrad_affine_materialised(const float* x, const float* aff, const float* dy, float* dw, int N, int C, int H, int W,
                              int O, int oh, int ow, int k, int stride, int pad) {
  (void)oh; (void)ow;
  if ((H * W) % 4 != 0) {
    set_error("conv2d_wgrad_affine: H * W must be a multiple of 4");
    return DFHIP_INVALID_ARGUMENT;
  }
  const int64_t n = (int64_t)N * C * H * W;
  float* tmp = nullptr;
  int st = dfhip_malloc(n, &tmp);
  if (st != DFHIP_OK) return st;
  DF_LAUNCH(k_affine_nchw, dim3((unsigned)df_ceil_div(n / 4, 256)), dim3(256), 0, stream(), x, aff, tmp, C, H * W / 4,
            n / 4);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    (void)dfhip_free(tmp);
    set_error("kernel launch failed: %s", hipGetErrorString(e));
    return DFHIP_RUNTIME_ERROR;
  }
  const bool was = g_group_on;
  g_group_on = false;
  st = dfhip_conv2d_wgrad(tmp, dy, dw, N, C, H, W, O, k, stride, pad);
  g_group_on = was;
  (void)dfhip_free(tmp);                 // stream-ordered pool: later work on the stream may reuse it
  return st;
}

// ---- tap-major weight gradient: plan + launch ----
template <int KS, int XV, int DV, bool DENSE>
int launch_wgrad_tap(TGeo g, float* dw, size_t lds, int ps, int CT, int OT) {
  const int T = KS * KS;
  g.slabs = workspace((size_t)ps * OT * CT * T * 1024 * sizeof(float));
  if (!g.slabs) return DFHIP_OUT_OF_MEMORY;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_wgrad_tap<KS, XV, DV, DENSE>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  dim3 grid((unsigned)ps, (unsigned)CT, (unsigned)OT);
  static int stamp_mode = -1;
  if (stamp_mode < 0) stamp_mode = getenv("DFHIP_CONV_STAMP") ? atoi(getenv("DFHIP_CONV_STAMP")) : 0;
  const size_t nb = (size_t)ps * CT * OT;
  g.stamps = nullptr;
  if (stamp_mode) {
    g.stamps = (unsigned long long*)workspace2(nb * 32 * 8);
    (void)hipMemsetAsync(g.stamps, 0, nb * 32 * 8, stream());
  }
  DF_LAUNCH((k_conv_wgrad_tap<KS, XV, DV, DENSE>), grid, dim3(512), lds, stream(), g);
  DF_LAUNCH_CHECK();
  {
    int st_ = launch_slab_sum(make_slab_sum(g.slabs, dw, ps, OT, CT, T, g.O, g.C));
    if (st_ != DFHIP_OK) return st_;
  }
  if (stamp_mode) {
    std::vector<unsigned long long> h(nb * 32);
    (void)hipStreamSynchronize(stream());
    (void)hipMemcpy(h.data(), g.stamps, nb * 32 * 8, hipMemcpyDeviceToHost);
    double seg[32] = {0};
    int nseg = 0;
    for (size_t b = 0; b < nb; b++) {
      const unsigned long long* r = &h[b * 32];
      int n = 0;
      while (n < 16 && r[n]) n++;
      for (int i = 1; i < n; i++) seg[i] += (double)(r[i] - r[i - 1]);
      nseg = std::max(nseg, n);
    }
    fprintf(stderr, "[stampt] k %d XV %d DV %d grid %d x %d x %d groups %d (BI %d BR %d) steps/group %d lds %zu; mean segments:",
            KS, XV, DV, ps, CT, OT, g.ngroups, g.BI, g.BR, g.ksg, lds);
    for (int i = 1; i < nseg; i++) fprintf(stderr, " %.0f", seg[i] / nb);
    fprintf(stderr, "\n[stampt]   group 0 per wave (start, end) since block start:");
    for (int w = 0; w < 8; w++) {
      double a = 0, e = 0;
      for (size_t b = 0; b < nb; b++) {
        a += (double)(h[b * 32 + 24 + w] - h[b * 32]);
        e += (double)(h[b * 32 + 16 + w] - h[b * 32]);
      }
      fprintf(stderr, " (%.0f, %.0f)", a / nb, e / nb);
    }
    fprintf(stderr, "\n");
  }
  return DFHIP_OK;
}

// LDS channel stride: the smallest value >= n that is 2 (mod 64)
static int bank_stride(int n) { return ((n - 2 + 63) / 64) * 64 + 2; }

// returns -1 when the shape is not eligible for the tap-major kernel
int run_wgrad_tap(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int O, int oh, int ow,
                  int k, int stride, int pad) {
  if (!(k == 1 || k == 3) || C < 16 || (N & 1)) return -1;
  if ((H * W) % 4 != 0 || (oh * ow) % 4 != 0) return -1;
  if (pad >= k) return -1;
  const int CT = (int)df_ceil_div(C, 32), OT = (int)df_ceil_div(O, 32);
  const int64_t steps_total = (int64_t)(N / 2) * oh * ow;
  const int ps0 = std::max(1, num_cus() / (CT * OT));
  constexpr int gpb = 2;   // groups per block aimed at
  const double target = (double)std::max<int64_t>(16, steps_total / ps0 / std::max(1, gpb));
  TGeo g = {};
  double best = 1e30;
  size_t best_lds = 0;
  for (int BR = oh; BR >= 1; BR--) {
    if (oh % BR) continue;
    const int gr = oh / BR;
    const int RLr = gr == 1 ? H : std::min(H, (BR - 1) * stride + k);
    const int RL = RLr * W, BRBC = BR * ow;
    if (RL % 4 || BRBC % 4) continue;
    if (gr > 1 && (W % 4)) continue;
    for (int BI = 2; BI <= N && BI <= 128; BI *= 2) {
      if (N % BI) break;
      const int ksg = BI / 2 * BRBC;
      if (ksg % 16) continue;                      // an even number of steps for each of the 8 waves
      const int xitems = 32 * (BI / 2) * (RL / 4), ditems = 32 * (BI / 2) * (BRBC / 4);
      if (xitems > 512 * 3 || ditems > 512 * 2) break;
      const int XST = bank_stride(BI * RL), DST = bank_stride(BI * BRBC);
      const size_t red = (k == 3 ? 4 : 1) * (size_t)8 * 4 * 64 * 16;
      const size_t lds = std::max((size_t)2 * 32 * (XST + DST) * 4, red);
      if (lds > 160 * 1024) break;
      double score = std::fabs(std::log((double)ksg / target)) - 1e-3 * BR;
      if (score < best) {
        best = score; best_lds = lds;
        g.BI = BI; g.BR = BR; g.gr = gr; g.RLr = RLr; g.XST = XST; g.DST = DST; g.ksg = ksg;
      }
    }
  }
  if (best > 1e29) return -1;
  g.x = x; g.dy = dy;
  g.N = N; g.C = C; g.H = H; g.W = W; g.O = O; g.oh = oh; g.ow = ow; g.k = k; g.stride = stride; g.pad = pad;
  g.ngroups = (N / g.BI) * g.gr;
  const int P = g.BI / 2, RL4 = g.RLr * W / 4, BRBC4 = g.BR * ow / 4;
  g.m_brow = magic_u32((unsigned)(g.BR * ow));
  g.m_ow = magic_u32((unsigned)ow);
  g.m_xc = magic_u32((unsigned)(P * RL4));
  g.m_xp = magic_u32((unsigned)RL4);
  g.m_dc = magic_u32((unsigned)(P * BRBC4));
  g.m_dp = magic_u32((unsigned)BRBC4);
  const int ps = std::max(1, std::min(g.ngroups, ps0));
  const int xv = (int)df_ceil_div(32 * P * RL4, 512), dv = (int)df_ceil_div(32 * P * BRBC4, 512);
  // images of 8 x 8 and more lose little to the padding taps; smaller ones skip them
  constexpr int dense_min = 8;
  const bool dense = k == 3 && std::min(oh, ow) >= dense_min;
  if (k == 3 && dense) {
    if (xv <= 1 && dv <= 1) return launch_wgrad_tap<3, 1, 1, true>(g, dw, best_lds, ps, CT, OT);
    if (xv <= 2 && dv <= 1) return launch_wgrad_tap<3, 2, 1, true>(g, dw, best_lds, ps, CT, OT);
    return launch_wgrad_tap<3, 3, 2, true>(g, dw, best_lds, ps, CT, OT);
  }
  if (k == 3) {
    if (xv <= 1 && dv <= 1) return launch_wgrad_tap<3, 1, 1, false>(g, dw, best_lds, ps, CT, OT);
    if (xv <= 2 && dv <= 1) return launch_wgrad_tap<3, 2, 1, false>(g, dw, best_lds, ps, CT, OT);
    return launch_wgrad_tap<3, 3, 2, false>(g, dw, best_lds, ps, CT, OT);
  }
  if (xv <= 2 && dv <= 1) return launch_wgrad_tap<1, 2, 1, true>(g, dw, best_lds, ps, CT, OT);
  return launch_wgrad_tap<1, 3, 2, true>(g, dw, best_lds, ps, CT, OT);
}

}  // namespace

namespace dfhip {
// what a backward sweep has deferred (dfhip_wgrad_group_end ends the sweep with it; the perceptron's backward launch, which
// waits for the optimizer BEHIND the sweep, is not part of it)
static int flush_sweep() {
  if (!g_queue.empty()) {
    int st = group_flush();
    if (st != DFHIP_OK) return st;
  }
  if (g_stem_pending.active) {                      // no grouped slab sum took it along
    g_stem_pending.active = false;
    int st = stem_combine_launch(g_stem_pending.c, g_stem_pending.blocks);
    if (st != DFHIP_OK) return st;
  }
  if (g_head_pending.active) {                      // no grouped slab sum took them along
    g_head_pending.active = false;
    int st = head_grads_launch(g_head_pending.h);
    if (st != DFHIP_OK) return st;
  }
  if (g_bias_pending.n > 0) {                       // (cannot happen while a rider is only accepted for a queued layer's dY)
    SGroup sg = {};
    sg.bias = g_bias_pending;
    sg.bias_blocks = sg.bias.blk_end[sg.bias.n - 1];
    g_bias_pending.n = 0;
    ProfScope prof("reduce_to_shape", 0.0, 0.0);
    DF_LAUNCH(k_wgrad_group_sum, dim3((unsigned)sg.bias_blocks), dim3(256), 0, stream(), sg);
    DF_LAUNCH_CHECK();
  }
  if (!g_side_pending.empty()) {
    int st = side_join();
    if (st != DFHIP_OK) return st;
  }
  if (!g_def.active) return DFHIP_OK;
  g_def.active = false;
  g_def.slot = -1;
  return launch_slab_sum(g_def.a);
}
int flush_deferred() {
  const int st = flush_sweep();
  if (st != DFHIP_OK) return st;
  return mlp_bwd_flush();                           // (dfhip_mlp.hip: a perceptron's backward launch waiting for its optimizer)
}
bool deferred_touches(const void* p) {
  if (!p) return false;
  if (mlp_bwd_pending_touches(p)) return true;
  if (g_stem_pending.active) {
    const StemCombine& c = g_stem_pending.c;
    const float* q = (const float*)p;
    auto in = [&](const float* lo, int64_t n) { return lo && q >= lo && q < lo + n; };
    if (in(c.dw, (int64_t)c.O * c.KK) || in(c.dgamma, c.O) || in(c.dbeta, c.O) || in(c.dbias, c.O) || p == c.slabs) return true;
  }
  if (g_head_pending.active) {
    const HeadGrads& h = g_head_pending.h;
    const float* q = (const float*)p;
    auto in = [&](const float* lo, int64_t n) { return lo && q >= lo && q < lo + n; };
    if (in(h.dw, (int64_t)h.C * h.Pn) || in(h.db, h.Pn) || in(h.dl, (int64_t)h.N * h.Pn) || in(h.pooled, (int64_t)h.N * h.C) ||
        in(h.loss, 1) || in(h.rowloss, h.N)) return true;
  }
  for (int i = 0; i < g_bias_pending.n; i++) {
    const float* q = (const float*)p;
    if (q >= g_bias_pending.dst[i] && q < g_bias_pending.dst[i] + g_bias_pending.C[i]) return true;
  }
  for (const auto& sp : g_side_pending)              // a gradient in flight on the side stream: join first
    for (const auto& r : sp.dws)
      if ((const float*)p >= r.first && (const float*)p < r.second) return true;
  const float* q = (const float*)p;
  for (const auto& e : g_queue)
    if (q >= e.dw && q < e.dw + (int64_t)e.d.O * e.d.C * (e.tstride ? e.tstride : e.T)) return true;
  if (!g_def.active) return false;
  const float* lo = g_def.a.dw;
  return q >= lo && q < lo + (int64_t)g_def.a.O * g_def.a.C * g_def.a.T;
}
int deferred_scratch_slot() { return g_def.active ? g_def.slot : -1; }
bool group_collecting() { return g_group_on && !g_queue.empty() && !g_group_stream && g_side_pending.empty(); }
bool stem_combine_pending() { return g_stem_pending.active; }
bool head_grads_defer(const HeadGrads& h) {
  // (the head's backward is the FIRST node of a sweep: the group is open, nothing is queued yet; dfhip_wgrad_group_end flushes
  // what the group left behind)
  if (!g_group_on || g_group_stream || !g_side_pending.empty() || g_head_pending.active || overlap_cus()) return false;
  g_head_pending.h = h;
  g_head_pending.active = true;
  return true;
}
// dfhip_reduce_to_shape on (N, C, H, W) -> (1, C, 1, 1) inside a sweep: taken along by the group's slab-sum launch when `src` is
// the dY of a layer that is queued in the group (so the array is alive until the group is flushed: the caller keeps a queued
// layer's operands) and its planes are whole 16-byte pieces, a power of two of them
bool bias_sum_defer(const float* src, float* dst, int N, int C, int HW) {
  if (!group_collecting() || overlap_cus() || g_bias_pending.n >= kBiasSums || N <= 0 || C <= 0 || HW < 4) return false;
  if ((HW & (HW - 1)) != 0 || (((uintptr_t)src) & 15) != 0) return false;
  bool queued = false;
  for (const auto& e : g_queue)
    if (e.d.dy == src && e.d.N == N && e.d.O == C && e.d.oh * e.ow == HW && !e.d.bnz) queued = true;
  if (!queued) return false;
  BiasSums& b = g_bias_pending;
  const int i = b.n++;
  int lq = 0;
  while ((4 << lq) < HW) lq++;
  b.src[i] = src; b.dst[i] = dst; b.N[i] = N; b.C[i] = C; b.lq[i] = lq;
  b.blk_end[i] = (i ? b.blk_end[i - 1] : 0) + C;
  return true;
}
bool conv_matmul_dtype_is_bf16() { return g_matmul_dtype == 1; }
bool stem_combine_defer(const StemCombine& c, int blocks) {
  if (!group_collecting() || g_stem_pending.active) return false;
  g_stem_pending.c = c;
  g_stem_pending.blocks = blocks;
  g_stem_pending.active = true;
  return true;
}
}  // namespace dfhip

extern "C" {

int dfhip_conv2d_wgrad_bn(const float* x, const float* affine, const float* dy, const float* z, const float* table,
                          float* dw, int N, int C, int H, int W, int O, int k, int stride, int pad, int* done) {
  DF_INIT_KEEP();
  DF_TOUCH(x); DF_TOUCH(dy); DF_TOUCH(dw);
  DF_CHECK_ARG(x && dy && z && table && dw && done, "conv2d_wgrad_bn: null array");
  *done = 0;
  int oh, ow;
  int s = check_conv_args("conv2d_wgrad_bn", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  if (!g_group_on) return DFHIP_OK;
  const int st = run_wgrad_rows(x, dy, dw, N, C, H, W, O, oh, ow, k, stride, pad, affine, z, table);
  if (st < 0) return DFHIP_OK;
  if (st != DFHIP_OK) return st;
  *done = 1;
  return DFHIP_OK;
}

int dfhip_bn_bwd_table(const dfhip_bn_bwd_in* in, int O) {
  DF_INIT_KEEP();
  DF_CHECK_ARG(in && in->table && O > 0, "bn_bwd_table: null descriptor / table");
  int s = check_bn_in("bn_bwd_table", in);
  if (s != DFHIP_OK) return s;
  ProfScope prof("bn2d_bwd", 0.0, 4.0 * O * (2.0 * in->nrows + 8.0));
  DF_CHECK_ARG(O % 4 == 0 && O <= 256, "bn_bwd_table: O must be a multiple of 4, at most 256");
  DF_LAUNCH(k_bn_bwd_table, dim3(1), dim3(256), 0, stream(), *in, O);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_bn2d_backward_apply_table(const float* dy, const float* z, const float* table, float* dz, int N, int O,
                                    int HW) {
  DF_INIT_KEEP();
  DF_TOUCH(dy); DF_TOUCH(dz);
  DF_CHECK_ARG(dy && z && table && dz && N > 0 && O > 0 && HW > 0, "bn2d_backward_apply_table: null array / empty shape");
  return bn_bwd_materialise(dy, z, table, dz, N, O, HW);
}

int dfhip_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W,
                       int O, int k, int stride, int pad) {
  DF_INIT_KEEP();          // the row-unit kernel takes the pending slab sum along; other paths flush
  DF_TOUCH(x);
  DF_TOUCH(dy);
  DF_TOUCH(dw);
  int oh, ow;
  int s = check_conv_args("conv2d_wgrad", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(x && dy && dw, "conv2d_wgrad: null array");
  if (g_group_on) {        // queued for the grouped launch (shapes the row-unit kernel does not take run now)
    int st = run_wgrad_rows(x, dy, dw, N, C, H, W, O, oh, ow, k, stride, pad);
    if (st >= 0) return st;
    st = flush_deferred();   // (outside this call's profiling bracket)
    if (st != DFHIP_OK) return st;
  }
  double flops = 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + (double)N * O * oh * ow);
  ProfScope prof("conv2d_wgrad_igemm", flops, bytes);
  WGeo g = {};
  g.x = x; g.dy = dy;
  g.N = N; g.C = C; g.H = H; g.W = W; g.O = O; g.oh = oh; g.ow = ow;
  g.k = k; g.stride = stride; g.pad = pad;
  {
    int st = run_wgrad_rows(x, dy, dw, N, C, H, W, O, oh, ow, k, stride, pad);
    if (st >= 0) return st;
    st = flush_deferred();
    if (st != DFHIP_OK) return st;
    st = run_wgrad_tap(x, dy, dw, N, C, H, W, O, oh, ow, k, stride, pad);
    if (st >= 0) return st;
    st = run_wgrad_v3(g, dw);
    if (st >= 0) return st;
  }
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

// dW of a convolution whose forward pass applied y = x * scale + shift (per source channel) to its operand on load
// (dfhip_conv2d_forward_bn_input): the grouped direct body applies the same pair to its source rows; every other
// situation writes the operand out once and takes the ordinary paths.
int dfhip_conv2d_wgrad_affine(const float* x, const float* affine, const float* dy, float* dw, int N, int C, int H,
                              int W, int O, int k, int stride, int pad) {
  DF_INIT_KEEP();
  DF_TOUCH(x);
  DF_TOUCH(dy);
  DF_TOUCH(dw);
  int oh, ow;
  int s = check_conv_args("conv2d_wgrad_affine", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(x && affine && dy && dw, "conv2d_wgrad_affine: null array");
  if (g_group_on) {
    int st = run_wgrad_rows(x, dy, dw, N, C, H, W, O, oh, ow, k, stride, pad, affine);
    if (st >= 0) return st;
  }
  return wgrad_affine_materialised(x, affine, dy, dw, N, C, H, W, O, oh, ow, k, stride, pad);
}

// Backward pass of the stem pool(relu(bn(conv(x)))) of the CIFAR models (test/ResNet_CIFAR10_cuda.py:84-87) when
// x needs no gradient: the statistics pass of dfhip_bn_relu_maxpool2_backward (dgamma, dbeta), then the weight
// gradient of the convolution with d loss / d z rebuilt from z inside the kernel that consumes it.  z (N, O, oh, ow)
// is the convolution output, y / g (N, O, oh/2, ow/2) the pooled output and its gradient.  *done = 0 (nothing
// launched) when the shape has no such kernel: the caller then runs dfhip_bn_relu_maxpool2_backward and
// dfhip_conv2d_wgrad.
int dfhip_stem_backward(const float* x, const float* z, const float* y, const float* g, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_var, float* dw, float* dgamma,
                        float* dbeta, int N, int C, int H, int W, int O, int k, int stride, int pad, float eps,
                        int* done) {
  DF_INIT();
  DF_CHECK_ARG(x && z && y && g && save_mean && save_var && dw && done, "stem_backward: null array");
  *done = 0;
  int oh, ow;
  int s = check_conv_args("stem_backward", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  if (O > 32 || (oh & 1) || (ow & 7) || stride != 1 || C > 4) return DFHIP_OK;
  WGeo wg = {};
  wg.x = x; wg.dy = z;
  wg.N = N; wg.C = C; wg.H = H; wg.W = W; wg.O = O; wg.oh = oh; wg.ow = ow;
  wg.k = k; wg.stride = stride; wg.pad = pad;
  {
    // dry run of the plan: the statistics pass must not be launched for a shape the fused kernel then refuses
    WGeo t = wg;
    const int k2 = k * k, KW = C * k2;
    if ((ow & 1) || (H * W) % 4 != 0 || (oh * ow) % 4 != 0) return DFHIP_OK;
    const int KT = (k == 1 || KW <= 32) ? 1 : 5;
    const int CH = std::min(C, (KT * 32 + k2 - 2) / k2 + 1);
    bool ok = false;
    int gi = 1;
    for (int tp : {256, 128, 64, 32}) {
      plan_groups(N, oh, ow, tp, t.BI, t.BR, t.BC, gi, t.gr, t.gc);
      if (t.BC != ow || t.gc != 1 || N % t.BI != 0 || oh % t.BR != 0) continue;
      if ((t.BI * t.BR) % 8) continue;
      int RLr = t.gr == 1 ? H : std::min(H, (t.BR - 1) * stride + k);
      int RL = RLr * W, BRBC = t.BR * t.BC;
      if (RL % 4 || BRBC % 4) continue;
      if (t.gr > 1 && (W % 4 || ow % 4)) continue;
      int PE = t.BI * RL, BMP = t.BI * BRBC, guard = (k * W + k + 3) & ~3;
      size_t buf = (size_t)32 * (BMP + 4) + 2 * guard + (size_t)CH * PE;
      size_t lds = std::max(2 * buf, (size_t)2 * 8 * 16 * 64) * 4;
      if (lds <= 150 * 1024 && CH * PE / 4 <= 1024 && 32 * BMP / 4 <= 512 * 4) { ok = true; break; }
    }
    if (!ok) return DFHIP_OK;
  }
  const float* partial = nullptr;
  int S = 0;
  {
    ProfScope prof("bn_relu_pool_bwd", 0.0, (4.0 + 2.0) * (double)N * O * oh * ow);
    int st = stem_bwd_stats(z, y, g, gamma, beta, save_mean, save_var, N, O, oh, ow, eps, &partial, &S);
    if (st != DFHIP_OK) return st;
  }
  wg.st.y = y; wg.st.g = g; wg.st.gamma = gamma; wg.st.beta = beta; wg.st.mean = save_mean; wg.st.var = save_var;
  wg.st.partial = partial; wg.st.S = S; wg.st.eps = eps; wg.st.dgamma = dgamma; wg.st.dbeta = dbeta;
  double flops = 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + 1.5 * (double)N * O * oh * ow);
  ProfScope prof("conv2d_wgrad_igemm", flops, bytes);
  int st = run_wgrad_v3(wg, dw, true);
  if (st < 0) {
    set_error("stem_backward: the plan test and the kernel's own disagree (N %d C %d H %d W %d O %d k %d)", N, C, H, W, O, k);
    return DFHIP_RUNTIME_ERROR;
  }
  if (st != DFHIP_OK) return st;
  *done = 1;
  return DFHIP_OK;
}

// Grouped weight gradients.  Between _begin and _end, dfhip_conv2d_wgrad calls whose shape the
// row-unit kernel takes are QUEUED instead of launched; _end (or any entry point that flushes
// deferred work, or that touches a queued dw) runs all of them as one launch plus one slab-sum
// launch.  Contract: the x and dy arrays of a queued call stay allocated and unmodified until the
// flush; dw holds the gradient only after it.
int dfhip_set_conv_matmul_dtype(int dtype) {
  DF_INIT();
  DF_CHECK_ARG(dtype == 0 || dtype == 1, "set_conv_matmul_dtype: 0 (fp32) or 1 (bf16 operands, fp32 accumulation)");
  g_matmul_dtype = dtype;
  return DFHIP_OK;
}

int dfhip_conv_matmul_dtype(void) { return g_matmul_dtype; }

int dfhip_wgrad_group_begin(void) {
  DF_INIT();
  g_group_on = true;
  if (overlap_cus() && !capturing()) return chain_begin();
  return DFHIP_OK;
}

int dfhip_wgrad_group_end(void) {
  DF_INIT_KEEP();
  g_group_on = false;
  if (overlap_cus()) {
    // (DFHIP_OVERLAP_CUS is a DIAGNOSTIC switch: copies, synchronize and the collectives do not follow the chain stream.)
    // The side stream's launches join the chain, the chain joins the library stream, what is still queued runs on all CUs;
    // the chain ends whatever the join returned (an error used to leave every later launch on the chain stream)
    int st = DFHIP_OK;
    if (!g_side_pending.empty()) st = side_join();
    const int st2 = chain_end();
    if (st != DFHIP_OK) return st;
    if (st2 != DFHIP_OK) return st2;
  }
  return flush_sweep();
}

}  // extern "C"
