// Conv2d forward and input gradient as implicit GEMMs on the CDNA4 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), NCHW in and out.  (Weight gradients: dfhip_wgrad.hip; the overwrite-mode
// input-gradient family: dfhip_dgrad_ow.hip; register-operand forward kernels: dfhip_conv_rows.hip.)
//
// Reference algorithm being replaced (all Python over the device primitives):
//   forward  nn/functional.py:316-344  pad -> im2col (k*k strided gather/scatter pairs,
//            :263-283) -> permute/compact -> matmul(M=N*oh*ow, K=C*k*k, P=O) -> NCHW view
//   dgrad    tensor.py:708 (dC . B^T) -> __im2col2d.grad_fn (functional.py:285-294, the
//            overlapping windows are ASSIGNED in (i,j) row-major order: last writer wins)
//            -> __pad2d.grad_fn (:307-313)
//   wgrad    tensor.py:716 (A^T . dC)                                        [dfhip_wgrad.hip]
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
#include <cmath>
#include <vector>

#include "dfhip_internal.h"
#include "dfhip_bnbwd.h"
#include "dfhip_stem_combine.h"

using namespace dfhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// n / d for n*d < 2^32 with magic = ceil(2^32 / d); magic == 0 encodes d == 1
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic) {
  return magic ? __umulhi(n, magic) : n;
}

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
  int PT;                  // v3: 32-pixel tiles per workgroup (8 / PT k-split slices)
  unsigned long long* stamps;  // diagnostic builds only (DFHIP_CONV_STAMP): per-block time stamps
  unsigned m_perimg, m_pc, m_sd;   // exact division by multiplication (magic = ceil(2^32/d), 0 = divide by 1)
  unsigned m_brbc, m_bc, m_birl4, m_rl4, m_run4;
  // forward only: per-workgroup BatchNorm partials of the output, [gridDim.x][CD][3] = (count, mean,
  // centred second moment) over the workgroup's pixels, or null
  float* stats;
  int vec_store;   // 1: destination pixels of a tile are contiguous and 16-byte aligned in runs of 4
  const float* addsrc;   // dgrad: tensor of the destination's shape added to the result (gradient accumulation), or null
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
// forward / input-gradient, main kernel (source channels % 8 == 0, k in {1,3}).
//
// One workgroup = one job = (pixel group of PT 32-pixel tiles) x (32 destination channels) over
// the whole reduction.  All 8 waves run the matrix cores (two per SIMD, so one wave's LDS wait
// hides behind the other's MFMAs): wave w owns pixel tile  w % PT  and slice  w / PT  of the
// KSPLIT = 8 / PT slices of every channel chunk; the KSPLIT partial accumulators of a tile are
// summed through LDS at the end (no partial slabs in HBM, no second kernel).  The host picks PT
// so that every ResNet-18 layer at batch 256 is ~256-512 jobs of 144 MFMAs per wave.
//
// Staging moves the source planes AS THEY ARE (no halo, no gather): for one channel chunk the
// LDS image is [channel][image][rows x W], filled with 16-byte loads whose addresses are a
// scalar chunk base + a 32-bit lane offset computed once per job.  Padding, dilation (strided
// input gradient) and the "last writer" tap selection of the overwrite mode are handled on the
// READ side instead: every lane keeps, per tap, the LDS offset of its pixel's source element
// and one validity bit; an invalid (pixel, tap) reads offset 0 and is zeroed by a select, and a
// tap that is invalid for every lane of the wave is skipped.  A chunk is loaded into registers
// while the previous one is in the matrix cores and stored to the other LDS buffer before the
// barrier: a thread moves ~1 patch vector and ~1-5 weight vectors per chunk.
//
// Weights keep their GLOBAL order in LDS (a commit is a plain float4 store):
//   forward:  W[d0+d][cb+c][t] -> Wl[(c*k2 + t) * 33 + d]: transposed, row stride 33, so that the 32
//             A-operand lanes (d) of a read and the 64 lanes of a staging store (consecutive (c,t)
//             of one d) both fall into distinct banks (PMC: the [d][c*k2+t] image with a 16-byte
//             row pitch had as many bank-conflict cycles as active LDS cycles)
//   (was)     W[d0+d][cb+c][t] -> Wl[d * RS + c*k2 + t], RS = CCP*k2 + 4  (A-operand lanes run
//             along d: 4-way bank conflict on one ds_read_b32 per MFMA, well inside LDS budget)
//   dgrad:    W[cb+c][d0+d][t] -> Wl[c * 32*k2 + d*k2 + t]                (stride k2: conflict-free)
// ---------------------------------------------------------------------------------------
constexpr int kPV = 2;   // patch float4 vectors per thread per chunk
constexpr int kWV = 5;   // weight float4 vectors per thread per chunk

template <int KS, int MODE, int NPK>
__global__ __launch_bounds__(512, 2) void k_conv_igemm_v4(Geo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int k = KS, k2 = KS * KS;
  const int CCP = g.CC;                       // channels per chunk: 8, 16 or 32
  const int RS = MODE == 0 ? CCP * k2 + 4 : 32 * k2;
  const int PEp = g.PE;                       // floats per channel in the LDS image: BI * RL
  // channel pitch: the image plus one 16-byte slot that stays ZERO -- an invalid (pixel, tap) reads that
  // slot instead of being zeroed by a select in front of every MFMA
  const int PEs = PEp + 4;
  const int xs_floats = CCP * PEs;
  constexpr int WT = 33;                      // forward: row pitch of the transposed weight image
  const int ws_floats = MODE == 0 ? CCP * k2 * WT : CCP * RS;
  float* Xs0 = smem;
  float* Ws0 = Xs0 + 2 * xs_floats;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int PT = g.PT, KSPLIT = 8 / g.PT;
  const int pt = wave % PT, ks = wave / PT;
  [[maybe_unused]] int stamp_i = 0;
#ifdef DFHIP_STAMPS
#define DF_STAMP()                                                                              \
  do {                                                                                          \
    if (g.stamps && lane == 0 && (wave == 0 || wave == 7) && stamp_i < 30)                      \
      g.stamps[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave == 7)) * 32 + stamp_i++] = \
          __builtin_readcyclecounter();                                                         \
  } while (0)
#else
#define DF_STAMP() do { } while (0)
#endif
  DF_STAMP();

  int gx = blockIdx.x;
  const int grx = gx % g.gr; gx /= g.gr;
  const int n0 = gx * g.BI, r0 = grx * g.BR;
  const int d0 = blockIdx.y * 32;
  const int nchunks = g.CS / CCP;
  const int64_t src_plane = (int64_t)g.HS * g.WS;
  // source rows this group needs (clipped to the image): [su_lo, su_lo + RLr)
  const int RL = PEp / g.BI;                  // floats per (image, channel) in LDS = RLr * WS
  // first staged source row: the first row the band can touch, pulled back so that the RLr staged
  // rows never run past the image
  const int su_lo = min(max(0, (r0 * g.so + g.u0 >= 0 ? (r0 * g.so + g.u0) / g.sd : 0)), g.HS - RL / g.WS);

  // ---- staging descriptors (chunk-invariant) ----
  // patch: vector v of the chunk image -> (channel c, image, q4); LDS float index = 4 * v
  const int RL4 = RL >> 2;
  const int pvecs = CCP * g.BI * RL4;
  unsigned p_src[kPV];
  int p_lx[kPV];
#pragma unroll
  for (int u = 0; u < kPV; u++) {
    int v = tid + 512 * u;
    v = v < pvecs ? v : 0;
    int c = (int)fast_div((unsigned)v, g.m_birl4);
    int rem = v - c * (g.BI * RL4);
    p_lx[u] = c * PEs + 4 * rem;
    int im = (int)fast_div((unsigned)rem, g.m_rl4), q4 = rem - im * RL4;
    int imc = min(im, g.N - 1 - n0);            // images past the batch end are never read
    p_src[u] = (unsigned)(((int64_t)imc * g.CS + c) * src_plane + (int64_t)su_lo * g.WS + 4 * q4);
  }
  // the first chunk's source planes are requested NOW (they come from HBM, the weights from L2): the
  // weight descriptors below are computed while those loads are in flight
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 pv0, pv1, wv0, wv1, wv2, wv3, wv4;
#define DF_ISSUE_X(chunk)                                                                        \
  do {                                                                                           \
    const float* xb = g.src + ((int64_t)n0 * g.CS + (int64_t)(chunk) * CCP) * src_plane;         \
    pv0 = *reinterpret_cast<const f32x4*>(xb + p_src[0]);                                        \
    pv1 = *reinterpret_cast<const f32x4*>(xb + p_src[1]);                                        \
  } while (0)
  DF_ISSUE_X(0);
  __builtin_amdgcn_sched_barrier(0);
  const int run4 = (MODE == 0 ? CCP * k2 : 32 * k2) >> 2;
  const int wvecs = run4 * (MODE == 0 ? 32 : CCP);
  const int dmax = min(32, g.CD - d0);
  unsigned w_src[kWV];
  int w_lx[kWV];
#pragma unroll
  for (int u = 0; u < kWV; u++) {
    int f = tid + 512 * u;
    const bool live = f < wvecs;
    f = live ? f : 0;
    int row = (int)fast_div((unsigned)f, g.m_run4), e4 = f - row * run4;
    if (MODE == 0) w_src[u] = (unsigned)(((d0 + min(row, dmax - 1)) * g.wC) * k2 + 4 * e4);
    else w_src[u] = (unsigned)((row * g.wC + d0) * k2 + 4 * e4);     // host guarantees CD % 32 == 0
    w_lx[u] = live ? ((MODE == 0 && row >= dmax) ? -2 : row * RS + 4 * e4) : -1;
  }

#define DF_ISSUE_W(chunk)                                                                        \
  do {                                                                                           \
    const float* wb = g.wgt + (int64_t)(chunk) * CCP * (MODE == 0 ? k2 : g.wC * k2);            \
    wv0 = *reinterpret_cast<const f32x4*>(wb + w_src[0]);                                        \
    wv1 = *reinterpret_cast<const f32x4*>(wb + w_src[1]);                                        \
    wv2 = *reinterpret_cast<const f32x4*>(wb + w_src[2]);                                        \
    wv3 = *reinterpret_cast<const f32x4*>(wb + w_src[3]);                                        \
    wv4 = *reinterpret_cast<const f32x4*>(wb + w_src[4]);                                        \
  } while (0)
#define DF_ISSUE(chunk) do { DF_ISSUE_W(chunk); DF_ISSUE_X(chunk); } while (0)
#define DF_WSTORE(u, val)                                                                        \
  do {                                                                                           \
    if (MODE == 0) {                                                                             \
      if (w_lx[u] != -1) {                     /* transposed image: four scalar stores */        \
        int f_ = tid + 512 * (u);                                                                \
        int row_ = (int)fast_div((unsigned)f_, g.m_run4), e4_ = f_ - row_ * run4;                \
        const f32x4 v_ = w_lx[u] >= 0 ? (val) : (f32x4){0.f, 0.f, 0.f, 0.f};                     \
        float* q_ = Wl_ + (4 * e4_) * WT + row_;                                                 \
        q_[0] = v_.x; q_[WT] = v_.y; q_[2 * WT] = v_.z; q_[3 * WT] = v_.w;                       \
      }                                                                                          \
    } else if (w_lx[u] >= 0) *reinterpret_cast<f32x4*>(Wl_ + w_lx[u]) = (val);                   \
  } while (0)
#define DF_COMMIT(buf)                                                                           \
  do {                                                                                           \
    float* Xs_ = Xs0 + (buf) * xs_floats;                                                        \
    float* Wl_ = Ws0 + (buf) * ws_floats;                                                        \
    if (tid < pvecs) *reinterpret_cast<f32x4*>(Xs_ + p_lx[0]) = pv0;                             \
    if (tid + 512 < pvecs) *reinterpret_cast<f32x4*>(Xs_ + p_lx[1]) = pv1;                       \
    DF_WSTORE(0, wv0); DF_WSTORE(1, wv1); DF_WSTORE(2, wv2); DF_WSTORE(3, wv3); DF_WSTORE(4, wv4); \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) acc[rr] = 0.0f;

  DF_STAMP();
  DF_ISSUE_W(0);
  DF_STAMP();
  // (computed while the first chunk is in flight)
  // ---- per-lane, per-tap read descriptors of this wave's pixel tile ----
  const int q = pt * 32 + l31;
  const int img = (int)fast_div((unsigned)q, g.m_brbc);
  const int qrem = q - img * (g.BR * g.BC);
  const int r = (int)fast_div((unsigned)qrem, g.m_bc), col = qrem - r * g.BC;
  const bool pvalid = q < g.BI * g.BR * g.BC && n0 + img < g.N && r0 + r < g.HD;
  int poff[k2];
  unsigned vmask = 0;
  {
    int tapsel = -2;
    if (MODE == 2 && pvalid) {
      int ti = last_tap(r0 + r + g.pad, k, g.stride, g.oh);
      int tj = last_tap(col + g.pad, k, g.stride, g.ow);
      tapsel = (ti < 0 || tj < 0) ? -1 : ti * k + tj;
    }
#pragma unroll
    for (int t = 0; t < k2; t++) {
      const int i = t / k, j = t - i * k;
      const int ti = MODE == 0 ? i : k - 1 - i, tj = MODE == 0 ? j : k - 1 - j;
      int un = (r0 + r) * g.so + g.u0 + ti, vn = col * g.so + g.u0 + tj;
      bool ok = pvalid && un >= 0 && vn >= 0;
      int su = ok ? (int)fast_div((unsigned)un, g.m_sd) : 0, sv = ok ? (int)fast_div((unsigned)vn, g.m_sd) : 0;
      ok = ok && su * g.sd == un && sv * g.sd == vn && su < g.HS && sv < g.WS;
      if (MODE == 2) ok = ok && tapsel == t;
      poff[t] = ok ? img * RL + (su - su_lo) * g.WS + sv : PEp;      // PEp: the channel's zero slot
      vmask |= ok ? (1u << t) : 0u;
    }
  }

  if (tid < 2 * CCP) {                           // the zero slot of every channel of both buffers
    const int zb = tid >= CCP ? 1 : 0, zc = tid - zb * CCP;
    *reinterpret_cast<f32x4*>(Xs0 + zb * xs_floats + zc * PEs + PEp) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  DF_COMMIT(0);
  DF_STAMP();
  __syncthreads();
  DF_STAMP();

  // k pairs of a tap that this wave owns (>= 1); a compile-time count lets the tap x pair nest
  // unroll into straight-line code whose LDS reads are hoisted ahead of the MFMAs
  const int npk = NPK ? NPK : (CCP >> 1) / KSPLIT;
  const int c_first = 2 * ks * npk + half;    // this lane's first channel inside a chunk
  const int astep = MODE == 0 ? 2 * k2 * WT : 2 * RS;
  const int bstep = 2 * PEs;
  for (int ch = 0; ch < nchunks; ch++) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) DF_ISSUE(ch + 1);    // lands while this chunk is in the matrix cores
    const float* Xb = Xs0 + buf * xs_floats + c_first * PEs;
    const float* aq = Ws0 + buf * ws_floats + (MODE == 0 ? c_first * k2 * WT + l31 : c_first * RS + l31 * k2);
    if constexpr (MODE != 2 && NPK > 0) {
      // forward / accumulate: all taps straight-line (whole-wave-invalid taps are rare here), the
      // operands of tap t+1 are read into a second register set before the MFMAs of tap t issue
      float av[2][NPK ? NPK : 1], bv[2][NPK ? NPK : 1];
#pragma unroll
      for (int p = 0; p < NPK; p++) {
        av[0][p] = aq[p * astep];
        bv[0][p] = (Xb + poff[0])[p * bstep];
      }
#pragma unroll
      for (int t = 0; t < k2; t++) {
        if (t + 1 < k2) {
          const float* bp = Xb + poff[t + 1];
          const float* ap = aq + (MODE == 0 ? (t + 1) * WT : (t + 1));
#pragma unroll
          for (int p = 0; p < NPK; p++) {
            av[(t + 1) & 1][p] = ap[p * astep];
            bv[(t + 1) & 1][p] = bp[p * bstep];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NPK; p++)      // (an invalid lane's operand came from the zero slot)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][p], bv[t & 1][p], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll
    for (int t = 0; t < k2; t++) {
      const bool ok = (vmask >> t) & 1u;
      // overwrite mode: a tap nobody in the wave selected is skipped (most are)
      if (MODE != 2 || __ballot(ok) != 0ull) {
        const float* bp = Xb + poff[t];
        const float* ap = aq + (MODE == 0 ? t * WT : t);
        if (NPK) {
#pragma unroll
          for (int p = 0; p < (NPK ? NPK : 1); p++) {
            float a = ap[p * astep];
            float b = bp[p * bstep];
            b = ok ? b : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          }
        } else {
#pragma unroll 2
          for (int p = 0; p < npk; p++) {
            float a = ap[p * astep];
            float b = bp[p * bstep];
            b = ok ? b : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          }
        }
      }
    }
    DF_STAMP();
    if (ch + 1 < nchunks) DF_COMMIT(buf ^ 1);  // the other buffer was released at the last barrier
    DF_STAMP();
    __syncthreads();
    DF_STAMP();
  }
#undef DF_ISSUE
#undef DF_ISSUE_X
#undef DF_ISSUE_W
#undef DF_COMMIT
#undef DF_WSTORE

  // sum the KSPLIT partial accumulators of every pixel tile through LDS (staging buffers are free)
  float* red = smem;                    // [PT][KSPLIT-1][16][64]
  if (KSPLIT > 1) {
    if (ks > 0) {
      float* dstp = red + ((size_t)(pt * (KSPLIT - 1) + ks - 1) * 16) * 64 + lane;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) dstp[rr * 64] = acc[rr];
    }
    __syncthreads();
    if (ks == 0) {
      for (int s2 = 0; s2 < KSPLIT - 1; s2++) {
        const float* srcp = red + ((size_t)(pt * (KSPLIT - 1) + s2) * 16) * 64 + lane;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) acc[rr] += srcp[rr * 64];
      }
    }
  }

  // ---- epilogue: every finished tile goes through LDS as T[32 channels][36] so that it leaves
  // with 16-byte stores along the pixels (4 stores per lane instead of 16) and, for the forward
  // pass, so that its BatchNorm partials are row sums instead of cross-lane shuffles ----
  constexpr int TS = 36;
  const bool vec_ok = g.vec_store != 0;          // pixels of a tile are contiguous in runs of 4
  if (ks == 0 && g.bias) {
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
      int d = d0 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      acc[rr] += g.bias[min(d, g.CD - 1)];
    }
  }
  const bool want_stats = MODE == 0 && g.stats != nullptr;
  if (vec_ok || want_stats) {
    __syncthreads();                             // staging buffers / k-split sums are dead
    float* Tt = smem + pt * (32 * TS);
    if (ks == 0) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) Tt[((rr & 3) + 8 * (rr >> 2) + 4 * half) * TS + l31] = acc[rr];
    }
    __syncthreads();
    if (ks == 0 && vec_ok) {
      const int64_t dplane = (int64_t)g.HD * g.WD;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int f = lane + 64 * u;
        const int o = f >> 3, q4 = f & 7;
        const int qq = pt * 32 + 4 * q4;
        const int im = (int)fast_div((unsigned)qq, g.m_brbc);
        const int rem = qq - im * (g.BR * g.BC);
        const int rr_ = (int)fast_div((unsigned)rem, g.m_bc), cc_ = rem - rr_ * g.BC;
        const bool ok = qq < g.BI * g.BR * g.BC && n0 + im < g.N && r0 + rr_ < g.HD && d0 + o < g.CD;
        if (ok) {
          f32x4 v = *reinterpret_cast<const f32x4*>(Tt + o * TS + 4 * q4);
          const int64_t at = ((int64_t)(n0 + im) * g.CD + d0 + o) * dplane + (int64_t)(r0 + rr_) * g.WD + cc_;
          if (g.addsrc) v += *reinterpret_cast<const f32x4*>(g.addsrc + at);
          *reinterpret_cast<f32x4*>(g.dst + at) = v;
        }
      }
    }
  }
  if (ks == 0 && !vec_ok && pvalid) {
    const int64_t dplane = (int64_t)g.HD * g.WD;
    const int64_t at0 = (int64_t)(n0 + img) * g.CD * dplane + (int64_t)(r0 + r) * g.WD + col;
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
      int d = d0 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      if (d < g.CD) {
        float v = acc[rr];
        if (g.addsrc) v += g.addsrc[at0 + (int64_t)d * dplane];
        g.dst[at0 + (int64_t)d * dplane] = v;
      }
    }
  }
  if (want_stats) {
    // BatchNorm partials of this workgroup's output: lane (channel = lane & 31, half) walks 16 of
    // the tile's 32 pixels, the halves meet through one swap; exact two-pass (mean, centred
    // second moment) per tile, then Chan's combination of the PT tiles.
    float* sb = smem + PT * (32 * TS);           // [PT][32 channels][3]
    if (ks == 0) {
      const unsigned vm = (unsigned)__ballot(pvalid);          // validity of the tile's 32 pixels
      const float cnt = (float)__popc(vm);
      const float* row = smem + pt * (32 * TS) + l31 * TS + 16 * half;
      float vals[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * q4);
        vals[4 * q4] = t.x; vals[4 * q4 + 1] = t.y; vals[4 * q4 + 2] = t.z; vals[4 * q4 + 3] = t.w;
      }
      const unsigned hm = vm >> (16 * half);
      float t = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; i++) t += ((hm >> i) & 1u) ? vals[i] : 0.0f;
      t += __shfl_xor(t, 32);
      const float mean_t = cnt > 0.0f ? t / cnt : 0.0f;
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const float dlt = ((hm >> i) & 1u) ? vals[i] - mean_t : 0.0f;
        q += dlt * dlt;
      }
      q += __shfl_xor(q, 32);
      if (half == 0) {
        float* o = sb + (pt * 32 + l31) * 3;
        o[0] = cnt; o[1] = mean_t; o[2] = q;
      }
    }
    __syncthreads();
    if (tid < 32 && d0 + tid < g.CD) {
      float n = 0.0f, mu = 0.0f, m2 = 0.0f;
      for (int t = 0; t < PT; t++) {
        const float* o = sb + (t * 32 + tid) * 3;
        const float nb = o[0];
        if (nb > 0.0f) {
          const float nn = n + nb, delta = o[1] - mu;
          mu += delta * (nb / nn);
          m2 += o[2] + delta * delta * (n * nb / nn);
          n = nn;
        }
      }
      float* o = g.stats + ((int64_t)blockIdx.x * g.CD + d0 + tid) * 3;
      o[0] = n; o[1] = mu; o[2] = m2;
    }
  }
  DF_STAMP();
#undef DF_STAMP
}

// ---------------------------------------------------------------------------------------
// forward, few-channel 3x3 / stride 1 / pad 1 convolution on 32-wide images (the ResNet stem:
// 3 -> 32 channels on 32 x 32): K = C*9 <= 36, so a 32-pixel row is ceil(K/2) <= 18 MFMAs and the
// layer is bound by its 33.5 MB of output, not by the matrix cores.  A workgroup (8 waves) takes a band
// of up to 32 rows of one image: the band's input rows (+ one above and below, + one column either side,
// all zero outside the image) go to LDS with one 16-byte load per thread and round, and a wave owns an
// image ROW at a time (its 32 lanes are the row's pixels, W == 32): the weight operand of its 32 output
// channels sits in registers for the whole kernel, the pixel operand of a (channel, tap) is one LDS read
// (no masks: the padding is in the tile), and the 32 x 32 result leaves as sixteen 128-byte stores per
// half-wave.  The address unit then sees the stores and little else: with the operands as global loads
// (14 per row, 2-4 lines each) the kernel spent 2 us waiting for its first row and 12.5 us in all.
// ---------------------------------------------------------------------------------------
// partials (or nullptr): BatchNorm partials of y, one row per workgroup (blockIdx.x) -- (count, mean, centred second
// moment) per output channel, the layout of dfhip_conv2d_forward_stats.  A lane keeps sums of (y - s) and (y - s)^2
// over the rows it walks, s = the channel's value at the wave's first pixel (a sample of the channel: the second
// moment about it is as well conditioned as about the mean); the 32 pixels of a half-wave are combined by a halving
// butterfly (31 exchanges for the 32 sums of a lane, each lane ends with one of them), the eight waves through LDS.
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
template <int XM>
__device__ __forceinline__ float stem_swz(float v) {     // the value of lane (l ^ XM) of the same 32-lane group
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XM << 10) | 0x1f));
}
template <int C>
__global__ __launch_bounds__(512) void k_conv_rows3(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y,
                                                    int N, int H, int O, int rb, int bands, float* __restrict__ partials) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int KK = C * 9, NP = (KK + 1) / 2;
  constexpr int CH = kStemRows * kStemPitch;            // floats per channel tile
  constexpr int NQ = C * kStemRows * 8, QR = (NQ + 511) / 512;      // 16-byte pieces of the tile, rounds
  constexpr int NWR = (32 * KK + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int d0 = blockIdx.y * 32;
  __shared__ float Wl[32 * KK];
  __shared__ __attribute__((aligned(16))) float Xl[C * CH];
  __shared__ float sred[8][2][32], sshift[8][2][16], scount[8], Bl[32];
  STEM_STAMP(6); STEM_STAMP(0);
  const int items = N * bands;
  int item = blockIdx.x;
  // requests of the first tile and of the weights leave together
  auto tile_request = [&](int it, f32x4 (&v)[QR]) {
    const int n = it / bands, r0 = (it - n * bands) * rb;
    const float* xn = x + (int64_t)n * C * H * 32;
#pragma unroll
    for (int q0 = 0; q0 < QR; q0++) {
      const int q = min(tid + 512 * q0, NQ - 1);
      const int pc = q & 7, t = q >> 3, lr = t % kStemRows, c = t / kStemRows;
      const int row = min(max(r0 - 1 + lr, 0), H - 1);                      // (clamped: the commit zeroes what is outside)
      v[q0] = *reinterpret_cast<const f32x4*>(xn + ((int64_t)c * H + row) * 32 + pc * 4);
    }
  };
  auto tile_commit = [&](int it, const f32x4 (&v)[QR]) {
    const int n = it / bands, r0 = (it - n * bands) * rb;
    const int nr = min(rb, H - r0);
#pragma unroll
    for (int q0 = 0; q0 < QR; q0++) {
      const int q = tid + 512 * q0;
      const int pc = q & 7, t = q >> 3, lr = t % kStemRows, c = t / kStemRows;
      const int row = r0 - 1 + lr;
      const bool in = lr < nr + 2 && row >= 0 && row < H;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (q < NQ) *reinterpret_cast<f32x4*>(&Xl[c * CH + lr * kStemPitch + 4 + pc * 4]) = in ? v[q0] : z;
    }
  };
  f32x4 tv[QR];
  float wv[NWR];
  if (item < items) tile_request(item, tv);
  {
    const int nw = min(32, O - d0) * KK;
#pragma unroll
    for (int i0 = 0; i0 < NWR; i0++) {
      const int i = tid + 512 * i0;
      const float t = w[(int64_t)d0 * KK + min(i, nw - 1)];
      wv[i0] = i < nw ? t : 0.0f;
    }
#pragma unroll
    for (int i0 = 0; i0 < NWR; i0++) {
      const int i = tid + 512 * i0;
      if (i < 32 * KK) Wl[i] = wv[i0];
    }
    if (tid >= 480) {                                   // (the last lanes: they have the fewest weight pieces)
      const int dl = tid - 480;
      float t = 0.0f;
      if (bias) t = bias[min(d0 + dl, O - 1)];
      Bl[dl] = t;
    }
    // the columns either side of the image (never written again)
    if (tid < C * kStemRows * 2) {
      const int t = tid >> 1;
      Xl[(t / kStemRows) * CH + (t % kStemRows) * kStemPitch + ((tid & 1) ? 36 : 3)] = 0.0f;
    }
  }
  if (item < items) tile_commit(item, tv);
  __syncthreads();
  STEM_STAMP(1);
  float a[NP];
  int koff[NP];                                      // LDS index of (channel, tap) for output row 0 of the band
#pragma unroll
  for (int p = 0; p < NP; p++) {
    // k index 2p (lower half-wave) / 2p+1 (upper): (channel, tap row, tap column); a k beyond K reads tap 0 against a
    // zero weight
    const int kk = 2 * p + half;
    a[p] = kk < KK ? Wl[l31 * KK + kk] : 0.0f;
    const int ke = kk < KK ? kk : 0;
    const int cc = ke / 9, ii = (ke % 9) / 3, jj = ke % 3;
    koff[p] = cc * CH + ii * kStemPitch + jj + 3 + l31;
  }
  float bv[16];
#pragma unroll
  for (int rr = 0; rr < 16; rr++) {
    bv[rr] = Bl[(rr & 3) + 8 * (rr >> 2) + 4 * half];
  }
  float st_s[16], st_1[16], st_2[16];          // shift, sum of (y - shift), sum of (y - shift)^2
  float st_n = 0.0f;
  bool first = true;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) { st_s[rr] = 0.0f; st_1[rr] = 0.0f; st_2[rr] = 0.0f; }
  STEM_STAMP(2);
  while (item < items) {
    const int n = item / bands, r0 = (item - n * bands) * rb;
    const int nr = min(rb, H - r0);
    const int next = item + gridDim.x;
    if (next < items) tile_request(next, tv);
    float b[NP], bn[NP];
    if (wave < nr) {
#pragma unroll
      for (int p = 0; p < NP; p++) b[p] = Xl[koff[p] + wave * kStemPitch];
    }
    for (int r = wave; r < nr; r += 8) {
      if (r + 8 < nr) {
#pragma unroll
        for (int p = 0; p < NP; p++) bn[p] = Xl[koff[p] + (r + 8) * kStemPitch];
      }
      f32x16 acc;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) acc[rr] = bv[rr];
#pragma unroll
      for (int p = 0; p < NP; p++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b[p], acc, 0, 0, 0);
      float* yo = y + ((int64_t)n * O * H + r0 + r) * 32 + l31;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        const int d = d0 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        if (d < O) yo[(int64_t)d * H * 32] = acc[rr];
      }
      if (partials) {
        if (first) {
#pragma unroll
          for (int rr = 0; rr < 16; rr++)
            st_s[rr] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc[rr]), 0));   // lane 0 of the half
          first = false;
        }
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          const float dl = acc[rr] - st_s[rr];
          st_1[rr] += dl;
          st_2[rr] = fmaf(dl, dl, st_2[rr]);
        }
        st_n += 32.0f;
      }
      if (r + 8 < nr) {
#pragma unroll
        for (int p = 0; p < NP; p++) b[p] = bn[p];
      }
    }
    item = next;
    if (item < items) {
      __syncthreads();                           // every wave has read its rows of the tile
      tile_commit(item, tv);
      __syncthreads();
    }
  }
  STEM_STAMP(3); STEM_STAMP(4);
  if (partials) {
    // 32 sums per lane -> one per lane: at each step a lane hands one half of what it holds to its partner
    float v[32];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) { v[rr] = st_1[rr]; v[16 + rr] = st_2[rr]; }
#define DF_STEM_HALVE(XM, CNT)                                                         \
    {                                                                                  \
      const bool up = (l31 & XM) != 0;                                                 \
      _Pragma("unroll") for (int i = 0; i < CNT / 2; i++) {                            \
        const float lo = v[i], hi = v[i + CNT / 2];                                    \
        v[i] = (up ? hi : lo) + stem_swz<XM>(up ? lo : hi);                            \
      }                                                                                \
    }
    DF_STEM_HALVE(16, 32) DF_STEM_HALVE(8, 16) DF_STEM_HALVE(4, 8) DF_STEM_HALVE(2, 4) DF_STEM_HALVE(1, 2)
#undef DF_STEM_HALVE
    sred[wave][half][l31] = v[0];                // lane l holds sum number l: (y - s) of register l, (y - s)^2 of l - 16
    if (l31 == 0) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) sshift[wave][half][rr] = st_s[rr];
      if (half == 0) scount[wave] = st_n;
    }
    __syncthreads();
    if (tid < 32 && d0 + tid < O) {
      const int hh = (tid >> 2) & 1, rr = (tid & 3) + 4 * (tid >> 3);
      // the waves' sums are moved to one shift (wave 0's: every shift is a sample of the channel, so their distances
      // are of the size of its spread): sum (y - S) = a1 + n d, sum (y - S)^2 = a2 + 2 d a1 + n d^2 with d = s - S
      const float S = sshift[0][hh][rr];                 // (wave 0 has a row whenever the workgroup has an item)
      float n = 0.0f, A1 = 0.0f, A2 = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) {
        const float nb = scount[w8];
        const float a1 = sred[w8][hh][rr], a2 = sred[w8][hh][16 + rr];
        const float d = nb > 0.0f ? sshift[w8][hh][rr] - S : 0.0f;
        n += nb;
        A1 += fmaf(nb, d, a1);
        A2 += fmaf(d, fmaf(nb, d, 2.0f * a1), a2);
      }
      const float mu_s = n > 0.0f ? A1 / n : 0.0f;
      float* o = partials + ((int64_t)blockIdx.x * O + d0 + tid) * 3;
      o[0] = n; o[1] = S + mu_s; o[2] = n > 0.0f ? A2 - A1 * mu_s : 0.0f;
    }
  }
  STEM_STAMP(5); STEM_STAMP(7);
}


// ---------------------------------------------------------------------------------------
// forward, 1x1 / stride 2 / pad 0 (the ResNet shortcut convolutions): K = C is 16-64 MFMAs per
// 32 x 32 tile, so like the overwrite input gradient this is a kernel without staging phases.  A
// wave owns 32 consecutive output pixels x 32 output channels: its pixel operand x[n][c][2r][2s] is
// requested in full (C/2 loads per lane) together with the workgroup's weight tile (LDS, once), then
// C/2 MFMAs, then the staged kernel's epilogue: the tile through LDS, 16-byte stores, BatchNorm
// partials (count, mean, centred second moment per workgroup and channel).
// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_conv1x1s2_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ stats, int H, int lw, int O, int DT) {
  // DT: tiles of 32 output channels a workgroup computes from the pixel operand it loaded ONCE (the stride-2 gather
  // uses half of every sector it fetches; with one channel tile per workgroup the 64-channel 32 x 32 layer of the wide
  // ResNet fetched its input four times: 30 us for 1.07 GFLOP)
  constexpr int NP = C / 2, WT = 33, TS = 36;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                                 // [DT][c][WT] (transposed weight tiles)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int W = 1 << lw, HW = H << lw;
  const int olw = lw - 1, OW = W >> 1, OHW = HW >> 2;
  const int tile = blockIdx.x * 4 + wave;
  const int p = tile * 32 + l31;                    // flat output pixel (n, r, s)
  const int n = p / OHW, rem = p - n * OHW;
  const int r = rem >> olw, sc = rem & (OW - 1);
  float b[NP];
  {
    const float* xb = x + ((int64_t)n * C + half) * HW + ((2 * r) << lw) + 2 * sc;
#pragma unroll
    for (int kp = 0; kp < NP; kp++) b[kp] = xb[(int64_t)(2 * kp) * HW];
  }
  const int dbase = blockIdx.y * DT * 32;
  {
    constexpr int RUN4 = C / 4;                     // float4s per output channel
    const int dmax = min(32 * DT, O - dbase);
    for (int f = tid; f < 32 * DT * RUN4; f += 256) {
      const int row = f / RUN4, e4 = f - row * RUN4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < dmax) v = *reinterpret_cast<const float4*>(w + (int64_t)(dbase + row) * C + 4 * e4);
      float* q = Wl + (row >> 5) * (C * WT) + (4 * e4) * WT + (row & 31);
      q[0] = v.x; q[WT] = v.y; q[2 * WT] = v.z; q[3 * WT] = v.w;
    }
  }
  __syncthreads();
  float* Tt = smem + DT * (C * WT) + wave * (32 * TS);   // [4][32 channels][TS]
  float* sb = smem + DT * (C * WT) + 4 * (32 * TS);      // [4][32][3]
  for (int dt = 0; dt < DT; dt++) {
    const int d0 = dbase + dt * 32;
    if (d0 >= O) break;
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; rr++) acc[rr] = 0.0f;
    {
      const float* aq = Wl + dt * (C * WT) + half * WT + l31;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[(2 * kp) * WT], b[kp], acc, 0, 0, 0);
    }
    if (bias) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) acc[rr] += bias[min(d0 + (rr & 3) + 8 * (rr >> 2) + 4 * half, O - 1)];
    }
#pragma unroll
    for (int rr = 0; rr < 16; rr++) Tt[((rr & 3) + 8 * (rr >> 2) + 4 * half) * TS + l31] = acc[rr];
    // (each wave reads back only its own tile: no barrier needed before the stores)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int f = lane + 64 * u, o = f >> 3, q4 = f & 7;
      if (d0 + o < O) {
        const int pp = tile * 32 + 4 * q4;            // four consecutive output pixels (same image: OHW % 4 == 0)
        const int nn = pp / OHW, rr0 = pp - nn * OHW;
        const float4 v = *reinterpret_cast<const float4*>(Tt + o * TS + 4 * q4);
        *reinterpret_cast<float4*>(y + ((int64_t)nn * O + d0 + o) * OHW + rr0) = v;
      }
    }
    if (stats) {
      const float* row = Tt + l31 * TS + 16 * half;
      float vals[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const float4 t4 = *reinterpret_cast<const float4*>(row + 4 * q4);
        vals[4 * q4] = t4.x; vals[4 * q4 + 1] = t4.y; vals[4 * q4 + 2] = t4.z; vals[4 * q4 + 3] = t4.w;
      }
      float t = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; i++) t += vals[i];
      t += __shfl_xor(t, 32);
      const float mean_t = t * (1.0f / 32.0f);
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; i++) { const float dlt = vals[i] - mean_t; q += dlt * dlt; }
      q += __shfl_xor(q, 32);
      if (half == 0) {
        float* o3 = sb + (wave * 32 + l31) * 3;
        o3[0] = 32.0f; o3[1] = mean_t; o3[2] = q;
      }
      __syncthreads();
      if (tid < 32 && d0 + tid < O) {
        float cn = 0.0f, mu = 0.0f, m2 = 0.0f;
        for (int tt = 0; tt < 4; tt++) {
          const float* o3 = sb + (tt * 32 + tid) * 3;
          const float nb = o3[0], nn = cn + nb, delta = o3[1] - mu;
          mu += delta * (nb / nn);
          m2 += o3[2] + delta * delta * (cn * nb / nn);
          cn = nn;
        }
        float* o3 = stats + ((int64_t)blockIdx.x * O + d0 + tid) * 3;
        o3[0] = cn; o3[1] = mu; o3[2] = m2;
      }
      __syncthreads();                                // sb and the tiles are rewritten by the next channel tile
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same layer with ALL FOUR output positions in one workgroup (16 images x 16 output channels; 16x16x4 MFMA:
// D[16 channels][16 images] += A[16][4 source channels] . B[4][16]).  Every 36-byte weight record (o, c) is then read
// whole, in memory order and once per workgroup (the one-position kernel reads 16 bytes of each from four workgroups:
// the gather was 2.7 us of its 12.9), and the 2 x 2 input planes serve four positions.  The eight waves split the
// source channels and stream their own slices -- eight channels at a time -- through wave-private LDS areas with the
// next slice's requests in flight: no block barrier before the first MFMA.  Lane (kk, n) of the B operand is image n,
// channel 4 p + kk; output position q takes input pixel s through tap (sy - qy + 1, sx - qx + 1): sixteen MFMAs per
// k-step on nine weight operands.  Sums through LDS, 16-byte stores (the four positions of a channel are adjacent),
// BatchNorm partials over the 16 images x 4 positions of a channel.
// ---------------------------------------------------------------------------------------
// DG (accumulate-mode input gradient of the same layer): x is dY (C = the convolution's output channels), w the convolution's
// own [C][O][3][3] tensor -- a slice's records are then rows of 16 x 9 consecutive floats per contraction channel, stored with
// the tap flipped (the gradient's tap of (source s, position q) is 8 - the forward's) -- and `bias` is the gradient the
// tensor already holds, [N][O][2][2], added in the epilogue.
// BNC (forward only): x is the RAW output of the previous convolution and the operand is its training-mode BatchNorm
// (dfhip_conv2d_forward_bn_input; no activation in between).  No launch turns the producer's partial rows into a (scale, shift)
// table: a wave needs the pairs of its own CW channels only -- lane (g, c) adds up rows g, g + 64 / CW, ... of channel cb + c
// (requested in front of the first slice), the groups meet through lane exchanges, the operand lanes fetch their channel's
// pair from lane c.  Workgroup (0, 0) leaves the table and the statistics behind for the backward pass.  (The padding needs
// no care: a tap outside the image is never multiplied.)
template <int C, bool DG = false, bool BNC = false>
__global__ __launch_bounds__(512) void k_conv3x3_img2_all(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ stats, int O, BnInput bn) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
  static_assert(!(DG && BNC), "the BatchNorm operand is a forward form");
  constexpr int CW = C / 8;                     // channels per wave
  constexpr int NCHK = CW / 8;                  // slices of eight channels
  constexpr int WP = 17;                        // pitch of a (channel, tap) row of 16 output channels
  constexpr int WSL = 72 * WP + 16 * 9 * 4;     // floats of a wave's area: weights [72][17], planes [16 images][9 float4]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const DfTile wg = {(int)blockIdx.x, (int)blockIdx.y};          // (image tile, channel tile) in grid order
  const int n0 = wg.x * 16, o0 = wg.y * 16;
  float* Wl = smem + wave * WSL;
  float* Xs = Wl + 72 * WP;
  const int cb = wave * CW;                     // this wave's first source channel
  // requests of a slice: 16 records rows of 72 floats (eight channels x nine taps of one output channel) = 288 float4,
  // 16 planes rows of 32 floats (eight channels x four pixels of one image) = 128 float4
  f32x4 wq[5], xq[2];
  auto request = [&](int ch) {
    const int c0 = cb + 8 * ch;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      if constexpr (DG) {
        const int e = min(lane + 64 * i, 287), oj = e / 36, j = e - 36 * oj;
        wq[i] = *reinterpret_cast<const f32x4*>(w + ((int64_t)(c0 + oj) * O + o0) * 9 + 4 * j);
      } else {
        const int e = min(lane + 64 * i, 287), m = e / 18, j = e - 18 * m;
        wq[i] = *reinterpret_cast<const f32x4*>(w + ((int64_t)(o0 + m) * C + c0) * 9 + 4 * j);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int e = lane + 64 * i, im = e >> 3, j = e & 7;
      xq[i] = *reinterpret_cast<const f32x4*>(x + ((int64_t)(n0 + im) * C + c0 + j) * 4);
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int BG = 64 / CW, BRP = 8;          // row groups of the wave (2 at 32 channels, 4 at 16); rows per lane and pass
  static_assert(!BNC || CW == 16 || CW == 32, "a wave's channels are a half or a quarter of its lanes");
  f32x3u bn_row[BNC ? BRP : 1];
  float bn_shift = 0.0f, bn_g = 1.0f, bn_b = 0.0f, bn_rm = 0.0f, bn_rv = 0.0f;
  const int bc = lane % CW, bg = lane / CW;
  if constexpr (BNC) {
    const float* pc = bn.part + (cb + bc) * 3;
#pragma unroll
    for (int k2 = 0; k2 < BRP; k2++)
      bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(bg + BG * k2, bn.rows - 1) * (C * 3));
    bn_shift = pc[1];
    if (bn.gamma) bn_g = bn.gamma[cb + bc];
    if (bn.beta) bn_b = bn.beta[cb + bc];
    if (blockIdx.x == 0 && blockIdx.y == 0) {                         // (only the workgroup that updates them)
      if (bn.running_mean) bn_rm = bn.running_mean[cb + bc];
      if (bn.running_var) bn_rv = bn.running_var[cb + bc];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  request(0);
  float bn_sc = 1.0f, bn_sh = 0.0f;
  if constexpr (BNC) {
    float tn = 0.0f, t1 = 0.0f, t2 = 0.0f;
    auto add_rows = [&](int r0) {
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++) {
        const bool in = r0 + bg + BG * k2 < bn.rows;
        const float nb = in ? bn_row[k2][0] : 0.0f, d = nb > 0.0f ? bn_row[k2][1] - bn_shift : 0.0f;
        tn += nb;
        t1 = fmaf(nb, d, t1);
        t2 += fmaf(nb * d, d, in ? bn_row[k2][2] : 0.0f);
      }
    };
    add_rows(0);
    for (int r0 = BG * BRP; r0 < bn.rows; r0 += BG * BRP) {
      const float* pc = bn.part + (cb + bc) * 3;
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++)
        bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(r0 + bg + BG * k2, bn.rows - 1) * (C * 3));
      add_rows(r0);
    }
    if constexpr (BG == 4) { tn += __shfl_xor(tn, 16, 64); t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64); }
    tn += __shfl_xor(tn, 32, 64); t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
    const float mu_s = tn > 0.0f ? t1 / tn : 0.0f;
    const float mu = bn_shift + mu_s;
    const float var = tn > 0.0f ? fmaxf(t2 - t1 * mu_s, 0.0f) / tn : 0.0f;            // biased variance (batchnorm.py:41-42)
    const float rsd = 1.0f / sqrtf(var + bn.eps);
    bn_sc = bn_g * rsd;
    bn_sh = bn_b - mu * bn_sc;
    if (blockIdx.x == 0 && blockIdx.y == 0 && bg == 0) {
      const int c = cb + bc;
      if (bn.save_mean) bn.save_mean[c] = mu;
      if (bn.save_var) bn.save_var[c] = var;
      if (bn.running_mean) bn.running_mean[c] = bn_rm * (1.0f - bn.momentum) + mu * bn.momentum;
      if (bn.running_var) bn.running_var[c] = bn_rv * (1.0f - bn.momentum) + var * bn.momentum;
      bn.affine[2 * c] = bn_sc;
      bn.affine[2 * c + 1] = bn_sh;
    }
  }
#pragma unroll
  for (int ch = 0; ch < NCHK; ch++) {
    // commit: weight float4 (m, j) holds (channel, tap) rows 4 j .. 4 j + 3 of output channel m
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int e = lane + 64 * i, m = min(e, 287) / 18, j = min(e, 287) - 18 * m;
      if (e < 288) {
        if constexpr (DG) {
          const int oj = min(e, 287) / 36, jj = min(e, 287) - 36 * oj;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int f = 4 * jj + t, mo = f / 9, tap = f - 9 * mo;
            Wl[(oj * 9 + 8 - tap) * WP + mo] = wq[i][t];
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; t++) Wl[(4 * j + t) * WP + m] = wq[i][t];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int e = lane + 64 * i, im = e >> 3, j = e & 7;
      *reinterpret_cast<f32x4*>(Xs + (im * 9 + j) * 4) = xq[i];
    }
    if (ch + 1 < NCHK) request(ch + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float a[2][9];
    f32x4 b[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
      for (int t = 0; t < 9; t++) a[p][t] = Wl[((4 * p + kk) * 9 + t) * WP + n];
      b[p] = *reinterpret_cast<const f32x4*>(Xs + (n * 9 + 4 * p + kk) * 4);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (BNC) {
#pragma unroll
      for (int p = 0; p < 2; p++) {      // (lane c of the wave holds the pair of channel cb + c)
        const float sc = __shfl(bn_sc, 8 * ch + 4 * p + kk, 64), sh = __shfl(bn_sh, 8 * ch + 4 * p + kk, 64);
#pragma unroll
        for (int e = 0; e < 4; e++) b[p][e] = fmaf(b[p][e], sc, sh);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int sp = 0; sp < 4; sp++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int tap = ((sp >> 1) - (q >> 1) + 1) * 3 + ((sp & 1) - (q & 1) + 1);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][tap], b[p][sp], acc[q], 0, 0, 0);
        }
  }
  // the eight waves' sums: waves 1..7 hand theirs over, wave 0 adds them in order
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(smem);
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) ex[((wave - 1) * 4 + q) * 64 + lane] = acc[q];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int ww = 0; ww < 7; ww++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += ex[(ww * 4 + q) * 64 + lane];
  // register v of lane (kk, n) is output channel o0 + 4 kk + v of image n0 + n; its four positions are adjacent
#pragma unroll
  for (int v = 0; v < 4; v++) {
    if constexpr (DG) {
      f32x4 r = {acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
      if (bias) r += *reinterpret_cast<const f32x4*>(bias + ((int64_t)(n0 + n) * O + o0 + 4 * kk + v) * 4);
      *reinterpret_cast<f32x4*>(y + ((int64_t)(n0 + n) * O + o0 + 4 * kk + v) * 4) = r;
    } else {
      const float bv = bias ? bias[o0 + 4 * kk + v] : 0.0f;
      f32x4 r = {acc[0][v] + bv, acc[1][v] + bv, acc[2][v] + bv, acc[3][v] + bv};
      acc[0][v] = r[0]; acc[1][v] = r[1]; acc[2][v] = r[2]; acc[3][v] = r[3];
      *reinterpret_cast<f32x4*>(y + ((int64_t)(n0 + n) * O + o0 + 4 * kk + v) * 4) = r;
    }
  }
  if (stats) {
    // exact two-pass over the 64 values of a channel: the lane's four positions, then the 16 images of its DPP row
    auto row16 = [](float t) {
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xf, 0xf, true));   // row_ror:8
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xf, 0xf, true));   // row_ror:4
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x122, 0xf, 0xf, true));   // row_ror:2
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x121, 0xf, 0xf, true));   // row_ror:1
      return t;
    };
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float mean = row16((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) * (1.0f / 64.0f);
      float qs = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) { const float d = acc[q][v] - mean; qs = fmaf(d, d, qs); }
      qs = row16(qs);
      if (n == 0) {
        float* o3 = stats + ((int64_t)wg.x * O + o0 + 4 * kk + v) * 3;
        o3[0] = 64.0f; o3[1] = mean; o3[2] = qs;
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

}  // namespace
namespace dfhip {
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
}  // namespace dfhip
namespace {

}  // namespace
namespace dfhip {
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
}  // namespace dfhip
namespace {

constexpr size_t kLdsBudget = 64 * 1024;  // keeps two workgroups resident per CU


template <int KS, int MODE, int NPK>
int launch_igemm_v4n(const Geo& g, dim3 grid, size_t lds) {
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv_igemm_v4<KS, MODE, NPK>,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr_lds = std::max(lds, (size_t)65536);
  }
  DF_LAUNCH((k_conv_igemm_v4<KS, MODE, NPK>), grid, dim3(512), lds, stream(), g);
  return DFHIP_OK;
}

template <int KS, int MODE>
int launch_igemm_v4(const Geo& g, dim3 grid, size_t lds) {
  const int npk = (g.CC >> 1) / (8 / g.PT);
  if (npk == 1) return launch_igemm_v4n<KS, MODE, 1>(g, grid, lds);
  if (npk == 2) return launch_igemm_v4n<KS, MODE, 2>(g, grid, lds);
  if (npk == 4) return launch_igemm_v4n<KS, MODE, 4>(g, grid, lds);
  if (npk == 8) return launch_igemm_v4n<KS, MODE, 8>(g, grid, lds);
  return launch_igemm_v4n<KS, MODE, 0>(g, grid, lds);
}

int finish_slabs(const Geo& g, int splits, float* final_dst, const float* bias, int64_t out_elems) {
  if (splits > 1) {
    int nb = (int)std::min<int64_t>(df_ceil_div(out_elems, 256), (int64_t)num_cus() * 8);
    DF_LAUNCH(k_conv_slab_sum, dim3(nb), dim3(256), 0, stream(), g.dst, bias, final_dst,
                       out_elems, splits, (int64_t)g.HD * g.WD, g.CD);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

static int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// main kernel; returns -1 when the shape is not eligible (the generic kernel takes it then)
int run_igemm_v4(Geo g, int dest_channels, float* final_dst, const float* bias, float* partials = nullptr,
                 int cap_rows = 0, int* rows_out = nullptr) {
  if (g.CS % 8 != 0 || !(g.k == 1 || g.k == 3 || g.k == 5)) return -1;
  if (g.mode != 0 && g.CD % 32 != 0) return -1;    // dgrad weight rows are staged as whole 32-channel runs
  if ((g.HS * g.WS) % 4 != 0) return -1;            // channel planes must stay 16-byte aligned
  const int k2 = g.k * g.k;
  const int dtiles = (int)df_ceil_div(dest_channels, 32);
  int gi = 1, chosen = 0, ccp = 0;
  size_t lds = 0;
  for (int pt : {8, 4, 2, 1}) {
    const int ksplit = 8 / pt;
    plan_groups(g.N, g.HD, g.WD, 32 * pt, g.BI, g.BR, g.BC, gi, g.gr, g.gc);
    if (g.BC != g.WD || g.gc != 1) continue;        // whole destination rows only
    if (g.HD % g.BR != 0) continue;                 // equal row bands (same source row count)
    // source rows a group touches, clipped to the image; identical for every band when computed
    // for the widest (interior) case, so use the un-clipped span and clip per group in the kernel
    int span_lo = floor_div(g.u0, g.sd), span_hi = floor_div((g.BR - 1) * g.so + g.u0 + g.k - 1, g.sd);
    int RLr = std::min(g.HS, span_hi - span_lo + 1);
    if (g.gr == 1) RLr = g.HS;                      // whole-image groups stage the whole plane
    int RL = RLr * g.WS;
    if (RL % 4 != 0 || (g.gr > 1 && (g.WS % 4 != 0))) continue;
    g.PE = g.BI * RL;
    int64_t jobs = (int64_t)gi * g.gr * dtiles;
    constexpr int jobmul = 1;
    if (jobs < (int64_t)num_cus() * jobmul && pt > 1) continue;
    constexpr size_t lds_cap_single = (size_t)(150) * 1024;
    auto fits = [&](int cc) {
      size_t wl = g.mode == 0 ? (size_t)33 * cc * k2 : (size_t)cc * 32 * k2;
      size_t stage_b = 2 * ((size_t)cc * (g.PE + 4) + wl) * 4;    // (+4: the zero slot of a channel)
      size_t red_b = (size_t)pt * (ksplit - 1) * 16 * 64 * 4;
      lds = std::max(stage_b, red_b);
      int pvecs = cc * g.PE / 4, wvecs = 8 * cc * k2;
      // two workgroups per CU only pay when there are more jobs than CUs; otherwise take the LDS
      const size_t cap = jobs > (int64_t)num_cus() ? 76 * 1024 : lds_cap_single;
      return lds <= cap && pvecs <= 512 * kPV && wvecs <= 512 * kWV;
    };
    constexpr int maxchunks = 2;
    int c = std::max(8, 2 * ksplit);
    while (c < 32 && g.CS / c > maxchunks && g.CS % (2 * c) == 0 && fits(2 * c)) c *= 2;
    if (g.CS % c != 0 || !fits(c)) continue;
    chosen = pt;
    ccp = c;
    break;
  }
  if (!chosen) return -1;
  g.PT = chosen;
  g.CC = ccp;
  g.m_sd = magic_u32((unsigned)g.sd);
  g.m_brbc = magic_u32((unsigned)(g.BR * g.BC));
  g.m_bc = magic_u32((unsigned)g.BC);
  {
    const int RL4 = g.PE / g.BI / 4;
    g.m_birl4 = magic_u32((unsigned)(g.BI * RL4));
    g.m_rl4 = magic_u32((unsigned)RL4);
    g.m_run4 = magic_u32((unsigned)((g.mode == 0 ? ccp * k2 : 32 * k2) / 4));
  }
  g.cs_per_split = g.CS;
  g.slab_stride = 0;
  g.dst = final_dst;
  g.bias = bias;
  dim3 grid((unsigned)(gi * g.gr), (unsigned)dtiles, 1);
  int st = DFHIP_OK;
  g.stats = nullptr;
  g.vec_store = ((g.BR * g.BC) % 4 == 0 && (g.HD * g.WD) % 4 == 0) ? 1 : 0;
  if (partials) {
    if ((int)grid.x > cap_rows) return -1;
    g.stats = partials;
    if (rows_out) *rows_out = (int)grid.x;
  }
  lds = std::max(lds, (size_t)chosen * (32 * 36 + 32 * 3) * sizeof(float));
  g.stamps = nullptr;
  static int stamp_mode = -1;
  if (stamp_mode < 0) stamp_mode = getenv("DFHIP_CONV_STAMP") ? atoi(getenv("DFHIP_CONV_STAMP")) : 0;
  const size_t nstamp = (size_t)grid.x * grid.y * 2 * 32;
  if (stamp_mode) {
    g.stamps = (unsigned long long*)workspace2(nstamp * 8);
    (void)hipMemsetAsync(g.stamps, 0, nstamp * 8, stream());
  }
#define DF_V4(KSV)                                                        \
  (g.mode == 0 ? launch_igemm_v4<KSV, 0>(g, grid, lds)                    \
               : (g.mode == 1 ? launch_igemm_v4<KSV, 1>(g, grid, lds)     \
                              : launch_igemm_v4<KSV, 2>(g, grid, lds)))
  if (g.k == 1) st = DF_V4(1);
  else if (g.k == 3) st = DF_V4(3);
  else {    // 5x5 (the CNN workload's second convolution)
    st = DF_V4(5);           // (the tap count at compile time: -20 % on the CNN's 5x5 layer against the run-time loop)
  }
#undef DF_V4
  if (st != DFHIP_OK) return st;
  DF_LAUNCH_CHECK();
  if (stamp_mode) {
    std::vector<unsigned long long> h(nstamp);
    (void)hipStreamSynchronize(stream());
    (void)hipMemcpy(h.data(), g.stamps, nstamp * 8, hipMemcpyDeviceToHost);
    fprintf(stderr, "[stamp4] mode %d k %d PT %d CC %d PE %d grid %u x %u lds %zu\n", g.mode, g.k, g.PT, g.CC, g.PE,
            grid.x, grid.y, lds);
    {
      // whole-grid view: span from the first block's first stamp to the last block's last stamp,
      // and the mean duration of every segment of wave 0 over all blocks
      const size_t nb = nstamp / 64;
      unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
      double seg[32] = {0};
      int nseg = 0;
      for (size_t b = 0; b < nb; b++) {
        const unsigned long long* r = &h[(b * 2) * 32];
        int n = 0;
        while (n < 32 && r[n]) n++;
        if (!n) continue;
        tmin = std::min(tmin, r[0]);
        smax = std::max(smax, r[0]);
        tmax = std::max(tmax, r[n - 1]);
        for (int i = 1; i < n; i++) seg[i] += (double)(r[i] - r[i - 1]);
        nseg = std::max(nseg, n);
      }
      fprintf(stderr, "[stamp4]  grid span %lld, last block start +%lld; mean segments:", (long long)(tmax - tmin),
              (long long)(smax - tmin));
      for (int i = 1; i < nseg; i++) fprintf(stderr, " %.0f", seg[i] / nb);
      fprintf(stderr, "\n");
    }
  }
  return DFHIP_OK;
}

// 3x3 / stride 1 / pad 1 on 2 x 2 images (128 or 256 source channels, batch a multiple of 32); -1 when not eligible
int run_conv_img2(const Geo& g, int O, float* y, const float* bias, float* partials, int cap_rows, int* rows_out) {
  if (g.mode != 0 || g.k != 3 || g.so != 1 || g.sd != 1 || g.u0 != -1) return -1;
  if (g.HS != 2 || g.WS != 2 || g.HD != 2 || g.WD != 2 || (g.N % 32) != 0) return -1;
  if (!(g.CS == 128 || g.CS == 256)) return -1;
  // all four positions per workgroup (k_conv3x3_img2_all): 16 images x 16 channels; other channel counts / more rows than
  // the partials buffer holds take the staged kernel (the one-position-per-workgroup kernel of round 2 is gone: superseded)
  if (O % 16 != 0 || g.N % 16 != 0 || (partials && g.N / 16 > cap_rows)) return -1;
  const int rows4 = g.N / 16;
  if (rows_out) *rows_out = partials ? rows4 : 0;
  dim3 grid4((unsigned)rows4, (unsigned)(O / 16));
  const size_t lds4 = std::max((size_t)8 * (72 * 17 + 16 * 9 * 4), (size_t)7 * 4 * 64 * 4) * sizeof(float);
  // XCD arrangement: blockIdx.y selects 16 output channels' weights, blockIdx.x 16 images' planes
  if (g.CS == 128) DF_LAUNCH(k_conv3x3_img2_all<128>, grid4, dim3(512), lds4, stream(), g.src, g.wgt, bias, y, partials, O, BnInput{});
  else DF_LAUNCH(k_conv3x3_img2_all<256>, grid4, dim3(512), lds4, stream(), g.src, g.wgt, bias, y, partials, O, BnInput{});
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// the same layer with the training-mode BatchNorm of its input applied on the operand loads, the table built by every wave
// for its own channels (k_conv3x3_img2_all<.., BNC>); -1 when the shape is not one it takes
int run_conv_img2_bn(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int O, int k,
                     int stride, int pad, float* partials, int cap_rows, int* rows_out, const BnInput& bn, bool dry_run) {
  if (k != 3 || stride != 1 || pad != 1 || H != 2 || W != 2 || (N % 32) != 0) return -1;
  if (!(C == 128 || C == 256) || O % 16 != 0 || !bn.part || bn.rows < 1 || !bn.affine) return -1;
  const int rows4 = N / 16;
  if (partials && rows4 > cap_rows) return -1;
  if (rows_out) *rows_out = partials ? rows4 : 0;
  if (dry_run) return DFHIP_OK;
  dim3 grid4((unsigned)rows4, (unsigned)(O / 16));
  const size_t lds4 = std::max((size_t)8 * (72 * 17 + 16 * 9 * 4), (size_t)7 * 4 * 64 * 4) * sizeof(float);
  if (C == 128) DF_LAUNCH((k_conv3x3_img2_all<128, false, true>), grid4, dim3(512), lds4, stream(), x, w, bias, y, partials, O, bn);
  else DF_LAUNCH((k_conv3x3_img2_all<256, false, true>), grid4, dim3(512), lds4, stream(), x, w, bias, y, partials, O, bn);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// shortcut convolution (1x1 / stride 2 / pad 0, 32-256 source channels); -1 when not eligible
int run_conv1x1s2(const Geo& g, int O, float* y, const float* bias, float* partials, int cap_rows, int* rows_out) {
  if (g.mode != 0 || g.k != 1 || g.so != 2 || g.sd != 1 || g.u0 != 0) return -1;
  if (!(g.CS == 32 || g.CS == 64 || g.CS == 128 || g.CS == 256)) return -1;
  if ((g.HS & 1) || (g.WS & 1) || g.HD * 2 != g.HS || g.WD * 2 != g.WS) return -1;
  int lw = 0;
  while ((1 << lw) < g.WS) lw++;
  if ((1 << lw) != g.WS) return -1;
  const int64_t opix = (int64_t)g.N * g.HD * g.WD;
  if (opix % 128 != 0 || (g.HD * g.WD) % 4 != 0) return -1;
  const int rows = (int)(opix / 128);
  if (partials && rows > cap_rows) return -1;
  if ((int64_t)rows * df_ceil_div(O, 32) < num_cus() / 2) return -1;   // too few workgroups: the k-split kernel wins
  if (rows_out) *rows_out = partials ? rows : 0;
  // channel tiles per workgroup: as many as still leave two workgroups per CU (and 64 KB of LDS per workgroup)
  const int OT = (int)df_ceil_div(O, 32);
  constexpr int dtmax = 8;
  int DT = 1;
  auto lds_of = [&](int dt) { return ((size_t)dt * g.CS * 33 + 4 * 32 * 36 + 4 * 96) * sizeof(float); };
  // (more tiles per workgroup with 87-155 KB of LDS and one workgroup per CU were measured: 128 channels 23.6 -> 26.2 us)
  while (DT * 2 <= std::min(OT, dtmax) && (int64_t)rows * df_ceil_div(OT, DT * 2) >= 2 * (int64_t)num_cus() &&
         lds_of(DT * 2) <= 64 * 1024)
    DT *= 2;
  const size_t lds = lds_of(DT);
  dim3 grid((unsigned)rows, (unsigned)df_ceil_div(OT, DT));
  if (g.CS == 32) DF_LAUNCH(k_conv1x1s2_fwd<32>, grid, dim3(256), lds, stream(), g.src, g.wgt, bias, y, partials, g.HS, lw, O, DT);
  else if (g.CS == 64) DF_LAUNCH(k_conv1x1s2_fwd<64>, grid, dim3(256), lds, stream(), g.src, g.wgt, bias, y, partials, g.HS, lw, O, DT);
  else if (g.CS == 128) DF_LAUNCH(k_conv1x1s2_fwd<128>, grid, dim3(256), lds, stream(), g.src, g.wgt, bias, y, partials, g.HS, lw, O, DT);
  else DF_LAUNCH(k_conv1x1s2_fwd<256>, grid, dim3(256), lds, stream(), g.src, g.wgt, bias, y, partials, g.HS, lw, O, DT);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// few-channel 3x3 stem on 32-wide images; returns -1 when the shape is not one it takes
int run_conv_rows3(const Geo& g, int dest_channels, float* final_dst, const float* bias, float* partials = nullptr,
                   int cap_rows = 0, int* rows_out = nullptr) {
  if (g.mode != 0 || g.k != 3 || g.so != 1 || g.sd != 1 || g.u0 != -1) return -1;
  if (g.CS < 1 || g.CS > 4 || g.WS != 32 || g.WD != 32 || g.HS != g.HD) return -1;
  if ((int64_t)g.N * df_ceil_div(g.HS, 8) >= ((int64_t)1 << 31)) return -1;        // band index in an int
  // a workgroup per band of rb rows of an image (the whole image when there are enough images to fill the chip);
  // more bands than partial rows / twice the CUs: the workgroups walk them
  int rb = 32;
  while (rb > 8 && (int64_t)g.N * df_ceil_div(g.HS, rb) < num_cus()) rb /= 2;
  const int bands = df_ceil_div(g.HS, rb);
  const int64_t items = (int64_t)g.N * bands;
  const bool want_part = partials && rows_out && cap_rows > 0;
  const int64_t cap = want_part ? std::min<int64_t>(cap_rows, (int64_t)num_cus() * 2) : (int64_t)num_cus() * 2;
  dim3 grid((unsigned)std::min<int64_t>(items, cap), (unsigned)df_ceil_div(dest_channels, 32));
  float* part = want_part ? partials : nullptr;
  if (rows_out) *rows_out = part ? (int)grid.x : 0;
#define DF_STEM(CV)                                                                                   \
  case CV:                                                                                            \
    DF_LAUNCH(k_conv_rows3<CV>, grid, dim3(512), 0, stream(), g.src, g.wgt, bias, final_dst, g.N, g.HS, \
                       dest_channels, rb, bands, part);                                               \
    break;
  switch (g.CS) { DF_STEM(1) DF_STEM(2) DF_STEM(3) DF_STEM(4) default: return -1; }
#undef DF_STEM
  DF_LAUNCH_CHECK();
#ifdef DFHIP_STEM_STAMPS
  {
    static int calls = 0;
    if (++calls % 50 == 0) {
      std::vector<unsigned long long> h((size_t)grid.x * 8 * 8);
      (void)hipStreamSynchronize(stream());
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stem_stamps), h.size() * sizeof(unsigned long long));
      double seg[6] = {0}, cmax = 0;
      unsigned long long r0 = ~0ull, r1 = 0;
      const size_t nw = (size_t)grid.x * 8;
      for (size_t b = 0; b < nw; b++) {
        const unsigned long long* r = &h[b * 8];
        for (int k = 1; k <= 5; k++) seg[k] += (double)(r[k] - r[k - 1]);
        cmax = std::max(cmax, (double)(r[5] - r[0]));
        r0 = std::min(r0, r[6]); r1 = std::max(r1, r[7]);
      }
      fprintf(stderr, "[stem-stamp] grid %u stats %d: span %.2f us; mean cycles per wave: tile + weights %.0f, operand setup %.0f, "
              "rows %.0f, - %.0f, stats tail %.0f; slowest wave %.0f\n", grid.x, part != nullptr,
              (double)(r1 - r0) / 100.0, seg[1] / nw, seg[2] / nw, seg[3] / nw, seg[4] / nw, seg[5] / nw, cmax);
    }
  }
#endif
  return DFHIP_OK;
}

int run_igemm(Geo g, int dest_channels, float* final_dst, const float* bias) {
  {
    int st = run_igemm_v4(g, dest_channels, final_dst, bias);
    if (st >= 0) return st;
    st = run_conv_rows3(g, dest_channels, final_dst, bias);
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
  DF_LAUNCH(k_conv_igemm, grid, dim3(256), lds, stream(), g);
  DF_LAUNCH_CHECK();
  return finish_slabs(g, splits, final_dst, bias, out_elems);
}



}  // namespace

extern "C" {

static int conv2d_forward_impl(const float* x, const float* w, const float* bias, float* y, int N, int C, int H,
                               int W, int O, int k, int stride, int pad, float* partials, int cap_rows,
                               int* rows_out) {
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
  if (rows_out) *rows_out = 0;
  {
    int rows = 0;
    const bool want = partials && rows_out;
    int st = run_conv_rowmma(x, w, bias, y, N, C, H, W, O, k, stride, pad, want ? partials : nullptr, cap_rows, &rows);
    if (st < 0) st = run_conv1x1s2(g, O, y, bias, want ? partials : nullptr, cap_rows, &rows);
    if (st < 0) st = run_conv_img2(g, O, y, bias, want ? partials : nullptr, cap_rows, &rows);
    if (st >= 0) {
      if (want) *rows_out = rows;
      return st;
    }
  }
  if (partials && rows_out) {
    int rows = 0;
    int st = run_igemm_v4(g, O, y, bias, partials, cap_rows, &rows);
    if (st < 0) st = run_conv_rows3(g, O, y, bias, partials, cap_rows, &rows);     // the 1-4 channel stem
    if (st >= 0) {
      *rows_out = rows;
      return st;
    }
  }
  return run_igemm(g, O, y, bias);
}

int dfhip_conv2d_forward(const float* x, const float* w, const float* bias, float* y, int N, int C,
                         int H, int W, int O, int k, int stride, int pad) {
  DF_INIT();
  return conv2d_forward_impl(x, w, bias, y, N, C, H, W, O, k, stride, pad, nullptr, 0, nullptr);
}

// forward + BatchNorm partials of y: partials[rows][O][3] = (count, mean, centred second moment) of
// each output channel over the pixels of one workgroup; *rows_out = 0 when the shape took a kernel
// that does not produce them (the caller then uses dfhip_bn2d_forward_train)
int dfhip_conv2d_forward_stats(const float* x, const float* w, const float* bias, float* y, int N, int C,
                               int H, int W, int O, int k, int stride, int pad, float* partials,
                               int cap_rows, int* rows_out) {
  DF_INIT();
  DF_CHECK_ARG(partials && rows_out && cap_rows > 0, "conv2d_forward_stats: null partials / rows_out");
  return conv2d_forward_impl(x, w, bias, y, N, C, H, W, O, k, stride, pad, partials, cap_rows, rows_out);
}

// y = conv2d(x, w; 3x3, stride 2, pad 1) and yd = conv2d(x, wd; 1x1, stride 2, pad 0) -- conv1 and downsample[0] of a
// residual block, which read the same tensor (test/ResNet_CIFAR10_cuda.py:31-47) -- as ONE launch: the shortcut reads
// exactly the 3x3 convolution's centre-tap pixels and rides as a tenth tap.  BatchNorm partials of both outputs as
// dfhip_conv2d_forward_stats (partials / partials_d: cap_rows x O x 3, same row count for both).  *done = 0 (nothing
// launched) when the shape has no such kernel: the caller then runs the two convolutions separately.
int dfhip_conv2d_forward_s2_pair(const float* x, const float* w, const float* wd, float* y, float* yd, int N, int C,
                                 int H, int W, int O, float* partials, float* partials_d, int cap_rows, int* rows_out,
                                 int* done) {
  DF_INIT();
  DF_CHECK_ARG(x && w && wd && y && yd && done, "conv2d_forward_s2_pair: null array");
  DF_CHECK_ARG((partials == nullptr) == (partials_d == nullptr), "conv2d_forward_s2_pair: partials of both outputs or of none");
  *done = 0;
  if (rows_out) *rows_out = 0;
  int oh, ow;
  int s = check_conv_args("conv2d_forward_s2_pair", N, C, H, W, O, 3, 2, 1, oh, ow);
  if (s != DFHIP_OK) return s;
  int rows = 0;
  const bool want = partials && rows_out;
  if (run_conv3x3s2_pair(x, w, wd, y, yd, N, C, H, W, O, want ? partials : nullptr, want ? partials_d : nullptr,
                         cap_rows, &rows, true) < 0)
    return DFHIP_OK;
  double flops = 2.0 * N * oh * ow * (double)C * 10 * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * 10 + 2.0 * N * O * oh * ow);
  ProfScope prof("conv2d_fwd_igemm", flops, bytes);
  int st = run_conv3x3s2_pair(x, w, wd, y, yd, N, C, H, W, O, want ? partials : nullptr, want ? partials_d : nullptr,
                              cap_rows, &rows);
  if (st < 0) {
    set_error("conv2d_forward_s2_pair: the shape test and the kernel's own disagree");
    return DFHIP_RUNTIME_ERROR;
  }
  if (st != DFHIP_OK) return st;
  if (want) *rows_out = rows;
  *done = 1;
  return DFHIP_OK;
}

static int conv2d_dgrad_impl(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H,
                             int W, int O, int k, int stride, int pad, int mode, const BnBwdIn* bi = nullptr,
                             BnStatsOut* so = nullptr) {
  int oh, ow;
  int s = check_conv_args("conv2d_dgrad", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null array");
  DF_CHECK_ARG(mode == 0 || mode == 1, "conv2d_dgrad: mode must be 0 (overwrite) or 1 (accumulate)");
  // accumulate mode: the implicit GEMM's 2 M K P.  overwrite mode: what the reference's k*k assignments leave is ONE
  // tap per input pixel (functional.py:285-294), i.e. 2 N H W C O executed flops -- 1/k^2 of the GEMM the reference
  // computes and then mostly discards; the family is bound by its bytes, not by the matrix cores.
  double flops = mode == 0 ? 2.0 * N * H * W * (double)C * O : 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W * (add ? 2 : 1) + (double)O * C * k * k + (double)N * O * oh * ow);
  // the BatchNorm-backward forms read more (round 6: the per-kernel table booked 19.6 MB for a launch that reads 42): z of the
  // pending BatchNorm under the dY operand, the inputs of the layers whose row sums the launch leaves behind, the rows
  if (bi) bytes += 4.0 * ((double)N * O * oh * ow + (double)bi->nrows * O * bi->K);
  if (so) bytes += 4.0 * so->n * (double)N * C * H * W;
  ProfScope prof(mode == 0 ? "conv2d_dgrad_overwrite_igemm" : "conv2d_dgrad_igemm", flops, bytes);
  Geo g = {};
  g.src = dy; g.wgt = w;
  g.N = N; g.CS = O; g.HS = oh; g.WS = ow;
  g.CD = C; g.HD = H; g.WD = W;
  g.k = k; g.so = 1; g.sd = stride; g.u0 = pad - (k - 1);
  g.mode = mode == 0 ? 2 : 1; g.wC = C; g.pad = pad; g.stride = stride; g.oh = oh; g.ow = ow;
  if (mode == 0) {
    // overwrite mode: the kernels without phases of dfhip_dgrad_ow.hip (one surviving tap per input pixel)
    const int st = run_dgrad_ow(dy, w, add, dx, N, C, H, W, O, k, stride, pad, bi, so);
    if (st != -1) {
      if (st == -2) { set_error("conv2d_dgrad: no kernel with the BatchNorm descriptors for this shape (query dfhip_conv2d_bn_bwd_supported first)"); return DFHIP_INVALID_ARGUMENT; }
      return st;
    }
  }
  if (bi || so) {
    set_error("conv2d_dgrad: no kernel with the BatchNorm descriptors for this shape (query dfhip_conv2d_bn_bwd_supported first)");
    return DFHIP_INVALID_ARGUMENT;
  }
  if (mode == 1) {
    // 2 x 2 images: the forward's all-positions kernel with the roles of the channel counts swapped (k_conv3x3_img2_all<DG>)
    if (k == 3 && stride == 1 && pad == 1 && H == 2 && W == 2 && N % 16 == 0 && C % 16 == 0 &&
        (O == 128 || O == 256)) {
      dim3 grid4((unsigned)(N / 16), (unsigned)(C / 16));
      const size_t lds4 = std::max((size_t)8 * (72 * 17 + 16 * 9 * 4), (size_t)7 * 4 * 64 * 4) * sizeof(float);
      if (O == 128) DF_LAUNCH((k_conv3x3_img2_all<128, true>), grid4, dim3(512), lds4, stream(), dy, w, add, dx, (float*)nullptr, C, BnInput{});
      else DF_LAUNCH((k_conv3x3_img2_all<256, true>), grid4, dim3(512), lds4, stream(), dy, w, add, dx, (float*)nullptr, C, BnInput{});
      DF_LAUNCH_CHECK();
      return DFHIP_OK;
    }
    int st = run_conv_rowmma_dgrad(dy, w, add, dx, N, C, H, W, O, k, stride, pad);
    if (st >= 0) return st;
    if (k == 3 && stride == 2 && pad == 1) {                 // one product per (pixel, tap) pair that exists
      st = run_dgrad_s2_acc(dy, w, nullptr, nullptr, add, dx, N, C, H, W, O);
      if (st >= 0) return st;
    }
  }
  if (add) {
    g.addsrc = add;
    int st = run_igemm_v4(g, C, dx, nullptr);
    if (st >= 0) return st;
    g.addsrc = nullptr;                      // the generic kernel has no such epilogue: add afterwards
    st = run_igemm(g, C, dx, nullptr);
    if (st != DFHIP_OK) return st;
    return dfhip_ewise_add(dx, add, dx, (int64_t)N * C * H * W);
  }
  return run_igemm(g, C, dx, nullptr);
}

int dfhip_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int C, int H, int W,
                       int O, int k, int stride, int pad, int mode) {
  DF_INIT_KEEP();          // sits between two weight gradients of a backward pass
  DF_TOUCH(dy);
  DF_TOUCH(w);
  DF_TOUCH(dx);
  return conv2d_dgrad_impl(dy, w, nullptr, dx, N, C, H, W, O, k, stride, pad, mode);
}

// the overwrite-mode pair launch (dfhip_dgrad_ow.hip) inside its profiling bracket; *done = 0: no kernel for the shape
static int dgrad_pair_overwrite(const float* dy, const float* w, const float* dy2, const float* w2, const float* add,
                                float* dx, int N, int C, int H, int W, int O, const BnBwdIn* bi, const BnBwdIn* bi2,
                                BnStatsOut* so, int* done) {
  const int oh = H / 2, ow = W / 2;
  const int t = run_dgrad_ow_pair(dy, w, dy2, w2, add, dx, N, C, H, W, O, bi, bi2, so, true);
  if (t < 0) {
    if (bi || bi2 || so) { set_error("conv2d_dgrad_pair: no kernel with the BatchNorm descriptors for this shape"); return DFHIP_INVALID_ARGUMENT; }
    return DFHIP_OK;
  }
  const double flops = 2.0 * N * H * W * (double)C * O + 2.0 * N * oh * ow * (double)C * O;
  double bytes = 4.0 * ((double)N * C * H * W * (add ? 2 : 1) + (double)O * C * 10 + 2.0 * N * O * oh * ow);
  if (bi) bytes += 4.0 * ((double)N * O * oh * ow + (double)bi->nrows * O * bi->K);
  if (bi2) bytes += 4.0 * ((double)N * O * oh * ow + (double)bi2->nrows * O * bi2->K);
  if (so) bytes += 4.0 * so->n * (double)N * C * H * W;
  ProfScope prof("conv2d_dgrad_overwrite_igemm", flops, bytes);
  const int st = run_dgrad_ow_pair(dy, w, dy2, w2, add, dx, N, C, H, W, O, bi, bi2, so);
  if (st < 0) { set_error("conv2d_dgrad_pair: the shape test and the kernel's own disagree"); return DFHIP_RUNTIME_ERROR; }
  if (st != DFHIP_OK) return st;
  *done = 1;
  return DFHIP_OK;
}

// dx = dgrad(3x3 / stride 2 / pad 1; dy, w) + dgrad(1x1 / stride 2 / pad 0; dy2, w2) (+ add): the two consumers of a
// residual block's input whose shortcut is a convolution (test/ResNet_CIFAR10_cuda.py:58-63: conv1 and downsample[0]
// both read x), reference-exact ("overwrite") mode, as ONE launch -- the shortcut's gradient lands on the even pixels,
// parity class 0 of the stride-2 kernel.  *done = 0 (nothing launched) when the shape has no such kernel: the caller
// then runs dfhip_conv2d_dgrad and dfhip_conv2d_dgrad_add.
int dfhip_conv2d_dgrad_pair(const float* dy, const float* w, const float* dy2, const float* w2, const float* add,
                            float* dx, int N, int C, int H, int W, int O, int mode, int* done) {
  DF_INIT_KEEP();
  DF_TOUCH(dy); DF_TOUCH(w); DF_TOUCH(dy2); DF_TOUCH(w2); DF_TOUCH(add); DF_TOUCH(dx);
  DF_CHECK_ARG(dy && w && dy2 && w2 && dx && done, "conv2d_dgrad_pair: null array");
  *done = 0;
  int oh, ow;
  int s = check_conv_args("conv2d_dgrad_pair", N, C, H, W, O, 3, 2, 1, oh, ow);
  if (s != DFHIP_OK) return s;
  if (mode == 1) {
    // accumulate mode: the stride-2 kernel that multiplies only the (pixel, tap) pairs that exist, shortcut included
    double flopsa = 2.0 * N * oh * ow * (double)C * 10 * O;
    double bytesa = 4.0 * ((double)N * C * H * W * (add ? 2 : 1) + (double)O * C * 10 + 2.0 * N * O * oh * ow);
    if (run_dgrad_s2_acc(dy, w, dy2, w2, add, dx, N, C, H, W, O, true) < 0) return DFHIP_OK;    // (shape test only)
    ProfScope profa("conv2d_dgrad_igemm", flopsa, bytesa);
    const int sta = run_dgrad_s2_acc(dy, w, dy2, w2, add, dx, N, C, H, W, O);
    if (sta != DFHIP_OK) return sta < 0 ? DFHIP_RUNTIME_ERROR : sta;
    *done = 1;
    return DFHIP_OK;
  }
  if (mode != 0) return DFHIP_OK;
  return dgrad_pair_overwrite(dy, w, dy2, w2, add, dx, N, C, H, W, O, nullptr, nullptr, nullptr, done);
}

// ---- the BatchNorm-backward forms (dfhip_bnbwd.h) ----
static int check_bn_out(const char* fn, const BnStatsOut* o) {
  if (!o) return DFHIP_OK;
  DF_CHECK_ARG((o->n == 1 || o->n == 2) && o->rows && o->cap_rows >= 1, "%s: dfhip_bn_stats_out n %d, cap_rows %d",
               fn, o->n, o->cap_rows);
  for (int j = 0; j < o->n; j++) DF_CHECK_ARG(o->z[j] && o->mean[j], "%s: dfhip_bn_stats_out with a null array", fn);
  return DFHIP_OK;
}

int dfhip_conv2d_bn_bwd_supported(int N, int C, int H, int W, int O, int k, int stride, int pad, int pair,
                                  int* dgrad_ok, int* wgrad_ok, int* stats_rows) {
  DF_CHECK_ARG(dgrad_ok && wgrad_ok && stats_rows, "conv2d_bn_bwd_supported: null result");
  *dgrad_ok = *wgrad_ok = *stats_rows = 0;
  int oh, ow;
  if (check_conv_args("conv2d_bn_bwd_supported", N, C, H, W, O, k, stride, pad, oh, ow) != DFHIP_OK) return DFHIP_OK;
  static int enabled = getenv("DFHIP_BN_BWD_FUSED") ? atoi(getenv("DFHIP_BN_BWD_FUSED")) : 1;
  if (!enabled) return DFHIP_OK;
  BnBwdIn bi = {};
  BnStatsOut so = {};
  so.n = 1;
  const int st = pair ? run_dgrad_ow_pair(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, C, H, W, O, &bi, &bi, &so, true)
                      : run_dgrad_ow(nullptr, nullptr, nullptr, nullptr, N, C, H, W, O, k, stride, pad, &bi, &so, true);
  // (the table of a pending backward is built from rows of O * K <= 768 floats; the rows are one per workgroup)
  *dgrad_ok = st == DFHIP_OK && O * 3 <= 768 && so.nrows <= kBnRowsMax;
  *stats_rows = *dgrad_ok ? so.nrows : 0;
  // the grouped direct body of the weight gradient: whole 32-channel tiles, planes of 2 .. 16 pixels a side (DF_ROWS_KINDS)
  const bool k3 = k == 3 && pad == 1, k1 = k == 1 && pad == 0 && stride == 2;
  *wgrad_ok = (k3 || k1) && C % 32 == 0 && O % 32 == 0 && N % 16 == 0 && H == W &&
              (W == 16 || W == 8 || W == 4 || (W == 2 && stride == 1)) && (stride == 1 || stride == 2) &&
              (stride == 1 || ow * 2 == W) && (H * W) % 4 == 0 && (oh * ow) % 4 == 0;
  return DFHIP_OK;
}

int dfhip_conv2d_dgrad_bn(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H, int W,
                          int O, int k, int stride, int pad, const dfhip_bn_bwd_in* in, dfhip_bn_stats_out* out) {
  DF_INIT_KEEP();
  DF_TOUCH(dy); DF_TOUCH(w); DF_TOUCH(add); DF_TOUCH(dx);
  int s = check_bn_in("conv2d_dgrad_bn", in);
  if (s != DFHIP_OK) return s;
  s = check_bn_out("conv2d_dgrad_bn", out);
  if (s != DFHIP_OK) return s;
  return conv2d_dgrad_impl(dy, w, add, dx, N, C, H, W, O, k, stride, pad, 0, in, out);
}

int dfhip_conv2d_dgrad_pair_bn(const float* dy, const float* w, const float* dy2, const float* w2, const float* add,
                               float* dx, int N, int C, int H, int W, int O, const dfhip_bn_bwd_in* in,
                               const dfhip_bn_bwd_in* in2, dfhip_bn_stats_out* out) {
  DF_INIT_KEEP();
  DF_TOUCH(dy); DF_TOUCH(w); DF_TOUCH(dy2); DF_TOUCH(w2); DF_TOUCH(add); DF_TOUCH(dx);
  DF_CHECK_ARG(dy && w && dy2 && w2 && dx, "conv2d_dgrad_pair_bn: null array");
  int oh, ow;
  int s = check_conv_args("conv2d_dgrad_pair_bn", N, C, H, W, O, 3, 2, 1, oh, ow);
  if (s != DFHIP_OK) return s;
  s = check_bn_in("conv2d_dgrad_pair_bn", in);
  if (s != DFHIP_OK) return s;
  s = check_bn_in("conv2d_dgrad_pair_bn", in2);
  if (s != DFHIP_OK) return s;
  s = check_bn_out("conv2d_dgrad_pair_bn", out);
  if (s != DFHIP_OK) return s;
  int done = 0;
  s = dgrad_pair_overwrite(dy, w, dy2, w2, add, dx, N, C, H, W, O, in, in2, out, &done);
  if (s != DFHIP_OK) return s;
  if (!done) { set_error("conv2d_dgrad_pair_bn: no kernel for this shape (query dfhip_conv2d_bn_bwd_supported first)"); return DFHIP_INVALID_ARGUMENT; }
  return DFHIP_OK;
}




// dx = conv2d_dgrad(dy, w) + add: the gradient accumulation of a tensor with two consumers
// (tensor.py:444 "parent.grad + contrib") done in the epilogue of the input-gradient kernel
int dfhip_conv2d_dgrad_add(const float* dy, const float* w, const float* add, float* dx, int N, int C,
                           int H, int W, int O, int k, int stride, int pad, int mode) {
  DF_INIT_KEEP();
  DF_TOUCH(dy);
  DF_TOUCH(w);
  DF_TOUCH(add);
  DF_TOUCH(dx);
  DF_CHECK_ARG(add, "conv2d_dgrad_add: null addend");
  return conv2d_dgrad_impl(dy, w, add, dx, N, C, H, W, O, k, stride, pad, mode);
}



// conv2d forward whose operand is the training-mode BatchNorm of x (no activation in between), x being the raw
// output of the previous convolution and `partials_in` the BatchNorm partials that convolution left behind
// (dfhip_conv2d_forward_stats).  Returns DFHIP_OK with *done = 1 when a kernel with that operand path took the
// shape; *done = 0 (nothing launched, nothing written) otherwise -- the caller then runs BatchNorm and the
// convolution separately.  By-products as dfhip_bn2d_forward_partials: save_mean / save_var (C), the running
// statistics, plus affine_out (C x 2: scale, shift) for dfhip_conv2d_wgrad_affine.
int dfhip_conv2d_forward_bn_input(const float* x, const float* partials_in, int rows_in, const float* gamma,
                                  const float* beta, float eps, float momentum, float* save_mean, float* save_var,
                                  float* running_mean, float* running_var, float* affine_out, const float* w,
                                  const float* bias, float* y, int N, int C, int H, int W, int O, int k, int stride,
                                  int pad, float* partials, int cap_rows, int* rows_out, int* done) {
  DF_INIT();
  int oh, ow;
  int s = check_conv_args("conv2d_forward_bn_input", N, C, H, W, O, k, stride, pad, oh, ow);
  if (s != DFHIP_OK) return s;
  DF_CHECK_ARG(x && w && y && partials_in && save_mean && save_var && affine_out && done,
               "conv2d_forward_bn_input: null array");
  *done = 0;
  if (rows_out) *rows_out = 0;
  if (rows_in < 1) return DFHIP_OK;     // (the documented contract for what the entry does not take: *done = 0, the caller writes conv(bn(x)))
  BnInput bn = {};
  bn.part = partials_in; bn.rows = rows_in; bn.gamma = gamma; bn.beta = beta; bn.eps = eps; bn.momentum = momentum;
  bn.save_mean = save_mean; bn.save_var = save_var; bn.running_mean = running_mean; bn.running_var = running_var;
  bn.affine = affine_out;
  double flops = 2.0 * N * oh * ow * (double)C * k * k * O;
  double bytes = 4.0 * ((double)N * C * H * W + (double)O * C * k * k + (double)N * O * oh * ow);
  // (shape test first: a call that launches nothing must not count as a launch of the family)
  const bool want = partials && rows_out;
  int rows = 0;
  if (H == 2 && W == 2) {
    // 2 x 2 images: the all-positions kernel, every wave builds the pairs of its own channels from the rows
    if (run_conv_img2_bn(x, w, bias, y, N, C, H, W, O, k, stride, pad, want ? partials : nullptr, cap_rows, &rows, bn, true) < 0)
      return DFHIP_OK;
    ProfScope prof("conv2d_fwd_igemm", flops, bytes);
    int st2 = run_conv_img2_bn(x, w, bias, y, N, C, H, W, O, k, stride, pad, want ? partials : nullptr, cap_rows, &rows, bn, false);
    if (st2 != DFHIP_OK) return st2 < 0 ? DFHIP_RUNTIME_ERROR : st2;
    if (want) *rows_out = rows;
    *done = 1;
    return DFHIP_OK;
  }
  if (run_conv_rowmma(x, w, bias, y, N, C, H, W, O, k, stride, pad, want ? partials : nullptr, cap_rows, &rows, &bn, true) < 0)
    return DFHIP_OK;
  // the (scale, shift) table is built by every workgroup of the convolution from the producer's rows (the register-operand
  // kernels: the rows are requested in front of the weights; the image-per-lane kernel of the 4 x 4 layers: every wave
  // for its own sixteen channels) -- a launch of its own for it cost 4.2 us + a kernel boundary (round 4)
  int st = DFHIP_OK;
  ProfScope prof("conv2d_fwd_igemm", flops, bytes);
  st = run_conv_rowmma(x, w, bias, y, N, C, H, W, O, k, stride, pad, want ? partials : nullptr, cap_rows, &rows, &bn);
  if (st < 0) {
    set_error("conv2d_forward_bn_input: the shape test and the kernel's own disagree (N %d C %d H %d W %d O %d)", N, C, H, W, O);
    return DFHIP_RUNTIME_ERROR;
  }
  if (st != DFHIP_OK) return st;
  if (want) *rows_out = rows;
  *done = 1;
  return DFHIP_OK;
}




}  // extern "C"
