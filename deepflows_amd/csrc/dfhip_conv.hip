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
                                                          float* __restrict__ stats, int O, int xo, BnInput bn) {
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
  const DfTile wg = xcd_tile(xo);               // (image tile, channel tile): XCD-aware order, dfhip_internal.h
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
struct SGroup {
  int n;
  int vb_end[kMaxGroup];
  SlabSum s[kMaxGroup];
  int stem_blocks;              // the first workgroups of the launch: the stem's one-pass backward, last step (dfhip_stem_combine.h)
  StemCombine stem;
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
template <int KS, int PAD, int STRIDE, int W, int OW, int UR, int NP, bool PAIR, bool BN>
__device__ __forceinline__ void wgrad_rows_direct(const GDesc& d, float* smem, const int bx, const int by,
                                                  const int bz) {
  static_assert(!PAIR || (KS == 3 && STRIDE == 2 && PAD == 1), "the shortcut rides on the 3x3 / stride-2 body");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int T = KS * KS, NW = kRowsWaves;
  constexpr int NXR = (UR - 1) * STRIDE + KS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = d.H, oh = d.oh;
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
  constexpr int NUSED = wg_used_rows<KS, STRIDE, UR>();
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
      const int xr = wg_nth_used_row<KS, STRIDE, UR>((kk - PA2 - PAZ) / PBR);
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
    un.rs = r * STRIDE - PAD;
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
  constexpr int NIT = NP * UR * KS * OW;                     // (pair, ur, i, s) iterations of a unit, <= KS MFMAs each
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
        for (int i = 0; i < KS; i++) {
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
          if (affine && (ur == 0 || i + STRIDE >= KS)) {     // first use of source row ur * STRIDE + i
#pragma unroll
            for (int cc = 0; cc < W; cc++) c.b[p][ur * STRIDE + i][cc] = fmaf(c.b[p][ur * STRIDE + i][cc], aff_sc, aff_sh);
          }
#pragma unroll
          for (int s = 0; s < OW; s++) {
            const int it = ((p * UR + ur) * KS + i) * OW + s;
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

template <int KS, int PAD, int STRIDE, int W, int OW, int UR, int NP>
__device__ __forceinline__ void wgrad_direct_any(const GDesc& d, float* smem, const int bx, const int by, const int bz) {
  const bool bn = d.bnz != nullptr || d.bnz2 != nullptr;
  if constexpr (KS == 3 && STRIDE == 2) {
    if (d.dy2) {
      if (bn) wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, true, true>(d, smem, bx, by, bz);
      else wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, true, false>(d, smem, bx, by, bz);
      return;
    }
  }
  if constexpr (W <= 16) {       // (the BatchNorm operand exists for the planes of the CIFAR models' residual stages)
    if (bn) { wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, false, true>(d, smem, bx, by, bz); return; }
  }
  wgrad_rows_direct<KS, PAD, STRIDE, W, OW, UR, NP, false, false>(d, smem, bx, by, bz);
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
__global__ __launch_bounds__(64 * kRowsWaves) void k_wgrad_group(WGroup a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
#define DF_CASE(ID, KSV, PADV, SV, WV, OWV, URV, NPV)                                              \
  case ID:                                                                                         \
    if (d.direct) wgrad_direct_any<KSV, PADV, SV, WV, OWV, URV, NPV>(d, smem, bx, by, bz);        \
    else wgrad_rows_body<KSV, PADV, SV, WV, OWV, URV, true>(g, smem, bx, by, bz, d.ps, d.CT, d.OT); \
    break;
    DF_ROWS_KINDS(DF_CASE)
#undef DF_CASE
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
          float* dst = d.dw + ((int64_t)(o + e) * d.C + c) * TT;
#pragma unroll
          for (int t = 0; t < TT; t++)
            dst[t] = e == 0 ? acc[t].x : (e == 1 ? acc[t].y : (e == 2 ? acc[t].z : acc[t].w));
        }
    }
  }
}

__global__ __launch_bounds__(256) void k_wgrad_group_sum(SGroup a) {
  __shared__ float4 part[256 * 9];
  // (the stem's workgroups are a chain of dependent round trips over its 256 slabs: dispatched first, they run beside the
  // streaming ones instead of behind them)
  if ((int)blockIdx.x < a.stem_blocks) {             // (uniform per workgroup)
    stem_combine_body(a.stem, (int)blockIdx.x);
    return;
  }
  const int b = (int)blockIdx.x - a.stem_blocks;
  int li = 0;
  while (li + 1 < a.n && b >= a.vb_end[li]) li++;
  const SlabSum& d = a.s[li];
  const int vb = b - (li ? a.vb_end[li - 1] : 0);
  if (d.T == 9) wgrad_group_sum_body<9>(d, vb, part);
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

unsigned magic_u32(unsigned d) { return d <= 1 ? 0u : (unsigned)((((uint64_t)1 << 32) + d - 1) / d); }

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
  static int k5_on = getenv("DFHIP_IGEMM_K5") ? atoi(getenv("DFHIP_IGEMM_K5")) : 1;
  if (g.CS % 8 != 0 || !(g.k == 1 || g.k == 3 || (g.k == 5 && k5_on))) return -1;
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
    static int jobmul = getenv("DFHIP_JOBMUL") ? atoi(getenv("DFHIP_JOBMUL")) : 1;
    if (jobs < (int64_t)num_cus() * jobmul && pt > 1) continue;
    static size_t lds_cap_single = (size_t)(getenv("DFHIP_LDS_SINGLE") ? atoi(getenv("DFHIP_LDS_SINGLE")) : 150) * 1024;
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
    static int maxchunks = getenv("DFHIP_MAXCHUNKS") ? atoi(getenv("DFHIP_MAXCHUNKS")) : 2;
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
  static int vec_store = getenv("DFHIP_VEC_STORE") ? atoi(getenv("DFHIP_VEC_STORE")) : 1;
  g.vec_store = (vec_store && (g.BR * g.BC) % 4 == 0 && (g.HD * g.WD) % 4 == 0) ? 1 : 0;
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
    static int k5_spec = getenv("DFHIP_IGEMM_K5_SPEC") ? atoi(getenv("DFHIP_IGEMM_K5_SPEC")) : 1;
    if (k5_spec) st = DF_V4(5);
    else st = g.mode == 0 ? launch_igemm_v4n<5, 0, 0>(g, grid, lds)
                          : (g.mode == 1 ? launch_igemm_v4n<5, 1, 0>(g, grid, lds) : launch_igemm_v4n<5, 2, 0>(g, grid, lds));
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
  static int enabled = getenv("DFHIP_IMG2_LEAN") ? atoi(getenv("DFHIP_IMG2_LEAN")) : 1;
  if (!enabled || g.mode != 0 || g.k != 3 || g.so != 1 || g.sd != 1 || g.u0 != -1) return -1;
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
  const int xo = xcd_arrangement(rows4, O / 16, 4.0 * g.N * g.CS * 4, 4.0 * O * g.CS * 9);
  if (g.CS == 128) DF_LAUNCH(k_conv3x3_img2_all<128>, grid4, dim3(512), lds4, stream(), g.src, g.wgt, bias, y, partials, O, xo, BnInput{});
  else DF_LAUNCH(k_conv3x3_img2_all<256>, grid4, dim3(512), lds4, stream(), g.src, g.wgt, bias, y, partials, O, xo, BnInput{});
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// the same layer with the training-mode BatchNorm of its input applied on the operand loads, the table built by every wave
// for its own channels (k_conv3x3_img2_all<.., BNC>); -1 when the shape is not one it takes
int run_conv_img2_bn(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int O, int k,
                     int stride, int pad, float* partials, int cap_rows, int* rows_out, const BnInput& bn, bool dry_run) {
  static int enabled = getenv("DFHIP_IMG2_LEAN") ? atoi(getenv("DFHIP_IMG2_LEAN")) : 1;
  static int bn_on = getenv("DFHIP_IMG2_BN_INPUT") ? atoi(getenv("DFHIP_IMG2_BN_INPUT")) : 1;
  if (!enabled || !bn_on || k != 3 || stride != 1 || pad != 1 || H != 2 || W != 2 || (N % 32) != 0) return -1;
  if (!(C == 128 || C == 256) || O % 16 != 0 || !bn.part || bn.rows < 1 || !bn.affine) return -1;
  const int rows4 = N / 16;
  if (partials && rows4 > cap_rows) return -1;
  if (rows_out) *rows_out = partials ? rows4 : 0;
  if (dry_run) return DFHIP_OK;
  dim3 grid4((unsigned)rows4, (unsigned)(O / 16));
  const size_t lds4 = std::max((size_t)8 * (72 * 17 + 16 * 9 * 4), (size_t)7 * 4 * 64 * 4) * sizeof(float);
  const int xo = xcd_arrangement(rows4, O / 16, 4.0 * N * C * 4, 4.0 * O * C * 9);
  if (C == 128) DF_LAUNCH((k_conv3x3_img2_all<128, false, true>), grid4, dim3(512), lds4, stream(), x, w, bias, y, partials, O, xo, bn);
  else DF_LAUNCH((k_conv3x3_img2_all<256, false, true>), grid4, dim3(512), lds4, stream(), x, w, bias, y, partials, O, xo, bn);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// shortcut convolution (1x1 / stride 2 / pad 0, 32-256 source channels); -1 when not eligible
int run_conv1x1s2(const Geo& g, int O, float* y, const float* bias, float* partials, int cap_rows, int* rows_out) {
  static int enabled = getenv("DFHIP_SHORTCUT_LEAN") ? atoi(getenv("DFHIP_SHORTCUT_LEAN")) : 1;
  if (!enabled || g.mode != 0 || g.k != 1 || g.so != 2 || g.sd != 1 || g.u0 != 0) return -1;
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
  static int dtmax = getenv("DFHIP_SHORTCUT_DT") ? atoi(getenv("DFHIP_SHORTCUT_DT")) : 8;
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
  static int enabled = getenv("DFHIP_STEM_ROWS") ? atoi(getenv("DFHIP_STEM_ROWS")) : 1;
  if (!enabled || g.mode != 0 || g.k != 3 || g.so != 1 || g.sd != 1 || g.u0 != -1) return -1;
  if (g.CS < 1 || g.CS > 4 || g.WS != 32 || g.WD != 32 || g.HS != g.HD) return -1;
  if ((int64_t)g.N * df_ceil_div(g.HS, 8) >= ((int64_t)1 << 31)) return -1;        // band index in an int
  // a workgroup per band of rb rows of an image (the whole image when there are enough images to fill the chip);
  // more bands than partial rows / twice the CUs: the workgroups walk them
  static int rb_env = getenv("DFHIP_STEM_BAND") ? atoi(getenv("DFHIP_STEM_BAND")) : 0;
  int rb = rb_env > 0 ? std::min(rb_env, 32) : 32;
  if (rb_env <= 0) {
    while (rb > 8 && (int64_t)g.N * df_ceil_div(g.HS, rb) < num_cus()) rb /= 2;
  }
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
  static int enabled = getenv("DFHIP_WGRAD_STEM3") ? atoi(getenv("DFHIP_WGRAD_STEM3")) : 1;
  if (!enabled || g.k != 3 || g.stride != 1 || g.pad != 1 || g.W != 32 || g.ow != 32 || g.oh != g.H) return -1;
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
  // halve the dY operand reads per MFMA; DFHIP_WGRAD_KT5=0 restores two)
  static int kt5 = getenv("DFHIP_WGRAD_KT5") ? atoi(getenv("DFHIP_WGRAD_KT5")) : 1;
  const int KT = (g.k == 1 || KW <= 32) ? 1 : (g.O <= 32 ? 5 : ((kt5 && g.k == 5 && KW >= 320) ? 5 : 2));
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
  static int defer = getenv("DFHIP_WGRAD_DEFER") ? atoi(getenv("DFHIP_WGRAD_DEFER")) : 1;
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
  if (!defer) return launch_slab_sum(mine);
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
  int W, ow, k, stride, pad;     // the rest of the call's arguments (a small group is re-planned as stand-alone launches)
};
std::vector<GQueued> g_queue;
bool g_group_on = false;
struct StemPending { bool active = false; StemCombine c; int blocks = 0; } g_stem_pending;
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

// Workgroup length of a grouped layer (operands from global memory).  The body runs one workgroup per CU (470 registers), so
// the launch ends when the CU with the longest queue does: a layer is cut so that a workgroup is about `target` cycles of
// its executed MFMAs (64 issue cycles each, four waves) -- long enough that the ~12 000 cycles around the MFMA stream (first
// operands, cross-wave sum, slab stores) stay small, short enough that the longest workgroup is a fraction of the launch
// (with the 256 / (tiles * 16) rule the 64-channel layers ran 400 000-cycle workgroups: the whole launch).
static double replan_direct(GQueued& e, int target) {
  GDesc& d = e.d;
  int UR = 1, NP = 1;
  switch (d.kind) {
#define DF_UR(ID, KSV, PADV, SV, WV, OWV, URV, NPV) case ID: UR = URV; NP = NPV; break;
    DF_ROWS_KINDS(DF_UR)
#undef DF_UR
    default: break;
  }
  // cycles of a wave per unit, measured with the in-kernel counters (-DDFHIP_STAMPS, DFHIP_CONV_STAMP=1; batch 256): the
  // MFMA issue slots of the unit plus what its operand loads and hand-over cost -- units of few MFMAs (1x1 kernels, 2 x 2
  // planes) are bound by the latency of their loads, not by their MFMA count
  static const double kUnitCycles[13] = {9260, 8720, 7640, 5560, 4780, 4680, 4490, 9400, 7200, 2840, 19000, 9800, 9400};
  const int upr = d.oh / UR, U = (d.N / (2 * NP)) * upr;
  const double unit_cycles = d.kind >= 0 && d.kind < 13 ? kUnitCycles[d.kind] : 5000.0;
  if (target > 0) {
    const double per_wave = std::max(1.0, (target - 14000.0) / unit_cycles);
    const int want = 4 * (int)std::max(1.0, std::floor(per_wave + 0.5));
    const int ps = std::max(1, (int)std::floor((double)U / want + 0.5));
    d.units = std::max(4, ((int)df_ceil_div(U, ps) + 3) & ~3);
    d.ps = (int)df_ceil_div(U, d.units);
  }
  return d.units / 4 * unit_cycles + 14000.0;
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
  static int target = getenv("DFHIP_WGRAD_GROUP_CYCLES") ? atoi(getenv("DFHIP_WGRAD_GROUP_CYCLES")) : 0;
  for (auto& e : q) {
    e.work = e.d.direct ? replan_direct(e, target) : (double)e.d.N * e.d.oh * e.d.oh * e.T / e.d.ps * 16.0;
    e.rides_on = -1;
    e.d.dy2 = nullptr; e.d.slabs2 = nullptr;
    e.d.bnz2 = nullptr; e.d.bntab2 = nullptr;
  }
  // longest workgroups first (they are dispatched in grid order): the launch ends on short ones
  std::stable_sort(q.begin(), q.end(), [](const GQueued& a, const GQueued& b) { return a.work > b.work; });
  // a block's 1x1 / stride-2 shortcut reads the same tensor as its 3x3 / stride-2 convolution: the shortcut's weight gradient
  // rides on that layer's workgroups (wgrad_rows_direct<PAIR>) -- it keeps its slabs (one per workgroup of the host layer) and
  // its entry in the slab sum, and has no workgroups of its own
  static int pair_on = getenv("DFHIP_WGRAD_PAIR") ? atoi(getenv("DFHIP_WGRAD_PAIR")) : 1;
  if (pair_on && q.size() <= (size_t)kMaxGroup) {
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
      ProfScope prof("conv2d_wgrad_grouped", flops, bytes);
      DF_LAUNCH(k_wgrad_group, dim3((unsigned)wgs), dim3(64 * kRowsWaves), lds, gst, wg);
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
      ProfScope prof("wgrad_slab_sum", 0.0, sbytes);
      DF_LAUNCH(k_wgrad_group_sum, dim3((unsigned)(vbs + sg.stem_blocks)), dim3(256), 0, gst, sg);
      DF_LAUNCH_CHECK();
    }
  }
  return DFHIP_OK;
}

int side_flush() {
  if (g_queue.empty()) return DFHIP_OK;
  if (!g_side) {
    DF_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
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
int run_wgrad_rows(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int O, int oh, int ow,
                   int k, int stride, int pad, const float* aff, const float* bnz, const float* bntab) {
  static int enabled = getenv("DFHIP_WGRAD_ROWS") ? atoi(getenv("DFHIP_WGRAD_ROWS")) : 1;
  if (!enabled) return -1;
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
  static int gpb1 = getenv("DFHIP_WGRAD_GPB") ? atoi(getenv("DFHIP_WGRAD_GPB")) : 2;   // groups per block aimed at
  static int gpbg = getenv("DFHIP_WGRAD_GROUP_GPB") ? atoi(getenv("DFHIP_WGRAD_GROUP_GPB")) : 2;
  const int gpb = grouped ? gpbg : gpb1;
  // output rows per unit (must match the instantiations)
  const int UR = k == 1 ? 1 : (stride == 1 ? (W >= 16 ? 1 : (W == 8 ? 2 : (W == 4 ? 4 : 2))) : (W >= 16 ? 1 : 2));
  const double per_block = (double)(N / 2) * (oh / UR) / ps0;
  const double target = std::max((double)kRowsWaves, per_block / std::max(1, gpb));
  static int gcap = getenv("DFHIP_WGRAD_GROUP_LDS") ? atoi(getenv("DFHIP_WGRAD_GROUP_LDS")) : 160;
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
  static int direct_allowed = getenv("DFHIP_WGRAD_DIRECT") ? atoi(getenv("DFHIP_WGRAD_DIRECT")) : 1;
  if (!staged_plan && !(grouped && direct_allowed && C % 32 == 0 && O % 32 == 0)) return -1;
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
    static int direct_on = getenv("DFHIP_WGRAD_DIRECT") ? atoi(getenv("DFHIP_WGRAD_DIRECT")) : 1;
    const bool direct = direct_on && C % 32 == 0 && O % 32 == 0 && (N / 2) % NPk == 0;
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
  static int enabled = getenv("DFHIP_WGRAD_TAP") ? atoi(getenv("DFHIP_WGRAD_TAP")) : 1;
  if (!enabled) return -1;
  if (!(k == 1 || k == 3) || C < 16 || (N & 1)) return -1;
  if ((H * W) % 4 != 0 || (oh * ow) % 4 != 0) return -1;
  if (pad >= k) return -1;
  const int CT = (int)df_ceil_div(C, 32), OT = (int)df_ceil_div(O, 32);
  const int64_t steps_total = (int64_t)(N / 2) * oh * ow;
  const int ps0 = std::max(1, num_cus() / (CT * OT));
  static int gpb = getenv("DFHIP_WGRAD_GPB") ? atoi(getenv("DFHIP_WGRAD_GPB")) : 2;   // groups per block aimed at
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
  static int dense_min = getenv("DFHIP_WGRAD_DENSE_MIN") ? atoi(getenv("DFHIP_WGRAD_DENSE_MIN")) : 8;
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
int flush_deferred() {
  if (!g_queue.empty()) {
    int st = group_flush();
    if (st != DFHIP_OK) return st;
  }
  if (g_stem_pending.active) {                      // no grouped slab sum took it along
    g_stem_pending.active = false;
    int st = stem_combine_launch(g_stem_pending.c, g_stem_pending.blocks);
    if (st != DFHIP_OK) return st;
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
bool deferred_touches(const void* p) {
  if (!p) return false;
  if (g_stem_pending.active) {
    const StemCombine& c = g_stem_pending.c;
    const float* q = (const float*)p;
    auto in = [&](const float* lo, int64_t n) { return lo && q >= lo && q < lo + n; };
    if (in(c.dw, (int64_t)c.O * c.KK) || in(c.dgamma, c.O) || in(c.dbeta, c.O) || in(c.dbias, c.O) || p == c.slabs) return true;
  }
  for (const auto& sp : g_side_pending)              // a gradient in flight on the side stream: join first
    for (const auto& r : sp.dws)
      if ((const float*)p >= r.first && (const float*)p < r.second) return true;
  const float* q = (const float*)p;
  for (const auto& e : g_queue)
    if (q >= e.dw && q < e.dw + (int64_t)e.d.O * e.d.C * e.T) return true;
  if (!g_def.active) return false;
  const float* lo = g_def.a.dw;
  return q >= lo && q < lo + (int64_t)g_def.a.O * g_def.a.C * g_def.a.T;
}
int deferred_scratch_slot() { return g_def.active ? g_def.slot : -1; }
bool group_collecting() { return g_group_on && !g_queue.empty() && !g_group_stream && g_side_pending.empty(); }
bool stem_combine_pending() { return g_stem_pending.active; }
bool stem_combine_defer(const StemCombine& c, int blocks) {
  if (!group_collecting() || g_stem_pending.active) return false;
  g_stem_pending.c = c;
  g_stem_pending.blocks = blocks;
  g_stem_pending.active = true;
  return true;
}
}  // namespace dfhip

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
    static int img2dg = getenv("DFHIP_DGRAD_IMG2_ACC") ? atoi(getenv("DFHIP_DGRAD_IMG2_ACC")) : 1;
    if (img2dg && k == 3 && stride == 1 && pad == 1 && H == 2 && W == 2 && N % 16 == 0 && C % 16 == 0 &&
        (O == 128 || O == 256)) {
      dim3 grid4((unsigned)(N / 16), (unsigned)(C / 16));
      const size_t lds4 = std::max((size_t)8 * (72 * 17 + 16 * 9 * 4), (size_t)7 * 4 * 64 * 4) * sizeof(float);
      const int xo = xcd_arrangement(N / 16, C / 16, 4.0 * N * O * 4, 4.0 * O * C * 9);
      if (O == 128) DF_LAUNCH((k_conv3x3_img2_all<128, true>), grid4, dim3(512), lds4, stream(), dy, w, add, dx, (float*)nullptr, C, xo, BnInput{});
      else DF_LAUNCH((k_conv3x3_img2_all<256, true>), grid4, dim3(512), lds4, stream(), dy, w, add, dx, (float*)nullptr, C, xo, BnInput{});
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
  const double bytes = 4.0 * ((double)N * C * H * W * (add ? 2 : 1) + (double)O * C * 10 + 2.0 * N * O * oh * ow);
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
  static int pair_on = getenv("DFHIP_DGRAD_PAIR") ? atoi(getenv("DFHIP_DGRAD_PAIR")) : 1;
  if (pair_on && mode == 1) {
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
static int check_bn_in(const char* fn, const BnBwdIn* b) {
  if (!b) return DFHIP_OK;
  DF_CHECK_ARG(b->z && b->rows && b->mean && b->var, "%s: dfhip_bn_bwd_in with a null array", fn);
  DF_CHECK_ARG(b->nrows >= 1 && b->nrows <= kBnRowsMax && b->K >= 2 && b->K <= 3 && b->kidx >= 1 && b->kidx < b->K,
               "%s: dfhip_bn_bwd_in rows %d (1..%d), K %d, kidx %d", fn, b->nrows, kBnRowsMax, b->K, b->kidx);
  return DFHIP_OK;
}
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
  static int direct_on = getenv("DFHIP_WGRAD_DIRECT") ? atoi(getenv("DFHIP_WGRAD_DIRECT")) : 1;
  *wgrad_ok = direct_on && (k3 || k1) && C % 32 == 0 && O % 32 == 0 && N % 16 == 0 && H == W &&
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
  static int enabled = getenv("DFHIP_BN_INPUT") ? atoi(getenv("DFHIP_BN_INPUT")) : 1;
  if (!enabled) return DFHIP_OK;
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
  // the (scale, shift) table: built by every workgroup of the convolution from the producer's rows (the register-operand
  // kernels: the rows are requested in front of the weights; the image-per-lane kernel of the 4 x 4 layers: every wave
  // for its own sixteen channels), or by a launch of its own (DFHIP_BN_INPUT_COMBINE=0)
  static int combine_env = getenv("DFHIP_BN_INPUT_COMBINE") ? atoi(getenv("DFHIP_BN_INPUT_COMBINE")) : 1;
  bn.combine = combine_env && rows_in >= 1 ? 1 : 0;
  int st = DFHIP_OK;
  if (!bn.combine) {
    st = bn_finalize_partials(bn, C);
    if (st < 0) return DFHIP_OK;
    if (st != DFHIP_OK) return st;
  }
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
  static int enabled = getenv("DFHIP_STEM_FUSED_BWD") ? atoi(getenv("DFHIP_STEM_FUSED_BWD")) : 1;
  if (!enabled || O > 32 || (oh & 1) || (ow & 7) || stride != 1 || C > 4) return DFHIP_OK;
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
int dfhip_wgrad_group_begin(void) {
  DF_INIT();
  g_group_on = true;
  return DFHIP_OK;
}

int dfhip_wgrad_group_end(void) {
  DF_INIT_KEEP();
  g_group_on = false;
  return flush_deferred();
}

}  // extern "C"
