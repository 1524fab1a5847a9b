// Conv2d forward, 3x3 / stride 2 / pad 1 with 32-128 source channels TOGETHER WITH the 1x1 / stride 2 / pad 0
// shortcut convolution of the same residual block (test/ResNet_CIFAR10_cuda.py:31-47: conv1 and downsample[0] both
// read the block input; reference path nn/functional.py:316-344 twice): the shortcut reads exactly the pixels of the
// 3x3 convolution's centre tap -- x[2r][2s] -- so it is a TENTH tap of the same register-operand loop with a weight
// tensor and an accumulator of its own, not a second launch.
//
// v_mfma_f32_16x16x4_f32 as in dfhip_conv_rows.hip: D[16 output channels][16 pixels] += A[16][4 source channels] .
// B[4][16].  A tile is 16 consecutive OUTPUT pixels of the flattened (n, r, s) space (two rows of an 8-wide output,
// a whole 4 x 4 output image, four 2 x 2 output images).  Lane (kk, n) of the B operand of tap (i, j) holds
// x[img][4p + kk][2r + i - 1][2s + j - 1]: with a stride of two nothing is shared between taps by a lane shift, so
// every tap is a buffer load of its own (offsets resolved once per lane; a tap that falls into the padding gets an
// offset past the buffer and reads as zero) -- nine loads per channel group, NO vector instruction in the loop
// (beside fp32 MFMAs every VALU instruction costs matrix-pipe time).  The weight operand of the workgroup's output
// channels goes to LDS in MFMA lane order, ten operands per channel group (nine taps + the shortcut), staged a
// chunk ahead as in the stride-1 kernel.  A wave owns one pixel tile and OWW output-channel tiles; BatchNorm
// partials of both outputs (count, mean, centred second moment per workgroup and channel).
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "dfhip_internal.h"

using namespace dfhip;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16b(float v) {
  v += dpp_mov2<0x128>(v);   // row_ror:8
  v += dpp_mov2<0x124>(v);   // row_ror:4
  v += dpp_mov2<0x122>(v);   // row_ror:2
  v += dpp_mov2<0x121>(v);   // row_ror:1
  return v;
}
#ifdef DFHIP_S2_STAMPS
__device__ unsigned long long g_s2_stamps[1024 * 8 * 8];
#define S2_STAMP(i)                                                                                     \
  do {                                                                                                  \
    if (lane == 0) g_s2_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + (i)] =          \
        (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();                     \
  } while (0)
#else
#define S2_STAMP(i) do { } while (0)
#endif
constexpr int kBlk2 = 260;   // LDS pitch (floats) of a block of four operands x 64 lanes (+4: bank spread of the commits)

// C: source channels; OWW: output-channel tiles (16 channels) per wave; TS: pixel tiles per workgroup; OWG: groups of
// OWW channel tiles per workgroup (a workgroup is TS x OWG waves); CG: channel groups per weight chunk (even); XA:
// chunks of pixel operands requested ahead (the loads of a chunk must be back before its MFMAs: with one chunk of
// read-ahead a chunk's ~1 300 cycles of MFMAs did not cover the memory latency and the loop ran at a third of the pipe rate).
// lwo: log2 of the OUTPUT width (3, 2, 1); H: input height = width = 2 << lwo.
// KS: waves that share a tile and split every chunk's channel groups between them (1 or 2; their accumulators are
// summed through LDS at the end): two waves per SIMD where the problem has only 1 024 tile x channel-tile units, and
// each wave requests only its own groups' pixels.
template <int C, int OWW, int TS, int OWG, int CG, int XA, int KS>
__global__ __launch_bounds__(64 * TS * OWG * KS) void k_conv3x3s2_pair(const float* __restrict__ x,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ wd,
                                                                 float* __restrict__ y, float* __restrict__ yd,
                                                                 float* __restrict__ partials,
                                                                 float* __restrict__ partials_d, int lwo, int O) {
  constexpr int NWAVE = TS * OWG * KS, NTHR = 64 * NWAVE;
  constexpr int CGW = CG / KS;                 // channel groups of a chunk per wave
  static_assert(CG % KS == 0 && (CGW * 10) % 4 == 0, "a wave's share of a chunk is whole operand blocks");
  constexpr int OT = OWG * OWW;                // output-channel tiles of the workgroup
  constexpr int NG = C / 4;                    // channel groups (one MFMA k-step per tap each)
  constexpr int TAPS = 10;                     // nine taps + the shortcut
  constexpr int NCH = NG / CG;                 // weight chunks
  constexpr int NBLK = NG * TAPS / 4;          // operand blocks per output-channel tile
  constexpr int CBLK = CG * TAPS / 4;          // ... per chunk
  static_assert(NG % CG == 0 && (CG * TAPS) % 4 == 0, "chunks are whole operand blocks");
  extern __shared__ __attribute__((aligned(16))) float Wl[];   // [OT][NBLK blocks][64 lanes][4] at pitch kBlk2
  __shared__ float sred[2][NWAVE][16 * OWW][2];
  const int WO = 1 << lwo, HWO = WO << lwo;    // output plane
  const int W = WO << 1, HW = HWO << 2;        // input plane (H = W)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const int ts = wave % TS, og = (wave / TS) % OWG, ksp = wave / (TS * OWG);
  const int ot0 = og * OWW;                    // first channel tile of this wave within the workgroup
  const int o_wg = blockIdx.y * (OT * 16);     // first output channel of the workgroup
  const int T = blockIdx.x * TS + ts;          // this wave's pixel tile
  S2_STAMP(6);
  S2_STAMP(0);

  // ---- weight chunk staging ----
  // main: OT x 16 rows (output channels) of 36 * CG contiguous floats per chunk; shortcut: OT x 16 rows of 4 * CG.
  // thread -> 16-byte pieces; LDS float index of every element resolved once (operand f = p * 10 + t of lane
  // (kk, m) lives at block f / 4, slot (kk * 16 + m) * 4 + f % 4)
  constexpr int ROW4 = 9 * CG, TOT4 = OT * 16 * ROW4, PER = (TOT4 + NTHR - 1) / NTHR;
  constexpr int ROWD4 = CG, TOTD4 = OT * 16 * ROWD4, PERD = (TOTD4 + NTHR - 1) / NTHR;
  const float* wsrc[PER];
  const float* dsrc[PERD];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int q = tid + NTHR * u;
    const int qq = q < TOT4 ? q : q % TOT4;    // (threads past the end re-do a piece: every load keeps a consumer)
    const int tl = qq / (16 * ROW4), r = qq - tl * (16 * ROW4);
    const int m = r / ROW4, e4 = r - m * ROW4;
    wsrc[u] = w + ((int64_t)(o_wg + tl * 16 + m) * C * 9) + 4 * e4;
  }
#pragma unroll
  for (int u = 0; u < PERD; u++) {
    const int q = tid + NTHR * u;
    const int qq = q < TOTD4 ? q : q % TOTD4;
    const int tl = qq / (16 * ROWD4), r = qq - tl * (16 * ROWD4);
    const int m = r / ROWD4, e4 = r - m * ROWD4;
    dsrc[u] = wd + ((int64_t)(o_wg + tl * 16 + m) * C) + 4 * e4;
  }
  f32x4 wv[2][PER], dv[2][PERD];
#define S2_WLOAD(chunk)                                                                                   \
  do {                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < PER; u++)                                                       \
      wv[(chunk) & 1][u] = *reinterpret_cast<const f32x4*>(wsrc[u] + (chunk) * (36 * CG));                \
    _Pragma("unroll") for (int u = 0; u < PERD; u++)                                                      \
      dv[(chunk) & 1][u] = *reinterpret_cast<const f32x4*>(dsrc[u] + (chunk) * (4 * CG));                 \
  } while (0)
#define S2_WCOMMIT(chunk)                                                                                 \
  do {                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < PER; u++) {                                                     \
      _Pragma("unroll") for (int e = 0; e < 4; e++)                                                       \
        Wl[woff[u][e] + (chunk) * (CBLK * kBlk2)] = wv[(chunk) & 1][u][e];                                \
    }                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < PERD; u++) {                                                    \
      _Pragma("unroll") for (int e = 0; e < 4; e++)                                                       \
        Wl[doff[u][e] + (chunk) * (CBLK * kBlk2)] = dv[(chunk) & 1][u][e];                                \
    }                                                                                                     \
  } while (0)
  S2_WLOAD(0);
  if (NCH > 1) S2_WLOAD(1);
  __builtin_amdgcn_sched_barrier(0);           // the weight loads stay in front of the pixel loads

  // ---- per-lane source offsets of the nine taps (bytes from x, channel group 0) ----
  constexpr unsigned kOut = 0x80000000u;
  const int P = T * 16 + n;                    // flat output pixel (n, r, s)
  const int img = P / HWO, rem = P - img * HWO;
  const int r = rem >> lwo, s = rem & (WO - 1);
  // per source row i: the taps j = 1, 2 are the adjacent pixels (2s, 2s + 1) -- ONE 8-byte load -- and j = 0 is pixel
  // 2s - 1 (the address unit takes ~24 cycles per wave load of this kernel's shape whatever its width: nine 4-byte
  // loads per channel group kept it busier than the matrix cores)
  unsigned xo2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int yy = 2 * r + i - 1;
    const bool rowok = yy >= 0 && yy < W;
    xo2[i] = rowok ? (unsigned)((((int64_t)img * C + kk) * W + yy) * W + 2 * s) * 4u : kOut;
  }
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
  float raw[NG / KS][9];                       // this wave's groups: chunk ch holds raw[ch * CGW ..]
#define S2_XLOAD(chunk)                                                                                   \
  do {                                                                                                    \
    _Pragma("unroll") for (int g_ = 0; g_ < CGW; g_++) {                                                  \
      const int q_ = (chunk) * CGW + g_;                    /* slot in raw */                             \
      const int p_ = (chunk) * CG + ksp * CGW + g_;         /* channel group */                           \
      _Pragma("unroll") for (int i_ = 0; i_ < 3; i_++) {                                                  \
        const f32x2 t2_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, (int)xo2[i_], p_ * (16 * HW), 0)); \
        raw[q_][i_ * 3 + 1] = t2_[0]; raw[q_][i_ * 3 + 2] = t2_[1];                                       \
        raw[q_][i_ * 3] = 0.0f;          /* pixel 2s - 1 is the neighbouring lane's 2(s-1) + 1: a lane shift below */ \
      }                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                     \
  } while (0)
#pragma unroll
  for (int c0 = 0; c0 < XA && c0 < NCH; c0++) S2_XLOAD(c0);
  S2_STAMP(1);
  // LDS positions of the staged weights: resolved while the first loads are in flight
  int woff[PER][4], doff[PERD][4];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int q = tid + NTHR * u;
    const int qq = q < TOT4 ? q : q % TOT4;
    const int tl = qq / (16 * ROW4), r = qq - tl * (16 * ROW4);
    const int m = r / ROW4, e4 = r - m * ROW4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int idx = 4 * e4 + e;
      const int c = idx / 9, t = idx - 9 * c;
      const int f = (c >> 2) * TAPS + t;
      woff[u][e] = tl * (NBLK * kBlk2) + (f >> 2) * kBlk2 + (c & 3) * 64 + m * 4 + (f & 3);
    }
  }
#pragma unroll
  for (int u = 0; u < PERD; u++) {
    const int q = tid + NTHR * u;
    const int qq = q < TOTD4 ? q : q % TOTD4;
    const int tl = qq / (16 * ROWD4), r = qq - tl * (16 * ROWD4);
    const int m = r / ROWD4, e4 = r - m * ROWD4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = 4 * e4 + e;                // channel within the chunk
      const int f = (c >> 2) * TAPS + 9;
      doff[u][e] = tl * (NBLK * kBlk2) + (f >> 2) * kBlk2 + (c & 3) * 64 + m * 4 + (f & 3);
    }
  }
  S2_WCOMMIT(0);
  S2_STAMP(2);
  __syncthreads();
  S2_STAMP(3);

  f32x4 acc[OWW][4], accd[OWW];
#pragma unroll
  for (int a = 0; a < OWW; a++) {
#pragma unroll
    for (int c = 0; c < 4; c++) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accd[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  constexpr int WBLK = CBLK / KS;              // operand blocks of a chunk this wave reads
  const f32x4* Wl4 = reinterpret_cast<const f32x4*>(Wl + ot0 * (NBLK * kBlk2)) + ksp * WBLK * (kBlk2 / 4) + lane;
  f32x4 a4[2][OWW][WBLK];
#pragma unroll
  for (int a = 0; a < OWW; a++)
#pragma unroll
    for (int q = 0; q < WBLK; q++) a4[0][a][q] = Wl4[a * (NBLK * kBlk2 / 4) + q * (kBlk2 / 4)];
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    // chunk ch + 1 (requested a chunk ago) is committed now and made visible by a barrier in the middle of this
    // chunk's MFMAs; chunk ch + 2 and the pixels of chunk ch + 1 are requested
    if (ch + 1 < NCH) S2_WCOMMIT(ch + 1);
    if (ch + 2 < NCH) S2_WLOAD(ch + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (ch + XA < NCH) S2_XLOAD(ch + XA);
#pragma unroll
    for (int g = 0; g < CGW; g++) {
      const int p = ch * CGW + g;                // slot in raw
      if (g == CGW / 2 && ch + 1 < NCH) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < OWW; a++)
#pragma unroll
          for (int q = 0; q < WBLK; q++)
            a4[(ch + 1) & 1][a][q] = Wl4[a * (NBLK * kBlk2 / 4) + ((ch + 1) * CBLK + q) * (kBlk2 / 4)];
      }
      // tap column 0 = the left neighbour's column 2 (one DPP move + one select per source row; the address unit
      // was the busier side: six wave loads per channel group against twenty MFMAs -- a 4-byte load of pixel 2s - 1 per
      // source row instead: +1 us)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float sh = dpp_mov2<0x111>(raw[p][i * 3 + 2]);    // row_shr:1, zero into lane 0 of the 16
        raw[p][i * 3] = s > 0 ? sh : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int f = g * TAPS + t;
#pragma unroll
        for (int a = 0; a < OWW; a++)
          acc[a][t & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[ch & 1][a][f >> 2][f & 3], raw[p][t], acc[a][t & 3], 0, 0, 0);
      }
      {
        const int f = g * TAPS + 9;              // the shortcut: the centre tap's pixels, its own weights
#pragma unroll
        for (int a = 0; a < OWW; a++)
          accd[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[ch & 1][a][f >> 2][f & 3], raw[p][4], accd[a], 0, 0, 0);
      }
    }
  }
#undef S2_WLOAD
#undef S2_WCOMMIT
#undef S2_XLOAD

  S2_STAMP(4);
  // ---- the waves that split a tile's channel groups: the second one hands its sums over through LDS ----
  f32x4 resv[OWW], resdv[OWW];
#pragma unroll
  for (int a = 0; a < OWW; a++) {
    resv[a] = acc[a][0];
    resv[a] += acc[a][1]; resv[a] += acc[a][2]; resv[a] += acc[a][3];
    resdv[a] = accd[a];
  }
  if constexpr (KS == 2) {
    __syncthreads();                           // every wave is done with the weight image: it becomes the exchange area
    f32x4* ex = reinterpret_cast<f32x4*>(Wl) + ((ts + TS * og) * OWW * 2) * 64 + lane;
    if (ksp == 1) {
#pragma unroll
      for (int a = 0; a < OWW; a++) { ex[(2 * a) * 64] = resv[a]; ex[(2 * a + 1) * 64] = resdv[a]; }
    }
    __syncthreads();
    if (ksp == 0) {
#pragma unroll
      for (int a = 0; a < OWW; a++) { resv[a] += ex[(2 * a) * 64]; resdv[a] += ex[(2 * a + 1) * 64]; }
    }
  }
  const bool owner = ksp == 0;                 // stores and statistics of the tile
  const int swave = ts + TS * og;              // row of the statistics exchange
  // ---- epilogue: accumulator register v of lane (kk, n) is channel o + 4 kk + v of the tile's pixel n ----
#pragma unroll
  for (int a = 0; a < OWW; a++) {
    if (!owner) break;
    const f32x4 res = resv[a];
    const f32x4 resd = resdv[a];
    const int o0 = o_wg + (ot0 + a) * 16;
    const int64_t at = ((int64_t)img * O + o0 + 4 * kk) * HWO + rem;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      y[at + (int64_t)v * HWO] = res[v];
      yd[at + (int64_t)v * HWO] = resd[v];
    }
    if (partials) {
      // exact two-pass over the wave's 16 pixels per channel (both outputs), Chan's combination across the TS waves
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float m1 = row_sum16b(res[v]) * (1.0f / 16.0f), m2 = row_sum16b(resd[v]) * (1.0f / 16.0f);
        const float d1 = res[v] - m1, d2 = resd[v] - m2;
        const float q1 = row_sum16b(d1 * d1), q2 = row_sum16b(d2 * d2);
        if (n == 0) {
          sred[0][swave][a * 16 + 4 * kk + v][0] = m1; sred[0][swave][a * 16 + 4 * kk + v][1] = q1;
          sred[1][swave][a * 16 + 4 * kk + v][0] = m2; sred[1][swave][a * 16 + 4 * kk + v][1] = q2;
        }
      }
    }
  }
  if (partials) {
    __syncthreads();
    if (tid < 2 * OT * 16) {
      const int which = tid / (OT * 16), e = tid - which * (OT * 16);
      const int tl = e >> 4, chn = e & 15;       // channel tile of the workgroup, channel in it
      const int og2 = tl / OWW, a = tl - og2 * OWW;
      float nn = 16.0f, mu = sred[which][og2 * TS][a * 16 + chn][0], m2 = sred[which][og2 * TS][a * 16 + chn][1];
#pragma unroll
      for (int w4 = 1; w4 < TS; w4++) {
        const float tot = nn + 16.0f, delta = sred[which][og2 * TS + w4][a * 16 + chn][0] - mu;
        mu += delta * (16.0f / tot);
        m2 += sred[which][og2 * TS + w4][a * 16 + chn][1] + delta * delta * (nn * 16.0f / tot);
        nn = tot;
      }
      float* o = (which ? partials_d : partials) + ((int64_t)blockIdx.x * O + o_wg + tl * 16 + chn) * 3;
      o[0] = nn; o[1] = mu; o[2] = m2;
    }
  }
  S2_STAMP(5);
  S2_STAMP(7);
}

template <int C, int OWW, int TS, int OWG, int CG, int XA, int KS = 1>
int launch_s2(const float* x, const float* w, const float* wd, float* y, float* yd, float* partials, float* partials_d,
              int lwo, int O, dim3 grid) {
  const size_t lds = (size_t)(OWG * OWW) * (C / 4 * 10 / 4) * kBlk2 * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv3x3s2_pair<C, OWW, TS, OWG, CG, XA, KS>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)65536)));
    attr = true;
  }
  DF_LAUNCH((k_conv3x3s2_pair<C, OWW, TS, OWG, CG, XA, KS>), grid, dim3(64 * TS * OWG * KS), lds, stream(), x, w, wd, y, yd, partials,
            partials_d, lwo, O);
  DF_LAUNCH_CHECK();
#ifdef DFHIP_S2_STAMPS
  {
    static int calls = 0;
    if (++calls % 20 == 0) {
      const size_t nw = (size_t)grid.x * grid.y * TS * OWG * KS;
      std::vector<unsigned long long> h((size_t)grid.x * grid.y * 8 * 8);
      (void)hipStreamSynchronize(stream());
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_s2_stamps), h.size() * sizeof(unsigned long long));
      double seg[6] = {0}, cmax = 0;
      unsigned long long r0 = ~0ull, r1 = 0;
      for (size_t b = 0; b < (size_t)grid.x * grid.y; b++)
        for (int wv2 = 0; wv2 < TS * OWG * KS; wv2++) {
          const unsigned long long* r = &h[(b * 8 + wv2) * 8];
          for (int k = 1; k <= 5; k++) seg[k] += (double)(r[k] - r[k - 1]);
          cmax = std::max(cmax, (double)(r[5] - r[0]));
          r0 = std::min(r0, r[6]); r1 = std::max(r1, r[7]);
        }
      fprintf(stderr, "[s2-stamp] C %d OWW %d TS %d OWG %d CG %d XA %d KS %d grid %u x %u: span %.2f us; mean cycles per wave: issue %.0f, "
              "w-commit %.0f, barrier %.0f, mfma-loop %.0f, epilogue %.0f; slowest wave %.0f\n",
              C, OWW, TS, OWG, CG, XA, KS, grid.x, grid.y, (double)(r1 - r0) / 100.0, seg[1] / nw, seg[2] / nw, seg[3] / nw,
              seg[4] / nw, seg[5] / nw, cmax);
    }
  }
#endif
  return DFHIP_OK;
}


// ---------------------------------------------------------------------------------------
// The pair on 4 x 4 images (ResNet layer4's first block, 128 -> 256 channels, outputs 2 x 2) with an IMAGE on the lane,
// as k_conv3x3_img4_fwd (dfhip_conv_rows.hip): D[16 output channels][16 images], an accumulator per output pixel and
// output, and a tap whose source lies outside the image is not multiplied -- 25 of the 36 (pixel, tap) pairs of the
// stride-2 convolution exist, + the shortcut's four.  16 images x 16 output channels (of both convolutions) per
// workgroup, eight waves x 16 source channels streamed eight at a time through wave-private LDS areas (weight
// records and planes in memory order), two waves per SIMD: the register-operand kernel above runs this shape with one
// wave per SIMD and its loop at 55 % of the pipe rate.
// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512) void k_conv3x3s2_img4_pair(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ wd, float* __restrict__ y,
                                                             float* __restrict__ yd, float* __restrict__ partials,
                                                             float* __restrict__ partials_d, int O) {
  constexpr int CW = C / 8, NCHK = CW / 8, WP = 17, XP = 33;
  constexpr int WSL = 72 * WP + 8 * WP + 16 * XP * 4;      // weights [72][17], shortcut weights [8][17], planes [16][33] float4
  extern __shared__ __attribute__((aligned(16))) float smem_p4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const DfTile wg = {(int)blockIdx.x, (int)blockIdx.y};          // (image tile, channel tile) in grid order
  const int n0 = wg.x * 16, o0 = wg.y * 16;
  float* Wl = smem_p4 + wave * WSL;
  float* Dl = Wl + 72 * WP;
  float* Xs = Dl + 8 * WP;
  const int cb = wave * CW;
  f32x4 wq[5], xq[8];
  f32x2 dq;
  auto request = [&](int ch) {
    const int c0 = cb + 8 * ch;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int e = min(lane + 64 * i, 287), m = e / 18, j = e - 18 * m;
      wq[i] = *reinterpret_cast<const f32x4*>(w + ((int64_t)(o0 + m) * C + c0) * 9 + 4 * j);
    }
    // shortcut weights: 16 output channels x 8 channels = 64 pairs of floats (lane: channel o = lane / 4, pair lane % 4)
    dq = *reinterpret_cast<const f32x2*>(wd + (int64_t)(o0 + (lane >> 2)) * C + c0 + 2 * (lane & 3));
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = lane + 64 * i, im = e >> 5, j = e & 31;
      xq[i] = *reinterpret_cast<const f32x4*>(x + ((int64_t)(n0 + im) * C + c0) * 16 + 4 * j);
    }
  };
  f32x4 acc[4], accd[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; accd[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  request(0);
#pragma unroll
  for (int ch = 0; ch < NCHK; ch++) {
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int e = lane + 64 * i, m = min(e, 287) / 18, j = min(e, 287) - 18 * m;
      if (e < 288) {
#pragma unroll
        for (int t = 0; t < 4; t++) Wl[(4 * j + t) * WP + m] = wq[i][t];
      }
    }
    Dl[(2 * (lane & 3)) * WP + (lane >> 2)] = dq[0];
    Dl[(2 * (lane & 3) + 1) * WP + (lane >> 2)] = dq[1];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = lane + 64 * i, im = e >> 5, j = e & 31;
      *reinterpret_cast<f32x4*>(Xs + (im * XP + j) * 4) = xq[i];
    }
    if (ch + 1 < NCHK) request(ch + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float a[2][9], ad[2];
    f32x4 xr[2][4];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
      for (int t = 0; t < 9; t++) a[p][t] = Wl[((4 * p + kk) * 9 + t) * WP + n];
      ad[p] = Dl[(4 * p + kk) * WP + n];
#pragma unroll
      for (int r = 0; r < 4; r++) xr[p][r] = *reinterpret_cast<const f32x4*>(Xs + (n * XP + (4 * p + kk) * 4 + r) * 4);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int oy = 0; oy < 2; oy++)
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              const int sy = 2 * oy + i - 1, sx = 2 * ox + j - 1;
              if (sy >= 0 && sy < 4 && sx >= 0 && sx < 4)
                acc[oy * 2 + ox] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][i * 3 + j], xr[p][sy][sx], acc[oy * 2 + ox], 0, 0, 0);
            }
          accd[oy * 2 + ox] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[p], xr[p][2 * oy][2 * ox], accd[oy * 2 + ox], 0, 0, 0);
        }
  }
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(smem_p4);
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) { ex[((wave - 1) * 8 + q) * 64 + lane] = acc[q]; ex[((wave - 1) * 8 + 4 + q) * 64 + lane] = accd[q]; }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll 1
  for (int ww = 0; ww < 7; ww++)
#pragma unroll
    for (int q = 0; q < 4; q++) { acc[q] += ex[(ww * 8 + q) * 64 + lane]; accd[q] += ex[(ww * 8 + 4 + q) * 64 + lane]; }
  // register v of lane (kk, n) is output channel o0 + 4 kk + v of image n0 + n; its four output pixels are adjacent
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int64_t at = ((int64_t)(n0 + n) * O + o0 + 4 * kk + v) * 4;
    *reinterpret_cast<f32x4*>(y + at) = (f32x4){acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
    *reinterpret_cast<f32x4*>(yd + at) = (f32x4){accd[0][v], accd[1][v], accd[2][v], accd[3][v]};
  }
  if (partials) {
    // exact two-pass over the 64 values of a channel (both outputs): the lane's four pixels, then the 16 images
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float m1 = row_sum16b((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) * (1.0f / 64.0f);
      const float m2 = row_sum16b((accd[0][v] + accd[1][v]) + (accd[2][v] + accd[3][v])) * (1.0f / 64.0f);
      float q1 = 0.0f, q2 = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float d1 = acc[q][v] - m1, d2 = accd[q][v] - m2;
        q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2);
      }
      q1 = row_sum16b(q1); q2 = row_sum16b(q2);
      if (n == 0) {
        float* o1 = partials + ((int64_t)wg.x * O + o0 + 4 * kk + v) * 3;
        float* o2 = partials_d + ((int64_t)wg.x * O + o0 + 4 * kk + v) * 3;
        o1[0] = 64.0f; o1[1] = m1; o1[2] = q1;
        o2[0] = 64.0f; o2[1] = m2; o2[2] = q2;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Accumulate-mode input gradient of a 3x3 / stride 2 / pad 1 convolution (the mathematically complete sum, tensor.py:708 with
// `+=` instead of the reference's `=`; DESIGN.md "Overwrite and accumulate"), optionally with the gradient of the block's 1x1 /
// stride 2 shortcut in the same launch.  Input pixel (y, x) = (2u + ey, 2v + ex) receives from output pixel (u + du, v + dv)
// through tap (ky, kx) only where y + 1 - ky is even: class (ey, ex) has (1 + ey)(1 + ex) taps -- ey = 0: ky = 1, du = 0;
// ey = 1: (ky, du) = (0, 1) and (2, 0) -- i.e. 9 products per 4 pixels, where the implicit GEMM over the zero-stuffed dY
// multiplies 36.  A tile is 32 pixels of ONE class (consecutive positions of the flattened (n, u, v) space: the lanes read a
// dense run of dY), a workgroup stages only its class's weight slices, and the classes' workgroups are sized by their tap
// counts.  32x32x2 MFMA as in k_dgrad_ow_s1: D[32 channels][32 pixels] += W[32][2 o] . dY[2 o][32].
// SPLIT (O >= 128): the four waves of a workgroup share a tile and split o; their partial tiles meet in LDS.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x16s __attribute__((ext_vector_type(16)));
struct S2Acc {
  const float* dy; const float* w; const float* dy2; const float* w2; const float* add; float* dx;
  int C, H, lw;              // dx is (N, C, H, H), H = 1 << lw
  int tiles;                 // tiles of 32 pixels per class
  int cls_end[4];            // blockIdx.x ranges of the classes, in the order 3 (four taps), 1, 2, 0
};

template <int O, bool SPLIT, bool PAIR, int EY, int EX>
__device__ __forceinline__ void dgrad_s2_acc_class(const S2Acc& a, float* smem, const int slot, const int nslots) {
  constexpr int NT = (1 + EY) * (1 + EX);                 // taps of the class
  constexpr bool SC = PAIR && EY == 0 && EX == 0;         // the shortcut lands on the even-even pixels
  constexpr int NSL = NT + (SC ? 1 : 0);
  constexpr int OQ = SPLIT ? O / 4 : O, NP = OQ / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int C = a.C, H = a.H, W = a.H, low = a.lw - 1, oh = H >> 1, ow = W >> 1;
  const int ohw = oh * ow;
  const int c0 = blockIdx.y * 32;
  float* Wl = smem;                                       // [NSL][O][32]
  float* red = smem + NSL * O * 32;                       // SPLIT: [3][16][64]
  // weight slices of this class: tap t = ty * (1 + EX) + tx; ty -> ky = EY ? 2 * ty : 1, du = EY ? 1 - ty : 0
  {
    // (twelve (o, c) pairs of a thread in flight: <= 48 loads, under the 63 the wait counter can tell apart)
    constexpr int PAIRS = O * 32 / 256, PB = PAIRS < 12 ? PAIRS : (PAIRS % 12 == 0 ? 12 : 8);
    static_assert(PAIRS % PB == 0, "weight staging rounds");
    for (int i0 = 0; i0 < PAIRS; i0 += PB) {
      float wv[PB][NSL];
#pragma unroll
      for (int i = 0; i < PB; i++) {
        const int e = tid + 256 * (i0 + i), o = e >> 5, c = e & 31;
        const float* q = a.w + ((int64_t)o * C + c0 + c) * 9;
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int ty = t / (1 + EX), tx = t % (1 + EX);
          const int ky = EY ? 2 * ty : 1, kx = EX ? 2 * tx : 1;
          wv[i][t] = q[ky * 3 + kx];
        }
        if constexpr (SC) wv[i][NT] = a.w2[(int64_t)o * C + c0 + c];
      }
#pragma unroll
      for (int i = 0; i < PB; i++) {
        const int e = tid + 256 * (i0 + i), o = e >> 5, c = e & 31;
#pragma unroll
        for (int t = 0; t < NSL; t++) Wl[(t * O + o) * 32 + c] = wv[i][t];
      }
    }
  }
  __syncthreads();
  const int ob = SPLIT ? wave * OQ : 0;
  const int first = SPLIT ? slot : slot * 4 + wave, step = SPLIT ? nslots : nslots * 4;
  for (int tile = first; tile < a.tiles; tile += step) {
    const int p = tile * 32 + l31;
    const int n = p / ohw, rem = p - n * ohw, u = rem >> low, v = rem & (ow - 1);
    // every operand of the tile is requested up front (unconditional loads at a clamped position, zeroed afterwards)
    float sv[NSL][NP];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int ty = t / (1 + EX), tx = t % (1 + EX);
      const int du = EY ? 1 - ty : 0, dv = EX ? 1 - tx : 0;
      const int su = min(u + du, oh - 1), sx = min(v + dv, ow - 1);
      const float* sp = a.dy + ((int64_t)n * O + ob + half) * ohw + su * ow + sx;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) sv[t][kp] = sp[(int64_t)(2 * kp) * ohw];
    }
    if constexpr (SC) {
      const float* sp = a.dy2 + ((int64_t)n * O + ob + half) * ohw + u * ow + v;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) sv[NT][kp] = sp[(int64_t)(2 * kp) * ohw];
    }
    const int y = 2 * u + EY, x = 2 * v + EX;
    const int64_t obase = (((int64_t)n * C + c0) * H + y) * W + x;
    const int64_t HW = (int64_t)H * W;
    float held[16];
    const bool stores = !SPLIT || wave == 0;
#pragma unroll
    for (int r = 0; r < 16; r++)
      held[r] = (a.add && stores) ? a.add[obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW] : 0.0f;
    f32x16s acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < NSL; t++) {
      bool inside = true;
      if (t < NT) {
        const int ty = t / (1 + EX), tx = t % (1 + EX);
        const int du = EY ? 1 - ty : 0, dv = EX ? 1 - tx : 0;
        inside = u + du < oh && v + dv < ow;
      }
      const float* wq = Wl + (t * O + ob + half) * 32 + l31;
#pragma unroll
      for (int kp = 0; kp < NP; kp++)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[2 * kp * 32], inside ? sv[t][kp] : 0.0f, acc, 0, 0, 0);
    }
    if constexpr (SPLIT) {
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int ww = 0; ww < 3; ww++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[r] += red[(ww * 16 + r) * 64 + lane];
      }
    }
    if (stores) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        a.dx[obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW] = acc[r] + held[r];
    }
    if constexpr (SPLIT) __syncthreads();               // the exchange area is free again
  }
}

template <int O, bool SPLIT, bool PAIR>
__global__ __launch_bounds__(256) void k_dgrad_s2_acc(S2Acc a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < a.cls_end[0]) dgrad_s2_acc_class<O, SPLIT, PAIR, 1, 1>(a, smem, b, a.cls_end[0]);
  else if (b < a.cls_end[1]) dgrad_s2_acc_class<O, SPLIT, PAIR, 0, 1>(a, smem, b - a.cls_end[0], a.cls_end[1] - a.cls_end[0]);
  else if (b < a.cls_end[2]) dgrad_s2_acc_class<O, SPLIT, PAIR, 1, 0>(a, smem, b - a.cls_end[1], a.cls_end[2] - a.cls_end[1]);
  else dgrad_s2_acc_class<O, SPLIT, PAIR, 0, 0>(a, smem, b - a.cls_end[2], a.cls_end[3] - a.cls_end[2]);
}

template <int O, bool SPLIT, bool PAIR>
int launch_dgrad_s2_acc(const S2Acc& a, int gx, int C) {
  const size_t lds = ((size_t)4 * O * 32 + (SPLIT ? 3 * 16 * 64 : 0)) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_s2_acc<O, SPLIT, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(lds, (size_t)65536)));
    attr = true;
  }
  DF_LAUNCH((k_dgrad_s2_acc<O, SPLIT, PAIR>), dim3((unsigned)gx, (unsigned)(C / 32)), dim3(256), lds, stream(), a);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

}  // namespace

namespace dfhip {

// dx (N, C, H, H) = [add +] the complete (accumulate-mode) input gradient of conv3x3/s2/p1 from dy (N, O, H/2, H/2)
// [+ that of the block's conv1x1/s2/p0 shortcut from dy2, w2]; -1 when the shape is not one this kernel takes
int run_dgrad_s2_acc(const float* dy, const float* w, const float* dy2, const float* w2, const float* add, float* dx, int N,
                     int C, int H, int W, int O, bool dry_run) {
  if (H != W || H < 4 || (H & (H - 1)) || C % 32 != 0 || !(O == 64 || O == 128 || O == 256)) return -1;
  const int64_t cp = (int64_t)N * (H / 2) * (W / 2);       // pixels per class
  if (cp % 128 != 0) return -1;
  if (dry_run) return DFHIP_OK;
  int lw = 0;
  while ((1 << lw) < W) lw++;
  S2Acc a = {};
  a.dy = dy; a.w = w; a.dy2 = dy2; a.w2 = w2; a.add = add; a.dx = dx;
  a.C = C; a.H = H; a.lw = lw; a.tiles = (int)(cp / 32);
  const bool split = O >= 128, pair = dy2 != nullptr;
  // workgroups per class, in the order (1,1), (0,1), (1,0), (0,0).  Unsplit: four tiles per workgroup whatever the class (the
  // long ones are dispatched first).  SPLIT: a workgroup walks tiles; the classes share about one workgroup per CU and channel
  // tile in proportion to their taps
  const int weight[4] = {4, 2, 2, pair ? 2 : 1};
  int cnt[4];
  if (!split) {
    for (int q = 0; q < 4; q++) cnt[q] = a.tiles / 4;
  } else {
    // (measured and dropped: a walk sized by tiles x taps with 65 instead of 61 workgroups per channel tile -- 26.5 -> 31 us at
    // 256 channels, every workgroup stages its class's slices, up to 131 KB --; half / a quarter of the workgroups -- 35 / 59
    // us, a tile costs ~9 us of latency --; the next tile's operands requested before the current tile's MFMAs -- 33 us, a
    // workgroup with one tile then loads two)
    const int budget = std::max(9, num_cus() / (C / 32));
    const int wsum = weight[0] + weight[1] + weight[2] + weight[3];
    for (int q = 0; q < 4; q++) cnt[q] = std::max(1, std::min(a.tiles, budget * weight[q] / wsum));
  }
  int gx = 0;
  for (int q = 0; q < 4; q++) { gx += cnt[q]; a.cls_end[q] = gx; }
  if (O == 64) return pair ? launch_dgrad_s2_acc<64, false, true>(a, gx, C) : launch_dgrad_s2_acc<64, false, false>(a, gx, C);
  if (O == 128) return pair ? launch_dgrad_s2_acc<128, true, true>(a, gx, C) : launch_dgrad_s2_acc<128, true, false>(a, gx, C);
  return pair ? launch_dgrad_s2_acc<256, true, true>(a, gx, C) : launch_dgrad_s2_acc<256, true, false>(a, gx, C);
}

// x (N, C, H, H) -> y = conv3x3/s2/p1(x, w) and yd = conv1x1/s2/p0(x, wd), both (N, O, H/2, H/2); -1 when the shape is
// not one this kernel takes (dry_run: only the shape test)
int run_conv3x3s2_pair(const float* x, const float* w, const float* wd, float* y, float* yd, int N, int C, int H, int W,
                       int O, float* partials, float* partials_d, int cap_rows, int* rows_out, bool dry_run) {
  if (H != W) return -1;
  constexpr int img4 = 1;
  if (img4 && C == 128 && H == 4 && N % 16 == 0 && O % 16 == 0 && (!partials || N / 16 <= cap_rows) &&
      (int64_t)(N / 16) * (O / 16) >= num_cus() / 2) {
    // an image on the lane, taps outside the image not multiplied (k_conv3x3s2_img4_pair)
    if (rows_out) *rows_out = partials ? N / 16 : 0;
    if (dry_run) return DFHIP_OK;
    dim3 grid4((unsigned)(N / 16), (unsigned)(O / 16));
    const size_t lds4 = std::max((size_t)8 * (72 * 17 + 8 * 17 + 16 * 33 * 4), (size_t)7 * 8 * 64 * 4) * sizeof(float);
    static bool attr4 = false;
    if (!attr4) {
      DF_HIP(hipFuncSetAttribute((const void*)k_conv3x3s2_img4_pair<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
      attr4 = true;
    }
    DF_LAUNCH(k_conv3x3s2_img4_pair<128>, grid4, dim3(512), lds4, stream(), x, w, wd, y, yd, partials, partials_d, O);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  // (a workgroup = eight pixel tiles x ONE group of channel tiles -- 32 -> 64 channels two tiles per wave, 64 -> 128 one: a pixel
  // tile is requested by one wave of the workgroup only and the weight image is half as large: 11.8 -> 11.3 and 11.3 -> 10.1 us
  // against four tiles x two groups / two channel tiles per wave (round 5; the 64-channel form has fewer weight pieces than
  // staging threads: the spare threads' piece index is taken modulo the piece count).  Measured and dropped before: chunks of
  // four channel groups (13.4 -> 14.9 us), three chunks of pixels ahead, two waves splitting a tile's groups.)
  int otw, ts, lwo;
  if (C == 32 && H == 16) { otw = 2; ts = 8; lwo = 3; }
  else if (C == 64 && H == 8) { otw = 1; ts = 8; lwo = 2; }
  else if (C == 128 && H == 4) { otw = 1; ts = 4; lwo = 1; }
  else return -1;
  const int ocw = otw * 16;                    // output channels per workgroup
  const int64_t opix = (int64_t)N * (H / 2) * (W / 2);
  if (O % ocw != 0 || opix % (16 * ts) != 0) return -1;
  const int64_t gx = opix / (16 * ts);
  if (partials && gx > cap_rows) return -1;
  if (gx * (O / ocw) < num_cus() / 2) return -1;
  if (rows_out) *rows_out = partials ? (int)gx : 0;
  if (dry_run) return DFHIP_OK;
  dim3 grid((unsigned)gx, (unsigned)(O / ocw));
  if (C == 32) return launch_s2<32, 2, 8, 1, 2, 2>(x, w, wd, y, yd, partials, partials_d, lwo, O, grid);
  if (C == 64) return launch_s2<64, 1, 8, 1, 2, 2>(x, w, wd, y, yd, partials, partials_d, lwo, O, grid);
  return launch_s2<128, 1, 4, 1, 2, 2>(x, w, wd, y, yd, partials, partials_d, lwo, O, grid);
}

}  // namespace dfhip
