// Overwrite-mode ("last writer wins") input gradients of the 3x3 / 1x1 convolutions, the kernel family
// `conv2d_dgrad_overwrite_igemm`, and the BatchNorm-backward forms that ride on it (dfhip_bnbwd.h).
//
// Reference being replaced: tensor.py:708 (dC . B^T) -> __im2col2d.grad_fn (nn/functional.py:285-294: the k*k
// overlapping windows are ASSIGNED in (i, j) row-major order) -> __pad2d.grad_fn (:307-313); with the BatchNorm
// descriptors also the autograd chain of nn/modules/batchnorm.py:30-55 on either side of the convolution.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
#include <type_traits>

#include "dfhip_bnbwd.h"

using namespace dfhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// BatchNorm descriptors of one launch, as the kernels see them
// taps (1, 1), (1, 2) and (2, 1), (2, 2) of a 3x3 weight record are two pairs of neighbouring floats: two 8-byte loads (4-byte
// aligned: the records are 36 bytes apart) instead of four 4-byte ones -- the weight slices of a workgroup were 8-64 load
// instructions per thread, each a strided gather over ~18 lines
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void load_taps_4578(const float* __restrict__ q, float (&out)[4]) {
  const f32x2u a = *reinterpret_cast<const f32x2u*>(q + 4), b = *reinterpret_cast<const f32x2u*>(q + 7);
  out[0] = a[0]; out[1] = a[1]; out[2] = b[0]; out[3] = b[1];
}

// diagnostic build only (-DDFHIP_OW_STAMPS): per-wave cycle stamps of k_dgrad_ow_s1 (s_memtime; 6 / 7: s_memrealtime at the start /
// end of the wave), dumped for every 20th launch of each instantiation: where a 8-12 us launch's cycles go
#ifdef DFHIP_OW_STAMPS
__device__ unsigned long long g_ow_stamps[1024 * 8 * 8];
#define OW_STAMP(i)                                                                                         \
  do {                                                                                                      \
    if ((threadIdx.x & 63) == 0 && (blockIdx.y * gridDim.x + blockIdx.x) < 1024)                            \
      g_ow_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + (i)] =             \
          (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();                       \
  } while (0)
static void ow_stamp_dump(const char* what, int wgs, int nw, int* calls) {
  if (++*calls % 20) return;
  const int nb = wgs < 1024 ? wgs : 1024;
  std::vector<unsigned long long> h((size_t)nb * 8 * 8);
  (void)hipStreamSynchronize(stream());
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_ow_stamps), h.size() * sizeof(unsigned long long));
  double seg[6] = {0};
  unsigned long long r0 = ~0ull, r1 = 0, rs = 0;
  size_t cnt = 0;
  for (int b = 0; b < nb; b++)
    for (int w = 0; w < nw; w++) {
      const unsigned long long* r = &h[((size_t)b * 8 + w) * 8];
      for (int k = 1; k <= 5; k++) seg[k] += (double)(r[k] - r[k - 1]);
      r0 = std::min(r0, r[6]); r1 = std::max(r1, r[7]); rs = std::max(rs, r[6]);
      cnt++;
    }
  fprintf(stderr, "[ow-stamp] %s, %d workgroups x %d waves: first start -> last end %.2f us, last wave starts +%.2f us; mean cycles per wave: "
          "until the loads are issued %.0f, weights to LDS + table %.0f, barrier %.0f, operand transform + MFMAs %.0f, stores + rows %.0f\n",
          what, wgs, nw, (double)(r1 - r0) / 100.0, (double)(rs - r0) / 100.0, seg[1] / cnt, seg[2] / cnt, seg[3] / cnt, seg[4] / cnt,
          seg[5] / cnt);
}
#else
#define OW_STAMP(i) do { } while (0)
#endif

struct OwBn {
  BnBwdIn in, in2;             // of dy (and of the shortcut's dy2); rows == nullptr: dz = dy
  const float* zt[2];          // the (<= 2) layers whose output gradient dx is: their inputs and batch means
  const float* mt[2];
  float* rows;                 // one row of sums per workgroup: [gridDim.x][C][1 + NST]
};

// ---------------------------------------------------------------------------------------
// "overwrite" input gradient of a 3x3 / stride 1 / pad 1 convolution, as what it is: the reference's
// k*k strided assignments (functional.py:285-294, last writer wins) leave, for input pixel (y, x),
// only the tap (ti, tj) = (y >= 1 ? 2 : 1, x >= 1 ? 2 : 1) and with it the source pixel
// (max(y-1, 0), max(x-1, 0)):
//     dX[n][c][y][x] = sum_o dY[n][o][max(y-1,0)][max(x-1,0)] * W[o][c][ti][tj]
// -- a 1x1 convolution of a shifted, edge-clamped dY with one of four weight slices (interior,
// top row, left column, corner), 1/9 of the products of the true gradient.  With so little
// arithmetic the staged kernel above is all fixed cost (12 us for 2.5 us of MFMAs); here a wave owns
// a 32-pixel x 32-channel tile outright: the dY operand of its pixels is read ONCE from global
// memory into registers (O/2 coalesced loads), each of the <= 4 weight slices present in the tile
// is applied with the lanes of the other slices masked, and the tile leaves as sixteen 128-byte
// stores per half-wave (plus the gradient the tensor already holds, tensor.py:444).  No LDS, no
// barrier.  Used while O <= 64 (dY operand in registers) and a tile is 32 pixels of one image.
// ---------------------------------------------------------------------------------------
// GEO 0: 3x3 / stride 1 / pad 1 (above).  GEO 1: 3x3 / stride 2 / pad 1 -- the surviving tap of pixel
// (y, x) is (1 + (y & 1), 1 + (x & 1)) and its source (y >> 1, x >> 1): a nearest-neighbour
// up-sampling of dY followed by one of four 1x1 convolutions by pixel parity.  GEO 2: 1x1 / stride 2 /
// pad 0 -- even (y, x) receive sum_o dY[o][y/2][x/2] W[o][c], the others stay zero.  GEO 3 (round 6; the CNN workload's
// conv2, test/CNN_CIFAR10_cuda.py:66): 5x5 / stride 1 / pad 2 -- the last of the 25 assignments that reaches input pixel
// (y, x) is tap (2 + min(y, 2), 2 + min(x, 2)), its source (max(y - 2, 0), max(x - 2, 0)): nine weight slices by
// (min(y, 2), min(x, 2)), 1/25 of the products of the true gradient (the staged implicit-GEMM kernel this replaces for that
// layer multiplied all 25 taps' worth of zeros: 27 us for 25 MB).
template <int GEO>
struct OwGeo {
  static constexpr int NSL = GEO == 2 ? 1 : (GEO == 3 ? 9 : 4);      // weight slices
  static constexpr int NSA = GEO == 3 ? 9 : 4;                      // (registers of a thread's slice values)
  __device__ static void pixel(int y, int x, int& u, int& v, int& cls) {
    if (GEO == 0) { u = max(y - 1, 0); v = max(x - 1, 0); cls = (y >= 1 ? 2 : 0) + (x >= 1 ? 1 : 0); }
    else if (GEO == 1) { u = y >> 1; v = x >> 1; cls = ((y & 1) << 1) + (x & 1); }
    else if (GEO == 3) { u = max(y - 2, 0); v = max(x - 2, 0); cls = 3 * min(y, 2) + min(x, 2); }
    else { u = y >> 1; v = x >> 1; cls = ((y | x) & 1) ? -1 : 0; }
  }
  // lane -> pixel of tile t.  Default: 32 consecutive pixels of the flattened (n, y, x) space.  Stride 2
  // with pg: the four tiles 4b..4b+3 are the four parity classes of one block of 128 flat pixels (128/W
  // flat rows), so a tile holds ONE class -- a quarter of the MFMAs of a mixed tile -- and its lanes read
  // a dense run of dY; the classes' stride-2 stores complete each other's cache lines.
  __device__ static void decode(int t, int l31, int lw, int H, int pg, int& n, int& y, int& x) {
    if (GEO == 1 && pg) {
      const int b = t >> 2, py = (t >> 1) & 1, px = t & 1;
      const int ry = l31 >> (lw - 1), rx = l31 & ((1 << (lw - 1)) - 1);
      const int R = b * (128 >> lw) + 2 * ry + py;          // flat row (n * H + y); H is even
      n = R / H; y = R - n * H; x = 2 * rx + px;
    } else if (GEO == 0 && pg) {
      // Stride 1 with pg: the tiles of a group of 32 / W images (H tiles) are its image rows 0 -- one tile -- and then the
      // pixels of rows 1 .. H - 1 in memory order, H - 1 tiles.  A tile then never holds image row 0 TOGETHER with later
      // rows, i.e. never all four tap classes: every wave runs two MFMA passes (x = 0 or not) where the waves of the
      // tiles that held row 0 ran four -- and with one wave per SIMD those waves were the length of the launch.
      const int W = 1 << lw, G = 32 >> lw;
      const int g = t / H, k = t - g * H;
      if (k == 0) { n = g * G + (l31 >> lw); y = 0; x = l31 & (W - 1); }
      else {
        const int q = (k - 1) * 32 + l31, per = (H - 1) << lw;
        const int im = q / per, r2 = q - im * per;
        n = g * G + im; y = 1 + (r2 >> lw); x = r2 & (W - 1);
      }
    } else {
      const int HW = H << lw, p = t * 32 + l31;
      n = p / HW;
      const int rem = p - n * HW;
      y = rem >> lw; x = rem & ((1 << lw) - 1);
    }
  }
  // the NSL weights of one (o, c) pair, from the (O, C, k, k) tensor
  __device__ static void slices(const float* __restrict__ w, int64_t oc, float (&out)[NSA]) {
    if constexpr (GEO == 2) { out[0] = w[oc]; out[1] = out[2] = out[3] = 0.0f; }
    else if constexpr (GEO == 3) {
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) out[3 * a + b] = w[oc * 25 + (2 + a) * 5 + 2 + b];
    } else load_taps_4578(w + oc * 9, out);
  }
};

// PAIR (GEO 1 only): the block's 1x1 / stride 2 shortcut convolution reads the same input tensor, and its input
// gradient lands on the pixels with even (y, x) -- parity class 0 of this kernel, same source pixel (y >> 1, x >> 1):
// dy2 / w2 (O, C, 1, 1) are a FIFTH slice with an operand of its own, and the two gradients of the tensor leave as one
// sum (tensor.py:444) instead of a second launch that re-reads and re-writes the even pixels.
// BNI: dy (and dy2) are gradients of BatchNorm OUTPUTS whose backward pass has not been applied; the operand is formed
// on load from (dy, z) and the per-channel table built from the producer's rows (dfhip_bnbwd.h).  NST: dx is the output
// gradient of NST BatchNorm layers; the tile's per-channel sums of dx and dx (z_t - mean_t) leave as partial rows.
// NW: waves (32-pixel tiles) per workgroup.  4 everywhere but for the 32-channel BatchNorm builds (NW = 8): a workgroup leaves
// ONE row of sums, and every workgroup of the consuming launch reads all rows -- with 512 workgroups of 4 waves on the
// 16 x 16 x 32-channel layers that was 512 x 64 KB = 32 MB of row reads per launch (PMC: 40 MB fetched for 19.6 MB algorithmic),
// with 256 workgroups of 8 waves it is 256 x 32 KB.
#ifdef DFHIP_RIDER_PROBE
// Diagnostic build (round 5, tools/probes/rider_probe.sh): extra workgroups of this kernel -- the ones past the last tile --
// spin on independent fp32 MFMAs for g_rider_iters x 4 instructions per wave: what a matrix-bound rider that shares the CUs
// of a chain kernel costs that kernel, and how much matrix work it gets done meanwhile.
__device__ int g_rider_iters;
__device__ __forceinline__ void rider_spin() {
  typedef float r4 __attribute__((ext_vector_type(4)));
  r4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = 0.3f + threadIdx.x * 1e-3f;
  const int iters = g_rider_iters;
  if (iters < 0) {                                   // control: a rider that only SLEEPS for -iters x 0.1 us (no pipe is used)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(-iters) * 10ull) __builtin_amdgcn_s_sleep(8);
    return;
  }
  for (int i = 0; i < iters; i++) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.25f, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.125f, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.0625f, a3, 0, 0, 0);
  }
  const float r = a0[0] + a1[1] + a2[2] + a3[3];
  asm volatile("" ::"v"(r));                         // (the sums are "used": the loop stays)
}
#endif
template <int O, int GEO, bool PAIR, bool BNI, int NST, int NW>
__global__ __launch_bounds__(64 * NW) void k_dgrad_ow_s1(const float* __restrict__ dy, const float* __restrict__ w,
                                                     const float* __restrict__ add, float* __restrict__ dx,
                                                     int C, int H, int lw, int tiles, int pg,
                                                     const float* __restrict__ dy2, const float* __restrict__ w2,
                                                     const OwBn bn) {
  static_assert(!PAIR || GEO == 1, "the shortcut rides on the stride-2 3x3 kernel");
  constexpr int NP = O / 2, NSL = OwGeo<GEO>::NSL;
  constexpr int KS = 1 + NST;
  constexpr int NT = 64 * NW;                       // threads: NW waves, a 32-pixel tile each
  constexpr bool BN = BNI || NST > 0;               // a BatchNorm build: more loads than a wave keeps in flight, see below
  // the weight slices of this channel tile: [slice][o][c] (the nine slices of a 5x5 kernel: 74 KB at 64 channels, dynamic LDS)
  // (128 channels of dY: the four slices of a 3x3 kernel are 64 KB -- dynamic as well)
  constexpr bool DYN = GEO == 3 || O > 64;
  __shared__ float Wl_static[DYN ? 1 : NSL + (PAIR ? 1 : 0)][DYN ? 1 : O][32];
  extern __shared__ __attribute__((aligned(16))) float Wl_dynamic[];
  float (*Wl)[O][32] = DYN ? reinterpret_cast<float (*)[O][32]>(Wl_dynamic) : reinterpret_cast<float (*)[O][32]>(Wl_static);
  __shared__ float4 tab[BNI ? (PAIR ? 2 : 1) : 1][BNI ? O : 1];
  __shared__ float wsum[NST ? NW : 1][NST ? 32 : 1][KS];
  __shared__ float mtl[NST ? NST : 1][32];
  __shared__ __attribute__((aligned(16))) float tpart[BNI ? bn_part_floats(64 * NW) : 4];
#ifdef DFHIP_RIDER_PROBE
  if ((int)blockIdx.x * NW >= tiles) { rider_spin(); return; }      // (uniform per workgroup)
#endif
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  OW_STAMP(6); OW_STAMP(0);
  const int HW = H << lw;
  const int c0 = blockIdx.y * 32;
  const int tile = min((int)blockIdx.x * NW + wave, tiles - 1);   // 32 consecutive pixels of one image (clamped: a
  const bool live = (int)blockIdx.x * NW + wave < tiles;          //  wave past the end still takes part in the barriers)
  int n, y, x;
  OwGeo<GEO>::decode(tile, l31, lw, H, pg, n, y, x);
  const int rem = (y << lw) + x;
  int u, v, cls;                                    // source pixel, slice q <-> tap (1 + (q >> 1), 1 + (q & 1))
  OwGeo<GEO>::pixel(y, x, u, v, cls);
  const int slw = (GEO == 0 || GEO == 3) ? lw : lw - 1;   // source plane: H x W (stride 1), or H/2 x W/2
  const int SHW = (GEO == 0 || GEO == 3) ? HW : HW >> 2;
  const int64_t soff = ((int64_t)n * O + half) * SHW + (u << slw) + v;
  const int64_t obase = ((int64_t)n * C + c0) * HW + rem;
  constexpr int PAIRS = O * 32 / NT;
  static_assert(O * 32 % NT == 0, "whole rounds of weight staging");
  float wv[PAIRS][OwGeo<GEO>::NSA], wv2[PAIRS];
  auto load_weights = [&]() {       // the thread's share of the weight slices: (o, c) pairs, the four taps 4, 5, 7, 8 of each
#pragma unroll
    for (int i = 0; i < PAIRS; i++) {
      const int e = threadIdx.x + NT * i, o = e >> 5, c = e & 31;
      OwGeo<GEO>::slices(w, (int64_t)o * C + c0 + c, wv[i]);
      wv2[i] = PAIR ? w2[(int64_t)o * C + c0 + c] : 0.0f;
    }
  };
  float sv[NP], zv[BNI ? NP : 1], sv2[PAIR ? NP : 1], zv2[(PAIR && BNI) ? NP : 1];
  auto load_operands = [&]() {      // the dY operand of the wave's pixels (and the BatchNorm inputs under it)
    const float* sp = dy + soff;
#pragma unroll
    for (int kp = 0; kp < NP; kp++) sv[kp] = sp[(int64_t)(2 * kp) * SHW];
    if constexpr (BNI) {
      const float* zp = bn.in.z + soff;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) zv[kp] = zp[(int64_t)(2 * kp) * SHW];
    }
    if constexpr (PAIR) {
      const float* sp2 = dy2 + soff;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) sv2[kp] = sp2[(int64_t)(2 * kp) * SHW];
      if constexpr (BNI) {
        const float* zp = bn.in2.z + soff;
#pragma unroll
        for (int kp = 0; kp < NP; kp++) zv2[kp] = zp[(int64_t)(2 * kp) * SHW];
      }
    }
  };
  float held[16], zt[NST ? NST : 1][16];
  auto load_epilogue = [&]() {      // the gradient the tensor already holds; the inputs of the layers this gradient is for
#pragma unroll
    for (int r = 0; r < 16; r++)
      held[r] = add ? add[obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW] : 0.0f;
    if constexpr (NST > 0) {
#pragma unroll
      for (int j = 0; j < NST; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) zt[j][r] = bn.zt[j][obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW];
      }
    }
  };
  if constexpr (!BN) {
    // everything this wave needs from global memory is requested up front: its dY operand, the gradient the tensor already
    // holds (added in the epilogue; requested now, not after the MFMAs), its share of the weight slices
    load_operands();
    load_epilogue();
    load_weights();
  } else {
    // The BatchNorm builds have more to fetch than the 63 loads a wave keeps in flight.  First what the barrier needs -- the
    // weight slices, the rows of the pending BatchNorm (all threads add them up while the operands are on their way: the
    // per-channel table) and the batch means of the layers this gradient is for --, then the operands; what only the epilogue
    // needs is requested behind the barrier and lands under the MFMAs.
    load_weights();
    if constexpr (NST > 0) {
      if ((int)threadIdx.x >= NT - 64 && (int)threadIdx.x < NT - 64 + 32 * NST) {
        const int j = (threadIdx.x - (NT - 64)) >> 5, ch = threadIdx.x & 31;
        mtl[j][ch] = bn.mt[j][c0 + ch];
      }
    }
    if constexpr (BNI) {
      const bool first = lin == 0;
      BnTableBuilder<NT> tb;
      tb.issue(bn.in, O);
      __builtin_amdgcn_sched_barrier(0);
      load_operands();
      __builtin_amdgcn_sched_barrier(0);
      tb.finish(bn.in, O, tab[0], tpart, first);
      if constexpr (PAIR) {
        __syncthreads();                               // (the scratch of the first table is free)
        tb.issue(bn.in2, O);
        tb.finish(bn.in2, O, tab[1], tpart, first);
      }
    } else {
      load_operands();
    }
  }
  OW_STAMP(1);
#pragma unroll
  for (int i = 0; i < PAIRS; i++) {
    const int e = threadIdx.x + NT * i, o = e >> 5, c = e & 31;
#pragma unroll
    for (int q = 0; q < NSL; q++) Wl[q][o][c] = wv[i][q];
    if constexpr (PAIR) Wl[NSL][o][c] = wv2[i];
  }
  OW_STAMP(2);
  __syncthreads();
  OW_STAMP(3);
#ifndef DFHIP_OW_STAMPS
  if constexpr (NST == 0) { if (!live) return; }
#endif
  if constexpr (BN) {
    __builtin_amdgcn_sched_barrier(0);
    load_epilogue();
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (BNI) {
#pragma unroll
    for (int kp = 0; kp < NP; kp++) sv[kp] = bn_bwd_point(sv[kp], zv[kp], tab[0][2 * kp + half]);
    if constexpr (PAIR) {
#pragma unroll
      for (int kp = 0; kp < NP; kp++) sv2[kp] = bn_bwd_point(sv2[kp], zv2[kp], tab[1][2 * kp + half]);
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
  for (int q = 0; q < NSL; q++) {
    const bool mine = cls == q;
    if (__ballot(mine) == 0ull) continue;
    float a[NP];
#pragma unroll
    for (int kp = 0; kp < NP; kp++) a[kp] = Wl[q][2 * kp + half][l31];
#pragma unroll
    for (int kp = 0; kp < NP; kp++)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], mine ? sv[kp] : 0.0f, acc, 0, 0, 0);
  }
  if constexpr (PAIR) {
    const bool mine = cls == 0;
    if (__ballot(mine) != 0ull) {
      float a[NP];
#pragma unroll
      for (int kp = 0; kp < NP; kp++) a[kp] = Wl[NSL][2 * kp + half][l31];
#pragma unroll
      for (int kp = 0; kp < NP; kp++)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], mine ? sv2[kp] : 0.0f, acc, 0, 0, 0);
    }
  }
  OW_STAMP(4);
  float outv[16];
#pragma unroll
  for (int r = 0; r < 16; r++) outv[r] = acc[r] + held[r];
  if (live) {
#pragma unroll
    for (int r = 0; r < 16; r++) dx[obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW] = outv[r];
  }
  if constexpr (NST > 0) {
    // per-channel sums of the tile: over the 32 pixels of the half-wave, then over the four waves; the workgroup's row
    float s[KS][16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float o_ = live ? outv[r] : 0.0f;
      s[0][r] = o_;
#pragma unroll
      for (int j = 0; j < NST; j++) s[1 + j][r] = o_ * (zt[j][r] - mtl[j][(r & 3) + 8 * (r >> 2) + 4 * half]);
    }
#pragma unroll
    for (int k = 0; k < KS; k++) half_reduce16(s[k], lane);
    if ((l31 & 1) == 0) {
      const int r = 8 * ((l31 >> 4) & 1) + 4 * ((l31 >> 3) & 1) + 2 * ((l31 >> 2) & 1) + ((l31 >> 1) & 1);
      const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int k = 0; k < KS; k++) wsum[wave][ch][k] = s[k][0];
    }
    __syncthreads();
    if ((int)threadIdx.x < 32 * KS) {
      const int ch = threadIdx.x / KS, k = threadIdx.x % KS;
      float t = (wsum[0][ch][k] + wsum[1][ch][k]) + (wsum[2][ch][k] + wsum[3][ch][k]);
      if constexpr (NW == 8) t += (wsum[4][ch][k] + wsum[5][ch][k]) + (wsum[6][ch][k] + wsum[7][ch][k]);
      bn.rows[((int64_t)blockIdx.x * C + c0) * KS + threadIdx.x] = t;
    }
  }
  OW_STAMP(5); OW_STAMP(7);
}

// The same for 128 / 256 output channels: ONE tile per workgroup, its four waves split the reduction
// over o four ways (each keeps O/8 dY values per lane) and their partial tiles are summed through LDS
// in wave order.  The weight slices (4 x O x 32 floats) sit in dynamic LDS.
template <int O, int GEO, bool PAIR = false>
__global__ __launch_bounds__(256) void k_dgrad_ow_s1_ks(const float* __restrict__ dy, const float* __restrict__ w,
                                                        const float* __restrict__ add, float* __restrict__ dx,
                                                        int C, int H, int lw, int tiles, int pg,
                                                        const float* __restrict__ dy2 = nullptr,
                                                        const float* __restrict__ w2 = nullptr) {
  static_assert(!PAIR || GEO == 1, "the shortcut rides on the stride-2 3x3 kernel");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int OQ = O / 4, NP = OQ / 2, NSL = OwGeo<GEO>::NSL, NSLP = NSL + (PAIR ? 1 : 0);
  float* Wl = smem;                                 // [NSLP][O][32]
  float* red = smem + NSLP * O * 32;                // [3][16][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int HW = H << lw;
  const int c0 = blockIdx.y * 32;
  const int slw = (GEO == 0 || GEO == 3) ? lw : lw - 1;   // source plane: H x W (stride 1), or H/2 x W/2
  const int SHW = (GEO == 0 || GEO == 3) ? HW : HW >> 2;
  const int ob = wave * OQ;                         // this wave's slice of the reduction
  // the weight slices of this channel tile, once per workgroup (it walks several pixel tiles)
  constexpr int PAIRS = O * 32 / 256;
  constexpr int WB = PAIRS < 16 ? PAIRS : 16;
  static_assert(PAIRS % WB == 0, "whole rounds of weight staging");
  for (int i0 = 0; i0 < PAIRS; i0 += WB) {          // up to sixteen (o, c) pairs in flight per thread
    float wv[WB][OwGeo<GEO>::NSA], wv2[WB];
#pragma unroll
    for (int i = 0; i < WB; i++) {
      const int e = threadIdx.x + 256 * (i0 + i), o = e >> 5, c = e & 31;
      OwGeo<GEO>::slices(w, (int64_t)o * C + c0 + c, wv[i]);
      wv2[i] = PAIR ? w2[(int64_t)o * C + c0 + c] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < WB; i++) {
      const int e = threadIdx.x + 256 * (i0 + i), o = e >> 5, c = e & 31;
#pragma unroll
      for (int q = 0; q < NSL; q++) Wl[(q * O + o) * 32 + c] = wv[i][q];
      if constexpr (PAIR) Wl[(NSL * O + o) * 32 + c] = wv2[i];
    }
  }
  __syncthreads();
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    int n, y, x;
    OwGeo<GEO>::decode(tile, l31, lw, H, pg, n, y, x);
    const int rem = (y << lw) + x;
    int u, v, cls;
    OwGeo<GEO>::pixel(y, x, u, v, cls);
    float sv[NP];
    {
      const float* sp = dy + ((int64_t)n * O + ob + half) * SHW + (u << slw) + v;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) sv[kp] = sp[(int64_t)(2 * kp) * SHW];
    }
    float sv2[PAIR ? NP : 1];
    if constexpr (PAIR) {
      const float* sp = dy2 + ((int64_t)n * O + ob + half) * SHW + (u << slw) + v;
#pragma unroll
      for (int kp = 0; kp < NP; kp++) sv2[kp] = sp[(int64_t)(2 * kp) * SHW];
    }
    const int64_t obase = ((int64_t)n * C + c0) * HW + rem;
    float held[16];                                 // (wave 0 stores the tile: it requests the held gradient now)
#pragma unroll
    for (int r = 0; r < 16; r++)
      held[r] = (add && wave == 0) ? add[obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW] : 0.0f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int q = 0; q < NSL; q++) {
      const bool mine = cls == q;
      if (__ballot(mine) == 0ull) continue;
      const float* wq = Wl + (q * O + ob + half) * 32 + l31;
      float a[NP];
#pragma unroll
      for (int kp = 0; kp < NP; kp++) a[kp] = wq[2 * kp * 32];
#pragma unroll
      for (int kp = 0; kp < NP; kp++)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], mine ? sv[kp] : 0.0f, acc, 0, 0, 0);
    }
    if constexpr (PAIR) {
      const bool mine = cls == 0;
      if (__ballot(mine) != 0ull) {
        const float* wq = Wl + (NSL * O + ob + half) * 32 + l31;
        float a[NP];
#pragma unroll
        for (int kp = 0; kp < NP; kp++) a[kp] = wq[2 * kp * 32];
#pragma unroll
        for (int kp = 0; kp < NP; kp++)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], mine ? sv2[kp] : 0.0f, acc, 0, 0, 0);
      }
    }
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
#pragma unroll
      for (int r = 0; r < 16; r++)
        dx[obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * HW] = acc[r] + held[r];
    }
    __syncthreads();                                // red is reused by the next tile
  }
}

// ---------------------------------------------------------------------------------------
// The overwrite-mode input gradient of a 3x3 / stride 1 / pad 1 convolution on 2 x 2 IMAGES (ResNet layer4): all
// four input pixels keep a tap whose source is output pixel (0, 0) -- (y, x) -> (max(y - 1, 0), max(x - 1, 0)) --
// so dX[n][c][q] = sum_o dY[n][o][0][0] W[o][c][{4, 5, 7, 8}[q]]: a plain product of a 256 x O matrix with an
// O x 4C one whose result IS the NCHW tensor.  The general kernel above gives a tile 8 images x 4 pixels and runs
// four passes with three quarters of the lanes masked (12 us per launch for 0.13 GFLOP); here a workgroup owns 16
// images x 16 channels x the four pixels: v_mfma_f32_16x16x4_f32 with an image on the lane, one accumulator per
// pixel, the four waves split o and are summed through LDS, and a lane leaves with 16-byte stores (its four
// pixels of a channel are adjacent in memory).
// ---------------------------------------------------------------------------------------
template <int O>
__global__ __launch_bounds__(256) void k_dgrad_ow_img2(const float* __restrict__ dy, const float* __restrict__ w,
                                                       const float* __restrict__ add, float* __restrict__ dx, int C) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int OQ = O / 4, NK = OQ / 4;             // o per wave, MFMA k-steps per wave
  float* Wl = smem;                                 // [4 slices][O][16 channels]
  float* red = smem + 4 * O * 16;                   // [3 waves][16 values][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const DfTile wg = {(int)blockIdx.x, (int)blockIdx.y};          // (image tile, channel tile) in grid order
  const int n0 = wg.x * 16, c0 = wg.y * 16;
  const int ob = wave * OQ;
  // dY operand of this wave's o range: pixel (0, 0) of image n0 + n, channel ob + 4 p + kk
  float sv[NK];
  {
    const float* sp = dy + ((int64_t)(n0 + n) * O + ob + kk) * 4;
#pragma unroll
    for (int p = 0; p < NK; p++) sv[p] = sp[(int64_t)p * 16];
  }
  // the gradient the tensor already holds: wave 0 stores the tile and requests its pieces now
  f32x4 held[4];
  const int64_t obase = ((int64_t)(n0 + n) * C + c0 + 4 * kk) * 4;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    held[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (add && wave == 0) held[v] = *reinterpret_cast<const f32x4*>(add + obase + 4 * v);
  }
  // weight slices: (o, c) pairs, taps 4, 5, 7, 8 of each; 16 pairs per thread in two rounds of eight
  constexpr int PAIRS = O * 16 / 256;
#pragma unroll
  for (int i0 = 0; i0 < PAIRS; i0 += 8) {
    float wv[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = tid + 256 * (i0 + i), o = e >> 4, c = e & 15;
      const float* q = w + ((int64_t)o * C + c0 + c) * 9;
      load_taps_4578(q, wv[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = tid + 256 * (i0 + i), o = e >> 4, c = e & 15;
#pragma unroll
      for (int q = 0; q < 4; q++) Wl[(q * O + o) * 16 + c] = wv[i][q];
    }
  }
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NK; p++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(q * O + ob + 4 * p + kk) * 16 + n], sv[p], acc[q], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int v = 0; v < 4; v++) red[((wave - 1) * 16 + q * 4 + v) * 64 + lane] = acc[q][v];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int ww = 0; ww < 3; ww++)
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[q][v] += red[(ww * 16 + q * 4 + v) * 64 + lane];
    // register v of lane (kk, n) is channel c0 + 4 kk + v of image n0 + n; its four pixels are adjacent
#pragma unroll
    for (int v = 0; v < 4; v++) {
      f32x4 r = {acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
      r += held[v];
      *reinterpret_cast<f32x4*>(dx + obase + 4 * v) = r;
    }
  }
}

// The same idea on 4 x 4 IMAGES (ResNet layer3): input pixel (y, x) keeps the tap whose source is output pixel
// (max(y - 1, 0), max(x - 1, 0)) -- one of nine -- with one of four weight slices.  The general kernel's 32-pixel tile
// (two images) holds all four slices and runs four passes with most lanes masked; here a workgroup owns 16 images x
// 16 channels x the sixteen pixels: an image on the lane, one accumulator per pixel (16 MFMAs per k-step, none of
// them masked), the nine source pixels of a k-step are nine loads, the four waves split o and are summed through
// LDS, and a lane leaves with 16-byte stores (the pixels of a channel are adjacent in memory).
// GEO 0: 3x3 / stride 1 / pad 1, dY is 4 x 4 (nine sources).  GEO 1: 3x3 / stride 2 / pad 1, dY is 2 x 2: pixel (y, x)
// keeps tap (1 + (y & 1), 1 + (x & 1)) of source (y >> 1, x >> 1) (four sources); PAIR (GEO 1): the block's 1x1 /
// stride 2 shortcut adds sum_o dY2[o][y/2][x/2] W2[o][c] on the pixels with even (y, x) -- a fifth slice.
// GEO 2: the GEO 1 pattern on one 4 x 4 block (blockIdx.z) of an 8 x 8 image whose dY is 4 x 4 (layer3's first
// block: 256 workgroups where whole images would give 64).  The dY planes are requested in memory order and turned
// into the image-per-lane order through a wave-private LDS strip (see the operand section below).
// HS 2 (GEO 0 / 1): blockIdx.z selects image rows 2 z, 2 z + 1 -- eight accumulators per lane, twice the workgroups (the
// 128-channel layers gave 128 workgroups of four waves: half the CUs had nothing, the other half one wave per SIMD);
// a pixel's products are added in the same order either way.
constexpr int kImg4Strip = 16 * 260;     // floats of a wave's transposition strip (16 images x (256 + 4))
template <int O, int GEO, bool PAIR, int HS = 1>
__global__ __launch_bounds__(256) void k_dgrad_ow_img4(const float* __restrict__ dy, const float* __restrict__ w,
                                                       const float* __restrict__ add, float* __restrict__ dx, int C,
                                                       const float* __restrict__ dy2, const float* __restrict__ w2, int S) {
  static_assert(!PAIR || GEO >= 1, "the shortcut rides on the stride-2 forms");
  static_assert(HS == 1 || (HS == 2 && GEO != 2), "row halves: whole 4 x 4 images only");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int OQ = O / 4, NK = OQ / 4;             // o per wave, MFMA k-steps per wave
  constexpr int SW = GEO == 0 ? 3 : 2, NS = SW * SW;  // sources per image: SW x SW of a (GEO 0: 4, GEO 1: 2)-wide plane
  constexpr int NSL = PAIR ? 5 : 4;
  float* Wl = smem;                                 // [NSL slices][O][16 channels]; later the exchange area [3][16][64] f32x4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const DfTile wg = {(int)blockIdx.x, (int)blockIdx.y};          // (image tile, channel tile) in grid order
  const int n0 = wg.x * 16, c0 = wg.y * 16;
  const int ob = wave * OQ;
  // GEO 2: the image is S x S (S = 8), this workgroup's 4 x 4 block of it is (qy, qx); its sources are a 2 x 2 block of
  // the (S/2) x (S/2) dY plane
  const int Q = GEO == 2 ? (S >> 2) : 1;
  const int qy = GEO == 2 ? (int)blockIdx.z / Q : 0, qx = GEO == 2 ? (int)blockIdx.z % Q : 0;
  // dY operands of this wave's o range: the sources of image n0 + n, channel ob + 4 p + kk.  A lane is an IMAGE, so
  // read directly every lane of a request is a line of its own and the address unit takes them one per cycle (24-48
  // requests x 64 lines per wave); instead the wave requests its (16 images x channels) block in the order it lies in
  // memory -- a request = 1 KB = 8 lines -- and turns it through a wave-private LDS strip.
  float sv[NK][NS], sv2[PAIR ? NK : 1][4];
  {
    constexpr int WLF = (NSL * O * 16 > 3 * 16 * 64 * 4) ? NSL * O * 16 : 3 * 16 * 64 * 4;
    float* strip = smem + WLF + wave * kImg4Strip;
    if constexpr (GEO == 0) {
      // chunks of 16 channels: per image 16 x 64 bytes in a row; strip row = an image (256 floats + 4: bank spread)
      constexpr int NCH = OQ / 16;
      static_assert(OQ % 16 == 0, "whole chunks");
      f32x4 t[2][16];
#pragma unroll
      for (int i = 0; i < 16; i++)
        t[0][i] = *reinterpret_cast<const f32x4*>(dy + ((int64_t)(n0 + i) * O + ob) * 16 + lane * 4);
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        if (ch + 1 < NCH) {
#pragma unroll
          for (int i = 0; i < 16; i++)
            t[(ch + 1) & 1][i] = *reinterpret_cast<const f32x4*>(dy + ((int64_t)(n0 + i) * O + ob + 16 * (ch + 1)) * 16 + lane * 4);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) *reinterpret_cast<f32x4*>(strip + i * 260 + lane * 4) = t[ch & 1][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int pp = 0; pp < 4; pp++)
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(strip + n * 260 + (4 * pp + kk) * 16 + u * 4);
            sv[4 * ch + pp][u * 3] = r[0]; sv[4 * ch + pp][u * 3 + 1] = r[1]; sv[4 * ch + pp][u * 3 + 2] = r[2];
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    } else if constexpr (GEO == 2) {
      // 4 x 4 planes (S = 8) as in GEO 0: chunks of 16 channels, strip row = an image; this block's 2 x 2 sources are
      // two 8-byte pieces of the lane's plane
      constexpr int NCH = OQ / 16;
      static_assert(OQ % 16 == 0, "whole chunks");
      auto turn2 = [&](const float* src, float (&dst)[NK][4]) {
        f32x4 t[2][16];
#pragma unroll
        for (int i = 0; i < 16; i++)
          t[0][i] = *reinterpret_cast<const f32x4*>(src + ((int64_t)(n0 + i) * O + ob) * 16 + lane * 4);
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
          if (ch + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < 16; i++)
              t[(ch + 1) & 1][i] = *reinterpret_cast<const f32x4*>(src + ((int64_t)(n0 + i) * O + ob + 16 * (ch + 1)) * 16 + lane * 4);
          }
#pragma unroll
          for (int i = 0; i < 16; i++) *reinterpret_cast<f32x4*>(strip + i * 260 + lane * 4) = t[ch & 1][i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int pp = 0; pp < 4; pp++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              const f32x2 r = *reinterpret_cast<const f32x2*>(strip + n * 260 + (4 * pp + kk) * 16 + (2 * qy + u) * 4 + 2 * qx);
              dst[4 * ch + pp][u * 2] = r[0]; dst[4 * ch + pp][u * 2 + 1] = r[1];
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      };
      turn2(dy, sv);
      if constexpr (PAIR) turn2(dy2, sv2);
    } else {
      // the 2 x 2 planes: per image OQ x 16 bytes in a row; 64 / OQ images per request; strip row = an image (OQ + 1 float4)
      constexpr int IPL = 64 / OQ >= 1 ? 64 / OQ : 1, NL = 16 / IPL;
      static_assert(OQ <= 64 && 64 % OQ == 0, "an image's row fits a request");
      const int li = lane / OQ, lo = lane % OQ;
      auto turn = [&](const float* src, float (&dst)[NK][4]) {
        f32x4 t[NL];
#pragma unroll
        for (int i = 0; i < NL; i++)
          t[i] = *reinterpret_cast<const f32x4*>(src + ((int64_t)(n0 + i * IPL + li) * O + ob + lo) * 4);
#pragma unroll
        for (int i = 0; i < NL; i++) *reinterpret_cast<f32x4*>(strip + ((i * IPL + li) * (OQ + 1) + lo) * 4) = t[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < NK; p++) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(strip + (n * (OQ + 1) + 4 * p + kk) * 4);
          dst[p][0] = r[0]; dst[p][1] = r[1]; dst[p][2] = r[2]; dst[p][3] = r[3];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      };
      turn(dy, sv);
      if constexpr (PAIR) turn(dy2, sv2);
    }
  }
  // weight slices: (o, c) pairs, taps 4, 5, 7, 8 of each (+ the shortcut's weight)
  constexpr int PAIRS = O * 16 / 256, PB = PAIRS < 8 ? PAIRS : 8;
  static_assert(PAIRS % PB == 0, "weight staging rounds");
#pragma unroll
  for (int i0 = 0; i0 < PAIRS; i0 += PB) {
    float wv[PB][4], wv2[PB];
#pragma unroll
    for (int i = 0; i < PB; i++) {
      const int e = tid + 256 * (i0 + i), o = e >> 4, c = e & 15;
      const float* q = w + ((int64_t)o * C + c0 + c) * 9;
      load_taps_4578(q, wv[i]);
      wv2[i] = PAIR ? w2[(int64_t)o * C + c0 + c] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < PB; i++) {
      const int e = tid + 256 * (i0 + i), o = e >> 4, c = e & 15;
#pragma unroll
      for (int q = 0; q < 4; q++) Wl[(q * O + o) * 16 + c] = wv[i][q];
      if constexpr (PAIR) Wl[(4 * O + o) * 16 + c] = wv2[i];
    }
  }
  __syncthreads();
  constexpr int NY = HS == 2 ? 2 : 4, NQ = NY * 4;  // image rows / pixels of this workgroup
  const int hz = HS == 2 ? (int)blockIdx.z : 0;
  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto products = [&](auto y0c) {
    constexpr int Y0 = decltype(y0c)::value;        // first image row of the workgroup
#pragma unroll
    for (int p = 0; p < NK; p++) {
      float a[NSL];
#pragma unroll
      for (int q = 0; q < NSL; q++) a[q] = Wl[(q * O + ob + 4 * p + kk) * 16 + n];
#pragma unroll
      for (int yy = 0; yy < NY; yy++)
#pragma unroll
        for (int x = 0; x < 4; x++) {
          const int y = Y0 + yy;
          int u, v, cls;
          if (GEO == 0) { u = y >= 1 ? y - 1 : 0; v = x >= 1 ? x - 1 : 0; cls = (y >= 1 ? 2 : 0) + (x >= 1 ? 1 : 0); }
          else { u = y >> 1; v = x >> 1; cls = ((y & 1) << 1) + (x & 1); }
          acc[yy * 4 + x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cls], sv[p][u * SW + v], acc[yy * 4 + x], 0, 0, 0);
        }
    }
    if constexpr (PAIR) {
      // the shortcut's gradient: pixel (2u, 2v) <- dY2[u][v]
#pragma unroll
      for (int p = 0; p < NK; p++) {
        const float a5 = Wl[(4 * O + ob + 4 * p + kk) * 16 + n];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int v = 0; v < 2; v++)
            if (2 * u >= Y0 && 2 * u < Y0 + NY)
              acc[(2 * u - Y0) * 4 + 2 * v] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(a5, sv2[p][u * 2 + v], acc[(2 * u - Y0) * 4 + 2 * v], 0, 0, 0);
      }
    }
  };
  if (HS == 2 && hz == 1) products(std::integral_constant<int, 2>{});
  else products(std::integral_constant<int, 0>{});
  __syncthreads();                                  // the weight image is dead: it becomes the exchange area
  f32x4* ex = reinterpret_cast<f32x4*>(smem);
  // All four waves leave their partial tiles in LDS, then the 256 threads walk the workgroup's output in MEMORY order --
  // (image, channel, row): 16-byte pieces that are neighbours in dx -- and add the four partials of a piece in wave
  // order.  A lane of the accumulator is an image: stored straight from the registers every lane of a store instruction
  // is a line of its own (64 lines, and as many for the gradient the tensor already holds; one wave storing the sums
  // that way: 7.7 -> 7.3 us x 3, 10.3 -> 9.9, 10.9 -> 10.7 us in round 4).
#pragma unroll
  for (int q = 0; q < NQ; q++) ex[(wave * NQ + q) * 64 + lane] = acc[q];
  __syncthreads();
  const float* exf = reinterpret_cast<const float*>(smem);
  const int rowp = GEO == 2 ? S : 4, plane = GEO == 2 ? S * S : 16;
  const int64_t wbase = ((int64_t)n0 * C + c0) * plane + (GEO == 2 ? 4 * qy * S + 4 * qx : 0) + 8 * hz;
#pragma unroll
  for (int i = 0; i < 16 * 16 * NY / 256; i++) {
    const int f = tid + 256 * i;
    const int y = f % NY, ch = (f / NY) % 16, im = f / (NY * 16);
    const int src = ((ch >> 2) * 16 + im) * 4 + (ch & 3);         // float index of (lane, v) inside a tile's 64 x 4 block
    f32x4 r;
#pragma unroll
    for (int xx = 0; xx < 4; xx++) {
      const int q = y * 4 + xx;
      float t = exf[(0 * NQ + q) * 256 + src];
      t += exf[(1 * NQ + q) * 256 + src];
      t += exf[(2 * NQ + q) * 256 + src];
      t += exf[(3 * NQ + q) * 256 + src];
      r[xx] = t;
    }
    const int64_t at = wbase + ((int64_t)im * C + ch) * plane + rowp * y;
    if (add) r += *reinterpret_cast<const f32x4*>(add + at);
    *reinterpret_cast<f32x4*>(dx + at) = r;
  }
}

template <int O, int GEO, bool PAIR>
int launch_dgrad_img4(const float* dy, const float* w, const float* add, float* dx, int N, int C, const float* dy2,
                      const float* w2, int S = 4) {
  // whole images give (N / 16) x (C / 16) workgroups: where that leaves CUs without one, a workgroup takes two image rows
  const bool halves = GEO != 2 && (N / 16) * (C / 16) < num_cus();
  dim3 grid((unsigned)(N / 16), (unsigned)(C / 16), (unsigned)(GEO == 2 ? (S / 4) * (S / 4) : (halves ? 2 : 1)));
  const size_t base = std::max((size_t)(PAIR ? 5 : 4) * O * 16, (size_t)4 * 16 * 64 * 4) * sizeof(float);
  const size_t lds = base + (size_t)4 * kImg4Strip * sizeof(float);
  // blockIdx.y selects 16 input channels' weight columns (NSL slices x O), blockIdx.x 16 images of dY
  static bool attr = false;
  if (!attr) {
    DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_img4<O, GEO, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)std::max(base + (size_t)4 * kImg4Strip * sizeof(float), (size_t)65536)));
    if constexpr (GEO != 2)
      DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_img4<O, GEO, PAIR, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)std::max(base + (size_t)4 * kImg4Strip * sizeof(float), (size_t)65536)));
    attr = true;
  }
  if constexpr (GEO == 2) {
    DF_LAUNCH((k_dgrad_ow_img4<O, GEO, PAIR>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, dy2, w2, S);
  } else {
    if (halves) DF_LAUNCH((k_dgrad_ow_img4<O, GEO, PAIR, 2>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, dy2, w2, S);
    else DF_LAUNCH((k_dgrad_ow_img4<O, GEO, PAIR>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, dy2, w2, S);
  }
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// the BatchNorm descriptors of a launch whose grid is (gx, gy) workgroups of 32-channel tiles
int make_ow_bn(OwBn& b, const BnBwdIn* bi, const BnBwdIn* bi2, BnStatsOut* so, int gx, int gy, int C) {
  (void)gy; (void)C;
  b = OwBn{};
  if (bi) b.in = *bi;
  if (bi2) b.in2 = *bi2;
  if (!so) return DFHIP_OK;
  for (int j = 0; j < so->n; j++) { b.zt[j] = so->z[j]; b.mt[j] = so->mean[j]; }
  DF_CHECK_ARG(gx <= so->cap_rows && gx <= kBnRowsMax, "bn rows: room for %d rows needed, %d given (at most %d)", gx,
               so->cap_rows, kBnRowsMax);
  b.rows = so->rows;
  so->nrows = gx;
  return DFHIP_OK;
}

template <int O, int GEO, bool PAIR>
int launch_ow_s1(const float* dy, const float* w, const float* add, float* dx, int C, int H, int lw, int tiles, int pg,
                 const float* dy2, const float* w2, const BnBwdIn* bi, const BnBwdIn* bi2, BnStatsOut* so, bool dry_run = false) {
  OwBn bn;
  const bool any = bi || bi2 || so;
  if (!any) {
    if (dry_run) return DFHIP_OK;
    dim3 grid((unsigned)df_ceil_div(tiles, 4), (unsigned)(C / 32));
    bn = OwBn{};
    DF_LAUNCH((k_dgrad_ow_s1<O, GEO, PAIR, false, 0, 4>), grid, dim3(256), 0, stream(), dy, w, add, dx, C, H, lw, tiles, pg,
              dy2, w2, bn);
    DF_LAUNCH_CHECK();
#ifdef DFHIP_OW_STAMPS
    { static int calls = 0; char what[96]; snprintf(what, sizeof(what), "O %d GEO %d PAIR %d plain", O, GEO, (int)PAIR);
      ow_stamp_dump(what, (int)(grid.x * grid.y), 4, &calls); }
#endif
    return DFHIP_OK;
  }
  if constexpr (GEO == 2) {
    return -2;
  } else {
    // eight waves per workgroup for the 32-channel layers (half the rows, half the readers of the rows: see the kernel)
    constexpr bool CAN8 = O == 32 && !PAIR;
    constexpr int nw = CAN8 ? 8 : 4;
    dim3 grid((unsigned)df_ceil_div(tiles, nw), (unsigned)(C / 32));
    if (so) so->nrows = (int)grid.x;
    if (dry_run) return DFHIP_OK;
    int st = make_ow_bn(bn, bi, bi2, so, (int)grid.x, (int)grid.y, C);
    if (st != DFHIP_OK) return st;
#ifdef DFHIP_RIDER_PROBE
    {
      static int wgs = getenv("DFHIP_RIDER_WGS") ? atoi(getenv("DFHIP_RIDER_WGS")) : 0;
      static int iters = getenv("DFHIP_RIDER_ITERS") ? atoi(getenv("DFHIP_RIDER_ITERS")) : 0;
      static bool told = false;
      if (!told) {
        const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_rider_iters), &iters, sizeof(int));
        fprintf(stderr, "[rider] O %d GEO %d: %d extra workgroups x %d iterations (hipMemcpyToSymbol: %s), grid %u x %u\n", O, GEO, wgs,
                iters, hipGetErrorString(e), grid.x, grid.y);
        told = true;
      }
      if (wgs > 0) grid.x += (unsigned)(wgs / (int)grid.y);      // (DFHIP_RIDER_WGS in all, over the grid's rows)
    }
#endif
    const int nst = so ? so->n : 0;
    const bool bni = bi || bi2;
    // (of a PAIR only one gradient may be pending: the other one's descriptor is the identity, its z operand re-reads dY)
    if (bni && !bi) { bn.in = BnBwdIn{}; bn.in.z = dy; }
    if (bni && PAIR && !bi2) { bn.in2 = BnBwdIn{}; bn.in2.z = dy2; }
#define DF_OW(BNIV, NSTV, NWV)                                                                                          \
  DF_LAUNCH((k_dgrad_ow_s1<O, GEO, PAIR, BNIV, NSTV, NWV>), grid, dim3(64 * NWV), 0, stream(), dy, w, add, dx, C, H, lw, \
            tiles, pg, dy2, w2, bn)
#define DF_OW_NW(BNIV, NSTV) DF_OW(BNIV, NSTV, nw)
    if (bni) { if (nst == 0) DF_OW_NW(true, 0); else if (nst == 1) DF_OW_NW(true, 1); else DF_OW_NW(true, 2); }
    else { if (nst == 1) DF_OW_NW(false, 1); else DF_OW_NW(false, 2); }
#undef DF_OW_NW
#undef DF_OW
    DF_LAUNCH_CHECK();
#ifdef DFHIP_OW_STAMPS
    { static int calls[2][3] = {{0}}; char what[96];
      snprintf(what, sizeof(what), "O %d GEO %d PAIR %d pending %d rows-out %d", O, GEO, (int)PAIR, (int)bni, nst);
      ow_stamp_dump(what, (int)(grid.x * grid.y), nw, &calls[bni ? 1 : 0][nst]); }
#endif
    return DFHIP_OK;
  }
}

}  // namespace

namespace dfhip {

int run_dgrad_ow(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H, int W, int O, int k,
                 int stride, int pad, const BnBwdIn* bi, BnStatsOut* so, bool dry_run) {
  const int oh = (H + 2 * pad - k) / stride + 1, ow = (W + 2 * pad - k) / stride + 1;
  const bool bnform = bi || so;
  int lw = 0;
  while ((1 << lw) < W) lw++;
  // (a tile is 32 consecutive pixels of the flattened (n, y, x) space; every lane decodes its own pixel)
  const int geo = (k == 3 && stride == 1 && pad == 1) ? 0 : (k == 3 && stride == 2 && pad == 1) ? 1
                  : (k == 1 && stride == 2 && pad == 0) ? 2 : (k == 5 && stride == 1 && pad == 2) ? 3 : -1;
  const bool lean_ok = geo >= 0 && C % 32 == 0 && (1 << lw) == W && H >= 2 && W >= 2 &&
                       ((int64_t)N * H * W) % 32 == 0 &&
                       (geo == 0 || geo == 3 || (H % 2 == 0 && oh * 2 == H && ow * 2 == W));
  if (geo == 3) {
    // 5x5 / stride 1 / pad 2 (the CNN workload's conv2): nine weight slices in dynamic LDS, one workgroup per CU walks the tiles
    if (!(lean_ok && (O == 32 || O == 64) && W >= 4 && H >= 3)) return -1;
    if (bnform) return -2;
    if (dry_run) return DFHIP_OK;
    const int tiles = (int)((int64_t)N * H * W / 32);
    // a wave owns a 32-pixel tile outright (k_dgrad_ow_s1), four tiles per workgroup, the nine slices in dynamic LDS
    dim3 grid((unsigned)df_ceil_div(tiles, 4), (unsigned)(C / 32));
    const size_t lds = (size_t)9 * O * 32 * sizeof(float);
    OwBn bn = OwBn{};
#define DF_K5(OV)                                                                                             \
  do {                                                                                                        \
    static bool attr = false;                                                                                 \
    if (!attr) {                                                                                              \
      DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_s1<OV, 3, false, false, 0, 4>,                       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)65536))); \
      attr = true;                                                                                            \
    }                                                                                                         \
    DF_LAUNCH((k_dgrad_ow_s1<OV, 3, false, false, 0, 4>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, H, lw, tiles, 0, \
              (const float*)nullptr, (const float*)nullptr, bn);                                               \
  } while (0)
    if (O == 32) DF_K5(32); else DF_K5(64);
#undef DF_K5
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  const int tiles = lean_ok ? (int)((int64_t)N * H * W / 32) : 0;
  // stride 2: parity-grouped tiles when whole 128-pixel blocks tile the tensor
  const int pg = (geo == 1 && ((int64_t)N * H * W) % 128 == 0 && W <= 64) ? 1 : 0;
  if (geo == 0 && H == 2 && W == 2 && N % 16 == 0 && C % 16 == 0 && (O == 128 || O == 256 || O == 512)) {
    if (bnform) return -2;
    if (dry_run) return DFHIP_OK;
    // 2 x 2 images: a plain product (k_dgrad_ow_img2)
    dim3 grid((unsigned)(N / 16), (unsigned)(C / 16));
    const size_t lds = ((size_t)4 * O * 16 + 3 * 16 * 64) * sizeof(float);
    // blockIdx.y selects 16 input channels' weight columns (all O, four taps), blockIdx.x 16 images of dY
#define DF_I2(OV)                                                                                          \
  do {                                                                                                     \
    static bool attr = false;                                                                              \
    if (!attr) {                                                                                           \
      DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_img2<OV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)std::max(lds, (size_t)65536)));                                      \
      attr = true;                                                                                         \
    }                                                                                                      \
    DF_LAUNCH(k_dgrad_ow_img2<OV>, grid, dim3(256), lds, stream(), dy, w, add, dx, C);                \
  } while (0)
    if (O == 128) DF_I2(128); else if (O == 256) DF_I2(256); else DF_I2(512);
#undef DF_I2
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  if ((geo == 0 || geo == 1) && H == 4 && W == 4 && N % 16 == 0 && C % 16 == 0 &&
      ((geo == 0 && (O == 64 || O == 128)) || (geo == 1 && (O == 128 || O == 256)))) {
    if (bnform) return -2;
    if (dry_run) return DFHIP_OK;
    // 4 x 4 images: 16 images x 16 channels x 16 pixels per workgroup, no masked lanes (k_dgrad_ow_img4)
    const float* none = nullptr;
    if (geo == 0) return O == 64 ? launch_dgrad_img4<64, 0, false>(dy, w, add, dx, N, C, none, none)
                                 : launch_dgrad_img4<128, 0, false>(dy, w, add, dx, N, C, none, none);
    return O == 128 ? launch_dgrad_img4<128, 1, false>(dy, w, add, dx, N, C, none, none)
                    : launch_dgrad_img4<256, 1, false>(dy, w, add, dx, N, C, none, none);
  }
  if (lean_ok && geo == 0 && O == 128 && !bnform && (W == 16 || W == 8) && N % (32 / W) == 0 &&
      (int)((int64_t)N * H * W / 32) % H == 0) {
    // 128 channels of dY on 8- and 16-wide planes (the CNN workload's conv3): a wave owns a 32-pixel tile outright, as on the
    // 32- and 64-channel layers (k_dgrad_ow_s1), with the four slices in dynamic LDS -- 15.3 -> 10.4 us against the split-K form below
    if (dry_run) return DFHIP_OK;
    dim3 grid((unsigned)df_ceil_div(tiles, 4), (unsigned)(C / 32));
    const size_t lds = (size_t)4 * 128 * 32 * sizeof(float);
    static bool attr = false;
    if (!attr) {
      DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_s1<128, 0, false, false, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      attr = true;
    }
    OwBn bn = OwBn{};
    DF_LAUNCH((k_dgrad_ow_s1<128, 0, false, false, 0, 4>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, H, lw, tiles, 1,
              (const float*)nullptr, (const float*)nullptr, bn);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  if (lean_ok && (O == 128 || O == 256)) {
    if (bnform) return -2;
    if (dry_run) return DFHIP_OK;
    // about two workgroups per CU (one at 256 channels: 140 KB of LDS); each walks tiles / grid.x tiles
    // (1x1: the slice is small, one tile per workgroup)
    const int want = geo == 2 ? tiles : std::max(1, num_cus() * (O == 128 ? 2 : 1) / (C / 32));
    dim3 grid((unsigned)std::min(tiles, want), (unsigned)(C / 32));
    const size_t lds = ((size_t)(geo == 2 ? 1 : 4) * O * 32 + 3 * 16 * 64) * sizeof(float);
#define DF_KS(OV, GV)                                                                                         \
  do {                                                                                                        \
    static bool attr = false;                                                                                 \
    if (!attr) {                                                                                              \
      DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_s1_ks<OV, GV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)std::max(lds, (size_t)65536)));                                         \
      attr = true;                                                                                            \
    }                                                                                                         \
    DF_LAUNCH((k_dgrad_ow_s1_ks<OV, GV>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, H, lw, tiles, pg, \
              (const float*)nullptr, (const float*)nullptr);                                                    \
  } while (0)
    if (O == 128) { if (geo == 0) DF_KS(128, 0); else if (geo == 1) DF_KS(128, 1); else DF_KS(128, 2); }
    else { if (geo == 0) DF_KS(256, 0); else if (geo == 1) DF_KS(256, 1); else DF_KS(256, 2); }
#undef DF_KS
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  if (lean_ok && (O == 32 || O == 64)) {
    if (bnform && geo == 2) return -2;
    const float* none = nullptr;
    // stride 1, 16- and 8-wide images: tiles that never mix image row 0 with later rows (OwGeo::decode, `pg`)
    const bool row0 = geo == 0 && (W == 16 || W == 8) && N % (32 / W) == 0 && tiles % H == 0;
    const int pgs = geo == 0 ? (row0 ? 1 : 0) : pg;
#define DF_A(OV, GV) return launch_ow_s1<OV, GV, false>(dy, w, add, dx, C, H, lw, tiles, pgs, none, none, bi, nullptr, so, dry_run)
    if (O == 32) { if (geo == 0) DF_A(32, 0); else if (geo == 1) DF_A(32, 1); else DF_A(32, 2); }
    else { if (geo == 0) DF_A(64, 0); else if (geo == 1) DF_A(64, 1); else DF_A(64, 2); }
#undef DF_A
  }
  return -1;
}

int run_dgrad_ow_pair(const float* dy, const float* w, const float* dy2, const float* w2, const float* add, float* dx,
                      int N, int C, int H, int W, int O, const BnBwdIn* bi, const BnBwdIn* bi2, BnStatsOut* so,
                      bool dry_run) {
  const int oh = (H + 2 - 3) / 2 + 1, ow = (W + 2 - 3) / 2 + 1;
  const bool bnform = bi || bi2 || so;
  int lw = 0;
  while ((1 << lw) < W) lw++;
  if (H == 4 && W == 4 && N % 16 == 0 && C % 16 == 0 && (O == 128 || O == 256)) {
    // 4 x 4 images (layer4's first block): the image-per-lane kernel with the shortcut as a fifth slice
    if (bnform) return -2;
    if (dry_run) return DFHIP_OK;
    return O == 128 ? launch_dgrad_img4<128, 1, true>(dy, w, add, dx, N, C, dy2, w2)
                    : launch_dgrad_img4<256, 1, true>(dy, w, add, dx, N, C, dy2, w2);
  }
  if (H == 8 && W == 8 && N % 16 == 0 && C % 16 == 0 && (O == 64 || O == 128)) {
    // 8 x 8 images (layer3's first block): four 4 x 4 blocks per image, each the image-per-lane kernel of the case above
    if (bnform) return -2;
    if (dry_run) return DFHIP_OK;
    return O == 64 ? launch_dgrad_img4<64, 2, true>(dy, w, add, dx, N, C, dy2, w2, 8)
                   : launch_dgrad_img4<128, 2, true>(dy, w, add, dx, N, C, dy2, w2, 8);
  }
  const bool ok = C % 32 == 0 && (1 << lw) == W && H >= 2 && W >= 2 && ((int64_t)N * H * W) % 32 == 0 && H % 2 == 0 &&
                  oh * 2 == H && ow * 2 == W && (O == 32 || O == 64 || O == 128);   // (256: five slices do not fit the LDS)
  if (!ok) return -1;
  if (O == 128 && bnform) return -2;
  const int tiles = (int)((int64_t)N * H * W / 32);
  const int pg = (((int64_t)N * H * W) % 128 == 0 && W <= 64) ? 1 : 0;
  if (O == 128 && dry_run) return DFHIP_OK;
  if (O == 128) {
    // (five slices = 92 KB of LDS: ONE workgroup per CU, which walks twice the tiles of the unpaired launch)
    const int want = std::max(1, num_cus() / (C / 32));
    dim3 grid((unsigned)std::min(tiles, want), (unsigned)(C / 32));
    const size_t lds = ((size_t)5 * O * 32 + 3 * 16 * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
      DF_HIP(hipFuncSetAttribute((const void*)k_dgrad_ow_s1_ks<128, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)std::max(lds, (size_t)65536)));
      attr = true;
    }
    DF_LAUNCH((k_dgrad_ow_s1_ks<128, 1, true>), grid, dim3(256), lds, stream(), dy, w, add, dx, C, H, lw, tiles, pg, dy2, w2);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  if (O == 32) return launch_ow_s1<32, 1, true>(dy, w, add, dx, C, H, lw, tiles, pg, dy2, w2, bi, bi2, so, dry_run);
  return launch_ow_s1<64, 1, true>(dy, w, add, dx, C, H, lw, tiles, pg, dy2, w2, bi, bi2, so, dry_run);
}

}  // namespace dfhip
