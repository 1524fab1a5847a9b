// The stem of the CIFAR ResNet -- pool(relu(bn(conv(x)))) on a 3-channel 32 x 32 input (test/ResNet_CIFAR10_cuda.py:
// 84-87; reference path nn/functional.py:316-374, nn/modules/batchnorm.py:30-55) -- WITHOUT its full-resolution
// tensor.  The convolution output z (N x 32 x 32 x 32: 33.5 MB at batch 256, the largest tensor of the network) was
// written once and read three times (pooling forward, statistics pass and weight gradient backward): 134 MB of the
// 185 MB the four stem kernels moved.  It no longer exists in memory:
//
//   forward   BatchNorm is a monotone map per channel (non-decreasing for gamma >= 0, non-increasing for gamma < 0:
//             every rounded step of ((v - mean) * rsd) * gamma + beta is monotone) and so is ReLU, hence
//                 max over the 2 x 2 window of relu(bn(z_i)) == relu(bn(max_i z_i))        (gamma >= 0)
//                                                              == relu(bn(min_i z_i))        (gamma <  0)
//             bit for bit.  k_stem_fwd_sel runs the convolution (the MFMA sequence of k_conv_rows3: image tile with its
//             padding in LDS, a wave per row) on row PAIRS, keeps the BatchNorm partials of z as before and writes
//             only the selected extreme of every window (8.4 MB); stem_pool_apply (dfhip_fused.hip) combines the
//             partials and applies bn + relu to it.
//   backward  the gradient goes to EVERY window position whose relu(bn(z_i)) equals the pooled value (tensor.py:779-791)
//             and through the ReLU where output == input (tensor.py:872-877): that needs every z_i, so both backward
//             kernels REBUILD z from the 3 MB input with the same MFMA sequence (operands in the same k order: the
//             same bits) instead of reading it -- k_stem_bwd<.., false> for the sums of dy and dy * xhat (dbeta,
//             dgamma), k_stem_bwd<.., true> for the weight gradient with d loss / d z formed per element exactly as
//             the separate operators form it.  The convolution is computed TRANSPOSED there (pixels on the
//             accumulator rows, channels on the lanes): a lane is an output channel, so the channel's constants are
//             registers of the lane and the rebuilt gradient is already the A operand of the weight-gradient MFMA.
//
// Shapes: 3x3 / stride 1 / pad 1, 1-3 input channels, <= 32 output channels, 32-wide images of even height <= 32, no
// bias, an input that needs no gradient.  Everything else stays on dfhip_conv2d_forward_stats +
// dfhip_bn_relu_maxpool2_forward_partials / dfhip_stem_backward.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "dfhip_internal.h"
#include "dfhip_stem_combine.h"

using namespace dfhip;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// diagnostic build only (-DDFHIP_STEM2_STAMPS): per-wave cycle stamps, dumped by the entry points every 20th call
#ifdef DFHIP_STEM2_STAMPS
__device__ unsigned long long g_st2[512 * 8 * 8];
#define ST2(i)                                                                                          \
  do {                                                                                                  \
    if ((threadIdx.x & 63) == 0) g_st2[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] =                \
        (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();                     \
  } while (0)
static void st2_dump(const char* what, int gx) {
  static int calls_f = 0, calls_b = 0;                 // (forward and backward alternate: a counter each)
  int& calls = what[0] == 'f' ? calls_f : calls_b;
  if (++calls % 20) return;
  std::vector<unsigned long long> h((size_t)gx * 8 * 8);
  (void)hipStreamSynchronize(stream());
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_st2), h.size() * sizeof(unsigned long long));
  double seg[6] = {0};
  unsigned long long r0 = ~0ull, r1 = 0;
  const size_t nw = (size_t)gx * 8;
  for (size_t b = 0; b < nw; b++) {
    const unsigned long long* r = &h[b * 8];
    for (int k = 1; k <= 5; k++) seg[k] += (double)(r[k] - r[k - 1]);
    r0 = std::min(r0, r[6]); r1 = std::max(r1, r[7]);
  }
  fprintf(stderr, "[stem2-stamp] %s grid %d: span %.2f us; mean cycles per wave between stamps: %.0f %.0f %.0f %.0f %.0f\n", what, gx,
          (double)(r1 - r0) / 100.0, seg[1] / nw, seg[2] / nw, seg[3] / nw, seg[4] / nw, seg[5] / nw);
}
#else
#define ST2(i) do { } while (0)
static void st2_dump(const char*, int) {}
#endif

// LDS tile of a channel: 32 + KS - 1 rows of 40 floats (KS / 2 zero rows above and below), image column c at 4 + c,
// KS / 2 zero columns either side
constexpr int kPitch = 40;
template <int KS> struct TileGeo {
  static constexpr int PAD = KS / 2, ROWS = 32 + KS - 1, CH = ROWS * kPitch;
  // LDS index of (channel, tap) kk of an output pixel at row 0, column 0 of the image
  static __device__ __host__ constexpr int koff(int kk) {
    return (kk / (KS * KS)) * CH + ((kk % (KS * KS)) / KS) * kPitch + (kk % KS) + 4 - PAD;
  }
};

template <int XM>
__device__ __forceinline__ float swz(float v) {    // the value of lane (l ^ XM) of the same 32-lane group
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XM << 10) | 0x1f));
}

// max without the canonicalising v_max x, x pair the compiler puts in front of fmaxf (NaNs are not an issue here)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// the value of the horizontal neighbour (lane ^ 1): a DPP quad permutation, no LDS round trip
__device__ __forceinline__ float lane_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
}
// d loss / d bn-output at one position of a pooling window (the three operators' tie rules, tensor.py:779-791 and
// 872-877): the gradient arrives where relu(bo) equals the pooled value and passes the ReLU where relu(bo) == bo, i.e.
// where bo >= 0 -- and then relu(bo) is bo itself (or +0 for -0, which compares equal): two comparisons decide it
__device__ __forceinline__ float stem_dy2(float bo, float pooled, float g) {
  return (bo >= 0.0f && pooled == bo) ? g : 0.0f;
}

// the image tile: 16-byte pieces requested by every thread, committed with the zero rows above / below the image
template <int C, int KS, int QR>
__device__ __forceinline__ void tile_request(const float* __restrict__ xn, int H, int tid, f32x4 (&v)[QR]) {
  typedef TileGeo<KS> G;
  constexpr int NQ = C * G::ROWS * 8;
#pragma unroll
  for (int q0 = 0; q0 < QR; q0++) {
    const int q = min(tid + 512 * q0, NQ - 1);
    const int pc = q & 7, t = q >> 3, lr = t % G::ROWS, c = t / G::ROWS;
    const int row = min(max(lr - G::PAD, 0), H - 1);                   // (clamped: the commit zeroes what is outside)
    v[q0] = *reinterpret_cast<const f32x4*>(xn + ((int64_t)c * H + row) * 32 + pc * 4);
  }
}
template <int C, int KS, int QR>
__device__ __forceinline__ void tile_commit(float* __restrict__ Xl, int H, int tid, const f32x4 (&v)[QR]) {
  typedef TileGeo<KS> G;
  constexpr int NQ = C * G::ROWS * 8;
#pragma unroll
  for (int q0 = 0; q0 < QR; q0++) {
    const int q = tid + 512 * q0;
    const int pc = q & 7, t = q >> 3, lr = t % G::ROWS, c = t / G::ROWS;
    const bool in = lr >= G::PAD && lr < H + G::PAD;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (q < NQ) *reinterpret_cast<f32x4*>(&Xl[c * G::CH + lr * kPitch + 4 + pc * 4]) = in ? v[q0] : z;
  }
}
template <int C, int KS>
__device__ __forceinline__ void tile_side_columns(float* __restrict__ Xl, int tid) {
  typedef TileGeo<KS> G;
  constexpr int SIDE = 2 * G::PAD;                   // zero columns per tile row (never written again)
  for (int e = tid; e < C * G::ROWS * SIDE; e += 512) {
    const int t = e / SIDE, sc = e % SIDE;
    const int col = sc < G::PAD ? 4 - G::PAD + sc : 36 + (sc - G::PAD);
    Xl[(t / G::ROWS) * G::CH + (t % G::ROWS) * kPitch + col] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------
// forward: convolution on row pairs, BatchNorm partials of z, the selected extreme of every 2 x 2 window
// ---------------------------------------------------------------------------------------
template <int C, int KS>
__global__ __launch_bounds__(512) void k_stem_fwd_sel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      float* __restrict__ zsel, float* __restrict__ partials, int N,
                                                      int H, int O) {
  typedef TileGeo<KS> G;
  constexpr int KK = C * KS * KS, NP = (KK + 1) / 2;
  constexpr bool AREG = NP <= 16;                   // the weight operands of a lane stay in registers (3x3); else LDS
  constexpr int NQ = C * G::ROWS * 8, QR = (NQ + 511) / 512;
  constexpr int NWR = (32 * KK + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  __shared__ float Wl[32 * KK + 32];
  __shared__ __attribute__((aligned(16))) float Xl[C * G::CH];
  __shared__ float sred[8][2][32], sshift[8][2][16], scount[8], Gl[32], Bl[32];
  __shared__ __attribute__((aligned(16))) float Tz[8][32 * 16];     // a wave's pooled row: [channel][16 windows], stored as 16-byte pieces
  ST2(6); ST2(0);
  int item = blockIdx.x;
  f32x4 tv[QR];
  if (item < N) tile_request<C, KS, QR>(x + (int64_t)item * C * H * 32, H, tid, tv);
  {
    float wv[NWR];
    const int nw = O * KK;
#pragma unroll
    for (int i0 = 0; i0 < NWR; i0++) {
      const int i = tid + 512 * i0;
      const float t = w[min(i, nw - 1)];
      wv[i0] = i < nw ? t : 0.0f;
    }
#pragma unroll
    for (int i0 = 0; i0 < NWR; i0++) {
      const int i = tid + 512 * i0;
      if (i < 32 * KK) Wl[i] = wv[i0];
    }
    if (tid < 32) Wl[32 * KK + tid] = 0.0f;         // the weight a k slot past K reads (an odd K: the last MFMA's second slot)
    if (tid >= 480) {
      const int dl = tid - 480;
      float t = 1.0f, bb = 0.0f;
      if (gamma) t = gamma[min(dl, O - 1)];
      if (bias) bb = bias[min(dl, O - 1)];
      Gl[dl] = t < 0.0f ? -1.0f : 1.0f;             // which extreme of a window survives bn + relu + max
      Bl[dl] = dl < O ? bb : 0.0f;
    }
    tile_side_columns<C, KS>(Xl, tid);
  }
  if (item < N) tile_commit<C, KS, QR>(Xl, H, tid, tv);
  __syncthreads();
  ST2(1);
  // weight operand of MFMA p: k index 2 p (lower half-wave) / 2 p + 1 (upper); a k past K is a zero weight
  auto wsrc = [&](int p) { const int kk = 2 * p + half; return kk < KK ? l31 * KK + kk : 32 * KK + l31; };
  float a[AREG ? NP : 1];
  if constexpr (AREG) {
#pragma unroll
    for (int p = 0; p < NP; p++) a[p] = Wl[wsrc(p)];
  }
  // LDS index of (channel, tap) 2 p + half for output row 0: a compile-time constant per half-wave, selected where it is
  // used (index registers would push the kernel over its 256: it spilled)
  auto koff = [&](int p) {
    const int k0 = 2 * p < KK ? 2 * p : 0, k1 = 2 * p + 1 < KK ? 2 * p + 1 : 0;
    return (half ? G::koff(k1) : G::koff(k0)) + l31;
  };
  constexpr int PCH = AREG ? NP : 4;                // MFMA pairs between two scheduling fences (register pressure, 5x5)
  // bit rr: the channel of accumulator register rr has a negative gamma (its windows keep their minimum)
  unsigned negmask = 0;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) negmask |= (Gl[(rr & 3) + 8 * (rr >> 2) + 4 * half] < 0.0f ? 1u : 0u) << rr;
  float st_s[16], st_1[16], st_2[16];               // shift, sum of (z - shift), sum of (z - shift)^2 per lane
  float st_n = 0.0f;
  bool first = true;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) { st_s[rr] = 0.0f; st_1[rr] = 0.0f; st_2[rr] = 0.0f; }
  const int pairs = H >> 1;
  ST2(2);
  while (item < N) {
    const int n = item;
    const int next = item + gridDim.x;
    if (next < N) tile_request<C, KS, QR>(x + (int64_t)next * C * H * 32, H, tid, tv);
    for (int pr = wave; pr < pairs; pr += 8) {
      f32x16 acc0, acc1;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {              // (the convolution's bias of the register's channel, or 0)
        const float bb = Bl[(rr & 3) + 8 * (rr >> 2) + 4 * half];
        acc0[rr] = bb; acc1[rr] = bb;
      }
#pragma unroll
      for (int p = 0; p < NP; p++) {
        float ap;
        if constexpr (AREG) ap = a[p]; else ap = Wl[wsrc(p)];
        const float b0 = Xl[koff(p) + (2 * pr) * kPitch], b1 = Xl[koff(p) + (2 * pr + 1) * kPitch];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap, b1, acc1, 0, 0, 0);
        if (p % PCH == PCH - 1) __builtin_amdgcn_sched_barrier(0);     // (keeps the operand reads of later pairs behind)
      }
      if (partials) {
        if (first) {
#pragma unroll
          for (int rr = 0; rr < 16; rr++)
            st_s[rr] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc0[rr]), 0));   // lane 0 of the half
          first = false;
        }
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          const float d0 = acc0[rr] - st_s[rr], d1 = acc1[rr] - st_s[rr];
          st_1[rr] += d0;
          st_2[rr] = fmaf(d0, d0, st_2[rr]);
          st_1[rr] += d1;
          st_2[rr] = fmaf(d1, d1, st_2[rr]);
        }
        st_n += 64.0f;
      }
      // the window's extreme: vertical in registers, horizontal from the neighbouring lane (DPP); a flipped sign bit
      // turns a minimum into a maximum (exact).  Both lanes of a pair hold the window's value and store it to the same
      // address (no exec juggling per store); the scratch tensor has 32 channel planes whatever O is.
      float m[16];
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        const unsigned sb = ((negmask >> rr) & 1u) << 31;
        m[rr] = vmax(__uint_as_float(__float_as_uint(acc0[rr]) ^ sb), __uint_as_float(__float_as_uint(acc1[rr]) ^ sb));
      }
#pragma unroll
      for (int rr = 0; rr < 16; rr++) m[rr] = vmax(m[rr], lane_xor1(m[rr]));
      // through the wave's LDS strip: two 16-byte stores per lane (16 channel segments of 64 bytes each) instead of sixteen
      // 4-byte ones (two segments each) -- the store instructions of the eight waves were the busiest unit of the loop
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        const unsigned sb = ((negmask >> rr) & 1u) << 31;
        Tz[wave][((rr & 3) + 8 * (rr >> 2) + 4 * half) * 16 + (l31 >> 1)] = __uint_as_float(__float_as_uint(m[rr]) ^ sb);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        const int chn = 16 * h2 + (lane >> 2);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(&Tz[wave][chn * 16 + (lane & 3) * 4]);
        *reinterpret_cast<f32x4*>(zsel + (((int64_t)n * 32 + chn) * pairs + pr) * 16 + (lane & 3) * 4) = v4;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    item = next;
    if (item < N) {
      __syncthreads();                               // every wave has read its rows of the tile
      tile_commit<C, KS, QR>(Xl, H, tid, tv);
      __syncthreads();
    }
  }
  ST2(3);
  if (partials) {
    // 32 sums per lane -> one per lane (k_conv_rows3's butterfly), the eight waves through LDS, one shift per channel
    float v[32];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) { v[rr] = st_1[rr]; v[16 + rr] = st_2[rr]; }
#define DF_HALVE(XM, CNT)                                                              \
    {                                                                                  \
      const bool up = (l31 & XM) != 0;                                                 \
      _Pragma("unroll") for (int i = 0; i < CNT / 2; i++) {                            \
        const float lo = v[i], hi = v[i + CNT / 2];                                    \
        v[i] = (up ? hi : lo) + swz<XM>(up ? lo : hi);                                 \
      }                                                                                \
    }
    DF_HALVE(16, 32) DF_HALVE(8, 16) DF_HALVE(4, 8) DF_HALVE(2, 4) DF_HALVE(1, 2)
#undef DF_HALVE
    sred[wave][half][l31] = v[0];
    if (l31 == 0) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) sshift[wave][half][rr] = st_s[rr];
      if (half == 0) scount[wave] = st_n;
    }
    __syncthreads();
    if (tid < 32 && tid < O) {
      const int hh = (tid >> 2) & 1, rr = (tid & 3) + 4 * (tid >> 3);
      // (wave 0 has a row pair whenever the workgroup had an image; a workgroup without one writes a zero count)
      const float S = sshift[0][hh][rr];
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
      float* o = partials + ((int64_t)blockIdx.x * O + tid) * 3;
      o[0] = n; o[1] = S + mu_s; o[2] = n > 0.0f ? A2 - A1 * mu_s : 0.0f;
    }
  }
  ST2(4); ST2(5); ST2(7);
}

// ---------------------------------------------------------------------------------------
// backward: z rebuilt TRANSPOSED (lane = output channel, accumulator rows = pixels of an image row).
//   WGRAD == false: slice sums of dy and dy * xhat per channel -> partial[O][S][2], S = gridDim.x
//   WGRAD == true : the weight gradient; partial holds the S slices of the pass above; slabs [gridDim.x][O][K]
// A step is a PAIR of image rows (they share one row of the pooled planes y and g); a wave takes steps wave,
// wave + 8, ... of an image.  y / g rows are requested coalesced (a wave load = 16 channels x 64 bytes) and turned
// into the lane-per-channel order through the wave's own LDS area.
// ---------------------------------------------------------------------------------------
struct StemB {
  const float* x;
  const float* w;
  const float* y;
  const float* g;
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  const float* bias;   // the convolution's bias (or nullptr)
  float* partial;      // [O][S][2]
  float* slabs;        // WGRAD: [gridDim.x][O][K + 1] (the last column: the bias gradient)
  float* dgamma;
  float* dbeta;
  int N, H, O, S;
  float eps;
};

// MODE 0: statistics pass, 1: weight gradient from the statistics, 2: BOTH in one pass (3x3) -- the weight gradient is
// linear in the two sums the statistics pass exists for:
//     dW = gamma rsd (A - k1 S - k2 B),   A = sum dy (x) patch,  B = sum xhat (x) patch,  S = sum patch,
//     k1 = sum dy / m,  k2 = sum dy xhat / m
// so one kernel accumulates A and B (two MFMA streams on the same rebuilt z), S, sum dy and sum dy xhat per workgroup and
// k_stem_combine forms dW, dgamma, dbeta and the bias gradient from the totals: one convolution rebuild instead of two,
// one pass over y and g instead of two, one launch fewer.
template <int C, int KS, int MODE>
__global__ __launch_bounds__(512) void k_stem_bwd(StemB g) {
  constexpr bool WGRAD = MODE == 1, ONE = MODE == 2;
  typedef TileGeo<KS> G;
  constexpr int KK = C * KS * KS, NP = (KK + 1) / 2;
  constexpr bool AREG = NP <= 16;                     // weight operands in registers (3x3) or read from LDS per MFMA
  constexpr int KT = (KK + 31) / 32;                  // column tiles of the weight-gradient tile (32 (channel, tap) each)
  constexpr int NQ = C * G::ROWS * 8, QR = (NQ + 511) / 512;
  constexpr int YP = 20;                              // LDS pitch of a channel's 16 pooled pixels (bank spread)
  constexpr int WSLOT = 2 * 32 * YP;                  // a wave's private transposition area: y and g
  extern __shared__ __attribute__((aligned(16))) float smem_sb[];
  float* Xl = smem_sb;                                // [C][G::CH]
  float* Wl = Xl + C * G::CH;                         // [32][KK] + 32 zeros
  float* area = Wl + 32 * KK + 32 + ((32 * KK) & 1);  // 8 wave areas; later the cross-wave reduction [8][16][64] / sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  float* Ys = area + wave * WSLOT;
  float* Gs = Ys + 32 * YP;
  ST2(6); ST2(0);
  const int H = g.H, O = g.O;
  const int o = min(l31, O - 1);                      // (a lane past the last channel repeats it: its results are not stored)
  const int steps = H >> 1;
  // the channel's constants, requested up front
  const float c_mean = g.mean[o], c_var = g.var[o];
  const float gw = g.gamma ? g.gamma[o] : 1.0f, gb = g.beta ? g.beta[o] : 0.0f;
  // WGRAD: the S slices of (sum dy, sum dy xhat) of channel tid / 16: lanes 0..15 of a DPP row take every 16th slice
  float sdy = 0.0f, sdyx = 0.0f;
  if constexpr (WGRAD) {
    // (unconditional loads at a clamped slice, sixteen in flight: a load under `if (u < S)` is a round trip of its own)
    const float* pr = g.partial + (int64_t)min(tid >> 4, O - 1) * g.S * 2;
    f32x2 pv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pv[i] = *reinterpret_cast<const f32x2*>(pr + 2 * min((tid & 15) + 16 * i, g.S - 1));
#pragma unroll
    for (int i = 0; i < 16; i++)
      if ((tid & 15) + 16 * i < g.S) { sdy += pv[i][0]; sdyx += pv[i][1]; }
  }
  // pooled rows of a step: 32 channels x 16 pixels each, two wave loads per plane; the requests of BOTH steps of an
  // image leave before its tile is committed (issued inside the step they were a round trip in front of every step's
  // first vector instruction)
  f32x4 ry[2][2], rg[2][2];
  auto request_rows = [&](int n) {
    const float* yn = g.y + (int64_t)n * O * steps * 16 + (lane & 3) * 4;
    const float* gn = g.g + (int64_t)n * O * steps * 16 + (lane & 3) * 4;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int sidx = min(wave + 8 * k, steps - 1);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int64_t at = (int64_t)min(16 * q + (lane >> 2), O - 1) * steps * 16 + sidx * 16;
        ry[k][q] = *reinterpret_cast<const f32x4*>(yn + at);
        rg[k][q] = *reinterpret_cast<const f32x4*>(gn + at);
      }
    }
  };
  int item = blockIdx.x;
  f32x4 tv[QR];
  if (item < g.N) tile_request<C, KS, QR>(g.x + (int64_t)item * C * H * 32, H, tid, tv);
  const float c_bias = g.bias ? g.bias[o] : 0.0f;
  {
    const int nw = O * KK;
    for (int i = tid; i < 32 * KK + 32; i += 512) Wl[i] = i < nw ? g.w[i] : 0.0f;
    tile_side_columns<C, KS>(Xl, tid);
  }
  float k1 = 0.0f, k2 = 0.0f;
  if constexpr (WGRAD) {
    // sum over the 16 lanes of the row (fixed order: rotations inside the DPP row), then to the lanes of the channel
    auto row16 = [](float t) {
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xf, 0xf, true));   // row_ror:8
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xf, 0xf, true));   // row_ror:4
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x122, 0xf, 0xf, true));   // row_ror:2
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x121, 0xf, 0xf, true));   // row_ror:1
      return t;
    };
    sdy = row16(sdy);
    sdyx = row16(sdyx);
    float* tot = area + 8 * WSLOT;                    // [32][2], its own area
    if ((tid & 15) == 0) { tot[2 * (tid >> 4)] = sdy; tot[2 * (tid >> 4) + 1] = sdyx; }
  }
  if (item < g.N) request_rows(item);
  if (item < g.N) tile_commit<C, KS, QR>(Xl, H, tid, tv);
  __syncthreads();
  ST2(1);
  const float rsd = 1.0f / sqrtf(c_var + g.eps);
  if constexpr (WGRAD) {
    const float* tot = area + 8 * WSLOT;
    const float t1 = tot[2 * o], t2 = tot[2 * o + 1];
    if (blockIdx.x == 0 && tid < O) {
      if (g.dgamma) g.dgamma[tid] = t2;
      if (g.dbeta) g.dbeta[tid] = t1;
    }
    const float m = (float)g.N * (float)H * 32.0f;
    k1 = t1 / m;
    k2 = t2 / m;
  }
  // operands of the transposed convolution: A = the pixel's (channel, tap) value, lane = pixel; B = the weights of
  // output channel l31 -- the forward kernel's registers with the roles swapped, the same k order
  auto wsrc = [&](int p) { const int kk = 2 * p + half; return kk < KK ? l31 * KK + kk : 32 * KK + l31; };
  auto koff = [&](int p) {
    const int k0 = 2 * p < KK ? 2 * p : 0, k1 = 2 * p + 1 < KK ? 2 * p + 1 : 0;
    return (half ? G::koff(k1) : G::koff(k0)) + l31;
  };
  float a[AREG ? NP : 1];
  if constexpr (AREG) {
#pragma unroll
    for (int p = 0; p < NP; p++) a[p] = Wl[wsrc(p)];
  }
  // WGRAD: the (channel, tap) columns of this lane in the weight-gradient tiles (column 32 t + l31 of tile t; a lane
  // past K reads tap 0, its column is not stored)
  int koffw[KT];
#pragma unroll
  for (int t = 0; t < KT; t++) {
    const int kw = 32 * t + l31 < KK ? 32 * t + l31 : 0;
    koffw[t] = (kw / (KS * KS)) * G::CH + ((kw % (KS * KS)) / KS) * kPitch + (kw % KS) + 4 - G::PAD + 4 * half;
  }
  f32x16 acc[KT], accB[ONE ? KT : 1];
#pragma unroll
  for (int t = 0; t < KT; t++)
#pragma unroll
    for (int rr = 0; rr < 16; rr++) acc[t][rr] = 0.0f;
#pragma unroll
  for (int t = 0; t < (ONE ? KT : 1); t++)
#pragma unroll
    for (int rr = 0; rr < 16; rr++) accB[t][rr] = 0.0f;
  float s1 = 0.0f, s2 = 0.0f, sb = 0.0f;            // sb (WGRAD): the lane's sum of d loss / d z = the bias gradient
  float sx[ONE ? KT : 1], sxh = 0.0f;               // ONE: the lane's column sums of the patches; its sum of xhat
#pragma unroll
  for (int t = 0; t < (ONE ? KT : 1); t++) sx[t] = 0.0f;
  ST2(2);
  const int nsw = (steps - wave + 7) >> 3;           // <= 2: the image has at most 16 row pairs
  while (item < g.N) {
    const int next = item + gridDim.x;
    if (next < g.N) tile_request<C, KS, QR>(g.x + (int64_t)next * C * H * 32, H, tid, tv);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (k >= nsw) break;
      const int sidx = wave + 8 * k;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        *reinterpret_cast<f32x4*>(Ys + (16 * q + (lane >> 2)) * YP + (lane & 3) * 4) = ry[k][q];
        *reinterpret_cast<f32x4*>(Gs + (16 * q + (lane >> 2)) * YP + (lane & 3) * 4) = rg[k][q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // accumulator row rr of lane (half, l31) is pixel (rr & 3) + 8 (rr >> 2) + 4 half of the row: its pooled column
      // is 4 (rr >> 2) + 2 half + ((rr & 3) >> 1) -- the pair at 4 q + 2 half for q = rr >> 2
      f32x2 yv[4], gv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        yv[q] = *reinterpret_cast<const f32x2*>(Ys + l31 * YP + 4 * q + 2 * half);
        gv[q] = *reinterpret_cast<const f32x2*>(Gs + l31 * YP + 4 * q + 2 * half);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      constexpr int PCHB = AREG ? NP : 8;
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        // this row's transposed convolution (the accumulator starts from the lane's bias: the forward kernel's
        // sequence), consumed before the next row's is formed
        f32x16 zt;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) zt[rr] = c_bias;
#pragma unroll
        for (int p = 0; p < NP; p++) {
          float ap;
          if constexpr (AREG) ap = a[p]; else ap = Wl[wsrc(p)];
          zt = __builtin_amdgcn_mfma_f32_32x32x2f32(Xl[koff(p) + (2 * sidx + h2) * kPitch], ap, zt, 0, 0, 0);
          if (p % PCHB == PCHB - 1) __builtin_amdgcn_sched_barrier(0);
        }
        const int xrow = (2 * sidx + h2) * kPitch;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          const float z_ = zt[rr];
          const float xh_ = __fmul_rn(__fsub_rn(z_, c_mean), rsd);                  // bn_point's first two steps
          const float bo_ = __fadd_rn(__fmul_rn(xh_, gw), gb);
          const float dy_ = stem_dy2(bo_, yv[rr >> 2][(rr & 3) >> 1], gv[rr >> 2][(rr & 3) >> 1]);
          if constexpr (WGRAD) {
            const float t = (dy_ - k1 - xh_ * k2) * gw * rsd;
            sb += t;
#pragma unroll
            for (int ct = 0; ct < KT; ct++)
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(t, Xl[koffw[ct] + xrow + (rr & 3) + 8 * (rr >> 2)], acc[ct], 0, 0, 0);
          } else if constexpr (ONE) {
            s1 += dy_;
            s2 += dy_ * xh_;
            sxh += xh_;
#pragma unroll
            for (int ct = 0; ct < KT; ct++) {
              const float xv = Xl[koffw[ct] + xrow + (rr & 3) + 8 * (rr >> 2)];
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dy_, xv, acc[ct], 0, 0, 0);
              accB[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(xh_, xv, accB[ct], 0, 0, 0);
              sx[ct] += xv;
            }
          } else {
            s1 += dy_;
            s2 += dy_ * xh_;
          }
        }
      }
    }
    item = next;
    if (item < g.N) {
      request_rows(item);
      __syncthreads();
      tile_commit<C, KS, QR>(Xl, H, tid, tv);
      __syncthreads();
    }
  }
  ST2(3);
  __syncthreads();                                   // every wave is done with its transposition area
  ST2(4);
  if constexpr (WGRAD) {
    // the eight waves' tiles through LDS, a column tile at a time; slab [blockIdx.x][O][KK + 1], column KK = the bias
    // gradient (the lanes' sums of d loss / d z: lane = channel, its two halves and the eight waves added in order)
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(area);       // [8][16][64]
    float* out = g.slabs + (int64_t)blockIdx.x * O * (KK + 1);
#pragma unroll
    for (int ct = 0; ct < KT; ct++) {
      if (ct > 0) __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 16; rr++) red[wave][rr][lane] = acc[ct][rr];
      __syncthreads();
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        const int idx = tid + 512 * h2, rr = idx >> 6, ln = idx & 63;
        float v = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; w8++) v += red[w8][rr][ln];
        const int oo = (rr & 3) + 8 * (rr >> 2) + 4 * (ln >> 5), kk = 32 * ct + (ln & 31);
        if (oo < O && kk < KK) out[(int64_t)oo * (KK + 1) + kk] = v;
      }
    }
    __syncthreads();
    float* sums = area;                              // [8 waves][64 lanes]
    sums[wave * 64 + lane] = sb;
    __syncthreads();
    if (tid < O) {
      float t = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) t += sums[w8 * 64 + tid] + sums[w8 * 64 + 32 + tid];
      out[(int64_t)tid * (KK + 1) + KK] = t;
    }
  } else if constexpr (ONE) {
    // slab [blockIdx.x]: A [O][KK], B [O][KK], then per channel (sum dy, sum dy xhat, sum xhat), then S [KK]
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(area);
    const int SL = 2 * O * KK + 3 * O + KK;
    float* out = g.slabs + (int64_t)blockIdx.x * SL;
#pragma unroll
    for (int which = 0; which < 2; which++) {
#pragma unroll
      for (int ct = 0; ct < KT; ct++) {
        if (which + ct > 0) __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; rr++) red[wave][rr][lane] = which ? accB[ct][rr] : acc[ct][rr];
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
          const int idx = tid + 512 * h2, rr = idx >> 6, ln = idx & 63;
          float v = 0.0f;
#pragma unroll
          for (int w8 = 0; w8 < 8; w8++) v += red[w8][rr][ln];
          const int oo = (rr & 3) + 8 * (rr >> 2) + 4 * (ln >> 5), kk = 32 * ct + (ln & 31);
          if (oo < O && kk < KK) out[(int64_t)which * O * KK + (int64_t)oo * KK + kk] = v;
        }
      }
    }
    __syncthreads();
    float* sums = area;                              // [8 waves][64 lanes][3 + KT]
    constexpr int NS = 3 + KT;
    sums[(wave * 64 + lane) * NS] = s1;
    sums[(wave * 64 + lane) * NS + 1] = s2;
    sums[(wave * 64 + lane) * NS + 2] = sxh;
#pragma unroll
    for (int ct = 0; ct < KT; ct++) sums[(wave * 64 + lane) * NS + 3 + ct] = sx[ct];
    __syncthreads();
    if (tid < 3 * O) {
      const int ch = tid / 3, which = tid % 3;
      float t = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) t += sums[(w8 * 64 + ch) * NS + which] + sums[(w8 * 64 + 32 + ch) * NS + which];
      out[2 * O * KK + tid] = t;
    }
    if (tid >= 128 && tid < 128 + KK) {
      const int kk = tid - 128, ct = kk >> 5, ln = kk & 31;
      float t = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) t += sums[(w8 * 64 + ln) * NS + 3 + ct] + sums[(w8 * 64 + 32 + ln) * NS + 3 + ct];
      out[2 * O * KK + 3 * O + kk] = t;
    }
  } else {
    float* sums = area;                              // [8 waves][64 lanes][2]
    sums[(wave * 64 + lane) * 2] = s1;
    sums[(wave * 64 + lane) * 2 + 1] = s2;
    __syncthreads();
    if (tid < 2 * O) {
      const int ch = tid >> 1, which = tid & 1;
      float t = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) t += sums[(w8 * 64 + ch) * 2 + which] + sums[(w8 * 64 + 32 + ch) * 2 + which];
      g.partial[((int64_t)ch * g.S + blockIdx.x) * 2 + which] = t;
    }
  }
  ST2(5); ST2(7);
}

// out[i] = sum over the slabs in a fixed order.  A block owns 32 consecutive elements; its 256 threads are 8 slab
// groups x 32 elements, a thread takes the slabs sg, sg + 8, ... (eight loads in flight), the groups meet in LDS.
// element i of a slab is (channel i / (KK + 1), column i % (KK + 1)): columns < KK go to dw[channel][column], column KK to
// db[channel] (when there is a bias)
__global__ __launch_bounds__(256) void k_stem_slab_sum(const float* __restrict__ slabs, float* __restrict__ dw,
                                                       float* __restrict__ db, int n, int splits, int KK) {
  stem_slab_sum_body(slabs, dw, db, n, splits, KK, (int)blockIdx.x);      // (dfhip_stem_combine.h: also a rider of the group's slab sum)
}

// One-pass backward: the totals over the workgroups' slabs and dW, dgamma, dbeta, dbias from them (dfhip_stem_combine.h;
// inside a weight-gradient group the same body runs as extra workgroups of the group's slab-sum launch)
__global__ __launch_bounds__(256) void k_stem_combine(StemCombine c) { stem_combine_body(c, (int)blockIdx.x); }

size_t stem_bwd_lds(int C, int k) {
  const int KK = C * k * k;
  return ((size_t)C * (32 + k - 1) * kPitch + 32 * KK + 32 + ((32 * KK) & 1) + std::max(8 * 2 * 32 * 20 + 64, 8 * 16 * 64)) *
         sizeof(float);
}

bool stem_shape(int N, int C, int H, int W, int O, int k, int stride, int pad) {
  return N >= 1 && (k == 3 || k == 5) && stride == 1 && pad == k / 2 && W == 32 && C >= 1 && C <= 3 && O >= 1 &&
         O <= 32 && H >= 2 && H <= 32 && !(H & 1);
}

// the instantiations: C = 1 .. 3 input channels, 3x3 or 5x5
#define DF_STEM_CASES(X) X(1, 3) X(2, 3) X(3, 3) X(1, 5) X(2, 5) X(3, 5)
#define DF_STEM_CASES5(X) X(1, 5) X(2, 5) X(3, 5)      // the two-pass backward exists for the 5x5 stems only

}  // namespace

extern "C" {

int dfhip_stem_forward_noz(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                           float* y, float* save_mean, float* save_var, float* running_mean, float* running_var, int N,
                           int C, int H, int W, int O, int k, int stride, int pad, float eps, float momentum, int* done) {
  DF_INIT();
  DF_CHECK_ARG(x && w && y && save_mean && save_var && done, "stem_forward_noz: null array");
  *done = 0;
  if (!stem_shape(N, C, H, W, O, k, stride, pad)) return DFHIP_OK;
  const int gx = (int)std::min<int64_t>(N, std::min(2 * num_cus(), 512));
  const size_t nsel = (size_t)N * 32 * (H / 2) * 16;      // (32 channel planes per image whatever O is)
  float* zsel = workspace((nsel + (size_t)gx * O * 3) * sizeof(float));
  if (!zsel) return DFHIP_OUT_OF_MEMORY;
  float* partials = zsel + nsel;
  {
    const double flops = 2.0 * N * H * 32.0 * C * k * k * O;
    ProfScope prof("conv2d_fwd_igemm", flops, 4.0 * ((double)N * C * H * 32 + (double)O * C * k * k + (double)N * O * (H / 2) * 16));
#define DF_CASE(CV, KV)                                                                                          \
    if (C == CV && k == KV)                                                                                      \
      DF_LAUNCH((k_stem_fwd_sel<CV, KV>), dim3((unsigned)gx), dim3(512), 0, stream(), x, w, bias, gamma, zsel, partials, N, H, O);
    DF_STEM_CASES(DF_CASE)
#undef DF_CASE
    DF_LAUNCH_CHECK();
    st2_dump("fwd_sel", gx);
  }
  {
    ProfScope prof("bn_relu_pool_fwd", 0.0, 8.0 * (double)N * O * (H / 2) * 16);
    int s = stem_pool_apply(zsel, 32, partials, gx, gamma, beta, y, save_mean, save_var, running_mean, running_var, N, O,
                            (H / 2) * 16, eps, momentum);
    if (s != DFHIP_OK) return s;
  }
  *done = 1;
  return DFHIP_OK;
}

}  // extern "C"
namespace dfhip {
int stem_combine_launch(const StemCombine& c, int blocks) {
  DF_LAUNCH(k_stem_combine, dim3((unsigned)blocks), dim3(256), 0, stream(), c);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}
}  // namespace dfhip
extern "C" {

int dfhip_stem_backward_noz(const float* x, const float* w, const float* bias, const float* y, const float* g,
                            const float* gamma, const float* beta, const float* save_mean, const float* save_var,
                            float* dw, float* dbias, float* dgamma, float* dbeta, int N, int C, int H, int W, int O, int k,
                            int stride, int pad, float eps, int* done) {
  DF_INIT_KEEP();
  DF_CHECK_ARG(x && w && y && g && save_mean && save_var && dw && done, "stem_backward_noz: null array");
  *done = 0;
  // queued weight gradients stay queued when the one-pass form runs inside their group: the group's slab-sum launch then
  // does this call's last step (the slabs of totals live in the second scratch area, the group's own in the first)
  // (round 6: the two-pass form of the 5x5 stems rides too -- its last step is a plain slab sum)
  const bool ride = (k == 3 || k == 5) && group_collecting() && stem_shape(N, C, H, W, O, k, stride, pad);
  if (!ride) {
    int f__ = flush_deferred();
    if (f__ != DFHIP_OK) return f__;
  }
  if (!stem_shape(N, C, H, W, O, k, stride, pad)) return DFHIP_OK;
  const int KK = C * k * k;
  // (k_stem_bwd<.., 1> adds up at most 16 lanes x 16 slices: the slices of the first pass are capped at 256 whatever the device)
  const int gx = (int)std::min<int64_t>(N, std::min(num_cus(), 256));
  const bool one = k == 3;   // (5x5: two accumulator sets of three tiles do not fit the registers)
  const size_t SL = (size_t)2 * O * KK + 3 * O + KK;
  const size_t ws_bytes = std::max((size_t)O * gx * 2 + (size_t)gx * O * (KK + 1), (size_t)gx * SL) * sizeof(float);
  float* ws = ride ? workspace2(ws_bytes) : workspace(ws_bytes);
  if (!ws) return DFHIP_OUT_OF_MEMORY;
  StemB b = {};
  b.x = x; b.w = w; b.y = y; b.g = g; b.gamma = gamma; b.beta = beta; b.mean = save_mean; b.var = save_var; b.bias = bias;
  b.partial = ws; b.slabs = ws + (size_t)O * gx * 2; b.dgamma = dgamma; b.dbeta = dbeta;
  b.N = N; b.H = H; b.O = O; b.S = gx; b.eps = eps;
  const size_t lds = stem_bwd_lds(C, k);
  static bool attr = false;
  if (!attr) {
    const int cap = (int)stem_bwd_lds(3, 5);
#define DF_CASE(CV, KV)                                                                                                   \
    DF_HIP(hipFuncSetAttribute((const void*)k_stem_bwd<CV, KV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
    DF_HIP(hipFuncSetAttribute((const void*)k_stem_bwd<CV, KV, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    DF_STEM_CASES5(DF_CASE)
#undef DF_CASE
    DF_HIP(hipFuncSetAttribute((const void*)k_stem_bwd<1, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    DF_HIP(hipFuncSetAttribute((const void*)k_stem_bwd<2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    DF_HIP(hipFuncSetAttribute((const void*)k_stem_bwd<3, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr = true;
  }
  const double npix = (double)N * O * H * 32;
  if (one) {
    // one pass: statistics and both halves of the weight gradient from one rebuild of the convolution
    b.slabs = ws;
    ProfScope prof("conv2d_wgrad_igemm", 2.0 * N * H * 32.0 * KK * O, 4.0 * ((double)N * C * H * 32 + (double)O * KK + npix / 2));
    if (C == 1) DF_LAUNCH((k_stem_bwd<1, 3, 2>), dim3((unsigned)gx), dim3(512), lds, stream(), b);
    else if (C == 2) DF_LAUNCH((k_stem_bwd<2, 3, 2>), dim3((unsigned)gx), dim3(512), lds, stream(), b);
    else DF_LAUNCH((k_stem_bwd<3, 3, 2>), dim3((unsigned)gx), dim3(512), lds, stream(), b);
    DF_LAUNCH_CHECK();
    st2_dump("bwd_onepass", gx);
    StemCombine c = {};
    c.slabs = ws; c.gamma = gamma; c.var = save_var; c.dw = dw; c.dbias = dbias; c.dgamma = dgamma; c.dbeta = dbeta;
    c.O = O; c.KK = KK; c.splits = gx; c.m = (float)N * (float)H * 32.0f; c.eps = eps;
    const int cblocks = (int)df_ceil_div(O * KK, 8);
    // (a weight-gradient group is collecting: its slab-sum launch, which follows this call, takes the combine along)
    if (!(ride && stem_combine_defer(c, cblocks))) {
      int st = stem_combine_launch(c, cblocks);
      if (st != DFHIP_OK) return st;
    }
    *done = 1;
    return DFHIP_OK;
  }
  {
    ProfScope prof("bn_relu_pool_bwd", 0.0, 4.0 * ((double)N * C * H * 32 + npix / 2));
#define DF_CASE(CV, KV) \
    if (C == CV && k == KV) DF_LAUNCH((k_stem_bwd<CV, KV, 0>), dim3((unsigned)gx), dim3(512), lds, stream(), b);
    DF_STEM_CASES5(DF_CASE)
#undef DF_CASE
    DF_LAUNCH_CHECK();
    st2_dump("bwd_stats", gx);
  }
  {
    ProfScope prof("conv2d_wgrad_igemm", 2.0 * N * H * 32.0 * KK * O, 4.0 * ((double)N * C * H * 32 + (double)O * KK + npix / 2));
#define DF_CASE(CV, KV) \
    if (C == CV && k == KV) DF_LAUNCH((k_stem_bwd<CV, KV, 1>), dim3((unsigned)gx), dim3(512), lds, stream(), b);
    DF_STEM_CASES5(DF_CASE)
#undef DF_CASE
    DF_LAUNCH_CHECK();
    st2_dump("bwd_wgrad", gx);
    const int n = O * (KK + 1);
    StemCombine c = {};
    c.plain = 1; c.slabs = b.slabs; c.dw = dw; c.dbias = dbias; c.O = O; c.KK = KK; c.splits = gx;
    if (!(ride && stem_combine_defer(c, (int)df_ceil_div(n, 32)))) {
      DF_LAUNCH(k_stem_slab_sum, dim3((unsigned)df_ceil_div(n, 32)), dim3(256), 0, stream(), (const float*)b.slabs, dw, dbias, n,
                gx, KK);
      DF_LAUNCH_CHECK();
    }
  }
  *done = 1;
  return DFHIP_OK;
}

}  // extern "C"
