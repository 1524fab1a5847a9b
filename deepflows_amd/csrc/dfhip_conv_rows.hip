// Conv2d forward, 3x3 / stride 1 / pad 1 on 16-, 8- and 4-wide images with 32-128 source channels
// (ResNet layer1-3; reference path nn/functional.py:316-344: pad -> im2col -> matmul): every operand
// of the matrix cores comes out of REGISTERS, there is no staged image, no chunk pipeline and no
// barrier inside the reduction.
//
// v_mfma_f32_16x16x4_f32: D[16 output channels][16 pixels] += A[16][4 source channels] . B[4][16].
// The 16 pixels of a tile are 16 consecutive pixels of the flattened (n, y, x) space: one row of a
// 16-wide image, two rows of an 8-wide one, a whole 4 x 4 image.  Lane (kk = lane / 16, n = lane % 16)
// of the B operand holds x[img][4p + kk][y + i - 1][x + j - 1] for tap (i, j):
//   * the centre column (j = 1) of a vertical variant is ONE coalesced global load (4 channel rows of
//     64 B); a 16-wide image re-uses the row registers of the neighbouring output rows, a 4 x 4 image
//     gets its vertical variants with a 4-lane DPP shift (the zero fill IS the padding);
//   * the left / right columns are the same register shifted by one lane (`row_shr:1` / `row_shl:1`,
//     16-lane DPP rows = pixel tiles, zero fill = the padding of a 16-wide image; narrower images
//     clear the lanes that would read across a row end).
// The weight operand of the workgroup's 16 output channels goes to LDS once, already in MFMA lane
// order (a wave reads four consecutive (channel group, tap) operands with one conflict-free
// ds_read_b128); a wave owns NT pixel tiles, so an operand read feeds NT MFMAs.  Four waves per
// workgroup share the weight image, two workgroups per CU; the waves never meet again until the
// BatchNorm partials (count, mean, centred second moment per workgroup and channel, the layout of
// dfhip_conv2d_forward_stats) are combined at the very end.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "dfhip_internal.h"

using namespace dfhip;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
#ifdef DFHIP_ROWS_NODPP
  return v;      // diagnostic build (timing only, wrong results): no shifts
#endif
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int kShr1 = 0x111, kShl1 = 0x101, kShr4 = 0x114, kShl4 = 0x104;
// LDS pitch (floats) of one block of four operands x 64 lanes: 4 floats of padding put the 16 blocks a staging
// store instruction touches on different banks (256 would put them all on the same four)
constexpr int kBlk = 260;
constexpr int kRor1 = 0x121, kRor2 = 0x122, kRor4 = 0x124, kRor8 = 0x128;

// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_mov<kRor8>(v);
  v += dpp_mov<kRor4>(v);
  v += dpp_mov<kRor2>(v);
  v += dpp_mov<kRor1>(v);
  return v;
}

#ifdef DFHIP_ROWS_STAMPS
// diagnostic build only: per-wave cycle stamps (s_memtime) and one real-time stamp pair per wave
__device__ unsigned long long g_rows_stamps[512 * 8 * 8];
#define RS_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (lane == 0) g_rows_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + (i)] =          \
        (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();                        \
  } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif

// LW: log2 of the image width (4, 3, 2); C: source channels; NT: pixel tiles per wave; VDPP: the tile is the
// whole image (H * W == 16): vertical variants by DPP instead of loads.  A workgroup is 8 waves = TS tile sets x
// OW output-channel tiles; the weights of its OW tiles go through LDS in chunks of CG channel groups (chunk
// k + 1 is requested before and committed after the MFMAs of chunk k: one barrier per chunk, every chunk has
// LDS space of its own); the pixel operands of chunk k + 1 are requested at the same point.
// DG: the INPUT GRADIENT of the same convolution in its mathematically correct ("accumulate") form -- a 3x3 / stride
// 1 / pad 1 convolution of dY with the weights transposed and flipped: x = dY (C = the convolution's output
// channels), O = its input channels, operand (d, s, t) = w[s][d][8 - t]; `bias` is then a TENSOR of the output's
// shape that is added to the result (the gradient the input already holds, tensor.py:444), not a vector.
// BNI: x is the RAW output of the previous convolution and the operand of this one is its training-mode BatchNorm
// (batchnorm.py:30-55, no activation in between: test/ResNet_CIFAR10_cuda.py:31-35): bn.affine holds the per-channel
// (scale, shift) of that BatchNorm (dfhip_bn2d_finalize_partials: a few-microsecond launch that combines the
// partials the producer left behind), a workgroup copies it to LDS, and a pixel operand becomes
// fma(raw, scale, shift) on its way out of the load registers -- the normalised tensor is never written or read.
// (Combining the partials inside this kernel instead -- every workgroup for itself, 100 KB of rows each -- cost
// 2-3.7 us per launch: 256 workgroups pulling the same lines through L2 at the same moment.)
// BF (opt-in, dfhip_set_conv_matmul_dtype(1); outside the 1e-4 parity contract; forward only): the contraction on
// v_mfma_f32_16x16x16_bf16, fp32 accumulators.  A weight chunk is CG = 4 channel groups = 16 source channels = the K of ONE bf16
// instruction, and in its operand layout lane (m, kk) supplies k = 4 kk .. 4 kk + 3 -- with k = 4 kk + e standing for channel
// 4 (p0 + e) + kk these are the four values the lane holds for the chunk's four groups in the fp32 kernel.  So the loads, the
// BatchNorm operand transform, the lane shifts and the weight staging are the fp32 kernel's; the four groups' operands of a
// tap are rounded (nearest even) into two registers and one instruction of 16 cycles replaces four of 32.
template <int LW, int C, int NT, bool VDPP, int TS, int OW, int CG, bool DG, bool BNI, bool BF = false>
__global__ __launch_bounds__(512) void k_conv3x3_rowmma(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ partials, int H, int O, BnInput bn) {
  static_assert(TS * OW == 8, "a workgroup is eight waves");
  static_assert(!BF || (CG == 4 && !VDPP && !DG), "the bf16 contraction: a chunk is one K = 16, forward, rows of 8 / 16 pixels");
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float Wl[];   // [OW][C/4*9/4 blocks][64 lanes][4] at pitch kBlk
  __shared__ float sred[8][16][2];
  __shared__ __attribute__((aligned(8))) float bn_tab[BNI ? C : 1][2];
  __shared__ float bn_acc[BNI ? 512 / C : 1][BNI ? C : 1][3];
  constexpr int W = 1 << LW;
  constexpr int NG = C / 4;                    // channel groups (one MFMA k-step each)
  constexpr int NCH = NG / CG;                 // weight chunks
  constexpr int NBLK = NG * 9 / 4;             // operand blocks per output-channel tile
  constexpr int CBLK = CG * 9 / 4;             // ... per chunk
  constexpr int NL = LW == 4 ? NT + 2 : (VDPP ? NT : 3 * NT);   // global loads per group and wave
  static_assert(NG % CG == 0 && (CG * 9) % 4 == 0, "chunks are whole operand blocks");
  const int HW = H << LW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const int ts = wave % TS, otl = wave / TS;
  const int o0 = (blockIdx.y * OW + otl) * 16;
  const int T0 = (blockIdx.x * TS + ts) * NT;  // first pixel tile of this wave
  RS_STAMP(6);
  RS_STAMP(0);
  // ---- BNI: the (scale, shift) pair of this thread's channel is requested first (dfhip_bn2d_finalize_partials wrote
  // the table from the producer's partials) ----
  // (Built, measured and dropped: a wave owning C / 8 channels with its lanes split over the rows and lane exchanges
  // instead of the LDS round and the second barrier -- the per-lane rows make every load sixteen short segments:
  // 16.5 -> 17.1 and 17.9 -> 20.4 us; every wave finishing the sums for itself behind the first barrier -- no second
  // barrier, eight times the divisions: 16.5 -> 16.5, 17.9 -> 18.3 us.)
  // There is no launch for the table -- every workgroup adds the producer's rows up for itself (thread = a channel
  // x a sixteenth / an eighth of the rows, 12-byte loads requested here, in front of the weights; the shifted sums of
  // k_bn_finalize_partials, combined through LDS in a fixed order), workgroup (0, 0) leaves the table, the batch
  // statistics and the running statistics behind for the backward pass.
  f32x2 bn_pair = {1.0f, 0.0f};
  constexpr int BNG = 512 / C, BRP = 16;            // row groups; rows per thread and pass
  f32x3u bn_row[BNI ? BRP : 1];
  float bn_shift = 0.0f, bn_g = 1.0f, bn_b = 0.0f, bn_rm = 0.0f, bn_rv = 0.0f;
  const int bn_c = tid % C, bn_gr = tid / C;
  if constexpr (BNI) {
    const float* pc = bn.part + bn_c * 3;
    // (every load instruction of the workgroup queues in front of its weight and pixel loads: half of them when the rows are few)
    if (bn.rows <= BNG * BRP / 2) {
#pragma unroll
      for (int k2 = 0; k2 < BRP / 2; k2++)
        bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(bn_gr + BNG * k2, bn.rows - 1) * (C * 3));
#pragma unroll
      for (int k2 = BRP / 2; k2 < BRP; k2++) bn_row[k2] = (f32x3u){0.0f, 0.0f, 0.0f};
    } else {
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++)
        bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(bn_gr + BNG * k2, bn.rows - 1) * (C * 3));
    }
    bn_shift = pc[1];                               // row 0's mean
    if (tid < C) {
      if (bn.gamma) bn_g = bn.gamma[tid];
      if (bn.beta) bn_b = bn.beta[tid];
      if (blockIdx.x == 0 && blockIdx.y == 0) {                     // (only the workgroup that updates them)
        if (bn.running_mean) bn_rm = bn.running_mean[tid];
        if (bn.running_var) bn_rv = bn.running_var[tid];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- weight chunk staging: thread -> a 16-byte piece of the chunk's weights; TOT4 pieces per chunk ----
  //   forward: OW x 16 rows (output channels) of 36 * CG contiguous floats (CG groups of 4 source channels x 9 taps)
  //   DG:      4 * CG source channels (rows of w) x OW runs of 144 contiguous floats (16 destination channels x 9 taps)
  constexpr int ROW4 = 9 * CG, TOT4 = OW * 16 * ROW4, PER = (TOT4 + 511) / 512;
  const float* wsrc[PER];
  // LDS float index of every element of the piece within chunk 0 (operand f = p * 9 + t of lane (kk, m) lives at block
  // f / 4, slot (kk * 16 + m) * 4 + f % 4); resolved ONCE: a commit is then PER * 4 plain ds_write_b32 at a compile-time
  // chunk offset -- with the index arithmetic inside the commit both waves of a SIMD spent ~600 cycles per chunk on
  // it at the same time, with nothing in the matrix pipe
  int woff[PER][4];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    // (threads past the end re-do the piece 512 places back: every load has an UNCONDITIONAL consumer -- under a
    // condition the compiler sinks the load into the commit, behind the MFMAs it was meant to overlap)
    const int q = tid + 512 * u;
    const int qq = q < TOT4 ? q : q % TOT4;
    if constexpr (!DG) {
      const int tl = qq / (16 * ROW4), r = qq - tl * (16 * ROW4);
      const int m = r / ROW4, e4 = r - m * ROW4;
      wsrc[u] = w + ((int64_t)((blockIdx.y * OW + tl) * 16 + m) * C * 9) + 4 * e4;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int idx = 4 * e4 + e;
        const int c = idx / 9, t = idx - 9 * c;
        const int f = (c >> 2) * 9 + t;
        woff[u][e] = tl * (NBLK * kBlk) + (f >> 2) * kBlk + (c & 3) * 64 + m * 4 + (f & 3);
      }
    } else {
      const int sl = qq / (OW * 36), r = qq - sl * (OW * 36);
      const int tl = r / 36, e4 = r - tl * 36;
      wsrc[u] = w + ((int64_t)sl * O + (blockIdx.y * OW + tl) * 16) * 9 + 4 * e4;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int idx = 4 * e4 + e;
        const int m = idx / 9, t = idx - 9 * m;          // destination channel in the tile, tap of w
        const int f = (sl >> 2) * 9 + (8 - t);           // flipped tap
        woff[u][e] = tl * (NBLK * kBlk) + (f >> 2) * kBlk + (sl & 3) * 64 + m * 4 + (f & 3);
      }
    }
  }
  const int64_t wchunk = DG ? (int64_t)(4 * CG) * O * 9 : 36 * CG;     // floats between two chunks of a piece
  f32x4 wv[2][PER];                            // two chunks in flight
#define RM_WSLOT(chunk) ((chunk) & 1)
#define RM_WLOAD(chunk)                                                                         \
  do {                                                                                          \
    _Pragma("unroll") for (int u = 0; u < PER; u++)                                             \
      wv[RM_WSLOT(chunk)][u] = *reinterpret_cast<const f32x4*>(wsrc[u] + (chunk) * wchunk);     \
  } while (0)
#define RM_WCOMMIT(chunk)                                                                       \
  do {                                                                                          \
    _Pragma("unroll") for (int u = 0; u < PER; u++) {                                           \
      _Pragma("unroll") for (int e = 0; e < 4; e++)                                             \
        Wl[woff[u][e] + (chunk) * (CBLK * kBlk)] = wv[RM_WSLOT(chunk)][u][e];                   \
    }                                                                                           \
  } while (0)
  RM_WLOAD(0);
  if (NCH > 1) RM_WLOAD(1);
  __builtin_amdgcn_sched_barrier(0);           // the weight loads stay in front of the pixel loads

  // ---- per-lane source offsets (elements from x, channel group 0) and validity of every load ----
  // (a load whose source row is outside the image gets an offset past the buffer's size: the buffer load
  // returns zero for it, which is the padding)
  constexpr unsigned kOut = 0x80000000u;
  unsigned xo[NL];
  float vm[NL];                               // BNI: 1 / 0, the load's source row is inside the image
  float m_l[NT], m_r[NT];                     // narrower images: 0 for the lanes whose left / right neighbour is across a row end
  int img_t[NT], rem_t[NT];
  if constexpr (LW == 4) {
    const int img = T0 / H, y0 = T0 - img * H;           // NT consecutive rows of one image (H % NT == 0)
#pragma unroll
    for (int r = 0; r < NT + 2; r++) {
      const int yy = y0 - 1 + r;
      xo[r] = (yy >= 0 && yy < H) ? (unsigned)((((int64_t)img * C + kk) * H + yy) * W + n) * 4u : kOut;
      vm[r] = (yy >= 0 && yy < H) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) { img_t[t] = img; rem_t[t] = (y0 + t) * W; m_l[t] = 1.0f; m_r[t] = 1.0f; }
  } else {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int P = (T0 + t) * 16;                         // first pixel of the tile: never straddles images
      const int img = P / HW, rem0 = P - img * HW;
      img_t[t] = img; rem_t[t] = rem0;
      const int rem = rem0 + n, yy = rem >> LW, xx = rem & (W - 1);
      m_l[t] = xx == 0 ? 0.0f : 1.0f; m_r[t] = xx == W - 1 ? 0.0f : 1.0f;
      if constexpr (VDPP) {
        xo[t] = (unsigned)(((int64_t)img * C + kk) * HW + rem) * 4u;
        vm[t] = 1.0f;
      } else {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int ys = yy + i - 1;
          xo[t * 3 + i] = (ys >= 0 && ys < H) ? (unsigned)((((int64_t)img * C + kk) * H + ys) * W + xx) * 4u : kOut;
          vm[t * 3 + i] = (ys >= 0 && ys < H) ? 1.0f : 0.0f;
        }
      }
    }
  }
  // pixel operands: buffer loads, the lane's offset is one VGPR per load kind, the channel group's a scalar
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);   // (tensors are < 2^31 bytes: check_conv_args)
  float raw[NG][NL];
#ifdef DFHIP_ROWS_NOXLOAD      // diagnostic build (timing only, wrong results): no pixel loads
#define RM_XVALUE(u, p_) __int_as_float((int)xo[u] + (p_))
#else
#define RM_XVALUE(u, p_) __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (int)xo[u], (p_) * (16 * HW), 0))
#endif
#define RM_XLOAD(chunk)                                                                                          \
  do {                                                                                                           \
    _Pragma("unroll") for (int g_ = 0; g_ < CG; g_++) {                                                          \
      const int p_ = (chunk) * CG + g_;                                                                          \
      _Pragma("unroll") for (int u = 0; u < NL; u++)                                                             \
        raw[p_][u] = RM_XVALUE(u, p_);                                                                           \
      __builtin_amdgcn_sched_barrier(0);       /* group by group: the first group lands first */                 \
    }                                                                                                            \
  } while (0)
  RM_XLOAD(0);
  RS_STAMP(1);
  if constexpr (BNI) {
    float tn = 0.0f, t1 = 0.0f, t2 = 0.0f;
    auto add_rows = [&](int r0) {
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++) {
        const bool in = r0 + bn_gr + BNG * k2 < bn.rows;
        const float nb = in ? bn_row[k2][0] : 0.0f, d = nb > 0.0f ? bn_row[k2][1] - bn_shift : 0.0f;
        tn += nb;
        t1 = fmaf(nb, d, t1);
        t2 += fmaf(nb * d, d, in ? bn_row[k2][2] : 0.0f);
      }
    };
    add_rows(0);
    for (int r0 = BNG * BRP; r0 < bn.rows; r0 += BNG * BRP) {     // (more rows than one pass holds: uniform, rare)
      const float* pc = bn.part + bn_c * 3;
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++)
        bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(r0 + bn_gr + BNG * k2, bn.rows - 1) * (C * 3));
      add_rows(r0);
    }
    bn_acc[bn_gr][bn_c][0] = tn; bn_acc[bn_gr][bn_c][1] = t1; bn_acc[bn_gr][bn_c][2] = t2;
    // (summed per channel behind the barrier of the first weight commit; a second barrier publishes the table)
  }
  RM_WCOMMIT(0);
  RS_STAMP(2);
  __syncthreads();
  RS_STAMP(3);

  // four accumulation chains per wave whatever NT is (the taps of a tile go round its 4 / NT chains): with two,
  // the matrix pipe of a SIMD idled a third of the time
  constexpr int NCN = NT >= 4 ? 1 : 4 / NT;
  f32x4 acc[NT][NCN];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int c = 0; c < NCN; c++) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const f32x4* Wl4 = reinterpret_cast<const f32x4*>(Wl + otl * (NBLK * kBlk)) + lane;
  f32x4 a4[2][CBLK];                                       // weight operands of two chunks: CG groups x 9 taps each
#pragma unroll
  for (int q = 0; q < CBLK; q++) a4[0][q] = Wl4[q * (kBlk / 4)];
  if constexpr (BNI) {
    if (tid < C) {
      float n = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
      for (int g2 = 0; g2 < BNG; g2++) { n += bn_acc[g2][tid][0]; a1 += bn_acc[g2][tid][1]; a2 += bn_acc[g2][tid][2]; }
      const float mu_s = n > 0.0f ? a1 / n : 0.0f;
      const float mu = bn_shift + mu_s;
      const float var = n > 0.0f ? fmaxf(a2 - a1 * mu_s, 0.0f) / n : 0.0f;          // biased variance (batchnorm.py:41-42)
      const float rsd = 1.0f / sqrtf(var + bn.eps);
      bn_pair[0] = bn_g * rsd;
      bn_pair[1] = bn_b - mu * bn_pair[0];
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (bn.save_mean) bn.save_mean[tid] = mu;
        if (bn.save_var) bn.save_var[tid] = var;
        if (bn.running_mean) bn.running_mean[tid] = bn_rm * (1.0f - bn.momentum) + mu * bn.momentum;
        if (bn.running_var) bn.running_var[tid] = bn_rv * (1.0f - bn.momentum) + var * bn.momentum;
        bn.affine[2 * tid] = bn_pair[0];
        bn.affine[2 * tid + 1] = bn_pair[1];
      }
    }
    if (tid < C) { bn_tab[tid][0] = bn_pair[0]; bn_tab[tid][1] = bn_pair[1]; }
    __syncthreads();
  }
  float bsh[2][NT][3][3];
  auto make_shifts = [&](float (&b)[NT][3][3], const float (&rw)[NL]) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      if constexpr (VDPP) {
        // a 4 x 4 image per tile: tap (i, j) is the register shifted by 4 (i - 1) + (j - 1) lanes, ONE DPP
        // move each (the zero fill is the vertical padding), times the lane mask of the column
        const float v = rw[t];
        b[t][0][0] = dpp_mov<0x115>(v) * m_l[t];   // row_shr:5
        b[t][0][1] = dpp_mov<kShr4>(v);
        b[t][0][2] = dpp_mov<0x113>(v) * m_r[t];   // row_shr:3
        b[t][1][0] = dpp_mov<kShr1>(v) * m_l[t];
        b[t][1][1] = v;
        b[t][1][2] = dpp_mov<kShl1>(v) * m_r[t];
        b[t][2][0] = dpp_mov<0x103>(v) * m_l[t];   // row_shl:3
        b[t][2][1] = dpp_mov<kShl4>(v);
        b[t][2][2] = dpp_mov<0x105>(v) * m_r[t];   // row_shl:5
      } else if constexpr (LW < 4) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const float v = rw[t * 3 + i];
          b[t][i][0] = dpp_mov<kShr1>(v) * m_l[t];
          b[t][i][1] = v;
          b[t][i][2] = dpp_mov<kShl1>(v) * m_r[t];
        }
      }
    }
  };
  // BNI: the (scale, shift) pair of a lane's channel of group q, read from the LDS table two groups before its use;
  // a load whose source row is outside the image returned 0 and must stay 0 (the padding is of the NORMALISED tensor)
  f32x2 tabr[BNI ? NG : 1];
  auto bn_apply = [&](float (&dst)[NL], const float (&rw)[NL], const f32x2 t2) {
#pragma unroll
    for (int u = 0; u < NL; u++) {
      float v = BNI ? fmaf(rw[u], t2[0], t2[1]) : rw[u];
      if constexpr (BNI && !VDPP) {
        if (LW == 4 ? (u == 0 || u == NT + 1) : (u % 3 != 1)) v *= vm[u];
      }
      dst[u] = v;
    }
  };
  if constexpr (BNI) {
    tabr[0] = *reinterpret_cast<const f32x2*>(&bn_tab[kk][0]);
    if (NG > 1) tabr[1] = *reinterpret_cast<const f32x2*>(&bn_tab[4 + kk][0]);
  }
  if constexpr (LW < 4) {
    float nr[NL];
    bn_apply(nr, raw[0], tabr[0]);
    make_shifts(bsh[0], nr);
  }
  float bfc[BF && LW == 4 ? 3 : 1][BF && LW == 4 ? NT + 2 : 1][BF && LW == 4 ? CG : 1];              // bf16: (column variant, row, group)
  float bfb[BF && LW != 4 ? NT : 1][BF && LW != 4 ? 3 : 1][BF && LW != 4 ? 3 : 1][BF && LW != 4 ? CG : 1]; // (tile, tap row, tap column, group)
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    // chunk ch + 1 (requested a chunk ago) is committed now and made visible by a barrier in the MIDDLE of this
    // chunk's MFMAs, where no wave waits for anything else; chunk ch + 2 and the pixels of ch + 1 are requested
    if (ch + 1 < NCH) RM_WCOMMIT(ch + 1);
    if (ch + 2 < NCH) RM_WLOAD(ch + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (ch + 1 < NCH) RM_XLOAD(ch + 1);
    for (int g = 0; g < CG; g++) {
      const int p = ch * CG + g;
      if (g == CG / 2 && ch + 1 < NCH) {
#ifndef DFHIP_ROWS_NOBAR        // (diagnostic build, timing only: no barrier inside the reduction)
        __syncthreads();
#endif
        // the next chunk's operands are visible now: read them here, half a chunk before their first use
#pragma unroll
        for (int q = 0; q < CBLK; q++) a4[(ch + 1) & 1][q] = Wl4[((ch + 1) * CBLK + q) * (kBlk / 4)];
      }
      if constexpr (BNI) {
        if (p + 2 < NG) tabr[p + 2] = *reinterpret_cast<const f32x2*>(&bn_tab[4 * (p + 2) + kk][0]);
      }
      // the nine pixel operands of every tile from this group's registers
      if constexpr (BF) {
        // bf16: the group's operands are made as in the fp32 kernel and kept; the chunk's last group packs the four
        // groups' values of every (tile, tap) and issues the chunk's 9 NT instructions
        float nr[NL];
        bn_apply(nr, raw[p], tabr[BNI ? p : 0]);
        if constexpr (LW == 4) {
#pragma unroll
          for (int r = 0; r < NT + 2; r++) {
            bfc[1][r][g] = nr[r];
            bfc[0][r][g] = dpp_mov<kShr1>(nr[r]);
            bfc[2][r][g] = dpp_mov<kShl1>(nr[r]);
          }
        } else {
          float bl[NT][3][3];
          make_shifts(bl, nr);
#pragma unroll
          for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
              for (int j = 0; j < 3; j++) bfb[t][i][j][g] = bl[t][i][j];
        }
        if (g == CG - 1) {
#pragma unroll
          for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
              const int t9 = i * 3 + j;
              const bf16x4 pa = {(__bf16)a4[ch & 1][(0 * 9 + t9) >> 2][(0 * 9 + t9) & 3], (__bf16)a4[ch & 1][(1 * 9 + t9) >> 2][(1 * 9 + t9) & 3],
                                 (__bf16)a4[ch & 1][(2 * 9 + t9) >> 2][(2 * 9 + t9) & 3], (__bf16)a4[ch & 1][(3 * 9 + t9) >> 2][(3 * 9 + t9) & 3]};
#pragma unroll
              for (int t = 0; t < NT; t++) {
                bf16x4 pb;
                if constexpr (LW == 4) pb = (bf16x4){(__bf16)bfc[j][t + i][0], (__bf16)bfc[j][t + i][1], (__bf16)bfc[j][t + i][2], (__bf16)bfc[j][t + i][3]};
                else pb = (bf16x4){(__bf16)bfb[t][i][j][0], (__bf16)bfb[t][i][j][1], (__bf16)bfb[t][i][j][2], (__bf16)bfb[t][i][j][3]};
                acc[t][t9 % NCN] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pb, acc[t][t9 % NCN], 0, 0, 0);
              }
            }
          }
        }
      } else if constexpr (LW == 4) {
        float c1[NT + 2], c0[NT + 2], c2[NT + 2], nr[NL];
        bn_apply(nr, raw[p], tabr[BNI ? p : 0]);
#pragma unroll
        for (int r = 0; r < NT + 2; r++) {
          c1[r] = nr[r];
          c0[r] = dpp_mov<kShr1>(c1[r]);
          c2[r] = dpp_mov<kShl1>(c1[r]);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int f = g * 9 + i * 3 + j;
            const float a = a4[ch & 1][f >> 2][f & 3];
#pragma unroll
            for (int t = 0; t < NT; t++) {
              const float b = j == 0 ? c0[t + i] : (j == 1 ? c1[t + i] : c2[t + i]);
              acc[t][(i * 3 + j) % NCN] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t][(i * 3 + j) % NCN], 0, 0, 0);
            }
          }
        }
      } else {
        // the shifted operands of group p + 1 are made WHILE the MFMAs of group p issue (they read only registers
        // that were loaded a chunk ago): an MFMA whose operand comes from the instruction in front of it stalls the
        // wave's issue, and with it the matrix pipe, for the latency of that instruction
        __builtin_amdgcn_sched_barrier(0);
        if (p + 1 < NG) {
          float nr[NL];
          bn_apply(nr, raw[p + 1], tabr[BNI ? p + 1 : 0]);
          make_shifts(bsh[(p + 1) & 1], nr);
        }
        float (&b)[NT][3][3] = bsh[p & 1];
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int f = g * 9 + i * 3 + j;
            const float a = a4[ch & 1][f >> 2][f & 3];
#pragma unroll
            for (int t = 0; t < NT; t++) {
              acc[t][(i * 3 + j) % NCN] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t][i][j], acc[t][(i * 3 + j) % NCN], 0, 0, 0);
            }
          }
        }
        // issue order of the group: one shift behind every MFMA
#pragma unroll
        for (int k2 = 0; k2 < 9 * NT; k2++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, BNI ? 2 : 1, 0);
        }
      }
    }
  }
#undef RM_WLOAD
#undef RM_WSLOT
#undef RM_WCOMMIT
#undef RM_XLOAD
#undef RM_XVALUE

  f32x4 res[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    res[t] = acc[t][0];
#pragma unroll
    for (int c = 1; c < NCN; c++) res[t] += acc[t][c];
  }
  RS_STAMP(4);
  // ---- epilogue: accumulator register v of lane (kk, n) is channel o0 + 4 * kk + v, pixel n of the tile ----
  if (bias && !DG) {
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float bv = bias[o0 + 4 * kk + v];
#pragma unroll
      for (int t = 0; t < NT; t++) res[t][v] += bv;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int64_t at = ((int64_t)img_t[t] * O + o0 + 4 * kk) * HW + rem_t[t] + n;
    if (DG && bias) {
#pragma unroll
      for (int v = 0; v < 4; v++) res[t][v] += bias[at + (int64_t)v * HW];
    }
#pragma unroll
    for (int v = 0; v < 4; v++) y[at + (int64_t)v * HW] = res[t][v];
  }
  if (partials) {
    // BatchNorm partials: exact two-pass over the wave's NT * 16 pixels per channel, then Chan's
    // combination of the TS waves that share the output channels
    constexpr float cnt_w = 16.0f * NT;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      float s = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; t++) s += res[t][v];
      const float mean_w = row_sum16(s) * (1.0f / cnt_w);
      float q = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const float d = res[t][v] - mean_w;
        q += d * d;
      }
      q = row_sum16(q);
      if (n == 0) {
        sred[wave][4 * kk + v][0] = mean_w;
        sred[wave][4 * kk + v][1] = q;
      }
    }
    __syncthreads();
    if (tid < 16 * OW) {
      const int tl = tid >> 4, ch = tid & 15;
      float nn = cnt_w, mu = sred[tl * TS][ch][0], m2 = sred[tl * TS][ch][1];
#pragma unroll
      for (int w4 = 1; w4 < TS; w4++) {
        const float tot = nn + cnt_w, delta = sred[tl * TS + w4][ch][0] - mu;
        mu += delta * (cnt_w / tot);
        m2 += sred[tl * TS + w4][ch][1] + delta * delta * (nn * cnt_w / tot);
        nn = tot;
      }
      float* o = partials + ((int64_t)blockIdx.x * O + (blockIdx.y * OW + tl) * 16 + ch) * 3;
      o[0] = nn; o[1] = mu; o[2] = m2;
    }
  }
  RS_STAMP(5);
  RS_STAMP(7);
}

template <int LW, int C, int NT, bool VDPP, int TS, int OW, int CG, bool DG = false, bool BNI = false, bool BF = false>
int launch_rowmma(const float* x, const float* w, const float* bias, float* y, float* partials, int H, int O,
                  dim3 grid, const BnInput* bnp = nullptr) {
  BnInput bn = {};
  if (bnp) bn = *bnp;
  const size_t lds = (size_t)OW * (C / 4 * 9 / 4) * kBlk * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv3x3_rowmma<LW, C, NT, VDPP, TS, OW, CG, DG, BNI, BF>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)65536)));
    attr = true;
  }
  DF_LAUNCH((k_conv3x3_rowmma<LW, C, NT, VDPP, TS, OW, CG, DG, BNI, BF>), grid, dim3(512), lds, stream(), x, w, bias, y,
                     partials, H, O, bn);
  DF_LAUNCH_CHECK();
#ifdef DFHIP_ROWS_STAMPS
  {
    static int calls = 0;
    if (++calls % 20 == 0) {
      const size_t nw = (size_t)grid.x * grid.y * 8;
      std::vector<unsigned long long> h(nw * 8);
      (void)hipStreamSynchronize(stream());
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_rows_stamps), nw * 8 * sizeof(unsigned long long));
      double seg[6] = {0}, cyc_sum = 0, cyc_max = 0, rt_sum = 0, rt_max = 0;
      unsigned long long r0 = ~0ull, r1 = 0, rs = 0;
      for (size_t i = 0; i < nw; i++) {
        const unsigned long long* r = &h[i * 8];
        for (int k = 1; k <= 5; k++) seg[k] += (double)(r[k] - r[k - 1]);
        r0 = std::min(r0, r[6]); r1 = std::max(r1, r[7]); rs = std::max(rs, r[6]);
        const double cyc = (double)(r[5] - r[0]), rt = (double)(r[7] - r[6]);
        cyc_sum += cyc; cyc_max = std::max(cyc_max, cyc); rt_sum += rt; rt_max = std::max(rt_max, rt);
      }
      fprintf(stderr, "[rows-clock] cycles per wave mean %.0f max %.0f; wave life mean %.2f us max %.2f us -> %.3f GHz\n",
              cyc_sum / nw, cyc_max, rt_sum / nw / 100.0, rt_max / 100.0, cyc_sum / rt_sum / 10.0);
      fprintf(stderr, "[rows-stamp] LW %d C %d NT %d TS %d OW %d CG %d grid %u x %u: first start -> last end %.2f us, last "
              "wave starts +%.2f us; mean cycles per wave: issue %.0f, w-commit %.0f, barrier %.0f, mfma-loop %.0f, epilogue %.0f\n",
              LW, C, NT, TS, OW, CG, grid.x, grid.y, (double)(r1 - r0) / 100.0, (double)(rs - r0) / 100.0,
              seg[1] / nw, seg[2] / nw, seg[3] / nw, seg[4] / nw, seg[5] / nw);
    }
  }
#endif
  return DFHIP_OK;
}


// ---------------------------------------------------------------------------------------
// 4 x 4 images (ResNet layer3, 128 -> 128 channels) with an IMAGE on the lane -- the forward twin of the
// image-per-lane input-gradient kernels and of k_conv3x3_img2_all (dfhip_conv.hip).  v_mfma_f32_16x16x4_f32:
// D[16 output channels][16 images] += A[16][4 source channels] . B[4][16]; an accumulator per output pixel, and a
// (pixel, tap) pair whose source falls outside the image is simply NOT multiplied: 100 of the 144 pairs of a 4 x 4
// image exist, so the layer is 0.84 instead of 1.21 GFLOP of MFMAs, and no lane shift, mask or select is needed to
// build a tap operand (the register-operand kernel spends 8 DPP moves + 6 multiplies per 9 MFMAs on this width:
// beside fp32 MFMAs that is matrix-pipe time).  A workgroup is 16 images x 16 output channels x one HALF of the image
// (rows 2h, 2h + 1: 256 workgroups at batch 256); its eight waves split the source channels (16 each) and stream them
// eight at a time through wave-private LDS areas: the weight records (16 output channels x 8 channels x 9 taps) and
// the image planes (16 images x 8 channels x 64 bytes) both in memory order, then read back in MFMA lane order -- no
// block barrier before the first MFMA.  `aff` (or nullptr): the BatchNorm of the input applied on its way out of the
// strip (dfhip_conv2d_forward_bn_input: (scale, shift) per channel); the padding needs no care, it is never read.
// Sums through LDS, 16-byte stores, BatchNorm partials over the 16 images x 8 pixels of a channel.
// ---------------------------------------------------------------------------------------
template <int HH>
__device__ __forceinline__ void img4_fwd_mfmas(const float (&a)[9], const f32x4 (&xr)[4], f32x4 (&acc)[8]) {
#pragma unroll
  for (int yy = 0; yy < 2; yy++)
#pragma unroll
    for (int xx = 0; xx < 4; xx++)
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int sy = 2 * HH + yy + i - 1, sx = xx + j - 1;
          if (sy >= 0 && sy < 4 && sx >= 0 && sx < 4)
            acc[yy * 4 + xx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * 3 + j], xr[yy + i][sx], acc[yy * 4 + xx], 0, 0, 0);
        }
}

template <int C, bool COMB>
__global__ __launch_bounds__(512) void k_conv3x3_img4_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ stats, int O,
                                                          const float* __restrict__ aff, BnInput bn) {
  constexpr int CW = C / 8, NCHK = CW / 8, WP = 17, XP = 33;      // XP: float4 pitch of an image's strip row
  static_assert(CW == 16, "a wave's channels are one DPP row of the combine below");
  constexpr int WSL = 72 * WP + 16 * XP * 4;                       // floats of a wave's area
  extern __shared__ __attribute__((aligned(16))) float smem_i4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const DfTile wg = {(int)blockIdx.x, (int)blockIdx.y};          // (image tile, channel tile) in grid order
  const int n0 = wg.x * 16, o0 = wg.y * 16, h = blockIdx.z;
  float* Wl = smem_i4 + wave * WSL;
  float* Xs = Wl + 72 * WP;
  __shared__ __attribute__((aligned(8))) float bn_wtab[COMB ? 8 : 1][COMB ? CW : 1][2];   // a wave's own pairs
  const int cb = wave * CW;
  f32x4 wq[5], xq[8];
  auto request = [&](int ch) {
    const int c0 = cb + 8 * ch;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int e = min(lane + 64 * i, 287), m = e / 18, j = e - 18 * m;
      wq[i] = *reinterpret_cast<const f32x4*>(w + ((int64_t)(o0 + m) * C + c0) * 9 + 4 * j);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = lane + 64 * i, im = e >> 5, j = e & 31;      // float4 j of the image's 8 channels x 16 pixels
      xq[i] = *reinterpret_cast<const f32x4*>(x + ((int64_t)(n0 + im) * C + c0) * 16 + 4 * j);
    }
  };
  f32x4 acc[8];
#pragma unroll
  for (int q = 0; q < 8; q++) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // COMB: no launch has turned the producer's rows into the (scale, shift) table -- a wave needs the
  // pairs of its own sixteen channels only: lane (g, c) adds up rows g, g + 4, ... of channel cb + c (requested in
  // front of the first chunk), the four groups meet through two lane exchanges, and the operand lanes fetch their
  // channel's pair from lane c with a bpermute; no barrier, nothing through LDS.  Workgroup (0, 0, 0) leaves the
  // table and the statistics behind for the backward pass.
  constexpr int BRP = 8;
  f32x3u bn_row[COMB ? BRP : 1];
  float bn_shift = 0.0f, bn_g = 1.0f, bn_b = 0.0f, bn_rm = 0.0f, bn_rv = 0.0f;
  constexpr bool combine = COMB;
  if constexpr (combine) {
    const float* pc = bn.part + (cb + n) * 3;
#pragma unroll
    for (int k2 = 0; k2 < BRP; k2++)
      bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(kk + 4 * k2, bn.rows - 1) * (C * 3));
    bn_shift = pc[1];
    if (bn.gamma) bn_g = bn.gamma[cb + n];
    if (bn.beta) bn_b = bn.beta[cb + n];
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {     // (only the workgroup that updates them)
      if (bn.running_mean) bn_rm = bn.running_mean[cb + n];
      if (bn.running_var) bn_rv = bn.running_var[cb + n];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  request(0);
  f32x2 bn_pair = {1.0f, 0.0f};
  auto finish_table = [&]() {
    float tn = 0.0f, t1 = 0.0f, t2 = 0.0f;
    auto add_rows = [&](int r0) {
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++) {
        const bool in = r0 + kk + 4 * k2 < bn.rows;
        const float nb = in ? bn_row[k2][0] : 0.0f, d = nb > 0.0f ? bn_row[k2][1] - bn_shift : 0.0f;
        tn += nb;
        t1 = fmaf(nb, d, t1);
        t2 += fmaf(nb * d, d, in ? bn_row[k2][2] : 0.0f);
      }
    };
    add_rows(0);
    for (int r0 = 4 * BRP; r0 < bn.rows; r0 += 4 * BRP) {
      const float* pc = bn.part + (cb + n) * 3;
#pragma unroll
      for (int k2 = 0; k2 < BRP; k2++)
        bn_row[k2] = *reinterpret_cast<const f32x3u*>(pc + (int64_t)min(r0 + kk + 4 * k2, bn.rows - 1) * (C * 3));
      add_rows(r0);
    }
    // groups 0 + 1 and 2 + 3, then the two sums: the same order in every lane
    auto xlane = [&](float v, int m) { return __shfl_xor(v, m, 64); };
    tn += xlane(tn, 16); t1 += xlane(t1, 16); t2 += xlane(t2, 16);
    tn += xlane(tn, 32); t1 += xlane(t1, 32); t2 += xlane(t2, 32);
    const float mu_s = tn > 0.0f ? t1 / tn : 0.0f;
    const float mu = bn_shift + mu_s;
    const float var = tn > 0.0f ? fmaxf(t2 - t1 * mu_s, 0.0f) / tn : 0.0f;            // biased variance (batchnorm.py:41-42)
    const float rsd = 1.0f / sqrtf(var + bn.eps);
    bn_pair[0] = bn_g * rsd;
    bn_pair[1] = bn_b - mu * bn_pair[0];
    if (kk == 0) { bn_wtab[wave][n][0] = bn_pair[0]; bn_wtab[wave][n][1] = bn_pair[1]; }   // (read behind the wave barriers below)
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && kk == 0) {
      const int c = cb + n;
      if (bn.save_mean) bn.save_mean[c] = mu;
      if (bn.save_var) bn.save_var[c] = var;
      if (bn.running_mean) bn.running_mean[c] = bn_rm * (1.0f - bn.momentum) + mu * bn.momentum;
      if (bn.running_var) bn.running_var[c] = bn_rv * (1.0f - bn.momentum) + var * bn.momentum;
      bn.affine[2 * c] = bn_pair[0];
      bn.affine[2 * c + 1] = bn_pair[1];
    }
  };
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
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = lane + 64 * i, im = e >> 5, j = e & 31;
      *reinterpret_cast<f32x4*>(Xs + (im * XP + j) * 4) = xq[i];
    }
    const int c0 = cb + 8 * ch;
    if (ch + 1 < NCHK) request(ch + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float a[2][9];
    f32x4 xr[2][4];                              // the four source rows 2h - 1 .. 2h + 2 (one of them outside: never used)
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
      for (int t = 0; t < 9; t++) a[p][t] = Wl[((4 * p + kk) * 9 + t) * WP + n];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = min(max(2 * h - 1 + r, 0), 3);
        xr[p][r] = *reinterpret_cast<const f32x4*>(Xs + (n * XP + (4 * p + kk) * 4 + row) * 4);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (combine) {
      // (the sums are finished HERE, behind the first slice's commit: the rows' latency hides under it)
      if (ch == 0) {
        finish_table();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (aff) {
#pragma unroll
      for (int p = 0; p < 2; p++) {
        f32x2 sc;
        if constexpr (combine) {
          sc = *reinterpret_cast<const f32x2*>(&bn_wtab[wave][8 * ch + 4 * p + kk][0]);
        } else {
          sc = *reinterpret_cast<const f32x2*>(aff + 2 * (c0 + 4 * p + kk));
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) xr[p][r][e] = fmaf(xr[p][r][e], sc[0], sc[1]);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
      if (h == 0) img4_fwd_mfmas<0>(a[p], xr[p], acc);
      else img4_fwd_mfmas<1>(a[p], xr[p], acc);
    }
  }
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(smem_i4);
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 8; q++) ex[((wave - 1) * 8 + q) * 64 + lane] = acc[q];
  }
  __syncthreads();
  if (wave != 0) return;
  // (one wave's tile at a time: fully unrolled the compiler requests all 56 vectors before the first addition and
  // spills)
#pragma unroll 1
  for (int ww = 0; ww < 7; ww++)
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] += ex[(ww * 8 + q) * 64 + lane];
  // register v of lane (kk, n) is output channel o0 + 4 kk + v of image n0 + n; a row's four pixels are adjacent
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const float bv = bias ? bias[o0 + 4 * kk + v] : 0.0f;
#pragma unroll
    for (int yy = 0; yy < 2; yy++) {
      f32x4 r = {acc[yy * 4][v] + bv, acc[yy * 4 + 1][v] + bv, acc[yy * 4 + 2][v] + bv, acc[yy * 4 + 3][v] + bv};
      acc[yy * 4][v] = r[0]; acc[yy * 4 + 1][v] = r[1]; acc[yy * 4 + 2][v] = r[2]; acc[yy * 4 + 3][v] = r[3];
      *reinterpret_cast<f32x4*>(y + ((int64_t)(n0 + n) * O + o0 + 4 * kk + v) * 16 + (2 * h + yy) * 4) = r;
    }
  }
  if (stats) {
    // exact two-pass over the 128 values of a channel: the lane's eight pixels, then the 16 images of its DPP row
#pragma unroll
    for (int v = 0; v < 4; v++) {
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; q++) t += acc[q][v];
      const float mean = row_sum16(t) * (1.0f / 128.0f);
      float qs = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; q++) { const float d = acc[q][v] - mean; qs = fmaf(d, d, qs); }
      qs = row_sum16(qs);
      if (n == 0) {
        float* o3 = stats + ((int64_t)(wg.x * 2 + h) * O + o0 + 4 * kk + v) * 3;
        o3[0] = 128.0f; o3[1] = mean; o3[2] = qs;
      }
    }
  }
}

}  // namespace

namespace dfhip {

// 4 x 4 images with an image on the lane (k_conv3x3_img4_fwd); -1 when the shape is not one it takes
static int run_conv_img4_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int O,
                             float* partials, int cap_rows, int* rows_out, const BnInput* bn, bool dry_run) {
  if (C != 128 || N % 16 != 0 || O % 16 != 0) return -1;
  const int rows = N / 16 * 2;
  if (partials && rows > cap_rows) return -1;
  if ((int64_t)rows * (O / 16) < num_cus() / 2) return -1;
  if (bn && !bn->affine) return -1;
  if (rows_out) *rows_out = partials ? rows : 0;
  if (dry_run) return DFHIP_OK;
  dim3 grid((unsigned)(N / 16), (unsigned)(O / 16), 2);
  const size_t lds = std::max((size_t)8 * (72 * 17 + 16 * 33 * 4), (size_t)7 * 8 * 64 * 4) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DF_HIP(hipFuncSetAttribute((const void*)k_conv3x3_img4_fwd<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DF_HIP(hipFuncSetAttribute((const void*)k_conv3x3_img4_fwd<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  // blockIdx.y selects 16 output channels' weights, blockIdx.x 16 images' planes (both halves of an image tile are
  // neighbours in z: they share the planes of the middle rows)
  if (bn)
    DF_LAUNCH((k_conv3x3_img4_fwd<128, true>), grid, dim3(512), lds, stream(), x, w, bias, y, partials, O,
              (const float*)bn->affine, *bn);
  else
    DF_LAUNCH((k_conv3x3_img4_fwd<128, false>), grid, dim3(512), lds, stream(), x, w, bias, y, partials, O,
              bn ? (const float*)bn->affine : (const float*)nullptr, BnInput{});
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// returns -1 when the shape is not one this kernel takes
int run_conv_rowmma(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int O,
                    int k, int stride, int pad, float* partials, int cap_rows, int* rows_out, const BnInput* bn,
                    bool dry_run) {
  if (k != 3 || stride != 1 || pad != 1) return -1;
  if (W == 4 && H == 4 && C == 128) {
    const int st4 = run_conv_img4_fwd(x, w, bias, y, N, C, O, partials, cap_rows, rows_out, bn, dry_run);
    if (st4 >= 0) return st4;
  }
  const int64_t HW = (int64_t)H * W;
  if (HW % 16 != 0) return -1;
  const int64_t tiles = (int64_t)N * HW / 16;
  // (width, channels) -> tiles per wave, tile sets x channel tiles per workgroup
  int nt, ts, ow;
  // 16- and 8-wide layers that leave BatchNorm partials or read a BatchNorm operand: eight tile sets x one channel tile per
  // workgroup (twice the pixel loads, half the weight image and half as many partial rows for the consumer of the statistics:
  // 15.1 -> 14.5 and 16.1 -> 15.2 us plain, -0.4 us with a BatchNorm operand); without statistics four tile sets x two tiles.
  // Measured and dropped (rounds 2-5): weight chunks of eight channel groups, two chunks of pixels ahead, every weight chunk
  // committed in the prologue (one barrier), half the tiles per wave on twice the workgroups (+4.5 us with a BatchNorm operand).
  const bool wide = (!bn && partials) || bn;
  if (W == 16 && C == 32) { nt = 4; ts = wide ? 8 : 4; ow = wide ? 1 : 2; }
  else if (W == 8 && C == 64) { nt = 2; ts = wide ? 8 : 4; ow = wide ? 1 : 2; }
  else if (W == 4 && H == 4 && C == 128) { nt = 1; ts = 8; ow = 1; }
  else return -1;
  if (O % (16 * ow) != 0 || tiles % (ts * nt) != 0 || (W == 16 && H % nt != 0)) return -1;
  const int64_t gx = tiles / (ts * nt);
  if (partials && gx > cap_rows) return -1;
  if (gx * (O / (16 * ow)) < num_cus() / 2) return -1;   // too few workgroups: the staged kernel's k-split wins
  if (rows_out) *rows_out = partials ? (int)gx : 0;
  dim3 grid((unsigned)gx, (unsigned)(O / (16 * ow)));
  if (dry_run) return DFHIP_OK;              // eligible: nothing launched
  // opt-in bf16 contraction (dfhip_set_conv_matmul_dtype(1)): the eight-tile-set forms of the 16- and 8-wide layers
  const bool bf = conv_matmul_dtype_is_bf16() && wide;
  if (bn) {
    if (!bn->affine) return -1;
    if (W == 16 && bf) return launch_rowmma<4, 32, 4, false, 8, 1, 4, false, true, true>(x, w, bias, y, partials, H, O, grid, bn);
    if (W == 8 && bf) return launch_rowmma<3, 64, 2, false, 8, 1, 4, false, true, true>(x, w, bias, y, partials, H, O, grid, bn);
    if (W == 16) return launch_rowmma<4, 32, 4, false, 8, 1, 4, false, true>(x, w, bias, y, partials, H, O, grid, bn);
    if (W == 8) return launch_rowmma<3, 64, 2, false, 8, 1, 4, false, true>(x, w, bias, y, partials, H, O, grid, bn);
    return launch_rowmma<2, 128, 1, true, 8, 1, 8, false, true>(x, w, bias, y, partials, H, O, grid, bn);
  }
  if (W == 16 && bf) return launch_rowmma<4, 32, 4, false, 8, 1, 4, false, false, true>(x, w, bias, y, partials, H, O, grid);
  if (W == 8 && bf) return launch_rowmma<3, 64, 2, false, 8, 1, 4, false, false, true>(x, w, bias, y, partials, H, O, grid);
  if (W == 16) return wide ? launch_rowmma<4, 32, 4, false, 8, 1, 4>(x, w, bias, y, partials, H, O, grid)
                           : launch_rowmma<4, 32, 4, false, 4, 2, 4>(x, w, bias, y, partials, H, O, grid);
  if (W == 8) return wide ? launch_rowmma<3, 64, 2, false, 8, 1, 4>(x, w, bias, y, partials, H, O, grid)
                          : launch_rowmma<3, 64, 2, false, 4, 2, 4>(x, w, bias, y, partials, H, O, grid);
  return launch_rowmma<2, 128, 1, true, 8, 1, 8>(x, w, bias, y, partials, H, O, grid);
}


// the accumulate-mode input gradient of a 3x3 / stride 1 / pad 1 convolution on the same kernel (DG): dy (N, O, H, W),
// w (O, C, 3, 3) -> dx (N, C, H, W) (+ add); -1 when the shape is not eligible
int run_conv_rowmma_dgrad(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H, int W,
                          int O, int k, int stride, int pad) {
  if (k != 3 || stride != 1 || pad != 1) return -1;
  const int64_t HW = (int64_t)H * W;
  if (HW % 16 != 0) return -1;
  const int64_t tiles = (int64_t)N * HW / 16;
  // (eight tile sets x one channel tile per workgroup on the 16- and 8-wide layers, as the forward launches)
  int nt, ts = 8, ow = 1;
  if (W == 16 && O == 32) nt = 4;
  else if (W == 8 && O == 64) nt = 2;
  else if (W == 4 && H == 4 && O == 128) nt = 1;
  else return -1;
  if (C % (16 * ow) != 0 || tiles % (ts * nt) != 0 || (W == 16 && H % nt != 0)) return -1;
  const int64_t gx = tiles / (ts * nt);
  if (gx * (C / (16 * ow)) < num_cus() / 2) return -1;
  dim3 grid((unsigned)gx, (unsigned)(C / (16 * ow)));
  if (W == 16) return launch_rowmma<4, 32, 4, false, 8, 1, 4, true>(dy, w, add, dx, nullptr, H, C, grid);
  if (W == 8) return launch_rowmma<3, 64, 2, false, 8, 1, 4, true>(dy, w, add, dx, nullptr, H, C, grid);
  return launch_rowmma<2, 128, 1, true, 8, 1, 8, true>(dy, w, add, dx, nullptr, H, C, grid);
}

}  // namespace dfhip
