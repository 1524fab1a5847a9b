// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32): the `matmul` primitive
// (reference: ndarray_backend_cuda.cu:443-466, a 4x4-thread naive kernel with a global-memory
// K loop) and the transposed-operand forms matmul.grad_fn needs (tensor.py:699-716), which in
// the reference cost a compact() copy of the transposed operand before every call.
//
// Structure (per 256-thread workgroup = 4 waves in a 2x2 grid):
//   * block tile (2*TM*32) x (2*TN*32), K step 32, two LDS buffers;
//   * global -> registers -> LDS staging with the lanes running along whichever operand
//     dimension is contiguous in memory (so both the plain and the transposed forms read
//     coalesced); the next K step is fetched into registers while the current one is in the
//     matrix cores, one barrier per step;
//   * LDS tiles are K-major ([k][row], row contiguous), so a fragment read is 32 consecutive
//     floats per half-wave: conflict-free ds_read_b32, one read per operand per MFMA;
//   * the f32 MFMA is exact fp32 (a k-ordered fmaf chain), 64 cycles per instruction per SIMD:
//     one wave per SIMD with its TM*TN accumulator tiles already saturates the pipe;
//   * small outputs with a long reduction are split along K into slabs that a second kernel
//     sums in a fixed order (deterministic, no float atomics).
#include <algorithm>

#include "dfhip_internal.h"

using namespace dfhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

template <int TM, int TN, bool A_KCONTIG, bool B_KCONTIG>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A,
                                              const float* __restrict__ B,
                                              const float* __restrict__ bias,
                                              float* __restrict__ C, int M, int K, int P,
                                              int64_t a_sm, int64_t a_sk, int64_t b_sk,
                                              int64_t b_sp, int k_per_split, int64_t slab_stride) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int EA = BM * BK / 256, EB = BN * BK / 256;  // elements per thread per step
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                   // [2][BK][LDA]
  float* Bs = smem + 2 * BK * LDA;    // [2][BK][LDB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, p0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  C += (int64_t)blockIdx.z * slab_stride;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  float ra[EA], rb[EB];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; e++) {
      int r, k;
      if (A_KCONTIG) { k = tid & 31; r = (tid >> 5) + 8 * e; }
      else { r = tid % BM; k = tid / BM + (256 / BM) * e; }
      int gm = m0 + r, gk = k0 + k;
      ra[e] = (gm < M && gk < kend) ? A[(int64_t)gm * a_sm + (int64_t)gk * a_sk] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < EB; e++) {
      int r, k;
      if (B_KCONTIG) { k = tid & 31; r = (tid >> 5) + 8 * e; }
      else { r = tid % BN; k = tid / BN + (256 / BN) * e; }
      int gp = p0 + r, gk = k0 + k;
      rb[e] = (gp < P && gk < kend) ? B[(int64_t)gk * b_sk + (int64_t)gp * b_sp] : 0.0f;
    }
  };
  auto store_tile = [&](int buf) {
    float* as = As + buf * BK * LDA;
    float* bs = Bs + buf * BK * LDB;
#pragma unroll
    for (int e = 0; e < EA; e++) {
      int r, k;
      if (A_KCONTIG) { k = tid & 31; r = (tid >> 5) + 8 * e; }
      else { r = tid % BM; k = tid / BM + (256 / BM) * e; }
      as[k * LDA + r] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < EB; e++) {
      int r, k;
      if (B_KCONTIG) { k = tid & 31; r = (tid >> 5) + 8 * e; }
      else { r = tid % BN; k = tid / BN + (256 / BN) * e; }
      bs[k * LDB + r] = rb[e];
    }
  };

  const int nsteps = (kend - kbeg + BK - 1) / BK;
  if (nsteps > 0) {
    load_tile(kbeg);
    store_tile(0);
  }
  __syncthreads();
  const int half = lane >> 5, l31 = lane & 31;
  for (int s = 0; s < nsteps; s++) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_tile(kbeg + (s + 1) * BK);
    const float* as = As + buf * BK * LDA + half * LDA + wm * (TM * 32) + l31;
    const float* bs = Bs + buf * BK * LDB + half * LDB + wn * (TN * 32) + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = as[kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; j++) b[j] = bs[kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nsteps) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      int col = p0 + (wn * TN + j) * 32 + l31;
      float bv = (bias != nullptr && col < P) ? bias[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M && col < P) C[(int64_t)row * P + col] = acc[i][j][r] + bv;
      }
    }
}

// Launch-bound contractions (a 256 x 256 x 10 classifier and its gradients): one launch, no LDS
// staging, no slab kernel.  A workgroup owns a 32 x 32 output tile, its eight waves split K into
// eight contiguous ranges and read their operands straight from global memory (everything is
// L2-resident at these sizes; all loads of a wave are independent), the eight partial tiles are
// summed through LDS in wave order.
// relu_out (or nullptr): max(C, 0) is stored there as well (F.relu(F.linear(...)): the pre-activation stays for the
// backward pass, which passes the gradient where output == input, tensor.py:872-877).
// MASK: which operand is an upstream gradient to be taken only where a ReLU passes it -- 1 = A, 2 = B: element e of
// that operand counts only where mr[e] == mz[e] (mr = the ReLU's output, mz = its input, laid out like the operand).
template <int MASK = 0>
__device__ __forceinline__ void gemm_small_tile(const float* __restrict__ A, const float* __restrict__ B,
                                                const float* __restrict__ bias, float* __restrict__ C, int M, int K,
                                                int P, int64_t a_sm, int64_t a_sk, int64_t b_sk, int64_t b_sp,
                                                int bx, int by, float (*red)[16][64],
                                                float* __restrict__ relu_out = nullptr,
                                                const float* __restrict__ mz = nullptr,
                                                const float* __restrict__ mr = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int m = by * 32 + l31, p = bx * 32 + l31;
  const int kq = (((K + 7) / 8) + 3) & ~3;                 // k per wave: a multiple of 4 (16-byte operand vectors)
  const int kbeg = wave * kq, kend = min(K, kbeg + kq);
  const int64_t a0 = (int64_t)min(m, M - 1) * a_sm, b0 = (int64_t)min(p, P - 1) * b_sp;
  const float* ap = A + a0;
  const float* bp = B + b0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
  // an operand whose k runs along memory (a lane = a row of 784 floats, say) is read 16 bytes at a time: a scalar load
  // per (lane, k) there is a wave load that touches 64 lines for 4 bytes each, and the address unit, not the matrix
  // cores, paces the loop; both half-waves read the same vector and take the k of their parity out of it
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const bool avec = MASK != 1 && a_sk == 1 && (a_sm & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (kbeg & 3) == 0;
  const bool bvec = MASK != 2 && b_sk == 1 && (b_sp & 3) == 0 && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (kbeg & 3) == 0;
  // (measured and dropped: every load unconditional at a clamped address and four 16-k steps requested together --
  // 64 loads in flight per wave -- ran 1.5-2x SLOWER than this loop on every shape of the MLP workload)
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    float a[8], b[8];
    if (avec) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int kq4 = k0 + 4 * q;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (kq4 + 3 < kend) t = *reinterpret_cast<const f32x4*>(ap + kq4);
        else {
#pragma unroll
          for (int e = 0; e < 4; e++) t[e] = kq4 + e < kend ? ap[kq4 + e] : 0.0f;
        }
        a[2 * q] = m < M ? (half ? t[1] : t[0]) : 0.0f;
        a[2 * q + 1] = m < M ? (half ? t[3] : t[2]) : 0.0f;
      }
    }
    if (bvec) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int kq4 = k0 + 4 * q;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (kq4 + 3 < kend) t = *reinterpret_cast<const f32x4*>(bp + kq4);
        else {
#pragma unroll
          for (int e = 0; e < 4; e++) t[e] = kq4 + e < kend ? bp[kq4 + e] : 0.0f;
        }
        b[2 * q] = p < P ? (half ? t[1] : t[0]) : 0.0f;
        b[2 * q + 1] = p < P ? (half ? t[3] : t[2]) : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int k = k0 + 2 * u + half;
      const bool ok = k < kend;
      if (!avec) a[u] = (ok && m < M) ? ap[(int64_t)k * a_sk] : 0.0f;
      if (!bvec) b[u] = (ok && p < P) ? bp[(int64_t)k * b_sk] : 0.0f;
      if constexpr (MASK == 1) {
        const int64_t e = a0 + (int64_t)min(k, kend - 1) * a_sk;
        a[u] = (mr[e] == mz[e]) ? a[u] : 0.0f;
      }
      if constexpr (MASK == 2) {
        const int64_t e = b0 + (int64_t)min(k, kend - 1) * b_sk;
        b[u] = (mr[e] == mz[e]) ? b[u] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    for (int w = 0; w < 7; w++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] += red[w][r][lane];
    if (p < P) {
      const float bb = bias ? bias[p] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = by * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M) {
          const float z = acc[r] + bb;
          C[(int64_t)row * P + p] = z;
          if (relu_out) relu_out[(int64_t)row * P + p] = (z != z) ? z : (z > 0.0f ? z : 0.0f);
        }
      }
    }
  }
}

__global__ __launch_bounds__(512) void k_gemm_small(const float* __restrict__ A, const float* __restrict__ B,
                                                    const float* __restrict__ bias, float* __restrict__ C,
                                                    int M, int K, int P, int64_t a_sm, int64_t a_sk,
                                                    int64_t b_sk, int64_t b_sp, float* __restrict__ relu_out) {
  __shared__ float red[7][16][64];
  gemm_small_tile(A, B, bias, C, M, K, P, a_sm, a_sk, b_sk, b_sp, blockIdx.x, blockIdx.y, red, relu_out);
}

// Backward of a small Linear layer (y = x.W + b, linear.py:28-59 / tensor.py:699-716) as ONE launch:
// blocks [0, nx) compute tiles of dX = dY.W^T, blocks [nx, nx + nw) tiles of dW = X^T.dY, the last
// block db = column sums of dY (each output may be null).  Three launch-bound launches otherwise.
// MASKED: dy is the gradient of relu(y) -- the layer's activation was fused into its forward launch -- and counts
// only where mr == mz (mr = relu(y), mz = y; both M x P like dy): the mask is applied on the operand loads.
template <bool MASKED>
__global__ __launch_bounds__(512) void k_linear_bwd_small(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ dw, float* __restrict__ db, int M, int K,
                                                          int P, int nx, int nw, const float* __restrict__ mz,
                                                          const float* __restrict__ mr) {
  __shared__ float red[7][16][64];
  const int b = blockIdx.x;
  if (b < nx) {                                     // dX (M x K) = dY (M x P) . W^T (P x K);  W is (K, P)
    const int tx = (K + 31) / 32;
    gemm_small_tile<MASKED ? 1 : 0>(dy, w, nullptr, dx, M, P, K, P, 1, 1, P, b % tx, b / tx, red, nullptr, mz, mr);
  } else if (b < nx + nw) {                         // dW (K x P) = X^T (K x M) . dY (M x P);  X is (M, K)
    const int t = b - nx, tx = (P + 31) / 32;
    gemm_small_tile<MASKED ? 2 : 0>(x, dy, nullptr, dw, K, M, P, 1, K, P, 1, t % tx, t / tx, red, nullptr, mz, mr);
  } else {                                          // db[p] = sum_m dY[m][p]: a block per 32 columns, 16 row groups
    const int pc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int p = (b - nx - nw) * 32 + pc;
    float acc = 0.0f;
    for (int m0 = rg; m0 < M; m0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {               // (clamped, unconditional: the eight loads are in flight together)
        const int64_t e = (int64_t)min(m0 + 16 * u, M - 1) * P + min(p, P - 1);
        v[u] = dy[e];
        if (MASKED) v[u] = (mr[e] == mz[e]) ? v[u] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (m0 + 16 * u < M) acc += v[u];
    }
    red[0][rg][pc] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && p < P) {
      float t = 0.0f;
      for (int r = 0; r < 16; r++) t += red[0][r][pc];
      db[p] = t;
    }
  }
}

// out[i] = sum_s slabs[s][i] (+ bias[i % P]) in slab order
__global__ __launch_bounds__(256) void k_slab_sum(const float* __restrict__ slabs,
                                                  const float* __restrict__ bias,
                                                  float* __restrict__ out, int64_t n, int splits,
                                                  int P) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    float v = slabs[i];
    for (int s = 1; s < splits; s++) v += slabs[(int64_t)s * n + i];
    if (bias) v += bias[i % P];
    out[i] = v;
  }
}

template <int TM, int TN>
int launch_cfg(const float* a, const float* b, const float* bias, float* out, int M, int K, int P,
               int ta, int tb, int splits) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  size_t lds = (size_t)2 * BK * ((BM + 1) + (BN + 1)) * sizeof(float);
  int64_t a_sm = ta ? 1 : K, a_sk = ta ? M : 1;
  int64_t b_sk = tb ? 1 : P, b_sp = tb ? K : 1;
  int k_per_split = (int)df_ceil_div(df_ceil_div(K, splits), BK) * BK;
  if (k_per_split < BK) k_per_split = BK;
  splits = (int)std::max<int64_t>(1, df_ceil_div(K, k_per_split));
  float* dst = out;
  int64_t slab = 0;
  if (splits > 1) {
    slab = (int64_t)M * P;
    dst = workspace((size_t)slab * splits * sizeof(float));
    if (!dst) return DFHIP_OUT_OF_MEMORY;
  }
  dim3 grid((unsigned)df_ceil_div(P, BN), (unsigned)df_ceil_div(M, BM), (unsigned)splits);
  const float* kbias = splits > 1 ? nullptr : bias;
#define DF_GEMM_LAUNCH(AK, BKC)                                                                   \
  do {                                                                                            \
    static bool attr_set = false;                                                                 \
    if (!attr_set) {                                                                              \
      DF_HIP(hipFuncSetAttribute((const void*)k_gemm<TM, TN, AK, BKC>,                            \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
      attr_set = true;                                                                            \
    }                                                                                             \
    DF_LAUNCH((k_gemm<TM, TN, AK, BKC>), grid, dim3(256), lds, stream(), a, b, kbias,    \
                       dst, M, K, P, a_sm, a_sk, b_sk, b_sp, k_per_split, slab);                  \
  } while (0)
  bool ak = !ta, bk = tb != 0;
  if (ak && bk) DF_GEMM_LAUNCH(true, true);
  else if (ak && !bk) DF_GEMM_LAUNCH(true, false);
  else if (!ak && bk) DF_GEMM_LAUNCH(false, true);
  else DF_GEMM_LAUNCH(false, false);
#undef DF_GEMM_LAUNCH
  DF_LAUNCH_CHECK();
  if (splits > 1) {
    int64_t n = slab;
    int blocks = (int)std::min<int64_t>(df_ceil_div(n, 256), (int64_t)num_cus() * 8);
    DF_LAUNCH(k_slab_sum, dim3(blocks), dim3(256), 0, stream(), dst, bias, out, n, splits, P);
    DF_LAUNCH_CHECK();
  }
  return DFHIP_OK;
}

// contractions the one-launch kernels take: at most kSmallFlops of work and a reduction of 64 .. 4 096 (the eight waves
// of a tile split it); everything at these sizes is L2-resident and the alternative is a k-split launch + a slab sum
constexpr double kSmallFlops = 1.5e8;

int gemm_dispatch(const float* a, const float* b, const float* bias, float* out, int64_t m,
                  int64_t k, int64_t p, int ta, int tb, float* relu_out = nullptr) {
  DF_INIT();
  DF_CHECK_ARG(m >= 0 && k >= 0 && p >= 0, "matmul: negative dimension");
  DF_CHECK_ARG(m < ((int64_t)1 << 31) && k < ((int64_t)1 << 31) && p < ((int64_t)1 << 31),
               "matmul: dimension too large");
  if (m == 0 || p == 0) return DFHIP_OK;
  DF_CHECK_ARG(out && (k == 0 || (a && b)), "matmul: null array");
  ProfScope prof("gemm_mfma_f32", 2.0 * (double)m * (double)k * (double)p,
                 4.0 * ((double)m * k + (double)k * p + (double)m * p));
  int M = (int)m, K = (int)k, P = (int)p;
  int cus = num_cus();
  const bool small = 2.0 * (double)m * (double)k * (double)p < (relu_out ? kSmallFlops : 8.0e6) && K >= 64 && K <= 4096;
  if (small) {
    DF_LAUNCH(k_gemm_small, dim3((unsigned)df_ceil_div(P, 32), (unsigned)df_ceil_div(M, 32)), dim3(512), 0,
                       stream(), a, b, bias, out, M, K, P, ta ? (int64_t)1 : (int64_t)K, ta ? (int64_t)M : (int64_t)1,
                       tb ? (int64_t)1 : (int64_t)P, tb ? (int64_t)K : (int64_t)1, relu_out);
    DF_LAUNCH_CHECK();
    return DFHIP_OK;
  }
  DF_CHECK_ARG(!relu_out, "gemm: the fused activation exists on the one-launch kernel only");
  int64_t t128 = df_ceil_div(M, 128) * df_ceil_div(P, 128);
  int64_t t64 = df_ceil_div(M, 64) * df_ceil_div(P, 64);
  if (t128 >= 2 * (int64_t)cus && P > 64) return launch_cfg<2, 2>(a, b, bias, out, M, K, P, ta, tb, 1);
  if (P <= 64 && df_ceil_div(M, 128) >= 2 * (int64_t)cus)
    return launch_cfg<2, 1>(a, b, bias, out, M, K, P, ta, tb, 1);
  int splits = 1;
  if (t64 < (int64_t)cus * 2 && K >= 256) {
    int64_t want = df_ceil_div((int64_t)cus * 4, t64);
    int64_t maxs = K / 128;
    splits = (int)std::max<int64_t>(1, std::min<int64_t>(want, maxs));
  }
  return launch_cfg<1, 1>(a, b, bias, out, M, K, P, ta, tb, splits);
}

}  // namespace

extern "C" {

int dfhip_matmul(const float* a, const float* b, float* out, int64_t m, int64_t n, int64_t p) {
  return gemm_dispatch(a, b, nullptr, out, m, n, p, 0, 0);
}

int dfhip_gemm(const float* a, const float* b, const float* bias, float* out, int64_t m, int64_t n,
               int64_t p, int ta, int tb) {
  return gemm_dispatch(a, b, bias, out, m, n, p, ta, tb);
}

static int linear_backward_small(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                 int64_t m, int64_t k, int64_t p, const float* mz, const float* mr) {
  ProfScope prof("gemm_mfma_f32", 2.0 * (double)m * (double)k * (double)p * ((dx ? 1 : 0) + (dw ? 1 : 0)),
                 4.0 * (2.0 * (double)m * k + 2.0 * (double)k * p + 2.0 * (double)m * p));
  const int nx = dx ? (int)(df_ceil_div(k, 32) * df_ceil_div(m, 32)) : 0;
  const int nw = dw ? (int)(df_ceil_div(p, 32) * df_ceil_div(k, 32)) : 0;
  const int nb = db ? (int)df_ceil_div(p, 32) : 0;
  if (nx + nw + nb == 0) return DFHIP_OK;
  if (mz) DF_LAUNCH(k_linear_bwd_small<true>, dim3((unsigned)(nx + nw + nb)), dim3(512), 0, stream(), x, w, dy, dx, dw, db,
                    (int)m, (int)k, (int)p, nx, nw, mz, mr);
  else DF_LAUNCH(k_linear_bwd_small<false>, dim3((unsigned)(nx + nw + nb)), dim3(512), 0, stream(), x, w, dy, dx, dw, db,
                 (int)m, (int)k, (int)p, nx, nw, mz, mr);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_linear_relu_forward(const float* x, const float* w, const float* bias, float* z, float* r, int64_t m,
                              int64_t k, int64_t p, int* done) {
  DF_CHECK_ARG(x && w && z && r && done, "linear_relu_forward: null array");
  *done = 0;
  if (!(2.0 * (double)m * (double)k * (double)p < kSmallFlops && k >= 64 && k <= 4096 && m > 0 && p > 0))
    return DFHIP_OK;                               // no one-launch kernel for this size: nothing is launched
  int st = gemm_dispatch(x, w, bias, z, m, k, p, 0, 0, r);
  if (st == DFHIP_OK) *done = 1;
  return st;
}

int dfhip_linear_relu_backward(const float* x, const float* w, const float* z, const float* r, const float* dr,
                               float* dx, float* dw, float* db, int64_t m, int64_t k, int64_t p, int* done) {
  DF_INIT();
  DF_CHECK_ARG(m > 0 && k > 0 && p > 0, "linear_relu_backward: non-positive dimension");
  DF_CHECK_ARG(z && r && dr && done && (!dx || w) && (!dw || x), "linear_relu_backward: null array");
  *done = 0;
  // the dW tiles reduce over m, the dX tiles over p
  if (2.0 * (double)m * (double)k * (double)p >= kSmallFlops || m > 4096 || p > 4096) return DFHIP_OK;
  int st = linear_backward_small(x, w, dr, dx, dw, db, m, k, p, z, r);
  if (st == DFHIP_OK) *done = 1;
  return st;
}

int dfhip_linear_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                          int64_t m, int64_t k, int64_t p) {
  DF_INIT();
  DF_CHECK_ARG(m > 0 && k > 0 && p > 0, "linear_backward: non-positive dimension");
  DF_CHECK_ARG(dy && (!dx || w) && (!dw || x), "linear_backward: null array");
  if (2.0 * (double)m * (double)k * (double)p < kSmallFlops && m <= 4096 && p <= 4096)
    return linear_backward_small(x, w, dy, dx, dw, db, m, k, p, nullptr, nullptr);
  // larger layers: the three contractions on the tiled kernel
  if (dx) { int st = gemm_dispatch(dy, w, nullptr, dx, m, p, k, 0, 1); if (st != DFHIP_OK) return st; }
  if (dw) { int st = gemm_dispatch(x, dy, nullptr, dw, k, m, p, 1, 0); if (st != DFHIP_OK) return st; }
  if (db) {
    const int64_t src_shape[2] = {m, p}, dst_shape[2] = {1, p};
    return dfhip_reduce_to_shape(dy, db, 2, src_shape, 2, dst_shape);
  }
  return DFHIP_OK;
}

}  // extern "C"
