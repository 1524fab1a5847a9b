// The memory-bound device primitives of the BackendTensor protocol, hand-written for
// gfx950: fill, compact, ewise_setitem, scalar_setitem, 6 ewise-binary, 7 scalar, 3 unary,
// reduce_sum, reduce_max (reference kernels: ndarray_backend_cuda.cu:127-432, 469-509;
// value semantics: backend_tensor.py:66-170 NumpyBackend).
//
// All of these are HBM-bound (8-12 B per element): 16-byte per-lane accesses, 256-thread
// workgroups, grid capped at a few blocks per CU with a grid-stride loop, wave64 shuffles
// for the row reductions.
#include <math.h>

#include "dfhip_internal.h"

using namespace dfhip;

namespace {

constexpr int kThreads = 256;

inline int grid_for(int64_t work_items, int per_thread = 1) {
  int64_t blocks = df_ceil_div(work_items, (int64_t)kThreads * per_thread);
  int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- elementwise functors ---------------------------------------------------------------
struct OpAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct OpMul { __device__ float operator()(float a, float b) const { return a * b; } };
struct OpDiv { __device__ float operator()(float a, float b) const { return a / b; } };
// np.maximum propagates NaN (backend_tensor.py:127-131)
struct OpMax {
  __device__ float operator()(float a, float b) const {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
  }
};
struct OpEq { __device__ float operator()(float a, float b) const { return a == b ? 1.0f : 0.0f; } };
struct OpGe { __device__ float operator()(float a, float b) const { return a >= b ? 1.0f : 0.0f; } };
struct OpPow { __device__ float operator()(float a, float b) const { return powf(a, b); } };
struct OpLog { __device__ float operator()(float a) const { return logf(a); } };
struct OpExp { __device__ float operator()(float a) const { return expf(a); } };
struct OpTanh { __device__ float operator()(float a) const { return tanhf(a); } };

template <class Op>
__global__ __launch_bounds__(kThreads) void k_ewise(const float* __restrict__ a,
                                                    const float* __restrict__ b,
                                                    float* __restrict__ out, int64_t n, Op op) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (int64_t i = tid; i < n4; i += step) {
    float4 x = a4[i], y = b4[i], r;
    r.x = op(x.x, y.x); r.y = op(x.y, y.y); r.z = op(x.z, y.z); r.w = op(x.w, y.w);
    o4[i] = r;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) out[i] = op(a[i], b[i]);
}

template <class Op>
__global__ __launch_bounds__(kThreads) void k_scalar(const float* __restrict__ a, float s,
                                                     float* __restrict__ out, int64_t n, Op op) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (int64_t i = tid; i < n4; i += step) {
    float4 x = a4[i], r;
    r.x = op(x.x, s); r.y = op(x.y, s); r.z = op(x.z, s); r.w = op(x.w, s);
    o4[i] = r;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) out[i] = op(a[i], s);
}

template <class Op>
__global__ __launch_bounds__(kThreads) void k_unary(const float* __restrict__ a,
                                                    float* __restrict__ out, int64_t n, Op op) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (int64_t i = tid; i < n4; i += step) {
    float4 x = a4[i], r;
    r.x = op(x.x); r.y = op(x.y); r.z = op(x.z); r.w = op(x.w);
    o4[i] = r;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) out[i] = op(a[i]);
}

__global__ __launch_bounds__(kThreads) void k_fill(float* __restrict__ out, int64_t n, float v) {
  int64_t n4 = n >> 2;
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  float4* o4 = reinterpret_cast<float4*>(out);
  float4 r = make_float4(v, v, v, v);
  for (int64_t i = tid; i < n4; i += step) o4[i] = r;
  for (int64_t i = (n4 << 2) + tid; i < n; i += step) out[i] = v;
}

template <class Op, class Fn>
int launch_binary(const char* name, const float* a, const float* b, float* out, int64_t n, Fn) {
  DF_INIT_KEEP();          // gradient accumulation sits between two weight gradients of a backward pass
  DF_TOUCH(a);
  DF_TOUCH(b);
  DF_TOUCH(out);
  DF_CHECK_ARG(n >= 0, "%s: negative size", name);
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(a && b && out, "%s: null array", name);
  DF_CHECK_ARG(aligned16(a) && aligned16(b) && aligned16(out), "%s: arrays must be 16-byte aligned", name);
  ProfScope prof("ewise", 0.0, 12.0 * (double)n);
  DF_LAUNCH((k_ewise<Op>), dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), a, b, out, n, Op());
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

template <class Op>
int launch_scalar(const char* name, const float* a, float s, float* out, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0, "%s: negative size", name);
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(a && out, "%s: null array", name);
  DF_CHECK_ARG(aligned16(a) && aligned16(out), "%s: arrays must be 16-byte aligned", name);
  ProfScope prof("ewise", 0.0, 8.0 * (double)n);
  DF_LAUNCH((k_scalar<Op>), dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), a, s, out, n, Op());
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

template <class Op>
int launch_unary(const char* name, const float* a, float* out, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0, "%s: negative size", name);
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(a && out, "%s: null array", name);
  DF_CHECK_ARG(aligned16(a) && aligned16(out), "%s: arrays must be 16-byte aligned", name);
  ProfScope prof("ewise", 0.0, 8.0 * (double)n);
  DF_LAUNCH((k_unary<Op>), dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), a, out, n, Op());
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

// ---- strided views ------------------------------------------------------------------------
struct View {
  int ndim;
  int64_t shape[DFHIP_MAX_DIMS];
  int64_t strides[DFHIP_MAX_DIMS];
  int64_t offset;
};

// Collapse adjacent dims that are contiguous with each other and drop size-1 dims: most
// protocol calls (permutes of NCHW activations, stride-0 broadcasts) end up with 2-4 dims.
int make_view(const char* name, int ndim, const int64_t* shape, const int64_t* strides,
              int64_t offset, int64_t n_expected, View* v) {
  DF_CHECK_ARG(ndim >= 0 && ndim <= DFHIP_MAX_DIMS, "%s: number of dims %d exceeds %d", name, ndim,
               DFHIP_MAX_DIMS);
  int64_t total = 1;
  for (int i = 0; i < ndim; i++) {
    DF_CHECK_ARG(shape[i] >= 0, "%s: negative extent", name);
    total *= shape[i];
  }
  DF_CHECK_ARG(total == n_expected, "%s: view has %lld elements, array has %lld", name,
               (long long)total, (long long)n_expected);
  int64_t s[DFHIP_MAX_DIMS], t[DFHIP_MAX_DIMS];
  int m = 0;
  for (int i = 0; i < ndim; i++) {
    if (shape[i] == 1) continue;
    if (m > 0 && t[m - 1] == shape[i] * strides[i]) {  // outer dim steps exactly over inner
      s[m - 1] *= shape[i];
      t[m - 1] = strides[i];
    } else {
      s[m] = shape[i];
      t[m] = strides[i];
      m++;
    }
  }
  if (m == 0) {
    s[0] = 1;
    t[0] = 1;
    m = 1;
  }
  v->ndim = m;
  for (int i = 0; i < m; i++) {
    v->shape[i] = s[i];
    v->strides[i] = t[i];
  }
  v->offset = offset;
  return DFHIP_OK;
}

__device__ __forceinline__ int64_t view_index(const View& v, int64_t g) {
  int64_t idx = v.offset;
#pragma unroll
  for (int d = DFHIP_MAX_DIMS - 1; d >= 0; d--) {
    if (d < v.ndim) {
      int64_t q = g / v.shape[d];
      int64_t r = g - q * v.shape[d];
      idx += r * v.strides[d];
      g = q;
    }
  }
  return idx;
}

__device__ __forceinline__ int64_t view_index32(const View& v, uint32_t g) {
  int64_t idx = v.offset;
#pragma unroll
  for (int d = DFHIP_MAX_DIMS - 1; d >= 0; d--) {
    if (d < v.ndim) {
      uint32_t sd = (uint32_t)v.shape[d];
      uint32_t q = g / sd;
      uint32_t r = g - q * sd;
      idx += (int64_t)r * v.strides[d];
      g = q;
    }
  }
  return idx;
}

// mode 0: out[g] = a[idx(g)] (compact); 1: out[idx(g)] = a[g]; 2: out[idx(g)] = val
template <int MODE, bool SMALL>
__global__ __launch_bounds__(kThreads) void k_strided(const float* __restrict__ a,
                                                      float* __restrict__ out, int64_t n, View v,
                                                      float val) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = tid; g < n; g += step) {
    int64_t idx = SMALL ? view_index32(v, (uint32_t)g) : view_index(v, g);
    if (MODE == 0) out[g] = a[idx];
    else if (MODE == 1) out[idx] = a[g];
    else out[idx] = val;
  }
}

// 2-D transpose-like compact through an LDS tile: out[r][c] = a[off + r*sr + c*sc] where the
// source is contiguous along r (sr == 1): reads coalesce along r, writes along c.
__global__ __launch_bounds__(kThreads) void k_transpose2d(const float* __restrict__ a,
                                                          float* __restrict__ out, int64_t R,
                                                          int64_t Cn, int64_t sc, int64_t off) {
  __shared__ float tile[32][33];
  int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int64_t r = r0 + tx, c = c0 + j;
    if (r < R && c < Cn) tile[j][tx] = a[off + r + c * sc];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int64_t r = r0 + j, c = c0 + tx;
    if (r < R && c < Cn) out[r * Cn + c] = tile[tx][j];
  }
}

// ---- row reductions ------------------------------------------------------------------------
template <bool IS_MAX>
__device__ __forceinline__ float red_op(float a, float b) {
  if (IS_MAX) return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));  // np.max propagates NaN
  return a + b;
}
template <bool IS_MAX>
__device__ __forceinline__ float red_init() { return IS_MAX ? -INFINITY : 0.0f; }

template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = red_op<IS_MAX>(v, __shfl_down(v, o, 64));
  return v;
}

// rows of up to 16 elements: one thread per row
template <bool IS_MAX>
__global__ __launch_bounds__(kThreads) void k_reduce_thread(const float* __restrict__ a,
                                                            float* __restrict__ out, int64_t n_out,
                                                            int rs) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = tid; g < n_out; g += step) {
    const float* p = a + g * rs;
    float v = p[0];
    for (int i = 1; i < rs; i++) v = red_op<IS_MAX>(v, p[i]);
    out[g] = v;
  }
}

// one wave per row
template <bool IS_MAX>
__global__ __launch_bounds__(kThreads) void k_reduce_wave(const float* __restrict__ a,
                                                          float* __restrict__ out, int64_t n_out,
                                                          int64_t rs) {
  int lane = threadIdx.x & 63;
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t g = wave; g < n_out; g += nwaves) {
    const float* p = a + g * rs;
    float v = red_init<IS_MAX>();
    for (int64_t i = lane; i < rs; i += 64) v = red_op<IS_MAX>(v, p[i]);
    v = wave_reduce<IS_MAX>(v);
    if (lane == 0) out[g] = v;
  }
}

// one block per (row, split): partial of a slice of the row; splits == 1 writes the result
template <bool IS_MAX>
__global__ __launch_bounds__(kThreads) void k_reduce_block(const float* __restrict__ a,
                                                           float* __restrict__ out, int64_t rs,
                                                           int splits, int64_t chunk) {
  __shared__ float part[kThreads / 64];
  int64_t row = blockIdx.x / splits;
  int sp = blockIdx.x % splits;
  int64_t lo = sp * chunk, hi = lo + chunk < rs ? lo + chunk : rs;
  const float* p = a + row * rs;
  float v = red_init<IS_MAX>();
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) v = red_op<IS_MAX>(v, p[i]);
  v = wave_reduce<IS_MAX>(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = part[0];
    for (int w = 1; w < kThreads / 64; w++) r = red_op<IS_MAX>(r, part[w]);
    out[blockIdx.x] = r;
  }
}

template <bool IS_MAX>
int reduce_rows(const char* name, const float* a, float* out, int64_t n_out, int64_t rs) {
  DF_INIT();
  DF_CHECK_ARG(n_out >= 0 && rs >= 0, "%s: negative size", name);
  if (n_out == 0) return DFHIP_OK;
  DF_CHECK_ARG(a && out, "%s: null array", name);
  DF_CHECK_ARG(rs > 0, "%s: reduce_size must be > 0", name);
  ProfScope prof("reduce", 0.0, 4.0 * (double)n_out * (double)(rs + 1));
  if (rs <= 16) {
    DF_LAUNCH((k_reduce_thread<IS_MAX>), dim3(grid_for(n_out)), dim3(kThreads), 0, stream(),
                       a, out, n_out, (int)rs);
  } else if (rs <= 4096 || n_out >= 4096) {
    DF_LAUNCH((k_reduce_wave<IS_MAX>), dim3(grid_for(n_out * 64)), dim3(kThreads), 0,
                       stream(), a, out, n_out, rs);
  } else {
    int64_t want = df_ceil_div((int64_t)num_cus() * 4, n_out);
    int64_t maxs = df_ceil_div(rs, (int64_t)8192);
    int splits = (int)(want < maxs ? want : maxs);
    if (splits < 1) splits = 1;
    int64_t chunk = (df_ceil_div(rs, (int64_t)splits) + 3) & ~(int64_t)3;
    if (splits == 1) {
      DF_LAUNCH((k_reduce_block<IS_MAX>), dim3((unsigned)n_out), dim3(kThreads), 0, stream(),
                         a, out, rs, 1, chunk);
    } else {
      float* tmp = workspace((size_t)n_out * splits * sizeof(float));
      if (!tmp) return DFHIP_OUT_OF_MEMORY;
      DF_LAUNCH((k_reduce_block<IS_MAX>), dim3((unsigned)(n_out * splits)), dim3(kThreads), 0,
                         stream(), a, tmp, rs, splits, chunk);
      DF_LAUNCH_CHECK();
      DF_LAUNCH((k_reduce_wave<IS_MAX>), dim3(grid_for(n_out * 64)), dim3(kThreads), 0,
                         stream(), tmp, out, n_out, (int64_t)splits);
    }
  }
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

}  // namespace

extern "C" {

int dfhip_fill(float* out, int64_t n, float val) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0, "fill: negative size");
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(out, "fill: null array");
  ProfScope prof("fill", 0.0, 4.0 * (double)n);
  if (!aligned16(out)) {
    DF_CHECK_ARG(false, "fill: array must be 16-byte aligned");
  }
  DF_LAUNCH(k_fill, dim3(grid_for(n, 4)), dim3(kThreads), 0, stream(), out, n, val);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_compact(const float* a, float* out, int64_t n_out, int ndim, const int64_t* shape,
                  const int64_t* strides, int64_t offset) {
  DF_INIT_KEEP();          // gradients are compacted between two weight gradients of a backward pass
  DF_TOUCH(a);
  DF_TOUCH(out);
  View v;
  int s = make_view("compact", ndim, shape, strides, offset, n_out, &v);
  if (s != DFHIP_OK) return s;
  if (n_out == 0) return DFHIP_OK;
  DF_CHECK_ARG(a && out, "compact: null array");
  ProfScope prof("compact", 0.0, 8.0 * (double)n_out);
  if (v.ndim == 1 && v.strides[0] == 1) {
    DF_HIP(hipMemcpyAsync(out, a + offset, (size_t)n_out * 4, hipMemcpyDeviceToDevice, stream()));
    return DFHIP_OK;
  }
  if (v.ndim == 2 && v.strides[0] == 1 && v.strides[1] > 1 && v.shape[0] >= 32 && v.shape[1] >= 8) {
    dim3 grid((unsigned)df_ceil_div(v.shape[1], 32), (unsigned)df_ceil_div(v.shape[0], 32));
    if (grid.y <= 65535) {
      DF_LAUNCH(k_transpose2d, grid, dim3(kThreads), 0, stream(), a, out, v.shape[0],
                         v.shape[1], v.strides[1], offset);
      DF_LAUNCH_CHECK();
      return DFHIP_OK;
    }
  }
  if (n_out < (int64_t)1 << 31)
    DF_LAUNCH((k_strided<0, true>), dim3(grid_for(n_out)), dim3(kThreads), 0, stream(), a, out,
                       n_out, v, 0.0f);
  else
    DF_LAUNCH((k_strided<0, false>), dim3(grid_for(n_out)), dim3(kThreads), 0, stream(), a, out,
                       n_out, v, 0.0f);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_ewise_setitem(const float* a, float* out, int64_t n, int ndim, const int64_t* shape,
                        const int64_t* strides, int64_t offset) {
  DF_INIT();
  View v;
  int s = make_view("ewise_setitem", ndim, shape, strides, offset, n, &v);
  if (s != DFHIP_OK) return s;
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(a && out, "ewise_setitem: null array");
  ProfScope prof("setitem", 0.0, 8.0 * (double)n);
  if (n < (int64_t)1 << 31)
    DF_LAUNCH((k_strided<1, true>), dim3(grid_for(n)), dim3(kThreads), 0, stream(), a, out, n, v, 0.0f);
  else
    DF_LAUNCH((k_strided<1, false>), dim3(grid_for(n)), dim3(kThreads), 0, stream(), a, out, n, v, 0.0f);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_scalar_setitem(int64_t n, float val, float* out, int ndim, const int64_t* shape,
                         const int64_t* strides, int64_t offset) {
  DF_INIT();
  View v;
  int s = make_view("scalar_setitem", ndim, shape, strides, offset, n, &v);
  if (s != DFHIP_OK) return s;
  if (n == 0) return DFHIP_OK;
  DF_CHECK_ARG(out, "scalar_setitem: null array");
  ProfScope prof("setitem", 0.0, 4.0 * (double)n);
  if (n < (int64_t)1 << 31)
    DF_LAUNCH((k_strided<2, true>), dim3(grid_for(n)), dim3(kThreads), 0, stream(), nullptr, out, n, v, val);
  else
    DF_LAUNCH((k_strided<2, false>), dim3(grid_for(n)), dim3(kThreads), 0, stream(), nullptr, out, n, v, val);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}

int dfhip_ewise_add(const float* a, const float* b, float* o, int64_t n) { return launch_binary<OpAdd>("ewise_add", a, b, o, n, 0); }
int dfhip_ewise_mul(const float* a, const float* b, float* o, int64_t n) { return launch_binary<OpMul>("ewise_mul", a, b, o, n, 0); }
int dfhip_ewise_div(const float* a, const float* b, float* o, int64_t n) { return launch_binary<OpDiv>("ewise_div", a, b, o, n, 0); }
int dfhip_ewise_maximum(const float* a, const float* b, float* o, int64_t n) { return launch_binary<OpMax>("ewise_maximum", a, b, o, n, 0); }
int dfhip_ewise_eq(const float* a, const float* b, float* o, int64_t n) { return launch_binary<OpEq>("ewise_eq", a, b, o, n, 0); }
int dfhip_ewise_ge(const float* a, const float* b, float* o, int64_t n) { return launch_binary<OpGe>("ewise_ge", a, b, o, n, 0); }

int dfhip_scalar_add(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpAdd>("scalar_add", a, s, o, n); }
int dfhip_scalar_mul(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpMul>("scalar_mul", a, s, o, n); }
int dfhip_scalar_div(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpDiv>("scalar_div", a, s, o, n); }
int dfhip_scalar_power(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpPow>("scalar_power", a, s, o, n); }
int dfhip_scalar_maximum(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpMax>("scalar_maximum", a, s, o, n); }
int dfhip_scalar_eq(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpEq>("scalar_eq", a, s, o, n); }
int dfhip_scalar_ge(const float* a, float s, float* o, int64_t n) { return launch_scalar<OpGe>("scalar_ge", a, s, o, n); }

int dfhip_ewise_log(const float* a, float* o, int64_t n) { return launch_unary<OpLog>("ewise_log", a, o, n); }
int dfhip_ewise_exp(const float* a, float* o, int64_t n) { return launch_unary<OpExp>("ewise_exp", a, o, n); }
int dfhip_ewise_tanh(const float* a, float* o, int64_t n) { return launch_unary<OpTanh>("ewise_tanh", a, o, n); }

int dfhip_reduce_sum(const float* a, float* out, int64_t n_out, int64_t rs) {
  return reduce_rows<false>("reduce_sum", a, out, n_out, rs);
}
int dfhip_reduce_max(const float* a, float* out, int64_t n_out, int64_t rs) {
  return reduce_rows<true>("reduce_max", a, out, n_out, rs);
}

}  // extern "C"
