/*
 * dfhip.h — C ABI of libdfhip.so, the MI355X (gfx950) `hip` device for DeepFlows'
 * BackendTensor protocol.
 *
 * Every entry point below is what the reference's device-module binding for this
 * path would bind (the reference binds its CUDA device through pybind11 in
 * DeepFlows/backend/backend_src/ndarray_backend_cuda.cu:515-717; the Python call
 * sites are in DeepFlows/backend/backend_tensor.py).  The citation after each
 * declaration names the reference interface it replaces.
 *
 * Conventions
 *   - all buffers are device pointers to fp32 (the reference is fp32-only,
 *     ndarray_backend_cuda.cu:19 `scalar_t`); sizes are in ELEMENTS;
 *   - shapes / strides / offsets are int64 in elements, strides may be 0
 *     (broadcast) or negative (flip), at most DFHIP_MAX_DIMS dims
 *     (ndarray_backend_cuda.cu:107-124 `CudaVec`, MAX_VEC_SIZE 8);
 *   - every function returns a dfhip_status; on failure a thread-local message
 *     is available from dfhip_last_error().  The Python shim maps the status to
 *     the exception class pybind11 raised for the CUDA module
 *     (invalid_argument→ValueError, out_of_range→IndexError,
 *     runtime_error→RuntimeError, bad_alloc→MemoryError);
 *   - all work is enqueued on ONE non-blocking HIP stream owned by the library
 *     (the reference uses the default stream, ndarray_backend_cuda.cu:138);
 *     only dfhip_d2h and dfhip_sync block the host;
 *   - no PyTorch types, no C++ types: plain pointers and sizes.
 */
#ifndef DFHIP_H
#define DFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFHIP_MAX_DIMS 8

typedef enum {
  DFHIP_OK = 0,
  DFHIP_INVALID_ARGUMENT = 1, /* -> ValueError  (std::invalid_argument / domain_error) */
  DFHIP_OUT_OF_RANGE = 2,     /* -> IndexError  (std::out_of_range)                     */
  DFHIP_RUNTIME_ERROR = 3,    /* -> RuntimeError (hip launch / memcpy / alloc failure)  */
  DFHIP_OUT_OF_MEMORY = 4     /* -> MemoryError (std::bad_alloc)                        */
} dfhip_status;

/* ---- runtime ----------------------------------------------------------------------- */
int dfhip_init(int device_id);                 /* hipSetDevice + stream + pool; idempotent  */
int dfhip_device_count(int* out);
const char* dfhip_last_error(void);
int dfhip_sync(void);                          /* hipStreamSynchronize on the library stream */
void* dfhip_stream(void);                      /* the hipStream_t all kernels are launched on */
int dfhip_device_name(char* out, int cap);

/* Array(size): ndarray_backend_cuda.cu:48-83 (CudaArray: cudaMalloc / cudaFree deleter).
 * Here: a size-bucketed, stream-ordered caching pool (no hipMalloc in steady state).   */
int dfhip_malloc(int64_t n_floats, float** out);
int dfhip_free(float* p);
int dfhip_pool_stats(int64_t* bytes_reserved, int64_t* bytes_in_use, int64_t* n_hipmalloc);
int dfhip_pool_trim(void);

/* from_numpy / to_numpy: ndarray_backend_cuda.cu:667-716 (blocking cudaMemcpy).       */
int dfhip_h2d(const float* host, float* dst, int64_t n);
int dfhip_d2h(const float* src, float* host, int64_t n);
int dfhip_d2d(const float* src, float* dst, int64_t n);

/* ---- the 25 device primitives (backend_tensor.py:66-170 NumpyBackend;                */
/*      ndarray_backend_cuda.cu:127-509 kernels) ---------------------------------------- */
/* fill: .cu:127-144 */
int dfhip_fill(float* out, int64_t n, float val);
/* compact: .cu:147-176; out[g] = a[offset + sum_i idx_i(g)*strides_i], g row-major      */
int dfhip_compact(const float* a, float* out, int64_t n_out, int ndim,
                  const int64_t* shape, const int64_t* strides, int64_t offset);
/* ewise_setitem: .cu:178-198; out[offset + sum idx_i(g)*strides_i] = a[g] (scatter)     */
int dfhip_ewise_setitem(const float* a, float* out, int64_t n, int ndim,
                        const int64_t* shape, const int64_t* strides, int64_t offset);
/* scalar_setitem: .cu:200-221 */
int dfhip_scalar_setitem(int64_t n, float val, float* out, int ndim,
                         const int64_t* shape, const int64_t* strides, int64_t offset);
/* ewise binary: .cu:224-229 add, 259-262 mul, 285-288 div, 325-328 maximum,
 * 351-354 eq, 377-380 ge (eq/ge produce 1.0f / 0.0f)                                   */
int dfhip_ewise_add(const float* a, const float* b, float* out, int64_t n);
int dfhip_ewise_mul(const float* a, const float* b, float* out, int64_t n);
int dfhip_ewise_div(const float* a, const float* b, float* out, int64_t n);
int dfhip_ewise_maximum(const float* a, const float* b, float* out, int64_t n);
int dfhip_ewise_eq(const float* a, const float* b, float* out, int64_t n);
int dfhip_ewise_ge(const float* a, const float* b, float* out, int64_t n);
/* scalar binary: .cu:246-249 add, 272-275 mul, 298-307 div, 312-315 power,
 * 338-341 maximum, 364-367 eq, 390-393 ge.  Values follow the NumPy device
 * (backend_tensor.py:109-143): scalar_div by 0 gives inf/nan, it does not throw.       */
int dfhip_scalar_add(const float* a, float s, float* out, int64_t n);
int dfhip_scalar_mul(const float* a, float s, float* out, int64_t n);
int dfhip_scalar_div(const float* a, float s, float* out, int64_t n);
int dfhip_scalar_power(const float* a, float s, float* out, int64_t n);
int dfhip_scalar_maximum(const float* a, float s, float* out, int64_t n);
int dfhip_scalar_eq(const float* a, float s, float* out, int64_t n);
int dfhip_scalar_ge(const float* a, float s, float* out, int64_t n);
/* unary: .cu:403-406 log, 416-419 exp, 429-432 tanh.  log follows NumPy
 * (backend_tensor.py:145-146): log(0) = -inf, log(<0) = nan.                           */
int dfhip_ewise_log(const float* a, float* out, int64_t n);
int dfhip_ewise_exp(const float* a, float* out, int64_t n);
int dfhip_ewise_tanh(const float* a, float* out, int64_t n);
/* matmul: .cu:443-466; out[M x P] = a[M x N] . b[N x P], row-major, fp32 MFMA          */
int dfhip_matmul(const float* a, const float* b, float* out, int64_t m, int64_t n, int64_t p);
/* reduce_sum / reduce_max: .cu:469-509; out[g] = reduce of reduce_size contiguous inputs */
int dfhip_reduce_sum(const float* a, float* out, int64_t n_out, int64_t reduce_size);
int dfhip_reduce_max(const float* a, float* out, int64_t n_out, int64_t reduce_size);

/* ---- fused hot-path entry points ------------------------------------------------------
 * The reference composes these from the primitives above in Python; a
 * front-end uses them when `hasattr(device.mod, ...)`.                                   */

/* GEMM with transposed operands, replaces compact()+matmul in tensor.py:699-716
 * (matmul.grad_fn).  out[M x P] = op(a) . op(b); ta: a is stored [N x M]; tb: b is
 * stored [P x N].  If `bias` is not NULL (length P) it is added to every row
 * (F.linear, nn/functional.py:8-12).                                                   */
int dfhip_gemm(const float* a, const float* b, const float* bias, float* out,
               int64_t m, int64_t n, int64_t p, int ta, int tb);

/* Backward of y = x.W + b (linear.py:28-59; matmul.grad_fn tensor.py:699-716 and the bias un-broadcast
 * tensor.py:462-483): dx (m,k) = dy.W^T, dw (k,p) = x^T.dy, db (p) = column sums of dy; any output may
 * be NULL.  One launch when the layer is small (a classifier head), the three contractions otherwise. */
int dfhip_linear_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                          int64_t m, int64_t k, int64_t p);
/* F.relu(F.linear(x, W, b)) (nn/functional.py:8-16) as ONE launch for layers up to 1.5e8 flops: z (m,p) = x.W + b is kept
 * for the backward pass, r = max(z, 0).  Backward of that pair: the upstream gradient dr counts only where r == z
 * (tensor.py:872-877: the gradient passes where output == input, 0 included); the mask is applied on the operand loads
 * of the three contractions (dx = (dr*mask).W^T, dw = x^T.(dr*mask), db = column sums), one launch.  *done = 0 (nothing
 * launched) when the layer is too large for the one-launch kernels: the caller then runs gemm + relu / relu_backward +
 * linear_backward. */
int dfhip_linear_relu_forward(const float* x, const float* w, const float* bias, float* z, float* r, int64_t m,
                              int64_t k, int64_t p, int* done);
int dfhip_linear_relu_backward(const float* x, const float* w, const float* z, const float* r, const float* dr,
                               float* dx, float* dw, float* db, int64_t m, int64_t k, int64_t p, int* done);

/* Conv2d forward: nn/functional.py:316-344 (pad -> im2col -> GEMM) + conv.py:104-108
 * (bias (1,O,1,1), may be NULL).  x (N,C,H,W), w (O,C,k,k), y (N,O,oh,ow), all NCHW.  */
int dfhip_conv2d_forward(const float* x, const float* w, const float* bias, float* y,
                         int N, int C, int H, int W, int O, int k, int stride, int pad);
/* dx = dfhip_conv2d_dgrad(dy, w) + add (add has dx's shape): the autograd accumulation
 * "parent.grad + contrib" (tensor.py:444) of a tensor with two consumers, e.g. the input of a
 * residual block, done in the epilogue of the input-gradient kernel instead of a separate pass. */
int dfhip_conv2d_dgrad_add(const float* dy, const float* w, const float* add, float* dx, int N, int C,
                           int H, int W, int O, int k, int stride, int pad, int mode);

/* y = conv2d(x, w; 3x3 / stride 2 / pad 1) and yd = conv2d(x, wd; 1x1 / stride 2 / pad 0): conv1 and downsample[0] of
 * a residual block read the same tensor (test/ResNet_CIFAR10_cuda.py:31-47; nn/functional.py:316-344 twice) -- one
 * launch, the shortcut as a tenth tap of the 3x3 loop (it reads that convolution's centre-tap pixels).  x (N,C,H,W);
 * w (O,C,3,3); wd (O,C,1,1); y, yd (N,O,H/2,W/2).  partials / partials_d (both or neither): BatchNorm partials of y /
 * yd as dfhip_conv2d_forward_stats, *rows_out rows each.  *done = 0 (nothing launched) when the shape has no such
 * kernel: the caller then runs dfhip_conv2d_forward_stats twice. */
int dfhip_conv2d_forward_s2_pair(const float* x, const float* w, const float* wd, float* y, float* yd, int N, int C,
                                 int H, int W, int O, float* partials, float* partials_d, int cap_rows, int* rows_out,
                                 int* done);

/* dx = dgrad(3x3 / stride 2 / pad 1; dy, w) + dgrad(1x1 / stride 2 / pad 0; dy2, w2) (+ add, may be NULL): the two
 * gradients the input of a residual block receives when its shortcut is a convolution (test/ResNet_CIFAR10_cuda.py:
 * 58-63: conv1 and downsample[0] both read x; tensor.py:444 sums them) as one launch.  mode as dfhip_conv2d_dgrad:
 * 0 = each with the reference's last-writer-wins rule (functional.py:285-294), 1 = the complete sums (a pixel of
 * parity class (ey, ex) receives through (1 + ey)(1 + ex) taps; only those products are formed).  dy, dy2 (N, O, H/2,
 * W/2); w (O, C, 3, 3); w2 (O, C, 1, 1); dx (N, C, H, W).  *done = 0 (nothing launched) when the shape has no such
 * kernel: the caller then runs dfhip_conv2d_dgrad followed by dfhip_conv2d_dgrad_add. */
int dfhip_conv2d_dgrad_pair(const float* dy, const float* w, const float* dy2, const float* w2, const float* add,
                            float* dx, int N, int C, int H, int W, int O, int mode, int* done);

/* weight gradient: tensor.py:716 (A^T . dC) for the conv GEMM; dw (O,C,k,k)            */
int dfhip_conv2d_wgrad(const float* x, const float* dy, float* dw,
                       int N, int C, int H, int W, int O, int k, int stride, int pad);

/* conv2d of the training-mode BatchNorm2d of x, with NO activation in between -- `out = bn1(conv1(x)); out =
 * conv2(out)` of the reference's residual block (test/ResNet_CIFAR10_cuda.py:31-35; nn/modules/batchnorm.py:30-55
 * followed by nn/functional.py:316-344): x is conv1's raw output, partials_in (rows_in x C x 3) the BatchNorm
 * partials dfhip_conv2d_forward_stats left behind for it.  Every workgroup of the convolution combines the
 * partials into a per-channel (scale, shift) table and applies it to its operand on load: the normalised tensor is
 * never written or read.  By-products as dfhip_bn2d_forward_partials: save_mean / save_var (C; batch mean, biased
 * variance), running_mean / running_var (C, updated in place, may be NULL), plus affine_out (C x 2: scale, shift).
 * *done = 1 when a kernel with this operand path took the shape; 0 (nothing launched or written) otherwise -- the
 * caller then runs dfhip_bn2d_forward_partials and dfhip_conv2d_forward_stats.  partials / cap_rows / rows_out:
 * BatchNorm partials of y, as in dfhip_conv2d_forward_stats (partials may be NULL). */
int dfhip_conv2d_forward_bn_input(const float* x, const float* partials_in, int rows_in, const float* gamma,
                                  const float* beta, float eps, float momentum, float* save_mean, float* save_var,
                                  float* running_mean, float* running_var, float* affine_out, const float* w,
                                  const float* bias, float* y, int N, int C, int H, int W, int O, int k, int stride,
                                  int pad, float* partials, int cap_rows, int* rows_out, int* done);
/* weight gradient of that convolution: dw = A^T . dC (tensor.py:716) with A built from x * scale + shift per source
 * channel (affine: C x 2, the table dfhip_conv2d_forward_bn_input wrote).  Queued like dfhip_conv2d_wgrad inside
 * a group; x, affine and dy must then stay allocated until the group has run. */
int dfhip_conv2d_wgrad_affine(const float* x, const float* affine, const float* dy, float* dw,
                              int N, int C, int H, int W, int O, int k, int stride, int pad);

/* Opt-in reduced-precision contraction (OUTSIDE the 1e-4 parity contract, never the default): dtype 1 =
 * the operands of two kernel families' matrix products are rounded to bf16 (nearest even) while they are packed, accumulation
 * stays fp32: the grouped convolution weight gradients (dfhip_wgrad_group_begin / _end; v_mfma_f32_32x32x8_bf16, rows of >= 4
 * output pixels) and the register-operand 3x3 / stride 1 forward convolutions of 16- and 8-wide planes with BatchNorm
 * statistics or a BatchNorm operand (dfhip_conv2d_forward_stats, dfhip_conv2d_forward_bn_input; v_mfma_f32_16x16x16_bf16).
 * Everything else -- the other forward kernels, input gradients, BatchNorm, the optimizer -- computes in fp32 as before.
 * dtype 0 = fp32 (the reference's arithmetic, backend_tensor.py:154-158). */
int dfhip_set_conv_matmul_dtype(int dtype);
int dfhip_conv_matmul_dtype(void);

/* Grouped weight gradients (extension; the reference computes A^T . dC inside each node's grad_fn,
 * tensor.py:716, but nothing reads a weight's .grad before backward() returns, tensor.py:421-499).
 * Between _begin and _end, dfhip_conv2d_wgrad calls are queued where the fast kernel takes the
 * shape, and _end -- or any call that touches a queued dw -- runs all of them as ONE launch plus
 * one slab-sum launch.  The x and dy arrays of a queued call must stay allocated and unmodified
 * until then.                                                                                   */
int dfhip_wgrad_group_begin(void);
int dfhip_wgrad_group_end(void);

/* conv2d_forward that also leaves BatchNorm partials of y behind: partials[rows][O][3] =
 * (count, mean, centred second moment) of every output channel over the pixels of one workgroup
 * (rows <= cap_rows).  *rows_out = 0 when the shape was served by a kernel without this epilogue;
 * the caller then falls back to dfhip_bn2d_forward_train.  Fuses the first pass of
 * nn/modules/batchnorm.py:36-42 (batch mean / variance) into nn/functional.py:conv2d. */
int dfhip_conv2d_forward_stats(const float* x, const float* w, const float* bias, float* y, int N, int C,
                               int H, int W, int O, int k, int stride, int pad, float* partials,
                               int cap_rows, int* rows_out);

/* input gradient: tensor.py:708 (dC . B^T) followed by __im2col2d.grad_fn
 * (nn/functional.py:285-294) and __pad2d.grad_fn (:307-313).
 * mode 0 = "overwrite": reference-exact last-writer-wins col2im (functional.py:292-293);
 * mode 1 = "accumulate": the mathematically correct sum over overlapping windows.      */
int dfhip_conv2d_dgrad(const float* dy, const float* w, float* dx,
                       int N, int C, int H, int W, int O, int k, int stride, int pad, int mode);

/* ---- BatchNorm2d backward without a launch of its own --------------------------------------------------------
 * Autograd of nn/modules/batchnorm.py:30-55 (training mode, biased variance): for y = gamma (z - mean) r + beta,
 * r = 1 / sqrt(var + eps), m = N*H*W elements per channel,
 *     dz = A dy + B (z - mean) + Cc,  A = gamma r,  B = -A r k2,  Cc = -A k1,
 *     k1 = sum(dy) / m,  k2 = r sum(dy (z - mean)) / m,   dgamma = r sum(dy (z - mean)),  dbeta = sum(dy).
 * The two sums need all of dy.  Instead of a statistics pass and an apply pass behind a launch boundary,
 *   - the kernel that PRODUCES dy (the input gradient of the next convolution) leaves per-channel partial rows
 *     (sum dy, sum dy (z - mean)) behind (dfhip_bn_stats_out), one row per workgroup, at most DFHIP_BN_ROWS_MAX;
 *   - the kernels that CONSUME dz (input gradient and weight gradient of the convolution whose output z is) add the
 *     rows up and form dz on their operand loads (dfhip_bn_bwd_in); dz itself is never written.
 * All arrays are device arrays; the structs are read by the host side of the call only. */
#define DFHIP_BN_ROWS_MAX 512
typedef struct dfhip_bn_bwd_in {
  const float* z;        /* the BatchNorm layer's input, same shape as dy                                         */
  const float* rows;     /* nrows x O x K partial sums left behind by the producer of dy                          */
  int nrows, K, kidx;    /* (row[0], row[kidx]) = (sum dy, sum dy (z - mean)) over the row's share of the tensor  */
  const float* mean;     /* O: batch mean / biased variance saved by the forward pass                             */
  const float* var;
  const float* gamma;    /* O, or NULL (1)                                                                        */
  float eps;
  float inv_m;           /* 1 / (N * H * W of dy)                                                                 */
  float* dgamma;         /* O each, may be NULL: written by the call that adds up the rows (dfhip_conv2d_dgrad_bn, */
  float* dbeta;          /* dfhip_conv2d_dgrad_pair_bn, dfhip_bn_bwd_table)                                       */
  float* table;          /* O x 4: (A, B, mean, Cc) per channel, written by the same call, read by               */
                         /* dfhip_conv2d_wgrad_bn and dfhip_bn2d_backward_apply_table                              */
} dfhip_bn_bwd_in;
typedef struct dfhip_bn_stats_out {
  int n;                 /* 1 or 2 BatchNorm layers whose OUTPUT gradient the produced tensor is (2: bn2(z2) + bn_d(z_d), */
  const float* z[2];     /* the tail of a residual block whose shortcut is convolution + BatchNorm); their inputs ...     */
  const float* mean[2];  /* ... and batch means                                                                           */
  float* rows;           /* cap_rows x C x (1 + n): (sum dx, sum dx (z0 - mean0)[, sum dx (z1 - mean1)])                  */
  int cap_rows;          /* >= the rows the call writes (*stats_rows of dfhip_conv2d_bn_bwd_supported)                    */
  int nrows;             /* out: rows written                                                                             */
} dfhip_bn_stats_out;
/* *dgrad_ok / *wgrad_ok = 1 when dfhip_conv2d_dgrad_bn (pair = 0; dfhip_conv2d_dgrad_pair_bn: pair = 1) / a grouped
 * dfhip_conv2d_wgrad_bn has a kernel with the BatchNorm descriptors for this convolution (overwrite mode only);
 * *stats_rows = the rows that input-gradient call writes into a dfhip_bn_stats_out. */
int dfhip_conv2d_bn_bwd_supported(int N, int C, int H, int W, int O, int k, int stride, int pad, int pair,
                                  int* dgrad_ok, int* wgrad_ok, int* stats_rows);
/* dfhip_conv2d_dgrad / dfhip_conv2d_dgrad_add (add may be NULL) in mode 0 (tensor.py:708 + functional.py:285-294) with
 * `in` (may be NULL): dy is the gradient of a BatchNorm OUTPUT, its backward is applied on the operand loads; and
 * `out` (may be NULL): dx is the output gradient of out->n BatchNorm layers, their rows are left behind.
 * DFHIP_INVALID_ARGUMENT when the shape has no such kernel (dfhip_conv2d_bn_bwd_supported says so beforehand). */
int dfhip_conv2d_dgrad_bn(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H, int W,
                          int O, int k, int stride, int pad, const dfhip_bn_bwd_in* in, dfhip_bn_stats_out* out);
/* dfhip_conv2d_dgrad_pair in mode 0 with the descriptors of dy (in), of the shortcut's dy2 (in2) and of dx (out);
 * each may be NULL. */
int dfhip_conv2d_dgrad_pair_bn(const float* dy, const float* w, const float* dy2, const float* w2, const float* add,
                               float* dx, int N, int C, int H, int W, int O, const dfhip_bn_bwd_in* in,
                               const dfhip_bn_bwd_in* in2, dfhip_bn_stats_out* out);
/* dfhip_conv2d_wgrad (affine == NULL) / dfhip_conv2d_wgrad_affine whose dy is the gradient of a BatchNorm output:
 * z and table (O x 4) as in dfhip_bn_bwd_in, the table already written.  Only inside a group (dfhip_wgrad_group_begin);
 * *done = 0 (nothing queued) otherwise or when the grouped kernel does not take the shape: the caller then writes dz
 * out (dfhip_bn2d_backward_apply_table) and calls dfhip_conv2d_wgrad. */
int dfhip_conv2d_wgrad_bn(const float* x, const float* affine, const float* dy, const float* z, const float* table,
                          float* dw, int N, int C, int H, int W, int O, int k, int stride, int pad, int* done);
/* the table, dgamma and dbeta of `in` as a launch of its own (for a convolution that needs no input gradient) */
int dfhip_bn_bwd_table(const dfhip_bn_bwd_in* in, int O);
/* dz (N, O, HW) = A dy + B (z - mean) + Cc from the table: the pending backward written out */
int dfhip_bn2d_backward_apply_table(const float* dy, const float* z, const float* table, float* dz, int N, int O,
                                    int HW);

/* BatchNorm2d training forward from the partials of dfhip_conv2d_forward_stats (x is that
 * convolution's output): statistics by Chan's combination of the partial rows, then
 * y = (x - mean) / sqrt(var + eps) * gamma + beta (+ residual, the shortcut of a residual block:
 * test/ResNet_CIFAR10_cuda.py BasicBlock.forward "out + shortcut(x)").  residual may be NULL. */
int dfhip_bn2d_forward_partials(const float* x, const float* partials, int rows, const float* gamma,
                                const float* beta, const float* residual, float* y, float* save_mean,
                                float* save_var, float* running_mean, float* running_var, int N, int C,
                                int HW, float eps, float momentum);

/* BatchNorm2d training forward: nn/modules/batchnorm.py:30-55.  Biased variance; running
 * stats r = r*(1-momentum) + batch*momentum (both may be NULL).  save_mean / save_var
 * (C floats each) feed the backward.                                                    */
int dfhip_bn2d_forward_train(const float* x, const float* gamma, const float* beta, float* y,
                             float* save_mean, float* save_var, float* running_mean,
                             float* running_var, int N, int C, int HW, float eps, float momentum);
int dfhip_bn2d_forward_eval(const float* x, const float* gamma, const float* beta, float* y,
                            const float* running_mean, const float* running_var,
                            int N, int C, int HW, float eps);
/* backward of that expression graph (autograd through batchnorm.py:32-55): dx, dgamma,
 * dbeta (any of dx/dgamma/dbeta may be NULL).                                           */
int dfhip_bn2d_backward(const float* dy, const float* x, const float* gamma,
                        const float* save_mean, const float* save_var, float* dx,
                        float* dgamma, float* dbeta, int N, int C, int HW, float eps);

/* Two BatchNorm2d layers that meet in one sum -- y = bnA(xA) + bnB(xB), the tail of a residual block whose
 * shortcut is convolution + BatchNorm (test/ResNet_CIFAR10_cuda.py:38-47): batchnorm.py:30-55 twice and the
 * "out + identity" in one launch (bnB's output is never written), and the two backward passes, which
 * receive the same dy, in one launch.  Statistics from the convolutions' partials as in
 * dfhip_bn2d_forward_partials.                                                                      */
int dfhip_bn2d_forward_partials_pair(const float* xa, const float* partials_a, int rows_a, const float* gamma_a,
                                     const float* beta_a, float* save_mean_a, float* save_var_a,
                                     float* running_mean_a, float* running_var_a, float eps_a, float momentum_a,
                                     const float* xb, const float* partials_b, int rows_b, const float* gamma_b,
                                     const float* beta_b, float* save_mean_b, float* save_var_b,
                                     float* running_mean_b, float* running_var_b, float eps_b, float momentum_b,
                                     float* y, int N, int C, int HW);
int dfhip_bn2d_backward_pair(const float* dy, const float* xa, const float* gamma_a, const float* save_mean_a,
                             const float* save_var_a, float* dxa, float* dgamma_a, float* dbeta_a, float eps_a,
                             const float* xb, const float* gamma_b, const float* save_mean_b,
                             const float* save_var_b, float* dxb, float* dgamma_b, float* dbeta_b, float eps_b,
                             int N, int C, int HW);

/* BatchNorm2d (training) -> ReLU -> MaxPool2d(2) in one forward launch after the statistics pass,
 * and its backward (statistics + apply), for the stem of the CIFAR models
 * (test/ResNet_CIFAR10_cuda.py: pool1(relu(bn1(conv1(x))))).  x (N,C,H,W) is the convolution
 * output, y (N,C,H/2,W/2) the pooled tensor, g its gradient; dx the gradient w.r.t. x.  Same
 * values as the three separate operators: batchnorm.py:30-55, functional.py:15-16 / tensor.py:872-877
 * (gradient where output == input), functional.py:347-374 / tensor.py:779-791 (every tied
 * maximum gets the gradient).  H even, W a multiple of 8. */
int dfhip_bn_relu_maxpool2_forward(const float* x, const float* gamma, const float* beta, float* y,
                                   float* save_mean, float* save_var, float* running_mean,
                                   float* running_var, int N, int C, int H, int W, float eps, float momentum);
/* Backward of the whole stem pool(relu(bn(conv(x)))) (test/ResNet_CIFAR10_cuda.py:84-87) when x needs no gradient:
 * the statistics pass (dgamma, dbeta) and then the convolution's weight gradient (tensor.py:716) with
 * d loss / d (convolution output) rebuilt from the convolution output z inside the kernel that consumes it -- the
 * full-resolution gradient tensor is never written.  z (N, O, oh, ow); y, g (N, O, oh/2, ow/2) the pooled output
 * and its gradient; dw (O, C, k, k); dgamma, dbeta (O).  *done = 0 (nothing launched) when the shape has no such
 * kernel: the caller then runs dfhip_bn_relu_maxpool2_backward and dfhip_conv2d_wgrad. */
int dfhip_stem_backward(const float* x, const float* z, const float* y, const float* g, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_var, float* dw, float* dgamma,
                        float* dbeta, int N, int C, int H, int W, int O, int k, int stride, int pad, float eps,
                        int* done);
/* The same stem for the shapes of the CIFAR models (3x3 / pad 1 or 5x5 / pad 2, stride 1, 1-3 input channels, <= 32
 * output channels, 32-wide images of even height <= 32; bias (O) and its gradient dbias optional) WITHOUT the full-resolution convolution output: forward writes only
 * y (N, O, H/2, 16) and the statistics -- BatchNorm and ReLU are monotone per channel, so the pooled value is
 * relu(bn(extreme of the window)) bit for bit and the convolution keeps only that extreme; backward rebuilds the
 * convolution output from x with the same MFMA sequence (the same bits) to apply the reference's tie rules
 * (tensor.py:779-791, 872-877) and forms dw (O, C, 3, 3), dgamma, dbeta (O) as dfhip_stem_backward does.  Replaces
 * dfhip_conv2d_forward_stats + dfhip_bn_relu_maxpool2_forward_partials / dfhip_stem_backward for that shape
 * (nn/functional.py:316-374 + nn/modules/batchnorm.py:30-55 in the reference).  *done = 0 (nothing launched) for any
 * other shape.  Inside a weight-gradient group (dfhip_wgrad_group_begin / _end) with gradients queued, the 3x3 backward
 * leaves its last step -- totals over the workgroups' slabs -> dw, dbias, dgamma, dbeta -- to the group's slab-sum launch:
 * like a queued dfhip_conv2d_wgrad, those four arrays hold their values after the group has been flushed (by _end, by any
 * entry point that flushes deferred work, or by one that is handed one of them). */
int dfhip_stem_forward_noz(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                           float* y, float* save_mean, float* save_var, float* running_mean, float* running_var, int N,
                           int C, int H, int W, int O, int k, int stride, int pad, float eps, float momentum, int* done);
int dfhip_stem_backward_noz(const float* x, const float* w, const float* bias, const float* y, const float* g,
                            const float* gamma, const float* beta, const float* save_mean, const float* save_var,
                            float* dw, float* dbias, float* dgamma, float* dbeta, int N, int C, int H, int W, int O, int k,
                            int stride, int pad, float eps, int* done);
/* the same with the batch statistics from the partials the convolution that produced x left behind
 * (dfhip_conv2d_forward_stats: partials[rows][C][3]): no statistics pass over x */
int dfhip_bn_relu_maxpool2_forward_partials(const float* x, const float* partials, int rows, const float* gamma,
                                            const float* beta, float* y, float* save_mean, float* save_var,
                                            float* running_mean, float* running_var, int N, int C, int H, int W,
                                            float eps, float momentum);
int dfhip_bn_relu_maxpool2_backward(const float* x, const float* y, const float* g, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_var,
                                    float* dx, float* dgamma, float* dbeta, int N, int C, int H, int W,
                                    float eps);

/* Input pipeline on the device.
 * dfhip_upload_async / dfhip_upload_wait: double-buffered pinned upload on a copy stream, so the
 * H2D of batch k+1 overlaps the training step of batch k (utils/data/dataloader.py:60-139 hands
 * over pageable NumPy arrays; Tensor(..., device='cuda') copies them synchronously).
 * dfhip_augment_batch: test/ResNet_CIFAR10_cuda.py:129-148 augment_batch — reflect pad, random crop,
 * horizontal flip, optional erase rectangle, clip to [-1, 1]; decisions[n] = {ys, xs, flip, erase_y,
 * erase_x, erase_h (0 = none), erase_w, 0} as int32 in device memory (drawn on the host).
 * dfhip_onehot_smooth: :180-184, onehot * (1 - eps) + eps / K from int32 labels. */
int dfhip_upload_async(const void* host, float* dst, int64_t bytes, int slot);
int dfhip_upload_wait(int slot);
/* dfhip_upload_release: every kernel that reads the destination of slot's last upload has been enqueued -- the next
 * dfhip_upload_async through this slot waits for THIS point of the library stream, not for what is enqueued later
 * (the training step of that batch runs beside the next copy). */
int dfhip_upload_release(int slot);
int dfhip_augment_batch(const float* in, float* out, const int* decisions, int N, int C, int H, int W, int pad);
int dfhip_onehot_smooth(const int* labels, float* out, int N, int K, double eps);

/* MaxPool2d with kernel == stride, no padding: nn/functional.py:347-374; backward is
 * tensor.max.grad_fn (tensor.py:779-791): every tied maximum receives the gradient.     */
int dfhip_maxpool2d_forward(const float* x, float* y, int NC, int H, int W, int k);
int dfhip_maxpool2d_backward(const float* x, const float* y, const float* dy, float* dx,
                             int NC, int H, int W, int k);

/* ReLU = tensor.maximum(x, 0) (nn/functional.py:15-16); backward (y == x) * g
 * (tensor.py:872-877): x == 0 passes the gradient.                                      */
int dfhip_relu_forward(const float* x, float* y, int64_t n);
int dfhip_relu_backward(const float* x, const float* y, const float* g, float* dx, int64_t n);

/* cross_entropy(reduction='mean') on dense targets: nn/functional.py:104-115.
 * loss is one float; logits/target are (rows, cols) row-major.                          */
int dfhip_softmax_ce_forward(const float* logits, const float* target, float* loss,
                             int rows, int cols);
/* dlogits = gscale * (softmax * sum(target) - target) / rows, gscale read from the
 * device scalar `gloss` (the upstream gradient of the (1,)-shaped loss).                */
int dfhip_softmax_ce_backward(const float* logits, const float* target, const float* gloss,
                              float* dlogits, int rows, int cols);

/* dfhip_softmax_ce_forward that also writes dlogits for an upstream gradient of exactly 1 (what loss.backward()
 * seeds, tensor.py:428), so the backward pass of a training step needs no launch for this node.  *done = 0 when the
 * shape is outside the one-workgroup kernel (rows <= 4096, cols <= 32): only the loss was computed. */
int dfhip_softmax_ce_forward_grad(const float* logits, const float* target, float* loss, float* dlogits,
                                  int rows, int cols, int* done);

/* Classifier head of test/ResNet_CIFAR10_cuda.py:96-99 as one launch each way: y (N, P) = (scale * sum over the HW
 * pixels of x (N, C, HW)) . w (C, P) + b (P or NULL), P <= 16.  `scale` carries the divide-by-total-count rule of
 * the two tensor.mean(axis=2) calls (backend_tensor.py:659-662); pooled (N, C) is kept for the backward pass:
 * dx (N, C, HW) = gscale * (dy . w^T) broadcast over the pixels, dw (C, P) = pooled^T . dy (tensor.py:716),
 * db (P) = column sums of dy; dx / dw / db may be NULL. */
int dfhip_pool_linear_forward(const float* x, const float* w, const float* b, float* pooled, float* y,
                              int N, int C, int HW, int P, float scale);
int dfhip_pool_linear_backward(const float* dy, const float* pooled, const float* w, float* dx, float* dw,
                               float* db, int N, int C, int HW, int P, float gscale);

/* The classifier head AND its loss from one launch (training step): logits (N, P) as dfhip_pool_linear_forward,
 * loss (1) = cross-entropy(logits, target) as dfhip_softmax_ce_forward (nn/functional.py:104-115: stable log-softmax x dense
 * target, sum / N), dlogits (N, P) as dfhip_softmax_ce_forward_grad (upstream gradient exactly 1, tensor.py:428) and
 * dx (N, C, HW or NULL) = gscale * (dlogits . w^T) broadcast over the pixels as dfhip_pool_linear_backward -- everything of
 * test/ResNet_CIFAR10_cuda.py:96-99 + :186 and of its backward pass that belongs to ONE image; pooled (N, C) is kept.
 * The loss is a sum over the workgroups: the launch leaves the N row losses in `rowloss`, and loss[0] is NOT valid until
 * dfhip_head_grads (the backward pass: no launch inside a weight-gradient group) or dfhip_loss_finish (one small launch: a
 * reader that comes first) has added them up in dfhip_softmax_ce_forward's order -- an arrival counter inside the launch
 * costs every workgroup two round trips to the memory side (measured: 5.5 -> 10.3 us).  *done = 0 and nothing launched
 * when the shape is outside the kernel (P <= 16, C <= 512, N <= 4096): the caller runs the three entry points above. */
int dfhip_head_ce_forward(const float* x, const float* w, const float* b, const float* target, float* pooled,
                          float* logits, float* loss, float* dlogits, float* dx, float* rowloss, int N, int C, int HW,
                          int P, float scale, float gscale, int* done);

/* The same launch for a classifier behind a Dropout layer -- criterion(fc(drop(x)), target) with x (N, C) two-dimensional
 * (test/CNN_CIFAR10_cuda.py:104-107 + its CrossEntropyLoss; BASELINE.json config 3): xd (N, C) = x * mask / keep
 * (nn/modules/dropout.py:31, dfhip_dropout_apply's two roundings) is what the Linear reads and is kept for dfhip_head_grads
 * (as `pooled`); dx (N, C or NULL) is the gradient at Dropout's INPUT, (dlogits . w^T) / keep * mask.  logits, loss, dlogits,
 * rowloss as dfhip_head_ce_forward.  *done = 0 and nothing launched outside P <= 16, C <= 2048, N <= 4096. */
int dfhip_droplin_ce_forward(const float* x, const float* mask, float keep, const float* w, const float* b, const float* target,
                             float* xd, float* logits, float* loss, float* dlogits, float* dx, float* rowloss, int N, int C,
                             int P, int* done);

/* dw (C, P) = pooled^T . dlogits (tensor.py:716), db (P) = column sums of dlogits (tensor.py:462-483); either may be NULL.
 * rowloss / loss (or NULL, NULL): also the mean of the row losses (see dfhip_head_ce_forward).
 * Between dfhip_wgrad_group_begin and _end the sums are extra workgroups of the group's slab-sum launch (no launch of their
 * own; the input arrays must stay alive until dfhip_wgrad_group_end returns), otherwise one launch. */
int dfhip_head_grads(const float* dlogits, const float* pooled, float* dw, float* db, const float* rowloss, float* loss,
                     int N, int C, int P);

/* loss[0] = (sum of the N row losses of dfhip_head_ce_forward) / N, dfhip_softmax_ce_forward's order of additions. */
int dfhip_loss_finish(const float* rowloss, float* loss, int N);

/* A two-layer perceptron's training step, one launch each way (test/MLP_MNIST_cuda.py:37-59: fc1 -> F.relu -> fc2, then
 * CrossEntropyLoss; BASELINE.json config 2).  Forward: zpre (M, HD or NULL) = x (M, K) . w1 (K, HD) + b1 (HD or NULL), h = relu(zpre)
 * (nn/functional.py:8-16), logits (M, P) = h . w2 (HD, P) + b2 (P or NULL), the row losses of cross_entropy(logits, target)
 * (nn/functional.py:104-115) in rowloss (M), dlogits (M, P) for an upstream gradient of exactly 1 (tensor.py:428) and
 * dz1 (M, HD) = d loss / d (x . w1 + b1) = (dlogits . w2^T) where relu's output equals its input (tensor.py:872-877: also
 * at 0) -- everything of both passes that belongs to one row of the batch.  *done = 0 and nothing launched when the shape
 * is outside the kernel (K % 16 == 0, HD in {64, 128, 256}, P <= 16, M <= 4096).
 * Backward: dw1 (K, HD) = x^T . dz1, dw2 (HD, P) = h^T . dlogits (tensor.py:716), db1 (HD or NULL) / db2 (P or NULL) =
 * column sums of dz1 / dlogits (tensor.py:462-483), and loss[0] (or NULL) = the mean of the row losses in
 * dfhip_softmax_ce_forward's order.  x needs no gradient on this path (the caller takes the per-layer entry points if it
 * does). */
int dfhip_mlp2_ce_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* target, float* zpre, float* h, float* logits, float* dlogits, float* dz1,
                          float* rowloss, int M, int K, int HD, int P, int* done);
int dfhip_mlp2_ce_backward(const float* x, const float* h, const float* dz1, const float* dlogits, float* dw1, float* db1,
                           float* dw2, float* db2, const float* rowloss, float* loss, int M, int K, int HD, int P);

/* Dropout in training mode as one launch each way (nn/modules/dropout.py:31 `x * mask / (1 - p)`, the mask drawn on the
 * host): backward = 0: y = (x * mask) / keep; backward = 1 (x is the upstream gradient): y = (x / keep) * mask -- the two
 * primitive nodes' results bit for bit (backend_tensor.py:551-586, tensor.py:640-676).  keep = 1 - p. */
int dfhip_dropout_apply(const float* x, const float* mask, float* y, int64_t n, float keep, int backward);

/* Sum `src` over the axes in which dst_shape is 1 (or missing leading axes): replaces
 * the host un-broadcast of tensor.py:462-483.  Shapes are right-aligned.                */
int dfhip_reduce_to_shape(const float* src, float* dst, int src_ndim, const int64_t* src_shape,
                          int dst_ndim, const int64_t* dst_shape);
/* Single-axis reduce of a compact (outer, red, inner) array without the permute copy of
 * backend_tensor.py:624-657; op 0 = sum, 1 = max; result multiplied by `scale`
 * (BackendTensor.mean divides by the TOTAL element count, backend_tensor.py:659-662).   */
int dfhip_reduce_axis(const float* src, float* dst, int64_t outer, int64_t red, int64_t inner,
                      int op, float scale);
/* dst[(o, r, i)] = src[(o, i)] * scale : backward of the above (tensor.py:763-766).     */
int dfhip_broadcast_axis(const float* src, float* dst, int64_t outer, int64_t red,
                         int64_t inner, float scale);
/* dst += alpha * src (gradient accumulation, tensor.py:486-493).                        */
int dfhip_axpy(float* dst, const float* src, float alpha, int64_t n);

/* Adam for a captured (hipGraph) training step: dfhip_adam_step with the ten fp32 scalars of the
 * update (lr, beta1, beta2, 1-beta1, 1-beta2, 1-beta1^t, 1-beta2^t, eps, weight_decay, grad_scale;
 * optim/adam.py:42-60) read from device memory instead of the kernel arguments.  hyper_dev is
 * 16 + 16 * cap floats: a header (entry index, a block counter, cap) and `cap` entries, the scalars of steps
 * first_step, first_step + 1, ...  dfhip_adam_step_dev applies the entry the index points at and its last block
 * moves the index on, so a graph replays it unchanged step after step with no launch in between;
 * dfhip_adam_hyper_table (not inside a capture: it drains the stream and copies) is called once per `cap`
 * steps or when a scalar changes.  dfhip_adam_hyper: the one-entry form (a one-thread launch that writes the
 * scalars of step `step`; hyper_dev >= 32 floats). */
int dfhip_adam_hyper(float* hyper_dev, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int step, double grad_scale);
int dfhip_adam_hyper_table(float* hyper_dev, int cap, double lr, double beta1, double beta2, double eps,
                           double weight_decay, int first_step, double grad_scale);
/* Where the device is in a table written by dfhip_adam_hyper_table: *index = entries consumed so far (every
 * dfhip_adam_step_dev moves it on, on the device), *cap = its length.  Drains the stream (not inside a capture).  The
 * host mirror of optim/adam.py:34 (`self.t`) checks it against its own count before it rewrites a table: a replayed
 * step the host did not count -- or counted without running -- would otherwise train on with another step's bias
 * corrections (entries past the end are clamped to the last one). */
int dfhip_adam_hyper_index(const float* hyper_dev, int* index, int* cap);
int dfhip_adam_step_dev(int n_tensors, float* const* params, const float* const* grads,
                        float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                        const float* hyper_dev);

/* Multi-tensor optimizer steps.  ptr tables are HOST arrays of device pointers, sizes in
 * elements.  grad_scale multiplies every gradient first (1/world for data parallel).
 * Adam: optim/adam.py:28-64 (t = step count starting at 1).  SGD: optim/sgd.py:16-24.
 * Hyper-parameters arrive as doubles (Python floats) and are rounded to fp32 at exactly the
 * points where the reference's NumPy device rounds them (np.float32(s), backend_tensor.py:
 * 109-125), e.g. 1-beta1 and 1-beta1**t are formed in double first.                      */
int dfhip_adam_step(int n_tensors, float* const* params, const float* const* grads,
                    float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                    double lr, double beta1, double beta2, double eps, double weight_decay,
                    int step, double grad_scale);
int dfhip_sgd_step(int n_tensors, float* const* params, const float* const* grads,
                   float* const* momentum_buf, const int64_t* sizes, double lr, double momentum,
                   double weight_decay, int nesterov, double grad_scale);
/* The same update with its scalars in a 16-float device array that dfhip_sgd_hyper writes (a one-thread launch on the
 * library stream, refused inside a capture): a captured step replays dfhip_sgd_step_dev unchanged while a
 * learning-rate scheduler (optim/scheduler.py:13-61 mutates optimizer.lr) keeps moving the scalars from the host.
 * momentum_buf may be NULL only if the momentum written by dfhip_sgd_hyper is 0. */
int dfhip_sgd_hyper(float* hyper_dev, double lr, double momentum, double weight_decay, int nesterov,
                    double grad_scale);
int dfhip_sgd_step_dev(int n_tensors, float* const* params, const float* const* grads,
                       float* const* momentum_buf, const int64_t* sizes, const float* hyper_dev);

/* Pack / unpack a list of tensors into one flat bucket (gradient all-reduce payload).   */
int dfhip_pack(int n_tensors, const float* const* srcs, const int64_t* sizes, float* flat);
int dfhip_unpack(int n_tensors, float* const* dsts, const int64_t* sizes, const float* flat);

/* ---- data-parallel exchange over RCCL / xGMI (new functionality; the reference has no
 *      collective) ---------------------------------------------------- */
int dfhip_comm_unique_id(char* out128);                       /* ncclGetUniqueId          */
int dfhip_comm_init(int rank, int world, const char* id128);  /* ncclCommInitRank         */
int dfhip_allreduce_sum(float* buf, int64_t n);               /* The same all-reduce on a communication stream: it waits for the work already queued on the
 * library stream and then runs beside whatever is queued afterwards (the rest of the backward pass);
 * dfhip_comm_wait makes the library stream wait for all of them (call it before the optimizer). */
int dfhip_allreduce_sum_async(float* buf, int64_t n);
int dfhip_comm_wait(void);
/* Poll the communicator's asynchronous error state (ncclCommGetAsyncError): a peer that died or a link that failed does
 * not fail the collective that was already enqueued, the local stream just never finishes.  DFHIP_RUNTIME_ERROR (with
 * the RCCL error text in dfhip_last_error) when the communicator is broken -- the caller ends its process with a failure
 * status and the launcher reaps the other ranks.  dfhip_comm_wait and dfhip_comm_allreduce_max (the barrier, which waits
 * for its stream with this poll and a DFHIP_COMM_TIMEOUT_S bound instead of a bare hipStreamSynchronize) call it too. */
int dfhip_comm_check(void);

/* in place, library stream */
/* max over the ranks of one host float (one-element ncclAllReduce(max) + stream synchronisation): the barrier and
 * the "slowest rank" reduction of a multi-GPU run without any host-side process group */
int dfhip_comm_allreduce_max(float* value);
/* path of the shared object the RCCL symbols were resolved from ("" before dfhip_comm_unique_id / _init) */
int dfhip_comm_rccl_path(char* out, int cap);
/* Diagnostic: the shader clock (MHz, median over all SIMDs) the chip holds while every SIMD issues fp32 MFMAs back to
 * back for `milliseconds`, and the TFLOP/s that loop delivered (tflops may be NULL).  bench.py prices the matrix-core
 * families against the data-sheet peak and reports this figure beside it. */
int dfhip_clock_probe(double milliseconds, double* mhz, double* tflops);

/* size of the communicator and this process's rank in it, as RCCL reports them (ncclCommCount / ncclCommUserRank) */
int dfhip_comm_count(int* ranks, int* rank);
int dfhip_comm_destroy(void);

/* Host helper (no device call): the float32 mask np.random.binomial(1, keep, n) would draw next from NumPy's legacy
 * global generator (nn/modules/dropout.py:16-18 of the reference), continued from a copy of its state -- key[624]
 * and pos as returned by np.random.get_state() -- and the state advanced exactly as NumPy would have advanced it
 * (written back to key / pos; the caller hands it to np.random.set_state).  One double per element, thresholded. */
int dfhip_host_bernoulli_mask(uint32_t* key, int* pos, double keep, int64_t n, float* out);
/* the same mask generated straight into a pinned staging slot and copied to the device array `dst` on the library
 * stream (Dropout's from_numpy of nn/modules/dropout.py:28 without the intermediate host array) */
int dfhip_upload_bernoulli_mask(uint32_t* key, int* pos, double keep, int64_t n, float* dst);
/* `count` masks drawn one after the other from the same stream (the Dropout layers of a captured step in forward order:
 * nn/modules/dropout.py:16-18 once per layer) -- bytes over PCIe on the copy stream beside the step that is running, one
 * launch on the library stream that turns them into the float arrays dst[i] */
int dfhip_upload_bernoulli_masks(uint32_t* key, int* pos, int count, const double* keep, const int64_t* n, float* const* dst);
/* vector width of the helpers above: 0 plain C, 1 AVX2, 2 AVX-512 (F + BW + VL), -1 the widest the host has (the
 * default).  A level the host lacks is clamped; returns the level in effect.  For the tests: all three must agree. */
int dfhip_host_rng_isa(int level);

/* ---- whole-step capture (hipGraph) -------------------------------------------------- */
int dfhip_graph_begin(void);                   /* hipStreamBeginCapture on the library stream */
int dfhip_graph_end(void** graph_exec);        /* end capture + instantiate                  */
int dfhip_graph_launch(void* graph_exec);
int dfhip_graph_destroy(void* graph_exec);

/* ---- measurement helpers ------------------------------------------------------------ */
int dfhip_event_create(void** ev);
int dfhip_event_record(void* ev);              /* on the library stream */
int dfhip_event_elapsed_ms(void* start, void* stop, float* ms);  /* synchronises on stop */
int dfhip_event_destroy(void* ev);
/* per-kernel-family HIP-event timing: when enabled every launch of the family is
 * bracketed by events on the library stream (used by bench.py for roofline.achieved).   */
int dfhip_prof_enable(int on);
int dfhip_prof_reset(void);
/* hold the library stream (a one-thread kernel polling pinned host memory, gives up after timeout_ms <= 200) while
 * the host queues an eagerly issued step, then let it go: the event pairs then time launches that run back to back
 * as in a replayed graph instead of the host's gaps between them.  No counterpart in the reference. */
int dfhip_prof_gate_close(int timeout_ms);
int dfhip_prof_gate_open(void);
/* returns number of families written; names are static strings.  total_ms: between the two stream events
 * around a family call (kernel + dispatch); kernel_ms / kernel_launches (may be NULL): sum and count of the
 * family's kernels timed by event pairs bound to their dispatches (start / end stamps of the kernel itself, the
 * figure rocprofv3's kernel trace reports) */
int dfhip_prof_read(int cap, const char** names, double* total_ms, int64_t* calls,
                    double* flops, double* bytes, double* kernel_ms, int64_t* kernel_launches);
/* The same per KERNEL instantiation (name as the profiler prints it, e.g. "k_conv3x3_rowmma<4, 32, ...>"): launches,
 * the kernels' own durations summed, and the algorithmic flops / bytes of the family calls whose first launch it was.
 * Returns the number of rows written (<= cap); the name pointers stay valid until the library is unloaded. */
int dfhip_prof_read_kernels(int cap, const char** names, const char** families, double* kernel_ms, int64_t* launches,
                            double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* DFHIP_H */
