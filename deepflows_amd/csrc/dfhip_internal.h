// Internal helpers shared by the libdfhip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/dfhip.h"

namespace dfhip {

// thread-local error text behind dfhip_last_error()
void set_error(const char* fmt, ...);
hipStream_t stream();          // the single library stream
bool initialised();
bool profiling();             // per-family HIP-event profiling is switched on (dfhip_prof_enable)
bool capturing();             // a stream capture (dfhip_graph_begin) is open
int ensure_init();             // lazy dfhip_init(LOCAL_RANK or 0)
int num_cus();
int overlap_cus();            // DFHIP_OVERLAP_CUS: CUs (a multiple of 8) of the backward chain's stream, 0 = one stream
int make_masked_stream(hipStream_t* out, int first_bit, int nbits);
int host_bernoulli_mask_bits(uint32_t* key, int* pos, double keep, int64_t n, uint8_t* scratch, uint8_t* bits);   // dfhip_hostrng.hip: the Dropout mask as bits
bool bias_sum_defer(const float* src, float* dst, int N, int C, int HW);   // dfhip_wgrad.hip: a bias gradient as workgroups of the sweep's slab sum
bool conv_matmul_dtype_is_bf16();   // dfhip_set_conv_matmul_dtype(1): bf16 operands in the convolutions' matrix products (opt-in)
int chain_begin();            // the library's launches go to the chain stream (after everything queued so far) ...
int chain_end();              // ... and back (the library stream waits for the chain)

// RAII HIP-event bracket around one kernel family launch (active only if profiling on)
struct ProfScope {
  int slot, prev;
  hipEvent_t e0, e1;
  bool on;
  ProfScope(const char* family, double flops, double bytes);
  ~ProfScope();
};

// Inside an active ProfScope every launch also gets an event pair BOUND TO ITS DISPATCH (hipExtLaunchKernelGGL:
// the pair reads the kernel's own start / end time stamps, what rocprofv3's kernel trace reports); false when no
// scope is active.
bool prof_kernel_events(const void* host_function, hipEvent_t* start, hipEvent_t* stop);

// scratch workspace (grown on demand, never shrunk); stream-ordered reuse
float* workspace(size_t bytes);
// second, independent scratch area (for kernels that need two live scratch buffers)
float* workspace2(size_t bytes);
float* workspace_slot(int slot, size_t bytes);
// third area for small per-channel partials (never holds deferred slabs, so using it flushes nothing)
float* workspace_small(size_t bytes);

// Deferred work: the slab sum of a weight gradient rides on the NEXT weight-gradient launch (its
// blocks add the slabs while their own first loads are in flight).  Every C entry point flushes it
// (DF_INIT) except the few that are called between two weight gradients of a backward pass, which
// keep it (DF_INIT_KEEP) and flush only if one of their arrays is the pending destination
// (DF_TOUCH).
int flush_deferred();
bool deferred_touches(const void* p);
int deferred_scratch_slot();          // scratch slot holding the pending slabs, -1 if none
bool group_collecting();              // between dfhip_wgrad_group_begin / _end with weight gradients queued on the library stream
bool stem_combine_pending();          // the stem's one-pass backward left its last step to the group's slab sum (its totals
                                      // live in the second scratch area until then; dfhip_stem_combine.h)

// A training-mode BatchNorm2d standing between two convolutions, applied by the SECOND one on its operand loads
// (dfhip_conv2d_forward_bn_input): the partials the first convolution left behind and where the by-products go.
struct BnInput {
  const float* part;           // rows x C x (count, mean, centred second moment)
  int rows;
  const float* gamma;          // C (nullptr: 1)
  const float* beta;           // C (nullptr: 0)
  float eps, momentum;
  float* save_mean;            // C: batch mean / biased variance for the backward pass
  float* save_var;
  float* running_mean;         // C or nullptr: updated in place (batchnorm.py:44-52)
  float* running_var;
  float* affine;               // C x (scale, shift): y = x * scale + shift, for the weight gradient
};

// dfhip_conv.hip, shared with dfhip_wgrad.hip: pixel-group shape of a staged kernel, argument check of every convolution
// entry point (sets the error text, computes oh / ow), exact division by multiplication (magic = ceil(2^32 / d), 0 = d <= 1)
void plan_groups(int N, int HD, int WD, int tp, int& BI, int& BR, int& BC, int& gi, int& gr, int& gc);
int check_conv_args(const char* name, int N, int C, int H, int W, int O, int k, int stride, int pad, int& oh, int& ow);
inline unsigned magic_u32(unsigned d) { return d <= 1 ? 0u : (unsigned)((((uint64_t)1 << 32) + d - 1) / d); }


// dfhip_conv_rows.hip: register-operand 3x3 / stride 1 / pad 1 forward kernel; -1 when the shape is not eligible
// (bn: the operand is the BatchNorm of x, see BnInput)
int run_conv_rowmma(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int O,
                    int k, int stride, int pad, float* partials, int cap_rows, int* rows_out,
                    const BnInput* bn = nullptr, bool dry_run = false);   // dry_run: only the shape test

// dfhip_conv_s2.hip: 3x3 / stride 2 / pad 1 forward together with the block's 1x1 / stride 2 shortcut convolution
// (one launch, two outputs, BatchNorm partials of both); -1 when the shape is not eligible
int run_conv3x3s2_pair(const float* x, const float* w, const float* wd, float* y, float* yd, int N, int C, int H, int W,
                       int O, float* partials, float* partials_d, int cap_rows, int* rows_out, bool dry_run = false);

int run_conv_rowmma_dgrad(const float* dy, const float* w, const float* add, float* dx, int N, int C, int H, int W,
                          int O, int k, int stride, int pad);

// dfhip_conv_s2.hip: accumulate-mode input gradient of 3x3 / stride 2 / pad 1 (+ the block's 1x1 / stride 2 shortcut when dy2,
// w2 are given), one product per (pixel, tap) pair that exists; -1 when the shape is not eligible
int run_dgrad_s2_acc(const float* dy, const float* w, const float* dy2, const float* w2, const float* add, float* dx, int N,
                     int C, int H, int W, int O, bool dry_run = false);

}  // namespace dfhip

// ---- tile of a 2-D grid ----------------------------------------------------------------------------------------------
// (Rounds 3-5 carried an XCD-aware renumbering of the image-per-lane kernels' workgroups here -- every XCD's L2 holding 1/8 of
// a layer3-4 weight tensor instead of all of it: the L2 fills dropped as sized (2 x 2 forward 20.0 -> 9.0 MB per launch), no
// launch got shorter and the step longer (0.6875 -> 0.7003 ms: the weights come out of the Infinity Cache either way, and with
// the arrangement the 32 workgroups of an XCD ask for the same lines at the same moment).  Switched off in round 5, removed in
// round 6: profiles/r03b_xcd_*, docs/DESIGN_round_5.md item 4.)
struct DfTile { int x, y; };
// ---- point-wise pieces of the stem (BatchNorm -> ReLU -> MaxPool 2x2) shared by its kernels ----
#if defined(__HIPCC__)
__device__ __forceinline__ float bn_point(float v, float mean, float rsd, float w, float b) {
  // (rsd = 1 / sd, the same value in the forward and in the backward kernel: both see the same outputs)
  return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v, mean), rsd), w), b);
}
__device__ __forceinline__ float relu_point(float a) { return (a != a) ? a : (a > 0.0f ? a : 0.0f); }
__device__ __forceinline__ float stem_dy(float v, float pooled, float g, float mean, float rsd, float w, float b) {
  const float bo = bn_point(v, mean, rsd, w, b);
  const float r = relu_point(bo);
  const float gp = (pooled == r ? 1.0f : 0.0f) * g;
  return (r == bo ? 1.0f : 0.0f) * gp;
}
#endif

// backward pass of the stem fused into its weight gradient (dfhip_stem_backward): what the weight-gradient kernel needs
// to rebuild d loss / d (convolution output) from the convolution output z itself while it stages its dY operand
struct StemBwd {
  const float* y;              // pooled output (N, O, oh/2, ow/2)
  const float* g;              // its gradient
  const float* gamma;          // O (nullptr: 1)
  const float* beta;           // O (nullptr: 0)
  const float* mean;           // O: batch mean / biased variance of z
  const float* var;
  const float* partial;        // [O][S][2]: per-slice sums of dy and dy * xhat (k_bn_relu_pool_bwd<false>)
  int S;
  float eps;
  float* dgamma;               // O (may be nullptr)
  float* dbeta;
};
// dfhip_fused.hip: the statistics pass of the stem's backward (sums of dy and dy * xhat per channel slice);
// returns the partial array and its slice count
int stem_bwd_stats(const float* z, const float* y, const float* g, const float* gamma, const float* beta,
                   const float* mean, const float* var, int N, int C, int H, int W, float eps, const float** partial,
                   int* S);

// dfhip_fused.hip: y = relu(bn(sel)) with the statistics from convolution partials -- the second launch of the stem
// without its full-resolution tensor (dfhip_stem.hip)
int stem_pool_apply(const float* sel, int sel_planes, const float* partials, int rows, const float* gamma, const float* beta, float* y,
                    float* save_mean, float* save_var, float* running_mean, float* running_var, int N, int C, int P,
                    float eps, float momentum);

// every kernel launch of the library
#define DF_LAUNCH(kern, grid, block, lds, strm, ...)                                                  \
  do {                                                                                                \
    hipEvent_t pa__, pb__;                                                                            \
    if (dfhip::prof_kernel_events((const void*)(kern), &pa__, &pb__))                                 \
      hipExtLaunchKernelGGL(kern, grid, block, lds, strm, pa__, pb__, 0, __VA_ARGS__);                \
    else                                                                                              \
      hipLaunchKernelGGL(kern, grid, block, lds, strm, __VA_ARGS__);                                  \
  } while (0)

#define DF_CHECK_ARG(cond, ...)                      \
  do {                                               \
    if (!(cond)) {                                   \
      dfhip::set_error(__VA_ARGS__);                 \
      return DFHIP_INVALID_ARGUMENT;                 \
    }                                                \
  } while (0)

#define DF_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) {                                                             \
      dfhip::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                       __LINE__);                                                        \
      return e__ == hipErrorOutOfMemory ? DFHIP_OUT_OF_MEMORY : DFHIP_RUNTIME_ERROR;     \
    }                                                                                    \
  } while (0)

#define DF_LAUNCH_CHECK()                                                              \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      dfhip::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),     \
                       __FILE__, __LINE__);                                            \
      return DFHIP_RUNTIME_ERROR;                                                      \
    }                                                                                  \
  } while (0)

#define DF_INIT_KEEP()                         \
  do {                                         \
    int s__ = dfhip::ensure_init();            \
    if (s__ != DFHIP_OK) return s__;           \
  } while (0)

#define DF_INIT()                              \
  do {                                         \
    int s__ = dfhip::ensure_init();            \
    if (s__ != DFHIP_OK) return s__;           \
    s__ = dfhip::flush_deferred();             \
    if (s__ != DFHIP_OK) return s__;           \
  } while (0)

#define DF_TOUCH(p)                                  \
  do {                                               \
    if (dfhip::deferred_touches((const void*)(p))) { \
      int s__ = dfhip::flush_deferred();             \
      if (s__ != DFHIP_OK) return s__;               \
    }                                                \
  } while (0)

static inline int64_t df_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
