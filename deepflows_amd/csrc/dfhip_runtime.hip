// libdfhip runtime: device/stream ownership, the caching pool behind Array(size),
// host<->device copies, hipGraph capture, HIP-event profiling and the RCCL communicator.
//
// Replaces the runtime half of the reference's CUDA module
// (DeepFlows/backend/backend_src/ndarray_backend_cuda.cu:48-83 CudaArray = raw
// cudaMalloc/cudaFree per Array, :667-716 blocking cudaMemcpy, default stream everywhere).
// MI355X design: ONE process per GPU, ONE explicit non-blocking stream, a size-bucketed
// stream-ordered pool (a training step allocates ~3.6k arrays, SURVEY F8 — none of them may
// reach hipMalloc in steady state), capture of the whole step into a hipGraph.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

#include <cxxabi.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dfhip_internal.h"
#include "dfhip_head.h"

namespace dfhip {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::mutex g_mu;
static bool g_init = false;
static int g_device = 0;
static hipStream_t g_stream = nullptr;
static int g_num_cus = 256;
static char g_devname[256] = "";

// Round 5: CU-partitioned streams.  Two streams of this stack run their kernels at the same time only when their CU masks
// are DISJOINT (tools/probes/two_streams2.hip: a chain of dependent kernels beside one long kernel takes the sum of the two on
// plain streams -- any number of hardware queues -- and the maximum on masked streams that share no CU).  A mask bit stands
// for one CU of EVERY XCD (tools/probes/cu_mask_map.hip): bits 0..8n-1 = n CUs per XCD.  DFHIP_OVERLAP_CUS=n: during a
// backward sweep (dfhip_wgrad_group_begin .. _end) the library's launches go to a "chain" stream on n CUs and the grouped
// weight gradients (DFHIP_WGRAD_SIDE) to a stream on the other 256 - n.
static hipStream_t g_chain = nullptr;
static hipEvent_t g_chain_ev = nullptr;
static bool g_chain_on = false;
hipStream_t stream() { return g_chain_on ? g_chain : g_stream; }
int overlap_cus() {
  static int n = getenv("DFHIP_OVERLAP_CUS") ? atoi(getenv("DFHIP_OVERLAP_CUS")) : 0;
  return (n >= 8 && n <= 248) ? (n & ~7) : 0;
}
int make_masked_stream(hipStream_t* out, int first_bit, int nbits) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = first_bit; c < first_bit + nbits && c < 256; c++) mask[c / 32] |= 1u << (c % 32);
  DF_HIP(hipExtStreamCreateWithCUMask(out, 8, mask));
  return DFHIP_OK;
}
int chain_begin() {
  if (!overlap_cus() || g_chain_on) return DFHIP_OK;
  if (!g_chain) {
    int st = make_masked_stream(&g_chain, 0, overlap_cus());
    if (st != DFHIP_OK) return st;
    DF_HIP(hipEventCreateWithFlags(&g_chain_ev, hipEventDisableTiming));
  }
  DF_HIP(hipEventRecord(g_chain_ev, g_stream));
  DF_HIP(hipStreamWaitEvent(g_chain, g_chain_ev, 0));
  g_chain_on = true;
  return DFHIP_OK;
}
int chain_end() {
  if (!g_chain_on) return DFHIP_OK;
  g_chain_on = false;
  DF_HIP(hipEventRecord(g_chain_ev, g_chain));
  DF_HIP(hipStreamWaitEvent(g_stream, g_chain_ev, 0));
  return DFHIP_OK;
}
bool initialised() { return g_init; }
int num_cus() { return g_num_cus; }

static int do_init(int device_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_init) {
    if (device_id >= 0 && device_id != g_device) {
      set_error("dfhip already initialised on device %d (asked for %d)", g_device, device_id);
      return DFHIP_INVALID_ARGUMENT;
    }
    return DFHIP_OK;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    return DFHIP_RUNTIME_ERROR;
  }
  if (device_id < 0) {
    const char* lr = getenv("LOCAL_RANK");
    device_id = lr ? atoi(lr) % count : 0;
  }
  if (device_id >= count) {
    set_error("device %d out of range (%d visible)", device_id, count);
    return DFHIP_INVALID_ARGUMENT;
  }
  DF_HIP(hipSetDevice(device_id));
  if (const char* cm = getenv("DFHIP_CU_MASK")) {
    // diagnostic (round 5, tools/probes/two_streams2.hip): the library stream on a subset of the CUs -- "lo<N>" = mask bits
    // 0..N-1, "hi<N>" = bits 256-N..255 (which CUs those are: tools/probes/cu_mask_map.hip)
    // (DIAGNOSTIC ONLY.  The value is checked: a prefix other than lo / hi, a missing or out-of-range count used to give an
    // all-zero mask -- ADVICE r5)
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t len = strlen(cm);
    const bool lo = len >= 3 && cm[0] == 'l' && cm[1] == 'o', hi = len >= 3 && cm[0] == 'h' && cm[1] == 'i';
    const int n = (lo || hi) ? atoi(cm + 2) : 0;
    if (n < 1 || n > 256) {
      set_error("DFHIP_CU_MASK=%s: expected lo<N> or hi<N> with 1 <= N <= 256", cm);
      return DFHIP_INVALID_ARGUMENT;
    }
    for (int c = 0; c < 256; c++)
      if ((lo && c < n) || (hi && c >= 256 - n)) mask[c / 32] |= 1u << (c % 32);
    DF_HIP(hipExtStreamCreateWithCUMask(&g_stream, 8, mask));
  } else {
    DF_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  }
  hipDeviceProp_t prop;
  DF_HIP(hipGetDeviceProperties(&prop, device_id));
  g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  snprintf(g_devname, sizeof(g_devname), "%s (%s)", prop.name, prop.gcnArchName);
  g_device = device_id;
  g_init = true;
  return DFHIP_OK;
}

int ensure_init() { return g_init ? DFHIP_OK : do_init(-1); }


// ---------------------------------------------------------------------------------------
// Caching pool.  Exact-size free lists (rounded to 256 B); a training loop requests the
// same sizes every step, so after one warm-up step nothing reaches hipMalloc.  Frees are
// stream-ordered: a buffer handed back is only ever reused by later work on the same
// stream, so no event is needed.  Buffers served while a capture is open become owned by
// that capture (they are recycled only inside it and are never handed to eager code
// afterwards, because the instantiated graph keeps writing to them on every replay).
// ---------------------------------------------------------------------------------------
struct Block {
  size_t bytes;
  int owner;  // 0 = general pool, >0 = capture id
};
static std::unordered_map<void*, Block> g_blocks;                         // every live/free block
static std::unordered_map<size_t, std::vector<void*>> g_free;             // general pool
static std::unordered_map<int, std::unordered_map<size_t, std::vector<void*>>> g_cap_free;
static int g_capture_open = 0;  // id of the capture being recorded (0 = none)
static int g_capture_seq = 0;
static int64_t g_reserved = 0, g_in_use = 0, g_n_hipmalloc = 0;

static size_t round_bytes(int64_t n_floats) {
  size_t b = (size_t)(n_floats > 0 ? n_floats : 1) * sizeof(float);
  return (b + 255) & ~(size_t)255;
}

static int pool_alloc(size_t bytes, void** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  void* p = nullptr;
  if (g_capture_open) {
    auto& fl = g_cap_free[g_capture_open][bytes];
    if (!fl.empty()) {
      p = fl.back();
      fl.pop_back();
    }
  }
  if (!p) {
    auto it = g_free.find(bytes);
    if (it != g_free.end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
    }
  }
  if (!p) {
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      // give cached blocks back to the driver once, then retry
      for (auto& kv : g_free) {
        for (void* q : kv.second) {
          g_reserved -= (int64_t)g_blocks[q].bytes;
          g_blocks.erase(q);
          (void)hipFree(q);
        }
        kv.second.clear();
      }
      e = hipMalloc(&p, bytes);
      if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return DFHIP_OUT_OF_MEMORY;
      }
    }
    g_n_hipmalloc++;
    {
    }
    g_reserved += (int64_t)bytes;
    g_blocks[p] = Block{bytes, 0};
  }
  g_blocks[p].owner = g_capture_open;
  g_in_use += (int64_t)bytes;
  *out = p;
  return DFHIP_OK;
}

static int pool_free(void* p) {
  if (!p) return DFHIP_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_blocks.find(p);
  if (it == g_blocks.end()) {
    set_error("dfhip_free: pointer %p was not allocated by dfhip_malloc", p);
    return DFHIP_INVALID_ARGUMENT;
  }
  g_in_use -= (int64_t)it->second.bytes;
  if (it->second.owner)
    g_cap_free[it->second.owner][it->second.bytes].push_back(p);
  else
    g_free[it->second.bytes].push_back(p);
  return DFHIP_OK;
}

struct ScratchSlot {
  void* p = nullptr;
  size_t bytes = 0;
};
static ScratchSlot g_ws[3];
static float* scratch(int i, size_t bytes) {
  // A scratch area is simply a pool block that is returned to the pool as soon as a
  // bigger one is needed; stream order makes the hand-over safe.
  if (deferred_scratch_slot() == i) (void)flush_deferred();   // its slabs live in this slot
  if (i == 1 && stem_combine_pending()) (void)flush_deferred();   // the stem's totals live in this one
  bytes = (bytes + 255) & ~(size_t)255;
  ScratchSlot& s = g_ws[i];
  bool wrong_owner = false;
  if (s.p) {
    std::lock_guard<std::mutex> lk(g_mu);
    wrong_owner = g_blocks[s.p].owner != g_capture_open;
  }
  if (s.p && s.bytes >= bytes && !wrong_owner) return (float*)s.p;
  if (s.p) pool_free(s.p);
  s.p = nullptr;
  s.bytes = 0;
  void* p = nullptr;
  if (pool_alloc(bytes, &p) != DFHIP_OK) return nullptr;
  s.p = p;
  s.bytes = bytes;
  return (float*)p;
}
float* workspace(size_t bytes) { return scratch(0, bytes); }
float* workspace2(size_t bytes) { return scratch(1, bytes); }
float* workspace_slot(int slot, size_t bytes) { return scratch(slot & 1, bytes); }
float* workspace_small(size_t bytes) { return scratch(2, bytes); }

// ---------------------------------------------------------------------------------------
// profiling: HIP events around kernel families on the library stream
// ---------------------------------------------------------------------------------------
struct ProfSlot {
  const char* name;
  double ms = 0, flops = 0, bytes = 0;
  int64_t calls = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  // event pairs bound to the dispatches themselves: kernel-only time of the family's launches
  double kernel_ms = 0;
  int64_t kernel_launches = 0;
  struct KPending { hipEvent_t a, b; int rec; };
  std::vector<KPending> kpending;
};
// per KERNEL (the instantiation's own name, from DF_LAUNCH): launches and the kernels' own durations, plus the
// algorithmic flops / bytes of the scopes whose first launch it was
struct KernelRec {
  const void* fn;              // the launch site's host function
  std::string name;            // the instantiation's own name, demangled (hipKernelNameRefByPtr)
  int slot;                    // family
  double ms = 0, flops = 0, bytes = 0;
  int64_t launches = 0;
};
static std::vector<KernelRec> g_krec;
static double g_scope_flops = 0, g_scope_bytes = 0;     // of the innermost scope, not yet handed to a kernel
static int g_prof_scope = -1;           // slot of the innermost active ProfScope
static bool g_prof_on = false;
bool profiling() { return g_prof_on; }
bool capturing() { return g_capture_open != 0; }
static std::vector<ProfSlot> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
static void prof_resolve(ProfSlot& s) {
  for (auto& pr : s.pending) {
    float ms = 0;
    if (hipEventSynchronize(pr.second) == hipSuccess &&
        hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
      s.ms += ms;
    g_event_pool.push_back(pr.first);
    g_event_pool.push_back(pr.second);
  }
  s.pending.clear();
  for (auto& pr : s.kpending) {
    float ms = 0;
    if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      s.kernel_ms += ms;
      s.kernel_launches++;
      if (pr.rec >= 0) { g_krec[pr.rec].ms += ms; g_krec[pr.rec].launches++; }
    }
    g_event_pool.push_back(pr.a);
    g_event_pool.push_back(pr.b);
  }
  s.kpending.clear();
}
static int kernel_record(const void* fn, int slot) {
  if (!fn) return -1;
  for (size_t i = 0; i < g_krec.size(); i++)
    if (g_krec[i].slot == slot && g_krec[i].fn == fn) return (int)i;
  KernelRec r;
  r.fn = fn;
  r.slot = slot;
  // "void (anonymous namespace)::k_x<1, true>(float const*, int)" -> "k_x<1, true>" (as tools/pmc_summary.py shortens
  // the profiler's names)
  const char* mangled = hipKernelNameRefByPtr(fn, g_stream);
  std::string t = mangled ? mangled : "?";
  int status = 0;
  char* dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
  if (dem && status == 0) t = dem;
  free(dem);
  for (const char* drop : {"(anonymous namespace)::", "dfhip::"}) {
    size_t q;
    while ((q = t.find(drop)) != std::string::npos) t.erase(q, strlen(drop));
  }
  if (t.rfind("void ", 0) == 0) t = t.substr(5);
  int depth = 0;
  for (size_t i = 0; i < t.size(); i++) {
    if (t[i] == '<') depth++;
    else if (t[i] == '>') depth--;
    else if (t[i] == '(' && depth == 0) { t = t.substr(0, i); break; }
  }
  r.name = t;
  g_krec.push_back(r);
  return (int)g_krec.size() - 1;
}
// DFHIP_KERNEL_LOG=<file> (diagnostic): every kernel instantiation this process launches is appended to the file once, by
// name -- the union over the test-suite and the benches says which instantiations of the library are reachable by default
// (tools/kernel_coverage.py compares it with the stubs the shared object holds)
static void note_kernel(const void* fn) {
  static std::unordered_map<const void*, bool> seen;
  if (!seen.emplace(fn, true).second) return;
  static const char* path = getenv("DFHIP_KERNEL_LOG");
  const char* mangled = hipKernelNameRefByPtr(fn, g_stream);
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "%s\n", mangled ? mangled : "?");
    fclose(f);
  }
}
bool prof_kernel_events(const void* fn, hipEvent_t* start, hipEvent_t* stop) {
  static const bool klog = getenv("DFHIP_KERNEL_LOG") != nullptr;
  if (klog) note_kernel(fn);
  if (g_prof_scope < 0 || g_capture_open) return false;
  hipEvent_t a = get_event(), b = get_event();
  if (!a || !b) return false;
  const int rec = kernel_record(fn, g_prof_scope);
  if (rec >= 0) {
    g_krec[rec].flops += g_scope_flops;              // the scope's algorithmic work goes to its first launch
    g_krec[rec].bytes += g_scope_bytes;
    g_scope_flops = g_scope_bytes = 0;
  }
  g_prof[g_prof_scope].kpending.push_back({a, b, rec});
  *start = a;
  *stop = b;
  return true;
}
ProfScope::ProfScope(const char* family, double flops, double bytes) : slot(-1), e0(nullptr), e1(nullptr), on(false) {
  if (!g_prof_on || g_capture_open) return;
  for (size_t i = 0; i < g_prof.size(); i++)
    if (g_prof[i].name == family || strcmp(g_prof[i].name, family) == 0) slot = (int)i;
  if (slot < 0) {
    ProfSlot s;
    s.name = family;
    g_prof.push_back(s);
    slot = (int)g_prof.size() - 1;
  }
  g_prof[slot].calls++;
  g_prof[slot].flops += flops;
  g_prof[slot].bytes += bytes;
  e0 = get_event();
  e1 = get_event();
  if (!e0 || !e1) return;
  on = true;
  prev = g_prof_scope;
  g_prof_scope = slot;
  g_scope_flops = flops;
  g_scope_bytes = bytes;
  (void)hipEventRecord(e0, g_stream);
}
ProfScope::~ProfScope() {
  if (!on) return;
  g_prof_scope = prev;
  (void)hipEventRecord(e1, g_stream);
  ProfSlot& s = g_prof[slot];
  s.pending.emplace_back(e0, e1);
  if (s.pending.size() > 2048 || s.kpending.size() > 4096) prof_resolve(s);
}

}  // namespace dfhip

using namespace dfhip;

extern "C" {

int dfhip_init(int device_id) { return do_init(device_id); }

int dfhip_device_count(int* out) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  *out = (e == hipSuccess) ? c : 0;
  return DFHIP_OK;
}

const char* dfhip_last_error(void) { return g_err; }

int dfhip_sync(void) {
  DF_INIT();
  DF_HIP(hipStreamSynchronize(g_stream));
  return DFHIP_OK;
}

void* dfhip_stream(void) {
  (void)flush_deferred();
  return (void*)g_stream;
}

// ---- clock probe: the shader clock the chip holds under a dense fp32-MFMA load -------------------------------------
// One wave per SIMD on every CU issues independent 16x16x4 fp32 MFMAs (four accumulator chains, register operands,
// data that is neither zero nor constant); lane 0 of each wave stamps the shader-cycle counter (s_memtime) and the
// 100 MHz real-time counter around the loop: cycles / real time = the clock during the loop (MI355X_MICROARCH.md,
// DVFS give-back item 6).  The matrix-pipe roofline of a kernel is 64 flop per cycle and SIMD at THIS clock; the
// 157.3 TFLOP/s of the data sheet is the same figure at 2.4 GHz.
__global__ __launch_bounds__(256) void k_clock_probe(unsigned long long* __restrict__ out, int iters, float seed) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  // two operand pairs of opposite sign, used alternately: the sums stay bounded and no vector instruction sits
  // between the MFMAs (a VALU write to an MFMA operand costs wait states)
  const float a0 = seed + 0.001f * (float)lane, a1 = -a0, b = 1.0f - 0.002f * (float)lane;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; it += 2) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  const f32x4 cs = c0 + c1 + c2 + c3;
  const float sink = cs[0] + cs[1] + cs[2] + cs[3];
  if (lane == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    out[2 * w] = t1 - t0;
    out[2 * w + 1] = (r1 - r0) | ((unsigned long long)(sink == 12345.678f) << 63);     // (keeps the MFMAs alive)
  }
}

int dfhip_clock_probe(double milliseconds, double* mhz, double* tflops) {
  DF_INIT();
  DF_CHECK_ARG(mhz && milliseconds > 0.0 && milliseconds <= 2000.0, "clock_probe: bad arguments");
  const int waves = num_cus() * 4;
  unsigned long long* d = nullptr;
  DF_HIP(hipMalloc((void**)&d, (size_t)waves * 2 * sizeof(unsigned long long)));
  std::vector<unsigned long long> h((size_t)waves * 2);
  // launches of ~0.25 ms (4 MFMAs x 32 cycles per iteration at <= 2.4 GHz) back to back for the requested time; the
  // stamps of the LAST launch are read
  const int iters = 4700;                        // (even: the loop body is two iterations)
  const int launches = std::max(1, (int)(milliseconds / 0.25));
  for (int l = 0; l < launches; l++) {
    hipLaunchKernelGGL(k_clock_probe, dim3((unsigned)num_cus()), dim3(256), 0, g_stream, d, iters, 0.37f + 0.01f * (float)l);
  }
  DF_LAUNCH_CHECK();
  DF_HIP(hipMemcpyAsync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, g_stream));
  DF_HIP(hipStreamSynchronize(g_stream));
  (void)hipFree(d);
  std::vector<double> clk;
  for (int w = 0; w < waves; w++) {
    const double cyc = (double)h[2 * (size_t)w], ticks = (double)(h[2 * (size_t)w + 1] & ~(1ull << 63));
    if (ticks > 0) clk.push_back(cyc / ticks * 100.0);       // MHz: the real-time counter runs at 100 MHz
  }
  DF_CHECK_ARG(!clk.empty(), "clock_probe: no stamps came back");
  std::sort(clk.begin(), clk.end());
  *mhz = clk[clk.size() / 2];
  if (tflops) {
    // what the loop delivered: 4 MFMAs x 2048 flop per iteration and wave
    double best = 0.0;
    for (int w = 0; w < waves; w++) {
      const double ticks = (double)(h[2 * (size_t)w + 1] & ~(1ull << 63));
      if (ticks > best) best = ticks;
    }
    *tflops = best > 0 ? (double)waves * iters * 4.0 * 2048.0 / (best * 1e-8) / 1e12 : 0.0;
  }
  return DFHIP_OK;
}

int dfhip_device_name(char* out, int cap) {
  DF_INIT();
  snprintf(out, cap, "%s", g_devname);
  return DFHIP_OK;
}

int dfhip_malloc(int64_t n_floats, float** out) {
  DF_INIT_KEEP();
  DF_CHECK_ARG(n_floats >= 0, "Array size must be >= 0 (got %lld)", (long long)n_floats);
  void* p = nullptr;
  int s = pool_alloc(round_bytes(n_floats), &p);
  *out = (float*)p;
  return s;
}

int dfhip_free(float* p) {
  if (mlp_bwd_defer_free(p)) return DFHIP_OK;       // (an operand of a launch that is still deferred: freed behind it, dfhip_mlp.hip)
  DF_TOUCH(p);
  return pool_free((void*)p);
}

int dfhip_pool_stats(int64_t* bytes_reserved, int64_t* bytes_in_use, int64_t* n_hipmalloc) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (bytes_reserved) *bytes_reserved = g_reserved;
  if (bytes_in_use) *bytes_in_use = g_in_use;
  if (n_hipmalloc) *n_hipmalloc = g_n_hipmalloc;
  return DFHIP_OK;
}

int dfhip_pool_trim(void) {
  DF_INIT();
  DF_HIP(hipStreamSynchronize(g_stream));
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_free) {
    for (void* q : kv.second) {
      g_reserved -= (int64_t)g_blocks[q].bytes;
      g_blocks.erase(q);
      (void)hipFree(q);
    }
    kv.second.clear();
  }
  return DFHIP_OK;
}

// Host buffers handed over by NumPy are pageable: stage through pinned memory so the copy
// is a real async DMA on the library stream, then wait (the protocol's from_numpy /
// to_numpy are blocking calls, ndarray_backend_cuda.cu:667-716).
static void* g_pinned = nullptr;
static size_t g_pinned_bytes = 0;
static int pinned(size_t bytes, void** out) {
  if (g_pinned_bytes < bytes) {
    if (g_pinned) (void)hipHostFree(g_pinned);
    g_pinned = nullptr;
    g_pinned_bytes = 0;
    size_t want = bytes < (size_t)(8 << 20) ? (size_t)(8 << 20) : bytes;
    DF_HIP(hipHostMalloc(&g_pinned, want, hipHostMallocDefault));
    g_pinned_bytes = want;
  }
  *out = g_pinned;
  return DFHIP_OK;
}

// Host -> device without draining the stream: a ring of pinned staging slots, each guarded by an
// event recorded behind its copy; the host blocks only when it laps the ring (the reference's
// per-step uploads — dropout masks, batches — used to stall the whole pipeline here).
struct H2dSlot {
  void* p = nullptr;
  size_t bytes = 0;
  hipEvent_t done = nullptr;
  bool busy = false;
};
static H2dSlot g_h2d[4];
static int g_h2d_next = 0;

// device-side copy out of pinned host memory (the kernel reads it over PCIe): stays on the compute queue of the
// library stream -- an SDMA copy between two graph launches costs a hop to the copy engine's queue and back
__global__ __launch_bounds__(256) void k_copy_from_host(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = src[(n4 << 2) + threadIdx.x];
}

int dfhip_h2d(const float* host, float* dst, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0 && (n == 0 || (host && dst)), "h2d: null pointer");
  if (n == 0) return DFHIP_OK;
  // A copy recorded into a graph would re-read one of the four RECYCLED pinned slots on every replay (whatever a
  // later upload left there): refuse it.  Host data a captured step needs (e.g. Dropout masks) is written into
  // graph-owned device arrays by eager uploads in front of each replay (graph_step.CapturedTrainStep).
  if (g_capture_open) {
    set_error("h2d inside a graph capture: upload into a graph-owned array before the replay instead");
    return DFHIP_RUNTIME_ERROR;
  }
  size_t bytes = (size_t)n * sizeof(float);
  const size_t chunk = (size_t)64 << 20;
  for (size_t off = 0; off < bytes; off += chunk) {
    size_t b = bytes - off < chunk ? bytes - off : chunk;
    H2dSlot& sl = g_h2d[g_h2d_next];
    g_h2d_next = (g_h2d_next + 1) & 3;
    if (!sl.done) DF_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (sl.busy) {
      DF_HIP(hipEventSynchronize(sl.done));
      sl.busy = false;
    }
    if (sl.bytes < b) {
      if (sl.p) (void)hipHostFree(sl.p);
      sl.p = nullptr;
      sl.bytes = 0;
      size_t want = b < (size_t)(1 << 20) ? (size_t)(1 << 20) : b;
      DF_HIP(hipHostMalloc(&sl.p, want, hipHostMallocDefault));
      sl.bytes = want;
    }
    memcpy(sl.p, (const char*)host + off, b);
    // up to 32 MB: a kernel on the library stream reads the pinned slot (no hop to the copy engine's queue between
    // two launches of the stream -- with per-step uploads in front of graph replays the SDMA path cost 0.1-0.7 ms per step)
    if (b <= ((size_t)32 << 20) && ((((uintptr_t)dst) + off) & 15) == 0) {
      const int64_t nf = (int64_t)(b / sizeof(float));
      const int blocks = (int)std::min<int64_t>(1024, (nf / 4 + 255) / 256 + 1);
      hipLaunchKernelGGL(k_copy_from_host, dim3(blocks), dim3(256), 0, g_stream, (const float*)sl.p,
                         (float*)((char*)dst + off), nf);
      DF_LAUNCH_CHECK();
    } else {
      DF_HIP(hipMemcpyAsync((char*)dst + off, sl.p, b, hipMemcpyHostToDevice, g_stream));
    }
    DF_HIP(hipEventRecord(sl.done, g_stream));
    sl.busy = true;
  }
  return DFHIP_OK;
}

// ---- Dropout masks (the generator itself: dfhip_hostrng.hip) ------------------------------------------------------
// the mask generated straight into a pinned staging slot and sent to `dst` on the library stream: no intermediate host
// array (a fresh 2 MB NumPy array per step is 512 first-touch page faults) and no second copy.  A 16-byte aligned
// destination (every array of the pool) takes the bit path of dfhip_upload_bernoulli_masks below.
int dfhip_upload_bernoulli_masks(uint32_t* key, int* pos, int count, const double* keep, const int64_t* n, float* const* dst);
int dfhip_upload_bernoulli_mask(uint32_t* key, int* pos, double keep, int64_t n, float* dst) {
  DF_INIT();
  DF_CHECK_ARG(key && pos && dst && n >= 0, "upload_bernoulli_mask: bad arguments");
  if (n == 0) return DFHIP_OK;
  if (g_capture_open) {
    set_error("upload_bernoulli_mask inside a graph capture: register the mask as a graph input instead");
    return DFHIP_RUNTIME_ERROR;
  }
  if ((((uintptr_t)dst) & 15) == 0) return dfhip_upload_bernoulli_masks(key, pos, 1, &keep, &n, &dst);
  // (a destination that is not 16-byte aligned: the mask as floats through a plain copy)
  const size_t b = (size_t)n * sizeof(float);
  H2dSlot& sl = g_h2d[g_h2d_next];
  g_h2d_next = (g_h2d_next + 1) & 3;
  if (!sl.done) DF_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  if (sl.busy) {
    DF_HIP(hipEventSynchronize(sl.done));
    sl.busy = false;
  }
  if (sl.bytes < b) {
    if (sl.p) (void)hipHostFree(sl.p);
    sl.p = nullptr;
    sl.bytes = 0;
    size_t want = b < (size_t)(1 << 20) ? (size_t)(1 << 20) : b;
    DF_HIP(hipHostMalloc(&sl.p, want, hipHostMallocDefault));
    sl.bytes = want;
  }
  int st = dfhip_host_bernoulli_mask(key, pos, keep, n, (float*)sl.p);
  if (st != DFHIP_OK) return st;
  DF_HIP(hipMemcpyAsync(dst, sl.p, b, hipMemcpyHostToDevice, g_stream));
  DF_HIP(hipEventRecord(sl.done, g_stream));
  sl.busy = true;
  return DFHIP_OK;
}

int dfhip_d2h(const float* src, float* host, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(n >= 0 && (n == 0 || (host && src)), "d2h: null pointer");
  if (n == 0) return DFHIP_OK;
  size_t bytes = (size_t)n * sizeof(float);
  const size_t chunk = (size_t)64 << 20;
  for (size_t off = 0; off < bytes; off += chunk) {
    size_t b = bytes - off < chunk ? bytes - off : chunk;
    void* stage = nullptr;
    int s = pinned(b, &stage);
    if (s != DFHIP_OK) return s;
    DF_HIP(hipMemcpyAsync(stage, (const char*)src + off, b, hipMemcpyDeviceToHost, g_stream));
    DF_HIP(hipStreamSynchronize(g_stream));
    memcpy((char*)host + off, stage, b);
  }
  return DFHIP_OK;
}

// ---- asynchronous, double-buffered upload (input pipeline) -------------------------------------
// Two pinned slots and a copy stream: dfhip_upload_async(slot) copies the host bytes into the
// slot's pinned buffer and enqueues the H2D on the copy stream (it returns as soon as the memcpy
// into pinned memory is done); dfhip_upload_wait(slot) makes the library stream wait for that
// copy.  The host fills slot k+1 while the GPU trains on slot k; a slot is reused only after the
// copy that last read it has finished (event wait on the host, normally already signalled).
struct UploadSlot {
  void* pinned = nullptr;
  size_t bytes = 0;
  hipEvent_t done = nullptr;     // H2D of this slot finished
  hipEvent_t consumed = nullptr; // the library stream has passed the kernels that read the destination
  bool in_flight = false;
  bool released = false;         // `consumed` was recorded behind the readers of the previous upload (dfhip_upload_release)
  const void* last_dst = nullptr;  // the array the release point belongs to: the flag is honoured for an upload INTO IT only
};
static UploadSlot g_up[2];
static hipStream_t g_copy_stream = nullptr;

int dfhip_upload_async(const void* host, float* dst, int64_t bytes, int slot) {
  DF_INIT_KEEP();
  DF_CHECK_ARG(slot == 0 || slot == 1, "upload_async: slot must be 0 or 1");
  DF_CHECK_ARG(bytes >= 0 && (bytes == 0 || (host && dst)), "upload_async: null pointer");
  if (bytes == 0) return DFHIP_OK;
  if (!g_copy_stream) DF_HIP(hipStreamCreateWithFlags(&g_copy_stream, hipStreamNonBlocking));
  UploadSlot& u = g_up[slot];
  if (!u.done) {
    DF_HIP(hipEventCreateWithFlags(&u.done, hipEventDisableTiming));
    DF_HIP(hipEventCreateWithFlags(&u.consumed, hipEventDisableTiming));
  }
  if (u.in_flight) DF_HIP(hipEventSynchronize(u.done));      // the pinned buffer is free again
  if (u.bytes < (size_t)bytes) {
    if (u.pinned) (void)hipHostFree(u.pinned);
    u.pinned = nullptr;
    u.bytes = 0;
    DF_HIP(hipHostMalloc(&u.pinned, (size_t)bytes, hipHostMallocDefault));
    u.bytes = (size_t)bytes;
  }
  memcpy(u.pinned, host, (size_t)bytes);
  // The destination may still be read by kernels the library stream has queued for its previous use.  Where the caller
  // marked the end of those readers (dfhip_upload_release) the copy waits for exactly that point -- and runs beside
  // whatever was queued after it: the training step of the batch before --; otherwise it waits for everything queued so far
  // (round 3's behaviour: the copy of batch k + 1 then started only when step k had finished, ~0.1 ms per step exposed).
  // (the release point says "the readers of the slot's LAST destination are queued": an upload through the same slot into
  // another array -- a second pipeline of another batch shape shares g_up -- waits for everything queued so far instead)
  if (!u.released || u.last_dst != (const void*)dst) DF_HIP(hipEventRecord(u.consumed, g_stream));
  u.released = false;
  u.last_dst = dst;
  DF_HIP(hipStreamWaitEvent(g_copy_stream, u.consumed, 0));
  DF_HIP(hipMemcpyAsync(dst, u.pinned, (size_t)bytes, hipMemcpyHostToDevice, g_copy_stream));
  DF_HIP(hipEventRecord(u.done, g_copy_stream));
  u.in_flight = true;
  return DFHIP_OK;
}

int dfhip_upload_wait(int slot) {
  DF_INIT_KEEP();
  DF_CHECK_ARG(slot == 0 || slot == 1, "upload_wait: slot must be 0 or 1");
  UploadSlot& u = g_up[slot];
  if (u.in_flight) DF_HIP(hipStreamWaitEvent(g_stream, u.done, 0));
  return DFHIP_OK;
}

int dfhip_upload_release(int slot) {
  DF_INIT_KEEP();
  DF_CHECK_ARG(slot == 0 || slot == 1, "upload_release: slot must be 0 or 1");
  UploadSlot& u = g_up[slot];
  if (!u.done) return DFHIP_OK;                     // (nothing was ever uploaded through this slot)
  DF_HIP(hipEventRecord(u.consumed, g_stream));
  u.released = true;
  return DFHIP_OK;
}

// ---- all Dropout masks of a step in one call (a captured step's graph inputs) ------------------------------------------
// The masks of a step are drawn one after the other into pinned memory as BITS (0.5 M elements = 64 KB) and become the float
// arrays the captured kernels read with ONE launch in front of the graph: the kernel fetches the bits over PCIe itself
// (a byte per thread, eight floats out).  Measured on the CNN workload (0.391 ms per step without any upload): a kernel per
// mask fetching 0 / 1 BYTES stood 23 us in front of every replay and ONE such kernel for both masks as long (the fetch of
// 0.5 MB in 4-byte requests is what takes the time, not the launch); the same bytes sent by the copy engine on a second
// stream beside the running step and expanded from device memory cost 44 us (the hop to the copy queue and back).
namespace {
constexpr int kMaskBatch = 8;
struct MaskTable {
  const uint8_t* src[kMaskBatch];
  float* dst[kMaskBatch];
  int64_t n[kMaskBatch];
};
struct MaskStage {
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t expanded = nullptr;  // the library stream has passed the kernel that read `pinned`
  bool pending = false;
};
MaskStage g_mask_stage[4];
int g_mask_stage_next = 0;
}  // namespace

__global__ __launch_bounds__(256) void k_masks_expand(MaskTable t) {
  const uint8_t* __restrict__ src = t.src[blockIdx.y];
  float* __restrict__ dst = t.dst[blockIdx.y];
  const int64_t n = t.n[blockIdx.y], n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const unsigned b = src[i];
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4((float)(b & 1), (float)((b >> 1) & 1), (float)((b >> 2) & 1), (float)((b >> 3) & 1));
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4((float)((b >> 4) & 1), (float)((b >> 5) & 1), (float)((b >> 6) & 1), (float)((b >> 7) & 1));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = (float)((src[n8] >> threadIdx.x) & 1);
}

int dfhip_upload_bernoulli_masks(uint32_t* key, int* pos, int count, const double* keep, const int64_t* n, float* const* dst) {
  DF_INIT();
  DF_CHECK_ARG(key && pos && count >= 0 && (count == 0 || (keep && n && dst)), "upload_bernoulli_masks: bad arguments");
  if (g_capture_open) {
    set_error("upload_bernoulli_masks inside a graph capture: register the masks as graph inputs instead");
    return DFHIP_RUNTIME_ERROR;
  }
  DF_CHECK_ARG(*pos >= 0 && *pos <= 624, "upload_bernoulli_masks: bad generator position");
  bool aligned = true;
  for (int i = 0; i < count; i++) {
    DF_CHECK_ARG(n[i] >= 0 && (n[i] == 0 || dst[i]) && keep[i] >= 0.0 && keep[i] <= 1.0, "upload_bernoulli_masks: bad mask");
    if ((((uintptr_t)dst[i]) & 15) != 0) aligned = false;
  }
  if (!aligned) {      // (one mask at a time, the same stream order: the unaligned ones as floats through a plain copy)
    for (int i = 0; i < count; i++) {
      const int st = dfhip_upload_bernoulli_mask(key, pos, keep[i], n[i], dst[i]);
      if (st != DFHIP_OK) return st;
    }
    return DFHIP_OK;
  }
  for (int i0 = 0; i0 < count; i0 += kMaskBatch) {
    const int cnt = std::min(kMaskBatch, count - i0);
    size_t off[kMaskBatch], bits = 0;
    int64_t longest = 0;
    for (int i = 0; i < cnt; i++) {
      off[i] = bits;
      bits += (((size_t)n[i0 + i] + 7) / 8 + 63) & ~(size_t)63;
      longest = std::max(longest, n[i0 + i]);
    }
    if (longest == 0) continue;
    const size_t total = bits + (size_t)longest;        // the bits of every mask, then the bytes of the one being drawn
    MaskStage& st = g_mask_stage[g_mask_stage_next];
    g_mask_stage_next = (g_mask_stage_next + 1) & 3;
    if (!st.expanded) DF_HIP(hipEventCreateWithFlags(&st.expanded, hipEventDisableTiming));
    if (st.pending) {                                   // the pinned buffer is free again
      DF_HIP(hipEventSynchronize(st.expanded));
      st.pending = false;
    }
    if (st.pinned_bytes < total) {
      if (st.pinned) (void)hipHostFree(st.pinned);
      st.pinned = nullptr;
      st.pinned_bytes = 0;
      const size_t want = std::max(total, (size_t)1 << 20);
      DF_HIP(hipHostMalloc(&st.pinned, want, hipHostMallocDefault));
      st.pinned_bytes = want;
    }
    MaskTable t;
    for (int i = 0; i < kMaskBatch; i++) { t.src[i] = nullptr; t.dst[i] = nullptr; t.n[i] = 0; }
    for (int i = 0; i < cnt; i++) {
      uint8_t* at = (uint8_t*)st.pinned + off[i];
      const int s = host_bernoulli_mask_bits(key, pos, keep[i0 + i], n[i0 + i], (uint8_t*)st.pinned + bits, at);
      if (s != DFHIP_OK) return s;
      t.src[i] = at;
      t.dst[i] = dst[i0 + i];
      t.n[i] = n[i0 + i];
    }
    const int blocks = (int)std::min<int64_t>(512, (longest / 8 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_masks_expand, dim3(blocks, cnt), dim3(256), 0, g_stream, t);
    DF_LAUNCH_CHECK();
    DF_HIP(hipEventRecord(st.expanded, g_stream));
    st.pending = true;
  }
  return DFHIP_OK;
}

int dfhip_d2d(const float* src, float* dst, int64_t n) {
  DF_INIT();
  if (n == 0) return DFHIP_OK;
  DF_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, g_stream));
  return DFHIP_OK;
}

// ---- hipGraph capture ------------------------------------------------------------------
struct GraphExec {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int capture_id;
};

// the blocks a capture owned go back to the general pool (graph destroyed, or the capture failed)
static void release_capture_blocks(int capture_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_blocks)
    if (kv.second.owner == capture_id) kv.second.owner = 0;
  auto it = g_cap_free.find(capture_id);
  if (it != g_cap_free.end()) {
    for (auto& kv : it->second)
      for (void* p : kv.second) g_free[kv.first].push_back(p);
    g_cap_free.erase(it);
  }
}

int dfhip_graph_begin(void) {
  DF_INIT();
  DF_CHECK_ARG(!g_capture_open, "a capture is already open");
  DF_HIP(hipStreamSynchronize(g_stream));
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_capture_open = ++g_capture_seq;
  }
  hipError_t e = hipStreamBeginCapture(g_stream, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    g_capture_open = 0;
    set_error("hipStreamBeginCapture failed: %s", hipGetErrorString(e));
    return DFHIP_RUNTIME_ERROR;
  }
  return DFHIP_OK;
}

int dfhip_graph_end(void** graph_exec) {
  DF_CHECK_ARG(g_capture_open, "no capture open");
  // the capture must be closed even when it has been invalidated (an illegal call inside it, e.g. a
  // blocking copy): otherwise every later launch on the stream fails
  const int f__ = flush_deferred();
  int id = g_capture_open;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(g_stream, &graph);
  g_capture_open = 0;
  (void)hipGetLastError();                       // clear the sticky capture error
  if (e != hipSuccess || !graph || f__ != DFHIP_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
      // an invalidated capture leaves the stream unusable: nothing was executed since
      // dfhip_graph_begin synchronised it, so it is simply replaced
      (void)hipStreamDestroy(g_stream);
      g_stream = nullptr;
      (void)hipGetLastError();
      if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("capture failed and the library stream could not be re-created");
        return DFHIP_RUNTIME_ERROR;
      }
    }
    release_capture_blocks(id);                   // no GraphExec will ever own them
    set_error("capture failed: %s", e != hipSuccess ? hipGetErrorString(e) : "a call inside the capture failed");
    return DFHIP_RUNTIME_ERROR;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    release_capture_blocks(id);
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return DFHIP_RUNTIME_ERROR;
  }
  GraphExec* g = new GraphExec{graph, exec, id};
  *graph_exec = g;
  return DFHIP_OK;
}

int dfhip_graph_launch(void* graph_exec) {
  DF_CHECK_ARG(graph_exec, "null graph");
  { int f__ = flush_deferred(); if (f__ != DFHIP_OK) return f__; }
  GraphExec* g = (GraphExec*)graph_exec;
  DF_HIP(hipGraphLaunch(g->exec, g_stream));
  return DFHIP_OK;
}

int dfhip_graph_destroy(void* graph_exec) {
  if (!graph_exec) return DFHIP_OK;
  GraphExec* g = (GraphExec*)graph_exec;
  (void)hipStreamSynchronize(g_stream);
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  release_capture_blocks(g->capture_id);
  delete g;
  return DFHIP_OK;
}

// ---- events / profiling ----------------------------------------------------------------
int dfhip_event_create(void** ev) {
  DF_INIT();
  hipEvent_t e;
  DF_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return DFHIP_OK;
}
int dfhip_event_record(void* ev) {
  { int f__ = flush_deferred(); if (f__ != DFHIP_OK) return f__; }
  DF_HIP(hipEventRecord((hipEvent_t)ev, g_stream));
  return DFHIP_OK;
}
int dfhip_event_elapsed_ms(void* start, void* stop, float* ms) {
  DF_HIP(hipEventSynchronize((hipEvent_t)stop));
  DF_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DFHIP_OK;
}
int dfhip_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return DFHIP_OK;
}

int dfhip_prof_enable(int on) {
  g_prof_on = on != 0;
  return DFHIP_OK;
}
// A profiled eager step is issued through Python: the host needs about as long per launch as the kernels run, so
// the queue runs dry and every event pair also measures the host's gap between the two records.  The gate holds the
// stream (one thread polling a word of pinned host memory) while the host queues the whole step, events included;
// opened, the launches run back to back as they do inside a replayed graph.  It gives up after `timeout_ms` (a
// blocking call issued behind a closed gate must not hang the process).
static int* g_gate_flag = nullptr;
static int g_gate_gen = 0;
__global__ void k_prof_gate(const int* flag, int gen, unsigned long long max_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < gen) {
    if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) break;
    __builtin_amdgcn_s_sleep(64);
  }
}
int dfhip_prof_gate_close(int timeout_ms) {
  if (g_capture_open) return DFHIP_OK;
  if (!g_gate_flag) {
    DF_HIP(hipHostMalloc((void**)&g_gate_flag, sizeof(int), hipHostMallocMapped));
    *g_gate_flag = 0;
  }
  (void)flush_deferred();
  if (timeout_ms <= 0 || timeout_ms > 200) timeout_ms = 200;
  g_gate_gen++;
  hipLaunchKernelGGL(k_prof_gate, dim3(1), dim3(1), 0, g_stream, (const int*)g_gate_flag, g_gate_gen,
                     (unsigned long long)timeout_ms * 100000ull);
  DF_LAUNCH_CHECK();
  return DFHIP_OK;
}
int dfhip_prof_gate_open(void) {
  if (g_gate_flag) __atomic_store_n(g_gate_flag, g_gate_gen, __ATOMIC_RELEASE);
  return DFHIP_OK;
}
int dfhip_prof_reset(void) {
  for (auto& s : g_prof) {
    prof_resolve(s);
    s.ms = s.flops = s.bytes = s.kernel_ms = 0;
    s.calls = s.kernel_launches = 0;
  }
  for (auto& k : g_krec) { k.ms = k.flops = k.bytes = 0; k.launches = 0; }
  return DFHIP_OK;
}
// per kernel instantiation: name, family, launches, summed duration of the kernels' own start / end stamps, and the
// algorithmic flops / bytes of the family calls whose first launch it was
int dfhip_prof_read_kernels(int cap, const char** names, const char** families, double* kernel_ms, int64_t* launches,
                            double* flops, double* bytes) {
  (void)flush_deferred();
  for (auto& s : g_prof) prof_resolve(s);
  int n = 0;
  for (auto& k : g_krec) {
    if (n >= cap) break;
    if (k.launches == 0) continue;
    names[n] = k.name.c_str();
    families[n] = g_prof[k.slot].name;
    kernel_ms[n] = k.ms;
    launches[n] = k.launches;
    flops[n] = k.flops;
    bytes[n] = k.bytes;
    n++;
  }
  return n;
}
int dfhip_prof_read(int cap, const char** names, double* total_ms, int64_t* calls,
                    double* flops, double* bytes, double* kernel_ms, int64_t* kernel_launches) {
  (void)flush_deferred();
  int n = 0;
  for (auto& s : g_prof) {
    prof_resolve(s);
    if (n >= cap) break;
    if (s.calls == 0) continue;
    names[n] = s.name;
    total_ms[n] = s.ms;
    calls[n] = s.calls;
    flops[n] = s.flops;
    bytes[n] = s.bytes;
    if (kernel_ms) kernel_ms[n] = s.kernel_ms;
    if (kernel_launches) kernel_launches[n] = s.kernel_launches;
    n++;
  }
  return n;
}

// ---- RCCL (loaded lazily: a single-GPU process never pays for librccl) -----------------
typedef struct { char internal[128]; } df_nccl_uid;
typedef void* df_nccl_comm;
static void* g_rccl = nullptr;
static int (*p_ncclGetUniqueId)(df_nccl_uid*) = nullptr;
static int (*p_ncclCommInitRank)(df_nccl_comm*, int, df_nccl_uid, int) = nullptr;
static int (*p_ncclAllReduce)(const void*, void*, size_t, int, int, df_nccl_comm, hipStream_t) = nullptr;
static int (*p_ncclCommDestroy)(df_nccl_comm) = nullptr;
static const char* (*p_ncclGetErrorString)(int) = nullptr;
static int (*p_ncclCommCount)(df_nccl_comm, int*) = nullptr;
static int (*p_ncclCommUserRank)(df_nccl_comm, int*) = nullptr;
static int (*p_ncclCommGetAsyncError)(df_nccl_comm, int*) = nullptr;
static df_nccl_comm g_comm = nullptr;
static int g_world = 1;

static int load_rccl() {
  if (g_rccl) return DFHIP_OK;
  // ROCm's own RCCL first (by path), whatever the process has mapped under the soname only after that: a launcher
  // that imported a framework with a bundled librccl must not decide which collective library the step uses.
  // DFHIP_RCCL_PATH overrides.
  const char* names[] = {getenv("DFHIP_RCCL_PATH") ? getenv("DFHIP_RCCL_PATH") : "/opt/rocm/lib/librccl.so.1",
                         "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    g_rccl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl) break;
  }
  if (!g_rccl) {
    set_error("cannot load librccl: %s", dlerror());
    return DFHIP_RUNTIME_ERROR;
  }
  p_ncclGetUniqueId = (decltype(p_ncclGetUniqueId))dlsym(g_rccl, "ncclGetUniqueId");
  p_ncclCommInitRank = (decltype(p_ncclCommInitRank))dlsym(g_rccl, "ncclCommInitRank");
  p_ncclAllReduce = (decltype(p_ncclAllReduce))dlsym(g_rccl, "ncclAllReduce");
  p_ncclCommDestroy = (decltype(p_ncclCommDestroy))dlsym(g_rccl, "ncclCommDestroy");
  p_ncclGetErrorString = (decltype(p_ncclGetErrorString))dlsym(g_rccl, "ncclGetErrorString");
  p_ncclCommCount = (decltype(p_ncclCommCount))dlsym(g_rccl, "ncclCommCount");
  p_ncclCommUserRank = (decltype(p_ncclCommUserRank))dlsym(g_rccl, "ncclCommUserRank");
  p_ncclCommGetAsyncError = (decltype(p_ncclCommGetAsyncError))dlsym(g_rccl, "ncclCommGetAsyncError");
  if (!p_ncclGetUniqueId || !p_ncclCommInitRank || !p_ncclAllReduce || !p_ncclCommDestroy) {
    set_error("librccl is missing a required symbol");
    return DFHIP_RUNTIME_ERROR;
  }
  return DFHIP_OK;
}
#define DF_NCCL(call)                                                                   \
  do {                                                                                  \
    int r__ = (call);                                                                   \
    if (r__ != 0) {                                                                     \
      set_error("%s failed: %s", #call, p_ncclGetErrorString ? p_ncclGetErrorString(r__) : "?"); \
      return DFHIP_RUNTIME_ERROR;                                                       \
    }                                                                                   \
  } while (0)

int dfhip_comm_unique_id(char* out128) {
  int s = load_rccl();
  if (s != DFHIP_OK) return s;
  df_nccl_uid id;
  DF_NCCL(p_ncclGetUniqueId(&id));
  memcpy(out128, id.internal, 128);
  return DFHIP_OK;
}

int dfhip_comm_init(int rank, int world, const char* id128) {
  DF_INIT();
  DF_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank/world %d/%d", rank, world);
  DF_CHECK_ARG(!g_comm, "communicator already initialised");
  int s = load_rccl();
  if (s != DFHIP_OK) return s;
  df_nccl_uid id;
  memcpy(id.internal, id128, 128);
  DF_NCCL(p_ncclCommInitRank(&g_comm, world, id, rank));
  g_world = world;
  return DFHIP_OK;
}

// A peer that died or a link that failed does not fail the collective call that was enqueued before: it shows up as the
// communicator's ASYNCHRONOUS error (ncclCommGetAsyncError) while the local stream simply never finishes.  Every host-side
// wait of the exchange path polls it, so that a broken job ends with an error (the launcher then reaps the other ranks by
// PID) instead of hanging in hipStreamSynchronize.  ncclSuccess = 0, ncclInProgress = 7.
static int comm_async_check(const char* where) {
  if (!g_comm || !p_ncclCommGetAsyncError) return DFHIP_OK;
  int st = 0;
  const int r = p_ncclCommGetAsyncError(g_comm, &st);
  if (r != 0 || (st != 0 && st != 7)) {
    const int bad = r != 0 ? r : st;
    set_error("RCCL reports an asynchronous error (%s) at %s: the job cannot continue", p_ncclGetErrorString ? p_ncclGetErrorString(bad) : "?",
              where);
    return DFHIP_RUNTIME_ERROR;
  }
  return DFHIP_OK;
}

// hipStreamSynchronize with the communicator polled meanwhile; DFHIP_COMM_TIMEOUT_S (default 300) bounds the wait
static int comm_sync_stream(hipStream_t s, const char* where) {
  static double limit = getenv("DFHIP_COMM_TIMEOUT_S") ? atof(getenv("DFHIP_COMM_TIMEOUT_S")) : 300.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; spins++) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return comm_async_check(where);
    if (q != hipErrorNotReady) {
      set_error("%s: the stream failed: %s", where, hipGetErrorString(q));
      return DFHIP_RUNTIME_ERROR;
    }
    if ((spins & 255) == 255) {
      const int st = comm_async_check(where);
      if (st != DFHIP_OK) return st;
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited > limit) {
        set_error("%s: no completion after %.0f s (DFHIP_COMM_TIMEOUT_S): a rank of the job is gone or stuck", where, waited);
        return DFHIP_RUNTIME_ERROR;
      }
      if (waited > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
}

int dfhip_comm_check(void) {
  DF_INIT_KEEP();
  return comm_async_check("dfhip_comm_check");
}

int dfhip_allreduce_sum(float* buf, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(g_comm, "dfhip_comm_init has not been called");
  if (n == 0) return DFHIP_OK;
  ProfScope prof("rccl_allreduce", 0.0, 2.0 * (double)n * 4.0);
  // ncclFloat32 = 7, ncclSum = 0
  DF_NCCL(p_ncclAllReduce(buf, buf, (size_t)n, 7, 0, g_comm, g_stream));
  return DFHIP_OK;
}

// Overlapped exchange: the all-reduce is enqueued on a communication stream that waits for
// everything already queued on the library stream (the packed gradients) and runs beside the rest
// of the backward pass; dfhip_comm_wait makes the library stream wait for every all-reduce issued
// this way (before the optimizer reads the bucket).
static hipStream_t g_comm_stream = nullptr;
static hipEvent_t g_comm_ready = nullptr, g_comm_done = nullptr;
static bool g_comm_pending = false;

int dfhip_allreduce_sum_async(float* buf, int64_t n) {
  DF_INIT();
  DF_CHECK_ARG(g_comm, "dfhip_comm_init has not been called");
  if (n == 0) return DFHIP_OK;
  if (!g_comm_stream) {
    DF_HIP(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
    DF_HIP(hipEventCreateWithFlags(&g_comm_ready, hipEventDisableTiming));
    DF_HIP(hipEventCreateWithFlags(&g_comm_done, hipEventDisableTiming));
  }
  DF_HIP(hipEventRecord(g_comm_ready, g_stream));
  DF_HIP(hipStreamWaitEvent(g_comm_stream, g_comm_ready, 0));
  DF_NCCL(p_ncclAllReduce(buf, buf, (size_t)n, 7, 0, g_comm, g_comm_stream));
  DF_HIP(hipEventRecord(g_comm_done, g_comm_stream));
  g_comm_pending = true;
  return DFHIP_OK;
}

int dfhip_comm_wait(void) {
  DF_INIT();
  if (g_comm_pending) {
    DF_HIP(hipStreamWaitEvent(g_stream, g_comm_done, 0));
    g_comm_pending = false;
  }
  return comm_async_check("dfhip_comm_wait");
}

// max over ranks of one host float (step time of the slowest rank) / barrier: a one-element all-reduce on the
// library stream followed by a stream synchronisation -- no host-side process group anywhere
int dfhip_comm_allreduce_max(float* value) {
  DF_INIT();
  DF_CHECK_ARG(g_comm && value, "comm_allreduce_max: no communicator / null value");
  static float* d = nullptr;
  if (!d) DF_HIP(hipMalloc((void**)&d, sizeof(float)));
  DF_HIP(hipMemcpyAsync(d, value, sizeof(float), hipMemcpyHostToDevice, g_stream));
  // ncclFloat32 = 7, ncclMax = 2
  DF_NCCL(p_ncclAllReduce(d, d, 1, 7, 2, g_comm, g_stream));
  DF_HIP(hipMemcpyAsync(value, d, sizeof(float), hipMemcpyDeviceToHost, g_stream));
  return comm_sync_stream(g_stream, "dfhip_comm_allreduce_max (barrier)");
}

int dfhip_comm_count(int* ranks, int* rank) {
  // what the communicator itself says (ncclCommCount / ncclCommUserRank), not what the launcher's environment said
  DF_CHECK_ARG(g_comm && ranks && rank, "comm_count: no communicator / null output");
  DF_CHECK_ARG(p_ncclCommCount && p_ncclCommUserRank, "librccl has no ncclCommCount / ncclCommUserRank");
  DF_NCCL(p_ncclCommCount(g_comm, ranks));
  DF_NCCL(p_ncclCommUserRank(g_comm, rank));
  return DFHIP_OK;
}

int dfhip_comm_rccl_path(char* out, int cap) {
  // which shared object the collective symbols came from (diagnostic: bench.py reports it)
  if (!out || cap <= 0) return DFHIP_INVALID_ARGUMENT;
  out[0] = 0;
  Dl_info info;
  if (p_ncclAllReduce && dladdr((void*)p_ncclAllReduce, &info) && info.dli_fname) snprintf(out, (size_t)cap, "%s", info.dli_fname);
  return DFHIP_OK;
}

int dfhip_comm_destroy(void) {
  (void)flush_deferred();
  if (g_comm) {
    if (g_comm_stream) (void)hipStreamSynchronize(g_comm_stream);
    (void)hipStreamSynchronize(g_stream);
    p_ncclCommDestroy(g_comm);
    g_comm = nullptr;
  }
  return DFHIP_OK;
}

}  // extern "C"
