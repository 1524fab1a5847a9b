// Dropout masks: NumPy's global generator, continued in C (host code only: no device call in this file).
//
// The reference draws a mask with np.random.binomial(1, keep, shape) on the host (nn/modules/dropout.py:16-18).  For
// n = 1 NumPy's sampler consumes ONE double of the legacy MT19937 stream per element and thresholds it, and a double is
// ((a >> 5) * 2^26 + (b >> 6)) / 2^53 of two consecutive 32-bit outputs a, b: the mask is a pure function of the generator
// state.  host_bernoulli_mask continues that stream from a copy of the state (key[624], pos) the caller took with
// np.random.get_state() and hands the advanced state back -- same masks, same position of the global generator afterwards.
//
// At 0.5 M elements per step the CNN workload's replayed step is bound by exactly this loop (round 6: 0.40 ms of host
// work beside 0.39 ms of GPU work), so the three passes over a block of 624 words are written with intrinsics at the
// widest vector the host has (AVX-512 / AVX2 / plain C, picked at run time; dfhip_host_rng_isa forces one for the tests):
//   regenerate  the twist of the 624 state words: three runs without a dependency inside a vector (the distance to the
//               word a step reads back is 227)
//   temper      the four xor-shifts of every word
//   threshold   k = (a >> 5) * 2^26 + (b >> 6) is an integer below 2^53 and u = k / 2^53 is exact, so  u > thr  is the
//               INTEGER comparison  k > floor(thr * 2^53)  (thr * 2^53 is exact too): a pair of words is one 64-bit
//               lane, k two shifts and a mask of it -- no conversion to double at all
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "dfhip_internal.h"

using namespace dfhip;

namespace {
constexpr int kMtN = 624, kMtM = 397;
constexpr uint32_t kUp = 0x80000000u, kLo = 0x7fffffffu, kMa = 0x9908b0dfu;

inline uint32_t twist(uint32_t a, uint32_t b, uint32_t far) {
  const uint32_t y = (a & kUp) | (b & kLo);
  return far ^ (y >> 1) ^ ((0u - (y & 1u)) & kMa);
}
inline uint32_t temper1(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
// one element of the mask from the two words of its double; `flip` = 1 when NumPy's sampler mirrors the test (keep > 0.5)
inline unsigned bit1(uint32_t a, uint32_t b, uint64_t T, unsigned flip) {
  const uint64_t k = ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
  return (unsigned)(k > T) ^ flip;
}

// ---- plain C -------------------------------------------------------------------------------------------------------
void regen_base(uint32_t* mt) {
  int i = 0;
  for (; i < kMtN - kMtM; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + kMtM]);
  for (; i < kMtN - 1; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + (kMtM - kMtN)]);
  mt[kMtN - 1] = twist(mt[kMtN - 1], mt[0], mt[kMtM - 1]);
}
void temper_base(const uint32_t* src, uint32_t* dst, int n) {
  for (int i = 0; i < n; i++) dst[i] = temper1(src[i]);
}
template <typename OUT>
void pairs_base(const uint32_t* q, int64_t pairs, uint64_t T, unsigned flip, OUT* o) {
  for (int64_t i = 0; i < pairs; i++) o[i] = (OUT)bit1(q[2 * i], q[2 * i + 1], T, flip);
}

// ---- AVX2 ----------------------------------------------------------------------------------------------------------
#define DF_AVX2 __attribute__((target("avx2")))
DF_AVX2 inline __m256i twist8(const uint32_t* a, const uint32_t* far) {
  const __m256i up = _mm256_set1_epi32((int)kUp), lo = _mm256_set1_epi32((int)kLo), ma = _mm256_set1_epi32((int)kMa);
  const __m256i y = _mm256_or_si256(_mm256_and_si256(_mm256_loadu_si256((const __m256i*)a), up),
                                    _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(a + 1)), lo));
  const __m256i odd = _mm256_srai_epi32(_mm256_slli_epi32(y, 31), 31);
  return _mm256_xor_si256(_mm256_xor_si256(_mm256_loadu_si256((const __m256i*)far), _mm256_srli_epi32(y, 1)),
                          _mm256_and_si256(odd, ma));
}
DF_AVX2 void regen_avx2(uint32_t* mt) {
  int i = 0;
  for (; i + 8 <= kMtN - kMtM; i += 8) _mm256_storeu_si256((__m256i*)(mt + i), twist8(mt + i, mt + i + kMtM));
  for (; i < kMtN - kMtM; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + kMtM]);
  for (; i + 8 <= kMtN - 1; i += 8) _mm256_storeu_si256((__m256i*)(mt + i), twist8(mt + i, mt + i + (kMtM - kMtN)));
  for (; i < kMtN - 1; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + (kMtM - kMtN)]);
  mt[kMtN - 1] = twist(mt[kMtN - 1], mt[0], mt[kMtM - 1]);
}
DF_AVX2 void temper_avx2(const uint32_t* src, uint32_t* dst, int n) {
  const __m256i c1 = _mm256_set1_epi32((int)0x9d2c5680u), c2 = _mm256_set1_epi32((int)0xefc60000u);
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    __m256i y = _mm256_loadu_si256((const __m256i*)(src + i));
    y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
    y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), c1));
    y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), c2));
    y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
    _mm256_storeu_si256((__m256i*)(dst + i), y);
  }
  for (; i < n; i++) dst[i] = temper1(src[i]);
}
// four pairs = four 64-bit lanes (little endian: a in the low word): k = ((v & 0xffffffe0) << 21) | (v >> 38)
DF_AVX2 inline int gt4(const uint32_t* q, __m256i T) {
  const __m256i v = _mm256_loadu_si256((const __m256i*)q);
  const __m256i k = _mm256_or_si256(_mm256_slli_epi64(_mm256_and_si256(v, _mm256_set1_epi64x(0xffffffe0ll)), 21), _mm256_srli_epi64(v, 38));
  return _mm256_movemask_pd(_mm256_castsi256_pd(_mm256_cmpgt_epi64(k, T)));
}
struct Lut4 {
  uint32_t bytes[16];
  float floats[16][4];
  constexpr Lut4() : bytes(), floats() {
    for (int m = 0; m < 16; m++) {
      bytes[m] = (uint32_t)(m & 1) | (uint32_t)((m >> 1) & 1) << 8 | (uint32_t)((m >> 2) & 1) << 16 | (uint32_t)((m >> 3) & 1) << 24;
      for (int j = 0; j < 4; j++) floats[m][j] = (float)((m >> j) & 1);
    }
  }
};
constexpr Lut4 kLut4;
DF_AVX2 void pairs_avx2_u8(const uint32_t* q, int64_t pairs, uint64_t T, unsigned flip, uint8_t* o) {
  const __m256i Tv = _mm256_set1_epi64x((long long)T);
  const int fl = flip ? 15 : 0;
  int64_t i = 0;
  for (; i + 8 <= pairs; i += 8) {
    const uint32_t lo = kLut4.bytes[gt4(q + 2 * i, Tv) ^ fl], hi = kLut4.bytes[gt4(q + 2 * i + 8, Tv) ^ fl];
    const uint64_t both = (uint64_t)lo | (uint64_t)hi << 32;
    memcpy(o + i, &both, 8);
  }
  for (; i < pairs; i++) o[i] = (uint8_t)bit1(q[2 * i], q[2 * i + 1], T, flip);
}
DF_AVX2 void pairs_avx2_f32(const uint32_t* q, int64_t pairs, uint64_t T, unsigned flip, float* o) {
  const __m256i Tv = _mm256_set1_epi64x((long long)T);
  const int fl = flip ? 15 : 0;
  int64_t i = 0;
  for (; i + 4 <= pairs; i += 4) _mm_storeu_ps(o + i, _mm_loadu_ps(kLut4.floats[gt4(q + 2 * i, Tv) ^ fl]));
  for (; i < pairs; i++) o[i] = (float)bit1(q[2 * i], q[2 * i + 1], T, flip);
}

// ---- AVX-512 (F + BW + VL) -----------------------------------------------------------------------------------------
#define DF_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl")))
DF_AVX512 inline __m512i twist16(const uint32_t* a, const uint32_t* far) {
  const __m512i va = _mm512_loadu_si512(a), vb = _mm512_loadu_si512(a + 1);
  // y = (a & UP) | (b & LO): bitwise select with the constant LO (truth table 0xCA: c ? b : a with c = LO -> imm for (a, b, c))
  const __m512i y = _mm512_ternarylogic_epi32(_mm512_set1_epi32((int)kLo), vb, va, 0xCA);
  const __mmask16 odd = _mm512_test_epi32_mask(y, _mm512_set1_epi32(1));
  const __m512i t = _mm512_xor_si512(_mm512_loadu_si512(far), _mm512_srli_epi32(y, 1));
  return _mm512_mask_xor_epi32(t, odd, t, _mm512_set1_epi32((int)kMa));
}
DF_AVX512 void regen_avx512(uint32_t* mt) {
  int i = 0;
  for (; i + 16 <= kMtN - kMtM; i += 16) _mm512_storeu_si512(mt + i, twist16(mt + i, mt + i + kMtM));
  for (; i < kMtN - kMtM; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + kMtM]);
  for (; i + 16 <= kMtN - 1; i += 16) _mm512_storeu_si512(mt + i, twist16(mt + i, mt + i + (kMtM - kMtN)));
  for (; i < kMtN - 1; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + (kMtM - kMtN)]);
  mt[kMtN - 1] = twist(mt[kMtN - 1], mt[0], mt[kMtM - 1]);
}
DF_AVX512 inline __m512i temper16(__m512i y) {
  const __m512i c1 = _mm512_set1_epi32((int)0x9d2c5680u), c2 = _mm512_set1_epi32((int)0xefc60000u);
  y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
  y = _mm512_ternarylogic_epi32(y, _mm512_slli_epi32(y, 7), c1, 0x78);     // y ^ (s & c)
  y = _mm512_ternarylogic_epi32(y, _mm512_slli_epi32(y, 15), c2, 0x78);
  return _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
}
DF_AVX512 void temper_avx512(const uint32_t* src, uint32_t* dst, int n) {
  int i = 0;
  for (; i + 16 <= n; i += 16) _mm512_storeu_si512(dst + i, temper16(_mm512_loadu_si512(src + i)));
  for (; i < n; i++) dst[i] = temper1(src[i]);
}
DF_AVX512 inline __mmask8 gt8(const uint32_t* q, __m512i T) {
  const __m512i v = _mm512_loadu_si512(q);
  // ((v << 21) & (0xffffffe0 << 21)) | (v >> 38): truth table (a & c) | b = 0xEC for (a = v << 21, b = v >> 38, c = mask)
  const __m512i k = _mm512_ternarylogic_epi64(_mm512_slli_epi64(v, 21), _mm512_srli_epi64(v, 38),
                                              _mm512_set1_epi64((long long)(0xffffffe0ull << 21)), 0xEC);
  return _mm512_cmpgt_epi64_mask(k, T);
}
DF_AVX512 void pairs_avx512_u8(const uint32_t* q, int64_t pairs, uint64_t T, unsigned flip, uint8_t* o) {
  const __m512i Tv = _mm512_set1_epi64((long long)T);
  const uint64_t fl = flip ? ~0ull : 0ull;
  int64_t i = 0;
  for (; i + 64 <= pairs; i += 64) {
    uint64_t m = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) m |= (uint64_t)gt8(q + 2 * i + 16 * j, Tv) << (8 * j);
    _mm512_storeu_si512(o + i, _mm512_maskz_set1_epi8((__mmask64)(m ^ fl), 1));
  }
  for (; i + 8 <= pairs; i += 8)
    _mm_storel_epi64((__m128i*)(o + i), _mm_maskz_set1_epi8((__mmask16)((gt8(q + 2 * i, Tv) ^ (unsigned)fl) & 0xff), 1));
  for (; i < pairs; i++) o[i] = (uint8_t)bit1(q[2 * i], q[2 * i + 1], T, flip);
}
DF_AVX512 void pairs_avx512_f32(const uint32_t* q, int64_t pairs, uint64_t T, unsigned flip, float* o) {
  const __m512i Tv = _mm512_set1_epi64((long long)T);
  const unsigned fl = flip ? 0xffffu : 0u;
  const __m512 one = _mm512_set1_ps(1.0f);
  int64_t i = 0;
  for (; i + 16 <= pairs; i += 16) {
    const unsigned m = (unsigned)gt8(q + 2 * i, Tv) | (unsigned)gt8(q + 2 * i + 16, Tv) << 8;
    _mm512_storeu_ps(o + i, _mm512_maskz_mov_ps((__mmask16)(m ^ fl), one));
  }
  for (; i < pairs; i++) o[i] = (float)bit1(q[2 * i], q[2 * i + 1], T, flip);
}

// ---- the vector width in use ---------------------------------------------------------------------------------------
struct RngOps {
  void (*regen)(uint32_t*);
  void (*temper)(const uint32_t*, uint32_t*, int);
  void (*pairs_u8)(const uint32_t*, int64_t, uint64_t, unsigned, uint8_t*);
  void (*pairs_f32)(const uint32_t*, int64_t, uint64_t, unsigned, float*);
};
const RngOps kOps[3] = {{regen_base, temper_base, pairs_base<uint8_t>, pairs_base<float>},
                        {regen_avx2, temper_avx2, pairs_avx2_u8, pairs_avx2_f32},
                        {regen_avx512, temper_avx512, pairs_avx512_u8, pairs_avx512_f32}};
int best_isa() {
  static const int best = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl")) ? 2
                          : __builtin_cpu_supports("avx2") ? 1 : 0;
  return best;
}
int g_isa = -1;                                   // -1: the widest the host has
inline const RngOps& ops() { return kOps[g_isa < 0 ? best_isa() : g_isa]; }
inline void call_pairs(const RngOps& r, const uint32_t* q, int64_t p, uint64_t T, unsigned f, uint8_t* o) { r.pairs_u8(q, p, T, f, o); }
inline void call_pairs(const RngOps& r, const uint32_t* q, int64_t p, uint64_t T, unsigned f, float* o) { r.pairs_f32(q, p, T, f, o); }

template <typename OUT>
int mask_from_stream(uint32_t* key, int* pos, double keep, int64_t n, OUT* out) {
  const RngOps& r = ops();
  // tempered outputs are produced a block at a time into `buf`; `have` of them are not consumed yet
  alignas(64) uint32_t buf[kMtN + 16];
  int p = *pos, have = 0, at = 0;
  const bool low = keep <= 0.5;                     // NumPy's inversion sampler mirrors the test for keep > 0.5
  const double thr = low ? 1.0 - keep : keep;
  const uint64_t T = (uint64_t)(thr * 9007199254740992.0);     // floor(thr * 2^53): the product is exact
  const unsigned flip = low ? 0u : 1u;
  uint32_t carry_a = 0;
  bool has_carry = false;
  int64_t done = 0;
  while (done < n) {
    if (have == 0) {
      if (p == kMtN) { r.regen(key); p = 0; }
      have = kMtN - p;
      r.temper(key + p, buf, have);
      p = kMtN;
      at = 0;
    }
    if (has_carry) {                                // a double straddling two blocks
      out[done++] = (OUT)bit1(carry_a, buf[at], T, flip);
      at++; have--;
      has_carry = false;
      continue;
    }
    int64_t pairs = have / 2;
    if (pairs > n - done) pairs = n - done;
    call_pairs(r, buf + at, pairs, T, flip, out + done);
    done += pairs;
    at += (int)(2 * pairs);
    have -= (int)(2 * pairs);
    if (done < n && have == 1) {                    // odd word left in the block: first half of the next double
      carry_a = buf[at];
      has_carry = true;
      have = 0;
    }
  }
  // words of the last block that were tempered but not consumed go back: the generator stands right behind the last
  // word used
  *pos = p - have;
  return DFHIP_OK;
}
// ---- 0 / 1 bytes -> bits (element i = bit i & 7 of byte i >> 3): what crosses PCIe is an eighth of the bytes ----------
void pack_base(const uint8_t* b, int64_t n, uint8_t* bits) {
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t v;
    memcpy(&v, b + i, 8);
    bits[i >> 3] = (uint8_t)((v * 0x0102040810204080ull) >> 56);      // byte j (0 / 1) lands on bit 56 + j
  }
  if (i < n) {
    unsigned m = 0;
    for (int j = 0; i + j < n; j++) m |= (unsigned)(b[i + j] & 1) << j;
    bits[i >> 3] = (uint8_t)m;
  }
}
DF_AVX2 void pack_avx2(const uint8_t* b, int64_t n, uint8_t* bits) {
  int64_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(_mm256_loadu_si256((const __m256i*)(b + i)), 7));
    memcpy(bits + (i >> 3), &m, 4);
  }
  pack_base(b + i, n - i, bits + (i >> 3));
}
DF_AVX512 void pack_avx512(const uint8_t* b, int64_t n, uint8_t* bits) {
  int64_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m512i v = _mm512_loadu_si512(b + i);
    const uint64_t m = (uint64_t)_mm512_test_epi8_mask(v, v);
    memcpy(bits + (i >> 3), &m, 8);
  }
  pack_base(b + i, n - i, bits + (i >> 3));
}
void (*const kPack[3])(const uint8_t*, int64_t, uint8_t*) = {pack_base, pack_avx2, pack_avx512};
}  // namespace

namespace dfhip {
// the mask as bits ((n + 7) / 8 bytes at `bits`); `scratch` holds n bytes
int host_bernoulli_mask_bits(uint32_t* key, int* pos, double keep, int64_t n, uint8_t* scratch, uint8_t* bits) {
  const int st = mask_from_stream<uint8_t>(key, pos, keep, n, scratch);
  if (st != DFHIP_OK) return st;
  kPack[g_isa < 0 ? best_isa() : g_isa](scratch, n, bits);
  return DFHIP_OK;
}
}  // namespace dfhip

int dfhip_host_bernoulli_mask(uint32_t* key, int* pos, double keep, int64_t n, float* out) {
  if (!key || !pos || !out || n < 0 || *pos < 0 || *pos > kMtN || !(keep >= 0.0 && keep <= 1.0)) {
    set_error("host_bernoulli_mask: bad arguments");
    return DFHIP_INVALID_ARGUMENT;
  }
  return mask_from_stream<float>(key, pos, keep, n, out);
}

int dfhip_host_rng_isa(int level) {
  if (level >= 0) g_isa = level > best_isa() ? best_isa() : level;
  else g_isa = -1;
  return g_isa < 0 ? best_isa() : g_isa;
}
