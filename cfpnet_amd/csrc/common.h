// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 16-byte vector accesses).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/cfpnet_hip.h"

typedef unsigned short bf16_t;  // storage type of a bf16 element
typedef _Float16 f16_t;         // storage type of an IEEE half element (CFP_F16): 10 mantissa bits, |x| <= 65504

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ float h2f(f16_t v) { return (float)v; }
__device__ __forceinline__ f16_t f2h(float f) {   // round-to-nearest-even, saturating: an overflow stays finite
  return (f16_t)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);
}

// One 16-byte vector of T: 4 x f32 or 8 x bf16 / f16.
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  }
  __device__ static __forceinline__ void store(float* p, const float* v) {
    f32x4 x = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = x;
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    u32x4 x = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(x[i] << 16);
      v[2 * i + 1] = __uint_as_float(x[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    u32x4 x;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = x;
  }
};

template <> struct Vec<f16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const f16_t* p, float* v) {
    f16x8 x = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
  __device__ static __forceinline__ void store(f16_t* p, const float* v) {
    f16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f2h(v[i]);
    *reinterpret_cast<f16x8*>(p) = x;
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return f2h(v); }
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }

// Pre-split weight rows of the f16x3 kernels (conv_igemm_x3.hip, cfp_pack_w_x3): row `row` holds nk = ceil(K / 32) K-steps of 64 halves
// [hi(32) | lo(32)], channel r of a step at position 8 * ((r % 16) / 4) + r % 4 + 4 * (r / 16).  Four consecutive channels c .. c + 3
// (c % 4 == 0) are therefore four consecutive halves in each half-row: two 8-byte stores.
__device__ __forceinline__ void x3_store4(f16_t* __restrict__ wout, long long row, int nk, int c, const float* v) {
  const int ks = c >> 5, r = c & 31;
  f16_t* d = wout + (row * nk + ks) * 64 + ((r & 15) >> 2) * 8 + ((r >> 4) << 2);
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) { hi[e] = f2h(v[e]); lo[e] = (f16_t)(v[e] - (float)hi[e]); }
  *reinterpret_cast<f16x4*>(d) = hi;
  *reinterpret_cast<f16x4*>(d + 32) = lo;
}

// ---- f16x3 split (float32 storage, split-precision matrix math: conv_igemm_x3.hip, conv3x3_halo_x3.hip, loftr_tail_x3.hip) ----
// eight float32 -> (hi, lo) half operands.  hi by truncation (v_cvt_pkrtz never rounds a finite value up to infinity), x - hi is exact in
// float32, lo rounds it to nearest-even: |x - hi - lo| <= 2^-11 |x - hi| <= 2^-21 |x|.
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, f16x8& hi, f16x8& lo) {
  typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4& x = h == 0 ? x0 : x1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const h2_t t = __builtin_amdgcn_cvt_pkrtz(x[2 * q], x[2 * q + 1]);
      const f16x2 t2 = __builtin_bit_cast(f16x2, t);
      const float r0 = __builtin_fmaf((float)t2[0], -1.f, x[2 * q]);
      const float r1 = __builtin_fmaf((float)t2[1], -1.f, x[2 * q + 1]);
      hi[4 * h + 2 * q] = t2[0]; hi[4 * h + 2 * q + 1] = t2[1];
      lo[4 * h + 2 * q] = (_Float16)r0; lo[4 * h + 2 * q + 1] = (_Float16)r1;
    }
  }
}

// Two f32 -> one packed dword of H (low half = first element).
template <typename H> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)a, (__bf16)b};          // one v_cvt_pk_bf16_f32 (the scalar form cost two extra shuffles per pair)
  return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float a, float b) {
  f16x2 h = {f2h(a), f2h(b)};
  return __builtin_bit_cast(uint32_t, h);
}
// 16-bit storage element <-> raw bits (LDS planes, packed stores)
template <typename H> __device__ __forceinline__ H from_bits(unsigned short b) { return __builtin_bit_cast(H, b); }
template <typename H> __device__ __forceinline__ unsigned short to_bits(H v) { return __builtin_bit_cast(unsigned short, v); }

// D = A(16x32) * B(32x16) + C on the matrix core for either 16-bit format (same rate, same lane layout).
template <typename H> __device__ __forceinline__ f32x4 mfma16(const s16x8& a, const s16x8& b, const f32x4& c) {
  if constexpr (sizeof(H) == 2 && !__is_same(H, bf16_t))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case CFP_ACT_RELU: return x > 0.f ? x : 0.f;
    case CFP_ACT_LRELU: return x > 0.f ? x : 0.01f * x;
    case CFP_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
    case CFP_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case CFP_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __expf(-x));
    default: return x;
  }
}

// Activation with the kind as a compile-time constant, and a dispatcher that hoists the (wave-uniform)
// runtime switch out of per-element loops: `with_act(act, [&](auto A) { ... act_c<A.value>(x) ... })`.
template <int ACT> __device__ __forceinline__ float act_c(float x) {
  if constexpr (ACT == CFP_ACT_RELU) return x > 0.f ? x : 0.f;
  else if constexpr (ACT == CFP_ACT_LRELU) return x > 0.f ? x : 0.01f * x;
  else if constexpr (ACT == CFP_ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));   // v_rcp_f32: 1 ulp, no div sequence
  else if constexpr (ACT == CFP_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
  else if constexpr (ACT == CFP_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-x));
  else return x;
}
// GELU(erf) for results that are stored in 16 bits: erf by Abramowitz & Stegun 7.1.26 (|error| < 5e-7 in float32, four orders of
// magnitude under the storage rounding) -- 14 VALU instructions instead of erff's 33; the LKPM MLP applies it to 4 D values per token
// (19.7 M per launch at 1/4 scale: ~16 us of VALU issue with erff).  Float32 storage keeps act_c's erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.f));
  const float p = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.f - p * __expf(-u * u);
  return 0.5f * x * (1.f + copysignf(e, x));
}
template <int ACT> __device__ __forceinline__ float act_c16(float x) {      // act_c for kernels whose results leave in 16 bits
  if constexpr (ACT == CFP_ACT_GELU) return gelu_fast(x);
  else return act_c<ACT>(x);
}
template <int V> struct IntC { static constexpr int value = V; };
template <typename F> __device__ __forceinline__ void with_act(int act, F&& f) {
  switch (act) {
    case CFP_ACT_RELU: f(IntC<CFP_ACT_RELU>{}); break;
    case CFP_ACT_LRELU: f(IntC<CFP_ACT_LRELU>{}); break;
    case CFP_ACT_SILU: f(IntC<CFP_ACT_SILU>{}); break;
    case CFP_ACT_GELU: f(IntC<CFP_ACT_GELU>{}); break;
    case CFP_ACT_SIGMOID: f(IntC<CFP_ACT_SIGMOID>{}); break;
    default: f(IntC<CFP_ACT_NONE>{}); break;
  }
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : __expf(x); }  // elu(x)+1

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in every lane: four `v_add_f32 ... row_ror` -- rotations inside the
// row are a VALU operand modifier, where `__shfl_xor` compiles to `ds_bpermute_b32` (an LDS-crossbar round trip per step).
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

// Whole-wave reductions, result in every lane: DPP rotations inside the four 16-lane rows (VALU operand modifiers), then the four row
// results by v_readlane in a fixed order -- 11 instructions and no LDS traffic, where six `__shfl_xor` steps are six dependent
// ds_bpermute round trips (~100 cycles each; softmax_expect does three reductions per pixel row).
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// XCD-aware block index: hardware deals workgroups round-robin over the 8 XCDs (blocks b and b + 8 share an L2), so
// logically adjacent tiles that re-read the same lines land on 8 different L2s.  This bijective remap gives every XCD a
// contiguous chunk of the logical order instead (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- host side -------------------------------------------------------------------------
void cfp_set_error(const std::string& msg);
int cfp_check_launch(const char* what);

#define CFP_REQUIRE(cond, code, msg)        \
  do {                                      \
    if (!(cond)) {                          \
      cfp_set_error(std::string(msg));      \
      return (code);                        \
    }                                       \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool is16(int dtype) { return dtype == CFP_BF16 || dtype == CFP_F16; }
static inline bool dtype_ok(int dtype) { return dtype == CFP_F32 || is16(dtype); }
static inline int vec_elems(int dtype) { return is16(dtype) ? 8 : 4; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Division of a non-negative 32-bit index by a launch-constant divisor without the division sequence: with s = ceil(log2 d) and
// magic = floor(2^(31+s) / d) + 1 < 2^32,  floor(m / d) = mulhi(m, magic) >> (s - 1)  exactly for 0 <= m < 2^31 (the error term
// magic * d - 2^(31+s) lies in (0, d]).  d = 1 is magic 0.  The elementwise kernels split a flat vector index into (row, channel
// vector) once per 16 bytes moved; the compiler's 64-bit division there is a few dozen VALU instructions per vector.
struct FastDiv { unsigned d, magic, shift; };
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f{d, 0u, 0u};
  if (d <= 1) return f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.magic = (unsigned)(((1ull << (31 + s)) / d) + 1ull);
  f.shift = s - 1;
  return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned fd_div(unsigned m, const FastDiv& f) { return f.magic ? (__umulhi(m, f.magic) >> f.shift) : m; }
// flat index -> (row, channel vector) with CV = f.d vectors per row
__device__ __forceinline__ void fd_rowcol(unsigned i, const FastDiv& f, unsigned& row, unsigned& cv) { row = fd_div(i, f); cv = i - row * f.d; }
#endif

