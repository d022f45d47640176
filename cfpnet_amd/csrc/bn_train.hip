// Training-mode BatchNorm (batch statistics) + activation, forward and backward, on NHWC rows.
//
//   reference: every nn.BatchNorm2d / nn.BatchNorm1d of the model in model.train() (train.py:119-131): timm's
//   BatchNormAct2d in the encoder (encoder.py:57-69), decoder.py:45-50, transformer.py:240-244, convnext.py:30,
//   encoder.py:10-12 -- followed by SiLU / LeakyReLU / ReLU / nothing.
//
//   forward   mean_c, var_c (biased) over the B*H*W rows;  y = act((x - mean) * rsqrt(var + eps) * gamma + beta)
//             running_mean/var updated with `momentum` (running_var from the UNBIASED variance, like torch)
//   backward  dz = dy * act'(z);  dbeta = sum dz;  dgamma = sum dz * xhat;
//             dx = gamma * invstd * (dz - dbeta / n - xhat * dgamma / n)
//
// Three passes over the activation forward (sum, centred sum of squares, apply) and two backward (reduce, apply): all
// HBM-bound column reductions / elementwise sweeps over 16-byte vectors.  Reductions are per-split partials in f32
// combined in a fixed order (run-to-run deterministic); the variance is the two-pass form (no E[x^2] - E[x]^2
// cancellation).
#include "common.h"
#include <type_traits>

namespace {

inline int rc_colbits(int C, int ve) {          // column lanes of the column reductions: 2^bits >= C / ve, at most 32
  int b = 0;
  while ((1 << b) < C / ve && b < 5) ++b;
  return b;
}

// The float32 parity mode of the training path uses the full-precision expf and true division: the inference kernels' v_exp_f32 /
// v_rcp_f32 forms carry ~1e-6 relative error, which the batch-statistics BatchNorms of this network amplify into ~1e-2 of a gradient
// tensor -- visible against a float32 reference, invisible under 16-bit storage (see sigmoid_t below).
// FAST (the 16-bit storage modes): sigmoid through v_exp_f32 / v_rcp_f32 -- ~1e-6 relative, three orders of magnitude under the
// rounding of the stored activations / gradients, and ~3x fewer VALU instructions per element: these "bandwidth-bound" sweeps were
// VALU-bound (a wave64 VALU op occupies its SIMD for 4 cycles; SiLU's derivative with expf + a true division is ~30 of them per
// element, 8 elements per 16-byte load).  The float32 parity mode keeps expf and the true division.
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float z) {
  if constexpr (FAST) return __frcp_rn(1.f + __expf(-z));
  else return 1.f / (1.f + expf(-z));
}
template <int ACT, bool FAST = false> __device__ __forceinline__ float act_grad_c(float z) {      // act_grad, selected at compile time
  if constexpr (ACT == CFP_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  else if constexpr (ACT == CFP_ACT_LRELU) return z > 0.f ? 1.f : 0.01f;
  else if constexpr (ACT == CFP_ACT_SILU) { const float s = sigmoid_t<FAST>(z); return s * (1.f + z * (1.f - s)); }
  else if constexpr (ACT == CFP_ACT_GELU) return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * expf(-0.5f * z * z);
  else if constexpr (ACT == CFP_ACT_SIGMOID) { const float s = sigmoid_t<FAST>(z); return s * (1.f - s); }
  else return 1.f;
}
template <int ACT, bool FAST = false> __device__ __forceinline__ float act_precise(float x) {
  if constexpr (ACT == CFP_ACT_SILU) return FAST ? x * sigmoid_t<FAST>(x) : x / (1.f + expf(-x));
  else if constexpr (ACT == CFP_ACT_SIGMOID) return sigmoid_t<FAST>(x);
  else return act_c<ACT>(x);
}

// MODE 0: s1 = sum x                         (aux unused)
// MODE 1: s1 = sum (x - mean)^2              (aux0 = mean)
// MODE 2: s1 = sum dz, s2 = sum dz * xhat    (aux0 = mean, aux1 = invstd, aux2 = scale, aux3 = shift; dz = dy * act'(x*scale+shift))
// MODE 3: s1 = sum dy, s2 = sum dy * xhat    (LayerNorm: aux0 = per-ROW [rows][2] (mean, rstd))
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ x, int ld, const T* __restrict__ dy, int dy_ld, long long rows,
                                                        int C, const float* __restrict__ aux0, const float* __restrict__ aux1,
                                                        const float* __restrict__ aux2, const float* __restrict__ aux3, int act,
                                                        float* __restrict__ partial, long long rows_per_split, int colbits) {
  constexpr int VE = Vec<T>::N;
  constexpr int NS = MODE >= 2 ? 2 : 1;
  __shared__ float red[256][8 * 2];
  const int tid = threadIdx.x;
  // 2^colbits column lanes (<= 32), the other threads are row lanes: narrow tensors (C = 16 .. 64 at full resolution, the most
  // rows of the network) keep every lane busy instead of 1/8 of them
  const int RC_COLS = 1 << colbits, RC_LANES = 256 >> colbits;
  const int cl = tid & (RC_COLS - 1), rl = tid >> colbits;
  const int c0 = (blockIdx.x * RC_COLS + cl) * VE;
  const bool c_ok = c0 < C;
  const int cc = c_ok ? c0 : 0;
  const long long r0 = (long long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
  float a0[VE], a1[VE], a2[VE], a3[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    a0[e] = (MODE == 1 || MODE == 2) ? aux0[cc + e] : 0.f;
    a1[e] = (MODE == 2) ? aux1[cc + e] : 0.f;
    a2[e] = (MODE == 2) ? aux2[cc + e] : 0.f;
    a3[e] = (MODE == 2) ? aux3[cc + e] : 0.f;
  }
  float s1[VE], s2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  constexpr int U = 4;
  with_act(MODE == 2 ? act : CFP_ACT_NONE, [&](auto A) {          // the activation's derivative chosen once, not per element
  for (long long r = r0 + rl; r < r1; r += RC_LANES * U) {
    float v[U][VE], g[U][VE], rmean[U], rrstd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = min(r + (long long)u * RC_LANES, r1 - 1);
      Vec<T>::load(x + rr * ld + cc, v[u]);
      if (MODE >= 2) Vec<T>::load(dy + rr * dy_ld + cc, g[u]);
      if (MODE == 3) { rmean[u] = aux0[rr * 2]; rrstd[u] = aux0[rr * 2 + 1]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + (long long)u * RC_LANES >= r1) continue;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        if (MODE == 0) s1[e] += v[u][e];
        else if (MODE == 1) { const float d = v[u][e] - a0[e]; s1[e] = fmaf(d, d, s1[e]); }
        else if (MODE == 2) {
          const float dz = g[u][e] * act_grad_c<decltype(A)::value, !std::is_same<T, float>::value>(v[u][e] * a2[e] + a3[e]);
          s1[e] += dz;
          s2[e] = fmaf(dz, (v[u][e] - a0[e]) * a1[e], s2[e]);
        } else {
          s1[e] += g[u][e];
          s2[e] = fmaf(g[u][e], (v[u][e] - rmean[u]) * rrstd[u], s2[e]);
        }
      }
    }
  }
  });
#pragma unroll
  for (int e = 0; e < VE; ++e) { red[tid][e * 2] = s1[e]; red[tid][e * 2 + 1] = s2[e]; }
  __syncthreads();
  // fixed-order combination of the row lanes
  for (int i = tid; i < RC_COLS * VE * NS; i += 256) {
    const int which = i / (RC_COLS * VE), k = i - which * (RC_COLS * VE);
    const int c_l = k / VE, e = k - c_l * VE;
    const int c = (blockIdx.x * RC_COLS + c_l) * VE + e;
    if (c >= C) continue;
    float s = 0.f;
    for (int l = 0; l < RC_LANES; ++l) s += red[(l << colbits) + c_l][e * 2 + which];
    partial[((long long)blockIdx.y * NS + which) * C + c] = s;
  }
}

// ---- batch statistics in ONE pass over x ---------------------------------------------------------------------------------
// (count, mean, M2 = sum (x - mean)^2) triples merge exactly (Chan et al.):  n = nA + nB, d = mB - mA,
//   mean = mA + d * nB / n,   M2 = M2A + M2B + d^2 * nA * nB / n.
// A thread sums (x - c) and (x - c)^2 with c = the first value IT reads (a sample of the same channel, so |mean - c| is a few
// standard deviations at most and s2 - s1^2/n loses no more than a few ulps -- the textbook E[x^2] - E[x]^2 cancellation needs
// |mean| >> std relative to the shift), converts to a triple, and triples are merged in fixed orders: row lanes in LDS, row
// splits by one wave per channel.  Same tolerance as the two-pass form in the tests, one read of x instead of two.
struct Mom { float n, mean, m2; };
__device__ __forceinline__ Mom mom_merge(const Mom& a, const Mom& b) {
  const float n = a.n + b.n;
  if (n == 0.f) return Mom{0.f, 0.f, 0.f};
  const float d = b.mean - a.mean, f = b.n / n;
  return Mom{n, a.mean + d * f, a.m2 + b.m2 + d * d * a.n * f};
}

template <typename T>
__global__ __launch_bounds__(256) void colmoments_kernel(const T* __restrict__ x, int ld, long long rows, int C, float* __restrict__ partial,
                                                         long long rows_per_split, int colbits) {
  constexpr int VE = Vec<T>::N;
  __shared__ float red[256][8 * 2];
  __shared__ float cnt[256];
  const int tid = threadIdx.x;
  const int cols = 1 << colbits, lanes = 256 >> colbits;
  const int cl = tid & (cols - 1), rl = tid >> colbits;
  const int c0 = (blockIdx.x * cols + cl) * VE;
  const int cc = c0 < C ? c0 : 0;
  const long long r0 = (long long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
  float sh[VE], s1[VE], s2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  Vec<T>::load(x + min(r0 + rl, r1 - 1) * ld + cc, sh);
  float n = 0.f;
  constexpr int U = 4;
  for (long long r = r0 + rl; r < r1; r += lanes * U) {
    float v[U][VE];
#pragma unroll
    for (int u = 0; u < U; ++u) Vec<T>::load(x + min(r + (long long)u * lanes, r1 - 1) * ld + cc, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + (long long)u * lanes >= r1) continue;
      n += 1.f;
#pragma unroll
      for (int e = 0; e < VE; ++e) { const float d = v[u][e] - sh[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
    }
  }
  const float inv = n > 0.f ? 1.f / n : 0.f;
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    const float m = s1[e] * inv;
    red[tid][e * 2] = sh[e] + m;                       // the thread's mean
    red[tid][e * 2 + 1] = fmaxf(s2[e] - s1[e] * m, 0.f);   // and M2
  }
  cnt[tid] = n;
  __syncthreads();
  for (int i = tid; i < cols * VE; i += 256) {
    const int c_l = i / VE, e = i - c_l * VE;
    const int c = (blockIdx.x * cols + c_l) * VE + e;
    if (c >= C) continue;
    Mom a{0.f, 0.f, 0.f};
    for (int l = 0; l < lanes; ++l) {
      const int t = (l << colbits) + c_l;
      a = mom_merge(a, Mom{cnt[t], red[t][e * 2], red[t][e * 2 + 1]});
    }
    partial[((long long)blockIdx.y * 2 + 0) * C + c] = a.mean;
    partial[((long long)blockIdx.y * 2 + 1) * C + c] = a.m2;
  }
}

// One wave per channel merges the splits' triples (lanes stride over the splits, then a butterfly: a fixed tree), and finishes
// the layer's constants in the same launch: mean, biased variance, 1/std, the folded scale / shift of the apply pass and the
// running statistics (momentum update with the unbiased variance, like torch).
// WPC = waves per channel: 1 (four channels per workgroup), or 4 for the thousands of row tiles a convolution's epilogue leaves at the
// full-resolution layers (7 072 - 14 144 partials per channel and only 16 - 128 channels: one wave per channel was a ~55-round chain of
// dependent loads on a handful of workgroups); the four waves' results meet in LDS in wave order.
template <int WPC>
__global__ __launch_bounds__(256) void colmoments_final_kernel(const float* __restrict__ partial, int nsplit, long long rows,
                                                               long long rows_per_split, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float* __restrict__ mean, float* __restrict__ var, float* __restrict__ invstd,
                                                               float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ float wres[4][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = WPC == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (c >= C) return;                                   // WPC = 4: whole workgroups only (grid = C)
  const int first = WPC == 1 ? lane : (int)threadIdx.x, step = 64 * WPC;
  // four independent merge chains per lane (splits j, j + step, j + 2 step, j + 3 step of every 4 step): with the thousands of row tiles a
  // convolution's epilogue leaves (cfp_conv2d_nhwc_moments) one chain of dependent loads + divisions per lane was the whole launch
  Mom a4[4] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  for (int j0 = first; j0 < nsplit; j0 += 4 * step) {
    float pm[4], pq[4], pn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + step * u;
      const bool ok = j < nsplit;
      const long long r0 = (long long)j * rows_per_split;
      pn[u] = ok ? (float)(min(rows, r0 + rows_per_split) - r0) : 0.f;
      pm[u] = ok ? partial[((long long)j * 2 + 0) * C + c] : 0.f;
      pq[u] = ok ? partial[((long long)j * 2 + 1) * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] = mom_merge(a4[u], Mom{pn[u], pm[u], pq[u]});
  }
  Mom a = mom_merge(mom_merge(a4[0], a4[1]), mom_merge(a4[2], a4[3]));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    Mom b{__shfl_xor(a.n, o, 64), __shfl_xor(a.mean, o, 64), __shfl_xor(a.m2, o, 64)};
    a = (lane & o) ? mom_merge(b, a) : mom_merge(a, b);      // both partners compute the same (lower lane first) merge
  }
  if (WPC == 4) {
    if (lane == 0) { wres[wave][0] = a.n; wres[wave][1] = a.mean; wres[wave][2] = a.m2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    a = Mom{wres[0][0], wres[0][1], wres[0][2]};
#pragma unroll
    for (int w = 1; w < 4; ++w) a = mom_merge(a, Mom{wres[w][0], wres[w][1], wres[w][2]});
  } else if (lane != 0) {
    return;
  }
  const float v = a.m2 / (float)rows;
  const float is = 1.f / sqrtf(v + eps);
  const float g = gamma ? gamma[c] : 1.f, b0 = beta ? beta[c] : 0.f;
  mean[c] = a.mean; var[c] = v; invstd[c] = is;
  scale[c] = g * is; shift[c] = b0 - a.mean * g * is;
  const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
  // a batch whose statistics are not finite (an fp16 activation overflow: the step the optimizer then skips, cfp_grad_clip_factor) must
  // not poison the running statistics -- they outlive the step and are folded into every later validation engine
  const bool finite = fabsf(a.mean) <= 3.0e38f && fabsf(v) <= 3.0e38f;      // false for inf and NaN
  if (running_mean && finite) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * a.mean;
  if (running_var && finite) running_var[c] = (1.f - momentum) * running_var[c] + momentum * v * unbias;
}

// MODE 0 -> mean; MODE 1 -> var (biased) ; MODE 2/3 -> dbeta (which 0), dgamma (which 1).
// One wave per output: lanes stride over the splits, then a butterfly -- a fixed order, so still bit-reproducible.
__global__ __launch_bounds__(256) void colfinal_kernel(const float* __restrict__ partial, int nsplit, int ns, int C, float inv_n,
                                                       float* __restrict__ out0, float* __restrict__ out1) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= C * ns) return;
  const int which = i / C, c = i - which * C;
  float s = 0.f;
  for (int j = lane; j < nsplit; j += 64) s += partial[((long long)j * ns + which) * C + c];
  s = wave_sum(s);
  if (lane == 0) (which == 0 ? out0 : out1)[c] = s * inv_n;
}

// Elementwise sweeps with per-channel constants: a thread keeps ONE 16-byte channel vector (its constants live in registers)
// and walks rows -- 32 vector columns x 8 row lanes per workgroup, grid = (column blocks, row chunks).
template <typename T>
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const T* __restrict__ x, int ld, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act, T* __restrict__ out, int out_ld,
                                                              long long rows, int C, long long rows_per_chunk, int colbits,
                                                              const T* __restrict__ res, int res_ld) {
  constexpr int VE = Vec<T>::N;
  const int cols = 1 << colbits, lanes = 256 >> colbits;       // narrow tensors: fewer column lanes, more row lanes
  const int cl = threadIdx.x & (cols - 1), rl = threadIdx.x >> colbits;
  const int c = (blockIdx.x * cols + cl) * VE;
  if (c >= C) return;
  float sc[VE], sh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { sc[e] = scale[c + e]; sh[e] = shift[c + e]; }
  const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  with_act(act, [&](auto A) {
    constexpr int U = 4;
    for (long long r = r0 + rl; r < r1; r += lanes * U) {
      float v[U][VE], rs[U][VE];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long rr = min(r + (long long)u * lanes, r1 - 1);
        Vec<T>::load(x + rr * ld + c, v[u]);
        if (res) Vec<T>::load(res + rr * res_ld + c, rs[u]);        // uniform: the skip connection added after the activation
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long rr = r + (long long)u * lanes;
        if (rr >= r1) continue;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          v[u][e] = act_precise<decltype(A)::value, !std::is_same<T, float>::value>(v[u][e] * sc[e] + sh[e]);
          if (res) v[u][e] = to_f32<T>(from_f32<T>(v[u][e])) + rs[u][e];      // same rounding as the separate add of a stored tensor
        }
        Vec<T>::store(out + rr * out_ld + c, v[u]);
      }
    }
  });
}

// dx = gamma*invstd * (dz - dbeta/n - xhat * dgamma/n),  dz = dy * act'(x*scale+shift)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, int ld, const T* __restrict__ dy, int dy_ld,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_n,
                                                           int act, T* __restrict__ dx, int dx_ld, long long rows, int C,
                                                           long long rows_per_chunk, int colbits) {
  constexpr int VE = Vec<T>::N;
  const int cols = 1 << colbits, lanes = 256 >> colbits;       // narrow tensors: fewer column lanes, more row lanes
  const int cl = threadIdx.x & (cols - 1), rl = threadIdx.x >> colbits;
  const int c = (blockIdx.x * cols + cl) * VE;
  if (c >= C) return;
  float sc[VE], sh[VE], mu[VE], is[VE], k0[VE], k1[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    sc[e] = scale[c + e]; sh[e] = shift[c + e]; mu[e] = mean[c + e]; is[e] = invstd[c + e];
    k0[e] = dbeta[c + e] * inv_n; k1[e] = dgamma[c + e] * inv_n;
  }
  const long long r0 = (long long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  constexpr int U = 2;
  with_act(act, [&](auto A) {
  for (long long r = r0 + rl; r < r1; r += lanes * U) {
    float v[U][VE], g[U][VE];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = min(r + (long long)u * lanes, r1 - 1);
      Vec<T>::load(x + rr * ld + c, v[u]);
      Vec<T>::load(dy + rr * dy_ld + c, g[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + (long long)u * lanes;
      if (rr >= r1) continue;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float dz = g[u][e] * act_grad_c<decltype(A)::value, !std::is_same<T, float>::value>(v[u][e] * sc[e] + sh[e]);
        const float xh = (v[u][e] - mu[e]) * is[e];
        v[u][e] = sc[e] * (dz - k0[e] - xh * k1[e]);                     // scale = gamma * invstd
      }
      Vec<T>::store(dx + rr * dx_ld + c, v[u]);
    }
  }
  });
}

// LayerNorm backward, data part: one row per LPR lanes (LPR = C / VE, a power of two <= 64), like the forward kernel.
//   xhat = (x - mean) * rstd;  g = dy * gamma;  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat));  stats[row] = (mean, rstd)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, int ld, const T* __restrict__ dy, int dy_ld,
                                                            const float* __restrict__ gamma, float eps, T* __restrict__ dx, int dx_ld,
                                                            float* __restrict__ stats, long long rows, int C, int accumulate) {
  constexpr int VE = Vec<T>::N;
  const int LPR = C / VE;
  const int rows_per_block = 256 / LPR;
  const int lr = threadIdx.x % LPR, rb = threadIdx.x / LPR;
  float gm[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) gm[e] = gamma[lr * VE + e];
  const float inv_c = 1.f / (float)C;
  for (long long row0 = (long long)blockIdx.x * rows_per_block; row0 < rows; row0 += (long long)gridDim.x * rows_per_block) {
    const long long row = row0 + rb;
    const bool ok = row < rows;
    const long long rr = ok ? row : rows - 1;
    float v[VE], g[VE];
    Vec<T>::load(x + rr * ld + lr * VE, v);
    Vec<T>::load(dy + rr * dy_ld + lr * VE, g);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) s += v[e];
    for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
    for (int o = LPR >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * inv_c + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      v[e] = (v[e] - mean) * rstd;          // xhat
      g[e] *= gm[e];
      sg += g[e];
      sgx = fmaf(g[e], v[e], sgx);
    }
    for (int o = LPR >> 1; o > 0; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
    sg *= inv_c; sgx *= inv_c;
    if (ok) {
      float o_[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o_[e] = rstd * (g[e] - sg - v[e] * sgx);
      if (accumulate) {
        float a_[VE];
        Vec<T>::load(dx + row * dx_ld + lr * VE, a_);
#pragma unroll
        for (int e = 0; e < VE; ++e) o_[e] += a_[e];
      }
      Vec<T>::store(dx + row * dx_ld + lr * VE, o_);
      if (lr == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    }
  }
}

// dz = dy * act'(z)  (z = the pre-activation the forward kernel saw)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ z, int ld, const T* __restrict__ dy, int dy_ld, int act,
                                                      T* __restrict__ dz, int dz_ld, long long rows, int C, FastDiv fcv) {
  constexpr int VE = Vec<T>::N;
  const unsigned total = (unsigned)(rows * fcv.d);
  with_act(act, [&](auto A) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned ru, cvu;
    fd_rowcol(i, fcv, ru, cvu);
    const long long r = ru;
    const int c = (int)cvu * VE;
    float v[VE], g[VE];
    Vec<T>::load(z + r * ld + c, v);
    Vec<T>::load(dy + r * dy_ld + c, g);
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = g[e] * act_grad_c<decltype(A)::value, !std::is_same<T, float>::value>(v[e]);
    Vec<T>::store(dz + r * dz_ld + c, v);
  }
  });
}

int g_red_target = 1024;         // cfp_debug_set key 20: workgroups a column reduction aims for
int g_ew_target = 2048;          // cfp_debug_set key 21: workgroups an elementwise sweep aims for
inline int red_splits(long long rows, int C, int ve) {
  const int colbits = rc_colbits(C, ve);
  const int colblk = cdiv(C, ve << colbits);
  long long ns = g_red_target / colblk;
  const long long mx = (rows + 8 * (256 >> colbits) - 1) / (8 * (256 >> colbits));      // at least 8 rows per row lane
  if (ns > mx) ns = mx;
  if (ns > 4096) ns = 4096;
  if (ns < 1) ns = 1;
  return (int)ns;
}
struct EwGrid { unsigned gx, gy; long long rpc; int colbits; };
inline EwGrid ew_cols(long long rows, int C, int ve) {
  const int cv = C / ve;
  int colbits = 0;
  while ((1 << colbits) < cv && colbits < 5) ++colbits;            // 1 .. 32 column lanes
  const int cols = 1 << colbits, lanes = 256 >> colbits;
  const unsigned gx = (unsigned)cdiv(cv, cols);
  long long ns = g_ew_target / gx;                                   // ~8 workgroups per CU over the launch
  const long long mx = (rows + 4 * lanes - 1) / (4 * lanes);
  if (ns > mx) ns = mx;
  if (ns > 65535) ns = 65535;
  if (ns < 1) ns = 1;
  const long long rpc = (rows + ns - 1) / ns;
  return EwGrid{gx, (unsigned)((rows + rpc - 1) / rpc), rpc, colbits};
}
inline int ew_grid(long long total) { long long b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

template <int MODE>
int launch_reduce(const void* x, int ld, const void* dy, int dy_ld, long long rows, int C, const float* a0, const float* a1,
                  const float* a2, const float* a3, int act, int dtype, float* partial, float* out0, float* out1, float inv_n,
                  hipStream_t s, cfp_wgrad_job* job = nullptr) {
  const int ve = vec_elems(dtype);
  const int ns = red_splits(rows, C, ve);
  const long long rps = (rows + ns - 1) / ns;
  const int nsplit = (int)((rows + rps - 1) / rps);
  const int colbits = rc_colbits(C, ve);
  const dim3 grid(cdiv(C, ve << colbits), nsplit);
#define RL(T) hipLaunchKernelGGL((colreduce_kernel<T, MODE>), grid, dim3(256), 0, s, (const T*)x, ld, (const T*)dy, dy_ld, rows, C, a0, a1, a2, a3, \
                                 act, partial, rps, colbits)
  if (dtype == CFP_BF16) RL(bf16_t); else if (dtype == CFP_F16) RL(f16_t); else RL(float);
#undef RL
  const int nsum = MODE >= 2 ? 2 : 1;
  if (job) {      // partial is [split][out0 (C) | out1 (C)]: the slab layout of cfp_wgrad_reduce_jobs (inv_n must be 1)
    job->slabs = partial; job->dw = out0; job->db = nsum == 2 ? out1 : nullptr; job->n = (long long)C * nsum; job->n_dw = C; job->nsplit = nsplit;
    job->ew = nsplit < 8 ? 256 : nsplit < 32 ? 64 : 32; job->beta = 0.f; job->beta_b = 0.f;
    return 0;
  }
  hipLaunchKernelGGL(colfinal_kernel, dim3(cdiv(C * nsum, 4)), dim3(256), 0, s, partial, nsplit, nsum, C, inv_n, out0, out1);
  return 0;
}

inline void launch_moments_final(const float* partial, int nsplit, long long rows, long long rps, int C, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, float* mean, float* var, float* invstd,
                                 float* scale, float* shift, hipStream_t s) {
  if (nsplit >= 256)
    hipLaunchKernelGGL(colmoments_final_kernel<4>, dim3(C), dim3(256), 0, s, partial, nsplit, rows, rps, C, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, var, invstd, scale, shift);
  else
    hipLaunchKernelGGL(colmoments_final_kernel<1>, dim3(cdiv(C, 4)), dim3(256), 0, s, partial, nsplit, rows, rps, C, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, var, invstd, scale, shift);
}

}  // namespace

void cfp_bn_debug_set(int key, int value) { (key == 20 ? g_red_target : g_ew_target) = value; }

extern "C" size_t cfp_bn_ws_bytes(int C) { return C > 0 ? (size_t)4096 * 2 * C * sizeof(float) : 0; }

#define BN_COMMON(name)                                                                                                      \
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, name ": bad dtype");                                                              \
  const int ve = vec_elems(dtype);                                                                                           \
  CFP_REQUIRE(rows > 0 && C > 0 && C % ve == 0 && ld % ve == 0 && ld >= C, CFP_ESHAPE, name ": bad shape");                  \
  hipStream_t s = reinterpret_cast<hipStream_t>(stream)

extern "C" int cfp_bn_train_stats(const void* x, int ld, long long rows, int C, int dtype, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, float* mean, float* var, float* invstd,
                                  float* scale, float* shift, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(x && mean && var && invstd && scale && shift && ws && aligned16(x), CFP_EINVAL, "cfp_bn_train_stats: bad pointer");
  BN_COMMON("cfp_bn_train_stats");
  CFP_REQUIRE(ws_bytes >= cfp_bn_ws_bytes(C), CFP_EINVAL, "cfp_bn_train_stats: workspace too small");
  float* partial = reinterpret_cast<float*>(ws);
  {
    const int ns = red_splits(rows, C, ve);
    const long long rps = (rows + ns - 1) / ns;
    const int nsplit = (int)((rows + rps - 1) / rps);
    const int colbits = rc_colbits(C, ve);
    const dim3 grid(cdiv(C, ve << colbits), nsplit);
#define ML(T) hipLaunchKernelGGL(colmoments_kernel<T>, grid, dim3(256), 0, s, (const T*)x, ld, rows, C, partial, rps, colbits)
    if (dtype == CFP_BF16) ML(bf16_t); else if (dtype == CFP_F16) ML(f16_t); else ML(float);
#undef ML
    launch_moments_final(partial, nsplit, rows, rps, C, gamma, beta, eps, momentum, running_mean, running_var, mean, var, invstd, scale, shift, s);
  }
  return cfp_check_launch("cfp_bn_train_stats");
}

// The same layer constants from per-split (mean, M2) partials a producer already wrote (cfp_conv2d_nhwc_moments): split j holds the rows
// [j * rows_per_split, min(rows, (j + 1) * rows_per_split)), layout [split][2][C].  One launch instead of a pass over the tensor + one.
extern "C" int cfp_bn_train_stats_partials(const float* partial, int nsplit, long long rows, long long rows_per_split, int C, const float* gamma,
                                           const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                           float* var, float* invstd, float* scale, float* shift, cfp_stream_t stream) {
  CFP_REQUIRE(partial && mean && var && invstd && scale && shift, CFP_EINVAL, "cfp_bn_train_stats_partials: bad pointer");
  CFP_REQUIRE(nsplit > 0 && rows > 0 && rows_per_split > 0 && C > 0 && (long long)(nsplit - 1) * rows_per_split < rows &&
                  (long long)nsplit * rows_per_split >= rows, CFP_ESHAPE, "cfp_bn_train_stats_partials: splits do not cover the rows");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  launch_moments_final(partial, nsplit, rows, rows_per_split, C, gamma, beta, eps, momentum, running_mean, running_var, mean, var, invstd, scale, shift, s);
  return cfp_check_launch("cfp_bn_train_stats_partials");
}

extern "C" int cfp_scale_shift_act(const void* x, int ld, const float* scale, const float* shift, int act, void* out, int out_ld,
                                   long long rows, int C, int dtype, cfp_stream_t stream) {
  return cfp_scale_shift_act_res(x, ld, scale, shift, act, nullptr, 0, out, out_ld, rows, C, dtype, stream);
}

extern "C" int cfp_scale_shift_act_res(const void* x, int ld, const float* scale, const float* shift, int act, const void* res, int res_ld,
                                       void* out, int out_ld, long long rows, int C, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(x && scale && shift && out && aligned16(x) && aligned16(out) && aligned16(res), CFP_EINVAL, "cfp_scale_shift_act: bad pointer");
  BN_COMMON("cfp_scale_shift_act");
  CFP_REQUIRE(out_ld % ve == 0 && out_ld >= C && (!res || (res_ld % ve == 0 && res_ld >= C)), CFP_ESHAPE, "cfp_scale_shift_act: bad pitch");
  const EwGrid eg = ew_cols(rows, C, ve);
  const long long rpc = eg.rpc;
  const dim3 grid(eg.gx, eg.gy);
#define SL(T) hipLaunchKernelGGL(scale_shift_act_kernel<T>, grid, dim3(256), 0, s, (const T*)x, ld, scale, shift, act, (T*)out, out_ld, rows, C, rpc, eg.colbits, \
                                 (const T*)res, res_ld)
  if (dtype == CFP_BF16) SL(bf16_t); else if (dtype == CFP_F16) SL(f16_t); else SL(float);
#undef SL
  return cfp_check_launch("cfp_scale_shift_act");
}

extern "C" int cfp_bn_train_bwd(const void* x, int ld, const void* dy, int dy_ld, long long rows, int C, int dtype, const float* mean,
                                const float* invstd, const float* scale, const float* shift, int act, float* dgamma, float* dbeta,
                                void* dx, int dx_ld, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(x && dy && mean && invstd && scale && shift && dgamma && dbeta && dx && ws && aligned16(x) && aligned16(dy) && aligned16(dx),
              CFP_EINVAL, "cfp_bn_train_bwd: bad pointer");
  BN_COMMON("cfp_bn_train_bwd");
  CFP_REQUIRE(dy_ld % ve == 0 && dy_ld >= C && dx_ld % ve == 0 && dx_ld >= C, CFP_ESHAPE, "cfp_bn_train_bwd: bad pitch");
  CFP_REQUIRE(ws_bytes >= cfp_bn_ws_bytes(C), CFP_EINVAL, "cfp_bn_train_bwd: workspace too small");
  float* partial = reinterpret_cast<float*>(ws);
  // dbeta = sum dz, dgamma = sum dz * xhat (unscaled sums: inv_n = 1)
  launch_reduce<2>(x, ld, dy, dy_ld, rows, C, mean, invstd, scale, shift, act, dtype, partial, dbeta, dgamma, 1.f, s);
  const EwGrid eg = ew_cols(rows, C, ve);
  const long long rpc = eg.rpc;
  const dim3 grid(eg.gx, eg.gy);
  const float inv_n = 1.f / (float)rows;
#define BL(T) hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, grid, dim3(256), 0, s, (const T*)x, ld, (const T*)dy, dy_ld, mean, invstd, scale, shift, \
                                 dbeta, dgamma, inv_n, act, (T*)dx, dx_ld, rows, C, rpc, eg.colbits)
  if (dtype == CFP_BF16) BL(bf16_t); else if (dtype == CFP_F16) BL(f16_t); else BL(float);
#undef BL
  return cfp_check_launch("cfp_bn_train_bwd");
}

/* Column sum over rows: the bias gradient of a conv / linear layer (sum over pixels of dy). */
extern "C" int cfp_colsum(const void* x, int ld, long long rows, int C, int dtype, float* out, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(x && out && ws && aligned16(x), CFP_EINVAL, "cfp_colsum: bad pointer");
  BN_COMMON("cfp_colsum");
  CFP_REQUIRE(ws_bytes >= cfp_bn_ws_bytes(C), CFP_EINVAL, "cfp_colsum: workspace too small");
  launch_reduce<0>(x, ld, nullptr, 0, rows, C, nullptr, nullptr, nullptr, nullptr, 0, dtype, reinterpret_cast<float*>(ws), out, nullptr, 1.f, s);
  return cfp_check_launch("cfp_colsum");
}

extern "C" int cfp_act_bwd(const void* z, int ld, const void* dy, int dy_ld, int act, void* dz, int dz_ld, long long rows, int C, int dtype,
                           cfp_stream_t stream) {
  CFP_REQUIRE(z && dy && dz && aligned16(z) && aligned16(dy) && aligned16(dz), CFP_EINVAL, "cfp_act_bwd: bad pointer");
  BN_COMMON("cfp_act_bwd");
  CFP_REQUIRE(dy_ld % ve == 0 && dy_ld >= C && dz_ld % ve == 0 && dz_ld >= C, CFP_ESHAPE, "cfp_act_bwd: bad pitch");
  const dim3 grid(ew_grid(rows * (C / ve)));
  CFP_REQUIRE(rows * (C / ve) < (1ll << 31), CFP_ESHAPE, "cfp_act_bwd: too many elements");
  const FastDiv fcv = make_fastdiv((unsigned)(C / ve));
#define AL(T) hipLaunchKernelGGL(act_bwd_kernel<T>, grid, dim3(256), 0, s, (const T*)z, ld, (const T*)dy, dy_ld, act, (T*)dz, dz_ld, rows, C, fcv)
  if (dtype == CFP_BF16) AL(bf16_t); else if (dtype == CFP_F16) AL(f16_t); else AL(float);
#undef AL
  return cfp_check_launch("cfp_act_bwd");
}

extern "C" size_t cfp_layernorm_bwd_ws_bytes(long long rows, int C) {
  return (rows > 0 && C > 0) ? cfp_bn_ws_bytes(C) + (size_t)rows * 2 * sizeof(float) : 0;
}

/* Backward of nn.LayerNorm over the channel axis: dx (+= when accumulate), dgamma, dbeta. */
static int layernorm_bwd_impl(const void* x, int ld, const void* dy, int dy_ld, const float* gamma, float eps, void* dx, int dx_ld,
                              int accumulate, float* dgamma, float* dbeta, long long rows, int C, int dtype, void* ws, size_t ws_bytes,
                              cfp_wgrad_job* job, cfp_stream_t stream) {
  CFP_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && ws && aligned16(x) && aligned16(dy) && aligned16(dx), CFP_EINVAL,
              "cfp_layernorm_bwd: bad pointer");
  BN_COMMON("cfp_layernorm_bwd");
  const int lpr = C / ve;
  CFP_REQUIRE(lpr <= 64 && (lpr & (lpr - 1)) == 0, CFP_ESHAPE, "cfp_layernorm_bwd: C / vector width must be a power of two <= 64");
  CFP_REQUIRE(dy_ld % ve == 0 && dy_ld >= C && dx_ld % ve == 0 && dx_ld >= C, CFP_ESHAPE, "cfp_layernorm_bwd: bad pitch");
  CFP_REQUIRE(ws_bytes >= cfp_layernorm_bwd_ws_bytes(rows, C), CFP_EINVAL, "cfp_layernorm_bwd: workspace too small");
  float* partial = reinterpret_cast<float*>(ws);
  float* stats = partial + cfp_bn_ws_bytes(C) / sizeof(float);
  long long blocks = (rows + (256 / lpr) - 1) / (256 / lpr);
  if (blocks > 4096) blocks = 4096;
#define LL(T) hipLaunchKernelGGL(layernorm_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, (const T*)x, ld, (const T*)dy, dy_ld, gamma, eps, \
                                 (T*)dx, dx_ld, stats, rows, C, accumulate)
  if (dtype == CFP_BF16) LL(bf16_t); else if (dtype == CFP_F16) LL(f16_t); else LL(float);
#undef LL
  launch_reduce<3>(x, ld, dy, dy_ld, rows, C, stats, nullptr, nullptr, nullptr, 0, dtype, partial, dbeta, dgamma, 1.f, s, job);
  return cfp_check_launch("cfp_layernorm_bwd");
}

extern "C" int cfp_layernorm_bwd(const void* x, int ld, const void* dy, int dy_ld, const float* gamma, float eps, void* dx, int dx_ld,
                                 int accumulate, float* dgamma, float* dbeta, long long rows, int C, int dtype, void* ws, size_t ws_bytes,
                                 cfp_stream_t stream) {
  return layernorm_bwd_impl(x, ld, dy, dy_ld, gamma, eps, dx, dx_ld, accumulate, dgamma, dbeta, rows, C, dtype, ws, ws_bytes, nullptr, stream);
}

/* The same with the finishing sum of the parameter-gradient partials left to cfp_wgrad_reduce_jobs (a job record like the dense weight
 * gradients': [split][dbeta | dgamma] slabs); `ws` must stay untouched until the job has been reduced. */
extern "C" int cfp_layernorm_bwd_deferred(const void* x, int ld, const void* dy, int dy_ld, const float* gamma, float eps, void* dx, int dx_ld,
                                          int accumulate, float* dgamma, float* dbeta, long long rows, int C, int dtype, void* ws,
                                          size_t ws_bytes, cfp_wgrad_job* job, cfp_stream_t stream) {
  CFP_REQUIRE(job, CFP_EINVAL, "cfp_layernorm_bwd_deferred: null job");
  return layernorm_bwd_impl(x, ld, dy, dy_ld, gamma, eps, dx, dx_ld, accumulate, dgamma, dbeta, rows, C, dtype, ws, ws_bytes, job, stream);
}
