// Linear attention (elu+1 feature map), training form: forward that keeps the per-(group, head) KV state, and backward.
//
//   reference: LinearAttention.forward (src/models/attention.py:20-52) and its autograd:
//     Q = elu(q)+1, K = elu(k)+1, values = v / S
//     KV = sum_s K_s (x) values_s            [d x d]     Ksum = sum_s K_s
//     Z_l = 1 / (Q_l . Ksum + eps)           out_l = (Q_l KV) * Z_l * S
//   Tokens are grouped contiguously: q [N*L, heads*d], k, v [N*S, heads*d] (the zone / window / inside-outside groupings
//   are row gathers done before, cfp_index_rows).  One workgroup per (group n, head h).
//
//   backward, with A_l = Q_l KV:   dA_l = dout_l * S * Z_l        e_l = -Z_l^2 * S * (dout_l . A_l)
//     dQ_l = KV dA_l + e_l Ksum     dKV = sum_l Q_l (x) dA_l       dKsum = sum_l e_l Q_l
//     dK_s = dKV values_s + dKsum   dvalues_s = dKV^T K_s          dv = dvalues / S;   dq = dQ * elu'(q), dk = dK * elu'(k)
//
// The two reductions over tokens (KV over keys, dKV over queries) are the same "augmented outer product" M[i][j] =
// sum_r X_r[i] * Y_r[j], j = 0..d with Y_r[d] = 1 (Ksum) or e_r (dKsum): rows are staged through LDS in chunks and every
// thread owns fixed (i, j) entries, so the summation order is fixed (bit-reproducible).
#include "common.h"

namespace {

constexpr int ACH = 128;     // token rows per LDS chunk

__device__ __forceinline__ float elu1_grad(float x) { return x > 0.f ? 1.f : expf(x); }
__device__ __forceinline__ float elu1p(float x) { return x > 0.f ? x + 1.f : expf(x); }      // full-precision elu(x)+1 (training parity)

// NT threads per workgroup (256, or one wave for groups of at most 64 tokens), RCH token rows per LDS chunk
template <int D, int NT>
struct Outer {
  static constexpr int NO = D * (D + 1);
  static constexpr int G = NO >= NT ? 1 : NT / NO;
  static constexpr int OPT = (NO + NT - 1) / NT;
  static constexpr int RCH = NT == 64 ? 64 : ACH;
};

// acc += sum over the chunk's rows of X[r][i] * Y[r][j]   (sX: [ACH][D], sY: [ACH][D+1])
template <int D, int NT, int PX = D>
__device__ __forceinline__ void outer_accumulate(const float* sX, const float* sY, int nrows, float (&acc)[Outer<D, NT>::OPT]) {
  using O = Outer<D, NT>;
  const int tid = threadIdx.x;
  if (O::G == 1) {
#pragma unroll
    for (int u = 0; u < O::OPT; ++u) {
      const int o = tid + u * NT;
      if (o < O::NO) {
        const int i = o / (D + 1), j = o - i * (D + 1);
        float a = acc[u];
        for (int r = 0; r < nrows; ++r) a = fmaf(sX[r * PX + i], sY[r * (D + 1) + j], a);
        acc[u] = a;
      }
    }
  } else {
    const int g = tid / O::NO, o = tid - g * O::NO;
    if (g < O::G) {
      const int i = o / (D + 1), j = o - i * (D + 1);
      float a = acc[0];
      for (int r = g; r < nrows; r += O::G) a = fmaf(sX[r * PX + i], sY[r * (D + 1) + j], a);
      acc[0] = a;
    }
  }
}

// combine the per-thread partial sums into sM[D][D+1]
template <int D, int NT>
__device__ __forceinline__ void outer_finish(const float (&acc)[Outer<D, NT>::OPT], float* sRed, float* sM) {
  using O = Outer<D, NT>;
  const int tid = threadIdx.x;
  if (O::G == 1) {
#pragma unroll
    for (int u = 0; u < O::OPT; ++u) {
      const int o = tid + u * NT;
      if (o < O::NO) sM[o] = acc[u];
    }
  } else {
    const int g = tid / O::NO, o = tid - g * O::NO;
    if (g < O::G) sRed[g * O::NO + o] = acc[0];
    __syncthreads();
    if (tid < O::NO) {
      float a = 0.f;
      for (int gg = 0; gg < O::G; ++gg) a += sRed[gg * O::NO + tid];
      sM[tid] = a;
    }
  }
  __syncthreads();
}

// SPLIT 0: one workgroup per (group, head) does both phases.  When there are few groups (global attention: N = batch) the token
// axis is split over blockIdx.y instead: SPLIT 1 = the key phase of one chunk of S, partial KV to `partial`; SPLIT 2 = the
// query phase of one chunk of L, which first adds the partials up in chunk order (so the result does not depend on the split
// only through the fixed summation tree: bit-reproducible for a given shape).
template <typename T, int D, int SPLIT, int NT>
__global__ __launch_bounds__(NT) void linattn_fwd_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ k, int k_ld,
                                                          const T* __restrict__ v, int v_ld, T* __restrict__ out, int out_ld,
                                                          float* __restrict__ kv_save, int L, int S, int heads, float eps,
                                                          float* __restrict__ partial, int nchunks, int chunk_len) {
  using O = Outer<D, NT>;
  constexpr int RCH = O::RCH;
  __shared__ float sX[SPLIT == 2 ? 1 : RCH * D], sY[SPLIT == 2 ? 1 : RCH * (D + 1)], sM[O::NO], sRed[O::G > 1 ? O::G * O::NO : 1];
  const int tid = threadIdx.x;
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const float fS = (float)S;
  float acc[O::OPT];
#pragma unroll
  for (int u = 0; u < O::OPT; ++u) acc[u] = 0.f;
  const int s_begin = SPLIT == 1 ? blockIdx.y * chunk_len : 0, s_end = SPLIT == 1 ? min(S, s_begin + chunk_len) : (SPLIT == 2 ? 0 : S);
  for (int s0 = s_begin; s0 < s_end; s0 += RCH) {
    const int nr = min(RCH, s_end - s0);
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, c = e - r * D;
      const long long row = (long long)n * S + s0 + r;
      sX[r * D + c] = elu1p(to_f32<T>(k[row * k_ld + h * D + c]));
      sY[r * (D + 1) + c] = to_f32<T>(v[row * v_ld + h * D + c]) / fS;
      if (c == 0) sY[r * (D + 1) + D] = 1.f;
    }
    __syncthreads();
    outer_accumulate<D, NT>(sX, sY, nr, acc);
  }
  __syncthreads();
  if (SPLIT == 2) {
    for (int e = tid; e < O::NO; e += NT) {
      float a = 0.f;
      for (int c = 0; c < nchunks; ++c) a += partial[((long long)blockIdx.x * nchunks + c) * O::NO + e];
      sM[e] = a;
    }
    __syncthreads();
  } else {
    outer_finish<D, NT>(acc, sRed, sM);
  }
  if (SPLIT == 1) {
    for (int e = tid; e < O::NO; e += NT) partial[((long long)blockIdx.x * gridDim.y + blockIdx.y) * O::NO + e] = sM[e];
    return;
  }
  if (SPLIT == 0 || blockIdx.y == 0)
    for (int e = tid; e < O::NO; e += NT) kv_save[(long long)blockIdx.x * O::NO + e] = sM[e];
  // queries: one token per thread
  const int l_begin = SPLIT == 2 ? blockIdx.y * chunk_len : 0, l_end = SPLIT == 2 ? min(L, l_begin + chunk_len) : L;
  for (int l = l_begin + tid; l < l_end; l += NT) {
    const long long row = (long long)n * L + l;
    float Q[D];
    float den = eps;
#pragma unroll
    for (int i = 0; i < D; ++i) { Q[i] = elu1p(to_f32<T>(q[row * q_ld + h * D + i])); den = fmaf(Q[i], sM[i * (D + 1) + D], den); }
    const float z = 1.f / den;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < D; ++i) a = fmaf(Q[i], sM[i * (D + 1) + j], a);
      out[row * out_ld + h * D + j] = from_f32<T>(a * z * fS);
    }
  }
}

// SPLIT as in the forward: 0 = both phases in one workgroup per (group, head); 1 = the query phase of one chunk of L (dq, partial
// [dKV | dKsum] to `partial`); 2 = the key phase of one chunk of S after adding the partials up in chunk order.
template <typename T, int D, int SPLIT, int NT>
__global__ __launch_bounds__(NT) void linattn_bwd_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ k, int k_ld,
                                                          const T* __restrict__ v, int v_ld, const T* __restrict__ dout, int do_ld,
                                                          const float* __restrict__ kv_save, T* __restrict__ dq, int dq_ld,
                                                          T* __restrict__ dk, int dk_ld, T* __restrict__ dv, int dv_ld, int L, int S,
                                                          int heads, float eps, float* __restrict__ partial, int nchunks, int chunk_len) {
  using O = Outer<D, NT>;
  constexpr int RCH = O::RCH;
  __shared__ float sX[SPLIT == 2 ? 1 : RCH * D], sY[SPLIT == 2 ? 1 : RCH * (D + 1)], sM[SPLIT == 2 ? 1 : O::NO], sG[O::NO],
      sRed[O::G > 1 ? O::G * O::NO : 1];
  const int tid = threadIdx.x;
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const float fS = (float)S;
  if (SPLIT != 2)
    for (int e = tid; e < O::NO; e += NT) sM[e] = kv_save[(long long)blockIdx.x * O::NO + e];
  float acc[O::OPT];
#pragma unroll
  for (int u = 0; u < O::OPT; ++u) acc[u] = 0.f;
  __syncthreads();
  // ---- queries: dq, and the rows (Q_l, dA_l, e_l) of the dKV / dKsum reduction
  const int l_begin = SPLIT == 1 ? blockIdx.y * chunk_len : 0, l_end = SPLIT == 1 ? min(L, l_begin + chunk_len) : (SPLIT == 2 ? 0 : L);
  for (int l0 = l_begin; l0 < l_end; l0 += RCH) {
    const int nr = min(RCH, l_end - l0);
    __syncthreads();
    if (tid < nr) {
      const long long row = (long long)n * L + l0 + tid;
      float Q[D], qr[D], A[D], g[D];
      float den = eps;
#pragma unroll
      for (int i = 0; i < D; ++i) { qr[i] = to_f32<T>(q[row * q_ld + h * D + i]); Q[i] = elu1p(qr[i]); den = fmaf(Q[i], sM[i * (D + 1) + D], den); }
      const float z = 1.f / den;
      float dotA = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) a = fmaf(Q[i], sM[i * (D + 1) + j], a);
        A[j] = a;
        g[j] = to_f32<T>(dout[row * do_ld + h * D + j]);
        dotA = fmaf(g[j], a, dotA);
      }
      const float e_l = -z * z * fS * dotA;
#pragma unroll
      for (int j = 0; j < D; ++j) { g[j] *= fS * z; sY[tid * (D + 1) + j] = g[j]; }     // dA_l
      sY[tid * (D + 1) + D] = e_l;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        float d = e_l * sM[i * (D + 1) + D];
#pragma unroll
        for (int j = 0; j < D; ++j) d = fmaf(sM[i * (D + 1) + j], g[j], d);
        dq[row * dq_ld + h * D + i] = from_f32<T>(d * elu1_grad(qr[i]));
        sX[tid * D + i] = Q[i];
      }
    }
    __syncthreads();
    outer_accumulate<D, NT>(sX, sY, nr, acc);
  }
  __syncthreads();
  if (SPLIT == 2) {
    for (int e = tid; e < O::NO; e += NT) {
      float a = 0.f;
      for (int c = 0; c < nchunks; ++c) a += partial[((long long)blockIdx.x * nchunks + c) * O::NO + e];
      sG[e] = a;
    }
    __syncthreads();
  } else {
    outer_finish<D, NT>(acc, sRed, sG);          // sG = [dKV | dKsum]
  }
  if (SPLIT == 1) {
    for (int e = tid; e < O::NO; e += NT) partial[((long long)blockIdx.x * gridDim.y + blockIdx.y) * O::NO + e] = sG[e];
    return;
  }
  // ---- keys: dk, dv
  const int s_begin = SPLIT == 2 ? blockIdx.y * chunk_len : 0, s_end = SPLIT == 2 ? min(S, s_begin + chunk_len) : S;
  for (int s = s_begin + tid; s < s_end; s += NT) {
    const long long row = (long long)n * S + s;
    float K[D], kr[D], val[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      kr[i] = to_f32<T>(k[row * k_ld + h * D + i]);
      K[i] = elu1p(kr[i]);
      val[i] = to_f32<T>(v[row * v_ld + h * D + i]) / fS;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float d = sG[i * (D + 1) + D];
#pragma unroll
      for (int j = 0; j < D; ++j) d = fmaf(sG[i * (D + 1) + j], val[j], d);
      dk[row * dk_ld + h * D + i] = from_f32<T>(d * elu1_grad(kr[i]));
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < D; ++i) d = fmaf(K[i], sG[i * (D + 1) + j], d);
      dv[row * dv_ld + h * D + j] = from_f32<T>(d / fS);
    }
  }
}


// ---- head dims 16 / 32: staged form ------------------------------------------------------------------------------------------
// The kernels above keep a token's whole head (Q, A, dA, ... = 4 D floats) in the registers of ONE lane and unroll D x (D+1)
// products per lane: at D = 32 that is 256 VGPRs plus 1.3 KB of scratch per lane and one wave per SIMD (296 us for the 2 304
// 16-token zone groups of the 1/16 scale).  Here every per-token product is a thread-per-OUTPUT loop over operands staged in
// LDS (pitch D+1: conflict-free both along a row and down a column), so all lanes work whatever the group size and nothing
// spills.  Same mathematics, same fixed summation orders (each output is one sequential fma chain).
template <typename T, int D, int SPLIT, int NT>
__global__ __launch_bounds__(NT) void linattn_fwd2_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ k, int k_ld,
                                                           const T* __restrict__ v, int v_ld, T* __restrict__ out, int out_ld,
                                                           float* __restrict__ kv_save, int L, int S, int heads, float eps,
                                                           float* __restrict__ partial, int nchunks, int chunk_len) {
  using O = Outer<D, NT>;
  constexpr int RCH = NT == 64 ? 32 : 64, P = D + 1;
  __shared__ float sX[RCH * P], sY[SPLIT == 2 ? 1 : RCH * P], sM[O::NO], sZ[RCH], sRed[O::G > 1 ? O::G * O::NO : 1];
  const int tid = threadIdx.x;
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const float fS = (float)S;
  float acc[O::OPT];
#pragma unroll
  for (int u = 0; u < O::OPT; ++u) acc[u] = 0.f;
  const int s_begin = SPLIT == 1 ? blockIdx.y * chunk_len : 0, s_end = SPLIT == 1 ? min(S, s_begin + chunk_len) : (SPLIT == 2 ? 0 : S);
  for (int s0 = s_begin; s0 < s_end; s0 += RCH) {
    const int nr = min(RCH, s_end - s0);
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, c = e - r * D;
      const long long row = (long long)n * S + s0 + r;
      sX[r * P + c] = elu1p(to_f32<T>(k[row * k_ld + h * D + c]));
      sY[r * P + c] = to_f32<T>(v[row * v_ld + h * D + c]) / fS;
      if (c == 0) sY[r * P + D] = 1.f;
    }
    __syncthreads();
    outer_accumulate<D, NT, P>(sX, sY, nr, acc);
  }
  __syncthreads();
  if (SPLIT == 2) {
    for (int e = tid; e < O::NO; e += NT) {
      float a = 0.f;
      for (int c = 0; c < nchunks; ++c) a += partial[((long long)blockIdx.x * nchunks + c) * O::NO + e];
      sM[e] = a;
    }
    __syncthreads();
  } else {
    outer_finish<D, NT>(acc, sRed, sM);
  }
  if (SPLIT == 1) {
    for (int e = tid; e < O::NO; e += NT) partial[((long long)blockIdx.x * gridDim.y + blockIdx.y) * O::NO + e] = sM[e];
    return;
  }
  if (SPLIT == 0 || blockIdx.y == 0)
    for (int e = tid; e < O::NO; e += NT) kv_save[(long long)blockIdx.x * O::NO + e] = sM[e];
  const int l_begin = SPLIT == 2 ? blockIdx.y * chunk_len : 0, l_end = SPLIT == 2 ? min(L, l_begin + chunk_len) : L;
  for (int l0 = l_begin; l0 < l_end; l0 += RCH) {
    const int nr = min(RCH, l_end - l0);
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, c = e - r * D;
      sX[r * P + c] = elu1p(to_f32<T>(q[((long long)n * L + l0 + r) * q_ld + h * D + c]));
    }
    __syncthreads();
    if (tid < nr) {
      float den = eps;
#pragma unroll 8
      for (int i = 0; i < D; ++i) den = fmaf(sX[tid * P + i], sM[i * P + D], den);
      sZ[tid] = 1.f / den;
    }
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, j = e - r * D;
      float a = 0.f;
#pragma unroll 8
      for (int i = 0; i < D; ++i) a = fmaf(sX[r * P + i], sM[i * P + j], a);
      out[((long long)n * L + l0 + r) * out_ld + h * D + j] = from_f32<T>(a * sZ[r] * fS);
    }
  }
}

template <typename T, int D, int SPLIT, int NT>
__global__ __launch_bounds__(NT) void linattn_bwd2_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ k, int k_ld,
                                                           const T* __restrict__ v, int v_ld, const T* __restrict__ dout, int do_ld,
                                                           const float* __restrict__ kv_save, T* __restrict__ dq, int dq_ld,
                                                           T* __restrict__ dk, int dk_ld, T* __restrict__ dv, int dv_ld, int L, int S,
                                                           int heads, float eps, float* __restrict__ partial, int nchunks, int chunk_len) {
  using O = Outer<D, NT>;
  constexpr int RCH = NT == 64 ? 32 : 64, P = D + 1;
  __shared__ float sX[RCH * P], sY[RCH * P], sT[SPLIT == 2 ? 1 : RCH * P], sM[SPLIT == 2 ? 1 : O::NO], sG[O::NO], sZ[RCH], sE[RCH],
      sRed[O::G > 1 ? O::G * O::NO : 1];
  const int tid = threadIdx.x;
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const float fS = (float)S;
  if (SPLIT != 2)
    for (int e = tid; e < O::NO; e += NT) sM[e] = kv_save[(long long)blockIdx.x * O::NO + e];
  float acc[O::OPT];
#pragma unroll
  for (int u = 0; u < O::OPT; ++u) acc[u] = 0.f;
  __syncthreads();
  // ---- queries: dq, and the rows (Q_l, dA_l, e_l) of the dKV / dKsum reduction
  const int l_begin = SPLIT == 1 ? blockIdx.y * chunk_len : 0, l_end = SPLIT == 1 ? min(L, l_begin + chunk_len) : (SPLIT == 2 ? 0 : L);
  for (int l0 = l_begin; l0 < l_end; l0 += RCH) {
    const int nr = min(RCH, l_end - l0);
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, c = e - r * D;
      const long long row = (long long)n * L + l0 + r;
      sX[r * P + c] = elu1p(to_f32<T>(q[row * q_ld + h * D + c]));
      sY[r * P + c] = to_f32<T>(dout[row * do_ld + h * D + c]);
    }
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {           // T[r][i] = sum_j KV[i][j] dout[r][j]
      const int r = e / D, i = e - r * D;
      float a = 0.f;
#pragma unroll 8
      for (int j = 0; j < D; ++j) a = fmaf(sM[i * P + j], sY[r * P + j], a);
      sT[r * P + i] = a;
    }
    __syncthreads();
    if (tid < nr) {                                    // Z_l, e_l:  dout_l . A_l = Q_l . T_l
      float den = eps, dot = 0.f;
#pragma unroll 8
      for (int i = 0; i < D; ++i) {
        const float qv = sX[tid * P + i];
        den = fmaf(qv, sM[i * P + D], den);
        dot = fmaf(qv, sT[tid * P + i], dot);
      }
      const float z = 1.f / den;
      sZ[tid] = fS * z;
      sE[tid] = -z * z * fS * dot;
    }
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {           // dQ = S Z T + e Ksum, times elu'(q);  dA = dout S Z in place
      const int r = e / D, i = e - r * D;
      const long long row = (long long)n * L + l0 + r;
      const float d = sZ[r] * sT[r * P + i] + sE[r] * sM[i * P + D];
      dq[row * dq_ld + h * D + i] = from_f32<T>(d * elu1_grad(to_f32<T>(q[row * q_ld + h * D + i])));
      sY[r * P + i] *= sZ[r];
    }
    if (tid < nr) sY[tid * P + D] = sE[tid];
    __syncthreads();
    outer_accumulate<D, NT, P>(sX, sY, nr, acc);
  }
  __syncthreads();
  if (SPLIT == 2) {
    for (int e = tid; e < O::NO; e += NT) {
      float a = 0.f;
      for (int c = 0; c < nchunks; ++c) a += partial[((long long)blockIdx.x * nchunks + c) * O::NO + e];
      sG[e] = a;
    }
    __syncthreads();
  } else {
    outer_finish<D, NT>(acc, sRed, sG);          // sG = [dKV | dKsum]
  }
  if (SPLIT == 1) {
    for (int e = tid; e < O::NO; e += NT) partial[((long long)blockIdx.x * gridDim.y + blockIdx.y) * O::NO + e] = sG[e];
    return;
  }
  // ---- keys: dk, dv
  const int s_begin = SPLIT == 2 ? blockIdx.y * chunk_len : 0, s_end = SPLIT == 2 ? min(S, s_begin + chunk_len) : S;
  for (int s0 = s_begin; s0 < s_end; s0 += RCH) {
    const int nr = min(RCH, s_end - s0);
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, c = e - r * D;
      const long long row = (long long)n * S + s0 + r;
      sX[r * P + c] = elu1p(to_f32<T>(k[row * k_ld + h * D + c]));
      sY[r * P + c] = to_f32<T>(v[row * v_ld + h * D + c]) / fS;
    }
    __syncthreads();
    for (int e = tid; e < nr * D; e += NT) {
      const int r = e / D, i = e - r * D;
      const long long row = (long long)n * S + s0 + r;
      float a = sG[i * P + D];
#pragma unroll 8
      for (int j = 0; j < D; ++j) a = fmaf(sG[i * P + j], sY[r * P + j], a);
      dk[row * dk_ld + h * D + i] = from_f32<T>(a * elu1_grad(to_f32<T>(k[row * k_ld + h * D + i])));
      float b = 0.f;                                   // here the output index is j = i
#pragma unroll 8
      for (int ii = 0; ii < D; ++ii) b = fmaf(sX[r * P + ii], sG[ii * P + i], b);
      dv[row * dv_ld + h * D + i] = from_f32<T>(b / fS);
    }
  }
}

}  // namespace

extern "C" size_t cfp_linattn_state_bytes(int N, int heads, int d) { return (N > 0 && heads > 0 && d > 0) ? (size_t)N * heads * d * (d + 1) * sizeof(float) : 0; }

#define LA_CHECK(name)                                                                                                     \
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, name ": bad dtype");                                                            \
  CFP_REQUIRE(N > 0 && L > 0 && S > 0 && heads > 0 && (d == 4 || d == 8 || d == 16 || d == 32), CFP_ESHAPE, name ": head dim must be 4/8/16/32"); \
  CFP_REQUIRE((long long)N * heads < (1ll << 31), CFP_ESHAPE, name ": too many groups");                                   \
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);                                                                   \
  const dim3 grid((unsigned)((long long)N * heads))

// Few groups (global attention: N = batch, 64-128 workgroups for 256 CUs): split the token axes over blockIdx.y.
struct LaSplit { int c1, len1, c2, len2; };
static bool la_plan(long long groups, int R, int A, LaSplit* p) {      // R: tokens of the reduction phase, A: of the apply phase
  if (groups >= 512) return false;
  const int want = (int)((1024 + groups - 1) / groups);
  int c1 = std::min(want, cdiv(R, ACH));
  p->len1 = cdiv(cdiv(R, c1), ACH) * ACH;
  p->c1 = cdiv(R, p->len1);
  int c2 = std::min(want, cdiv(A, 256));
  p->len2 = cdiv(cdiv(A, c2), 256) * 256;
  p->c2 = cdiv(A, p->len2);
  return p->c1 > 1 || p->c2 > 1;
}

extern "C" size_t cfp_linattn_ws_bytes(int N, int L, int S, int heads, int d) {
  if (N <= 0 || L <= 0 || S <= 0 || heads <= 0 || d <= 0) return 0;
  const long long groups = (long long)N * heads;
  LaSplit f, b;
  size_t need = 0;
  if (la_plan(groups, S, L, &f)) need = std::max(need, (size_t)groups * f.c1 * d * (d + 1) * sizeof(float));
  if (la_plan(groups, L, S, &b)) need = std::max(need, (size_t)groups * b.c1 * d * (d + 1) * sizeof(float));
  return need;
}

extern "C" int cfp_linattn_fwd(const void* q, int q_ld, const void* k, int k_ld, const void* v, int v_ld, void* out, int out_ld, float* state,
                               int N, int L, int S, int heads, int d, float eps, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(q && k && v && out && state, CFP_EINVAL, "cfp_linattn_fwd: null pointer");
  LA_CHECK("cfp_linattn_fwd");
  const int C = heads * d;
  CFP_REQUIRE(q_ld >= C && k_ld >= C && v_ld >= C && out_ld >= C, CFP_ESHAPE, "cfp_linattn_fwd: pitch smaller than heads*d");
  LaSplit sp;
  const bool split = la_plan((long long)N * heads, S, L, &sp) && ws &&
                     ws_bytes >= (size_t)N * heads * sp.c1 * d * (d + 1) * sizeof(float);
  float* partial = reinterpret_cast<float*>(ws);
  const bool wave = L <= 64 && S <= 64;      // zone / window groups of a few tokens: one wave per (group, head)
#define FARGS(T) (const T*)q, q_ld, (const T*)k, k_ld, (const T*)v, v_ld, (T*)out, out_ld, state, L, S, heads, eps
#define L1(T, DD, KERN) do {                                                                                                        \
    if (!split && wave) hipLaunchKernelGGL((KERN<T, DD, 0, 64>), grid, dim3(64), 0, s, FARGS(T), nullptr, 0, 0);                     \
    else if (!split) hipLaunchKernelGGL((KERN<T, DD, 0, 256>), grid, dim3(256), 0, s, FARGS(T), nullptr, 0, 0);                      \
    else {                                                                                                                          \
      hipLaunchKernelGGL((KERN<T, DD, 1, 256>), dim3(grid.x, sp.c1), dim3(256), 0, s, FARGS(T), partial, sp.c1, sp.len1);            \
      hipLaunchKernelGGL((KERN<T, DD, 2, 256>), dim3(grid.x, sp.c2), dim3(256), 0, s, FARGS(T), partial, sp.c1, sp.len2);            \
    }                                                                                                                               \
  } while (0)
#define LD(T) do { if (d == 4) L1(T, 4, linattn_fwd_kernel); else if (d == 8) L1(T, 8, linattn_fwd2_kernel);                         \
                   else if (d == 16) L1(T, 16, linattn_fwd2_kernel); else L1(T, 32, linattn_fwd2_kernel); } while (0)
  if (dtype == CFP_BF16) LD(bf16_t); else if (dtype == CFP_F16) LD(f16_t); else LD(float);
#undef LD
#undef L1
#undef FARGS
  return cfp_check_launch("cfp_linattn_fwd");
}

extern "C" int cfp_linattn_bwd(const void* q, int q_ld, const void* k, int k_ld, const void* v, int v_ld, const void* dout, int do_ld,
                               const float* state, void* dq, int dq_ld, void* dk, int dk_ld, void* dv, int dv_ld, int N, int L, int S,
                               int heads, int d, float eps, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(q && k && v && dout && state && dq && dk && dv, CFP_EINVAL, "cfp_linattn_bwd: null pointer");
  LA_CHECK("cfp_linattn_bwd");
  const int C = heads * d;
  CFP_REQUIRE(q_ld >= C && k_ld >= C && v_ld >= C && do_ld >= C && dq_ld >= C && dk_ld >= C && dv_ld >= C, CFP_ESHAPE,
              "cfp_linattn_bwd: pitch smaller than heads*d");
  LaSplit sp;
  const bool split = la_plan((long long)N * heads, L, S, &sp) && ws &&
                     ws_bytes >= (size_t)N * heads * sp.c1 * d * (d + 1) * sizeof(float);
  float* partial = reinterpret_cast<float*>(ws);
  const bool wave = L <= 64 && S <= 64;
#define BARGS(T) (const T*)q, q_ld, (const T*)k, k_ld, (const T*)v, v_ld, (const T*)dout, do_ld, state, (T*)dq, dq_ld, (T*)dk, dk_ld, (T*)dv, \
                 dv_ld, L, S, heads, eps
#define L1(T, DD, KERN) do {                                                                                                        \
    if (!split && wave) hipLaunchKernelGGL((KERN<T, DD, 0, 64>), grid, dim3(64), 0, s, BARGS(T), nullptr, 0, 0);                     \
    else if (!split) hipLaunchKernelGGL((KERN<T, DD, 0, 256>), grid, dim3(256), 0, s, BARGS(T), nullptr, 0, 0);                      \
    else {                                                                                                                          \
      hipLaunchKernelGGL((KERN<T, DD, 1, 256>), dim3(grid.x, sp.c1), dim3(256), 0, s, BARGS(T), partial, sp.c1, sp.len1);            \
      hipLaunchKernelGGL((KERN<T, DD, 2, 256>), dim3(grid.x, sp.c2), dim3(256), 0, s, BARGS(T), partial, sp.c1, sp.len2);            \
    }                                                                                                                               \
  } while (0)
#define LD(T) do { if (d == 4) L1(T, 4, linattn_bwd_kernel); else if (d == 8) L1(T, 8, linattn_bwd2_kernel);                         \
                   else if (d == 16) L1(T, 16, linattn_bwd2_kernel); else L1(T, 32, linattn_bwd2_kernel); } while (0)
  if (dtype == CFP_BF16) LD(bf16_t); else if (dtype == CFP_F16) LD(f16_t); else LD(float);
#undef LD
#undef L1
#undef BARGS
  return cfp_check_launch("cfp_linattn_bwd");
}
