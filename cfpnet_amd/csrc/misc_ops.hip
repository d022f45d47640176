// Bandwidth-bound helper kernels: reductions over H*W, the SE gate, LayerNorm, bilinear
// resampling (with the zone crop / scatter folded into its addressing), broadcast adds, layout
// conversion.  All of them move 16 bytes per lane per access along the channel axis.
#include "common.h"

namespace {

// ---- channel sums: partial[b][s][c] = sum over the s-th slice of the HW rows ------------
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ in, int in_ld, float* __restrict__ partial,
                                                          int HW, int C, int nsplit) {
  constexpr int VE = Vec<T>::N;
  __shared__ float red[256 * 8];
  const int b = blockIdx.y, s = blockIdx.x;
  const int CV = C / VE;
  const int rows_per = (HW + nsplit - 1) / nsplit;
  const int r_begin = s * rows_per;
  const int r_end = min(HW, r_begin + rows_per);
  const int lanes_r = 256 / CV > 0 ? 256 / CV : 1;   // row lanes when CV <= 256
  // channel vectors are walked in chunks of up to 256
  for (int cv0 = 0; cv0 < CV; cv0 += 256) {
    const int ncv = min(256, CV - cv0);
    const int rl = 256 / ncv;                 // row lanes for this chunk
    const int cv = threadIdx.x % ncv, rr = threadIdx.x / ncv;
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    if (rr < rl) {
      int r = r_begin + rr;
      for (; r + 3 * rl < r_end; r += 4 * rl) {          // four rows in flight (one per iteration = one memory round trip per row)
        float v[4][VE];
#pragma unroll
        for (int u = 0; u < 4; ++u) Vec<T>::load(in + ((long long)b * HW + r + u * rl) * in_ld + (cv0 + cv) * VE, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < VE; ++e) acc[e] += v[u][e];
      }
      for (; r < r_end; r += rl) {
        float v[VE];
        Vec<T>::load(in + ((long long)b * HW + r) * in_ld + (cv0 + cv) * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) red[threadIdx.x * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < ncv) {
      float t[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) t[e] = 0.f;
      for (int j = 0; j < rl; ++j)
#pragma unroll
        for (int e = 0; e < VE; ++e) t[e] += red[(j * ncv + threadIdx.x) * 8 + e];
      float* dst = partial + ((long long)b * nsplit + s) * C + (cv0 + threadIdx.x) * VE;
#pragma unroll
      for (int e = 0; e < VE; ++e) dst[e] = t[e];
    }
    __syncthreads();
  }
  (void)lanes_r;
}

// ---- SE block --------------------------------------------------------------------------
// se_hidden: mean (from the partial sums) -> FC(C->R) + bias -> SiLU.  One wave per hidden unit;
// lanes walk the channels, so weight rows are read as coalesced 256-byte runs and the dot product
// ends in a wavefront shuffle reduction.
__global__ __launch_bounds__(256) void se_hidden_kernel(const float* __restrict__ partial, int nsplit, float inv_hw,
                                                        const float* __restrict__ wr, const float* __restrict__ br,
                                                        float* __restrict__ hidden, int C, int R) {
  // grid (ceil(R/4), B): every wave of a workgroup produces one hidden unit, so the 4 x ceil(R/4) x B
  // dot products run in parallel instead of one after the other (they are latency-, not bandwidth-bound).
  // All loads are 16-byte and issued in groups of four before the first use.
  extern __shared__ __attribute__((aligned(16))) float mean[];   // [C]
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  for (int c4 = threadIdx.x; c4 < C4; c4 += 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 3 < nsplit; j += 4) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j) * C + c4 * 4);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j + 1) * C + c4 * 4);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j + 2) * C + c4 * 4);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j + 3) * C + c4 * 4);
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; j < nsplit; ++j) s += *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(mean + c4 * 4) = s * inv_hw;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float* wrow = wr + (long long)r * C;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int c4 = lane;
  for (; c4 + 192 < C4; c4 += 256) {     // four independent 16-byte loads in flight per lane
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + c4 * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrow + (c4 + 64) * 4);
    const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrow + (c4 + 128) * 4);
    const f32x4 w3 = *reinterpret_cast<const f32x4*>(wrow + (c4 + 192) * 4);
    acc += w0 * *reinterpret_cast<const f32x4*>(mean + c4 * 4);
    acc += w1 * *reinterpret_cast<const f32x4*>(mean + (c4 + 64) * 4);
    acc += w2 * *reinterpret_cast<const f32x4*>(mean + (c4 + 128) * 4);
    acc += w3 * *reinterpret_cast<const f32x4*>(mean + (c4 + 192) * 4);
  }
  for (; c4 < C4; c4 += 64) acc += *reinterpret_cast<const f32x4*>(wrow + c4 * 4) * *reinterpret_cast<const f32x4*>(mean + c4 * 4);
  const float s = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
  if (lane == 0) hidden[(long long)b * R + r] = act_c<CFP_ACT_SILU>(s + br[r]);
}

// se_scale: x[b, hw, c] *= sigmoid(hidden[b] . we_t[:, c] + be[c]).  A thread keeps one 16-byte
// channel vector, computes its gate once (R x VE FMAs, expand weights stored [R][C] so the
// reads are coalesced) and then walks rows.
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(T* __restrict__ x, int ld, const float* __restrict__ hidden,
                                                       const float* __restrict__ we_t, const float* __restrict__ be,
                                                       int HW, int C, int R, int row_lanes) {
  constexpr int VE = Vec<T>::N;
  extern __shared__ float hid[];    // [R]
  const int b = blockIdx.y;
  for (int r = threadIdx.x; r < R; r += 256) hid[r] = hidden[(long long)b * R + r];
  __syncthreads();
  const int CV = C / VE;
  const int items = CV * row_lanes;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int cv = it % CV, rl = it / CV;
    float g[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) g[e] = be[cv * VE + e];
    for (int r = 0; r < R; ++r) {
      const float h = hid[r];
      const float* wp = we_t + (long long)r * C + cv * VE;
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = fmaf(h, wp[e], g[e]);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) g[e] = 1.f / (1.f + __expf(-g[e]));
    for (int row = rl; row < HW; row += row_lanes) {
      T* p = x + ((long long)b * HW + row) * ld + cv * VE;
      float v[VE];
      Vec<T>::load(p, v);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] *= g[e];
      Vec<T>::store(p, v);
    }
  }
}

// se_fold: per-image project weights  wout[b][n][c] = w[n][c] * sigmoid(hidden[b] . we_t[:, c] + be[c]).
// Same thread layout as se_scale (one 16-byte channel vector per thread, gate computed once), but
// the rows walked are the Cout rows of the project weight instead of the H*W rows of the activation.
// Loads are issued four rows at a time (the loop bodies carry no dependence through memory).
template <typename T>
__global__ __launch_bounds__(256) void se_fold_kernel(const T* __restrict__ w, T* __restrict__ wout, const float* __restrict__ hidden,
                                                      const float* __restrict__ we_t, const float* __restrict__ be, int Cout, int C,
                                                      int R, int row_lanes) {
  constexpr int VE = Vec<T>::N;
  extern __shared__ float hid[];    // [R]
  const int b = blockIdx.y;
  for (int r = threadIdx.x; r < R; r += 256) hid[r] = hidden[(long long)b * R + r];
  __syncthreads();
  const int CV = C / VE;
  const int items = CV * row_lanes;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int cv = it % CV, rl = it / CV;
    float g[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) g[e] = be[cv * VE + e];
    int r = 0;
    for (; r + 3 < R; r += 4) {
      float wp[4][VE];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < VE; e += 4) Vec<float>::load(we_t + (long long)(r + k) * C + cv * VE + e, wp[k] + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float h = hid[r + k];
#pragma unroll
        for (int e = 0; e < VE; ++e) g[e] = fmaf(h, wp[k][e], g[e]);
      }
    }
    for (; r < R; ++r) {
      const float h = hid[r];
      const float* wp = we_t + (long long)r * C + cv * VE;
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = fmaf(h, wp[e], g[e]);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) g[e] = 1.f / (1.f + __expf(-g[e]));
    int n = rl;
    for (; n + 3 * row_lanes < Cout; n += 4 * row_lanes) {
      float v[4][VE];
#pragma unroll
      for (int k = 0; k < 4; ++k) Vec<T>::load(w + (long long)(n + k * row_lanes) * C + cv * VE, v[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < VE; ++e) v[k][e] *= g[e];
        Vec<T>::store(wout + ((long long)b * Cout + n + k * row_lanes) * C + cv * VE, v[k]);
      }
    }
    for (; n < Cout; n += row_lanes) {
      float v[VE];
      Vec<T>::load(w + (long long)n * C + cv * VE, v);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] *= g[e];
      Vec<T>::store(wout + ((long long)b * Cout + n) * C + cv * VE, v);
    }
  }
}

// se_gate_fold: the whole squeeze-excite tail of an inverted-residual block in ONE launch, written for latency
// (every load batch of a phase is issued before its first use; 16 waves share the serial phases):
//   phase 0  mean[c]   = sum_s partial[b][s][c] / HW                                (all C channels, LDS)
//   phase 1  hidden[r] = silu(mean . w_reduce[r] + b_reduce[r])       wave w takes r = w, w+16, ...   (LDS)
//   phase 2  gate[c]   = sigmoid(hidden . w_expand_t[:, c] + b_expand[c])   for the 256 channels of this slab
//   phase 3  wout[b][n][c] = w[n][c] * gate[c]                         for every project row n, this slab
// grid (ceil(C/256), B): every workgroup recomputes mean/hidden of its image (R x C MACs: cheaper than a
// kernel boundary).  Replaces cfp_se_hidden + cfp_se_fold on the bf16/f32 hot path.
template <typename T>
__global__ __launch_bounds__(1024) void se_gate_fold_kernel(const float* __restrict__ partial, int nsplit, float inv_hw,
                                                            const float* __restrict__ wr, const float* __restrict__ br,
                                                            const float* __restrict__ we_t, const float* __restrict__ be,
                                                            const T* __restrict__ w, T* __restrict__ wout, int Cout, int C, int R, int x3) {
  constexpr int VE = Vec<T>::N;
  const int nk3 = (C + 31) >> 5;            // x3 (float32 weights only): wout is the pre-split f16x3 operand (cfp_pack_w_x3 layout)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* mean = sm;                         // [C]
  float* hid = sm + C;                      // [R] (padded to 64)
  float* gpart = hid + ((R + 63) & ~63);    // [4][256]
  float* gate = gpart + 4 * 256;            // [256]
  const int b = blockIdx.y, slab = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C4 = C >> 2;
  // ---- phase 0
  for (int c4 = tid; c4 < C4; c4 += 1024) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 3 < nsplit; j += 4) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j) * C + c4 * 4);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j + 1) * C + c4 * 4);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j + 2) * C + c4 * 4);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j + 3) * C + c4 * 4);
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; j < nsplit; ++j) s += *reinterpret_cast<const f32x4*>(partial + ((long long)b * nsplit + j) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(mean + c4 * 4) = s * inv_hw;
  }
  __syncthreads();
  // ---- phase 1: 16 waves, up to 4 hidden units each, all weight loads of a wave in flight together
  {
    constexpr int MAXL = 8;                  // C <= 8 * 256 floats per lane-stride pass; R <= 64 = 16 waves x 4
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += 2) {      // two hidden units per wave at a time (64 VGPRs of weights in flight)
      f32x4 wv[2][MAXL];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = wave + 16 * (k0 + k);
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
          const int c4 = lane + 64 * l;
          wv[k][l] = (r < R && c4 < C4) ? *reinterpret_cast<const f32x4*>(wr + (long long)r * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = wave + 16 * (k0 + k);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
          const int c4 = lane + 64 * l;
          if (c4 < C4) acc += wv[k][l] * *reinterpret_cast<const f32x4*>(mean + c4 * 4);
        }
        const float s = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (lane == 0 && r < R) hid[r] = act_c<CFP_ACT_SILU>(s + br[r]);
      }
    }
  }
  __syncthreads();
  // ---- phase 2: 256 channels x 4 quarter-ranges of r
  {
    const int cl = tid & 255, qr = tid >> 8;
    const int c = slab * 256 + cl;
    const int rq = (R + 3) >> 2;
    const int r0 = qr * rq, r1 = min(R, r0 + rq);
    float s = 0.f;
    if (c < C) {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = (r0 + u < r1) ? we_t[(long long)(r0 + u) * C + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) s = fmaf((r0 + u < r1) ? hid[r0 + u] : 0.f, t[u], s);
      for (int r = r0 + 16; r < r1; ++r) s = fmaf(hid[r], we_t[(long long)r * C + c], s);
    }
    gpart[qr * 256 + cl] = s;
  }
  __syncthreads();
  if (tid < 256) {
    const int c = slab * 256 + tid;
    const float z = (gpart[tid] + gpart[256 + tid]) + (gpart[512 + tid] + gpart[768 + tid]) + (c < C ? be[c] : 0.f);
    gate[tid] = act_c<CFP_ACT_SIGMOID>(z);
  }
  __syncthreads();
  // ---- phase 3
  {
    constexpr int VPS = 256 / VE;           // vectors per slab row
    const int vl = tid % VPS, rl = tid / VPS, nrl = 1024 / VPS;
    const int cv = slab * VPS + vl;
    if (cv * VE < C) {
      float g[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = gate[vl * VE + e];
      int n = rl;
      for (; n + 3 * nrl < Cout; n += 4 * nrl) {
        float v[4][VE];
#pragma unroll
        for (int k = 0; k < 4; ++k) Vec<T>::load(w + (long long)(n + k * nrl) * C + cv * VE, v[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int e = 0; e < VE; ++e) v[k][e] *= g[e];
          if constexpr (VE == 4) { if (x3) { x3_store4(reinterpret_cast<f16_t*>(wout), (long long)b * Cout + n + k * nrl, nk3, cv * 4, v[k]); continue; } }
          Vec<T>::store(wout + ((long long)b * Cout + n + k * nrl) * C + cv * VE, v[k]);
        }
      }
      for (; n < Cout; n += nrl) {
        float v[VE];
        Vec<T>::load(w + (long long)n * C + cv * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] *= g[e];
        if constexpr (VE == 4) { if (x3) { x3_store4(reinterpret_cast<f16_t*>(wout), (long long)b * Cout + n, nk3, cv * 4, v); continue; } }
        Vec<T>::store(wout + ((long long)b * Cout + n) * C + cv * VE, v);
      }
    }
  }
}

// se_gate_fold2 (round 3): the squeeze-excite tail when the depthwise kernel has already taken the reduce FC's dot products.
// The reduce layer is linear in the channel sums, so every workgroup of dw3x3_stream_kernel (image b, 64-channel block, row
// range) contributes  hpart[b][k][r] = sum_{c in block} w_reduce[r][c] * (sum of its output pixels of channel c)  and what is
// left here is
//   phase A  hidden[r] = silu(inv_hw * sum_k hpart[b][k][r] + b_reduce[r])        K = blocks x ranges partials, added in k order
//   phase B  gate[c]   = sigmoid(hidden . w_expand_t[:, c] + b_expand[c])          the 128 channels of this slab
//   phase C  wout[b][n][c] = w[n][c] * gate[c]                                     this workgroup's share of the project rows
// grid (ceil(C/128), B, Z): Z splits the project rows so that the launch fills the chip; every load batch of a phase is in flight
// together.  se_gate_fold_kernel above needed the R x C reduce weights (323 KB at C = 1392) in EVERY workgroup and ran 48 workgroups
// of 1024 threads: 15.8 us per launch, 24 launches per forward (profiles/r2n_bench_b8_kernel_stats.txt).
template <typename T>
__global__ __launch_bounds__(256) void se_gate_fold2_kernel(const float* __restrict__ hpart, int K, float inv_hw, const float* __restrict__ br,
                                                            const float* __restrict__ we_t, const float* __restrict__ be,
                                                            const float* __restrict__ w, T* __restrict__ wout, int Cout, int C, int R, int rows_per_z, int x3) {
  constexpr int VE = Vec<T>::N;
  const int nk3 = (C + 31) >> 5;            // x3 (float32 only): wout is the pre-split f16x3 operand (cfp_pack_w_x3 layout)
  __shared__ float hsum[4][64];
  __shared__ float hid[64];
  __shared__ float gpart[2][128];
  __shared__ float gate[128];
  const int b = blockIdx.y, slab = blockIdx.x, z = blockIdx.z;
  const int tid = threadIdx.x;
  // ---- phase A: thread (r = tid & 63, quarter kq = tid >> 6) adds its partials k = kq, kq + 4, ... in order (16 loads in flight)
  {
    const int r = tid & 63, kq = tid >> 6;
    float s = 0.f;
    if (r < R) {
      const float* hp = hpart + (long long)b * K * R + r;
      for (int k0 = kq; k0 < K; k0 += 64) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = (k0 + 4 * u < K) ? hp[(long long)(k0 + 4 * u) * R] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += t[u];
      }
    }
    hsum[kq][r] = s;
  }
  __syncthreads();
  if (tid < 64) hid[tid] = tid < R ? act_c<CFP_ACT_SILU>(((hsum[0][tid] + hsum[1][tid]) + (hsum[2][tid] + hsum[3][tid])) * inv_hw + br[tid]) : 0.f;
  __syncthreads();
  // ---- phase B: channel cl = tid & 127, half of the hidden units rh = tid >> 7 (up to 32 loads in flight)
  {
    const int cl = tid & 127, rh = tid >> 7;
    const int c = slab * 128 + cl;
    float s = 0.f;
    if (c < C) {
      float t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = (rh * 32 + u < R) ? we_t[(long long)(rh * 32 + u) * C + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 32; ++u) s = fmaf(hid[rh * 32 + u], t[u], s);
    }
    gpart[rh][cl] = s;
  }
  __syncthreads();
  if (tid < 128) {
    const int c = slab * 128 + tid;
    gate[tid] = act_c<CFP_ACT_SIGMOID>(gpart[0][tid] + gpart[1][tid] + (c < C ? be[c] : 0.f));
  }
  __syncthreads();
  // ---- phase C: vector lane vl (128 / VE per row), row lane rl; 4 rows in flight per thread
  {
    constexpr int VPS = 128 / VE;
    constexpr int NRL = 256 / VPS;
    const int vl = tid % VPS, rl = tid / VPS;
    const int cv = slab * VPS + vl;
    if (cv * VE < C) {
      float g[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = gate[vl * VE + e];
      const int n1 = min(Cout, (z + 1) * rows_per_z);
      for (int n = z * rows_per_z + rl; n < n1; n += 4 * NRL) {
        float v[4][VE];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (n + k * NRL < n1) {      // FLOAT32 project weights: the folded weight is rounded to the storage type once, not twice
#pragma unroll
            for (int e = 0; e < VE; e += 4) Vec<float>::load(w + (long long)(n + k * NRL) * C + cv * VE + e, v[k] + e);
          }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (n + k * NRL < n1) {
#pragma unroll
            for (int e = 0; e < VE; ++e) v[k][e] *= g[e];
            if constexpr (VE == 4) { if (x3) { x3_store4(reinterpret_cast<f16_t*>(wout), (long long)b * Cout + n + k * NRL, nk3, cv * 4, v[k]); continue; } }
            Vec<T>::store(wout + ((long long)b * Cout + n + k * NRL) * C + cv * VE, v[k]);
          }
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(T* __restrict__ x, int ld, const float* __restrict__ gate,
                                                             int HW, int C, long long total) {
  constexpr int VE = Vec<T>::N;
  const int CV = C / VE;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int cv = (int)(i % CV);
    long long row = i / CV;
    int b = (int)(row / HW);
    float v[VE];
    T* p = x + row * ld + cv * VE;
    Vec<T>::load(p, v);
    const float* g = gate + (long long)b * C + cv * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] *= g[e];
    Vec<T>::store(p, v);
  }
}

// ---- LayerNorm: LPR lanes per row, each lane one 16-byte vector -------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ in, int in_ld, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const T* __restrict__ res, int res_ld, T* __restrict__ out,
                                                        int out_ld, int rows, int C) {
  constexpr int VE = Vec<T>::N;
  const int LPR = C / VE;               // power of two, <= 64
  const int rows_per_block = 256 / LPR;
  const int lr = threadIdx.x % LPR;
  const int rb = threadIdx.x / LPR;
  float g[VE], bt[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { g[e] = gamma[lr * VE + e]; bt[e] = beta[lr * VE + e]; }
  const float inv_c = 1.f / (float)C;
  for (long long row0 = (long long)blockIdx.x * rows_per_block; row0 < rows; row0 += (long long)gridDim.x * rows_per_block) {
    long long row = row0 + rb;
    bool ok = row < rows;
    float v[VE];
    if (ok) Vec<T>::load(in + row * in_ld + lr * VE, v);
    else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) s += v[e];
    for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) { float d = v[e] - mean; q = fmaf(d, d, q); }
    for (int o = LPR >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * inv_c + eps);
    if (ok) {
      float o_[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o_[e] = (v[e] - mean) * rstd * g[e] + bt[e];
      if (res) {
        float r_[VE];
        Vec<T>::load(res + row * res_ld + lr * VE, r_);
#pragma unroll
        for (int e = 0; e < VE; ++e) o_[e] += r_[e];
      }
      Vec<T>::store(out + row * out_ld + lr * VE, o_);
    }
  }
}

// ---- bilinear resampling, align_corners=True ---------------------------------------------
struct ResizeP {
  const void* src; void* dst; const uint8_t* zone_valid;
  int src_ld, Hs, Ws, sy0, sx0, sh, sw;
  int dst_ld, Hd, Wd, dy0, dx0, dh, dw;
  int zn, p1, p2, accumulate, B, C;
  float scale_y, scale_x;
  FastDiv fcv, fdw, fdh;      // element index -> (b, dy, dx, channel vector) without 64-bit divisions (the first version's four cost more than the taps)
};

template <typename T>
__global__ __launch_bounds__(256) void resize_kernel(ResizeP p) {
  constexpr int VE = Vec<T>::N;
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  T* __restrict__ dst = reinterpret_cast<T*>(p.dst);
  const unsigned total = (unsigned)p.B * p.dh * p.dw * p.fcv.d;      // < 2^31: host check
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned t, cvu, q, dxu, bu, dyu;
    fd_rowcol(i, p.fcv, t, cvu);
    fd_rowcol(t, p.fdw, q, dxu);
    fd_rowcol(q, p.fdh, bu, dyu);
    const int cv = (int)cvu, dx = (int)dxu, dy = (int)dyu, b = (int)bu;
    const int oy = p.dy0 + dy, ox = p.dx0 + dx;
    if (oy < 0 || oy >= p.Hd || ox < 0 || ox >= p.Wd) continue;
    // source coordinate inside the source rectangle (torch: src = scale * dst_index)
    const float fy = p.scale_y * (float)dy, fx = p.scale_x * (float)dx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.sh - 1 ? 1 : 0), x1 = x0 + (x0 < p.sw - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    float tap[4][VE];
    const int ys[2] = {y0, y1}, xs[2] = {x0, x1};
    // validity first (the four zone flags are requested together), then four UNCONDITIONAL loads from clamped addresses and a select on
    // the values: `if (ok) load` compiled to one branch + memory round trip per tap, up to eight in a row with the zone flags
    bool ok[4]; unsigned char zv[4]; const T* sp[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ry = ys[a], rx = xs[c];              // rectangle coordinates
        const int gy = p.sy0 + ry, gx = p.sx0 + rx;    // source-map coordinates
        ok[a * 2 + c] = gy >= 0 && gy < p.Hs && gx >= 0 && gx < p.Ws;
        const int gyc = min(max(gy, 0), p.Hs - 1), gxc = min(max(gx, 0), p.Ws - 1);
        sp[a * 2 + c] = src + ((long long)(b * p.Hs + gyc) * p.Ws + gxc) * p.src_ld + cv * VE;
        zv[a * 2 + c] = 1;
        if (p.zone_valid)                              // uniform
          zv[a * 2 + c] = p.zone_valid[(long long)b * p.zn * p.zn + min(max(ry / p.p1, 0), p.zn - 1) * p.zn + min(max(rx / p.p2, 0), p.zn - 1)];
      }
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) Vec<T>::load(sp[t4], tap[t4]);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const bool keep = ok[t4] && zv[t4] != 0;
#pragma unroll
      for (int e = 0; e < VE; ++e) tap[t4][e] = keep ? tap[t4][e] : 0.f;
    }
    float o[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e)
      o[e] = ly0 * (lx0 * tap[0][e] + lx1 * tap[1][e]) + ly1 * (lx0 * tap[2][e] + lx1 * tap[3][e]);
    T* dp = dst + ((long long)(b * p.Hd + oy) * p.Wd + ox) * p.dst_ld + cv * VE;
    if (p.accumulate) {
      float d[VE];
      Vec<T>::load(dp, d);
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] += d[e];
    }
    Vec<T>::store(dp, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_rowtable_kernel(const T* __restrict__ in, int in_ld, const float* __restrict__ table,
                                                           T* __restrict__ out, int out_ld, long long rows, int C, int H, int W,
                                                           int Wt, int oy, int ox, const int* __restrict__ dev_off, int Ht,
                                                           FastDiv fcv, FastDiv fw, FastDiv fh) {
  constexpr int VE = Vec<T>::N;
  const unsigned total = (unsigned)(rows * fcv.d);      // < 2^31 (host check): 32-bit indices and magic-number divisions -- the first
  if (dev_off) {          // window origin read from device memory (a captured graph replays with a new random window), clamped to the table
    oy = min(max(dev_off[0], 0), Ht - H);
    ox = min(max(dev_off[1], 0), Wt - W);
  }                                                      // version's five 64-bit divisions per element made this 10 MB pass cost 12 us
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned r, cv, q, x, b, y;
    fd_rowcol(i, fcv, r, cv);
    fd_rowcol(r, fw, q, x);
    fd_rowcol(q, fh, b, y);
    const long long trow = (long long)(y + oy) * Wt + x + ox;
    float v[VE];
    Vec<T>::load(in + (long long)r * in_ld + cv * VE, v);
    const float* tp = table + trow * C + cv * VE;
    if constexpr (VE == 8) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += t0[e]; v[4 + e] += t1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += tp[e];
    }
    Vec<T>::store(out + (long long)r * out_ld + cv * VE, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ in, int in_ld, T* __restrict__ out, int out_ld,
                                                        long long rows, int C, FastDiv fcv) {
  constexpr int VE = Vec<T>::N;
  const unsigned total = (unsigned)(rows * fcv.d);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned ru, cvu;
    fd_rowcol(i, fcv, ru, cvu);
    const long long r = ru;
    const int cv = (int)cvu;
    *reinterpret_cast<u32x4*>(out + r * out_ld + cv * VE) = *reinterpret_cast<const u32x4*>(in + r * in_ld + cv * VE);
  }
}

// two independent strided row copies in one launch (blockIdx.y picks one): a channel concatenation, or its backward split
struct Copy2P { const void* in[2]; void* out[2]; int in_ld[2], out_ld[2]; FastDiv fcv[2]; };
template <typename T>
__global__ __launch_bounds__(256) void copy_rows2_kernel(Copy2P p, long long rows) {
  constexpr int VE = Vec<T>::N;
  const int k = blockIdx.y;
  const T* __restrict__ in = reinterpret_cast<const T*>(p.in[k]);
  T* __restrict__ out = reinterpret_cast<T*>(p.out[k]);
  const int in_ld = p.in_ld[k], out_ld = p.out_ld[k];
  const FastDiv fcv = p.fcv[k];
  const unsigned total = (unsigned)(rows * fcv.d);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned ru, cvu;
    fd_rowcol(i, fcv, ru, cvu);
    *reinterpret_cast<u32x4*>(out + (long long)ru * out_ld + cvu * VE) = *reinterpret_cast<const u32x4*>(in + (long long)ru * in_ld + cvu * VE);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rgb_to_nhwc8_kernel(const float* __restrict__ rgb, T* __restrict__ out, int HW, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = (unsigned long long)i / (unsigned)HW, hw = i - b * HW;      // one division (images x pixels < 2^32 in every caller would allow 32 bits; kept general)
    const float* p = rgb + b * 3 * HW + hw;
    float v[8] = {p[0], p[HW], p[2 * (long long)HW], 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) Vec<T>::store(out + i * 8, v);
    else { Vec<float>::store(reinterpret_cast<float*>(out) + i * 8, v); Vec<float>::store(reinterpret_cast<float*>(out) + i * 8 + 4, v + 4); }
  }
}

// 16-bit storage only: channels 0-2 = the image rounded to the storage type, channels 3-5 = what that rounding lost (x - hi, itself
// rounded), 6-7 = 0.  A stem whose weight rows repeat the three real channels in slots 3-5 then sums w * (hi + lo) in its float32
// accumulators: the network sees the float32 input the reference sees (src/models/encoder.py:71-73 feeds float32 RGB) for free --
// the K axis of the stem GEMM is padded to 8 channel slots per tap anyway.
template <typename T>
__global__ __launch_bounds__(256) void rgb_to_nhwc8_hilo_kernel(const float* __restrict__ rgb, T* __restrict__ out, int HW, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = (unsigned long long)i / (unsigned)HW, hw = i - b * HW;      // one division (images x pixels < 2^32 in every caller would allow 32 bits; kept general)
    const float* p = rgb + b * 3 * HW + hw;
    float v[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = p[c * (long long)HW];
      const float hi = to_f32<T>(from_f32<T>(x));
      v[c] = hi;
      v[3 + c] = x - hi;
    }
    v[6] = 0.f; v[7] = 0.f;
    Vec<T>::store(out + i * 8, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scalar_to_rows8_kernel(const float* __restrict__ in, T* __restrict__ out, long long rows) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows; i += (long long)gridDim.x * 256) {
    float v[8] = {in[i], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) Vec<T>::store(out + i * 8, v);
    else { Vec<float>::store(reinterpret_cast<float*>(out) + i * 8, v); Vec<float>::store(reinterpret_cast<float*>(out) + i * 8 + 4, v + 4); }
  }
}

inline int ew_blocks(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define CHECK_DTYPE(name) CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, name ": bad dtype")

extern "C" int cfp_channel_sum(const void* in, int in_ld, float* partial, int B, int HW, int C, int nsplit, int dtype,
                               cfp_stream_t stream) {
  CHECK_DTYPE("cfp_channel_sum");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(in && partial && aligned16(in), CFP_EINVAL, "cfp_channel_sum: bad pointer");
  CFP_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0 && in_ld % ve == 0 && in_ld >= C && nsplit > 0 && nsplit <= 65535,
              CFP_ESHAPE, "cfp_channel_sum: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(nsplit, B), dim3(256), 0, s, (const bf16_t*)in, in_ld, partial, HW, C, nsplit);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(channel_sum_kernel<f16_t>, dim3(nsplit, B), dim3(256), 0, s, (const f16_t*)in, in_ld, partial, HW, C, nsplit);
  else
    hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(nsplit, B), dim3(256), 0, s, (const float*)in, in_ld, partial, HW, C, nsplit);
  return cfp_check_launch("cfp_channel_sum");
}

extern "C" int cfp_se_hidden(const float* partial, int nsplit, float inv_hw, const float* w_reduce, const float* b_reduce,
                             float* hidden, int B, int C, int R, cfp_stream_t stream) {
  CFP_REQUIRE(partial && w_reduce && b_reduce && hidden, CFP_EINVAL, "cfp_se_hidden: null pointer");
  CFP_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && R > 0 && nsplit > 0 && (size_t)C * 4 <= 64 * 1024, CFP_ESHAPE, "cfp_se_hidden: bad shape");
  CFP_REQUIRE(aligned16(partial) && aligned16(w_reduce), CFP_EINVAL, "cfp_se_hidden: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(se_hidden_kernel, dim3(cdiv(R, 4), B), dim3(256), (size_t)C * sizeof(float), reinterpret_cast<hipStream_t>(stream),
                     partial, nsplit, inv_hw, w_reduce, b_reduce, hidden, C, R);
  return cfp_check_launch("cfp_se_hidden");
}

extern "C" int cfp_se_fold(const void* w_proj, void* w_out, const float* hidden, const float* w_expand_t, const float* b_expand,
                           int B, int Cout, int C, int R, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_se_fold");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(w_proj && w_out && hidden && w_expand_t && b_expand && aligned16(w_proj) && aligned16(w_out) && aligned16(w_expand_t),
              CFP_EINVAL, "cfp_se_fold: bad pointer");
  CFP_REQUIRE(B > 0 && B <= 65535 && Cout > 0 && C > 0 && C % 8 == 0 && R > 0 && R <= 4096, CFP_ESHAPE, "cfp_se_fold: bad shape");
  const int cv = C / ve;
  int row_lanes = (256 * 256 / B) / cv;          // ~one wave of the chip per call
  if (row_lanes > Cout) row_lanes = Cout;
  if (row_lanes < 1) row_lanes = 1;
  const int blocks = cdiv((long long)cv * row_lanes, 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(se_fold_kernel<bf16_t>, dim3(blocks, B), dim3(256), (size_t)R * sizeof(float), s, (const bf16_t*)w_proj,
                       (bf16_t*)w_out, hidden, w_expand_t, b_expand, Cout, C, R, row_lanes);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(se_fold_kernel<f16_t>, dim3(blocks, B), dim3(256), (size_t)R * sizeof(float), s, (const f16_t*)w_proj,
                       (f16_t*)w_out, hidden, w_expand_t, b_expand, Cout, C, R, row_lanes);
  else
    hipLaunchKernelGGL(se_fold_kernel<float>, dim3(blocks, B), dim3(256), (size_t)R * sizeof(float), s, (const float*)w_proj,
                       (float*)w_out, hidden, w_expand_t, b_expand, Cout, C, R, row_lanes);
  return cfp_check_launch("cfp_se_fold");
}

extern "C" int cfp_se_gate_fold(const float* partial, int nsplit, float inv_hw, const float* w_reduce, const float* b_reduce,
                                const float* w_expand_t, const float* b_expand, const void* w_proj, void* w_out, int B, int Cout,
                                int C, int R, int dtype, cfp_stream_t stream) {
  const int x3 = dtype == CFP_F32X3 ? 1 : 0;      // float32 project weights in, per-image pre-split f16x3 operands out
  if (x3) dtype = CFP_F32;
  CHECK_DTYPE("cfp_se_gate_fold");
  CFP_REQUIRE(partial && w_reduce && b_reduce && w_expand_t && b_expand && w_proj && w_out, CFP_EINVAL, "cfp_se_gate_fold: null pointer");
  CFP_REQUIRE(aligned16(partial) && aligned16(w_reduce) && aligned16(w_proj) && aligned16(w_out), CFP_EINVAL,
              "cfp_se_gate_fold: pointers must be 16-byte aligned");
  CFP_REQUIRE(B > 0 && B <= 65535 && nsplit > 0 && Cout > 0 && C > 0 && C % 8 == 0 && C <= 2048 && R > 0 && R <= 64, CFP_ESHAPE,
              "cfp_se_gate_fold: need C % 8 == 0, C <= 2048, R <= 64");
  const size_t lds = (size_t)(C + ((R + 63) & ~63) + 5 * 256) * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(cdiv(C, 256), B);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(se_gate_fold_kernel<bf16_t>, grid, dim3(1024), lds, s, partial, nsplit, inv_hw, w_reduce, b_reduce, w_expand_t,
                       b_expand, (const bf16_t*)w_proj, (bf16_t*)w_out, Cout, C, R, 0);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(se_gate_fold_kernel<f16_t>, grid, dim3(1024), lds, s, partial, nsplit, inv_hw, w_reduce, b_reduce, w_expand_t,
                       b_expand, (const f16_t*)w_proj, (f16_t*)w_out, Cout, C, R, 0);
  else
    hipLaunchKernelGGL(se_gate_fold_kernel<float>, grid, dim3(1024), lds, s, partial, nsplit, inv_hw, w_reduce, b_reduce, w_expand_t,
                       b_expand, (const float*)w_proj, (float*)w_out, Cout, C, R, x3);
  return cfp_check_launch("cfp_se_gate_fold");
}

int g_fold_zcap = 16;     // cfp_debug_set key 37: most row splits of the project weights per 128-channel slab (binds for single images only: 8 -> 16 measured 2.960 -> 2.948 ms, inside the noise but never behind)
void cfp_fold_debug_set(int v) { g_fold_zcap = v; }
extern "C" int cfp_se_gate_fold2(const float* hpart, int K, float inv_hw, const float* b_reduce, const float* w_expand_t, const float* b_expand,
                                 const float* w_proj, void* w_out, int B, int Cout, int C, int R, int dtype, cfp_stream_t stream) {
  const int x3 = dtype == CFP_F32X3 ? 1 : 0;
  if (x3) dtype = CFP_F32;
  CHECK_DTYPE("cfp_se_gate_fold2");
  CFP_REQUIRE(hpart && b_reduce && w_expand_t && b_expand && w_proj && w_out, CFP_EINVAL, "cfp_se_gate_fold2: null pointer");
  CFP_REQUIRE(aligned16(w_proj) && aligned16(w_out), CFP_EINVAL, "cfp_se_gate_fold2: pointers must be 16-byte aligned");
  CFP_REQUIRE(B > 0 && B <= 65535 && K > 0 && Cout > 0 && C > 0 && C % 8 == 0 && R > 0 && R <= 64, CFP_ESHAPE,
              "cfp_se_gate_fold2: need C % 8 == 0, R <= 64");
  const int slabs = cdiv(C, 128);
  int Z = cdiv(384, slabs * B);                       // enough workgroups to fill 256 CUs; each recomputes the (cheap) phases A and B
  if (Z > g_fold_zcap) Z = g_fold_zcap;
  if (Z > Cout) Z = Cout;
  if (Z < 1) Z = 1;
  const int rows_per_z = cdiv(Cout, Z);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(slabs, B, cdiv(Cout, rows_per_z));
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(se_gate_fold2_kernel<bf16_t>, grid, dim3(256), 0, s, hpart, K, inv_hw, b_reduce, w_expand_t, b_expand, w_proj,
                       (bf16_t*)w_out, Cout, C, R, rows_per_z, 0);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(se_gate_fold2_kernel<f16_t>, grid, dim3(256), 0, s, hpart, K, inv_hw, b_reduce, w_expand_t, b_expand, w_proj,
                       (f16_t*)w_out, Cout, C, R, rows_per_z, 0);
  else
    hipLaunchKernelGGL(se_gate_fold2_kernel<float>, grid, dim3(256), 0, s, hpart, K, inv_hw, b_reduce, w_expand_t, b_expand, w_proj,
                       (float*)w_out, Cout, C, R, rows_per_z, x3);
  return cfp_check_launch("cfp_se_gate_fold2");
}

extern "C" int cfp_se_scale(void* x, int ld, const float* hidden, const float* w_expand_t, const float* b_expand, int B, int HW,
                            int C, int R, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_se_scale");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(x && hidden && w_expand_t && b_expand && aligned16(x), CFP_EINVAL, "cfp_se_scale: bad pointer");
  CFP_REQUIRE(B > 0 && B <= 65535 && HW > 0 && C > 0 && C % 8 == 0 && R > 0 && R <= 4096 && ld % ve == 0 && ld >= C, CFP_ESHAPE,
              "cfp_se_scale: bad shape");
  const int cv = C / ve;
  int row_lanes = (256 * 256 / B) / cv;          // ~one wave of the chip per image batch
  if (row_lanes > HW) row_lanes = HW;
  if (row_lanes < 1) row_lanes = 1;
  const int blocks = cdiv((long long)cv * row_lanes, 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(se_scale_kernel<bf16_t>, dim3(blocks, B), dim3(256), (size_t)R * sizeof(float), s, (bf16_t*)x, ld, hidden,
                       w_expand_t, b_expand, HW, C, R, row_lanes);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(se_scale_kernel<f16_t>, dim3(blocks, B), dim3(256), (size_t)R * sizeof(float), s, (f16_t*)x, ld, hidden,
                       w_expand_t, b_expand, HW, C, R, row_lanes);
  else
    hipLaunchKernelGGL(se_scale_kernel<float>, dim3(blocks, B), dim3(256), (size_t)R * sizeof(float), s, (float*)x, ld, hidden,
                       w_expand_t, b_expand, HW, C, R, row_lanes);
  return cfp_check_launch("cfp_se_scale");
}

extern "C" int cfp_scale_channels(void* x, int ld, const float* gate, int B, int HW, int C, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_scale_channels");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(x && gate && aligned16(x), CFP_EINVAL, "cfp_scale_channels: bad pointer");
  CFP_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0 && ld % ve == 0 && ld >= C, CFP_ESHAPE, "cfp_scale_channels: bad shape");
  long long total = (long long)B * HW * (C / ve);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(scale_channels_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (bf16_t*)x, ld, gate, HW, C, total);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(scale_channels_kernel<f16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (f16_t*)x, ld, gate, HW, C, total);
  else
    hipLaunchKernelGGL(scale_channels_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (float*)x, ld, gate, HW, C, total);
  return cfp_check_launch("cfp_scale_channels");
}

extern "C" int cfp_layernorm(const void* in, int in_ld, const float* gamma, const float* beta, float eps,
                             const void* residual, int res_ld, void* out, int out_ld, int rows, int C, int dtype,
                             cfp_stream_t stream) {
  CHECK_DTYPE("cfp_layernorm");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(in && gamma && beta && out && aligned16(in) && aligned16(out) && aligned16(residual), CFP_EINVAL,
              "cfp_layernorm: bad pointer");
  const int lpr = C / ve;
  CFP_REQUIRE(rows > 0 && C > 0 && C % ve == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && in_ld % ve == 0 && out_ld % ve == 0 &&
                  in_ld >= C && out_ld >= C && (!residual || (res_ld % ve == 0 && res_ld >= C)),
              CFP_ESHAPE, "cfp_layernorm: C / vector width must be a power of two <= 64");
  const int rpb = 256 / lpr;
  long long blocks = ((long long)rows + rpb - 1) / rpb;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)in, in_ld, gamma, beta, eps,
                       (const bf16_t*)residual, res_ld, (bf16_t*)out, out_ld, rows, C);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(layernorm_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16_t*)in, in_ld, gamma, beta, eps,
                       (const f16_t*)residual, res_ld, (f16_t*)out, out_ld, rows, C);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)in, in_ld, gamma, beta, eps,
                       (const float*)residual, res_ld, (float*)out, out_ld, rows, C);
  return cfp_check_launch("cfp_layernorm");
}

extern "C" int cfp_resize_bilinear(const void* src, int src_ld, int Hs, int Ws, int sy0, int sx0, int sh, int sw,
                                   void* dst, int dst_ld, int Hd, int Wd, int dy0, int dx0, int dh, int dw,
                                   const uint8_t* zone_valid, int zn, int p1, int p2, int accumulate, int B, int C,
                                   int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_resize_bilinear");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(src && dst && aligned16(src) && aligned16(dst), CFP_EINVAL, "cfp_resize_bilinear: bad pointer");
  CFP_REQUIRE(B > 0 && C > 0 && C % 8 == 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 &&
                  src_ld % ve == 0 && dst_ld % ve == 0 && src_ld >= C && dst_ld >= C,
              CFP_ESHAPE, "cfp_resize_bilinear: bad shape");
  CFP_REQUIRE(!zone_valid || (zn > 0 && p1 > 0 && p2 > 0 && sh <= zn * p1 && sw <= zn * p2), CFP_ESHAPE,
              "cfp_resize_bilinear: zone grid smaller than the source rectangle");
  ResizeP p;
  p.src = src; p.dst = dst; p.zone_valid = zone_valid;
  p.src_ld = src_ld; p.Hs = Hs; p.Ws = Ws; p.sy0 = sy0; p.sx0 = sx0; p.sh = sh; p.sw = sw;
  p.dst_ld = dst_ld; p.Hd = Hd; p.Wd = Wd; p.dy0 = dy0; p.dx0 = dx0; p.dh = dh; p.dw = dw;
  p.zn = zn; p.p1 = p1; p.p2 = p2; p.accumulate = accumulate; p.B = B; p.C = C;
  p.scale_y = dh > 1 ? (float)(sh - 1) / (float)(dh - 1) : 0.f;
  p.scale_x = dw > 1 ? (float)(sw - 1) / (float)(dw - 1) : 0.f;
  long long total = (long long)B * dh * dw * (C / ve);
  CFP_REQUIRE(total < (1ll << 31), CFP_ESHAPE, "cfp_resize_bilinear: too many elements");
  p.fcv = make_fastdiv((unsigned)(C / ve)); p.fdw = make_fastdiv((unsigned)dw); p.fdh = make_fastdiv((unsigned)dh);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(resize_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, p);
  else if (dtype == CFP_F16) hipLaunchKernelGGL(resize_kernel<f16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(resize_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, p);
  return cfp_check_launch("cfp_resize_bilinear");
}

static int add_rowtable_impl(const void* in, int in_ld, const float* table, void* out, int out_ld, int rows, int C, int H, int W, int Wt,
                             int oy, int ox, const int* dev_off, int Ht, int dtype, cfp_stream_t stream, const char* who) {
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, std::string(who) + ": bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(in && table && out && aligned16(in) && aligned16(out), CFP_EINVAL, std::string(who) + ": bad pointer");
  CFP_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0 && Wt >= W + ox && oy >= 0 && ox >= 0 && in_ld % ve == 0 &&
                  out_ld % ve == 0 && in_ld >= C && out_ld >= C && (!dev_off || Ht >= H), CFP_ESHAPE, std::string(who) + ": bad shape");
  long long total = (long long)rows * (C / ve);
  CFP_REQUIRE(total < (1ll << 31) && aligned16(table), CFP_ESHAPE, std::string(who) + ": too many elements / table not 16-byte aligned");
  const FastDiv fcv = make_fastdiv((unsigned)(C / ve)), fw = make_fastdiv((unsigned)W), fh = make_fastdiv((unsigned)H);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define AR(T) hipLaunchKernelGGL(add_rowtable_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)in, in_ld, table, (T*)out, out_ld, \
                                 (long long)rows, C, H, W, Wt, oy, ox, dev_off, Ht, fcv, fw, fh)
  if (dtype == CFP_BF16) AR(bf16_t); else if (dtype == CFP_F16) AR(f16_t); else AR(float);
#undef AR
  return cfp_check_launch(who);
}

extern "C" int cfp_add_rowtable(const void* in, int in_ld, const float* table, void* out, int out_ld, int rows, int C,
                                int H, int W, int Wt, int oy, int ox, int dtype, cfp_stream_t stream) {
  return add_rowtable_impl(in, in_ld, table, out, out_ld, rows, C, H, W, Wt, oy, ox, nullptr, 0, dtype, stream, "cfp_add_rowtable");
}

extern "C" int cfp_add_rowtable_dev(const void* in, int in_ld, const float* table, void* out, int out_ld, int rows, int C, int H, int W,
                                    int Ht, int Wt, const int* oyox, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(oyox, CFP_EINVAL, "cfp_add_rowtable_dev: null offset pointer");
  return add_rowtable_impl(in, in_ld, table, out, out_ld, rows, C, H, W, Wt, 0, 0, oyox, Ht, dtype, stream, "cfp_add_rowtable_dev");
}

extern "C" int cfp_copy_rows(const void* in, int in_ld, void* out, int out_ld, int rows, int C, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_copy_rows");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(in && out && aligned16(in) && aligned16(out), CFP_EINVAL, "cfp_copy_rows: bad pointer");
  CFP_REQUIRE(rows > 0 && C > 0 && C % ve == 0 && in_ld % ve == 0 && out_ld % ve == 0 && in_ld >= C && out_ld >= C, CFP_ESHAPE,
              "cfp_copy_rows: bad shape");
  long long total = (long long)rows * (C / ve);
  CFP_REQUIRE(total < (1ll << 31), CFP_ESHAPE, "cfp_copy_rows: too many elements");
  const FastDiv fcv = make_fastdiv((unsigned)(C / ve));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16)
    hipLaunchKernelGGL(copy_rows_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)in, in_ld, (bf16_t*)out, out_ld, (long long)rows, C, fcv);
  else if (dtype == CFP_F16)
    hipLaunchKernelGGL(copy_rows_kernel<f16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const f16_t*)in, in_ld, (f16_t*)out, out_ld, (long long)rows, C, fcv);
  else
    hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)in, in_ld, (float*)out, out_ld, (long long)rows, C, fcv);
  return cfp_check_launch("cfp_copy_rows");
}

extern "C" int cfp_copy_rows2(const void* in0, int in0_ld, void* out0, int out0_ld, int C0, const void* in1, int in1_ld, void* out1, int out1_ld,
                              int C1, int rows, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_copy_rows2");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(in0 && out0 && in1 && out1 && aligned16(in0) && aligned16(out0) && aligned16(in1) && aligned16(out1), CFP_EINVAL,
              "cfp_copy_rows2: bad pointer");
  CFP_REQUIRE(rows > 0 && C0 > 0 && C1 > 0 && C0 % ve == 0 && C1 % ve == 0 && in0_ld % ve == 0 && out0_ld % ve == 0 && in1_ld % ve == 0 &&
                  out1_ld % ve == 0 && in0_ld >= C0 && out0_ld >= C0 && in1_ld >= C1 && out1_ld >= C1, CFP_ESHAPE, "cfp_copy_rows2: bad shape");
  const long long total = (long long)rows * (std::max(C0, C1) / ve);
  CFP_REQUIRE(total < (1ll << 31), CFP_ESHAPE, "cfp_copy_rows2: too many elements");
  Copy2P p;
  p.in[0] = in0; p.in[1] = in1; p.out[0] = out0; p.out[1] = out1;
  p.in_ld[0] = in0_ld; p.in_ld[1] = in1_ld; p.out_ld[0] = out0_ld; p.out_ld[1] = out1_ld;
  p.fcv[0] = make_fastdiv((unsigned)(C0 / ve)); p.fcv[1] = make_fastdiv((unsigned)(C1 / ve));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(ew_blocks(total), 2);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(copy_rows2_kernel<bf16_t>, grid, dim3(256), 0, s, p, (long long)rows);
  else if (dtype == CFP_F16) hipLaunchKernelGGL(copy_rows2_kernel<f16_t>, grid, dim3(256), 0, s, p, (long long)rows);
  else hipLaunchKernelGGL(copy_rows2_kernel<float>, grid, dim3(256), 0, s, p, (long long)rows);
  return cfp_check_launch("cfp_copy_rows2");
}

extern "C" int cfp_rgb_to_nhwc8(const float* rgb, void* out, int B, int H, int W, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_rgb_to_nhwc8");
  CFP_REQUIRE(rgb && out && aligned16(out), CFP_EINVAL, "cfp_rgb_to_nhwc8: bad pointer");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0, CFP_ESHAPE, "cfp_rgb_to_nhwc8: bad shape");
  long long total = (long long)B * H * W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(rgb_to_nhwc8_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, rgb, (bf16_t*)out, H * W, total);
  else if (dtype == CFP_F16) hipLaunchKernelGGL(rgb_to_nhwc8_kernel<f16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, rgb, (f16_t*)out, H * W, total);
  else hipLaunchKernelGGL(rgb_to_nhwc8_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, rgb, (float*)out, H * W, total);
  return cfp_check_launch("cfp_rgb_to_nhwc8");
}

extern "C" int cfp_rgb_to_nhwc8_hilo(const float* rgb, void* out, int B, int H, int W, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_rgb_to_nhwc8_hilo: 16-bit storage types only");
  CFP_REQUIRE(rgb && out && aligned16(out), CFP_EINVAL, "cfp_rgb_to_nhwc8_hilo: bad pointer");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0, CFP_ESHAPE, "cfp_rgb_to_nhwc8_hilo: bad shape");
  long long total = (long long)B * H * W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(rgb_to_nhwc8_hilo_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, rgb, (bf16_t*)out, H * W, total);
  else hipLaunchKernelGGL(rgb_to_nhwc8_hilo_kernel<f16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, rgb, (f16_t*)out, H * W, total);
  return cfp_check_launch("cfp_rgb_to_nhwc8_hilo");
}

extern "C" int cfp_scalar_to_rows8(const float* in, void* out, int rows, int dtype, cfp_stream_t stream) {
  CHECK_DTYPE("cfp_scalar_to_rows8");
  CFP_REQUIRE(in && out && aligned16(out), CFP_EINVAL, "cfp_scalar_to_rows8: bad pointer");
  CFP_REQUIRE(rows > 0, CFP_ESHAPE, "cfp_scalar_to_rows8: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(scalar_to_rows8_kernel<bf16_t>, dim3(ew_blocks(rows)), dim3(256), 0, s, in, (bf16_t*)out, (long long)rows);
  else if (dtype == CFP_F16) hipLaunchKernelGGL(scalar_to_rows8_kernel<f16_t>, dim3(ew_blocks(rows)), dim3(256), 0, s, in, (f16_t*)out, (long long)rows);
  else hipLaunchKernelGGL(scalar_to_rows8_kernel<float>, dim3(ew_blocks(rows)), dim3(256), 0, s, in, (float*)out, (long long)rows);
  return cfp_check_launch("cfp_scalar_to_rows8");
}
