// Squeeze-excite in the training step (timm SqueezeExcite inside every inverted-residual block of the RGB encoder,
// /root/reference/src/models/encoder.py:57-69 -> timm efficientnet_blocks.SqueezeExcite: x * sigmoid(conv_expand(silu(conv_reduce(mean_HW(x)))))).
//
// On the tape this was ~8 launches forward and ~16 backward per block (channel sums, scale, two GEMVs with bias, two activations, the
// broadcast multiply; backward of each), every one a [16, C] or [16, R] problem at the ~5 us launch floor: 24 blocks x 24 launches = 9 % of
// the step.  Here the gate and its backward are three kernels; with the channel sums, the channel dot and the two broadcast multiplies that
// stay as they were the block costs 3 + 4 launches.
//
//   forward   mean[b,c] = sum_s part[b,s,c] / HW;  z1 = W1 mean + b1;  a1 = silu(z1);  z2 = W2 a1 + b2;  gate = sigmoid(z2)
//   backward  dz2 = dgate * gate * (1 - gate);  db2 = sum_b dz2;  dW2[c,r] = sum_b dz2[b,c] a1[b,r];  da1 = W2^T dz2;
//             dz1 = da1 * silu'(z1);  db1 = sum_b dz1;  dW1[r,c] = sum_b dz1[b,r] mean[b,c];  dmean = W1^T dz1;  add = dmean / HW
//
// W1 [Rp][C], b1 [Rp], W2 [C][Rp], b2 [C]: float32, Rp = R padded to a multiple of 4 with zero rows / columns (<= 64).
// Full-precision expf / division like the other training kernels; every sum in a fixed order.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoid_p(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_p(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_grad_p(float z) { const float s = sigmoid_p(z); return s * (1.f + z * (1.f - s)); }

constexpr int SE_RMAX = 64;

// Sum of v[i] over the 64 lanes for every i, delivered as: lane l returns the total of v[l].  A butterfly in which the partners split the
// remaining indices (the lane with bit o set keeps the upper half): 32 + 16 + ... + 1 = 63 shuffles instead of the 64 x 6 of one
// wave_sum per value (ds_bpermute is the bottleneck: the straightforward form took ~30 us for sixteen waves).
template <int O>
__device__ __forceinline__ void wave_rs_step(float (&v)[64], int lane) {
  const bool up = (lane & O) != 0;
#pragma unroll
  for (int i = 0; i < O; ++i) {
    const float send = up ? v[i] : v[i + O];
    const float keep = up ? v[i + O] : v[i];
    v[i] = keep + __shfl_xor(send, O, 64);
  }
}
__device__ __forceinline__ float wave_reduce_scatter64(float (&v)[64], int lane) {      // every index is a compile-time constant: v stays in registers
  wave_rs_step<32>(v, lane); wave_rs_step<16>(v, lane); wave_rs_step<8>(v, lane);
  wave_rs_step<4>(v, lane); wave_rs_step<2>(v, lane); wave_rs_step<1>(v, lane);
  return v[0];
}

// grid (ceil(C / 256), B), 1024 threads: every workgroup recomputes mean and hidden of its image (cheaper than a kernel boundary) and
// produces the gate of its 256-channel slab; slab 0 also stores mean and z1 for the backward.
__global__ __launch_bounds__(1024) void se_train_fwd_kernel(const float* __restrict__ part, int nsplit, float inv_hw, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float* __restrict__ mean_out,
                                                            float* __restrict__ z1_out, float* __restrict__ gate_out, int C, int Rp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* mean = sm;                         // [C]
  float* a1 = sm + C;                       // [64]
  float* gpart = a1 + SE_RMAX;              // [4][256]
  const int b = blockIdx.y, slab = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C4 = C >> 2;
  for (int c4 = tid; c4 < C4; c4 += 1024) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 3 < nsplit; j += 4) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(part + ((long long)b * nsplit + j) * C + c4 * 4);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(part + ((long long)b * nsplit + j + 1) * C + c4 * 4);
      const f32x4 q2 = *reinterpret_cast<const f32x4*>(part + ((long long)b * nsplit + j + 2) * C + c4 * 4);
      const f32x4 q3 = *reinterpret_cast<const f32x4*>(part + ((long long)b * nsplit + j + 3) * C + c4 * 4);
      s += q0; s += q1; s += q2; s += q3;
    }
    for (; j < nsplit; ++j) s += *reinterpret_cast<const f32x4*>(part + ((long long)b * nsplit + j) * C + c4 * 4);
    s = s * inv_hw;
    *reinterpret_cast<f32x4*>(mean + c4 * 4) = s;
    if (slab == 0) *reinterpret_cast<f32x4*>(mean_out + (long long)b * C + c4 * 4) = s;
  }
  __syncthreads();
  // hidden units: wave w takes r = w, w + 16, ...; all weight loads of a unit pair in flight together
  {
    constexpr int MAXL = 8;                 // C <= 8 * 256
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += 2) {
      f32x4 wv[2][MAXL];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = wave + 16 * (k0 + k);
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
          const int c4 = lane + 64 * l;
          wv[k][l] = (r < Rp && c4 < C4) ? *reinterpret_cast<const f32x4*>(w1 + (long long)r * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
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
        if (lane == 0 && r < SE_RMAX) {
          const float z = r < Rp ? s + b1[r] : 0.f;
          a1[r] = r < Rp ? silu_p(z) : 0.f;
          if (slab == 0 && r < Rp) z1_out[(long long)b * Rp + r] = z;
        }
      }
    }
  }
  __syncthreads();
  // gate of this slab: 256 channels x 4 quarter-ranges of r
  {
    const int cl = tid & 255, qr = tid >> 8;
    const int c = slab * 256 + cl;
    float s = 0.f;
    if (c < C) {
      const float* row = w2 + (long long)c * Rp;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r0 = qr * 16 + u * 4;
        if (r0 < Rp) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(row + r0);
          s = fmaf(a1[r0], wv[0], s); s = fmaf(a1[r0 + 1], wv[1], s); s = fmaf(a1[r0 + 2], wv[2], s); s = fmaf(a1[r0 + 3], wv[3], s);
        }
      }
    }
    gpart[qr * 256 + cl] = s;
  }
  __syncthreads();
  if (tid < 256) {
    const int c = slab * 256 + tid;
    if (c < C) gate_out[(long long)b * C + c] = sigmoid_p((gpart[tid] + gpart[256 + tid]) + (gpart[512 + tid] + gpart[768 + tid]) + b2[c]);
  }
}

// grid B, 1024 threads: dz2[b, :] and dz1[b, :] of one image
__global__ __launch_bounds__(1024) void se_train_bwd1_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                             const float* __restrict__ z1, const float* __restrict__ w2,
                                                             float* __restrict__ dz2_out, float* __restrict__ dz1_out, int C, int Rp) {
  __shared__ float red[16][SE_RMAX];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[SE_RMAX];
#pragma unroll
  for (int r = 0; r < SE_RMAX; ++r) acc[r] = 0.f;
  for (int c = tid; c < C; c += 1024) {
    const float g = gate[(long long)b * C + c];
    const float dz = dgate[(long long)b * C + c] * g * (1.f - g);
    dz2_out[(long long)b * C + c] = dz;
    const float* row = w2 + (long long)c * Rp;
#pragma unroll
    for (int r4 = 0; r4 < SE_RMAX / 4; ++r4) {
      if (r4 * 4 < Rp) {                    // uniform
        const f32x4 wv = *reinterpret_cast<const f32x4*>(row + r4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r4 * 4 + e] = fmaf(dz, wv[e], acc[r4 * 4 + e]);
      }
    }
  }
  red[wave][lane] = wave_reduce_scatter64(acc, lane);
  __syncthreads();
  if (tid < Rp) {
    float da = 0.f;
    for (int w = 0; w < 16; ++w) da += red[w][tid];
    dz1_out[(long long)b * Rp + tid] = da * silu_grad_p(z1[(long long)b * Rp + tid]);
  }
}

// grid ceil(C / 64), 256 threads = 64 channels x 4 sub-lanes: the parameter gradients and the gradient of the mean.  Sub-lane q owns the
// hidden units 16 q .. 16 q + 15 of its channel for dW2 / dW1 and the images q, q + 4, ... for the gradient of the mean; a1 and dz1 sit in
// LDS and are read as 16-byte vectors (the first version -- one thread per channel walking all 64 units with scalar LDS reads -- took ~45 us).
__global__ __launch_bounds__(256) void se_train_bwd2_kernel(const float* __restrict__ dz2, const float* __restrict__ dz1, const float* __restrict__ z1,
                                                            const float* __restrict__ mean, const float* __restrict__ w1, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                            float* __restrict__ add, float inv_hw, float beta, int B, int C, int Rp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sa1 = sm;                          // [B][Rp]
  float* sdz1 = sm + B * Rp;                // [B][Rp]
  const int tid = threadIdx.x;
  for (int i = tid; i < B * Rp; i += 256) { sa1[i] = silu_p(z1[i]); sdz1[i] = dz1[i]; }
  __syncthreads();
  if (blockIdx.x == 0 && tid < Rp) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sdz1[b * Rp + tid];
    db1[tid] = beta != 0.f ? beta * db1[tid] + s : s;
  }
  const int cl = tid & 63, q = tid >> 6;
  const int c = blockIdx.x * 64 + cl;
  if (c >= C) return;
  const int r0 = q * 16;                    // this sub-lane's hidden units
  const bool r_on = r0 < Rp;                // Rp % 4 == 0: units r0 + 4 u .. + 3 exist iff r0 + 4 u < Rp
  // ---- dW2[c][r] = sum_b dz2[b,c] a1[b,r],  db2[c] = sum_b dz2[b,c];  dW1[r][c] = sum_b dz1[b,r] mean[b,c]
  f32x4 a2[4], a1w[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { a2[u] = f32x4{0.f, 0.f, 0.f, 0.f}; a1w[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float sb = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {      // sixteen images' values requested together (one image per iteration = one round trip each)
    float dv[16], mv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int b = min(b0 + i, B - 1);
      dv[i] = dz2[(long long)b * C + c];
      mv[i] = mean[(long long)b * C + c];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int b = b0 + i;
      if (b >= B) continue;
      sb += dv[i];
      if (r_on) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (r0 + 4 * u < Rp) {
            a2[u] += *reinterpret_cast<const f32x4*>(sa1 + b * Rp + r0 + 4 * u) * dv[i];
            a1w[u] += *reinterpret_cast<const f32x4*>(sdz1 + b * Rp + r0 + 4 * u) * mv[i];
          }
        }
      }
    }
  }
  if (q == 0) db2[c] = beta != 0.f ? beta * db2[c] + sb : sb;
  if (r_on) {
    float* row = dw2 + (long long)c * Rp + r0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + 4 * u < Rp) {
        f32x4 o = a2[u];
        if (beta != 0.f) o += *reinterpret_cast<const f32x4*>(row + 4 * u) * beta;
        *reinterpret_cast<f32x4*>(row + 4 * u) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* dd = dw1 + (long long)(r0 + 4 * u + e) * C + c;
          *dd = beta != 0.f ? beta * *dd + a1w[u][e] : a1w[u][e];
        }
      }
    }
  }
  // ---- add[b][c] = (sum_r dz1[b,r] W1[r][c]) / HW for the images b = q, q + 4, ... (up to 16 per sub-lane: B <= 64)
  float dm[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dm[i] = 0.f;
  float wc[SE_RMAX];                         // this channel's column of W1, all loads in flight together
#pragma unroll
  for (int r = 0; r < SE_RMAX; ++r) wc[r] = w1[(long long)min(r, Rp - 1) * C + c];
#pragma unroll
  for (int r4 = 0; r4 < SE_RMAX / 4; ++r4) {
    if (r4 * 4 < Rp) {                       // uniform
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int b = q + 4 * i;
        if (b < B) {
          const f32x4 dzv = *reinterpret_cast<const f32x4*>(sdz1 + b * Rp + r4 * 4);
          dm[i] += (dzv[0] * wc[r4 * 4] + dzv[1] * wc[r4 * 4 + 1]) + (dzv[2] * wc[r4 * 4 + 2] + dzv[3] * wc[r4 * 4 + 3]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int b = q + 4 * i;
    if (b < B) add[(long long)b * C + c] = dm[i] * inv_hw;
  }
}

}  // namespace

extern "C" int cfp_se_train_fwd(const float* partial, int nsplit, float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
                                float* mean, float* z1, float* gate, int B, int C, int Rp, cfp_stream_t stream) {
  CFP_REQUIRE(partial && w1 && b1 && w2 && b2 && mean && z1 && gate, CFP_EINVAL, "cfp_se_train_fwd: null pointer");
  CFP_REQUIRE(aligned16(partial) && aligned16(w1) && aligned16(w2) && aligned16(mean), CFP_EINVAL, "cfp_se_train_fwd: pointers must be 16-byte aligned");
  CFP_REQUIRE(B > 0 && B <= 65535 && nsplit > 0 && C > 0 && C % 4 == 0 && C <= 2048 && Rp > 0 && Rp % 4 == 0 && Rp <= SE_RMAX, CFP_ESHAPE,
              "cfp_se_train_fwd: need C % 4 == 0, C <= 2048, Rp % 4 == 0, Rp <= 64");
  const size_t lds = (size_t)(C + SE_RMAX + 4 * 256) * sizeof(float);
  hipLaunchKernelGGL(se_train_fwd_kernel, dim3(cdiv(C, 256), B), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), partial, nsplit, inv_hw,
                     w1, b1, w2, b2, mean, z1, gate, C, Rp);
  return cfp_check_launch("cfp_se_train_fwd");
}

extern "C" int cfp_se_train_bwd(const float* dgate, const float* gate, const float* z1, const float* mean, const float* w1, const float* w2,
                                float* dw1, float* db1, float* dw2, float* db2, float* add, float* ws, float inv_hw, float beta, int B, int C,
                                int Rp, cfp_stream_t stream) {
  CFP_REQUIRE(dgate && gate && z1 && mean && w1 && w2 && dw1 && db1 && dw2 && db2 && add && ws, CFP_EINVAL, "cfp_se_train_bwd: null pointer");
  CFP_REQUIRE(aligned16(w2) && aligned16(dw2) && aligned16(ws), CFP_EINVAL, "cfp_se_train_bwd: pointers must be 16-byte aligned");
  CFP_REQUIRE(B > 0 && B <= 64 && C > 0 && C % 4 == 0 && C <= 2048 && Rp > 0 && Rp % 4 == 0 && Rp <= SE_RMAX, CFP_ESHAPE,
              "cfp_se_train_bwd: need B <= 64, C % 4 == 0, C <= 2048, Rp % 4 == 0, Rp <= 64");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* dz2 = ws;                          // [B][C]
  float* dz1 = ws + (size_t)B * C;          // [B][Rp]
  hipLaunchKernelGGL(se_train_bwd1_kernel, dim3(B), dim3(1024), 0, s, dgate, gate, z1, w2, dz2, dz1, C, Rp);
  hipLaunchKernelGGL(se_train_bwd2_kernel, dim3(cdiv(C, 64)), dim3(256), (size_t)2 * B * Rp * sizeof(float), s, dz2, dz1, z1, mean, w1, dw1, db1,
                     dw2, db2, add, inv_hw, beta, B, C, Rp);
  return cfp_check_launch("cfp_se_train_bwd");
}

extern "C" size_t cfp_se_train_ws_floats(int B, int C, int Rp) { return (B > 0 && C > 0 && Rp > 0) ? (size_t)B * (C + Rp) : 0; }
