// Adaptive-bins head: bin-width regressor (tiny, one workgroup per image) and the per-pixel
// 256-way softmax + expectation over bin centres.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void bin_regressor_kernel(const float* __restrict__ partial, int nsplit, float inv_hw,
                                                            const float* __restrict__ w1x1, const float* __restrict__ w0,
                                                            const float* __restrict__ b0, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float min_val, float max_val,
                                                            int norm, float* __restrict__ edges, float* __restrict__ centers,
                                                            int C, int hidden, int nbins) {
  extern __shared__ float sm[];
  float* a = sm;                 // [max(C, hidden, nbins)]
  float* t = sm + 512;           // second buffer
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int j = 0; j < nsplit; ++j) s += partial[((long long)b * nsplit + j) * C + c];
    a[c] = s * inv_hw;
  }
  __syncthreads();
  // conv1x1 without bias commutes with the spatial mean (decoder.py:24-25)
  for (int o = tid; o < C; o += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(a[c], w1x1[(long long)o * C + c], s);
    t[o] = s;
  }
  __syncthreads();
  for (int o = tid; o < hidden; o += 256) {
    float s = b0[o];
    for (int c = 0; c < C; ++c) s = fmaf(t[c], w0[(long long)o * C + c], s);
    a[o] = s > 0.f ? s : 0.01f * s;
  }
  __syncthreads();
  for (int o = tid; o < hidden; o += 256) {
    float s = b1[o];
    for (int c = 0; c < hidden; ++c) s = fmaf(a[c], w1[(long long)o * hidden + c], s);
    t[o] = s > 0.f ? s : 0.01f * s;
  }
  __syncthreads();
  for (int o = tid; o < nbins; o += 256) {
    float s = b2[o];
    for (int c = 0; c < hidden; ++c) s = fmaf(t[c], w2[(long long)o * hidden + c], s);
    a[o] = s;
  }
  __syncthreads();
  if (tid == 0) {
    // sequential on purpose: torch.cumsum / the L1 normalisation run in bin order on the host too
    float total = 0.f;
    if (norm == 1) {
      float mx = a[0];
      for (int i = 1; i < nbins; ++i) mx = fmaxf(mx, a[i]);
      for (int i = 0; i < nbins; ++i) { a[i] = expf(a[i] - mx); total += a[i]; }
    } else {
      for (int i = 0; i < nbins; ++i) {
        float y = norm == 0 ? fmaxf(a[i], 0.f) + 0.1f : 1.f / (1.f + expf(-a[i]));
        a[i] = y;
        total += y;
      }
    }
    float e = min_val;
    float* eb = edges + (long long)b * (nbins + 1);
    float* cb = centers + (long long)b * nbins;
    eb[0] = e;
    for (int i = 0; i < nbins; ++i) {
      float wdt = (max_val - min_val) * (a[i] / total);
      float e2 = e + wdt;
      eb[i + 1] = e2;
      cb[i] = 0.5f * (e + e2);
      e = e2;
    }
  }
}

// One wave per pixel: lane holds NB/64 consecutive logits.  Probabilities of a 64-pixel tile are
// transposed through LDS so the NCHW prob write is 16-byte vectors along the pixel axis.
template <typename T, int NBINS>
__global__ __launch_bounds__(256) void bin_softmax_kernel(const T* __restrict__ logits, int ld, const float* __restrict__ centers,
                                                          T* __restrict__ prob, float* __restrict__ pred, int HW) {
  constexpr int PL = NBINS / 64;            // logits per lane
  constexpr int TP = sizeof(T) == 2 ? 64 : 32;   // pixels per tile
  constexpr int PITCH = TP + (sizeof(T) == 2 ? 8 : 4);
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T* sP = reinterpret_cast<T*>(smraw);      // [NBINS][PITCH]
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * TP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float cen[PL];
#pragma unroll
  for (int j = 0; j < PL; ++j) cen[j] = centers[(long long)b * NBINS + lane * PL + j];
  for (int pi = wave; pi < TP; pi += 4) {
    const int pix = p0 + pi;
    if (pix >= HW) break;
    const T* lp = logits + ((long long)b * HW + pix) * ld + lane * PL;
    float v[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) v[j] = to_f32<T>(lp[j]);
    float mx = v[0];
#pragma unroll
    for (int j = 1; j < PL; ++j) mx = fmaxf(mx, v[j]);
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PL; ++j) { v[j] = __expf(v[j] - mx); s += v[j]; }
    s = wave_sum(s);
    const float inv = 1.f / s;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < PL; ++j) { v[j] *= inv; dot = fmaf(v[j], cen[j], dot); }
    dot = wave_sum(dot);
    if (lane == 0) pred[(long long)b * HW + pix] = dot;
    if (prob) {
#pragma unroll
      for (int j = 0; j < PL; ++j) sP[(lane * PL + j) * PITCH + pi] = from_f32<T>(v[j]);
    }
  }
  if (!prob) return;
  __syncthreads();
  constexpr int VE = Vec<T>::N;
  constexpr int CH = TP / VE;               // vectors per channel row of the tile
  for (int q = threadIdx.x; q < NBINS * CH; q += 256) {
    const int n = q / CH, ch = q % CH;
    const int pix = p0 + ch * VE;
    if (pix >= HW) continue;
    T* dp = prob + ((long long)b * NBINS + n) * HW + pix;
    if (pix + VE <= HW && (((long long)b * NBINS + n) * HW + pix) % VE == 0) {
      *reinterpret_cast<u32x4*>(dp) = *reinterpret_cast<const u32x4*>(sP + n * PITCH + ch * VE);
    } else {
      for (int e = 0; e < VE && pix + e < HW; ++e) dp[e] = sP[n * PITCH + ch * VE + e];
    }
  }
}

}  // namespace

extern "C" int cfp_bin_regressor(const float* partial, int nsplit, float inv_hw, const float* w1x1, const float* w0,
                                 const float* b0, const float* w1, const float* b1, const float* w2, const float* b2,
                                 float min_val, float max_val, int norm, float* edges, float* centers, int B, int C,
                                 int hidden, int nbins, cfp_stream_t stream) {
  CFP_REQUIRE(partial && w1x1 && w0 && b0 && w1 && b1 && w2 && b2 && edges && centers, CFP_EINVAL, "cfp_bin_regressor: null pointer");
  CFP_REQUIRE(B > 0 && nsplit > 0 && C > 0 && C <= 512 && hidden > 0 && hidden <= 512 && nbins > 0 && nbins <= 512 &&
                  norm >= 0 && norm <= 2, CFP_ESHAPE, "cfp_bin_regressor: bad shape (C, hidden, nbins <= 512)");
  hipLaunchKernelGGL(bin_regressor_kernel, dim3(B), dim3(256), 1024 * sizeof(float), reinterpret_cast<hipStream_t>(stream), partial,
                     nsplit, inv_hw, w1x1, w0, b0, w1, b1, w2, b2, min_val, max_val, norm, edges, centers, C, hidden, nbins);
  return cfp_check_launch("cfp_bin_regressor");
}

extern "C" int cfp_bin_softmax(const void* logits, int ld, const float* centers, void* prob, float* pred, int B, int HW,
                               int nbins, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dtype == CFP_F32 || dtype == CFP_BF16, CFP_EINVAL, "cfp_bin_softmax: bad dtype");
  CFP_REQUIRE(logits && centers && pred, CFP_EINVAL, "cfp_bin_softmax: null pointer");
  CFP_REQUIRE(B > 0 && B <= 65535 && HW > 0 && (nbins == 256 || nbins == 128 || nbins == 64) && ld >= nbins, CFP_ESHAPE,
              "cfp_bin_softmax: nbins must be 64, 128 or 256");
  CFP_REQUIRE(aligned16(prob), CFP_EINVAL, "cfp_bin_softmax: prob must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define SM_LAUNCH(T, NBI)                                                                                          \
  do {                                                                                                             \
    constexpr int TP = sizeof(T) == 2 ? 64 : 32;                                                                   \
    constexpr int PITCH = TP + (sizeof(T) == 2 ? 8 : 4);                                                           \
    size_t lds = (size_t)NBI * PITCH * sizeof(T);                                                                  \
    hipLaunchKernelGGL((bin_softmax_kernel<T, NBI>), dim3(cdiv(HW, TP), B), dim3(256), lds, s, (const T*)logits, ld, \
                       centers, (T*)prob, pred, HW);                                                               \
  } while (0)
#define SM_SWITCH(T) switch (nbins) { case 256: SM_LAUNCH(T, 256); break; case 128: SM_LAUNCH(T, 128); break; default: SM_LAUNCH(T, 64); break; }
  if (dtype == CFP_BF16) { SM_SWITCH(bf16_t); } else { SM_SWITCH(float); }
#undef SM_SWITCH
#undef SM_LAUNCH
  return cfp_check_launch("cfp_bin_softmax");
}

extern "C" int cfp_bin_head_fused(const void* x, int x_ld, const void* w, const float* bias, const float* centers,
                                  void* prob, float* pred, int B, int HW, int Cin, int dtype, cfp_stream_t stream) {
  cfp_set_error("cfp_bin_head_fused: not built in this version");
  return CFP_EINVAL;
}
