// Adaptive-bins head: bin-width regressor (tiny, one workgroup per image) and the per-pixel
// 256-way softmax + expectation over bin centres.
#include "igemm_core.h"

namespace {

// dense layer for one image: out[o] = act(bias[o] + sum_c in[c] * wt[c][o]).  Weights are stored
// TRANSPOSED ([n_in][n_out]) so thread o's loads are unit-stride across threads.  The block is
// 1024 threads: the input range is cut in 4 slices (one per 256-thread group) and every thread
// issues its loads 16 at a time, so a layer costs a handful of memory round trips instead of n_in / 4.
__device__ __forceinline__ void fc_layer(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                                         float* __restrict__ outv, float* __restrict__ part, int n_in, int n_out, bool lrelu) {
  const int g = threadIdx.x >> 8, o0 = threadIdx.x & 255;
  const int per = (n_in + 3) >> 2;
  const int c0 = g * per, c1 = min(n_in, c0 + per);
  for (int o = o0; o < n_out; o += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int c = c0; c < c1; c += 32) {          // 32 weight loads in flight per thread: a layer is 1-2 memory round trips
      float w[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) w[j] = (c + j < c1) ? wt[(long long)(c + j) * n_out + o] : 0.f;
#pragma unroll
      for (int j = 0; j < 32; j += 4) {
        s0 = fmaf((c + j < c1) ? in[c + j] : 0.f, w[j], s0);
        s1 = fmaf((c + j + 1 < c1) ? in[c + j + 1] : 0.f, w[j + 1], s1);
        s2 = fmaf((c + j + 2 < c1) ? in[c + j + 2] : 0.f, w[j + 2], s2);
        s3 = fmaf((c + j + 3 < c1) ? in[c + j + 3] : 0.f, w[j + 3], s3);
      }
    }
    part[g * 512 + o] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
    float s = (part[o] + part[512 + o]) + (part[1024 + o] + part[1536 + o]) + (bias ? bias[o] : 0.f);
    outv[o] = (lrelu && s < 0.f) ? 0.01f * s : s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void bin_regressor_kernel(const float* __restrict__ partial, int nsplit, float inv_hw,
                                                             const float* __restrict__ w1x1, const float* __restrict__ w0,
                                                             const float* __restrict__ b0, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float min_val, float max_val,
                                                             int norm, float* __restrict__ edges, float* __restrict__ centers,
                                                             int C, int hidden, int nbins) {
  __shared__ float a[512], t[512], part[4 * 512];
  const int b = blockIdx.x, tid = threadIdx.x;
  // spatial mean from the per-slice partial sums: slices are summed in order by 4 groups, then combined
  {
    const int g = tid >> 8, c0 = tid & 255;
    const int per = (nsplit + 3) >> 2;
    for (int c = c0; c < C; c += 256) {
      float s = 0.f;
      for (int j = g * per; j < min(nsplit, (g + 1) * per); ++j) s += partial[((long long)b * nsplit + j) * C + c];
      part[g * 512 + c] = s;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) a[c] = ((part[c] + part[512 + c]) + (part[1024 + c] + part[1536 + c])) * inv_hw;
    __syncthreads();
  }
  fc_layer(a, w1x1, nullptr, t, part, C, C, false);   // conv1x1 without bias commutes with the spatial mean
  fc_layer(t, w0, b0, a, part, C, hidden, true);
  fc_layer(a, w1, b1, t, part, hidden, hidden, true);
  fc_layer(t, w2, b2, a, part, hidden, nbins, false);
  // normalisation: y_i in parallel, then bin order sums on one lane (torch.cumsum order)
  if (norm != 1) {
    for (int i = tid; i < nbins; i += 1024) a[i] = norm == 0 ? fmaxf(a[i], 0.f) + 0.1f : 1.f / (1.f + expf(-a[i]));
    __syncthreads();
  }
  if (tid == 0) {
    float total = 0.f;
    if (norm == 1) {
      float mx = a[0];
      for (int i = 1; i < nbins; ++i) mx = fmaxf(mx, a[i]);
      for (int i = 0; i < nbins; ++i) { a[i] = expf(a[i] - mx); total += a[i]; }
    } else {
      for (int i = 0; i < nbins; ++i) total += a[i];
    }
    t[0] = total;
  }
  __syncthreads();
  {
    const float total = t[0];
    __syncthreads();
    for (int i = tid; i < nbins; i += 1024) t[i] = (max_val - min_val) * (a[i] / total);
    __syncthreads();
  }
  if (tid == 0) {
    float e = min_val;
    float* eb = edges + (long long)b * (nbins + 1);
    float* cb = centers + (long long)b * nbins;
    eb[0] = e;
    for (int i = 0; i < nbins; ++i) {
      float e2 = e + t[i];
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
  hipLaunchKernelGGL(bin_regressor_kernel, dim3(B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), partial,
                     nsplit, inv_hw, w1x1, w0, b0, w1, b1, w2, b2, min_val, max_val, norm, edges, centers, C, hidden, nbins);
  return cfp_check_launch("cfp_bin_regressor");
}

extern "C" int cfp_bin_softmax(const void* logits, int ld, const float* centers, void* prob, float* pred, int B, int HW,
                               int nbins, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_bin_softmax: bad dtype");
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
  if (dtype == CFP_BF16) { SM_SWITCH(bf16_t); } else if (dtype == CFP_F16) { SM_SWITCH(f16_t); } else { SM_SWITCH(float); }
#undef SM_SWITCH
#undef SM_LAUNCH
  return cfp_check_launch("cfp_bin_softmax");
}

// ---- fused bin head ----------------------------------------------------------------------
// logits = x @ w^T + bias (1x1 conv, Cin -> 256) on the matrix cores; the 128 x 256 logit tile never
// leaves the registers: row softmax + expectation with 16-lane shuffle butterflies on the MFMA
// accumulator layout, probabilities transposed through LDS so the NCHW write is 16-byte vectors
// along the pixel axis.
namespace {
constexpr int HBM_ = 128, HBN_ = 256, HPITCH = HBM_ + 8;

template <typename H>
__global__ __launch_bounds__(256) void bin_head_fused_kernel(ConvP p, const float* __restrict__ bias,
                                                             const float* __restrict__ centers, H* __restrict__ prob,
                                                             float* __restrict__ pred, int HW) {
  constexpr int TM = 2, TN = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * HBM_;
  f32x4 acc[TM][TN];
  igemm_mainloop<H, HBM_, HBN_, 4, 1>(p, m0, 0, 0, (p.K + 31) / 32, smem, acc);
  __syncthreads();
  H* sP = reinterpret_cast<H*>(smem);      // [256][HPITCH]
  float bs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bs[j] = bias[j * 16 + fr];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row0 = wave * 32 + i * 16 + fq * 4;           // this lane's 4 rows: row0 .. row0+3
    const int mrow = m0 + row0;
    const int bidx = min(mrow, p.M - 1) / HW;               // rows of one lane never straddle images when HW % 4 == 0
    float cen[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cen[j] = centers[(long long)bidx * HBN_ + j * 16 + fr];
    float pr[4][TN];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < TN; ++j) { pr[r][j] = acc[i][j][r] + bs[j]; mx = fmaxf(mx, pr[r][j]); }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) { pr[r][j] = __expf(pr[r][j] - mx); s += pr[r][j]; }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
      const float inv = 1.f / s;
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) { pr[r][j] *= inv; dot = fmaf(pr[r][j], cen[j], dot); }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) dot += __shfl_xor(dot, o, 64);
      if (fr == 0 && mrow + r < p.M) pred[mrow + r] = dot;
    }
    if (prob) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        uint32_t lo = pack2<H>(pr[0][j], pr[1][j]);
        uint32_t hi = pack2<H>(pr[2][j], pr[3][j]);
        uint2 v = {lo, hi};
        *reinterpret_cast<uint2*>(sP + (j * 16 + fr) * HPITCH + row0) = v;
      }
    }
  }
  if (!prob) return;
  __syncthreads();
  constexpr int CH = HBM_ / 8;
  for (int q = tid; q < HBN_ * CH; q += 256) {
    const int n = q / CH, ch = q % CH;
    const int m = m0 + ch * 8;
    if (m >= p.M) continue;
    const int b = m / HW, hw = m % HW;                 // HW % 8 == 0: a chunk stays inside one image
    *reinterpret_cast<u32x4*>(prob + ((long long)b * HBN_ + n) * HW + hw) = *reinterpret_cast<const u32x4*>(sP + n * HPITCH + ch * 8);
  }
}
}  // namespace

extern "C" int cfp_bin_head_fused(const void* x, int x_ld, const void* w, const float* bias, const float* centers,
                                  void* prob, float* pred, int B, int HW, int Cin, int dtype, cfp_stream_t stream) {
  if (dtype == CFP_F32X3) {      // float32 tensors, f16x3 matrix math, float32 prob: conv_igemm_x3.hip
    CFP_REQUIRE(x && w && bias && centers && pred, CFP_EINVAL, "cfp_bin_head_fused: null pointer");
    CFP_REQUIRE(B > 0 && HW > 0 && HW % 4 == 0 && Cin > 0 && Cin % 4 == 0 && x_ld % 4 == 0 && x_ld >= Cin && (long long)B * HW < (1ll << 31), CFP_ESHAPE,
                "cfp_bin_head_fused: bad shape (HW and Cin must be multiples of 4)");
    CFP_REQUIRE(aligned16(x) && aligned16(w) && aligned16(prob) && aligned16(bias) && aligned16(centers), CFP_EINVAL, "cfp_bin_head_fused: pointers must be 16-byte aligned");
    int rc = bin_head_x3_launch(x, x_ld, w, bias, centers, (float*)prob, pred, B, HW, Cin, reinterpret_cast<hipStream_t>(stream));
    CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_bin_head_fused: f16x3 kernel launch failed");
    return cfp_check_launch("cfp_bin_head_fused");
  }
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_bin_head_fused: bf16/f16 or CFP_F32X3 (use cfp_conv2d_nhwc + cfp_bin_softmax for plain f32)");
  CFP_REQUIRE(x && w && bias && centers && pred, CFP_EINVAL, "cfp_bin_head_fused: null pointer");
  CFP_REQUIRE(B > 0 && HW > 0 && HW % 8 == 0 && Cin > 0 && Cin % 8 == 0 && x_ld % 8 == 0 && x_ld >= Cin &&
                  (long long)B * HW < (1ll << 31), CFP_ESHAPE, "cfp_bin_head_fused: bad shape (HW and Cin must be multiples of 8)");
  CFP_REQUIRE(aligned16(x) && aligned16(w) && aligned16(prob), CFP_EINVAL, "cfp_bin_head_fused: pointers must be 16-byte aligned");
  ConvP p;
  p.in = x; p.w = w; p.out = nullptr; p.res = nullptr; p.scale = nullptr; p.shift = nullptr;
  p.in_ld = x_ld; p.out_ld = 0; p.res_ld = 0;
  p.B = 1; p.H = 1; p.W = B * HW; p.Cin = Cin; p.Ho = 1; p.Wo = B * HW; p.Cout = HBN_;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad_t = 0; p.pad_l = 0; p.M = B * HW; p.K = Cin; p.act = 0; p.pointwise = 1;
  p.ln_gamma = nullptr; p.ln_beta = nullptr; p.ln_eps = 0.f; p.rows_per_batch = 0; p.w_bstride = 0; p.f16 = dtype == CFP_F16; p.dil = 1; p.mom = nullptr; p.probe = 0;
  size_t lds = (size_t)HBN_ * HPITCH * sizeof(bf16_t);
  size_t ops_lds = 2 * (HBM_ + HBN_) * 64;
  if (lds < ops_lds) lds = ops_lds;
  hipError_t e = dtype == CFP_F16
      ? hipFuncSetAttribute((const void*)bin_head_fused_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipFuncSetAttribute((const void*)bin_head_fused_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { cfp_set_error(std::string("cfp_bin_head_fused: ") + hipGetErrorString(e)); return CFP_EHIP; }
  if (dtype == CFP_F16)
    hipLaunchKernelGGL(bin_head_fused_kernel<f16_t>, dim3(cdiv(p.M, HBM_)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p, bias,
                       centers, (f16_t*)prob, pred, HW);
  else
    hipLaunchKernelGGL(bin_head_fused_kernel<bf16_t>, dim3(cdiv(p.M, HBM_)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p, bias,
                       centers, (bf16_t*)prob, pred, HW);
  return cfp_check_launch("cfp_bin_head_fused");
}

// ---- spatial mean of a 3x3 convolution's output WITHOUT running the convolution -----------------------------------------
// The bin-width regressor starts with mean_HW(conv1x1(unet)) (decoder.py:28-29), and unet = decoder.conv0(t) is a LINEAR 3x3
// convolution (bias, no BatchNorm, no activation, zero padding 1; decoder.py:126).  Its spatial sum is therefore a function of nine
// shifted sums of its 32-channel INPUT t:   sum_px unet[m] = HW * b[m] + sum_{tap, c} W[m][tap][c] * S_tap[c],  where S_tap is the sum
// of t over the image minus the border row / column the tap (dy, dx) never reads (plus the corner pixel subtracted twice):
//     S(dy,dx) = T - [dy=+1] R_first - [dy=-1] R_last - [dx=+1] C_first - [dx=-1] C_last + corner.
// T comes from cfp_channel_sum over t (39 MB instead of the 157 MB of unet); the kernel below adds up the four border lines itself,
// uses the float32 master weights -- and the whole regressor branch now depends on t only, so it runs BESIDE conv0 instead of after
// it: its latency chain (sums -> 4 dense layers -> bin centres) leaves the critical path conv0 -> head.
namespace {

template <typename T>
__global__ __launch_bounds__(1024) void conv3x3_mean_kernel(const float* __restrict__ partial, int nsplit, const T* __restrict__ in, int in_ld,
                                                            const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ msum,
                                                            int H, int W, int C, int Cout) {
  extern __shared__ float sm[];
  float* q5 = sm;                  // [5][C]: all pixels, first row, last row, first column, last column
  float* corner = sm + 5 * C;      // [4][C]: (0,0), (0,W-1), (H-1,0), (H-1,W-1)
  float* S = corner + 4 * C;       // [9][C]
  float* red = S + 9 * C;          // [5][1024]
  const int b = blockIdx.x, tid = threadIdx.x;
  const T* img = in + (long long)b * H * W * in_ld;
  // five sums per channel, each by 1024 / C lanes in a fixed order: the channel sums of the splits, then the four border lines.
  // All loads of a thread are independent (four accumulators per line, everything requested before it is used) and the five
  // reductions share one pair of barriers: as a chain of dependent loads this kernel took 57 us for 8 workgroups.
  const int L = 1024 / C;          // host: C divides 1024
  const int c = tid % C, ln = tid / C;
  float s5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = ln;
    for (; j + 3 * L < nsplit; j += 4 * L) {
      const float v0 = partial[((long long)b * nsplit + j) * C + c], v1 = partial[((long long)b * nsplit + j + L) * C + c];
      const float v2 = partial[((long long)b * nsplit + j + 2 * L) * C + c], v3 = partial[((long long)b * nsplit + j + 3 * L) * C + c];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; j < nsplit; j += L) a0 += partial[((long long)b * nsplit + j) * C + c];
    s5[0] = (a0 + a1) + (a2 + a3);
  }
#pragma unroll
  for (int k = 1; k < 5; ++k) {
    const int n = k <= 2 ? W : H;
    const long long stride = (k <= 2 ? 1ll : (long long)W) * in_ld;                       // along the row (k = 1, 2) or down the column (k = 3, 4)
    const T* line = img + (k == 2 ? (long long)(H - 1) * W * in_ld : 0) + (k == 4 ? (long long)(W - 1) * in_ld : 0) + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = ln;
    for (; j + 3 * L < n; j += 4 * L) {
      const T v0 = line[j * stride], v1 = line[(j + L) * stride], v2 = line[(j + 2 * L) * stride], v3 = line[(j + 3 * L) * stride];
      a0 += to_f32<T>(v0); a1 += to_f32<T>(v1); a2 += to_f32<T>(v2); a3 += to_f32<T>(v3);
    }
    for (; j < n; j += L) a0 += to_f32<T>(line[j * stride]);
    s5[k] = (a0 + a1) + (a2 + a3);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k * 1024 + tid] = s5[k];
  __syncthreads();
  for (int i = tid; i < 5 * C; i += 1024) {
    const int k = i / C, cc = i - k * C;
    float t = 0.f;
    for (int j = 0; j < L; ++j) t += red[k * 1024 + j * C + cc];
    q5[i] = t;
  }
  __syncthreads();
  for (int i = tid; i < 4 * C; i += 1024) {
    const int k = i / C, cc = i - k * C;
    const int y = (k >> 1) ? H - 1 : 0, x = (k & 1) ? W - 1 : 0;
    corner[i] = to_f32<T>(img[((long long)y * W + x) * in_ld + cc]);
  }
  __syncthreads();
  for (int i = tid; i < 9 * C; i += 1024) {
    const int tap = i / C, cc = i - tap * C;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    float s = q5[cc];
    if (dy == 1) s -= q5[C + cc];
    if (dy == -1) s -= q5[2 * C + cc];
    if (dx == 1) s -= q5[3 * C + cc];
    if (dx == -1) s -= q5[4 * C + cc];
    if (dy != 0 && dx != 0) s += corner[((dy == -1 ? 2 : 0) + (dx == -1 ? 1 : 0)) * C + cc];
    S[i] = s;
  }
  __syncthreads();
  // out[m] = HW * bias[m] + W[m][0 .. 9C) . S: eight lanes per output, each a contiguous eighth of the row, its weights fetched as
  // 16-byte vectors that are all in flight before the first multiply
  const int K = 9 * C;
  const int per = (K + 7) >> 3;
  for (int m = tid >> 3; m < Cout; m += 128) {
    const int part = tid & 7;
    const int k0 = part * per, k1 = min(K, k0 + per);
    float s = 0.f;
    if ((per & 3) == 0 && per <= 64 && k1 - k0 == per) {
      const f32x4* wv = reinterpret_cast<const f32x4*>(w + (long long)m * K + k0);
      f32x4 buf[16];
#pragma unroll
      for (int q4 = 0; q4 < 16; ++q4) buf[q4] = (q4 * 4 < per) ? wv[q4] : f32x4{0.f, 0.f, 0.f, 0.f};
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int q4 = 0; q4 < 16; ++q4) {
        if (q4 * 4 < per) {
          a0 = fmaf(buf[q4][0], S[k0 + q4 * 4], a0); a1 = fmaf(buf[q4][1], S[k0 + q4 * 4 + 1], a1);
          a2 = fmaf(buf[q4][2], S[k0 + q4 * 4 + 2], a2); a3 = fmaf(buf[q4][3], S[k0 + q4 * 4 + 3], a3);
        }
      }
      s = (a0 + a1) + (a2 + a3);
    } else {
      for (int k = k0; k < k1; ++k) s = fmaf(w[(long long)m * K + k], S[k], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0) msum[(long long)b * Cout + m] = s + (float)(H * W) * (bias ? bias[m] : 0.f);
  }
}

}  // namespace

extern "C" int cfp_conv3x3_mean(const float* partial, int nsplit, const void* in, int in_ld, const float* w, const float* bias, float* msum,
                                int B, int H, int W, int C, int Cout, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(partial && in && w && msum, CFP_EINVAL, "cfp_conv3x3_mean: null pointer");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_conv3x3_mean: bad dtype");
  CFP_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0 && C <= 512 && 1024 % C == 0 && Cout > 0 && nsplit > 0 && in_ld >= C, CFP_ESHAPE,
              "cfp_conv3x3_mean: bad shape (C must divide 1024)");
  const size_t lds = (size_t)(18 * C + 5 * 1024) * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(conv3x3_mean_kernel<bf16_t>, dim3(B), dim3(1024), lds, s, partial, nsplit, (const bf16_t*)in, in_ld, w, bias, msum, H, W, C, Cout);
  else if (dtype == CFP_F16) hipLaunchKernelGGL(conv3x3_mean_kernel<f16_t>, dim3(B), dim3(1024), lds, s, partial, nsplit, (const f16_t*)in, in_ld, w, bias, msum, H, W, C, Cout);
  else hipLaunchKernelGGL(conv3x3_mean_kernel<float>, dim3(B), dim3(1024), lds, s, partial, nsplit, (const float*)in, in_ld, w, bias, msum, H, W, C, Cout);
  return cfp_check_launch("cfp_conv3x3_mean");
}
