// Depthwise convolutions in NHWC.
//
// dw3x3  : the encoder's 24 depthwise 3x3 convs (C = 224 ... 1392).  4.3 FLOP/byte -> HBM bound.
//          One thread owns a 16-byte channel vector of PW consecutive output pixels; consecutive
//          threads own consecutive channel vectors, so every load/store instruction of a wave is
//          a run of full 16-byte-per-lane accesses along the channel axis.  Row re-use (3x) is
//          served by L2; column re-use by registers.
// dwlarge: LKPM's 7x7 / 15x15 / 31x31 depthwise convs (C = 128 / 64 / 32).  Up to 480 FLOP/byte
//          -> vector-FMA bound; design notes at the kernel.
#include "common.h"

namespace {

template <typename T, int STRIDE, int PW>
__global__ __launch_bounds__(256) void dw3x3_kernel(const T* __restrict__ in, int in_ld, const T* __restrict__ w,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    T* __restrict__ out, int out_ld, int B, int H, int W, int C,
                                                    int pad_t, int pad_l, int Ho, int Wo, int act) {
  constexpr int VE = Vec<T>::N;
  constexpr int IW = (PW - 1) * STRIDE + 3;   // input columns feeding PW outputs
  const int CV = C / VE;
  const int WG = (Wo + PW - 1) / PW;
  const long long total = (long long)B * Ho * WG * CV;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int cv = (int)(idx % CV);
    long long t = idx / CV;
    int wg = (int)(t % WG);
    t /= WG;
    int ho = (int)(t % Ho);
    int b = (int)(t / Ho);
    const int c0 = cv * VE;
    const int wo0 = wg * PW;
    const int hi0 = ho * STRIDE - pad_t;
    const int wi0 = wo0 * STRIDE - pad_l;

    float acc[PW][VE];
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[p][e] = 0.f;

#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = hi0 + kh;
      if (hi < 0 || hi >= H) continue;
      float wv[3][VE];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) Vec<T>::load(w + (kh * 3 + kw) * C + c0, wv[kw]);
      const T* row = in + ((long long)(b * H + hi) * W) * in_ld + c0;
#pragma unroll
      for (int x = 0; x < IW; ++x) {
        const int wi = wi0 + x;
        float v[VE];
        if (wi >= 0 && wi < W) {
          Vec<T>::load(row + (long long)wi * in_ld, v);
        } else {
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] = 0.f;
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          const int kw = x - p * STRIDE;   // compile-time after unrolling
          if (kw >= 0 && kw < 3) {
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[p][e] = fmaf(v[e], wv[kw][e], acc[p][e]);
          }
        }
      }
    }
    float sc[VE], sh[VE];
    Vec<float>::load(scale + c0, sc);
    Vec<float>::load(shift + c0, sh);
    if constexpr (VE == 8) {
      Vec<float>::load(scale + c0 + 4, sc + 4);
      Vec<float>::load(shift + c0 + 4, sh + 4);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p) {
      const int wo = wo0 + p;
      if (wo >= Wo) break;
      float o[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = apply_act(acc[p][e] * sc[e] + sh[e], act);
      Vec<T>::store(out + ((long long)(b * Ho + ho) * Wo + wo) * out_ld + c0, o);
    }
  }
}

// ---- large kernel ------------------------------------------------------------------------
// Workgroup = 512 threads = 8 waves; wave w owns channel c0 + w of a 32-row x 32-column output
// tile, lane = (column, 16-row strip).  The input tile + halo sits in LDS planar per channel
// ([8][32+K-1][32+K-1] f32) so the 64 lanes of a wave read 2 x 32 consecutive floats:
// conflict-free ds_read_b32.  The channel is wave-uniform, so the K*K weights come through the
// scalar cache (s_load) and cost no LDS or VALU slot: the inner loop is 16 v_fma per LDS read.
// The vertical sliding window lives in a 32-register ring addressed with compile-time indices (K
// is a template parameter, the ky loop is fully unrolled): no register moves, and each row is
// fetched LPRE steps before its first use so the LDS latency hides behind ~128 FMAs.
constexpr int LTH = 32, LTW = 32, LC = 8, LSTRIP = 16, LPRE = 8;

template <typename T, int K>
__global__ __launch_bounds__(512) void dwlarge_kernel(const T* __restrict__ in, int in_ld,
                                                      const float* __restrict__ w, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, T* __restrict__ out,
                                                      int out_ld, int B, int H, int W, int C, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HALO = (K - 1) / 2;
  constexpr int TPH = LTH + K - 1, TPW = LTW + K - 1;
  constexpr int VE = Vec<T>::N;
  constexpr int VPP = LC / VE;
  float* sIn = lds;                         // [LC][TPH][TPW]

  const int tiles_x = (W + LTW - 1) / LTW, tiles_y = (H + LTH - 1) / LTH;
  const int cgs = C / LC;
  int bid = blockIdx.x;
  const int cg = bid % cgs; bid /= cgs;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c0 = cg * LC;
  const int y0 = ty * LTH - HALO, x0 = tx * LTW - HALO;
  const int tid = threadIdx.x;

  for (int i = tid; i < TPH * TPW * VPP; i += 512) {
    const int v = i % VPP, pix = i / VPP;
    const int py = pix / TPW, px = pix % TPW;
    const int y = y0 + py, x = x0 + px;
    float vals[VE];
    if (y >= 0 && y < H && x >= 0 && x < W) {
      Vec<T>::load(in + ((long long)(b * H + y) * W + x) * in_ld + c0 + v * VE, vals);
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) vals[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) sIn[(v * VE + e) * (TPH * TPW) + pix] = vals[e];
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int col = lane & 31, strip = lane >> 5;
  const float* __restrict__ wc = w + (long long)(c0 + wave) * (K * K);   // [kx][ky], wave-uniform
  const float* base = sIn + wave * (TPH * TPW) + (strip * LSTRIP) * TPW + col;
  constexpr int NROW = LSTRIP + K - 1;     // input rows a strip touches
  float acc[LSTRIP];
#pragma unroll
  for (int i = 0; i < LSTRIP; ++i) acc[i] = 0.f;

  for (int kx = 0; kx < K; ++kx) {
    const float* colp = base + kx;
    float win[32];
#pragma unroll
    for (int i = 0; i < LSTRIP - 1 + LPRE; ++i)
      if (i < NROW) win[i & 31] = colp[i * TPW];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      constexpr int AHEAD = LSTRIP - 1 + LPRE;
      if (ky + AHEAD < NROW) win[(ky + AHEAD) & 31] = colp[(ky + AHEAD) * TPW];
      const float wv = wc[kx * K + ky];
#pragma unroll
      for (int i = 0; i < LSTRIP; ++i) acc[i] = fmaf(win[(ky + i) & 31], wv, acc[i]);
    }
  }
  const float sc = scale[c0 + wave], sh = shift[c0 + wave];
  __syncthreads();                          // everyone is done reading sIn: reuse it as [pixel][8]
  float* sOut = lds;
#pragma unroll
  for (int i = 0; i < LSTRIP; ++i) sOut[((strip * LSTRIP + i) * LTW + col) * LC + wave] = apply_act(acc[i] * sc + sh, act);
  __syncthreads();
  for (int pix = tid; pix < LTH * LTW; pix += 512) {
    const int py = pix / LTW, px = pix % LTW;
    const int y = ty * LTH + py, x = tx * LTW + px;
    if (y < H && x < W) {
      float v[LC];
      Vec<float>::load(sOut + pix * LC, v);
      Vec<float>::load(sOut + pix * LC + 4, v + 4);
      T* dp = out + ((long long)(b * H + y) * W + x) * out_ld + c0;
#pragma unroll
      for (int q = 0; q < VPP; ++q) Vec<T>::store(dp + q * VE, v + q * VE);
    }
  }
}

// generic odd k (3..31): 16x16 tile, weights staged in LDS
constexpr int LT = 16;

template <typename T>
__global__ __launch_bounds__(256) void dwlarge_generic_kernel(const T* __restrict__ in, int in_ld,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, T* __restrict__ out,
                                                              int out_ld, int B, int H, int W, int C, int k, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int halo = (k - 1) / 2;
  const int TP = LT + k - 1;
  float* sIn = lds;                     // [TP][TP][LC]
  float* sW = lds + TP * TP * LC;       // [ky*k+kx][LC]
  const int tiles_x = (W + LT - 1) / LT, tiles_y = (H + LT - 1) / LT;
  const int cgs = C / LC;
  int bid = blockIdx.x;
  const int cg = bid % cgs; bid /= cgs;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c0 = cg * LC;
  const int y0 = ty * LT - halo, x0 = tx * LT - halo;
  const int tid = threadIdx.x;
  for (int i = tid; i < k * k * LC; i += 256) {
    const int c = i % LC, t = i / LC, ky = t / k, kx = t % k;
    sW[i] = w[(long long)(c0 + c) * k * k + kx * k + ky];
  }
  constexpr int VE = Vec<T>::N;
  constexpr int VPP = LC / VE;
  for (int i = tid; i < TP * TP * VPP; i += 256) {
    int v = i % VPP, pix = i / VPP;
    int py = pix / TP, px = pix % TP;
    int y = y0 + py, x = x0 + px;
    float vals[VE];
    if (y >= 0 && y < H && x >= 0 && x < W) {
      Vec<T>::load(in + ((long long)(b * H + y) * W + x) * in_ld + c0 + v * VE, vals);
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) vals[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) sIn[pix * LC + v * VE + e] = vals[e];
  }
  __syncthreads();
  const int c = tid & 7;
  const int col = (tid >> 3) & 15;
  const int r0 = (tid >> 7) * 8;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int kx = 0; kx < k; ++kx) {
    const float* colp = sIn + ((r0 * TP) + col + kx) * LC + c;
    float win[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) win[i + 1] = colp[i * TP * LC];
    for (int ky = 0; ky < k; ++ky) {
#pragma unroll
      for (int i = 0; i < 7; ++i) win[i] = win[i + 1];
      win[7] = colp[(ky + 7) * TP * LC];
      const float wv = sW[(ky * k + kx) * LC + c];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(win[i], wv, acc[i]);
    }
  }
  const float sc = scale[c0 + c], sh = shift[c0 + c];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int y = ty * LT + r0 + i, x = tx * LT + col;
    if (y < H && x < W) out[((long long)(b * H + y) * W + x) * out_ld + c0 + c] = from_f32<T>(apply_act(acc[i] * sc + sh, act));
  }
}

template <typename T, int K>
hipError_t launch_dwlarge(const void* in, int in_ld, const float* w, const float* scale, const float* shift, void* out,
                          int out_ld, int B, int H, int W, int C, int act, hipStream_t s) {
  constexpr int TPH = LTH + K - 1, TPW = LTW + K - 1;
  size_t lds = (size_t)LC * TPH * TPW * sizeof(float);
  if (lds < (size_t)LTH * LTW * LC * sizeof(float)) lds = (size_t)LTH * LTW * LC * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)dwlarge_kernel<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long long blocks = (long long)B * cdiv(H, LTH) * cdiv(W, LTW) * (C / LC);
  hipLaunchKernelGGL((dwlarge_kernel<T, K>), dim3((unsigned)blocks), dim3(512), lds, s, (const T*)in, in_ld, w, scale, shift,
                     (T*)out, out_ld, B, H, W, C, act);
  return hipSuccess;
}

template <typename T>
hipError_t launch_dwlarge_any(const void* in, int in_ld, const float* w, const float* scale, const float* shift, void* out,
                              int out_ld, int B, int H, int W, int C, int k, int act, hipStream_t s) {
  switch (k) {
    case 7: return launch_dwlarge<T, 7>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, act, s);
    case 15: return launch_dwlarge<T, 15>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, act, s);
    case 31: return launch_dwlarge<T, 31>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, act, s);
    default: break;
  }
  const int TP = LT + k - 1;
  size_t lds = (size_t)(TP * TP * LC + k * k * LC) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)dwlarge_generic_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long long blocks = (long long)B * cdiv(H, LT) * cdiv(W, LT) * (C / LC);
  hipLaunchKernelGGL(dwlarge_generic_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, s, (const T*)in, in_ld, w, scale, shift,
                     (T*)out, out_ld, B, H, W, C, k, act);
  return hipSuccess;
}

}  // namespace

extern "C" int cfp_dwconv3x3_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                                  void* out, int out_ld, int B, int H, int W, int C, int stride, int pad_t, int pad_l,
                                  int Ho, int Wo, int act, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(in && w && out && scale && shift, CFP_EINVAL, "cfp_dwconv3x3_nhwc: null pointer");
  CFP_REQUIRE(dtype == CFP_F32 || dtype == CFP_BF16, CFP_EINVAL, "cfp_dwconv3x3_nhwc: bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(stride == 1 || stride == 2, CFP_ESHAPE, "cfp_dwconv3x3_nhwc: stride must be 1 or 2");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && in_ld % ve == 0 && out_ld % ve == 0 &&
                  in_ld >= C && out_ld >= C, CFP_ESHAPE, "cfp_dwconv3x3_nhwc: bad shape");
  CFP_REQUIRE(aligned16(in) && aligned16(w) && aligned16(out) && aligned16(scale) && aligned16(shift), CFP_EINVAL,
              "cfp_dwconv3x3_nhwc: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  constexpr int PW = 4;
  long long total = (long long)B * Ho * cdiv(Wo, PW) * (C / ve);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
#define DW_LAUNCH(T, S)                                                                                            \
  hipLaunchKernelGGL((dw3x3_kernel<T, S, PW>), dim3(blocks), dim3(256), 0, s, (const T*)in, in_ld, (const T*)w,    \
                     scale, shift, (T*)out, out_ld, B, H, W, C, pad_t, pad_l, Ho, Wo, act)
  if (dtype == CFP_BF16) { if (stride == 1) DW_LAUNCH(bf16_t, 1); else DW_LAUNCH(bf16_t, 2); }
  else { if (stride == 1) DW_LAUNCH(float, 1); else DW_LAUNCH(float, 2); }
#undef DW_LAUNCH
  return cfp_check_launch("cfp_dwconv3x3_nhwc");
}

extern "C" int cfp_dwconv_large_nhwc(const void* in, int in_ld, const float* w, const float* scale,
                                     const float* shift, void* out, int out_ld, int B, int H, int W, int C, int k,
                                     int act, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(in && w && out && scale && shift, CFP_EINVAL, "cfp_dwconv_large_nhwc: null pointer");
  CFP_REQUIRE(dtype == CFP_F32 || dtype == CFP_BF16, CFP_EINVAL, "cfp_dwconv_large_nhwc: bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(k >= 3 && k <= 31 && (k & 1), CFP_ESHAPE, "cfp_dwconv_large_nhwc: k must be odd, 3..31");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && in_ld % ve == 0 && out_ld % ve == 0 && in_ld >= C &&
                  out_ld >= C, CFP_ESHAPE, "cfp_dwconv_large_nhwc: bad shape");
  CFP_REQUIRE(aligned16(in) && aligned16(out), CFP_EINVAL, "cfp_dwconv_large_nhwc: pointers must be 16-byte aligned");
  CFP_REQUIRE((long long)B * cdiv(H, 16) * cdiv(W, 16) * (C / 8) < (1ll << 31), CFP_ESHAPE, "cfp_dwconv_large_nhwc: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = dtype == CFP_BF16
      ? launch_dwlarge_any<bf16_t>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, k, act, s)
      : launch_dwlarge_any<float>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, k, act, s);
  if (e != hipSuccess) { cfp_set_error(std::string("cfp_dwconv_large_nhwc: ") + hipGetErrorString(e)); return CFP_EHIP; }
  return cfp_check_launch("cfp_dwconv_large_nhwc");
}
