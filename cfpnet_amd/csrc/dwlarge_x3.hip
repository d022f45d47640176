// Large-kernel depthwise convolution (Block14.dwconv2 + bn1 + relu, convnext.py:30,45-47; k = 31 / 15 / 7) in the DEFAULT numerics of the
// drop-in boundary: float32 tensors, the banded-Toeplitz GEMM of dwconv.hip's dwlarge_mfma_kernel on v_mfma_f32_16x16x32_f16 with every
// operand taken as hi + lo (A_hi B_hi + A_hi B_lo + A_lo B_hi, float32 accumulate; conv_igemm_x3.hip has the arithmetic and its error).
//
// Until round 4 this mode ran the float32 VALU kernel (dwlarge_kernel<float, 31>: 179 us per launch at the 1/4 scale, two launches per
// forward, chip filling: 269 us of the 4.7 ms step with four batches in flight, tools/ablate_time.py --x3).  Structure = the 16-bit kernel's:
//   * a workgroup = one (image, TH x TW pixel patch, channel group); the patch + halo of every channel is staged in LDS as PLANES
//     [channel][hi | lo][PH][PITCH] of halves -- split ONCE on the way in -- with a row pitch of 2 (mod 4) sixteen-byte slots (the
//     A-fragment reads of 16 rows are conflict-free, dwconv.hip dwl_pitch);
//   * a wave owns one channel: for every kernel row ky the kx taps are a banded-Toeplitz B operand [k-chunk][16 output columns], built on
//     the host as hi and lo tables (cfp_dwconv_large_toeplitz's layout, twice), streamed from global memory a few kernel rows ahead;
//     A fragment = 16 patch rows x 32 columns of the channel plane at row offset ky;
//   * FOUR channels (= one 16-byte float32 vector per pixel) and four waves per workgroup, where the 16-bit kernel has eight: the planes
//     are twice as many bytes per channel;
//   * epilogue: BN scale / shift + activation into float32 planes that reuse the LDS, then a coalesced NHWC copy-out (16 bytes per pixel).
#include "common.h"

namespace {

constexpr int dwl3_pitch(int slots) { return (slots % 4 == 2) ? slots : dwl3_pitch(slots + 1); }

template <int K, int TH, int TW>
__global__ __launch_bounds__(256) void dwlarge_x3_kernel(const float* __restrict__ in, int in_ld, const f16_t* __restrict__ tb, long long tb_lo,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out,
                                                         int out_ld, int B, int H, int W, int C, int act) {
  constexpr int HALO = (K - 1) / 2;
  constexpr int LM = (HALO + 7) / 8 * 8;
  constexpr int NH = (16 + LM + HALO + 31) / 32;
  constexpr int NTX = TW / 16, NT = (TH / 16) * NTX;
  constexpr int PH = TH + K - 1;
  constexpr int PWV = TW + LM + HALO;
  constexpr int PWA = (TW - 16) + NH * 32;
  constexpr int PW = (PWA > PWV ? PWA : PWV);
  constexpr int PITCH = dwl3_pitch((PW + 7) / 8) * 8;      // halves per plane row
  constexpr int PLANE = PH * PITCH;                        // halves per plane
  static_assert(TH * TW * 4 * 4 <= 8 * PLANE * 2, "the float32 output planes reuse the operand LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  f16_t* planes = reinterpret_cast<f16_t*>(dsm);           // [4 channels][hi, lo][PH][PITCH]
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int cgs = C / 4;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = bid % cgs; bid /= cgs;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c0 = cg * 4;
  const int y0 = ty * TH - HALO, x0 = tx * TW - LM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;

  // B fragments (hi and lo bands) of the first kernel rows are requested before the staging so their latency hides under it
  const f16_t* __restrict__ tbc = tb + (long long)(c0 + wave) * K * NH * 64 * 8 + lane * 8;
  constexpr int PF = 2;                                    // kernel rows per prefetch group
  f16x8 bnh[PF][NH], bnl[PF][NH];
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const long long o = (long long)((u < K ? u : K - 1) * NH + h) * 512;
      bnh[u][h] = *reinterpret_cast<const f16x8*>(tbc + o);
      bnl[u][h] = *reinterpret_cast<const f16x8*>(tbc + tb_lo + o);
    }

  // ---- staging: NHWC float32 -> 4 x (hi, lo) planes of halves; zeros outside the image and in the k-padding columns ----------------
  for (int i = tid; i < PH * PITCH; i += 256) {
    const int py = i / PITCH, px = i - py * PITCH;
    const int y = y0 + py, x = x0 + px;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (px < PWV && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      v = *reinterpret_cast<const f32x4*>(in + ((long long)(b * H + y) * W + x) * in_ld + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f16_t hi = f2h(v[e]);
      planes[(2 * e) * PLANE + i] = hi;
      planes[(2 * e + 1) * PLANE + i] = (f16_t)(v[e] - (float)hi);
    }
  }
  __syncthreads();

  const f16_t* plh = planes + (2 * wave) * PLANE;
  const f16_t* pll = plh + PLANE;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int abase = j * PITCH + q * 8;
#pragma unroll 1
  for (int ky0 = 0; ky0 < K; ky0 += PF) {
    f16x8 bch[PF][NH], bcl[PF][NH];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int h = 0; h < NH; ++h) { bch[u][h] = bnh[u][h]; bcl[u][h] = bnl[u][h]; }
    if (ky0 + PF < K) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kyn = min(ky0 + PF + u, K - 1);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const long long o = (long long)(kyn * NH + h) * 512;
          bnh[u][h] = *reinterpret_cast<const f16x8*>(tbc + o);
          bnl[u][h] = *reinterpret_cast<const f16x8*>(tbc + tb_lo + o);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ky = ky0 + u;
      if (ky < K) {                                        // uniform
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int o = abase + ((t / NTX) * 16 + ky) * PITCH + (t % NTX) * 16 + h * 32;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(plh + o);
            const f16x8 al = *reinterpret_cast<const f16x8*>(pll + o);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bch[u][h], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bcl[u][h], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bch[u][h], acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();      // every wave is done with the operand planes: LDS becomes four float32 output planes [channel][TH][TW + 1]
  // ---- epilogue: D lane = (column j, rows 4q .. 4q+3) of each tile -----------------------------------
  constexpr int OP = TW + 1;
  float* oplanes = reinterpret_cast<float*>(dsm);
  const float sc = scale[c0 + wave], sh = shift[c0 + wave];
  float* ow = oplanes + wave * TH * OP;
  with_act(act, [&](auto A) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) ow[((t / NTX) * 16 + 4 * q + r) * OP + (t % NTX) * 16 + j] = act_c<decltype(A)::value>(acc[t][r] * sc + sh);
  });
  __syncthreads();
  for (int i = tid; i < TH * TW; i += 256) {
    const int py = i / TW, px = i - py * TW;
    const int y = ty * TH + py, x = tx * TW + px;
    if (y < H && x < W) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = oplanes[e * TH * OP + py * OP + px];
      *reinterpret_cast<f32x4*>(out + ((long long)(b * H + y) * W + x) * out_ld + c0) = v;
    }
  }
}

template <int K, int TH, int TW>
hipError_t launch_dwlarge_x3(const void* in, int in_ld, const void* tb, long long tb_lo, const float* scale, const float* shift, void* out, int out_ld,
                             int B, int H, int W, int C, int act, hipStream_t s) {
  constexpr int HALO = (K - 1) / 2, LM = (HALO + 7) / 8 * 8, NH = (16 + LM + HALO + 31) / 32;
  constexpr int PH = TH + K - 1, PWV = TW + LM + HALO, PWA = (TW - 16) + NH * 32, PW = (PWA > PWV ? PWA : PWV);
  constexpr int PITCH = dwl3_pitch((PW + 7) / 8) * 8;
  constexpr size_t lds = (size_t)8 * PH * PITCH * 2;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)dwlarge_x3_kernel<K, TH, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr = true;
  }
  long long blocks = (long long)B * cdiv(H, TH) * cdiv(W, TW) * (C / 4);
  hipLaunchKernelGGL((dwlarge_x3_kernel<K, TH, TW>), dim3((unsigned)blocks), dim3(256), lds, s, (const float*)in, in_ld, (const f16_t*)tb, tb_lo, scale,
                     shift, (float*)out, out_ld, B, H, W, C, act);
  return hipSuccess;
}

}  // namespace

// cfp_dwconv_large_mfma_nhwc for dtype CFP_F32X3 (dwconv.hip dispatches here): float32 in / out, `toeplitz` = [hi table | lo table] of halves,
// each cfp_dwconv_large_toeplitz_elems(C, k) long.
int g_dwl3_small31 = 1;      // default since round 5: 74.6 vs 113.2 us alone, 70.3 vs 90.5 with four copies (tools/probes/dwl3_tile_probe.py), bit-identical
void cfp_dwl3_debug_set(int value) { g_dwl3_small31 = value; }
hipError_t dwlarge_x3_launch(const void* in, int in_ld, const void* toeplitz, long long table_elems, const float* scale, const float* shift, void* out,
                             int out_ld, int B, int H, int W, int C, int k, int act, hipStream_t s) {
  // k = 31: the 64 x 32 pixel tile takes 120 KB of LDS (one workgroup per CU: staging, products and the output pass of a workgroup serialise),
  // the 32 x 32 tile 79 KB (two per CU) at 1.3x the halo rows per output row -- cfp_debug_set(30, v): 0 = 64 x 32, 1 = 32 x 32
  if (k == 31 && g_dwl3_small31) return launch_dwlarge_x3<31, 32, 32>(in, in_ld, toeplitz, table_elems, scale, shift, out, out_ld, B, H, W, C, act, s);
  if (k == 31) return launch_dwlarge_x3<31, 64, 32>(in, in_ld, toeplitz, table_elems, scale, shift, out, out_ld, B, H, W, C, act, s);
  if (k == 15) return launch_dwlarge_x3<15, 32, 32>(in, in_ld, toeplitz, table_elems, scale, shift, out, out_ld, B, H, W, C, act, s);
  return launch_dwlarge_x3<7, 32, 32>(in, in_ld, toeplitz, table_elems, scale, shift, out, out_ld, B, H, W, C, act, s);
}
