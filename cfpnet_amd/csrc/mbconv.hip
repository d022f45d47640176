// Inverted-residual front half as ONE kernel (stride-1 blocks of the RGB encoder, timm InvertedResidual as restated in
// oracle/cfpnet_oracle.py:88-96; reference call site /root/reference/src/models/encoder.py:66-69):
//
//     conv_pw 1x1 (Cin -> mid) + BN1 + SiLU   ->   conv_dw 3x3 (depthwise) + BN2 + SiLU   ->   mid2  (+ per-tile channel sums for squeeze-excite)
//
// Before: a pointwise GEMM wrote the expanded tensor (up to 16 MB per block at batch 8), the depthwise kernel read it back in 128-byte
// channel slices 1.3-2.8 KB apart (2.3x fetch amplification).  Here the expanded tensor exists only as a [pixels][64
// channels] tile in LDS:
//   * a workgroup owns a TH x TW tile of output pixels of one image and a contiguous range of mid channels; the input pixels of the
//     tile + a one-pixel halo ([(TH+2)(TW+2)][Cin], 16-bit) are loaded ONCE into LDS as the MFMA A operand;
//   * per chunk of 64 mid channels: the chunk's expand weights arrive by LDS-DMA (the host stores them in the LDS image layout:
//     rows of KP + 8 elements, so a linear copy gives bank-conflict-free rows) while the previous chunk's depthwise phase runs;
//     GEMM [pixels incl. halo] x [64] x Cin on the matrix cores -> BN1 + SiLU -> halo pixels outside the image forced to ZERO (the
//     depthwise conv pads its INPUT, i.e. this tensor, with zeros) -> 16-bit tile in LDS;
//   * depthwise 3x3 on that tile with the diagonal-weight MFMA trick of dw3x3_mfma_kernel (5 MFMAs per 16 channels x 16 pixels),
//     BN2 + SiLU, channel sums, results staged in LDS and written as 16-byte vectors.
// The expand GEMM is recomputed for the halo ((TH+2)(TW+2) / (TH TW) = 1.6x at 6x10): irrelevant at ~2 GFLOP per block.
//
// STATUS (round 2): correct (tests/test_ops_gpu.py::test_mbconv_expand_dw_fused, and the whole model with CFP_MBCONV_FUSED=1) but NOT
// faster than the two kernels it replaces: 27 / 38 / 40 / 38 us against 27 / 35 / 40 / 24 us alone (30x40 mid 448 / 672 / 816, 15x20
// mid 1392, batch 8) and 19-32 us against 12-23 us with four copies side by side (tools/mbconv_bench.py).  A workgroup runs its
// phases one after the other on four waves (GEMM -> BN + SiLU -> 16-bit tile -> depthwise -> BN + SiLU -> store), the SiLU of the
// expanded tensor is evaluated on 1.6x the pixels, and with 76-100 KB of LDS at most two workgroups share a CU: what the fusion saves in
// HBM traffic it loses in latency chains.  The engine keeps it OFF by default; making it pay needs wave specialisation (GEMM waves
// feeding depthwise waves through a ring of tiles) -- DESIGN.md 8.
#include "common.h"

namespace {

constexpr int MB_CH = 64;                  // mid channels per chunk
constexpr int MB_PM = MB_CH * 2 + 16;      // pixel pitch of the mid tile in bytes (144 / 16 = 9: odd -> conflict-free 16-byte reads of 16 rows)
constexpr int MB_MAXPX = 128;              // (TH + 2) * (TW + 2) <= 128 halo'd pixels (8 m-tiles), TH * TW <= 64 output pixels

struct MbP {
  const void* x; const void* wpw; const float* s1; const float* t1;
  const void* wdw; const float* s2; const float* t2;
  void* out; float* partial;
  int x_ld, out_ld;
  int B, H, W, Cin, KP, mid;
  int TH, TW, tiles_y, tiles_x;
  int ntg;                                 // 16-channel tiles per channel group (blockIdx.y)
};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
__device__ __attribute__((aligned(16))) unsigned int g_mb_zero[4] = {0u, 0u, 0u, 0u};

template <typename HT>
__global__ __launch_bounds__(256) void mbconv_expand_dw_kernel(MbP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int pitchX = p.KP * 2 + 16;                      // bytes per pixel row of the x tile and per weight row
  const int CW = p.TW + 2;                                // halo'd tile width
  const int PXI = (p.TH + 2) * CW;                        // halo'd pixels
  const int MT = (PXI + 15) >> 4;                         // m-tiles of the expand GEMM
  unsigned char* sX = smem;                               // [MT*16][pitchX]
  unsigned char* sW = sX + MT * 16 * pitchX;              // [64][pitchX]: one buffer (two workgroups per CU hide the DMA better than a second buffer)
  unsigned char* sM = sW + MB_CH * pitchX;                // [MT*16][MB_PM]
  unsigned char* sO = sM + MT * 16 * MB_PM;               // [64 output pixels][128 bytes]
  float* sSum = reinterpret_cast<float*>(sO + 64 * 128);  // [64]

  const int tile = blockIdx.x % (p.tiles_y * p.tiles_x), b = blockIdx.x / (p.tiles_y * p.tiles_x);
  const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const int y0 = ty * p.TH, x0 = tx * p.TW;
  const int th = min(p.TH, p.H - y0), tw = min(p.TW, p.W - x0);
  const int NT = p.mid >> 4;
  const int nt_begin = blockIdx.y * p.ntg, nt_end = min(NT, nt_begin + p.ntg);
  const HT* __restrict__ x = reinterpret_cast<const HT*>(p.x);
  const HT* __restrict__ wpw = reinterpret_cast<const HT*>(p.wpw);
  const HT* __restrict__ wdw = reinterpret_cast<const HT*>(p.wdw);
  HT* __restrict__ out = reinterpret_cast<HT*>(p.out);

  // weights of chunk c -> buffer c & 1: rows [ch0, ch0 + 16 n) x pitchX bytes are contiguous in global memory
  auto issue_w = [&](int c) {
    const int n0 = nt_begin + c * 4;
    const int rows = min(4, nt_end - n0) * 16;
    const int bytes = rows * pitchX;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(wpw) + (long long)n0 * 16 * pitchX;
    unsigned char* dst = sW;
    for (int o = wave * 1024; o < bytes; o += 4096) {
      const int off = o + lane * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)(off < bytes ? src + off : reinterpret_cast<const unsigned char*>(g_mb_zero)), (lptr_t)(dst + o), 16, 0, 0);
    }
  };
  const int nchunks = (nt_end - nt_begin + 3) >> 2;
  if (nchunks > 0) issue_w(0);

  // ---- the input tile + halo, once ----------------------------------------------------------------------------------------------
  {
    const int kc = p.KP >> 3;                              // 16-byte chunks per pixel row (incl. the zero K padding)
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (int i = tid; i < MT * 16 * kc; i += 256) {
      const int px = i / kc, c8 = i - px * kc;
      const int iy = px / CW, ix = px - iy * CW;
      const int y = y0 - 1 + iy, xx = x0 - 1 + ix;
      const bool ok = px < PXI && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W && c8 * 8 < p.Cin;
      u32x4 v = zero4;
      if (ok) v = *reinterpret_cast<const u32x4*>(x + ((long long)(b * p.H + y) * p.W + xx) * p.x_ld + c8 * 8);
      *reinterpret_cast<u32x4*>(sX + px * pitchX + c8 * 16) = v;
    }
  }
  // which of this lane's GEMM rows (m-tile wave + 4 i, row 4 fq + r) are pixels INSIDE the image: bit 4 i + r
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int px = (wave + 4 * i) * 16 + fq * 4 + r;
      const int iy = px / CW, ix = px - iy * CW;
      const int y = y0 - 1 + iy, xx = x0 - 1 + ix;
      if (px < PXI && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) inside |= 1u << (4 * i + r);
    }

  for (int c = 0; c < nchunks; ++c) {
    const int n0 = nt_begin + c * 4;
    const int ntc = min(4, nt_end - n0);                   // 16-channel tiles of this chunk
    const int ch0 = n0 * 16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                       // weights of chunk c landed (and the x tile, first time); sM / sO / sSum of chunk c-1 consumed
    const unsigned char* cW = sW;

    // ---- expand GEMM: [pixels incl. halo] x [64 channels], K = Cin (zero-padded to KP) --------------------------------------------
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = p.KP >> 5;
    for (int ks = 0; ks < nks; ++ks) {
      s16x8 bf[4], af[2];
#pragma unroll
      for (int n = 0; n < 4; ++n) bf[n] = *reinterpret_cast<const s16x8*>(cW + (n * 16 + fr) * pitchX + (ks * 4 + fq) * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mt = wave + 4 * i;
        af[i] = *reinterpret_cast<const s16x8*>(sX + (min(mt, MT - 1) * 16 + fr) * pitchX + (ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = mfma16<HT>(af[i], bf[n], acc[i][n]);
    }
    // BN1 + SiLU; pixels outside the image are the depthwise conv's zero padding
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (n < ntc) {
        const float sc = p.s1[ch0 + n * 16 + fr], sh = p.t1[ch0 + n * 16 + fr];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mt = wave + 4 * i;
          if (mt < MT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = ((inside >> (4 * i + r)) & 1u) ? act_c<CFP_ACT_SILU>(acc[i][n][r] * sc + sh) : 0.f;
              *reinterpret_cast<HT*>(sM + (mt * 16 + fq * 4 + r) * MB_PM + (n * 16 + fr) * 2) = from_f32<HT>(v);
            }
          }
        }
      }
    }
    __syncthreads();
    if (c + 1 < nchunks) issue_w(c + 1);                   // every wave is done with this chunk's weights: the next chunk's arrive during the depthwise phase

    // ---- depthwise 3x3 + BN2 + SiLU on the tile: wave w owns the chunk's w-th 16-channel group --------------------------------------
    if (wave < ntc) {
      const int cg = ch0 + wave * 16;                      // first channel of the group
      s16x8 afr[5];                                        // diagonal weight fragments (dw3x3_mfma_kernel): A[i][(t', c')] = w[2 pr + t'][i] * delta(c', i)
      int toff[5];
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
        const int tap = 2 * pr + (fq >> 1);
        const bool on = tap < 9 && (fr >> 3) == (fq & 1);
        const short wv = on ? (short)to_bits<HT>(wdw[min(tap, 8) * p.mid + cg + fr]) : (short)0;
#pragma unroll
        for (int e = 0; e < 8; ++e) afr[pr][e] = (e == (fr & 7)) ? wv : (short)0;
        const int tc = min(tap, 8);
        toff[pr] = ((tc / 3) * CW + tc % 3) * MB_PM + (wave * 16 + (fq & 1) * 8) * 2;
      }
      float sc2[4], sh2[4], csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc2[r] = p.s2[cg + fq * 4 + r]; sh2[r] = p.t2[cg + fq * 4 + r]; }
      const int npo = p.TH * p.TW;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        if (pb * 16 < npo) {                               // uniform
          const int po = pb * 16 + fr;
          const int oy = po / p.TW, ox = po - oy * p.TW;
          const bool valid = po < npo && oy < th && ox < tw;
          const int base = valid ? (oy * CW + ox) * MB_PM : 0;
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int pr = 0; pr < 5; ++pr) {
            const s16x8 bfr = *reinterpret_cast<const s16x8*>(sM + base + toff[pr]);
            a = mfma16<HT>(afr[pr], bfr, a);
          }
          float y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            y[r] = act_c<CFP_ACT_SILU>(a[r] * sc2[r] + sh2[r]);
            csum[r] += valid ? y[r] : 0.f;
          }
          if (valid) {
            uint2 pk;
            pk.x = pack2<HT>(y[0], y[1]);
            pk.y = pack2<HT>(y[2], y[3]);
            *reinterpret_cast<uint2*>(sO + po * 128 + (wave * 16 + fq * 4) * 2) = pk;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) csum[r] += __shfl_xor(csum[r], o, 64);
      }
      if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sSum[wave * 16 + fq * 4 + r] = csum[r];
      }
    }
    __syncthreads();

    // ---- copy-out: 128 bytes per output pixel, 16-byte vectors -----------------------------------------------------------------------
    for (int i = tid; i < p.TH * p.TW * 8; i += 256) {
      const int po = i >> 3, c8 = i & 7;
      const int oy = po / p.TW, ox = po - oy * p.TW;
      if (oy < th && ox < tw && c8 * 8 < ntc * 16)
        *reinterpret_cast<u32x4*>(out + ((long long)(b * p.H + y0 + oy) * p.W + x0 + ox) * p.out_ld + ch0 + c8 * 8) =
            *reinterpret_cast<const u32x4*>(sO + po * 128 + c8 * 16);
    }
    if (p.partial && tid < ntc * 16) p.partial[((long long)b * p.tiles_y * p.tiles_x + tile) * p.mid + ch0 + tid] = sSum[tid];
  }
}

struct MbPlan { int TH, TW, G, lds; };

MbPlan mb_plan(int B, int H, int W, int Cin, int mid) {
  MbPlan pl{0, 0, 0, 0};
  const int KP = (Cin + 31) / 32 * 32;
  const int pitchX = KP * 2 + 16;
  // tile: up to 64 output pixels, (TH + 2)(TW + 2) <= 128; prefer a width that divides W
  int bestTH = 0, bestTW = 0; double best = 1e30;
  for (int tw = 4; tw <= 16; ++tw)
    for (int th = 2; th <= 16; ++th) {
      if (th * tw > 64 || (th + 2) * (tw + 2) > MB_MAXPX) continue;
      const double tiles = (double)cdiv(H, th) * cdiv(W, tw);
      const double cost = tiles * (((th + 2) * (tw + 2) + 15) / 16 * 16 + 40);   // GEMM rows incl. halo and m-tile padding + a fixed cost per tile
      if (cost < best) { best = cost; bestTH = th; bestTW = tw; }
    }
  pl.TH = bestTH; pl.TW = bestTW;
  const int MT = ((bestTH + 2) * (bestTW + 2) + 15) / 16;
  pl.lds = MT * 16 * pitchX + MB_CH * pitchX + MT * 16 * MB_PM + 64 * 128 + 64 * 4;
  const int tiles = B * cdiv(H, bestTH) * cdiv(W, bestTW);
  const int NT = mid / 16;
  int G = cdiv(500, tiles);                                 // about two workgroups per CU
  if (G < 1) G = 1;
  int ntg = cdiv(cdiv(NT, G), 4) * 4;                       // whole chunks per group
  pl.G = cdiv(NT, ntg);
  return pl;
}

}  // namespace

extern "C" int cfp_mbconv_plan(int B, int H, int W, int Cin, int mid, int* tiles_per_image, int* KP) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || mid <= 0) return CFP_ESHAPE;
  const MbPlan pl = mb_plan(B, H, W, Cin, mid);
  if (tiles_per_image) *tiles_per_image = cdiv(H, pl.TH) * cdiv(W, pl.TW);
  if (KP) *KP = (Cin + 31) / 32 * 32;
  return pl.lds <= 160 * 1024 ? CFP_OK : CFP_ESHAPE;
}

extern "C" int cfp_mbconv_expand_dw(const void* x, int x_ld, const void* wpw, const float* s1, const float* t1, const void* wdw,
                                    const float* s2, const float* t2, void* out, int out_ld, float* partial, int B, int H, int W, int Cin,
                                    int mid, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_mbconv_expand_dw: bf16 / f16 only");
  CFP_REQUIRE(x && wpw && s1 && t1 && wdw && s2 && t2 && out, CFP_EINVAL, "cfp_mbconv_expand_dw: null pointer");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && mid > 0 && mid % 16 == 0 && x_ld % 8 == 0 && x_ld >= Cin && out_ld % 8 == 0 &&
                  out_ld >= mid, CFP_ESHAPE, "cfp_mbconv_expand_dw: bad shape (Cin % 8, mid % 16)");
  CFP_REQUIRE(aligned16(x) && aligned16(wpw) && aligned16(out) && aligned16(s1) && aligned16(t1) && aligned16(s2) && aligned16(t2), CFP_EINVAL,
              "cfp_mbconv_expand_dw: pointers must be 16-byte aligned");
  const MbPlan pl = mb_plan(B, H, W, Cin, mid);
  CFP_REQUIRE(pl.TH > 0 && pl.lds <= 160 * 1024, CFP_ESHAPE, "cfp_mbconv_expand_dw: tile does not fit the LDS (Cin too large)");
  MbP p;
  p.x = x; p.wpw = wpw; p.s1 = s1; p.t1 = t1; p.wdw = wdw; p.s2 = s2; p.t2 = t2; p.out = out; p.partial = partial;
  p.x_ld = x_ld; p.out_ld = out_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.KP = (Cin + 31) / 32 * 32; p.mid = mid;
  p.TH = pl.TH; p.TW = pl.TW; p.tiles_y = cdiv(H, pl.TH); p.tiles_x = cdiv(W, pl.TW);
  p.ntg = cdiv(cdiv(mid / 16, pl.G), 4) * 4;
  const dim3 grid(B * p.tiles_y * p.tiles_x, cdiv(mid / 16, p.ntg));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MB_LAUNCH(T)                                                                                                             \
  do {                                                                                                                           \
    static bool attr = false;                                                                                                    \
    if (!attr) {                                                                                                                 \
      if (hipFuncSetAttribute((const void*)mbconv_expand_dw_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { \
        cfp_set_error("cfp_mbconv_expand_dw: cannot set the LDS size");                                                          \
        return CFP_EHIP;                                                                                                         \
      }                                                                                                                          \
      attr = true;                                                                                                               \
    }                                                                                                                            \
    hipLaunchKernelGGL(mbconv_expand_dw_kernel<T>, grid, dim3(256), (size_t)pl.lds, s, p);                                       \
  } while (0)
  if (dtype == CFP_F16) MB_LAUNCH(f16_t); else MB_LAUNCH(bf16_t);
#undef MB_LAUNCH
  return cfp_check_launch("cfp_mbconv_expand_dw");
}
