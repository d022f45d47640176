// bf16 3x3 stride-1 convolution with an LDS-resident input halo tile ("direct" convolution on the
// matrix cores).
//
// The implicit GEMM of conv_igemm2.hip fetches every input pixel nine times (once per filter tap)
// from L2 into LDS: for this network's big 3x3 layers that is 0.9-1.4 GB of L2->LDS traffic per
// launch and it, not the MFMA rate, sets their time (measured: 6-9 TB/s of operand traffic at
// 25-45 % MFMA utilisation).  Here a workgroup owns a TH x 16 patch of output pixels and stages the
// (TH+2) x 18 input halo ONCE per 64-channel chunk; the nine taps are nine shifted views of that
// tile, so the A-operand traffic drops 4-7x and the weights become the larger stream.
//
//   * K order: chunk of 64 input channels (outer) -> kernel row kh -> kernel column kw -> 2 x k32.
//     One pipeline step = (chunk, kh): the weight slab [3 kw][BN][64 ch] for it is DMA'd while the
//     previous step computes; the halo tile of the NEXT chunk is DMA'd during kh = 0 of this one.
//   * both operands use `global_load_lds_dwordx4` into 128-byte rows (one row = one halo pixel or
//     one (kw, cout) weight row) with the chunk XOR swizzle of conv_igemm2.hip, so the 16 lanes of
//     an MFMA operand read (16 consecutive pixels / 16 consecutive couts) are conflict-free;
//     out-of-image halo pixels and the channel tail read the zero word.
//   * MFMA v_mfma_f32_16x16x32_bf16; an M-tile is 16 consecutive output pixels of one row.
//   * epilogue as in conv_igemm2.hip: scale/shift, activation, residual, 16-byte channel vectors.
#include "igemm_core.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16b[4] = {0u, 0u, 0u, 0u};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// UP = true (cfp_upsample_cat_conv3x3, decoder.py:51-58 UpSampleBN: interpolate(x, size=skip, bilinear, align_corners=True) -> cat -> conv3x3):
// the 64-channel chunks below p.up_C are not fetched but COMPUTED into the halo tile -- four 16-byte taps of the low-resolution map,
// blended in float32 with resize_kernel's own arithmetic and rounded to the storage type as the stored upsampled tensor would have
// been, so the result is bit-identical to cfp_resize_bilinear + cfp_conv2d_nhwc.  The blend of chunk c + 1 is spread over the three
// kernel-row steps of chunk c (loads issued before a step's MFMAs, blend + LDS store after them); the upsampled tensor (79 MB at
// batch 8 for up4) is never written or read back, and every halo pixel is blended once per chunk, not once per tap.
template <typename H, int TH, int BN, int WM, int WN, bool UP = false>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(ConvP p) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int TW = 16;
  constexpr int BM = TH * TW;
  constexpr int TM = TH / WM;              // 16-pixel tile rows per wave
  constexpr int TN = BN / WN / 16;
  constexpr int HW_ = TW + 2;              // halo width
  constexpr int HP = (TH + 2) * HW_;       // halo pixels
  constexpr int NAG = (HP + 7) / 8;        // 8-pixel DMA groups
  constexpr int NAH = (NAG + 3) / 4;       // per wave
  constexpr int A_BYTES = NAH * 4 * 1024;
  constexpr int NBG = BN * 3 / 8;          // 8-row DMA groups of the [3][BN] weight slab
  constexpr int NBW = (NBG + 3) / 4;
  constexpr int B_BYTES = ((NBG + 3) / 4) * 4 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto sA = [&](int buf) -> unsigned char* { return smem + buf * A_BYTES; };
  auto sB = [&](int buf) -> unsigned char* { return smem + 2 * A_BYTES + buf * B_BYTES; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;

  // ---- tile coordinates: N-tiles of one pixel patch are adjacent workgroups (they share the halo in L2)
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % tiles_n; bid /= tiles_n;
  const int tx_ = bid % tiles_x; bid /= tiles_x;
  const int ty_ = bid % tiles_y;
  const int b = bid / tiles_y;
  const int n0 = tile_n * BN, x0 = tx_ * TW, y0 = ty_ * TH;

  const H* __restrict__ in = reinterpret_cast<const H*>(p.in) + (long long)b * p.H * p.W * p.in_ld;
  const H* __restrict__ wt = reinterpret_cast<const H*>(p.w);
  const H* zsrc = reinterpret_cast<const H*>(g_zero16b);

  // ---- per-lane DMA bookkeeping ---------------------------------------------------------------
  int a_off[NAH];       // element offset of the halo pixel inside the image, or -1
#pragma unroll
  for (int i = 0; i < NAH; ++i) {
    const int hp = (i * 4 + wave) * 8 + rsub;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int y = y0 - p.pad_t + hy, x = x0 - p.pad_l + hx;
    const bool ok = hp < HP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    a_off[i] = ok ? (y * p.W + x) * p.in_ld + lc * 8 : -1;
  }
  int b_off[NBW];       // element offset of (cout row, kw) in the weight tensor (+ lane's chunk), or -1
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int br = (((j * 4 + wave) % NBG) * 8) + rsub;   // row of the [3][BN] slab
    const int kw = br / BN, n = n0 + (br - kw * BN);
    b_off[j] = n < p.Cout ? n * p.K + kw * p.Cin + lc * 8 : -1;
  }
  const int nchunks = (p.Cin + 63) >> 6;
  const int nsteps = nchunks * 3;
  // bilinear taps of this lane's halo pixels (UP): element offset of tap (0, 0) in the low-resolution image, flags for +1 column / +1 row,
  // the two fractions; u_off < 0: pixel outside the image (zeros, like the padding of the stored tensor)
  int u_off[UP ? NAH : 1], u_dx[UP ? NAH : 1], u_dy[UP ? NAH : 1];
  float u_lx[UP ? NAH : 1], u_ly[UP ? NAH : 1];
  const H* __restrict__ low = nullptr;
  int nup = 0;                                  // chunks computed from the low-resolution map
  if constexpr (UP) {
    low = reinterpret_cast<const H*>(p.up_src) + (long long)b * p.up_H * p.up_W * p.up_ld;
    nup = p.up_C >> 6;
#pragma unroll
    for (int i = 0; i < NAH; ++i) {
      const int hp = (i * 4 + wave) * 8 + rsub;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int y = y0 - p.pad_t + hy, x = x0 - p.pad_l + hx;
      const bool ok = hp < HP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const float fy = p.up_sy * (float)y, fx = p.up_sx * (float)x;      // torch: src = scale * dst_index (align_corners=True)
      const int ys = (int)fy, xs = (int)fx;
      u_dy[i] = (ys < p.up_H - 1 ? 1 : 0) * p.up_W * p.up_ld;
      u_dx[i] = (xs < p.up_W - 1 ? 1 : 0) * p.up_ld;
      u_ly[i] = fy - (float)ys; u_lx[i] = fx - (float)xs;
      u_off[i] = ok ? (ys * p.up_W + xs) * p.up_ld + lc * 8 : -1;
    }
  }
  // blend items i = i0, i0 + istep, ... of chunk `chunk` into halo buffer `buf` (four loads per item in flight together)
  auto blend_a = [&](int chunk, int buf, int i0, int istep) {
    if constexpr (UP) {
      const int c0 = chunk * 64;
#pragma unroll
      for (int i = 0; i < NAH; ++i) {
        if (i % istep != i0) continue;
        float t[4][8], o[8];
        const bool ok = u_off[i] >= 0;
        const H* s00 = low + (ok ? u_off[i] : lc * 8) + c0;
        Vec<H>::load(s00, t[0]); Vec<H>::load(s00 + u_dx[i], t[1]); Vec<H>::load(s00 + u_dy[i], t[2]); Vec<H>::load(s00 + u_dy[i] + u_dx[i], t[3]);
        const float ly1 = u_ly[i], lx1 = u_lx[i], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ok ? ly0 * (lx0 * t[0][e] + lx1 * t[1][e]) + ly1 * (lx0 * t[2][e] + lx1 * t[3][e]) : 0.f;
        Vec<H>::store(reinterpret_cast<H*>(sA(buf) + (i * 4 + wave) * 1024 + lane * 16), o);
      }
    }
  };

  auto issue_a = [&](int chunk, int buf) {
    const int c0 = chunk * 64;
    const bool cok = c0 + lc * 8 < p.Cin;
#pragma unroll
    for (int i = 0; i < NAH; ++i) glds16((cok && a_off[i] >= 0) ? in + a_off[i] + c0 : zsrc, sA(buf) + (i * 4 + wave) * 1024);
  };
  auto issue_b = [&](int step, int buf) {
    const int chunk = step / 3, kh = step - chunk * 3;
    const int c0 = chunk * 64;
    const bool cok = c0 + lc * 8 < p.Cin;
    const int koff = kh * 3 * p.Cin + c0;
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      glds16((cok && b_off[j] >= 0) ? wt + b_off[j] + koff : zsrc, sB(buf) + ((j * 4 + wave) % NBG) * 1024);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (UP && nup > 0) blend_a(0, 0, 0, 1); else issue_a(0, 0);
  issue_b(0, 0);
  for (int s = 0; s < nsteps; ++s) {
    const int chunk = s / 3, kh = s - chunk * 3;
    if constexpr (UP) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // + this wave's blended halo stores
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // slab s (+ halo of this chunk) landed; everyone is past step s-1
    asm volatile("" ::: "memory");
    if (s + 1 < nsteps) issue_b(s + 1, (s + 1) & 1);
    const bool next_up = UP && chunk + 1 < nup;
    if (kh == 0 && chunk + 1 < nchunks && !next_up) issue_a(chunk + 1, (chunk + 1) & 1);

    const unsigned char* cA = sA(chunk & 1);
    const unsigned char* cB = sB(s & 1) + (wn * (BN / WN)) * 128;
    const int nsub = (p.Cin - chunk * 64) > 32 ? 2 : 1;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (sub < nsub) {   // wave-uniform
          s16x8 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int hp = (wm * TM + i + kh) * HW_ + fr + kw;
            af[i] = *reinterpret_cast<const s16x8*>(cA + hp * 128 + (((sub * 4 + fq) ^ (hp & 7)) * 16));
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int br = kw * BN + j * 16 + fr;     // + wn * (BN / WN) folded into cB (a multiple of 16)
            bfr[j] = *reinterpret_cast<const s16x8*>(cB + br * 128 + (((sub * 4 + fq) ^ (fr & 7)) * 16));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = mfma16<H>(af[i], bfr[j], acc[i][j]);
        }
      }
    }
    if (next_up) blend_a(chunk + 1, (chunk + 1) & 1, kh, 3);      // a third of the next chunk's halo pixels per kernel-row step
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // operand buffers are free: LDS becomes the C tile

  constexpr int CP = BN + 8;
  static_assert(BM * CP * 2 <= 2 * A_BYTES + 2 * B_BYTES, "C tile must fit in the operand LDS");
  H* sC = reinterpret_cast<H*>(smem);
  float sc[TN], sh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / WN) + j * 16 + fr;
    const bool ok = n < p.Cout;
    sc[j] = (ok && p.scale) ? p.scale[n] : 1.f;
    sh[j] = (ok && p.shift) ? p.shift[n] : 0.f;
  }
  with_act(p.act, [&](auto A) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * TM + i) * 16 + fq * 4 + r;
          const int col = wn * (BN / WN) + j * 16 + fr;
          sC[row * CP + col] = from_f32<H>(act_c16<decltype(A)::value>(acc[i][j][r] * sc[j] + sh[j]));
        }
  });
  __syncthreads();
  constexpr int CH = BN / 8;
  H* __restrict__ out = reinterpret_cast<H*>(p.out);
  const H* __restrict__ res = reinterpret_cast<const H*>(p.res);
  for (int q = tid; q < BM * CH; q += 256) {
    const int row = q / CH, ch = q % CH;
    const int y = y0 + row / TW, x = x0 + row % TW;
    const int n = n0 + ch * 8;
    if (y >= p.Ho || x >= p.Wo || n >= p.Cout) continue;
    const long long m = ((long long)b * p.Ho + y) * p.Wo + x;
    u32x4 v = *reinterpret_cast<const u32x4*>(sC + row * CP + ch * 8);
    if (res) {
      float a[8], r8[8];
      Vec<H>::load(reinterpret_cast<const H*>(&v), a);
      Vec<H>::load(res + m * p.res_ld + n, r8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += r8[e];
      Vec<H>::store(reinterpret_cast<H*>(&v), a);
    }
    *reinterpret_cast<u32x4*>(out + m * p.out_ld + n) = v;
  }
}

template <typename H, int TH, int BN, int WM, int WN, bool UP = false>
int launch3(const ConvP& p, hipStream_t s) {
  constexpr int HP = (TH + 2) * 18, NAG = (HP + 7) / 8, NAH = (NAG + 3) / 4, A_BYTES = NAH * 4 * 1024;
  constexpr int NBG = BN * 3 / 8, B_BYTES = ((NBG + 3) / 4) * 4 * 1024;
  constexpr size_t lds = 2 * A_BYTES + 2 * B_BYTES;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = conv3x3_direct_kernel<H, TH, BN, WM, WN, UP>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
  const long long tiles = (long long)p.B * cdiv(p.Ho, TH) * cdiv(p.Wo, 16) * cdiv(p.Cout, BN);
  if (tiles >= (1ll << 31)) return -1;
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), lds, s, p);
  return 0;
}

struct Cfg3 { int th, bn; };
constexpr Cfg3 kCfg3[] = {{8, 128}, {8, 64}, {16, 64}, {16, 32}, {8, 32}, {16, 16}};
constexpr int kNumCfg3 = sizeof(kCfg3) / sizeof(kCfg3[0]);

}  // namespace

int conv3x3_num_variants() { return kNumCfg3; }
void conv3x3_variant_shape(int v, int* th, int* bn) { *th = kCfg3[v].th; *bn = kCfg3[v].bn; }

// Requirements (checked by the caller): bf16, KH = KW = 3, stride 1, Ho = H + pads - 2 etc. as in ConvP,
// Cin % 8 == 0, H * W * in_ld < 2^31, Cout * K < 2^31.
#define L3U(...) (p.f16 ? launch3<f16_t, __VA_ARGS__, true>(p, s) : launch3<bf16_t, __VA_ARGS__, true>(p, s))
// The same with the upsample + concatenation loader (p.up_src set; up_C % 64 == 0, up_C <= Cin).
int conv3x3_up_launch(int v, const ConvP& p, hipStream_t s) {
  switch (v) {
    case 0: return L3U(8, 128, 2, 2);
    case 1: return L3U(8, 64, 2, 2);
    case 4: return L3U(8, 32, 4, 1);
    case 5: return L3U(16, 16, 4, 1);
    default: return -3;
  }
}

#define L3(...) (p.f16 ? launch3<f16_t, __VA_ARGS__>(p, s) : launch3<bf16_t, __VA_ARGS__>(p, s))
int conv3x3_launch(int v, const ConvP& p, hipStream_t s) {
  switch (v) {
    case 0: return L3(8, 128, 2, 2);
    case 1: return L3(8, 64, 2, 2);
    case 2: return L3(16, 64, 4, 1);
    case 3: return L3(16, 32, 4, 1);
    case 4: return L3(8, 32, 4, 1);
    case 5: return L3(16, 16, 4, 1);
    default: return -3;
  }
}
