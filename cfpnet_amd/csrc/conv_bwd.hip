// Backward of the dense convolution / linear layer: weight gradient, and the weight re-layout the data gradient needs.
//
//   reference: autograd of nn.Conv2d / nn.Linear / nn.Conv1d(k=1) in the training step (train.py:119-131).
//
// Weight gradient    dW[co][kh][kw][ci] = sum over output pixels m = (b, ho, wo) of
//                    dY[m][co] * X[b, ho*s - pad_t + kh, wo*s - pad_l + kw, ci]
// is a GEMM whose REDUCTION axis is the pixel axis (up to 6e5 rows at 1/2 scale) and whose output is small
// (Cout x KH*KW*Cin), so it is split over pixel chunks: grid = output tiles x chunks, every workgroup reduces its chunk
// into a 64 x 64 tile on the matrix cores and writes it to its slab; a second kernel adds the slabs in a fixed order
// (bit-reproducible, no atomics).  Both operands have the reduction index as their ROW index in memory (NHWC), i.e. the
// MFMA wants them transposed; with v_mfma_f32_16x16x4_f32 an operand is one scalar per lane, so the transposition is
// just which LDS word a lane reads.  16-bit activations / gradients are widened to f32 while staging: products of
// bf16/fp16 values are exact in f32 and the accumulation is f32 either way (the 16-bit matrix-core form with
// ds_read_b64_tr_b16 operand reads is the speed step, not a different result).
//
// Data gradient      dX = conv(dilate(dY, stride), flip(W)^T): `cfp_conv2d_weight_flip` produces
//                    Wt[ci][KH-1-kh][KW-1-kw][co] = W[co][kh][kw][ci], after which dX is a plain cfp_conv2d_nhwc call
//                    (stride 1, padding K-1-pad, input dilation = the forward stride).
#include "common.h"
#include <cstring>
#include <type_traits>

namespace {

constexpr int WB = 64;          // output tile: 64 (Cout) x 64 (K')
constexpr int WM = 32;          // pixel rows per staging step
constexpr int WP = WB + 16;     // LDS row pitch in floats: 80 = 16 mod 32 -> the two k rows of a 32-lane half hit disjoint banks

struct WgP {
  const void* x; const void* dy; float* slabs;
  int bias;                                    // != 0: slab = [dW (Cout x K) | db (Cout)], db = column sums of dY (16-bit kernels)
  int x_ld, dy_ld;
  int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int M, K, rows_per_split, nsplit, ntiles, xcd_order;
  unsigned mg_hw, sh_hw, mg_w, sh_w;           // floor(m / (Ho*Wo)) and floor(r / Wo) as multiply-high + shift (m < 2^31)
};

// q = floor(m / d) for 0 <= m < 2^31 without an integer division (~40 VALU instructions each, and the im2col row split needs
// two per staged vector): with s = ceil(log2 d) and magic = floor(2^(31+s) / d) + 1 < 2^32,  q = mulhi(m, magic) >> (s - 1)
// (the error term e = magic * d - 2^(31+s) is in (0, d], so m * e < 2^(31+s) and the quotient is exact).  d = 1: magic 0.
inline void fastdiv_make(unsigned d, unsigned* magic, unsigned* shift) {
  if (d <= 1) { *magic = 0u; *shift = 0u; return; }
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  *magic = (unsigned)(((1ull << (31 + s)) / d) + 1ull);
  *shift = s - 1;
}
__device__ __forceinline__ int fastdiv(int m, unsigned magic, unsigned shift) {
  return magic ? (int)(__umulhi((unsigned)m, magic) >> shift) : m;
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgP p) {
  constexpr int VE = Vec<T>::N;
  constexpr int VPR = WB / VE;                 // vectors per tile row
  constexpr int NV = WM * VPR / 256;           // vectors per thread and operand per step (2 for f32, 1 for 16-bit)
  __shared__ float sD[WM * WP];                // dY tile  [m][co]
  __shared__ float sX[WM * WP];                // im2col tile [m][k']
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_k = (p.K + WB - 1) / WB;
  const int tile_co = blockIdx.x / tiles_k, tile_k = blockIdx.x - tile_co * tiles_k;
  const int co0 = tile_co * WB, k0 = tile_k * WB;
  const int m_begin = blockIdx.y * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

  // this thread's staging slots: row r (0..31), vector v of the row
  int s_row[NV], s_col[NV];
  int x_kh[NV], x_kw[NV], x_ci[NV];
  bool k_ok[NV], co_ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = tid + i * 256;
    s_row[i] = q / VPR;
    s_col[i] = (q - s_row[i] * VPR) * VE;
    const int kk = k0 + s_col[i];
    k_ok[i] = kk < p.K;
    const int tap = kk / p.Cin;                 // (kh, kw); Cin % VE == 0 keeps a vector inside one tap
    x_ci[i] = kk - tap * p.Cin;
    x_kh[i] = tap / p.KW;
    x_kw[i] = tap - x_kh[i] * p.KW;
    co_ok[i] = co0 + s_col[i] < p.Cout;
  }
  const int HoWo = p.Ho * p.Wo;

  float rD[NV][VE], rX[NV][VE];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int m = m0 + s_row[i];
      const bool m_ok = m < m_end;
      const int mm = m_ok ? m : m_begin;
      const int b = fastdiv(mm, p.mg_hw, p.sh_hw), r = mm - b * HoWo;
      const int ho = fastdiv(r, p.mg_w, p.sh_w), wo = r - ho * p.Wo;
      const int hi = ho * p.stride - p.pad_t + x_kh[i], wi = wo * p.stride - p.pad_l + x_kw[i];
      const bool in_ok = m_ok && k_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const int hic = min(max(hi, 0), p.H - 1), wic = min(max(wi, 0), p.W - 1);
      // unconditional loads from clamped addresses, the value is selected afterwards (no branch per load)
      Vec<T>::load(X + ((long long)(b * p.H + hic) * p.W + wic) * p.x_ld + (k_ok[i] ? x_ci[i] : 0), rX[i]);
      Vec<T>::load(DY + (long long)mm * p.dy_ld + (co_ok[i] ? co0 + s_col[i] : 0), rD[i]);
      const bool d_ok = m_ok && co_ok[i];
#pragma unroll
      for (int e = 0; e < VE; ++e) { rX[i][e] = in_ok ? rX[i][e] : 0.f; rD[i][e] = d_ok ? rD[i][e] : 0.f; }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int e = 0; e < VE; e += 4) {
        *reinterpret_cast<f32x4*>(&sX[s_row[i] * WP + s_col[i] + e]) = f32x4{rX[i][e], rX[i][e + 1], rX[i][e + 2], rX[i][e + 3]};
        *reinterpret_cast<f32x4*>(&sD[s_row[i] * WP + s_col[i] + e]) = f32x4{rD[i][e], rD[i][e + 1], rD[i][e + 2], rD[i][e + 3]};
      }
    }
  };

  // wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of the tile: 2 x 2 accumulators of 16 x 16
  const int qa = (wave >> 1) * 32, qb = (wave & 1) * 32;
  const int fr = lane & 15, fk = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (m_begin < m_end) fetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += WM) {
    __syncthreads();                           // the previous step's fragment reads are done
    stash();
    __syncthreads();
    if (m0 + WM < m_end) fetch(m0 + WM);       // next step's loads fly while this step is on the matrix cores
#pragma unroll
    for (int ks = 0; ks < WM / 4; ++ks) {
      const int row = ks * 4 + fk;
      float a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sD[row * WP + qa + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[j] = sX[row * WP + qb + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
  }

  // accumulator layout: column (k') = lane & 15, row (co) = 4 * (lane >> 4) + r
  float* __restrict__ slab = p.slabs + (long long)blockIdx.y * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + qa + i * 16 + fk * 4 + r, kk = k0 + qb + j * 16 + fr;
        if (co < p.Cout && kk < p.K) slab[(long long)co * p.K + kk] = acc[i][j][r];
      }
}

// 16-bit inputs on the 16-bit matrix cores (v_mfma_f32_16x16x32_bf16/_f16): the tiles are staged as they lie in memory
// ([pixel row][channel]) and the operands -- which want the pixel axis along k -- are read with the hardware transpose
// read ds_read_b64_tr_b16 (a 16-lane group fetches 4 rows x 16 columns and every lane receives one column of it).
constexpr int W16M = 128;           // pixel rows per staging step of the 64 x 64 tile (four k = 32 MFMA steps per barrier pair)
typedef __attribute__((ext_vector_type(4))) short s16x4;

// PW: pointwise stride-1 convolution / Linear (the majority of the layers): im2col row m IS input row m, no (b, ho, wo) split.
// TCO x TK: output tile (64 or 128 each).  With 64 x 64 tiles the big launches (the 3x3 convolutions at 1/2 and 1/4 scale: 2 - 9 * 10^5
// pixel rows) are bound by L2 -> LDS traffic -- dY is re-read once per K tile, the im2col operand once per Cout tile: 8.4 GB for the
// head's 128 -> 128 conv, 717 us = 15 % of the MFMA peak -- so those run 128-wide tiles (4x less operand traffic per FLOP, 16
// accumulator tiles per wave); the ~200 small layers keep the 64 x 64 tile for its parallelism.
template <typename H, bool PW, int TCO, int TK>
__global__ __launch_bounds__(256) void conv_wgrad16_kernel(WgP p) {
  constexpr int MS = (TCO == 64 && TK == 64) ? W16M : 64;     // pixel rows per staging step
  constexpr int PD = TCO + 8, PX = TK + 8;                    // LDS row pitches in elements (16-byte aligned rows, 2-way bank conflicts at worst)
  __shared__ __attribute__((aligned(16))) unsigned short sD[MS * PD];    // dY tile  [m][co]
  __shared__ __attribute__((aligned(16))) unsigned short sX[MS * PX];    // im2col tile [m][k']
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_k = (p.K + TK - 1) / TK;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), and all output tiles of one pixel chunk read
  // the same dY rows and (shifted) x rows.  With (tile, chunk) = (blockIdx.x, blockIdx.y) the tiles of a chunk landed on 8 different
  // XCDs and each fetched the chunk itself: 9x the compulsory HBM traffic for a 3x3 conv (712 us for the head's weight gradient).
  // Here XCD x owns the chunks x, x + 8, ... and walks all tiles of a chunk back to back, so they meet in ONE L2.
  int tile, split;
  if (p.xcd_order) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tile = slot % p.ntiles; split = (slot / p.ntiles) * 8 + xcd;
  } else {                                                    // many tiles per chunk: plain (tile fastest) order
    tile = blockIdx.x % p.ntiles; split = blockIdx.x / p.ntiles;
  }
  if (split >= p.nsplit) return;                              // padding of the chunk count to a multiple of 8 (uniform: before any barrier)
  const int tile_co = tile / tiles_k, tile_k = tile - tile_co * tiles_k;
  const int co0 = tile_co * TCO, k0 = tile_k * TK;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const unsigned short* __restrict__ X = reinterpret_cast<const unsigned short*>(p.x);
  const unsigned short* __restrict__ DY = reinterpret_cast<const unsigned short*>(p.dy);
  constexpr int NVD = MS * (TCO / 8) / 256, NVX = MS * (TK / 8) / 256;      // 16-byte vectors per thread and step of each operand
  int d_row[NVD], d_col[NVD], x_row[NVX], x_col[NVX], x_kh[NVX], x_kw[NVX], x_ci[NVX];
  bool k_ok[NVX], co_ok[NVD];
#pragma unroll
  for (int i = 0; i < NVD; ++i) {
    const int q = tid + i * 256;
    d_row[i] = q / (TCO / 8);
    d_col[i] = (q - d_row[i] * (TCO / 8)) * 8;
    co_ok[i] = co0 + d_col[i] < p.Cout;
  }
#pragma unroll
  for (int i = 0; i < NVX; ++i) {
    const int q = tid + i * 256;
    x_row[i] = q / (TK / 8);
    x_col[i] = (q - x_row[i] * (TK / 8)) * 8;
    const int kk = k0 + x_col[i];
    k_ok[i] = kk < p.K;
    const int tap = kk / p.Cin;
    x_ci[i] = kk - tap * p.Cin;
    x_kh[i] = tap / p.KW;
    x_kw[i] = tap - x_kh[i] * p.KW;
  }
  const int HoWo = p.Ho * p.Wo;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 rD[NVD], rX[NVX];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
      const int m = m0 + x_row[i];
      const bool m_ok = m < m_end;
      const int mm = m_ok ? m : m_begin;
      bool in_ok;
      long long xrow;
      if (PW) {
        in_ok = m_ok && k_ok[i];
        xrow = mm;
      } else {
        const int b = fastdiv(mm, p.mg_hw, p.sh_hw), r = mm - b * HoWo;
        const int ho = fastdiv(r, p.mg_w, p.sh_w), wo = r - ho * p.Wo;
        const int hi = ho * p.stride - p.pad_t + x_kh[i], wi = wo * p.stride - p.pad_l + x_kw[i];
        in_ok = m_ok && k_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const int hic = min(max(hi, 0), p.H - 1), wic = min(max(wi, 0), p.W - 1);
        xrow = (long long)(b * p.H + hic) * p.W + wic;
      }
      const u32x4 vx = *reinterpret_cast<const u32x4*>(X + xrow * p.x_ld + (k_ok[i] ? x_ci[i] : 0));
      rX[i] = in_ok ? vx : zero4;
    }
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
      const int m = m0 + d_row[i];
      const bool m_ok = m < m_end;
      const int mm = m_ok ? m : m_begin;
      const u32x4 vd = *reinterpret_cast<const u32x4*>(DY + (long long)mm * p.dy_ld + (co_ok[i] ? co0 + d_col[i] : 0));
      rD[i] = (m_ok && co_ok[i]) ? vd : zero4;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < NVX; ++i) *reinterpret_cast<u32x4*>(&sX[x_row[i] * PX + x_col[i]]) = rX[i];
#pragma unroll
    for (int i = 0; i < NVD; ++i) *reinterpret_cast<u32x4*>(&sD[d_row[i] * PD + d_col[i]]) = rD[i];
  };
  constexpr int TI = TCO / 32, TJ = TK / 32;          // 16 x 16 accumulator tiles per wave along Cout / K'
  const int qa = (wave >> 1) * (TCO / 2), qb = (wave & 1) * (TK / 2);
  const int g = lane >> 4, idx = lane & 15, tq = idx >> 2, tp = idx & 3;
  // this lane's address inside a 4-row x 16-column block (rows 8g .. of a k = 32 step): row tq, columns 4 tp .. 4 tp + 3
  const int blkD = (8 * g + tq) * PD + 4 * tp, blkX = (8 * g + tq) * PX + 4 * tp;
  auto tr = [&](const unsigned short* base) -> s16x4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  };
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient db[co] = sum_m dY[m][co] for free: one more B fragment of ones against the dY fragments that are already in registers
  // (all 16 columns of the result are that sum); only the first K tile's left waves do it.  Replaces a column-reduction + finaliser pair.
  const bool do_bias = p.bias && tile_k == 0 && (wave & 1) == 0;
  const short one16 = std::is_same<H, bf16_t>::value ? (short)0x3F80 : (short)0x3C00;      // 1.0 in bf16 / fp16
  const s16x8 ones = {one16, one16, one16, one16, one16, one16, one16, one16};
  f32x4 accb[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (m_begin < m_end) fetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += MS) {
    __syncthreads();
    stash();
    __syncthreads();
    if (m0 + MS < m_end) fetch(m0 + MS);
#pragma unroll
    for (int ks = 0; ks < MS / 32; ++ks) {
      s16x8 a[TI], bb[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const unsigned short* base = sD + ks * 32 * PD + blkD + qa + i * 16;
        const s16x4 lo = tr(base), hi = tr(base + 4 * PD);
        a[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const unsigned short* base = sX + ks * 32 * PX + blkX + qb + j * 16;
        const s16x4 lo = tr(base), hi = tr(base + 4 * PX);
        bb[j] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = mfma16<H>(a[i], bb[j], acc[i][j]);
      if (do_bias) {                            // wave-uniform
#pragma unroll
        for (int i = 0; i < TI; ++i) accb[i] = mfma16<H>(a[i], ones, accb[i]);
      }
    }
  }
  const int fr = lane & 15, fk = lane >> 4;
  const long long slab_elems = (long long)p.Cout * p.K + (p.bias ? p.Cout : 0);
  float* __restrict__ slab = p.slabs + (long long)split * slab_elems;
  if (do_bias && fr == 0) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + qa + i * 16 + fk * 4 + r;
        if (co < p.Cout) slab[(long long)p.Cout * p.K + co] = accb[i][r];
      }
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + qa + i * 16 + fk * 4 + r, kk = k0 + qb + j * 16 + fr;
        if (co < p.Cout && kk < p.K) slab[(long long)co * p.K + kk] = acc[i][j][r];
      }
}

// dw = beta * dw + sum over the splits, in a fixed order.  EW consecutive elements per workgroup, 256 / EW lanes over the splits:
// a small weight tensor split 768 ways (32x32 Linear over 2*10^5 rows) is a long dependent chain per element, so the splits are
// walked by up to 8 lanes with 4 loads in flight each, and the lanes meet in LDS in lane order.
// (`n_dw` < n: the slab is [dW | db]; elements from n_dw on go to db with their own beta.)
template <int EW>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, long long n, float* __restrict__ dw,
                                                           float beta, long long n_dw, float* __restrict__ db, float beta_b) {
  constexpr int LANES = 256 / EW;
  __shared__ float red[LANES][EW];
  const int e = threadIdx.x % EW, sl = threadIdx.x / EW;
  for (long long i0 = (long long)blockIdx.x * EW; i0 < n; i0 += (long long)gridDim.x * EW) {
    const long long i = i0 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four loads in flight; the combination order is still fixed
    if (i < n) {
      int j = sl;
      for (; j + 3 * LANES < nsplit; j += 4 * LANES) {
        s0 += slabs[(long long)j * n + i]; s1 += slabs[(long long)(j + LANES) * n + i];
        s2 += slabs[(long long)(j + 2 * LANES) * n + i]; s3 += slabs[(long long)(j + 3 * LANES) * n + i];
      }
      for (; j < nsplit; j += LANES) s0 += slabs[(long long)j * n + i];
    }
    float s = (s0 + s1) + (s2 + s3);
    if (LANES > 1) {
      red[sl][e] = s;
      __syncthreads();
      if (sl == 0) {
        s = 0.f;
#pragma unroll
        for (int l = 0; l < LANES; ++l) s += red[l][e];
      }
    }
    if (sl == 0 && i < n) {
      if (i < n_dw) dw[i] = beta != 0.f ? beta * dw[i] + s : s;
      else db[i - n_dw] = beta_b != 0.f ? beta_b * db[i - n_dw] + s : s;
    }
    if (LANES > 1) __syncthreads();
  }
}

// The same reduction for MANY layers in one launch (cfp_wgrad_reduce_jobs): the slab sets of up to WJ_MAX weight-gradient launches
// travel by value in the kernel arguments (a graph node then carries them; no table in device memory), a workgroup finds its job by
// a uniform binary search over the block prefix and walks it exactly like wgrad_reduce_kernel<EW> does -- same lanes, same order, so
// the gradients are bit-identical to the per-layer launches.
constexpr int WJ_MAX = 48;
struct WJob { const float* slabs; float* dw; float* db; long long n, n_dw; int nsplit, ewbits; float beta, beta_b; };
struct WJobBatch { WJob j[WJ_MAX]; int start[WJ_MAX + 1]; int njobs; };
static_assert(sizeof(WJobBatch) <= 4000, "the job table travels in the kernel arguments");

__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const WJobBatch b) {
  __shared__ float red[256];
  int lo = 0, hi = b.njobs;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= b.start[mid]) lo = mid; else hi = mid;
  }
  const float* __restrict__ slabs = b.j[lo].slabs;
  float* __restrict__ dw = b.j[lo].dw;
  float* __restrict__ db = b.j[lo].db;
  const long long n = b.j[lo].n, n_dw = b.j[lo].n_dw;
  const int nsplit = b.j[lo].nsplit, ewbits = b.j[lo].ewbits;
  const float beta = b.j[lo].beta, beta_b = b.j[lo].beta_b;
  const int nblk = b.start[lo + 1] - b.start[lo], lb = (int)blockIdx.x - b.start[lo];
  const int EW = 1 << ewbits, LANES = 256 >> ewbits;
  const int e = threadIdx.x & (EW - 1), sl = threadIdx.x >> ewbits;
  if (ewbits == 8 && ((n | n_dw) & 3) == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(dw) |
                                                 reinterpret_cast<uintptr_t>(db)) & 15) == 0) {
    // one lane per element (the large tensors: where the bytes are): four consecutive elements per thread through 16-byte loads -- the
    // same four chains per element in the same order, a quarter of the load instructions
    for (long long i = ((long long)lb * 256 + threadIdx.x) * 4; i < n; i += (long long)nblk * 1024) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      int j = 0;
      for (; j + 3 < nsplit; j += 4) {
        s0 += *reinterpret_cast<const f32x4*>(slabs + (long long)j * n + i);
        s1 += *reinterpret_cast<const f32x4*>(slabs + (long long)(j + 1) * n + i);
        s2 += *reinterpret_cast<const f32x4*>(slabs + (long long)(j + 2) * n + i);
        s3 += *reinterpret_cast<const f32x4*>(slabs + (long long)(j + 3) * n + i);
      }
      for (; j < nsplit; ++j) s0 += *reinterpret_cast<const f32x4*>(slabs + (long long)j * n + i);
      f32x4 sum = (s0 + s1) + (s2 + s3);
      float* dst = i < n_dw ? dw + i : db + (i - n_dw);
      const float bt = i < n_dw ? beta : beta_b;
      if (bt != 0.f) sum += bt * *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = sum;
    }
    return;
  }
  for (long long i0 = (long long)lb * EW; i0 < n; i0 += (long long)nblk * EW) {
    const long long i = i0 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int j = sl;
      for (; j + 3 * LANES < nsplit; j += 4 * LANES) {
        s0 += slabs[(long long)j * n + i]; s1 += slabs[(long long)(j + LANES) * n + i];
        s2 += slabs[(long long)(j + 2 * LANES) * n + i]; s3 += slabs[(long long)(j + 3 * LANES) * n + i];
      }
      for (; j < nsplit; j += LANES) s0 += slabs[(long long)j * n + i];
    }
    float s = (s0 + s1) + (s2 + s3);
    if (LANES > 1) {
      red[sl * EW + e] = s;
      __syncthreads();
      if (sl == 0) {
        s = 0.f;
        for (int l = 0; l < LANES; ++l) s += red[l * EW + e];
      }
    }
    if (sl == 0 && i < n) {
      if (i < n_dw) dw[i] = beta != 0.f ? beta * dw[i] + s : s;
      else db[i - n_dw] = beta_b != 0.f ? beta_b * db[i - n_dw] + s : s;
    }
    if (LANES > 1) __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void weight_flip_kernel(const T* __restrict__ w, T* __restrict__ wt, int Cout, int KH, int KW, int Cin) {
  const long long n = (long long)Cout * KH * KW * Cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    // i indexes the OUTPUT [ci][kh'][kw'][co] so that the writes are coalesced
    const int co = (int)(i % Cout);
    long long t = i / Cout;
    const int kw2 = (int)(t % KW); t /= KW;
    const int kh2 = (int)(t % KH);
    const int ci = (int)(t / KH);
    wt[i] = w[(((long long)co * KH + (KH - 1 - kh2)) * KW + (KW - 1 - kw2)) * Cin + ci];
  }
}

// Every convolution's flipped copy in ONE launch (the training step flips ~290 weight tensors per step, 5 us each when launched
// one by one).  desc[t] = {src offset, dst offset (elements from the base pointers), Cout, KH, KW, Cin, first block, blocks};
// a workgroup finds its tensor by bisection over the first-block column and flips FLIP_PER_BLOCK consecutive outputs of it.
constexpr int FLIP_PER_BLOCK = 2048;
template <typename T>
__global__ __launch_bounds__(256) void weight_flip_batch_kernel(const T* __restrict__ src, T* __restrict__ dst, const long long* __restrict__ desc,
                                                                int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * 8 + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + lo * 8;
  const T* __restrict__ w = src + d[0];
  T* __restrict__ wt = dst + d[1];
  const int Cout = (int)d[2], KH = (int)d[3], KW = (int)d[4], Cin = (int)d[5];
  const long long total = (long long)Cout * KH * KW * Cin;
  const long long i0 = ((long long)blockIdx.x - d[6]) * FLIP_PER_BLOCK;
  for (long long i = i0 + threadIdx.x; i < min(total, i0 + FLIP_PER_BLOCK); i += 256) {
    const int co = (int)(i % Cout);
    long long t = i / Cout;
    const int kw2 = (int)(t % KW); t /= KW;
    const int kh2 = (int)(t % KH);
    const int ci = (int)(t / KH);
    wt[i] = w[(((long long)co * KH + (KH - 1 - kh2)) * KW + (KW - 1 - kw2)) * Cin + ci];
  }
}

int g_wgrad_target = 768;          // cfp_debug_set key 22: workgroups a 64 x 64-tile weight-gradient launch aims for
inline int wgrad_nsplit(int Cout, int K, int M) {
  const long long tiles = (long long)cdiv(Cout, WB) * cdiv(K, WB);
  long long ns = (g_wgrad_target + tiles - 1) / tiles;                  // ~3 workgroups per CU in total; every split costs a slab to add up
  const long long max_ns = cdiv(M, 4 * WM);                  // at least 128 rows per chunk
  if (ns > max_ns) ns = max_ns;
  if (ns > 1024) ns = 1024;
  if (ns < 1) ns = 1;
  return (int)ns;
}

// 16-bit launches with many pixel rows: 128-wide tiles along every axis that has them (see conv_wgrad16_kernel)
struct WgPlan { int tco, tk, mstep, nsplit; };
inline WgPlan wgrad_plan16(int Cout, int K, int M) {
  static const bool big_off = [] { const char* e = getenv("CFP_WGRAD_BIG"); return e && e[0] == '0'; }();      // A/B switch
  WgPlan pl{WB, WB, W16M, 0};
  if (M >= 100000 && !big_off) {
    if (Cout > 64) pl.tco = 128;
    if (K > 64) pl.tk = 128;
  }
  if (pl.tco == WB && pl.tk == WB) { pl.nsplit = wgrad_nsplit(Cout, K, M); return pl; }
  pl.mstep = 64;
  const long long tiles = (long long)cdiv(Cout, pl.tco) * cdiv(K, pl.tk);
  // every workgroup of the launch resident at once, XCD by XCD: these kernels hold 3 workgroups per CU (152 VGPRs), an XCD has 32 CUs and
  // owns the chunks x, x + 8, ...  One workgroup too many per XCD costs a whole extra round (the head's 774 workgroups on 768 slots: 627 us)
  long long ns = 8 * std::max<long long>(1, 96 / tiles);
  const long long max_ns = cdiv(M, 8 * pl.mstep);            // at least 512 rows per chunk
  if (ns > max_ns) ns = max_ns;
  if (ns > 1024) ns = 1024;
  if (ns < 1) ns = 1;
  pl.nsplit = (int)ns;
  return pl;
}

}  // namespace

void cfp_wgrad_debug_set(int value) { g_wgrad_target = value; }

extern "C" size_t cfp_conv2d_wgrad_ws_bytes(int Cout, int K, int M) {
  if (Cout <= 0 || K <= 0 || M <= 0) return 0;
  const int ns = std::max(wgrad_nsplit(Cout, K, M), wgrad_plan16(Cout, K, M).nsplit);      // the dtype is not known here: the larger plan
  return (size_t)ns * ((size_t)Cout * K + Cout) * sizeof(float);            // + the bias-gradient column of cfp_conv2d_wgrad_bias
}

extern "C" int cfp_conv2d_wgrad_bias(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int Cin,
                                     int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, float beta_b,
                                     int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);

extern "C" int cfp_conv2d_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws,
                                size_t ws_bytes, cfp_stream_t stream) {
  return cfp_conv2d_wgrad_bias(x, x_ld, dy, dy_ld, dw, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, beta, 0.f, dtype, ws,
                               ws_bytes, stream);
}

static int wgrad_impl(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int Cin,
                      int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, float beta_b,
                      int dtype, void* ws, size_t ws_bytes, cfp_wgrad_job* job, cfp_stream_t stream) {
  CFP_REQUIRE(x && dy && dw && ws, CFP_EINVAL, "cfp_conv2d_wgrad: null pointer");
  CFP_REQUIRE(!db || is16(dtype), CFP_EINVAL, "cfp_conv2d_wgrad_bias: the fused bias gradient is a 16-bit path (float32: cfp_colsum)");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_conv2d_wgrad: bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && Ho > 0 && Wo > 0, CFP_ESHAPE,
              "cfp_conv2d_wgrad: non-positive dimension");
  CFP_REQUIRE(Cin % ve == 0 && Cout % ve == 0 && x_ld % ve == 0 && dy_ld % ve == 0 && x_ld >= Cin && dy_ld >= Cout, CFP_ESHAPE,
              "cfp_conv2d_wgrad: channel counts / pitches must be multiples of the 16-byte vector");
  CFP_REQUIRE(aligned16(x) && aligned16(dy), CFP_EINVAL, "cfp_conv2d_wgrad: pointers must be 16-byte aligned");
  const long long M = (long long)B * Ho * Wo, K = (long long)KH * KW * Cin;
  CFP_REQUIRE(M < (1ll << 31) && K < (1ll << 31) && (long long)Cout * K < (1ll << 31), CFP_ESHAPE, "cfp_conv2d_wgrad: problem too large");
  CFP_REQUIRE(ws_bytes >= cfp_conv2d_wgrad_ws_bytes(Cout, (int)K, (int)M), CFP_EINVAL, "cfp_conv2d_wgrad: workspace too small");
  CFP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 3) == 0, CFP_EINVAL, "cfp_conv2d_wgrad: workspace must be 4-byte aligned");
  WgP p;
  p.x = x; p.dy = dy; p.slabs = reinterpret_cast<float*>(ws); p.x_ld = x_ld; p.dy_ld = dy_ld;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.Ho = Ho; p.Wo = Wo; p.M = (int)M; p.K = (int)K;
  fastdiv_make((unsigned)(Ho * Wo), &p.mg_hw, &p.sh_hw);
  fastdiv_make((unsigned)Wo, &p.mg_w, &p.sh_w);
  const WgPlan pl = is16(dtype) ? wgrad_plan16(Cout, (int)K, (int)M) : WgPlan{WB, WB, WM, wgrad_nsplit(Cout, (int)K, (int)M)};
  p.nsplit = pl.nsplit;
  const int mstep = pl.mstep;
  p.rows_per_split = cdiv(cdiv(M, p.nsplit), mstep) * mstep;
  p.nsplit = cdiv(M, p.rows_per_split);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  p.bias = db ? 1 : 0;
  if (p.nsplit == 1 && beta == 0.f && !db) p.slabs = dw;     // a single slab IS the result: written in place, no second kernel
  p.ntiles = cdiv(Cout, pl.tco) * cdiv(K, pl.tk);
  p.xcd_order = p.ntiles <= 24 ? 1 : 0;
  const dim3 grid(p.ntiles, p.nsplit);                        // float32 kernel: (tile, chunk)
  const dim3 grid16((unsigned)p.ntiles * (unsigned)(cdiv(p.nsplit, 8) * 8));      // 16-bit kernels: one axis, XCD-aware order inside
  const bool pw = KH == 1 && KW == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == H && Wo == W;
#define WG16(H, PWV, A, Bk) hipLaunchKernelGGL((conv_wgrad16_kernel<H, PWV, A, Bk>), grid16, dim3(256), 0, s, p)
#define WG16_T(H, PWV) do { if (pl.tco == 128 && pl.tk == 128) WG16(H, PWV, 128, 128); else if (pl.tco == 128) WG16(H, PWV, 128, 64); \
                            else if (pl.tk == 128) WG16(H, PWV, 64, 128); else WG16(H, PWV, 64, 64); } while (0)
  if (dtype == CFP_BF16 && pw) WG16_T(bf16_t, true);
  else if (dtype == CFP_BF16) WG16_T(bf16_t, false);
  else if (dtype == CFP_F16 && pw) WG16_T(f16_t, true);
  else if (dtype == CFP_F16) WG16_T(f16_t, false);
  else hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, s, p);
#undef WG16_T
#undef WG16
  const long long n_dw = (long long)Cout * K, n = n_dw + (db ? Cout : 0);
  if (job) std::memset(job, 0, sizeof(*job));                  // nsplit = 0: nothing left to reduce
  if (p.nsplit > 1 || beta != 0.f || db) {
    // enough workgroups first, then as few split lanes as that allows
    const int ew = (n >= 256 * 512 || p.nsplit < 8) ? 256 : (n >= 64 * 512 || p.nsplit < 32) ? 64 : 32;
    const int blocks = (int)std::min<long long>(2048, (n + ew - 1) / ew);
    if (job) {                                                 // the caller reduces later, with other layers (cfp_wgrad_reduce_jobs)
      job->slabs = p.slabs; job->dw = dw; job->db = db; job->n = n; job->n_dw = n_dw; job->nsplit = p.nsplit; job->ew = ew;
      job->beta = beta; job->beta_b = beta_b;
    } else if (ew == 256) hipLaunchKernelGGL(wgrad_reduce_kernel<256>, dim3(blocks), dim3(256), 0, s, p.slabs, p.nsplit, n, dw, beta, n_dw, db, beta_b);
    else if (ew == 64) hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3(blocks), dim3(256), 0, s, p.slabs, p.nsplit, n, dw, beta, n_dw, db, beta_b);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3(blocks), dim3(256), 0, s, p.slabs, p.nsplit, n, dw, beta, n_dw, db, beta_b);
  }
  return cfp_check_launch("cfp_conv2d_wgrad");
}

extern "C" int cfp_conv2d_wgrad_bias(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int Cin,
                                     int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, float beta_b,
                                     int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  return wgrad_impl(x, x_ld, dy, dy_ld, dw, db, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, beta, beta_b, dtype, ws, ws_bytes,
                    nullptr, stream);
}

extern "C" int cfp_conv2d_wgrad_deferred(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int Cin,
                                         int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, float beta_b,
                                         int dtype, void* ws, size_t ws_bytes, cfp_wgrad_job* job, cfp_stream_t stream) {
  CFP_REQUIRE(job, CFP_EINVAL, "cfp_conv2d_wgrad_deferred: null job");
  return wgrad_impl(x, x_ld, dy, dy_ld, dw, db, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, beta, beta_b, dtype, ws, ws_bytes,
                    job, stream);
}

extern "C" int cfp_wgrad_reduce_jobs(const cfp_wgrad_job* jobs, int njobs, cfp_stream_t stream) {
  CFP_REQUIRE(njobs >= 0 && (jobs || njobs == 0), CFP_EINVAL, "cfp_wgrad_reduce_jobs: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  WJobBatch b;
  int nb = 0, blocks = 0;
  auto flush = [&]() {
    if (nb == 0) return;
    b.njobs = nb; b.start[nb] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b);
    nb = 0; blocks = 0;
  };
  for (int i = 0; i < njobs; ++i) {
    const cfp_wgrad_job& j = jobs[i];
    if (j.nsplit <= 0) continue;                               // written in place by its launch
    CFP_REQUIRE(j.slabs && j.dw && j.n > 0 && j.n_dw > 0 && j.n_dw <= j.n && (j.db || j.n_dw == j.n) &&
                (j.ew == 32 || j.ew == 64 || j.ew == 256), CFP_EINVAL, "cfp_wgrad_reduce_jobs: bad job");
    for (int k = 0; k < nb; ++k)                               // two jobs of one launch must not write the same tensor: a new launch
      if (b.j[k].dw == j.dw || (j.db && b.j[k].db == j.db)) { flush(); break; }
    b.j[nb] = WJob{j.slabs, j.dw, j.db, j.n, j.n_dw, j.nsplit, j.ew == 256 ? 8 : j.ew == 64 ? 6 : 5, j.beta, j.beta_b};
    b.start[nb] = blocks;
    blocks += (int)std::min<long long>(2048, (j.n + j.ew - 1) / j.ew);
    if (++nb == WJ_MAX) flush();
  }
  flush();
  return cfp_check_launch("cfp_wgrad_reduce_jobs");
}

extern "C" int cfp_conv2d_weight_flip(const void* w, void* wt, int Cout, int KH, int KW, int Cin, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(w && wt && w != wt, CFP_EINVAL, "cfp_conv2d_weight_flip: bad pointer");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_conv2d_weight_flip: bad dtype");
  CFP_REQUIRE(Cout > 0 && KH > 0 && KW > 0 && Cin > 0 && (long long)Cout * KH * KW * Cin < (1ll << 31), CFP_ESHAPE,
              "cfp_conv2d_weight_flip: bad shape");
  const long long n = (long long)Cout * KH * KW * Cin;
  int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_F32)
    hipLaunchKernelGGL(weight_flip_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)w, (float*)wt, Cout, KH, KW, Cin);
  else
    hipLaunchKernelGGL(weight_flip_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, (const unsigned short*)w, (unsigned short*)wt,
                       Cout, KH, KW, Cin);
  return cfp_check_launch("cfp_conv2d_weight_flip");
}

extern "C" int cfp_weight_flip_blocks(long long elems) { return elems > 0 ? (int)((elems + FLIP_PER_BLOCK - 1) / FLIP_PER_BLOCK) : 0; }

extern "C" int cfp_conv2d_weight_flip_batch(const void* src_base, void* dst_base, const long long* desc, int n, int total_blocks, int dtype,
                                            cfp_stream_t stream) {
  CFP_REQUIRE(src_base && dst_base && desc && src_base != dst_base, CFP_EINVAL, "cfp_conv2d_weight_flip_batch: bad pointer");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_conv2d_weight_flip_batch: bad dtype");
  CFP_REQUIRE(n > 0 && total_blocks > 0, CFP_ESHAPE, "cfp_conv2d_weight_flip_batch: empty batch");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_F32)
    hipLaunchKernelGGL(weight_flip_batch_kernel<float>, dim3(total_blocks), dim3(256), 0, s, (const float*)src_base, (float*)dst_base, desc, n);
  else
    hipLaunchKernelGGL(weight_flip_batch_kernel<unsigned short>, dim3(total_blocks), dim3(256), 0, s, (const unsigned short*)src_base,
                       (unsigned short*)dst_base, desc, n);
  return cfp_check_launch("cfp_conv2d_weight_flip_batch");
}
