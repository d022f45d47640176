// Shared implicit-GEMM main loop (see conv_igemm.hip for the design notes).
#pragma once
#include "common.h"

struct ConvP {
  const void* in;
  const void* w;
  void* out;
  const void* res;
  const float* scale;
  const float* shift;
  int in_ld, out_ld, res_ld;
  int B, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, stride, pad_t, pad_l;
  int M, K;
  int act;
  int pointwise;  // KH == KW == 1, stride 1, no padding: A rows are plain pixel rows
  // --- second-generation (bf16) kernel only; zero / null elsewhere ---
  const float* ln_gamma;   // fused LayerNorm over the Cout axis after scale/shift/act (needs Cout == tile width)
  const float* ln_beta;
  float ln_eps;
  int rows_per_batch;      // > 0: M-tiles do not straddle images and image b uses weights w + b * w_bstride
  long long w_bstride;     // elements
  int f16;                 // 16-bit storage is IEEE half (CFP_F16) instead of bf16
  int k2;                  // gen-2, pointwise only: > 0 = two-term weights.  Every weight row is [hi | lo], each half padded to k2 K-steps
                           // (k2 * 64 elements); the kernel walks 2 * k2 K-steps and reads the SAME activations for both halves
  // --- conv3x3_direct only: bilinear upsample + skip concatenation folded into the halo loader (cfp_upsample_cat_conv3x3, UpSampleBN) ---
  const void* up_src;      // low-resolution source [B, up_H, up_W, up_ld]: input channels [0, up_C) are its align_corners=True bilinear
  int up_ld, up_C, up_H, up_W;   // upsampling to H x W, computed in the loader; channels [up_C, Cin) come from `in` (the skip tensor, virtual base: in + c addresses skip channel c - up_C)
  float up_sy, up_sx;      // (up_H - 1) / (H - 1), (up_W - 1) / (W - 1)
  int probe;               // timing probes of the gen-2 loop (cfp_debug_set key 16; results are garbage): 1 = no operand DMA after the prologue,
                           // 2 = no fragment reads / MFMAs, 3 = neither (barriers and waits only)
  float* mom;              // optional (training): per ROW TILE channel moments of the stored output, [tile_m][2][Cout] = (mean, M2 about that mean)
                           // of the tile's valid rows -- the batch-statistics BatchNorm that follows merges them instead of reading the tensor again
  int dil;                 // input dilation (gen-1 kernels only): the input is read as if `dil - 1` zeros sat between its pixels --
                           // the data gradient of a stride-`dil` convolution; H / W stay the REAL input size
};

// conv_igemm2.hip
int igemm2_num_variants();
void igemm2_variant_shape(int v, int* bm, int* bn, int* stages);
int igemm2_launch(int v, const ConvP& p, float* slabs, int splits, hipStream_t s);

// conv_igemm_x3.hip (float32 storage, f16x3 matrix math; `p.w` = the packed operand of cfp_pack_w_x3)
int igemm_x3_num_variants();
void igemm_x3_variant_shape(int v, int* bm, int* bn, int* stages);
constexpr int CFP_TICKET_SLOTS = 1024;       // output tiles a ticketed split-K launch may have (CFP_CONV_TICKET_BYTES / 4, cfpnet_hip.h)
int igemm_x3_launch(int v, const ConvP& p, float* slabs, int splits, hipStream_t s, unsigned* tickets = nullptr);
int bin_head_x3_launch(const void* x, int x_ld, const void* w, const float* bias, const float* centers, float* prob, float* pred, int B, int HW,
                       int Cin, hipStream_t s);

// conv3x3_halo_x3.hip (float32 storage, f16x3 matrix math, whole-depth halo in LDS; the weights are cfp_pack_w_x3's operand)
int conv3x3_halo_x3_num_variants();
bool conv3x3_halo_x3_takes(const ConvP& p);
size_t conv3x3_halo_x3_lds(int v, const ConvP& p);
int conv3x3_halo_x3_launch(int v, const ConvP& p, hipStream_t s);

// conv3x3_direct.hip
int conv3x3_num_variants();
void conv3x3_variant_shape(int v, int* th, int* bn);
int conv3x3_launch(int v, const ConvP& p, hipStream_t s);

// conv3x3_halo.hip
int conv3x3_halo_num_variants();
void conv3x3_halo_debug_stages(int v);
void conv3x3_halo_debug_odd_pitch(int v);
bool conv3x3_halo_takes(const ConvP& p);
int conv3x3_halo_launch(int v, const ConvP& p, hipStream_t s);

// Channel moments of a 16-bit output tile that sits in LDS as [rows][CP] (the values as STORED, i.e. rounded: what a statistics pass
// over the tensor would read).  TPC = NTHR / BN adjacent lanes share a channel: each takes every TPC-th row with sums shifted by the
// tile's first row (no cancellation), then the (n, mean, M2) triples are merged exactly in a fixed butterfly order.
template <typename T, int BN, int NTHR>
__device__ __forceinline__ void tile_moments(const T* sC, int CP, int rows_valid, int n0, int Cout, float* __restrict__ mom_tile, int tid) {
  constexpr int TPC = NTHR / BN;
  static_assert(NTHR % BN == 0 && TPC >= 1 && TPC <= 64 && (TPC & (TPC - 1)) == 0, "threads per channel");
  const int c = tid / TPC, part = tid % TPC;
  const float shift = to_f32<T>(sC[c]);
  float n = 0.f, s1 = 0.f, s2 = 0.f;
  for (int r = part; r < rows_valid; r += TPC) {
    const float v = to_f32<T>(sC[r * CP + c]) - shift;
    s1 += v; s2 = fmaf(v, v, s2); n += 1.f;
  }
  float mean = n > 0.f ? shift + s1 / n : 0.f;
  float m2 = n > 0.f ? fmaxf(s2 - s1 * s1 / n, 0.f) : 0.f;
#pragma unroll
  for (int o = 1; o < TPC; o <<= 1) {
    const float bn = __shfl_xor(n, o, 64), bmean = __shfl_xor(mean, o, 64), bm2 = __shfl_xor(m2, o, 64);
    // merge(lower lane, upper lane), computed identically by both partners
    const bool up = (part & o) != 0;
    const float an = up ? bn : n, amean = up ? bmean : mean, am2 = up ? bm2 : m2;
    const float cn = up ? n : bn, cmean = up ? mean : bmean, cm2 = up ? m2 : bm2;
    const float tn = an + cn;
    if (tn > 0.f) {
      const float d = cmean - amean, f = cn / tn;
      mean = amean + d * f; m2 = am2 + cm2 + d * d * an * f; n = tn;
    } else { mean = 0.f; m2 = 0.f; n = 0.f; }
  }
  if (part == 0 && n0 + c < Cout) { mom_tile[n0 + c] = mean; mom_tile[Cout + n0 + c] = m2; }
}

__device__ __forceinline__ int swz(int row, int chunk) {
  // physical 16-byte chunk of (row, logical chunk); g = [0,3,2,1][(row >> 2) & 3]
  int q = (row >> 2) & 3;
  int g = (4 - q) & 3;
  return chunk ^ g;
}

// Accumulates C[m0.., n0..] over K-steps [ks_begin, ks_end) into acc (16x16 MFMA tiles).
// smem: 2 * (BM + BN) * 64 bytes.  All 256 threads of the workgroup must call it.
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_mainloop(const ConvP& p, int m0, int n0, int ks_begin, int ks_end,
                                               unsigned char* smem, f32x4 (&acc)[BM / WM / 16][BN / WN / 16]) {
  constexpr int VE = Vec<T>::N;           // elements per 16-byte chunk
  constexpr int BK = 4 * VE;              // 64-byte rows
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int A_ROWS = BM / 64;         // rows of A per thread per K-step
  constexpr int B_ROWS = (BN + 63) / 64;  // rows of W per thread per K-step
  constexpr bool kBf16 = sizeof(T) == 2;
  auto sA = [&](int st) -> unsigned char* { return smem + st * ((BM + BN) * 64); };
  auto sB = [&](int st) -> unsigned char* { return smem + st * ((BM + BN) * 64) + BM * 64; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ wt = reinterpret_cast<const T*>(p.w);

  // ---- per-thread row bookkeeping for the A tile --------------------------------------
  const int ld_row = tid >> 2;  // 0..63
  const int ld_chunk = tid & 3;
  long long a_base[A_ROWS];     // element offset of (b, hi0, wi0) -- may be "virtual" (negative hi0)
  int a_hi0[A_ROWS], a_wi0[A_ROWS];
  bool a_ok[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    int m = m0 + ld_row + 64 * i;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    if (p.pointwise) {
      a_base[i] = (long long)mm * p.in_ld;
      a_hi0[i] = 0;
      a_wi0[i] = 0;
    } else {
      int wo = mm % p.Wo;
      int t = mm / p.Wo;
      int ho = t % p.Ho;
      int b = t / p.Ho;
      a_hi0[i] = ho * p.stride - p.pad_t;
      a_wi0[i] = wo * p.stride - p.pad_l;
      a_base[i] = (long long)b * p.H * p.W * p.in_ld;
    }
  }
  bool b_ok[B_ROWS];
  long long b_base[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    int r = ld_row + 64 * i;
    int n = n0 + r;
    b_ok[i] = (r < BN) && (n < p.Cout);
    b_base[i] = (long long)(b_ok[i] ? n : 0) * p.K;
  }

  u32x4 ra[A_ROWS], rb[B_ROWS];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // im2col position of this thread's chunk, advanced from K-step to K-step (they are issued in order) instead of two integer
  // divisions per step; a jump (first step, split-K start) recomputes it
  int t_k = -1, t_cc = 0, t_kh = 0, t_kw = 0;
  const bool dil_pow2 = (p.dil & (p.dil - 1)) == 0;
  const int dil_shift = 31 - __clz(p.dil);
  auto load_tiles = [&](int k0) {
    int k = k0 + ld_chunk * VE;
    bool kok = k < p.K;
    int kh = 0, kw = 0, cc = k;
    if (!p.pointwise) {
      if (k == t_k + BK && t_k >= 0) {
        t_cc += BK;
        while (t_cc >= p.Cin) {
          t_cc -= p.Cin;
          if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
        }
      } else {
        int tap = k / p.Cin;
        t_cc = k - tap * p.Cin;
        t_kh = tap / p.KW;
        t_kw = tap - t_kh * p.KW;
      }
      t_k = k;
      cc = t_cc; kh = t_kh; kw = t_kw;
    }
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
      bool ok = a_ok[i] && kok;
      long long off = a_base[i] + cc;
      if (!p.pointwise) {
        if (p.dil > 1 && dil_pow2) {   // uniform: (hi, wi) are coordinates of the zero-stuffed input; stride 2 in every shipped model
          ok = ok && hi >= 0 && wi >= 0 && ((hi | wi) & (p.dil - 1)) == 0;
          hi >>= dil_shift; wi >>= dil_shift;
        } else if (p.dil > 1) {
          ok = ok && hi >= 0 && wi >= 0 && hi % p.dil == 0 && wi % p.dil == 0;
          hi /= p.dil; wi /= p.dil;
        }
        ok = ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        off += ((long long)hi * p.W + wi) * p.in_ld;
      }
      ra[i] = ok ? *reinterpret_cast<const u32x4*>(in + off) : zero4;
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      rb[i] = (b_ok[i] && kok) ? *reinterpret_cast<const u32x4*>(wt + b_base[i] + k) : zero4;
    }
  };
  auto store_tiles = [&](int st) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      int r = ld_row + 64 * i;
      *reinterpret_cast<u32x4*>(sA(st) + r * 64 + swz(r, ld_chunk) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      int r = ld_row + 64 * i;
      if (r < BN) *reinterpret_cast<u32x4*>(sB(st) + r * 64 + swz(r, ld_chunk) * 16) = rb[i];
    }
  };

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tiles(ks_begin * BK);
  store_tiles(0);
  __syncthreads();

  const int fr = lane & 15;   // fragment row within a 16-row block
  const int fq = lane >> 4;   // k-chunk (bf16) / k index (f32)

  for (int ks = ks_begin; ks < ks_end; ++ks) {
    const int st = (ks - ks_begin) & 1;
    if (ks + 1 < ks_end) load_tiles((ks + 1) * BK);

    const unsigned char* cA = sA(st) + (wm * (BM / WM)) * 64;
    const unsigned char* cB = sB(st) + (wn * (BN / WN)) * 64;
    if constexpr (kBf16) {
      s16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int r = i * 16 + fr;   // wave row offsets are multiples of 16, so (r>>2)&3 is unchanged by them
        af[i] = *reinterpret_cast<const s16x8*>(cA + r * 64 + swz(r, fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int r = j * 16 + fr;
        bfr[j] = *reinterpret_cast<const s16x8*>(cB + r * 64 + swz(r, fq) * 16);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma16<T>(af[i], bfr[j], acc[i][j]);
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {   // four 16x16x4 steps cover BK = 16
        float af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          int r = i * 16 + fr;
          af[i] = *reinterpret_cast<const float*>(cA + r * 64 + swz(r, kk) * 16 + fq * 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          int r = j * 16 + fr;
          bfr[j] = *reinterpret_cast<const float*>(cB + r * 64 + swz(r, kk) * 16 + fq * 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (ks + 1 < ks_end) store_tiles(st ^ 1);
    __syncthreads();
  }
}
