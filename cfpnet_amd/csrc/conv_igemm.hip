// Implicit-GEMM convolution / linear layer on the CDNA4 matrix cores.
//
//   C[m, n] = sum_k A[m, k] * Wt[n, k]        m = (b, ho, wo)   k = (kh, kw, ci)   n = cout
//
// A is never materialised: each 16-byte chunk of a row of A (8 bf16 / 4 f32 consecutive input
// channels of one filter tap) is one global load from the NHWC activation, or zeros at the
// image border.  Weights are pre-packed [Cout][KH*KW*Cin], so both operands are K-contiguous.
//
// Workgroup = 256 threads = 4 waves laid out WM x WN over a BM x BN output tile; every wave
// owns (BM/WM) x (BN/WN) outputs as 16x16 MFMA accumulators.
//   bf16 : v_mfma_f32_16x16x32_bf16, BK = 32 (64-byte tile rows)
//   f32  : v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), BK = 16 -- the parity mode
// Tiles are staged global -> registers -> LDS with the next K-step's loads in flight during the
// MFMAs (double-buffered LDS, one barrier per K-step).  LDS rows are 64 bytes with a 16-byte
// chunk XOR swizzle chosen so every ds_read_b128 lane group of the 16x16x32 operand pattern
// hits 64 distinct banks.
// Epilogue: per-channel scale/shift (folded BatchNorm or bias), activation, optional residual;
// bf16 results go through LDS so global stores are full 16-byte vectors along the channel axis.
#include "common.h"

extern "C" int cfp_conv2d_variant(int M, int Cout);

namespace {

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
};

__device__ __forceinline__ int swz(int row, int chunk) {
  // physical 16-byte chunk of (row, logical chunk); g = [0,3,2,1][(row >> 2) & 3]
  int q = (row >> 2) & 3;
  int g = (4 - q) & 3;
  return chunk ^ g;
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int VE = Vec<T>::N;           // elements per 16-byte chunk
  constexpr int BK = 4 * VE;              // 64-byte rows
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int A_ROWS = BM / 64;         // rows of A per thread per K-step
  constexpr int B_ROWS = (BN + 63) / 64;  // rows of W per thread per K-step
  constexpr bool kBf16 = sizeof(T) == 2;

  // [2 stages][(BM + BN) rows][64 bytes]; reused as the C staging tile in the bf16 epilogue
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * 64];
  auto sA = [&](int st) -> unsigned char* { return smem + st * ((BM + BN) * 64); };
  auto sB = [&](int st) -> unsigned char* { return smem + st * ((BM + BN) * 64) + BM * 64; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

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

  auto load_tiles = [&](int k0) {
    int k = k0 + ld_chunk * VE;
    bool kok = k < p.K;
    int kh = 0, kw = 0, cc = k;
    if (!p.pointwise) {
      int tap = k / p.Cin;
      cc = k - tap * p.Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
      bool ok = a_ok[i] && kok;
      long long off = a_base[i] + cc;
      if (!p.pointwise) {
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

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int fr = lane & 15;   // fragment row within a 16-row block
  const int fq = lane >> 4;   // k-chunk (bf16) / k index (f32)

  for (int ks = 0; ks < nk; ++ks) {
    const int st = ks & 1;
    if (ks + 1 < nk) load_tiles((ks + 1) * BK);

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
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
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
    if (ks + 1 < nk) store_tiles(st ^ 1);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------
  // accumulator layout (16x16): col = lane & 15, row = (lane >> 4) * 4 + reg
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
  float sc[TN], sh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int n = n0 + wn * (BN / WN) + j * 16 + fr;
    bool ok = n < p.Cout;
    sc[j] = (ok && p.scale) ? p.scale[n] : 1.f;
    sh[j] = (ok && p.shift) ? p.shift[n] : 0.f;
  }

  if constexpr (kBf16) {
    // C-tile pitch (elements): padded by one 16-byte chunk when the operand LDS has room
    constexpr int CP = (BM * (BN + 8) * 2 <= 2 * (BM + BN) * 64) ? BN + 8 : BN;
    static_assert(BM * CP * 2 <= 2 * (BM + BN) * 64, "C tile must fit in the operand LDS");
    bf16_t* sC = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = wm * (BM / WM) + i * 16 + fq * 4 + r;
          int col = wn * (BN / WN) + j * 16 + fr;
          float v = apply_act(acc[i][j][r] * sc[j] + sh[j], p.act);
          sC[row * CP + col] = f2bf(v);
        }
    __syncthreads();
    constexpr int CH = BN / 8;   // 16-byte chunks per tile row
    for (int q = tid; q < BM * CH; q += 256) {
      int row = q / CH, ch = q % CH;
      int m = m0 + row, n = n0 + ch * 8;
      if (m >= p.M || n >= p.Cout) continue;
      u32x4 v = *reinterpret_cast<const u32x4*>(sC + row * CP + ch * 8);
      if (res) {
        float a[8], b[8];
        Vec<bf16_t>::load(reinterpret_cast<const bf16_t*>(&v), a);
        Vec<bf16_t>::load(reinterpret_cast<const bf16_t*>(res) + (long long)m * p.res_ld + n, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        Vec<bf16_t>::store(reinterpret_cast<bf16_t*>(&v), a);
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(out) + (long long)m * p.out_ld + n) = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = m0 + wm * (BM / WM) + i * 16 + fq * 4 + r;
          int n = n0 + wn * (BN / WN) + j * 16 + fr;
          if (m < p.M && n < p.Cout) {
            float v = apply_act(acc[i][j][r] * sc[j] + sh[j], p.act);
            if (res) v += to_f32<T>(res[(long long)m * p.res_ld + n]);
            out[(long long)m * p.out_ld + n] = from_f32<T>(v);
          }
        }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
void launch(const ConvP& p, hipStream_t s) {
  int tiles = cdiv(p.M, BM) * cdiv(p.Cout, BN);
  hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN>), dim3(tiles), dim3(256), 0, s, p);
}

template <typename T>
void dispatch(const ConvP& p, hipStream_t s) {
  // Tile choice: widest N tile that the layer fills; fall back to a smaller M tile when the
  // 128x128 grid would leave most of the 256 CUs idle.
  switch (cfp_conv2d_variant(p.M, p.Cout)) {
    case 0: return launch<T, 256, 16, 4, 1>(p, s);
    case 1: return launch<T, 256, 32, 4, 1>(p, s);
    case 2: return launch<T, 128, 64, 2, 2>(p, s);
    default: return launch<T, 128, 128, 2, 2>(p, s);
  }
}

}  // namespace

extern "C" int cfp_conv2d_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                               const void* residual, int res_ld, void* out, int out_ld, int B, int H, int W, int Cin,
                               int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                               int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(in && w && out, CFP_EINVAL, "cfp_conv2d_nhwc: null pointer");
  CFP_REQUIRE(dtype == CFP_F32 || dtype == CFP_BF16, CFP_EINVAL, "cfp_conv2d_nhwc: bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && Ho > 0 && Wo > 0,
              CFP_ESHAPE, "cfp_conv2d_nhwc: non-positive dimension");
  CFP_REQUIRE(Cin % ve == 0 && in_ld % ve == 0 && in_ld >= Cin, CFP_ESHAPE,
              "cfp_conv2d_nhwc: Cin / in_ld must be multiples of the 16-byte vector");
  CFP_REQUIRE(Cout % ve == 0 && out_ld % ve == 0 && out_ld >= Cout, CFP_ESHAPE,
              "cfp_conv2d_nhwc: Cout / out_ld must be multiples of the 16-byte vector");
  CFP_REQUIRE(!residual || (res_ld % ve == 0 && res_ld >= Cout), CFP_ESHAPE, "cfp_conv2d_nhwc: bad res_ld");
  CFP_REQUIRE(aligned16(in) && aligned16(w) && aligned16(out) && aligned16(residual), CFP_EINVAL,
              "cfp_conv2d_nhwc: pointers must be 16-byte aligned");
  CFP_REQUIRE((Ho - 1) * stride - pad_t + KH - 1 < H + KH && (Wo - 1) * stride - pad_l + KW - 1 < W + KW, CFP_ESHAPE,
              "cfp_conv2d_nhwc: output size inconsistent with input size");
  CFP_REQUIRE((long long)B * Ho * Wo < (1ll << 31) && (long long)KH * KW * Cin < (1ll << 31), CFP_ESHAPE,
              "cfp_conv2d_nhwc: problem too large");
  ConvP p;
  p.in = in; p.w = w; p.out = out; p.res = residual; p.scale = scale; p.shift = shift;
  p.in_ld = in_ld; p.out_ld = out_ld; p.res_ld = res_ld;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.M = B * Ho * Wo; p.K = KH * KW * Cin; p.act = act;
  p.pointwise = (KH == 1 && KW == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == H && Wo == W) ? 1 : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) dispatch<bf16_t>(p, s); else dispatch<float>(p, s);
  return cfp_check_launch("cfp_conv2d_nhwc");
}

// Which tile configuration cfp_conv2d_nhwc picks for a problem (for per-kernel accounting in
// bench.py): 0 = 256x16, 1 = 256x32, 2 = 128x64, 3 = 128x128.
extern "C" int cfp_conv2d_variant(int M, int Cout) {
  if (Cout <= 16) return 0;
  if (Cout <= 32) return 1;
  if (Cout <= 64) return 2;
  long long t128 = (long long)cdiv(M, 128) * cdiv(Cout, 128);
  bool waste = (Cout % 128) != 0 && (Cout % 128) <= 64;
  return (t128 < 192 || waste) ? 2 : 3;
}
