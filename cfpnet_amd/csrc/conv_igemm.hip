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
// MFMAs (double-buffered LDS, one barrier per K-step; igemm_core.h).  LDS rows are 64 bytes with
// a 16-byte chunk XOR swizzle chosen so every ds_read_b128 lane group of the 16x16x32 operand
// pattern hits 64 distinct banks.
// Epilogue: per-channel scale/shift (folded BatchNorm or bias), activation, optional residual;
// bf16 results go through LDS so global stores are full 16-byte vectors along the channel axis.
//
// Split-K: layers with few output tiles but a long K (the GSA "sr" convs: M = 240..1040,
// K = 4608..5184; the 1/32-scale pointwise convs) would run a handful of workgroups through
// 100+ serial K-steps.  They are cut into `splits` K-ranges writing f32 partial slabs
// [split][M][Cout] to a caller-provided workspace; a second kernel sums the slabs in split order
// (deterministic) and applies the epilogue.
#include "igemm_core.h"

extern "C" int cfp_conv2d_variant(int M, int Cout);

namespace {

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int VE = Vec<T>::N;
  constexpr int BK = 4 * VE;
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr bool kBf16 = sizeof(T) == 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lbid / tiles_n;
  const int tile_n = lbid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x4 acc[TM][TN];
  igemm_mainloop<T, BM, BN, WM, WN>(p, m0, n0, 0, (p.K + BK - 1) / BK, smem, acc);

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
    T* sC = reinterpret_cast<T*>(smem);
    with_act(p.act, [&](auto A) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = wm * (BM / WM) + i * 16 + fq * 4 + r;
            int col = wn * (BN / WN) + j * 16 + fr;
            sC[row * CP + col] = from_f32<T>(act_c16<decltype(A)::value>(acc[i][j][r] * sc[j] + sh[j]));
          }
    });
    __syncthreads();
    if (p.mom != nullptr) tile_moments<T, BN, 256>(sC, CP, min(BM, p.M - m0), n0, p.Cout, p.mom + (long long)tile_m * 2 * p.Cout, tid);
    constexpr int CH = BN / 8;   // 16-byte chunks per tile row
    for (int q = tid; q < BM * CH; q += 256) {
      int row = q / CH, ch = q % CH;
      int m = m0 + row, n = n0 + ch * 8;
      if (m >= p.M || n >= p.Cout) continue;
      u32x4 v = *reinterpret_cast<const u32x4*>(sC + row * CP + ch * 8);
      if (res) {
        float a[8], b[8];
        Vec<T>::load(reinterpret_cast<const T*>(&v), a);
        Vec<T>::load(res + (long long)m * p.res_ld + n, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        Vec<T>::store(reinterpret_cast<T*>(&v), a);
      }
      *reinterpret_cast<u32x4*>(out + (long long)m * p.out_ld + n) = v;
    }
  } else {
    with_act(p.act, [&](auto A) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = act_c<decltype(A)::value>(acc[i][j][r] * sc[j] + sh[j]);
    });
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = m0 + wm * (BM / WM) + i * 16 + fq * 4 + r;
          int n = n0 + wn * (BN / WN) + j * 16 + fr;
          if (m < p.M && n < p.Cout) {
            float v = acc[i][j][r];
            if (res) v += to_f32<T>(res[(long long)m * p.res_ld + n]);
            out[(long long)m * p.out_ld + n] = from_f32<T>(v);
          }
        }
  }
}

// Split-K: grid = tiles x splits; raw f32 accumulators to slab[split][m][n].
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_splitk_kernel(ConvP p, float* __restrict__ slabs, int splits) {
  constexpr int VE = Vec<T>::N;
  constexpr int BK = 4 * VE;
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, fr = lane & 15, fq = lane >> 4;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + BK - 1) / BK;
  const int per = (nk + splits - 1) / splits;
  const int k0 = sp * per, k1 = min(nk, k0 + per);
  f32x4 acc[TM][TN];
  if (k0 < k1) {   // uniform over the workgroup
    igemm_mainloop<T, BM, BN, WM, WN>(p, m0, n0, k0, k1, smem, acc);
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float* slab = slabs + (long long)sp * p.M * p.Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm * (BM / WM) + i * 16 + fq * 4 + r;
        int n = n0 + wn * (BN / WN) + j * 16 + fr;
        if (m < p.M && n < p.Cout) slab[(long long)m * p.Cout + n] = acc[i][j][r];
      }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, ConvP p) {
  constexpr int VE = Vec<T>::N;
  const int CV = p.Cout / VE;
  const long long total = (long long)p.M * CV;
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const long long m = i / CV;
    const int n = cv * VE;
    float v[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = 0.f;
    for (int s = 0; s < splits; ++s) {
      const float* sp = slabs + ((long long)s * p.M + m) * p.Cout + n;
#pragma unroll
      for (int e = 0; e < VE; e += 4) {
        f32x4 x = *reinterpret_cast<const f32x4*>(sp + e);
        v[e] += x[0]; v[e + 1] += x[1]; v[e + 2] += x[2]; v[e + 3] += x[3];
      }
    }
    with_act(p.act, [&](auto A) {
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float sc = p.scale ? p.scale[n + e] : 1.f, sh = p.shift ? p.shift[n + e] : 0.f;
        if constexpr (sizeof(T) == 2) v[e] = act_c16<decltype(A)::value>(v[e] * sc + sh);      // like the gen-2 epilogue: same results whether K was split or not
        else v[e] = act_c<decltype(A)::value>(v[e] * sc + sh);
      }
    });
    if constexpr (sizeof(T) == 4) {
      if (p.ln_gamma != nullptr) {
        // LayerNorm over the Cout channels of the row, fused into the finishing sum (round 5: the global attention's patch conv is a split-K
        // GEMM followed by nn.LayerNorm -- twins.py sr + norm).  A row is CV = Cout / 4 consecutive lanes (host: a power of two <= 64 and
        // Cout a multiple of 4), all in one wave and all in this iteration together; two-pass statistics as cfp_layernorm.
        float sum = (v[0] + v[1]) + (v[2] + v[3]);
        for (int o = 1; o < CV; o <<= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)p.Cout;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) { v[e] -= mean; q = fmaf(v[e], v[e], q); }
        for (int o = 1; o < CV; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q / (float)p.Cout + p.ln_eps);
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = v[e] * rstd * p.ln_gamma[n + e] + p.ln_beta[n + e];
      }
    }
    if (res) {
      float r[VE];
      Vec<T>::load(res + m * p.res_ld + n, r);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += r[e];
    }
    Vec<T>::store(out + m * p.out_ld + n, v);
  }
}


// Few-row f32 GEMM: out[m, n] = act(scale[n] * sum_k x[m, k] w[n, k] + shift[n]) (+ res) for M <= 64 rows -- the squeeze-excite
// FCs of the training step on [B, C] vectors and their data gradients, where a 128x64 tile leaves one or two workgroups walking
// K alone (31 us per call).  WPC waves per output column split K in 16-byte pieces (the weight row is read once, the <= 16
// input rows come from L2; all 17 loads of a step are issued before the first FMA), partial sums meet in a wave reduction and,
// for WPC = 4 (long K, few columns), in LDS in wave order.  256 threads = 4 / WPC columns per workgroup.
template <int WPC>
__global__ __launch_bounds__(256) void rowgemm_f32_kernel(ConvP p) {
  __shared__ float red[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = WPC == 4 ? blockIdx.x : blockIdx.x * 4 + wave;
  const bool n_ok = n < p.Cout;
  const int nn = n_ok ? n : 0;
  const float* __restrict__ x = reinterpret_cast<const float*>(p.in);
  const float* __restrict__ w = reinterpret_cast<const float*>(p.w) + (long long)nn * p.K;
  const float* __restrict__ res = reinterpret_cast<const float*>(p.res);
  float* __restrict__ out = reinterpret_cast<float*>(p.out);
  const float sc = p.scale ? p.scale[nn] : 1.f, sh = p.shift ? p.shift[nn] : 0.f;
  const int kstart = (WPC == 4 ? threadIdx.x : lane) * 4, kstep = WPC == 4 ? 1024 : 256;
  for (int m0 = 0; m0 < p.M; m0 += 16) {
    const int mc = min(16, p.M - m0);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = kstart; k < p.K; k += kstep) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
      f32x4 xv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) xv[i] = *reinterpret_cast<const f32x4*>(x + (long long)(m0 + min(i, mc - 1)) * p.in_ld + k);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(xv[i][0], wv[0], fmaf(xv[i][1], wv[1], fmaf(xv[i][2], wv[2], fmaf(xv[i][3], wv[3], acc[i]))));
    }
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float t = wave_sum(acc[i]);
      if (lane == i) mine = t;
    }
    if (WPC == 4) {
      if (m0 > 0) __syncthreads();
      if (lane < 16) red[wave][lane] = mine;
      __syncthreads();
      mine = lane < 16 ? ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane] : 0.f;
    }
    if (lane < mc && n_ok && (WPC == 1 || wave == 0)) {
      float v = 0.f;
      with_act(p.act, [&](auto A) { v = act_c<decltype(A)::value>(mine * sc + sh); });
      const long long m = m0 + lane;
      if (res) v += res[m * p.res_ld + n];
      out[m * p.out_ld + n] = v;
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
void launch(const ConvP& p, float* slabs, int splits, hipStream_t s) {
  int tiles = cdiv(p.M, BM) * cdiv(p.Cout, BN);
  if (splits <= 1) {
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN>), dim3(tiles), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((conv_igemm_splitk_kernel<T, BM, BN, WM, WN>), dim3(tiles * splits), dim3(256), 0, s, p, slabs, splits);
    long long total = (long long)p.M * (p.Cout / Vec<T>::N);
    int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, s, slabs, splits, p);
  }
}

int tile_count(int variant, int M, int Cout) {
  static const int bm[4] = {256, 256, 128, 128}, bn[4] = {16, 32, 64, 128};
  return cdiv(M, bm[variant]) * cdiv(Cout, bn[variant]);
}

// K-splits for a problem: only when the grid would leave most CUs idle and K is long.
int pick_splits(int M, int Cout, int K, int dtype) {
  const int bk = is16(dtype) ? 32 : 16;
  const int nk = cdiv(K, bk);
  const int tiles = tile_count(cfp_conv2d_variant(M, Cout), M, Cout);
  if (tiles >= 128 || nk < 16) return 1;
  int s = cdiv(256, tiles);
  if (s > nk / 6) s = nk / 6;
  if (s > 32) s = 32;
  return s < 2 ? 1 : s;
}

template <typename T>
void dispatch(const ConvP& p, float* slabs, int splits, hipStream_t s) {
  switch (cfp_conv2d_variant(p.M, p.Cout)) {
    case 0: return launch<T, 256, 16, 4, 1>(p, slabs, splits, s);
    case 1: return launch<T, 256, 32, 4, 1>(p, slabs, splits, s);
    case 2: return launch<T, 128, 64, 2, 2>(p, slabs, splits, s);
    default: return launch<T, 128, 128, 2, 2>(p, slabs, splits, s);
  }
}

}  // namespace

// ---- debug knobs (tools/conv_bench.py): key 0 = force gen-2 variant (-1 auto), key 1 = force K-splits
// (-1 auto), key 2 = 1 routes bf16 through the first-generation kernel.  Not thread-safe; test use only.
static int g_force_variant = -1, g_force_splits = -1, g_use_v1 = 0;
int g_tput = 0;                 // cfp_debug_set key 17 (tools): 1 = every call plans for several batches in flight; callers say it per call with the
                                // CFP_CONV_IN_FLIGHT bit of cfp_conv2d_nhwc_ex's flags (Engine does while it captures in-flight slots).
                                // Side by side, what a launch costs is the resources it holds, not its own latency, and the sweep with four
                                // copies running together (tools/conv_bench.py --sweep --inflight 4, three runs, profiles/r3_conv_sweep_inflight4.json)
                                // prefers larger tiles than the isolated sweep the default plan is fitted on
int g_probe = 0;                // cfp_debug_set key 16: ConvP.probe
int g_x3_ad = 0;                // cfp_debug_set key 29: f16x3 implicit GEMMs with four row waves take their A-direct form (A values global -> registers)
int g_x3_small_m = 4800, g_x3_small_k = 1 << 30;      // cfp_debug_set keys 33 / 34: single-image plans take the 32 x 64 tile up to this many rows / this K
int g_x3_occ = 1;            // cfp_debug_set key 32: 0 = in-flight plans keep the unconstrained instantiations of the 64 x 64 / 128 x 32 / 64 x 128 tiles
static int x3_occ_of(int v) { return v == 13 ? 34 : v == 16 ? 35 : v == 15 ? 36 : v; }
static int x3_ad_of(int v) { return v == 26 ? 28 : v == 14 ? 29 : v == 13 ? 30 : v == 23 ? 31 : v == 16 ? 32 : v; }
int g_small_s2 = 1;             // cfp_debug_set key 15: 0 = three-stage 64x64 tiles for the small GEMMs (the round-2 plan)
int g_up_halo = 1;              // cfp_debug_set key 14: cfp_upsample_cat_conv3x3 through the halo kernel: 0 never, 1 where planned, 2 wherever it can run
int g_halo_s2 = 1;              // cfp_debug_set key 18: 0 = stride-2 convs never take the halo kernel
int g_halo = 1;                 // cfp_debug_set key 12: 0 = never take the whole-depth halo kernel, 2 = wherever it can run
// Where conv3x3_halo.hip beats the implicit GEMMs (tools/conv_bench.py --halo at batch 8, us halo / without: 614400 px x 16 ch, K = 360:
// 28 / 43; K = 144: 24 / 31; x 32 ch: 28 / 40; 153600 px x 160 ch: 38 / 50; 614400 x 128: 81 / 84) and where it does not (153600 px
// x 32 / 64 ch: 18-25 / 16-24, 38400 px: 13-23 / 11-23): the thin-output layers at full resolution and the wide expand convs.
// stride 2 (tools/conv_bench.py --halo, us with / without): stem 614400 x 40 x 72 25.5-27 / 29, 153600 x 64 x 144 23.4 / 26, 38400 x 160 x 360 20.8 / 23.4
extern int g_bin_head_x3_rows;  // conv_igemm_x3.hip
int g_x3_ln_fused = 1;          // cfp_debug_set key 26: 0 = the f16x3 GEMMs run a following LayerNorm as a second kernel
int g_halo_x3 = 1;              // cfp_debug_set key 24: 0 = the f16x3 3x3 convolutions never take the halo kernel
// where conv3x3_halo_x3.hip replaces the f16x3 implicit GEMM: many pixels (the halo and the weights of a channel block are fetched once
// per workgroup instead of nine times / once per 128 rows) and an input depth whose halo fits LDS
// (tools/conv_bench_x3.py --halo, us halo / implicit GEMM, alone and with four copies side by side; profiles/r4_conv_bench_x3_halo_*.txt):
// it wins where the input is SHALLOW -- 614400 px: 32 -> 128 ch 152 / 188 in flight (167 / 207 alone), 40 -> 16 63 / 83, 32 -> 32 60 / 73,
// 16 -> 16 23 / 41; 153600 px 40 -> 160 62 / 97; 38400 px 56 -> 224 33 / 41 -- and loses from 64 input channels up (the head's 128 -> 128:
// 692 / 573, 153600 px 168 -> 64: 182 / 136, 38400 px 128 -> 64: 39 / 23): a deep halo leaves one workgroup per CU, whose load phase
// (95-124 KB through registers) nothing overlaps, where the implicit GEMM pipelines its operand stream over K.
// alone (one graph at a time) the halo kernel already wins at half the pixels (batch 1: 19200 px x 160 x 360 19.3 vs 24.2 us, profiles/r4_conv_bench_x3_b1_alone.txt)
static bool halo_x3_wins(long long M, int Cin, int Cout, bool tput) { return M >= (tput ? 30000 : 15000) && Cin <= 56; }
// Deep inputs (Cin % 32 == 0, >= 64): which tile of the chunk-pipelined kernel, or -1 = the implicit GEMM.  Round 5: the single-chunk-buffer tiles
// (36: 128 channels x 8 x 16 pixels, 61 KB; 38: 64 channels x 16 x 16 pixels, 68 KB) hold TWO workgroups per CU, whose load / split / MFMA / store
// phases cover each other -- tools/probes/chunk_x3_inflight.py, us per call with four copies side by side (alone):
//   614400 px 128 -> 128 (head conv)  464 (490) against 529 (577) for the 256-pixel double-buffered tile 22;  76800 px (batch 1): 62 (84) vs 79 (95)
//   153600 px  64 -> 64   35 (54) vs 44 (54) for tile 24;   64 -> 32: 30 (46) vs 40 (51);   64 -> 128: 68 (74) vs 88 (96, implicit GEMM)
//    38400 px 128 -> 128  32 (59) vs 47 (80, implicit GEMM); 256 -> 128: 61 (92) vs 88 (143); 128 -> 64: 16 (42) vs 21 (38): in flight only
//     9600 px 256 -> 256  35 (72) vs 45 (73, implicit GEMM); 128 / 256 -> 128: the implicit GEMM stays ahead (12.4 / 22.4 vs 13.7 / 23.7)
// Round 5, late (tools/conv_bench_x3.py --halo with the single-buffer tiles in the sweep, batch 1 alone and batch 8 with four copies in flight):
//   * ONE input chunk (Cin = 32) is a problem for these tiles too: 32 -> 32 through tile 45 beats the whole-depth halo kernel (614400 px 50.5 vs 56.7 us in
//     flight, 76800 px alone 14.0 vs 14.6), and a single image's 32 -> 128 (the head's conv0, 76800 px) runs 28.4 us through tile 44 against 34.1;
//   * a single graph's 65 ... 128 output channels below 200 000 pixels: tile 44 (64 channels x 8 x 16 pixels, two channel blocks) instead of 36 --
//     the head conv of one image is 600 workgroups of tile 36 on 512 slots (two rounds for 1.17 rounds of work); 1 200 half-size ones: 74.7 vs 82.5 us;
//   * 32 output channels alone: tile 45 from 15 000 pixels (19200 px 64 -> 32: 14.5 vs 16.1 us through the implicit GEMM).
static int chunk_x3_variant(long long M, int Cin, int Cout, bool tput) {
  if (Cin % 32 != 0) return -1;
  if (Cin == 32) {
    if (Cout <= 32) return M >= 30000 ? 45 : -1;
    if (Cout > 64 && Cout <= 128 && !tput) return (M >= 30000 && M < 200000) ? 44 : -1;
    return -1;
  }
  if (Cout > 128) return (tput && M >= 9000) ? 36 : -1;
  if (Cout > 64) return M >= 30000 ? ((!tput && M < 200000) ? 44 : 36) : -1;
  if (Cout <= 32) return M >= (tput ? 30000 : 15000) ? 45 : -1;      // 32 channels x 8 x 16 pixels, 37 KB (four per CU): 614400 px 96 -> 32 (up4's first conv on the padded
                                                    // concatenation) 127 (136) vs 174 (187) for the 64-channel tile 38; 153600 px 64 -> 32: 20 (35) vs 30 (45)
  if (M >= 100000) return 38;
  if (M >= 30000) return tput ? 38 : 24;
  return -1;
}
static bool halo_wins_s2(long long M, int Cout) { return g_halo == 2 || (g_halo == 1 && g_halo_s2 && M >= 30000 && Cout <= 160); }
static bool halo_wins(long long M, int Cout, bool tput) {
  if (g_halo == 2) return true;
  if (g_halo == 1 && tput && M >= 100000 && Cout <= 64) return true;      // 153600 px x 32 / 64 ch: 8 x 16 pixel tiles, 10.6 -> 7.6 / 12.9 -> 10.8 / 20.8 -> 16.7 us per call in flight
  return g_halo == 1 && ((M >= 300000 && (Cout <= 32 || Cout == 128)) || (M >= 100000 && Cout > 128 && Cout <= 160) ||
                         (M >= 30000 && Cout > 160 && Cout <= 256));          // 38400 px x 224 ch (two 128-channel blocks): 19.8 / 22.9
}
void cfp_tail_x3_debug_set(int value);      // loftr_tail_x3.hip: key 35 = waves per workgroup of the fused tails (0 = by the row count)
void cfp_fold_debug_set(int v);
void cfp_tail16_debug_set(int value);
void cfp_attn_apply_debug_set(int value);
void cfp_dwl3_debug_set(int value);          // dwlarge_x3.hip: key 30 = 1: the 32 x 32 pixel tile for k = 31 (two workgroups per CU)
void cfp_dwl_wgrad_debug_set(int value);     // train_misc2.hip: key 23 = 1 keeps the VALU kernels for the large depthwise weight gradient
void cfp_wgrad_debug_set(int value);         // conv_bwd.hip: key 22 = workgroups a small weight-gradient launch aims for (sets the slab count)
void cfp_bn_debug_set(int key, int value);   // bn_train.hip: 20 / 21 = workgroup targets of the column reductions / elementwise sweeps
void cfp_attn_debug_set(int value);          // attention.hip: key 19 = waves the key / value reduction aims for when it splits a group's keys
void cfp_dw_debug_set(int key, int value);   // dwconv.hip: key 3 = channel vectors per workgroup, 4 = rows per strip (0 = automatic), 5 = 1 forces the VALU kernel
extern "C" int cfp_debug_set(int key, int value) {
  switch (key) {
    case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 11: cfp_dw_debug_set(key, value); return CFP_OK;
    case 30: cfp_dwl3_debug_set(value); return CFP_OK;
    case 32: g_x3_occ = value; return CFP_OK;
    case 33: g_x3_small_m = value; return CFP_OK;
    case 34: g_x3_small_k = value; return CFP_OK;
    case 35: cfp_tail_x3_debug_set(value); return CFP_OK;
    case 37: cfp_fold_debug_set(value); return CFP_OK;
    case 38: cfp_attn_apply_debug_set(value); return CFP_OK;
    case 39: cfp_tail16_debug_set(value); return CFP_OK;
    case 17: g_tput = value; return CFP_OK;
    case 16: g_probe = (g_probe & 16) | value; return CFP_OK;
    case 29: g_x3_ad = value; return CFP_OK;
    case 28: g_probe = (g_probe & ~16) | (value ? 16 : 0); return CFP_OK;      // f16x3 GEMM: 1 = the plain (not fragment-pipelined) K loop
    case 15: g_small_s2 = value; return CFP_OK;
    case 14: g_up_halo = value; return CFP_OK;
    case 19: cfp_attn_debug_set(value); return CFP_OK;
    case 20: case 21: cfp_bn_debug_set(key, value); return CFP_OK;
    case 22: cfp_wgrad_debug_set(value); return CFP_OK;
    case 23: cfp_dwl_wgrad_debug_set(value); return CFP_OK;
    case 18: g_halo_s2 = value; return CFP_OK;
    case 12: g_halo = value; return CFP_OK;
    case 24: g_halo_x3 = value; return CFP_OK;
    case 25: g_bin_head_x3_rows = value; return CFP_OK;
    case 26: g_x3_ln_fused = value; return CFP_OK;
    case 27: conv3x3_halo_debug_odd_pitch(value); return CFP_OK;
    case 13: conv3x3_halo_debug_stages(value); return CFP_OK;
    case 0: g_force_variant = value; return CFP_OK;
    case 1: g_force_splits = value; return CFP_OK;
    case 2: g_use_v1 = value; return CFP_OK;
    default: cfp_set_error("cfp_debug_set: unknown key"); return CFP_EINVAL;
  }
}

// Which tile configuration the first-generation (f32) kernel picks: 0 = 256x16, 1 = 256x32,
// 2 = 128x64, 3 = 128x128.
extern "C" int cfp_conv2d_variant(int M, int Cout) {
  if (Cout <= 16) return 0;
  if (Cout <= 32) return 1;
  if (Cout <= 64) return 2;
  long long t128 = (long long)cdiv(M, 128) * cdiv(Cout, 128);
  bool waste = (Cout % 128) != 0 && (Cout % 128) <= 64;   // e.g. Cout = 136, 160, 448
  return (t128 < 192 || waste) ? 2 : 3;
}

namespace {

// Plan for a bf16 problem: kernel generation, tile variant, K-splits.  The rules are fitted on the
// per-shape sweep of tools/conv_bench.py over this network's 80 distinct conv/linear problems at
// batch 8 (profiles/r1_conv_sweep.json: every variant x 1..16 splits, timed inside a replayed HIP
// graph).  What the sweep says:
//   * what decides is workgroups resident per CU, not prefetch depth: 2-stage 64x64 / 128x64 /
//     64x128 tiles (32-48 KB LDS, 3-5 workgroups per CU) beat the 3-stage ones almost everywhere;
//   * 128x128 (2 stages, 2 workgroups/CU) pays only for >= 10^7 outputs with K >= 512 or a long K;
//   * short-K, many-row problems are bandwidth-bound: the first-generation kernel (K-step 32,
//     12-24 KB LDS, up to 6 workgroups/CU) wins or ties there;
//   * the direct 3x3 kernel (LDS halo tile) wins with its smallest tile (8x16 pixels x 32 couts,
//     2 workgroups/CU) on the Cout <= 32 layers with K >= 576 and on the 1/16-scale DAPM convs;
//   * few-row / long-K problems (GSA sr convs) want 8 K-splits.
struct Plan2 { int variant, splits; bool gen1; int direct; };   // direct >= 0: conv3x3_direct variant

Plan2 plan2(long long M, int N, int K, int rpb, int B, bool allow_split, bool conv3x3s1 = false, bool tput = false) {
  Plan2 pl{4, 1, false, -1};
  const int nv = igemm2_num_variants();
  const bool big = M >= 100000, mid = M >= 30000;
  if (N <= 16) {
    if (big && K <= 144) pl.gen1 = true; else pl.variant = 16;
  } else if (N <= 32) {
    if (conv3x3s1 && K >= 576 && mid) pl.direct = 4;
    else if ((big && K <= 128) || (mid && K <= 64)) pl.gen1 = true;
    else pl.variant = mid ? 16 : 9;
  } else if (N <= 64) {
    if ((big && K <= 160) || (mid && K <= 64)) pl.gen1 = true;
    else pl.variant = big ? 14 : mid ? 13 : 4;
  } else {
    if ((big && K <= 288) || (mid && K <= 64)) pl.gen1 = true;
    else if ((M * N >= 614400ll * 128 && K >= 512) || (mid && K >= 2048)) pl.variant = 1;
    else if (mid) pl.variant = (N % 128 != 0 && N % 128 <= 64) ? 14 : 15;
    else if ((N >= 256 && K >= 2048) || N >= 1024) pl.variant = 12;
    else if (conv3x3s1 && N <= 128 && K >= 1152 && M >= 8000) pl.direct = 4;
    else pl.variant = 4;
  }
  if (rpb > 0) { pl.gen1 = false; pl.direct = -1; if (pl.variant != 12) pl.variant = 4; }
  // long K on fewer 64x64 tiles than CUs (the 1/32-scale project GEMMs: 160 tiles x 22 K-steps): two K groups per workgroup
  // (tools/conv_bench.py --kgroups: 12.8 -> 9.3 us at 2400 x 232 x 1392, 8.7 -> 7.1 at K = 816; no gain once two workgroups share a CU)
  if (pl.variant == 4 && !pl.gen1 && pl.direct < 0 && K >= 768) {
    const long long tm = rpb > 0 ? (long long)B * ((rpb + 63) / 64) : (M + 63) / 64;
    if (tm * ((N + 63) / 64) <= 256) pl.variant = 19;
  }
  if (allow_split && K >= 2048 && (M <= 1100 || ((M + 63) / 64) * ((N + 63) / 64) <= 48)) { pl.variant = 4; pl.splits = 8; pl.gen1 = false; pl.direct = -1; }
  // the 64x64 tile with two stages instead of three wherever K is not split: alone it is 3 % faster over the network's small GEMMs
  // (tools/conv_bench.py --kgroups, k13 vs k4), and at 32 KB instead of 48 KB of LDS five of them share a CU with the other batches'
  // kernels instead of three
  if (pl.variant == 4 && pl.splits == 1 && !pl.gen1 && pl.direct < 0 && g_small_s2) pl.variant = 13;
  // ... and the 64x128 tile with two stages instead of three (2400 x 1392 x 232: 9.3 vs 9.7 us alone, 4.7 vs 5.5 per call in flight); wide
  // pointwise layers at the 1/16 scale on 128x64 tiles (9600 x 816 x 136: equal alone, 8.4 vs 9.6 in flight)
  if (pl.variant == 12 && pl.splits == 1 && !pl.gen1 && pl.direct < 0 && g_small_s2) pl.variant = 15;
  if (pl.variant == 13 && !pl.gen1 && pl.direct < 0 && rpb == 0 && M >= 9000 && M < 30000 && N >= 512 && K <= 256 && g_small_s2) pl.variant = 14;
  // single images: the deep decoder convs on a few row tiles (1200 x 256 x 3528 / x 2304) run 21.4 / 14.6 us through the direct kernel's 8 x 16 pixel tiles
  // against 37.8 / 25.8 us on 64 x 128 GEMM tiles (tools/conv_bench.py --sweep --batch 1, profiles/r4_conv_bench_bf16_b1_alone.txt)
  if (!tput && !g_tput && conv3x3s1 && rpb == 0 && !pl.gen1 && pl.splits == 1 && pl.direct < 0 && M <= 2400 && N >= 256 && K >= 2048) pl.direct = 4;
  if ((tput || g_tput) && !pl.gen1 && pl.splits == 1) {
    if (pl.direct < 0 && M <= 20000 && N >= 256 && K >= 2048) pl.variant = 1;                    // 9600 x 256 x 3528: 20.6 vs 31.0 us per call in flight (alone: 51 vs 42)
    if (pl.direct < 0 && pl.variant == 13 && rpb == 0 && M >= 30000 && M < 100000 && N <= 64 && K >= 512) pl.variant = 14;      // 38400 x 64 x 1152: 7.8 vs 10.1
  }
  if (g_force_variant >= 200 && conv3x3s1 && g_force_variant - 200 < conv3x3_num_variants()) { pl.direct = g_force_variant - 200; pl.gen1 = false; pl.splits = 1; }
  else if (g_force_variant >= 0) pl.direct = -1;
  if (g_force_variant >= 0 && g_force_variant < nv) { pl.variant = g_force_variant; pl.gen1 = false; }
  if (g_force_splits >= 1) pl.splits = g_force_splits;
  if (pl.splits > 1 && pl.variant >= 19 && pl.variant <= 21 && g_force_variant < 0) pl.variant = 4;      // K groups and split-K exclude each other
  if (pl.direct >= 0) { pl.gen1 = false; pl.splits = 1; }
  return pl;
}

int pick_ln_variant(int Cout, long long M) {   // the tile must span exactly Cout channels
  switch (Cout) {
    case 128: return M <= 20000 ? 12 : 1;
    case 64: return 2;
    case 32: return 8;
    case 16: return 11;
    default: return -1;
  }
}

}  // namespace

// Plan of the f16x3 kernels (float32 storage, conv_igemm_x3.hip): the gen-2 rules on the equivalent 16-bit problem.  A K-step of 32
// channels moves the bytes and takes the LDS reads of a 64-channel 16-bit step, so the rules see twice the K.
static Plan2 plan_x3(long long M, int N, int K, int rpb, int B, bool allow_split, bool tput, bool piw_split = true) {
  Plan2 pl = plan2(M, N, 2 * (cdiv(K, 32) * 32), rpb, B, allow_split, false, tput);
  // short-K, many-row problems (the rules' "first-generation" answer) and the direct 3x3 kernel have no f16x3 form: row-heavy tiles
  if (pl.gen1 || pl.direct >= 0) pl.variant = N <= 16 ? 10 : N <= 32 ? 16 : N <= 64 ? 14 : K <= 64 ? 16 : 1;
  pl.gen1 = false; pl.direct = -1;
  // tools/conv_bench_x3.py (alone and four copies side by side, profiles/r4_conv_bench_x3_*.txt): the 128 x 128 tile as four row waves
  // (every A fragment converted once) beats the 2 x 2 layout everywhere (head conv 562 vs 614 us alone, 537 vs 569 in flight), and in
  // flight it also beats the 64 x 128 tile on the long-K decoder convs (9600 x 256 x 3528: 53 vs 58 us, 38400 x 224 x 504: 36 vs 43)
  if (pl.variant == 0 || pl.variant == 1) pl.variant = 26;
  if (tput && pl.variant == 15 && M >= 9000 && N >= 128 && N <= 256 && K >= 500) pl.variant = 26;
  if (pl.variant < 0 || pl.variant >= igemm_x3_num_variants()) pl.variant = 13;
  // single images (tools/conv_bench_x3.py --batch 1, profiles/r4_conv_bench_x3_b1_alone.txt): few row tiles and a long K -> the two-K-group
  // tile halves the serial chain (1200 x 256 x 3528: 43.7 vs 77.5 us, x 2304: 30.4 vs 49.7); a few hundred rows x many channels -> 32-row tiles
  // (300 x 1392 x 232: 7.7 vs 10.5 us)
  if (!tput && rpb == 0 && pl.splits <= 1 && M <= 2400 && K >= 2000 && N >= 128 && (pl.variant == 15 || pl.variant == 13)) pl.variant = 19;
  // round 5, late: REAL K splits measured against the two K groups (tools/probes/splitk_b1_probe.py -- the sweep tool's forced splits had silently
  // run un-split for want of a workspace): 1200 x 256 x 3528 22.3 us (eight splits + the reduce launch) vs 42.2, 1200 x 256 x 2304 20.4 (four) vs 27.5,
  // 4800 x 128 x 2808 29.5 (four) vs 32.8; from 1 152 deep down the K groups stay ahead (4800 x 128 x 1152: 17.1 vs 20.0)
  if (!tput && rpb == 0 && allow_split && pl.splits <= 1 && K >= 2000 && (M <= 1200 || (M <= 4800 && K >= 2500))) {
    pl.variant = 13; pl.splits = (M <= 1200 && K >= 3000) ? 8 : 4;
  }
  if (!tput && rpb == 0 && pl.splits <= 1 && M <= 512 && N >= 512 && K <= 512) pl.variant = 17;
  // round 5, late (full sweep of a single image's launches, tools/conv_bench_x3.py --batch 1): up to 4 800 rows the 32 x 64 tile (twice the workgroups,
  // four column waves) is ahead of 64 x 64 on every short-K layer (4800 x 56 x 224: 7.3 vs 8.6 us, 1200 x 512 x 128: 7.8 vs 9.4, 4800 x 256 x 64: 8.0 vs
  // 9.4; whole forward, same box: 3.03 vs 3.15 ms -- far more than the isolated launches predict); at 19 200 rows it is level or behind.  The 12 x 12 / 9 x 9 / 6 x 6 patch convolutions of the global attention (30 ... 130 rows, K ~ 5 000,
  // eight K splits): 16.8 / 17.9 / 16.8 us against 17.9 / 19.8 / 18.0 through 64-row tiles that are mostly padding
  if (!tput && pl.splits <= 1 && pl.variant == 13 && M <= g_x3_small_m && K <= g_x3_small_k) pl.variant = 17;
  if (!tput && rpb == 0 && pl.splits >= 4 && M <= 160 && (pl.variant == 4 || pl.variant == 13)) pl.variant = 17;
  // per-image weights (the squeeze-excite-folded project GEMMs) on a handful of row tiles with a long K -- single images: 300 x 232 x 1392 is
  // 20 tiles of 44 serial K-steps at ~0.8 us each (18.2 us, profiles/r4_conv_bench_x3_b1_alone.txt) on 20 of 256 CUs.  K splits instead of the
  // two K groups (round 5; the caller passes the slab workspace, cfp_conv2d_plan with rows_per_batch > 0 tells it the split count)
  if (rpb > 0 && !tput && piw_split && pl.splits <= 1) {
    const long long tiles = (long long)B * cdiv(rpb, 64) * cdiv(N, 64);
    const int nks = cdiv(K, 32);
    if (tiles <= 64 && nks >= 20) { pl.variant = 4; pl.splits = nks >= 40 ? 8 : 4; }
  }
  if (g_x3_ad && pl.splits <= 1) pl.variant = x3_ad_of(pl.variant);
  if (tput && g_x3_occ && pl.splits <= 1) pl.variant = x3_occ_of(pl.variant);      // one more resident workgroup per CU (conv_igemm_x3.hip, OCC)
  return pl;
}

extern "C" int cfp_conv2d_plan(int M, int Cout, int K, int KH, int stride, int dtype, int rows_per_batch, int B, int* variant,
                               int* splits) {
  if (dtype == CFP_F32X3) {
    const int cin3 = K / 9;
    const bool chunk3 = chunk_x3_variant(M, cin3, Cout, g_tput != 0) >= 0;
    if (KH == 3 && stride == 1 && K % 9 == 0 && cin3 % 8 == 0 && rows_per_batch <= 0 && g_halo_x3 && g_force_variant < 0 &&
        (chunk3 || halo_x3_wins(M, cin3, Cout, g_tput != 0))) {
      if (variant) *variant = 500;          // conv3x3_halo_x3.hip (the tile is chosen from Cout and the LDS the halo takes)
      if (splits) *splits = 1;
      return CFP_OK;
    }
    Plan2 pl = plan_x3(M, Cout, K, rows_per_batch, B, rows_per_batch <= 0, g_tput != 0);
    if (variant) *variant = 400 + pl.variant;
    if (splits) *splits = pl.splits;
    return CFP_OK;
  }
  if (is16(dtype) && !g_use_v1) {
    Plan2 pl = plan2(M, Cout, K, rows_per_batch, B, rows_per_batch <= 0, KH == 3 && stride == 1 && K % 9 == 0);
    const int cin = K / 9;
    if (KH == 3 && stride == 1 && K % 9 == 0 && cin % 8 == 0 && cin >= 8 && cin <= 64 && Cout % 8 == 0 && Cout <= 512 && rows_per_batch <= 0 &&
        (g_force_variant < 0 ? halo_wins(M, Cout, g_tput != 0) : g_force_variant >= 300)) {
      if (variant) *variant = 300;          // conv3x3_halo.hip (the tile is chosen from Cout and the pixel count)
      if (splits) *splits = 1;
    } else if (pl.direct >= 0) {
      if (variant) *variant = 200 + pl.direct;
      if (splits) *splits = 1;
    } else if (pl.gen1) {
      if (variant) *variant = cfp_conv2d_variant(M, Cout);
      if (splits) *splits = pick_splits(M, Cout, K, dtype);
    } else {
      if (variant) *variant = 100 + pl.variant;
      if (splits) *splits = pl.splits;
    }
  } else {
    if (variant) *variant = cfp_conv2d_variant(M, Cout);
    if (splits) *splits = pick_splits(M, Cout, K, dtype);
  }
  return CFP_OK;
}

extern "C" size_t cfp_conv2d_ws_bytes(int M, int Cout, int K, int dtype) {
  if (M <= 0 || Cout <= 0 || K <= 0) return 0;
  int v, s;
  cfp_conv2d_plan(M, Cout, K, 1, 1, dtype, 0, 1, &v, &s);
  return s <= 1 ? 0 : (size_t)s * M * Cout * sizeof(float);
}

extern "C" int cfp_layernorm(const void* in, int in_ld, const float* gamma, const float* beta, float eps,
                             const void* residual, int res_ld, void* out, int out_ld, int rows, int C, int dtype,
                             cfp_stream_t stream);

static int conv2d_impl(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                                  const void* residual, int res_ld, void* out, int out_ld, int B, int H, int W, int Cin,
                                  int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                                  int dtype, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                  int per_image_weights, void* ws, size_t ws_bytes, cfp_stream_t stream, int dil,
                                  float* mom = nullptr, size_t mom_floats = 0, int* mom_nsplit = nullptr, int* mom_rps = nullptr) {
  CFP_REQUIRE(in && w && out, CFP_EINVAL, "cfp_conv2d_nhwc: null pointer");
  if (mom_nsplit) *mom_nsplit = 0;
  if (mom_rps) *mom_rps = 0;
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_conv2d_nhwc: bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && Ho > 0 && Wo > 0,
              CFP_ESHAPE, "cfp_conv2d_nhwc: non-positive dimension");
  CFP_REQUIRE(Cin % ve == 0 && in_ld % ve == 0 && in_ld >= Cin, CFP_ESHAPE,
              "cfp_conv2d_nhwc: Cin / in_ld must be multiples of the 16-byte vector");
  CFP_REQUIRE(Cout % ve == 0 && out_ld % ve == 0 && out_ld >= Cout, CFP_ESHAPE,
              "cfp_conv2d_nhwc: Cout / out_ld must be multiples of the 16-byte vector");
  CFP_REQUIRE(!residual || (res_ld % ve == 0 && res_ld >= Cout), CFP_ESHAPE, "cfp_conv2d_nhwc: bad res_ld");
  CFP_REQUIRE(aligned16(in) && aligned16(w) && aligned16(out) && aligned16(residual) && aligned16(ws), CFP_EINVAL,
              "cfp_conv2d_nhwc: pointers must be 16-byte aligned");
  CFP_REQUIRE(dil >= 1 && (dil == 1 || (stride == 1 && !per_image_weights && !ln_gamma)), CFP_EINVAL, "cfp_conv2d_nhwc: bad input dilation");
  const long long Hd = (long long)(H - 1) * dil + 1, Wd = (long long)(W - 1) * dil + 1;      // zero-stuffed input size
  // (a data gradient may extend past the stuffed input: forward pixels the strided window never reached get a zero gradient)
  CFP_REQUIRE(dil > 1 || ((Ho - 1) * stride - pad_t + KH - 1 < Hd + KH && (Wo - 1) * stride - pad_l + KW - 1 < Wd + KW), CFP_ESHAPE,
              "cfp_conv2d_nhwc: output size inconsistent with input size");
  CFP_REQUIRE((long long)B * Ho * Wo < (1ll << 31) && (long long)KH * KW * Cin < (1ll << 31), CFP_ESHAPE,
              "cfp_conv2d_nhwc: problem too large");
  CFP_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), CFP_EINVAL, "cfp_conv2d_nhwc: ln_gamma / ln_beta must come together");
  CFP_REQUIRE(!ln_gamma || (aligned16(ln_gamma) && aligned16(ln_beta)), CFP_EINVAL, "cfp_conv2d_nhwc: LayerNorm parameters must be 16-byte aligned");
  ConvP p;
  p.in = in; p.w = w; p.out = out; p.res = residual; p.scale = scale; p.shift = shift;
  p.in_ld = in_ld; p.out_ld = out_ld; p.res_ld = res_ld;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.M = B * Ho * Wo; p.K = KH * KW * Cin; p.act = act; p.f16 = dtype == CFP_F16; p.dil = dil;
  p.pointwise = (KH == 1 && KW == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == H && Wo == W && dil == 1) ? 1 : 0;
  p.ln_gamma = nullptr; p.ln_beta = nullptr; p.ln_eps = ln_eps; p.rows_per_batch = 0; p.w_bstride = 0; p.k2 = 0;
  p.up_src = nullptr; p.up_ld = p.up_C = p.up_H = p.up_W = 0; p.up_sy = p.up_sx = 0.f;
  p.mom = nullptr; p.probe = g_probe;
  // channel moments of the output for the BatchNorm that follows (training): only the kernels that end with the output tile in LDS
  // produce them (gen-2 without split-K, gen-1 16-bit); every other route reports 0 row tiles and the caller runs its statistics pass
  const bool want_mom = mom && mom_nsplit && mom_rps && is16(dtype) && !residual && !ln_gamma && !per_image_weights && act == CFP_ACT_NONE && dil == 1;
  auto offer_mom = [&](int bm) {
    const long long tiles = cdiv(p.M, bm);
    if (!want_mom || (size_t)tiles * 2 * Cout > mom_floats) return false;
    p.mom = mom; *mom_nsplit = (int)tiles; *mom_rps = bm;
    return true;
  };
  const bool w2 = (per_image_weights & CFP_CONV_W2) != 0;
  const bool x3 = (per_image_weights & CFP_CONV_X3) != 0;
  const bool tput = (per_image_weights & CFP_CONV_IN_FLIGHT) != 0 || g_tput != 0;
  const bool tickets = (per_image_weights & CFP_CONV_WS_TICKETS) != 0;
  per_image_weights &= CFP_CONV_PER_IMAGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  if (x3) {
    // float32 storage, split-precision matrix math: `w` is the pre-split operand of cfp_pack_w_x3 (conv_igemm_x3.hip)
    CFP_REQUIRE(dtype == CFP_F32 && dil == 1 && !w2 && !mom, CFP_EINVAL, "cfp_conv2d_nhwc: CFP_CONV_X3 is for float32 storage (forward convolutions)");
    CFP_REQUIRE(KH < 256 && KW < 256 && (long long)H * W * in_ld < (1ll << 31) && (long long)cdiv(p.K, 32) * 64 * Cout < (1ll << 40), CFP_ESHAPE,
                "cfp_conv2d_nhwc: problem too large for the f16x3 kernel");
    CFP_REQUIRE(aligned16(scale) && aligned16(shift), CFP_EINVAL, "cfp_conv2d_nhwc: scale / shift must be 16-byte aligned (read as 4-float vectors)");
    const int rpb = per_image_weights ? Ho * Wo : 0;
    // 3x3 stride-1 layers with the whole-depth halo in LDS (conv3x3_halo_x3.hip): cfp_debug_set(0, 500 + v) forces its tile v, 500 + 99 its
    // automatic tile, 400 + v keeps the implicit GEMM
    // deep inputs (Cin % 32 == 0, >= 64): the chunk-pipelined form where it measured ahead of the implicit GEMM with four copies side by side
    // (tools/conv_bench_x3.py --halo --inflight 4: 153600 px 64 -> 64 ch 43.6 vs 49.4 us, 38400 px 128 -> 64 20.4 vs 22.0, head conv 556 vs 565)
    // (batch 1: the head conv, 76800 px, 84 us through chunk tile 24 against 118 us through the 128 x 128 implicit GEMM)
    const int chunk_v = g_halo_x3 ? chunk_x3_variant(p.M, Cin, Cout, tput) : -1;
    if (!per_image_weights && !ln_gamma && conv3x3_halo_x3_takes(p) &&
        (g_force_variant >= 500 || (g_force_variant < 0 && g_halo_x3 && (chunk_v >= 0 || halo_x3_wins(p.M, Cin, Cout, tput))))) {
      const int hv = g_force_variant >= 500 && g_force_variant < 599 ? g_force_variant - 500 : chunk_v;
      int rc = conv3x3_halo_x3_launch(hv, p, s);
      if (rc == 0) return cfp_check_launch("cfp_conv2d_nhwc");
      CFP_REQUIRE(g_force_variant < 500, CFP_EHIP, "cfp_conv2d_nhwc: the forced f16x3 halo variant cannot run this problem");
    }
    Plan2 pl = plan_x3(p.M, Cout, p.K, rpb, B, rpb == 0, tput);
    if (g_force_variant >= 400 && g_force_variant - 400 < igemm_x3_num_variants()) pl.variant = g_force_variant - 400;
    if (g_force_splits >= 1) pl.splits = g_force_splits;
    if (pl.splits > 1 && pl.variant >= 19) pl.variant = 4;
    if (rpb > 0) { p.rows_per_batch = rpb; p.w_bstride = (long long)Cout * cdiv(p.K, 32) * 64; }
    if (pl.splits > 1 && (!ws || ws_bytes < (size_t)pl.splits * p.M * Cout * sizeof(float) + (tickets ? CFP_CONV_TICKET_BYTES : 0))) {
      pl.splits = 1;
      if (rpb > 0) pl = plan_x3(p.M, Cout, p.K, rpb, B, false, tput, false);      // no workspace: the un-split plan of this problem (K groups)
      pl.splits = 1;
    }
    // LayerNorm over Cout in the epilogue when a four-row-wave tile spans exactly Cout channels; otherwise as a second kernel
    int ln_v = -1;
    if (ln_gamma && rpb == 0 && g_x3_ln_fused) ln_v = Cout == 128 ? (p.M >= 30000 ? 26 : 27) : Cout == 64 ? (p.M >= 100000 ? 14 : 13) : Cout == 32 ? 16 : Cout == 16 ? 11 : -1;
    // ... or, when K is split, inside the finishing sum (splitk_reduce_kernel): a row must be a power-of-two group of lanes of one wave
    const int cv4 = Cout / 4;
    const bool ln_in_reduce = ln_gamma && pl.splits > 1 && Cout % 4 == 0 && cv4 <= 64 && (cv4 & (cv4 - 1)) == 0 && g_x3_ln_fused;
    if (ln_in_reduce) {
      p.ln_gamma = ln_gamma; p.ln_beta = ln_beta;      // (the split GEMM ignores them: it only writes slabs)
    } else if (ln_v >= 0) {
      CFP_REQUIRE(aligned16(ln_gamma) && aligned16(ln_beta), CFP_EINVAL, "cfp_conv2d_nhwc: LayerNorm parameters must be 16-byte aligned");
      pl.variant = g_x3_ad ? x3_ad_of(ln_v) : ln_v; pl.splits = 1; p.ln_gamma = ln_gamma; p.ln_beta = ln_beta;
    } else if (ln_gamma) {
      p.res = nullptr;      // the residual is added after the LayerNorm kernel
    }
    // CFP_CONV_WS_TICKETS: the first CFP_CONV_TICKET_BYTES of `ws` are the caller's zeroed ticket area and the slabs follow -- the last workgroup
    // to reach an output tile finishes it (sum in split order + epilogue), no second launch
    unsigned* tk = nullptr;
    float* slabs = (float*)ws;
    if (tickets && pl.splits > 1) {
      int bm, bn, st;
      igemm_x3_variant_shape(pl.variant, &bm, &bn, &st);
      const long long tiles = (rpb > 0 ? (long long)B * cdiv(rpb, bm) : cdiv(p.M, bm)) * cdiv(Cout, bn);
      slabs = (float*)((char*)ws + CFP_CONV_TICKET_BYTES);      // the ticket area is never slab space, whether this launch takes tickets or not
      // (a LayerNorm that rides in the finishing sum keeps the reduce launch: the ticketed finish has no row-wide statistics)
      if (!ln_in_reduce && tiles <= CFP_TICKET_SLOTS && (long long)pl.splits * p.M * Cout * 4 < (1ll << 31) - 16) tk = (unsigned*)ws;      // (slabs behind one buffer descriptor)
    }
    int rc = igemm_x3_launch(pl.variant, p, slabs, pl.splits, s, tk);
    CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_conv2d_nhwc: f16x3 kernel launch failed");
    if (pl.splits > 1 && !tk) {
      long long total = (long long)p.M * (Cout / 4);
      int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
      hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)slabs, pl.splits, p);
    }
    int e = cfp_check_launch("cfp_conv2d_nhwc");
    if (e != CFP_OK || !ln_gamma || ln_v >= 0 || ln_in_reduce) return e;
    return cfp_layernorm(out, out_ld, ln_gamma, ln_beta, ln_eps, residual, res_ld, out, out_ld, p.M, Cout, dtype, stream);
  }

  CFP_REQUIRE(!w2 || (p.pointwise && is16(dtype) && !per_image_weights && dil == 1 && !g_use_v1), CFP_EINVAL,
              "cfp_conv2d_nhwc: two-term weights (CFP_CONV_W2) are for 16-bit pointwise layers with shared weights");
  const bool gen2 = dil == 1 && is16(dtype) && !g_use_v1 && p.K <= 16384 && KH < 256 && KW < 256 &&
                    ((long long)B * H * W + (long long)pad_t * W + pad_l) * in_ld * 2 < (1ll << 31) - 65536 &&      // 32-bit byte offsets into
                    (long long)Cout * p.K * 2 < (1ll << 31) - 65536;                                                    // buffer descriptors
  if (gen2) {
    CFP_REQUIRE(aligned16(scale) && aligned16(shift), CFP_EINVAL, "cfp_conv2d_nhwc: scale / shift must be 16-byte aligned (read as 4-float vectors)");
    const int rpb = per_image_weights ? Ho * Wo : 0;
    int ln_variant = ln_gamma ? pick_ln_variant(Cout, p.M) : -1;
    const bool c33 = KH == 3 && KW == 3 && stride == 1 && !ln_gamma && rpb == 0 && (long long)Cout * p.K < (1ll << 31);
    Plan2 pl = plan2(p.M, Cout, w2 ? 2 * cdiv(p.K, 64) * 64 : p.K, rpb, B, rpb == 0 && ln_variant < 0, c33, tput);
    if (w2) { pl.gen1 = false; pl.direct = -1; p.k2 = cdiv(p.K, 64); }
    // few input channels, many pixels: the whole-depth halo kernel (conv3x3_halo.hip); cfp_debug_set(0, 300 + v) forces its variant v,
    // any other forced variant / the gen-1 switch keeps the implicit GEMMs (A/B, tests)
    const bool c33s2 = KH == 3 && KW == 3 && stride == 2 && !ln_gamma && rpb == 0;
    if (!w2 && (c33 || c33s2) && !want_mom && conv3x3_halo_takes(p) &&
        (g_force_variant < 0 ? (Cin <= 64 && (stride == 1 ? halo_wins(p.M, Cout, tput) : halo_wins_s2(p.M, Cout))) : g_force_variant >= 300)) {
      int hv = g_force_variant >= 300 ? g_force_variant - 300 : -1;
      if (hv < 0 && stride == 2 && Cout > 64) hv = 3;
      if (hv < 0 && tput) hv = (p.M < 300000 && Cout <= 32) ? 7 : (p.M < 300000 && Cout <= 64) ? 3 : (Cout > 160 && Cout <= 224) ? 6 : -1;
      int rc = conv3x3_halo_launch(hv, p, s);
      if (rc == 0) return cfp_check_launch("cfp_conv2d_nhwc");
      CFP_REQUIRE(g_force_variant < 0, CFP_EHIP, "cfp_conv2d_nhwc: the forced halo variant cannot run this problem");
    }
    if (pl.direct >= 0 && !want_mom) {
      int rc = conv3x3_launch(pl.direct, p, s);
      CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_conv2d_nhwc: direct 3x3 kernel launch failed");
      return cfp_check_launch("cfp_conv2d_nhwc");
    }
    if (pl.direct >= 0) { pl.direct = -1; pl.variant = 14; pl.splits = 1; }      // statistics wanted: the implicit GEMM (it ends with the tile in LDS)
    if (pl.gen1 && !ln_gamma) goto gen1_path;
    if (ln_variant >= 0 && g_force_variant < 0) pl.variant = ln_variant;
    if (ln_variant >= 0) { p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; pl.splits = 1; }
    if (rpb > 0) { p.rows_per_batch = rpb; p.w_bstride = (long long)Cout * p.K; pl.splits = 1; }
    if (pl.splits > 1 && (!ws || ws_bytes < (size_t)pl.splits * p.M * Cout * sizeof(float))) pl.splits = 1;
    const bool ln_after = ln_gamma && ln_variant < 0;   // Cout has no exact-width tile: LayerNorm as a second kernel
    if (ln_after) p.res = nullptr;
    if (pl.splits == 1) { int bm, bn, st; igemm2_variant_shape(pl.variant, &bm, &bn, &st); offer_mom(bm); }
    int rc = igemm2_launch(pl.variant, p, (float*)ws, pl.splits, s);
    CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_conv2d_nhwc: gen-2 kernel launch failed");
    if (pl.splits > 1) {
      long long total = (long long)p.M * (Cout / 8);
      int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
      if (p.f16) hipLaunchKernelGGL(splitk_reduce_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (const float*)ws, pl.splits, p);
      else hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const float*)ws, pl.splits, p);
    }
    int e = cfp_check_launch("cfp_conv2d_nhwc");
    if (e != CFP_OK || !ln_after) return e;
    return cfp_layernorm(out, out_ld, ln_gamma, ln_beta, ln_eps, residual, res_ld, out, out_ld, p.M, Cout, dtype, stream);
  }

gen1_path:
  CFP_REQUIRE(!w2, CFP_ESHAPE, "cfp_conv2d_nhwc: two-term weights need the gen-2 kernel (tensor too large for 32-bit offsets)");
  // first-generation kernels (f32 parity mode, short-K bf16): per-image weights run image by image,
  // LayerNorm as a second kernel
  if (dtype == CFP_F32 && p.pointwise && p.M <= 64 && !per_image_weights && !ln_gamma && !g_use_v1) {
    if (p.K >= 512 && Cout <= 1024) hipLaunchKernelGGL(rowgemm_f32_kernel<4>, dim3(Cout), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(rowgemm_f32_kernel<1>, dim3(cdiv(Cout, 4)), dim3(256), 0, s, p);
    return cfp_check_launch("cfp_conv2d_nhwc");
  }
  if (ln_gamma) p.res = nullptr;
  const int nimg = per_image_weights ? B : 1;
  const size_t esz = is16(dtype) ? 2 : 4;
  for (int b = 0; b < nimg; ++b) {
    ConvP q = p;
    if (per_image_weights) {
      q.B = 1; q.M = Ho * Wo;
      q.in = (const char*)in + (size_t)b * H * W * in_ld * esz;
      q.out = (char*)out + (size_t)b * Ho * Wo * out_ld * esz;
      if (q.res) q.res = (const char*)q.res + (size_t)b * Ho * Wo * res_ld * esz;
      q.w = (const char*)w + (size_t)b * Cout * p.K * esz;
    }
    int splits = pick_splits(q.M, Cout, q.K, dtype);
    if (splits > 1 && (!ws || ws_bytes < (size_t)splits * q.M * Cout * sizeof(float))) splits = 1;
    if (splits == 1 && is16(dtype) && !per_image_weights) {
      const int v1 = cfp_conv2d_variant(q.M, Cout);
      if (offer_mom(v1 <= 1 ? 256 : 128)) q.mom = p.mom;
    }
    if (dtype == CFP_BF16) dispatch<bf16_t>(q, (float*)ws, splits, s); else if (dtype == CFP_F16) dispatch<f16_t>(q, (float*)ws, splits, s); else dispatch<float>(q, (float*)ws, splits, s);
  }
  int e = cfp_check_launch("cfp_conv2d_nhwc");
  if (e != CFP_OK || !ln_gamma) return e;
  return cfp_layernorm(out, out_ld, ln_gamma, ln_beta, ln_eps, residual, res_ld, out, out_ld, p.M, Cout, dtype, stream);
}

extern "C" int cfp_conv2d_nhwc_ex(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                                  const void* residual, int res_ld, void* out, int out_ld, int B, int H, int W, int Cin,
                                  int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                                  int dtype, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                  int per_image_weights, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  return conv2d_impl(in, in_ld, w, scale, shift, residual, res_ld, out, out_ld, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo,
                     act, dtype, ln_gamma, ln_beta, ln_eps, per_image_weights, ws, ws_bytes, stream, 1);
}

// cfp_conv2d_nhwc + the per-row-tile channel moments of its (rounded, stored) output for a batch-statistics BatchNorm that follows
// (training: timm / torch `conv -> BatchNorm2d` in model.train()).  mom: [row tiles][2][Cout] float32 (mean, M2 about it); on return
// *nsplit = row tiles written (0: this problem's kernel does not produce them -- run cfp_bn_train_stats) and *rows_per_split = rows
// per tile; feed them to cfp_bn_train_stats_partials.
extern "C" int cfp_conv2d_nhwc_moments(const void* in, int in_ld, const void* w, const float* bias, void* out, int out_ld, int B, int H, int W,
                                       int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int dtype,
                                       void* ws, size_t ws_bytes, float* mom, size_t mom_floats, int* nsplit, int* rows_per_split,
                                       cfp_stream_t stream) {
  CFP_REQUIRE(mom && nsplit && rows_per_split, CFP_EINVAL, "cfp_conv2d_nhwc_moments: null moments pointer");
  return conv2d_impl(in, in_ld, w, nullptr, bias, nullptr, 0, out, out_ld, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo,
                     CFP_ACT_NONE, dtype, nullptr, nullptr, 0.f, 0, ws, ws_bytes, stream, 1, mom, mom_floats, nsplit, rows_per_split);
}

int conv3x3_up_launch(int v, const ConvP& p, hipStream_t s);      // conv3x3_direct.hip

// UpSampleBN's first half in one launch (decoder.py:51-58): F.interpolate(low, size=(H, W), bilinear, align_corners=True) -> cat([up, skip],
// dim=1) -> conv3x3 (+ folded BatchNorm + activation).  Neither the upsampled tensor nor the concatenation exists in memory: the direct
// 3x3 kernel computes the upsampled channel chunks into its LDS halo tile and fetches the skip chunks.  Bit-identical to
// cfp_resize_bilinear + cfp_conv2d_nhwc on the concatenation buffer.  -> CFP_ESHAPE when the shape is not taken (caller keeps the pair).
extern "C" int cfp_upsample_cat_conv3x3(const void* low, int low_ld, int Hs, int Ws, int Cup, const void* skip, int skip_ld, int Cskip,
                                        const void* w, const float* scale, const float* shift, void* out, int out_ld, int B, int H, int W,
                                        int Cout, int act, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(low && skip && w && out, CFP_EINVAL, "cfp_upsample_cat_conv3x3: null pointer");
  if (dtype == CFP_F32X3) {
    // float32 tensors, f16x3 matrix math (round 5): the chunk-pipelined kernel with two sources (conv3x3_halo_x3.hip, UP = true).  `w` is the
    // pre-split operand packed over the PADDED channel axis [Cout][9][Cup + 32 ceil(Cskip / 32)] (ops.pack_w_x3_cat)
    CFP_REQUIRE(B > 0 && H > 1 && W > 1 && Hs > 0 && Ws > 0 && Cup > 0 && Cup % 32 == 0 && Cskip >= 4 && Cskip % 4 == 0 && Cout % 4 == 0 && Cout > 0 &&
                    low_ld % 4 == 0 && low_ld >= Cup && skip_ld % 4 == 0 && skip_ld >= Cskip && out_ld % 4 == 0 && out_ld >= Cout, CFP_ESHAPE,
                "cfp_upsample_cat_conv3x3 (f32x3): need Cup % 32 == 0, Cskip % 4 == 0 and 16-byte channel vectors");
    CFP_REQUIRE(aligned16(low) && aligned16(skip) && aligned16(w) && aligned16(out) && aligned16(scale) && aligned16(shift), CFP_EINVAL,
                "cfp_upsample_cat_conv3x3: pointers must be 16-byte aligned");
    CFP_REQUIRE(((long long)H * W) * skip_ld < (1ll << 31) - 65536 && ((long long)Hs * Ws) * low_ld < (1ll << 31) - 65536 && (long long)B * H * W < (1ll << 31),
                CFP_ESHAPE, "cfp_upsample_cat_conv3x3: tensor too large for 32-bit offsets");
    ConvP p{};
    p.in = skip; p.w = w; p.out = out; p.res = nullptr; p.scale = scale; p.shift = shift;
    p.in_ld = skip_ld; p.out_ld = out_ld; p.res_ld = 0;
    p.B = B; p.H = H; p.W = W; p.Cin = Cup + Cskip; p.Ho = H; p.Wo = W; p.Cout = Cout;
    p.KH = 3; p.KW = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1;
    p.M = B * H * W; p.K = 9 * (Cup + cdiv(Cskip, 32) * 32); p.act = act; p.dil = 1;
    p.up_src = low; p.up_ld = low_ld; p.up_C = Cup; p.up_H = Hs; p.up_W = Ws;
    p.up_sy = (float)(Hs - 1) / (float)(H - 1);        // cfp_resize_bilinear's own expression
    p.up_sx = (float)(Ws - 1) / (float)(W - 1);
    const int v = g_force_variant >= 540 && g_force_variant <= 542 ? g_force_variant - 500 : (Cout <= 32 ? 40 : Cout <= 64 ? 41 : 42);
    int rc = conv3x3_halo_x3_launch(v, p, reinterpret_cast<hipStream_t>(stream));
    CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_upsample_cat_conv3x3 (f32x3): kernel launch failed");
    return cfp_check_launch("cfp_upsample_cat_conv3x3");
  }
  CFP_REQUIRE(is16(dtype), CFP_ESHAPE, "cfp_upsample_cat_conv3x3: bf16 / f16 storage or CFP_F32X3");
  CFP_REQUIRE(B > 0 && H > 1 && W > 1 && Hs > 0 && Ws > 0 && Cup > 0 && Cup % 64 == 0 && Cskip > 0 && Cskip % 8 == 0 && Cout % 8 == 0 &&
                  low_ld % 8 == 0 && low_ld >= Cup && skip_ld % 8 == 0 && skip_ld >= Cskip && out_ld % 8 == 0 && out_ld >= Cout, CFP_ESHAPE,
              "cfp_upsample_cat_conv3x3: need Cup % 64 == 0 and 16-byte channel vectors");
  CFP_REQUIRE(aligned16(low) && aligned16(skip) && aligned16(w) && aligned16(out), CFP_EINVAL, "cfp_upsample_cat_conv3x3: pointers must be 16-byte aligned");
  const int Cin = Cup + Cskip;
  CFP_REQUIRE(((long long)B * H * W) * skip_ld * 2 < (1ll << 31) - 65536 && ((long long)Hs * Ws) * low_ld * 2 < (1ll << 31) - 65536 &&
                  (long long)Cout * 9 * Cin * 2 < (1ll << 31) - 65536 && (long long)B * H * W < (1ll << 31), CFP_ESHAPE,
              "cfp_upsample_cat_conv3x3: tensor too large for 32-bit offsets");
  ConvP p;
  const size_t esz = 2;
  p.in = (const char*)skip - (size_t)Cup * esz;      // virtual base: channel c >= Cup of the concatenation is skip channel c - Cup
  p.w = w; p.out = out; p.res = nullptr; p.scale = scale; p.shift = shift;
  p.in_ld = skip_ld; p.out_ld = out_ld; p.res_ld = 0;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.Cout = Cout;
  p.KH = 3; p.KW = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1;
  p.M = B * H * W; p.K = 9 * Cin; p.act = act; p.f16 = dtype == CFP_F16; p.dil = 1; p.pointwise = 0;
  p.ln_gamma = nullptr; p.ln_beta = nullptr; p.ln_eps = 0.f; p.rows_per_batch = 0; p.w_bstride = 0; p.k2 = 0;
  p.mom = nullptr; p.probe = 0;
  p.up_src = low; p.up_ld = low_ld; p.up_C = Cup; p.up_H = Hs; p.up_W = Ws;
  p.up_sy = (float)(Hs - 1) / (float)(H - 1);        // cfp_resize_bilinear's own expression (bit-identical source coordinates)
  p.up_sx = (float)(Ws - 1) / (float)(W - 1);
  // few concatenated channels and a thin output at many pixels (up4: 64 + 24 -> 32 at 240 x 320): the whole-depth halo kernel blends
  // into its resident tile once per workgroup; the 64-channel-chunk direct kernel otherwise.  cfp_debug_set(14, 0) keeps the direct kernel.
  if (g_up_halo && conv3x3_halo_takes(p) && Cout <= 64 && (g_up_halo == 2 || (long long)B * H * W >= 300000)) {
    int rc = conv3x3_halo_launch(g_force_variant >= 300 ? g_force_variant - 300 : -1, p, reinterpret_cast<hipStream_t>(stream));
    if (rc == 0) return cfp_check_launch("cfp_upsample_cat_conv3x3");
  }
  const int v = Cout <= 16 ? 5 : (Cout <= 32 ? 4 : (Cout <= 64 ? 1 : 0));
  int rc = conv3x3_up_launch(v, p, reinterpret_cast<hipStream_t>(stream));
  CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_upsample_cat_conv3x3: kernel launch failed");
  return cfp_check_launch("cfp_upsample_cat_conv3x3");
}

// Data gradient of a convolution: dX [B,H,W,Cin] from dY [B,Ho,Wo,Cout] and the flipped weights of cfp_conv2d_weight_flip
// ([Cin][KH][KW][Cout]).  dX = conv_stride1(zero-stuff(dY, stride), Wt) with padding K-1-pad, accumulate into dX when
// `accumulate` (the skip connections' gradient) by passing dX as the residual.
extern "C" int cfp_conv2d_dgrad(const void* dy, int dy_ld, const void* wt, void* dx, int dx_ld, int B, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate, int dtype, void* ws,
                                size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(pad_t >= 0 && pad_l >= 0 && pad_t < KH && pad_l < KW, CFP_ESHAPE, "cfp_conv2d_dgrad: padding must be smaller than the kernel");
  return conv2d_impl(dy, dy_ld, wt, nullptr, nullptr, accumulate ? dx : nullptr, dx_ld, dx, dx_ld, B, Ho, Wo, Cout, Cin, KH, KW, 1,
                     KH - 1 - pad_t, KW - 1 - pad_l, H, W, CFP_ACT_NONE, dtype, nullptr, nullptr, 0.f, 0, ws, ws_bytes, stream, stride);
}

extern "C" int cfp_conv2d_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                               const void* residual, int res_ld, void* out, int out_ld, int B, int H, int W, int Cin,
                               int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                               int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  return cfp_conv2d_nhwc_ex(in, in_ld, w, scale, shift, residual, res_ld, out, out_ld, B, H, W, Cin, Cout, KH, KW, stride,
                            pad_t, pad_l, Ho, Wo, act, dtype, nullptr, nullptr, 0.f, 0, ws, ws_bytes, stream);
}
