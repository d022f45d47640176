// bf16 implicit-GEMM convolution / linear layer, second generation: LDS-DMA staged, multi-stage.
//
// Same GEMM view as conv_igemm.hip (C[m,n] = sum_k A[m,k] Wt[n,k], A never materialised), but built
// for the regime this network actually runs in on a 256-CU chip at batch 8: most layers are a
// single wave of workgroups (or less), so one workgroup per CU has to hide its own load latency.
//
//   * K-step = 64 bf16 (128-byte LDS rows).  Both operand tiles go global -> LDS directly with
//     `global_load_lds_dwordx4` (no VGPR staging, no ds_write): lane l of a wave-instruction
//     lands at base + 16*l, i.e. 8 rows x 8 chunks.  Out-of-image taps, the M / N / K tails all
//     read a 16-byte zero word instead (the source address is per lane), so the halo costs no
//     branch and no LDS store.
//     (Round 2 tried `buffer_load_dwordx4 ... lds` through descriptors instead -- 32-bit offsets, hardware zero fill for the halo,
//     wave-uniform tap offsets in soffset: 0.6 instead of 4.6 VALU instructions per MFMA in the 128x128 loop -- and measured it
//     SLOWER: 2.64 vs 2.59 ms per batch-8 step, up3.a 67 vs 58 us (gpurun_out/ab_loader.txt, tools/probes/).  The loop is bound
//     by LDS bandwidth and DMA latency, not VALU issue; the descriptor form is kept only in head_fused.hip.)
//   * LDS rows are XOR-swizzled on the 16-byte chunk (physical = logical ^ (row & 7)); because the
//     DMA destination is lane-linear the swizzle is applied to the per-lane SOURCE chunk, and
//     again on the ds_read_b128 side.  Each 16-lane read group of the 16x16x32 operand pattern
//     then covers all 64 banks.
//   * STAGES (3 or 4) K-steps are in flight: the wait before consuming step k is a counted
//     `s_waitcnt vmcnt(N)` that leaves the younger stages outstanding, followed by one raw
//     `s_barrier` per K-step (a __syncthreads() would drain the DMA queue).
//   * im2col addressing: every lane always fetches the same logical chunk column, so its (kh, kw,
//     channel) position advances by 64 channels per K-step with compare-and-subtract, no division
//     (an LDS lookup table would make hipcc drain the DMA queue before each table read).
//   * epilogue: scale/shift (folded BatchNorm / bias), activation, optional LayerNorm over the
//     channel axis (when one tile spans all output channels), optional residual; results leave
//     through LDS as 16-byte vectors along the channel axis.
//   * `rows_per_batch > 0`: M-tiles never straddle an image and image b uses weights
//     w + b * w_bstride (squeeze-excite gates folded into per-image project weights).
//   * split-K as in conv_igemm.hip: f32 slabs + the shared reduce kernel.
#include "igemm_core.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KG = 2 (round 3): EIGHT waves, two groups of four.  Both groups own the whole tile; group g takes the K-steps k0 + g, k0 + g + 2, ...
// through its own LDS stages, and the groups' accumulators are added through LDS before the epilogue.  For the launches that cannot
// fill the chip (a 64 x 64 tile of a 9600 x 136 x 816 layer is 450 workgroups, each walking 13 K-steps at ~0.57 us per step whatever
// the stage count -- tools/conv_bench.py) this halves the K loop's latency at the cost of one LDS round trip, without the slabs and
// the reduce launch of the split-K form.
template <typename H, int BM, int BN, int WM, int WN, int STAGES, bool SPLITK, int KG = 1>
__global__ __launch_bounds__(256 * KG) void igemm2_kernel(ConvP p, float* __restrict__ slabs, int splits) {
  static_assert(WM * WN == 4, "four waves per group");
  static_assert(KG == 1 || (KG == 2 && !SPLITK), "K groups are the in-workgroup alternative to split-K");
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int NA = BM / 32;                    // A DMA instructions per wave per K-step
  constexpr int NBG = BN / 8;                    // 8-row groups of the W tile
  constexpr int NB = (NBG + 3) / 4;              // W DMA instructions per wave per K-step
  constexpr int LPS = NA + NB;                   // DMA instructions per wave per stage
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert((STAGES - 2) * LPS <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 2 ? wave8 >> 2 : 0;        // K group
  const int wave = wave8 & 3;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;              // logical K-chunk this lane fetches
  unsigned char* gsm = smem + kg * (STAGES * STAGE_BYTES);      // this group's operand stages

  // ---- tile coordinates -------------------------------------------------------------------
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = xcd_remap(blockIdx.x, gridDim.x);   // an XCD's L2 sees a contiguous run of tiles (shared A rows / halos / weights)
  int sp = 0;
  if constexpr (SPLITK) { sp = bid % splits; bid /= splits; }
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  int m0, m_end;
  const H* __restrict__ wt = reinterpret_cast<const H*>(p.w);
  if (p.rows_per_batch > 0) {
    const int tpb = (p.rows_per_batch + BM - 1) / BM;
    const int b = tile_m / tpb;
    m0 = b * p.rows_per_batch + (tile_m % tpb) * BM;
    m_end = (b + 1) * p.rows_per_batch;
    wt += (long long)b * p.w_bstride;
  } else {
    m0 = tile_m * BM;
    m_end = p.M;
  }
  const int n0 = tile_n * BN;
  const int nk_all = p.k2 > 0 ? 2 * p.k2 : ((p.K + 63) >> 6);
  const int wrow = p.k2 > 0 ? 2 * p.k2 * 64 : p.K;          // elements per weight row (two-term weights: [hi | lo], each k2 K-steps long)
  int k0 = 0, k1 = nk_all;
  if constexpr (SPLITK) {
    const int per = (nk_all + splits - 1) / splits;
    k0 = sp * per;
    k1 = min(nk_all, k0 + per);
  }
  const H* __restrict__ in = reinterpret_cast<const H*>(p.in);

  // ---- per-lane row bookkeeping -----------------------------------------------------------
  const H* a_ptr[NA];
  int a_hi0[NA], a_wi0[NA];
  unsigned a_okmask = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + (i * 4 + wave) * 8 + rsub;
    const bool ok = m < m_end;
    if (ok) a_okmask |= 1u << i;
    const int mm = ok ? m : m0;
    if (p.pointwise) {
      a_ptr[i] = in + (long long)mm * p.in_ld;
      a_hi0[i] = 0;
      a_wi0[i] = 0;
    } else {
      const int wo = mm % p.Wo;
      const int t = mm / p.Wo;
      const int ho = t % p.Ho;
      const int b = t / p.Ho;
      a_hi0[i] = ho * p.stride - p.pad_t;
      a_wi0[i] = wo * p.stride - p.pad_l;
      a_ptr[i] = in + (((long long)b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.in_ld;   // virtual when in the halo
    }
  }
  const H* b_ptr[NB];
  unsigned b_okmask = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + ((j * 4 + wave) % NBG) * 8 + rsub;
    const bool ok = n < p.Cout;
    if (ok) b_okmask |= 1u << j;
    b_ptr[j] = wt + (long long)(ok ? n : 0) * wrow;
  }
  const H* zsrc = reinterpret_cast<const H*>(g_zero16);
  // im2col position (tap row/column, channel offset) of this lane's chunk in the NEXT K-step to be
  // issued; advanced by 64 channels per K-step without divisions (K-steps are issued in order).
  int i_cc = 0, i_kh = 0, i_kw = 0;
  if (!p.pointwise) {
    const int kk = ((k0 + kg) * 8 + lc) * 8;
    const int tap = kk / p.Cin;
    i_cc = kk - tap * p.Cin;
    i_kh = tap / p.KW;
    i_kw = tap - i_kh * p.KW;
  }

  auto issue = [&](int ks, int buf) {
    unsigned char* sA = gsm + buf * STAGE_BYTES;
    unsigned char* sB = sA + BM * 128;
    const int kk = (ks * 8 + lc) * 8;
    bool kok = kk < p.K;
    if (p.pointwise) {
      // two-term weights (p.k2 > 0): the second half of the K loop meets the SAME activations again
      const int kka = ((p.k2 > 0 && ks >= p.k2 ? ks - p.k2 : ks) * 8 + lc) * 8;
      const bool koka = kka < p.K;
      if (p.k2 > 0) kok = true;                      // weight rows are zero-padded to whole K-steps
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = koka && ((a_okmask >> i) & 1u);
        glds16(ok ? a_ptr[i] + kka : zsrc, sA + (i * 4 + wave) * 1024);
      }
    } else {
      const int off = (i_kh * p.W + i_kw) * p.in_ld + i_cc;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = kok && ((a_okmask >> i) & 1u) && (unsigned)(a_hi0[i] + i_kh) < (unsigned)p.H &&
                        (unsigned)(a_wi0[i] + i_kw) < (unsigned)p.W;
        glds16(ok ? a_ptr[i] + off : zsrc, sA + (i * 4 + wave) * 1024);
      }
      i_cc += 64 * KG;
      while (i_cc >= p.Cin) {
        i_cc -= p.Cin;
        if (++i_kw == p.KW) { i_kw = 0; ++i_kh; }
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = kok && ((b_okmask >> j) & 1u);
      glds16(ok ? b_ptr[j] + kk : zsrc, sB + ((j * 4 + wave) % NBG) * 1024);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: STAGES-1 K-steps in flight (of this group's steps k0 + kg, k0 + kg + KG, ...) ------------------
  const int nkg = max(0, (k1 - k0 - kg + KG - 1) / KG);      // this group's K-steps
  const int nit = (k1 - k0 + KG - 1) / KG;                     // barrier rounds = group 0's steps (the other group has as many or one fewer)
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nkg) issue(k0 + kg + s * KG, s);

  const int a_row0 = wm * (BM / WM), b_row0 = wn * (BN / WN);
  for (int it = 0; it < nit; ++it) {
    const int buf = it % STAGES;
    const int ahead = min(nkg - 1 - it, STAGES - 2);   // younger stages that may stay in flight
    if (ahead >= 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * LPS>();
    else if (ahead == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                     // stage `buf` landed for every wave; stage buf-1 fully consumed
    asm volatile("" ::: "memory");
    if (KG == 2 && it >= nkg) continue;               // the second group's missing last step (odd step count): it only keeps the barrier count
    if (it + STAGES - 1 < nkg && !(p.probe & 1)) issue(k0 + kg + (it + STAGES - 1) * KG, (it + STAGES - 1) % STAGES);
    if (p.probe & 2) continue;

    const unsigned char* cA = gsm + buf * STAGE_BYTES + a_row0 * 128;
    const unsigned char* cB = gsm + buf * STAGE_BYTES + BM * 128 + b_row0 * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 af[TM], bfr[TN];
      const int pc = ((s * 4 + fq) ^ (fr & 7)) * 16;   // tile row offsets are multiples of 16 rows
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(cA + (i * 16 + fr) * 128 + pc);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const s16x8*>(cB + (j * 16 + fr) * 128 + pc);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma16<H>(bfr[j], af[i], acc[i][j]);      // transposed: acc[r] = 4 CONSECUTIVE CHANNELS (n = 4 fq + r) of pixel row fr
    }
  }
  wait_vmcnt<0>();
  __syncthreads();   // every wave is done with the operand stages: LDS becomes the C tile
  if constexpr (KG == 2) {
    // the second group's accumulators travel through LDS (float32, lane-contiguous 16-byte pieces) and are added by the first
    static_assert(BM * BN * 4 <= KG * STAGES * STAGE_BYTES, "accumulator exchange must fit in the operand LDS");
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + wave * (TM * TN * 64) + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) xch[(i * TN + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 o = xch[(i * TN + j) * 64];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += o[r];
        }
    }
    __syncthreads();
  }

  // The accumulators are held transposed (weights as the MFMA's row operand): a lane owns four consecutive output channels of one
  // pixel, so the epilogue moves 8-byte (16-bit) / 16-byte (float32 slab) vectors instead of sixteen 2-byte LDS writes per thread.
  if constexpr (SPLITK) {
    float* slab = slabs + (long long)sp * p.M * p.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = m0 + a_row0 + i * 16 + fr;
        const int n = n0 + b_row0 + j * 16 + fq * 4;
        if (m < m_end && n < p.Cout) *reinterpret_cast<f32x4*>(slab + (long long)m * p.Cout + n) = acc[i][j];      // Cout % 8 == 0, n % 4 == 0
      }
    return;
  } else {
    constexpr int CP = BN + 8;   // C-tile pitch in elements (one 16-byte chunk of padding)
    static_assert(BM * CP * 2 <= STAGES * STAGE_BYTES, "C tile must fit in the operand LDS");
    H* sC = reinterpret_cast<H*>(smem);
    f32x4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + b_row0 + j * 16 + fq * 4;
      const bool ok = n < p.Cout;
      sc[j] = (ok && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[j] = (ok && p.shift) ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (kg == 0) with_act(p.act, [&](auto A) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = a_row0 + i * 16 + fr;
          const int col = b_row0 + j * 16 + fq * 4;
          float y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = act_c16<decltype(A)::value>(acc[i][j][r] * sc[j][r] + sh[j][r]);
          uint2 pk;
          pk.x = pack2<H>(y[0], y[1]);
          pk.y = pack2<H>(y[2], y[3]);
          *reinterpret_cast<uint2*>(sC + row * CP + col) = pk;
        }
    });
    __syncthreads();
    if (p.mom != nullptr) tile_moments<H, BN, 256 * KG>(sC, CP, min(BM, m_end - m0), n0, p.Cout, p.mom + (long long)tile_m * 2 * p.Cout, tid);
    constexpr int CH = BN / 8;   // 16-byte chunks per tile row (a power of two <= 16)
    H* __restrict__ out = reinterpret_cast<H*>(p.out);
    const H* __restrict__ res = reinterpret_cast<const H*>(p.res);
    const bool ln = p.ln_gamma != nullptr;   // host guarantees Cout == BN
    float g[8], bt[8];
    if (ln) {
      const int ch = tid % CH;
      Vec<float>::load(p.ln_gamma + ch * 8, g); Vec<float>::load(p.ln_gamma + ch * 8 + 4, g + 4);
      Vec<float>::load(p.ln_beta + ch * 8, bt); Vec<float>::load(p.ln_beta + ch * 8 + 4, bt + 4);
    }
    for (int q = tid; q < BM * CH; q += 256 * KG) {
      const int row = q / CH, ch = q % CH;
      const int m = m0 + row, n = n0 + ch * 8;
      const bool live = m < m_end && n < p.Cout;
      float a[8];
      Vec<H>::load(sC + row * CP + ch * 8, a);
      if (ln) {   // all CH lanes of a row are in one wave and take this branch together
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e];
#pragma unroll
        for (int o = CH >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.f / (float)BN);
        float qq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = a[e] - mean; qq = fmaf(d, d, qq); }
#pragma unroll
        for (int o = CH >> 1; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
        const float rstd = rsqrtf(qq * (1.f / (float)BN) + p.ln_eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (a[e] - mean) * rstd * g[e] + bt[e];
      }
      if (!live) continue;
      if (res) {
        float b[8];
        Vec<H>::load(res + (long long)m * p.res_ld + n, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
      }
      if (ln || res) {
        Vec<H>::store(out + (long long)m * p.out_ld + n, a);
      } else {
        *reinterpret_cast<u32x4*>(out + (long long)m * p.out_ld + n) = *reinterpret_cast<const u32x4*>(sC + row * CP + ch * 8);
      }
    }
  }
}

struct Cfg { int bm, bn, stages, kg = 1; };
// variant ids (cfp_conv2d_variant2): keep in sync with launch_variant below
constexpr Cfg kCfg[] = {
    {128, 128, 3},  // 0
    {128, 128, 2},  // 1
    {128, 64, 3},   // 2
    {128, 64, 4},   // 3
    {64, 64, 3},    // 4
    {64, 64, 4},    // 5
    {256, 32, 3},   // 6
    {256, 32, 2},   // 7
    {128, 32, 3},   // 8
    {128, 32, 4},   // 9
    {256, 16, 2},   // 10
    {128, 16, 4},   // 11
    {64, 128, 3},   // 12
    {64, 64, 2},    // 13
    {128, 64, 2},   // 14
    {64, 128, 2},   // 15
    {128, 32, 2},   // 16
    {32, 64, 3},    // 17: few-row / long-K problems with per-image weights (the squeeze-excite project GEMMs at 1/32: 300 rows per image):
    {32, 128, 3},   // 18  twice the workgroups of the 64-row tiles, so that 8 images x 232 channels fill the chip
    {64, 64, 2, 2},   // 19: two K groups (eight waves): long-K launches of a few hundred tiles
    {64, 64, 3, 2},   // 20
    {64, 128, 2, 2},  // 21
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

template <typename H, int BM, int BN, int WM, int WN, int STAGES, int KG = 1>
int launch2(const ConvP& p, float* slabs, int splits, hipStream_t s) {
  const size_t lds = (size_t)KG * STAGES * (BM + BN) * 128;
  if (lds > 160 * 1024) return -1;
  if (KG == 2 && splits > 1) return -4;
  long long tiles_m = p.rows_per_batch > 0 ? (long long)p.B * cdiv(p.rows_per_batch, BM) : cdiv(p.M, BM);
  long long tiles = tiles_m * cdiv(p.Cout, BN);
  if (splits <= 1) {
    auto k = igemm2_kernel<H, BM, BN, WM, WN, STAGES, false, KG>;
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256 * KG), lds, s, p, slabs, 1);
  } else if constexpr (KG == 1) {
    auto k = igemm2_kernel<H, BM, BN, WM, WN, STAGES, true>;
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)(tiles * splits)), dim3(256), lds, s, p, slabs, splits);
  }
  return 0;
}

}  // namespace

int igemm2_num_variants() { return kNumCfg; }
void igemm2_variant_shape(int v, int* bm, int* bn, int* stages) { *bm = kCfg[v].bm; *bn = kCfg[v].bn; *stages = kCfg[v].stages; }

// Launch variant v.  Returns 0, or a negative value if the variant cannot run this problem.
#define L2(...) (p.f16 ? launch2<f16_t, __VA_ARGS__>(p, slabs, splits, s) : launch2<bf16_t, __VA_ARGS__>(p, slabs, splits, s))
int igemm2_launch(int v, const ConvP& p, float* slabs, int splits, hipStream_t s) {
  switch (v) {
    case 0: return L2(128, 128, 2, 2, 3);
    case 1: return L2(128, 128, 2, 2, 2);
    case 2: return L2(128, 64, 2, 2, 3);
    case 3: return L2(128, 64, 2, 2, 4);
    case 4: return L2(64, 64, 2, 2, 3);
    case 5: return L2(64, 64, 2, 2, 4);
    case 6: return L2(256, 32, 4, 1, 3);
    case 7: return L2(256, 32, 4, 1, 2);
    case 8: return L2(128, 32, 4, 1, 3);
    case 9: return L2(128, 32, 4, 1, 4);
    case 10: return L2(256, 16, 4, 1, 2);
    case 11: return L2(128, 16, 4, 1, 4);
    case 12: return L2(64, 128, 2, 2, 3);
    case 13: return L2(64, 64, 2, 2, 2);
    case 14: return L2(128, 64, 2, 2, 2);
    case 15: return L2(64, 128, 2, 2, 2);
    case 16: return L2(128, 32, 4, 1, 2);
    case 17: return L2(32, 64, 1, 4, 3);
    case 18: return L2(32, 128, 1, 4, 3);
    case 19: return L2(64, 64, 2, 2, 2, 2);
    case 20: return L2(64, 64, 2, 2, 3, 2);
    case 21: return L2(64, 128, 2, 2, 2, 2);
    default: return -3;
  }
}
