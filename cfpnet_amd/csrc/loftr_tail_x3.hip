// Fused tail of a LoFTR encoder layer (transformer.py:45-71, attention.py:48-49) in the DEFAULT numerics of the drop-in boundary: float32
// tensors, every GEMM of the chain as A_hi W_hi + A_hi W_lo + A_lo W_hi on v_mfma_f32_16x16x32_f16 (conv_igemm_x3.hip has the arithmetic).
//
//   q   = x @ Wq^T                                        q projection of this wave's own rows (transformer.py:45)
//   msg = (elu1(q) KV[g]) / (elu1(q) . Ksum[g] + eps) * S linear-attention apply, per head (float32 VALU)
//   y1  = LayerNorm1(msg @ Wm^T)                          merge + norm1
//   h   = relu([x | y1] @ W0^T)                           mlp.0
//   out = LayerNorm2(h @ W2^T) + x                        mlp.2 + norm2 + residual
//
// Unfused (the float32 mode's path, and this mode's until round 4) this is six launches per layer -- q GEMM, apply, merge GEMM + LayerNorm,
// mlp.0 GEMM, mlp.2 GEMM + LayerNorm -- 18 layers per forward: 72 of the 154 small f16x3 GEMM launches and the 24 attention-apply launches,
// 1.5 ms of the 5.1 ms step with four batches in flight (tools/ablate_time.py --x3).  Structure = loftr_tail.hip's (the 16-bit kernel): a wave
// owns 16 token rows from the projection to the final store, its tiles live in a private LDS region, only the weights are shared (LDS-DMA,
// 128-byte swizzled rows [hi(32) | lo(32)] of cfp_pack_w_x3's operand, double buffered, one barrier per 32-channel K-step).  Differences:
//   * the tiles are FLOAT32 (row pitch D + 8 floats = 2 (mod 4) sixteen-byte slots: the A-fragment reads of 16 rows are conflict-free);
//     a lane reads its two channel quads (4 fq .., 16 + 4 fq ..) of the K-step and splits them in registers -- one fragment per K-step
//     and wave (16 rows), ~20 VALU instructions beside 3 x NT MFMAs;
//   * mlp.0 / mlp.2 run in two halves of the hidden width: half of h (D channels) is produced, then consumed as a K range of mlp.2 into
//     accumulators that stay in registers -- the hidden tile is D wide instead of 2 D, so that four waves' tiles and the weight stages
//     fit the LDS at D = 128 (152 KB with the three weight stages);
//   * nothing is rounded on the way: msg, y1, h stay float32 (the unfused float32 path stores them in float32 as well).
#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16t[4] = {0u, 0u, 0u, 0u};
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

struct TailX3P {
  const float* q; const float* kv; const float* ksum; const float* x; float* out;
  const f16_t* wq;               // optional: q_proj weights (pre-split); the kernel then projects q for its own rows and `q` is unused
  const f16_t* wm; const f16_t* w0; const f16_t* w2;      // pre-split operands of cfp_pack_w_x3: [D][D], [2D][2D], [D][2D]
  const float* g1; const float* b1; const float* g2; const float* b2;
  int q_ld, x_ld, out_ld;
  int rows, Hq, Wq, qth, qtw, ggy, ggx;
  FastDiv fwq, fhq, fqth, fqtw;
  float v_length, eps, ln_eps;
};

// One GEMM of the chain for this wave's 16 rows: acc[j] += A[16 x 32 nks] * W[N x ..]^T over the K-steps [ks0, ks0 + nks) of the weight rows
// (N = NT * 16 rows starting at W, `wrow` halves per row).  `arow(ks)` returns the wave-private float32 row pointer (lane's row fr) of the
// A operand's 32 channels of K-step ks.  All four waves of the workgroup must call it together (they share the weight stages).
// Weight stages per kernel: three (two K-steps of DMA in flight) unless one fewer lets ANOTHER WORKGROUP share the CU -- these kernels run
// their phases (projection, attention apply, GEMM chain, LayerNorms) behind barriers, so a co-resident workgroup is what fills the gaps (the
// lesson of DESIGN.md 4.5): the LoFTR tail at D = 32 is 43 KB with three stages (three per CU) and 39 KB with two (four per CU; its GEMMs
// are one or two K-steps long anyway); the LKPM tail at D = 64 is 61 KB -> 53 KB (two -> three per CU).
template <int D> constexpr int tail_nst_loftr() { return D == 32 ? 2 : 3; }
template <int D> constexpr int tail_nst_lkpm() { return D == 64 ? 2 : 3; }
template <int NT, int BSTAGE, bool ZERO, int NST, int WAVES, typename AF>
__device__ __forceinline__ void tail_gemm_x3(f32x4 (&acc)[NT], const f16_t* __restrict__ W, int wrow, int ks0, int nks, AF arow, unsigned char* sB,
                                             int wave, int lane) {
  constexpr int N = NT * 16;
  constexpr int NBG = N / 8;                  // 8-row DMA groups of a weight stage
  constexpr int NBW = (NBG + WAVES - 1) / WAVES;      // LDS-DMA instructions per wave and stage
  static_assert(N * 128 <= BSTAGE, "weight stage");
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;
  auto issue = [&](int ks, int st) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int g = (j * WAVES + wave) % NBG;
      const int n = g * 8 + rsub;
      glds16(W + (long long)n * wrow + (ks * 8 + lc) * 8, sB + st * BSTAGE + g * 1024);      // rows are zero-padded to whole K-steps
    }
  };
  if (ZERO) {
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Hand-over of the shared weight stages.  `s_barrier` has no memory semantics for the compiler and LDS reads are asynchronous: without the
  // wait + clobber IN FRONT of the barrier the previous GEMM's last fragment reads (one K-step GEMMs at D = 32 are straight-line code once
  // inlined) may be scheduled -- or still be in flight -- behind it, while a faster wave already streams the next weights into the stage they
  // read.  Found as a timing-dependent mismatch of a few 16-row tiles at D = 32 (tools/probes/x3_tail_stability.py: 109 of 76 800 rows in one
  // of 20 runs); D = 64 / 128 never showed it.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();               // every wave is done with the previous GEMM's stages
  asm volatile("" ::: "memory");
  // THREE weight stages, two K-steps of DMA in flight (round 4, late): a K-step here is 3 NT MFMAs (~0.1-0.2 us) against a ~0.7-1 us round
  // trip of its weight tile from L2 -- with two stages every step of the chain (32 of them at D = 128) waited out that round trip
  // (batch-1 forward 3.50 -> 3.41 ms).  Counted wait: the loads of a stage are this wave's NBW youngest vector-memory operations when the
  // next stage has been issued behind it, and they complete in order; the last step waits for everything.  Issuing the first stages of the
  // NEXT GEMM of the chain right after a K loop (under the LayerNorm / attention / GELU work between the GEMMs) was built too and measured
  // no gain (3.42 ms; it costs a barrier per GEMM) -- not kept.
  constexpr int AHEAD = NST - 1;              // K-steps of DMA in flight beside the one being computed
  static_assert(NST == 2 || NST == 3, "wait ladder below");
  issue(ks0, 0);
  if (AHEAD > 1 && nks > 1) issue(ks0 + 1, 1);
  const int pc0 = ((fq) ^ (fr & 7)) * 16, pc1 = ((4 + fq) ^ (fr & 7)) * 16;
  for (int i = 0; i < nks; ++i) {
    if (AHEAD > 1 && i + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW) : "memory");      // (the loads of step i + 1 may still fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (i + AHEAD < nks) issue(ks0 + i + AHEAD, (i + AHEAD) % NST);      // the stage of step i - 1: everybody has read it (barrier above)
    const unsigned char* cB = sB + (i % NST) * BSTAGE;
    const float* ar = arow(ks0 + i);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(ar + 4 * fq);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(ar + 16 + 4 * fq);
    f16x8 ahi, alo;
    split8(x0, x1, ahi, alo);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f16x8 bh = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc0);
      const f16x8 bl = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc1);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh, acc[j], 0, 0, 0);      // acc[r] = row fq * 4 + r, column j * 16 + fr
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh, acc[j], 0, 0, 0);
    }
  }
}

// LayerNorm over the N = NT * 16 columns of each of this lane's 4 rows (row = fq * 4 + r, col = j * 16 + fr): two-pass statistics over the
// 16 lanes of a DPP row, as cfp_layernorm.
template <int NT>
__device__ __forceinline__ void tail_layernorm_x3(f32x4 (&acc)[NT], const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int fr) {
  constexpr float inv_n = 1.f / (float)(NT * 16);
  float g[NT], bt[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { g[j] = gamma[j * 16 + fr]; bt[j] = beta[j * 16 + fr]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) s += acc[j][r];
    s = row16_sum(s);
    const float mean = s * inv_n;
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { const float dlt = acc[j][r] - mean; qq = fmaf(dlt, dlt, qq); }
    qq = row16_sum(qq);
    const float rstd = rsqrtf(qq * inv_n + eps);
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j][r] = (acc[j][r] - mean) * rstd * g[j] + bt[j];
  }
}

// WAVES (1, 2 or 4) waves of 16 token rows per workgroup.  Every wave reads the WHOLE weight tile of a K-step from LDS (16 KB at D = 128), so four
// waves share one LDS pipe for four times the bytes: the GEMM phases of the four-wave kernel are LDS-read bound (64 KB per step at 128 B / clk
// against 24 MFMAs per wave).  Few token rows (a single image: 19 four-wave workgroups at D = 128 on 256 CUs) therefore run as MORE, NARROWER
// workgroups -- each streams all the weights from L2 itself, which is cheap while the chip is mostly idle -- and many rows keep four waves
// (the weight stream is then shared by 64 rows).  Same arithmetic per row in every layout.
template <int D, int HEADS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void loftr_tail_x3_kernel(TailX3P p) {
  constexpr int d = D / HEADS;
  constexpr int NT = D / 16;
  constexpr int PA = D + 8;                                // row pitch in floats: (D + 8) / 4 = 2 (mod 4) sixteen-byte slots
  constexpr int TILE = 16 * PA;                            // floats per tile
  constexpr int WAVE_LDS = 3 * TILE * 4;                   // msg / y1 | x | q / h-half tiles of one wave
  constexpr int BSTAGE = D * 128;                          // weight stage: [D rows][hi(32) | lo(32)]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int NST = tail_nst_loftr<D>();
  unsigned char* sB = smem;
  float* tMsg = reinterpret_cast<float*>(smem + NST * BSTAGE + wave * WAVE_LDS);
  float* tX = tMsg + TILE;
  float* tH = tX + TILE;
  const long long row0 = (long long)blockIdx.x * (16 * WAVES) + wave * 16;
  constexpr int wrow1 = (D / 32) * 64, wrow2 = (2 * D / 32) * 64;      // halves per packed weight row for K = D and K = 2 D

  // ---- x tile -> LDS (16-byte vectors) ------------------------------------------------------------------------------------------
  constexpr int XCH = D / 4;
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < p.rows) v = *reinterpret_cast<const f32x4*>(p.x + m * p.x_ld + ch * 4);
    *reinterpret_cast<f32x4*>(tX + r * PA + ch * 4) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- optional q projection for this wave's rows -----------------------------------------------------------------------------------
  const bool own_q = p.wq != nullptr;
  if (own_q) {
    f32x4 acc[NT];
    tail_gemm_x3<NT, BSTAGE, true, NST, WAVES>(acc, p.wq, wrow1, 0, D / 32, [&](int ks) { return tX + fr * PA + ks * 32; }, sB, wave, lane);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tH[(fq * 4 + r) * PA + j * 16 + fr] = acc[j][r];      // the hidden tile is free until mlp.0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- linear-attention apply: lane = (row, head slot) ------------------------------------------------------------------------------
  {
    const int r = fr;
    const long long m = row0 + r;
    const bool ok = m < p.rows;
    const long long mm = ok ? m : 0;
    const unsigned t = fd_div((unsigned)mm, p.fwq), xq = (unsigned)mm - t * (unsigned)p.Wq;
    const unsigned b = fd_div(t, p.fhq), yq = t - b * (unsigned)p.Hq;
    const long long g = ((long long)b * p.ggy + fd_div(yq, p.fqth)) * p.ggx + fd_div(xq, p.fqtw);
#pragma unroll
    for (int hs = 0; hs < HEADS / 4; ++hs) {
      const int h = fq + 4 * hs;
      const float* __restrict__ kv = p.kv + (g * HEADS + h) * d * d;
      const float* __restrict__ ks = p.ksum + (g * HEADS + h) * d;
      const float* qp = own_q ? tH + r * PA + h * d : p.q + mm * p.q_ld + h * d;
      float qv[d];
#pragma unroll
      for (int c = 0; c < d; c += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(qp + c);
        qv[c] = t4[0]; qv[c + 1] = t4[1]; qv[c + 2] = t4[2]; qv[c + 3] = t4[3];
      }
      float o[d];
#pragma unroll
      for (int j = 0; j < d; ++j) o[j] = 0.f;
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < d; ++i) {
        const float qe = elu1(qv[i]);
        z = fmaf(qe, ks[i], z);
#pragma unroll
        for (int j = 0; j < d; j += 4) {
          const f32x4 kk = *reinterpret_cast<const f32x4*>(kv + i * d + j);
          o[j] = fmaf(qe, kk[0], o[j]); o[j + 1] = fmaf(qe, kk[1], o[j + 1]);
          o[j + 2] = fmaf(qe, kk[2], o[j + 2]); o[j + 3] = fmaf(qe, kk[3], o[j + 3]);
        }
      }
      const float zi = 1.f / (z + p.eps);                    // (o * 1/(z+eps)) * S, as attention.py:48-49
#pragma unroll
      for (int j = 0; j < d; ++j) tMsg[r * PA + h * d + j] = ok ? o[j] * zi * p.v_length : 0.f;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- merge + norm1 -----------------------------------------------------------------------------------------------------------------
  {
    f32x4 acc[NT];
    tail_gemm_x3<NT, BSTAGE, true, NST, WAVES>(acc, p.wm, wrow1, 0, D / 32, [&](int ks) { return tMsg + fr * PA + ks * 32; }, sB, wave, lane);
    tail_layernorm_x3<NT>(acc, p.g1, p.b1, p.ln_eps, fr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of msg are complete
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tMsg[(fq * 4 + r) * PA + j * 16 + fr] = acc[j][r];      // y1 replaces msg
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- mlp.0 ([x | y1], K = 2 D -> 2 D, ReLU) and mlp.2 (K = 2 D -> D) in two halves of the hidden width -------------------------------
  f32x4 acc2[NT];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 acc[NT];
    tail_gemm_x3<NT, BSTAGE, true, NST, WAVES>(acc, p.w0 + (long long)half * D * wrow2, wrow2, 0, 2 * D / 32,
                                   [&](int ks) { return ks * 32 < D ? tX + fr * PA + ks * 32 : tMsg + fr * PA + (ks * 32 - D); }, sB, wave, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (second half) this wave's mlp.2 reads of the previous half are complete
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tH[(fq * 4 + r) * PA + j * 16 + fr] = fmaxf(acc[j][r], 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (half == 0)
      tail_gemm_x3<NT, BSTAGE, true, NST, WAVES>(acc2, p.w2, wrow2, 0, D / 32, [&](int ks) { return tH + fr * PA + ks * 32; }, sB, wave, lane);
    else
      tail_gemm_x3<NT, BSTAGE, false, NST, WAVES>(acc2, p.w2, wrow2, D / 32, D / 32, [&](int ks) { return tH + fr * PA + (ks * 32 - D); }, sB, wave, lane);
  }
  tail_layernorm_x3<NT>(acc2, p.g2, p.b2, p.ln_eps, fr);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = fq * 4 + r, col = j * 16 + fr;
      tMsg[row * PA + col] = acc2[j][r] + tX[row * PA + col];      // stage the output tile
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    if (m < p.rows) *reinterpret_cast<f32x4*>(p.out + m * p.out_ld + ch * 4) = *reinterpret_cast<const f32x4*>(tMsg + r * PA + ch * 4);
  }
}

int g_tail_waves = 0;      // cfp_debug_set key 35: 0 = by the row count, else 1 / 2 / 4 waves per workgroup (A/B)
// waves per workgroup for `rows` token rows: few rows -> narrow workgroups (see the kernel)
inline int tail_waves(long long rows) {
  if (g_tail_waves == 1 || g_tail_waves == 2 || g_tail_waves == 4) return g_tail_waves;
  return rows <= 4800 ? 1 : rows < 8192 ? 2 : 4;      // (a batch of 8 at D = 128, 9 600 rows, measured slower with two waves: 6.50 vs 6.32 ms per forward)
}

template <int D, int HEADS, int WAVES>
int launch_tail_x3w(const TailX3P& p, hipStream_t s) {
  constexpr size_t lds = tail_nst_loftr<D>() * (D * 128) + WAVES * (3 * 16 * (D + 8) * 4);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = loftr_tail_x3_kernel<D, HEADS, WAVES>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)cdiv(p.rows, 16 * WAVES)), dim3(64 * WAVES), lds, s, p);
  return 0;
}
template <int D, int HEADS>
int launch_tail_x3(const TailX3P& p, hipStream_t s) {
  const int w = tail_waves(p.rows);
  return w == 1 ? launch_tail_x3w<D, HEADS, 1>(p, s) : w == 2 ? launch_tail_x3w<D, HEADS, 2>(p, s) : launch_tail_x3w<D, HEADS, 4>(p, s);
}

// ---- LKPM tail (Block14.forward after the depthwise conv, convnext.py:48-58): LayerNorm(1e-6) -> pwconv1 (D -> 4 D) -> GELU -> pwconv2
// (4 D -> D) -> + input, for the 16 token rows of a wave, in the default numerics.  The hidden width runs in FOUR quarters of D channels:
// a quarter of h is produced (bias + exact erf GELU), then consumed as a K range of pwconv2 into accumulators that stay in registers.
struct LkpmX3P {
  const float* t; const float* xin; float* out;
  const f16_t* w1; const f16_t* w2;             // pre-split operands of cfp_pack_w_x3: [4 D][D], [D][4 D]
  const float* lg; const float* lb; const float* b1; const float* b2;
  int t_ld, x_ld, out_ld, rows;
  float ln_eps;
};

template <int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lkpm_tail_x3_kernel(LkpmX3P p) {
  constexpr int NT = D / 16;
  constexpr int PA = D + 8;
  constexpr int TILE = 16 * PA;
  constexpr int WAVE_LDS = 2 * TILE * 4;                   // normalised input / output tile | hidden quarter
  constexpr int BSTAGE = D * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  unsigned char* sB = smem;
  constexpr int NST = tail_nst_lkpm<D>();
  float* tA = reinterpret_cast<float*>(smem + NST * BSTAGE + wave * WAVE_LDS);
  float* tH = tA + TILE;
  const long long row0 = (long long)blockIdx.x * (16 * WAVES) + wave * 16;
  constexpr int XCH = D / 4;
  constexpr int wrow1 = (D / 32) * 64, wrow2 = (4 * D / 32) * 64;

  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < p.rows) v = *reinterpret_cast<const f32x4*>(p.t + m * p.t_ld + ch * 4);
    *reinterpret_cast<f32x4*>(tA + r * PA + ch * 4) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {      // LayerNorm over the D channels of each row: lane = (row fr, quarter fq of the channels)
    constexpr int Q = D / 4;
    float v[Q];
#pragma unroll
    for (int c = 0; c < Q; c += 4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(tA + fr * PA + fq * Q + c);
      v[c] = t4[0]; v[c + 1] = t4[1]; v[c + 2] = t4[2]; v[c + 3] = t4[3];
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < Q; ++c) s += v[c];
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / (float)D);
    float qq = 0.f;
#pragma unroll
    for (int c = 0; c < Q; ++c) { const float dl = v[c] - mean; qq = fmaf(dl, dl, qq); }
    qq += __shfl_xor(qq, 16, 64); qq += __shfl_xor(qq, 32, 64);
    const float rstd = rsqrtf(qq * (1.f / (float)D) + p.ln_eps);
#pragma unroll
    for (int c = 0; c < Q; c += 4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[c + e] - mean) * rstd * p.lg[fq * Q + c + e] + p.lb[fq * Q + c + e];
      *reinterpret_cast<f32x4*>(tA + fr * PA + fq * Q + c) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  f32x4 acc2[NT];
#pragma unroll
  for (int part = 0; part < 4; ++part) {
    f32x4 acc[NT];
    tail_gemm_x3<NT, BSTAGE, true, NST, WAVES>(acc, p.w1 + (long long)part * D * wrow1, wrow1, 0, D / 32, [&](int ks) { return tA + fr * PA + ks * 32; }, sB, wave, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's pwconv2 reads of the previous quarter are complete
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float bj = p.b1[part * D + j * 16 + fr];
#pragma unroll
      for (int r = 0; r < 4; ++r) tH[(fq * 4 + r) * PA + j * 16 + fr] = act_c<CFP_ACT_GELU>(acc[j][r] + bj);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (part == 0)
      tail_gemm_x3<NT, BSTAGE, true, NST, WAVES>(acc2, p.w2, wrow2, 0, D / 32, [&](int ks) { return tH + fr * PA + ks * 32; }, sB, wave, lane);
    else
      tail_gemm_x3<NT, BSTAGE, false, NST, WAVES>(acc2, p.w2, wrow2, part * (D / 32), D / 32, [&](int ks) { return tH + fr * PA + (ks * 32 - part * D); }, sB, wave, lane);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float bj = p.b2[j * 16 + fr];
#pragma unroll
    for (int r = 0; r < 4; ++r) tA[(fq * 4 + r) * PA + j * 16 + fr] = acc2[j][r] + bj;      // the normalised input is consumed: stage the output tile
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    if (m < p.rows) {
      f32x4 a = *reinterpret_cast<const f32x4*>(tA + r * PA + ch * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.xin + m * p.x_ld + ch * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += b[e];
      *reinterpret_cast<f32x4*>(p.out + m * p.out_ld + ch * 4) = a;
    }
  }
}

template <int D, int WAVES>
int launch_lkpm_x3w(const LkpmX3P& p, hipStream_t s) {
  constexpr size_t lds = tail_nst_lkpm<D>() * (D * 128) + WAVES * (2 * 16 * (D + 8) * 4);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = lkpm_tail_x3_kernel<D, WAVES>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)cdiv(p.rows, 16 * WAVES)), dim3(64 * WAVES), lds, s, p);
  return 0;
}
template <int D>
int launch_lkpm_x3(const LkpmX3P& p, hipStream_t s) {
  const int w = tail_waves(p.rows);
  return w == 1 ? launch_lkpm_x3w<D, 1>(p, s) : w == 2 ? launch_lkpm_x3w<D, 2>(p, s) : launch_lkpm_x3w<D, 4>(p, s);
}

}  // namespace

void cfp_tail_x3_debug_set(int value) { g_tail_waves = value; }

// cfp_lkpm_tail for dtype CFP_F32X3 (loftr_tail.hip dispatches here)
int lkpm_tail_x3_launch(const void* t, int t_ld, const void* xin, int x_ld, void* out, int out_ld, const void* w1, const float* b1, const void* w2,
                        const float* b2, const float* ln_g, const float* ln_b, float ln_eps, int rows, int D, hipStream_t s) {
  LkpmX3P p;
  p.t = (const float*)t; p.xin = (const float*)xin; p.out = (float*)out; p.w1 = (const f16_t*)w1; p.w2 = (const f16_t*)w2;
  p.lg = ln_g; p.lb = ln_b; p.b1 = b1; p.b2 = b2; p.t_ld = t_ld; p.x_ld = x_ld; p.out_ld = out_ld; p.rows = rows; p.ln_eps = ln_eps;
  if (D == 32) return launch_lkpm_x3<32>(p, s);
  if (D == 64) return launch_lkpm_x3<64>(p, s);
  if (D == 128) return launch_lkpm_x3<128>(p, s);
  return -2;
}

// cfp_loftr_tail for dtype CFP_F32X3 (loftr_tail.hip dispatches here): float32 q / x / out, weights = cfp_pack_w_x3 operands.
int loftr_tail_x3_launch(const void* q, int q_ld, const float* kv, const float* ksum, const void* x, int x_ld, void* out, int out_ld,
                         const void* w_q, const void* w_merge, const void* w_mlp0, const void* w_mlp2, const float* ln1_g, const float* ln1_b,
                         const float* ln2_g, const float* ln2_b, float ln_eps, int NB, int Hq, int Wq, int qth, int qtw, float v_length,
                         float eps, int heads, int D, hipStream_t s) {
  TailX3P p;
  p.q = (const float*)q; p.kv = kv; p.ksum = ksum; p.x = (const float*)x; p.out = (float*)out;
  p.wq = (const f16_t*)w_q; p.wm = (const f16_t*)w_merge; p.w0 = (const f16_t*)w_mlp0; p.w2 = (const f16_t*)w_mlp2;
  p.g1 = ln1_g; p.b1 = ln1_b; p.g2 = ln2_g; p.b2 = ln2_b;
  p.q_ld = q_ld; p.x_ld = x_ld; p.out_ld = out_ld;
  p.rows = NB * Hq * Wq; p.Hq = Hq; p.Wq = Wq; p.qth = qth; p.qtw = qtw; p.ggy = cdiv(Hq, qth); p.ggx = cdiv(Wq, qtw);
  p.fwq = make_fastdiv((unsigned)Wq); p.fhq = make_fastdiv((unsigned)Hq); p.fqth = make_fastdiv((unsigned)qth); p.fqtw = make_fastdiv((unsigned)qtw);
  p.v_length = v_length; p.eps = eps; p.ln_eps = ln_eps;
  if (D == 32 && heads == 4) return launch_tail_x3<32, 4>(p, s);
  if (D == 32 && heads == 8) return launch_tail_x3<32, 8>(p, s);
  if (D == 64 && heads == 4) return launch_tail_x3<64, 4>(p, s);
  if (D == 64 && heads == 8) return launch_tail_x3<64, 8>(p, s);
  if (D == 128 && heads == 4) return launch_tail_x3<128, 4>(p, s);
  if (D == 128 && heads == 8) return launch_tail_x3<128, 8>(p, s);
  return -2;
}
