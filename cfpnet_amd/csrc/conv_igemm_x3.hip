// Implicit-GEMM convolution / linear layer for FLOAT32 STORAGE with split-precision matrix math ("f16x3"):
//
//   x = x_hi + x_lo   (x_hi = the value rounded toward zero to IEEE half, x_lo = half(x - x_hi): 21-22 significant bits together)
//   A . W  ~=  A_hi . W_hi  +  A_hi . W_lo  +  A_lo . W_hi          three v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block, f32 accumulate
//
// i.e. 3/16 of the bf16 / fp16 matrix rate instead of the 1/16 of v_mfma_f32_16x16x4_f32 that the float32 parity kernels (conv_igemm.hip)
// run at, with a product error of ~2^-21 (the dropped lo x lo term and the two representation errors) -- three orders of magnitude
// inside the north-star tolerance (1e-3 relative L1 on the depth map) on every weight family, where one-term fp16 / bf16 storage is
// at 0.8e-3 ... 1.2e-2 depending on the conditioning of the network (tests/test_forward_gpu.py).
//
// Layout and pipeline are those of the second-generation 16-bit kernel (conv_igemm2.hip): both operand tiles travel global -> LDS with
// `global_load_lds_dwordx4`, 128-byte LDS rows, XOR chunk swizzle applied on the DMA source side, STAGES K-steps in flight, one raw
// barrier per K-step.  What differs:
//   * a K-step is 32 input channels.  An A row in LDS is the 32 FLOAT32 values as they sit in the NHWC tensor (eight 16-byte chunks);
//     lane (fr, fq) reads chunks fq and fq + 4 (the two reads of the 16-bit kernel, so the same banks) = channels 4 fq .. 4 fq + 3 and
//     16 + 4 fq .. 16 + 4 fq + 3, and splits its eight values into the hi / lo operands in registers: 4 x v_cvt_pkrtz_f16_f32,
//     8 x v_cvt_f32_f16 + 4 x v_pk_add_f32 (x - hi, exact), 4 x v_cvt_pk_f16_f32 per fragment (~20 vector instructions beside the
//     fragment's 3 TN MFMAs).  Tiles with WN = 1 convert every A element exactly once.
//   * the weights are PRE-SPLIT by cfp_pack_w_x3 (or by the squeeze-excite fold kernels, per image): row n holds, per K-step, 64 halves
//     [hi(32) | lo(32)] in the lane order above (position 8 fq + e <-> channel 4 fq + e for e < 4, 16 + 4 fq + e - 4 for e >= 4), zero
//     padded to whole K-steps -- the B fragments are plain 16-byte LDS reads.
//   * the epilogue stores float32 straight from the accumulators (a lane owns 4 consecutive output channels of one pixel: 16 bytes).
#include "igemm_core.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16_x3[4] = {0u, 0u, 0u, 0u};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The K loop: accumulates C[m0.., n0..] over K-steps [k0, k1) into acc.  smem: KG * STAGES * (BM + BN) * 128 bytes.  All 256 * KG threads call it.
// INTERLEAVE: fragment i of wave-row wm covers tile rows (i * WM + wm) * 16 .. + 15 instead of (wm * TM + i) * 16 .. (the fused bin head wants
// the 64 rows of one i contiguous).
template <int BM, int BN, int WM, int WN, int STAGES, int KG, bool INTERLEAVE = false>
__device__ __forceinline__ void x3_mainloop(const ConvP& p, const f16_t* __restrict__ wt, int m0, int m_end, int n0, int k0, int k1,
                                            unsigned char* smem, f32x4 (&acc)[BM / WM / 16][BN / WN / 16]) {
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int NA = BM / 32;                    // A DMA instructions per wave per K-step
  constexpr int NBG = BN / 8;                    // 8-row groups of the W tile
  constexpr int NB = (NBG + 3) / 4;              // W DMA instructions per wave per K-step
  constexpr int LPS = NA + NB;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert((STAGES - 2) * LPS <= 63, "vmcnt field");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 2 ? wave8 >> 2 : 0;
  const int wave = wave8 & 3;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;              // logical 16-byte chunk this lane fetches: 4 float32 channels of A, 8 halves of W
  unsigned char* gsm = smem + kg * (STAGES * STAGE_BYTES);
  const int nk_all = (p.K + 31) >> 5;
  const int wrow = nk_all * 64;                  // halves per packed weight row
  const float* __restrict__ in = reinterpret_cast<const float*>(p.in);

  const float* a_ptr[NA];
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
  const f16_t* b_ptr[NB];
  unsigned b_okmask = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + ((j * 4 + wave) % NBG) * 8 + rsub;
    const bool ok = n < p.Cout;
    if (ok) b_okmask |= 1u << j;
    b_ptr[j] = wt + (long long)(ok ? n : 0) * wrow;
  }
  const void* zsrc = reinterpret_cast<const void*>(g_zero16_x3);
  // im2col position of this lane's chunk in the next K-step to be issued, advanced by 32 channels per K-step without divisions
  int i_cc = 0, i_kh = 0, i_kw = 0;
  if (!p.pointwise) {
    const int kk = ((k0 + kg) * 8 + lc) * 4;
    const int tap = kk / p.Cin;
    i_cc = kk - tap * p.Cin;
    i_kh = tap / p.KW;
    i_kw = tap - i_kh * p.KW;
  }

  auto issue = [&](int ks, int buf) {
    unsigned char* sA = gsm + buf * STAGE_BYTES;
    unsigned char* sB = sA + BM * 128;
    const int kk = (ks * 8 + lc) * 4;            // first of this lane's four channels in the im2col K axis
    const bool kok = kk < p.K;
    if (p.pointwise) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = kok && ((a_okmask >> i) & 1u);
        glds16(ok ? (const void*)(a_ptr[i] + kk) : zsrc, sA + (i * 4 + wave) * 1024);
      }
    } else {
      const int off = (i_kh * p.W + i_kw) * p.in_ld + i_cc;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = kok && ((a_okmask >> i) & 1u) && (unsigned)(a_hi0[i] + i_kh) < (unsigned)p.H &&
                        (unsigned)(a_wi0[i] + i_kw) < (unsigned)p.W;
        glds16(ok ? (const void*)(a_ptr[i] + off) : zsrc, sA + (i * 4 + wave) * 1024);
      }
      i_cc += 32 * KG;
      while (i_cc >= p.Cin) {
        i_cc -= p.Cin;
        if (++i_kw == p.KW) { i_kw = 0; ++i_kh; }
      }
    }
    const int kw16 = (ks * 8 + lc) * 8;          // weight rows are zero-padded to whole K-steps: no K-tail test
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = (b_okmask >> j) & 1u;
      glds16(ok ? (const void*)(b_ptr[j] + kw16) : zsrc, sB + ((j * 4 + wave) % NBG) * 1024);
    }
  };

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkg = max(0, (k1 - k0 - kg + KG - 1) / KG);      // this group's K-steps
  const int nit = (k1 - k0 + KG - 1) / KG;                     // barrier rounds
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nkg) issue(k0 + kg + s * KG, s);

  const int b_row0 = wn * (BN / WN);
  const int pc0 = ((fq) ^ (fr & 7)) * 16, pc1 = ((4 + fq) ^ (fr & 7)) * 16;      // tile row offsets are multiples of 16 rows
  if constexpr (KG == 1 && TN <= 2 * TM + 4 && !INTERLEAVE) {
    // FRAGMENT-PIPELINED loop (round 4): the fragments of K-step it + 1 are read from LDS -- and its A values split -- WHILE the MFMAs of
    // step it run, from a second register set; the barrier at the top of an iteration then says "everybody has read stage it, stage it + 1
    // has landed".  Before, a wave read its 2 (TM + TN) fragments, waited for them, split, and only then issued its 3 TM TN MFMAs: with one
    // or two waves per SIMD the matrix core idled through every read phase (tools/igemm_x3_probe.py: compute without DMA took 2.2x its MFMA
    // time).  Same products in the same order: results are bit-identical to the plain loop (cfp_debug_set(28, 1) keeps it for A/B runs).
    if (p.probe == 0) {
      f16x8 bh[2][TN], bl[2][TN], ah[2][TM], al[2][TM];
      f32x4 xr[TM][2];
      auto read_frags = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
        const unsigned char* cA = gsm + buf * STAGE_BYTES;
        const unsigned char* cB = gsm + buf * STAGE_BYTES + BM * 128 + b_row0 * 128;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[S][j] = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc0);
          bl[S][j] = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc1);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = (wm * TM + i) * 16 + fr;
          xr[i][0] = *reinterpret_cast<const f32x4*>(cA + r * 128 + pc0);
          xr[i][1] = *reinterpret_cast<const f32x4*>(cA + r * 128 + pc1);
        }
      };
      auto split_frags = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) split8(xr[i][0], xr[i][1], ah[S][i], al[S][i]);
      };
      // stage 0 -> register set 0
      {
        const int ahead = min(nkg - 1, STAGES - 2);             // stages younger than stage 0 that may stay in flight
        if (ahead >= 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * LPS>();
        else if (ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (STAGES - 1 < nkg) issue(k0 + (STAGES - 1), (STAGES - 1) % STAGES);
      read_frags(IntC<0>{}, 0);
      split_frags(IntC<0>{});
      auto body = [&](auto CUR, int it) {
        constexpr int C = decltype(CUR)::value, N = C ^ 1;
        const bool more = it + 1 < nit;
        if (more) {
          const int ahead = min(nkg - 2 - it, STAGES - 2);      // younger stages than it + 1 that may stay in flight
          if (ahead >= 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * LPS>();
          else if (ahead == 1) wait_vmcnt<LPS>();
          else wait_vmcnt<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of stage `it` are in its registers
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (it + STAGES < nkg) issue(k0 + it + STAGES, it % STAGES);      // into the stage everybody has just finished reading
          read_frags(IntC<N>{}, (it + 1) % STAGES);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[C][j], ah[C][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[C][j], al[C][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[C][j], ah[C][i], acc[i][j], 0, 0, 0);
          }
        if (more) split_frags(IntC<N>{});
      };
      int it = 0;
      for (; it + 1 < nit; it += 2) { body(IntC<0>{}, it); body(IntC<1>{}, it + 1); }
      if (it < nit) body(IntC<0>{}, it);
      wait_vmcnt<0>();
      return;
    }
  }
  for (int it = 0; it < nit; ++it) {
    const int buf = it % STAGES;
    const int ahead = min(nkg - 1 - it, STAGES - 2);
    if (ahead >= 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * LPS>();
    else if (ahead == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of the previous step are complete (see loftr_tail_x3.hip)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (KG == 2 && it >= nkg) continue;
    if (it + STAGES - 1 < nkg && !(p.probe & 1)) issue(k0 + kg + (it + STAGES - 1) * KG, (it + STAGES - 1) % STAGES);
    if (p.probe & 2) continue;      // timing probes (cfp_debug_set key 16; results are garbage): 1 = no DMA after the prologue, 2 = no reads / MFMAs,
                                    // 4 = no hi / lo split (one MFMA per block), 8 = split but one MFMA per block

    const unsigned char* cA = gsm + buf * STAGE_BYTES;
    const unsigned char* cB = gsm + buf * STAGE_BYTES + BM * 128 + b_row0 * 128;
    if constexpr (TN > 2 * TM) {
      // wide wave tiles (the fused bin head: 16 column tiles): the A fragments are split first and stay, the weight fragments stream
      // through eight registers at a time
      f16x8 ahi[TM], alo[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (INTERLEAVE ? (i * WM + wm) : (wm * TM + i)) * 16 + fr;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(cA + r * 128 + pc0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(cA + r * 128 + pc1);
        split8(x0, x1, ahi[i], alo[i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f16x8 bh = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc0);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ahi[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, alo[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ahi[i], acc[i][j], 0, 0, 0);
        }
      }
      continue;
    }
    f16x8 bhi[TN], blo[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bhi[j] = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc0);
      blo[j] = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc1);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = (INTERLEAVE ? (i * WM + wm) : (wm * TM + i)) * 16 + fr;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(cA + r * 128 + pc0);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(cA + r * 128 + pc1);
      f16x8 ahi, alo;
      if (p.probe & 4) {
        ahi = __builtin_bit_cast(f16x8, x0); alo = __builtin_bit_cast(f16x8, x1);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhi[j], ahi, acc[i][j], 0, 0, 0);
        continue;
      }
      split8(x0, x1, ahi, alo);
      if (p.probe & 8) {
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhi[j], ahi, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][0][e & 3] += (float)alo[e];
        continue;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {      // transposed (weights as the row operand): acc[r] = 4 CONSECUTIVE CHANNELS (n = 4 fq + r) of pixel row fr
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blo[j], ahi, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhi[j], alo, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhi[j], ahi, acc[i][j], 0, 0, 0);
      }
    }
  }
  wait_vmcnt<0>();
}


// ---- A-direct K loop (round 4, late): the 4 x 1 wave layouts keep a wave's A rows private, so staging them in LDS buys nothing -- and the LDS
// port is what bounds these kernels (fragment reads + DMA writes need 1.2-2.3x the MFMA clocks of a K-step).  Here the A values go
// straight from global memory into the registers of the lane that feeds them to the matrix core (lane (fr, fq) of row block i: row 16 i + fr,
// 16-byte chunks fq and fq + 4 of the K-step -- one wave instruction covers sixteen 64-byte row segments), two K-steps ahead; only the W tile
// travels through LDS.  LDS traffic per K-step drops by the whole A part (read + DMA write): 128 x 128: 112 -> 80 KB, 128 x 64: 72 -> 40 KB.
// Same products in the same order as the staged loops: bit-identical results.  Two W stages, no K groups, no split-K.
//
// Order of the vector-memory operations (the counted wait relies on it; the empty asm statements keep the compiler from moving the plain
// loads across the DMA builtins):  W(0) A(0) W(1) A(1) | top of iteration it, after the barrier: W(it + 2) A(it + 2).  At the top of
// iteration it the youngest operations are W(it + 1) A(it + 1): the W stage has landed once at most the 2 TM loads of A(it + 1) are
// outstanding; those are waited for (by the compiler) where they are split, after the MFMAs of step it.
template <int BM, int BN>
__device__ __forceinline__ void x3_mainloop_ad(const ConvP& p, const f16_t* __restrict__ wt, int m0, int m_end, int n0, int nk,
                                               unsigned char* smem, f32x4 (&acc)[BM / 64][BN / 16]) {
  constexpr int TM = BM / 64, TN = BN / 16;
  constexpr int NBG = BN / 8;
  constexpr int NB = (NBG + 3) / 4;
  constexpr int STAGE_BYTES = BN * 128;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;
  const int wrow = ((p.K + 31) >> 5) * 64;
  const float* __restrict__ in = reinterpret_cast<const float*>(p.in);
  const void* zsrc = reinterpret_cast<const void*>(g_zero16_x3);

  const float* a_ptr[TM];
  int a_hi0[TM], a_wi0[TM];
  unsigned a_okmask = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wave * TM + i) * 16 + fr;
    const bool ok = m < m_end;
    if (ok) a_okmask |= 1u << i;
    const int mm = ok ? m : m0;
    {      // one path for 1x1 and k x k (a pointwise layer is a 1 x 1 window: no uniform branch around the loads below)
      const int wo = mm % p.Wo;
      const int t = mm / p.Wo;
      const int ho = t % p.Ho;
      const int b = t / p.Ho;
      a_hi0[i] = ho * p.stride - p.pad_t;
      a_wi0[i] = wo * p.stride - p.pad_l;
      a_ptr[i] = in + (((long long)b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.in_ld;   // virtual when in the halo
    }
  }
  // im2col position of this lane's two chunks in the next K-step to be loaded
  int c_cc[2] = {0, 0}, c_kh[2] = {0, 0}, c_kw[2] = {0, 0};
  {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = (fq + 4 * h) * 4;
      const int tap = kk / p.Cin;
      c_cc[h] = kk - tap * p.Cin;
      c_kh[h] = tap / p.KW;
      c_kw[h] = tap - c_kh[h] * p.KW;
    }
  }
  const f16_t* b_ptr[NB];
  unsigned b_okmask = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + ((j * 4 + wave) % NBG) * 8 + rsub;
    const bool ok = n < p.Cout;
    if (ok) b_okmask |= 1u << j;
    b_ptr[j] = wt + (long long)(ok ? n : 0) * wrow;
  }
  auto issue_w = [&](int ks, int buf) {
    unsigned char* sB = smem + buf * STAGE_BYTES;
    const int kw16 = (ks * 8 + lc) * 8;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = (b_okmask >> j) & 1u;
      glds16(ok ? (const void*)(b_ptr[j] + kw16) : zsrc, sB + ((j * 4 + wave) % NBG) * 1024);
    }
    asm volatile("" ::: "memory");
  };
  f32x4 xr[2][TM][2];
  auto load_a = [&](auto SET, int ks) {      // K-steps must be asked for in order (the im2col position advances by one step per call)
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = (ks * 8 + fq + 4 * h) * 4;
      const bool kok = kk < p.K;
      {
        const int off = (c_kh[h] * p.W + c_kw[h]) * p.in_ld + c_cc[h];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const bool ok = kok && ((a_okmask >> i) & 1u) && (unsigned)(a_hi0[i] + c_kh[h]) < (unsigned)p.H &&
                          (unsigned)(a_wi0[i] + c_kw[h]) < (unsigned)p.W;
          xr[S][i][h] = *reinterpret_cast<const f32x4*>(ok ? (const void*)(a_ptr[i] + off) : zsrc);
        }
        c_cc[h] += 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) {      // Cin >= 8 (host): at most four taps per 32 channels; selects, no divergent loop
          const bool wrap = c_cc[h] >= p.Cin;
          c_cc[h] -= wrap ? p.Cin : 0;
          const int kw1 = c_kw[h] + (wrap ? 1 : 0);
          const bool roww = kw1 == p.KW;
          c_kw[h] = roww ? 0 : kw1;
          c_kh[h] += roww ? 1 : 0;
        }
      }
    }
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nk <= 0) return;

  const int pc0 = ((fq) ^ (fr & 7)) * 16, pc1 = ((4 + fq) ^ (fr & 7)) * 16;
  f16x8 bh[2][TN], bl[2][TN], ah[2][TM], al[2][TM];
  auto read_w = [&](auto SET, int buf) {
    constexpr int S = decltype(SET)::value;
    const unsigned char* cB = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[S][j] = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc0);
      bl[S][j] = *reinterpret_cast<const f16x8*>(cB + (j * 16 + fr) * 128 + pc1);
    }
  };
  auto split_a = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) split8(xr[S][i][0], xr[S][i][1], ah[S][i], al[S][i]);
  };
  issue_w(0, 0);
  load_a(IntC<0>{}, 0);
  if (nk > 1) {
    issue_w(1, 1);
    load_a(IntC<1>{}, 1);
    wait_vmcnt<NB + 2 * TM>();          // W(0) and A(0) have landed; W(1) / A(1) may still be in flight
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_w(IntC<0>{}, 0);
  split_a(IntC<0>{});
  // CUR: register set of step it; MORE: step it + 1 exists; LOAD: step it + 2 exists
  auto body = [&](auto CUR, auto MORE, auto LOAD, int it) {
    constexpr int C = decltype(CUR)::value, N = C ^ 1;
    if constexpr (decltype(MORE)::value) {
      wait_vmcnt<2 * TM>();                                    // W(it + 1) has landed (see the order above)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of W stage `it` are in its registers
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (decltype(LOAD)::value) {
        issue_w(it + 2, it & 1);                               // into the stage everybody has just finished reading
        load_a(IntC<C>{}, it + 2);                             // set C held A(it), split during the previous iteration
      }
      read_w(IntC<N>{}, (it + 1) & 1);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[C][j], ah[C][i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[C][j], al[C][i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[C][j], ah[C][i], acc[i][j], 0, 0, 0);
      }
    if constexpr (decltype(MORE)::value) split_a(IntC<N>{});
  };
  constexpr IntC<0> F{};
  constexpr IntC<1> T{};
  int it = 0;
  for (; it + 3 < nk; it += 2) { body(F, T, T, it); body(T, T, T, it + 1); }
  const int left = nk - it;            // 1, 2 or 3 (it is even)
  if (left == 3) { body(F, T, T, it); body(T, T, F, it + 1); body(F, F, F, it + 2); }
  else if (left == 2) { body(F, T, F, it); body(T, F, F, it + 1); }
  else body(F, F, F, it);
  wait_vmcnt<0>();
}

// OCC > 0: waves per SIMD the register allocation must leave room for (second argument of __launch_bounds__).  Left alone the compiler lands a
// few registers past a step of the occupancy ladder (512 / waves, in eights): 134 VGPRs for the 64 x 64 and 128 x 32 tiles (3 waves where 128 is
// 4), 178 for 64 x 128 (2 where 168 is 3).  Squeezed, they spill 3 ... 10 registers to scratch.  Measured on one box (tools/mode_bench.py, float32
// storage): with four batches in flight the extra resident workgroup per CU wins (4.200 vs 4.259 ms per batch of 8: these GEMMs spend half their
// wave-cycles waiting on K loops of 4 ... 20 steps); a single graph LOSES (6.56 vs 6.42 ms, batch 1 3.22 vs 3.18: too few workgroups to fill
// the slots, and each is a little slower).  So these are separate instantiations (variant ids 34 - 36) that the plan picks under the
// CFP_CONV_IN_FLIGHT hint only.  128 x 64 at 168 registers spills 18 and measured slower either way: no such variant.
template <int BM, int BN, int WM, int WN, int STAGES, bool SPLITK, int KG = 1, bool AD = false, int OCC = 0>
__global__ __launch_bounds__(256 * KG, (OCC > 0 ? OCC : 1)) void igemm_x3_kernel(ConvP p, float* slabs, int splits, unsigned* tickets) {
  static_assert(WM * WN == 4, "four waves per group");
  static_assert(!AD || (WM == 4 && WN == 1 && KG == 1 && !SPLITK && STAGES == 2), "A-direct: 4 x 1 waves, two stages, whole K");
  static_assert(KG == 1 || (KG == 2 && !SPLITK), "K groups are the in-workgroup alternative to split-K");
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 2 ? wave8 >> 2 : 0;
  const int wave = wave8 & 3;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;

  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  int sp = 0;
  if constexpr (SPLITK) { sp = bid % splits; bid /= splits; }
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int nk_all = (p.K + 31) >> 5;
  int m0, m_end;
  const f16_t* __restrict__ wt = reinterpret_cast<const f16_t*>(p.w);
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
  int k0 = 0, k1 = nk_all;
  if constexpr (SPLITK) {
    const int per = (nk_all + splits - 1) / splits;
    k0 = sp * per;
    k1 = min(nk_all, k0 + per);
  }
  f32x4 acc[TM][TN];
  if constexpr (AD) x3_mainloop_ad<BM, BN>(p, wt, m0, m_end, n0, nk_all, smem, acc);
  else x3_mainloop<BM, BN, WM, WN, STAGES, KG>(p, wt, m0, m_end, n0, k0, k1, smem, acc);
  const int a_row0 = wm * (BM / WM), b_row0 = wn * (BN / WN);
  if constexpr (KG == 2) {
    __syncthreads();   // every wave is done with the operand stages: LDS carries the second group's accumulators to the first
    static_assert(BM * BN * 4 <= KG * STAGES * STAGE_BYTES, "accumulator exchange must fit in the operand LDS");
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + wave * (TM * TN * 64) + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) xch[(i * TN + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 o = xch[(i * TN + j) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += o[r];
      }
  }

  if constexpr (SPLITK) {
    if (tickets == nullptr) {                   // the finishing sum is a second launch (splitk_reduce_kernel)
      float* slab = slabs + (long long)sp * p.M * p.Cout;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int m = m0 + a_row0 + i * 16 + fr;
          const int n = n0 + b_row0 + j * 16 + fq * 4;
          if (m < m_end && n < p.Cout) *reinterpret_cast<f32x4*>(slab + (long long)m * p.Cout + n) = acc[i][j];
        }
      return;
    }
    // Ticketed finish: the workgroup that arrives LAST at its output tile sums the slabs -- every one read back from memory in split order, so the
    // value is splitk_reduce_kernel's bit for bit whichever workgroup that is -- and runs the epilogue below.
    // Visibility across the eight XCDs (private L2s) WITHOUT fences: a device-scope release fence is buffer_wbl2 -- a walk over the whole L2 per
    // wave, measured +23 us per launch with 1 280 waves -- so the slab traffic itself is device-coherent instead: stores and loads carry the
    // sc1 scope bit (write-through to memory / never served from a stale L2 line), every wave waits for its stores to be acknowledged
    // (vmcnt 0) before the barrier, and only then is the ticket taken (a device-scope atomic).
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    constexpr int SC1 = 16;                     // cache-policy bit of the buffer intrinsics: sc1 = agent (device) scope on gfx940+
    const long long slab_elems = (long long)p.M * p.Cout;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, (int)(slab_elems * splits * 4), 0x00020000);     // host: < 2^31 bytes
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = m0 + a_row0 + i * 16 + fr;
        const int n = n0 + b_row0 + j * 16 + fq * 4;
        if (m < m_end && n < p.Cout)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc[i][j]), rs, (int)((sp * slab_elems + (long long)m * p.Cout + n) * 4), 0, SC1);
      }
    wait_vmcnt<0>();
    __syncthreads();
    int* s_last = reinterpret_cast<int*>(smem);       // (the operand stages are free after the barrier; a static __shared__ would push the
    if (tid == 0) {                                   //  kernel past the 160 KB the dynamic allocation may ask for)
      const unsigned t = __hip_atomic_fetch_add(tickets + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_last = t == (unsigned)(splits - 1);
      if (*s_last) __hip_atomic_store(tickets + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // handed back zeroed for the next launch
    }
    __syncthreads();
    if (!*s_last) return;
    // U splits' loads are issued together (each is a trip to memory: one at a time would be a chain of `splits` latencies), added in split order
    constexpr int U = TM * TN >= 16 ? 1 : TM * TN >= 8 ? 2 : 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < splits; q0 += U) {
      f32x4 x[U][TM][TN];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int m = m0 + a_row0 + i * 16 + fr;
            const int n = n0 + b_row0 + j * 16 + fq * 4;
            const bool ok = m < m_end && n < p.Cout && q0 + u < splits;
            // (a refused element reads past the descriptor's range: zeros, and its sum is never stored)
            const long long off = ok ? ((q0 + u) * slab_elems + (long long)m * p.Cout + n) * 4 : 0x7ffffff0ll;
            x[u][i][j] = __builtin_bit_cast(f32x4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, SC1));
          }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (q0 + u < splits) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i][j][r] += x[u][i][j][r];
        }
    }
  }
  {
    float* __restrict__ out = reinterpret_cast<float*>(p.out);
    const float* __restrict__ res = reinterpret_cast<const float*>(p.res);
    if constexpr (WN == 1 && KG == 1) {
      if (p.ln_gamma != nullptr) {
        // LayerNorm over the output channels fused in (host: BN == Cout, so the tile holds whole rows): out = LN(act(conv * scale + shift)) *
        // gamma + beta + residual (transformer.py:63,68-70).  A pixel's BN channels sit in TN x 4 registers of the four lanes fr, fr + 16,
        // fr + 32, fr + 48: the two-pass statistics (mean, then squared deviations, as cfp_layernorm) cost two cross-lane steps each.
        with_act(p.act, [&](auto A) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + a_row0 + i * 16 + fr;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const int n = j * 16 + fq * 4;
              const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
              const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int r = 0; r < 4; ++r) { acc[i][j][r] = act_c<decltype(A)::value>(acc[i][j][r] * sc[r] + sh[r]); sum += acc[i][j][r]; }
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.f / (float)BN);
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean; qq = fmaf(d, d, qq); }
            qq += __shfl_xor(qq, 16, 64);
            qq += __shfl_xor(qq, 32, 64);
            const float rstd = rsqrtf(qq * (1.f / (float)BN) + p.ln_eps);
            if (m >= m_end) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const int n = j * 16 + fq * 4;
              const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + n), bt = *reinterpret_cast<const f32x4*>(p.ln_beta + n);
              f32x4 y;
#pragma unroll
              for (int r = 0; r < 4; ++r) y[r] = (acc[i][j][r] - mean) * rstd * g[r] + bt[r];
              if (res) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(res + (long long)m * p.res_ld + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] += rv[r];
              }
              *reinterpret_cast<f32x4*>(out + (long long)m * p.out_ld + n) = y;
            }
          }
        });
        return;
      }
    }
    with_act(p.act, [&](auto A) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + b_row0 + j * 16 + fq * 4;
        if (n >= p.Cout) continue;            // Cout % 4 == 0: a lane's four channels are in or out together
        const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + a_row0 + i * 16 + fr;
          if (m >= m_end) continue;
          f32x4 y;
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = act_c<decltype(A)::value>(acc[i][j][r] * sc[r] + sh[r]);
          if (res) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(res + (long long)m * p.res_ld + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] += rv[r];
          }
          *reinterpret_cast<f32x4*>(out + (long long)m * p.out_ld + n) = y;
        }
      }
    });
  }
}

struct Cfg { int bm, bn, stages, kg = 1; bool ad = false; };
// same variant ids as conv_igemm2.hip (the plan function is shared); a K-step here is 32 channels
constexpr Cfg kCfg[] = {
    {128, 128, 3}, {128, 128, 2}, {128, 64, 3}, {128, 64, 4}, {64, 64, 3}, {64, 64, 4}, {256, 32, 3}, {256, 32, 2}, {128, 32, 3}, {128, 32, 4},
    {256, 16, 2}, {128, 16, 4}, {64, 128, 3}, {64, 64, 2}, {128, 64, 2}, {64, 128, 2}, {128, 32, 2}, {32, 64, 3}, {32, 128, 3},
    {64, 64, 2, 2}, {64, 64, 3, 2}, {64, 128, 2, 2},
    // f16x3 only (ids >= 22): larger row tiles (fewer L2 -> LDS bytes per MFMA) and the other wave layouts of the common shapes
    {256, 128, 2}, {256, 64, 2}, {64, 64, 2}, {128, 64, 2}, {128, 128, 2}, {64, 128, 2},
    // A-direct (ids >= 28): 4 x 1 waves, the A values go global -> registers, only the W tile through LDS
    {128, 128, 2, 1, true}, {128, 64, 2, 1, true}, {64, 64, 2, 1, true}, {256, 64, 2, 1, true}, {128, 32, 2, 1, true}, {256, 128, 2, 1, true},
    // ids 34-36: ids 13 / 16 / 15 compiled for 4 / 4 / 3 waves per SIMD (in-flight plans; the same arithmetic in the same order)
    {64, 64, 2}, {128, 32, 2}, {64, 128, 2},
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

template <int BM, int BN, int WM, int WN, int STAGES, int KG = 1, bool AD = false, int OCC = 0>
int launch_x3(const ConvP& p, float* slabs, int splits, hipStream_t s, unsigned* tickets) {
  const size_t lds = AD ? (size_t)STAGES * BN * 128 : (size_t)KG * STAGES * (BM + BN) * 128;
  if (lds > 160 * 1024) return -1;
  if ((KG == 2 || AD || OCC > 0) && splits > 1) return -4;
  if (AD && p.Cin < 8) return -5;
  long long tiles_m = p.rows_per_batch > 0 ? (long long)p.B * cdiv(p.rows_per_batch, BM) : cdiv(p.M, BM);
  long long tiles = tiles_m * cdiv(p.Cout, BN);
  if (splits <= 1) {
    auto k = igemm_x3_kernel<BM, BN, WM, WN, STAGES, false, KG, AD, OCC>;
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256 * KG), lds, s, p, slabs, 1, (unsigned*)nullptr);
  } else if constexpr (KG == 1 && !AD && OCC == 0) {
    auto k = igemm_x3_kernel<BM, BN, WM, WN, STAGES, true>;
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
    if (tickets && tiles > CFP_TICKET_SLOTS) return -6;
    hipLaunchKernelGGL(k, dim3((unsigned)(tiles * splits)), dim3(256), lds, s, p, slabs, splits, tickets);
  }
  return 0;
}

// ---- fused bin head (f16x3): conv_out 1x1 (Cin -> 256 logits) + softmax over the bins + expectation with the bin centres, float32 prob
// written NCHW (deltar.py:45-62: conv_out = Conv2d(128, n_bins, 1) + Softmax(dim=1), pred = sum(prob * centers)).  The 128 x 256 logit tile
// never leaves the registers (transposed accumulators: a lane holds 64 of its pixel's 256 logits, the other three quarters sit in lanes
// fr + 16, + 32, + 48); the probabilities of 64 pixels at a time cross LDS so that the NCHW store is 16-byte vectors along the pixel axis,
// 256 contiguous bytes per bin.  Saves the logits' round trip (629 MB written + read at batch 8) and a launch.
constexpr int HB_N = 256, HB_PP = 64 + 4;      // HB_PP: floats per bin row of the 64-pixel probability tile in LDS

// HB_M = 64: 80 KB of LDS, two workgroups per CU -- one's softmax / store phase overlaps the other's K loop (measured against HB_M = 128,
// one workgroup per CU with a third fewer weight re-reads: tools/head_bench.py --x3)
template <int HB_M>
__global__ __launch_bounds__(256) void bin_head_x3_kernel(ConvP p, const float* __restrict__ bias, const float* __restrict__ centers,
                                                          float* __restrict__ prob, float* __restrict__ pred, int HW) {
  constexpr int TM = HB_M / 64, TN = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * HB_M;
  f32x4 acc[TM][TN];
  x3_mainloop<HB_M, HB_N, 4, 1, 2, 1, true>(p, reinterpret_cast<const f16_t*>(p.w), m0, p.M, 0, 0, (p.K + 31) >> 5, smem, acc);
  float* sP = reinterpret_cast<float*>(smem);      // [256][HB_PP]
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();      // the operand stages (first round) / the previous round's probability tile are done with
    const int m = m0 + i * 64 + wave * 16 + fr;      // this lane's pixel (INTERLEAVE row map)
    const int bidx = min(m, p.M - 1) / HW;
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + j * 16 + fq * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[i][j][r] += bs[r]; mx = fmaxf(mx, acc[i][j][r]); }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[i][j][r] = __expf(acc[i][j][r] - mx); s += acc[i][j][r]; }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float inv = 1.f / s;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 cen = *reinterpret_cast<const f32x4*>(centers + (long long)bidx * HB_N + j * 16 + fq * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[i][j][r] *= inv; dot = fmaf(acc[i][j][r], cen[r], dot); }
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (fq == 0 && m < p.M) pred[m] = dot;
    if (prob == nullptr) continue;
    const int pl = wave * 16 + fr;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sP[(j * 16 + fq * 4 + r) * HB_PP + pl] = acc[i][j][r];
    __syncthreads();
    for (int q = tid; q < HB_N * 16; q += 256) {
      const int n = q >> 4, ch = q & 15;
      const int mm = m0 + i * 64 + ch * 4;
      if (mm >= p.M) continue;
      const int b = mm / HW, hw = mm - b * HW;      // HW % 4 == 0: four pixels stay inside one image
      // written once, read by nobody on this chip soon (630 MB at batch 8): streaming store, leaves L2 / the Infinity Cache to the operands
      __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(sP + n * HB_PP + ch * 4), reinterpret_cast<f32x4*>(prob + ((long long)b * HB_N + n) * HW + hw));
    }
  }
}

// f32 [rows][K] -> packed halves [rows][ceil(K / 32) * 64]
__global__ __launch_bounds__(256) void pack_w_x3_kernel(const float* __restrict__ w, f16_t* __restrict__ out, long long rows, int K, int nk) {
  const long long total = rows * nk * 32;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i & 31);
    const long long t = i >> 5;
    const int ks = (int)(t % nk);
    const long long n = t / nk;
    const int k = ks * 32 + r;
    const float x = k < K ? w[n * K + k] : 0.f;
    const f16_t hi = f2h(x);
    const f16_t lo = (f16_t)(x - (float)hi);
    const int pos = ((r & 15) >> 2) * 8 + (r & 3) + ((r >> 4) << 2);
    f16_t* row = out + (n * nk + ks) * 64;
    row[pos] = hi;
    row[32 + pos] = lo;
  }
}

}  // namespace

int igemm_x3_num_variants() { return kNumCfg; }
void igemm_x3_variant_shape(int v, int* bm, int* bn, int* stages) { *bm = kCfg[v].bm; *bn = kCfg[v].bn; *stages = kCfg[v].stages; }

// Launch variant v (conv_igemm2.hip's ids).  Returns 0, or a negative value if the variant cannot run this problem.
int igemm_x3_launch(int v, const ConvP& p, float* slabs, int splits, hipStream_t s, unsigned* tickets) {
  switch (v) {
    case 0: return launch_x3<128, 128, 2, 2, 3>(p, slabs, splits, s, tickets);
    case 1: return launch_x3<128, 128, 2, 2, 2>(p, slabs, splits, s, tickets);
    case 2: return launch_x3<128, 64, 4, 1, 3>(p, slabs, splits, s, tickets);
    case 3: return launch_x3<128, 64, 4, 1, 4>(p, slabs, splits, s, tickets);
    case 4: return launch_x3<64, 64, 4, 1, 3>(p, slabs, splits, s, tickets);
    case 5: return launch_x3<64, 64, 4, 1, 4>(p, slabs, splits, s, tickets);
    case 6: return launch_x3<256, 32, 4, 1, 3>(p, slabs, splits, s, tickets);
    case 7: return launch_x3<256, 32, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 8: return launch_x3<128, 32, 4, 1, 3>(p, slabs, splits, s, tickets);
    case 9: return launch_x3<128, 32, 4, 1, 4>(p, slabs, splits, s, tickets);
    case 10: return launch_x3<256, 16, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 11: return launch_x3<128, 16, 4, 1, 4>(p, slabs, splits, s, tickets);
    case 12: return launch_x3<64, 128, 2, 2, 3>(p, slabs, splits, s, tickets);
    case 13: return launch_x3<64, 64, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 14: return launch_x3<128, 64, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 15: return launch_x3<64, 128, 2, 2, 2>(p, slabs, splits, s, tickets);
    case 16: return launch_x3<128, 32, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 17: return launch_x3<32, 64, 1, 4, 3>(p, slabs, splits, s, tickets);
    case 18: return launch_x3<32, 128, 1, 4, 3>(p, slabs, splits, s, tickets);
    case 19: return launch_x3<64, 64, 4, 1, 2, 2>(p, slabs, splits, s, tickets);
    case 20: return launch_x3<64, 64, 4, 1, 3, 2>(p, slabs, splits, s, tickets);
    case 21: return launch_x3<64, 128, 2, 2, 2, 2>(p, slabs, splits, s, tickets);
    case 22: return launch_x3<256, 128, 2, 2, 2>(p, slabs, splits, s, tickets);
    case 23: return launch_x3<256, 64, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 24: return launch_x3<64, 64, 2, 2, 2>(p, slabs, splits, s, tickets);
    case 25: return launch_x3<128, 64, 2, 2, 2>(p, slabs, splits, s, tickets);
    case 26: return launch_x3<128, 128, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 27: return launch_x3<64, 128, 4, 1, 2>(p, slabs, splits, s, tickets);
    case 28: return launch_x3<128, 128, 4, 1, 2, 1, true>(p, slabs, splits, s, tickets);
    case 29: return launch_x3<128, 64, 4, 1, 2, 1, true>(p, slabs, splits, s, tickets);
    case 30: return launch_x3<64, 64, 4, 1, 2, 1, true>(p, slabs, splits, s, tickets);
    case 31: return launch_x3<256, 64, 4, 1, 2, 1, true>(p, slabs, splits, s, tickets);
    case 32: return launch_x3<128, 32, 4, 1, 2, 1, true>(p, slabs, splits, s, tickets);
    case 33: return launch_x3<256, 128, 4, 1, 2, 1, true>(p, slabs, splits, s, tickets);
    case 34: return launch_x3<64, 64, 4, 1, 2, 1, false, 4>(p, slabs, splits, s, tickets);
    case 35: return launch_x3<128, 32, 4, 1, 2, 1, false, 4>(p, slabs, splits, s, tickets);
    case 36: return launch_x3<64, 128, 2, 2, 2, 1, false, 3>(p, slabs, splits, s, tickets);
    default: return -3;
  }
}

extern "C" size_t cfp_pack_w_x3_elems(long long rows, int K) { return rows <= 0 || K <= 0 ? 0 : (size_t)rows * ((K + 31) / 32) * 64; }

// Convolution / linear weights [rows][K] float32 (K = KH * KW * Cin in (kh, kw, ci) order) -> the pre-split operand of the f16x3 kernels:
// [rows][ceil(K / 32)][hi(32) | lo(32)] IEEE halves in the kernels' lane order, zero padded to whole K-steps.  `rows` may be B * Cout
// (per-image weights).
extern "C" int cfp_pack_w_x3(const float* w, void* out, long long rows, int K, cfp_stream_t stream) {
  CFP_REQUIRE(w && out, CFP_EINVAL, "cfp_pack_w_x3: null pointer");
  CFP_REQUIRE(rows > 0 && K > 0 && aligned16(out), CFP_ESHAPE, "cfp_pack_w_x3: bad shape / alignment");
  const int nk = (K + 31) / 32;
  const long long total = rows * nk * 32;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_w_x3_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, reinterpret_cast<f16_t*>(out), rows, K, nk);
  return cfp_check_launch("cfp_pack_w_x3");
}

int g_bin_head_x3_rows = 64;      // cfp_debug_set key 25: pixel rows per workgroup of the fused f16x3 bin head (64 / 128)
// cfp_bin_head_fused for dtype CFP_F32X3 (head.hip dispatches here): x float32 [B*HW, x_ld], w = cfp_pack_w_x3 of conv_out's [256, Cin].
int bin_head_x3_launch(const void* x, int x_ld, const void* w, const float* bias, const float* centers, float* prob, float* pred, int B, int HW,
                       int Cin, hipStream_t s) {
  ConvP p;
  p.in = x; p.w = w; p.out = nullptr; p.res = nullptr; p.scale = nullptr; p.shift = nullptr;
  p.in_ld = x_ld; p.out_ld = 0; p.res_ld = 0;
  p.B = 1; p.H = 1; p.W = B * HW; p.Cin = Cin; p.Ho = 1; p.Wo = B * HW; p.Cout = HB_N;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad_t = 0; p.pad_l = 0; p.M = B * HW; p.K = Cin; p.act = 0; p.pointwise = 1;
  p.ln_gamma = nullptr; p.ln_beta = nullptr; p.ln_eps = 0.f; p.rows_per_batch = 0; p.w_bstride = 0; p.f16 = 0; p.k2 = 0; p.dil = 1; p.mom = nullptr; p.probe = 0;
  p.up_src = nullptr; p.up_ld = p.up_C = p.up_H = p.up_W = 0; p.up_sy = p.up_sx = 0.f;
  static_assert((size_t)HB_N * HB_PP * 4 <= (size_t)2 * (64 + HB_N) * 128, "probability tile must fit in the operand LDS");
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)bin_head_x3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    if (hipFuncSetAttribute((const void*)bin_head_x3_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    attr = true;
  }
  if (g_bin_head_x3_rows == 128)
    hipLaunchKernelGGL(bin_head_x3_kernel<128>, dim3(cdiv(p.M, 128)), dim3(256), (size_t)2 * (128 + HB_N) * 128, s, p, bias, centers, prob, pred, HW);
  else
    hipLaunchKernelGGL(bin_head_x3_kernel<64>, dim3(cdiv(p.M, 64)), dim3(256), (size_t)2 * (64 + HB_N) * 128, s, p, bias, centers, prob, pred, HW);
  return 0;
}
