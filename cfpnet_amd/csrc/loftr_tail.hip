// Fused tail of a LoFTR encoder layer (transformer.py:48-71, attention.py:48-49), bf16:
//
//   msg = (Q' KV[g]) / (Q' . Ksum[g] + eps) * S          linear-attention apply, per head
//   y1  = LayerNorm1(msg @ Wm^T)                          merge + norm1
//   h   = relu([x | y1] @ W0^T)                           mlp.0 (the concat is a split K range)
//   out = LayerNorm2(h @ W2^T) + x                        mlp.2 + norm2 + residual
//
// Unfused this is 4-5 launches per layer (apply, merge+LN, mlp0, mlp2+LN) over tensors of a few MB,
// 18 layers per forward.  Here a wave owns 16 token rows from the apply to the final store: its
// msg / y1 / x / h tiles live in a private LDS region (row pitch padded by 16 bytes, so the MFMA
// A-operand read of 16 rows is conflict-free), and only the weights are shared: each GEMM streams
// its [N][64] weight slabs global -> LDS with `global_load_lds_dwordx4` (XOR-swizzled 128-byte rows,
// double buffered, one barrier per 64-wide K step).  Rows never leave their wave, so both
// LayerNorms are a 16-lane shuffle reduction over the accumulator registers.
// Rounding points match the unfused path (msg, the GEMM outputs before each LayerNorm and h are
// rounded to bf16), so the two paths agree to accumulation order.
#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16c[4] = {0u, 0u, 0u, 0u};
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <typename H> struct TailP {
  const H* q; const float* kv; const float* ksum; const H* x; H* out;
  const H* wq;                   // optional: q_proj weights [D][D]; the kernel then computes q = x @ wq^T for its own rows and `q` is unused
  const H* wm; const H* w0; const H* w2;
  const float* g1; const float* b1; const float* g2; const float* b2;
  int q_ld, x_ld, out_ld;
  int rows, Hq, Wq, qth, qtw, ggy, ggx;
  FastDiv fwq, fhq, fqth, fqtw;  // rows < 2^31: the token -> (image, y, x) -> key-group split without 64-bit divisions (four of them per lane and
                                 // row tile were ~600 VALU instructions: about a third of the kernel at D = 32)
  float v_length, eps, ln_eps;
};

// One GEMM of the chain for this wave's 16 rows: acc[j] += A[16 x K] * W[N x K]^T, N = NT * 16.
// `afrag(k)` returns the lane's A fragment for columns [k, k + 32) of the wave-private operand.
// All four waves of the workgroup must call it together (they share the weight slabs).
template <typename H, int NT, int BSTAGE, int WAVES, typename AF>
__device__ __forceinline__ void tail_gemm(f32x4 (&acc)[NT], const H* __restrict__ W, int K, AF afrag, unsigned char* sB,
                                          int wave, int lane) {
  constexpr int N = NT * 16;
  constexpr int NBG = N / 8;                  // 8-row DMA groups of a weight slab
  constexpr int NBW = (NBG + WAVES - 1) / WAVES;      // LDS-DMA instructions per wave and slab
  // weight slabs of THIS GEMM are N * 128 bytes; the region holds 2 * BSTAGE bytes (two slabs of the widest GEMM, N = 2 D): the N = D GEMMs
  // fit three of theirs in it and keep two K-steps of DMA in flight (round 4: a K-step is a few MFMAs against a ~0.7-1 us round trip of
  // its weight tile from L2; counted wait as in loftr_tail_x3.hip)
  constexpr int SST = N * 128;
  constexpr int STG = (3 * SST <= 2 * BSTAGE) ? 3 : 2;
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;
  const H* zsrc = reinterpret_cast<const H*>(g_zero16c);
  const int nk = (K + 63) >> 6;
  auto issue = [&](int ks, int st) {
    const int kk = ks * 64 + lc * 8;
    const bool kok = kk < K;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int g = (j * WAVES + wave) % NBG;
      const int n = g * 8 + rsub;
      glds16(kok ? W + (long long)n * K + kk : zsrc, sB + st * SST + g * 1024);
    }
  };
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the wait + clobber in front of the barrier: `s_barrier` has no memory semantics for the compiler and LDS reads are asynchronous, so the
  // previous GEMM's last fragment reads must be forced complete before a faster wave may stream the next weights into the slab they read
  // (round 4: the float32 twin of this kernel, loftr_tail_x3.hip, showed the race at D = 32; this one never did, same protocol now)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();               // every wave is done with the previous GEMM's slabs
  asm volatile("" ::: "memory");
  issue(0, 0);
  if (STG == 3 && nk > 1) issue(1, 1);
  for (int ks = 0; ks < nk; ++ks) {
    if (STG == 3 && ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW) : "memory");      // slab ks landed, slab ks + 1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (STG == 3) { if (ks + 2 < nk) issue(ks + 2, (ks + 2) % 3); }      // slab (ks - 1) % 3: everybody has read it (barrier above)
    else if (ks + 1 < nk) issue(ks + 1, (ks + 1) & 1);
    const unsigned char* cB = sB + (ks % STG) * SST;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int k = ks * 64 + sub * 32;
      if (k < K) {                            // uniform
        const s16x8 a = afrag(k);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const s16x8 b = *reinterpret_cast<const s16x8*>(cB + (j * 16 + fr) * 128 + (((sub * 4 + fq) ^ (fr & 7)) * 16));
          acc[j] = mfma16<H>(a, b, acc[j]);
        }
      }
    }
  }
}

// LayerNorm over the N = NT*16 columns of each of this lane's 4 rows (row = fq*4 + r, col = j*16 + fr);
// the values are first rounded to bf16 (the unfused path stores the GEMM output in bf16).
template <typename H, int NT>
__device__ __forceinline__ void tail_layernorm(f32x4 (&acc)[NT], const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float eps, int fr) {
  constexpr float inv_n = 1.f / (float)(NT * 16);
  float g[NT], bt[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { g[j] = gamma[j * 16 + fr]; bt[j] = beta[j * 16 + fr]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[j][r] = to_f32<H>(from_f32<H>(acc[j][r])); s += acc[j][r]; }
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

// WAVES waves of 16 token rows per workgroup: 4, or 1 / 2 for few token rows (a single image), as loftr_tail_x3.hip (round 5) -- narrower
// workgroups fill more of an otherwise idle chip and shorten each one's chain; same arithmetic per row.
template <typename H, int D, int HEADS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void loftr_tail_kernel(TailP<H> p) {
  constexpr int d = D / HEADS;
  constexpr int PA = D + 8, PH = 2 * D + 8;               // row pitches (elements): +16 bytes
  constexpr int WAVE_LDS = (2 * PA + PH) * 16 * 2;         // msg/y1 | x | h tiles of one wave
  constexpr int BSTAGE = 2 * D * 128;                      // largest weight slab: [2D rows][64 k]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  unsigned char* sB = smem;
  H* tMsg = reinterpret_cast<H*>(smem + 2 * BSTAGE + wave * WAVE_LDS);
  H* tX = tMsg + 16 * PA;
  H* tH = tX + 16 * PA;
  const long long row0 = (long long)blockIdx.x * (16 * WAVES) + wave * 16;

  // ---- x tile -> LDS (16-byte vectors) -----------------------------------------------------------
  constexpr int XCH = D / 8;                               // 16-byte chunks per row
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < p.rows) v = *reinterpret_cast<const u32x4*>(p.x + m * p.x_ld + ch * 8);
    *reinterpret_cast<u32x4*>(tX + r * PA + ch * 8) = v;
  }

  // ---- optional q projection for this wave's rows (transformer.py:45: q = q_proj(x)): one more GEMM of the chain instead of a separate
  // launch that writes [rows, D] and is read back here; q is rounded to the storage type like the unfused GEMM's output ----------------
  const bool own_q = p.wq != nullptr;
  if (own_q) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 acc[D / 16];
    tail_gemm<H, D / 16, BSTAGE, WAVES>(acc, p.wq, D, [&](int k) { return *reinterpret_cast<const s16x8*>(tX + fr * PA + k + fq * 8); }, sB, wave, lane);
#pragma unroll
    for (int j = 0; j < D / 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tH[(fq * 4 + r) * PH + j * 16 + fr] = from_f32<H>(acc[j][r]);     // the hidden tile is free until mlp.0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- linear-attention apply: lane = (row, head slot) --------------------------------------------
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
      float qv[d];
      if (own_q) {                                           // uniform
        const H* qp = tH + r * PH + h * d;
        if constexpr (d >= 8) {
#pragma unroll
          for (int c = 0; c < d; c += 8) Vec<H>::load(qp + c, qv + c);
        } else {
#pragma unroll
          for (int c = 0; c < d; ++c) qv[c] = to_f32<H>(qp[c]);
        }
      } else {
        const H* qp = p.q + mm * p.q_ld + h * d;
        if constexpr (d >= 8) {
#pragma unroll
          for (int c = 0; c < d; c += 8) Vec<H>::load(qp + c, qv + c);
        } else {
#pragma unroll
          for (int c = 0; c < d; ++c) qv[c] = to_f32<H>(qp[c]);
        }
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
      for (int j = 0; j < d; ++j) tMsg[r * PA + h * d + j] = ok ? from_f32<H>(o[j] * zi * p.v_length) : from_bits<H>(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- merge + norm1 -------------------------------------------------------------------------------
  {
    f32x4 acc[D / 16];
    tail_gemm<H, D / 16, BSTAGE, WAVES>(acc, p.wm, D, [&](int k) { return *reinterpret_cast<const s16x8*>(tMsg + fr * PA + k + fq * 8); }, sB, wave, lane);
    tail_layernorm<H, D / 16>(acc, p.g1, p.b1, p.ln_eps, fr);
#pragma unroll
    for (int j = 0; j < D / 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tMsg[(fq * 4 + r) * PA + j * 16 + fr] = from_f32<H>(acc[j][r]);   // y1 replaces msg
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- mlp.0: [x | y1] (K = 2D) -> 2D, ReLU ---------------------------------------------------------
  {
    f32x4 acc[2 * D / 16];
    tail_gemm<H, 2 * D / 16, BSTAGE, WAVES>(acc, p.w0, 2 * D, [&](int k) {
      const H* src = k < D ? tX + fr * PA + k : tMsg + fr * PA + (k - D);
      return *reinterpret_cast<const s16x8*>(src + fq * 8);
    }, sB, wave, lane);
#pragma unroll
    for (int j = 0; j < 2 * D / 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tH[(fq * 4 + r) * PH + j * 16 + fr] = from_f32<H>(fmaxf(acc[j][r], 0.f));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- mlp.2 (K = 2D) -> D, norm2, + x ---------------------------------------------------------------
  {
    f32x4 acc[D / 16];
    tail_gemm<H, D / 16, BSTAGE, WAVES>(acc, p.w2, 2 * D, [&](int k) { return *reinterpret_cast<const s16x8*>(tH + fr * PH + k + fq * 8); }, sB, wave, lane);
    tail_layernorm<H, D / 16>(acc, p.g2, p.b2, p.ln_eps, fr);
#pragma unroll
    for (int j = 0; j < D / 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = fq * 4 + r, col = j * 16 + fr;
        tMsg[row * PA + col] = from_f32<H>(acc[j][r] + to_f32<H>(tX[row * PA + col]));     // stage the output tile
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    if (m < p.rows) *reinterpret_cast<u32x4*>(p.out + m * p.out_ld + ch * 8) = *reinterpret_cast<const u32x4*>(tMsg + r * PA + ch * 8);
  }
}

int g_tail16_waves = 0;      // cfp_debug_set key 39: 0 = by the row count, else 1 / 2 / 4 waves per workgroup (A/B)
template <typename H, int D, int HEADS, int WAVES>
int launch_tail_w(const TailP<H>& p, hipStream_t s) {
  constexpr size_t lds = 2 * (2 * D * 128) + WAVES * ((2 * (D + 8) + 2 * D + 8) * 16 * 2);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = loftr_tail_kernel<H, D, HEADS, WAVES>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)cdiv(p.rows, 16 * WAVES)), dim3(64 * WAVES), lds, s, p);
  return 0;
}
template <typename H, int D, int HEADS>
int launch_tail(const TailP<H>& p, hipStream_t s) {
  const int w = (g_tail16_waves == 1 || g_tail16_waves == 2 || g_tail16_waves == 4) ? g_tail16_waves : p.rows <= 4800 ? 1 : p.rows < 8192 ? 2 : 4;
  return w == 1 ? launch_tail_w<H, D, HEADS, 1>(p, s) : w == 2 ? launch_tail_w<H, D, HEADS, 2>(p, s) : launch_tail_w<H, D, HEADS, 4>(p, s);
}

// ---- LKPM tail (Block14.forward after the depthwise conv, convnext.py:48-58): LayerNorm(1e-6) -> pwconv1 (D -> 4D) -> GELU ->
// pwconv2 (4D -> D) -> + input, for the 16 token rows of a wave, with the same machinery as the LoFTR tail: the 4D-wide hidden tile
// lives in the wave's LDS region and never reaches HBM (unfused: a LayerNorm launch, two GEMM launches and 2 x 4D x 2 bytes per
// token of traffic -- 79 MB at the 1/4 scale of a batch of 8).  pwconv1 runs in two halves of 2D output channels so that its weight
// slabs ([2D][64], double buffered) and the four waves' tiles fit the LDS at D = 128.
template <typename H> struct LkpmP {
  const H* t; const H* xin; H* out;
  const H* w1; const H* w2;
  const float* lg; const float* lb; const float* b1; const float* b2;
  int t_ld, x_ld, out_ld, rows;
  float ln_eps;
};

template <typename H, int D>
__global__ __launch_bounds__(256) void lkpm_tail_kernel(LkpmP<H> p) {
  constexpr int PA = D + 8, PH = 4 * D + 8;               // row pitches (elements): +16 bytes
  constexpr int WAVE_LDS = (PA + PH) * 16 * 2;            // normalised-input / output tile | hidden tile of one wave
  constexpr int BSTAGE = 2 * D * 128;                     // largest weight slab: [2D rows][64 k]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  unsigned char* sB = smem;
  H* tA = reinterpret_cast<H*>(smem + 2 * BSTAGE + wave * WAVE_LDS);
  H* tH = tA + 16 * PA;
  const long long row0 = (long long)blockIdx.x * 64 + wave * 16;
  constexpr int XCH = D / 8;                               // 16-byte chunks per row

  // ---- t tile -> LDS, LayerNorm over the D channels of each row (lane = row fr, quarter fq of the channels) -------------------------
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < p.rows) v = *reinterpret_cast<const u32x4*>(p.t + m * p.t_ld + ch * 8);
    *reinterpret_cast<u32x4*>(tA + r * PA + ch * 8) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    constexpr int Q = D / 4;                               // channels per lane
    float v[Q];
#pragma unroll
    for (int c = 0; c < Q; c += 8) Vec<H>::load(tA + fr * PA + fq * Q + c, v + c);
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
    for (int c = 0; c < Q; c += 8) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[c + e] - mean) * rstd * p.lg[fq * Q + c + e] + p.lb[fq * Q + c + e];
      Vec<H>::store(tA + fr * PA + fq * Q + c, o);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- pwconv1 + GELU: two halves of 2D hidden channels ------------------------------------------------------------------------------
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 acc[2 * D / 16];
    tail_gemm<H, 2 * D / 16, BSTAGE, 4>(acc, p.w1 + (long long)half * 2 * D * D, D,
                                     [&](int k) { return *reinterpret_cast<const s16x8*>(tA + fr * PA + k + fq * 8); }, sB, wave, lane);
#pragma unroll
    for (int j = 0; j < 2 * D / 16; ++j) {
      const float bj = p.b1[half * 2 * D + j * 16 + fr];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        tH[(fq * 4 + r) * PH + half * 2 * D + j * 16 + fr] = from_f32<H>(act_c16<CFP_ACT_GELU>(acc[j][r] + bj));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- pwconv2 (K = 4D) + bias, staged in FLOAT32 over the (consumed) hidden tile; the residual is added on the way out, so the
  // output is rounded once, like the unfused GEMM epilogue does -----------------------------------------------------------------------
  constexpr int PF = D + 4;                                // floats per staged row (16 x PF x 4 <= 16 x PH x 2)
  float* tF = reinterpret_cast<float*>(tH);
  {
    f32x4 acc[D / 16];
    tail_gemm<H, D / 16, BSTAGE, 4>(acc, p.w2, 4 * D, [&](int k) { return *reinterpret_cast<const s16x8*>(tH + fr * PH + k + fq * 8); }, sB, wave, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tH are complete (tail_gemm consumed them)
#pragma unroll
    for (int j = 0; j < D / 16; ++j) {
      const float bj = p.b2[j * 16 + fr];
#pragma unroll
      for (int r = 0; r < 4; ++r) tF[(fq * 4 + r) * PF + j * 16 + fr] = acc[j][r] + bj;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < 16 * XCH; i += 64) {
    const int r = i / XCH, ch = i - r * XCH;
    const long long m = row0 + r;
    if (m < p.rows) {
      float a[8], b[8];
      Vec<float>::load(tF + r * PF + ch * 8, a);
      Vec<float>::load(tF + r * PF + ch * 8 + 4, a + 4);
      Vec<H>::load(p.xin + m * p.x_ld + ch * 8, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
      Vec<H>::store(p.out + m * p.out_ld + ch * 8, a);
    }
  }
}

template <typename H, int D>
int launch_lkpm(const LkpmP<H>& p, hipStream_t s) {
  constexpr size_t lds = 2 * (2 * D * 128) + 4 * (((D + 8) + (4 * D + 8)) * 16 * 2);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = lkpm_tail_kernel<H, D>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)cdiv(p.rows, 64)), dim3(256), lds, s, p);
  return 0;
}

}  // namespace

void cfp_tail16_debug_set(int value) { g_tail16_waves = value; }

int loftr_tail_x3_launch(const void* q, int q_ld, const float* kv, const float* ksum, const void* x, int x_ld, void* out, int out_ld,
                         const void* w_q, const void* w_merge, const void* w_mlp0, const void* w_mlp2, const float* ln1_g, const float* ln1_b,
                         const float* ln2_g, const float* ln2_b, float ln_eps, int NB, int Hq, int Wq, int qth, int qtw, float v_length,
                         float eps, int heads, int D, hipStream_t s);      // loftr_tail_x3.hip

int lkpm_tail_x3_launch(const void* t, int t_ld, const void* xin, int x_ld, void* out, int out_ld, const void* w1, const float* b1, const void* w2,
                        const float* b2, const float* ln_g, const float* ln_b, float ln_eps, int rows, int D, hipStream_t s);      // loftr_tail_x3.hip

extern "C" int cfp_loftr_tail(const void* q, int q_ld, const float* kv, const float* ksum, const void* x, int x_ld,
                              void* out, int out_ld, const void* w_q, const void* w_merge, const void* w_mlp0, const void* w_mlp2,
                              const float* ln1_g, const float* ln1_b, const float* ln2_g, const float* ln2_b, float ln_eps,
                              int NB, int Hq, int Wq, int qth, int qtw, float v_length, float eps, int heads, int D,
                              int dtype, cfp_stream_t stream) {
  if (dtype == CFP_F32X3) {      // float32 tensors, f16x3 matrix math, weights = cfp_pack_w_x3 operands: loftr_tail_x3.hip
    CFP_REQUIRE((q || w_q) && kv && ksum && x && out && w_merge && w_mlp0 && w_mlp2 && ln1_g && ln1_b && ln2_g && ln2_b, CFP_EINVAL,
                "cfp_loftr_tail: null pointer (q or w_q must be given)");
    CFP_REQUIRE(NB > 0 && Hq > 0 && Wq > 0 && qth > 0 && qtw > 0 && v_length > 0.f && (long long)NB * Hq * Wq < (1ll << 31), CFP_ESHAPE, "cfp_loftr_tail: bad grid");
    CFP_REQUIRE((D == 32 || D == 64 || D == 128) && (heads == 4 || heads == 8), CFP_ESHAPE, "cfp_loftr_tail: D must be 32/64/128 and heads 4/8");
    CFP_REQUIRE((w_q || (q_ld >= D && q_ld % 4 == 0)) && x_ld >= D && out_ld >= D && x_ld % 4 == 0 && out_ld % 4 == 0, CFP_ESHAPE,
                "cfp_loftr_tail: pitches must be >= D and multiples of 4");
    CFP_REQUIRE(aligned16(q) && aligned16(w_q) && aligned16(x) && aligned16(out) && aligned16(w_merge) && aligned16(w_mlp0) && aligned16(w_mlp2) &&
                    aligned16(kv), CFP_EINVAL, "cfp_loftr_tail: pointers must be 16-byte aligned");
    int rc3 = loftr_tail_x3_launch(q, q_ld, kv, ksum, x, x_ld, out, out_ld, w_q, w_merge, w_mlp0, w_mlp2, ln1_g, ln1_b, ln2_g, ln2_b, ln_eps, NB, Hq, Wq,
                                   qth, qtw, v_length, eps, heads, D, reinterpret_cast<hipStream_t>(stream));
    CFP_REQUIRE(rc3 == 0, CFP_EHIP, "cfp_loftr_tail: f16x3 launch failed");
    return cfp_check_launch("cfp_loftr_tail");
  }
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_loftr_tail: bf16 / f16 or CFP_F32X3 (the plain f32 parity mode uses the unfused kernels)");
  CFP_REQUIRE((q || w_q) && kv && ksum && x && out && w_merge && w_mlp0 && w_mlp2 && ln1_g && ln1_b && ln2_g && ln2_b, CFP_EINVAL,
              "cfp_loftr_tail: null pointer (q or w_q must be given)");
  CFP_REQUIRE(NB > 0 && Hq > 0 && Wq > 0 && qth > 0 && qtw > 0 && v_length > 0.f, CFP_ESHAPE, "cfp_loftr_tail: bad grid");
  CFP_REQUIRE((D == 32 || D == 64 || D == 128) && (heads == 4 || heads == 8), CFP_ESHAPE,
              "cfp_loftr_tail: D must be 32/64/128 and heads 4/8");
  CFP_REQUIRE((w_q || (q_ld >= D && q_ld % 8 == 0)) && x_ld >= D && out_ld >= D && x_ld % 8 == 0 && out_ld % 8 == 0, CFP_ESHAPE,
              "cfp_loftr_tail: pitches must be >= D and multiples of 8");
  CFP_REQUIRE(aligned16(q) && aligned16(w_q) && aligned16(x) && aligned16(out) && aligned16(w_merge) && aligned16(w_mlp0) && aligned16(w_mlp2) &&
                  aligned16(kv), CFP_EINVAL, "cfp_loftr_tail: pointers must be 16-byte aligned");
  CFP_REQUIRE((long long)NB * Hq * Wq < (1ll << 31), CFP_ESHAPE, "cfp_loftr_tail: too many rows");
  int rc = -2;
  auto run = [&](auto tag) {
    using H = decltype(tag);
    TailP<H> p;
    p.q = (const H*)q; p.kv = kv; p.ksum = ksum; p.x = (const H*)x; p.out = (H*)out;
    p.wq = (const H*)w_q; p.wm = (const H*)w_merge; p.w0 = (const H*)w_mlp0; p.w2 = (const H*)w_mlp2;
    p.g1 = ln1_g; p.b1 = ln1_b; p.g2 = ln2_g; p.b2 = ln2_b;
    p.q_ld = q_ld; p.x_ld = x_ld; p.out_ld = out_ld;
    p.rows = NB * Hq * Wq; p.Hq = Hq; p.Wq = Wq; p.qth = qth; p.qtw = qtw; p.ggy = cdiv(Hq, qth); p.ggx = cdiv(Wq, qtw);
    p.fwq = make_fastdiv((unsigned)Wq); p.fhq = make_fastdiv((unsigned)Hq); p.fqth = make_fastdiv((unsigned)qth); p.fqtw = make_fastdiv((unsigned)qtw);
    p.v_length = v_length; p.eps = eps; p.ln_eps = ln_eps;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (D == 32 && heads == 4) rc = launch_tail<H, 32, 4>(p, s);
    else if (D == 32 && heads == 8) rc = launch_tail<H, 32, 8>(p, s);
    else if (D == 64 && heads == 4) rc = launch_tail<H, 64, 4>(p, s);
    else if (D == 64 && heads == 8) rc = launch_tail<H, 64, 8>(p, s);
    else if (D == 128 && heads == 4) rc = launch_tail<H, 128, 4>(p, s);
    else if (D == 128 && heads == 8) rc = launch_tail<H, 128, 8>(p, s);
  };
  if (dtype == CFP_F16) run(f16_t{}); else run(bf16_t{});
  CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_loftr_tail: launch failed");
  return cfp_check_launch("cfp_loftr_tail");
}

extern "C" int cfp_lkpm_tail(const void* t, int t_ld, const void* xin, int x_ld, void* out, int out_ld, const void* w1, const float* b1,
                             const void* w2, const float* b2, const float* ln_g, const float* ln_b, float ln_eps, int rows, int D, int dtype,
                             cfp_stream_t stream) {
  if (dtype == CFP_F32X3) {      // float32 tensors, f16x3 matrix math, weights = cfp_pack_w_x3 operands: loftr_tail_x3.hip
    CFP_REQUIRE(t && xin && out && w1 && b1 && w2 && b2 && ln_g && ln_b, CFP_EINVAL, "cfp_lkpm_tail: null pointer");
    CFP_REQUIRE(rows > 0 && (D == 32 || D == 64 || D == 128), CFP_ESHAPE, "cfp_lkpm_tail: D must be 32/64/128");
    CFP_REQUIRE(t_ld >= D && x_ld >= D && out_ld >= D && t_ld % 4 == 0 && x_ld % 4 == 0 && out_ld % 4 == 0, CFP_ESHAPE,
                "cfp_lkpm_tail: pitches must be >= D and multiples of 4");
    CFP_REQUIRE(aligned16(t) && aligned16(xin) && aligned16(out) && aligned16(w1) && aligned16(w2), CFP_EINVAL, "cfp_lkpm_tail: pointers must be 16-byte aligned");
    int rc3 = lkpm_tail_x3_launch(t, t_ld, xin, x_ld, out, out_ld, w1, b1, w2, b2, ln_g, ln_b, ln_eps, rows, D, reinterpret_cast<hipStream_t>(stream));
    CFP_REQUIRE(rc3 == 0, CFP_EHIP, "cfp_lkpm_tail: f16x3 launch failed");
    return cfp_check_launch("cfp_lkpm_tail");
  }
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_lkpm_tail: bf16 / f16 or CFP_F32X3 (the plain f32 parity mode uses the unfused kernels)");
  CFP_REQUIRE(t && xin && out && w1 && b1 && w2 && b2 && ln_g && ln_b, CFP_EINVAL, "cfp_lkpm_tail: null pointer");
  CFP_REQUIRE(rows > 0 && (D == 32 || D == 64 || D == 128), CFP_ESHAPE, "cfp_lkpm_tail: D must be 32/64/128");
  CFP_REQUIRE(t_ld >= D && x_ld >= D && out_ld >= D && t_ld % 8 == 0 && x_ld % 8 == 0 && out_ld % 8 == 0, CFP_ESHAPE,
              "cfp_lkpm_tail: pitches must be >= D and multiples of 8");
  CFP_REQUIRE(aligned16(t) && aligned16(xin) && aligned16(out) && aligned16(w1) && aligned16(w2), CFP_EINVAL, "cfp_lkpm_tail: pointers must be 16-byte aligned");
  int rc = -2;
  auto run = [&](auto tag) {
    using H = decltype(tag);
    LkpmP<H> p;
    p.t = (const H*)t; p.xin = (const H*)xin; p.out = (H*)out; p.w1 = (const H*)w1; p.w2 = (const H*)w2;
    p.lg = ln_g; p.lb = ln_b; p.b1 = b1; p.b2 = b2; p.t_ld = t_ld; p.x_ld = x_ld; p.out_ld = out_ld; p.rows = rows; p.ln_eps = ln_eps;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (D == 32) rc = launch_lkpm<H, 32>(p, s);
    else if (D == 64) rc = launch_lkpm<H, 64>(p, s);
    else rc = launch_lkpm<H, 128>(p, s);
  };
  if (dtype == CFP_F16) run(f16_t{}); else run(bf16_t{});
  CFP_REQUIRE(rc == 0, CFP_EHIP, "cfp_lkpm_tail: launch failed");
  return cfp_check_launch("cfp_lkpm_tail");
}
