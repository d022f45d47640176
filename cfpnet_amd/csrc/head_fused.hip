// The adaptive-bins head as ONE kernel:  depth_head.conv3x3 (128 -> 128, the network's largest GEMM: 25 % of all MACs)
//   -> conv_out 1x1 (128 -> 256 bin logits) -> per-pixel softmax -> prob (NCHW) and pred = sum_k prob_k * centre_k.
// Reference: /root/reference/src/models/decoder.py:22-27 (range attention maps `ram = conv3x3(x)`), deltar.py:18-19,51-61.
//
// Before: igemm2<128,128> wrote `ram` (157 MB at batch 8) and bin_head_fused read it back.  Here `ram` never leaves the
// registers.  The trick is to run both GEMMs TRANSPOSED (channels / bins on the MFMA row axis, pixels on the column axis):
//
//   GEMM1  ram^T[ch][px]    = W3[ch][K = 9*128] * im2col^T[K][px]       acc1[i][j][r] = ram(ch = 16 i + 4 fq + r, px = 16 j + fr)
//   GEMM2  logit^T[bin][px] = Wout[bin][ch]     * ram^T[ch][px]
//
// The B operand of a 16x16x32 MFMA wants, in lane (fr, fq), eight K-values of column fr.  Which eight is free as long as the A
// operand uses the same K order -- and the accumulators of GEMM1 already hold, for pixel fr, the channels 16 i + 4 fq + {0..3}.
// Two accumulator tiles (i = 2 kb, 2 kb + 1) packed to 16 bit ARE the B fragment of K-block kb of GEMM2 in the channel order
// c(fq, e) = 32 kb + 16 (e >> 2) + 4 fq + (e & 3); the host stores Wout with its K axis permuted the same way (ops.permute_wout).
// No LDS round trip, no shuffle.  With every wave owning all 128 channels of 32 pixels (tiles 8 x 2) there is no cross-wave
// reduction either, and the softmax over the 256 bins of a pixel is a register reduction + two lane-group shuffles.
//
//   * main loop = the gen-2 implicit GEMM (conv_igemm2.hip): K-step 64, LDS-DMA staging, XOR-swizzled 128-byte rows, one raw
//     barrier per K-step, two stages -- but Cin = 128 makes the tap of a K-step wave-uniform, so all im2col arithmetic is scalar:
//     per lane only a byte offset and a 9-bit tap-validity mask per pixel row.  Loads are `buffer_load_dwordx4 ... lds` through
//     descriptors: the tap offset travels in soffset (an SGPR), halo / tail lanes get an out-of-range voffset and the hardware
//     writes zeros (probed: tools/probes/buffer_lds_oob.hip) -> 3 VALU per A-row DMA, 0 per weight DMA (the generic kernel
//     spends ~12: 64-bit pointer select + range predicates).
//   * while the last K-step computes, the stage that is already free receives the first half of Wout by DMA; the second half
//     follows into the other stage and lands during the first half of GEMM2.
//   * optional exactness islands (flags): Wout as hi + lo 16-bit planes and ram as hi + lo B fragments (2 extra MFMAs per
//     GEMM2 tile: +11 % / +22 % matrix work of a kernel that is not matrix-bound) remove the rounding of `ram` and of the
//     conv_out weights from the logits.
//   * prob leaves through LDS transposed to [bin][pixel] so the NCHW write is 16-byte vectors along the pixel axis.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int HF_BM = 128;              // pixels per workgroup
constexpr int HF_C = 128;               // channels in / out of the 3x3 conv
constexpr int HF_NB = 256;              // bins
constexpr int HF_K = 9 * HF_C;
constexpr int HF_STEPS = HF_K / 64;     // 18 K-steps
constexpr int HF_STAGE = (HF_BM + HF_C) * 128;          // 32 KB: pixel rows + weight rows of one K-step
constexpr int HF_PPITCH = HF_BM + 8;                    // prob staging pitch (elements)
constexpr int HF_PBYTES = HF_NB * HF_PPITCH * 2;        // 69 632 B >= 2 stages: operand stages, later the prob staging tile
constexpr int HF_CONST = 4 * 256 * 4;                   // 4 KB behind it: conv_out bias | conv3x3 scale, shift | bin centres of the tile's (two) images
constexpr int HF_LDS = HF_PBYTES + HF_CONST;            // 72 KB: two workgroups per CU
constexpr unsigned HF_OOB = 0x80000000u;
static_assert(HF_PBYTES >= 2 * HF_STAGE, "prob staging must cover both operand stages");

struct HeadP {
  const void* x; const void* w3; const float* scale3; const float* shift3;
  const void* wout; const float* bias_out; const float* centers;
  void* prob; float* pred; void* ram_out;
  int x_ld, B, H, W, M, HW;
  int wout_lo;      // Wout carries a second plane (lo = W - hi) right after the first
  int ram_lo;       // feed ram to GEMM2 as hi + lo
  int probe;        // timing probes (tools/head_bench.py --probe): 1 = descriptors with zero records (no fetch), 2 = stop after GEMM1, 4 = stop after GEMM2
};

#ifdef HF_GLDS
__device__ __attribute__((aligned(16))) unsigned int g_hf_zero[4] = {0u, 0u, 0u, 0u};
#endif
// Stage hand-over: this wave's LDS-DMA loads have landed (vmcnt) AND its LDS-side operations have drained (lgkmcnt), then the
// workgroup barrier.  With `s_waitcnt vmcnt(0)` + a raw s_barrier alone the kernel produced a wrong 32-pixel group (one wave's
// pixels) about once per 640 forwards when three captured forwards ran concurrently -- never alone, never with same-input repeats
// (tools/probes/inflight_race*.py, race_ab.sh: 5 / 3200 wrong results vs 0 / 3200 with this form): the zero words the hardware
// writes for out-of-range lanes of a `buffer_load ... lds` are apparently not covered by vmcnt.  HF_RAW_BARRIER restores the old form.
#ifdef HF_RAW_BARRIER
#define HF_BARRIER() __builtin_amdgcn_s_barrier()
template <int N> __device__ __forceinline__ void hf_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#else
#define HF_BARRIER() __syncthreads()
template <int N> __device__ __forceinline__ void hf_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
#endif

template <typename H, bool RAMLO>
__global__ __launch_bounds__(256, 2) void depth_head_fused_kernel(HeadP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;                      // logical 16-byte chunk of the 128-byte K-step row this lane fetches
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = bid * HF_BM;

  // descriptors (wave-uniform: kernel arguments only).  x is addressed relative to one row + one pixel BEFORE its start so
  // that the (signed) tap offset becomes a non-negative soffset; a valid tap never dereferences anything before x.
  const unsigned guard = (unsigned)(p.W + 1) * (unsigned)p.x_ld * 2u;
  const unsigned xbytes = (unsigned)p.M * (unsigned)p.x_ld * 2u;
  const int live = (p.probe & 1) ? 0 : 1;                // probe: zero records = every load is dropped by the range check
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.x - guard), 0, (int)(xbytes + guard) * live, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, HF_C * HF_K * 2 * live, 0x00020000);
  const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.wout, 0, HF_NB * HF_C * 2 * (p.wout_lo ? 2 : 1) * live, 0x00020000);

  // ---- per-lane row bookkeeping: 4 pixel rows and 4 weight rows per K-step ----------------------------------------------
  unsigned a_off[4], a_mask[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (i * 4 + wave) * 8 + rsub;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int wo = mm % p.W;
    const int t = mm / p.W;
    const int ho = t % p.H;
    a_off[i] = (unsigned)mm * (unsigned)p.x_ld * 2u + (unsigned)lc * 16u;
    unsigned mk = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        if (ok && (unsigned)(ho + kh - 1) < (unsigned)p.H && (unsigned)(wo + kw - 1) < (unsigned)p.W) mk |= 1u << (kh * 3 + kw);
    a_mask[i] = mk;
    b_off[i] = (unsigned)((i * 4 + wave) * 8 + rsub) * (unsigned)(HF_K * 2) + (unsigned)lc * 16u;
  }

  auto issue = [&](int ks, int buf) {
    unsigned char* sA = smem + buf * HF_STAGE;             // pixel rows
    unsigned char* sB = sA + HF_BM * 128;                  // weight rows
    const int tap = ks >> 1;                               // Cin = 128 = two K-steps per tap: all scalar
    const int kh = tap / 3, kw = tap - kh * 3;
#ifdef HF_GLDS
    {   // A/B variant: global_load_lds with a 64-bit pointer select against a zero word (the gen-2 GEMM's loader)
      using gptr_t = const __attribute__((address_space(1))) void*;
      const unsigned char* xt = reinterpret_cast<const unsigned char*>(p.x) + (long long)(((kh - 1) * p.W + (kw - 1)) * p.x_ld * 2 + (ks & 1) * 128);
      const unsigned char* wk = reinterpret_cast<const unsigned char*>(p.w3) + ks * 128;
      const unsigned bit2 = 1u << tap;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned char* src = (a_mask[i] & bit2) ? xt + a_off[i] : reinterpret_cast<const unsigned char*>(g_hf_zero);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(sA + (i * 4 + wave) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(wk + b_off[j]), (lds_ptr_t)(sB + (j * 4 + wave) * 1024), 16, 0, 0);
      return;
    }
#endif
    const unsigned soff = (unsigned)((kh * p.W + kw) * p.x_ld * 2 + (ks & 1) * 128);   // + guard - (W + 1) * ld * 2 = tap (kh-1, kw-1)
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned v = (a_mask[i] & bit) ? a_off[i] : HF_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(sA + (i * 4 + wave) * 1024), 16, (int)v, (int)soff, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sB + (j * 4 + wave) * 1024), 16, (int)b_off[j], ks * 128, 0, 0);
  };
  // one half (64 permuted channels) of Wout -> a 32 KB stage: 256 rows of 128 bytes, 8 DMA instructions per wave
  auto issue_wout = [&](int half, int buf, int plane) {
    unsigned char* dst = smem + buf * HF_STAGE;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int g = q * 4 + wave;                          // 8-row group
      const unsigned v = (unsigned)(g * 8 + rsub) * (unsigned)(HF_C * 2) + (unsigned)lc * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, (lds_ptr_t)(dst + g * 1024), 16, (int)v, half * 128 + plane * (HF_NB * HF_C * 2), 0, 0);
    }
  };

  f32x4 acc1[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  // the per-channel / per-bin vectors the later phases need go to LDS now (their global latency hides behind K-step 0; read back
  // with ds_read_b128 when needed -- loading them where they are used cost three exposed global round trips per workgroup)
  float* sK = reinterpret_cast<float*>(smem + HF_PBYTES);    // [0,256) bias_out | [256,384) scale3 | [384,512) shift3 | [512,768) centres(b0) | [768,1024) centres(b1)
  const int img0 = m0 / p.HW;
  {
    const int img1 = min(p.B - 1, (min(m0 + HF_BM, p.M) - 1) / p.HW);
    sK[tid] = p.bias_out[tid];
    if (tid < 128) sK[256 + tid] = p.scale3 ? p.scale3[tid] : 1.f;
    else sK[256 + tid] = p.shift3 ? p.shift3[tid - 128] : 0.f;
    sK[512 + tid] = p.centers[(long long)img0 * HF_NB + tid];
    sK[768 + tid] = p.centers[(long long)img1 * HF_NB + tid];
  }
  for (int ks = 0; ks < HF_STEPS; ++ks) {
    const int buf = ks & 1;
    hf_wait_vmcnt<0>();
    HF_BARRIER();                                          // stage `buf` landed for every wave; stage buf^1 fully consumed
    asm volatile("" ::: "memory");
    if (ks + 1 < HF_STEPS) issue(ks + 1, buf ^ 1);
    else issue_wout(0, buf ^ 1, 0);                        // the free stage receives Wout[:, first 64 channels]
    const unsigned char* cP = smem + buf * HF_STAGE + (wave * 32) * 128;    // this wave's 32 pixel rows
    const unsigned char* cW = smem + buf * HF_STAGE + HF_BM * 128;          // all 128 weight rows
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int pc = ((s * 4 + fq) ^ (fr & 7)) * 16;
      s16x8 wf[8], pf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) pf[j] = *reinterpret_cast<const s16x8*>(cP + (j * 16 + fr) * 128 + pc);
#pragma unroll
      for (int i = 0; i < 8; ++i) wf[i] = *reinterpret_cast<const s16x8*>(cW + (i * 16 + fr) * 128 + pc);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = mfma16<H>(wf[i], pf[j], acc1[i][j]);
    }
  }
  // HF_STEPS is even: the last K-step used stage 1, Wout half 0 is (being) written to stage 0
  hf_wait_vmcnt<0>();
  HF_BARRIER();                                            // stage 1 consumed by every wave; Wout half 0 landed
  asm volatile("" ::: "memory");
  issue_wout(1, 1, 0);                                     // lands while GEMM2 runs on half 0
  if (p.probe & 2) {
    hf_wait_vmcnt<0>();
    if (tid == 0 && acc1[0][0][0] == 123.456f) p.pred[0] = acc1[7][1][3];
    return;
  }

  // ---- ram = scale * acc + shift (depth_head.conv3x3 has a bias and no activation) -> B fragments of GEMM2 -------------------
  s16x8 bh[4][2], bl[RAMLO ? 4 : 1][2];
  {
    H* ram_out = reinterpret_cast<H*>(p.ram_out);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float sc[2][4], sh[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ch = (2 * kb + u) * 16 + fq * 4;
        Vec<float>::load(sK + 256 + ch, sc[u]);
        Vec<float>::load(sK + 384 + ch, sh[u]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[u * 4 + r] = acc1[2 * kb + u][j][r] * sc[u][r] + sh[u][r];
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const H h0 = from_f32<H>(v[2 * e]), h1 = from_f32<H>(v[2 * e + 1]);
          hi[e] = (uint32_t)to_bits<H>(h0) | ((uint32_t)to_bits<H>(h1) << 16);
          lo[e] = pack2<H>(v[2 * e] - to_f32<H>(h0), v[2 * e + 1] - to_f32<H>(h1));
        }
        bh[kb][j] = __builtin_bit_cast(s16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
        if constexpr (RAMLO) bl[kb][j] = __builtin_bit_cast(s16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
        if (ram_out) {                                     // test / tap hook: ram as the unfused path stores it
          const int m = m0 + wave * 32 + j * 16 + fr;
          if (m < p.M) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              uint2 w2 = {hi[2 * u], hi[2 * u + 1]};
              *reinterpret_cast<uint2*>(ram_out + (long long)m * HF_C + (2 * kb + u) * 16 + fq * 4) = w2;
            }
          }
        }
      }
    }
  }

  __builtin_amdgcn_sched_barrier(0);      // acc1 is dead from here: keep the 128 registers of acc2 from being set up above this line

  // ---- GEMM2: logit^T[bin][px], accumulators start at the conv_out bias ----------------------------------------------------------
  f32x4 acc2[16][2];
#pragma unroll
  for (int ti = 0; ti < 16; ++ti) {
    float b4[4];
    Vec<float>::load(sK + ti * 16 + fq * 4, b4);
#pragma unroll
    for (int j = 0; j < 2; ++j) acc2[ti][j] = f32x4{b4[0], b4[1], b4[2], b4[3]};
  }
  auto gemm2_half = [&](int half, const unsigned char* base, bool use_lo_b) {
#pragma unroll
    for (int kl = 0; kl < 2; ++kl) {
      const int kb = half * 2 + kl;
      const int pc = ((kl * 4 + fq) ^ (fr & 7)) * 16;
#pragma unroll
      for (int ti = 0; ti < 16; ++ti) {
        const s16x8 af = *reinterpret_cast<const s16x8*>(base + (ti * 16 + fr) * 128 + pc);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc2[ti][j] = mfma16<H>(af, bh[kb][j], acc2[ti][j]);
          if constexpr (RAMLO) { if (use_lo_b) acc2[ti][j] = mfma16<H>(af, bl[kb][j], acc2[ti][j]); }
        }
      }
    }
  };
  constexpr bool ram_lo = RAMLO;
  gemm2_half(0, smem, ram_lo);
  hf_wait_vmcnt<0>();
  HF_BARRIER();                                            // Wout half 1 landed; every wave is done with half 0 (stage 0)
  asm volatile("" ::: "memory");
  if (p.wout_lo) issue_wout(0, 0, 1);                      // lo plane of half 0 -> stage 0, during GEMM2 on half 1
  gemm2_half(1, smem + HF_STAGE, ram_lo);
  if (p.wout_lo) {                                         // Wout_lo * ram_hi: the lo plane only meets the hi fragments (lo * lo ~ 2^-22)
    hf_wait_vmcnt<0>();
    HF_BARRIER();
    asm volatile("" ::: "memory");
    issue_wout(1, 1, 1);
    gemm2_half(0, smem, false);
    hf_wait_vmcnt<0>();
    HF_BARRIER();
    asm volatile("" ::: "memory");
    gemm2_half(1, smem + HF_STAGE, false);
  }
  if (p.probe & 4) {
    if (tid == 0 && acc2[0][0][0] == 123.456f) p.pred[0] = acc2[15][1][3];
    return;
  }
  __syncthreads();                                         // LDS becomes the prob staging tile

  // ---- softmax over the 256 bins of each pixel + expectation ---------------------------------------------------------------------
  H* sP = reinterpret_cast<H*>(smem);                      // [256 bins][HF_PPITCH]
  H* prob = reinterpret_cast<H*>(p.prob);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pl = wave * 32 + j * 16 + fr;                // pixel within the tile
    const int m = m0 + pl;
    const int bimg = min(m, p.M - 1) / p.HW;
    float mx = -3.0e38f;
#pragma unroll
    for (int ti = 0; ti < 16; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc2[ti][j][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float s = 0.f;
#pragma unroll
    for (int ti = 0; ti < 16; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(acc2[ti][j][r] - mx); acc2[ti][j][r] = e; s += e; }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float inv = 1.f / s;
    float dot = 0.f;
#pragma unroll
    for (int ti = 0; ti < 16; ++ti) {
      float cen[4];
      Vec<float>::load(sK + (bimg == img0 ? 512 : 768) + ti * 16 + fq * 4, cen);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = acc2[ti][j][r] * inv;
        dot = fmaf(pr, cen[r], dot);
        if (prob) sP[(ti * 16 + fq * 4 + r) * HF_PPITCH + pl] = from_f32<H>(pr);
      }
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (fq == 0 && m < p.M) p.pred[m] = dot;
  }
  if (!prob) return;
  __syncthreads();
  // NCHW copy-out: 256 bin rows x 16 chunks of 8 pixels (HW % 8 == 0: a chunk stays inside one image)
#pragma unroll 4
  for (int q = tid; q < HF_NB * (HF_BM / 8); q += 256) {
    const int n = q >> 4, ch = q & 15;
    const int m = m0 + ch * 8;
    if (m >= p.M) continue;
    const int b = m / p.HW, hw = m - b * p.HW;
    *reinterpret_cast<u32x4*>(prob + ((long long)b * HF_NB + n) * p.HW + hw) = *reinterpret_cast<const u32x4*>(sP + n * HF_PPITCH + ch * 8);
  }
}

}  // namespace

extern "C" int cfp_depth_head_fused(const void* x, int x_ld, const void* w3, const float* scale3, const float* shift3,
                                    const void* wout_perm, const float* bias_out, const float* centers, void* prob, float* pred,
                                    void* ram_out, int B, int H, int W, int flags, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_depth_head_fused: bf16/f16 only (f32 parity mode runs conv + bin_softmax)");
  CFP_REQUIRE(x && w3 && wout_perm && bias_out && centers && pred, CFP_EINVAL, "cfp_depth_head_fused: null pointer");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && x_ld >= HF_C && x_ld % 8 == 0, CFP_ESHAPE, "cfp_depth_head_fused: bad shape (128 channels, x_ld % 8 == 0)");
  const long long M = (long long)B * H * W;
  CFP_REQUIRE((H * W) % 16 == 0 && H * W >= HF_BM, CFP_ESHAPE, "cfp_depth_head_fused: H*W must be a multiple of 16 and at least 128");
  CFP_REQUIRE((M + W + 1) * x_ld * 2 < (1ll << 31) - 4096, CFP_ESHAPE, "cfp_depth_head_fused: input larger than 2 GB");
  CFP_REQUIRE(aligned16(x) && aligned16(w3) && aligned16(wout_perm) && aligned16(prob) && aligned16(bias_out) && aligned16(centers) &&
                  aligned16(scale3) && aligned16(shift3) && aligned16(ram_out), CFP_EINVAL, "cfp_depth_head_fused: pointers must be 16-byte aligned");
  CFP_REQUIRE((flags & ~(3 | (7 << 8))) == 0, CFP_EINVAL, "cfp_depth_head_fused: unknown flags");
  HeadP p;
  p.x = x; p.w3 = w3; p.scale3 = scale3; p.shift3 = shift3; p.wout = wout_perm; p.bias_out = bias_out; p.centers = centers;
  p.prob = prob; p.pred = pred; p.ram_out = ram_out;
  p.x_ld = x_ld; p.B = B; p.H = H; p.W = W; p.M = (int)M; p.HW = H * W;
  p.wout_lo = (flags & CFP_HEAD_WOUT_HILO) ? 1 : 0; p.ram_lo = (flags & CFP_HEAD_RAM_HILO) ? 1 : 0; p.probe = (flags >> 8) & 7;
  const int grid = cdiv(M, HF_BM);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define HF_LAUNCH(T, RL)                                                                                                              \
  do {                                                                                                                                \
    static bool attr = false;                                                                                                         \
    if (!attr) {                                                                                                                      \
      if (hipFuncSetAttribute((const void*)depth_head_fused_kernel<T, RL>, hipFuncAttributeMaxDynamicSharedMemorySize, HF_LDS) != hipSuccess) { \
        cfp_set_error("cfp_depth_head_fused: cannot set the LDS size");                                                               \
        return CFP_EHIP;                                                                                                              \
      }                                                                                                                               \
      attr = true;                                                                                                                    \
    }                                                                                                                                 \
    hipLaunchKernelGGL((depth_head_fused_kernel<T, RL>), dim3(grid), dim3(256), HF_LDS, s, p);                                        \
  } while (0)
  if (dtype == CFP_F16) { if (p.ram_lo) HF_LAUNCH(f16_t, true); else HF_LAUNCH(f16_t, false); }
  else { if (p.ram_lo) HF_LAUNCH(bf16_t, true); else HF_LAUNCH(bf16_t, false); }
#undef HF_LAUNCH
  return cfp_check_launch("cfp_depth_head_fused");
}
