// Depthwise 3x3 (+ BN + activation + squeeze-excite channel sums), 16-bit storage: the software-pipelined form.
//
// Reference op: timm InvertedResidual `conv_dw -> bn2 -> act` (as restated in oracle/cfpnet_oracle.py, encoder()), whose
// output mean feeds SqueezeExcite.  24 launches per forward, 13-31 MB each at batch 8: HBM-bound by the roofline model
// (4.3 FLOP/byte), the kernel BASELINE.json's north_star puts the ">= 60 % of measured HBM roofline" target on.
//
// What bounded dw3x3_mfma_kernel (dwconv.hip, rounds 1-2): a launch is ONE wave of ~500 workgroups that all run
// load (5.7 us) -> compute (5-6 us, VALU-issue bound: the SiLU's v_exp + v_rcp) -> copy-out (1 us) in lock-step, so memory sits
// idle during the compute phase and the vector pipe during the load phase.  Here the three phases overlap INSIDE a workgroup:
//
//   * workgroup = (image, 64-channel block, range of RT output rows); its whole input image (rows_in x cols_in pixels, 128
//     bytes each) is requested at kernel start by LDS-DMA (`global_load_lds_dwordx4`: no VGPR staging, no ds_write, nothing
//     for a wave to wait on), in ROW order; halo pixels / channel tails read a 16-byte zero word instead (per-lane source);
//   * the rows are consumed in STEPS of S output rows: a step waits with a COUNTED `s_waitcnt vmcnt(n)` for exactly the DMA
//     instructions that carry its rows (every younger one -- the rows of later steps, the output stores of earlier steps --
//     stays in flight), one barrier, then computes while the later rows keep arriving;
//   * a step's results leave STRAIGHT FROM THE ACCUMULATOR REGISTERS (a lane holds 4 consecutive channels of one pixel: 8-byte
//     stores, 32 contiguous bytes per pixel and wave; the four waves of a workgroup complete each 128-byte line in L2), so the
//     stores of step s run beside the MFMA / SiLU work of the next units and the DMA of step s + 1 -- and nothing in the loop
//     WRITES LDS: hipcc's wait-count pass puts an `s_waitcnt vmcnt(0)` in front of every LDS store that follows an LDS-DMA
//     (write-after-write on LDS; seen in the first version of this kernel, which wrote results in place), which drains the queue;
//   * the per-lane constants (9 taps, BN scale / shift of the block's 64 channels) ride in the same DMA queue (a 2 KB header in
//     front of the image), so no register-destination global load exists in the kernel: the compiler has no reason to insert a
//     `vmcnt(0)` of its own, which would drain the queue.
//
// The arithmetic is that of dw3x3_mfma_kernel, instruction for instruction (diagonal-weight MFMA: A = diag(w_tap) per
// 16-channel group, B = the shifted input vectors straight from LDS, 5 MFMAs per 16 channels x 16 pixels; f32 BN + activation;
// per-lane channel sums reduced by a 16-lane butterfly), so outputs are bit-identical to it.
#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_dws_zero16[4] = {0u, 0u, 0u, 0u};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform RUNTIME n (the field is an immediate): a scalar jump table.  n above the table waits
// for the table's largest count, which is stricter, never weaker.
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
#define DWS_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    DWS_W(0) DWS_W(1) DWS_W(2) DWS_W(3) DWS_W(4) DWS_W(5) DWS_W(6) DWS_W(7) DWS_W(8) DWS_W(9) DWS_W(10) DWS_W(11) DWS_W(12) DWS_W(13)
    DWS_W(14) DWS_W(15) DWS_W(16) DWS_W(17) DWS_W(18) DWS_W(19) DWS_W(20) DWS_W(21) DWS_W(22) DWS_W(23) DWS_W(24) DWS_W(25) DWS_W(26)
    DWS_W(27) DWS_W(28) DWS_W(29) DWS_W(30) DWS_W(31) DWS_W(32) DWS_W(33) DWS_W(34) DWS_W(35) DWS_W(36) DWS_W(37) DWS_W(38) DWS_W(39)
    DWS_W(40) DWS_W(41) DWS_W(42) DWS_W(43) DWS_W(44) DWS_W(45) DWS_W(46) DWS_W(47)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef DWS_W
}

struct DwsP {
  const bf16_t* in; const bf16_t* w; const float* scale; const float* shift; bf16_t* out; float* partial;
  int in_ld, out_ld;
  int B, H, W, C, pad_t, pad_l, Ho, Wo, act;
  int RT;         // output rows per workgroup
  int nranges;    // row ranges per image = ceil(Ho / RT)
  int S;          // output rows per step
  int ncb;        // 64-channel blocks = ceil(C / 64)
  int ups;        // units (16 pixels x 16 channels per wave) per step
  int utab_off;   // byte offset of the unit table in LDS
  FastDiv dWo;    // flat output-pixel index -> (row, column)
};

constexpr int DWS_PP = 144;          // pixel pitch in the LDS image: 8 channel chunks of 16 bytes + one pad chunk (conflict spread)
constexpr int DWS_HDR = 2048;        // constants header: 72 tap chunks | 16 scale chunks | 16 shift chunks | zeros   (two DMA instructions)
constexpr int DWS_UN = 5;            // units in flight per batch

template <typename HT, int STRIDE, bool DIAG>
__global__ __launch_bounds__(256) void dw3x3_stream_kernel(DwsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* tile = lds + DWS_HDR;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  // XCD-aware order: channel blocks of one row range straddle each other's 128-byte lines, neighbouring ranges share halo rows
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = lin % p.ncb;
  const int rg = (lin / p.ncb) % p.nranges;
  const int b = lin / (p.ncb * p.nranges);
  const int CV = p.C >> 3;
  const int cv0 = cb * 8;
  const int ho_begin = rg * p.RT;
  const int rows = min(p.RT, p.Ho - ho_begin);
  const int rows_in = (rows - 1) * STRIDE + 3, cols_in = (p.Wo - 1) * STRIDE + 3;
  const int hi_base = ho_begin * STRIDE - p.pad_t, wi_base = -p.pad_l;
  const int nsteps = (rows + p.S - 1) / p.S;
  unsigned long long tk0 = 0, tk1 = 0, tr0 = 0, tkd = 0;
  float t_ready[4] = {0.f, 0.f, 0.f, 0.f}, t_done[4] = {0.f, 0.f, 0.f, 0.f};      // DIAG: per-step stamps (cycles since kernel start)
  if constexpr (DIAG) { tk0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- 1. every DMA instruction of the workgroup, in image order: instruction i covers LDS bytes [1024 i, 1024 i + 1024) --------
  const int nslots = rows_in * cols_in * 9;
  const int NI = 2 + ((nslots + 63) >> 6);                 // 2 header instructions + the image
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(g_dws_zero16);
  {
    if (wave < 2) {                                        // header: instruction `wave`, slot k = 64 wave + lane
      const int k = wave * 64 + lane;
      const unsigned char* src = zsrc;
      if (k < 72) {
        const int tap = k >> 3, ch = k & 7;
        if (cv0 + ch < CV) src = reinterpret_cast<const unsigned char*>(p.w + (long long)tap * p.C + (cv0 + ch) * 8);
      } else if (k < 104) {
        const int f4 = (k - 72) & 15;                      // 4 floats = 4 channels
        if (cv0 * 8 + f4 * 4 < p.C) src = reinterpret_cast<const unsigned char*>((k < 88 ? p.scale : p.shift) + cv0 * 8 + f4 * 4);
      }
      glds16(src, lds + wave * 1024);
    }
    // image instructions i = 2 + wave', 6 + wave', ... with wave' chosen so that the waves alternate over the WHOLE sequence
    // (instruction i is issued by wave i & 3): waves 0 / 1 issued the header as instructions 0 / 1
    int i = (wave < 2) ? wave + 4 : wave;
    int s = (i - 2) * 64 + lane;                           // slot of this lane in instruction i
    int pi = s / 9, ch = s - pi * 9;
    int ty = pi / cols_in, tx = pi - ty * cols_in;
    const bf16_t* img = p.in + (long long)b * p.H * p.W * p.in_ld;
    for (; i < NI; i += 4) {
      const int hi = hi_base + ty, wi = wi_base + tx;
      const bool ok = ch < 8 && cv0 + ch < CV && ty < rows_in && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(img + ((long long)hi * p.W + wi) * p.in_ld + (cv0 + ch) * 8) : zsrc;
      glds16(src, lds + i * 1024);
      // next instruction of this wave: slot += 256 = 28 pixels + 4 chunks
      ch += 4; tx += 28;
      if (ch >= 9) { ch -= 9; tx += 1; }
      while (tx >= cols_in) { tx -= cols_in; ++ty; }
    }
  }
  // DMA instructions this wave has issued with index <= L
  auto issued_upto = [&](int L) { return L >= wave ? ((L - wave) >> 2) + 1 : 0; };
  const int n_dma = issued_upto(NI - 1);

  // ---- 2. the unit table (no dependence on the DMA): for lane-pixel j of unit u of step s the window base in the LDS image and
  //         the byte offset of the output pixel from the workgroup's first output pixel, or -1 / unused ------------------------------
  const int nmain = p.Wo >> 4, tw = p.Wo & 15;
  int* utab = reinterpret_cast<int*>(lds + p.utab_off);           // [nsteps][ups][2][16]
  for (int e = tid; e < nsteps * p.ups * 16; e += 256) {
    const int jj = e & 15, u = (e >> 4) % p.ups, st = (e >> 4) / p.ups;
    int r, x; bool valid;
    if (u < p.S * nmain) {
      r = st * p.S + u / nmain;
      x = (u % nmain) * 16 + jj;
      valid = true;
    } else {                                   // gathered row tails of the step's rows, 16 per unit
      const int t = (u - p.S * nmain) * 16 + jj;
      const int rr = t / tw;                   // tw > 0 here
      r = st * p.S + rr;
      x = (p.Wo - tw) + (t - rr * tw);
      valid = rr < p.S;
    }
    valid = valid && r < rows;
    utab[(e >> 4) * 32 + jj] = valid ? ((r * STRIDE) * cols_in + x * STRIDE) * DWS_PP : -1;
    utab[(e >> 4) * 32 + 16 + jj] = valid ? (r * p.Wo + x) * p.out_ld * 2 : 0;
  }

  // ---- 3. steps ------------------------------------------------------------------------------------------------------------
  const int g = wave;                                       // 16-channel group of this wave
  const int cbase = (cv0 + 2 * g) * 8;
  const bool g_ok = cbase < p.C;
  s16x8 afr[5];
  int toff[5];
  float sc[4], sh[4];
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  int stores_issued = 0;                                    // global stores this wave has issued so far (wave-uniform)
  // this lane's 4 output channels (D fragment: rows 4q .. 4q+3 of column j) of the workgroup's first output pixel
  unsigned char* obase = reinterpret_cast<unsigned char*>(p.out + ((long long)(b * p.Ho + ho_begin) * p.Wo) * p.out_ld + cbase + 4 * q);

  with_act(p.act, [&](auto A) {
    for (int st = 0; st < nsteps; ++st) {
      // rows this step reads: image rows [0, row_hi); the DMA instructions that carry them: [0, L]
      const int row_hi = min(rows_in, (min((st + 1) * p.S, rows) - 1) * STRIDE + 3);
      const int L = 2 + ((row_hi * cols_in * 9 - 1) >> 6);
      wait_vmcnt_rt((n_dma - issued_upto(min(L, NI - 1))) + stores_issued);
      // raw barrier: a __syncthreads() carries a fence that drains the whole DMA queue.  lgkmcnt(0) as well: round 2's finding in
      // head_fused.hip about the safe hand-over form for LDS-DMA stages (and the unit table's ds_writes before step 0)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // the step's rows have landed for every wave
      asm volatile("" ::: "memory");
      if constexpr (DIAG) {
        const float t = (float)(__builtin_amdgcn_s_memtime() - tk0);
        if (st == 0) t_ready[0] = t; else if (st == 1) t_ready[1] = t; else if (st == 2) t_ready[2] = t; else if (st == 3) t_ready[3] = t;
      }
      if (st == 0) {
        if constexpr (DIAG) tk1 = __builtin_amdgcn_s_memtime();
        // per-lane constants out of the header
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
          const int tap = 2 * pr + (q >> 1);
          const bool on = g_ok && tap < 9 && (j >> 3) == (q & 1);
          const short wv = *reinterpret_cast<const short*>(lds + (min(tap, 8) * 8 + 2 * g + (j >> 3)) * 16 + (j & 7) * 2);
#pragma unroll
          for (int e = 0; e < 8; ++e) afr[pr][e] = (on && e == (j & 7)) ? wv : (short)0;
          const int tc = min(tap, 8);
          toff[pr] = ((tc / 3) * cols_in + tc % 3) * DWS_PP + (2 * g + (q & 1)) * 16;
        }
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(lds + 72 * 16 + (16 * g + 4 * q) * 4);
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(lds + 88 * 16 + (16 * g + 4 * q) * 4);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) { sc[r4] = g_ok ? s4[r4] : 1.f; sh[r4] = g_ok ? h4[r4] : 0.f; }
      }
      if (g_ok) {
        const int* ut = utab + st * p.ups * 32;
        for (int u0 = 0; u0 < p.ups; u0 += DWS_UN) {
          int ooff[DWS_UN]; bool ok[DWS_UN], present[DWS_UN]; f32x4 acc[DWS_UN];
          s16x8 bfr[DWS_UN][5];
#pragma unroll
          for (int t = 0; t < DWS_UN; ++t) {
            present[t] = false;
            if (u0 + t < p.ups) {                           // wave-uniform
              const int tb = ut[(u0 + t) * 32 + j];
              // a unit whose FIRST pixel is past the range's last row has no valid pixel at all (rows ascend inside a unit): skipped as
              // a whole -- its store instruction must not be counted, because an all-lanes-off store is branched around, not issued
              present[t] = __builtin_amdgcn_readfirstlane(tb) >= 0;
              if (present[t]) {
                ok[t] = tb >= 0;
                ooff[t] = ut[(u0 + t) * 32 + 16 + j];
                const int base = ok[t] ? tb : 0;            // invalid lanes read pixel 0 (finite data), results dropped
#pragma unroll
                for (int pr = 0; pr < 5; ++pr) bfr[t][pr] = *reinterpret_cast<const s16x8*>(tile + base + toff[pr]);
              }
            }
          }
#pragma unroll
          for (int t = 0; t < DWS_UN; ++t) {
            if (present[t]) {
              acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int pr = 0; pr < 5; ++pr) acc[t] = mfma16<HT>(afr[pr], bfr[t][pr], acc[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < DWS_UN; ++t) {
            if (present[t]) {
              float y[4];
              const float vf = ok[t] ? 1.f : 0.f;
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                y[r4] = p.act == 99 ? acc[t][r4] : act_c<decltype(A)::value>(acc[t][r4] * sc[r4] + sh[r4]);     // act 99: timing experiment
                csum[r4] = fmaf(y[r4], vf, csum[r4]);
              }
              uint2 pk;
              pk.x = pack2<HT>(y[0], y[1]);
              pk.y = pack2<HT>(y[2], y[3]);
              // ONE store instruction per present unit and wave (lane 0 is valid, so the instruction is always issued)
              if (ok[t]) *reinterpret_cast<uint2*>(obase + ooff[t]) = pk;
              ++stores_issued;
            }
          }
        }
      }
      if constexpr (DIAG) {
        const float t = (float)(__builtin_amdgcn_s_memtime() - tk0);
        if (st == 0) t_done[0] = t; else if (st == 1) t_done[1] = t; else if (st == 2) t_done[2] = t; else if (st == 3) t_done[3] = t;
      }
    }
  });
  if constexpr (DIAG) tkd = __builtin_amdgcn_s_memtime();

  if (p.partial != nullptr && g_ok) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) csum[r4] += __shfl_xor(csum[r4], o, 64);
    }
    if (j == 0) {
      float* dst = p.partial + ((long long)b * p.nranges + rg) * p.C + cbase + 4 * q;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) dst[r4] = csum[r4];
    }
  }
  if constexpr (DIAG) {
    if (p.partial != nullptr && tid == 0) {      // phase stamps AFTER the partial-sum area; they feed no output value
      const unsigned long long tk3 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
      float* dbg = p.partial + (long long)p.B * p.nranges * p.C + (long long)blockIdx.x * 16;
      dbg[0] = (float)(tk1 - tk0); dbg[1] = (float)(tkd - tk1); dbg[2] = (float)(tk3 - tkd);
      dbg[3] = (float)(tr0 & 0xffffff); dbg[4] = (float)(tr1 & 0xffffff); dbg[5] = 2.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { dbg[6 + k] = t_ready[k]; dbg[10 + k] = t_done[k]; }
    }
  }
}

}  // namespace

// Work decomposition: S output rows per step (so that a step's row tails pack into whole 16-pixel units), RT = k S rows per
// workgroup.  Cost model = halo read amplification x idle-CU penalty x pipeline depth penalty x unit packing efficiency,
// under the LDS cap that keeps two workgroups on a CU.
int g_dws_force_S = 0, g_dws_force_RT = 0;     // cfp_debug_set keys 7 / 8 (tools/dw_bench.py --sweep)

struct DwsPlan { int S, RT, nranges, ups; size_t lds; int utab_off; };

static bool dws_plan(int B, int Ho, int Wo, int C, int stride, DwsPlan& best) {
  const int cols_in = (Wo - 1) * stride + 3;
  const int ncb = cdiv(C, 64);
  const int nmain = Wo >> 4, tw = Wo & 15;
  double bc = 1e30;
  bool found = false;
  for (int S = 1; S <= 4 && S <= Ho; ++S) {
    if (g_dws_force_S && S != g_dws_force_S) continue;
    const int ups = S * nmain + (tw ? cdiv(S * tw, 16) : 0);
    const double pack = (double)ups * 16.0 / ((double)S * Wo);
    for (int k = 1; k <= 8; ++k) {
      const int RT = k * S;
      if (RT > Ho && k > 1) break;
      if (g_dws_force_RT && RT != g_dws_force_RT) continue;
      const int rows_in = (RT - 1) * stride + 3;
      const int nslots = rows_in * cols_in * 9;
      const size_t img = (size_t)cdiv(nslots, 64) * 1024;
      const int utab_off = DWS_HDR + (int)img;
      const size_t lds = utab_off + (size_t)k * ups * 128;
      if (lds > 72 * 1024) break;
      if (2 + cdiv(nslots, 64) > 4 * 40) break;             // DMA instructions per wave must fit the vmcnt field with room for the stores
      const int nranges = cdiv(Ho, RT);
      const long long blocks = (long long)ncb * B * nranges;
      const double halo = (double)rows_in / (RT * stride);
      const double fill = blocks >= 512 ? 1.0 : 512.0 / (double)blocks;
      const double pipe = k >= 3 ? 1.0 : (k == 2 ? 1.15 : 1.5);
      const double c = halo * fill * pipe * pack;
      if (c < bc) { bc = c; best = DwsPlan{S, RT, nranges, ups, lds, utab_off}; found = true; }
    }
  }
  return found;
}

void cfp_dws_debug_set(int key, int value) { if (key == 7) g_dws_force_S = value; else g_dws_force_RT = value; }

int cfp_dws_strips(int B, int Ho, int Wo, int C, int stride) {
  DwsPlan d;
  return dws_plan(B, Ho, Wo, C, stride, d) ? d.nranges : 0;
}

// -> CFP_OK, an error code, or 1 when the shape is not taken (caller falls back to dw3x3_mfma_kernel)
int cfp_dws_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld, float* partial,
                   int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, int dtype, cfp_stream_t stream,
                   const char* who) {
  DwsPlan d;
  if (!dws_plan(B, Ho, Wo, C, stride, d)) return 1;
  bool diag = act >= 100;
  if (diag) act -= 100;
  DwsP p;
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.out = (bf16_t*)out; p.partial = partial;
  p.in_ld = in_ld; p.out_ld = out_ld;
  p.B = B; p.H = H; p.W = W; p.C = C; p.pad_t = pad_t; p.pad_l = pad_l; p.Ho = Ho; p.Wo = Wo; p.act = act;
  p.RT = d.RT; p.nranges = d.nranges; p.S = d.S; p.ncb = cdiv(C, 64); p.ups = d.ups; p.utab_off = d.utab_off;
  p.dWo = make_fastdiv((unsigned)Wo);
  const long long blocks = (long long)p.ncb * B * d.nranges;
  CFP_REQUIRE(blocks < (1ll << 31), CFP_ESHAPE, std::string(who) + ": grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DWS_LAUNCH(HH, ST, DG)                                                                                                       \
  do {                                                                                                                               \
    static bool attr = false;                                                                                                        \
    if (!attr) {                                                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)dw3x3_stream_kernel<HH, ST, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); \
      if (e != hipSuccess) { cfp_set_error(std::string(who) + ": " + hipGetErrorString(e)); return CFP_EHIP; }                       \
      attr = true;                                                                                                                   \
    }                                                                                                                                \
    hipLaunchKernelGGL((dw3x3_stream_kernel<HH, ST, DG>), dim3((unsigned)blocks), dim3(256), d.lds, s, p);                           \
  } while (0)
#define DWS_ST(HH, DG) do { if (stride == 1) DWS_LAUNCH(HH, 1, DG); else DWS_LAUNCH(HH, 2, DG); } while (0)
  if (dtype == CFP_F16) { if (diag) DWS_ST(f16_t, true); else DWS_ST(f16_t, false); }
  else { if (diag) DWS_ST(bf16_t, true); else DWS_ST(bf16_t, false); }
#undef DWS_ST
#undef DWS_LAUNCH
  return cfp_check_launch(who);
}
