// Depthwise 3x3 (+ BN + activation + squeeze-excite channel sums), 16-bit storage: the persistent, software-pipelined form.
//
// Reference op: timm InvertedResidual `conv_dw -> bn2 -> act` (as restated in oracle/cfpnet_oracle.py, encoder()), whose
// output mean feeds SqueezeExcite.  24 launches per forward, 13-31 MB each at batch 8: HBM-bound by the roofline model
// (4.3 FLOP/byte), the kernel BASELINE.json's north_star puts the ">= 60 % of measured HBM roofline" target on.
//
// What bounded dw3x3_mfma_kernel (dwconv.hip, rounds 1-2): a launch is ONE wave of ~500 workgroups that all run
// load (5.7 us) -> compute (5-6 us, VALU-issue + LDS-read bound: the SiLU's v_exp + v_rcp, 20 bytes of LDS reads per output)
// -> copy-out (1 us) in lock-step, so memory sits idle during the compute phase and the vector pipe during the load phase;
// and the workgroup count (104 image x channel-block columns x a few row ranges) never matches the 256 CUs.  Here:
//
//   * PERSISTENT and exactly balanced: one 512-thread workgroup per CU; the output rows of all (image, 64-channel block)
//     columns form one list of NCOL x Ho rows and workgroup w owns rows [w TOT / NW, (w + 1) TOT / NW) of it -- at most three
//     SEGMENTS (row ranges of consecutive columns), 6-12 rows in all at batch 8;
//   * the whole input of the workgroup (every segment's rows + halo, 128 bytes per pixel) is requested at kernel start by
//     LDS-DMA (`global_load_lds_dwordx4`: no VGPR staging, no ds_write, nothing for a wave to wait on), in consumption
//     order; halo pixels / channel tails read a 16-byte zero word instead (per-lane source address);
//   * the rows are consumed in STEPS of S output rows: a step waits with a COUNTED `s_waitcnt vmcnt(n)` for exactly the DMA
//     instructions that carry its rows (every younger one -- the rows of later steps, the output stores of earlier steps --
//     stays in flight), one barrier, then computes while the later rows keep arriving: the first step needs 4 of ~16 rows;
//   * results leave STRAIGHT FROM THE ACCUMULATOR REGISTERS (a lane holds 4 consecutive channels of one pixel: 8-byte stores,
//     32 contiguous bytes per pixel and wave; the waves of a workgroup complete each 128-byte line in L2), so the stores of a
//     unit run beside the MFMA / SiLU work of the next ones -- and nothing in the loop WRITES LDS: hipcc's wait-count pass puts an
//     `s_waitcnt vmcnt(0)` in front of every LDS store that follows an LDS-DMA (write-after-write on LDS; seen in the first
//     version of this kernel, which wrote results in place), which drains the queue;
//   * the per-lane constants (9 taps, BN scale / shift of the segment's 64 channels) ride in the same DMA queue (a 2 KB header in
//     front of each segment's image), so no register-destination global load exists in the kernel: the compiler has no reason to
//     insert a `vmcnt(0)` of its own.
//
// The arithmetic is that of dw3x3_mfma_kernel, instruction for instruction (diagonal-weight MFMA: A = diag(w_tap) per
// 16-channel group, B = the shifted input vectors straight from LDS, 5 MFMAs per 16 channels x 16 pixels; f32 BN + activation;
// per-lane channel sums reduced by a 16-lane butterfly), so outputs are bit-identical to it.
#include <algorithm>

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
  int S;          // output rows per step
  int ncb;        // 64-channel blocks = ceil(C / 64)
  int ups;        // units (16 pixels x 16 channels) per step and channel group
  int nslot;      // partial-sum slots per column = the most workgroups any column is split over
  int TOT;        // B * ncb * Ho output rows in all
  int utab_off;   // byte offset of the unit table in LDS (behind the largest workgroup image)
};

constexpr int DWS_PP = 144;          // pixel pitch in the LDS image: 8 channel chunks of 16 bytes + one pad chunk (conflict spread)
constexpr int DWS_HDR = 2048;        // constants header: 72 tap chunks | 16 scale chunks | 16 shift chunks | zeros   (two DMA instructions)
constexpr int DWS_UN = 3;            // units in flight per batch and wave
constexpr int DWS_MAXSEG = 3;

// Segment k of a workgroup's row list [G0, G1): a row range of one (image, channel block) column.
struct DwsSeg { int col, r0, nr, rows_in, ni, ldsoff, i0, fs0, nsteps; };

template <int STRIDE>
__device__ __forceinline__ DwsSeg dws_segment(int k, int G0, int G1, int Ho, int cols_in, int S) {
  DwsSeg s{0, 0, 0, 0, 0, 0, 0, 0, 0};
  int g = G0;
  for (int t = 0; t <= k; ++t) {
    s.col = g / Ho;
    s.r0 = g - s.col * Ho;
    s.nr = min(Ho - s.r0, G1 - g);
    s.rows_in = (s.nr - 1) * STRIDE + 3;
    s.ni = 2 + ((s.rows_in * cols_in * 9 + 63) >> 6);
    s.nsteps = (s.nr + S - 1) / S;
    if (t < k) { g += s.nr; s.ldsoff += s.ni * 1024; s.i0 += s.ni; s.fs0 += s.nsteps; }
  }
  return s;
}

template <typename HT, int STRIDE, bool DIAG>
__global__ __launch_bounds__(512) void dw3x3_stream_kernel(DwsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
  const int j = lane & 15, q = lane >> 4;
  const int g = wave & 3, hf = wave >> 2;                        // 16-channel group, which half of a step's units
  const int NW = gridDim.x;
  // XCD-aware order: neighbouring row lists share halo rows and 128-byte lines (adjacent channel blocks): one L2 for them
  const int w = xcd_remap(blockIdx.x, NW);
  const int G0 = (int)((long long)w * p.TOT / NW), G1 = (int)((long long)(w + 1) * p.TOT / NW);
  const int CV = p.C >> 3;
  const int cols_in = (p.Wo - 1) * STRIDE + 3;
  const int wi_base = -p.pad_l;
  int nseg = 0;
  for (int gg = G0; gg < G1 && nseg < DWS_MAXSEG; ++nseg) gg += min(p.Ho - gg % p.Ho, G1 - gg);
  unsigned long long tk0 = 0, tr0 = 0;
  float t_ready[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, t_done[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // DIAG: per-step stamps
  if constexpr (DIAG) { tk0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- 1. every DMA instruction of the workgroup, in consumption order.  Instruction i (global index over the segments) is issued
  //         by wave i & 7 and fills the LDS kilobyte i --------------------------------------------------------------------------------
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(g_dws_zero16);
  int ni_total = 0;
  for (int k = 0; k < nseg; ++k) {
    const DwsSeg sg = dws_segment<STRIDE>(k, G0, G1, p.Ho, cols_in, p.S);
    const int b = sg.col / p.ncb, cv0 = (sg.col - b * p.ncb) * 8;
    const int hi_base = sg.r0 * STRIDE - p.pad_t;
    unsigned char* base = lds + sg.ldsoff;
    int il = (wave - sg.i0) & 7;                           // first local instruction of this wave
    if (il < 2) {                                          // header instruction il: slot kk = 64 il + lane
      const int kk = il * 64 + lane;
      const unsigned char* src = zsrc;
      if (kk < 72) {
        const int tap = kk >> 3, ch = kk & 7;
        if (cv0 + ch < CV) src = reinterpret_cast<const unsigned char*>(p.w + (long long)tap * p.C + (cv0 + ch) * 8);
      } else if (kk < 104) {
        const int f4 = (kk - 72) & 15;                     // 4 floats = 4 channels
        if (cv0 * 8 + f4 * 4 < p.C) src = reinterpret_cast<const unsigned char*>((kk < 88 ? p.scale : p.shift) + cv0 * 8 + f4 * 4);
      }
      glds16(src, base + il * 1024);
      il += 8;
    }
    if (il < sg.ni) {
      const int s = (il - 2) * 64 + lane;                  // slot of this lane: pixel s / 9, chunk s % 9 (chunk 8 = pad)
      int pi = s / 9, ch = s - pi * 9;
      int ty = pi / cols_in, tx = pi - ty * cols_in;
      // everything per instruction is 32-bit adds and compares: the valid window of (ty, tx), a running byte offset into the image
      const int ty_lo = max(0, -hi_base), ty_hi = min(sg.rows_in, p.H - hi_base);
      const int tx_lo = p.pad_l, tx_hi = min(cols_in, p.W + p.pad_l);
      const int chv = min(8, CV - cv0);                    // valid chunks of this channel block
      const int pixb = p.in_ld * 2, rowb = p.W * pixb;
      const unsigned char* img = reinterpret_cast<const unsigned char*>(p.in + ((long long)b * p.H * p.W + (long long)hi_base * p.W + wi_base) * p.in_ld + cv0 * 8);
      int off = ty * rowb + tx * pixb + ch * 16;
      for (; il < sg.ni; il += 8) {
        const bool ok = ch < chv && ty >= ty_lo && ty < ty_hi && tx >= tx_lo && tx < tx_hi;
        // pad chunks are never read: their lanes stay OFF (an LDS-DMA lane with EXEC = 0 moves nothing).  Only real halo pixels
        // fetch the zero word -- with every lane 8 of every instruction of every CU reading that one 16-byte address the DMA
        // stream ran at ~300 cycles per instruction (one L2 channel serving 25 000 requests per launch)
        if (ch < 8) glds16(ok ? img + off : zsrc, base + il * 1024);
        // this wave's next instruction: slot += 512 = 56 pixels + 8 chunks
        ch += 8; tx += 56; off += 56 * pixb + 128;
        if (ch >= 9) { ch -= 9; tx += 1; off += pixb - 144; }
        while (tx >= cols_in) { tx -= cols_in; ++ty; off += rowb - cols_in * pixb; }
      }
    }
    ni_total = sg.i0 + sg.ni;
  }
  // DMA instructions this wave has issued with global index <= L
  auto issued_upto = [&](int L) { return L >= wave ? ((L - wave) >> 3) + 1 : 0; };
  const int n_dma = issued_upto(ni_total - 1);
  unsigned long long tki = 0, tku = 0;
  if constexpr (DIAG) tki = __builtin_amdgcn_s_memtime();

  // ---- 2. the unit table (no dependence on the DMA): for lane-pixel j of unit u of flattened step fs the window base in the
  //         segment's LDS image and the byte offset of the output pixel from the segment's first output pixel, or -1 / unused ----------
  const int nmain = p.Wo >> 4, tw = p.Wo & 15;
  int* utab = reinterpret_cast<int*>(lds + p.utab_off);           // [steps][ups][2][16]
  for (int k = 0; k < nseg; ++k) {
    const DwsSeg sg = dws_segment<STRIDE>(k, G0, G1, p.Ho, cols_in, p.S);
    for (int e = tid; e < sg.nsteps * p.ups * 16; e += 512) {
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
      valid = valid && r < sg.nr;
      int* row = utab + (sg.fs0 * p.ups + (e >> 4)) * 32;
      row[jj] = valid ? ((r * STRIDE) * cols_in + x * STRIDE) * DWS_PP : -1;
      row[16 + jj] = valid ? (r * p.Wo + x) * p.out_ld * 2 : 0;
    }
  }

  if constexpr (DIAG) tku = __builtin_amdgcn_s_memtime();

  // ---- 3. segments x steps ---------------------------------------------------------------------------------------------------
  int stores_issued = 0;                                    // global stores this wave has issued so far (wave-uniform)
  float csk[DWS_MAXSEG][4];                                 // this lane's channel sums per segment (selected by predicated moves)
#pragma unroll
  for (int k = 0; k < DWS_MAXSEG; ++k)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) csk[k][r4] = 0.f;

  with_act(p.act, [&](auto A) {
    for (int k = 0; k < nseg; ++k) {
      const DwsSeg sg = dws_segment<STRIDE>(k, G0, G1, p.Ho, cols_in, p.S);
      const int b = sg.col / p.ncb, cv0 = (sg.col - b * p.ncb) * 8;
      const int cbase = (cv0 + 2 * g) * 8;
      const bool g_ok = cbase < p.C;
      const unsigned char* hdr = lds + sg.ldsoff;
      const unsigned char* tile = hdr + DWS_HDR;
      // this lane's 4 output channels (D fragment: rows 4q .. 4q+3 of column j) of the segment's first output pixel
      unsigned char* obase = reinterpret_cast<unsigned char*>(p.out + ((long long)(b * p.Ho + sg.r0) * p.Wo) * p.out_ld + cbase + 4 * q);
      s16x8 afr[5];
      int toff[5];
      float sc[4], sh[4];
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
      for (int st = 0; st < sg.nsteps; ++st) {
        const int fs = sg.fs0 + st;
        // rows this step reads: image rows [0, row_hi); the DMA instructions that carry them: global index <= L
        const int row_hi = min(sg.rows_in, (min((st + 1) * p.S, sg.nr) - 1) * STRIDE + 3);
        const int L = sg.i0 + 2 + ((row_hi * cols_in * 9 - 1) >> 6);
        wait_vmcnt_rt((n_dma - issued_upto(min(L, ni_total - 1))) + stores_issued);
        // raw barrier: a __syncthreads() carries a fence that drains the whole DMA queue.  lgkmcnt(0) as well: round 2's finding in
        // head_fused.hip about the safe hand-over form for LDS-DMA stages (and the unit table's ds_writes before the first step)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // the step's rows have landed for every wave
        asm volatile("" ::: "memory");
        if constexpr (DIAG) {
          const float t = (float)(__builtin_amdgcn_s_memtime() - tk0);
          if (fs == 0) t_ready[0] = t; else if (fs == 1) t_ready[1] = t; else if (fs == 2) t_ready[2] = t; else if (fs == 3) t_ready[3] = t;
          else if (fs == 4) t_ready[4] = t; else if (fs == 5) t_ready[5] = t;
        }
        if (st == 0) {                                      // per-lane constants out of the segment's header
#pragma unroll
          for (int pr = 0; pr < 5; ++pr) {
            const int tap = 2 * pr + (q >> 1);
            const bool on = g_ok && tap < 9 && (j >> 3) == (q & 1);
            const short wv = *reinterpret_cast<const short*>(hdr + (min(tap, 8) * 8 + 2 * g + (j >> 3)) * 16 + (j & 7) * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) afr[pr][e] = (on && e == (j & 7)) ? wv : (short)0;
            const int tc = min(tap, 8);
            toff[pr] = ((tc / 3) * cols_in + tc % 3) * DWS_PP + (2 * g + (q & 1)) * 16;
          }
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(hdr + 72 * 16 + (16 * g + 4 * q) * 4);
          const f32x4 h4 = *reinterpret_cast<const f32x4*>(hdr + 88 * 16 + (16 * g + 4 * q) * 4);
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) { sc[r4] = g_ok ? s4[r4] : 1.f; sh[r4] = g_ok ? h4[r4] : 0.f; }
        }
        if (g_ok) {
          const int* ut = utab + fs * p.ups * 32;
          // the two waves of a channel group share a step's units: unit u goes to half (u + fs) & 1
          for (int u0 = (hf + fs) & 1; u0 < p.ups; u0 += 2 * DWS_UN) {
            int ooff[DWS_UN]; bool ok[DWS_UN], present[DWS_UN]; f32x4 acc[DWS_UN];
            s16x8 bfr[DWS_UN][5];
#pragma unroll
            for (int t = 0; t < DWS_UN; ++t) {
              present[t] = false;
              const int u = u0 + 2 * t;
              if (u < p.ups) {                              // wave-uniform
                const int tb = ut[u * 32 + j];
                // a unit whose FIRST pixel is past the segment's last row has no valid pixel at all (rows ascend inside a unit): skipped
                // as a whole -- its store instruction must not be counted, because an all-lanes-off store is branched around, not issued
                present[t] = __builtin_amdgcn_readfirstlane(tb) >= 0;
                if (present[t]) {
                  ok[t] = tb >= 0;
                  ooff[t] = ut[u * 32 + 16 + j];
                  const int base = ok[t] ? tb : 0;          // invalid lanes read pixel 0 (finite data), results dropped
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
          if (fs == 0) t_done[0] = t; else if (fs == 1) t_done[1] = t; else if (fs == 2) t_done[2] = t; else if (fs == 3) t_done[3] = t;
          else if (fs == 4) t_done[4] = t; else if (fs == 5) t_done[5] = t;
        }
      }
#pragma unroll
      for (int kk = 0; kk < DWS_MAXSEG; ++kk)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) csk[kk][r4] = (kk == k) ? csum[r4] : csk[kk][r4];
    }
  });
  unsigned long long tkd = 0;
  if constexpr (DIAG) tkd = __builtin_amdgcn_s_memtime();

  // ---- 4. channel sums: 16-lane butterfly, the two halves of a group added through LDS (half 0 + half 1, a fixed order), one slot of
  //         partial[b][slot][C] per (column, workgroup); the workgroup that ends a column zeroes the slots nobody fills ------------------
  if (p.partial != nullptr) {
    float* red = reinterpret_cast<float*>(lds + p.utab_off);       // the unit table is dead: [seg][group][16 channels]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DWS_MAXSEG; ++k) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) csk[k][r4] += __shfl_xor(csk[k][r4], o, 64);
      }
      if (hf == 1 && j == 0) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) red[(k * 4 + g) * 16 + 4 * q + r4] = csk[k][r4];
      }
    }
    __syncthreads();
    if (hf == 0 && j == 0) {
      for (int k = 0; k < nseg; ++k) {
        const DwsSeg sg = dws_segment<STRIDE>(k, G0, G1, p.Ho, cols_in, p.S);
        const int b = sg.col / p.ncb, cb = sg.col - b * p.ncb;
        const int cbase = (cb * 8 + 2 * g) * 8;
        if (cbase >= p.C) continue;
        // slot = number of workgroup boundaries inside the column before this segment
        const long long X = (long long)sg.col * p.Ho;
        const int wfirst = (int)(((X + 1) * NW + p.TOT - 1) / p.TOT) - 1;
        const int slot = w - wfirst;
        float mine[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) mine[r4] = (k == 0 ? csk[0][r4] : (k == 1 ? csk[1][r4] : csk[2][r4])) + red[(k * 4 + g) * 16 + 4 * q + r4];
        float* dst = p.partial + ((long long)b * p.nslot + slot) * p.C + cbase + 4 * q;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) dst[r4] = mine[r4];
        if (sg.r0 + sg.nr == p.Ho) {
          for (int s2 = slot + 1; s2 < p.nslot; ++s2) {
            float* z = p.partial + ((long long)b * p.nslot + s2) * p.C + cbase + 4 * q;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) z[r4] = 0.f;
          }
        }
      }
    }
  }
  if constexpr (DIAG) {
    if (p.partial != nullptr && tid == 0) {      // stamps AFTER the partial-sum area; they feed no output value
      const unsigned long long tk3 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
      float* dbg = p.partial + (long long)p.B * p.nslot * p.C + (long long)blockIdx.x * 24;
      dbg[0] = t_ready[0]; dbg[1] = (float)(tkd - tk0) - t_ready[0]; dbg[2] = (float)(tk3 - tkd);
      dbg[18] = (float)(tki - tk0); dbg[19] = (float)(tku - tk0);
      dbg[3] = (float)(tr0 & 0xffffff); dbg[4] = (float)(tr1 & 0xffffff); dbg[5] = 3.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) { dbg[6 + k] = t_ready[k]; dbg[12 + k] = t_done[k]; }
    }
  }
}

}  // namespace

// Work decomposition: NW persistent workgroups (one per CU; two per CU when a workgroup's rows would not fit 150 KB of LDS), S
// output rows per step chosen so that a step's row tails pack into whole 16-pixel units.  The plan walks the workgroups' row lists
// on the host (a few hundred iterations) for the exact LDS need, the step count and the slot count of the partial sums.
int g_dws_force_S = 0, g_dws_force_NW = 0;     // cfp_debug_set keys 7 / 8 (tools/dw_bench.py --sweep-stream)

struct DwsPlan { int S, NW, ups, nslot, utab_off; size_t lds; };

static int dws_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

static bool dws_plan(int B, int Ho, int Wo, int C, int stride, DwsPlan& best) {
  const int cols_in = (Wo - 1) * stride + 3;
  const int ncb = cdiv(C, 64);
  const int nmain = Wo >> 4, tw = Wo & 15;
  const long long TOTll = (long long)B * ncb * Ho;
  if (TOTll >= (1ll << 30)) return false;
  const int TOT = (int)TOTll;
  // rows per step: best unit packing, then the smaller step
  int S = 1; double bp = 1e30;
  for (int s = 1; s <= 4 && s <= Ho; ++s) {
    if (g_dws_force_S && s != g_dws_force_S) continue;
    const int ups = s * nmain + (tw ? cdiv(s * tw, 16) : 0);
    const double pack = (double)ups * 16.0 / ((double)s * Wo) * (1.0 + 0.02 * s);
    if (pack < bp) { bp = pack; S = s; }
  }
  const int ups = S * nmain + (tw ? cdiv(S * tw, 16) : 0);
  const int cus = dws_cu_count();
  for (int mult = 1; mult <= 2; ++mult) {
    int NW = g_dws_force_NW ? g_dws_force_NW : cus * mult;
    if (NW > TOT) NW = TOT;                                  // at least one row each
    const size_t cap = (mult == 1 && !g_dws_force_NW) ? 150 * 1024 : 76 * 1024;
    size_t img_max = 0; int steps_max = 0, nslot = 1, dma_max = 0;
    bool ok = true;
    int col_prev = -1, col_cnt = 0;
    for (int w = 0; w < NW && ok; ++w) {
      const int G0 = (int)((long long)w * TOT / NW), G1 = (int)((long long)(w + 1) * TOT / NW);
      size_t img = 0; int steps = 0, nseg = 0, ni = 0;
      for (int g = G0; g < G1;) {
        const int col = g / Ho, r0 = g - col * Ho, nr = std::min(Ho - r0, G1 - g);
        if (++nseg > DWS_MAXSEG) { ok = false; break; }
        const int rows_in = (nr - 1) * stride + 3;
        const int n = 2 + cdiv((long long)rows_in * cols_in * 9, 64);
        img += (size_t)n * 1024; ni += n; steps += cdiv(nr, S);
        if (col == col_prev) ++col_cnt; else { col_prev = col; col_cnt = 1; }
        if (col_cnt > nslot) nslot = col_cnt;
        g += nr;
      }
      if (img > img_max) img_max = img;
      if (steps > steps_max) steps_max = steps;
      if (ni > dma_max) dma_max = ni;
    }
    if (!ok) continue;
    const size_t lds = img_max + (size_t)std::max(steps_max * ups * 128, DWS_MAXSEG * 4 * 16 * 4);
    if (lds > cap) continue;
    if (cdiv(dma_max, 8) > 38) continue;                     // DMA instructions per wave must fit the vmcnt field beside the stores
    best = DwsPlan{S, NW, ups, nslot, (int)img_max, lds};
    return true;
  }
  return false;
}

void cfp_dws_debug_set(int key, int value) { if (key == 7) g_dws_force_S = value; else g_dws_force_NW = value; }

int cfp_dws_strips(int B, int Ho, int Wo, int C, int stride) {
  DwsPlan d;
  return dws_plan(B, Ho, Wo, C, stride, d) ? d.nslot : 0;
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
  p.S = d.S; p.ncb = cdiv(C, 64); p.ups = d.ups; p.nslot = d.nslot; p.TOT = B * p.ncb * Ho; p.utab_off = d.utab_off;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DWS_LAUNCH(HH, ST, DG)                                                                                                       \
  do {                                                                                                                               \
    static bool attr = false;                                                                                                        \
    if (!attr) {                                                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)dw3x3_stream_kernel<HH, ST, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); \
      if (e != hipSuccess) { cfp_set_error(std::string(who) + ": " + hipGetErrorString(e)); return CFP_EHIP; }                       \
      attr = true;                                                                                                                   \
    }                                                                                                                                \
    hipLaunchKernelGGL((dw3x3_stream_kernel<HH, ST, DG>), dim3((unsigned)d.NW), dim3(512), d.lds, s, p);                             \
  } while (0)
#define DWS_ST(HH, DG) do { if (stride == 1) DWS_LAUNCH(HH, 1, DG); else DWS_LAUNCH(HH, 2, DG); } while (0)
  if (dtype == CFP_F16) { if (diag) DWS_ST(f16_t, true); else DWS_ST(f16_t, false); }
  else { if (diag) DWS_ST(bf16_t, true); else DWS_ST(bf16_t, false); }
#undef DWS_ST
#undef DWS_LAUNCH
  return cfp_check_launch(who);
}
