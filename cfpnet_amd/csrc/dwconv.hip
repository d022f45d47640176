// Depthwise convolutions in NHWC.
//
// dw3x3  : the encoder's 24 depthwise 3x3 convs (C = 224 ... 1392).  4.3 FLOP/byte -> HBM bound.
//          One thread owns a 16-byte channel vector of PW consecutive output pixels; consecutive
//          threads own consecutive channel vectors, so every load/store instruction of a wave is
//          a run of full 16-byte-per-lane accesses along the channel axis.  Row re-use (3x) is
//          served by L2; column re-use by registers.
// dwlarge: LKPM's 7x7 / 15x15 / 31x31 depthwise convs (C = 128 / 64 / 32).  Up to 480 FLOP/byte
//          -> vector-FMA bound; design notes at the kernel.
#include "common.h"

namespace {

// Raw 16-byte vector -> VE floats, or zeros when !ok (the select is on the VALUE: the load itself is
// unconditional from a clamped address, so a thread's 18 input + 9 weight loads are all in flight
// together instead of one branch + wait per tap).
template <typename T> __device__ __forceinline__ void unpack_masked(const u32x4& raw, bool ok, float* v);
template <> __device__ __forceinline__ void unpack_masked<bf16_t>(const u32x4& raw, bool ok, float* v) {
  const uint32_t m = ok ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t x = raw[i] & m;
    v[2 * i] = __uint_as_float(x << 16);
    v[2 * i + 1] = __uint_as_float(x & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void unpack_masked<f16_t>(const u32x4& raw, bool ok, float* v) {
  const uint32_t m = ok ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 h = __builtin_bit_cast(f16x2, raw[i] & m);
    v[2 * i] = (float)h[0];
    v[2 * i + 1] = (float)h[1];
  }
}
template <> __device__ __forceinline__ void unpack_masked<float>(const u32x4& raw, bool ok, float* v) {
  const uint32_t m = ok ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(raw[i] & m);
}

// Depthwise 3x3 (+ optional per-(image, row strip, channel) sums of the stored output for
// squeeze-excite: the H*W mean timm's SqueezeExcite takes of this tensor then needs no extra pass).
//
// HBM-bound (4.3 FLOP/byte), and at batch 8 a launch moves only 13-30 MB, so what matters is that
// every byte is requested from memory once and that all of a workgroup's requests are in flight
// together.  Workgroup = one image, one strip of R output rows (full width), CVB channel vectors
// (16-byte vectors: 8 bf16 / 4 f32):
//   1. the input strip + halo ((R-1)*S+3 rows x (Wo-1)*S+3 columns, zeros outside the image) goes
//      global -> LDS in one sweep of 16-byte loads, CVB*16 contiguous bytes per pixel (128/256 B);
//      read amplification = halo rows only (1.2-1.7x) instead of 9 taps / column reuse (4.5-6x);
//   2. thread = (channel vector, pixel slot); a unit of work is a run of 4 output pixels of one row:
//      3 x 6 conflict-free ds_read_b128 (a 16-lane read group covers 256 contiguous bytes) feed 4
//      outputs, weights held in registers, f32 FMAs, BN scale/shift + activation, one 16-byte
//      store per output pixel;
//   3. per-channel sums of the stored values are reduced over the pixel slots through LDS in slot
//      order (deterministic) and written as partial[b][strip][c].
// grid = (ceil(CV/CVB), B * nstrips).
template <typename T, int STRIDE, int CVB>
__global__ __launch_bounds__(256) void dw3x3_kernel(const T* __restrict__ in, int in_ld, const T* __restrict__ w,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    T* __restrict__ out, int out_ld, float* __restrict__ partial, int B, int H,
                                                    int W, int C, int pad_t, int pad_l, int Ho, int Wo, int act, int R,
                                                    int nstrips, const float* __restrict__ w_red, int RD, float* __restrict__ hpart) {
  constexpr int VE = Vec<T>::N;
  constexpr int NSLOT = 256 / CVB;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u32x4* tile = reinterpret_cast<u32x4*>(dsm);          // [rows_in][cols_in][CVB]
  const int CV = C / VE;
  const int strip = blockIdx.y % nstrips, b = blockIdx.y / nstrips;
  const int cv0 = blockIdx.x * CVB;
  const int ho_begin = strip * R;
  const int rows = min(R, Ho - ho_begin);
  const int rows_in = (rows - 1) * STRIDE + 3, cols_in = (Wo - 1) * STRIDE + 3;
  const int hi_base = ho_begin * STRIDE - pad_t, wi_base = -pad_l;
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, slot = tid / CVB;
  const bool cv_ok = cv0 + cvl < CV;
  const int c0 = (cv_ok ? cv0 + cvl : 0) * VE;

  // ---- 1. input strip + halo -> LDS (every load of the workgroup in flight together) ----------
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int items = rows_in * cols_in * CVB;
  for (int i = tid; i < items; i += 256) {
    const int px = i / CVB;                       // cvl == i % CVB because 256 % CVB == 0
    const int ty = px / cols_in, tx = px - ty * cols_in;
    const int hi = hi_base + ty, wi = wi_base + tx;
    const bool ok = cv_ok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
    const int hic = min(max(hi, 0), H - 1), wic = min(max(wi, 0), W - 1);
    u32x4 v = *reinterpret_cast<const u32x4*>(in + ((long long)(b * H + hic) * W + wic) * in_ld + c0);
    tile[i] = ok ? v : zero4;
  }
  u32x4 wraw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wraw[t] = *reinterpret_cast<const u32x4*>(w + t * C + c0);
  float sc[VE], sh[VE];
#pragma unroll
  for (int e = 0; e < VE; e += 4) { Vec<float>::load(scale + c0 + e, sc + e); Vec<float>::load(shift + c0 + e, sh + e); }
  float wv[9][VE];
#pragma unroll
  for (int t = 0; t < 9; ++t) unpack_masked<T>(wraw[t], true, wv[t]);
  __syncthreads();

  // ---- 2. compute from LDS ---------------------------------------------------------------------
  float csum[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) csum[e] = 0.f;
  constexpr int L = 4;                         // output pixels per unit (a run along the row)
  constexpr int IW = (L - 1) * STRIDE + 3;     // input columns feeding a run
  const int runs = (Wo + L - 1) / L;
  with_act(act, [&](auto A) {
    for (int u = slot; u < rows * runs; u += NSLOT) {
      const int r = u / runs, x0 = (u - r * runs) * L;
      float acc[L][VE];
#pragma unroll
      for (int p = 0; p < L; ++p)
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[p][e] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const u32x4* rp = tile + ((r * STRIDE + kh) * cols_in) * CVB + cvl;
#pragma unroll
        for (int xx = 0; xx < IW; ++xx) {
          const int col = min(x0 * STRIDE + xx, cols_in - 1);   // columns past the row end feed masked outputs only
          float v[VE];
          unpack_masked<T>(rp[col * CVB], true, v);
#pragma unroll
          for (int p = 0; p < L; ++p) {
            const int kw = xx - p * STRIDE;   // compile-time after unrolling
            if (kw >= 0 && kw < 3) {
#pragma unroll
              for (int e = 0; e < VE; ++e) acc[p][e] = fmaf(v[e], wv[kh * 3 + kw][e], acc[p][e]);
            }
          }
        }
      }
#pragma unroll
      for (int p = 0; p < L; ++p) {
        if (x0 + p < Wo) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            acc[p][e] = to_f32<T>(from_f32<T>(act_c<decltype(A)::value>(acc[p][e] * sc[e] + sh[e])));   // the stored (rounded) value
            csum[e] += acc[p][e];
          }
          if (cv_ok) Vec<T>::store(out + ((long long)(b * Ho + ho_begin + r) * Wo + x0 + p) * out_ld + c0, acc[p]);
        }
      }
    }
  });
  if (partial == nullptr && hpart == nullptr) return;   // uniform

  // ---- 3. channel sums over the pixel slots ---------------------------------------------------
  __syncthreads();                  // everyone is done reading the tile: reuse it
  float* red = reinterpret_cast<float*>(dsm);   // [NSLOT][CVB*VE]
#pragma unroll
  for (int e = 0; e < VE; ++e) red[slot * (CVB * VE) + cvl * VE + e] = csum[e];
  __syncthreads();
  float s = 0.f;
  if (tid < CVB * VE) {
    const int c = cv0 * VE + tid;
    if (c < C) {
      for (int j = 0; j < NSLOT; ++j) s += red[j * (CVB * VE) + tid];
      if (partial) partial[((long long)b * nstrips + strip) * C + c] = s;
    }
  }
  if (hpart == nullptr) return;     // uniform
  // squeeze-excite reduce FC applied to this workgroup's channel sums (it is linear in them): hpart[b][strip * ncb + block][r], the
  // part of  w_red[r][:] . sum[:]  this block's channels contribute; cfp_se_gate_fold2 adds the parts in index order
  __syncthreads();
  if (tid < CVB * VE) red[tid] = s;                 // channels past C hold 0
  __syncthreads();
  if (tid < RD) {
    float dot = 0.f;
    for (int j = 0; j < CVB * VE; ++j) {
      const int c = cv0 * VE + j;
      if (c < C) dot = fmaf(w_red[(long long)tid * C + c], red[j], dot);
    }
    hpart[(((long long)b * nstrips + strip) * gridDim.x + blockIdx.x) * RD + tid] = dot;
  }
}

// ---- depthwise 3x3 with the tap reduction on the matrix cores (bf16) ---------------------------
// In-kernel stamps of the VALU kernel above showed what bounds it at batch 8: per workgroup ~5 us
// waiting for the strip, then a compute phase that is VALU-ISSUE bound (bf16 unpack + 72 FMA +
// address arithmetic + SiLU: 0.5 VALU instructions per output element), and no overlap between the
// two because every resident workgroup is in the same phase.  A depthwise conv has no reduction
// over channels, but
//     out[c][p] = sum_t w[t][c] * in[p + off_t][c]
// is the MFMA  D[i][j] += sum_k A[i][k] B[k][j]  with i = channel (16), j = pixel (16),
// k = (tap slot t' in {0,1}, channel c' in 0..15),  A[i][(t',c')] = w[2P+t'][i] * delta(c', i)
// (a DIAGONAL weight block per tap, built once per wave in registers) and B[(t',c')][j] =
// in[pixel j + off_(2P+t')][c'].  Only 1/16 of the MACs are useful, but the matrix pipe is 16x
// the vector rate, the bf16 operands need no unpacking, the B fragment of a lane is exactly one
// 16-byte LDS read (8 channels of one tap of one pixel) at a CONSTANT offset from the run's base
// address, and the VALU is left with the epilogue: 9 taps = 5 MFMAs per (16 channels x 16 pixels).
//   * LDS tile [rows_in][cols_in] pixels with a pixel pitch of CVB*16 + 16 bytes: 16 consecutive
//     pixels read conflict-free without a swizzle, so every address is linear in (pixel, tap);
//   * a wave owns one 16-channel group and walks the strip in raster order, 16 pixels of a row at
//     a time (row tails narrower than 16 are gathered from 16/tail consecutive rows), four runs per
//     loop iteration so their LDS-read -> MFMA -> SiLU chains overlap;
//   * results are written IN PLACE into the tile at the window's top-left input pixel, which no
//     later run reads (raster order), so there is no second LDS tile and the copy-out is fully
//     coalesced 16-byte stores;
//   * channel sums: per-lane over its pixels, then a 16-lane shuffle reduction; no cross-wave step.
template <typename HT, int STRIDE, int CVB>
__global__ __launch_bounds__(256) void dw3x3_mfma_kernel(const bf16_t* __restrict__ in, int in_ld, const bf16_t* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         bf16_t* __restrict__ out, int out_ld, float* __restrict__ partial, int B,
                                                         int H, int W, int C, int pad_t, int pad_l, int Ho, int Wo, int act, int R,
                                                         int nstrips, const float* __restrict__ w_red, int RD, float* __restrict__ hpart,
                                                         int cs_off) {
  constexpr int G = CVB / 2;                      // 16-channel groups per workgroup
  constexpr int GPW = (G + 3) / 4;                // groups per wave
  constexpr int PP = CVB * 16 + 16;               // pixel pitch in bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  const int CV = C / 8;
  // XCD-aware order: the channel blocks of one strip straddle each other's 128-byte lines and neighbouring strips share halo rows; dealt
  // round-robin over the 8 XCDs every one of them fetched those lines into its own L2 (the 2.3x fetch amplification of DESIGN 4.2)
  const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx = lin % (int)gridDim.x, by = lin / (int)gridDim.x;
  const int strip = by % nstrips, b = by / nstrips;
  const int cv0 = bx * CVB;
  const int ho_begin = strip * R;
  const int rows = min(R, Ho - ho_begin);
  const int rows_in = (rows - 1) * STRIDE + 3, cols_in = (Wo - 1) * STRIDE + 3;
  const int hi_base = ho_begin * STRIDE - pad_t, wi_base = -pad_l;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;         // B/D column (pixel) and k-chunk / D row block
  // diagnostic path of the same kernel (tools/dw_bench.py --stamps): act >= 100 writes per-workgroup phase
  // stamps AFTER the partial-sum area; the stamps feed no output value
  const bool stamps = act >= 100;
  if (stamps) act -= 100;
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tr0 = 0;
  if (stamps) { tk0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }

  // per-lane constants are requested BEFORE the tile staging so that their latency hides under it
  short wv5[GPW][5];
  float sc[GPW][4], sh[GPW][4];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave + 4 * gi;
    const int cbase = (cv0 + 2 * g) * 8;
    const bool g_ok = g < G && cbase < C;
#pragma unroll
    for (int pr = 0; pr < 5; ++pr) {
      const int tap = 2 * pr + (q >> 1);
      const bool on = g_ok && tap < 9 && (j >> 3) == (q & 1);
      wv5[gi][pr] = on ? (short)w[min(tap, 8) * C + cbase + j] : (short)0;
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      sc[gi][r4] = g_ok ? scale[cbase + 4 * q + r4] : 1.f;
      sh[gi][r4] = g_ok ? shift[cbase + 4 * q + r4] : 0.f;
    }
  }

  // ---- 1. input strip + halo -> LDS ---------------------------------------------------------------
  {
    const int cvl = tid % CVB;
    const bool cv_ok = cv0 + cvl < CV;
    const int c0 = (cv_ok ? cv0 + cvl : 0) * 8;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int items = rows_in * cols_in * CVB;
    for (int i = tid; i < items; i += 256) {
      const int px = i / CVB;
      const int ty = px / cols_in, tx = px - ty * cols_in;
      const int hi = hi_base + ty, wi = wi_base + tx;
      const bool ok = cv_ok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const int hic = min(max(hi, 0), H - 1), wic = min(max(wi, 0), W - 1);
      u32x4 v = *reinterpret_cast<const u32x4*>(in + ((long long)(b * H + hic) * W + wic) * in_ld + c0);
      *reinterpret_cast<u32x4*>(tile + px * PP + cvl * 16) = ok ? v : zero4;
    }
    // (This loop compiles to load -> s_waitcnt vmcnt(0) -> ds_write per iteration, ~11 serial round trips per thread.  Issuing eight
    // loads before the first wait, with add-and-wrap indices instead of the divisions, was measured in round 2: the load phase stays at
    // 5.9-6.1 us (in-kernel stamps) and the kernel at 13-22 us -- the phase is bound by the 2.3x fetch amplification of 128-byte pieces at
    // a 1.3-2.8 KB stride, not by the number of loads a thread has in flight.  Not adopted.)
  }
  {
    const int nmain_ = Wo >> 4, tw_ = Wo & 15;
    const int RB_ = (tw_ == 0) ? 1 : ((16 % tw_ == 0) ? 16 / tw_ : 1);
    const int upb_ = RB_ * nmain_ + (tw_ ? 1 : 0);
    const int nun_ = ((rows + RB_ - 1) / RB_) * upb_;
    int* ut = reinterpret_cast<int*>(tile + ((rows_in * cols_in * PP + 15) & ~15));
    for (int e = tid; e < nun_ * 16; e += 256) {
      const int u = e >> 4, jj = e & 15;
      const int rb = u / upb_, k = u - rb * upb_;
      int r, x; bool valid;
      if (k < RB_ * nmain_) {
        r = rb * RB_ + k / nmain_;
        x = (k % nmain_) * 16 + jj;
        valid = r < rows;
      } else {
        const int jr = (RB_ > 1) ? jj / tw_ : 0;
        r = rb * RB_ + jr;
        x = (Wo - tw_) + (jj - jr * tw_);
        valid = r < rows && (jj - jr * tw_) < tw_ && jr < RB_;
      }
      ut[e] = valid ? ((r * STRIDE) * cols_in + x * STRIDE) * PP : -1;
    }
  }
  __syncthreads();
  if (stamps) tk1 = __builtin_amdgcn_s_memtime();

  // ---- 2. runs of 16 output pixels on the matrix cores -------------------------------------------------
  // unit list of the strip, in an order that keeps the in-place writes safe: for every block of RB rows,
  // first the full 16-pixel runs of those rows (raster order), then the gathered row tails.
  const int nmain = Wo >> 4;                      // full runs per row
  const int tw = Wo & 15;                         // row tail width
  const int RB = (tw == 0) ? 1 : ((16 % tw == 0) ? 16 / tw : 1);   // rows gathered into one tail run
  const int upb = RB * nmain + (tw ? 1 : 0);      // units per row block
  const int nrb = (rows + RB - 1) / RB;
  const int nunits = nrb * upb;
  // window base (byte offset of the top-left input pixel) of lane-pixel j of unit u, or -1: a table in LDS,
  // filled once per workgroup, so the run loop carries no integer divisions
  int* utab = reinterpret_cast<int*>(tile + ((rows_in * cols_in * PP + 15) & ~15));
  const int tapstep = PP;                         // one pixel to the right
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave + 4 * gi;                   // wave-uniform
    const int cbase = (cv0 + 2 * g) * 8;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < G && cbase < C && act != 99) {         // act 99: timing experiment (staging + copy-out only)
      s16x8 afr[5];                                // diagonal weight fragments: lane (row i = j, chunk q) holds A[i][k = 8q..8q+7]
      int toff[5];                                 // byte offset of this lane's tap (pair pr, slot q>>1) + its 16-byte vector
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) afr[pr][e] = (e == (j & 7)) ? wv5[gi][pr] : (short)0;
        const int tc = min(2 * pr + (q >> 1), 8);
        toff[pr] = ((tc / 3) * cols_in + tc % 3) * tapstep + (2 * g + (q & 1)) * 16;
      }
      const int ooff = (2 * g + (q >> 1)) * 16 + (q & 1) * 8;     // where this lane's 4 output channels live in a pixel
      with_act(act, [&](auto A) {
        constexpr int UN = 4;      // runs in flight per iteration: the kernel is bound by the LDS -> MFMA -> SiLU latency chain
        for (int u0 = 0; u0 < nunits; u0 += UN) {
          int base[UN]; float vf[UN]; f32x4 acc[UN];
#pragma unroll
          for (int t = 0; t < UN; ++t) {
            const int tb = utab[min(u0 + t, nunits - 1) * 16 + j];
            const bool valid = tb >= 0 && (u0 + t < nunits);
            vf[t] = valid ? 1.f : 0.f;
            base[t] = valid ? tb : 0;                  // invalid lanes read pixel 0 (finite data), results dropped
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          s16x8 bfr[UN][5];
#pragma unroll
          for (int t = 0; t < UN; ++t)
#pragma unroll
            for (int pr = 0; pr < 5; ++pr) bfr[t][pr] = *reinterpret_cast<const s16x8*>(tile + base[t] + toff[pr]);
#pragma unroll
          for (int pr = 0; pr < 5; ++pr)
#pragma unroll
            for (int t = 0; t < UN; ++t) acc[t] = mfma16<HT>(afr[pr], bfr[t][pr], acc[t]);
          float y[UN][4];
#pragma unroll
          for (int t = 0; t < UN; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              y[t][r4] = act_c<decltype(A)::value>(acc[t][r4] * sc[gi][r4] + sh[gi][r4]);
              csum[r4] = fmaf(y[t][r4], vf[t], csum[r4]);
            }
#pragma unroll
          for (int t = 0; t < UN; ++t) {
            if (vf[t] != 0.f) {        // D: lane holds channels cbase + 4q .. +3 of pixel j
              uint2 pk;
              pk.x = pack2<HT>(y[t][0], y[t][1]);
              pk.y = pack2<HT>(y[t][2], y[t][3]);
              *reinterpret_cast<uint2*>(tile + base[t] + ooff) = pk;
            }
          }
        }
      });
      if (partial != nullptr) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) csum[r4] += __shfl_xor(csum[r4], o, 64);
        }
        if (j == 0) {
          float* dst = partial + ((long long)b * nstrips + strip) * C + cbase + 4 * q;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) dst[r4] = csum[r4];
        }
      }
      if (hpart != nullptr) {          // the workgroup's channel sums, for the reduce-FC dot products below
        if (partial == nullptr) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) csum[r4] += __shfl_xor(csum[r4], o, 64);
          }
        }
        if (j == 0) {
          float* cs = reinterpret_cast<float*>(tile + cs_off);
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) cs[16 * g + 4 * q + r4] = csum[r4];
        }
      }
    } else {
      if (partial != nullptr && g < G && cbase < C && j == 0) {
        float* dst = partial + ((long long)b * nstrips + strip) * C + cbase + 4 * q;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) dst[r4] = 0.f;
      }
      if (hpart != nullptr && g < G && j == 0) {
        float* cs = reinterpret_cast<float*>(tile + cs_off);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) cs[16 * g + 4 * q + r4] = 0.f;
      }
    }
  }
  __syncthreads();
  if (stamps) tk2 = __builtin_amdgcn_s_memtime();

  // squeeze-excite (round 3): the reduce FC is linear in the channel sums, so this workgroup adds ITS part of every hidden unit,
  //   hpart[b][strip * blocks + block][r] = sum_{c in block} w_reduce[r][c] * csum[c],
  // and cfp_se_gate_fold2 only has to add K small vectors instead of reading the R x C weights in every workgroup.  Thread = (hidden
  // unit r = tid >> 2, quarter of the block's channels); the weight loads are issued BEFORE the copy-out so their latency hides under it.
  constexpr int NCQ = CVB * 2;                    // channels per thread = CVB * 8 / 4
  f32x4 wq[NCQ / 4];
  const int hr = tid >> 2, hq = tid & 3;
  if (hpart != nullptr) {
#pragma unroll
    for (int v = 0; v < NCQ / 4; ++v) {
      const int c = cv0 * 8 + hq * NCQ + v * 4;
      wq[v] = (hr < RD && c < C) ? *reinterpret_cast<const f32x4*>(w_red + (long long)hr * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- 3. coalesced copy-out of the in-place results -------------------------------------------------
  {
    const int cvl = tid % CVB;
    if (cv0 + cvl < CV) {
      const int c0 = (cv0 + cvl) * 8;
      const int npix = rows * Wo;
      for (int i = tid; i < npix * CVB; i += 256) {
        const int po = i / CVB;
        const int r = po / Wo, x = po - r * Wo;
        const int pxo = (r * STRIDE) * cols_in + x * STRIDE;
        *reinterpret_cast<u32x4*>(out + ((long long)(b * Ho + ho_begin + r) * Wo + x) * out_ld + c0) =
            *reinterpret_cast<const u32x4*>(tile + pxo * PP + cvl * 16);
      }
    }
  }
  if (hpart != nullptr) {
    const float* cs = reinterpret_cast<const float*>(tile + cs_off) + hq * NCQ;
    float sdot = 0.f;
#pragma unroll
    for (int v = 0; v < NCQ / 4; ++v) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cs + v * 4);
      sdot = fmaf(wq[v][0], c4[0], sdot); sdot = fmaf(wq[v][1], c4[1], sdot); sdot = fmaf(wq[v][2], c4[2], sdot); sdot = fmaf(wq[v][3], c4[3], sdot);
    }
    sdot += __shfl_xor(sdot, 1, 64);
    sdot += __shfl_xor(sdot, 2, 64);
    if (hq == 0 && hr < RD) hpart[(((long long)b * nstrips + strip) * gridDim.x + bx) * RD + hr] = sdot;
  }
  if (stamps && partial != nullptr && tid == 0) {
    const unsigned long long tk3 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    float* dbg = partial + (long long)B * nstrips * C + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 6;
    dbg[0] = (float)(tk1 - tk0); dbg[1] = (float)(tk2 - tk1); dbg[2] = (float)(tk3 - tk2);
    dbg[3] = (float)(tr0 & 0xffffff); dbg[4] = (float)(tr1 & 0xffffff); dbg[5] = 1.f;
  }
}

// ---- large kernel ------------------------------------------------------------------------
// Workgroup = 512 threads = 8 waves; wave w owns channel c0 + w of a 32-row x 32-column output
// tile, lane = (column, 16-row strip).  The input tile + halo sits in LDS planar per channel
// ([8][32+K-1][32+K-1] f32) so the 64 lanes of a wave read 2 x 32 consecutive floats:
// conflict-free ds_read_b32.  The channel is wave-uniform, so the K*K weights come through the
// scalar cache (s_load) and cost no LDS or VALU slot: the inner loop is 16 v_fma per LDS read.
// The vertical sliding window lives in a 32-register ring addressed with compile-time indices (K
// is a template parameter, the ky loop is fully unrolled): no register moves, and each row is
// fetched LPRE steps before its first use so the LDS latency hides behind ~128 FMAs.
constexpr int LTH = 32, LTW = 32, LC = 8, LSTRIP = 16, LPRE = 8;

template <typename T, int K>
__global__ __launch_bounds__(512) void dwlarge_kernel(const T* __restrict__ in, int in_ld,
                                                      const float* __restrict__ w, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, T* __restrict__ out,
                                                      int out_ld, int B, int H, int W, int C, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HALO = (K - 1) / 2;
  constexpr int TPH = LTH + K - 1, TPW = LTW + K - 1;
  constexpr int VE = Vec<T>::N;
  constexpr int VPP = LC / VE;
  float* sIn = lds;                         // [LC][TPH][TPW]

  const int tiles_x = (W + LTW - 1) / LTW, tiles_y = (H + LTH - 1) / LTH;
  const int cgs = C / LC;
  int bid = blockIdx.x;
  const int cg = bid % cgs; bid /= cgs;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c0 = cg * LC;
  const int y0 = ty * LTH - HALO, x0 = tx * LTW - HALO;
  const int tid = threadIdx.x;

  for (int i = tid; i < TPH * TPW * VPP; i += 512) {
    const int v = i % VPP, pix = i / VPP;
    const int py = pix / TPW, px = pix % TPW;
    const int y = y0 + py, x = x0 + px;
    float vals[VE];
    if (y >= 0 && y < H && x >= 0 && x < W) {
      Vec<T>::load(in + ((long long)(b * H + y) * W + x) * in_ld + c0 + v * VE, vals);
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) vals[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) sIn[(v * VE + e) * (TPH * TPW) + pix] = vals[e];
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int col = lane & 31, strip = lane >> 5;
  const float* __restrict__ wc = w + (long long)(c0 + wave) * (K * K);   // [kx][ky], wave-uniform
  const float* base = sIn + wave * (TPH * TPW) + (strip * LSTRIP) * TPW + col;
  constexpr int NROW = LSTRIP + K - 1;     // input rows a strip touches
  float acc[LSTRIP];
#pragma unroll
  for (int i = 0; i < LSTRIP; ++i) acc[i] = 0.f;

  for (int kx = 0; kx < K; ++kx) {
    const float* colp = base + kx;
    float win[32];
#pragma unroll
    for (int i = 0; i < LSTRIP - 1 + LPRE; ++i)
      if (i < NROW) win[i & 31] = colp[i * TPW];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      constexpr int AHEAD = LSTRIP - 1 + LPRE;
      if (ky + AHEAD < NROW) win[(ky + AHEAD) & 31] = colp[(ky + AHEAD) * TPW];
      const float wv = wc[kx * K + ky];
#pragma unroll
      for (int i = 0; i < LSTRIP; ++i) acc[i] = fmaf(win[(ky + i) & 31], wv, acc[i]);
    }
  }
  const float sc = scale[c0 + wave], sh = shift[c0 + wave];
  __syncthreads();                          // everyone is done reading sIn: reuse it as [pixel][8]
  float* sOut = lds;
#pragma unroll
  for (int i = 0; i < LSTRIP; ++i) sOut[((strip * LSTRIP + i) * LTW + col) * LC + wave] = apply_act(acc[i] * sc + sh, act);
  __syncthreads();
  for (int pix = tid; pix < LTH * LTW; pix += 512) {
    const int py = pix / LTW, px = pix % LTW;
    const int y = ty * LTH + py, x = tx * LTW + px;
    if (y < H && x < W) {
      float v[LC];
      Vec<float>::load(sOut + pix * LC, v);
      Vec<float>::load(sOut + pix * LC + 4, v + 4);
      T* dp = out + ((long long)(b * H + y) * W + x) * out_ld + c0;
#pragma unroll
      for (int q = 0; q < VPP; ++q) Vec<T>::store(dp + q * VE, v + q * VE);
    }
  }
}

// ---- large-kernel depthwise on the matrix cores (bf16): banded Toeplitz GEMM per kernel row ------------
// The VALU kernel above reaches 53 TFLOP/s (a third of the f32 vector peak) and is the single most expensive
// non-GEMM kernel of the forward.  Per channel, one kernel row ky is a 1-D correlation along x, i.e. a product with a
// banded Toeplitz matrix:
//     out[y][x] += sum_x' in[y + ky][x0 - LM + x'] * T_ky[x'][x],     T_ky[x'][x] = w[ky][x' - (LM - HALO) - x]
// which is the MFMA  D[i][j] += sum_k A[i][k] B[k][j]  with i = output row (16), j = output column (16), k = x'
// (16 + LM + HALO <= 64 input columns: one or two K = 32 steps).  About half of the MACs are useful (the band), against
// 1/16 for the diagonal trick of the 3x3 kernel: 31x31 needs 62 MFMAs per 16x16 output tile of one channel.
//   * workgroup = 8 waves = 8 channels of a 32 x 32 output patch; the input patch + halo is staged ONCE, transposed
//     NHWC -> per-channel planes of bf16 in LDS (row pitch an odd number of 16-byte slots: the 16-row A-fragment read is
//     conflict-free, every fragment is one aligned ds_read_b128);
//   * the B fragments (the Toeplitz bands in MFMA operand layout, [C][k][NH][64 lanes][8] bf16) are precomputed on the
//     host at weight-packing time; a wave streams its channel's 62 KB once per patch, fully coalesced, prefetched a
//     few kernel rows ahead, and applies each fragment to its 4 output tiles;
//   * epilogue: BN scale/shift + activation, results written in place into the wave's own plane, then a coalesced
//     NHWC copy-out (16 bytes = 8 channels per pixel).
// Row pitch of a plane in 16-byte slots.  The A fragment of lane (j, q) is 16 bytes at row j, chunk q; `ds_read_b128` is banked over
// 16-lane groups that pair the lanes j in {0-3, 12-15} of chunk q with the lanes j in {4-11} of chunk q + 1 (MI355X_MICROARCH.md, LDS
// table).  With S = 2 (mod 4) slots per row the first eight cover the EVEN slots of the 256-byte bank row once and the others, one
// slot further, the ODD ones: conflict-free.  (Round 1 padded to an ODD S, which is conflict-free for 16 lanes of ONE chunk only:
// SQ_LDS_BANK_CONFLICT 46 % of the LDS cycles of a kernel that re-reads an A fragment for every one of its MFMAs.)
constexpr int dwl_pitch(int slots) { return (slots % 4 == 2) ? slots : dwl_pitch(slots + 1); }

template <typename HT, int K, int TH, int TW>
__global__ __launch_bounds__(512) void dwlarge_mfma_kernel(const bf16_t* __restrict__ in, int in_ld, const bf16_t* __restrict__ tb,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           bf16_t* __restrict__ out, int out_ld, int B, int H, int W, int C, int act) {
  constexpr int HALO = (K - 1) / 2;
  constexpr int LM = (HALO + 7) / 8 * 8;                  // left margin, 16-byte aligned
  constexpr int NH = (16 + LM + HALO + 31) / 32;          // K = 32 steps per kernel row
  constexpr int NTX = TW / 16, NT = (TH / 16) * NTX;     // 16 x 16 output tiles per wave
  constexpr int PH = TH + K - 1;                          // plane rows
  constexpr int PWV = TW + LM + HALO;                     // plane columns holding real data
  constexpr int PWA = (TW - 16) + NH * 32;                // columns an A fragment may touch (zero weights beyond PWV)
  constexpr int PW = (PWA > PWV ? PWA : PWV);
  constexpr int PITCH = dwl_pitch((PW + 7) / 8) * 8;      // elements; 2 (mod 4) 16-byte slots: see dwl_pitch
  constexpr int PLANE = PH * PITCH;                       // elements per channel plane
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  bf16_t* planes = reinterpret_cast<bf16_t*>(dsm);        // [8][PH][PITCH]
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int cgs = C / 8;
  int bid = xcd_remap(blockIdx.x, gridDim.x);             // the channel groups of one patch share an XCD's L2
  const int cg = bid % cgs; bid /= cgs;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c0 = cg * 8;
  const int y0 = ty * TH - HALO, x0 = tx * TW - LM;       // image coordinates of plane (0, 0)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;

  // B fragments of the first kernel rows are requested before the staging so their latency hides under it
  const bf16_t* __restrict__ tbc = tb + (long long)(c0 + wave) * K * NH * 64 * 8 + lane * 8;
  constexpr int PF = 4;                                   // kernel rows per prefetch group
  s16x8 bnext[PF][NH];
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int h = 0; h < NH; ++h) bnext[u][h] = *reinterpret_cast<const s16x8*>(tbc + (long long)((u < K ? u : K - 1) * NH + h) * 512);

  // ---- staging: NHWC -> 8 bf16 planes, zeros outside the image and in the k-padding columns ----------------
  for (int i = tid; i < PH * PITCH; i += 512) {
    const int py = i / PITCH, px = i - py * PITCH;
    const int y = y0 + py, x = x0 + px;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (px < PWV && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      v = *reinterpret_cast<const u32x4*>(in + ((long long)(b * H + y) * W + x) * in_ld + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      planes[(2 * e) * PLANE + i] = (bf16_t)(v[e] & 0xffffu);
      planes[(2 * e + 1) * PLANE + i] = (bf16_t)(v[e] >> 16);
    }
  }
  __syncthreads();

  // ---- banded Toeplitz GEMM: this wave = channel c0 + wave, (TH/16) x (TW/16) output tiles of 16 x 16 -------------------
  const bf16_t* pl = planes + wave * PLANE;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int abase = j * PITCH + q * 8;                    // lane's row (tile row j = A row i) and k-chunk
#pragma unroll 1
  for (int ky0 = 0; ky0 < K; ky0 += PF) {
    s16x8 bcur[PF][NH];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int h = 0; h < NH; ++h) bcur[u][h] = bnext[u][h];
    if (ky0 + PF < K) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kyn = min(ky0 + PF + u, K - 1);
#pragma unroll
        for (int h = 0; h < NH; ++h) bnext[u][h] = *reinterpret_cast<const s16x8*>(tbc + (long long)(kyn * NH + h) * 512);
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ky = ky0 + u;
      if (ky < K) {                                       // uniform
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          s16x8 afr[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t)
            afr[t] = *reinterpret_cast<const s16x8*>(pl + abase + ((t / NTX) * 16 + ky) * PITCH + (t % NTX) * 16 + h * 32);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma16<HT>(afr[t], bcur[u][h], acc[t]);
        }
      }
    }
  }
  // ---- epilogue in place: D lane = (column j, rows 4q .. 4q+3) of each tile -----------------------------------
  const float sc = scale[c0 + wave], sh = shift[c0 + wave];
  bf16_t* plw = planes + wave * PLANE;
  with_act(act, [&](auto A) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        plw[((t / NTX) * 16 + 4 * q + r) * PITCH + (t % NTX) * 16 + j] = to_bits<HT>(from_f32<HT>(act_c<decltype(A)::value>(acc[t][r] * sc + sh)));
  });
  __syncthreads();
  // ---- coalesced NHWC copy-out -------------------------------------------------------------------------------
  for (int i = tid; i < TH * TW; i += 512) {
    const int py = i / TW, px = i - py * TW;
    const int y = ty * TH + py, x = tx * TW + px;
    if (y < H && x < W) {
      u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = (uint32_t)planes[(2 * e) * PLANE + py * PITCH + px] | ((uint32_t)planes[(2 * e + 1) * PLANE + py * PITCH + px] << 16);
      *reinterpret_cast<u32x4*>(out + ((long long)(b * H + y) * W + x) * out_ld + c0) = v;
    }
  }
}

template <typename HT, int K, int TH, int TW>
hipError_t launch_dwlarge_mfma(const void* in, int in_ld, const void* tb, const float* scale, const float* shift, void* out, int out_ld,
                               int B, int H, int W, int C, int act, hipStream_t s) {
  constexpr int HALO = (K - 1) / 2, LM = (HALO + 7) / 8 * 8, NH = (16 + LM + HALO + 31) / 32;
  constexpr int PH = TH + K - 1, PWV = TW + LM + HALO, PWA = (TW - 16) + NH * 32, PW = (PWA > PWV ? PWA : PWV);
  constexpr int PITCH = dwl_pitch((PW + 7) / 8) * 8;
  constexpr size_t lds = (size_t)8 * PH * PITCH * 2;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)dwlarge_mfma_kernel<HT, K, TH, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr = true;
  }
  long long blocks = (long long)B * cdiv(H, TH) * cdiv(W, TW) * (C / 8);
  hipLaunchKernelGGL((dwlarge_mfma_kernel<HT, K, TH, TW>), dim3((unsigned)blocks), dim3(512), lds, s, (const bf16_t*)in, in_ld, (const bf16_t*)tb, scale,
                     shift, (bf16_t*)out, out_ld, B, H, W, C, act);
  return hipSuccess;
}

// generic odd k (3..31): 16x16 tile, weights staged in LDS
constexpr int LT = 16;

template <typename T>
__global__ __launch_bounds__(256) void dwlarge_generic_kernel(const T* __restrict__ in, int in_ld,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, T* __restrict__ out,
                                                              int out_ld, int B, int H, int W, int C, int k, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int halo = (k - 1) / 2;
  const int TP = LT + k - 1;
  float* sIn = lds;                     // [TP][TP][LC]
  float* sW = lds + TP * TP * LC;       // [ky*k+kx][LC]
  const int tiles_x = (W + LT - 1) / LT, tiles_y = (H + LT - 1) / LT;
  const int cgs = C / LC;
  int bid = blockIdx.x;
  const int cg = bid % cgs; bid /= cgs;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c0 = cg * LC;
  const int y0 = ty * LT - halo, x0 = tx * LT - halo;
  const int tid = threadIdx.x;
  for (int i = tid; i < k * k * LC; i += 256) {
    const int c = i % LC, t = i / LC, ky = t / k, kx = t % k;
    sW[i] = w[(long long)(c0 + c) * k * k + kx * k + ky];
  }
  constexpr int VE = Vec<T>::N;
  constexpr int VPP = LC / VE;
  for (int i = tid; i < TP * TP * VPP; i += 256) {
    int v = i % VPP, pix = i / VPP;
    int py = pix / TP, px = pix % TP;
    int y = y0 + py, x = x0 + px;
    float vals[VE];
    if (y >= 0 && y < H && x >= 0 && x < W) {
      Vec<T>::load(in + ((long long)(b * H + y) * W + x) * in_ld + c0 + v * VE, vals);
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) vals[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) sIn[pix * LC + v * VE + e] = vals[e];
  }
  __syncthreads();
  const int c = tid & 7;
  const int col = (tid >> 3) & 15;
  const int r0 = (tid >> 7) * 8;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int kx = 0; kx < k; ++kx) {
    const float* colp = sIn + ((r0 * TP) + col + kx) * LC + c;
    float win[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) win[i + 1] = colp[i * TP * LC];
    for (int ky = 0; ky < k; ++ky) {
#pragma unroll
      for (int i = 0; i < 7; ++i) win[i] = win[i + 1];
      win[7] = colp[(ky + 7) * TP * LC];
      const float wv = sW[(ky * k + kx) * LC + c];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(win[i], wv, acc[i]);
    }
  }
  const float sc = scale[c0 + c], sh = shift[c0 + c];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int y = ty * LT + r0 + i, x = tx * LT + col;
    if (y < H && x < W) out[((long long)(b * H + y) * W + x) * out_ld + c0 + c] = from_f32<T>(apply_act(acc[i] * sc + sh, act));
  }
}

template <typename T, int K>
hipError_t launch_dwlarge(const void* in, int in_ld, const float* w, const float* scale, const float* shift, void* out,
                          int out_ld, int B, int H, int W, int C, int act, hipStream_t s) {
  constexpr int TPH = LTH + K - 1, TPW = LTW + K - 1;
  size_t lds = (size_t)LC * TPH * TPW * sizeof(float);
  if (lds < (size_t)LTH * LTW * LC * sizeof(float)) lds = (size_t)LTH * LTW * LC * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)dwlarge_kernel<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long long blocks = (long long)B * cdiv(H, LTH) * cdiv(W, LTW) * (C / LC);
  hipLaunchKernelGGL((dwlarge_kernel<T, K>), dim3((unsigned)blocks), dim3(512), lds, s, (const T*)in, in_ld, w, scale, shift,
                     (T*)out, out_ld, B, H, W, C, act);
  return hipSuccess;
}

template <typename T>
hipError_t launch_dwlarge_any(const void* in, int in_ld, const float* w, const float* scale, const float* shift, void* out,
                              int out_ld, int B, int H, int W, int C, int k, int act, hipStream_t s) {
  switch (k) {
    case 7: return launch_dwlarge<T, 7>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, act, s);
    case 15: return launch_dwlarge<T, 15>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, act, s);
    case 31: return launch_dwlarge<T, 31>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, act, s);
    default: break;
  }
  const int TP = LT + k - 1;
  size_t lds = (size_t)(TP * TP * LC + k * k * LC) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)dwlarge_generic_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long long blocks = (long long)B * cdiv(H, LT) * cdiv(W, LT) * (C / LC);
  hipLaunchKernelGGL(dwlarge_generic_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, s, (const T*)in, in_ld, w, scale, shift,
                     (T*)out, out_ld, B, H, W, C, k, act);
  return hipSuccess;
}

}  // namespace

// dw3x3_stream.hip
int cfp_dws_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld, float* partial,
                   int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, int dtype, cfp_stream_t stream,
                   const char* who);
int cfp_dws_strips(int B, int Ho, int Wo, int C, int stride);
void cfp_dws_debug_set(int key, int value);
// dw3x3_slide.hip
int cfp_dwl_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld, float* partial,
                   const float* w_red, int RD, float* hpart, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo,
                   int act, int dtype, cfp_stream_t stream, const char* who);
int cfp_dwl_slots(int B, int Ho, int Wo, int C, int stride);
void cfp_dwl_debug_set(int value);
// dw3x3_rows.hip: float32 storage, register-sliding rows (round 5)
int cfp_dwr_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld, float* partial,
                   const float* w_red, int RD, float* hpart, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, cfp_stream_t stream, const char* who);
int cfp_dwr_slots(int B, int H, int W, int Ho, int Wo, int C, int stride, int* ncb);
void cfp_dwr_debug_set(int key, int value);
int cfp_dwr_launch_slots(int B, int H, int W, int Ho, int Wo, int C, int stride, int in_ld, int out_ld);

namespace {
// Work decomposition of the depthwise 3x3 kernel: CVB channel vectors and R output rows per workgroup.
// Among the configurations whose input strip fits 64 KB of LDS, take the one with the lowest
// (halo read amplification) x (penalty for leaving CUs idle).
struct DwPlan { int cvb, R, nstrips; size_t lds; };
int g_dw_force_cvb = 0, g_dw_force_R = 0, g_dw_valu = 0;   // cfp_debug_set keys 3 / 4 / 5 (tools/dw_bench.py)
// key 6: 2 = dw3x3_slide_kernel (round 3, second design: register window over input columns, one barrier per workgroup),
//        0 = dw3x3_stream_kernel (round 3: persistent, LDS-DMA staged, row steps behind counted vmcnt waits; bit-identical outputs), 1 = dw3x3_mfma_kernel.
// The pipelined kernel is correct and MEASURED SLOWER at batch 8 (24 vs 19 us at 30x40x816, profiles/r3_dw3x3_stream.md): its steps
// run in lock-step (all 8 waves read LDS, then all run the SiLU), and per-lane LDS-DMA costs ~300 cycles per 1 KB instruction.  The
// sliding-window kernel built on those findings is the default: 14 vs 19 us at 30x40x816, 8.8 vs 12.6 at 15x20x1392.
int g_dw_no_stream = 2;
inline DwPlan dw_plan(int B, int Ho, int Wo, int C, int stride, int ve, bool out_tile = false) {
  DwPlan best{8, 1, Ho, 0};
  double bc = 1e30;
  const int CV = C / ve;
  const int cols_in = (Wo - 1) * stride + 3;
  const int cvbs[2] = {16, 8};
  for (int k = 0; k < 2; ++k) {
    const int cvb = cvbs[k];
    if (g_dw_force_cvb && cvb != g_dw_force_cvb) continue;
    for (int R = 1; R <= Ho; ++R) {
      if (g_dw_force_R && R != g_dw_force_R) continue;
      const int rows_in = (R - 1) * stride + 3;
      size_t lds = (size_t)rows_in * cols_in * (out_tile ? cvb * 16 + 16 : cvb * 16);   // out_tile: the MFMA kernel's padded pixel pitch
      if (out_tile) {                                                                    // + its unit table, sized as the kernel fills it:
        const int nmain = Wo >> 4, tw = Wo & 15;                                          // row tails narrower than 16 are gathered from RB rows,
        const int RB = tw == 0 ? 1 : (16 % tw == 0 ? 16 / tw : 1);                        // and a row block has RB * nmain + 1 units even when the
        const int upb = RB * nmain + (tw ? 1 : 0);                                        // strip holds fewer than RB rows (round 3: the old bound
        lds += 16 + (size_t)cdiv(R, RB) * upb * 64;                                       // R * (ceil(Wo/16) + 1) overflowed LDS for Wo = 65, R = 1)
        lds = ((lds + 15) & ~(size_t)15) + (size_t)cvb * 8 * sizeof(float);                // + the channel sums of the squeeze-excite dot products
      }
      const size_t red = (size_t)(256 / cvb) * cvb * ve * sizeof(float);
      if (lds < red) lds = red;
      if (lds > 64 * 1024) break;
      const long long blocks = (long long)cdiv(CV, cvb) * B * cdiv(Ho, R);
      const double halo = (double)rows_in / (R * stride);
      const double fill = blocks >= 512 ? 1.0 : 512.0 / (double)blocks;
      const double pad = (double)(cdiv(CV, cvb) * cvb) / CV;
      const int units = R * cdiv(Wo, 4), nslot = 256 / cvb;          // runs of 4 output pixels over the pixel slots
      const double util = (double)(cdiv(units, nslot) * nslot) / units;
      const double c = halo * fill * pad * util;
      if (c < bc) { bc = c; best = DwPlan{cvb, R, cdiv(Ho, R), lds}; }
    }
  }
  if (best.lds == 0) {   // a single row does not fit (very wide map): still correct, LDS request will fail loudly
    best.lds = (size_t)3 * cols_in * best.cvb * 16;
  }
  return best;
}

int dw3x3_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld,
                 float* partial, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, int dtype,
                 cfp_stream_t stream, const char* who, const float* w_red = nullptr, int RD = 0, float* hpart = nullptr) {
  CFP_REQUIRE(in && w && out && scale && shift, CFP_EINVAL, std::string(who) + ": null pointer");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, std::string(who) + ": bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(stride == 1 || stride == 2, CFP_ESHAPE, std::string(who) + ": stride must be 1 or 2");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && in_ld % ve == 0 && out_ld % ve == 0 &&
                  in_ld >= C && out_ld >= C, CFP_ESHAPE, std::string(who) + ": bad shape");
  CFP_REQUIRE(aligned16(in) && aligned16(w) && aligned16(out) && aligned16(scale) && aligned16(shift), CFP_EINVAL,
              std::string(who) + ": pointers must be 16-byte aligned");
  const bool mfma = is16(dtype) && C % 16 == 0 && !g_dw_valu;
  CFP_REQUIRE(hpart == nullptr || ((mfma || dtype == CFP_F32) && w_red && RD > 0 && RD <= 64 && aligned16(w_red)), CFP_ESHAPE,
              std::string(who) + ": the squeeze-excite partials need R <= 64 and, in 16-bit storage, C % 16 == 0");
  if (dtype == CFP_F32 && act != 99) {
    // float32 storage (the default f16x3 mode): the register-sliding kernel (dw3x3_rows.hip), no LDS, no barrier
    const int rc = cfp_dwr_launch(in, in_ld, w, scale, shift, out, out_ld, partial, w_red, RD, hpart, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, act, stream, who);
    if (rc != 1) return rc;
  }
  if (mfma && g_dw_no_stream == 2 && act != 99) {
    // the sliding-window kernel (dw3x3_slide.hip): register window over input columns, one barrier per workgroup
    const int rc = cfp_dwl_launch(in, in_ld, w, scale, shift, out, out_ld, partial, w_red, RD, hpart, B, H, W, C, stride, pad_t, pad_l, Ho, Wo,
                                  act, dtype, stream, who);
    if (rc != 1) return rc;
  }
  if (mfma && !g_dw_no_stream && hpart == nullptr) {
    // the software-pipelined kernel (dw3x3_stream.hip): same arithmetic, load / compute / store overlapped inside a workgroup
    const int rc = cfp_dws_launch(in, in_ld, w, scale, shift, out, out_ld, partial, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, act, dtype, stream, who);
    if (rc != 1) return rc;
  }
  const DwPlan d = dw_plan(B, Ho, Wo, C, stride, ve, mfma);
  CFP_REQUIRE((long long)B * d.nstrips <= 65535, CFP_ESHAPE, std::string(who) + ": grid too large");
  CFP_REQUIRE(d.lds <= 64 * 1024, CFP_ESHAPE, std::string(who) + ": map too wide for the LDS strip");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(cdiv(C / ve, d.cvb), B * d.nstrips);
#define DW_LAUNCH(T, S, V)                                                                                                \
  do {                                                                                                                    \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      hipError_t e = hipFuncSetAttribute((const void*)dw3x3_kernel<T, S, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
      if (e != hipSuccess) { cfp_set_error(std::string(who) + ": " + hipGetErrorString(e)); return CFP_EHIP; }            \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((dw3x3_kernel<T, S, V>), grid, dim3(256), d.lds, s, (const T*)in, in_ld, (const T*)w, scale, shift,  \
                       (T*)out, out_ld, partial, B, H, W, C, pad_t, pad_l, Ho, Wo, act, d.R, d.nstrips, w_red, RD, hpart);  \
  } while (0)
#define DW_CVB(T, S) do { if (d.cvb == 16) DW_LAUNCH(T, S, 16); else DW_LAUNCH(T, S, 8); } while (0)
#define DWM_LAUNCH(HH, S, V)                                                                                                  \
  do {                                                                                                                    \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      hipError_t e = hipFuncSetAttribute((const void*)dw3x3_mfma_kernel<HH, S, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
      if (e != hipSuccess) { cfp_set_error(std::string(who) + ": " + hipGetErrorString(e)); return CFP_EHIP; }            \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((dw3x3_mfma_kernel<HH, S, V>), grid, dim3(256), d.lds, s, (const bf16_t*)in, in_ld, (const bf16_t*)w, scale, \
                       shift, (bf16_t*)out, out_ld, partial, B, H, W, C, pad_t, pad_l, Ho, Wo, act, d.R, d.nstrips, w_red, RD, hpart, \
                       (int)(d.lds - (size_t)V * 8 * sizeof(float)));                                                       \
  } while (0)
#define DWM_CVB(HH, S) do { if (d.cvb == 16) DWM_LAUNCH(HH, S, 16); else DWM_LAUNCH(HH, S, 8); } while (0)
  if (mfma && dtype == CFP_F16) { if (stride == 1) DWM_CVB(f16_t, 1); else DWM_CVB(f16_t, 2); }
  else if (mfma) { if (stride == 1) DWM_CVB(bf16_t, 1); else DWM_CVB(bf16_t, 2); }
  else if (dtype == CFP_BF16) { if (stride == 1) DW_CVB(bf16_t, 1); else DW_CVB(bf16_t, 2); }
  else if (dtype == CFP_F16) { if (stride == 1) DW_CVB(f16_t, 1); else DW_CVB(f16_t, 2); }
  else { if (stride == 1) DW_CVB(float, 1); else DW_CVB(float, 2); }
#undef DWM_CVB
#undef DWM_LAUNCH
#undef DW_CVB
#undef DW_LAUNCH
  return cfp_check_launch(who);
}
}  // namespace

void cfp_dw_debug_set(int key, int value) {
  if (key == 3) g_dw_force_cvb = value; else if (key == 4) g_dw_force_R = value; else if (key == 5) g_dw_valu = value;
  else if (key == 6) g_dw_no_stream = value; else if (key == 9) cfp_dwl_debug_set(value); else if (key == 10 || key == 11) cfp_dwr_debug_set(key, value);
  else cfp_dws_debug_set(key, value);
}

extern "C" int cfp_dwconv3x3_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                                  void* out, int out_ld, int B, int H, int W, int C, int stride, int pad_t, int pad_l,
                                  int Ho, int Wo, int act, int dtype, cfp_stream_t stream) {
  return dw3x3_launch(in, in_ld, w, scale, shift, out, out_ld, nullptr, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, act, dtype,
                      stream, "cfp_dwconv3x3_nhwc");
}

extern "C" int cfp_dwconv3x3_strips(int B, int Ho, int Wo, int C, int stride, int dtype) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (stride != 1 && stride != 2)) return 0;
  if (dtype == CFP_F32) {
    const int n = cfp_dwr_slots(B, (Ho - 1) * stride + 3, (Wo - 1) * stride + 3, Ho, Wo, C, stride, nullptr);
    if (n > 0) return n;
  }
  if (is16(dtype) && C % 16 == 0 && !g_dw_valu && g_dw_no_stream == 2) {
    const int n = cfp_dwl_slots(B, Ho, Wo, C, stride);
    if (n > 0) return n;
  }
  if (is16(dtype) && C % 16 == 0 && !g_dw_valu && !g_dw_no_stream) {
    const int n = cfp_dws_strips(B, Ho, Wo, C, stride);
    if (n > 0) return n;
  }
  return dw_plan(B, Ho, Wo, C, stride, vec_elems(dtype), is16(dtype) && C % 16 == 0 && !g_dw_valu).nstrips;
}

extern "C" int cfp_dwconv3x3_launch_slots(int B, int H, int W, int Ho, int Wo, int C, int stride, int in_ld, int out_ld, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (stride != 1 && stride != 2) || dtype != CFP_F32) return 0;
  return cfp_dwr_launch_slots(B, H, W, Ho, Wo, C, stride, in_ld, out_ld);
}

extern "C" int cfp_dwconv3x3_sum_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                                      void* out, int out_ld, float* partial, int B, int H, int W, int C, int stride,
                                      int pad_t, int pad_l, int Ho, int Wo, int act, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(partial, CFP_EINVAL, "cfp_dwconv3x3_sum_nhwc: null pointer");
  return dw3x3_launch(in, in_ld, w, scale, shift, out, out_ld, partial, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, act, dtype,
                      stream, "cfp_dwconv3x3_sum_nhwc");
}

extern "C" int cfp_dwconv3x3_se_parts(int B, int Ho, int Wo, int C, int stride, int dtype) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (stride != 1 && stride != 2)) return 0;
  if (dtype == CFP_F32 && C % 8 == 0) {     // float32 storage (the default f16x3 mode): the register-sliding kernel, else the LDS-strip one
    int ncb = 0;
    const int n = cfp_dwr_slots(B, (Ho - 1) * stride + 3, (Wo - 1) * stride + 3, Ho, Wo, C, stride, &ncb);
    if (n > 0) return n * ncb;
    const DwPlan d = dw_plan(B, Ho, Wo, C, stride, 4, false);
    return d.nstrips * cdiv(C / 4, d.cvb);
  }
  if (!is16(dtype) || C % 16 != 0 || g_dw_valu) return 0;
  if (g_dw_no_stream == 2) {
    const int n = cfp_dwl_slots(B, Ho, Wo, C, stride);
    if (n > 0) return n * cdiv(C, 64);
  }
  const DwPlan d = dw_plan(B, Ho, Wo, C, stride, 8, true);
  return d.nstrips * cdiv(C / 8, d.cvb);
}

extern "C" int cfp_dwconv3x3_se_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld,
                                     const float* w_reduce, int R, float* hpart, int B, int H, int W, int C, int stride, int pad_t,
                                     int pad_l, int Ho, int Wo, int act, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(hpart && w_reduce, CFP_EINVAL, "cfp_dwconv3x3_se_nhwc: null pointer");
  return dw3x3_launch(in, in_ld, w, scale, shift, out, out_ld, nullptr, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, act, dtype, stream,
                      "cfp_dwconv3x3_se_nhwc", w_reduce, R, hpart);
}

extern "C" int cfp_dwconv_large_nhwc(const void* in, int in_ld, const float* w, const float* scale,
                                     const float* shift, void* out, int out_ld, int B, int H, int W, int C, int k,
                                     int act, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(in && w && out && scale && shift, CFP_EINVAL, "cfp_dwconv_large_nhwc: null pointer");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_dwconv_large_nhwc: bad dtype");
  const int ve = vec_elems(dtype);
  CFP_REQUIRE(k >= 3 && k <= 31 && (k & 1), CFP_ESHAPE, "cfp_dwconv_large_nhwc: k must be odd, 3..31");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && in_ld % ve == 0 && out_ld % ve == 0 && in_ld >= C &&
                  out_ld >= C, CFP_ESHAPE, "cfp_dwconv_large_nhwc: bad shape");
  CFP_REQUIRE(aligned16(in) && aligned16(out), CFP_EINVAL, "cfp_dwconv_large_nhwc: pointers must be 16-byte aligned");
  CFP_REQUIRE((long long)B * cdiv(H, 16) * cdiv(W, 16) * (C / 8) < (1ll << 31), CFP_ESHAPE, "cfp_dwconv_large_nhwc: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = dtype == CFP_BF16
      ? launch_dwlarge_any<bf16_t>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, k, act, s)
      : dtype == CFP_F16 ? launch_dwlarge_any<f16_t>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, k, act, s)
      : launch_dwlarge_any<float>(in, in_ld, w, scale, shift, out, out_ld, B, H, W, C, k, act, s);
  if (e != hipSuccess) { cfp_set_error(std::string("cfp_dwconv_large_nhwc: ") + hipGetErrorString(e)); return CFP_EHIP; }
  return cfp_check_launch("cfp_dwconv_large_nhwc");
}


extern "C" size_t cfp_dwconv_large_toeplitz_elems(int C, int k) {
  if (C <= 0 || (k != 7 && k != 15 && k != 31)) return 0;
  const int halo = (k - 1) / 2, lm = (halo + 7) / 8 * 8, nh = (16 + lm + halo + 31) / 32;
  return (size_t)C * k * nh * 64 * 8;
}

namespace {
// band table of cfp_dwconv_large_mfma_nhwc from float32 weights [C][k][k] (ky, kx) that live on the device (training: they change every
// step): out[c][ky][h][lane][e] = w[c][ky'][kx'] for kx = 32 h + 8 (lane / 16) + e - (LM - halo) - lane % 16 inside [0, k), else 0;
// flip: (ky', kx') = (k - 1 - ky, k - 1 - kx), the data gradient's kernel.  One launch instead of five torch ones.
template <typename T>
__global__ __launch_bounds__(256) void toeplitz_bands_kernel(const float* __restrict__ w, T* __restrict__ out, int k, int nh, int lm_minus_halo,
                                                             int flip, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long long t = i >> 9;
    const int h = (int)(t % nh); t /= nh;
    const int ky = (int)(t % k);
    const long long c = t / k;
    const int kx = 32 * h + 8 * (lane >> 4) + e - lm_minus_halo - (lane & 15);
    float v = 0.f;
    if (kx >= 0 && kx < k) v = w[(c * k + (flip ? k - 1 - ky : ky)) * k + (flip ? k - 1 - kx : kx)];
    out[i] = from_f32<T>(v);
  }
}
}  // namespace

extern "C" int cfp_dwconv_large_toeplitz(const float* w, void* out, int C, int k, int flip, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(w && out, CFP_EINVAL, "cfp_dwconv_large_toeplitz: null pointer");
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_dwconv_large_toeplitz: the band table is a 16-bit operand");
  CFP_REQUIRE(C > 0 && (k == 7 || k == 15 || k == 31), CFP_ESHAPE, "cfp_dwconv_large_toeplitz: k must be 7, 15 or 31");
  const int halo = (k - 1) / 2, lm = (halo + 7) / 8 * 8, nh = (16 + lm + halo + 31) / 32;
  const long long total = (long long)C * k * nh * 512;
  const int blocks = (int)std::min<long long>(2048, (total + 255) / 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == CFP_BF16) hipLaunchKernelGGL(toeplitz_bands_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)out, k, nh, lm - halo, flip, total);
  else hipLaunchKernelGGL(toeplitz_bands_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, w, (f16_t*)out, k, nh, lm - halo, flip, total);
  return cfp_check_launch("cfp_dwconv_large_toeplitz");
}

hipError_t dwlarge_x3_launch(const void* in, int in_ld, const void* toeplitz, long long table_elems, const float* scale, const float* shift, void* out,
                             int out_ld, int B, int H, int W, int C, int k, int act, hipStream_t s);      // dwlarge_x3.hip

extern "C" int cfp_dwconv_large_mfma_nhwc(const void* in, int in_ld, const void* toeplitz, const float* scale, const float* shift,
                                          void* out, int out_ld, int B, int H, int W, int C, int k, int act, int dtype,
                                          cfp_stream_t stream) {
  CFP_REQUIRE(in && toeplitz && out && scale && shift, CFP_EINVAL, "cfp_dwconv_large_mfma_nhwc: null pointer");
  if (dtype == CFP_F32X3) {      // float32 tensors, f16x3 matrix math, `toeplitz` = [hi table | lo table]: dwlarge_x3.hip
    CFP_REQUIRE(k == 7 || k == 15 || k == 31, CFP_ESHAPE, "cfp_dwconv_large_mfma_nhwc: k must be 7, 15 or 31");
    CFP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && in_ld >= C && out_ld >= C, CFP_ESHAPE,
                "cfp_dwconv_large_mfma_nhwc: bad shape");
    CFP_REQUIRE(aligned16(in) && aligned16(out) && aligned16(toeplitz), CFP_EINVAL, "cfp_dwconv_large_mfma_nhwc: pointers must be 16-byte aligned");
    CFP_REQUIRE((long long)B * cdiv(H, 32) * cdiv(W, 32) * (C / 4) < (1ll << 31), CFP_ESHAPE, "cfp_dwconv_large_mfma_nhwc: grid too large");
    hipError_t e3 = dwlarge_x3_launch(in, in_ld, toeplitz, (long long)cfp_dwconv_large_toeplitz_elems(C, k), scale, shift, out, out_ld, B, H, W, C, k, act,
                                      reinterpret_cast<hipStream_t>(stream));
    if (e3 != hipSuccess) { cfp_set_error(std::string("cfp_dwconv_large_mfma_nhwc: ") + hipGetErrorString(e3)); return CFP_EHIP; }
    return cfp_check_launch("cfp_dwconv_large_mfma_nhwc");
  }
  CFP_REQUIRE(is16(dtype), CFP_EINVAL, "cfp_dwconv_large_mfma_nhwc: bf16 / f16 or CFP_F32X3 (plain f32 uses cfp_dwconv_large_nhwc)");
  CFP_REQUIRE(k == 7 || k == 15 || k == 31, CFP_ESHAPE, "cfp_dwconv_large_mfma_nhwc: k must be 7, 15 or 31");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && in_ld % 8 == 0 && out_ld % 8 == 0 && in_ld >= C && out_ld >= C,
              CFP_ESHAPE, "cfp_dwconv_large_mfma_nhwc: bad shape");
  CFP_REQUIRE(aligned16(in) && aligned16(out) && aligned16(toeplitz), CFP_EINVAL, "cfp_dwconv_large_mfma_nhwc: pointers must be 16-byte aligned");
  CFP_REQUIRE((long long)B * cdiv(H, 32) * cdiv(W, 32) * (C / 8) < (1ll << 31), CFP_ESHAPE, "cfp_dwconv_large_mfma_nhwc: grid too large");
  (void)0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // (k = 31 as 32 x 32 pixel tiles -- 79 KB of LDS, two workgroups per CU -- is the f16x3 kernel's default since round 5; in 16-bit storage the whole
  // step measured 2.219 vs 2.199 ms with four batches in flight, i.e. no gain: the 64 x 32 tile stays here)
#define DWL(HH) (k == 31 ? launch_dwlarge_mfma<HH, 31, 64, 32>(in, in_ld, toeplitz, scale, shift, out, out_ld, B, H, W, C, act, s) \
               : k == 15 ? launch_dwlarge_mfma<HH, 15, 32, 32>(in, in_ld, toeplitz, scale, shift, out, out_ld, B, H, W, C, act, s) \
                         : launch_dwlarge_mfma<HH, 7, 32, 32>(in, in_ld, toeplitz, scale, shift, out, out_ld, B, H, W, C, act, s))
  hipError_t e = dtype == CFP_F16 ? DWL(f16_t) : DWL(bf16_t);
#undef DWL
  if (e != hipSuccess) { cfp_set_error(std::string("cfp_dwconv_large_mfma_nhwc: ") + hipGetErrorString(e)); return CFP_EHIP; }
  return cfp_check_launch("cfp_dwconv_large_mfma_nhwc");
}
