// Depthwise 3x3 (+ BN + activation + squeeze-excite channel sums / reduce-FC partials), 16-bit storage: the SLIDING-WINDOW form.
//
// Reference op: timm InvertedResidual `conv_dw -> bn2 -> act` (oracle/cfpnet_oracle.py, encoder()), 24 launches per forward; the kernel
// BASELINE.json's north_star puts the ">= 60 % of measured HBM roofline" target on.  What rounds 2-3 measured about its predecessors
// (profiles/r3_dw3x3_stream.md): the diagonal-weight MFMA formulation is right for the arithmetic (no unpacking, VALU left with the
// epilogue), but (i) it read 20 bytes of LDS per output element -- every tap fetched its own B fragment --, (ii) all waves of a
// workgroup ran LDS-read phase and SiLU phase in lock-step behind per-step barriers, (iii) per-lane LDS-DMA costs ~300 cycles per
// 1 KB instruction.  This kernel keeps the MFMA formulation and changes the three:
//
//   * a wave owns a RUN of 16 consecutive output ROWS of one 16-channel group (MFMA column j = row y0 + j) and SLIDES along x.  Taps
//     are paired along y -- (dy0, dy1) and (dy2, -) -- so the two B fragments of an input COLUMN serve three output columns (as
//     dx = 2, 1, 0) with three different diagonal weight fragments: per output column 2 fragment reads (2 KB per 256 outputs instead
//     of 5 KB) and 6 MFMAs (instead of 5); the window of three input columns lives in registers (loop unrolled by 3);
//   * ONE barrier per workgroup: the 4 waves (= 64 channels = 128 bytes per pixel, whole cache lines) load the task's input patch
//     (18 rows x XS + 2 columns) with plain 16-byte global loads, all in flight together, store it to LDS, synchronise once, and then
//     every wave slides on its own -- waves and workgroups drift apart, so LDS reads, MFMAs, SiLU and stores of different waves overlap;
//   * results leave from the accumulator registers (8-byte stores: a lane holds 4 consecutive channels of one pixel).
//
// Arithmetic per output element is the same MFMA products in a different summation order (6 MFMAs of 2 taps instead of 5), float32
// accumulation: results equal dw3x3_mfma_kernel's up to float32 re-association, i.e. almost always bit-identical after rounding to
// 16 bits (tests: <= 1 ulp of the storage type, and bit-exact on small integers).
#include <algorithm>

#include "common.h"

namespace {

struct DwlP {
  const bf16_t* in; const bf16_t* w; const float* scale; const float* shift; bf16_t* out;
  float* partial;            // [B][nslot][C] channel sums per (y-run, x-segment) task, or null
  const float* w_red; float* hpart; int RD;     // squeeze-excite reduce FC [RD][C] and its partial dot products [B][nslot * ncb][RD], or null
  int in_ld, out_ld;
  int B, H, W, C, pad_t, pad_l, Ho, Wo, act;
  int XS;                    // output columns per task
  int nxs, nyr, ncb;         // x-segments, y-runs (16 output rows), 64-channel blocks
  int colsA;                 // allocated input columns per LDS row = (XS - 1) * stride + 3
  int rowpitch;              // LDS bytes per input row: colsA * 144 rounded up to 16 x odd (16 consecutive rows hit 16 different bank groups)
  FastDiv dcols;             // item -> (row, column) of the patch
  float* dbg;                // diagnostic stamps [workgroup][8] (tools/dw_bench.py --stamps), or null; they feed no output value
};

constexpr int DWL_PP = 144;  // pixel pitch: 8 chunks of 16 bytes + 16 bytes of padding

template <typename HT, int STRIDE>
__global__ __launch_bounds__(256, 4) void dw3x3_slide_kernel(DwlP p) {
  constexpr int ROWS = 15 * STRIDE + 3;          // input rows of a 16-row run
  constexpr int NLD = STRIDE == 1 ? 8 : 10;      // 16-byte loads per thread (ROWS * colsA * 8 <= 256 * NLD, checked by the host plan)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave = 16-channel group of the block
  const int j = lane & 15, q = lane >> 4;
  // task = (image, channel block, y-run, x-segment); XCD-aware order: x-segments / y-runs / channel blocks of one image share halos and lines
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int xs = t % p.nxs; t /= p.nxs;
  const int yr = t % p.nyr; t /= p.nyr;
  const int cb = t % p.ncb;
  const int b = t / p.ncb;
  const int CV = p.C >> 3, cv0 = cb * 8;
  const int y0 = yr * 16, ny = min(16, p.Ho - y0);
  const int x0 = xs * p.XS, nx = min(p.XS, p.Wo - x0);
  const int iy0 = y0 * STRIDE - p.pad_t, ix0 = x0 * STRIDE - p.pad_l;
  const int nrows_in = (ny - 1) * STRIDE + 3, ncols_in = (nx - 1) * STRIDE + 3;
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tr0 = 0;
  if (p.dbg) { tk0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- per-lane constants: requested first, their latency hides under the patch loads ------------------------------------------
  const int cbase = (cv0 + 2 * g) * 8;
  const bool g_ok = cbase < p.C;
  // k-chunk q of the MFMA's reduction axis <-> (row of the tap pair, 8-channel half of the group).  Stride 1 puts the ROW in the low
  // bit: the 16-lane groups ds_read_b128 is banked over mix lanes of q and q ^ 1 ({0-3, 12-15, 20-27}, ...; MI355X_MICROARCH.md,
  // LDS table), and with the half in the low bit lanes (j = 12, q = 0) and (j = 11, q = 1) fell 16 bytes apart on the same 256-byte
  // bank row -- a 2-way conflict in every group, 25 % of the LDS cycles by SQ_LDS_BANK_CONFLICT.  With the row there, those two lanes
  // read the SAME 16 bytes (row 12 as dy 0 and as dy 1), which is a broadcast.  Stride 2 is conflict-free the other way round.
  constexpr bool ROW_LOW = STRIDE == 1;
  const int q_dy = ROW_LOW ? (q & 1) : (q >> 1), q_half = ROW_LOW ? (q >> 1) : (q & 1);
  short wv[3][2];                                // weight of tap (dy = 2 pr + q_dy, dx) for this lane's diagonal element
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int dy = 2 * pr + q_dy;
      const bool on = g_ok && dy < 3 && (j >> 3) == q_half;
      wv[dx][pr] = on ? (short)p.w[(long long)(min(dy, 2) * 3 + dx) * p.C + cbase + j] : (short)0;
    }
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (g_ok) { sc4 = *reinterpret_cast<const f32x4*>(p.scale + cbase + 4 * q); sh4 = *reinterpret_cast<const f32x4*>(p.shift + cbase + 4 * q); }
  // squeeze-excite: this thread's slice of the reduce FC (hidden unit tid >> 2, 16 of the block's 64 channels)
  f32x4 wq[4];
  const int hr = tid >> 2, hq = tid & 3;
  if (p.hpart != nullptr) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c = cv0 * 8 + hq * 16 + v * 4;
      wq[v] = (hr < p.RD && c < p.C) ? *reinterpret_cast<const f32x4*>(p.w_red + (long long)hr * p.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- 1. the input patch -> LDS: every load of the thread in flight, then the LDS stores, ONE barrier --------------------------
  {
    const unsigned char* img = reinterpret_cast<const unsigned char*>(p.in + (long long)b * p.H * p.W * p.in_ld + cv0 * 8);
    const int pixb = p.in_ld * 2;
    u32x4 v[NLD];
    int dst[NLD];
    const int nitems = ROWS * p.colsA * 8;
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * 256;
      const int ch = i & 7;
      unsigned ry, rx;
      fd_rowcol((unsigned)(i >> 3), p.dcols, ry, rx);
      const int iy = iy0 + (int)ry, ix = ix0 + (int)rx;
      const bool ok = i < nitems && (int)ry < nrows_in && (int)rx < ncols_in && cv0 + ch < CV && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      dst[n] = i < nitems ? (int)ry * p.rowpitch + (int)rx * DWL_PP + ch * 16 : -1;
      v[n] = u32x4{0u, 0u, 0u, 0u};
      if (ok) v[n] = *reinterpret_cast<const u32x4*>(img + ((long long)iy * p.W + ix) * pixb + ch * 16);
    }
#pragma unroll
    for (int n = 0; n < NLD; ++n)
      if (dst[n] >= 0) *reinterpret_cast<u32x4*>(lds + dst[n]) = v[n];
  }
  __syncthreads();
  if (p.dbg) tk1 = __builtin_amdgcn_s_memtime();

  // ---- 2. every wave slides along x on its own ---------------------------------------------------------------------------------------
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  if (g_ok) {
    s16x8 afr[3][2];                             // diagonal weight fragments: lane (row i = j, chunk q) holds A[i][k = 8q .. 8q+7]
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) afr[dx][pr][e] = (e == (j & 7)) ? wv[dx][pr] : (short)0;
    // B fragment pr of input column c: lane (j, q) reads 8 channels (half q_half of the group) of pixel (row j * STRIDE + dy, column c),
    // dy = 2 pr + q_dy clamped to the window (the matching A element is zero for dy = 3)
    int roff[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) roff[pr] = (j * STRIDE + min(2 * pr + q_dy, 2)) * p.rowpitch + (2 * g + q_half) * 16;
    const bool row_ok = j < ny;
    unsigned char* obase = reinterpret_cast<unsigned char*>(p.out + ((long long)(b * p.Ho + y0 + (row_ok ? j : 0)) * p.Wo + x0) * p.out_ld + cbase + 4 * q);
    const int opix = p.out_ld * 2;
    auto ldf = [&](int col, s16x8 (&f)[2]) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) f[pr] = *reinterpret_cast<const s16x8*>(lds + roff[pr] + col * DWL_PP);
    };
    // epilogue constants folded for the exponential: silu(z) = z / (1 + 2^(-z log2 e)), z = acc * sc + sh -> t = acc * sc2 + sh2
    constexpr float NL2E = -1.4426950408889634f;
    f32x4 sc2, sh2;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) { sc2[r4] = sc4[r4] * NL2E; sh2[r4] = sh4[r4] * NL2E; }
    const float vf = row_ok ? 1.f : 0.f;
    with_act(p.act, [&](auto A) {
      constexpr int ACT = decltype(A)::value;
      auto emit = [&](int x, const s16x8 (&f0)[2], const s16x8 (&f1)[2], const s16x8 (&f2)[2]) {
        // two independent accumulation chains (pairs (dy0, dy1) and (dy2, -)), added at the end
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        acc0 = mfma16<HT>(afr[0][0], f0[0], acc0); acc1 = mfma16<HT>(afr[0][1], f0[1], acc1);
        acc0 = mfma16<HT>(afr[1][0], f1[0], acc0); acc1 = mfma16<HT>(afr[1][1], f1[1], acc1);
        acc0 = mfma16<HT>(afr[2][0], f2[0], acc0); acc1 = mfma16<HT>(afr[2][1], f2[1], acc1);
        float y[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float a = acc0[r4] + acc1[r4];
          if constexpr (ACT == CFP_ACT_SILU) {
            // one fused multiply-add each for z and for the exponent (the VALU is what bounds this kernel: 8 of its ~30 instructions per
            // column are the quarter-rate v_exp / v_rcp, every other one counts)
            const float z = fmaf(a, sc4[r4], sh4[r4]);
            const float e = __builtin_amdgcn_exp2f(fmaf(a, sc2[r4], sh2[r4]));
            y[r4] = z * __builtin_amdgcn_rcpf(1.f + e);
          } else {
            y[r4] = act_c<ACT>(fmaf(a, sc4[r4], sh4[r4]));
          }
          csum[r4] = fmaf(y[r4], vf, csum[r4]);
        }
        uint2 pk;
        pk.x = pack2<HT>(y[0], y[1]);
        pk.y = pack2<HT>(y[2], y[3]);
        if (row_ok) *reinterpret_cast<uint2*>(obase + (long long)x * opix) = pk;
      };
      // window of input columns in four register slots: three in use by the current output column, the fourth being fetched one
      // column ahead (the LDS latency hides under the previous column's MFMAs and epilogue); loop unrolled by four
      s16x8 F0[2], F1[2], F2[2], F3[2];
      if constexpr (STRIDE == 1) {
        ldf(0, F0); ldf(1, F1); ldf(2, F2);
        for (int x = 0; x < nx; x += 4) {
          if (x + 1 < nx) ldf(x + 3, F3);
          emit(x, F0, F1, F2);
          if (x + 1 < nx) { if (x + 2 < nx) ldf(x + 4, F0); emit(x + 1, F1, F2, F3); }
          if (x + 2 < nx) { if (x + 3 < nx) ldf(x + 5, F1); emit(x + 2, F2, F3, F0); }
          if (x + 3 < nx) { if (x + 4 < nx) ldf(x + 6, F2); emit(x + 3, F3, F0, F1); }
        }
      } else {
        // stride 2: output column x reads input columns 2x, 2x+1, 2x+2 (slots: column c in slot c % 3); two new columns per step
        ldf(0, F0);
        for (int x = 0; x < nx; x += 3) {
          ldf(2 * x + 1, F1); ldf(2 * x + 2, F2);                 emit(x, F0, F1, F2);
          if (x + 1 < nx) { ldf(2 * x + 3, F0); ldf(2 * x + 4, F1); emit(x + 1, F2, F0, F1); }
          if (x + 2 < nx) { ldf(2 * x + 5, F2); ldf(2 * x + 6, F0); emit(x + 2, F1, F2, F0); }
        }
      }
    });
  }

  if (p.dbg && tid == 0) {
    tk2 = __builtin_amdgcn_s_memtime();
    const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
    float* d = p.dbg + (long long)blockIdx.x * 8;
    d[0] = (float)(tk1 - tk0); d[1] = (float)(tk2 - tk1); d[2] = 0.f; d[3] = (float)(tr0 & 0xffffff); d[4] = (float)(tr1 & 0xffffff); d[5] = 4.f;
  }
  // ---- 3. channel sums of the task (rows in the 16-lane butterfly's fixed order) ---------------------------------------------------------
  if (p.partial == nullptr && p.hpart == nullptr) return;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) csum[r4] += __shfl_xor(csum[r4], o, 64);
  }
  const int slot = yr * p.nxs + xs, nslot = p.nyr * p.nxs;
  if (p.partial != nullptr && g_ok && j == 0) {
    float* dstp = p.partial + ((long long)b * nslot + slot) * p.C + cbase + 4 * q;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) dstp[r4] = csum[r4];
  }
  if (p.hpart != nullptr) {
    __syncthreads();                             // every wave is done with the patch: its first bytes become the 64 channel sums
    float* cs = reinterpret_cast<float*>(lds);
    if (j == 0) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) cs[16 * g + 4 * q + r4] = g_ok ? csum[r4] : 0.f;
    }
    __syncthreads();
    float sdot = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cs + hq * 16 + v * 4);
      sdot = fmaf(wq[v][0], c4[0], sdot); sdot = fmaf(wq[v][1], c4[1], sdot); sdot = fmaf(wq[v][2], c4[2], sdot); sdot = fmaf(wq[v][3], c4[3], sdot);
    }
    sdot += __shfl_xor(sdot, 1, 64);
    sdot += __shfl_xor(sdot, 2, 64);
    if (hq == 0 && hr < p.RD) p.hpart[(((long long)b * nslot + slot) * p.ncb + cb) * p.RD + hr] = sdot;
  }
}

}  // namespace

// Work decomposition: y-runs of 16 output rows, x-segments of XS columns chosen so that the launch has >= ~3 workgroups per CU with
// the least halo; the patch must fit the per-thread load count of the kernel.
struct DwlPlan { int XS, nxs, nyr, colsA, rowpitch; size_t lds; };
int g_dwl_force_xs = 0;        // cfp_debug_set key 9 (tools/dw_bench.py --sweep-xs)

static bool dwl_plan(int B, int Ho, int Wo, int C, int stride, DwlPlan& d) {
  const int ncb = cdiv(C, 64), nyr = cdiv(Ho, 16);
  const int rows = 15 * stride + 3, nld = stride == 1 ? 8 : 10;
  const int xs_max = stride == 1 ? 12 : 4;
  int best = 0; double bc = 1e30;
  for (int XS = 2; XS <= xs_max && XS <= std::max(Wo, 2); ++XS) {
    const int colsA = (XS - 1) * stride + 3;
    if (rows * colsA * 8 > 256 * nld) continue;
    if (g_dwl_force_xs && XS != g_dwl_force_xs) continue;
    const int nxs = cdiv(Wo, XS);
    const long long tasks = (long long)B * ncb * nyr * nxs;
    const double halo = (double)((std::min(XS, Wo) - 1) * stride + 3) / (std::min(XS, Wo) * stride);
    const double waste = (double)(nxs * XS) / Wo;                      // ragged last segment
    const double fill = tasks >= 704 ? 1.0 : 704.0 / (double)tasks;    // ~2.75 workgroups per CU
    const double c = halo * waste * fill;
    if (c < bc) { bc = c; best = XS; }
  }
  if (best == 0) return false;
  d.XS = best; d.nxs = cdiv(Wo, best); d.nyr = nyr; d.colsA = (best - 1) * stride + 3;
  int rp = d.colsA * DWL_PP;
  if (((rp / 16) & 1) == 0) rp += 16;                                   // 16 x odd
  d.rowpitch = rp;
  d.lds = (size_t)rows * rp;
  if (d.lds < 1024) d.lds = 1024;
  return d.lds <= 64 * 1024 && (long long)B * ncb * nyr * d.nxs < (1ll << 31);
}

void cfp_dwl_debug_set(int value) { g_dwl_force_xs = value; }

int cfp_dwl_slots(int B, int Ho, int Wo, int C, int stride) {
  DwlPlan d;
  return dwl_plan(B, Ho, Wo, C, stride, d) ? d.nyr * d.nxs : 0;
}

// -> CFP_OK, an error code, or 1 when the shape is not taken (the caller falls back to dw3x3_mfma_kernel)
int cfp_dwl_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld, float* partial,
                   const float* w_red, int RD, float* hpart, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo,
                   int act, int dtype, cfp_stream_t stream, const char* who) {
  DwlPlan d;
  if (!dwl_plan(B, Ho, Wo, C, stride, d)) return 1;
  DwlP p;
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.out = (bf16_t*)out; p.partial = partial;
  p.w_red = w_red; p.hpart = hpart; p.RD = RD;
  p.in_ld = in_ld; p.out_ld = out_ld;
  p.B = B; p.H = H; p.W = W; p.C = C; p.pad_t = pad_t; p.pad_l = pad_l; p.Ho = Ho; p.Wo = Wo; p.act = act;
  p.XS = d.XS; p.nxs = d.nxs; p.nyr = d.nyr; p.ncb = cdiv(C, 64); p.colsA = d.colsA; p.rowpitch = d.rowpitch;
  p.dcols = make_fastdiv((unsigned)d.colsA);
  p.dbg = nullptr;
  if (act >= 100) {             // diagnostic launch: stamps behind the partial-sum area (the caller sized it: tools/dw_bench.py)
    act -= 100; p.act = act;
    if (partial) p.dbg = partial + (long long)B * d.nyr * d.nxs * C;
  }
  const long long tasks = (long long)B * p.ncb * d.nyr * d.nxs;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DWL_LAUNCH(HH, ST)                                                                                                          \
  do {                                                                                                                               \
    static bool attr = false;                                                                                                        \
    if (!attr) {                                                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)dw3x3_slide_kernel<HH, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
      if (e != hipSuccess) { cfp_set_error(std::string(who) + ": " + hipGetErrorString(e)); return CFP_EHIP; }                       \
      attr = true;                                                                                                                   \
    }                                                                                                                                \
    hipLaunchKernelGGL((dw3x3_slide_kernel<HH, ST>), dim3((unsigned)tasks), dim3(256), d.lds, s, p);                                 \
  } while (0)
  if (dtype == CFP_F16) { if (stride == 1) DWL_LAUNCH(f16_t, 1); else DWL_LAUNCH(f16_t, 2); }
  else { if (stride == 1) DWL_LAUNCH(bf16_t, 1); else DWL_LAUNCH(bf16_t, 2); }
#undef DWL_LAUNCH
  return cfp_check_launch(who);
}
