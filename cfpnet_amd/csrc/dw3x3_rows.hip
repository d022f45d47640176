// Depthwise 3x3 (+ BN + activation + squeeze-excite channel sums), FLOAT32 storage: the REGISTER-SLIDING form (round 5).
//
// Reference op: timm InvertedResidual `conv_dw -> bn2 -> act` (oracle/cfpnet_oracle.py, encoder(); /root/reference/src/models/encoder.py:66-69
// picks the blocks), 24 launches per forward -- the kernel family BASELINE.json's north_star puts ">= 60 % of measured HBM roofline" on.
// In the boundary's default mode (f32x3: float32 tensors) these launches ran the round-1 `dw3x3_kernel<float, ., 8>`: strip -> LDS in one
// sweep, ONE barrier, then compute -- load and compute phases of a workgroup never overlap, 33 % of its LDS cycles were bank conflicts
// (profiles/r4z_x3_pmc_sq.json) and the PMC traffic was 1.37x the algorithmic bytes (halo rows of every strip fetched again).
//
// At 4 bytes per element the arithmetic is a THIRD of the memory time (36 FMAs + BN + SiLU per four channels ~ 80 VALU instructions per
// 256 outputs against 2 KB moved), so nothing needs the matrix core or LDS here; what matters is that every wave has loads in flight
// all the time and that each byte leaves HBM once:
//
//   * a WAVE is a workgroup (no barrier, no LDS): lane = (channel vector cvl of 8, pixel slot of 8) -- a pixel's 8 channel vectors are
//     one 128-byte line, so every load / store instruction of the wave moves eight whole lines;
//   * a pixel slot is (run of R output rows, output column x); the lane SLIDES DOWN its column: the 3 x 3 window of float4 values lives
//     in registers (ring of 4 input rows at stride 1, 5 at stride 2), each output row loads ONE new input row (three columns: the two
//     neighbours come from L1 / L2 -- the next lanes fetch the same lines), issued one output row AHEAD of its use;
//   * borders and dead lanes cost nothing: loads go through a buffer descriptor of the image and a lane outside it adds 2^30 to its byte
//     offset -- the hardware returns zeros (TF-SAME padding) / drops the store;
//   * channel sums of the stored values (squeeze-excite): per lane over its rows, then three exchange steps over the wave's 8 pixel slots
//     in a fixed order, written as partial[b][slot][c] (cfp_se_gate_fold adds the slots in index order: deterministic).
//
// Arithmetic per output = dw3x3_kernel<float>'s own: taps in (ky, kx) order by fused multiply-add from 0, then acc * scale + shift, then
// the activation -- the stored tensor is bit-identical to the old kernel's (tests/test_ops_gpu.py::test_dw3x3_rows_*).
#include <algorithm>

#include "common.h"

namespace {

struct DwrP {
  const float* in; const float* w; const float* scale; const float* shift; float* out; float* partial;
  const float* w_red; float* hpart; int RD;      // squeeze-excite reduce FC [RD][C] and its partial dot products [B][nps * ncb][RD], or null
  int in_ld, out_ld, B, H, W, C, pad_t, pad_l, Ho, Wo, act;
  int R, nruns, CV, ncb, npx, nps;      // output rows per run, runs, C / 4, ceil(CV / 8), Wo * nruns, ceil(ceil(npx / 8) / 4) = SLOTS (workgroups per image and channel block)
  unsigned img_bytes, oimg_bytes;       // extent of one image of `in` / `out` in bytes (< 2^30)
  FastDiv dWo, dnps, dncb;
};

constexpr unsigned DWR_OOB = 0x40000000u;     // added to a byte offset: beyond every image (< 2^30 bytes), also when added twice

template <int STRIDE, int ACT>
__global__ __launch_bounds__(256) void dw3x3_rows_kernel(DwrP p) {
  constexpr int NB = STRIDE == 1 ? 4 : 5;       // ring of input rows: output row i reads rows S i .. S i + 2, rows S i + 3 .. are in flight
  const int lane = threadIdx.x & 63;
  const int wv_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // task = (image, 8-vector channel block, FOUR groups of 8 pixel slots: one per wave); XCD-aware: neighbouring tasks share lines in one L2.
  // The four waves never meet before the channel sums at the very end (one barrier): they exist to cut the number of partial-sum slots
  // the squeeze-excite tail has to add (cfp_se_gate_fold / fold2 walk them serially) by four.
  unsigned t = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
  unsigned psg4, cb, b;
  fd_rowcol(t, p.dnps, t, psg4);
  fd_rowcol(t, p.dncb, b, cb);
  const unsigned psg = psg4 * 4 + (unsigned)wv_id;
  const int cvl = lane & 7, pxl = lane >> 3;
  const int cv = (int)cb * 8 + cvl;
  const int ps = (int)psg * 8 + pxl;
  const bool live = cv < p.CV && ps < p.npx;
  unsigned run, xu;
  fd_rowcol((unsigned)min(ps, p.npx - 1), p.dWo, run, xu);
  const int x = (int)xu, y0 = (int)run * p.R;
  const int ny = live ? min(p.R, p.Ho - y0) : 0;
  const int c = min(cv, p.CV - 1) * 4;

  const float* img = p.in + (long long)b * p.H * p.W * p.in_ld;
  float* oimg = p.out + (long long)b * p.Ho * p.Wo * p.out_ld;
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)p.img_bytes, 0x00020000);
  const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)oimg, 0, (int)p.oimg_bytes, 0x00020000);

  unsigned colo[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int ix = x * STRIDE - p.pad_l + d;
    colo[d] = (live && (unsigned)ix < (unsigned)p.W) ? (unsigned)(ix * p.in_ld + c) * 4u : DWR_OOB;
  }
  const int iy0 = y0 * STRIDE - p.pad_t;
  const unsigned rowb = (unsigned)(p.W * p.in_ld) * 4u;
  const unsigned orow = (unsigned)(p.Wo * p.out_ld) * 4u;
  const unsigned obase = (unsigned)y0 * orow + (unsigned)(x * p.out_ld + c) * 4u;

  f32x4 wv[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(p.w + (long long)k * p.C + c);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c), sh = *reinterpret_cast<const f32x4*>(p.shift + c);

  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  auto ldrow = [&](int r, f32x4 (&buf)[3]) {
    const int iy = iy0 + r;
    const unsigned ro = (unsigned)iy < (unsigned)p.H ? (unsigned)iy * rowb : DWR_OOB;
#pragma unroll
    for (int d = 0; d < 3; ++d) buf[d] = __builtin_bit_cast(f32x4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(ro + colo[d]), 0, 0));
  };

  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  f32x4 rb[NB][3];
#pragma unroll
  for (int r = 0; r < 3; ++r) ldrow(r, rb[r]);
  const int R = (int)psg * 8 < p.npx ? p.R : 0;      // a wave past the last pixel slot (the workgroup's ragged tail) only joins the final barrier
  {
    for (int i0 = 0; i0 < R; i0 += NB) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + u;                     // i % NB == u
        if (i < R) {                              // wave-uniform
          if (i + 1 < R) {
#pragma unroll
            for (int s = 0; s < STRIDE; ++s) ldrow(STRIDE * i + 3 + s, rb[(STRIDE * u + 3 + s) % NB]);
          }
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const f32x4(&row)[3] = rb[(STRIDE * u + ky) % NB];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[e] = fmaf(row[kx][e], wv[ky * 3 + kx][e], acc[e]);
          }
          const bool on = i < ny;
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = act_c<ACT>(acc[e] * sc[e] + sh[e]);
            csum[e] += on ? y[e] : 0.f;
          }
          const unsigned oo = on ? obase + (unsigned)i * orow : DWR_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), rs_out, (int)oo, 0, 0);
        }
      }
    }
  }

  if (p.partial == nullptr && p.hpart == nullptr) return;
  // sum over the wave's 8 pixel slots (lane bits 3-5) in a fixed order, then over the four waves through LDS (wave order): wave 0 holds the
  // workgroup's 32 channel sums in lanes 0-7 (and in every lane with the same cvl)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = csum[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    csum[e] = v;
  }
  __shared__ f32x4 wsum[4][8];
  if (lane < 8) wsum[wv_id][lane] = csum;
  __syncthreads();
  if (wv_id != 0) return;
  csum = (wsum[0][cvl] + wsum[1][cvl]) + (wsum[2][cvl] + wsum[3][cvl]);
  if (p.partial != nullptr && pxl == 0 && cv < p.CV) *reinterpret_cast<f32x4*>(p.partial + ((long long)b * p.nps + psg4) * p.C + c) = csum;
  if (p.hpart == nullptr) return;
  // squeeze-excite: the reduce FC is linear in the channel sums, so the workgroup applies it to its own 32 channels (lane = hidden unit
  // r <= 64): hpart[b][slot * ncb + cb][r] = sum_c w_red[r][c] * sum[c]; cfp_se_gate_fold2 adds the parts in index order.  Dead channels hold 0.
  float dot = 0.f;
  const bool r_ok = lane < p.RD;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const int cvv = (int)cb * 8 + l;                                             // wave-uniform
    f32x4 wq = {0.f, 0.f, 0.f, 0.f};
    if (cvv < p.CV && r_ok) wq = *reinterpret_cast<const f32x4*>(p.w_red + (long long)lane * p.C + cvv * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) dot = fmaf(wq[e], lane_value(csum[e], l), dot);
  }
  if (r_ok) p.hpart[(((long long)b * p.nps + psg4) * p.ncb + cb) * p.RD + lane] = dot;
}

}  // namespace

// Work decomposition: runs of R output rows.  Short runs re-read halo rows, long runs leave the chip short of waves; ragged pixel-slot groups
// and channel blocks idle lanes.
struct DwrPlan { int R, nruns, npx, nps, ncb; };
int g_dwr_mode = 1;           // cfp_debug_set key 10: 0 = the round-1 LDS-strip kernel
int g_dwr_force_R = 0;        // cfp_debug_set key 11 (tools/dw_bench.py --sweep-r)

static bool dwr_plan(int B, int H, int W, int Ho, int Wo, int C, int stride, int in_ld, int out_ld, DwrPlan& d) {
  if (C % 4 != 0 || (long long)H * W * in_ld * 4 >= (1ll << 30) || (long long)Ho * Wo * out_ld * 4 >= (1ll << 30)) return false;
  const int CV = C / 4, ncb = cdiv(CV, 8);
  int best = 0; double bc = 1e30;
  for (int R = 1; R <= Ho; ++R) {
    if (g_dwr_force_R && R != std::min(g_dwr_force_R, Ho)) continue;
    const int nruns = cdiv(Ho, R);
    if (!g_dwr_force_R && cdiv(Ho, nruns) != R) continue;                // the shortest run length for this run count
    const long long npx = (long long)Wo * nruns, nps = (npx + 7) / 8;      // pixel-slot groups = live waves per image and channel block
    const long long waves = (long long)B * ncb * nps;
    // A time model in microseconds, fitted on tools/dw_bench_f32.py --sweep-r at batch 1 / 8 / 128 (profiles/r5a_dw_f32_sweep.txt):
    //   bytes: halo rows cost what first reads cost (a neighbouring run is rarely in flight nearby); ~5.5 TB/s while the chip is full;
    //   chain: a lane's R row steps are serial (~0.25 us each: load -> 36 FMAs -> store), once per resident round of 4 096 waves;
    //   tail:  every slot is one more partial vector the squeeze-excite kernel adds serially (~10 ns each).
    const double halo = (double)((R - 1) * stride + 3) / (R * stride);  // input rows read per input row used
    const double lanes = (double)(((nps + 3) / 4) * 32) / npx;         // idle lanes AND idle waves of the four-wave workgroups: they hold registers
                                                                         // (122 VGPRs: 16 waves per CU) that would otherwise keep loads in flight
    const double rows = (double)(nruns * R) / Ho;                       // ragged last run (its lanes idle through the tail rows)
    const double in_b = 4.0 * B * (Ho * stride) * (Wo * stride) * C, out_b = 4.0 * B * Ho * Wo * C;      // from the OUTPUT extent only: cfp_dwr_slots (asked before
                                                                                                         // the launch, without H / W) must see the same plan
    const double t_mem = (in_b * halo + out_b) * lanes * rows / 5.5e6;
    const double t_chain = std::max(1.0, (double)waves / 4096.0) * R * 0.25;
    const double t_tail = 0.01 * (double)((nps + 3) / 4) * ncb;
    const double cst = t_mem + t_chain + t_tail;
    if (cst < bc) { bc = cst; best = R; }
  }
  if (best == 0) return false;
  d.R = best; d.nruns = cdiv(Ho, best); d.npx = Wo * d.nruns; d.nps = cdiv(cdiv(d.npx, 8), 4); d.ncb = ncb;      // nps = workgroups (4 waves) per image and channel block
  return (long long)B * ncb * d.nps < (1ll << 31);
}

int cfp_dwr_launch_slots(int B, int H, int W, int Ho, int Wo, int C, int stride, int in_ld, int out_ld) {
  DwrPlan d;
  if (!g_dwr_mode || !dwr_plan(B, H, W, Ho, Wo, C, stride, in_ld, out_ld, d)) return 0;
  return d.nps;
}

void cfp_dwr_debug_set(int key, int value) { if (key == 10) g_dwr_mode = value; else if (key == 11) g_dwr_force_R = value; }

int cfp_dwr_slots(int B, int H, int W, int Ho, int Wo, int C, int stride, int* ncb) {
  DwrPlan d;
  if (!g_dwr_mode) return 0;
  if (!dwr_plan(B, H, W, Ho, Wo, C, stride, C, C, d)) return 0;
  if (ncb) *ncb = d.ncb;
  return d.nps;
}

// -> CFP_OK, an error code, or 1 when the shape is not taken (the caller falls back to dw3x3_kernel<float>)
int cfp_dwr_launch(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld, float* partial,
                   const float* w_red, int RD, float* hpart, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, cfp_stream_t stream, const char* who) {
  DwrPlan d, d0;
  if (!g_dwr_mode || (act != CFP_ACT_SILU && act != CFP_ACT_RELU && act != CFP_ACT_NONE)) return 1;
  const bool dense_ok = dwr_plan(B, H, W, Ho, Wo, C, stride, C, C, d0);          // what cfp_dwr_slots told the caller (it sized `partial` by it)
  if (!dwr_plan(B, H, W, Ho, Wo, C, stride, in_ld, out_ld, d) || !dense_ok || d0.nps != d.nps) {
    if ((partial || hpart) && dense_ok) { cfp_set_error(std::string(who) + ": row pitch too large for the float32 depthwise kernel"); return CFP_ESHAPE; }
    return 1;
  }
  DwrP p;
  p.in = (const float*)in; p.w = (const float*)w; p.scale = scale; p.shift = shift; p.out = (float*)out; p.partial = partial;
  p.w_red = w_red; p.hpart = hpart; p.RD = RD;
  p.in_ld = in_ld; p.out_ld = out_ld; p.B = B; p.H = H; p.W = W; p.C = C; p.pad_t = pad_t; p.pad_l = pad_l; p.Ho = Ho; p.Wo = Wo; p.act = act;
  p.R = d.R; p.nruns = d.nruns; p.CV = C / 4; p.ncb = d.ncb; p.npx = d.npx; p.nps = d.nps;
  p.img_bytes = (unsigned)(((long long)(H * W - 1) * in_ld + C) * 4);
  p.oimg_bytes = (unsigned)(((long long)(Ho * Wo - 1) * out_ld + C) * 4);
  p.dWo = make_fastdiv((unsigned)Wo); p.dnps = make_fastdiv((unsigned)d.nps); p.dncb = make_fastdiv((unsigned)d.ncb);
  const long long wgs = (long long)B * d.ncb * d.nps;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // the activation is a template argument (one register budget per variant: erff's temporaries would set it for all of them otherwise)
#define DWR(ST, AC) hipLaunchKernelGGL((dw3x3_rows_kernel<ST, AC>), dim3((unsigned)wgs), dim3(256), 0, s, p)
  if (act == CFP_ACT_SILU) { if (stride == 1) DWR(1, CFP_ACT_SILU); else DWR(2, CFP_ACT_SILU); }
  else if (act == CFP_ACT_RELU) { if (stride == 1) DWR(1, CFP_ACT_RELU); else DWR(2, CFP_ACT_RELU); }
  else { if (stride == 1) DWR(1, CFP_ACT_NONE); else DWR(2, CFP_ACT_NONE); }
#undef DWR
  return cfp_check_launch(who);
}
