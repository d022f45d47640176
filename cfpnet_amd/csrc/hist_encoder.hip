// The ToF histogram encoder as ONE kernel:  9 x (Conv1d k=1 -> BatchNorm1d -> ReLU) applied to every ToF
// sample point independently, widths 1 -> 32 -> 32 -> 32 -> 64 -> 64 -> 64 -> 128 -> 128 -> 128, tapped after layers 3, 6, 9.
// Reference: /root/reference/src/models/encoder.py:17-24 (PointNetEncoder), :31-35 (HistExtractor), :45-50 (HistogramEncoder).
//
// Round 1 ran it as 1 + 9 launches of the generic GEMM with 16-bit intermediates.  Here a workgroup owns 32 sample points and
// carries them through all nine layers: activations stay in LDS as FLOAT32 (two ping-pong tiles), every layer is
// `v_mfma_f32_16x16x4_f32` (exact f32 products, f32 accumulation -- the encoder's 55 k weights are never rounded to 16 bit), the
// weights stream from L2 straight into MFMA B fragments (221 KB for all layers, shared by every workgroup), BatchNorm is the folded
// scale/shift of the epilogue.  Only the three tapped activations leave the chip, in the engine's storage type.
//
// Fragment trick: lane (r = lane % 16, q = lane / 16) reads FOUR consecutive k of its row with one 16-byte load from each operand
// (activations: LDS, weights: global) and feeds element i of both to the i-th of four MFMAs -- the k index 16 kb + 4 q + i is the same
// on both sides, so the sum over k is complete, just in another order.
#include "common.h"

namespace {

constexpr int HE_P = 32;          // sample points per workgroup
constexpr int HE_PITCH = 136;     // floats per activation row in LDS: 34 sixteen-byte slots = 2 (mod 4).  `ds_read_b128` is banked over 16-lane
                                  // groups that pair rows {0-3, 12-15} of chunk q with rows {4-11} of chunk q + 1: the first eight rows then
                                  // cover the even slots of the 256-byte bank row, the others the odd ones (an ODD slot count -- 132 floats,
                                  // round 2 -- is conflict-free only for the 16 rows of ONE chunk: 42 % conflict cycles by the SQ counters)
constexpr int HE_LAYERS = 9;

struct HistEncP {
  const float* hist;              // [R] sample depths
  const float* blob;              // all parameters, float32: per layer W [Cout][Cin] | scale [Cout] | shift [Cout]
  int w_off[HE_LAYERS], s_off[HE_LAYERS], t_off[HE_LAYERS], cin[HE_LAYERS], cout[HE_LAYERS];
  void* out[3];                   // taps after layers 3, 6, 9: [R][cout] in the storage type
  const float* pe[3];             // optional [n_pe][cout] f32 table added to tap t on its way out (row = point index % n_pe): the
  int n_pe;                       // fusion blocks' `feat1 + positional_encodings2` (fusion.py:123-125), saving three launches
  int R;
};

template <typename T>
__global__ __launch_bounds__(256) void hist_encoder_kernel(HistEncP p) {
  __shared__ __attribute__((aligned(16))) float sX[2][HE_P][HE_PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int p0 = blockIdx.x * HE_P;

  // ---- layer 1: Cin = 1 (deltar.py:40 `hist_data.unsqueeze(-1)`): out[c] = relu(scale[c] * (w[c] * x) + shift[c]) ------------------
  {
    const int pt = tid & 31;
    const float x = (p0 + pt < p.R) ? p.hist[p0 + pt] : 0.f;
    const int C = p.cout[0];
    for (int c = tid >> 5; c < C; c += 8) {
      const float v = p.blob[p.s_off[0] + c] * (p.blob[p.w_off[0] + c] * x) + p.blob[p.t_off[0] + c];
      sX[0][pt][c] = fmaxf(v, 0.f);
    }
  }
  __syncthreads();

  int cur = 0, tap = 0;
  for (int l = 1; l < HE_LAYERS; ++l) {
    const int Cin = p.cin[l], Cout = p.cout[l];
    const float* __restrict__ W = p.blob + p.w_off[l];
    const float* __restrict__ sc = p.blob + p.s_off[l];
    const float* __restrict__ sh = p.blob + p.t_off[l];
    const int NT = Cout >> 4;
    for (int nt = wave; nt < NT; nt += 4) {                         // a wave owns 16 output channels x all 32 points
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const float* wrow = W + (long long)(nt * 16 + fr) * Cin + 4 * fq;
      // all weight fragments of this 16-channel tile are requested before the first MFMA (Cin <= 128: at most 8 of them): one L2
      // round trip per tile instead of one per 16 input channels (the loop below was latency-bound: 51 us per launch before)
      f32x4 b4[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) b4[j] = (j * 16 < Cin) ? *reinterpret_cast<const f32x4*>(wrow + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j * 16 < Cin) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(&sX[cur][fr][j * 16 + 4 * fq]);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(&sX[cur][16 + fr][j * 16 + 4 * fq]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], b4[j][i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], b4[j][i], acc[1], 0, 0, 0);
          }
        }
      }
      const int ch = nt * 16 + fr;
      const float s = sc[ch], t = sh[ch];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) sX[cur ^ 1][mt * 16 + 4 * fq + i][ch] = fmaxf(acc[mt][i] * s + t, 0.f);
    }
    __syncthreads();
    cur ^= 1;
    if (l % 3 == 2) {                                               // layers 3, 6, 9 (l = 2, 5, 8): a tap leaves the chip
      T* out = reinterpret_cast<T*>(p.out[tap]);
      const float* pe = p.pe[tap++];
      constexpr int VE = Vec<T>::N;
      const int cpr = Cout / VE;                                    // 16-byte vectors per row
      for (int q = tid; q < HE_P * cpr; q += 256) {
        const int pt = q / cpr, cv = q - pt * cpr;
        if (p0 + pt >= p.R) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = sX[cur][pt][cv * VE + e];
        if (pe) {
          const float* row = pe + (long long)((p0 + pt) % p.n_pe) * Cout + cv * VE;
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += row[e];
        }
        Vec<T>::store(out + (long long)(p0 + pt) * Cout + cv * VE, v);
      }
    }
  }
}

}  // namespace

extern "C" int cfp_hist_encoder(const float* hist, const float* blob, const int* layout, void* out0, void* out1, void* out2,
                                const float* pe0, const float* pe1, const float* pe2, int n_pe, int R, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(hist && blob && layout && out0 && out1 && out2, CFP_EINVAL, "cfp_hist_encoder: null pointer");
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_hist_encoder: bad dtype");
  CFP_REQUIRE(R > 0, CFP_ESHAPE, "cfp_hist_encoder: no sample points");
  CFP_REQUIRE(aligned16(blob) && aligned16(out0) && aligned16(out1) && aligned16(out2), CFP_EINVAL, "cfp_hist_encoder: pointers must be 16-byte aligned");
  HistEncP p;
  p.hist = hist; p.blob = blob; p.R = R;
  p.out[0] = out0; p.out[1] = out1; p.out[2] = out2;
  p.pe[0] = pe0; p.pe[1] = pe1; p.pe[2] = pe2; p.n_pe = n_pe;
  CFP_REQUIRE(!(pe0 || pe1 || pe2) || n_pe > 0, CFP_ESHAPE, "cfp_hist_encoder: positional tables need n_pe > 0");
  for (int l = 0; l < HE_LAYERS; ++l) {                            // layout: 5 host ints per layer (w_off, scale_off, shift_off, cin, cout), offsets in floats
    p.w_off[l] = layout[5 * l]; p.s_off[l] = layout[5 * l + 1]; p.t_off[l] = layout[5 * l + 2];
    p.cin[l] = layout[5 * l + 3]; p.cout[l] = layout[5 * l + 4];
    CFP_REQUIRE(p.cout[l] > 0 && p.cout[l] <= 128 && p.cout[l] % 16 == 0 && p.w_off[l] % 4 == 0 && p.w_off[l] >= 0 && p.s_off[l] >= 0 && p.t_off[l] >= 0,
                CFP_ESHAPE, "cfp_hist_encoder: widths must be multiples of 16 (<= 128), weight blocks 16-byte aligned");
    CFP_REQUIRE(l == 0 ? p.cin[l] == 1 : (p.cin[l] == p.cout[l - 1]), CFP_ESHAPE, "cfp_hist_encoder: layer widths do not chain (first layer takes 1 channel)");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(cdiv(R, HE_P));
  if (dtype == CFP_BF16) hipLaunchKernelGGL(hist_encoder_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == CFP_F16) hipLaunchKernelGGL(hist_encoder_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(hist_encoder_kernel<float>, grid, dim3(256), 0, s, p);
  return cfp_check_launch("cfp_hist_encoder");
}
