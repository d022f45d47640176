// Depth-evaluation metrics on the device: per image, one pass over prediction and ground truth.
//
//   reference: compute_errors (src/utils/metrics.py:4-24, identical copy evaluate_all.py:15-35) and the two
//   protocols wrapped around it:
//     mode 0  evaluate_all.py:38-41,80-84   pred = clip(pred, lo, hi) at model resolution, THEN bilinear
//             (align_corners=True) to the ground-truth size; valid = lo < gt < hi
//     mode 1  train.py:187-199 (validate)   bilinear first, then pred<lo -> lo, pred>hi -> hi, inf -> hi, nan -> lo;
//             valid = lo < gt < hi with the *_eval bounds
//   The reference moves both maps to the host and runs nine numpy reductions per image; here prediction and ground
//   truth are read once from HBM and nothing leaves the device until the caller asks for the numbers.
//
// Per-pixel terms are evaluated in float32 like numpy does on the float32 arrays; the sums are float64 (numpy:
// float32 pairwise sums), combined in a fixed order -> run-to-run deterministic, parity to ~1e-6 relative.
// Roofline: HBM, 4 B/pixel of ground truth + 4 B per model-resolution pixel of prediction (L2 serves the 4 taps).
#include "common.h"

namespace {

constexpr int kMetBlocks = 96;     // workgroups per image
constexpr int kMetTerms = 10;      // a1 a2 a3 abs_rel se log10 le2 le sq_rel n

struct MetP {
  const float* pred; const float* gt; double* partial; double* out;
  int B, Hp, Wp, H, W, interpolate, mode;
  float lo, hi, sy, sx;
};

__device__ __forceinline__ float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // NaN passes, like np.clip

__device__ __forceinline__ float met_pred(const MetP& p, const float* __restrict__ pb, int i) {
  float v;
  if (p.interpolate) {
    const int y = i / p.W, x = i - y * p.W;
    const float fy = p.sy * (float)y, fx = p.sx * (float)x;
    const int y0 = min((int)fy, p.Hp - 1), x0 = min((int)fx, p.Wp - 1);
    // ATen (UpSample.h compute_source_index_and_lambda): a dimension whose size does not change reads the SAME
    // pixel twice with weights (1, 0), so a non-finite value turns into NaN there and does not touch its neighbours
    const int y1 = p.Hp == p.H ? y0 : min(y0 + 1, p.Hp - 1), x1 = p.Wp == p.W ? x0 : min(x0 + 1, p.Wp - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    float t00 = pb[y0 * p.Wp + x0], t01 = pb[y0 * p.Wp + x1], t10 = pb[y1 * p.Wp + x0], t11 = pb[y1 * p.Wp + x1];
    if (p.mode == 0) { t00 = clipf(t00, p.lo, p.hi); t01 = clipf(t01, p.lo, p.hi); t10 = clipf(t10, p.lo, p.hi); t11 = clipf(t11, p.lo, p.hi); }
    v = hy * (hx * t00 + lx * t01) + ly * (hx * t10 + lx * t11);
  } else {
    v = pb[i];
    if (p.mode == 0) v = clipf(v, p.lo, p.hi);
  }
  if (p.mode == 1) {
    v = clipf(v, p.lo, p.hi);
    if (v != v) v = p.lo;
  }
  return v;
}

__global__ __launch_bounds__(256) void metrics_kernel(MetP p) {
  __shared__ double red[4][kMetTerms];
  const int b = blockIdx.y;
  const int hw = p.H * p.W;
  const float* pb = p.pred + (long long)b * p.Hp * p.Wp;
  const float* gb = p.gt + (long long)b * hw;
  double acc[kMetTerms];
#pragma unroll
  for (int k = 0; k < kMetTerms; ++k) acc[k] = 0.0;
  constexpr int U = 4;
  for (int i0 = blockIdx.x * 256 * U + threadIdx.x; i0 < hw; i0 += kMetBlocks * 256 * U) {
    float g[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = gb[min(i0 + u * 256, hw - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = met_pred(p, pb, min(i0 + u * 256, hw - 1));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * 256 >= hw || !(g[u] > p.lo && g[u] < p.hi)) continue;
      const float th = fmaxf(g[u] / v[u], v[u] / g[u]);
      const float d = g[u] - v[u];
      const float le = logf(g[u]) - logf(v[u]);
      acc[0] += th < 1.25f ? 1.0 : 0.0;
      acc[1] += th < 1.5625f ? 1.0 : 0.0;
      acc[2] += th < 1.953125f ? 1.0 : 0.0;
      acc[3] += (double)(fabsf(d) / g[u]);
      acc[4] += (double)(d * d);
      acc[5] += (double)fabsf(log10f(g[u]) - log10f(v[u]));
      acc[6] += (double)(le * le);
      acc[7] += (double)(-le);                 // err = log pred - log gt
      acc[8] += (double)((d * d) / g[u]);
      acc[9] += 1.0;
    }
  }
  // fixed-order reduction: butterfly inside the wave, then the four waves in order
#pragma unroll
  for (int k = 0; k < kMetTerms; ++k) {
    double a = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = a;
  }
  __syncthreads();
  if (threadIdx.x < kMetTerms)
    p.partial[((long long)b * kMetBlocks + blockIdx.x) * kMetTerms + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// out[b] = {a1, a2, a3, abs_rel, rmse, log_10, rmse_log, silog, sq_rel, n}  (the reference's dict order + the pixel count)
__global__ __launch_bounds__(64) void metrics_finalize_kernel(const double* __restrict__ partial, double* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  static_assert(kMetBlocks <= 128, "two partial rows per lane");
  double s[kMetTerms];
#pragma unroll
  for (int k = 0; k < kMetTerms; ++k) {          // all loads independent; the butterfly order is fixed
    double a = lane < kMetBlocks ? partial[((long long)b * kMetBlocks + lane) * kMetTerms + k] : 0.0;
    if (lane + 64 < kMetBlocks) a += partial[((long long)b * kMetBlocks + lane + 64) * kMetTerms + k];
    s[k] = a;
  }
#pragma unroll
  for (int k = 0; k < kMetTerms; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
  }
  if (lane == 0) {
    const double n = s[9];
    double* o = out + (long long)b * kMetTerms;
    const double mle = s[7] / n;
    o[0] = s[0] / n; o[1] = s[1] / n; o[2] = s[2] / n;
    o[3] = s[3] / n;
    o[4] = sqrt(s[4] / n);
    o[5] = s[5] / n;
    o[6] = sqrt(s[6] / n);
    o[7] = sqrt(s[6] / n - mle * mle) * 100.0;
    o[8] = s[8] / n;
    o[9] = n;
  }
}

}  // namespace

extern "C" size_t cfp_eval_metrics_ws_bytes(int B) { return B > 0 ? (size_t)B * kMetBlocks * kMetTerms * sizeof(double) : 0; }

extern "C" int cfp_eval_metrics(const float* pred, int Hp, int Wp, const float* gt, int H, int W, int B, int interpolate, int mode,
                                float lo, float hi, void* ws, size_t ws_bytes, double* out, cfp_stream_t stream) {
  CFP_REQUIRE(pred && gt && ws && out, CFP_EINVAL, "cfp_eval_metrics: null pointer");
  CFP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && H > 0 && W > 0, CFP_ESHAPE, "cfp_eval_metrics: non-positive dimension");
  CFP_REQUIRE((long long)H * W < (1ll << 31) && B <= 65535, CFP_ESHAPE, "cfp_eval_metrics: image or batch too large");
  CFP_REQUIRE(interpolate || (Hp == H && Wp == W), CFP_ESHAPE, "cfp_eval_metrics: sizes differ and interpolate is off");
  CFP_REQUIRE(mode == 0 || mode == 1, CFP_EINVAL, "cfp_eval_metrics: mode must be 0 (evaluate_all) or 1 (validate)");
  CFP_REQUIRE(lo < hi, CFP_EINVAL, "cfp_eval_metrics: empty depth range");
  CFP_REQUIRE(ws_bytes >= cfp_eval_metrics_ws_bytes(B), CFP_EINVAL, "cfp_eval_metrics: workspace too small");
  CFP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7) == 0, CFP_EINVAL, "cfp_eval_metrics: workspace must be 8-byte aligned");
  MetP p;
  p.pred = pred; p.gt = gt; p.partial = reinterpret_cast<double*>(ws); p.out = out;
  p.B = B; p.Hp = Hp; p.Wp = Wp; p.H = H; p.W = W; p.interpolate = interpolate; p.mode = mode; p.lo = lo; p.hi = hi;
  p.sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f;
  p.sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(metrics_kernel, dim3(kMetBlocks, B), dim3(256), 0, s, p);
  hipLaunchKernelGGL(metrics_finalize_kernel, dim3(B), dim3(64), 0, s, p.partial, out);
  return cfp_check_launch("cfp_eval_metrics");
}
