// Optimizer step of the training loop on flat f32 buffers (train.py:78-94,127-135):
//   * AdamW exactly as torch.optim.AdamW applies it (decoupled weight decay first, bias-corrected
//     moments, eps added to sqrt(v_hat)), one launch per parameter group over a contiguous segment
//     of the flat parameter / gradient / moment buffers -- 21.8 M parameters are 4 streams of
//     87 MB, i.e. an HBM-bound 16-byte-vector sweep, instead of ~600 small tensor updates;
//   * global gradient norm (clip_grad_norm_, train.py:130) as deterministic f64 partial sums and a
//     device-side clip factor that the AdamW kernel reads, so clipping needs no host sync.
#include "common.h"

namespace {

constexpr int kNormBlocks = 1024;

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ gscale) {
  const float gs = gscale ? gscale[0] : 1.f;
  if (gs < 0.f) return;            // overflow guard (cfp_grad_clip_factor): the gradient norm was not finite -> the whole step is skipped,
                                   // parameters and both moments stay as they are (what torch.cuda.amp.GradScaler does with an inf / nan step)
  const float decay = 1.f - lr * wd;
  const float step_size = lr / bc1;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gs;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= decay;
      mm[e] = beta1 * mm[e] + (1.f - beta1) * gg[e];
      vv[e] = beta2 * vv[e] + (1.f - beta2) * gg[e] * gg[e];
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] -= step_size * (mm[e] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gg = g[i] * gs;
    float pp = p[i] * decay;
    const float mm = beta1 * m[i] + (1.f - beta1) * gg;
    const float vv = beta2 * v[i] + (1.f - beta2) * gg * gg;
    pp -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

// 16-byte loads, four independent f64 chains per thread (the scalar-load form ran at 0.65 TB/s: 134 us for 21.8 M gradients)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, double* __restrict__ partial) {
  __shared__ double red[256];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 t = reinterpret_cast<const f32x4*>(x)[i];
    s0 = fma((double)t[0], (double)t[0], s0); s1 = fma((double)t[1], (double)t[1], s1);
    s2 = fma((double)t[2], (double)t[2], s2); s3 = fma((double)t[3], (double)t[3], s3);
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double t = (double)x[i];
    s0 += t * t;
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[0] = clip factor min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_), out[1] = norm; a norm that is not finite
// (an overflow somewhere in a 16-bit backward) makes out[0] = -1 -- cfp_adamw_step then skips the step -- and counts in out[2].
// One wave: lanes stride over the partials, then a fixed-order tree through LDS.
__global__ void clip_factor_kernel(const double* __restrict__ partial, int nblk, float max_norm, float* __restrict__ out) {
  __shared__ double red[64];
  if (blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const float norm = (float)sqrt(red[0]);
  out[1] = norm;
  if (!(norm < 3.0e38f)) { out[0] = -1.f; out[2] += 1.f; return; }      // inf or nan
  const float c = max_norm / (norm + 1e-6f);
  out[0] = c < 1.f ? c : 1.f;
}

}  // namespace

extern "C" int cfp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                              cfp_stream_t stream) {
  CFP_REQUIRE(param && grad && exp_avg && exp_avg_sq, CFP_EINVAL, "cfp_adamw_step: null pointer");
  CFP_REQUIRE(n > 0 && step >= 1, CFP_ESHAPE, "cfp_adamw_step: n and step must be positive");
  CFP_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), CFP_EINVAL,
              "cfp_adamw_step: buffers must be 16-byte aligned");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  return cfp_check_launch("cfp_adamw_step");
}

extern "C" size_t cfp_grad_clip_ws_bytes(void) { return (size_t)kNormBlocks * sizeof(double); }

extern "C" int cfp_grad_clip_factor(const float* grad, long long n, float max_norm, void* ws, size_t ws_bytes, float* out,
                                    cfp_stream_t stream) {
  CFP_REQUIRE(grad && ws && out && aligned16(grad), CFP_EINVAL, "cfp_grad_clip_factor: null or misaligned pointer");
  CFP_REQUIRE(n > 0 && max_norm > 0.f, CFP_ESHAPE, "cfp_grad_clip_factor: bad arguments");
  CFP_REQUIRE(ws_bytes >= cfp_grad_clip_ws_bytes() && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, CFP_EINVAL,
              "cfp_grad_clip_factor: workspace too small or misaligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, s, grad, n, reinterpret_cast<double*>(ws));
  hipLaunchKernelGGL(clip_factor_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const double*>(ws), kNormBlocks, max_norm, out);
  return cfp_check_launch("cfp_grad_clip_factor");
}
