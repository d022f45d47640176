// NYU training augmentation on the device: random crop, horizontal flip, gamma / brightness / colour jitter, clipping,
// uint8 -> float and ImageNet normalisation of the RGB image, millimetre -> metre conversion of the depth map -- one pass.
//
//   reference: src/dataloader/nyu.py:128-136 (np.array(image)/255, depth/1000), :204-213 (random_crop), :215-227
//   (train_preprocess: flip, 50 % augmentation), :229-245 (augment_image), :266-285 (ToTensor + Normalize).
//   The random draws (crop origin, flip, do_augment, gamma, brightness, 3 colour gains) stay on the host like in the
//   reference (python `random` / `np.random`) and arrive as per-sample parameter rows; the random rotation of nyu.py:122-126
//   (PIL bicubic / nearest) is not reproduced here and stays a host step if wanted.
//
// Arithmetic order and types as numpy evaluates them: x = u8 / 255.f (f32);  x = powf(x, (float)gamma);  x *= (float)brightness;
// x = (float)((double)x * colour[c]);  clip to [0, 1];  (x - mean[c]) / std[c] in f32.  HBM-bound: 3 B/pixel in, 12 B/pixel out.
#include "common.h"

namespace {

struct AugP {
  const unsigned char* rgb; const unsigned short* depth_mm; const int* pi; const float* pf; const double* pc;
  float* img; float* dep;
  int B, H0, W0, H, W;
  float mean[3], stdv[3];
};

__global__ __launch_bounds__(256) void nyu_augment_kernel(AugP p) {
  const long long hw = (long long)p.H * p.W, total = hw * p.B;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / hw);
    const long long r = i - (long long)b * hw;
    const int y = (int)(r / p.W), x = (int)(r - (long long)y * p.W);
    const int x0 = min(max(p.pi[b * 4], 0), p.W0 - p.W), y0 = min(max(p.pi[b * 4 + 1], 0), p.H0 - p.H);      // device-side parameters: clamped, never trusted
    const int flip = p.pi[b * 4 + 2], aug = p.pi[b * 4 + 3];
    const int sx = x0 + (flip ? p.W - 1 - x : x), sy = y0 + y;
    const long long src = ((long long)b * p.H0 + sy) * p.W0 + sx;
    const float gamma = p.pf[b * 2], bright = p.pf[b * 2 + 1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (float)p.rgb[src * 3 + c] / 255.f;
      if (aug) {
        v = powf(v, gamma);
        v = v * bright;
        v = (float)((double)v * p.pc[b * 3 + c]);
        v = fminf(fmaxf(v, 0.f), 1.f);
      }
      p.img[((long long)b * 3 + c) * hw + r] = (v - p.mean[c]) / p.stdv[c];
    }
    if (p.depth_mm) p.dep[(long long)b * hw + r] = (float)p.depth_mm[src] / 1000.f;
  }
}

}  // namespace

extern "C" int cfp_nyu_augment(const unsigned char* rgb_u8, const unsigned short* depth_mm, int B, int H0, int W0, const int* params_i,
                               const float* params_f, const double* colors, int H, int W, const float* mean3, const float* std3, float* image_out,
                               float* depth_out, cfp_stream_t stream) {
  CFP_REQUIRE(rgb_u8 && params_i && params_f && colors && image_out && mean3 && std3, CFP_EINVAL, "cfp_nyu_augment: null pointer");
  CFP_REQUIRE((depth_mm == nullptr) == (depth_out == nullptr), CFP_EINVAL, "cfp_nyu_augment: depth input and output must come together");
  CFP_REQUIRE(B > 0 && H0 >= H && W0 >= W && H > 0 && W > 0, CFP_ESHAPE, "cfp_nyu_augment: the crop must fit the source image");
  AugP p;
  p.rgb = rgb_u8; p.depth_mm = depth_mm; p.pi = params_i; p.pf = params_f; p.pc = colors; p.img = image_out; p.dep = depth_out;
  p.B = B; p.H0 = H0; p.W0 = W0; p.H = H; p.W = W;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }       // HOST arrays (3 floats each)
  const long long total = (long long)B * H * W;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(nyu_augment_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return cfp_check_launch("cfp_nyu_augment");
}
