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


// ---- random rotation (nyu.py:121-124: PIL Image.rotate, BILINEAR on the RGB image, NEAREST on the 16-bit depth) --------------
// Pillow's arithmetic restated (Image.rotate builds the destination->source matrix on the host; libImaging/Geometry.c:
// ImagingGenericTransform + affine_transform + bilinear_filter32RGB / nearest_filter16), all in float64 with the same operation
// order (the build has FMA contraction off), so the result equals Pillow's byte for byte.
struct RotP {
  const unsigned char* rgb; const unsigned short* dep; unsigned char* rgb_out; unsigned short* dep_out;
  const double* m;      // [B][6]
  int B, H, W;
};
__global__ __launch_bounds__(256) void nyu_rotate_kernel(RotP p) {
  const long long total = (long long)p.B * p.H * p.W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % p.W);
    const long long t = i / p.W;
    const int y = (int)(t % p.H), b = (int)(t / p.H);
    const double* m = p.m + b * 6;
    const double xc = (double)x + 0.5, yc = (double)y + 0.5;
    const double xin = m[0] * xc + m[1] * yc + m[2];
    const double yin = m[3] * xc + m[4] * yc + m[5];
    if (p.rgb) {
      unsigned char* o = p.rgb_out + i * 3;
      if (xin < 0.0 || xin >= (double)p.W || yin < 0.0 || yin >= (double)p.H) {
        o[0] = 0; o[1] = 0; o[2] = 0;
      } else {
        const double xf = xin - 0.5, yf = yin - 0.5;
        const int xq = (int)floor(xf), yq = (int)floor(yf);
        const double dx = xf - (double)xq, dy = yf - (double)yq;
        const int x0 = min(max(xq, 0), p.W - 1), x1 = min(max(xq + 1, 0), p.W - 1);
        const int y0 = min(max(yq, 0), p.H - 1);
        const bool has_y1 = yq + 1 >= 0 && yq + 1 < p.H;
        const unsigned char* r0 = p.rgb + ((long long)b * p.H + y0) * p.W * 3;
        const unsigned char* r1 = p.rgb + ((long long)b * p.H + (has_y1 ? yq + 1 : y0)) * p.W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double a0 = (double)r0[x0 * 3 + c], a1 = (double)r0[x1 * 3 + c];
          const double v1 = a0 + (a1 - a0) * dx;
          double v2 = v1;
          if (has_y1) {
            const double b0 = (double)r1[x0 * 3 + c], b1 = (double)r1[x1 * 3 + c];
            v2 = b0 + (b1 - b0) * dx;
          }
          o[c] = (unsigned char)(v1 + (v2 - v1) * dy);
        }
      }
    }
    if (p.dep) {
      const int xs = xin < 0.0 ? -1 : (int)xin, ys = yin < 0.0 ? -1 : (int)yin;
      const bool ok = xs >= 0 && xs < p.W && ys >= 0 && ys < p.H;
      p.dep_out[i] = ok ? p.dep[((long long)b * p.H + ys) * p.W + xs] : (unsigned short)0;
    }
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

extern "C" int cfp_nyu_rotate(const unsigned char* rgb_u8, const unsigned short* depth_mm, unsigned char* rgb_out, unsigned short* depth_out,
                              int B, int H, int W, const double* matrices, cfp_stream_t stream) {
  CFP_REQUIRE(matrices && (rgb_u8 || depth_mm), CFP_EINVAL, "cfp_nyu_rotate: null pointer");
  CFP_REQUIRE((rgb_u8 == nullptr) == (rgb_out == nullptr) && (depth_mm == nullptr) == (depth_out == nullptr), CFP_EINVAL,
              "cfp_nyu_rotate: every input needs its output");
  CFP_REQUIRE(rgb_u8 != rgb_out || !rgb_u8, CFP_EINVAL, "cfp_nyu_rotate: not in place");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0, CFP_ESHAPE, "cfp_nyu_rotate: bad shape");
  RotP p;
  p.rgb = rgb_u8; p.dep = depth_mm; p.rgb_out = rgb_out; p.dep_out = depth_out; p.m = matrices; p.B = B; p.H = H; p.W = W;
  const long long total = (long long)B * H * W;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(nyu_rotate_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return cfp_check_launch("cfp_nyu_rotate");
}
