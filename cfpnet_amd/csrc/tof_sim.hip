// ToF (L5 zone histogram) simulation from ground-truth depth: one workgroup per (image, zone).
//
//   reference: src/utils/dataloader.py:83-134 (get_hist_parallel) + :65-80 (sample_point_from_hist_parallel,
//   uniform branch); call site src/dataloader/nyu.py:154,179.  The reference runs this per sample on the CPU in
//   the data-loader workers (python loop over zones); here a whole batch is one launch reading each depth pixel once.
//
// Per zone (zone_px x zone_px patch of the depth map):
//   1. float32 histogram, bin = int(((x - 0) * bins) / (max_d - 0)) exactly as ATen's CPU histc evaluates it,
//      values outside [0, max_d] (and NaN) ignored, x == max_d folded into the last bin      -> LDS counters
//   2. bin 0 cleared, `floor_count` (20) subtracted and clamped at 0
//   3. the run of consecutive non-zero bins with the largest sum survives (first on ties)
//   4. n, mask = n > 0, mu and sigma over bin centres in float64 (ascending-bin order, no FMA contraction)
//   5. S samples w0*(mu-3sigma) + w1*(mu+3sigma) in float64 rounded to float32 (zero for masked-out zones)
//
// Roofline: HBM-bound by construction -- 4 bytes read per depth pixel inside the zone grid, ~200 bytes written per
// zone -- but at B=8 that is 6.4 MB, i.e. ~1.5 us of HBM time: the launch is latency bound (one pass over <= 16
// pixels per thread, LDS atomics, a <= 250-bin scan).
#include "common.h"

namespace {

constexpr int kTofThreads = 256;
constexpr int kTofMaxBins = 1024;
constexpr int kTofBatch = 8;         // pixels per thread in flight

// Step 5 for one (zone, sample): both branches of sample_point_from_hist_parallel (dataloader.py:65-80) in float64, rounded to
// float32.  UNIFORM: tensor_linspace(mu-3s, mu+3s) = w0*lo + w1*hi with the host's float32 linspace tables.  ICDF (the argparse
// default, `--sample_uniform` absent): torch.distributions.Normal(mu, s).icdf(ppf) = mu + s * erfinv(2 ppf - 1) * sqrt(2), where
// erfinv is evaluated in FLOAT32 on the 16 ppf points (the ppf tensor is float32, only the product is promoted) -- so that
// table (`w0` here) is an input evaluated on the host like the linspace tables; product order as in Normal.icdf.
__device__ __forceinline__ float tof_sample(double mu, double sg, float a, float b, int mode) {
  if (mode == CFP_TOF_SAMPLE_ICDF) return (float)(mu + (sg * (double)a) * 1.4142135623730951);
  const double lo = mu - 3.0 * sg, hi = mu + 3.0 * sg;
  return (float)((double)a * lo + (double)b * hi);
}

struct TofP {
  const float* depth; long long img_stride; int H, W;
  int zone_num, zone_px, sy0, sx0; const int* offsets; int offset_bound;
  float max_d; int bins; double bin_width; int floor_count;
  const float* w0; const float* w1; int nsamp; int sample_mode;
  double* fh; float* rect; unsigned char* mask; float* pts; int* hist_out;
};

__global__ __launch_bounds__(kTofThreads) void tof_hist_kernel(TofP p) {
  __shared__ int cnt[kTofMaxBins];
  __shared__ unsigned long long best;
  __shared__ double ms[2];
  const int tid = threadIdx.x;
  const int Z = p.zone_num * p.zone_num;
  const int b = blockIdx.x / Z, z = blockIdx.x - b * Z;
  const int zy = z / p.zone_num, zx = z - zy * p.zone_num;
  int off = p.offsets ? p.offsets[b] : 0;
  off = max(-p.offset_bound, min(p.offset_bound, off));
  const int sy = p.sy0 + off + zy * p.zone_px, sx = p.sx0 + off + zx * p.zone_px;

  for (int i = tid; i < p.bins; i += kTofThreads) cnt[i] = 0;
  if (tid == 0) best = 0ull;
  __syncthreads();

  // 1. histogram.  Lanes of a wave that hit the same bin are merged with a ballot so a flat zone (every pixel in
  //    two or three bins) does not serialise 64-way on one LDS counter.
  const float* img = p.depth + (long long)b * p.img_stride;
  const int npx = p.zone_px * p.zone_px;
  const float fb = (float)p.bins;
  for (int c0 = 0; c0 < npx; c0 += kTofThreads * kTofBatch) {
    // all loads of the batch are issued before the first bin is computed: one HBM round trip per 8 pixels/thread
    float v[kTofBatch];
#pragma unroll
    for (int u = 0; u < kTofBatch; ++u) {
      const int i = min(c0 + u * kTofThreads + tid, npx - 1);
      const int y = i / p.zone_px, x = i - y * p.zone_px;
      v[u] = img[(long long)(sy + y) * p.W + (sx + x)];
    }
#pragma unroll
    for (int u = 0; u < kTofBatch; ++u) {
      int bin = -1;
      if (c0 + u * kTofThreads + tid < npx && v[u] >= 0.f && v[u] <= p.max_d) {
        bin = (int)(((v[u] - 0.f) * fb) / (p.max_d - 0.f));
        if (bin == p.bins) bin = p.bins - 1;
      }
      unsigned long long todo = __ballot(bin >= 0);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lb = __builtin_amdgcn_readlane(bin, leader);
        const unsigned long long same = __ballot(bin == lb);
        if ((tid & 63) == leader) atomicAdd(&cnt[lb], (int)__popcll(same));
        todo &= ~same;
      }
    }
  }
  __syncthreads();

  // 2. ambient floor
  int hloc[kTofMaxBins / kTofThreads];
#pragma unroll
  for (int k = 0; k < kTofMaxBins / kTofThreads; ++k) {
    const int i = tid + k * kTofThreads;
    hloc[k] = (i < p.bins && i > 0) ? max(cnt[i] - p.floor_count, 0) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kTofMaxBins / kTofThreads; ++k) {
    const int i = tid + k * kTofThreads;
    if (i < p.bins) cnt[i] = hloc[k];
  }
  __syncthreads();

  // 3. every run start walks its run; key = (sum << 32) | ~start so the largest sum wins and, on ties, the lowest start
#pragma unroll
  for (int k = 0; k < kTofMaxBins / kTofThreads; ++k) {
    const int i = tid + k * kTofThreads;
    if (i < p.bins && hloc[k] > 0 && (i == 0 || cnt[i - 1] == 0)) {
      unsigned long long s = 0;
      for (int j = i; j < p.bins && cnt[j] > 0; ++j) s += (unsigned long long)cnt[j];
      atomicMax(&best, (s << 32) | (unsigned long long)(0xffffffffu - (unsigned)i));
    }
  }
  __syncthreads();
  const unsigned long long key = best;
  const int start = key ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : p.bins;
  int stop = start;
  while (stop < p.bins && cnt[stop] > 0) ++stop;          // every thread: <= run length LDS reads

  // 4. moments (one thread, ascending bins, float64)
  if (tid == 0) {
    const long long n = (long long)(key >> 32);
    const double nf = (double)((float)n + 1e-9f);           // the reference's `n + 1e-9` stays float32
    double acc = 0.0;
    for (int j = start; j < stop; ++j) {
      const double c = ((double)(float)((double)(j + 1) * p.bin_width) + (double)j * p.bin_width) / 2.0;
      acc += c * (double)cnt[j];
    }
    const double mu = acc / nf;
    double var = 0.0;
    for (int j = start; j < stop; ++j) {
      const double c = ((double)(float)((double)(j + 1) * p.bin_width) + (double)j * p.bin_width) / 2.0;
      const double d = c - mu;
      var += (double)cnt[j] * (d * d);
    }
    const double sigma = sqrt(var / nf) + 1e-9;
    ms[0] = mu; ms[1] = sigma;
    const long long o = (long long)b * Z + z;
    p.fh[o * 2] = mu; p.fh[o * 2 + 1] = sigma;
    p.mask[o] = n > 0 ? 1 : 0;
    p.rect[o * 4 + 0] = (float)sy; p.rect[o * 4 + 1] = (float)sx;
    p.rect[o * 4 + 2] = (float)(sy + p.zone_px); p.rect[o * 4 + 3] = (float)(sx + p.zone_px);
  }
  __syncthreads();

  // 5. samples
  const long long o = (long long)b * Z + z;
  if (tid < p.nsamp) {
    float v = 0.f;
    if (key) {
      v = tof_sample(ms[0], ms[1], p.w0[tid], p.w1 ? p.w1[tid] : 0.f, p.sample_mode);
    }
    p.pts[o * p.nsamp + tid] = v;
  }
  if (p.hist_out) {
    for (int i = tid; i < p.bins; i += kTofThreads) p.hist_out[o * p.bins + i] = (i >= start && i < stop) ? cnt[i] : 0;
  }
}

// stand-alone step 5 for callers that keep the reference's two-call structure
__global__ void tof_sample_kernel(const double* __restrict__ fh, const unsigned char* __restrict__ mask, const float* __restrict__ w0,
                                  const float* __restrict__ w1, long long nz, int nsamp, int sample_mode, float* __restrict__ pts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nz * nsamp) return;
  const long long z = i / nsamp;
  const int t = (int)(i - z * nsamp);
  float v = 0.f;
  if (mask[z]) {
    v = tof_sample(fh[z * 2], fh[z * 2 + 1], w0[t], w1 ? w1[t] : 0.f, sample_mode);
  }
  pts[i] = v;
}

}  // namespace

extern "C" int cfp_tof_sample_points(const double* fh, const unsigned char* mask, const float* w0, const float* w1, long long nzones,
                                     int nsamp, int sample_mode, float* pts, cfp_stream_t stream) {
  CFP_REQUIRE(fh && mask && w0 && pts && (w1 || sample_mode == CFP_TOF_SAMPLE_ICDF), CFP_EINVAL, "cfp_tof_sample_points: null pointer");
  CFP_REQUIRE(nzones >= 0 && nsamp > 0, CFP_ESHAPE, "cfp_tof_sample_points: bad sizes");
  CFP_REQUIRE(sample_mode == CFP_TOF_SAMPLE_UNIFORM || sample_mode == CFP_TOF_SAMPLE_ICDF, CFP_EINVAL, "cfp_tof_sample_points: bad sample_mode");
  if (nzones == 0) return CFP_OK;
  const long long total = nzones * nsamp;
  hipLaunchKernelGGL(tof_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), fh, mask,
                     w0, w1, nzones, nsamp, sample_mode, pts);
  return cfp_check_launch("cfp_tof_sample_points");
}

extern "C" int cfp_tof_hist_sim(const float* depth, long long img_stride, int B, int H, int W, int zone_num, int zone_px,
                                int sy0, int sx0, const int* offsets, int offset_bound, float max_distance, int bins,
                                double bin_width, int floor_count, const float* w0, const float* w1, int nsamp, int sample_mode,
                                double* fh, float* rect, unsigned char* mask, float* pts, int* hist_out, cfp_stream_t stream) {
  CFP_REQUIRE(depth && w0 && (w1 || sample_mode == CFP_TOF_SAMPLE_ICDF) && fh && rect && mask && pts, CFP_EINVAL, "cfp_tof_hist_sim: null pointer");
  CFP_REQUIRE(sample_mode == CFP_TOF_SAMPLE_UNIFORM || sample_mode == CFP_TOF_SAMPLE_ICDF, CFP_EINVAL, "cfp_tof_hist_sim: bad sample_mode");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && zone_num > 0 && zone_px > 0, CFP_ESHAPE, "cfp_tof_hist_sim: non-positive dimension");
  CFP_REQUIRE(img_stride >= (long long)H * W, CFP_ESHAPE, "cfp_tof_hist_sim: image stride smaller than H*W");
  CFP_REQUIRE(bins > 0 && bins <= kTofMaxBins, CFP_ESHAPE, "cfp_tof_hist_sim: bins must be in 1..1024");
  CFP_REQUIRE(nsamp > 0 && nsamp <= kTofThreads, CFP_ESHAPE, "cfp_tof_hist_sim: samples per zone must be in 1..256");
  CFP_REQUIRE(max_distance > 0.f && bin_width > 0.0 && floor_count >= 0, CFP_EINVAL, "cfp_tof_hist_sim: bad histogram parameters");
  CFP_REQUIRE(offset_bound >= 0 && (offsets || offset_bound == 0), CFP_EINVAL, "cfp_tof_hist_sim: offset bound without offsets");
  // the zone grid must stay inside the image for every admissible offset (the reference's slicing would shrink the
  // grid and fail in unfold/view instead)
  const long long span = (long long)zone_px * zone_num;
  CFP_REQUIRE((long long)sy0 - offset_bound >= 0 && (long long)sx0 - offset_bound >= 0 && sy0 + offset_bound + span <= H &&
                  sx0 + offset_bound + span <= W,
              CFP_ESHAPE, "cfp_tof_hist_sim: zone grid leaves the image");
  CFP_REQUIRE((long long)B * zone_num * zone_num < (1ll << 31), CFP_ESHAPE, "cfp_tof_hist_sim: too many zones");
  TofP p;
  p.depth = depth; p.img_stride = img_stride; p.H = H; p.W = W;
  p.zone_num = zone_num; p.zone_px = zone_px; p.sy0 = sy0; p.sx0 = sx0; p.offsets = offsets; p.offset_bound = offset_bound;
  p.max_d = max_distance; p.bins = bins; p.bin_width = bin_width; p.floor_count = floor_count;
  p.w0 = w0; p.w1 = w1; p.nsamp = nsamp; p.sample_mode = sample_mode;
  p.fh = fh; p.rect = rect; p.mask = mask; p.pts = pts; p.hist_out = hist_out;
  hipLaunchKernelGGL(tof_hist_kernel, dim3(B * zone_num * zone_num), dim3(kTofThreads), 0, reinterpret_cast<hipStream_t>(stream), p);
  return cfp_check_launch("cfp_tof_hist_sim");
}
