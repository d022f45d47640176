// Scale-invariant log loss (SILog) of the training step, forward and backward, f32.
//
//   p_i   = bilinear(pred -> target size, align_corners=True)          loss.py:10-11
//   g_i   = log p_i - log t_i            over the masked pixels          loss.py:13-16
//   loss  = 10 * sqrt(var_unbiased(g) + 0.15 * mean(g)^2)               loss.py:18-19
//
// Forward is two kernels: a grid-stride pass that interpolates, takes the logs, stores g_i and 1/p_i for
// the backward pass and reduces (sum g, sum g^2, count) in f64 per workgroup in a fixed order, and a
// one-workgroup finalize that combines the partials in order (run-to-run deterministic).
// Backward is a GATHER: every low-resolution prediction pixel collects d loss / d p_i from the few
// target-resolution pixels whose bilinear footprint touches it, so there are no atomics either.
//   d loss / d g_i = (50 / loss) * (2 (g_i - mean) / (n - 1) + 0.3 mean / n),    d g_i / d p_i = 1 / p_i
#include "common.h"

namespace {

constexpr int kLossBlocks = 512;

struct SilogP {
  const float* pred; const float* target; const unsigned char* mask;
  float* g; float* invp; double* partial; float* stats;
  int B, Hp, Wp, Ht, Wt, interpolate;
  float sy, sx;
};

__device__ __forceinline__ float interp_pred(const float* __restrict__ pb, int Hp, int Wp, int y, int x, float sy, float sx) {
  const float fy = sy * (float)y, fx = sx * (float)x;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = min(y0, Hp - 1); x0 = min(x0, Wp - 1);
  const int y1 = min(y0 + 1, Hp - 1), x1 = min(x0 + 1, Wp - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * pb[y0 * Wp + x0] + lx * pb[y0 * Wp + x1]) + ly * (hx * pb[y1 * Wp + x0] + lx * pb[y1 * Wp + x1]);
}

__global__ __launch_bounds__(256) void silog_fwd_kernel(SilogP p) {
  __shared__ double red[3][256];
  const long long hw = (long long)p.Ht * p.Wt, total = hw * p.B;
  double s = 0.0, s2 = 0.0, n = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / hw);
    const int r = (int)(i - (long long)b * hw);
    const int y = r / p.Wt, x = r - y * p.Wt;
    float gi = 0.f, ip = 0.f;
    if (p.mask == nullptr || p.mask[i]) {
      const float* pb = p.pred + (long long)b * p.Hp * p.Wp;
      const float pv = p.interpolate ? interp_pred(pb, p.Hp, p.Wp, y, x, p.sy, p.sx) : pb[r];
      gi = logf(pv) - logf(p.target[i]);
      ip = 1.f / pv;
      s += (double)gi; s2 += (double)gi * (double)gi; n += 1.0;
    }
    p.g[i] = gi;
    p.invp[i] = ip;      // 0 marks "not in the mask"
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2; red[2][threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    p.partial[blockIdx.x * 3 + 0] = red[0][0];
    p.partial[blockIdx.x * 3 + 1] = red[1][0];
    p.partial[blockIdx.x * 3 + 2] = red[2][0];
  }
}

// stats: [0] loss, [1] mean, [2] n, [3] Dg
// One wave: lanes stride over the block partials, then a fixed-order tree through LDS (a single thread walking them was 61 us on the
// critical path between the forward and the backward pass).
__global__ void silog_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ stats) {
  __shared__ double red[3][64];
  if (blockIdx.x != 0) return;
  double s = 0.0, s2 = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) { s += partial[i * 3]; s2 += partial[i * 3 + 1]; n += partial[i * 3 + 2]; }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2; red[2][threadIdx.x] = n;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  s = red[0][0]; s2 = red[1][0]; n = red[2][0];
  const double mean = s / n;
  const double var = (s2 - s * s / n) / (n - 1.0);      // unbiased, torch.var default
  const double dg = var + 0.15 * mean * mean;
  stats[0] = (float)(10.0 * sqrt(dg));
  stats[1] = (float)mean;
  stats[2] = (float)n;
  stats[3] = (float)dg;
}

struct SilogBwdP {
  const float* g; const float* invp; const float* stats; float* grad_pred;
  int B, Hp, Wp, Ht, Wt, interpolate;
  float sy, sx, grad_loss;
};

__global__ __launch_bounds__(256) void silog_bwd_kernel(SilogBwdP p) {
  const float loss = p.stats[0], mean = p.stats[1], n = p.stats[2];
  const float k = p.grad_loss * 50.f / loss;
  const float ca = k * 2.f / (n - 1.f), cb = k * 0.3f * mean / n;
  const long long hwp = (long long)p.Hp * p.Wp, total = hwp * p.B;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / hwp);
    const int r = (int)(i - (long long)b * hwp);
    const float* gb = p.g + (long long)b * p.Ht * p.Wt;
    const float* ib = p.invp + (long long)b * p.Ht * p.Wt;
    float acc = 0.f;
    if (!p.interpolate) {
      const float ip = ib[r];
      acc = ip != 0.f ? (ca * (gb[r] - mean) + cb) * ip : 0.f;
    } else {
      const int y = r / p.Wp, x = r - y * p.Wp;
      // target rows whose source coordinate sy*Y lies in (y-1, y+1): they touch prediction row y
      const int Y0 = p.sy > 0.f ? max(0, (int)ceilf(((float)y - 1.f) / p.sy) - 1) : 0;
      const int Y1 = p.sy > 0.f ? min(p.Ht - 1, (int)floorf(((float)y + 1.f) / p.sy) + 1) : p.Ht - 1;
      const int X0 = p.sx > 0.f ? max(0, (int)ceilf(((float)x - 1.f) / p.sx) - 1) : 0;
      const int X1 = p.sx > 0.f ? min(p.Wt - 1, (int)floorf(((float)x + 1.f) / p.sx) + 1) : p.Wt - 1;
      for (int Y = Y0; Y <= Y1; ++Y) {
        const float fy = p.sy * (float)Y;
        const int y0 = min((int)fy, p.Hp - 1), y1 = min(y0 + 1, p.Hp - 1);
        const float ly = fy - (float)y0;
        float wy = 0.f;                       // both taps can land on the same row at the bottom edge
        if (y == y0) wy += 1.f - ly;
        if (y == y1) wy += ly;
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
          const float fx = p.sx * (float)X;
          const int x0 = min((int)fx, p.Wp - 1), x1 = min(x0 + 1, p.Wp - 1);
          const float lx = fx - (float)x0;
          float wx = 0.f;
          if (x == x0) wx += 1.f - lx;
          if (x == x1) wx += lx;
          if (wx == 0.f) continue;
          const int t = Y * p.Wt + X;
          const float ip = ib[t];
          if (ip != 0.f) acc = fmaf(wy * wx, (ca * (gb[t] - mean) + cb) * ip, acc);
        }
      }
    }
    p.grad_pred[i] = acc;
  }
}

}  // namespace

extern "C" size_t cfp_silog_ws_bytes(int B, int Ht, int Wt) {
  if (B <= 0 || Ht <= 0 || Wt <= 0) return 0;
  const size_t n = (size_t)B * Ht * Wt;
  return 2 * n * sizeof(float) + 16 + (size_t)kLossBlocks * 3 * sizeof(double);
}

// ws layout: g [N] f32 | invp [N] f32 | (8-byte aligned) partial [kLossBlocks][3] f64
static inline double* silog_partial(void* ws, size_t n) {
  uintptr_t a = reinterpret_cast<uintptr_t>(ws) + 2 * n * sizeof(float);
  a = (a + 7) & ~(uintptr_t)7;
  return reinterpret_cast<double*>(a);
}

extern "C" int cfp_silog_loss_fwd(const float* pred, int Hp, int Wp, const float* target, const unsigned char* mask, int Ht, int Wt,
                                  int B, int interpolate, void* ws, size_t ws_bytes, float* stats, cfp_stream_t stream) {
  CFP_REQUIRE(pred && target && ws && stats, CFP_EINVAL, "cfp_silog_loss_fwd: null pointer");
  CFP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && Ht > 0 && Wt > 0, CFP_ESHAPE, "cfp_silog_loss_fwd: non-positive dimension");
  CFP_REQUIRE(interpolate || (Hp == Ht && Wp == Wt), CFP_ESHAPE, "cfp_silog_loss_fwd: sizes differ and interpolate is off");
  CFP_REQUIRE(ws_bytes >= cfp_silog_ws_bytes(B, Ht, Wt), CFP_EINVAL, "cfp_silog_loss_fwd: workspace too small");
  const size_t n = (size_t)B * Ht * Wt;
  SilogP p;
  p.pred = pred; p.target = target; p.mask = mask;
  p.g = reinterpret_cast<float*>(ws); p.invp = p.g + n; p.partial = silog_partial(ws, n); p.stats = stats;
  p.B = B; p.Hp = Hp; p.Wp = Wp; p.Ht = Ht; p.Wt = Wt; p.interpolate = interpolate;
  p.sy = Ht > 1 ? (float)(Hp - 1) / (float)(Ht - 1) : 0.f;
  p.sx = Wt > 1 ? (float)(Wp - 1) / (float)(Wt - 1) : 0.f;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(silog_fwd_kernel, dim3(kLossBlocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(silog_finalize_kernel, dim3(1), dim3(64), 0, s, p.partial, kLossBlocks, stats);
  return cfp_check_launch("cfp_silog_loss_fwd");
}

extern "C" int cfp_silog_loss_bwd(const void* ws, const float* stats, float grad_loss, int Hp, int Wp, int Ht, int Wt, int B,
                                  int interpolate, float* grad_pred, cfp_stream_t stream) {
  CFP_REQUIRE(ws && stats && grad_pred, CFP_EINVAL, "cfp_silog_loss_bwd: null pointer");
  CFP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && Ht > 0 && Wt > 0, CFP_ESHAPE, "cfp_silog_loss_bwd: non-positive dimension");
  CFP_REQUIRE(interpolate || (Hp == Ht && Wp == Wt), CFP_ESHAPE, "cfp_silog_loss_bwd: sizes differ and interpolate is off");
  const size_t n = (size_t)B * Ht * Wt;
  SilogBwdP p;
  p.g = reinterpret_cast<const float*>(ws); p.invp = p.g + n; p.stats = stats; p.grad_pred = grad_pred;
  p.B = B; p.Hp = Hp; p.Wp = Wp; p.Ht = Ht; p.Wt = Wt; p.interpolate = interpolate;
  p.sy = Ht > 1 ? (float)(Hp - 1) / (float)(Ht - 1) : 0.f;
  p.sx = Wt > 1 ? (float)(Wp - 1) / (float)(Wt - 1) : 0.f;
  p.grad_loss = grad_loss;
  const long long total = (long long)B * Hp * Wp;
  int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(silog_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return cfp_check_launch("cfp_silog_loss_bwd");
}
