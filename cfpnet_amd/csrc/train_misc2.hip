// Backward kernels, part 2: row gather by index map (crop / regroup / window partition / inside-outside partition and
// their adjoints), bilinear resize backward, the bin head (softmax + expectation over adaptive bin centres) forward and
// backward, and the bin-width -> centre chain.
//
//   reference: autograd of fusion.py:132-157 (crop, F.interpolate, rearrange, masked scatter), transformer.py:101-116,
//   215-234 (window partition, mask partition), decoder.py:56 (upsampling), deltar.py:51-61 (softmax, bin centres,
//   expectation) in model.train().
#include "common.h"

namespace {

inline int ew_grid3(long long total) { long long b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// out[i] = idx[i] >= 0 ? x[idx[i]] : 0   (+ out[i] when accumulate).  Every structural rearrangement of the model is an
// injective partial map of token rows, so the adjoint of a gather with map m is the gather with m's inverse map.
template <typename T>
__global__ __launch_bounds__(256) void index_rows_kernel(const T* __restrict__ x, int x_ld, const int* __restrict__ idx, T* __restrict__ out,
                                                         int out_ld, long long n_out, int C, int accumulate, FastDiv fcv) {
  constexpr int VE = Vec<T>::N;
  const unsigned total = (unsigned)(n_out * fcv.d);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned ru, cvu;
    fd_rowcol(i, fcv, ru, cvu);
    const long long r = ru;
    const int c = (int)cvu * VE;
    const int src = idx[r];
    float v[VE];
    if (src >= 0) Vec<T>::load(x + (long long)src * x_ld + c, v);
    else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = 0.f;
    }
    if (accumulate) {
      float o[VE];
      Vec<T>::load(out + r * out_ld + c, o);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += o[e];
    }
    Vec<T>::store(out + r * out_ld + c, v);
  }
}

// Backward of bilinear resize (align_corners=True) [B,Hs,Ws,C] -> [B,Hd,Wd,C] as a GATHER: every source pixel collects
// from the destination pixels whose 2x2 footprint contains it, with the forward kernel's own index / weight arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, int dy_ld, T* __restrict__ dx, int dx_ld, int B, int Hs,
                                                         int Ws, int Hd, int Wd, int C, float scale_y, float scale_x, float inv_y, float inv_x,
                                                         int accumulate) {
  constexpr int VE = Vec<T>::N;
  const int CV = C / VE;
  const long long total = (long long)B * Hs * Ws * CV;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long long t = i / CV;
    const int xs = (int)(t % Ws); t /= Ws;
    const int ys = (int)(t % Hs);
    const int b = (int)(t / Hs);
    // candidate destination rows: those with floor(scale_y * yd) in {ys - 1, ys}
    int yd0 = (int)floorf((float)(ys - 1) * inv_y) - 1, yd1 = (int)ceilf((float)(ys + 1) * inv_y) + 1;
    int xd0 = (int)floorf((float)(xs - 1) * inv_x) - 1, xd1 = (int)ceilf((float)(xs + 1) * inv_x) + 1;
    yd0 = max(yd0, 0); yd1 = min(yd1, Hd - 1); xd0 = max(xd0, 0); xd1 = min(xd1, Wd - 1);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int yd = yd0; yd <= yd1; ++yd) {
      const float fy = scale_y * (float)yd;
      const int y0 = (int)fy;
      const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
      const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
      const float wy = (y0 == ys ? ly0 : 0.f) + (y1 == ys ? ly1 : 0.f);
      if (wy == 0.f && y0 != ys && y1 != ys) continue;
      for (int xd = xd0; xd <= xd1; ++xd) {
        const float fx = scale_x * (float)xd;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
        const float wx = (x0 == xs ? lx0 : 0.f) + (x1 == xs ? lx1 : 0.f);
        if (x0 != xs && x1 != xs) continue;
        float g[VE];
        Vec<T>::load(dy + (((long long)b * Hd + yd) * Wd + xd) * dy_ld + cv * VE, g);
        const float w = wy * wx;
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = fmaf(w, g[e], acc[e]);
      }
    }
    T* dp = dx + (((long long)b * Hs + ys) * Ws + xs) * dx_ld + cv * VE;
    if (accumulate) {
      float o[VE];
      Vec<T>::load(dp, o);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += o[e];
    }
    Vec<T>::store(dp, acc);
  }
}

// ---- bin head --------------------------------------------------------------------------------------------------
// widths_normed [B][NB] -> edges [B][NB+1], centres [B][NB]  (deltar.py:53-59): one thread per image, sequential cumsum
__global__ void bin_centers_kernel(const float* __restrict__ wn, float min_val, float max_val, float* __restrict__ edges,
                                   float* __restrict__ centers, int B, int NB) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float e = min_val;
  edges[(long long)b * (NB + 1)] = e;
  for (int i = 0; i < NB; ++i) {
    const float n = e + (max_val - min_val) * wn[(long long)b * NB + i];
    centers[(long long)b * NB + i] = 0.5f * (e + n);
    edges[(long long)b * (NB + 1) + i + 1] = n;
    e = n;
  }
}
// adjoint: dwn[i] = (max-min) * (0.5 * dc[i] + sum_{j > i} dc[j])
__global__ void bin_centers_bwd_kernel(const float* __restrict__ dcenters, float min_val, float max_val, float* __restrict__ dwn, int B, int NB) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float tail = 0.f;
  for (int i = NB - 1; i >= 0; --i) {
    const float d = dcenters[(long long)b * NB + i];
    dwn[(long long)b * NB + i] = (max_val - min_val) * (0.5f * d + tail);
    tail += d;
  }
}

// pred[r] = sum_n softmax(logits[r])[n] * centers[b][n]; one wave per pixel row (NB = 64 * NPL bins), HW rows per image.
// Backward: dlogits[r][n] = p[n] * (c[n] - pred[r]) * dpred[r];  dcenters[b][n] = sum_r p[n] * dpred[r] (per-block partials).
template <typename T, int NPL>
__global__ __launch_bounds__(256) void softmax_expect_kernel(const T* __restrict__ logits, int ld, const float* __restrict__ centers,
                                                             float* __restrict__ pred, const float* __restrict__ dpred, T* __restrict__ dlogits,
                                                             int dl_ld, float* __restrict__ dc_partial, long long rows, int HW,
                                                             int rows_per_block) {
  constexpr int NB = 64 * NPL;
  __shared__ float sdc[4][NB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const bool bwd = dpred != nullptr;
  float dc[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) dc[k] = 0.f;
  const int b_blk = (int)(r0 / HW);       // rows_per_block divides HW: a block stays inside one image
  // lane owns the NPL CONTIGUOUS bins NPL * lane .. + NPL - 1: one 8 / 16-byte load per row instead of NPL strided 2-byte ones, and two
  // rows per wave in flight (the one-row loop was a chain of memory round trips: 263 us forward for 463 MB at 16 crops)
  float cen[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) cen[k] = centers[(long long)b_blk * NB + NPL * lane + k];
  constexpr int RU = 4;
  for (long long r = r0 + wave; r < r1; r += 4 * RU) {
    T raw[RU][NPL];
    float g[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long long rr = min(r + 4 * u, r1 - 1);
      __builtin_memcpy(raw[u], __builtin_assume_aligned(logits + rr * ld + NPL * lane, sizeof(T) * NPL), sizeof(T) * NPL);
      g[u] = bwd ? dpred[rr] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long long rr = r + 4 * u;
      if (rr >= r1) continue;                           // wave-uniform
      float v[NPL];
      float mx = -3.0e38f;
#pragma unroll
      for (int k = 0; k < NPL; ++k) { v[k] = to_f32<T>(raw[u][k]); mx = fmaxf(mx, v[k]); }
      mx = wave_max(mx);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {      // full-precision exp in the float32 parity mode; v_exp_f32 under 16-bit storage (1e-6 vs 1e-3)
        v[k] = sizeof(T) == 2 ? __expf(v[k] - mx) : expf(v[k] - mx);
        s += v[k];
      }
      s = wave_sum(s);
      const float inv = 1.f / s;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < NPL; ++k) { v[k] *= inv; dot = fmaf(v[k], cen[k], dot); }
      dot = wave_sum(dot);
      if (!bwd) {
        if (lane == 0) pred[rr] = dot;
      } else {
        T o[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          o[k] = from_f32<T>(v[k] * (cen[k] - dot) * g[u]);
          dc[k] = fmaf(v[k], g[u], dc[k]);
        }
        __builtin_memcpy(__builtin_assume_aligned(dlogits + rr * dl_ld + NPL * lane, sizeof(T) * NPL), o, sizeof(T) * NPL);
      }
    }
  }
  if (bwd) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) sdc[wave][NPL * lane + k] = dc[k];
    __syncthreads();
    for (int n = threadIdx.x; n < NB; n += 256)
      dc_partial[(long long)blockIdx.x * NB + n] = (sdc[0][n] + sdc[1][n]) + (sdc[2][n] + sdc[3][n]);
  }
}

// dcenters[b][n] = sum over the image's blocks of the partials (fixed order)
__global__ void dc_reduce_kernel(const float* __restrict__ partial, int blocks_per_image, int NB, float* __restrict__ dcenters) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (n >= NB) return;
  float s = 0.f;
  for (int j = 0; j < blocks_per_image; ++j) s += partial[((long long)b * blocks_per_image + j) * NB + n];
  dcenters[(long long)b * NB + n] = s;
}

inline int se_rows_per_block(int HW) {      // largest divisor of HW that is <= 256 (a block never straddles two images)
  for (int r = 256; r >= 1; --r)
    if (HW % r == 0) return r;
  return 1;
}

}  // namespace

#define T2_COMMON(name)                                          \
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, name ": bad dtype");  \
  const int ve = vec_elems(dtype);                               \
  hipStream_t s = reinterpret_cast<hipStream_t>(stream)

extern "C" int cfp_index_rows(const void* x, int x_ld, const int* idx, void* out, int out_ld, long long n_out, int C, int accumulate,
                              int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(x && idx && out && aligned16(x) && aligned16(out), CFP_EINVAL, "cfp_index_rows: bad pointer");
  T2_COMMON("cfp_index_rows");
  CFP_REQUIRE(n_out > 0 && C > 0 && C % ve == 0 && x_ld % ve == 0 && out_ld % ve == 0 && x_ld >= C && out_ld >= C, CFP_ESHAPE,
              "cfp_index_rows: bad shape");
  const dim3 grid(ew_grid3(n_out * (C / ve)));
  CFP_REQUIRE(n_out * (C / ve) < (1ll << 31), CFP_ESHAPE, "cfp_index_rows: too many elements");
  const FastDiv fcv = make_fastdiv((unsigned)(C / ve));
#define L(T) hipLaunchKernelGGL(index_rows_kernel<T>, grid, dim3(256), 0, s, (const T*)x, x_ld, idx, (T*)out, out_ld, n_out, C, accumulate, fcv)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  return cfp_check_launch("cfp_index_rows");
}

extern "C" int cfp_resize_bilinear_bwd(const void* dy, int dy_ld, void* dx, int dx_ld, int B, int Hs, int Ws, int Hd, int Wd, int C,
                                       int accumulate, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dy && dx && aligned16(dy) && aligned16(dx), CFP_EINVAL, "cfp_resize_bilinear_bwd: bad pointer");
  T2_COMMON("cfp_resize_bilinear_bwd");
  CFP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && C % ve == 0 && dy_ld % ve == 0 && dx_ld % ve == 0 && dy_ld >= C &&
                  dx_ld >= C, CFP_ESHAPE, "cfp_resize_bilinear_bwd: bad shape");
  const float sy = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f, sx = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
  const float iy = sy > 0.f ? 1.f / sy : (float)Hd, ix = sx > 0.f ? 1.f / sx : (float)Wd;
  const dim3 grid(ew_grid3((long long)B * Hs * Ws * (C / ve)));
#define L(T) hipLaunchKernelGGL(resize_bwd_kernel<T>, grid, dim3(256), 0, s, (const T*)dy, dy_ld, (T*)dx, dx_ld, B, Hs, Ws, Hd, Wd, C, sy, sx, iy, ix, \
                                accumulate)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  return cfp_check_launch("cfp_resize_bilinear_bwd");
}

extern "C" int cfp_bin_centers(const float* widths_normed, float min_val, float max_val, float* edges, float* centers, int B, int NB,
                               cfp_stream_t stream) {
  CFP_REQUIRE(widths_normed && edges && centers && B > 0 && NB > 0, CFP_EINVAL, "cfp_bin_centers: bad argument");
  hipLaunchKernelGGL(bin_centers_kernel, dim3(cdiv(B, 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), widths_normed, min_val, max_val,
                     edges, centers, B, NB);
  return cfp_check_launch("cfp_bin_centers");
}

extern "C" int cfp_bin_centers_bwd(const float* dcenters, float min_val, float max_val, float* dwidths_normed, int B, int NB,
                                   cfp_stream_t stream) {
  CFP_REQUIRE(dcenters && dwidths_normed && B > 0 && NB > 0, CFP_EINVAL, "cfp_bin_centers_bwd: bad argument");
  hipLaunchKernelGGL(bin_centers_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), dcenters, min_val, max_val,
                     dwidths_normed, B, NB);
  return cfp_check_launch("cfp_bin_centers_bwd");
}

extern "C" size_t cfp_softmax_expect_ws_bytes(int B, int HW, int NB) {
  if (B <= 0 || HW <= 0 || NB <= 0) return 0;
  return (size_t)B * (HW / se_rows_per_block(HW)) * NB * sizeof(float);
}

/* dpred == NULL: forward (pred out).  Otherwise backward: dlogits and dcenters out (pred unused). */
extern "C" int cfp_softmax_expect(const void* logits, int ld, const float* centers, float* pred, const float* dpred, void* dlogits, int dl_ld,
                                  float* dcenters, int B, int HW, int NB, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(logits && centers && aligned16(logits), CFP_EINVAL, "cfp_softmax_expect: bad pointer");
  T2_COMMON("cfp_softmax_expect");
  (void)ve;
  CFP_REQUIRE(B > 0 && HW > 0 && (NB == 64 || NB == 128 || NB == 256) && ld >= NB, CFP_ESHAPE, "cfp_softmax_expect: NB must be 64, 128 or 256");
  CFP_REQUIRE(ld % 4 == 0 && (!dlogits || (dl_ld % 4 == 0 && aligned16(dlogits))), CFP_ESHAPE, "cfp_softmax_expect: pitches must be multiples of 4 elements");
  const bool bwd = dpred != nullptr;
  CFP_REQUIRE(bwd ? (dlogits && dcenters && ws && dl_ld >= NB && ws_bytes >= cfp_softmax_expect_ws_bytes(B, HW, NB)) : (pred != nullptr), CFP_EINVAL,
              "cfp_softmax_expect: missing output / workspace");
  const int rpb = se_rows_per_block(HW);
  const long long rows = (long long)B * HW;
  const int blocks = (int)(rows / rpb);
  float* partial = reinterpret_cast<float*>(ws);
#define L(T, N) hipLaunchKernelGGL((softmax_expect_kernel<T, N>), dim3(blocks), dim3(256), 0, s, (const T*)logits, ld, centers, pred, dpred, (T*)dlogits, \
                                   dl_ld, partial, rows, HW, rpb)
#define LN(T) do { if (NB == 256) L(T, 4); else if (NB == 128) L(T, 2); else L(T, 1); } while (0)
  if (dtype == CFP_BF16) LN(bf16_t); else if (dtype == CFP_F16) LN(f16_t); else LN(float);
#undef LN
#undef L
  if (bwd) hipLaunchKernelGGL(dc_reduce_kernel, dim3(cdiv(NB, 64), B), dim3(64), 0, s, partial, HW / rpb, NB, dcenters);
  return cfp_check_launch("cfp_softmax_expect");
}

// ---- large-kernel depthwise weight gradient (LKPM dwconv, k = 7 / 15 / 31) ---------------------------------------------
// dw[c][ky][kx] = sum over (b, y, x) of dy[b,y,x,c] * x[b, y + ky - h, x + kx - h, c]   (h = (k-1)/2, zero padding)
// One workgroup per (32 x 32 pixel tile, 16-byte channel vector): dy tile and the haloed x tile are staged in LDS once,
// thread t owns taps t, t + 256, ...; every pixel's dy vector is an LDS broadcast, the x reads of neighbouring taps are
// neighbouring vectors.  Per-tile partial sums are added in tile order by a second kernel (bit-reproducible).
int g_dwl_wgrad_valu = 0;          // cfp_debug_set key 23: 1 = the VALU kernels for 16-bit storage too (A/B, tests)
namespace {
constexpr int LWT = 32;

template <typename T, int K>
__global__ __launch_bounds__(256) void dwlarge_wgrad_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dy, int dy_ld,
                                                            float* __restrict__ partial, int B, int H, int W, int C) {
  constexpr int VE = Vec<T>::N;
  constexpr int HALO = (K - 1) / 2, PW = LWT + K - 1;
  constexpr int NT = (K * K + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
  T* sx = reinterpret_cast<T*>(lsm);                       // [PW][PW][VE]
  T* sd = sx + PW * PW * VE;                               // [LWT][LWT][VE]
  const int tid = threadIdx.x;
  const int CV = C / VE;
  const int tiles_x = (W + LWT - 1) / LWT, tiles_y = (H + LWT - 1) / LWT;
  int bid = blockIdx.x;
  const int cv = bid % CV; bid /= CV;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * LWT, x0 = tx * LWT;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int i = tid; i < PW * PW; i += 256) {
    const int py = i / PW, px = i - py * PW;
    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
    const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long long)b * H + cy) * W + cx) * x_ld + cv * VE);
    *reinterpret_cast<u32x4*>(sx + i * VE) = ok ? v : zero4;
  }
  for (int i = tid; i < LWT * LWT; i += 256) {
    const int py = i / LWT, px = i - py * LWT;
    const int gy = y0 + py, gx = x0 + px;
    const bool ok = gy < H && gx < W;
    const int cy = min(gy, H - 1), cx = min(gx, W - 1);
    const u32x4 v = *reinterpret_cast<const u32x4*>(dy + (((long long)b * H + cy) * W + cx) * dy_ld + cv * VE);
    *reinterpret_cast<u32x4*>(sd + i * VE) = ok ? v : zero4;
  }
  __syncthreads();
  float acc[NT][VE];
  int toff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = min(tid + t * 256, K * K - 1);
    toff[t] = ((tap / K) * PW + tap % K) * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[t][e] = 0.f;
  }
  for (int py = 0; py < LWT; ++py) {
    for (int px = 0; px < LWT; ++px) {
      float g[VE];
      Vec<T>::load(sd + (py * LWT + px) * VE, g);
      const int base = (py * PW + px) * VE;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float v[VE];
        Vec<T>::load(sx + base + toff[t], v);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[t][e] = fmaf(g[e], v[e], acc[t][e]);
      }
    }
  }
  float* dst = partial + (long long)blockIdx.x * K * K * VE;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = tid + t * 256;
    if (tap < K * K) {
#pragma unroll
      for (int e = 0; e < VE; ++e) dst[tap * VE + e] = acc[t][e];
    }
  }
}

// k = 31: a thread owns FOUR CONSECUTIVE taps of one kernel row (31 rows x 8 threads) and slides a 4-vector window along the
// pixel row: per pixel one new x vector and the dy vector are read and unpacked instead of four x vectors (the generic kernel
// above spends more VALU issue on bf16 unpacking and LDS reads than on the 32 FMAs).  Same staging, same partial layout.
template <typename T>
__global__ __launch_bounds__(256) void dwlarge_wgrad31_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dy, int dy_ld,
                                                              float* __restrict__ partial, int B, int H, int W, int C) {
  constexpr int VE = Vec<T>::N, K = 31;
  constexpr int HALO = (K - 1) / 2, PW = LWT + K - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
  T* sx = reinterpret_cast<T*>(lsm);                       // [PW][PW][VE]
  T* sd = sx + PW * PW * VE;                               // [LWT][LWT][VE]
  const int tid = threadIdx.x;
  const int CV = C / VE;
  const int tiles_x = (W + LWT - 1) / LWT, tiles_y = (H + LWT - 1) / LWT;
  int bid = blockIdx.x;
  const int cv = bid % CV; bid /= CV;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * LWT, x0 = tx * LWT;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int i = tid; i < PW * PW; i += 256) {
    const int py = i / PW, px = i - py * PW;
    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
    const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long long)b * H + cy) * W + cx) * x_ld + cv * VE);
    *reinterpret_cast<u32x4*>(sx + i * VE) = ok ? v : zero4;
  }
  for (int i = tid; i < LWT * LWT; i += 256) {
    const int py = i / LWT, px = i - py * LWT;
    const int gy = y0 + py, gx = x0 + px;
    const bool ok = gy < H && gx < W;
    const int cy = min(gy, H - 1), cx = min(gx, W - 1);
    const u32x4 v = *reinterpret_cast<const u32x4*>(dy + (((long long)b * H + cy) * W + cx) * dy_ld + cv * VE);
    *reinterpret_cast<u32x4*>(sd + i * VE) = ok ? v : zero4;
  }
  __syncthreads();
  const int ky = min(tid >> 3, K - 1), kx0 = (tid & 7) * 4;          // threads 248..255 repeat row 30 and store nothing
  float acc[4][VE];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[j][e] = 0.f;
  for (int py = 0; py < LWT; ++py) {
    // x[py + ky][kx0 + px + j]: the last lanes' window runs up to 3 vectors past the row end (into the next row / the dy tile,
    // inside the LDS allocation); those products land in taps that are not stored
    const T* row = sx + ((py + ky) * PW + kx0) * VE;
    const T* drow = sd + py * LWT * VE;
    float w[4][VE];
#pragma unroll
    for (int j = 0; j < 4; ++j) Vec<T>::load(row + j * VE, w[j]);
#pragma unroll 2
    for (int p4 = 0; p4 < LWT; p4 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float g[VE];
        Vec<T>::load(drow + (p4 + u) * VE, g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < VE; ++e) acc[j][e] = fmaf(g[e], w[(u + j) & 3][e], acc[j][e]);
        Vec<T>::load(row + (p4 + u + 4) * VE, w[u]);           // x[.. + px + 4] replaces x[.. + px]
      }
    }
  }
  if (tid < K * 8) {
    float* dst = partial + (long long)blockIdx.x * K * K * VE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kx = kx0 + j;
      if (kx < K) {
#pragma unroll
        for (int e = 0; e < VE; ++e) dst[(ky * K + kx) * VE + e] = acc[j][e];
      }
    }
  }
}

// 16-bit storage (k = 31 / 15 / 7): the same sums on the matrix cores.  Per channel and per row r of the haloed x tile,
//   dW[ky][kx] += sum_x  X[r][x + kx] * dY[r - ky][x]           (x = 0..31: ONE k = 32 MFMA step)
// is a 16 x 16 x 32 product with A[m = kx][k = x] = X[r][x + kx] (a Toeplitz view of one x row) and B[k = x][n = ky] = dY[r - ky][x] (the
// dY rows the x row meets), accumulated over the 62 (46) rows r: 186 (46) MFMAs per channel and tile instead of 31 * 31 * 1024 FMAs on
// the vector ALUs (the VALU kernel above ran at 53 % of the float32 VALU peak: 334 us per launch at 16 x 104 x 136 x 32).
// The tile is staged TRANSPOSED into per-channel planes (the operands want x along k, the tensors are channels-last); a wave owns one
// of the 8 channels of the vector.  The A fragment of lane (m, q) starts at element 8 q + kx of the row -- any 16-bit offset -- so it
// reads the five dwords around it and funnel-shifts odd offsets by 16 bits (v_alignbit_b32); the B fragment is one aligned 16-byte
// read of row r - ky, zero where that row is outside the dY tile.  Products are exact (16-bit x 16-bit in float32), sums float32 like
// the VALU kernels; same partial layout, same reduce kernel.
template <typename H, int K>
__global__ __launch_bounds__(512, 2) void dwlarge_wgrad_mfma_kernel(const H* __restrict__ x, int x_ld, const H* __restrict__ dy, int dy_ld,
                                                                 float* __restrict__ partial, int B, int Hh, int W, int C) {
  constexpr int HALO = (K - 1) / 2, PW = LWT + K - 1;
  constexpr int NTL = (K + 15) / 16;                       // 16-wide tap tiles along kx and along ky
  constexpr int PX = 64, PD = 36;                          // plane row pitches in elements (x: 62 + the fragment's look-ahead; dY: 32 + 4: rows stay
                                                           // 8-byte aligned and k = 31 takes 81 920 B of LDS -- two workgroups per CU)
  constexpr int XPLANE = PW * PX, DPLANE = LWT * PD;
  static_assert(PW + 2 <= PX && (PX % 2) == 0 && (PD % 4) == 0, "plane pitches");
  extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
  unsigned short* sx = reinterpret_cast<unsigned short*>(lsm);            // [8][PW][PX]
  unsigned short* sd = sx + 8 * XPLANE;                                   // [8][LWT][PD]
  const int tid = threadIdx.x;
  const int CV = C / 8;
  const int tiles_x = (W + LWT - 1) / LWT, tiles_y = (Hh + LWT - 1) / LWT;
  // The CV workgroups of one pixel tile read the same cache lines (a pixel's channels are contiguous: each takes 16 of every 2 C bytes).
  // Hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so ids are laid out (tile group, channel vector, XCD): the CV
  // workgroups of a tile run on ONE XCD, eight launch slots apart, and three of four fetches hit its L2 (2 560 x 254 KB = 650 MB of
  // fabric traffic per launch at k = 31 otherwise).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cv = slot % CV;
  int bid = (slot / CV) * 8 + xcd;                          // pixel tile
  if (bid >= B * tiles_y * tiles_x) return;                 // padding of the tile count to a multiple of 8 (before any barrier)
  const long long pidx = (long long)bid * CV + cv;          // partial slot: [tile][cv], as the reduce kernel expects
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * LWT, x0 = tx * LWT;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // ---- staging: [pixel][8 channels] vectors -> 8 channel planes; every load of the thread is issued before the first LDS write (a loop of
  // load -> eight 2-byte writes was ten dependent memory round trips per workgroup) ----------------------------------------------
  constexpr int NX = (PW * PX + 511) / 512, ND = LWT * LWT / 512;
  u32x4 vx[NX], vd[ND];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int i = tid + j * 512;
    const int py = min(i / PX, PW - 1), px = i % PX;
    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
    const bool ok = i < PW * PX && px < PW && (unsigned)gy < (unsigned)Hh && (unsigned)gx < (unsigned)W;
    const int cy = min(max(gy, 0), Hh - 1), cx = min(max(gx, 0), W - 1);
    vx[j] = *reinterpret_cast<const u32x4*>(x + (((long long)b * Hh + cy) * W + cx) * x_ld + cv * 8);
    if (!ok) vx[j] = zero4;
  }
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int i = tid + j * 512;
    const int py = i / LWT, px = i - py * LWT;
    const int gy = y0 + py, gx = x0 + px;
    const bool ok = gy < Hh && gx < W;
    const int cy = min(gy, Hh - 1), cx = min(gx, W - 1);
    vd[j] = *reinterpret_cast<const u32x4*>(dy + (((long long)b * Hh + cy) * W + cx) * dy_ld + cv * 8);
    if (!ok) vd[j] = zero4;
  }
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int i = tid + j * 512;
    if (i < PW * PX) {
#pragma unroll
      for (int c = 0; c < 8; ++c) sx[c * XPLANE + i] = (unsigned short)(vx[j][c >> 1] >> ((c & 1) * 16));
    }
  }
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int i = tid + j * 512;
    const int py = i / LWT, px = i - py * LWT;
#pragma unroll
    for (int c = 0; c < 8; ++c) sd[c * DPLANE + py * PD + px] = (unsigned short)(vd[j][c >> 1] >> ((c & 1) * 16));
  }
  __syncthreads();
  // ---- one channel per wave ----------------------------------------------------------------------------------------------------
  const int lane = tid & 63, ch = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const unsigned int* xp = reinterpret_cast<const unsigned int*>(sx + ch * XPLANE);      // dword view of the channel's x plane
  const unsigned short* dp = sd + ch * DPLANE;
  f32x4 acc[NTL][NTL];
#pragma unroll
  for (int a = 0; a < NTL; ++a)
#pragma unroll
    for (int c = 0; c < NTL; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_dw[NTL];
  unsigned a_sh[NTL];
#pragma unroll
  for (int a = 0; a < NTL; ++a) {
    const int o = 8 * q + 16 * a + m;                      // first element of the lane's A fragment inside the row
    a_dw[a] = o >> 1;
    a_sh[a] = (unsigned)(o & 1) * 16u;
  }
#pragma unroll 2
  for (int r = 0; r < PW; ++r) {
    s16x8 af[NTL];
#pragma unroll
    for (int a = 0; a < NTL; ++a) {
      const unsigned int* src = xp + r * (PX / 2) + a_dw[a];
      unsigned int w[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) w[i] = src[i];
      u32x4 f;
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], a_sh[a]);
      af[a] = __builtin_bit_cast(s16x8, f);
    }
#pragma unroll
    for (int c = 0; c < NTL; ++c) {
      // dY rows r - ky, ky = 16 c + 0..15: the tile meets [0, 32) only for r in [16 c, 16 c + 46]  (wave-uniform)
      if (r < 16 * c || r > 16 * c + LWT + 14) continue;
      const int row = r - (16 * c + m);
      const bool in = (unsigned)row < (unsigned)LWT;
      const uint2* gp = reinterpret_cast<const uint2*>(dp + (in ? row : 0) * PD + 8 * q);      // two 8-byte reads (72-byte rows)
      const uint2 g0 = gp[0], g1 = gp[1];
      u32x4 g = {g0.x, g0.y, g1.x, g1.y};
      if (!in) g = zero4;
      const s16x8 bf = __builtin_bit_cast(s16x8, g);
#pragma unroll
      for (int a = 0; a < NTL; ++a) acc[a][c] = mfma16<H>(af[a], bf, acc[a][c]);      // D[kx = 16 a + 4 q + i][ky = 16 c + m]
    }
  }
  float* dst = partial + pidx * K * K * 8;
#pragma unroll
  for (int a = 0; a < NTL; ++a)
#pragma unroll
    for (int c = 0; c < NTL; ++c) {
      const int ky = 16 * c + m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kx = 16 * a + 4 * q + i;
        if (ky < K && kx < K) dst[(ky * K + kx) * 8 + ch] = acc[a][c][i];
      }
    }
}

// dw[c][tap] = beta * dw + sum over tiles (b, ty, tx) of partial[tile][cv][tap][e]; 64 outputs per workgroup, 4 lanes over the
// tiles with 4 loads in flight each (a chain of hundreds of dependent adds per output otherwise), combined in lane order.
__global__ __launch_bounds__(256) void dwlarge_wgrad_reduce_kernel(const float* __restrict__ partial, int ntiles, int CV, int VE, int KK,
                                                                   float* __restrict__ dw, float beta) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;     // over C * KK, laid out [c][tap]
  const int C = CV * VE;
  const bool ok = i < C * KK;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ok) {
    const int c = i / KK, tap = i - c * KK;
    const int cv = c / VE, e = c - cv * VE;
    const long long off = ((long long)cv * KK + tap) * VE + e, ts = (long long)CV * KK * VE;
    int t = sl;
    for (; t + 12 < ntiles; t += 16) {
      s0 += partial[t * ts + off]; s1 += partial[(t + 4) * ts + off]; s2 += partial[(t + 8) * ts + off]; s3 += partial[(t + 12) * ts + off];
    }
    for (; t < ntiles; t += 4) s0 += partial[t * ts + off];
  }
  red[sl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && ok) {
    const float s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    dw[i] = beta != 0.f ? beta * dw[i] + s : s;
  }
}

template <typename T, int K>
hipError_t launch_dwl_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* partial, int B, int H, int W, int C, hipStream_t s) {
  constexpr int VE = Vec<T>::N, PW = LWT + K - 1;
  constexpr size_t lds = (size_t)(PW * PW + LWT * LWT) * VE * sizeof(T);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)dwlarge_wgrad_kernel<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const long long blocks = (long long)B * cdiv(H, LWT) * cdiv(W, LWT) * (C / VE);
  if constexpr (sizeof(T) == 2) {
    if (!g_dwl_wgrad_valu) {
      constexpr size_t lds_m = (size_t)8 * ((LWT + K - 1) * 64 + LWT * 36) * 2;
      static bool attr_m = false;
      if (!attr_m) {
        hipError_t e = hipFuncSetAttribute((const void*)dwlarge_wgrad_mfma_kernel<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_m = true;
      }
      const long long blocks_m = (long long)cdiv(B * cdiv(H, LWT) * cdiv(W, LWT), 8) * 8 * (C / VE);      // tiles padded to whole XCD rounds
      hipLaunchKernelGGL((dwlarge_wgrad_mfma_kernel<T, K>), dim3((unsigned)blocks_m), dim3(512), lds_m, s, (const T*)x, x_ld, (const T*)dy, dy_ld,
                         partial, B, H, W, C);
      return hipSuccess;
    }
  }
  if constexpr (K == 31) {
    static bool attr31 = false;
    if (!attr31) {
      hipError_t e = hipFuncSetAttribute((const void*)dwlarge_wgrad31_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attr31 = true;
    }
    hipLaunchKernelGGL((dwlarge_wgrad31_kernel<T>), dim3((unsigned)blocks), dim3(256), lds, s, (const T*)x, x_ld, (const T*)dy, dy_ld, partial, B,
                       H, W, C);
  } else {
    hipLaunchKernelGGL((dwlarge_wgrad_kernel<T, K>), dim3((unsigned)blocks), dim3(256), lds, s, (const T*)x, x_ld, (const T*)dy, dy_ld, partial, B,
                       H, W, C);
  }
  return hipSuccess;
}
}  // namespace

void cfp_dwl_wgrad_debug_set(int value) { g_dwl_wgrad_valu = value; }

extern "C" size_t cfp_dwconv_large_wgrad_ws_bytes(int B, int H, int W, int C, int k) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0) return 0;
  return (size_t)B * cdiv(H, LWT) * cdiv(W, LWT) * C * k * k * sizeof(float);
}

extern "C" int cfp_dwconv_large_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int k,
                                      float beta, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream) {
  CFP_REQUIRE(x && dy && dw && ws && aligned16(x) && aligned16(dy), CFP_EINVAL, "cfp_dwconv_large_wgrad: bad pointer");
  T2_COMMON("cfp_dwconv_large_wgrad");
  CFP_REQUIRE(k == 7 || k == 15 || k == 31, CFP_ESHAPE, "cfp_dwconv_large_wgrad: k must be 7, 15 or 31");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % ve == 0 && x_ld % ve == 0 && dy_ld % ve == 0 && x_ld >= C && dy_ld >= C, CFP_ESHAPE,
              "cfp_dwconv_large_wgrad: bad shape");
  CFP_REQUIRE(ws_bytes >= cfp_dwconv_large_wgrad_ws_bytes(B, H, W, C, k), CFP_EINVAL, "cfp_dwconv_large_wgrad: workspace too small");
  float* partial = reinterpret_cast<float*>(ws);
  hipError_t e;
#define LK(T) (k == 31 ? launch_dwl_wgrad<T, 31>(x, x_ld, dy, dy_ld, partial, B, H, W, C, s) \
             : k == 15 ? launch_dwl_wgrad<T, 15>(x, x_ld, dy, dy_ld, partial, B, H, W, C, s) \
                       : launch_dwl_wgrad<T, 7>(x, x_ld, dy, dy_ld, partial, B, H, W, C, s))
  if (dtype == CFP_BF16) e = LK(bf16_t); else if (dtype == CFP_F16) e = LK(f16_t); else e = LK(float);
#undef LK
  if (e != hipSuccess) { cfp_set_error(std::string("cfp_dwconv_large_wgrad: ") + hipGetErrorString(e)); return CFP_EHIP; }
  const int ntiles = B * cdiv(H, LWT) * cdiv(W, LWT);
  hipLaunchKernelGGL(dwlarge_wgrad_reduce_kernel, dim3(cdiv((long long)C * k * k, 64)), dim3(256), 0, s, partial, ntiles, C / ve, ve, k * k, dw,
                     beta);
  return cfp_check_launch("cfp_dwconv_large_wgrad");
}

// ---- row normalisation of the bin widths: out = x / sum_c x  (decoder.py:36) and its backward -----------------------------
namespace {
__global__ void row_normalize_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int rows, int C) {
  // one wave per row; dy == NULL: forward.  backward: out = (dy - sum_c(dy * y)) / s with y = x / s
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[(long long)row * C + c];
  s = wave_sum(s);
  if (!dy) {
    for (int c = lane; c < C; c += 64) out[(long long)row * C + c] = x[(long long)row * C + c] / s;
    return;
  }
  float t = 0.f;
  for (int c = lane; c < C; c += 64) t = fmaf(dy[(long long)row * C + c], x[(long long)row * C + c] / s, t);
  t = wave_sum(t);
  for (int c = lane; c < C; c += 64) out[(long long)row * C + c] = (dy[(long long)row * C + c] - t) / s;
}
}  // namespace

extern "C" int cfp_row_normalize(const float* x, const float* dy, float* out, int rows, int C, cfp_stream_t stream) {
  CFP_REQUIRE(x && out && rows > 0 && C > 0, CFP_EINVAL, "cfp_row_normalize: bad argument");
  hipLaunchKernelGGL(row_normalize_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, dy, out, rows, C);
  return cfp_check_launch("cfp_row_normalize");
}
