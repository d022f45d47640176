// Small backward kernels of the training step: gradient accumulation, positional-encoding gradient, squeeze-excite
// backward pieces, depthwise 3x3 backward.  All HBM-bound sweeps over 16-byte NHWC vectors; reductions are per-split f32
// partials combined in a fixed order.
//
//   reference: autograd of the corresponding forward ops in model.train() (train.py:119-131) --
//   skip connections / torch.cat (decoder.py:57, transformer.py:66,238), `x + PE` (fusion.py:87-97), timm SqueezeExcite
//   and the depthwise conv_dw of InvertedResidual (encoder.py:57-69).
#include "common.h"

namespace {

inline int ew_grid2(long long total) { long long b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// out = a * x + b * y   (y may be null: out = a * x)
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ y, int y_ld, float a, float b,
                                                    T* __restrict__ out, int out_ld, long long rows, int C, FastDiv fcv) {
  constexpr int VE = Vec<T>::N;
  const unsigned total = (unsigned)(rows * fcv.d);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned ru, cvu;
    fd_rowcol(i, fcv, ru, cvu);
    const long long r = ru;
    const int c = (int)cvu * VE;
    float v[VE], w[VE];
    Vec<T>::load(x + r * x_ld + c, v);
    if (y) {
      Vec<T>::load(y + r * y_ld + c, w);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = a * v[e] + b * w[e];
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] *= a;
    }
    Vec<T>::store(out + r * out_ld + c, v);
  }
}

// dtable[(oy + y) * Wt + ox + x][c] (+)= sum_b dx[b][y][x][c]   -- gradient of `x + PE[oy:oy+H, ox:ox+W]` (f32 table)
// 2^ibits (pixel, channel vector) items per workgroup, the other threads are lanes over the batch: the second table of a fusion
// block is [16, D] against B * Z = 576 "images", one thread per item would add 576 rows alone (150 us); lanes meet in LDS in
// lane order.
template <typename T>
__global__ __launch_bounds__(256) void rowtable_grad_kernel(const T* __restrict__ dx, int ld, float* __restrict__ dtable, int B, int H, int W,
                                                            int C, int Wt, int oy, int ox, float beta, const int* __restrict__ dev_off, int Ht,
                                                            int ibits) {
  constexpr int VE = Vec<T>::N;
  __shared__ float red[256][8];
  const int CV = C / VE;
  if (dev_off) {
    oy = min(max(dev_off[0], 0), Ht - H);
    ox = min(max(dev_off[1], 0), Wt - W);
  }
  const int items = 1 << ibits, lanes = 256 >> ibits;
  const int il = threadIdx.x & (items - 1), bl = threadIdx.x >> ibits;
  const long long total = (long long)H * W * CV;
  const long long i = (long long)blockIdx.x * items + il;
  const bool ok = i < total;
  const long long px = ok ? i / CV : 0;
  const int c = ok ? (int)(i - px * CV) * VE : 0;
  float s[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s[e] = 0.f;
  if (ok)
    for (int b = bl; b < B; b += lanes) {
      float v[VE];
      Vec<T>::load(dx + ((long long)b * H * W + px) * ld + c, v);
#pragma unroll
      for (int e = 0; e < VE; ++e) s[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < VE; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (bl == 0 && ok) {
    const int y = (int)(px / W), x = (int)(px - (long long)y * W);
    float* d = dtable + ((long long)(oy + y) * Wt + ox + x) * C + c;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float t = 0.f;
      for (int l = 0; l < lanes; ++l) t += red[(l << ibits) + il][e];
      d[e] = beta != 0.f ? beta * d[e] + t : t;
    }
  }
}

// out[b][c] = sum over the HW rows of image b of x * y  (squeeze-excite: d gate = sum dy * x).  One workgroup per
// (image, 2^colbits vector columns): 256 / 2^colbits row lanes, fixed-order combination.  Fewer columns per workgroup when the
// launch would otherwise be a few dozen workgroups (C = 1392 at batch 16: 96 with 32 columns, 352 with 8).
template <typename T>
__global__ __launch_bounds__(256) void channel_dot_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ y, int y_ld,
                                                          float* __restrict__ out, int HW, int C, int colbits) {
  constexpr int VE = Vec<T>::N;
  __shared__ float red[256][8];
  const int cols = 1 << colbits, lanes = 256 >> colbits;
  const int tid = threadIdx.x, cl = tid & (cols - 1), rl = tid >> colbits;
  const int c0 = (blockIdx.x * cols + cl) * VE;
  const int cc = c0 < C ? c0 : 0;
  const long long base = (long long)blockIdx.y * HW;
  float s[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s[e] = 0.f;
  constexpr int U = 4;                       // four rows in flight per thread (one per iteration was a chain of ~HW / 8 round trips)
  for (int r = rl; r < HW; r += lanes * U) {
    float a[U][VE], b[U][VE];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = min(r + u * lanes, HW - 1);
      Vec<T>::load(x + (base + rr) * x_ld + cc, a[u]);
      Vec<T>::load(y + (base + rr) * y_ld + cc, b[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + u * lanes >= HW) continue;
#pragma unroll
      for (int e = 0; e < VE; ++e) s[e] = fmaf(a[u][e], b[u][e], s[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) red[tid][e] = s[e];
  __syncthreads();
  for (int i = tid; i < cols * VE; i += 256) {
    const int c_l = i / VE, e = i - c_l * VE;
    const int c = (blockIdx.x * cols + c_l) * VE + e;
    if (c >= C) continue;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[(l << colbits) + c_l][e];
    out[(long long)blockIdx.y * C + c] = t;
  }
}

// dx = dy * gate[b][c] + add[b][c]      (squeeze-excite backward: gate path + the broadcast gradient of the mean)
// A thread keeps its channel vector and walks rows (no index division per element); gate / add are re-read only when the
// image changes.  2^colbits column lanes, the rest of the workgroup are row lanes; blockIdx.y = row chunk.
template <typename T>
__global__ __launch_bounds__(256) void bcast_fma_kernel(const T* __restrict__ dy, int dy_ld, const float* __restrict__ gate,
                                                        const float* __restrict__ add, T* __restrict__ dx, int dx_ld, FastDiv fhw, int C,
                                                        int rows, int rows_per_chunk, int colbits) {
  constexpr int VE = Vec<T>::N;
  const int cols = 1 << colbits, lanes = 256 >> colbits;
  const int cl = threadIdx.x & (cols - 1), rl = threadIdx.x >> colbits;
  const int c = (blockIdx.x * cols + cl) * VE;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float g[VE], a[VE];
  int bcur = -1;
  constexpr int U = 4;
  for (int r = r0 + rl; r < r1; r += lanes * U) {
    float v[U][VE];
#pragma unroll
    for (int u = 0; u < U; ++u) Vec<T>::load(dy + (long long)min(r + u * lanes, r1 - 1) * dy_ld + c, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + u * lanes;
      if (rr >= r1) continue;
      const int b = (int)fd_div((unsigned)rr, fhw);      // rows < 2^31
      if (b != bcur) {
        bcur = b;
#pragma unroll
        for (int e = 0; e < VE; ++e) { g[e] = gate[(long long)b * C + c + e]; a[e] = add ? add[(long long)b * C + c + e] : 0.f; }
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) v[u][e] = v[u][e] * g[e] + a[e];
      Vec<T>::store(dx + (long long)rr * dx_ld + c, v[u]);
    }
  }
}

// depthwise 3x3 data gradient, stride S = 1 / 2: dx[hi][wi][c] = sum_{kh,kw} dy[(hi + pt - kh)/S][(wi + pl - kw)/S][c] * w[kh][kw][c]
// (S is a template parameter: with a runtime stride the twelve `% stride` / `/ stride` tests per output vector were ~800 VALU
// instructions of signed division against 72 FMAs)
template <typename T, int S>
__global__ __launch_bounds__(256) void dw3x3_dgrad_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ w, T* __restrict__ dx,
                                                          int dx_ld, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int Ho,
                                                          int Wo, int accumulate, FastDiv fcv, FastDiv fw, FastDiv fh) {
  constexpr int VE = Vec<T>::N;
  const unsigned total = (unsigned)B * H * W * fcv.d;      // < 2^31 (host check): 32-bit indices, magic-number divisions
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned pxu, cvu, tu, wiu, bu, hiu;
    fd_rowcol(i, fcv, pxu, cvu);
    fd_rowcol(pxu, fw, tu, wiu);
    fd_rowcol(tu, fh, bu, hiu);
    const long long px = pxu;
    const int c = (int)cvu * VE, wi = (int)wiu, hi = (int)hiu, b = (int)bu;
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hn = hi + pad_t - kh;
      if (hn < 0 || (S == 2 && (hn & 1))) continue;
      const int hq = S == 2 ? hn >> 1 : hn;
      if (hq >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wn = wi + pad_l - kw;
        if (wn < 0 || (S == 2 && (wn & 1))) continue;
        const int wq = S == 2 ? wn >> 1 : wn;
        if (wq >= Wo) continue;
        float g[VE], ww[VE];
        Vec<T>::load(dy + (((long long)b * Ho + hq) * Wo + wq) * dy_ld + c, g);
        Vec<T>::load(w + (long long)(kh * 3 + kw) * C + c, ww);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = fmaf(g[e], ww[e], acc[e]);
      }
    }
    if (accumulate) {
      float o[VE];
      Vec<T>::load(dx + px * dx_ld + c, o);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += o[e];
    }
    Vec<T>::store(dx + px * dx_ld + c, acc);
  }
}

// depthwise 3x3 weight gradient: partial[split][tap][c] = sum over the split's output pixels of dy * x(window tap)
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dy, int dy_ld,
                                                          float* __restrict__ partial, int B, int H, int W, int C, int stride, int pad_t,
                                                          int pad_l, int Ho, int Wo, long long rows_per_split, FastDiv fwo, FastDiv fho) {
  constexpr int VE = Vec<T>::N;
  __shared__ float red[8][32 * 8];
  const int tid = threadIdx.x, cl = tid & 31, rl = tid >> 5;
  const int c0 = (blockIdx.x * 32 + cl) * VE;
  const int cc = c0 < C ? c0 : 0;
  const long long M = (long long)B * Ho * Wo;
  const long long r0 = (long long)blockIdx.y * rows_per_split, r1 = min(M, r0 + rows_per_split);
  float s[9][VE];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VE; ++e) s[t][e] = 0.f;
  for (long long m = r0 + rl; m < r1; m += 8) {
    unsigned ttu, wou, bu, hou;                 // M < 2^31 (host check)
    fd_rowcol((unsigned)m, fwo, ttu, wou);
    fd_rowcol(ttu, fho, bu, hou);
    const int wo = (int)wou, ho = (int)hou, b = (int)bu;
    float g[VE];
    Vec<T>::load(dy + m * dy_ld + cc, g);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * stride - pad_t + kh;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * stride - pad_l + kw;
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const int hic = min(max(hi, 0), H - 1), wic = min(max(wi, 0), W - 1);
        float v[VE];
        Vec<T>::load(x + (((long long)b * H + hic) * W + wic) * x_ld + cc, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) s[kh * 3 + kw][e] = fmaf(g[e], ok ? v[e] : 0.f, s[kh * 3 + kw][e]);
      }
    }
  }
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VE; ++e) red[rl][cl * 8 + e] = s[t][e];
    __syncthreads();
    for (int i = tid; i < 32 * VE; i += 256) {
      const int c_l = i / VE, e = i - c_l * VE;
      const int c = (blockIdx.x * 32 + c_l) * VE + e;
      if (c >= C) continue;
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < 8; ++l) a += red[l][c_l * 8 + e];
      partial[((long long)blockIdx.y * 9 + t) * C + c] = a;
    }
  }
}

// out = beta * out + sum over the splits, fixed order: 32 consecutive outputs per workgroup, 8 lanes over the splits with 4 loads
// in flight each (hundreds of splits of a few thousand outputs: one thread per output would walk them alone), lanes meet in LDS.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nsplit, long long n, float* __restrict__ out,
                                                           float beta) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (long long i0 = (long long)blockIdx.x * 32; i0 < n; i0 += (long long)gridDim.x * 32) {
    const long long i = i0 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int j = sl;
      for (; j + 24 < nsplit; j += 32) {
        s0 += partial[(long long)j * n + i]; s1 += partial[(long long)(j + 8) * n + i];
        s2 += partial[(long long)(j + 16) * n + i]; s3 += partial[(long long)(j + 24) * n + i];
      }
      for (; j < nsplit; j += 8) s0 += partial[(long long)j * n + i];
    }
    red[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < n) {
      float s = 0.f;
#pragma unroll
      for (int l = 0; l < 8; ++l) s += red[l][e];
      out[i] = beta != 0.f ? beta * out[i] + s : s;
    }
    __syncthreads();
  }
}

inline int dw_wgrad_splits(long long M, int C, int ve) {
  const int colblk = cdiv(C, 32 * ve);
  long long ns = 1024 / colblk;
  const long long mx = (M + 31) / 32;                        // at least 32 rows (4 per row lane) per split
  if (ns > mx) ns = mx;
  if (ns > 2048) ns = 2048;
  if (ns < 1) ns = 1;
  return (int)ns;
}

}  // namespace

#define TM_COMMON(name)                                                                      \
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, name ": bad dtype");                              \
  const int ve = vec_elems(dtype);                                                           \
  hipStream_t s = reinterpret_cast<hipStream_t>(stream)

extern "C" int cfp_axpby(const void* x, int x_ld, const void* y, int y_ld, float a, float b, void* out, int out_ld, long long rows, int C,
                         int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(x && out && aligned16(x) && aligned16(y) && aligned16(out), CFP_EINVAL, "cfp_axpby: bad pointer");
  TM_COMMON("cfp_axpby");
  CFP_REQUIRE(rows > 0 && C > 0 && C % ve == 0 && x_ld % ve == 0 && out_ld % ve == 0 && x_ld >= C && out_ld >= C && (!y || (y_ld % ve == 0 && y_ld >= C)),
              CFP_ESHAPE, "cfp_axpby: bad shape");
  const dim3 grid(ew_grid2(rows * (C / ve)));
  CFP_REQUIRE(rows * (C / ve) < (1ll << 31), CFP_ESHAPE, "cfp_axpby: too many elements");
  const FastDiv fcv = make_fastdiv((unsigned)(C / ve));
#define L(T) hipLaunchKernelGGL(axpby_kernel<T>, grid, dim3(256), 0, s, (const T*)x, x_ld, (const T*)y, y_ld, a, b, (T*)out, out_ld, rows, C, fcv)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  return cfp_check_launch("cfp_axpby");
}

static int rowtable_grad_impl(const void* dx, int ld, float* dtable, int B, int H, int W, int C, int Wt, int oy, int ox, float beta,
                              const int* dev_off, int Ht, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dx && dtable && aligned16(dx), CFP_EINVAL, "cfp_rowtable_grad: bad pointer");
  TM_COMMON("cfp_rowtable_grad");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % ve == 0 && ld % ve == 0 && ld >= C && Wt >= ox + W && oy >= 0 && ox >= 0 &&
                  (!dev_off || Ht >= H), CFP_ESHAPE, "cfp_rowtable_grad: bad shape");
  const long long total = (long long)H * W * (C / ve);
  int ibits = 8;                                             // fewer items per workgroup (more batch lanes) while the launch is small
  while (ibits > 3 && (total >> ibits) < 512 && (256 >> (ibits - 1)) <= B) --ibits;
  CFP_REQUIRE(((total + (1 << ibits) - 1) >> ibits) < (1ll << 31), CFP_ESHAPE, "cfp_rowtable_grad: table too large");
  const dim3 grid((unsigned)cdiv(total, 1 << ibits));
#define L(T) hipLaunchKernelGGL(rowtable_grad_kernel<T>, grid, dim3(256), 0, s, (const T*)dx, ld, dtable, B, H, W, C, Wt, oy, ox, beta, dev_off, Ht, ibits)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  return cfp_check_launch("cfp_rowtable_grad");
}

extern "C" int cfp_rowtable_grad(const void* dx, int ld, float* dtable, int B, int H, int W, int C, int Wt, int oy, int ox, float beta,
                                 int dtype, cfp_stream_t stream) {
  return rowtable_grad_impl(dx, ld, dtable, B, H, W, C, Wt, oy, ox, beta, nullptr, 0, dtype, stream);
}

extern "C" int cfp_rowtable_grad_dev(const void* dx, int ld, float* dtable, int B, int H, int W, int C, int Ht, int Wt, const int* oyox,
                                     float beta, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(oyox, CFP_EINVAL, "cfp_rowtable_grad_dev: null offset pointer");
  return rowtable_grad_impl(dx, ld, dtable, B, H, W, C, Wt, 0, 0, beta, oyox, Ht, dtype, stream);
}

extern "C" int cfp_channel_dot(const void* x, int x_ld, const void* y, int y_ld, float* out, int B, int HW, int C, int dtype,
                               cfp_stream_t stream) {
  CFP_REQUIRE(x && y && out && aligned16(x) && aligned16(y), CFP_EINVAL, "cfp_channel_dot: bad pointer");
  TM_COMMON("cfp_channel_dot");
  CFP_REQUIRE(B > 0 && B <= 65535 && HW > 0 && C > 0 && C % ve == 0 && x_ld % ve == 0 && y_ld % ve == 0 && x_ld >= C && y_ld >= C, CFP_ESHAPE,
              "cfp_channel_dot: bad shape");
  int colbits = 5;
  while (colbits > 2 && (long long)cdiv(C, ve << colbits) * B < 512 && (256 >> colbits) * 4 <= HW) --colbits;
  const dim3 grid(cdiv(C, ve << colbits), B);
#define L(T) hipLaunchKernelGGL(channel_dot_kernel<T>, grid, dim3(256), 0, s, (const T*)x, x_ld, (const T*)y, y_ld, out, HW, C, colbits)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  return cfp_check_launch("cfp_channel_dot");
}

extern "C" int cfp_bcast_fma(const void* dy, int dy_ld, const float* gate, const float* add, void* dx, int dx_ld, int B, int HW, int C,
                             int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dy && gate && dx && aligned16(dy) && aligned16(dx), CFP_EINVAL, "cfp_bcast_fma: bad pointer");
  TM_COMMON("cfp_bcast_fma");
  CFP_REQUIRE(B > 0 && HW > 0 && C > 0 && C % ve == 0 && dy_ld % ve == 0 && dx_ld % ve == 0 && dy_ld >= C && dx_ld >= C, CFP_ESHAPE,
              "cfp_bcast_fma: bad shape");
  CFP_REQUIRE((long long)B * HW < (1ll << 31), CFP_ESHAPE, "cfp_bcast_fma: too many rows");
  const int rows = B * HW;
  const int cv = C / ve;
  int colbits = 0;
  while ((1 << colbits) < cv && colbits < 5) ++colbits;                 // 1 .. 32 column lanes
  const int lanes = 256 >> colbits;
  const unsigned gx = (unsigned)cdiv(cv, 1 << colbits);
  long long ns = std::max<long long>(1, 2048 / gx);                      // ~8 workgroups per CU over the launch
  ns = std::min<long long>(ns, cdiv(rows, 4 * lanes));
  ns = std::max<long long>(1, std::min<long long>(ns, 65535));
  const int rpc = cdiv(rows, (int)ns);
  const dim3 grid(gx, cdiv(rows, rpc));
#define L(T) hipLaunchKernelGGL(bcast_fma_kernel<T>, grid, dim3(256), 0, s, (const T*)dy, dy_ld, gate, add, (T*)dx, dx_ld, make_fastdiv((unsigned)HW), C, rows, rpc, colbits)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  return cfp_check_launch("cfp_bcast_fma");
}

extern "C" int cfp_dwconv3x3_dgrad(const void* dy, int dy_ld, const void* w, void* dx, int dx_ld, int B, int H, int W, int C, int stride,
                                   int pad_t, int pad_l, int Ho, int Wo, int accumulate, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dy && w && dx && aligned16(dy) && aligned16(w) && aligned16(dx), CFP_EINVAL, "cfp_dwconv3x3_dgrad: bad pointer");
  TM_COMMON("cfp_dwconv3x3_dgrad");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % ve == 0 && dy_ld % ve == 0 && dx_ld % ve == 0 && dy_ld >= C &&
                  dx_ld >= C && (stride == 1 || stride == 2), CFP_ESHAPE, "cfp_dwconv3x3_dgrad: bad shape");
  CFP_REQUIRE((long long)B * H * W * (C / ve) < (1ll << 31), CFP_ESHAPE, "cfp_dwconv3x3_dgrad: too many elements");
  const dim3 grid(ew_grid2((long long)B * H * W * (C / ve)));
  const FastDiv fcv = make_fastdiv((unsigned)(C / ve)), fw = make_fastdiv((unsigned)W), fh = make_fastdiv((unsigned)H);
#define L2(T, S) hipLaunchKernelGGL((dw3x3_dgrad_kernel<T, S>), grid, dim3(256), 0, s, (const T*)dy, dy_ld, (const T*)w, (T*)dx, dx_ld, B, H, W, C, stride, \
                                    pad_t, pad_l, Ho, Wo, accumulate, fcv, fw, fh)
#define L(T) do { if (stride == 1) L2(T, 1); else L2(T, 2); } while (0)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
#undef L2
  return cfp_check_launch("cfp_dwconv3x3_dgrad");
}

extern "C" size_t cfp_dwconv3x3_wgrad_ws_bytes(int C) { return C > 0 ? (size_t)2048 * 9 * C * sizeof(float) : 0; }

static int dw3x3_wgrad_impl(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int stride,
                            int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws, size_t ws_bytes, cfp_wgrad_job* job,
                            cfp_stream_t stream) {
  CFP_REQUIRE(x && dy && dw && ws && aligned16(x) && aligned16(dy), CFP_EINVAL, "cfp_dwconv3x3_wgrad: bad pointer");
  TM_COMMON("cfp_dwconv3x3_wgrad");
  CFP_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % ve == 0 && x_ld % ve == 0 && dy_ld % ve == 0 && x_ld >= C &&
                  dy_ld >= C && (stride == 1 || stride == 2), CFP_ESHAPE, "cfp_dwconv3x3_wgrad: bad shape");
  CFP_REQUIRE(ws_bytes >= cfp_dwconv3x3_wgrad_ws_bytes(C), CFP_EINVAL, "cfp_dwconv3x3_wgrad: workspace too small");
  const long long M = (long long)B * Ho * Wo;
  const int ns = dw_wgrad_splits(M, C, ve);
  const long long rps = (M + ns - 1) / ns;
  const int nsplit = (int)((M + rps - 1) / rps);
  float* partial = reinterpret_cast<float*>(ws);
  const dim3 grid(cdiv(C, 32 * ve), nsplit);
  CFP_REQUIRE(M < (1ll << 31), CFP_ESHAPE, "cfp_dwconv3x3_wgrad: too many output pixels");
  const FastDiv fwo = make_fastdiv((unsigned)Wo), fho = make_fastdiv((unsigned)Ho);
#define L(T) hipLaunchKernelGGL(dw3x3_wgrad_kernel<T>, grid, dim3(256), 0, s, (const T*)x, x_ld, (const T*)dy, dy_ld, partial, B, H, W, C, stride, \
                                pad_t, pad_l, Ho, Wo, rps, fwo, fho)
  if (dtype == CFP_BF16) L(bf16_t); else if (dtype == CFP_F16) L(f16_t); else L(float);
#undef L
  const long long n = 9ll * C;
  if (job) {      // the sum over the splits joins the batched reduction of the dense weight gradients (cfp_wgrad_reduce_jobs)
    job->slabs = partial; job->dw = dw; job->db = nullptr; job->n = n; job->n_dw = n; job->nsplit = nsplit;
    job->ew = nsplit < 8 ? 256 : nsplit < 32 ? 64 : 32; job->beta = beta; job->beta_b = 0.f;
  } else {
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)std::min<long long>(4096, cdiv(n, 32))), dim3(256), 0, s, partial, nsplit, n, dw, beta);
  }
  return cfp_check_launch("cfp_dwconv3x3_wgrad");
}

extern "C" int cfp_dwconv3x3_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int stride,
                                   int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws, size_t ws_bytes,
                                   cfp_stream_t stream) {
  return dw3x3_wgrad_impl(x, x_ld, dy, dy_ld, dw, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, beta, dtype, ws, ws_bytes, nullptr, stream);
}

extern "C" int cfp_dwconv3x3_wgrad_deferred(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int stride,
                                            int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws, size_t ws_bytes,
                                            cfp_wgrad_job* job, cfp_stream_t stream) {
  CFP_REQUIRE(job, CFP_EINVAL, "cfp_dwconv3x3_wgrad_deferred: null job");
  return dw3x3_wgrad_impl(x, x_ld, dy, dy_ld, dw, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, beta, dtype, ws, ws_bytes, job, stream);
}
