// Linear attention (feature map elu(x)+1) -- there is no QK^T matrix and no softmax here:
//   reduce:  KV[g,h] = sum_s K'_s^T (V_s / S),  Ksum[g,h] = sum_s K'_s      (d x d and d per head)
//   apply :  out_q   = (Q'_q KV) / (Q'_q . Ksum + eps) * S
// d = D / heads is 4..32, so the d x d accumulator of one (group, head) lives in the registers of
// one wave: lane = (key lane, row chunk); the cross-lane sum over key lanes is a butterfly of
// wavefront shuffles.  Which keys form a group (a ToF zone's 16 samples, a ws x ws window with
// its zero padding, the sub-sampled global keys, the inside-zone rectangle) is pure addressing.
#include "common.h"

namespace {

struct KvP {
  const void* k; const void* v;
  float* kv; float* ksum; float* ws;
  int k_ld, v_ld;
  int NB, Hk, Wk, th, tw, gy, gx;
  int cy0, cy1, cx0, cx1;
  int count_pad, heads, d, nsplit;
  float inv_len;
};

// IC rows of the d x d accumulator per lane; TPK = d / IC lanes per key; KL = 64 / TPK key lanes.
template <int D> struct KvCfg;
// The cross-lane sum over the KL key lanes is a butterfly of log2(KL) steps x (IC * d + IC) `ds_bpermute`s: with a lane holding all d
// rows (IC = d at d = 8) that was 432 of the kernel's 946 vector instructions.  Two rows per lane (TPK = d / 2 lanes share a key and
// load its V row together) leave 16 / 8 key lanes: 72 / 102 permutes at d = 8 / 16, the same two to four load rounds per lane.
template <> struct KvCfg<4>  { static constexpr int IC = 1; };
template <> struct KvCfg<8>  { static constexpr int IC = 2; };
template <> struct KvCfg<16> { static constexpr int IC = 2; };
template <> struct KvCfg<32> { static constexpr int IC = 2; };

template <typename T, int D>
__global__ __launch_bounds__(64) void kv_reduce_kernel(KvP p) {
  constexpr int IC = KvCfg<D>::IC;
  constexpr int TPK = D / IC;
  constexpr int KL = 64 / TPK;
  const T* __restrict__ K = reinterpret_cast<const T*>(p.k);
  const T* __restrict__ V = reinterpret_cast<const T*>(p.v);
  const int g = blockIdx.x, h = blockIdx.y, sp = blockIdx.z;
  const int lane = threadIdx.x;
  const int ip = lane % TPK, kl = lane / TPK;

  const int gpb = p.gy * p.gx;
  const int b = g / gpb, gi = g % gpb;
  const int ty = gi / p.gx, tx = gi % p.gx;
  // key rectangle of this group: tile, clipped to the grid and to the clip rectangle
  const int y0 = max(max(ty * p.th, p.cy0), 0), y1 = min(min((ty + 1) * p.th, p.cy1), p.Hk);
  const int x0 = max(max(tx * p.tw, p.cx0), 0), x1 = min(min((tx + 1) * p.tw, p.cx1), p.Wk);
  const int rh = max(y1 - y0, 0), rw = max(x1 - x0, 0);
  const int S = rh * rw;
  const int per = (S + p.nsplit - 1) / p.nsplit;
  const int s_begin = sp * per, s_end = min(S, s_begin + per);

  float acc[IC][D];
  float ks[IC];
#pragma unroll
  for (int i = 0; i < IC; ++i) {
    ks[i] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) acc[i][j] = 0.f;
  }
  // KU keys per iteration, their loads issued together: the loop is a chain of memory round trips (one per key before), not FMAs
  constexpr int KU = 4;
  // key s of the rectangle sits at (y0 + s / rw, x0 + s % rw): the lane's position advances by KL keys per load, so the division happens
  // once (the quotient / remainder of the step and of the first and last key) and every step after that is an add and one wrap -- the
  // per-key signed division was ~35 VALU instructions against 2 D = 8 .. 64 FMAs
  const int rwd = max(rw, 1);
  const int step_q = KL / rwd, step_r = KL - step_q * rwd;
  const int s_first = min(s_begin + kl, max(s_end - 1, 0));
  int cy = s_first / rwd, cx = s_first - cy * rwd;                      // (row, column) of key s0 + u * KL, before clamping
  const int ly = max(s_end - 1, 0) / rwd, lx = max(s_end - 1, 0) - ly * rwd;      // of the last key (what the clamped tail loads read)
  for (int s0 = s_begin + kl; s0 < s_end; s0 += KL * KU) {
    float kf[KU][IC], vf[KU][D];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const bool in = s0 + u * KL < s_end;
      const int yy = y0 + (in ? cy : ly), xx = x0 + (in ? cx : lx);
      cx += step_r; cy += step_q;
      if (cx >= rwd) { cx -= rwd; ++cy; }
      const long long row = ((long long)b * p.Hk + yy) * p.Wk + xx;
      const T* kp = K + row * p.k_ld + h * D + ip * IC;
      const T* vp = V + row * p.v_ld + h * D;
#pragma unroll
      for (int i = 0; i < IC; ++i) kf[u][i] = to_f32<T>(kp[i]);
#pragma unroll
      for (int j = 0; j < D; ++j) vf[u][j] = to_f32<T>(vp[j]);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (s0 + u * KL >= s_end) continue;              // same order of accumulation per lane as the one-key loop
#pragma unroll
      for (int j = 0; j < D; ++j) vf[u][j] *= p.inv_len;
#pragma unroll
      for (int i = 0; i < IC; ++i) {
        const float kk = elu1(kf[u][i]);
        ks[i] += kk;
#pragma unroll
        for (int j = 0; j < D; ++j) acc[i][j] = fmaf(kk, vf[u][j], acc[i][j]);
      }
    }
  }
  // butterfly over the key lanes (lanes with equal ip)
#pragma unroll
  for (int o = TPK; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < IC; ++i) {
      ks[i] += __shfl_xor(ks[i], o, 64);
#pragma unroll
      for (int j = 0; j < D; ++j) acc[i][j] += __shfl_xor(acc[i][j], o, 64);
    }
  }
  if (kl == 0) {
    // zero-padded window positions: K' = elu(0)+1 = 1, V = 0 (only the first split adds them)
    float padk = 0.f;
    if (p.count_pad && sp == 0) padk = (float)(p.th * p.tw - S);
    float* dkv; float* dks;
    if (p.nsplit == 1) {
      dkv = p.kv + ((long long)g * p.heads + h) * D * D;
      dks = p.ksum + ((long long)g * p.heads + h) * D;
    } else {
      float* base = p.ws + (((long long)g * p.heads + h) * p.nsplit + sp) * (D * D + D);
      dkv = base; dks = base + D * D;
    }
#pragma unroll
    for (int i = 0; i < IC; ++i) {
      dks[ip * IC + i] = ks[i] + padk;
#pragma unroll
      for (int j = 0; j < D; ++j) dkv[(ip * IC + i) * D + j] = acc[i][j];
    }
  }
}

__global__ __launch_bounds__(256) void kv_finalize_kernel(const float* __restrict__ ws, float* __restrict__ kv,
                                                          float* __restrict__ ksum, int nsplit, int d, long long items) {
  // one thread per element of [group*head][d*d + d]; the split partials are summed in split order, eight
  // independent loads in flight at a time (the serial chain of up to 64 dependent loads was the whole kernel)
  const int per = d * d + d;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items * per; i += (long long)gridDim.x * 256) {
    const long long gh = (long long)((unsigned long long)i / (unsigned)per);      // one division; the remainder by multiply-subtract
    const int e = (int)(i - gh * per);
    const float* base = ws + gh * nsplit * per + e;
    float s = 0.f;
    int j = 0;
    for (; j + 7 < nsplit; j += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = base[(long long)(j + u) * per];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; j < nsplit; ++j) s += base[(long long)j * per];
    if (e < d * d) kv[gh * d * d + e] = s; else ksum[gh * d + (e - d * d)] = s;
  }
}

struct ApP {
  const void* q; const float* kv; const float* ksum; void* out;
  int q_ld, out_ld, NB, Hq, Wq, qth, qtw, ggy, ggx;
  int ey0, ey1, ex0, ex1, heads;
  float v_length, eps;
  FastDiv fheads, fwq, fhq, fqth, fqtw;      // (token, head) index arithmetic without 64-bit divisions (five per element in the first version)
};

// JS lanes share one (token, head): lane js computes the D / JS outputs js * D / JS ... (and the normaliser, redundantly).  With one lane per
// (token, head) a d = 32 head is a chain of 1 024 dependent-latency FMAs fed by 256 sixteen-byte loads, and a single image has 4 800 such lanes --
// 19 workgroups, 24 us; eight lanes each shorten the chain eightfold and fill eight times the CUs.  Every output is the same sum in the same order.
template <typename T, int D, int JS>
__global__ __launch_bounds__(256) void attn_apply_kernel(ApP p) {
  constexpr int DJ = D / JS;
  const T* __restrict__ Q = reinterpret_cast<const T*>(p.q);
  T* __restrict__ O = reinterpret_cast<T*>(p.out);
  const unsigned total = (unsigned)p.NB * p.Hq * p.Wq * p.heads * JS;      // < 2^31: host check
  for (unsigned i0 = blockIdx.x * 256u + threadIdx.x; i0 < total; i0 += gridDim.x * 256u) {
    const unsigned i = i0 / JS;
    const int js = (int)(i0 - i * JS);
    unsigned toku, hu, tu, xu, bu, yu;
    fd_rowcol(i, p.fheads, toku, hu);
    fd_rowcol(toku, p.fwq, tu, xu);
    fd_rowcol(tu, p.fhq, bu, yu);
    const int h = (int)hu, x = (int)xu, y = (int)yu, b = (int)bu;
    const long long tok = toku;
    T* op = O + tok * p.out_ld + h * D + js * DJ;
    float o[DJ];
    if (y >= p.ey0 && y < p.ey1 && x >= p.ex0 && x < p.ex1) {
#pragma unroll
      for (int j = 0; j < DJ; ++j) o[j] = 0.f;
    } else {
      const long long g = ((long long)b * p.ggy + fd_div((unsigned)y, p.fqth)) * p.ggx + fd_div((unsigned)x, p.fqtw);
      const float* kv = p.kv + (g * p.heads + h) * D * D + js * DJ;
      const float* ks = p.ksum + (g * p.heads + h) * D;
      const T* qp = Q + tok * p.q_ld + h * D;
      float z = 0.f;
#pragma unroll
      for (int j = 0; j < DJ; ++j) o[j] = 0.f;
#pragma unroll
      for (int ii = 0; ii < D; ++ii) {
        const float qv = elu1(to_f32<T>(qp[ii]));
        z = fmaf(qv, ks[ii], z);
#pragma unroll
        for (int j = 0; j < DJ; ++j) o[j] = fmaf(qv, kv[ii * D + j], o[j]);
      }
      const float zi = 1.f / (z + p.eps);
#pragma unroll
      for (int j = 0; j < DJ; ++j) o[j] = o[j] * zi * p.v_length;
    }
#pragma unroll
    for (int j = 0; j < DJ; ++j) op[j] = from_f32<T>(o[j]);
  }
}

int g_ap_split_below = 65536;      // cfp_debug_set key 38: (token, head) pairs below which the outputs of a pair are split over several lanes
int g_kv_target_waves = 2048;      // cfp_debug_set key 19 (A/B)
int pick_nsplit(int S, long long groups_heads) {
  // enough waves to cover the chip, at least ~64 keys per wave
  long long want = (g_kv_target_waves + groups_heads - 1) / groups_heads;
  int by_keys = (S + 63) / 64;
  long long n = want < by_keys ? want : by_keys;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

}  // namespace

void cfp_attn_debug_set(int value) { g_kv_target_waves = value; }
void cfp_attn_apply_debug_set(int value) { g_ap_split_below = value; }

extern "C" size_t cfp_attn_kv_ws_floats(int NB, int Hk, int Wk, int th, int tw, int heads, int d) {
  if (NB <= 0 || Hk <= 0 || Wk <= 0 || th <= 0 || tw <= 0 || heads <= 0 || d <= 0) return 0;
  long long groups = (long long)NB * cdiv(Hk, th) * cdiv(Wk, tw);
  int S = (th < Hk ? th : Hk) * (tw < Wk ? tw : Wk);
  int ns = pick_nsplit(S, groups * heads);
  return (size_t)(groups * heads * ns * (d * d + d));
}

extern "C" int cfp_attn_kv_reduce(const void* k, int k_ld, const void* v, int v_ld, float* kv, float* ksum, float* ws,
                                  int NB, int Hk, int Wk, int th, int tw, int cy0, int cy1, int cx0, int cx1,
                                  int count_pad, float v_length, int heads, int d, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_attn_kv_reduce: bad dtype");
  CFP_REQUIRE(k && v && kv && ksum, CFP_EINVAL, "cfp_attn_kv_reduce: null pointer");
  CFP_REQUIRE(NB > 0 && Hk > 0 && Wk > 0 && th > 0 && tw > 0 && heads > 0 && (d == 4 || d == 8 || d == 16 || d == 32) &&
                  k_ld >= heads * d && v_ld >= heads * d && v_length > 0.f,
              CFP_ESHAPE, "cfp_attn_kv_reduce: bad shape (head dim must be 4, 8, 16 or 32)");
  KvP p;
  p.k = k; p.v = v; p.kv = kv; p.ksum = ksum; p.ws = ws; p.k_ld = k_ld; p.v_ld = v_ld;
  p.NB = NB; p.Hk = Hk; p.Wk = Wk; p.th = th; p.tw = tw; p.gy = cdiv(Hk, th); p.gx = cdiv(Wk, tw);
  p.cy0 = cy0; p.cy1 = cy1; p.cx0 = cx0; p.cx1 = cx1; p.count_pad = count_pad; p.heads = heads; p.d = d;
  p.inv_len = 1.0f / v_length;
  long long groups = (long long)NB * p.gy * p.gx;
  int S = (th < Hk ? th : Hk) * (tw < Wk ? tw : Wk);
  p.nsplit = pick_nsplit(S, groups * heads);
  CFP_REQUIRE(p.nsplit == 1 || ws, CFP_EINVAL, "cfp_attn_kv_reduce: workspace required");
  CFP_REQUIRE(groups < (1ll << 31) && heads <= 65535, CFP_ESHAPE, "cfp_attn_kv_reduce: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)groups, heads, p.nsplit);
#define KV_LAUNCH(T, D) hipLaunchKernelGGL((kv_reduce_kernel<T, D>), grid, dim3(64), 0, s, p)
#define KV_SWITCH(T) switch (d) { case 4: KV_LAUNCH(T, 4); break; case 8: KV_LAUNCH(T, 8); break; \
                                  case 16: KV_LAUNCH(T, 16); break; default: KV_LAUNCH(T, 32); break; }
  if (dtype == CFP_BF16) { KV_SWITCH(bf16_t) } else if (dtype == CFP_F16) { KV_SWITCH(f16_t) } else { KV_SWITCH(float) }
#undef KV_SWITCH
#undef KV_LAUNCH
  if (p.nsplit > 1) {
    long long items = groups * heads;
    long long total = items * (d * d + d);
    int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(kv_finalize_kernel, dim3(blocks), dim3(256), 0, s, ws, kv, ksum, p.nsplit, d, items);
  }
  return cfp_check_launch("cfp_attn_kv_reduce");
}

extern "C" int cfp_attn_apply(const void* q, int q_ld, const float* kv, const float* ksum, void* out, int out_ld,
                              int NB, int Hq, int Wq, int qth, int qtw, int ey0, int ey1, int ex0, int ex1,
                              float v_length, float eps, int heads, int d, int dtype, cfp_stream_t stream) {
  CFP_REQUIRE(dtype_ok(dtype), CFP_EINVAL, "cfp_attn_apply: bad dtype");
  CFP_REQUIRE(q && kv && ksum && out, CFP_EINVAL, "cfp_attn_apply: null pointer");
  CFP_REQUIRE(NB > 0 && Hq > 0 && Wq > 0 && qth > 0 && qtw > 0 && heads > 0 && (d == 4 || d == 8 || d == 16 || d == 32) &&
                  q_ld >= heads * d && out_ld >= heads * d, CFP_ESHAPE, "cfp_attn_apply: bad shape");
  ApP p;
  p.q = q; p.kv = kv; p.ksum = ksum; p.out = out; p.q_ld = q_ld; p.out_ld = out_ld;
  p.NB = NB; p.Hq = Hq; p.Wq = Wq; p.qth = qth; p.qtw = qtw; p.ggy = cdiv(Hq, qth); p.ggx = cdiv(Wq, qtw);
  p.ey0 = ey0; p.ey1 = ey1; p.ex0 = ex0; p.ex1 = ex1; p.heads = heads; p.v_length = v_length; p.eps = eps;
  long long total = (long long)NB * Hq * Wq * heads;
  CFP_REQUIRE(total < (1ll << 31), CFP_ESHAPE, "cfp_attn_apply: too many (token, head) pairs");
  p.fheads = make_fastdiv((unsigned)heads); p.fwq = make_fastdiv((unsigned)Wq); p.fhq = make_fastdiv((unsigned)Hq);
  p.fqth = make_fastdiv((unsigned)qth); p.fqtw = make_fastdiv((unsigned)qtw);
  const int js = total < g_ap_split_below ? (d >= 32 ? 8 : d >= 16 ? 4 : d >= 8 ? 2 : 1) : 1;
  CFP_REQUIRE(total * js < (1ll << 31), CFP_ESHAPE, "cfp_attn_apply: too many lanes");
  const long long lanes = total * js;
  int blocks = (int)((lanes + 255) / 256 > 8192 ? 8192 : (lanes + 255) / 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define AP_LAUNCH(T, D, JS) hipLaunchKernelGGL((attn_apply_kernel<T, D, JS>), dim3(blocks), dim3(256), 0, s, p)
#define AP_SWITCH(T) switch (d) { case 4: AP_LAUNCH(T, 4, 1); break; \
                                  case 8: if (js > 1) AP_LAUNCH(T, 8, 2); else AP_LAUNCH(T, 8, 1); break; \
                                  case 16: if (js > 1) AP_LAUNCH(T, 16, 4); else AP_LAUNCH(T, 16, 1); break; \
                                  default: if (js > 1) AP_LAUNCH(T, 32, 8); else AP_LAUNCH(T, 32, 1); break; }
  if (dtype == CFP_BF16) { AP_SWITCH(bf16_t) } else if (dtype == CFP_F16) { AP_SWITCH(f16_t) } else { AP_SWITCH(float) }
#undef AP_SWITCH
#undef AP_LAUNCH
  return cfp_check_launch("cfp_attn_apply");
}
