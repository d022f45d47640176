// 16-bit 3x3 stride-1 convolution for FEW input channels (Cin <= 64): the whole-depth input halo of a pixel tile stays in LDS.
//
// Reference ops: the EfficientNetV2 stem-side blocks of the RGB encoder (timm ConvBnAct / EdgeResidual: conv 3x3 -> BatchNorm -> SiLU
// [+ skip], oracle/cfpnet_oracle.py encoder()) and the decoder's 3x3 convolutions on 32 / 64 channels (decoder.py:51-58 UpSampleBN).
// They are this network's many-pixel, short-K problems (6 x 10^5 ... 4 x 10^4 pixels, K = 144 ... 576): by FLOPs they are nothing,
// by HBM bytes they should take 5-15 us each at batch 8, and as implicit GEMMs they take 16-54 us (profiles/r3_conv_sweep: 2.7x
// their ideal over the family) because the im2col A operand is fetched from L2 nine times -- once per tap -- and once more per
// N-tile: 442 MB of L2->LDS traffic for a 49 MB input.  conv3x3_direct.hip removed the nine but re-staged the halo per 64-channel
// chunk and per 32-cout tile behind three coarse pipeline steps, and lost to the implicit GEMM on every one of these shapes.
//
// This kernel:
//   * a workgroup owns TH x 16 output pixels (TH = 16 or 8) and all output channels, or one of 2-3 blocks of them; the (TH + 2) x 18 input halo with all Cin
//     channels is loaded ONCE (16-byte pieces, register-staged, pixel pitch an odd number of 16-byte slots so that the 16 lanes of
//     an MFMA operand read -- 16 consecutive pixels of a row -- fall on 16 different slots of the 256-byte bank row);
//   * K runs over (tap, 8-channel chunk) in the weight tensor's own order [Cout][kh][kw][Cin]: the B fragment of k-chunk c is the
//     halo pixel shifted by tap c / (Cin / 8), so the nine taps are address arithmetic on the resident tile (per lane, two
//     compare-and-subtract updates per 32-deep MFMA step);
//   * the weights are the only streamed operand: 64-deep K-steps of [Cout][64] rows, `global_load_lds_dwordx4` into XOR-swizzled
//     128-byte rows exactly as conv_igemm2.hip stages its W tile, 2-3 stages, counted `s_waitcnt vmcnt`, one raw barrier per K-step;
//   * a wave computes 4 pixel rows x NT 16-channel tiles; accumulators transposed (weights as the MFMA row operand): a lane owns 4
//     consecutive channels of one pixel and the epilogue (folded BatchNorm, activation, optional skip) stores 8 bytes from registers.
//
// Same products, float32 accumulation in a different order than the implicit GEMM: results agree to float32 re-association
// (tests: <= 1 ulp of the storage type against cfp_conv2d_nhwc's other kernels, bit-exact on small integers).
#include "igemm_core.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16h[4] = {0u, 0u, 0u, 0u};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct HaloP {
  int PP;          // halo pixel pitch in bytes (Cin * 2 rounded up to an odd number of 16-byte slots)
  int CPT;         // 16-byte chunks per pixel = Cin / 8
  int tiles_x, tiles_y;
  int n_blocks;    // workgroups per pixel tile: each owns NT * WN * 16 output channels (they re-read the halo from L2)
  FastDiv dcpt;    // piece -> (pixel, chunk)
};

// UP = true (cfp_upsample_cat_conv3x3, decoder.py:51-58 UpSampleBN): the 16-byte pieces of the halo that belong to channels below p.up_C
// are not fetched but BLENDED from four taps of the low-resolution map (resize_kernel's own float32 arithmetic, rounded to the storage
// type as the stored upsampled tensor would have been); the other channels come from the skip tensor.
// STRIDE = 2 (the stem and the first block of an encoder stage, TF-"same" padding): the halo is (2 TH + 1) x 33 input pixels, an output pixel's
// tap (dy, dx) is input pixel (2 y + dy, 2 x + dx) of it.
template <typename H, int NT, int WN, int STAGES, bool UP = false, int STRIDE = 1>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(ConvP p, HaloP hp) {
  constexpr int WM = 4 / WN;
  constexpr int TH = 4 * WM;                 // output rows per workgroup (a wave owns 4)
  constexpr int HC = 15 * STRIDE + 3;        // halo columns
  constexpr int HPIX = ((TH - 1) * STRIDE + 3) * HC;
  static_assert(!(UP && STRIDE != 1), "the upsampling loader is stride 1");
  constexpr int NPAD = NT * WN * 16;         // weight rows staged per K-step
  constexpr int NBG = NPAD / 8;              // 8-row DMA groups
  constexpr int NB = (NBG + 3) / 4;          // DMA instructions per wave per stage
  constexpr int WSTAGE = NPAD * 128;
  constexpr int LB = UP ? 4 : 6;             // halo pieces per thread and loader pass (UP: four taps each)
  static_assert((STAGES - 2) * NB <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;                              // STAGES weight stages
  unsigned char* sX = smem + STAGES * WSTAGE;            // the halo tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int n_base = (bid % hp.n_blocks) * NPAD; bid /= hp.n_blocks;      // channel blocks of one tile are neighbours: they share the halo in L2
  const int tx_ = bid % hp.tiles_x; bid /= hp.tiles_x;
  const int ty_ = bid % hp.tiles_y;
  const int b = bid / hp.tiles_y;
  const int x0 = tx_ * 16, y0 = ty_ * TH;

  const H* __restrict__ in = reinterpret_cast<const H*>(p.in) + (long long)b * p.H * p.W * p.in_ld;
  const H* __restrict__ wt = reinterpret_cast<const H*>(p.w);
  const H* zsrc = reinterpret_cast<const H*>(g_zero16h);
  const int nk = (p.K + 63) >> 6;

  // ---- weight stages: lane (row rsub of an 8-row group, logical chunk lc) -------------------------------------------------------
  const H* b_ptr[NB];
  unsigned b_okmask = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n_base + ((j * 4 + wave) % NBG) * 8 + rsub;
    const bool ok = n < p.Cout;
    if (ok) b_okmask |= 1u << j;
    b_ptr[j] = wt + (long long)(ok ? n : 0) * p.K;
  }
  auto issue = [&](int ks, int buf) {
    unsigned char* s = sW + buf * WSTAGE;
    const int kk = (ks * 8 + lc) * 8;
    const bool kok = kk < p.K;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = kok && ((b_okmask >> j) & 1u);
      glds16(ok ? b_ptr[j] + kk : zsrc, s + ((j * 4 + wave) % NBG) * 1024);
    }
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);

  // ---- the halo: all pieces of the thread in flight, then the LDS stores (the compiler drains the DMA queue before them; both are
  //      needed before the first MFMA anyway) ---------------------------------------------------------------------------------------
  {
    const int nitems = HPIX * hp.CPT;
    const H* __restrict__ low = nullptr;
    int upc = 0;
    if constexpr (UP) { low = reinterpret_cast<const H*>(p.up_src) + (long long)b * p.up_H * p.up_W * p.up_ld; upc = p.up_C >> 3; }
    for (int base = 0; base < nitems; base += 256 * LB) {
      u32x4 v[LB][UP ? 4 : 1];
      int dst[LB];
      float lyx[LB][UP ? 2 : 1];
      bool blend[LB];
#pragma unroll
      for (int n = 0; n < LB; ++n) {
        const int i = base + tid + n * 256;
        unsigned upx, uch;
        fd_rowcol((unsigned)i, hp.dcpt, upx, uch);
        const int px = (int)upx, ch = (int)uch;
        const int hy = px / HC, hx = px - hy * HC;
        const int y = y0 * STRIDE - p.pad_t + hy, x = x0 * STRIDE - p.pad_l + hx;
        const bool ok = i < nitems && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        dst[n] = i < nitems ? px * hp.PP + ch * 16 : -1;
        blend[n] = false;
#pragma unroll
        for (int t = 0; t < (UP ? 4 : 1); ++t) v[n][t] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (UP) {
          if (ch < upc) {
            blend[n] = ok;
            const float fy = p.up_sy * (float)y, fx = p.up_sx * (float)x;      // torch: src = scale * dst_index (align_corners=True)
            const int ys = (int)fy, xs = (int)fx;
            lyx[n][0] = fy - (float)ys; lyx[n][1] = fx - (float)xs;
            if (ok) {
              const int dyo = (ys < p.up_H - 1 ? 1 : 0) * p.up_W * p.up_ld, dxo = (xs < p.up_W - 1 ? 1 : 0) * p.up_ld;
              const H* s00 = low + (ys * p.up_W + xs) * p.up_ld + ch * 8;
              v[n][0] = *reinterpret_cast<const u32x4*>(s00); v[n][1] = *reinterpret_cast<const u32x4*>(s00 + dxo);
              v[n][2] = *reinterpret_cast<const u32x4*>(s00 + dyo); v[n][3] = *reinterpret_cast<const u32x4*>(s00 + dyo + dxo);
            }
            continue;
          }
        }
        if (ok) v[n][0] = *reinterpret_cast<const u32x4*>(in + (y * p.W + x) * p.in_ld + ch * 8);      // one image < 2^31 elements (host check)
      }
#pragma unroll
      for (int n = 0; n < LB; ++n) {
        if (dst[n] < 0) continue;
        if constexpr (UP) {
          if (blend[n]) {
            float t[4][8], o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) Vec<H>::load(reinterpret_cast<const H*>(&v[n][q]), t[q]);
            const float ly1 = lyx[n][0], lx1 = lyx[n][1], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ly0 * (lx0 * t[0][e] + lx1 * t[1][e]) + ly1 * (lx0 * t[2][e] + lx1 * t[3][e]);
            Vec<H>::store(reinterpret_cast<H*>(sX + dst[n]), o);
            continue;
          }
        }
        *reinterpret_cast<u32x4*>(sX + dst[n]) = v[n][0];
      }
    }
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // im2col position of this lane's k-chunk (c = 4 * step + fq): tap offset inside the halo and chunk inside the pixel
  const int ntap_chunks = 9 * hp.CPT;
  int c_cc, c_dx = 0, c_off = 0, c_idx = fq;
  {
    const int tap = fq / hp.CPT;
    c_cc = fq - tap * hp.CPT;
    c_dx = tap;                                  // fq <= 3 and CPT >= 1: tap <= 3; normalised below
    while (c_dx >= 3) { c_dx -= 3; c_off += HC * hp.PP; }
    c_off += c_dx * hp.PP;
  }
  const unsigned char* xrow = sX + ((wm * 4 * STRIDE) * HC + fr * STRIDE) * hp.PP;      // tap (0, 0) of output pixel (row wm * 4, column fr)
  const int growb = STRIDE * HC * hp.PP;                                                 // one output row further

  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks % STAGES;
    const int ahead = min(nk - 1 - ks, STAGES - 2);
    if (ahead >= 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * NB>();
    else if (ahead == 1) wait_vmcnt<NB>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's halo stores (first time round) and fragment reads of the previous step
    __builtin_amdgcn_s_barrier();          // stage `buf` (and the halo) landed for every wave; stage buf-1 fully consumed.  Raw: a __syncthreads() would drain the DMA queue
    asm volatile("" ::: "memory");
    if (ks + STAGES - 1 < nk) issue(ks + STAGES - 1, (ks + STAGES - 1) % STAGES);
    const unsigned char* cW = sW + buf * WSTAGE + (wn * NT * 16) * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 wf[NT], xf[4];
      const int pc = ((s * 4 + fq) ^ (fr & 7)) * 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const s16x8*>(cW + (j * 16 + fr) * 128 + pc);
      // chunks past the ninth tap meet zero weights: read any finite data -- the tile's first piece, the SAME address in every such lane
      // (identical addresses broadcast; per-lane addresses there collided with the live lanes of their read group)
      const bool live = c_idx < ntap_chunks;
      const int xo = c_off + c_cc * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) xf[g] = *reinterpret_cast<const s16x8*>(live ? xrow + g * growb + xo : sX);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[g][j] = mfma16<H>(wf[j], xf[g], acc[g][j]);   // acc[r] = channel 4 fq + r of tile j, pixel fr
      // next 32-deep step: four chunks further
      c_idx += 4;
      c_cc += 4;
      while (c_cc >= hp.CPT) {
        c_cc -= hp.CPT;
        c_off += hp.PP;
        if (++c_dx == 3) { c_dx = 0; c_off += (HC - 3) * hp.PP; }
      }
    }
  }

  // ---- epilogue: folded BatchNorm / bias, activation, optional skip; 8-byte stores from the accumulators -------------------------------
  H* __restrict__ out = reinterpret_cast<H*>(p.out) + (long long)b * p.Ho * p.Wo * p.out_ld;
  const H* __restrict__ res = p.res ? reinterpret_cast<const H*>(p.res) + (long long)b * p.Ho * p.Wo * p.res_ld : nullptr;
  const int x = x0 + fr;
  f32x4 sc[NT], sh[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n_base + (wn * NT + j) * 16 + fq * 4;
    const bool ok = n < p.Cout;
    sc[j] = (ok && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    sh[j] = (ok && p.shift) ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  with_act(p.act, [&](auto A) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int y = y0 + wm * 4 + g;
      const bool pix_ok = y < p.Ho && x < p.Wo;
      const long long pix = (long long)y * p.Wo + x;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n_base + (wn * NT + j) * 16 + fq * 4;
        float yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = act_c16<decltype(A)::value>(acc[g][j][r] * sc[j][r] + sh[j][r]);
        uint2 pk;
        pk.x = pack2<H>(yv[0], yv[1]);
        pk.y = pack2<H>(yv[2], yv[3]);
        if (!(pix_ok && n < p.Cout)) continue;
        if (res) {
          // the skip is added to the ROUNDED activation, as the other conv kernels do (they round into their LDS C tile first)
          const uint2 rr = *reinterpret_cast<const uint2*>(res + pix * p.res_ld + n);
          const H* ph = reinterpret_cast<const H*>(&pk);
          const H* rh = reinterpret_cast<const H*>(&rr);
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = to_f32<H>(ph[r]) + to_f32<H>(rh[r]);
          H oh[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) oh[r] = from_f32<H>(o[r]);
          pk = *reinterpret_cast<const uint2*>(oh);
        }
        *reinterpret_cast<uint2*>(out + pix * p.out_ld + n) = pk;
      }
    }
  });
}

int g_halo_odd_pitch = 0;    // cfp_debug_set key 27: 1 = the round-3 odd pixel pitch
int g_halo_stages = 0;      // cfp_debug_set key 13: force the number of weight stages (2-4), 0 = automatic

struct HCfg { int nt, wn; };
constexpr HCfg kHCfg[] = {
    {1, 1},  // 0: Cout <= 16, 16 x 16 pixels
    {2, 1},  // 1: <= 32
    {4, 1},  // 2: <= 64
    {2, 2},  // 3: <= 64, 8 x 16 pixels
    {4, 2},  // 4: <= 128
    {5, 2},  // 5: <= 160
    {7, 2},  // 6: <= 224
    {1, 2},  // 7: <= 32, 8 x 16 pixels
};
constexpr int kNumHCfg = sizeof(kHCfg) / sizeof(kHCfg[0]);

template <typename H, int NT, int WN, bool UP = false, int STRIDE = 1>
int launch_h(const ConvP& p, hipStream_t s) {
  constexpr int TH = 4 * (4 / WN);
  constexpr int NPAD = NT * WN * 16;
  HaloP hp;
  hp.n_blocks = cdiv(p.Cout, NPAD);
  hp.CPT = p.Cin / 8;
  hp.dcpt = make_fastdiv((unsigned)hp.CPT);
  if ((long long)p.H * p.W * p.in_ld >= (1ll << 31)) return -1;
  // Pixel pitch in 16-byte slots.  `ds_read_b128` is served in four 16-lane groups that pair the lanes (fr in {0-3, 12-15}, fq) with (fr in
  // {4-11}, fq ^ 1) (MI355X_MICROARCH.md, LDS table): the first set reads slots fr * S * P + o, the second fr * S * P + o + 1 (the next chunk of the
  // same pixel), and the 16 of them must differ mod 16.  Stride 1: P = 2 (mod 4) -- the first set then covers the even slots, the second the
  // odd ones (an ODD pitch, the classic padding and this kernel's rule until round 4, puts 43-50 % conflict cycles on these reads:
  // tools/halo_bank_model.py, measured 25-41 % of all LDS cycles, profiles/r3m_pmc_sq_inference.json).  One chunk per pixel (the sets then
  // read neighbouring PIXELS) and stride 2 (pixel step 2 P) want P odd.  Odd chunk counts keep 11-20 % on the steps where the two sets
  // straddle a tap (the shift between them is then even); no linear pitch removes both cases.
  int slots = hp.CPT;
  if (STRIDE == 1 && hp.CPT > 1) { while ((slots & 3) != 2) ++slots; }
  else if ((slots & 1) == 0) ++slots;
  if (g_halo_odd_pitch) { slots = hp.CPT; if ((slots & 1) == 0) ++slots; }      // cfp_debug_set key 27: the round-3 rule (A/B, PMC comparison)
  hp.PP = slots * 16;
  hp.tiles_x = cdiv(p.Wo, 16); hp.tiles_y = cdiv(p.Ho, TH);
  const int hpix = ((TH - 1) * STRIDE + 3) * (15 * STRIDE + 3);
  const size_t halo = (size_t)hpix * hp.PP;
  const long long tiles = (long long)p.B * hp.tiles_x * hp.tiles_y * hp.n_blocks;
  // weight stages: two.  More would hide more of the DMA latency behind MFMAs, but measured (tools/conv_bench.py --halo, us with
  // 2 / 3 / 4 stages: 614400 px x 128 ch 82 / 93 / 96, 153600 x 160 36 / 38 / 52, 614400 x 16 23 / 25 / 25, 38400 x 224 19 / 24 / 32)
  // the LDS they take costs more in resident workgroups than it gains -- the same finding as for the implicit GEMM's tiles
  int stages = g_halo_stages ? g_halo_stages : 2;
  const size_t lds = (size_t)stages * NPAD * 128 + halo;
  if (lds > 160 * 1024 || tiles >= (1ll << 31)) return -1;
#define HL(ST)                                                                                                                      \
  do {                                                                                                                              \
    auto k = conv3x3_halo_kernel<H, NT, WN, ST, UP, STRIDE>;                                                                        \
    static bool attr = false;                                                                                                       \
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; } \
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), lds, s, p, hp);                                                         \
  } while (0)
  if constexpr (UP || STRIDE != 1) { HL(2); } else { if (stages == 4) HL(4); else if (stages == 3) HL(3); else HL(2); }
#undef HL
  return 0;
}

}  // namespace

int conv3x3_halo_num_variants() { return kNumHCfg; }
void conv3x3_halo_debug_stages(int v) { g_halo_stages = v; }
void conv3x3_halo_debug_odd_pitch(int v) { g_halo_odd_pitch = v; }

// The problems this kernel takes: 3x3, stride 1 or 2, undilated, 16-bit, Cin a multiple of 8 and <= 128 (the plan uses it up to 64), no
// LayerNorm epilogue / per-image weights.
bool conv3x3_halo_takes(const ConvP& p) {
  return p.KH == 3 && p.KW == 3 && (p.stride == 1 || (p.stride == 2 && p.up_src == nullptr)) && p.dil <= 1 && p.Cin % 8 == 0 && p.Cin >= 8 && p.Cin <= 128 && p.Cout % 8 == 0 &&
         p.Cout <= 512 && p.ln_gamma == nullptr && p.rows_per_batch == 0 && p.k2 == 0 && p.K == 9 * p.Cin &&
         p.pad_t >= 0 && p.pad_l >= 0 && p.pad_t <= 2 && p.pad_l <= 2;
}

// variant < 0: chosen from Cout and the number of tiles.  Returns 0, or a negative value if the variant cannot run this problem.
int conv3x3_halo_launch(int v, const ConvP& p, hipStream_t s) {
  if (v < 0) {
    const long long t16 = (long long)p.B * cdiv(p.Wo, 16) * cdiv(p.Ho, 16);
    if (p.up_src != nullptr && p.Cout > 64) return -3;
    if (p.up_src != nullptr) v = p.Cout <= 32 ? 7 : 3;       // 8 x 16 pixel tiles: the blended halo is large, four workgroups per CU matter more (up4: 85 us, 16 x 16 tiles 104, direct kernel 112)
    else if (p.Cout <= 16) v = 0;
    else if (p.Cout <= 32) v = t16 >= 1024 ? 1 : 7;
    else if (p.Cout <= 64) v = t16 >= 1024 ? 2 : 3;
    else if (p.Cout <= 128) v = 4;
    else if (p.Cout <= 160) v = 5;
    else v = 4;                      // two or more 128-channel blocks
  }
  if (v >= kNumHCfg) return -3;
  if (p.up_src != nullptr) {      // upsample + concatenation in the loader: the thin-output tiles only (the decoder's first conv of a stage)
#define HU(NT, WN) (p.f16 ? launch_h<f16_t, NT, WN, true>(p, s) : launch_h<bf16_t, NT, WN, true>(p, s))
    switch (v) {
      case 0: return HU(1, 1);
      case 1: return HU(2, 1);
      case 2: return HU(4, 1);
      case 3: return HU(2, 2);
      case 7: return HU(1, 2);
      default: return -3;
    }
#undef HU
  }
  if (p.stride == 2) {
#define HS(NT, WN) (p.f16 ? launch_h<f16_t, NT, WN, false, 2>(p, s) : launch_h<bf16_t, NT, WN, false, 2>(p, s))
    switch (v) {
      case 0: return HS(1, 1);
      case 1: return HS(2, 1);
      case 2: return HS(4, 1);
      case 3: return HS(2, 2);
      case 4: return HS(4, 2);
      case 5: return HS(5, 2);
      case 7: return HS(1, 2);
      default: return -3;
    }
#undef HS
  }
#define HV(NT, WN) (p.f16 ? launch_h<f16_t, NT, WN>(p, s) : launch_h<bf16_t, NT, WN>(p, s))
  switch (v) {
    case 0: return HV(1, 1);
    case 1: return HV(2, 1);
    case 2: return HV(4, 1);
    case 3: return HV(2, 2);
    case 4: return HV(4, 2);
    case 5: return HV(5, 2);
    case 6: return HV(7, 2);
    case 7: return HV(1, 2);
    default: return -3;
  }
#undef HV
}
