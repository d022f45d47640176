// 3x3 stride-1 convolution for FLOAT32 STORAGE with split-precision (f16x3) matrix math, whole-depth input halo in LDS.
//
// Reference ops: the 3x3 convolutions of the decoder and the depth head (decoder.py:13,43-58,70-80 UpSampleBN / conv0, transformer.py:197-200
// DAPM convs) and of the RGB encoder's stem-side blocks (encoder.py:57-69) -- in the DEFAULT numerics of the drop-in boundary (float32
// tensors, A_hi W_hi + A_hi W_lo + A_lo W_hi on v_mfma_f32_16x16x32_f16; conv_igemm_x3.hip has the arithmetic and its error).
//
// Why: with several batches in flight the f16x3 implicit GEMMs are bound by the L2 -> LDS path (tools/igemm_x3_probe.py: the head conv's
// operand DMA alone takes 357 of its 625 us, at the ~60 GB/s per CU that path delivers; tools/conv_bench_x3.py: most launches sit at
// 60-100 % of that bound).  The 3x3 convolutions move 24 of the 35 GB per forward through it, nine times their input: the im2col A operand
// is fetched once per tap.  Here, as in conv3x3_halo.hip (the 16-bit kernel this one is modelled on):
//   * a workgroup owns TH x 16 output pixels (TH = 16 or 8) and NT x WN x 16 output channels; the (TH + 2) x 18 input halo with ALL Cin
//     channels is loaded ONCE, register-staged, and SPLIT ONCE on the way into LDS -- every element is converted a single time, where the
//     implicit GEMM converts it in the K loop of every wave, tap and N-tile that meets it.  LDS image: a HI plane and a LO plane, each
//     [pixel][Cin] halves with a pixel pitch of an ODD number of 16-byte slots (the planes are far apart on purpose: hi and lo of a quad
//     16 bytes apart were fused by hipcc into ds_read2_b64, which is banked mod 32 in 4 x 16-lane groups and ran at 32-47 % conflict
//     cycles -- profiles/r4e_x3_pmc_sq.json);
//   * K runs over (tap, channel) in the weight tensor's own order, 32 per step, against the SAME pre-split weight operand the implicit GEMM
//     takes (cfp_pack_w_x3: MFMA slot (fq, e) <-> k = 4 fq + e, 16 + 4 fq + e - 4): lane (pixel fr, fq) needs the channel quads 8 ks + fq and
//     8 ks + 4 + fq of the flattened (tap, channel / 4) axis -- two positions per lane advanced by compare-and-subtract, four ds_read_b64
//     per fragment (hi / lo of each quad).  Lanes fq and fq ^ 1 read the two quads of one 8-channel group: the 32 lanes of a ds_read_b64
//     group cover 16 pixels x 16 contiguous bytes at an odd slot pitch = all 64 banks once (MI355X_MICROARCH.md, LDS table);
//   * the weights are the only streamed operand (LDS-DMA, 128-byte rows [hi(32) | lo(32)], XOR-swizzled, STAGES deep, one raw barrier per
//     K-step) -- no conversion and no float32 operand in the K loop at all;
//   * a wave computes 4 pixel rows x NT 16-channel tiles, accumulators transposed (a lane owns 4 consecutive channels of one pixel):
//     folded BatchNorm / bias, activation, optional skip, 16-byte float32 stores.
// L2 -> LDS bytes per workgroup: halo + all weights of its channel block, e.g. conv0 (32 -> 128 channels at 240 x 320 x 8): 0.45 GB per
// launch against the implicit GEMM's 1.42 GB; up4's first conv (80 -> 32): 0.47 against 2.26.
#include "igemm_core.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16hx[4] = {0u, 0u, 0u, 0u};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct HaloX3P {
  int PP;          // halo pixel pitch in bytes PER PLANE: Cin * 2 rounded up to an odd number of 16-byte slots
  int LO;          // byte offset of the lo plane = halo pixels * PP
  int QPP;         // channel quads per pixel = Cin / 4 (even: Cin % 8 == 0)
  int tiles_x, tiles_y;
  int n_blocks;    // workgroups per pixel tile (each owns NT * WN * 16 output channels and re-reads the halo from L2)
  FastDiv dq;      // piece -> (pixel, quad)
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NT, int WN, int STAGES>
__global__ __launch_bounds__(256) void conv3x3_halo_x3_kernel(ConvP p, HaloX3P hp) {
  constexpr int WM = 4 / WN;
  constexpr int TH = 4 * WM;                 // output rows per workgroup (a wave owns 4)
  constexpr int HC = 18;                     // halo columns
  constexpr int HPIX = (TH + 2) * HC;
  constexpr int NPAD = NT * WN * 16;         // weight rows staged per K-step
  constexpr int NBG = NPAD / 8;
  constexpr int NB = (NBG + 3) / 4;
  constexpr int WSTAGE = NPAD * 128;
  constexpr int LB = 6;                      // halo pieces per thread and loader pass
  static_assert((STAGES - 2) * NB <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;
  unsigned char* sX = smem + STAGES * WSTAGE;

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

  const float* __restrict__ in = reinterpret_cast<const float*>(p.in) + (long long)b * p.H * p.W * p.in_ld;
  const f16_t* __restrict__ wt = reinterpret_cast<const f16_t*>(p.w);
  const void* zsrc = reinterpret_cast<const void*>(g_zero16hx);
  const int nk = (p.K + 31) >> 5;
  const int wrow = nk * 64;

  // ---- weight stages: lane (row rsub of an 8-row group, logical 16-byte chunk lc of the 128-byte K-step row) -----------------------
  const f16_t* b_ptr[NB];
  unsigned b_okmask = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n_base + ((j * 4 + wave) % NBG) * 8 + rsub;
    const bool ok = n < p.Cout;
    if (ok) b_okmask |= 1u << j;
    b_ptr[j] = wt + (long long)(ok ? n : 0) * wrow;
  }
  auto issue = [&](int ks, int buf) {
    unsigned char* s = sW + buf * WSTAGE;
    const int kk = (ks * 8 + lc) * 8;          // rows are zero-padded to whole K-steps
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = (b_okmask >> j) & 1u;
      glds16(ok ? (const void*)(b_ptr[j] + kk) : zsrc, s + ((j * 4 + wave) % NBG) * 1024);
    }
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);

  // ---- the halo: 16-byte float32 pieces (one channel quad of one pixel), all of the thread's loads of a pass in flight, then split and
  //      stored as [hi4 | .. | lo4 | ..] of the quad's 8-channel group ----------------------------------------------------------------
  {
    const int nitems = HPIX * hp.QPP;
    for (int base = 0; base < nitems; base += 256 * LB) {
      f32x4 v[LB];
      int dst[LB];
#pragma unroll
      for (int n = 0; n < LB; ++n) {
        const int i = base + tid + n * 256;
        unsigned upx, uq;
        fd_rowcol((unsigned)i, hp.dq, upx, uq);
        const int px = (int)upx, q = (int)uq;
        const int hy = px / HC, hx = px - hy * HC;
        const int y = y0 - p.pad_t + hy, x = x0 - p.pad_l + hx;
        const bool ok = i < nitems && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        dst[n] = i < nitems ? px * hp.PP + q * 8 : -1;
        v[n] = ok ? *reinterpret_cast<const f32x4*>(in + (y * p.W + x) * p.in_ld + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};      // one image < 2^31 elements (host check)
      }
#pragma unroll
      for (int n = 0; n < LB; ++n) {
        if (dst[n] < 0) continue;
        f16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = f2h(v[n][e]);                              // round-to-nearest, saturating
          lo[e] = (f16_t)(v[n][e] - (float)hi[e]);           // exact difference, rounded once: |x - hi - lo| <= 2^-22 |x|
        }
        *reinterpret_cast<f16x4*>(sX + dst[n]) = hi;
        *reinterpret_cast<f16x4*>(sX + hp.LO + dst[n]) = lo;
      }
    }
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // positions of this lane's two channel quads (k4 = 8 ks + fq and 8 ks + 4 + fq) in the flattened (tap, quad) axis: quad inside the pixel,
  // byte offset of the tap inside the halo
  const int nq_all = 9 * hp.QPP;
  int cq[2], dx[2], off[2], k4[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    k4[h] = fq + 4 * h;
    int tap = k4[h] / hp.QPP;
    cq[h] = k4[h] - tap * hp.QPP;
    const int ty = tap / 3;
    dx[h] = tap - ty * 3;
    off[h] = (ty * HC + dx[h]) * hp.PP;
  }
  const unsigned char* xrow = sX + ((wm * 4) * HC + fr) * hp.PP;      // tap (0, 0) of output pixel (row wm * 4, column fr)
  const int growb = HC * hp.PP;
  const int pc0 = ((fq) ^ (fr & 7)) * 16, pc1 = ((4 + fq) ^ (fr & 7)) * 16;

  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks % STAGES;
    const int ahead = min(nk - 1 - ks, STAGES - 2);
    if (ahead >= 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * NB>();
    else if (ahead == 1) wait_vmcnt<NB>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's halo stores (first time round) and the fragment reads of the previous step
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + STAGES - 1 < nk) issue(ks + STAGES - 1, (ks + STAGES - 1) % STAGES);
    const unsigned char* cW = sW + buf * WSTAGE + (wn * NT * 16) * 128;
    f16x8 whi[NT], wlo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      whi[j] = *reinterpret_cast<const f16x8*>(cW + (j * 16 + fr) * 128 + pc0);
      wlo[j] = *reinterpret_cast<const f16x8*>(cW + (j * 16 + fr) * 128 + pc1);
    }
    // quads past the ninth tap meet zero weights: read any finite data -- the tile's first piece, the same address in every such lane
    // (identical addresses broadcast; per-lane addresses would collide with the live lanes of the read group)
    const bool live0 = k4[0] < nq_all, live1 = k4[1] < nq_all;
    const int xo0 = off[0] + cq[0] * 8, xo1 = off[1] + cq[1] * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned char* xb0 = live0 ? xrow + g * growb + xo0 : sX;
      const unsigned char* xb1 = live1 ? xrow + g * growb + xo1 : sX;
      const f16x4 h0 = *reinterpret_cast<const f16x4*>(xb0), l0 = *reinterpret_cast<const f16x4*>(xb0 + hp.LO);
      const f16x4 h1 = *reinterpret_cast<const f16x4*>(xb1), l1 = *reinterpret_cast<const f16x4*>(xb1 + hp.LO);
      const f16x8 xhi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      const f16x8 xlo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
      for (int j = 0; j < NT; ++j) {      // acc[r] = channel 4 fq + r of tile j, pixel fr
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[j], xhi, acc[g][j], 0, 0, 0);
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xlo, acc[g][j], 0, 0, 0);
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xhi, acc[g][j], 0, 0, 0);
      }
    }
    // next K-step: eight quads further for both positions
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      k4[h] += 8;
      cq[h] += 8;
      while (cq[h] >= hp.QPP) {
        cq[h] -= hp.QPP;
        off[h] += hp.PP;
        if (++dx[h] == 3) { dx[h] = 0; off[h] += (HC - 3) * hp.PP; }
      }
    }
  }

  // ---- epilogue: folded BatchNorm / bias, activation, optional skip; 16-byte float32 stores from the accumulators ----------------------
  float* __restrict__ out = reinterpret_cast<float*>(p.out) + (long long)b * p.Ho * p.Wo * p.out_ld;
  const float* __restrict__ res = p.res ? reinterpret_cast<const float*>(p.res) + (long long)b * p.Ho * p.Wo * p.res_ld : nullptr;
  const int x = x0 + fr;
  with_act(p.act, [&](auto A) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n_base + (wn * NT + j) * 16 + fq * 4;
      if (n >= p.Cout) continue;
      const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int y = y0 + wm * 4 + g;
        if (!(y < p.Ho && x < p.Wo)) continue;
        const long long pix = (long long)y * p.Wo + x;
        f32x4 yv;
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = act_c<decltype(A)::value>(acc[g][j][r] * sc[r] + sh[r]);
        if (res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(res + pix * p.res_ld + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) yv[r] += rv[r];
        }
        *reinterpret_cast<f32x4*>(out + pix * p.out_ld + n) = yv;
      }
    }
  });
}

// ---- deep inputs: the halo by 32-CHANNEL CHUNKS, double buffered -------------------------------------------------------------------
// The whole-depth kernel above loses from 64 input channels up: its halo (95-124 KB) leaves one workgroup per CU whose load phase nothing
// overlaps.  Here (Cin % 32 == 0) the K loop runs chunk-major -- for every 32-channel chunk the nine taps, i.e. K-step (tap, chunk) of the
// weight rows in the order chunk 0: taps 0-8, chunk 1: taps 0-8, ... (a K-step of cfp_pack_w_x3's operand is 32 consecutive k = one tap's 32
// channels when Cin % 32 == 0, so any order is an address) -- and the halo of chunk c + 1 (pixels x 32 channels, split into a hi and a lo
// plane of 5 sixteen-byte slots per pixel) is fetched into registers at the start of chunk c and stored into the other LDS buffer after
// chunk c's ninth tap.  LDS: two chunk buffers + two weight stages (16 x 16 pixels, 128 output channels: 104 + 32 KB), so the tile can be
// 256 pixels at ANY depth and the weights are re-read once per 256 pixels instead of once per 128: head conv 1.9 GB through the L2 -> LDS
// path instead of the implicit GEMM's 5.7 GB.
//
// UP = true (round 5; cfp_upsample_cat_conv3x3 in float32 storage -- decoder.py:51-58 UpSampleBN's first conv): the input is the CONCATENATION
// [bilinear upsample (align_corners=True) of the low-resolution tensor p.up_src (p.up_C channels, a multiple of 32) | skip tensor p.in
// (p.Cin - p.up_C channels, a multiple of 4)], neither of which is materialised: chunks below p.up_C / 32 are BLENDED in the loader (four
// taps per piece, cfp_resize_bilinear's own expression in float32, then split like any other chunk), the chunks behind them come from the
// skip tensor with the channels past its end read as zeros.  The weight operand is packed over that padded channel axis
// ([Cout][9][up_C + 32 ceil(Cskip / 32)], ops.pack_w_x3_cat).  Removes the resize launch, the write + read of the upsampled tensor
// (157 MB at up4, batch 8) and the nine-fold tap re-fetch of the implicit GEMM these layers ran before (Cin = 80 / 168 / 312 / 392 is not a
// multiple of 32).
//
// NSTW = 3 (round 5): three weight stages with COUNTED waits.  The weights of K-step `it` are issued TWO steps ahead, so at the top of a step the
// stage it needs is strictly older than everything issued during the previous step (the next stage's DMA and, after tap 7, the next chunk's halo
// loads -- whichever side of the DMA builtin the compiler puts those plain loads): "all but the newest NB (+ NLD) operations have landed" is exact.
// With two stages the weights of a step were issued ONE step before it and the wait had to be a full one (the probe of tools/probes/
// chunk_x3_probe.py: 75 of the head conv's 645 us are exposed weight-DMA latency).
//
// SB = true (round 5): ONE chunk buffer instead of two.  The probe of the three-stage variants showed what bounds these kernels: <2,2> at 74 KB (two
// workgroups per CU) runs the head conv in 678 us, the same tile at 82 KB (one per CU) in 1 037 -- a workgroup's load / split / store, MFMA and
// output phases serialise, and only a SECOND resident workgroup fills them.  The 128-channel 8 x 16 pixel tile <4,2> needs 90 KB with two chunk
// buffers; with one it is 61 KB = two per CU, at the price of one more barrier per chunk (the next chunk is stored after everybody has left
// the current one) -- which the other workgroup covers.
template <int NT, int WN, bool UP = false, int NSTW = 2, bool SB = false>
__global__ __launch_bounds__(256) void conv3x3_chunk_x3_kernel(ConvP p, HaloX3P hp) {
  constexpr int WM = 4 / WN;
  constexpr int TH = 4 * WM;
  constexpr int HC = 18;
  constexpr int HPIX = (TH + 2) * HC;
  constexpr int NPAD = NT * WN * 16;
  constexpr int NBG = NPAD / 8;
  constexpr int NB = (NBG + 3) / 4;
  constexpr int WSTAGE = NPAD * 128;
  constexpr int PPC = 80;                                  // bytes per pixel and plane of a chunk: 32 halves + 16 bytes (5 slots: odd)
  constexpr int LO = HPIX * PPC;                           // lo plane behind the hi plane
  constexpr int CBUF = 2 * LO;                             // one chunk buffer
  constexpr int NLD = (HPIX * 8 + 255) / 256;              // 16-byte float32 pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(NSTW == 2 || NSTW == 3, "weight stages");
  static_assert(NB + NLD * (UP ? 4 : 1) <= 63, "vmcnt field");
  unsigned char* sW = smem;
  unsigned char* sX = smem + NSTW * WSTAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int rsub = lane >> 3;
  const int lc = (lane & 7) ^ rsub;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int n_base = (bid % hp.n_blocks) * NPAD; bid /= hp.n_blocks;
  const int tx_ = bid % hp.tiles_x; bid /= hp.tiles_x;
  const int ty_ = bid % hp.tiles_y;
  const int b = bid / hp.tiles_y;
  const int x0 = tx_ * 16, y0 = ty_ * TH;

  const float* __restrict__ in = reinterpret_cast<const float*>(p.in) + (long long)b * p.H * p.W * p.in_ld;
  const f16_t* __restrict__ wt = reinterpret_cast<const f16_t*>(p.w);
  const void* zsrc = reinterpret_cast<const void*>(g_zero16hx);
  const int NC0 = UP ? (p.up_C >> 5) : 0;                  // blended chunks (UP)
  const int c_skip = UP ? p.Cin - p.up_C : p.Cin;          // channels of `in` (UP: the skip tensor; any multiple of 4, zero-padded to chunks)
  const int NC = NC0 + ((c_skip + 31) >> 5);               // 32-channel chunks
  const float* __restrict__ low = UP ? reinterpret_cast<const float*>(p.up_src) + (long long)b * p.up_H * p.up_W * p.up_ld : nullptr;
  const int nit = 9 * NC;
  const int wrow = nit * 64;

  const f16_t* b_ptr[NB];
  unsigned b_okmask = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n_base + ((j * 4 + wave) % NBG) * 8 + rsub;
    const bool ok = n < p.Cout;
    if (ok) b_okmask |= 1u << j;
    b_ptr[j] = wt + (long long)(ok ? n : 0) * wrow;
  }
  auto issue_w = [&](int c, int tap, int buf) {            // K-step (tap, chunk c) of the weight rows
    unsigned char* s = sW + buf * WSTAGE;
    const int kk = ((tap * NC + c) * 8 + lc) * 8;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = (b_okmask >> j) & 1u;
      glds16(ok ? (const void*)(b_ptr[j] + kk) : zsrc, s + ((j * 4 + wave) % NBG) * 1024);
    }
  };
  // this thread's pieces of a chunk's halo: (pixel, quad) -> image offset / LDS offset (the same for every chunk)
  int src_off[NLD], dst_off[NLD];
  unsigned src_okmask = 0;
#pragma unroll
  for (int n = 0; n < NLD; ++n) {
    const int i = tid + n * 256;
    const int px = i >> 3, q = i & 7;
    const int hy = px / HC, hx = px - hy * HC;
    const int y = y0 - p.pad_t + hy, x = x0 - p.pad_l + hx;
    const bool ok = i < HPIX * 8 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    src_off[n] = ok ? (y * p.W + x) * p.in_ld + (UP ? 0 : q * 4) : 0;       // one image < 2^31 elements (host check); invalid pieces read the image's first quad
    if (ok) src_okmask |= 1u << n;
    dst_off[n] = i < HPIX * 8 ? px * PPC + q * 8 : -1;
  }
  // UP: this thread's pieces in the low-resolution source: top-left tap offset, the +1 steps (0 at the last row / column) and the two
  // interpolation weights -- cfp_resize_bilinear's own arithmetic (src = scale * dst_index, truncation, 1 - l)
  int up_off[UP ? NLD : 1], up_dxo[UP ? NLD : 1], up_dyo[UP ? NLD : 1];
  float up_ly[UP ? NLD : 1], up_lx[UP ? NLD : 1];
  if constexpr (UP) {
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * 256;
      const int px = i >> 3, q = i & 7;
      const int hy = px / HC, hx = px - hy * HC;
      const int y = min(max(y0 - p.pad_t + hy, 0), p.H - 1), x = min(max(x0 - p.pad_l + hx, 0), p.W - 1);      // (pieces outside the image are zeroed by src_okmask)
      const float fy = p.up_sy * (float)y, fx = p.up_sx * (float)x;
      const int ys = (int)fy, xs = (int)fx;
      up_ly[n] = fy - (float)ys; up_lx[n] = fx - (float)xs;
      up_dyo[n] = (ys < p.up_H - 1 ? 1 : 0) * p.up_W * p.up_ld;
      up_dxo[n] = (xs < p.up_W - 1 ? 1 : 0) * p.up_ld;
      up_off[n] = (ys * p.up_W + xs) * p.up_ld + q * 4;
    }
  }
  f32x4 hv[NLD][UP ? 4 : 1];
  // EVERY wave issues the same number of loads per chunk (clamped address, value selected afterwards): a wave whose last pieces lie outside
  // the halo must not skip the instruction
  auto load_chunk = [&](int c) {
    if (UP && c < NC0) {                                     // wave-uniform: a blended chunk -- four taps per piece
#pragma unroll
      for (int n = 0; n < NLD; ++n) {
        const float* s00 = low + up_off[UP ? n : 0] + c * 32;
        asm volatile("" : "+v"(s00));
        hv[n][0] = *reinterpret_cast<const f32x4*>(s00);
        if constexpr (UP) {
          hv[n][1] = *reinterpret_cast<const f32x4*>(s00 + up_dxo[n]);
          hv[n][2] = *reinterpret_cast<const f32x4*>(s00 + up_dyo[n]);
          hv[n][3] = *reinterpret_cast<const f32x4*>(s00 + up_dyo[n] + up_dxo[n]);
        }
      }
      return;
    }
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      // UP: src_off is the pixel only; the piece's four channels of the skip tensor, clamped (channels past its end are zeroed in store_chunk)
      const float* src = UP ? in + src_off[n] + min((c - NC0) * 32 + ((tid + n * 256) & 7) * 4, c_skip - 4) : in + src_off[n] + c * 32;
      asm volatile("" : "+v"(src));                          // keep the load unconditional (no exec-masked skip)
      hv[n][0] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_chunk = [&](int buf, int c) {
    unsigned char* d = sX + buf * CBUF;
    const bool blended = UP && c < NC0;                      // wave-uniform
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      if (dst_off[n] < 0) continue;
      bool okn = (src_okmask >> n) & 1u;
      f32x4 v = hv[n][0];
      if constexpr (UP) {
        if (blended) {
          const float ly1 = up_ly[n], lx1 = up_lx[n], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ly0 * (lx0 * hv[n][0][e] + lx1 * hv[n][1][e]) + ly1 * (lx0 * hv[n][2][e] + lx1 * hv[n][3][e]);
        } else {
          okn = okn && (c - NC0) * 32 + ((tid + n * 256) & 7) * 4 < c_skip;
        }
      }
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float x = okn ? v[e] : 0.f; hi[e] = f2h(x); lo[e] = (f16_t)(x - (float)hi[e]); }
      *reinterpret_cast<f16x4*>(d + dst_off[n]) = hi;
      *reinterpret_cast<f16x4*>(d + LO + dst_off[n]) = lo;
    }
  };

  issue_w(0, 0, 0);
  if (NSTW == 3 && nit > 1) issue_w(0, 1, 1);              // (nit = 9 NC >= 9: K-step 1 is tap 1 of chunk 0)
  load_chunk(0);
  store_chunk(0, 0);

  f32x4 acc[4][NT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xrow = ((wm * 4) * HC + fr) * PPC + fq * 8;      // tap (0, 0) of output pixel (row wm * 4, column fr), this lane's first quad
  constexpr int growb = HC * PPC;
  const int pc0 = ((fq) ^ (fr & 7)) * 16, pc1 = ((4 + fq) ^ (fr & 7)) * 16;

  int it = 0;
  int rbuf = 0;                                            // weight stage of the current step (it % NSTW)
  constexpr int NLDV = NLD * (UP ? 4 : 1);                 // vector-memory loads of one load_chunk (blended chunks: four taps per piece)
  for (int c = 0; c < NC; ++c) {
    const unsigned char* xc = sX + (SB ? 0 : (c & 1)) * CBUF + xrow;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap, ++it) {
      if constexpr (SB) {
        if (tap == 0 && c > 0) {                           // single buffer: everybody has left chunk c - 1 -> store chunk c (loaded at tap 7), meet again
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          store_chunk(0, c);
        }
      }
      if constexpr (NSTW == 2) {
        // the weights of this step were issued one step ago.  (No counted wait here: the halo loads of the next chunk, issued at tap 7, are
        // plain register loads the compiler may order either side of the DMA builtin, so "leave NLD in flight" would not be exact.)
        wait_vmcnt<0>();
      } else {
        // three stages: everything issued during the previous step may stay in flight -- the next step's weights (NB DMA instructions, when
        // there is a next step) and, at tap 8, the next chunk's halo loads (NLDV; a skip chunk of a two-source kernel issues fewer: UP keeps
        // the full wait there)
        if (it + 1 >= nit) wait_vmcnt<0>();
        else if (tap == 8 && c + 1 < NC) { if (UP) wait_vmcnt<0>(); else wait_vmcnt<NB + NLDV>(); }
        else wait_vmcnt<NB>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's halo stores / fragment reads of the previous step
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // timing probes (cfp_debug_set key 16; results are garbage): 1 = no weight DMA after the prologue, 2 = no fragment reads / MFMAs
      if (it + NSTW - 1 < nit && !(p.probe & 1)) {
        const int tn = tap + NSTW - 1;                     // K-step it + NSTW - 1 = (chunk, tap) of the stage everybody has just finished reading
        issue_w(tn >= 9 ? c + 1 : c, tn >= 9 ? tn - 9 : tn, NSTW == 2 ? ((it + 1) & 1) : (rbuf == 0 ? 2 : rbuf - 1));
      }
      if (tap == 7 && c + 1 < NC) load_chunk(c + 1);       // lands during this step's MFMAs; stored after the next (last) tap
      if (p.probe & 2) { if (!SB && tap == 8 && c + 1 < NC) store_chunk((c + 1) & 1, c + 1); rbuf = rbuf + 1 == NSTW ? 0 : rbuf + 1; continue; }
      const unsigned char* cW = sW + rbuf * WSTAGE + (wn * NT * 16) * 128;
      rbuf = rbuf + 1 == NSTW ? 0 : rbuf + 1;
      f16x8 whi[NT], wlo[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        whi[j] = *reinterpret_cast<const f16x8*>(cW + (j * 16 + fr) * 128 + pc0);
        wlo[j] = *reinterpret_cast<const f16x8*>(cW + (j * 16 + fr) * 128 + pc1);
      }
      const int ty = tap / 3, dx = tap - ty * 3;
      const unsigned char* xt = xc + (ty * HC + dx) * PPC;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned char* xb = xt + g * growb;
        const f16x4 h0 = *reinterpret_cast<const f16x4*>(xb), l0 = *reinterpret_cast<const f16x4*>(xb + LO);
        const f16x4 h1 = *reinterpret_cast<const f16x4*>(xb + 32), l1 = *reinterpret_cast<const f16x4*>(xb + LO + 32);
        const f16x8 xhi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const f16x8 xlo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[j], xhi, acc[g][j], 0, 0, 0);
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xlo, acc[g][j], 0, 0, 0);
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xhi, acc[g][j], 0, 0, 0);
        }
      }
      if (!SB && tap == 8 && c + 1 < NC) store_chunk((c + 1) & 1, c + 1);      // (every wave is past chunk c - 1: the buffer is free)
    }
  }

  float* __restrict__ out = reinterpret_cast<float*>(p.out) + (long long)b * p.Ho * p.Wo * p.out_ld;
  const float* __restrict__ res = p.res ? reinterpret_cast<const float*>(p.res) + (long long)b * p.Ho * p.Wo * p.res_ld : nullptr;
  const int x = x0 + fr;
  with_act(p.act, [&](auto A) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n_base + (wn * NT + j) * 16 + fq * 4;
      if (n >= p.Cout) continue;
      const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int y = y0 + wm * 4 + g;
        if (!(y < p.Ho && x < p.Wo)) continue;
        const long long pix = (long long)y * p.Wo + x;
        f32x4 yv;
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = act_c<decltype(A)::value>(acc[g][j][r] * sc[r] + sh[r]);
        if (res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(res + pix * p.res_ld + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) yv[r] += rv[r];
        }
        *reinterpret_cast<f32x4*>(out + pix * p.out_ld + n) = yv;
      }
    }
  });
}

template <int NT, int WN, bool UP = false, int NSTW = 2, bool SB = false>
int launch_cx(const ConvP& p, hipStream_t s) {
  constexpr int TH = 4 * (4 / WN);
  constexpr int NPAD = NT * WN * 16;
  HaloX3P hp;
  hp.n_blocks = cdiv(p.Cout, NPAD);
  hp.QPP = p.Cin / 4; hp.dq = make_fastdiv((unsigned)hp.QPP); hp.PP = 80; hp.LO = (TH + 2) * 18 * 80;
  if ((long long)p.H * p.W * p.in_ld >= (1ll << 31)) return -1;
  if (!UP && p.Cin % 32 != 0) return -1;
  if (UP && (p.up_src == nullptr || p.up_C % 32 != 0 || p.up_C <= 0 || (p.Cin - p.up_C) % 4 != 0 || p.Cin - p.up_C < 4 ||
             (long long)p.up_H * p.up_W * p.up_ld >= (1ll << 31) || p.H < 2 || p.W < 2)) return -1;
  hp.tiles_x = cdiv(p.Wo, 16); hp.tiles_y = cdiv(p.Ho, TH);
  const long long tiles = (long long)p.B * hp.tiles_x * hp.tiles_y * hp.n_blocks;
  const size_t lds = (size_t)NSTW * NPAD * 128 + (size_t)(SB ? 1 : 2) * 2 * hp.LO;
  if (lds > 160 * 1024 || tiles >= (1ll << 31)) return -1;
  auto k = conv3x3_chunk_x3_kernel<NT, WN, UP, NSTW, SB>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), lds, s, p, hp);
  return 0;
}

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
    {8, 1},  // 8: <= 128, 16 x 16 pixels
    {5, 1},  // 9: <= 80
};
constexpr int kNumHCfg = sizeof(kHCfg) / sizeof(kHCfg[0]);

template <int NT, int WN>
int launch_hx(const ConvP& p, hipStream_t s) {
  constexpr int TH = 4 * (4 / WN);
  constexpr int NPAD = NT * WN * 16;
  constexpr int STAGES = 2;
  HaloX3P hp;
  hp.n_blocks = cdiv(p.Cout, NPAD);
  hp.QPP = p.Cin / 4;
  hp.dq = make_fastdiv((unsigned)hp.QPP);
  if ((long long)p.H * p.W * p.in_ld >= (1ll << 31)) return -1;
  int slots = hp.QPP / 2;          // 16-byte slots of one plane's pixel (QPP is even)
  if ((slots & 1) == 0) ++slots;
  hp.PP = slots * 16;
  hp.LO = (TH + 2) * 18 * hp.PP;
  hp.tiles_x = cdiv(p.Wo, 16); hp.tiles_y = cdiv(p.Ho, TH);
  const size_t halo = (size_t)2 * hp.LO;
  const long long tiles = (long long)p.B * hp.tiles_x * hp.tiles_y * hp.n_blocks;
  const size_t lds = (size_t)STAGES * NPAD * 128 + halo;
  if (lds > 160 * 1024 || tiles >= (1ll << 31)) return -1;
  auto k = conv3x3_halo_x3_kernel<NT, WN, STAGES>;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), lds, s, p, hp);
  return 0;
}

}  // namespace

int conv3x3_halo_x3_num_variants() { return kNumHCfg; }

// The problems this kernel takes: 3x3, stride 1, undilated, float32 tensors with pre-split weights, Cin a multiple of 8, shared weights.
bool conv3x3_halo_x3_takes(const ConvP& p) {
  return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.dil == 1 && p.Cin % 8 == 0 && p.Cin >= 8 && p.Cout % 4 == 0 && p.rows_per_batch == 0 &&
         p.ln_gamma == nullptr && (long long)p.H * p.W * p.in_ld < (1ll << 31);
}

// Smallest LDS footprint (bytes) of variant v for this problem, or 0 if it cannot hold the halo.
size_t conv3x3_halo_x3_lds(int v, const ConvP& p) {
  if (v < 0 || v >= kNumHCfg) return 0;
  const int th = 4 * (4 / kHCfg[v].wn), npad = kHCfg[v].nt * kHCfg[v].wn * 16;
  int slots = p.Cin / 8;
  if ((slots & 1) == 0) ++slots;
  const size_t lds = (size_t)2 * npad * 128 + (size_t)2 * (th + 2) * 18 * slots * 16;
  return lds <= 160 * 1024 ? lds : 0;
}

// v < 0: automatic tile (widest channel block that fits, 16 x 16 pixels while two workgroups still share a CU).
int conv3x3_halo_x3_launch(int v, const ConvP& p, hipStream_t s) {
  if (v < 0) {
    const int c = p.Cout;
    const int cands16[] = {0, 1, 2, 9, 8};          // 16 x 16 pixel tiles by channel capacity 16 / 32 / 64 / 80 / 128
    const int cands8[] = {7, 3, 4, 5, 6};           // 8 x 16 pixel tiles: 32 / 64 / 128 / 160 / 224
    const int cap16[] = {16, 32, 64, 80, 128}, cap8[] = {32, 64, 128, 160, 224};
    int best = -1;
    const long long px = (long long)p.B * p.Ho * p.Wo;
    // measured exceptions to "widest block, 16 x 16 pixels": thin outputs at a quarter of the full resolution fill the chip better with
    // 8 x 16 pixel tiles (153600 px x 32 ch: 17.2 vs 19.6 us in flight); 224 channels as two 128-channel blocks (33 vs 39 us)
    if (c <= 32 && px < 300000 && conv3x3_halo_x3_lds(7, p)) best = 7;
    if (c > 160 && c <= 256 && conv3x3_halo_x3_lds(4, p)) best = 4;
    if (c > 128 && c <= 160 && px < 30000 && conv3x3_halo_x3_lds(3, p)) best = 3;      // a single image's 19200 px x 160: three 64-channel blocks, 19.7 vs 24.3 us (one 160-channel block)
    for (int i = 0; i < 5 && best < 0; ++i)
      if (c <= cap16[i] && conv3x3_halo_x3_lds(cands16[i], p) && conv3x3_halo_x3_lds(cands16[i], p) <= 80 * 1024) best = cands16[i];
    for (int i = 0; i < 5 && best < 0; ++i)
      if (c <= cap8[i] && conv3x3_halo_x3_lds(cands8[i], p)) best = cands8[i];
    if (best < 0) {      // wider than any block: several channel blocks per tile
      best = conv3x3_halo_x3_lds(4, p) ? 4 : (conv3x3_halo_x3_lds(3, p) ? 3 : -1);
    }
    if (best < 0) return -1;
    v = best;
  }
  switch (v) {
    case 0: return launch_hx<1, 1>(p, s);
    case 1: return launch_hx<2, 1>(p, s);
    case 2: return launch_hx<4, 1>(p, s);
    case 3: return launch_hx<2, 2>(p, s);
    case 4: return launch_hx<4, 2>(p, s);
    case 5: return launch_hx<5, 2>(p, s);
    case 6: return launch_hx<7, 2>(p, s);
    case 7: return launch_hx<1, 2>(p, s);
    case 8: return launch_hx<8, 1>(p, s);
    case 9: return launch_hx<5, 1>(p, s);
    // the chunk-pipelined form (Cin % 32 == 0): 20 + ...
    case 20: return launch_cx<2, 1>(p, s);      // <= 32 channels, 16 x 16 pixels
    case 21: return launch_cx<4, 1>(p, s);      // <= 64
    case 22: return launch_cx<8, 1>(p, s);      // <= 128
    case 23: return launch_cx<4, 2>(p, s);      // <= 128, 8 x 16 pixels
    case 24: return launch_cx<2, 2>(p, s);      // <= 64, 8 x 16 pixels
    case 25: return launch_cx<8, 2>(p, s);      // <= 256, 8 x 16 pixels
    // three weight stages with counted waits (round 5)
    case 30: return launch_cx<2, 1, false, 3>(p, s);
    case 31: return launch_cx<4, 1, false, 3>(p, s);
    case 32: return launch_cx<8, 1, false, 3>(p, s);
    case 33: return launch_cx<4, 2, false, 3>(p, s);
    case 34: return launch_cx<2, 2, false, 3>(p, s);
    case 35: return launch_cx<8, 2, false, 3>(p, s);
    // one chunk buffer: two workgroups per CU for the 128-channel tile (round 5)
    case 36: return launch_cx<4, 2, false, 2, true>(p, s);      // <= 128 channels, 8 x 16 pixels, 61 KB
    case 37: return launch_cx<8, 2, false, 2, true>(p, s);      // <= 256 channels, 8 x 16 pixels, 93 KB (one per CU; for comparison)
    case 38: return launch_cx<4, 1, false, 2, true>(p, s);      // <= 64 channels, 16 x 16 pixels, 68 KB
    case 39: return launch_cx<4, 2, false, 3, true>(p, s);      // 36 with three weight stages: 77 KB, still two per CU
    case 44: return launch_cx<2, 2, false, 2, true>(p, s);      // <= 64 channels, 8 x 16 pixels, 45 KB: three per CU
    case 45: return launch_cx<1, 2, false, 2, true>(p, s);      // <= 32 channels, 8 x 16 pixels, 37 KB: four per CU
    case 46: return launch_cx<2, 1, false, 2, true>(p, s);      // <= 32 channels, 16 x 16 pixels, 60 KB: two per CU
    case 43: return launch_cx<4, 1, false, 3, true>(p, s);      // 38 with three weight stages: 76 KB
    // two sources (cfp_upsample_cat_conv3x3): 8 x 16 pixel tiles (six pieces per thread and chunk: 96 registers of taps in flight)
    case 40: return launch_cx<1, 2, true>(p, s);      // <= 32 output channels per workgroup
    case 41: return launch_cx<2, 2, true>(p, s);      // <= 64
    case 42: return launch_cx<4, 2, true>(p, s);      // <= 128
    default: return -3;
  }
}
