/*
 * libcfpnet_hip.so -- C ABI of the MI355X (gfx950) CFPNet hot path.
 *
 * The reference (denyingmxd/CFPNet) is pure PyTorch: its "operator interface" for this path is
 * the set of torch ops `Deltar.forward` issues (src/models/deltar.py:34-67).  Each entry point
 * below replaces one family of those ops with a hand-written HIP kernel; the comment on every
 * declaration names the reference lines it stands in for.  The host side
 * (cfpnet_amd/engine.py, via ctypes -- see INTEGRATION.md) strings them together.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer; nothing is allocated or freed here
 *   - activations are NHWC ("tokens": [rows = B*H*W][channels]) with an explicit row pitch
 *     `*_ld` in ELEMENTS, so producers can write straight into a channel slice of a wider
 *     (concatenation) buffer; channel counts, pitches and slice offsets are multiples of
 *     8 elements (bf16) / 4 elements (f32) so that every access is a 16-byte vector
 *   - dtype: CFP_F32, CFP_BF16 or CFP_F16 (IEEE half, saturating stores) storage; all arithmetic accumulates in f32
 *   - all calls are asynchronous on `stream` (a hipStream_t), stateless and re-entrant
 *   - return 0 on success, a negative CFP_E* code otherwise; `cfp_last_error()` gives the
 *     message (thread-local).  No C++ exception crosses this boundary.
 */
#ifndef CFPNET_HIP_H
#define CFPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cfp_stream_t; /* hipStream_t */

enum { CFP_OK = 0, CFP_EINVAL = -1, CFP_ESHAPE = -2, CFP_EHIP = -3 };
enum { CFP_F32 = 0, CFP_BF16 = 1, CFP_F16 = 2 };
/* CFP_F32X3: float32 STORAGE with split-precision ("f16x3") matrix math -- accepted where a call plans or packs for that mode
 * (cfp_conv2d_plan, cfp_conv2d_ws_bytes, cfp_se_gate_fold's output format); kernels that do no matrix math take CFP_F32 tensors as they are. */
enum { CFP_F32X3 = 3 };
enum { CFP_TOF_SAMPLE_UNIFORM = 0, CFP_TOF_SAMPLE_ICDF = 1 };   /* sample_point_from_hist_parallel: --sample_uniform / default */
enum { CFP_ACT_NONE = 0, CFP_ACT_RELU = 1, CFP_ACT_LRELU = 2, CFP_ACT_SILU = 3, CFP_ACT_GELU = 4, CFP_ACT_SIGMOID = 5 };

int cfp_version(void);
const char* cfp_last_error(void);

/* Dense convolution / linear layer as an implicit GEMM on the matrix cores:
 *   out[m, n] = act( (sum_k A[m,k] * w[n,k]) * scale[n] + shift[n] ) + residual[m, n]
 * with m = (b, ho, wo), k = (kh, kw, ci), w packed [Cout][KH][KW][Cin].
 * Replaces nn.Conv2d 3x3 / 1x1 (+ folded BatchNorm + activation + skip add):
 *   decoder.py:43-58,70-80 (UpSampleBN, conv0..conv4), decoder.py:13 (depth_head.conv3x3),
 *   transformer.py:197-200,240-244 (DAPM convs), transformer.py:132 (GSA sr conv),
 *   nn.Linear in transformer.py:24-36 / convnext.py:32-34 / encoder.py:10-12 (KH=KW=1, H=1),
 *   and the encoder's conv_stem / conv / conv_exp / conv_pw / conv_pwl (encoder.py:57-69).
 * scale/shift may be NULL (1 / 0).  residual may be NULL.  Cin % 8 == 0 (bf16) or % 4 (f32). */
int cfp_conv2d_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                    const void* residual, int res_ld, void* out, int out_ld,
                    int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                    int pad_t, int pad_l, int Ho, int Wo, int act, int dtype,
                    void* ws, size_t ws_bytes, cfp_stream_t stream);

/* cfp_conv2d_nhwc with two more fusions (same reference ops, fewer launches):
 *   - ln_gamma/ln_beta != NULL: LayerNorm over the Cout axis (eps = ln_eps) applied after
 *     scale/shift/act, the residual is added AFTER the LayerNorm:
 *         out = LN(act(conv * scale + shift)) * gamma + beta + residual
 *     (transformer.py:63,68-70: merge -> norm1, mlp -> norm2 -> + x).  Fused into the GEMM
 *     epilogue when a tile spans exactly Cout channels (bf16, Cout in {16,32,64,128}), otherwise
 *     run as a second kernel on `out` in place.  CFP_CONV_X3 launches whose K is split run it inside
 *     the finishing sum of the splits (Cout / 4 a power of two <= 64; round 5).
 *   - per_image_weights != 0: `w` holds B weight matrices [B][Cout][K]; image b of the batch uses
 *     matrix b.  Used for the squeeze-excite gate folded into the project conv of the encoder's
 *     inverted-residual blocks (x * gate[b,:]) @ W^T == x @ (W * gate[b,:])^T, see cfp_se_fold.
 *   - per_image_weights & CFP_CONV_W2 (16-bit pointwise layers with shared weights): TWO-TERM weights.  Every row of `w` is
 *     [hi | lo] with hi = round16(W), lo = round16(W - hi), each half zero-padded to a multiple of 64 elements; the kernel runs
 *     the K loop over both halves against the same activations, so the layer sees its weights with ~22 (fp16) / ~16 (bf16)
 *     significant bits instead of 11 / 8.  Weight rounding of the pointwise layers is 0.8e-3 of the fp16 engine's 0.9e-3
 *     relative-L1 error (profiles/r2_precision_budget.md). */
/*   - per_image_weights & CFP_CONV_X3 (dtype CFP_F32): split-precision matrix math on float32 tensors.  Every value is taken as
 *     hi + lo (two IEEE halves, ~21 significant bits) and the product as A_hi W_hi + A_hi W_lo + A_lo W_hi on
 *     v_mfma_f32_16x16x32_f16 with float32 accumulation: 3/16 of the 16-bit matrix rate instead of the 1/16 of the float32 MFMA,
 *     product error ~2^-21.  `w` is then the PRE-SPLIT operand written by cfp_pack_w_x3 ([Cout][ceil(K/32)][hi(32) | lo(32)] halves;
 *     with CFP_CONV_PER_IMAGE: B such matrices, as cfp_se_gate_fold writes them for dtype CFP_F32X3).  This is the mode whose results
 *     stay inside the reference tolerance (1e-3 relative L1 on the depth map, /root/reference/src/models/deltar.py:34-67 in float32)
 *     for every weight family -- the default of the drop-in boundary. */
enum { CFP_CONV_PER_IMAGE = 1, CFP_CONV_W2 = 2, CFP_CONV_IN_FLIGHT = 4, CFP_CONV_X3 = 8, CFP_CONV_WS_TICKETS = 16 };   /* bits of cfp_conv2d_nhwc_ex's `per_image_weights` argument.
 * CFP_CONV_IN_FLIGHT: a hint -- this launch will run beside launches of other batches (several captured forwards in flight): the tile is then
 * chosen for the resources it holds rather than for its own latency (larger tiles).  Results do not depend on it: every tile walks K in the same order.
 * CFP_CONV_WS_TICKETS (CFP_CONV_X3 launches): the first CFP_CONV_TICKET_BYTES of `ws` are a TICKET AREA -- all zero when the workspace is first handed
 * over, and left all zero by every launch -- and the split-K slabs follow it (ws_bytes >= CFP_CONV_TICKET_BYTES + cfp_conv2d_ws_bytes(...)).  When K is
 * split, the workgroup that reaches an output tile last then finishes it (slabs summed in split order, scale / shift / act / residual), instead of a
 * second launch doing so: one kernel less per split layer, the same bits.  Launches that share a workspace must be ordered (one stream), as they must
 * for the slabs. */
#define CFP_CONV_TICKET_BYTES 4096
int cfp_conv2d_nhwc_ex(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                       const void* residual, int res_ld, void* out, int out_ld,
                       int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                       int pad_t, int pad_l, int Ho, int Wo, int act, int dtype,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, int per_image_weights,
                       void* ws, size_t ws_bytes, cfp_stream_t stream);

/* Weights [rows][K] float32 (K = KH*KW*Cin in (kh, kw, ci) order; rows = Cout, or B*Cout for per-image weights) -> the pre-split
 * operand of the CFP_CONV_X3 kernels: [rows][ceil(K/32)][hi(32) | lo(32)] IEEE halves in the kernels' lane order, zero padded to whole
 * 32-channel K-steps.  cfp_pack_w_x3_elems = halves in that buffer.  Replaces nothing in the reference (a weight re-layout, like the
 * [Cout][KH][KW][Cin] packing of every other mode); the layers are the nn.Conv2d / nn.Linear calls listed at cfp_conv2d_nhwc. */
size_t cfp_pack_w_x3_elems(long long rows, int K);
int cfp_pack_w_x3(const float* w, void* out, long long rows, int K, cfp_stream_t stream);

/* Scratch cfp_conv2d_nhwc wants for (M = B*Ho*Wo, Cout, K = KH*KW*Cin): non-zero only for layers it
 * runs split-K (few output tiles, long K: the GSA sr convs, the 1/32-scale pointwise convs).  With
 * ws == NULL or too small the layer runs un-split (same result up to f32 re-association). */
size_t cfp_conv2d_ws_bytes(int M, int Cout, int K, int dtype);

/* Kernel plan cfp_conv2d_nhwc uses for a problem (per-kernel accounting in bench.py and
 * tools/conv_bench.py).  *variant: 0..3 = first-generation tiles 256x16 / 256x32 / 128x64 /
 * 128x128 (f32); 100 + v = second-generation (bf16, LDS-DMA staged) variant v; 200 + v = direct 3x3
 * (LDS halo tile) variant v; 400 + v = f16x3 implicit-GEMM variant v, 500 = the f16x3 whole-depth-halo 3x3 kernel (dtype CFP_F32X3: float32
 * storage, CFP_CONV_X3).  *splits = K-splits.  KH/stride describe the filter (K = KH*KH*Cin);
 * rows_per_batch > 0 describes a per_image_weights call (B images of rows_per_batch rows). */
int cfp_conv2d_plan(int M, int Cout, int K, int KH, int stride, int dtype, int rows_per_batch, int B, int* variant,
                    int* splits);

/* First-generation tile choice for (M, Cout): 0 = 256x16, 1 = 256x32, 2 = 128x64, 3 = 128x128. */
int cfp_conv2d_variant(int M, int Cout);

/* Test/benchmark knobs, not for production use (process-global, not thread-safe):
 * key 0 = force second-generation variant v, or 200 + v = direct 3x3 variant v (-1 = automatic), key 1 = force K-splits (-1 = automatic),
 * key 2 = 1 routes bf16 through the first-generation kernel; keys 3 / 4 = depthwise 3x3 channel vectors per
 * workgroup (8 / 16) and output rows per strip (0 = automatic); key 5 = 1 forces the VALU depthwise kernel; keys 6-21 are
 * listed in README.md ("Kernel choices") and at the dispatch in csrc/conv_igemm.hip. */
int cfp_debug_set(int key, int value);

/* Depthwise 3x3 convolution, stride 1/2, explicit (TF-"SAME", possibly asymmetric) padding, fused
 * BatchNorm scale/shift + activation.  w packed [9][C].  HBM-bandwidth-bound.
 * Replaces timm InvertedResidual.conv_dw + bn2 + SiLU (encoder.py:66-69, 24 convs).
 * Kernels: float32 storage -> dw3x3_rows_kernel (round 5: register-sliding rows, no LDS; csrc/dw3x3_rows.hip; SiLU / ReLU / none), 16-bit storage with
 * C % 16 == 0 -> dw3x3_slide_kernel (diagonal-weight MFMA, csrc/dw3x3_slide.hip), otherwise the LDS-strip kernel of csrc/dwconv.hip. */
int cfp_dwconv3x3_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                       void* out, int out_ld, int B, int H, int W, int C, int stride, int pad_t, int pad_l,
                       int Ho, int Wo, int act, int dtype, cfp_stream_t stream);

/* cfp_dwconv3x3_nhwc that ALSO emits sums of the stored output per (image, row strip, channel):
 *   partial[b][s][c] (f32), s < cfp_dwconv3x3_strips(B, Ho, Wo, C, stride, dtype)
 * so the H*W mean that timm's SqueezeExcite takes of this tensor (x.mean((2,3))) needs no extra
 * pass: cfp_se_hidden(partial, nsplit = strips, ...) consumes it directly.  Strip order and the
 * in-strip reduction order are fixed: the sums are run-to-run deterministic. */
int cfp_dwconv3x3_strips(int B, int Ho, int Wo, int C, int stride, int dtype);
/* The number of slots the float32-storage LAUNCH will write for these full arguments (0: it does not take the shape and the older kernel's
 * own count applies).  Host-side only, no GPU work: tests/test_abi.py sweeps shapes and checks that it equals cfp_dwconv3x3_strips, which is
 * asked WITHOUT the input extent -- a mismatch would be an out-of-bounds write into `partial` (found and fixed in round 5). */
int cfp_dwconv3x3_launch_slots(int B, int H, int W, int Ho, int Wo, int C, int stride, int in_ld, int out_ld, int dtype);
int cfp_dwconv3x3_sum_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift,
                           void* out, int out_ld, float* partial, int B, int H, int W, int C, int stride,
                           int pad_t, int pad_l, int Ho, int Wo, int act, int dtype, cfp_stream_t stream);

/* Large-kernel depthwise convolution (odd k <= 31; 7 / 15 / 31 have dedicated kernels), stride 1,
 * "same" padding, fused bias + BatchNorm + ReLU.  w packed [C][kx][ky] as f32 (per-channel
 * contiguous, column-major taps: the kernel slides vertically and fetches weights through the
 * scalar cache).  Vector-FMA-bound.  Replaces Block14.dwconv2 + bn1 + relu (convnext.py:30,45-47). */
int cfp_dwconv_large_nhwc(const void* in, int in_ld, const float* w, const float* scale, const float* shift,
                          void* out, int out_ld, int B, int H, int W, int C, int k, int act, int dtype,
                          cfp_stream_t stream);

/* cfp_dwconv_large_nhwc on the matrix cores (bf16, k in {7, 15, 31}): every kernel row is a banded Toeplitz product,
 * 62 MFMAs per 16x16 output tile of one channel for k = 31.  `toeplitz` holds the bands in MFMA B-operand layout,
 *   toeplitz[c][ky][h][lane][e] = w[c][ky][kx],  kx = 32 h + 8 (lane >> 4) + e - (LM - halo) - (lane & 15)   (0 outside [0, k))
 * with halo = (k-1)/2, LM = halo rounded up to 8, h < NH = ceil((16 + LM + halo) / 32); cfp_dwconv_large_toeplitz_elems
 * gives its size in elements.  It is built once per weight tensor on the host (cfpnet_amd/engine.py). */
size_t cfp_dwconv_large_toeplitz_elems(int C, int k);
/* The same table built on the device from float32 weights [C][k][k] (training: the master weights change every step); flip != 0
 * gives the table of the 180-degree-rotated kernel (the data gradient).  out: cfp_dwconv_large_toeplitz_elems(C, k) 16-bit elements. */
int cfp_dwconv_large_toeplitz(const float* w, void* out, int C, int k, int flip, int dtype, cfp_stream_t stream);
int cfp_dwconv_large_mfma_nhwc(const void* in, int in_ld, const void* toeplitz, const float* scale, const float* shift,
                               void* out, int out_ld, int B, int H, int W, int C, int k, int act, int dtype,
                               cfp_stream_t stream);

/* Per-(batch, channel) sums over H*W, split in `nsplit` row slices: partial[b][s][c] (f32).
 * Consumers divide by H*W.  Replaces x.mean((2,3)) in the SE block and `.mean([2,3])` of
 * DepthRegression (decoder.py:24-25; conv1x1 and mean commute). */
int cfp_channel_sum(const void* in, int in_ld, float* partial, int B, int HW, int C, int nsplit, int dtype,
                    cfp_stream_t stream);

/* Squeeze-excite, first half: mean -> FC(C->R) + bias -> SiLU; hidden[b][r] f32.  w_reduce [R][C] f32.
 * Replaces timm SqueezeExcite.conv_reduce + act (inside the encoder.py:66-69 blocks). */
int cfp_se_hidden(const float* partial, int nsplit, float inv_hw, const float* w_reduce, const float* b_reduce,
                  float* hidden, int B, int C, int R, cfp_stream_t stream);

/* Squeeze-excite, second half, in place: x[b,hw,c] *= sigmoid(hidden[b] . w_expand_t[:, c] + b_expand[c]).
 * w_expand_t is the expand weight TRANSPOSED to [R][C] f32.  Replaces conv_expand + sigmoid gate + multiply. */
int cfp_se_scale(void* x, int ld, const float* hidden, const float* w_expand_t, const float* b_expand,
                 int B, int HW, int C, int R, int dtype, cfp_stream_t stream);

/* Squeeze-excite gate folded into the following project conv's weights:
 *   w_out[b][n][c] = w_proj[n][c] * sigmoid(hidden[b] . w_expand_t[:, c] + b_expand[c])
 * (x * gate[b,:]) @ W^T == x @ (W * gate[b,:])^T, so conv_expand + sigmoid + the gating multiply of
 * SqueezeExcite and conv_pwl become one per-image-weights GEMM (cfp_conv2d_nhwc_ex) with no pass
 * over the expanded activation.  hidden from cfp_se_hidden; w_expand_t [R][C] f32; w_proj [Cout][C]
 * and w_out [B][Cout][C] in `dtype`. */
int cfp_se_fold(const void* w_proj, void* w_out, const float* hidden, const float* w_expand_t, const float* b_expand,
                int B, int Cout, int C, int R, int dtype, cfp_stream_t stream);

/* Front half of a stride-1 inverted-residual block in one launch (csrc/mbconv.hip): conv_pw 1x1 (Cin -> mid) + BN1 + SiLU ->
 * conv_dw 3x3 (depthwise, padding 1) + BN2 + SiLU (timm InvertedResidual as the reference builds it, encoder.py:66-69) + the
 * per-tile channel sums squeeze-excite needs.  The expanded tensor never reaches HBM.
 *   x [B,H,W,x_ld] (Cin channels); wpw [mid][KP + 8] 16-bit with KP = Cin rounded up to 32 (cfp_mbconv_plan), zero padding behind
 *   the Cin real columns: the LDS image layout of the weights; s1 / t1, s2 / t2 [mid] f32: BatchNorm folded to scale / shift;
 *   wdw [9][mid] 16-bit (tap-major, like cfp_dwconv3x3_nhwc); out [B,H,W,out_ld] (mid channels); partial (may be NULL)
 *   [B][tiles_per_image][mid] f32 channel sums per spatial tile (tiles_per_image from cfp_mbconv_plan; feed cfp_se_gate_fold with
 *   nsplit = tiles_per_image).  bf16 / f16 only; Cin % 8 == 0, mid % 16 == 0.  cfp_mbconv_plan returns CFP_ESHAPE if the shape
 *   does not fit (the engine then runs cfp_conv2d_nhwc + cfp_dwconv3x3_sum_nhwc). */
int cfp_mbconv_plan(int B, int H, int W, int Cin, int mid, int* tiles_per_image, int* KP);
int cfp_mbconv_expand_dw(const void* x, int x_ld, const void* wpw, const float* s1, const float* t1, const void* wdw,
                         const float* s2, const float* t2, void* out, int out_ld, float* partial, int B, int H, int W, int Cin,
                         int mid, int dtype, cfp_stream_t stream);

/* cfp_se_hidden + cfp_se_fold in one launch (mean -> FC -> SiLU -> FC -> sigmoid -> per-image project weights),
 * structured for latency: this is what the engine calls between the depthwise conv and the project conv of every
 * inverted-residual block.  Same arguments as the two calls it replaces; C <= 2048, R <= 64. */
int cfp_se_gate_fold(const float* partial, int nsplit, float inv_hw, const float* w_reduce, const float* b_reduce,
                     const float* w_expand_t, const float* b_expand, const void* w_proj, void* w_out,
                     int B, int Cout, int C, int R, int dtype, cfp_stream_t stream);

/* cfp_dwconv3x3_nhwc + the squeeze-excite reduce FC applied to the workgroup's channel sums (16-bit storage with C % 16 == 0, or
 * float32 storage -- the default f16x3 mode -- with C % 8 == 0; R <= 64):
 * timm InvertedResidual conv_dw -> bn2 -> act, and of SqueezeExcite (x.mean((2, 3)) -> conv_reduce) the part that is LINEAR in the
 * sums: every workgroup (image b, row strip, channel block) writes
 *   hpart[b][k][r] = sum_{c in block} w_reduce[r][c] * (sum of its stored output pixels of channel c),   k = strip * blocks + block,
 * K = cfp_dwconv3x3_se_parts(...) partials per image (0: shape / dtype not supported).  w_reduce [R][C] f32.  cfp_se_gate_fold2 adds
 * them, scales by 1 / (Ho * Wo) and finishes the block.  Reference: encoder.py:66-69 (timm tf_efficientnetv2_b3 blocks[3..5]). */
int cfp_dwconv3x3_se_parts(int B, int Ho, int Wo, int C, int stride, int dtype);
int cfp_dwconv3x3_se_nhwc(const void* in, int in_ld, const void* w, const float* scale, const float* shift, void* out, int out_ld,
                          const float* w_reduce, int R, float* hpart, int B, int H, int W, int C, int stride, int pad_t, int pad_l,
                          int Ho, int Wo, int act, int dtype, cfp_stream_t stream);

/* The squeeze-excite tail when the depthwise kernel has already applied the reduce FC to its channel sums (the reduce layer is
 * linear in them): hpart [B][K][R] f32 partial dot products (cfp_dwconv3x3_se_nhwc), added in k order;
 *   hidden = silu(inv_hw * sum_k hpart + b_reduce);  gate = sigmoid(hidden . w_expand_t + b_expand);  w_out[b][n][c] = w_proj[n][c] * gate[c]
 * with w_proj the FLOAT32 project weights [Cout][C] (the folded weights are rounded to `dtype` once) and w_out [B][Cout][C] in `dtype`.
 * Replaces timm SqueezeExcite + the gate multiply as cfp_se_gate_fold does; one full-chip launch.  C % 8 == 0, R <= 64. */
int cfp_se_gate_fold2(const float* hpart, int K, float inv_hw, const float* b_reduce, const float* w_expand_t, const float* b_expand,
                      const float* w_proj, void* w_out, int B, int Cout, int C, int R, int dtype, cfp_stream_t stream);

/* x[b, hw, c] *= gate[b, c] in place. */
int cfp_scale_channels(void* x, int ld, const float* gate, int B, int HW, int C, int dtype, cfp_stream_t stream);

/* Row LayerNorm over C with affine, optional residual add afterwards:
 *   out[r,:] = LN(in[r,:]) * gamma + beta (+ residual[r,:])
 * Replaces nn.LayerNorm in transformer.py:38-39,63,68,70 and convnext.py:31,49. */
int cfp_layernorm(const void* in, int in_ld, const float* gamma, const float* beta, float eps,
                  const void* residual, int res_ld, void* out, int out_ld, int rows, int C, int dtype,
                  cfp_stream_t stream);

/* Linear attention, reduction half (attention.py:31-43):  for every key group g and head h
 *   KV[g,h] = sum_s (elu(K_s)+1)^T (V_s / v_length),   Ksum[g,h] = sum_s (elu(K_s)+1)
 * Keys live on an [NB, Hk, Wk] grid (row pitch ld); group (b, gy, gx) owns the th x tw tile
 * clipped to [cy0,cy1) x [cx0,cx1).  With count_pad != 0, tile positions outside the grid count
 * as zero-feature tokens (K = 1, V = 0): the zero-padded windows of transformer.py:101-107.
 * Covers hist2image (tile 1x16), LSA windows, GSA (one tile per batch) and DAPM (inside rectangle).
 * ws: f32 scratch of cfp_attn_kv_ws_floats() floats. */
size_t cfp_attn_kv_ws_floats(int NB, int Hk, int Wk, int th, int tw, int heads, int d);
int cfp_attn_kv_reduce(const void* k, int k_ld, const void* v, int v_ld, float* kv, float* ksum, float* ws,
                       int NB, int Hk, int Wk, int th, int tw, int cy0, int cy1, int cx0, int cx1,
                       int count_pad, float v_length, int heads, int d, int dtype, cfp_stream_t stream);

/* Linear attention, query half (attention.py:48-49):
 *   out[q,h,:] = (Q_q,h . KV[g(q),h]) / (Q_q,h . Ksum[g(q),h] + eps) * v_length,  Q = elu(q)+1
 * Queries live on an [NB, Hq, Wq] grid; g(q) = (b, y / qth, x / qtw).  Queries inside the
 * exclusion rectangle [ey0,ey1) x [ex0,ex1) get 0 (DAPM: only outside-zone tokens receive a
 * message, transformer.py:233-234). */
int cfp_attn_apply(const void* q, int q_ld, const float* kv, const float* ksum, void* out, int out_ld,
                   int NB, int Hq, int Wq, int qth, int qtw, int ey0, int ey1, int ex0, int ex1,
                   float v_length, float eps, int heads, int d, int dtype, cfp_stream_t stream);

/* Fused tail of a LoFTR encoder layer, one launch (bf16 only; D in {32,64,128}, heads in {4,8}):
 *   msg = cfp_attn_apply(q, kv, ksum)                                   attention.py:48-49
 *   y1  = LayerNorm(msg @ w_merge^T; ln1)                               transformer.py:63
 *   h   = relu([x | y1] @ w_mlp0^T)                                     transformer.py:66-67 (mlp.0 + ReLU)
 *   out = LayerNorm(h @ w_mlp2^T; ln2) + x                              transformer.py:67-71
 * q, x, out: [NB*Hq*Wq rows] with pitches q_ld / x_ld / out_ld; the query -> key-group map is that of
 * cfp_attn_apply (g = (b, y / qth, x / qtw)); no exclusion rectangle.  w_merge [D][D], w_mlp0 [2D][2D],
 * w_mlp2 [D][2D], all K-contiguous bf16.  w_q (optional, [D][D]): when given, q may be NULL and the kernel computes
 * q = x @ w_q^T (transformer.py:45) for its own rows, rounded to the storage type like a stored q.  Intermediates stay in LDS; rounding points (bf16 after the
 * apply, before each LayerNorm and after the ReLU) are those of the unfused sequence. */
int cfp_loftr_tail(const void* q, int q_ld, const float* kv, const float* ksum, const void* x, int x_ld,
                   void* out, int out_ld, const void* w_q, const void* w_merge, const void* w_mlp0, const void* w_mlp2,
                   const float* ln1_g, const float* ln1_b, const float* ln2_g, const float* ln2_b, float ln_eps,
                   int NB, int Hq, int Wq, int qth, int qtw, float v_length, float eps, int heads, int D,
                   int dtype, cfp_stream_t stream);

/* Bilinear resampling (align_corners=True) of a rectangle of an NHWC map into a rectangle of
 * another one.  The source rectangle may overhang the source map (reads 0 there: F.pad in
 * fusion.py:136).  If zone_valid != NULL, source texel (y,x) is multiplied by
 * zone_valid[b][(y/p1)*zn + x/p2] (fusion.py:144).  accumulate: 0 dst = v, 1 dst += v.
 * Only destination pixels inside the destination map are written.
 * Replaces F.interpolate in decoder.py:56 and fusion.py:141,148, the crop of fusion.py:138, the
 * zone regrouping of fusion.py:142,147,151 (which is pure addressing) and the masked
 * scatter-add of fusion.py:154-157. */
int cfp_resize_bilinear(const void* src, int src_ld, int Hs, int Ws, int sy0, int sx0, int sh, int sw,
                        void* dst, int dst_ld, int Hd, int Wd, int dy0, int dx0, int dh, int dw,
                        const uint8_t* zone_valid, int zn, int p1, int p2, int accumulate,
                        int B, int C, int dtype, cfp_stream_t stream);

/* out[r,:] = in[r,:] + table[((r / W) % H + oy) * Wt + (r % W) + ox, :]   (table f32 [*, C])
 * Positional encodings: fusion.py:92-96 (H x W window of the [Hmax*Wmax, D] table) and
 * fusion.py:123-124 (H=1, W=Wt=16). */
int cfp_add_rowtable(const void* in, int in_ld, const float* table, void* out, int out_ld, int rows, int C,
                     int H, int W, int Wt, int oy, int ox, int dtype, cfp_stream_t stream);

/* UpSampleBN's first half in ONE launch (decoder.py:51-58): F.interpolate(low, size=(H, W), mode="bilinear", align_corners=True) ->
 * torch.cat([up, skip], dim=1) -> conv3x3 (padding 1) + folded BatchNorm (scale / shift) + activation.  low [B,Hs,Ws,low_ld] (Cup channels),
 * skip [B,H,W,skip_ld] (Cskip channels), w [Cout][3][3][Cup + Cskip] (the concatenation's channel order), out [B,H,W,out_ld].
 * The upsampled tensor and the concatenation are never materialised: the direct 3x3 kernel computes the upsampled channel chunks into its
 * LDS halo tile (four taps blended in float32, rounded to `dtype` like a stored tensor) and fetches the skip chunks.  Bit-identical to
 * cfp_resize_bilinear + cfp_conv2d_nhwc.  bf16 / f16, Cup % 64 == 0, Cskip % 8 == 0, Cout % 8 == 0.
 * dtype = CFP_F32X3 (round 5): float32 tensors, f16x3 matrix math -- the chunk-pipelined kernel with two sources (conv3x3_halo_x3.hip); `w` is then
 * the pre-split operand of cfp_pack_w_x3 over the PADDED channel axis [Cout][9][Cup + 32 * ceil(Cskip / 32)] (zero weights on the padding);
 * Cup % 32 == 0, Cskip % 4 == 0, Cout % 4 == 0.  Equal to the pair it replaces to float32 round-off (another summation order); measured slower than
 * that pair at every decoder stage (profiles/r5_up_bench_x3.txt), so the engine takes it only under CFP_UP_FUSED_X3. */
int cfp_upsample_cat_conv3x3(const void* low, int low_ld, int Hs, int Ws, int Cup, const void* skip, int skip_ld, int Cskip,
                             const void* w, const float* scale, const float* shift, void* out, int out_ld, int B, int H, int W,
                             int Cout, int act, int dtype, cfp_stream_t stream);

/* Strided row copy out[r, 0:C] = in[r, 0:C]. */
int cfp_copy_rows(const void* in, int in_ld, void* out, int out_ld, int rows, int C, int dtype, cfp_stream_t stream);
/* Two such copies in one launch (a channel concatenation [a | b] -> out, or its backward split): copy k moves `rows` rows of Ck channels. */
int cfp_copy_rows2(const void* in0, int in0_ld, void* out0, int out0_ld, int C0, const void* in1, int in1_ld, void* out1, int out1_ld,
                   int C1, int rows, int dtype, cfp_stream_t stream);

/* rgb f32 NCHW [B,3,H,W] -> NHWC [B,H,W,8] (channels 3..7 zero) in `dtype`. */
int cfp_rgb_to_nhwc8(const float* rgb, void* out, int B, int H, int W, int dtype, cfp_stream_t stream);
/* The same for the 16-bit storage types with the input kept EXACT: channels 0-2 = rgb rounded to `dtype`, channels 3-5 = what the
 * rounding lost (rgb - hi, rounded), 6-7 = 0.  With the stem's weight rows repeating the three real channels in slots 3-5 the
 * float32 accumulators see w * (hi + lo): the float32 image the reference feeds its encoder (encoder.py:71-73), at no extra cost
 * (the stem GEMM pads every tap to 8 channel slots anyway).  bf16 / f16 only. */
int cfp_rgb_to_nhwc8_hilo(const float* rgb, void* out, int B, int H, int W, int dtype, cfp_stream_t stream);
/* f32 scalars [rows] -> [rows][8] with the value in column 0 (ToF sample depths, deltar.py:40). */
int cfp_scalar_to_rows8(const float* in, void* out, int rows, int dtype, cfp_stream_t stream);

/* ToF histogram encoder in one launch (csrc/hist_encoder.hip): HistogramEncoder.forward, encoder.py:45-50 = three HistExtractors
 * (:31-35) of three [Conv1d k=1 -> BatchNorm1d -> ReLU] layers each (PointNetEncoder, :17-24), applied to every sample point
 * independently.  hist [R] f32 (device): the R = B*Z*16 sample depths (deltar.py:40).  blob (device, f32): the parameters of the 9
 * layers; layout (HOST, 45 ints): per layer (w_off, scale_off, shift_off, cin, cout), offsets in floats into blob: W [cout][cin]
 * row-major, scale / shift [cout] = BatchNorm (+ conv bias) folded to y = relu(scale * (W x) + shift).  cin of layer 0 is 1, widths
 * are multiples of 16 up to 128.  out0 / out1 / out2: the activations after layers 3, 6, 9, [R][cout] in `dtype` (all arithmetic
 * is float32 whatever the storage type: v_mfma_f32_16x16x4_f32 on float32 activations held in LDS).  pe0 / pe1 / pe2 (device f32
 * [n_pe][cout], each may be NULL): `positional_encodings2` of the fusion block that consumes the tap, added on the way out with
 * row = sample index % n_pe (fusion.py:123-125: `feat1 + positional_encodings2`, n_pe = zone_sample_num). */
int cfp_hist_encoder(const float* hist, const float* blob, const int* layout, void* out0, void* out1, void* out2,
                     const float* pe0, const float* pe1, const float* pe2, int n_pe, int R, int dtype, cfp_stream_t stream);

/* LKPM tail in one launch (csrc/loftr_tail.hip): Block14.forward after its depthwise conv + BatchNorm + ReLU (convnext.py:48-58):
 *   out = xin + pwconv2(GELU(pwconv1(LayerNorm(t))))      LayerNorm over the D channels, eps = ln_eps (1e-6), exact-erf GELU
 * t [rows, t_ld] = the depthwise stage's output, xin [rows, x_ld] = the block's input (residual), w1 [4D][D], b1 [4D], w2 [D][4D],
 * b2 [D], ln_g / ln_b [D].  The 4D-wide hidden tensor never reaches HBM.  bf16 / f16 only; D in {32, 64, 128}. */
int cfp_lkpm_tail(const void* t, int t_ld, const void* xin, int x_ld, void* out, int out_ld, const void* w1, const float* b1,
                  const void* w2, const float* b2, const float* ln_g, const float* ln_b, float ln_eps, int rows, int D, int dtype,
                  cfp_stream_t stream);

/* Bin-width regressor + bin edges/centres, one workgroup per batch element, all f32:
 *   mean -> conv1x1 (no bias) -> Linear/LeakyReLU x2 -> Linear -> norm -> widths -> cumsum
 * Every weight matrix is passed TRANSPOSED, [n_in][n_out] (coalesced across output threads).
 * norm: 0 linear (relu + 0.1, L1-normalise), 1 softmax, 2 sigmoid (L1-normalise).
 * edges [B][nbins+1], centers [B][nbins].  Replaces decoder.py:23-36 + deltar.py:53-59. */
int cfp_bin_regressor(const float* partial, int nsplit, float inv_hw, const float* w1x1_t,
                      const float* w0_t, const float* b0, const float* w1_t, const float* b1,
                      const float* w2_t, const float* b2, float min_val, float max_val, int norm,
                      float* edges, float* centers, int B, int C, int hidden, int nbins, cfp_stream_t stream);

/* Spatial SUM of a linear 3x3 convolution's output (stride 1, zero padding 1, bias, no activation) without running the convolution:
 * from nine shifted sums of its INPUT (csrc/head.hip).  Used for DepthRegression's mean over conv1x1(unet) (decoder.py:28-29) where
 * unet = decoder.conv0(t) (decoder.py:126): the regressor branch then depends on t only and runs beside conv0.
 *   partial [B][nsplit][C] f32: channel sums of x over row splits (cfp_channel_sum); x [B,H,W,ld] (C channels, C divides 1024): the
 *   kernel reads its four border lines and corner pixels itself; w [Cout][3][3][C] float32, bias [Cout] (may be NULL) ->
 *   msum [B][Cout] = sum over all H*W output pixels (feed it to cfp_bin_regressor with nsplit = 1, inv_hw = 1/HW). */
int cfp_conv3x3_mean(const float* partial, int nsplit, const void* in, int in_ld, const float* w, const float* bias, float* msum,
                     int B, int H, int W, int C, int Cout, int dtype, cfp_stream_t stream);

/* Per-pixel softmax over nbins logits + expectation over bin centres:
 *   prob[b, n, hw] = softmax_n(logits[b*HW + hw, n]);  pred[b, hw] = sum_n prob * centers[b, n]
 * prob (NCHW, `dtype`) may be NULL.  pred is f32.  Replaces nn.Softmax(dim=1) of deltar.py:19
 * and deltar.py:61. */
int cfp_bin_softmax(const void* logits, int ld, const float* centers, void* prob, float* pred,
                    int B, int HW, int nbins, int dtype, cfp_stream_t stream);

/* The whole adaptive-bins head in one kernel (csrc/head_fused.hip):
 *   ram = conv3x3(x) (128 -> 128, decoder.py:22-27 `self.conv3x3`), logits = conv_out(ram) (1x1, 128 -> 256,
 *   deltar.py:18-19,51), prob = softmax over the 256 bins, pred = sum_n prob * centers (deltar.py:61).
 * `ram` and the logits never reach HBM.  x [B*H*W, x_ld] NHWC 16-bit with 128 channels; w3 [128][3*3*128] as cfp_conv2d_nhwc
 * lays weights out; scale3 / shift3 [128] f32 (may be NULL = 1 / 0; depth_head.conv3x3 has a bias and no activation);
 * wout_perm [256][128] with the input-channel axis permuted for the kernel's fragment order: position 32 kb + 8 q + e holds
 * channel 32 kb + 16 (e >> 2) + 4 q + (e & 3) (kb, q in 0..3, e in 0..7); with CFP_HEAD_WOUT_HILO a second [256][128] plane
 * follows holding round16(W - hi) so conv_out's weights act with ~22 significant bits; CFP_HEAD_RAM_HILO feeds ram to the
 * second GEMM as hi + lo (no 16-bit rounding of ram).  bias_out [256], centers [B,256] f32; prob [B,256,H*W] (NCHW, `dtype`,
 * may be NULL); pred [B*H*W] f32; ram_out [B*H*W,128] (`dtype`, may be NULL: test hook).  bf16 / f16 only; H*W % 16 == 0 and H*W >= 128. */
enum { CFP_HEAD_WOUT_HILO = 1, CFP_HEAD_RAM_HILO = 2 };
int cfp_depth_head_fused(const void* x, int x_ld, const void* w3, const float* scale3, const float* shift3,
                         const void* wout_perm, const float* bias_out, const float* centers, void* prob, float* pred,
                         void* ram_out, int B, int H, int W, int flags, int dtype, cfp_stream_t stream);

/* Fused bin head: logits = x @ w^T + bias never leave the chip:
 *   1x1 conv (Cin -> 256) on the matrix cores, row softmax, expectation, optional prob write.
 * Replaces conv_out (deltar.py:18-19,51) + deltar.py:61.  nbins == 256.  dtype CFP_BF16 / CFP_F16: 16-bit x, w [256][Cin] and prob;
 * dtype CFP_F32X3 (the default numerics of the drop-in boundary): float32 x, `w` = cfp_pack_w_x3 of the [256][Cin] weights, float32 prob
 * -- the reference's own output type -- written by the kernel; HW % 4 == 0. */
int cfp_bin_head_fused(const void* x, int x_ld, const void* w, const float* bias, const float* centers,
                       void* prob, float* pred, int B, int HW, int Cin, int dtype, cfp_stream_t stream);

/* ---- training step pieces -------------------------------------------------- */

/* SILogLoss.forward (loss.py:9-19), f32: pred [B,1,Hp,Wp] is bilinearly resized (align_corners=True) to the
 * target size when `interpolate`, g = log p - log t over the pixels with mask != 0 (mask may be NULL = all),
 * loss = 10 * sqrt(var_unbiased(g) + 0.15 * mean(g)^2).  stats[4] (device, f32) = {loss, mean(g), n, Dg}.
 * ws (cfp_silog_ws_bytes) keeps g and 1/p for the backward pass.  Reductions are f64 partial sums combined
 * in a fixed order (deterministic). */
size_t cfp_silog_ws_bytes(int B, int Ht, int Wt);
int cfp_silog_loss_fwd(const float* pred, int Hp, int Wp, const float* target, const unsigned char* mask, int Ht, int Wt,
                       int B, int interpolate, void* ws, size_t ws_bytes, float* stats, cfp_stream_t stream);
/* d loss / d pred (same ws / stats as the forward call), scaled by grad_loss; a gather, no atomics. */
int cfp_silog_loss_bwd(const void* ws, const float* stats, float grad_loss, int Hp, int Wp, int Ht, int Wt, int B,
                       int interpolate, float* grad_pred, cfp_stream_t stream);

/* torch.optim.AdamW step (train.py:82,131) over one contiguous segment of flat f32 buffers: `step` is the
 * 1-based step count of the group, beta1 may change per step (OneCycleLR cycles it, train.py:90-94).
 * grad_scale (device, may be NULL) multiplies the gradient first: the clip factor of cfp_grad_clip_factor. */
int cfp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                   cfp_stream_t stream);
/* nn.utils.clip_grad_norm_ (train.py:130) without a host sync: out[0] = min(1, max_norm / (||grad||_2 + 1e-6)),
 * out[1] = ||grad||_2, both on the device; `out` holds 3 floats.  OVERFLOW GUARD of the 16-bit training modes: when the norm is not
 * finite (an inf / nan anywhere in the gradient) out[0] = -1 and out[2] is incremented; cfp_adamw_step given such a grad_scale leaves
 * parameters and moments untouched (the step is skipped, like torch.cuda.amp.GradScaler does).  `grad` must be 16-byte aligned. */
size_t cfp_grad_clip_ws_bytes(void);
int cfp_grad_clip_factor(const float* grad, long long n, float max_norm, void* ws, size_t ws_bytes, float* out,
                         cfp_stream_t stream);

/* ---- data path either side of the model ------------------------------------------------- */

/* ToF zone-histogram simulation from ground-truth depth, a whole batch per launch.  Replaces the per-sample CPU
 * functions get_hist_parallel (src/utils/dataloader.py:83-134) and the uniform branch of
 * sample_point_from_hist_parallel (:65-80) that the data-loader workers call (src/dataloader/nyu.py:154,179).
 *   depth    [B] images of H x W f32, `img_stride` elements apart (device)
 *   zone grid: zone_num x zone_num squares of zone_px pixels whose top-left zone starts at (sy0 + off, sx0 + off);
 *            the reference uses sy0 = int((H - zone_px*zone_num)/2), 64 px zones when training and 56 px otherwise
 *            (:94-103).  offsets (device, [B] int32, may be NULL) is the per-sample `train_zone_random_offset`
 *            draw (:98-100), clamped to +-offset_bound; the grid must stay inside the image for that bound.
 *   histogram: torch.histc(bins, min=0, max=max_distance) in f32 (:104), bin 0 cleared and floor_count (20)
 *            subtracted (:108-109), strongest run of consecutive non-zero bins kept (:110-116), moments over bin
 *            centres built from multiples of bin_width (0.04) in f64 (:118,127-130).
 *   w0, w1   [nsamp] f32 (device): torch.linspace(1,0,nsamp) / torch.linspace(0,1,nsamp) as the HOST evaluates
 *            them (tensor_linspace, :42-57; ATen's vectorised linspace differs in the last bit between hosts);
 *            with sample_mode CFP_TOF_SAMPLE_ICDF w0 is the erfinv table of cfp_tof_sample_points and w1 may be NULL.
 * Outputs (device): fh [B,Z,2] f64 (mu, sigma); rect [B,Z,4] f32 (sy,sx,ey,ex); mask [B,Z] u8; pts [B,Z,nsamp] f32
 * (zero where mask is 0); hist_out [B,Z,bins] i32 (may be NULL): the counts that survive cluster selection. */
int cfp_tof_hist_sim(const float* depth, long long img_stride, int B, int H, int W, int zone_num, int zone_px, int sy0,
                     int sx0, const int* offsets, int offset_bound, float max_distance, int bins, double bin_width,
                     int floor_count, const float* w0, const float* w1, int nsamp, int sample_mode, double* fh, float* rect,
                     unsigned char* mask, float* pts, int* hist_out, cfp_stream_t stream);
/* The sampling step alone (sample_point_from_hist_parallel, dataloader.py:65-80), for callers that keep the reference's
 * two calls.  sample_mode CFP_TOF_SAMPLE_UNIFORM (--sample_uniform, :74-79): pts[z, t] = f32(w0[t]*(mu-3 sigma) +
 * w1[t]*(mu+3 sigma)) in f64.  CFP_TOF_SAMPLE_ICDF (the argparse default, :69-73): Normal(mu, sigma).icdf at the
 * nsamp ppf points arange(1e-3, 1, 0.998/(nsamp-1)): pts[z, t] = f32(mu + (sigma * w0[t]) * sqrt(2)) in f64 with
 * w0[t] = erfinv(2 ppf_t - 1) evaluated in FLOAT32 on the host, as torch does (the ppf tensor is f32); w1 is unused
 * and may be NULL.  pts is 0 where !mask[z]. */
int cfp_tof_sample_points(const double* fh, const unsigned char* mask, const float* w0, const float* w1, long long nzones,
                          int nsamp, int sample_mode, float* pts, cfp_stream_t stream);

/* Depth-evaluation metrics per image without leaving the device: compute_errors (src/utils/metrics.py:4-24 =
 * evaluate_all.py:15-35) fused with the protocol around it.
 *   mode 0 = evaluate_all.py:38-41,80-84: pred clipped to [lo, hi] at model resolution, then bilinear
 *            (align_corners=True) to H x W; valid pixels lo < gt < hi (args.min_depth / args.max_depth).
 *   mode 1 = train.py:187-199 (validate): bilinear first, then clamp to [lo, hi], nan -> lo; valid pixels
 *            lo < gt < hi (args.min_depth_eval / args.max_depth_eval).
 *   pred [B,Hp,Wp] f32, gt [B,H,W] f32 (device); interpolate = 0 requires equal sizes.
 * out [B,10] f64 (device) = {a1, a2, a3, abs_rel, rmse, log_10, rmse_log, silog, sq_rel, n_valid}; an image with
 * n_valid = 0 yields NaN metrics (the reference skips it).  Per-pixel terms f32, sums f64 in a fixed order. */
size_t cfp_eval_metrics_ws_bytes(int B);
int cfp_eval_metrics(const float* pred, int Hp, int Wp, const float* gt, int H, int W, int B, int interpolate, int mode,
                     float lo, float hi, void* ws, size_t ws_bytes, double* out, cfp_stream_t stream);

/* ---- training-step kernels: backward of the dense convolution, batch-statistics BatchNorm --------------------------
 * (the training row of SURVEY.md section 8: cfpnet_amd/autograd_hip.py chains them into the backward of the whole network) */

/* Weight gradient of nn.Conv2d / nn.Linear (autograd of the layers cfp_conv2d_nhwc replaces; train.py:125):
 * dw[Cout][KH*KW*Cin] (f32) = beta * dw + sum over output pixels of dy[m][co] * x[window(m)][ci].  x [B,H,W,Cin] and
 * dy [B,Ho,Wo,Cout] in `dtype` (16-bit inputs are widened to f32 on the way to the matrix cores; accumulation f32).
 * Split over pixel chunks into f32 slabs (ws) that a second kernel adds in a fixed order: bit-reproducible. */
size_t cfp_conv2d_wgrad_ws_bytes(int Cout, int K, int M);
int cfp_conv2d_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws,
                     size_t ws_bytes, cfp_stream_t stream);
/* The same, plus the bias gradient db[co] = beta_b * db + sum over output pixels of dY[m][co] from the SAME launch (16-bit dtypes only:
 * one more matrix-core product against a fragment of ones; float32 callers use cfp_colsum).  db may be NULL (= cfp_conv2d_wgrad). */
int cfp_conv2d_wgrad_bias(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, float beta_b,
                          int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);
/* The reduction of the slabs postponed and batched: cfp_conv2d_wgrad_deferred runs the slab launch of cfp_conv2d_wgrad_bias and
 * describes the remaining reduction in *job (host memory; job->nsplit = 0 when the launch already wrote dw in place) instead of
 * launching it; cfp_wgrad_reduce_jobs(jobs, n) then finishes up to 48 layers per launch (the job table travels in the kernel
 * arguments), in the same summation order as the per-layer reduction: bit-identical gradients, ~240 launches fewer per training
 * step.  `ws` of a deferred call must stay untouched until its job has been reduced; jobs that write the same dw / db are put
 * into separate launches, in the order given. */
typedef struct cfp_wgrad_job {
  const float* slabs; float* dw; float* db;
  long long n, n_dw;
  int nsplit, ew;
  float beta, beta_b;
} cfp_wgrad_job;
int cfp_conv2d_wgrad_deferred(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, float beta, float beta_b,
                              int dtype, void* ws, size_t ws_bytes, cfp_wgrad_job* job, cfp_stream_t stream);
int cfp_wgrad_reduce_jobs(const cfp_wgrad_job* jobs, int njobs, cfp_stream_t stream);
/* Two more producers of such jobs (their finishing sums are [split][n] slabs too): the depthwise 3x3 weight gradient
 * (cfp_dwconv3x3_wgrad) and the LayerNorm parameter gradients (cfp_layernorm_bwd: dbeta | dgamma).  Same contract: `ws` untouched
 * until cfp_wgrad_reduce_jobs has run; the results equal the immediate calls up to float32 summation order. */
int cfp_dwconv3x3_wgrad_deferred(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int stride,
                                 int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws, size_t ws_bytes,
                                 cfp_wgrad_job* job, cfp_stream_t stream);
int cfp_layernorm_bwd_deferred(const void* x, int ld, const void* dy, int dy_ld, const float* gamma, float eps, void* dx, int dx_ld,
                               int accumulate, float* dgamma, float* dbeta, long long rows, int C, int dtype, void* ws, size_t ws_bytes,
                               cfp_wgrad_job* job, cfp_stream_t stream);
/* wt[Cin][KH][KW][Cout] = w[Cout][KH-1-kh][KW-1-kw][Cin]: the weights the data gradient convolves with. */
int cfp_conv2d_weight_flip(const void* w, void* wt, int Cout, int KH, int KW, int Cin, int dtype, cfp_stream_t stream);
/* The same for n weight tensors in one launch (a training step flips every convolution's weights once).  `desc` is a DEVICE
 * array of n rows of 8 int64: {source offset, destination offset (elements from src_base / dst_base), Cout, KH, KW, Cin,
 * first workgroup, workgroups}; a tensor of E elements takes cfp_weight_flip_blocks(E) workgroups, rows sorted by first
 * workgroup, total_blocks = their sum. */
int cfp_weight_flip_blocks(long long elems);
int cfp_conv2d_weight_flip_batch(const void* src_base, void* dst_base, const long long* desc, int n, int total_blocks, int dtype,
                                 cfp_stream_t stream);
/* Data gradient: dx [B,H,W,Cin] (+= when accumulate) from dy [B,Ho,Wo,Cout] and the flipped weights, for the forward
 * geometry (KH,KW,stride,pad_t,pad_l): a stride-1 convolution over dy with `stride - 1` zeros stuffed between its pixels. */
int cfp_conv2d_dgrad(const void* dy, int dy_ld, const void* wt, void* dx, int dx_ld, int B, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate, int dtype, void* ws,
                     size_t ws_bytes, cfp_stream_t stream);

/* cfp_conv2d_nhwc (no activation / residual) that ALSO emits, per tile of output rows, the channel moments of its stored output for the
 * batch-statistics BatchNorm that follows it in model.train() (timm ConvBnAct / torch `Conv2d -> BatchNorm2d`): mom[tile][0][c] = mean,
 * mom[tile][1][c] = sum of squared deviations from that mean, over the tile's rows.  On return *nsplit = tiles written and
 * *rows_per_split = rows per tile -- or *nsplit = 0 when the kernel chosen for this problem does not produce them (float32, split-K,
 * mom_floats too small): then run cfp_bn_train_stats.  Saves one pass over the activation and one launch per layer. */
int cfp_conv2d_nhwc_moments(const void* in, int in_ld, const void* w, const float* bias, void* out, int out_ld, int B, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int dtype,
                            void* ws, size_t ws_bytes, float* mom, size_t mom_floats, int* nsplit, int* rows_per_split,
                            cfp_stream_t stream);
/* cfp_bn_train_stats from such partials ([split][2][C]; split j covers rows [j * rows_per_split, min(rows, (j + 1) * rows_per_split))). */
int cfp_bn_train_stats_partials(const float* partial, int nsplit, long long rows, long long rows_per_split, int C, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                float* var, float* invstd, float* scale, float* shift, cfp_stream_t stream);

/* Training-mode BatchNorm over `rows` NHWC rows (nn.BatchNorm2d/1d in model.train()): batch mean / biased variance
 * (one pass: shifted sums merged as exact (n, mean, M2) triples), running statistics updated with `momentum` (running_var from the unbiased variance), and the folded
 * per-channel scale = gamma*invstd, shift = beta - mean*scale that cfp_scale_shift_act applies with the activation. */
size_t cfp_bn_ws_bytes(int C);
int cfp_bn_train_stats(const void* x, int ld, long long rows, int C, int dtype, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, float* mean, float* var, float* invstd,
                       float* scale, float* shift, void* ws, size_t ws_bytes, cfp_stream_t stream);
/* out = act(x * scale[c] + shift[c]) (BatchNorm apply + SiLU / LeakyReLU / ReLU / GELU / sigmoid / none). */
int cfp_scale_shift_act(const void* x, int ld, const float* scale, const float* shift, int act, void* out, int out_ld,
                        long long rows, int C, int dtype, cfp_stream_t stream);
/* out = act(x * scale[c] + shift[c]) + res  (res may be NULL): the block's skip connection added in the same pass (the sum is rounded
 * like the separate add of the stored activation). */
int cfp_scale_shift_act_res(const void* x, int ld, const float* scale, const float* shift, int act, const void* res, int res_ld,
                            void* out, int out_ld, long long rows, int C, int dtype, cfp_stream_t stream);
/* Backward of act(BatchNorm(x)) in training mode: dgamma, dbeta (f32) and dx from the pre-BN input x and dy. */
int cfp_bn_train_bwd(const void* x, int ld, const void* dy, int dy_ld, long long rows, int C, int dtype, const float* mean,
                     const float* invstd, const float* scale, const float* shift, int act, float* dgamma, float* dbeta,
                     void* dx, int dx_ld, void* ws, size_t ws_bytes, cfp_stream_t stream);

/* out[c] = sum over rows of x[r][c] (f32): the bias gradient of a conv / linear layer.  ws: cfp_bn_ws_bytes(C). */
int cfp_colsum(const void* x, int ld, long long rows, int C, int dtype, float* out, void* ws, size_t ws_bytes, cfp_stream_t stream);
/* dz = dy * act'(z): backward of a stand-alone SiLU / ReLU / LeakyReLU / GELU / sigmoid given its input z. */
int cfp_act_bwd(const void* z, int ld, const void* dy, int dy_ld, int act, void* dz, int dz_ld, long long rows, int C, int dtype,
                cfp_stream_t stream);
/* Backward of nn.LayerNorm over the channel axis (transformer.py:38-39, convnext.py:31): dx (+= when accumulate),
 * dgamma, dbeta (f32) from the layer's input x and dy. */
size_t cfp_layernorm_bwd_ws_bytes(long long rows, int C);
int cfp_layernorm_bwd(const void* x, int ld, const void* dy, int dy_ld, const float* gamma, float eps, void* dx, int dx_ld,
                      int accumulate, float* dgamma, float* dbeta, long long rows, int C, int dtype, void* ws, size_t ws_bytes,
                      cfp_stream_t stream);

/* out = a*x + b*y (y may be NULL): gradient accumulation at skip connections, scaling. */
int cfp_axpby(const void* x, int x_ld, const void* y, int y_ld, float a, float b, void* out, int out_ld, long long rows, int C,
              int dtype, cfp_stream_t stream);
/* Gradient of `x + PE[oy:oy+H, ox:ox+W]` (fusion.py:87-97) w.r.t. the learned table: dtable[...] = beta*dtable + sum_b dx. */
int cfp_rowtable_grad(const void* dx, int ld, float* dtable, int B, int H, int W, int C, int Wt, int oy, int ox, float beta,
                      int dtype, cfp_stream_t stream);
/* Squeeze-excite gate in the training step (timm SqueezeExcite of the encoder's inverted-residual blocks, encoder.py:57-69), float32:
 *   forward : mean[b,c] = sum_s partial[b,s,c] * inv_hw;  z1 = W1 mean + b1;  gate = sigmoid(W2 silu(z1) + b2)      (one launch)
 *   backward: from dgate[b,c] = sum_hw dy * x (cfp_channel_dot): dW1, db1, dW2, db2 (out = beta * out + grad) and
 *             add[b,c] = d(loss)/d(mean) * inv_hw, the term cfp_bcast_fma adds to dy * gate                          (two launches)
 * W1 [Rp][C], b1 [Rp], W2 [C][Rp], b2 [C]; Rp = hidden width padded to a multiple of 4 with zero rows / columns, <= 64; C <= 2048;
 * B <= 64 in the backward.  mean [B][C], z1 [B][Rp], gate [B][C] are kept by the caller between the two calls;
 * ws: cfp_se_train_ws_floats(B, C, Rp) floats. */
size_t cfp_se_train_ws_floats(int B, int C, int Rp);
int cfp_se_train_fwd(const float* partial, int nsplit, float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
                     float* mean, float* z1, float* gate, int B, int C, int Rp, cfp_stream_t stream);
int cfp_se_train_bwd(const float* dgate, const float* gate, const float* z1, const float* mean, const float* w1, const float* w2,
                     float* dw1, float* db1, float* dw2, float* db2, float* add, float* ws, float inv_hw, float beta, int B, int C,
                     int Rp, cfp_stream_t stream);
/* out[b][c] = sum over the HW rows of image b of x*y: gradient of the squeeze-excite gate (sum dy * x). */
int cfp_channel_dot(const void* x, int x_ld, const void* y, int y_ld, float* out, int B, int HW, int C, int dtype, cfp_stream_t stream);
/* dx = dy * gate[b][c] + add[b][c] (add may be NULL): squeeze-excite backward w.r.t. the gated activation. */
int cfp_bcast_fma(const void* dy, int dy_ld, const float* gate, const float* add, void* dx, int dx_ld, int B, int HW, int C, int dtype,
                  cfp_stream_t stream);
/* Depthwise 3x3 backward (conv_dw of timm's InvertedResidual): data gradient (+= when accumulate) and weight gradient
 * dw[9][C] f32 = beta*dw + sum over output pixels of dy * x(tap). */
int cfp_dwconv3x3_dgrad(const void* dy, int dy_ld, const void* w, void* dx, int dx_ld, int B, int H, int W, int C, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, int accumulate, int dtype, cfp_stream_t stream);
size_t cfp_dwconv3x3_wgrad_ws_bytes(int C);
int cfp_dwconv3x3_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, float beta, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);

/* out[i] = idx[i] >= 0 ? x[idx[i]] : 0 (+= when accumulate), idx int32 on the device: crop / regroup / window partition /
 * inside-outside partition of token rows (fusion.py:132-157, transformer.py:101-116,215-234); the maps are injective, so
 * the adjoint of a gather is the gather with the inverse map. */
int cfp_index_rows(const void* x, int x_ld, const int* idx, void* out, int out_ld, long long n_out, int C, int accumulate, int dtype,
                   cfp_stream_t stream);
/* Backward of the full-map bilinear resize (align_corners=True) [B,Hs,Ws,C] -> [B,Hd,Wd,C]: dx (+= when accumulate). */
int cfp_resize_bilinear_bwd(const void* dy, int dy_ld, void* dx, int dx_ld, int B, int Hs, int Ws, int Hd, int Wd, int C,
                            int accumulate, int dtype, cfp_stream_t stream);
/* Adaptive bins (deltar.py:53-59): edges = cumsum(pad((max-min)*w, min)), centres = mid-points; and the adjoint. */
int cfp_bin_centers(const float* widths_normed, float min_val, float max_val, float* edges, float* centers, int B, int NB,
                    cfp_stream_t stream);
int cfp_bin_centers_bwd(const float* dcenters, float min_val, float max_val, float* dwidths_normed, int B, int NB, cfp_stream_t stream);
/* pred = sum_n softmax(logits)[n] * centres[b][n] (deltar.py:51,61) with logits [B*HW, NB] rows; with dpred != NULL the
 * backward instead: dlogits and dcentres (pred unused).  ws: cfp_softmax_expect_ws_bytes (backward only). */
size_t cfp_softmax_expect_ws_bytes(int B, int HW, int NB);
int cfp_softmax_expect(const void* logits, int ld, const float* centers, float* pred, const float* dpred, void* dlogits, int dl_ld,
                       float* dcenters, int B, int HW, int NB, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);

/* Linear attention, training form (attention.py:20-52 and its autograd) on contiguously grouped tokens:
 * q [N*L, heads*d], k, v [N*S, heads*d] -> out [N*L, heads*d]; `state` (cfp_linattn_state_bytes) keeps KV and Ksum of
 * every (group, head) for the backward call, which returns dq, dk, dv.  d in {4, 8, 16, 32}.
 * ws (cfp_linattn_ws_bytes, may be 0 / NULL): with fewer than 512 (group, head) pairs -- the global attentions, N = batch --
 * the token axes are split over more workgroups and the partial [d x (d+1)] states meet in ws, added in a fixed order;
 * without ws one workgroup per (group, head) walks all tokens (same values up to float32 summation order). */
size_t cfp_linattn_state_bytes(int N, int heads, int d);
size_t cfp_linattn_ws_bytes(int N, int L, int S, int heads, int d);
int cfp_linattn_fwd(const void* q, int q_ld, const void* k, int k_ld, const void* v, int v_ld, void* out, int out_ld, float* state,
                    int N, int L, int S, int heads, int d, float eps, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);
int cfp_linattn_bwd(const void* q, int q_ld, const void* k, int k_ld, const void* v, int v_ld, const void* dout, int do_ld,
                    const float* state, void* dq, int dq_ld, void* dk, int dk_ld, void* dv, int dv_ld, int N, int L, int S,
                    int heads, int d, float eps, int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);

/* Weight gradient of the large-kernel depthwise convolution of LKPM (convnext.py:30, k = 7/15/31, zero padding (k-1)/2):
 * dw[C][k][k] f32 = beta*dw + sum over pixels of dy * x(tap).  The data gradient is cfp_dwconv_large_nhwc with the kernel
 * flipped in both axes. */
size_t cfp_dwconv_large_wgrad_ws_bytes(int B, int H, int W, int C, int k);
int cfp_dwconv_large_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int B, int H, int W, int C, int k, float beta,
                           int dtype, void* ws, size_t ws_bytes, cfp_stream_t stream);

/* Bin-width normalisation out = x / sum_c x over f32 rows (decoder.py:36); with dy != NULL its backward
 * out = (dy - sum_c(dy * y)) / s instead (x = the forward input). */
int cfp_row_normalize(const float* x, const float* dy, float* out, int rows, int C, cfp_stream_t stream);

/* cfp_add_rowtable / cfp_rowtable_grad with the window origin (oy, ox) read from DEVICE memory (int32[2], clamped to the
 * Ht x Wt table): the random positional-encoding window of fusion.py:87-91 can change between replays of a captured graph. */
int cfp_add_rowtable_dev(const void* in, int in_ld, const float* table, void* out, int out_ld, int rows, int C, int H, int W,
                         int Ht, int Wt, const int* oyox, int dtype, cfp_stream_t stream);
int cfp_rowtable_grad_dev(const void* dx, int ld, float* dtable, int B, int H, int W, int C, int Ht, int Wt, const int* oyox,
                          float beta, int dtype, cfp_stream_t stream);

/* NYU training augmentation in one pass (src/dataloader/nyu.py:128-136,204-245,266-285): crop to H x W at (x0, y0), optional
 * horizontal flip, optional gamma / brightness / per-channel colour gain + clip, /255, ImageNet normalisation; depth
 * millimetres -> metres.  rgb_u8 [B,H0,W0,3] and depth_mm [B,H0,W0] uint16 (may be NULL together with depth_out) on the device;
 * params_i [B,4] int32 = (x0, y0, flip, do_augment), params_f [B,2] f32 = (gamma, brightness), colors [B,3] f64 on the
 * device -- the draws themselves stay on the host like in the reference; mean3 / std3 are HOST arrays of 3 floats.
 * image_out [B,3,H,W] f32, depth_out [B,1,H,W] f32.  Crop origins are clamped to the source image. */
int cfp_nyu_augment(const unsigned char* rgb_u8, const unsigned short* depth_mm, int B, int H0, int W0, const int* params_i,
                    const float* params_f, const double* colors, int H, int W, const float* mean3, const float* std3, float* image_out,
                    float* depth_out, cfp_stream_t stream);

/* The random rotation that precedes it (nyu.py:121-124, 200-202): PIL `Image.rotate(angle, BILINEAR)` on the RGB image and
 * `rotate(angle, NEAREST)` on the 16-bit depth, same size, zero fill, byte-exact with Pillow 12 (float64 arithmetic in
 * Pillow's operation order).  matrices [B,6] f64 on the device = Pillow's destination->source affine matrix per sample
 * (computed on the host from the drawn angle exactly as Image.rotate does, cfpnet_amd/augment.py: rotate_matrix).
 * rgb_u8 / rgb_out [B,H,W,3] uint8, depth_mm / depth_out [B,H,W] uint16; either pair may be NULL.  Not in place. */
int cfp_nyu_rotate(const unsigned char* rgb_u8, const unsigned short* depth_mm, unsigned char* rgb_out, unsigned short* depth_out,
                   int B, int H, int W, const double* matrices, cfp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CFPNET_HIP_H */
