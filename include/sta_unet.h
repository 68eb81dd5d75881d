/*
 * sta_unet.h — C-ABI of the HBM-bound glue kernels around the cross-attention path (same library,
 * same conventions as sta_xattn.h: raw device pointers, caller-owned buffers, stream-ordered, 0 / STA_E_*).
 *
 * These replace eager-PyTorch sequences in the CALLERS of the hot path (SURVEY.md §8 rows a3/a4/a6:
 * BasicTransformerBlock._forward, SpatialTransformer.forward, ResBlock._forward), each of which is a chain of
 * one-pass-per-op elementwise / normalisation kernels over [2I, C, H, W] or [2I, N, C] activations:
 * pure HBM traffic. One pass per chain instead of one per op. Inference (no autograd) only — the Python
 * wrappers fall back to the eager ops whenever gradients are being recorded.
 *
 * dtype: STA_BF16 / STA_F16 for activations and affine parameters; statistics and arithmetic in fp32.
 */
#ifndef STA_UNET_H
#define STA_UNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * y = act( GroupNorm_G( x + add[b, c] ) * gamma[c] + beta[c] ),  act = SiLU if silu != 0 else identity.
 *   x, y  [B][C][HW] (NCHW, contiguous), HW % 8 == 0, C % G == 0;  add [B][C] fp32 or NULL;  gamma, beta [C].
 * Replaces (reference openaimodel.py ResBlock._forward, util.py GroupNorm32, attention.py:335-337):
 *   `h + emb_out[..., None, None]` (with the producing conv's bias folded into `add`)  ->  GroupNorm32  ->  SiLU.
 * One workgroup per (b, group) slab; slabs up to 64 Ki elements are held in registers between the statistics
 * and the normalisation (one HBM read, exact two-pass variance), larger ones are read twice.
 */
int sta_groupnorm_silu(const void* x, const float* add, const void* gamma, const void* beta, void* y,
                       int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream);

/*
 * y[r][j] = x[r][j] * gelu(x[r][D + j])  (exact erf GELU), x [R][2D], y [R][D], D % 8 == 0.
 * Replaces GEGLU.forward after the projection (attention.py:43-45): chunk + gelu + mul = 3 passes -> 1.
 */
int sta_geglu(const void* x, void* y, long R, int D, int dtype, void* stream);

/*
 * s = x + f + bias (row-broadcast, bias may be NULL; f may be NULL: s = x);  y = LayerNorm(s) * gamma + beta.
 *   x, f, s, y [R][C], C % 8 == 0, C <= 2048. s may be NULL (not stored) or alias x.
 * Replaces `x = attn(norm(x)) + x` followed by the next `norm(x)` of BasicTransformerBlock._forward
 * (attention.py:274-299): residual add and LayerNorm in one pass. One wave per row.
 */
int sta_add_layernorm(const void* x, const void* f, const void* bias, const void* gamma, const void* beta,
                      void* s, void* y, long R, int C, float eps, int dtype, void* stream);

/*
 * sta_add_layernorm with y in QUERY-FRAGMENT order, the private layout between norm2 and the projection-fused
 * cross-attention (attention.py:279-281 -> :178): rows are taken in groups of 16 consecutive rows; group P, fragment
 * s (C/32 per group, 1 KiB each) holds at byte offset (P * C/32 + s) * 1024 + (16 g + c) * 16 the 8 values
 * y[16 P + c][32 s + 8 g .. + 7] — one fragment is one MFMA B operand of sta_xattn_fwd_proj_qfrag (lane = 16 g + c),
 * fetched by one fully coalesced 1-KiB load. Same bytes in total as row-major. s (if not NULL) stays row-major.
 * R % 16 == 0, C % 32 == 0, C <= 1024 (two chunks per lane above 512). Values are bit-identical to sta_add_layernorm's.
 */
int sta_add_layernorm_qfrag(const void* x, const void* f, const void* bias, const void* gamma, const void* beta,
                            void* s, void* y, long R, int C, float eps, int dtype, void* stream);

/*
 * The first half of the block's feed-forward in one pass (csrc/sta_ffgemm.hip):
 *     h = (y W_v^T + b_v) * gelu(y W_g^T + b_g)      GEGLU.forward (attention.py:42-45), exact-erf GELU as sta_geglu
 * y = norm3(x) read in QUERY-FRAGMENT order (sta_add_layernorm_qfrag / sta_to_out_ln_ofrag with y_qfrag), proj.weight
 * [2 * inner][C] re-laid out once per model by sta_ff_geglu_pack_w (sta_ff_geglu_packed_w_bytes bytes; 0 = unsupported:
 * C = 320, inner = 1280 only) and streamed through LDS; the [R][2 * inner] projection never exists in HBM (row-major: a library
 * GEMM that writes it + sta_geglu that reads it back).
 *   bias: [2 * inner] dtype (value half, then gate half) or NULL;  h: [R][inner] dtype, row-major (h_frag = 0) or in the
 *   fragment order of sta_ff_out_res_hfrag (h_frag = 1);  R % 16 == 0
 */
size_t sta_ff_geglu_packed_w_bytes(int C, int inner);
int sta_ff_geglu_pack_w(const void* w, void* packed, int C, int inner, int dtype, void* stream);
int sta_ff_geglu_qfrag(const void* y_qfrag, const void* packed_w, const void* bias, void* h, long R, int C, int inner,
                       int h_frag, int dtype, void* stream);
/*
 * ... and the second half with the block's last residual (attention.py:66-69 and the `+ x` of :299):
 *     out = x + h W2^T + b2
 * h read in FRAGMENT order (sta_ff_geglu_qfrag with h_frag = 1: per 16-row group inner/32 fragments of 1 KiB, lane 16 g + c =
 * row c, channels 32 s + 8 g .. + 7 — one contiguous KiB per store of the first kernel, one coalesced load of this one);
 * net[2].weight [C][inner] re-laid out once by sta_ff_out_pack_w (sta_ff_out_packed_w_bytes bytes; C = 320, inner = 1280 only).
 * Row-major this is a library GEMM + a separate residual add; here h crosses HBM once in each direction as whole KiB.
 *   bias: [C] or NULL;  x, out: [R][C] dtype row-major;  R % 16 == 0
 */
size_t sta_ff_out_packed_w_bytes(int C, int inner);
int sta_ff_out_pack_w(const void* w, void* packed, int C, int inner, int dtype, void* stream);
int sta_ff_out_res_hfrag(const void* h_frag, const void* packed_w, const void* bias, const void* x, void* out, long R, int C,
                         int inner, int dtype, void* stream);

/*
 * The 3x3, stride-1, padding-1 convolutions of the UNet trunk on NHWC activations as an implicit GEMM on the MFMAs
 * (ResBlock in_layers / out_layers, openaimodel.py:163-275; Upsample.conv :107-120; csrc/sta_conv.hip):
 *     out = conv(x) + bias + res            (`skip_connection(x) + out_layers(h)` of ResBlock._forward in the epilogue)
 *   x: [B][H >> up2][W >> up2][Cin] dtype;  out: [B][H][W][Cout] dtype;  zeros: >= 2 * Cin bytes of zeros (the padding halo);
 *   bias: [Cout] dtype or NULL;  res: [B][H][W][Cout] dtype or NULL (may not alias out);
 *   stats: NULL, or [B + 1][sta_conv3x3_stats_slots(H, W)][Cout][2] fp32: every wave stores the partial sum and sum of squares of the values it
 *   writes, per output channel (plain stores, fixed slots; the spare image absorbs a half-empty tile); sta_stats_finalize folds them into
 *   [B][Cout][2] — the statistics of the GroupNorm that consumes `out` (sta_groupnorm_silu_nhwc_cstats), whose own statistics pass drops;
 *   packed_w: the weight re-laid out once per model by sta_conv3x3_pack_w (element (o, i, ky, kx) of the source at
 *   o*so + i*si + ky*sy + kx*sx, strides in elements: any memory format of a [Cout][Cin][3][3] tensor);
 *   up2 = 1: the input is the nearest-neighbour 2x upsampling of x (Upsample.forward), read through (y >> 1, x >> 1) — the
 *   upsampled tensor never exists.
 * sta_conv3x3_nhwc_supported: W % 32 == 0 and H % 8 == 0, or W == 16 and H % 16 == 0, or H == W == 8; Cin % 64 == 0; Cout % 160 == 0 or
 * Cout % 128 == 0 (a workgroup owns 160 or 128 output channels);
 * out below 4 GiB. Everything else stays with the library convolution.
 */
int sta_conv3x3_nhwc_supported(int B, int H, int W, int Cin, int Cout);
int sta_conv3x3_stats_slots(int H, int W);
int sta_stats_finalize(const float* partial, float* stats, int B, int slots, int C, void* stream);
size_t sta_conv3x3_packed_w_bytes(int Cin, int Cout);
int sta_conv3x3_pack_w(const void* w, long so, long si, long sy, long sx, void* packed, int Cin, int Cout, int dtype, void* stream);
int sta_conv3x3_nhwc(const void* x, const void* packed_w, const void* zeros, const void* bias, const void* res, void* out, float* stats, int B,
                     int H, int W, int Cin, int Cout, int up2, int dtype, void* stream);

/*
 * The Linear layers / 1x1 convolutions of the transformer blocks and ResBlock skips as one row GEMM (csrc/sta_gemm.hip; reference
 * CrossAttention.to_q / to_k / to_v / to_out attention.py:158-215, FeedForward :48-69, SpatialTransformer.proj_in / proj_out :322-333,
 * ResBlock.skip_connection openaimodel.py:196-206):
 *     out[r][n] = sum_k x[r][k] w[n][k] + bias[n] + res[r][n]
 *   x: [R][K] dtype (tokens of a [B, N, C] tensor or NHWC pixels);  out, res: [R][N] dtype (res may be NULL, may not alias out);
 *   bias: [N] dtype or NULL;  zeros: >= 2 * K bytes of zeros;  packed_w: sta_linear_rows_pack_w of the weight (element (n, k) at
 *   n*sn + k*sk, strides in elements), once per model.
 * Supported: K % 64 == 0; N % 160 == 0 or N % 128 == 0; out below 4 GiB. Outside autograd only.
 */
int sta_linear_rows_supported(long R, int K, int N);
size_t sta_linear_rows_packed_w_bytes(int K, int N);
int sta_linear_rows_pack_w(const void* w, long sn, long sk, void* packed, int K, int N, int dtype, void* stream);
int sta_linear_rows(const void* x, const void* packed_w, const void* zeros, const void* bias, const void* res, void* out, long R, int K,
                    int N, int dtype, void* stream);
/* ... storing per-(image, slot) partial sums / sums of squares of the stored values to stats [R / rows_per_image + 1][rows_per_image / 64][N][2]
 * fp32 as sta_conv3x3_nhwc does (fold with sta_stats_finalize): rows_per_image % 256 == 0 (every 256-row tile inside one image) */
int sta_linear_rows_stats(const void* x, const void* packed_w, const void* zeros, const void* bias, const void* res, void* out, float* stats,
                          long rows_per_image, long R, int K, int N, int dtype, void* stream);
/* ... with x the column concatenation [xa | xb] of two row tensors ([R][Ka] and [R][K - Ka], Ka % 64 == 0) read in place — the 1x1
 * skip convolution over `torch.cat([h, skip], dim=1)` of the UNet's output blocks (openaimodel.py:740) without the concatenated tensor */
int sta_linear_rows_cat(const void* xa, const void* xb, int Ka, const void* packed_w, const void* zeros, const void* bias, const void* res,
                        void* out, long R, int K, int N, int dtype, void* stream);

/*
 * y = a + b + bias[c]  over NCHW tensors [B][C][HW] (HW % 8 == 0); b and/or bias may be NULL.
 * Replaces a convolution's separate bias pass plus the residual add that follows it
 * (`skip_connection(x) + out_layers(h)`, openaimodel.py ResBlock._forward; `proj_out(x) + x_in`, attention.py:346).
 */
int sta_add_bias_nchw(const void* a, const void* b, const void* bias, void* y, int B, int C, int HW,
                      int dtype, void* stream);

/*
 * NHWC ("channels_last") variants: x, y [B][HW][C]. GroupNorm needs a caller-owned workspace of
 * sta_groupnorm_nhwc_workspace_bytes(B, HW, G) bytes (per-chunk partial moments; written then read inside the
 * call, no initialisation needed). Limits: C % 8 == 0, C / G >= 8 or == 4 (an 8-channel column spans at most two groups), C <= 4096, G <= 64.
 * MIOpen's bf16 convolutions are NHWC kernels; keeping the UNet trunk in NHWC removes the two layout transposes
 * MIOpen wraps around every NCHW convolution and makes 'b c h w -> b (h w) c' (attention.py:338) a view.
 */
size_t sta_groupnorm_nhwc_workspace_bytes(int B, int HW, int G);
int sta_groupnorm_silu_nhwc(const void* x, const float* add, const void* gamma, const void* beta, void* y,
                            void* workspace, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream);
/* ... of the channel concatenation [xa | xb] (Ca and C - Ca channels, Ca % 8 == 0) read in place: y is the normalised concatenated
 * tensor [B][HW][C]; `torch.cat([h, skip], dim=1)` in front of an output block's first GroupNorm never exists */
int sta_groupnorm_silu_nhwc_cat(const void* xa, const void* xb, int Ca, const float* add, const void* gamma, const void* beta, void* y,
                                void* workspace, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream);
/* ... with the statistics ALREADY accumulated per channel by the producing kernel (stats_a [B][Ca][2], stats_b [B][C - Ca][2] or NULL
 * with xb; fp32 sum and sum of squares over the HW pixels): one pass instead of two. xb = NULL, Ca = C: a single input tensor. */
int sta_groupnorm_silu_nhwc_cstats(const void* xa, const void* xb, int Ca, const float* stats_a, const float* stats_b, const float* add,
                                   const void* gamma, const void* beta, void* y, int B, int C, int HW, int G, float eps, int silu, int dtype,
                                   void* stream);

/* y = a + b + bias (row-broadcast) over [rows][C] tensors (NHWC activations / token tensors); b, bias may be NULL. */
int sta_add_bias_rows(const void* a, const void* b, const void* bias, void* y, long rows, int C, int dtype, void* stream);

/*
 * Row-wise dynamic quantisation to OCP fp8 (e4m3fn) for the fp8-weight GEMMs of BASELINE configs[4]:
 *   scale[r] = max_c |x[r, c]| / 448 (1 if the row is all zero);  xq[r, c] = e4m3( x[r, c] / scale[r] )
 *   x [rows][C] dtype, xq [rows][C] bytes, scale [rows] fp32;  C % 8 == 0, C <= 5120.
 * Stands in front of the nn.Linear layers of the transformer blocks (reference attention.py:50,69,164-171) when their
 * weights are stored as e4m3 with one fp32 scale per output channel: the GEMM is then hipBLASLt's row-scaled e4m3 GEMM (a
 * library call; non-block-scaled fp8 runs at the 16-bit MFMA rate on gfx950, so this is a weight-memory option, not a
 * rate option: DESIGN.md section 5) and the two scale vectors are applied to its fp32 accumulators.
 */
int sta_quant_rows_fp8(const void* x, void* xq, float* scale, long rows, int C, int dtype, void* stream);

/*
 * Input gradients of the glue kernels, for the tracked (weight-optimisation) epochs (plms.py:220-277): the blend weights
 * are the only leaf and every model parameter is frozen, so each op needs d(input) only. csrc/sta_unet_bwd.hip.
 *
 * sta_groupnorm_silu_nhwc_bwd: dx of y = act(GroupNorm_G(x + add) * gamma + beta) on NHWC activations. fwd_workspace is the
 *   workspace the FORWARD call filled for the same x (its per-chunk moments; mean / rstd are rebuilt from it), bwd_workspace
 *   another sta_groupnorm_nhwc_workspace_bytes(B, HW, G) bytes. Same limits as the forward. (add gets no gradient: the
 *   timestep embedding does not depend on the blend weights.)
 * sta_geglu_bwd: dx [R][2D] of y = x[:, :D] * gelu(x[:, D:]) given dy [R][D].
 * sta_layernorm_bwd: ds = dLayerNorm(s)^T dy + dres for y = LayerNorm(s) * gamma + beta; s, dy, dres (may be NULL: the
 *   gradient reaching s through the residual connection), ds [R][C]; C % 8 == 0, C <= 2048.
 */
int sta_groupnorm_silu_nhwc_bwd(const void* x, const float* add, const void* gamma, const void* beta, const void* dy, void* dx,
                                const void* fwd_workspace, void* bwd_workspace, int B, int C, int HW, int G, float eps, int silu,
                                int dtype, void* stream);
int sta_geglu_bwd(const void* x, const void* dy, void* dx, long R, int D, int dtype, void* stream);
int sta_layernorm_bwd(const void* s, const void* gamma, const void* dy, const void* dres, void* ds, long R, int C, float eps,
                      int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
