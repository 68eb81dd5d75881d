/*
 * sta_xattn.h — C-ABI of the MI355X (gfx950) spatial-temporal cross-attention library.
 *
 * The reference (UCSB-NLP-Chang/Diffusion-SpaceTime-Attn) has no FFI layer: its hot path is the
 * eager-PyTorch sequence in attention_optimization/stable-diffusion/ldm/modules/attention.py.
 * Every entry point below therefore cites the Python statements it replaces; INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every tensor is a raw DEVICE pointer, contiguous, row-major;
 *   - the caller (PyTorch) owns every buffer; the library never allocates, frees or retains;
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*), never syncs;
 *   - return 0 on success, a negative STA_E_* code otherwise; sta_last_error() gives the text
 *     (thread-local). Nothing throws or aborts.
 *   - dtype: STA_BF16 or STA_F16 for q / k / v / out / dout / dq; coef, dcoef, maps are fp32.
 *
 * Context order everywhere ("n_ctx = K + 2"):
 *   ctx 0      = unconditional prompt ""  — attended by batch row 0 (uncond half of the CFG batch)
 *   ctx 1      = global prompt             — attended by batch row 1 (cond half)
 *   ctx 2 + i  = local prompt of object i  — attended by batch row 1, blended inside disc i
 * This is the set of (row, context) pairs that reach the output of
 * BasicTransformerBlock._forward (attention.py:278-294); gs_i[0] is computed there and never used.
 */
#ifndef STA_XATTN_H
#define STA_XATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STA_VERSION 0x000500 /* 0.5.0: one LDS-resident backward kernel for every head dim (round 6); 0.4.0: fragment-order entry points, trunk convolution / row GEMM / producer-side GroupNorm statistics (round 4) */

enum { STA_BF16 = 0, STA_F16 = 1 };

enum {
  STA_OK = 0,
  STA_E_ARG = -1,     /* bad argument (null pointer, shape out of the supported range) */
  STA_E_UNSUP = -2,   /* shape/dtype valid but not supported by this build */
  STA_E_LAUNCH = -3   /* HIP runtime reported an error at launch */
};

#define STA_MAX_KEYS 80     /* M <= 80 (CLIP: 77)                     */
#define STA_MAX_HEAD_DIM 160 /* d = C / heads <= 160, d % 8 == 0     */
#define STA_MAX_OBJECTS 8   /* K <= 8                                 */

/* The `hipcc --version` headline of the toolchain that built this library ("" if the build did not record it). The level-0 kernels sit on
 * MFMA hazard windows measured with one hipcc release; the host side (sta.lib.toolchain_validated) checks a library built by another
 * release numerically before its first use instead of trusting the lint alone. */
const char* sta_built_with(void);

/* Library version (STA_VERSION of the build). */
int sta_version(void);

/* Text of the last error on the calling thread ("" if none). Never NULL. */
const char* sta_last_error(void);

/*
 * Override of the launch heuristics (which forward kernel, workgroup shape, pixel tiles per workgroup, block
 * map). Nothing in the reference corresponds to it: it exists so that tests reach every kernel variant at
 * small sizes and tools/ can A/B variants; the product never calls it and the library reads no environment
 * variables. Process-global, takes effect at the next launch; value 0 restores the automatic choice.
 */
enum {
  STA_OPT_FWD_KERNEL = 0,   /* 1: LDS-resident ("staged") kernel, 2: wave-per-context ("split") kernel */
  STA_OPT_STAGED_TILES = 1, /* pixel tiles a staged workgroup walks (forward: 1..12, backward: 1..16) */
  STA_OPT_STAGED_WAVES = 2, /* waves per staged workgroup: 4, 8 or 12 (sta_xattn_fwd_proj, one head per workgroup: 4 or 8) */
  STA_OPT_STAGED_QT = 3,    /* 2: two 16-pixel sub-tiles per wave */
  STA_OPT_HEAD_MAJOR = 4,   /* 1: block b -> head b % heads; 2: XCD-contiguous tile ranges */
  STA_OPT_SPLIT_QT = 5,     /* sub-tiles per wave of the split kernel: 1, 2, 4 */
  STA_OPT_SELFATTN_32 = 6,  /* experiment builds only (-DSTA_EXPERIMENT_SELFATTN32): 2 = keep the 16x16x32-MFMA self-attention kernel at d = 40 */
  STA_OPT_PROJ_PAIR = 7,    /* sta_xattn_fwd_proj: 1 = head-pair kernel whenever the shape allows, 2 = one head per workgroup */
  STA_OPT_SELFATTN_WAVES = 8, /* sta_selfattn_fwd at d <= 48, log2-domain path: 8 = eight waves x one query tile (four waves per SIMD; measured
                                 slower, kept for tests / tools), anything else = four waves x two tiles */
  STA_OPT_SELFATTN_PIPE = 9, /* sta_selfattn_fwd at d = 40, 8 heads, log2-domain q, N % 64 == 0: 2 = the plain loop instead of the software-pipelined one; 3 = three query tiles per wave (192 queries per workgroup), 4 / 8 = four / eight waves x two tiles */
  STA_OPT_PROJ_LL2 = 10,    /* sta_xattn_fwd_proj where Wq + every context do not fit a CU's LDS but Wq + the two mandatory ones do (SD-v1 level 1,
                               C = 640, d = 80): 2 = refuse (the block then takes the GEMM + sta_xattn_fwd); default: local contexts from L2.
                               Also sta_xattn_fwd's LDS-resident kernel at d > 64 when a head's K + 2 contexts exceed a CU's LDS (levels 2 / mid,
                               level 1 from K = 4): 2 = contexts staged in groups per tile (the round-1 variant); default: the two mandatory
                               contexts resident, local ones as MFMA operands from L2, several tiles per workgroup */
  STA_OPT_BWD_KERNEL = 11,  /* unused since 0.5.0 (one backward kernel) */
  STA_OPT_BWD_SLOTS = 12,   /* sta_xattn_bwd: at most this many contexts in LDS (>= 2), the other local ones from L2 */
  STA_OPT_BWD_WAVES = 13,   /* experiment builds only (-DSTA_EXPERIMENT_BWD_WIDE): 8 = eight waves per workgroup */
  STA_OPT_COUNT = 14
};
int sta_set_option(int key, int value);

/*
 * Bytes of the packed K/V image for n_ctx contexts, `heads` heads of dim d (0 if unsupported).
 * The image holds, per (ctx, head), the MFMA operand fragments of K and V in the exact lane order
 * the kernels consume (DESIGN.md §"HBM layout"), for both forward and backward.
 */
size_t sta_xattn_packed_kv_bytes(int n_ctx, int heads, int d);

/*
 * Pack projected keys/values into the fragment image. Replaces the per-call
 *   k = self.to_k(context); v = self.to_v(context); rearrange(... '(b h) n d')   attention.py:180-183
 * re-layout (the projections themselves stay a GEMM outside this library). K and V do not depend
 * on the timestep, so the host calls this once per prompt per block, not once per UNet call.
 *   k, v   : [n_ctx][M][C]  dtype
 *   packed : sta_xattn_packed_kv_bytes(n_ctx, heads, C/heads) bytes, 16-byte aligned
 */
int sta_xattn_pack_kv(const void* k, const void* v, void* packed,
                      int n_ctx, int M, int C, int heads, int dtype, void* stream);

/*
 * Fused forward: for every pixel p and head h
 *   A_u = softmax(scale * q[0,p,h] K_0^T) V_0          (row 0, ctx 0)
 *   A_c = softmax(scale * q[1,p,h] K_1^T) V_1          (row 1, ctx 1)
 *   A_i = softmax(scale * q[1,p,h] K_{2+i}^T) V_{2+i}  (row 1, local i; only needed where mask_i[p])
 *   out[0,p] = A_u
 *   out[1,p] = A_c + sum_i coef[i] * mask[i,p] * (A_i - A_u)
 * i.e. the pre-projection form of attention.py:278-294 (CrossAttention.forward :175-197 for each
 * context + the masked blend :284-294). Because to_out is affine and mask is per-pixel, applying
 * to_out once to `out` equals the reference's post-projection blend (bias cancels in the difference).
 * Batching: the reference handles one image per call (n_samples must be 1, attention.py:282). Prompts are
 * independent, so this library takes n_img >= 1 images per launch; every tensor below gains a leading
 * image axis ([n_img][...]) and all images of a launch share K (prompts are grouped by object count).
 *   q      : [2][N][C] dtype   (to_q(norm2(x)), attention.py:178)
 *   packed : image from sta_xattn_pack_kv for n_ctx = n_img * (K + 2) (image-major)
 *   mask   : [N] uint8 bit field, bit i set iff pixel n lies inside disc i (attention.py:251-262; the
 *            K boolean [dim,dim] masks of the reference packed into one byte per pixel, K <= 8);
 *            may be NULL iff K == 0
 *   coef   : [K] fp32 device (W[:, step], plms.py:243);      may be NULL iff K == 0
 *   out    : [2][N][C] dtype
 *   maps   : NULL, or [K+2][heads][N][M] fp32 — the softmax probabilities ("attn", attention.py:194)
 *            for parity checks; never passed in timed runs. With maps != NULL no local context is
 *            skipped outside its disc.
 *   scale  : dim_head ** -0.5 (attention.py:163)
 */
int sta_xattn_fwd(const void* q, const void* packed, const uint8_t* mask, const float* coef,
                  void* out, float* maps,
                  int n_img, int N, int C, int heads, int M, int K, float scale, int dtype, void* stream);

/*
 * Forward with the QUERY PROJECTION inside: sta_xattn_fwd plus the GEMM in front of it,
 *   q = self.to_q(x)            attention.py:178  (x = norm2(hidden); the reference recomputes it K+1 times)
 * so the [2][N][C] query tensor is never written to or read from HBM. Inference only (no backward: the tracked
 * epochs keep q for sta_xattn_bwd). Operands are packed once:
 *   sta_xattn_pack_wq       to_q.weight [C][C] (bias-free Linear, attention.py:164) -> per-head MFMA fragments;
 *                           once per model, sta_xattn_packed_wq_bytes(C, heads) bytes
 *   sta_xattn_pack_kv_proj  like sta_xattn_pack_kv, forward-only image whose K fragments carry the head dim in
 *                           projected-query order; once per prompt per block,
 *                           sta_xattn_packed_kv_proj_bytes(n_ctx, heads, d) bytes
 *   sta_xattn_fwd_proj      y: [n_img][2][N][C] = norm2(hidden); everything else as sta_xattn_fwd (no maps output)
 * Supported when sta_xattn_fwd_proj_supported(C, heads, M, K) != 0: d <= 96, C % 160 == 0, 64 < M <= 80 and the head's
 * Wq slice plus all K+2 contexts fit the 160 KiB LDS of a CU (SD-v1 level 0, C = 320: K <= 4). Other shapes take
 * the GEMM + sta_xattn_fwd. At d = 40 with K <= 2 and an even head count a workgroup serves a head PAIR from one read
 * of the y rows (64 < M <= 77; per-(ctx, head) operand images of 13952 bytes in which one 16- or 8-byte LDS read is one
 * MFMA operand, packed alongside by the same two pack calls; csrc/sta_xattn_proj3.h holds the layout).
 */
int sta_xattn_fwd_proj_supported(int C, int heads, int M, int K);
size_t sta_xattn_packed_wq_bytes(int C, int heads);
int sta_xattn_pack_wq(const void* wq, void* packed, int C, int heads, int dtype, void* stream);
size_t sta_xattn_packed_kv_proj_bytes(int n_ctx, int heads, int d);
int sta_xattn_pack_kv_proj(const void* k, const void* v, void* packed,
                           int n_ctx, int M, int C, int heads, int dtype, void* stream);
int sta_xattn_fwd_proj(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                       const float* coef, void* out,
                       int n_img, int N, int C, int heads, int M, int K, float scale, int dtype, void* stream);
/* != 0 iff sta_xattn_fwd_proj takes this shape with the Wq slice and the two mandatory contexts resident in LDS and the K local
 * contexts read from L2 as MFMA operands (SD-v1 level 1: C = 640, d = 80, K >= 1; attention.py:178 + :175-197 + :278-294 as above). */
int sta_xattn_fwd_proj_locals_from_l2(int C, int heads, int M, int K);

/*
 * sta_xattn_fwd_proj reading y in QUERY-FRAGMENT order (include/sta_unet.h: sta_add_layernorm_qfrag writes it): y = norm2(hidden)
 * has one consumer at SD-v1 level 0 — this launch (attention.py:279-281 feed :178) — so its layout is private to the pair
 * of kernels, like the packed K / V / Wq images. Per 16-pixel group and batch row C/32 fragments of 1 KiB, fragment s holding
 * at byte (16 g + c) * 16 the values y[16 P + c][32 s + 8 g .. + 7]: one load instruction of a wave is one coalesced KiB
 * and IS the MFMA B operand (row-major y costs 16 half-used 128-byte lines per instruction, or a DPP hand-over).
 * Results are bit-identical to sta_xattn_fwd_proj on the same values (bf16 at SD-v1 level 0, all three layouts: the head-pair kernel folds
 * scale * log2(e) into its Wq fragments (W' = round16(W scale log2 e)) and takes its softmax without the running maximum, with a per-context range check of the
 * denominator that falls back to the standard softmax — same result class, exact for any logits). Taken by the head-pair kernel —
 * sta_xattn_fwd_proj_qfrag_supported(n_img, N, C, heads, M, K) != 0 iff d = 40, C in {160, 320}, K <= 2, 64 < M <= 77,
 * N % 16 == 0 and the launch has >= 256 pair workgroups — and by the locals-from-L2 kernel of SD-v1 level 1 (C = 640, d = 80,
 * K >= 1, N % 16 == 0: Wq slice + the two mandatory contexts fill the CU's LDS, local contexts are MFMA operands read from L2).
 */
int sta_xattn_fwd_proj_qfrag_supported(int n_img, int N, int C, int heads, int M, int K);
int sta_xattn_fwd_proj_qfrag(const void* y_frag, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                             const float* coef, void* out,
                             int n_img, int N, int C, int heads, int M, int K, float scale, int dtype, void* stream);

/*
 * The chain's last private layout: sta_xattn_fwd_proj_qfrag_ofrag = sta_xattn_fwd_proj_qfrag whose OUTPUT leaves in
 * OUT-FRAGMENT order, the MFMA B-operand order of sta_to_out_ln_ofrag below (per 16-pixel group and batch row ten 1-KiB
 * fragments: [2 pr], [2 pr + 1] = heads 2 pr / 2 pr + 1 of pair pr, 32 channels each, [8 + q] = the 8 remaining channels of the
 * four heads of pairs 2q, 2q + 1; csrc/sta_xattn_proj3.h::ofrag_channel names the channel behind every slot): every store
 * instruction of the attention kernel is one contiguous KiB instead of 160-byte pair segments at a 640-byte stride. C = 320 only.
 *
 * sta_to_out_ln_ofrag — the rest of the block's cross-attention section in one pass (csrc/sta_rowgemm.hip):
 *     s = x + blended . to_out.weight^T + to_out.bias          attention.py:215 (to_out), :294 (residual)
 *     y = LayerNorm(s) * gamma + beta                           attention.py:299 (norm3, feeding ff)
 * with `blended` read in out-fragment order and to_out's [R][C] result never written to HBM (row-major: a library GEMM that
 * writes it + sta_add_layernorm that reads it back). The weight is re-laid out once per model by sta_to_out_ln_pack_wo
 * (sta_to_out_ln_packed_wo_bytes(C, heads) bytes; 0 = unsupported: C = 320 with 8 heads only) and streamed through LDS.
 *   x, s, y : [R][C] dtype, row-major (s: the new residual stream, rounded to dtype before it is normalised, as sta_add_layernorm does)
 *   bias    : [C] dtype or NULL;  gamma, beta: [C] dtype;  R % 16 == 0
 */
int sta_xattn_fwd_proj_qfrag_ofrag(const void* y_frag, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                                   const float* coef, void* out_frag,
                                   int n_img, int N, int C, int heads, int M, int K, float scale, int dtype, void* stream);

/*
 * sta_xattn_fwd_proj / _qfrag / _qfrag_ofrag behind ONE entry point (`layout` 0 / 1 / 2) plus the head-pair kernel's STATISTICS AND
 * ADAPTIVE SWITCH (round 6). The pair kernel's softmax is optimistic (no running maximum; a per-context range check of the denominator
 * sends a wave through the standard softmax when it fails) — tuned to logits whose largest score lies in a window (fp16: about -3.5 .. +10
 * nats), which holds for the synthetic weights of the bench and is NOT known for real SD-v1-4 weights (a large BOS-token logit is typical).
 * `stats`: NULL (the three entry points above: always optimistic, nothing counted), or STA_P3_STATS_WORDS uint32 words in device memory,
 * zeroed once by the caller and then owned by the library across launches on one stream:
 *   [0] launches still to sit the optimistic softmax out     [1] sampled workgroups of this launch that have finished
 *   [2], [3] this launch's wave-level context evaluations / fall-backs in the SAMPLE (every 8th workgroup counts: 2048 workgroups adding
 *   to the same words cost 6 us per launch)      [4], [5] totals of [2], [3]      [6] launches counted     [7] launches that sat out
 * The last sampled workgroup of every launch folds the counts and, when more than an eighth of the evaluations fell back, makes the next 64
 * launches run the standard softmax only (device side: no host synchronisation, valid inside a captured graph) — hostile logits then cost
 * the standard kernel's time, not both paths per context. Shapes that do not take the pair kernel ignore `stats`.
 */
#define STA_P3_STATS_WORDS 8
int sta_xattn_fwd_proj_ex(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                          const float* coef, void* out,
                          int n_img, int N, int C, int heads, int M, int K, float scale, int dtype, int layout, void* stats, void* stream);
size_t sta_to_out_ln_packed_wo_bytes(int C, int heads);
int sta_to_out_ln_pack_wo(const void* wo, void* packed, int C, int heads, int kind, int dtype, void* stream);
int sta_to_out_ln_ofrag(const void* blended_ofrag, const void* packed_wo, const void* bias, const void* x,
                        const void* gamma, const void* beta, void* s, void* y,
                        long R, int C, int heads, float eps, int y_qfrag, int dtype, void* stream);
/*
 * The same pass behind the SELF-attention of the block (attn1: attention.py:274 `x = attn1(norm1(x)) + x`, then norm2 at :279):
 * sta_selfattn_fwd_sfrag = sta_selfattn_fwd whose output leaves in its own out-fragment order (per 16-pixel group ten 1-KiB
 * fragments: [h] = dims 0..31 of head h, [8 + q] = dims 32..39 of heads 4q..4q+3, one lane row each; C = 320, 8 heads,
 * N % 16 == 0), sta_to_out_ln_pack_wo(kind = 1) lays attn1.to_out.weight out for that order (kind = 0: the cross-attention
 * kernel's), and y_qfrag = 1 makes sta_to_out_ln_ofrag write y = norm2(s) in QUERY-fragment order — the input layout of
 * sta_xattn_fwd_proj_qfrag*: attn1.to_out's result and norm2's row-major output never exist in HBM either.
 */
int sta_selfattn_fwd_sfrag(const void* q, const void* k, const void* vt, void* out_frag, int B, int N, int C, int heads,
                           int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype, void* stream);

/* Bytes of fp32 workspace sta_xattn_bwd needs for the given shape (deterministic dcoef reduce). */
size_t sta_xattn_bwd_workspace_bytes(int n_img, int N, int heads, int K);

/*
 * Backward of sta_xattn_fwd w.r.t. q and coef (K/V/context gradients are not produced: the text
 * embeddings and projection weights are constants of the optimisation, plms.py:204-214; the
 * reference computes and discards them, diffusionmodules/util.py:140-145).
 *   dout   : [2][N][C] dtype — gradient w.r.t. `out`
 *   dq     : [2][N][C] dtype
 *   dcoef  : [K] fp32 (overwritten, not accumulated)
 *   workspace : sta_xattn_bwd_workspace_bytes(...) bytes (contents undefined on entry and exit)
 */
int sta_xattn_bwd(const void* q, const void* packed, const uint8_t* mask, const float* coef,
                  const void* dout, void* dq, float* dcoef, void* workspace,
                  int n_img, int N, int C, int heads, int M, int K, float scale, int dtype, void* stream);

/*
 * Flash-style self-attention of the same transformer block (attn1; CrossAttention.forward with
 * context = x, attention.py:175-197 called at :274): out = softmax(scale q k^T) v per head
 * without materialising the [heads, N, N] scores. SURVEY.md §8f rank 2.
 *   q   : [B][N][ldq]  dtype, head h in columns h*d .. h*d+d-1 (ldq >= C lets q live in a fused QKV buffer)
 *   k   : [B][N][ldk]  dtype
 *   vt  : V TRANSPOSED (the host computes W_v x^T instead of x W_v^T): element (b, c, n) at
 *         b*vt_batch_stride + c*vt_row_stride + n (elements; both multiples of 8). [B][C][N] is (N, C*N);
 *         ONE plain GEMM W_v . X^T over the flattened batch gives [C][B*N], i.e. (B*N, N) — no batched GEMM.
 *   out : [B][N][C]    dtype
 * Requires N % 8 == 0, d = C/heads <= 160, d % 8 == 0.
 * scale: the softmax scale (dim_head^-0.5). Passing scale = ln 2 (0.693147...) declares that q ALREADY carries
 * scale * log2(e) (fold it into W_q): the kernel then takes exp2 of the MFMA result directly (its running maximum is the
 * accumulators' initial value) — 11 % faster at d = 40 / 80, same result up to the one rounding of q.
 */
int sta_selfattn_fwd(const void* q, const void* k, const void* vt, void* out, int B, int N, int C, int heads,
                     int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype, void* stream);

/*
 * The same forward with an OPTIMISTIC softmax, both 16-bit types at the shapes of the software-pipelined kernel (SD-v1 level 0: C = 320,
 * 8 heads, N % 64 == 0, q in log2 units — sta_selfattn_optimistic_supported). The key loop of a workgroup starts at its OWN 64-key block
 * and wraps around; behind a query tile's first key block (whose exact maximum m_0 is subtracted) the loop keeps NO running maximum — a
 * tenth of its instructions, in a loop whose matrix and vector cycles add up (profiles/r05_level0.md): P = exp2(S - m_0) has fp32's
 * exponent range in bf16 and 2^16 of headroom over the query's own neighbourhood in fp16. Every denominator is range-checked at the end
 * ([2^-100, 2^100) bf16, [2^-100, 2^15) fp16: one P at fp16's largest number is enough to fail); a workgroup that fails writes
 * flags[workgroup] = 1 and the SECOND launch this call issues — the standard loop — recomputes exactly those workgroups (every other one
 * returns at once): exact for any logits. A call in which more than an eighth of the workgroups failed switches the optimistic loop off
 * for the next 64 calls on the same flags buffer (they cost the standard loop plus an empty launch), so activations that do not suit it
 * cost 1 / 65 of a launch on average, not a second loop per call. flags: sta_selfattn_optimistic_flags_bytes(B, N, heads) bytes of
 * device memory, caller-owned, ZERO before the first call (its first two words carry that state from call to call on one stream; the per-workgroup words start 128 bytes in), then
 * left alone. sfrag != 0: output in out-fragment order as sta_selfattn_fwd_sfrag.
 */
int sta_selfattn_optimistic_supported(int N, int C, int heads, float scale, int dtype);
size_t sta_selfattn_optimistic_flags_bytes(int B, int N, int heads);
int sta_selfattn_fwd_optimistic(const void* q, const void* k, const void* vt, void* out, void* flags, int B, int N, int C, int heads,
                                int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype, int sfrag, void* stream);

/*
 * The same forward for the differentiable path (the weight-optimisation epochs: plms.py:275-277 back-propagates
 * through attn1 of every checkpointed block, diffusionmodules/util.py:123-145): additionally writes
 *   lse : [B][heads][N] float32 = log2 of the row sums of exp(scale q k^T), i.e. P = exp2(scale*log2(e)*s - lse),
 * which is all the backward needs of the N x N scores.
 */
int sta_selfattn_fwd_lse(const void* q, const void* k, const void* vt, void* out, float* lse, int B, int N, int C, int heads,
                         int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype, void* stream);

/*
 * Backward of the self-attention above (replaces the autograd of attention.py:175-197 with context = x; the
 * reference materialises and differentiates the [heads, N, N] softmax): given dout, the forward's out and lse,
 *   dv = P^T dout,  dS = P o (dout v^T - delta),  delta = rowsum(dout o out),  dq = scale dS k,  dk = scale dS^T q
 * per head, P rebuilt from lse tile by tile (three launches: delta, dk/dv, dq; no atomics, deterministic).
 *   q, k, v      : [B][N][ld]  dtype rows, head h in columns h*d.. (ld >= C: slices of one fused QKV buffer are fine)
 *   qt, kt       : [B][C][N]   dtype, q and k transposed (contiguous)
 *   dout, out    : [B][N][C]   dtype (contiguous);   doutt : [B][C][N] = dout transposed
 *   lse          : [B][heads][N] float32 from sta_selfattn_fwd_lse;   delta : [B][heads][N] float32 workspace
 *   dq, dk, dv   : [B][N][ldg] dtype rows (ldg >= C, e.g. the three column blocks of one [B][N][3C] gradient)
 * Requires N % 64 == 0, d = C/heads <= 160, d % 8 == 0, ld % 8 == 0, ldg % 4 == 0.
 */
int sta_selfattn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* dout,
                     const void* doutt, const void* out, const float* lse, float* delta, void* dq, void* dk, void* dv,
                     int B, int N, int C, int heads, int ld, int ldg, float scale, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STA_XATTN_H */
