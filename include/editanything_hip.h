/* editanything_hip.h -- C ABI of libeditanything_hip.so (gfx950 / MI355X).
 *
 * The reference (sail-sg/EditAnything) is 100 % Python and has no FFI; its
 * operator swap points are the ones SURVEY.md section 8(b) "B3" lists.  Each entry
 * point below is what a binding for that swap point calls:
 *
 *   ea_attention_f16          <- xformers.ops.memory_efficient_attention(q,k,v)
 *                                ldm/modules/attention.py:223-233, the einsum path
 *                                :163-194, cldm/hack.py:72-111; SAM Attention.forward
 *                                (segment_anything, 3rd party) with decomposed rel-pos
 *   ea_relpos_tables_f16      <- segment_anything add_decomposed_rel_pos (3rd party)
 *   ea_sam_window_attn_f16    <- segment_anything Block.forward's windowed Attention (3rd party), bias fused
 *   ea_groupnorm_f16          <- GroupNorm32 + SiLU, ldm/modules/diffusionmodules/util.py:217-219,
 *                                openaimodel.py:200-204,221-224; attention.py:88-89; model.py:46-47
 *   ea_conv2d_f16             <- conv_nd 3x3/1x1 in ResBlock / Upsample / Downsample
 *                                openaimodel.py:108-152,200-231,254-274; model.py:68-149;
 *                                ControlNet hint block + zero-convs cldm/cldm.py:147-163,281-305
 *   ea_groupnorm_silu_conv3x3 <- ResBlock in_layers / out_layers as one call (openaimodel.py:254-274)
 *   ea_layernorm_rows_f16, ea_gather_add_rows_f32
 *                             <- segment_anything window_partition / window_unpartition around the windowed
 *                                Attention (3rd party), fused into norm1 and the residual add
 *   ea_sam_mask_postprocess   <- Sam.postprocess_masks + utils/amg.py calculate_stability_score /
 *                                batched_mask_to_box (segment_anything, 3rd party), one pass
 *   ea_layernorm_f16, ea_gemm_f16, ea_ln_gemm_f16
 *                             <- BasicTransformerBlock / GEGLU / SpatialTransformer Linears
 *                                ldm/modules/attention.py:49-76,152-160,263-275,316-339
 *   ea_sam_i2t_f16, ea_sam_upscale_tail_f16
 *                             <- segment_anything TwoWayAttentionBlock (image -> token attention + norm4) and
 *                                MaskDecoder.output_upscaling + hypernetwork product (3rd party), fused per token
 *   ea_cfg_ddim_step          <- DDIMSampler.p_sample_ddim, cldm/ddim_hacked.py:187-231
 *   ea_gather_rows            <- the per-step host work of DDIMSampler.ddim_sampling, cldm/ddim_hacked.py:137-169, 181-197
 *   ea_lincomb_f32            <- UniPCMultistepScheduler.step (diffusers, 3rd party; set at sam2image.py:42) and the
 *                                alpha-weighted latent blends of ...inpaint.py:2039-2051
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (PyTorch
 * allocates; the library never allocates, frees or synchronises); activations
 * are fp16 NHWC ("channels last": [B,H,W,C] == tokens [B,H*W,C]); weights are
 * fp16 [N][K] with K contiguous (conv: K = (ky*3+kx)*Cin + cin); biases fp32.
 * `stream` is a hipStream_t.  Return 0 on success, a negative EA_ERR_* code on
 * bad shapes / unsupported configs / launch failure; nothing throws.
 * Stateless and re-entrant: one process per GPU, any number of streams.
 */
#ifndef EDITANYTHING_HIP_H
#define EDITANYTHING_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EA_OK 0
#define EA_ERR_BAD_SHAPE (-1)
#define EA_ERR_BAD_ARG (-2)
#define EA_ERR_UNSUPPORTED (-3)
#define EA_ERR_WORKSPACE (-4)
#define EA_ERR_LAUNCH (-5)

#define EA_ACT_NONE 0
#define EA_ACT_SILU 1
#define EA_ACT_GELU 2  /* exact erf GELU */
#define EA_ACT_GEGLU 3 /* weight rows packed [G/2 value | G/2 gate] per G = geglu_block rows; N_out = N/2 */

/* Fused GEMM/conv epilogue: out = residual + scale*row_scale[m]*act(acc + bias + rowvec[m/rows_per_group]) */
typedef struct ea_epilogue {
  const float* bias;      /* [N] (or [M] if bias_per_row); may be NULL */
  int32_t bias_per_row;
  const float* rowvec;    /* [groups][rowvec_ld] e.g. time-embedding projection; may be NULL */
  int32_t rowvec_ld;
  int32_t rows_per_group; /* rows (pixels) per sample */
  int32_t act;            /* EA_ACT_* */
  float scale;            /* ControlNet conditioning scale; 1.0 otherwise */
  const float* row_scale; /* [M] spatial scale map (ControlNetModel2), may be NULL */
  const void* residual;   /* fp16 [M][ldr], may be NULL, may alias out */
  const float* residual32;/* fp32 [M][ldr], may be NULL, may alias out */
  int32_t ldr;
  void* out;              /* fp16 (out_f32=0) or fp32 [M][ldc] */
  int32_t ldc;
  int32_t out_f32;
  int32_t geglu_block;    /* EA_ACT_GEGLU packing granule G: 64 (0 means 64), 80 (needs N % 160 == 0, K % 64 == 0) or
                           * 32 (N % 128 == 0, K % 64 == 0: the register-direct epilogue, plain fp16 output only) */
  /* LayerNorm folded into the contraction (BasicTransformerBlock norm -> Linear, attention.py:271-275).  A is the
   * UN-normalised activation, W's rows carry gamma, bias carries W beta + b, and
   *     out = epilogue(rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n])
   * with mean / rstd from the row partials `ln_stats` the producing launch wrote (`row_stats_out`).  Only launches
   * for which ea_gemm_ln_fold_ok() returns 1 accept it (EA_ERR_UNSUPPORTED otherwise).  All NULL / 0 = off. */
  const float* ln_stats;  /* [ln_parts][M][2]: partial (sum, sum of squares) over the K columns of A's row m */
  int32_t ln_parts;
  const float* ln_colsum; /* [N]: sum over k of the gamma-folded fp16 weight row n */
  float ln_eps;
  /* Row statistics of THIS launch's fp16 output, for the next launch's fold: [ea_row_stats_parts(N)][M][2].  Written by
   * the epilogue where it can, by one extra small launch otherwise (split-K, generic kernel).  NULL = off. */
  float* row_stats_out;
  /* GroupNorm statistics of THIS launch's fp16 output, for the GroupNorm that consumes it (ResBlock in_layers conv ->
   * out_layers GroupNorm, out_layers conv + skip -> SpatialTransformer GroupNorm: openaimodel.py:254-274,
   * attention.py:308-311): per (sample, row chunk, group) partial (sum, sum of squares) of the ROUNDED outputs, written
   * by the epilogue in the layout ea_groupnorm_apply_f16 folds: [B][rows_per_sample / chunk_rows][N / gn_cpg][2] with
   * chunk_rows = ea_gemm_gn_stats_chunk_rows(...).  Only launches for which that query returns > 0 accept it
   * (EA_ERR_UNSUPPORTED otherwise).  NULL = off. */
  float* gn_stats_out;
  int32_t gn_rows_per_sample;   /* output rows (pixels) per sample: M = B * gn_rows_per_sample */
  int32_t gn_cpg;               /* channels per group (>= 8; gn_next_out: a multiple of 4) */
  /* The GroupNorm (+ SiLU) that CONSUMES this launch's fp16 output, applied by the split-K reduction itself (ResBlock
   * in_layers conv -> out_layers GroupNorm -> SiLU, out_layers conv + skip -> the next block's norm, at the 16 x 16 / 8 x 8
   * levels where the contraction is split along K: openaimodel.py:254-274, attention.py:308-311).  The reduction kernel is
   * laid out one workgroup per (sample, group): it sums the fp32 slices, applies bias / row vector / residual, writes the
   * fp16 output `out`, takes the group's statistics from those ROUNDED values while they are still in registers and writes
   * gn_next_out[m][n] = act((out[m][n] - mean) * rstd * gamma[n] + beta[n]) -- no statistics pass, no normalise pass.
   * Uses gn_rows_per_sample / gn_cpg above.  Only launches for which ea_gemm_gn_next_ok() returns 1 accept it
   * (EA_ERR_UNSUPPORTED otherwise).  gn_next_out NULL = off. */
  void* gn_next_out;            /* fp16 [M][ldc] */
  const float* gn_next_gamma;   /* [N] */
  const float* gn_next_beta;    /* [N] */
  float gn_next_eps;
  int32_t gn_next_silu;
  /* K-CONCATENATED SPLIT OPERANDS (the fp32-accurate SAM mode, sam_exact.py: an exact Linear as ONE launch).  With
   * A = [x_hi | x_lo | x_hi] (ea_split3_f32) and W = [W_lo | W_hi | W_hi] the contraction accumulates the two correction
   * products first; after `acc_scale_k` columns of K (a multiple of 64, 0 < acc_scale_k < K) the fp32 accumulators are
   * multiplied by `acc_scale` (2^-11: exact), then the hi x hi product is added:  x W^T = 2^-11 (x_hi W_lo^T + x_lo W_hi^T)
   * + x_hi W_hi^T.  Unsplit launches of the LDS-DMA kernel only (EA_ERR_UNSUPPORTED otherwise).  acc_scale_k = 0: off. */
  int32_t acc_scale_k;
  float acc_scale;
} ea_epilogue;

/* NHWC activation source for a convolution: channel-concat of x1 (c1 ch) and
 * optional x2 (c2 ch, optionally + x2_add), never materialised. */
typedef struct ea_conv_src {
  const void* x1; int32_t c1;
  const void* x2; int32_t c2;
  const void* x2_add;
  int32_t B, Hin, Win;
  int32_t ksize;   /* 1 or 3 */
  int32_t stride;  /* 1 or 2 */
  int32_t pad;     /* low-side zero padding (high side is implied by Hout/Wout) */
  int32_t ups;     /* 1: conv reads the nearest-2x upsampled input */
  int32_t Hout, Wout;
} ea_conv_src;

/* Tuning / A-B knobs of the contraction entry points, for tools and tests only (production leaves them zero).  The
 * setting belongs to the CALLING HOST THREAD (thread-local) -- no process-global mutable state, nothing is read from the
 * environment on the launch path.  NULL resets.  Fields: editanything_amd/csrc/ea_gemm.hip. */
typedef struct ea_tuning {
  int32_t force_generic;       /* 1: the register-staged generic kernel for everything */
  int32_t variant;             /* 0 auto, k: force instantiation k of the LDS-DMA kernels (1 / 9: ea_gemm2.h 128- / 64-row tiles,
                                  30: ea_gemm8.h 256 x 256 tiles, 33 / 34: the 64- / 128-row tiles with a 3-stage ring -- two K tiles in flight, for
                                  weights that come from HBM; the rest: tools builds) */
  int32_t splits;              /* 0 plan's own, s: force the split-K factor */
  int32_t bn;                  /* 0 plan's own, 128: force 128-wide column tiles */
  int32_t no_register_direct;  /* 1: LDS-slab epilogue where the register-direct one would run */
  int32_t debug;               /* K-loop / epilogue ablation selector */
} ea_tuning;
int ea_set_tuning(const ea_tuning* t);
/* 1 when the library was built with -DEA_TOOLS=1 (side builds of tools/, the tests' CPU emulation build): the opt-in
 * instantiations (variant 2-8, 10-13) and the `debug` ablation selectors exist; 0 in the shipped library, where forcing one
 * of those variants returns EA_ERR_UNSUPPORTED and `debug` is ignored. */
int ea_tools_build(void);

/* library / device info */
int ea_version(void);
int ea_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* Bytes of fp32 workspace ea_gemm_f16 / ea_conv2d_f16 need for this problem (split-K). */
size_t ea_gemm_workspace_bytes(int M, int N, int K, int batch);

/* LayerNorm fold support (see ea_epilogue): parts a producer with N output columns writes; whether a consumer launch
 * of this shape can apply the fold. */
int ea_row_stats_parts(int N);
int ea_gemm_ln_fold_ok(int M, int N, int K);
/* Rows per GroupNorm-statistics chunk (the wave tile height of the instantiation the planner picks) when a launch of
 * this shape can emit `gn_stats_out` from its epilogue, else 0: unsplit register-direct launches whose wave tiles hold
 * whole groups and whole-sample row ranges.  conv: 1 for ea_conv2d_f16 launches (M = B * Hout * Wout, K = ks*ks*Cin). */
int ea_gemm_gn_stats_chunk_rows(int M, int N, int K, int conv, int rows_per_sample, int cpg);
/* 1 if a launch of this shape (conv != 0: implicit-GEMM convolution) is split along K and its reduction can apply the
 * consuming GroupNorm (ea_epilogue.gn_next_out): whole samples of rows_per_sample rows, groups of cpg channels (cpg % 4 == 0),
 * one (sample, group) slab small enough for one workgroup's registers. */
int ea_gemm_gn_next_ok(int M, int N, int K, int conv, int rows_per_sample, int cpg);

/* C[b] = epilogue(A[b] (MxK, lda) * W[b]^T (NxK, ldw)), b < batch. */
int ea_gemm_f16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int batch,
                long long strideA, long long strideW, long long strideC, long long strideR,
                const ea_epilogue* epi, void* workspace, size_t ws_bytes, void* stream);

/* Implicit-GEMM convolution; M = B*Hout*Wout, K = ksize^2*(c1+c2), W [Cout][K]. */
int ea_conv2d_f16(const ea_conv_src* src, const void* W, int Cout, const ea_epilogue* epi,
                  void* workspace, size_t ws_bytes, void* stream);

/* TWIN launches: two contractions of ONE shape (same M, N, K / same convolution geometry; own operands, weights and
 * epilogues) issued as one grid when both plan onto the same kernel instantiation -- otherwise as two launches back to back.
 * Swap point: cldm/cldm.py:328-341 (ControlLDM.apply_model) runs `control_model(...)` and then `diffusion_model(...)`; the
 * ControlNet trunk (cldm.py:284-305) is a layer-for-layer copy of the UNet encoder (cldm.py:22-33,
 * openaimodel.py:716-760) on the same shapes, so every Linear / conv2d of one has a twin in the other with no dependency
 * between them.  As one grid the pair fills twice the workgroup slots where M is smallest (16 x 16 / 8 x 8 latents).
 * workspace: both problems' split-K partials (2 x ea_gemm_workspace_bytes, each half 256-byte aligned); a smaller workspace
 * makes the call fall back to the two-launch form.  Results are bit-identical to two single calls. */
int ea_gemm_f16_pair(const void* A0, const void* A1, int lda, const void* W0, const void* W1, int ldw, int M, int N, int K,
                     const ea_epilogue* epi0, const ea_epilogue* epi1, void* workspace, size_t ws_bytes, void* stream);
int ea_conv2d_f16_pair(const ea_conv_src* src0, const ea_conv_src* src1, const void* W0, const void* W1, int Cout,
                       const ea_epilogue* epi0, const ea_epilogue* epi1, void* workspace, size_t ws_bytes, void* stream);

/* ---- fp32-ACCURATE SAM (the reference never halves SAM: sam2image.py:69-70, editany_lora.py:87-94) on the fp16 matrix cores:
 * v = hi + 2^-11 lo (hi = fp16(v), lo = fp16(2^11 (v - hi))), products from three fp16 MFMAs with fp32 accumulation.
 * ea_split3_f32: x fp32 [M][K] (act = EA_ACT_GELU: exact erf GELU first, segment_anything MLPBlock) -> fp16 [M][3K] rows
 * [hi | lo | hi], the A operand of a contraction with ea_epilogue.acc_scale_k = 2K.
 * ea_layernorm_split3_f32: LayerNorm (fp32, segment_anything Block.norm1 / norm2, eps 1e-6) fused in front of that split;
 * out_rows (nullable): row m lands in row out_rows[m] of `out` (negative = dropped) -- window_partition's layout.
 * ea_attention_exact_f32: out = softmax(scale q k^T + bias_h[q][kh] + bias_w[q][kw]) v (segment_anything Attention.forward
 * with add_decomposed_rel_pos), q / k / v / out fp32, element (b, i, h, d) at ptr + b*s_b + i*s_n + h*D + d; D in {64, 80};
 * bias tables fp32 [B*H][N][S] (key j -> (j / S, j % S), S <= 64), both NULL = no bias.  Both products run on split
 * operands (the probabilities are split too), the softmax in fp32, online: no score matrix in memory. */
int ea_split3_f32(const float* x, void* out, long long M, int K, int act, void* stream);
int ea_layernorm_split3_f32(const float* x, const float* gamma, const float* beta, float eps, void* out, int M, int C,
                            const int* out_rows, void* stream);
int ea_attention_exact_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int N, int D,
                           long long s_b, long long s_n, long long o_sb, long long o_sn, float scale,
                           const float* bias_h, const float* bias_w, int S, void* stream);

/* GroupNorm (+ optional SiLU) on NHWC fp16, statistics in fp32.
 * workspace: ea_groupnorm_workspace_bytes(B, HW, C, groups). */
size_t ea_groupnorm_workspace_bytes(int B, int HW, int C, int groups);
int ea_groupnorm_f16(const void* x1, int c1, const void* x2, int c2, const void* x2_add,
                     const float* gamma, const float* beta, void* out, int B, int HW, int groups,
                     float eps, int silu, void* workspace, size_t ws_bytes, void* stream);
/* GroupNorm (+SiLU) from statistics a producing contraction left behind (`ea_epilogue.gn_stats_out`): the streaming
 * normalise pass only -- one read, one write, no statistics pass.  partial: [B][nchunk][groups][2]. */
int ea_groupnorm_apply_f16(const void* x, int C, const float* gamma, const float* beta, void* out, int B, int HW,
                           int groups, float eps, int silu, const float* partial, int nchunk, void* stream);

/* ResBlock half: out = epilogue(conv3x3(silu(groupnorm(cat(x1,x2+x2_add))))) -- one CALL, not one kernel: the GroupNorm (+SiLU)
 * launches write the normalised activation to `norm_out`, the convolution launch reads it back (folding the normalise + SiLU into
 * the convolution's operand staging would evaluate it once per tap: nine exponentials per input element against one read + one
 * write of it, DESIGN.md / profiles/HISTORY.md 8d).  What IS fused around it: the statistics of this norm come from the epilogue
 * of the launch that produced x1 where it can emit them (`gn_stats_out`, then the norm is the normalise pass alone), the split-K
 * reduction of a producer applies the consuming norm (`gn_next_out`), and the convolution's epilogue carries bias, the
 * time-embedding row vector, the skip add and the NEXT norm's statistics.
 * `norm_out` is caller scratch [B*Hin*Win*(c1+c2)] fp16 for the normalised activation. */
int ea_groupnorm_silu_conv3x3(const ea_conv_src* src, const float* gamma, const float* beta, int groups,
                              float eps, void* norm_out, const void* W, int Cout, const ea_epilogue* epi,
                              void* workspace, size_t ws_bytes, void* stream);

/* LayerNorm over the last dim of [M][C]; in_f32 selects an fp32 input (residual stream). Output fp16. */
int ea_layernorm_f16(const void* x, int in_f32, const float* gamma, const float* beta, void* out,
                     int M, int C, float eps, void* stream);

/* The same with an output row map: normalised row m is written to row out_rows[m] of `out` (negative = dropped).
 * SAM: norm1 writes straight into the zero-padded window_partition() layout. */
int ea_layernorm_rows_f16(const void* x, int in_f32, const float* gamma, const float* beta, void* out,
                          int M, int C, float eps, const int* out_rows, void* stream);

/* x[t][:] += src[rows[t]][:] for t < T (x fp32 [T][C] in place, src fp16 rows of C, rows[t] < 0 = skip).
 * SAM: window_unpartition() of the attention projection + residual add in one pass. */
int ea_gather_add_rows_f32(float* x, const void* src, const int* rows, int T, int C, void* stream);

/* LayerNorm followed by GEMM (BasicTransformerBlock norm -> to_q / GEGLU proj). `ln_out` scratch [M][K] fp16. */
int ea_ln_gemm_f16(const void* x, int in_f32, const float* gamma, const float* beta, float eps, void* ln_out,
                   const void* W, int ldw, int M, int N, int K, const ea_epilogue* epi,
                   void* workspace, size_t ws_bytes, void* stream);

/* Flash-style attention.  q/k/v element (b, i, h, d) at ptr + b*s_b + i*s_n + h*D + d (fp16).
 * out [B][Nq][H*D] fp16 (row stride o_sn).  Optional decomposed rel-pos bias (SAM):
 * bias_h/bias_w fp32 [B*H][Nq][S]; key j -> (j / S, j % S); S = 0 disables; S <= 32 (windows) or S == 64 (global
 * 64x64 grid; tables 16-byte aligned).  D in {40,64,80,160}; with bias D in {64,80}. */
int ea_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk,
                     int D, long long q_sb, long long q_sn, long long k_sb, long long k_sn, long long v_sb,
                     long long v_sn, long long o_sb, long long o_sn, float scale, const float* bias_h,
                     const float* bias_w, int S, void* stream);

/* SAM windowed attention with the decomposed rel-pos bias computed in-kernel (segment_anything Attention.forward +
 * add_decomposed_rel_pos over one window_partition()ed batch): one workgroup per (window, head), the whole S*S-token
 * window resident in LDS.  q/k/v element (w, i, h, d) at ptr + w*s_b + i*s_n + h*D + d; rel_h/rel_w fp16 [2S-1][D]
 * (16-byte aligned, already resized to 2S-1 rows); S <= 16; D in {64, 80}.  out [nWin][S*S][heads*D] fp16. */
int ea_sam_window_attn_f16(const void* q, const void* k, const void* v, void* out, int nWin, int heads, int S, int D,
                           long long q_sb, long long q_sn, long long k_sb, long long k_sn, long long v_sb,
                           long long v_sn, long long o_sb, long long o_sn, float scale, const void* rel_h,
                           const void* rel_w, void* stream);

/* SAM decomposed rel-pos tables: bias_h[bh][q][kh] = sum_c q[bh,q,c]*Rh[qh - kh + S - 1][c] (same for w),
 * q strided like ea_attention_f16, rel_h/rel_w fp16 [2S-1][D], tokens q = qh*S + qw. */
int ea_relpos_tables_f16(const void* q, int B, int H, int S, int D, long long q_sb, long long q_sn,
                         const void* rel_h, const void* rel_w, float* bias_h, float* bias_w, void* stream);

/* Row softmax on fp32 [rows][cols] -> fp16 (VAE single-head d=512 attention path). */
int ea_softmax_rows_f32_f16(const float* x, void* out, int rows, int cols, float scale, void* stream);

/* One fused elementwise kernel per sampler step (cldm/ddim_hacked.py:187-231):
 *   eps = eps_u + g*(eps_c - eps_u);  [v-pred: eps = sqrt(a_t)*v + sqrt(1-a_t)*x]
 *   x0 = (x - sqrt(1-a_t)*eps)/sqrt(a_t); x_prev = sqrt(a_prev)*x0 + sqrt(1-a_prev-sigma^2)*eps + sigma*noise
 *   optional inpaint blend: x_prev = mask*x_prev + (1-mask)*(sqrt(a_prev)*x_orig + sqrt(1-a_prev)*noise_orig)
 * All tensors fp32 [n]; eps_u may be NULL (no CFG); coef = {a_t, a_prev, sigma, guidance, vpred}. */
int ea_cfg_ddim_step(const float* x, const float* eps_c, const float* eps_u, const float* noise,
                     const float* coef, const float* mask, const float* x_orig, const float* noise_orig,
                     float* x_prev, float* pred_x0, long long n, void* stream);

/* Multistep sampler update / inpaint blend: out = sum_{k<5} coef[k] * s_k (NULL sources skipped); with `mask`:
 * out = mask * (that sum) + (1 - mask) * (coef[5] * alt0 + coef[6] * alt1).  All fp32 [n]; `coef` is a DEVICE buffer of
 * 7 floats (a captured step replays with new coefficients).  Replaces the per-step tensor arithmetic of diffusers'
 * UniPCMultistepScheduler.step (the scheduler sam2image.py:42 / editany_lora.py:384 install; third party) and the
 * re-noise blend of utils/stable_diffusion_controlnet_inpaint.py:1647-1664, 2039-2051. */
int ea_lincomb_f32(const float* s0, const float* s1, const float* s2, const float* s3, const float* s4,
                   const float* coef, const float* mask, const float* alt0, const float* alt1, float* out,
                   long long n, void* stream);

/* The per-step inputs of a captured denoising step, gathered by a DEVICE step index: for s < nseg (<= 8) row `*index` of tables[s]
 * (row_bytes[s] bytes, a multiple of 4; tables and destinations 4-byte aligned) is written dst_rows[s] times, back to back, to
 * dsts[s]; then *index += increment.  The four arrays are HOST arrays read during the call; `index` is a device int64.  One
 * launch for what the reference's sampler loops do on the host per step -- timestep, alpha / sigma coefficients
 * (cldm/ddim_hacked.py:181-197: index = total_steps - i - 1, a_t / a_prev / sigma_t picked per step), and here also every
 * ResBlock's time-embedding row (openaimodel.py:760-762, computed for all steps at once). */
int ea_gather_rows(const void* const* tables, void* const* dsts, const long long* row_bytes, const int* dst_rows, int nseg,
                   long long* index, int increment, void* stream);

/* SAM mask post-processing in one pass (Sam.postprocess_masks + calculate_stability_score + batched_mask_to_box of
 * segment_anything, third party): low_res fp32 [n][lh][lw] logits -> mask uint8 [n][H][W] (logit > threshold at the
 * original resolution, through the img_size^2 intermediate cropped to in_h x in_w, both resizes bilinear
 * align_corners=False) and stats int32 [n][6] = {#(> thr+off), #(> thr-off), xmin, ymin, xmax, ymax}.
 * `stats` must be initialised by the caller to {0, 0, W, H, -1, -1} per mask (accumulated with integer atomics).
 * `mask` may be NULL: statistics only -- the automatic mask generator filters its 3 x 1024 candidates on the statistics
 * and writes masks for the survivors of the NMS alone (a second call), instead of 3072 full-resolution masks. */
int ea_sam_mask_postprocess(const float* low_res, int n_masks, int lh, int lw, int img_size, int in_h, int in_w,
                            int H, int W, float threshold, float offset, unsigned char* mask, int* stats, void* stream);
/* The same over a SELECTION of the masks in `low_res`: slot i (of n_masks) processes low_res[index[i]] and writes
 * mask[i] / stats[i] (index: device int32 [n_masks]; NULL = identity).  lw <= 4096. */
int ea_sam_mask_postprocess_indexed(const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size,
                                    int in_h, int in_w, int H, int W, float threshold, float offset, unsigned char* mask,
                                    int* stats, void* stream);
/* The same with the kernel named (tests / tools: A/B of the two forms, which agree bit for bit): 0 = the library's choice,
 * 1 = the per-pixel kernel (any width), 2 = the tabled kernel (column / row tables of the two resizes in LDS, taps that
 * coincide read once; W <= 2048, else EA_ERR_UNSUPPORTED). */
int ea_sam_mask_postprocess_ex(const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size, int in_h,
                               int in_w, int H, int W, float threshold, float offset, unsigned char* mask, int* stats,
                               int kernel, void* stream);
/* show_anns' id map (sam2image.py:92-115 over SamAutomaticMaskGenerator.generate's list, sam2image.py:117-120) from the
 * records' low-resolution logits, without their full-resolution masks:
 *     idmap[y][x] = max(idmap[y][x], id_base + 1 + max{ s < n_masks : postprocessed mask s covers (y, x) })
 * with mask s = ea_sam_mask_postprocess_indexed's mask of slot s (same arithmetic, same bit per pixel); a pixel no mask covers
 * keeps its value.  idmap: device int32 [H][W], initialised by the caller (zeros for a whole list; id_base = the number of
 * records already painted when a list is processed in pieces).  W <= 2048. */
int ea_sam_id_map(const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size, int in_h, int in_w,
                  int H, int W, float threshold, int id_base, int* idmap, void* stream);

/* SAM mask decoder, image-token side (segment_anything TwoWayAttentionBlock / MaskDecoder, third party; reference call
 * sites sam2image.py:71,118, editany_lora.py:523-543), fused per token (csrc/ea_sam.hip).
 *
 * ea_sam_i2t_f16: one block's image -> token cross attention + residual + LayerNorm:
 *     k_out[b][t] = LN(k[b][t] + softmax_heads(scale * kp[b][t] . g2[b]^T + cbias[b]) . vo[b]^T + bo)
 * kp = keys + positional encoding and k = keys, fp16 [B][T][256] (batch strides kp_sb / k_sb in elements, 0 = one tensor
 * shared by every prompt); g2 fp16 [B][64][256] and cbias fp32 [B][64]: one score column per (head h, token j) at index
 * h * 8 + j (j < 7; column h * 8 + 7 is padding: cbias = -1e30 there); vo fp16 [B][256][64]: vo[b][n][s] multiplies score
 * column ea_sam_vo_perm(s) (the order the MFMA tile pairing leaves the probabilities in).  kp may be NULL: the operand is
 * then fp16(k + pe), formed in the kernel (pe fp16 [T][256]; no keys + pe tensor ever stored).  kp_out (optional, needs pe)
 * = k_out + pe.  C must be 256. */
int ea_sam_vo_perm(int s);
int ea_sam_i2t_f16(const void* kp, long long kp_sb, const void* k, long long k_sb, const void* pe, const void* g2,
                   const float* cbias, const void* vo, const float* bo, const float* ln_g, const float* ln_b, float eps,
                   float scale, void* k_out, void* kp_out, int B, int T, int C, void* stream);
/* ea_sam_t2i_f16: the token -> image cross attention of a block with its key / value projections folded into the token
 * side: ctx[b] = softmax_rows(scale * g[b] (k[b] + pe)^T) k[b], g fp16 [B][64][256] (one row per (head, token), padding rows
 * arbitrary), k fp16 [B][T][256] (k_sb = 0: shared), pe fp16 [T][256], ctx fp32 [B][64][256].  T % 64 == 0, C = 256. */
int ea_sam_t2i_f16(const void* k, long long k_sb, const void* pe, const void* g, float scale, float* ctx, int B, int T, int C,
                   void* stream);
/* ea_sam_upscale_f16 (round 6): MaskDecoder.output_upscaling WHOLE -- ConvTranspose2d(256 -> 64, k 2, s 2) as a per-token
 * product with w0 fp16 [256][256] (row (dy * 2 + dx) * 64 + c) + b0 fp32 [256], then exactly ea_sam_upscale_tail_f16 -- from the
 * image tokens k fp16 [B*h*w][256] to the mask logits, the intermediate [B*h*w*4][64] tensor never stored.  h * w % 16 == 0. */
int ea_sam_upscale_f16(const void* k, const void* w0, const float* b0, const float* ln_g, const float* ln_b, float eps,
                       const void* w1, const float* b1, const float* hyper, float* masks, int B, int h, int w, int m0, int nm,
                       void* stream);
/* The 7-token side of those two attentions (round 6; TwoWayAttentionBlock's q / k / v / out projections of the token
 * side, segment_anything modeling/transformer.py, third party), one launch per operand instead of einsum + cast + pad:
 * ea_sam_fold_heads_f16: out[b][s][c] = sum_e x[b][j][h * d_head + e] * w[h][e][c] with h * 8 + j = perm ? perm[s] : s, zero
 *   rows for the absent tokens j >= n.  x fp32 [B][n][heads * d_head], w fp32 [heads][d_head][C], perm int32 [64] on the
 *   device or NULL; out fp16 [B][64][C] (c_major = 0: ea_sam_t2i_f16's g, ea_sam_i2t_f16's g2) or [B][C][64] (c_major = 1:
 *   ea_sam_i2t_f16's vo with perm = ea_sam_vo_perm and w[h][e][c] = Wo[c][h * d_head + e]).  heads = 8, C = 256, d_head 16 | 32.
 * ea_sam_unfold_heads_f32: out[b][j][h * d_head + e] = bias[h * d_head + e] + sum_c ctx[b][h * 8 + j][c] * wt[c][h * d_head + e]:
 *   ea_sam_t2i_f16's ctx fp32 [B][64][C] through the value projection (wt = its weight transposed, fp32 [C][heads * d_head];
 *   bias fp32 or NULL) -> out fp32 [B][n][heads * d_head]. */
/* ea_sam_token_self_attn_f16: TwoWayAttentionBlock.self_attn's core on the prompt tokens: out[b] = per head
 * softmax(scale * q_h k_h^T) v_h over the n <= 8 tokens of prompt b, fp32 arithmetic on the fp16 projections q / k / v
 * [B][n][256] (dense) -> out fp16 [B][n][256].  heads = 8, C = 256 (head width 32). */
int ea_sam_token_self_attn_f16(const void* q, const void* k, const void* v, void* out, int B, int n, int heads, int C,
                               float scale, void* stream);
int ea_sam_fold_heads_f16(const float* x, const float* w, const int* perm, void* out, int B, int n, int heads, int d_head, int C,
                          int c_major, void* stream);
int ea_sam_unfold_heads_f32(const float* ctx, const float* wt, const float* bias, float* out, int B, int n, int heads, int d_head,
                            int C, void* stream);
/* ea_sam_upscale_tail_f16: MaskDecoder.output_upscaling from the first transposed conv's output on (u0 fp16 [B*h*w*4][64],
 * rows ordered (b, y, x, dy, dx)): LayerNorm2d(64, eps) + GELU, ConvTranspose2d(64 -> 32, k 2, s 2) as a per-row product
 * with w1 fp16 [128][64] (row (ddy * 2 + ddx) * 32 + c) + b1, GELU, and the product with the hypernetwork outputs
 * hyper fp32 [B][4][32] -> masks fp32 [B][nm][4h][4w], the outputs of hypernetworks m0 .. m0 + nm - 1 (MaskDecoder.forward
 * keeps 1..3 for multimask output, 0 otherwise). */
int ea_sam_upscale_tail_f16(const void* u0, const float* ln_g, const float* ln_b, float eps, const void* w1, const float* b1,
                            const float* hyper, float* masks, int B, int h, int w, int m0, int nm, void* stream);

/* Layout plumbing on device: NCHW fp32 -> NHWC fp16 with channel padding, and back. */
int ea_nchw_f32_to_nhwc_f16(const float* x, void* out, int B, int C, int H, int W, int Cpad, float mul,
                            float add, void* stream);
int ea_nhwc_f16_to_nchw_f32(const void* x, float* out, int B, int C, int H, int W, int Cstride,
                            float mul, float add, void* stream);

/* y = silu(x) on fp32 -> fp16 / fp32 small vectors (time-embedding MLP). */
int ea_silu_f32(const float* x, float* out, long long n, void* stream);
/* out = a + b (fp16), n % 8 == 0. */
int ea_add_f16(const void* a, const void* b, void* out, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
