// ea_sam.hip -- the image-token side of SAM's mask decoder as two fused gfx950 kernels (round 3).
//
// Reference call sites: `SamAutomaticMaskGenerator(sam).generate(image)` (sam2image.py:71,118; editany_lora.py:523) and
// `SamPredictor.predict` (editany_lora.py:527-543); the decoder itself is segment_anything's MaskDecoder / TwoWayTransformer
// (third party, un-vendored: SURVEY.md appendix C).  One image = 1024 point prompts x 4096 image tokens x 256 channels: the
// image side of the decoder is a stream of [B, 4096, 256] fp16 tensors (2 MB per prompt), and what it costs is passes
// over them.  Restated so that each pass does all the arithmetic that is local to a token:
//
//  * ea_sam_i2t_f16 -- one TwoWayAttentionBlock's "image -> token" cross attention + residual + LayerNorm (norm4):
//        keys <- LN(keys + softmax_j((keys + pe) Wq k_j) v_j Wo + bo)           (7 tokens j, 8 heads)
//    The 7-token side is folded into two small per-prompt matrices on the host side of the ABI: scores = (keys + pe) G2^T
//    + c with G2[h*8 + j] = Wq_h^T k_hj (a row of 256), and the attention output = P VO with VO[h*8 + j] = Wo_h v_hj.
//    Per token: a [256] x [256 x 64] product, eight 7-way softmaxes, a [64] x [64 x 256] product, residual, LayerNorm --
//    all in registers; the tensor is read once (twice: keys + pe as the operand, keys as the residual) and written once
//    (twice when the next block wants keys + pe too).  The unfused form runs a [B*4096 x 256 x 128] projection, an
//    attention call, a [B*4096 x 128 x 256] projection and a LayerNorm pass over the same tensor.
//  * ea_sam_upscale_tail_f16 -- output_upscaling's LayerNorm2d + GELU + second transposed conv + GELU + the product with
//    the hypernetwork outputs: from the first transposed conv's [B*4096*4, 64] rows straight to the [B, 4, 256, 256] mask
//    logits (three full passes and a batched product less).
//
// MFMA layout (ea_prims.h): operands swapped (D^T = W A^T), so a lane holds 4 consecutive output columns of one row; two
// 16-column tiles pair up through v_permlane16_swap into 8 consecutive columns per lane.  In ea_sam_i2t the score columns
// are laid out [head][8] (7 tokens + one pad column): after the pairing a lane holds exactly ONE head's scores, the softmax
// is lane-local, and the packed probabilities ARE the A fragment of the second product (whose weight rows the host
// stores in the matching order -- `ea_sam_vo_perm`).
#include "ea_prims.h"
#include "../../include/editanything_hip.h"

namespace {

struct SamI2tParams {
  const f16* kp; long long kp_sb;     // [B][T][256] keys + pe (stride 0: shared by all prompts)
  const f16* k; long long k_sb;       // [B][T][256] keys (residual)
  const f16* pe;                      // [T][256] or null
  const f16* g2;                      // [B][64][256]
  const float* cbias;                 // [B][64]
  const f16* vo;                      // [B][256][64] (k order: ea_sam_vo_perm)
  const float* bo; const float* ln_g; const float* ln_b;   // [256]
  float eps, scale;
  f16* k_out; f16* kp_out;            // [B][T][256]
  int B, T;
};

constexpr int SAM_C = 256;
constexpr int I2T_TOK_PER_WG = 512;   // 4 waves x 16 tokens x 8 rounds

// storage position s (0..31) of a 32-wide K step -> logical column inside the step, as the tile pairing leaves it in
// the lanes: lane group q4 = s / 8 holds columns (q4 & 1) * 16 + (q4 >> 1) * 8 + 0..7
__host__ __device__ inline int sam_perm32(int s) { return ((s >> 3) & 1) * 16 + (s >> 4) * 8 + (s & 7); }

__global__ __launch_bounds__(256, 2) void ea_sam_i2t_kernel(SamI2tParams p) {
  EA_SMEM(smem);
  // LDS: G2 [64][512 B] (16-B chunks XOR-swizzled with row & 15), VO [256][128 B] (chunks XOR (row >> 1) & 7), bo / gamma / beta
  char* s_g2 = smem;
  char* s_vo = smem + 64 * 512;
  float* s_vec = reinterpret_cast<float*>(smem + 64 * 512 + 256 * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int chunks_per_b = (p.T + I2T_TOK_PER_WG - 1) / I2T_TOK_PER_WG;
  const int b = blockIdx.x / chunks_per_b, chunk = blockIdx.x - b * chunks_per_b;
  {
    const f16* g2 = p.g2 + (long long)b * 64 * SAM_C;
    for (int i = tid; i < 64 * 32; i += 256) {       // 16-byte pieces
      const int row = i >> 5, ch = i & 31;
      *reinterpret_cast<f16x8*>(s_g2 + row * 512 + ((ch ^ (row & 15)) << 4)) = ea_ld8(g2 + row * SAM_C + ch * 8);
    }
    const f16* vo = p.vo + (long long)b * SAM_C * 64;
    for (int i = tid; i < 256 * 8; i += 256) {
      const int row = i >> 3, ch = i & 7;
      *reinterpret_cast<f16x8*>(s_vo + row * 128 + ((ch ^ ea_swz(row)) << 4)) = ea_ld8(vo + row * 64 + ch * 8);
    }
    for (int i = tid; i < SAM_C; i += 256) { s_vec[i] = p.bo[i]; s_vec[SAM_C + i] = p.ln_g[i]; s_vec[2 * SAM_C + i] = p.ln_b[i]; }
  }
  __syncthreads();
  // this lane's score-column bias: after the pairing it holds head (2*pr + (q4 & 1) * 2 + (q4 >> 1)) of pair pr -> columns
  // base .. base + 7 with base = 32 * pr + sam_perm32(8 * q4)
  float cb[2][8];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int e = 0; e < 8; ++e) cb[pr][e] = p.cbias[b * 64 + 32 * pr + sam_perm32(8 * q4) + e];

  const f16* kp_b = p.kp ? p.kp + (long long)b * p.kp_sb : nullptr;
  const f16* k_b = p.k + (long long)b * p.k_sb;
  f16* ko_b = p.k_out + (long long)b * p.T * SAM_C;
  f16* kpo_b = p.kp_out ? p.kp_out + (long long)b * p.T * SAM_C : nullptr;

  for (int round = 0; round < I2T_TOK_PER_WG / 64; ++round) {
    const int t0 = chunk * I2T_TOK_PER_WG + round * 64 + wave * 16;
    if (t0 >= p.T) break;                                   // (wave-uniform; no barrier below)
    const int t = (t0 + c16 < p.T) ? t0 + c16 : p.T - 1;    // ragged tail: clamp the row, mask the stores
    const bool row_ok = t0 + c16 < p.T;
    // ---- product 1: scores[16 x 64] = (keys + pe)[16 x 256] G2^T
    f16x8 fa[8];
    if (p.kp) {
#pragma unroll
      for (int s = 0; s < 8; ++s) fa[s] = ea_ld8(kp_b + (long long)t * SAM_C + 32 * s + 8 * q4);
    } else {
      // no stored keys + pe tensor: the operand is formed here, rounded to fp16 exactly like a stored sum would be
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f16x8 kv = ea_ld8(k_b + (long long)t * SAM_C + 32 * s + 8 * q4), pv = ea_ld8(p.pe + (long long)t * SAM_C + 32 * s + 8 * q4);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[s][e] = (f16)((float)kv[e] + (float)pv[e]);
      }
    }
    f32x4 sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 16 * j + c16, ch = 4 * s + q4;
        const f16x8 fb = *reinterpret_cast<const f16x8*>(s_g2 + row * 512 + ((ch ^ (row & 15)) << 4));
        sc[j] = ea_mfma_16x16x32(fb, fa[s], sc[j]);
      }
    // ---- pair the tiles: 8 consecutive columns = one head per lane; lane-local 7-way softmax (the pad column carries -1e30)
    f16x8 prob[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f32x4 a = sc[2 * pr], bq = sc[2 * pr + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { float x = a[r], y = bq[r]; ea_swap16(x, y); a[r] = x; bq[r] = y; }
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = a[r] * p.scale + cb[pr][r]; v[4 + r] = bq[r] * p.scale + cb[pr][4 + r]; }
      float mx = v[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) mx = fmaxf(mx, v[e]);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = ea_expf(v[e] - mx); sum += v[e]; }
      const float inv = 1.0f / sum;
#pragma unroll
      for (int e = 0; e < 8; ++e) prob[pr][e] = (f16)(v[e] * inv);
    }
    // ---- product 2: out[16 x 256] = P[16 x 64] VO^T (the lanes' probability vectors are the A fragments of the two K steps)
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = 16 * j + c16, ch = 4 * ks + q4;
        const f16x8 fb = *reinterpret_cast<const f16x8*>(s_vo + row * 128 + ((ch ^ ea_swz(row)) << 4));
        acc[j] = ea_mfma_16x16x32(fb, prob[ks], acc[j]);
      }
    // ---- + bo + residual, LayerNorm over the 256 columns of the row, outputs.  Paired layout: vector pr2 of lane group q4
    // covers columns 32 * pr2 + sam_perm32(8 * q4) .. + 7 of row c16.
    float x[8][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pr2 = 0; pr2 < 8; ++pr2) {
      f32x4 a = acc[2 * pr2], bq = acc[2 * pr2 + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { float u = a[r], w = bq[r]; ea_swap16(u, w); a[r] = u; bq[r] = w; }
      const int col = 32 * pr2 + sam_perm32(8 * q4);
      const f16x8 res = ea_ld8(k_b + (long long)t * SAM_C + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[pr2][r] = a[r] + s_vec[col + r] + (float)res[r];
        x[pr2][4 + r] = bq[r] + s_vec[col + 4 + r] + (float)res[4 + r];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1 += x[pr2][e]; s2 += x[pr2][e] * x[pr2][e]; }
    }
    s1 += ea_shfl_xor(s1, 16); s2 += ea_shfl_xor(s2, 16);
    s1 += ea_shfl_xor(s1, 32); s2 += ea_shfl_xor(s2, 32);
    const float mean = s1 * (1.0f / SAM_C);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / SAM_C) - mean * mean, 0.f) + p.eps);
#pragma unroll
    for (int pr2 = 0; pr2 < 8; ++pr2) {
      const int col = 32 * pr2 + sam_perm32(8 * q4);
      f16x8 h;
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { y[e] = (x[pr2][e] - mean) * rstd * s_vec[SAM_C + col + e] + s_vec[2 * SAM_C + col + e]; h[e] = (f16)y[e]; }
      if (row_ok) ea_st8(ko_b + (long long)t * SAM_C + col, h);
      if (kpo_b) {
        // keys + pe of the NEXT block, from the rounded keys (what a separate add over the stored fp16 tensor would give)
        const f16x8 pv = ea_ld8(p.pe + (long long)t * SAM_C + col);
        f16x8 h2;
#pragma unroll
        for (int e = 0; e < 8; ++e) h2[e] = (f16)((float)h[e] + (float)pv[e]);
        if (row_ok) ea_st8(kpo_b + (long long)t * SAM_C + col, h2);
      }
    }
  }
}

struct SamTailParams {
  const f16* u0;                      // [B*T*4][64]
  const float* ln_g; const float* ln_b; float eps;   // [64]
  const f16* w1; const float* b1;     // [128][64], [128]: N = (ddy * 2 + ddx) * 32 + c
  const float* hyper;                 // [B][4][32]
  float* masks;                       // [B][nm][4h][4w]: hypernetwork outputs m0 .. m0 + nm - 1
  int B, h, w, m0, nm;
};

__global__ __launch_bounds__(256, 2) void ea_sam_upscale_tail_kernel(SamTailParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int T = p.h * p.w;
  const long long rows_per_b = (long long)T * 4;
  // weight fragments and the per-K-position LayerNorm affine terms: loop invariant, in registers
  f16x8 fw[8][2];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fw[j][ks] = ea_ld8(p.w1 + (16 * j + c16) * 64 + 32 * ks + 8 * q4);
  float lg[2][8], lb[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) { lg[ks][e] = p.ln_g[32 * ks + 8 * q4 + e]; lb[ks][e] = p.ln_b[32 * ks + 8 * q4 + e]; }
  float bias[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[j][r] = p.b1[16 * j + 4 * q4 + r];
  const long long ntiles = (long long)p.B * rows_per_b / 16;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
    const long long row0 = tile * 16;
    const int b = (int)(row0 / rows_per_b);
    const long long row = row0 + c16;
    // ---- LayerNorm2d over the 64 channels of the row (fp32, eps 1e-6) + exact GELU -> A fragments
    f16x8 in0 = ea_ld8(p.u0 + row * 64 + 8 * q4), in1 = ea_ld8(p.u0 + row * 64 + 32 + 8 * q4);
    float v[2][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[0][e] = (float)in0[e]; v[1][e] = (float)in1[e]; s1 += v[0][e] + v[1][e]; s2 += v[0][e] * v[0][e] + v[1][e] * v[1][e]; }
    s1 += ea_shfl_xor(s1, 16); s2 += ea_shfl_xor(s2, 16);
    s1 += ea_shfl_xor(s1, 32); s2 += ea_shfl_xor(s2, 32);
    const float mean = s1 * (1.0f / 64.0f);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / 64.0f) - mean * mean, 0.f) + p.eps);
    f16x8 fa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[ks][e] = (f16)ea_gelu_erf((v[ks][e] - mean) * rstd * lg[ks][e] + lb[ks][e]);
    // ---- second transposed conv as a [16 x 64] x [64 x 128] product, + bias, GELU
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc[j] = ea_mfma_16x16x32(fw[j][ks], fa[ks], acc[j]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][r] = ea_gelu_erf(acc[j][r] + bias[j][r]);
    }
    // ---- product with the hypernetwork outputs: logits[mask m][sub-pixel s] = sum_c u1[s * 32 + c] * hyper[b][m][c];
    // this lane holds c in {4 q4 + r, 16 + 4 q4 + r}: partial sums, then a butterfly over the four lane groups
    const float* hy = p.hyper + (long long)b * 4 * 32;
    float out[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float h8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { h8[r] = hy[m * 32 + 4 * q4 + r]; h8[4 + r] = hy[m * 32 + 16 + 4 * q4 + r]; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) a += acc[2 * s][r] * h8[r] + acc[2 * s + 1][r] * h8[4 + r];
        a += ea_shfl_xor(a, 16);
        a += ea_shfl_xor(a, 32);
        out[m][s] = a;
      }
    }
    // lane group q4 writes mask q4: row -> (token (y, x), first up-sampling sub-pixel (dy, dx)); s = (ddy, ddx)
    const long long rb = row - (long long)b * rows_per_b;
    const int tok = (int)(rb >> 2), sub = (int)(rb & 3);
    const int y = tok / p.w, xx = tok - y * p.w;
    const int Y = 4 * y + 2 * (sub >> 1), X = 4 * xx + 2 * (sub & 1);
    const int W4 = 4 * p.w;
    float o[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = q4 == 0 ? out[0][s] : q4 == 1 ? out[1][s] : q4 == 2 ? out[2][s] : out[3][s];
    const int mo = q4 - p.m0;           // multimask output keeps hypernetworks 1..3, single-mask output 0 (MaskDecoder.forward)
    if (mo >= 0 && mo < p.nm) {
      float* dst = p.masks + (((long long)b * p.nm + mo) * (4 * p.h) + Y) * W4 + X;
      *reinterpret_cast<f32x2*>(dst) = f32x2{o[0], o[1]};
      *reinterpret_cast<f32x2*>(dst + W4) = f32x2{o[2], o[3]};
    }
  }
}

// ---- output_upscaling in ONE pass (round 6): first transposed conv + everything ea_sam_upscale_tail does.  Rounds 3-5 ran the
// first ConvTranspose2d(256 -> 64, 2, 2) as a plain contraction that wrote u0 = [B*T*4][64] fp16 (2.1 GB for 1024 prompts) for
// the tail kernel to read back: 1.4 + 2.5 ms per image, the largest item of the mask decoder.  Here its [256 x 256] weight
// lives in LDS as ready-made MFMA fragments (128 KB, loaded once per persistent workgroup), a wave (eight per workgroup, two per SIMD: one's matrix work under the other's GELUs) takes 16 image tokens,
// multiplies them through (128 MFMAs), and runs the tail on the accumulators -- four sub-pixels x (LayerNorm2d + GELU, second
// transposed conv, GELU, hypernetwork product) -- without u0 ever existing.  The accumulator layout (lane (c16, q4) holds
// channels 16 j + 4 q4 + r of token c16) is not the K order of the second product's fragments; the contraction does not care
// about K order, so the second weight's fragments (and the LayerNorm affine terms) are gathered in the order the lanes hold.
struct SamUpParams {
  const f16* k;                       // [B*T][256] image tokens after the transformer
  const f16* w0; const float* b0;     // [256][256] (row (dy * 2 + dx) * 64 + c), [256]
  const float* ln_g; const float* ln_b; float eps;   // [64]
  const f16* w1; const float* b1;     // [128][64], [128]
  const float* hyper;                 // [B][4][32]
  float* masks;                       // [B][nm][4h][4w]
  int B, h, w, m0, nm;
};

constexpr int UP_W0_LDS = 16 * 8 * 64 * 16;     // 128 KB: fragment (j, ks) of lane l at ((j * 8 + ks) * 64 + l) * 16

__global__ __launch_bounds__(512, 1) void ea_sam_upscale_fused_kernel(SamUpParams p) {
  EA_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int T = p.h * p.w;
  for (int f = tid; f < 16 * 8 * 64; f += 512) {
    const int l = f & 63, ks = (f >> 6) & 7, j = f >> 9;
    *reinterpret_cast<f16x8*>(smem + f * 16) = ea_ld8(p.w0 + (16 * j + (l & 15)) * SAM_C + 32 * ks + 8 * (l >> 4));
  }
  // second product's weight fragments in the accumulators' channel order: K position (ks, q4, e) = channel 16 (2 ks + (e >> 2)) + 4 q4 + (e & 3)
  f16x8 fw[8][2];
  float lg[2][8], lb[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int ch = 16 * (2 * ks + hf) + 4 * q4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f16x4 w4 = *reinterpret_cast<const f16x4*>(p.w1 + (16 * j + c16) * 64 + ch);
#pragma unroll
        for (int r = 0; r < 4; ++r) fw[j][ks][4 * hf + r] = w4[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { lg[ks][4 * hf + r] = p.ln_g[ch + r]; lb[ks][4 * hf + r] = p.ln_b[ch + r]; }
    }
  }
  __syncthreads();
  const long long ntiles = (long long)p.B * T / 16;
  const int W4 = 4 * p.w;
  for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long long)gridDim.x * 8) {
    const long long t0 = tile * 16;
    const int b = (int)(t0 / T);
    f16x8 fk[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) fk[ks] = ea_ld8(p.k + (t0 + c16) * SAM_C + 32 * ks + 8 * q4);
    const int tok = (int)(t0 - (long long)b * T) + c16;
    const int y = tok / p.w, xx = tok - y * p.w;
    const float* hy = p.hyper + (long long)b * 4 * 32;
#pragma unroll 1
    for (int sub = 0; sub < 4; ++sub) {
      // ---- first transposed conv, the 64 channels of sub-pixel `sub`: u[jj][r] = channel 16 jj + 4 q4 + r of token c16
      f32x4 u[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * sub + jj;
        u[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          u[jj] = ea_mfma_16x16x32(*reinterpret_cast<const f16x8*>(smem + ((j * 8 + ks) * 64 + lane) * 16), fk[ks], u[jj]);
#pragma unroll
        for (int r = 0; r < 4; ++r) u[jj][r] = (float)(f16)(u[jj][r] + p.b0[16 * j + 4 * q4 + r]);   // the unfused form stores u0 in fp16
      }
      // ---- LayerNorm2d over the 64 channels (fp32, eps 1e-6) + exact GELU -> fragments of the second product
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1 += u[jj][r]; s2 += u[jj][r] * u[jj][r]; }
      s1 += ea_shfl_xor(s1, 16); s2 += ea_shfl_xor(s2, 16);
      s1 += ea_shfl_xor(s1, 32); s2 += ea_shfl_xor(s2, 32);
      const float mean = s1 * (1.0f / 64.0f);
      const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / 64.0f) - mean * mean, 0.f) + p.eps);
      f16x8 fa[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[ks][e] = (f16)ea_gelu_erf((u[2 * ks + (e >> 2)][e & 3] - mean) * rstd * lg[ks][e] + lb[ks][e]);
      // ---- second transposed conv as a [16 x 64] x [64 x 128] product, + bias, GELU
      f32x4 acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc[j] = ea_mfma_16x16x32(fw[j][ks], fa[ks], acc[j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = ea_gelu_erf(acc[j][r] + p.b1[16 * j + 4 * q4 + r]);
      }
      // ---- product with the hypernetwork outputs (as in ea_sam_upscale_tail_kernel)
      float out[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float h8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { h8[r] = hy[m * 32 + 4 * q4 + r]; h8[4 + r] = hy[m * 32 + 16 + 4 * q4 + r]; }
#pragma unroll
        for (int s2i = 0; s2i < 4; ++s2i) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) a += acc[2 * s2i][r] * h8[r] + acc[2 * s2i + 1][r] * h8[4 + r];
          a += ea_shfl_xor(a, 16);
          a += ea_shfl_xor(a, 32);
          out[m][s2i] = a;
        }
      }
      const int Y = 4 * y + 2 * (sub >> 1), X = 4 * xx + 2 * (sub & 1);
      float o[4];
#pragma unroll
      for (int s2i = 0; s2i < 4; ++s2i) o[s2i] = q4 == 0 ? out[0][s2i] : q4 == 1 ? out[1][s2i] : q4 == 2 ? out[2][s2i] : out[3][s2i];
      const int mo = q4 - p.m0;
      if (mo >= 0 && mo < p.nm) {
        float* dst = p.masks + (((long long)b * p.nm + mo) * (4 * p.h) + Y) * W4 + X;
        *reinterpret_cast<f32x2*>(dst) = f32x2{o[0], o[1]};
        *reinterpret_cast<f32x2*>(dst + W4) = f32x2{o[2], o[3]};
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ token -> image
//  ctx[b] = softmax_rows(scale * g[b] (k[b] + pe)^T) k[b]          g [64 x 256] (one row per (head, token)), k [T x 256]
// The token -> image cross attention of a TwoWayAttentionBlock with the key / value projections folded into the 7-token
// side (amg.py `_t2i_folded`): a 64-query attention over T keys with head dimension 256 whose keys are (k + pe) and whose
// values are k itself.  One workgroup per prompt walks the image tokens 64 at a time: the tile is loaded into registers one
// tile ahead (k and pe, 16-byte loads), stored to LDS twice -- k + pe row-major for the score product, k row-major for the
// value product, which reads it TRANSPOSED with ds_read_b64_tr_b16 -- flash-style online softmax in registers (each of the 4
// waves owns 16 of the 64 score rows), probabilities paired into A fragments exactly like ea_sam_i2t.  Replaces, per
// block: a [B*64 x T x 256] score contraction with a 1-GB fp32 output, a row-softmax pass over it, and a batched
// [64 x T] x [T x 256] product (rocBLAS) -- three passes over 1-2 GB each -- with ONE pass over the 2 MB of k per prompt.
struct SamT2iParams {
  const f16* k; long long k_sb;       // [B][T][256] (stride 0: shared)
  const f16* pe;                      // [T][256]
  const f16* g;                       // [B][64][256]
  float* ctx;                         // [B][64][256]
  float scale;
  int B, T;
};

constexpr int T2I_TILE = 64;
constexpr int T2I_PITCH = 512 + 32;   // bytes per LDS row: 8 consecutive rows land on 8 different 32-byte bank groups (tr reads)

__global__ __launch_bounds__(256, 2) void ea_sam_t2i_kernel(SamT2iParams p) {
  EA_SMEM(smem);
  char* s_kp = smem;                                   // [64][PITCH] k + pe
  char* s_k = smem + T2I_TILE * T2I_PITCH;             // [64][PITCH] k
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.x;
  const f16* kb = p.k + (long long)b * p.k_sb;
  // this wave's 16 score rows of g as A fragments (loop invariant)
  f16x8 ga[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) ga[s] = ea_ld8(p.g + ((long long)b * 64 + wave * 16 + c16) * SAM_C + 32 * s + 8 * q4);
  // staging: thread t moves 16-byte pieces t, t + 256, ... of the [64 x 256] tile (32 pieces per row)
  f16x8 rk[8], rp[8];
  auto load_tile = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 5, ch = c & 31;
      rk[i] = ea_ld8(kb + (long long)(t0 + row) * SAM_C + ch * 8);
      rp[i] = ea_ld8(p.pe + (long long)(t0 + row) * SAM_C + ch * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 5, ch = c & 31;
      f16x8 kp;
#pragma unroll
      for (int e = 0; e < 8; ++e) kp[e] = (f16)((float)rk[i][e] + (float)rp[i][e]);
      *reinterpret_cast<f16x8*>(s_k + row * T2I_PITCH + ch * 16) = rk[i];
      *reinterpret_cast<f16x8*>(s_kp + row * T2I_PITCH + ch * 16) = kp;
    }
  };
  f32x4 acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  // per-lane part of the V^T fragment address (ea_prims.h ea_lds_read_tr16): 16-lane group q4 supplies the 8 tokens
  // sam_perm32(8 * q4) .. + 7 of a 32-token K step (the order the paired probabilities are in), lane j of the group the
  // token row j >> 2 (+ 4 for the second read) and channels 4 * (j & 3) .. + 3 of the 16-channel tile
  const int vt_off = (sam_perm32(8 * q4) + (c16 >> 2)) * T2I_PITCH + 8 * (c16 & 3);
  const int ntile = p.T / T2I_TILE;
  load_tile(0);
  for (int tl = 0; tl < ntile; ++tl) {
    __syncthreads();                       // the previous tile's readers are done
    store_tile();
    __syncthreads();
    if (tl + 1 < ntile) load_tile((tl + 1) * T2I_TILE);
    // ---- scores of this wave's 16 rows against the 64 tokens of the tile
    f32x4 sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x8 fb = *reinterpret_cast<const f16x8*>(s_kp + (16 * j + c16) * T2I_PITCH + (4 * s + q4) * 16);
        sc[j] = ea_mfma_16x16x32(fb, ga[s], sc[j]);
      }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc[j][r] *= p.scale; mx = fmaxf(mx, sc[j][r]); }
    mx = fmaxf(mx, ea_shfl_xor(mx, 16));
    mx = fmaxf(mx, ea_shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = ea_expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] *= alpha;
    // ---- probabilities, paired into the A fragments of the two 32-token K steps
    f16x8 prob[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f32x4 a = sc[2 * pr], bq = sc[2 * pr + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = ea_expf(a[r] - m_new); bq[r] = ea_expf(bq[r] - m_new);
        l_run += a[r] + bq[r];
        float x = a[r], y = bq[r];
        ea_swap16(x, y);
        prob[pr][r] = (f16)x; prob[pr][4 + r] = (f16)y;
      }
    }
    // ---- ctx += P k: the value operand is the k tile read transposed
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* vbase = s_k + 32 * ks * T2I_PITCH + vt_off;
      ea_static_for<16>([&](auto jn_) {
        constexpr int jn = decltype(jn_)::value;
        const f16x4 lo = ea_lds_read_tr16<32 * jn>(vbase);
        const f16x4 hi = ea_lds_read_tr16<32 * jn + 4 * T2I_PITCH>(vbase);
        ea_lds_tr_wait();
        f16x8 fb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { fb[e] = lo[e]; fb[4 + e] = hi[e]; }
        acc[jn] = ea_mfma_16x16x32(fb, prob[ks], acc[jn]);
      });
    }
  }
  // ---- normalise and write: lane (c16, q4) holds row wave * 16 + c16, columns 16 j + 4 q4 .. + 3
  l_run += ea_shfl_xor(l_run, 16);
  l_run += ea_shfl_xor(l_run, 32);
  const float inv = 1.0f / l_run;
  float* dst = p.ctx + ((long long)b * 64 + wave * 16 + c16) * SAM_C + 4 * q4;
#pragma unroll
  for (int j = 0; j < 16; ++j) *reinterpret_cast<f32x4*>(dst + 16 * j) = acc[j] * inv;
}


// ---- the 7-token side of the two cross attentions, folded per head (round 6).  ea_sam_t2i / ea_sam_i2t take the token side
// as per-prompt matrices with one row per (head, token): G[h*8 + j] = Wk_h^T q_hj, G2[h*8 + j] = Wq_h^T k_hj, VO[.][s] = Wo_h v_hj.
// Rounds 3-5 formed them with torch: a batched fp32 einsum (58 MB written for 1024 prompts), a cast, a zero fill and a strided
// copy into the padded fp16 operand -- four launches and ~150 us per operand, seven operands per decoder pass.  Here: one
// launch per operand, one workgroup per prompt, thread c owns output column c (its 16 weights per head in registers), the
// [64][256] fp16 result goes through LDS so that either layout leaves in 16-byte pieces.
struct SamFoldParams {
  const float* x;      // [B][n][heads * d]
  const float* w;      // [heads][d][256]
  const int* perm;     // storage position s -> h * 8 + j, or null (identity)
  f16* out;            // (b, s, c) at b * 64 * 256 + (cs ? c * 64 + s : s * 256 + c)
  int B, n, d, cs;
};

constexpr int FOLD_PITCH = SAM_C + 8;   // fp16 elements per staged row

template <int D>
__global__ __launch_bounds__(256) void ea_sam_fold_kernel(SamFoldParams p) {
  EA_SMEM(smem);
  float* xs = reinterpret_cast<float*>(smem);                          // [8][8 * D]
  int* inv = reinterpret_cast<int*>(smem + 8 * 8 * D * 4);             // [64]: h * 8 + j -> s
  f16* tile = reinterpret_cast<f16*>(smem + 8 * 8 * D * 4 + 256);      // [64][FOLD_PITCH]
  const int c = threadIdx.x, b = blockIdx.x;
  const int row = 8 * D;
  for (int i = c; i < 8 * row; i += 256) xs[i] = i < p.n * row ? p.x[(long long)b * p.n * row + i] : 0.0f;
  if (c < 64) inv[p.perm ? p.perm[c] : c] = c;
  __syncthreads();
#pragma unroll 1
  for (int h = 0; h < 8; ++h) {
    float w[D];
#pragma unroll
    for (int e = 0; e < D; ++e) w[e] = p.w[(h * D + e) * SAM_C + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float acc = 0.0f;
      const float* xr = xs + j * row + h * D;
#pragma unroll
      for (int e = 0; e < D; ++e) acc += xr[e] * w[e];
      tile[inv[h * 8 + j] * FOLD_PITCH + c] = (f16)acc;      // rows of the absent tokens (j >= n) are zero: xs is
    }
  }
  __syncthreads();
  f16* out = p.out + (long long)b * 64 * SAM_C;
  if (!p.cs) {
    for (int i = c; i < 64 * (SAM_C / 8); i += 256) {
      const int s = i / (SAM_C / 8), c8 = (i - s * (SAM_C / 8)) * 8;
      ea_st8(out + s * SAM_C + c8, *reinterpret_cast<const f16x8*>(tile + s * FOLD_PITCH + c8));
    }
  } else {
    for (int i = c; i < SAM_C * 8; i += 256) {
      const int cc = i >> 3, s8 = (i & 7) * 8;
      f16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[(s8 + e) * FOLD_PITCH + cc];
      ea_st8(out + cc * 64 + s8, v);
    }
  }
}

// ---- and back: the token -> image attention's context rows through the value projection,
//   out[b][j][h * d + e] = bias[h * d + e] + sum_c ctx[b][h * 8 + j][c] * w[c][h * d + e]
// (ctx = ea_sam_t2i_f16's fp32 [B][64][256]).  One workgroup per prompt; the 64 context rows are staged in LDS, thread
// (h, e) walks its weight row once for all of the head's tokens.
struct SamUnfoldParams {
  const float* ctx;    // [B][64][256]
  const float* w;      // [256][heads * d]: the projection's weight transposed
  const float* bias;   // [heads * d] or null
  float* out;          // [B][n][heads * d]
  int B, n, d;
};

template <int D>
__global__ __launch_bounds__(256) void ea_sam_unfold_kernel(SamUnfoldParams p) {
  EA_SMEM(smem);
  float* cs = reinterpret_cast<float*>(smem);                          // [64][256 + 4]
  constexpr int PITCH = SAM_C + 4;
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* ctx = p.ctx + (long long)b * 64 * SAM_C;
  for (int i = tid; i < 64 * (SAM_C / 4); i += 256) {
    const int s = i / (SAM_C / 4), c4 = (i - s * (SAM_C / 4)) * 4;
    *reinterpret_cast<f32x4*>(cs + s * PITCH + c4) = *reinterpret_cast<const f32x4*>(ctx + s * SAM_C + c4);
  }
  __syncthreads();
  // 8 * D (head, e) pairs; with D = 16 the 256 threads split each pair's tokens in two halves
  constexpr int PAIRS = 8 * D, SPLIT = 256 / PAIRS;       // D = 16: 128 pairs, 2 token groups; D = 32: 256 pairs, 1
  const int pair = tid % PAIRS, grp = tid / PAIRS;
  const int h = pair / D;
  const float* wr = p.w + pair;                            // w^T: [256][heads * d], a wave reads 256 consecutive bytes per c
  constexpr int NJ = 8 / SPLIT;                            // tokens grp, grp + SPLIT, ... of the head
  float acc[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) acc[i] = 0.0f;
  for (int c4 = 0; c4 < SAM_C; c4 += 4) {
    f32x4 w4;
#pragma unroll
    for (int e = 0; e < 4; ++e) w4[e] = wr[(c4 + e) * PAIRS];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(cs + (h * 8 + grp + SPLIT * i) * PITCH + c4);
      acc[i] += x4[0] * w4[0] + x4[1] * w4[1] + x4[2] * w4[2] + x4[3] * w4[3];
    }
  }
  const float bv = p.bias ? p.bias[pair] : 0.0f;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = grp + SPLIT * i;
    if (j < p.n) p.out[((long long)b * p.n + j) * PAIRS + pair] = acc[i] + bv;
  }
}


// ---- the tokens' self attention (TwoWayAttentionBlock.self_attn on the <= 8 prompt tokens, 8 heads): one workgroup per
// prompt, everything in fp32 from the fp16 projections (rounds 3-5: two batched fp32 matmuls of 8192 7 x 32 x 7 problems,
// ~95 us each, plus the casts and transposes around them).
struct SamSelfParams {
  const f16* q; const f16* k; const f16* v;   // [B][n][256] (row stride 256)
  f16* out;                                   // [B][n][256]
  int B, n;
  float scale;
};

__global__ __launch_bounds__(256) void ea_sam_token_self_attn_kernel(SamSelfParams p) {
  EA_SMEM(smem);
  float* qs = reinterpret_cast<float*>(smem);      // [8][256]
  float* ks = qs + 8 * SAM_C;
  float* vs = ks + 8 * SAM_C;
  float* sc = vs + 8 * SAM_C;                      // [8 heads][8][8] scores -> probabilities
  const int tid = threadIdx.x, b = blockIdx.x, n = p.n;
  const long long base = (long long)b * n * SAM_C;
  for (int i = tid; i < n * SAM_C; i += 256) {
    qs[i] = (float)p.q[base + i]; ks[i] = (float)p.k[base + i]; vs[i] = (float)p.v[base + i];
  }
  __syncthreads();
  for (int i = tid; i < 8 * n * n; i += 256) {
    const int h = i / (n * n), r = i - h * n * n, qi = r / n, kj = r - qi * n;
    const float* a = qs + qi * SAM_C + h * 32;
    const float* c = ks + kj * SAM_C + h * 32;
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < 32; ++e) acc += a[e] * c[e];
    sc[(h * 8 + qi) * 8 + kj] = acc * p.scale;
  }
  __syncthreads();
  if (tid < 8 * n) {
    const int h = tid / n, qi = tid - h * n;
    float* row = sc + (h * 8 + qi) * 8;
    float m = row[0];
    for (int j = 1; j < n; ++j) m = fmaxf(m, row[j]);
    float sum = 0.0f;
    for (int j = 0; j < n; ++j) { const float e = expf(row[j] - m); row[j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < n; ++j) row[j] *= inv;
  }
  __syncthreads();
  const int h = tid >> 5;                            // thread = output channel h * 32 + e
  for (int qi = 0; qi < n; ++qi) {
    const float* row = sc + (h * 8 + qi) * 8;
    float acc = 0.0f;
    for (int j = 0; j < n; ++j) acc += row[j] * vs[j * SAM_C + tid];
    p.out[base + qi * SAM_C + tid] = (f16)acc;
  }
}

}  // namespace

extern "C" int ea_sam_vo_perm(int s) { return (s & ~31) + sam_perm32(s & 31); }

extern "C" int ea_sam_i2t_f16(const void* kp, long long kp_sb, const void* k, long long k_sb, const void* pe, const void* g2,
                              const float* cbias, const void* vo, const float* bo, const float* ln_g, const float* ln_b, float eps,
                              float scale, void* k_out, void* kp_out, int B, int T, int C, void* stream) {
  if (!k || !g2 || !cbias || !vo || !bo || !ln_g || !ln_b || !k_out) return EA_ERR_BAD_ARG;
  if ((kp_out || !kp) && !pe) return EA_ERR_BAD_ARG;   // kp == NULL: the operand keys + pe is formed in the kernel
  if (C != SAM_C) return EA_ERR_UNSUPPORTED;
  if (B <= 0 || T <= 0 || (kp_sb & 7) || (k_sb & 7)) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)kp | (uintptr_t)k | (uintptr_t)pe | (uintptr_t)g2 | (uintptr_t)vo | (uintptr_t)k_out | (uintptr_t)kp_out) & 15) return EA_ERR_BAD_ARG;
  SamI2tParams p;
  p.kp = (const f16*)kp; p.kp_sb = kp_sb; p.k = (const f16*)k; p.k_sb = k_sb; p.pe = (const f16*)pe;
  p.g2 = (const f16*)g2; p.cbias = cbias; p.vo = (const f16*)vo; p.bo = bo; p.ln_g = ln_g; p.ln_b = ln_b;
  p.eps = eps; p.scale = scale; p.k_out = (f16*)k_out; p.kp_out = (f16*)kp_out; p.B = B; p.T = T;
  const int chunks = (T + I2T_TOK_PER_WG - 1) / I2T_TOK_PER_WG;
  const int smem = 64 * 512 + 256 * 128 + 3 * SAM_C * 4;
  auto kfn = ea_sam_i2t_kernel;
  ea_allow_big_lds(kfn, smem);
  EA_LAUNCH(kfn, dim3((unsigned)(B * chunks)), dim3(256), smem, stream, p);
  return ea_launch_status();
}

extern "C" int ea_sam_upscale_tail_f16(const void* u0, const float* ln_g, const float* ln_b, float eps, const void* w1, const float* b1,
                                       const float* hyper, float* masks, int B, int h, int w, int m0, int nm, void* stream) {
  if (!u0 || !ln_g || !ln_b || !w1 || !b1 || !hyper || !masks) return EA_ERR_BAD_ARG;
  if (B <= 0 || h <= 0 || w <= 0 || ((long long)h * w * 4) % 16 || m0 < 0 || nm <= 0 || m0 + nm > 4) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)u0 | (uintptr_t)w1) & 15 || ((uintptr_t)masks & 7)) return EA_ERR_BAD_ARG;
  SamTailParams p;
  p.u0 = (const f16*)u0; p.ln_g = ln_g; p.ln_b = ln_b; p.eps = eps; p.w1 = (const f16*)w1; p.b1 = b1; p.hyper = hyper;
  p.masks = masks; p.B = B; p.h = h; p.w = w; p.m0 = m0; p.nm = nm;
  const long long ntiles = (long long)B * h * w * 4 / 16;
  long long nb = (ntiles + 3) / 4;
  if (nb > 8192) nb = 8192;
  auto kfn = ea_sam_upscale_tail_kernel;
  EA_LAUNCH(kfn, dim3((unsigned)nb), dim3(256), 0, stream, p);
  return ea_launch_status();
}

extern "C" int ea_sam_t2i_f16(const void* k, long long k_sb, const void* pe, const void* g, float scale, float* ctx, int B, int T, int C,
                              void* stream) {
  if (!k || !pe || !g || !ctx) return EA_ERR_BAD_ARG;
  if (C != SAM_C) return EA_ERR_UNSUPPORTED;
  if (B <= 0 || T <= 0 || (T % T2I_TILE) || (k_sb & 7)) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)k | (uintptr_t)pe | (uintptr_t)g | (uintptr_t)ctx) & 15) return EA_ERR_BAD_ARG;
  SamT2iParams p;
  p.k = (const f16*)k; p.k_sb = k_sb; p.pe = (const f16*)pe; p.g = (const f16*)g; p.ctx = ctx; p.scale = scale; p.B = B; p.T = T;
  const int smem = 2 * T2I_TILE * T2I_PITCH;
  auto kfn = ea_sam_t2i_kernel;
  ea_allow_big_lds(kfn, smem);
  EA_LAUNCH(kfn, dim3((unsigned)B), dim3(256), smem, stream, p);
  return ea_launch_status();
}

extern "C" int ea_sam_fold_heads_f16(const float* x, const float* w, const int* perm, void* out, int B, int n, int heads, int d_head,
                                     int C, int c_major, void* stream) {
  if (!x || !w || !out) return EA_ERR_BAD_ARG;
  if (C != SAM_C || heads != 8 || (d_head != 16 && d_head != 32)) return EA_ERR_UNSUPPORTED;
  if (B <= 0 || n <= 0 || n > 8) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) return EA_ERR_BAD_ARG;
  SamFoldParams p;
  p.x = x; p.w = w; p.perm = perm; p.out = (f16*)out; p.B = B; p.n = n; p.d = d_head; p.cs = c_major ? 1 : 0;
  const int smem = 8 * 8 * d_head * 4 + 256 + 64 * FOLD_PITCH * 2;
  if (d_head == 16) { auto kfn = ea_sam_fold_kernel<16>; EA_LAUNCH(kfn, dim3((unsigned)B), dim3(256), smem, stream, p); }
  else { auto kfn = ea_sam_fold_kernel<32>; EA_LAUNCH(kfn, dim3((unsigned)B), dim3(256), smem, stream, p); }
  return ea_launch_status();
}

extern "C" int ea_sam_unfold_heads_f32(const float* ctx, const float* w, const float* bias, float* out, int B, int n, int heads,
                                       int d_head, int C, void* stream) {
  if (!ctx || !w || !out) return EA_ERR_BAD_ARG;
  if (C != SAM_C || heads != 8 || (d_head != 16 && d_head != 32)) return EA_ERR_UNSUPPORTED;
  if (B <= 0 || n <= 0 || n > 8) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)ctx | (uintptr_t)w) & 15) return EA_ERR_BAD_ARG;
  SamUnfoldParams p;
  p.ctx = ctx; p.w = w; p.bias = bias; p.out = out; p.B = B; p.n = n; p.d = d_head;
  const int smem = 64 * (SAM_C + 4) * 4;
  if (d_head == 16) {
    auto kfn = ea_sam_unfold_kernel<16>;
    ea_allow_big_lds(kfn, smem);
    EA_LAUNCH(kfn, dim3((unsigned)B), dim3(256), smem, stream, p);
  } else {
    auto kfn = ea_sam_unfold_kernel<32>;
    ea_allow_big_lds(kfn, smem);
    EA_LAUNCH(kfn, dim3((unsigned)B), dim3(256), smem, stream, p);
  }
  return ea_launch_status();
}

extern "C" int ea_sam_token_self_attn_f16(const void* q, const void* k, const void* v, void* out, int B, int n, int heads, int C,
                                          float scale, void* stream) {
  if (!q || !k || !v || !out) return EA_ERR_BAD_ARG;
  if (C != SAM_C || heads != 8) return EA_ERR_UNSUPPORTED;
  if (B <= 0 || n <= 0 || n > 8) return EA_ERR_BAD_SHAPE;
  SamSelfParams p;
  p.q = (const f16*)q; p.k = (const f16*)k; p.v = (const f16*)v; p.out = (f16*)out; p.B = B; p.n = n; p.scale = scale;
  auto kfn = ea_sam_token_self_attn_kernel;
  EA_LAUNCH(kfn, dim3((unsigned)B), dim3(256), (3 * 8 * SAM_C + 8 * 64) * 4, stream, p);
  return ea_launch_status();
}

extern "C" int ea_sam_upscale_f16(const void* k, const void* w0, const float* b0, const float* ln_g, const float* ln_b, float eps,
                                  const void* w1, const float* b1, const float* hyper, float* masks, int B, int h, int w, int m0, int nm,
                                  void* stream) {
  if (!k || !w0 || !b0 || !ln_g || !ln_b || !w1 || !b1 || !hyper || !masks) return EA_ERR_BAD_ARG;
  if (B <= 0 || h <= 0 || w <= 0 || m0 < 0 || nm <= 0 || m0 + nm > 4) return EA_ERR_BAD_SHAPE;
  if ((h * w) % 16) return EA_ERR_UNSUPPORTED;          // a wave's 16 tokens belong to one prompt
  if (((uintptr_t)k | (uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)masks) & 15) return EA_ERR_BAD_ARG;
  SamUpParams p;
  p.k = (const f16*)k; p.w0 = (const f16*)w0; p.b0 = b0; p.ln_g = ln_g; p.ln_b = ln_b; p.eps = eps; p.w1 = (const f16*)w1; p.b1 = b1;
  p.hyper = hyper; p.masks = masks; p.B = B; p.h = h; p.w = w; p.m0 = m0; p.nm = nm;
  const long long ntiles = (long long)B * h * w / 16;
  long long grid = (ntiles + 7) / 8;
  if (grid > 256) grid = 256;                            // persistent: one workgroup (8 waves) per CU holds the first weight in LDS
  auto kfn = ea_sam_upscale_fused_kernel;
  ea_allow_big_lds(kfn, UP_W0_LDS);
  EA_LAUNCH(kfn, dim3((unsigned)grid), dim3(512), UP_W0_LDS, stream, p);
  return ea_launch_status();
}
