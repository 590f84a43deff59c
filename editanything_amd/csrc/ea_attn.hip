// ea_attn.hip -- fused softmax(Q K^T * scale [+ rel-pos bias]) V for gfx950.
//
// Replaces, behind one C entry point, the reference's three interchangeable
// attention paths -- CrossAttention.forward (ldm/modules/attention.py:163-194,
// fp32 QK^T + softmax), MemoryEfficientCrossAttention / xformers (:216-243) and
// the sliced monkey patch (cldm/hack.py:72-111) -- and SAM's Attention.forward
// with decomposed relative-position bias (segment_anything image_encoder.py,
// third party: attn = (q*scale) k^T + rel_h[..., None] + rel_w[..., None, :]).
//
// Wave64 / MFMA mapping (not a warp-32 flash port):
//  * a workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns
//    32 queries; K/V tiles of 64 keys are staged once in LDS for all 4 waves.
//  * scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_f16 with
//    A = K tile, B = Q^T held in registers), so a lane holds 32 scores of ONE
//    query: the softmax row max / sum are in-lane reductions plus a single
//    cross-half (lane ^ 32) exchange -- no 32-lane shuffle trees.
//  * O^T = V^T P^T: the probabilities go straight from the S^T accumulator
//    registers into the B operand (the key permutation MFMA's C layout imposes is
//    matched on the V^T side, contraction order being free), so P never touches
//    LDS.  V is transposed while it is written to LDS.
//  * online softmax in fp32 (exp2 domain), fp16 P for the PV MFMA, fp32 O.
// q/k/v are read in place from the [B, N, H*D] projection outputs (arbitrary
// row/batch strides, e.g. a fused QKV buffer); no head split/merge copies.
#include "ea_platform.h"
#include "../../include/editanything_hip.h"
#include <string.h>

namespace {

struct AttnParams {
  const f16* q; const f16* k; const f16* v; f16* o;
  int B, H, Nq, Nk;
  long long q_sb, q_sn, k_sb, k_sn, v_sb, v_sn, o_sb, o_sn;
  float scale;
  const float* bias_h; const float* bias_w;
  int S;            // rel-pos grid side (key j -> (j / S, j % S)); 0 = no bias
  unsigned magic;   // ceil(2^22 / S): j / S == (j * magic) >> 22 exactly for j < 16384, S <= 128
};

constexpr int ATT_BQ = 128;  // queries per workgroup
constexpr int ATT_BK = 64;   // keys per tile

// BIAS: 0 none; 1 decomposed rel-pos bias through a per-workgroup LDS table (any S <= 32: SAM's 14x14 windows);
// 2 the S == ATT_BK == 64 case (SAM global attention): a key tile is exactly one key row, so bias_h is ONE value per
// query per tile and bias_w is the same 32 values per lane for every tile -> registers, no per-score memory access.
template <int D, int BIAS>
__global__ __launch_bounds__(256) void ea_attn_kernel(AttnParams p) {
  constexpr int DQK = (D + 15) / 16 * 16;  // QK^T contraction length (zero padded)
  constexpr int NKS = DQK / 16;
  constexpr int NDT = (D + 31) / 32;       // 32-wide tiles of the head dim for O^T
  constexpr int KROW = DQK * 2 + 16;       // bytes per K row in LDS (pad -> conflict-free b128 reads)
  constexpr int VROW = ATT_BK * 2 + 8;     // bytes per V^T row in LDS
  constexpr int KCH = DQK / 8;             // 16-B chunks per K row
  constexpr int VCH = D / 8;
  constexpr int NKLD = (ATT_BK * KCH + 255) / 256;
  constexpr int NVLD = (ATT_BK * VCH + 255) / 256;
  EA_SMEM(smem);
  char* ks = smem;
  char* vs = smem + ATT_BK * KROW;
  float* bt = reinterpret_cast<float*>(smem + ATT_BK * KROW + NDT * 32 * VROW);   // BIAS == 1: [128][2S + 1] fp32

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh % p.H;
  const int q_row = blockIdx.x * ATT_BQ + wave * 32 + l31;
  const bool q_ok = q_row < p.Nq;
  const int q_ld = q_ok ? q_row : p.Nq - 1;

  const f16* qp = p.q + b * p.q_sb + (long long)h * D;
  const f16* kp = p.k + b * p.k_sb + (long long)h * D;
  const f16* vp = p.v + b * p.v_sb + (long long)h * D;

  // zero the V^T rows that pad D up to NDT*32 (never rewritten afterwards)
  for (int i = tid; i < (NDT * 32 - D) * (VROW / 2); i += 256)
    reinterpret_cast<f16*>(vs + D * VROW)[i] = (f16)0.0f;

  // Q^T fragments (B operand): lane holds Q[q][16s + 8*half .. +7]
  f16x8 qf[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const int d0 = 16 * s + 8 * half;
    if (q_ok && d0 < D) qf[s] = ea_ld8(qp + (long long)q_ld * p.q_sn + d0);
    else qf[s] = ea_zero8();
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int e = 0; e < NDT; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[e][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float sc2 = p.scale * 1.4426950408889634f;

  constexpr float LOG2E = 1.4426950408889634f;
  const long long brow = ((long long)bh * p.Nq + q_ld) * p.S;
  const int bt_ld = 2 * p.S + 1;
  const float* btq = bt + (wave * 32 + l31) * bt_ld;
  float bwr[BIAS == 2 ? 2 : 1][BIAS == 2 ? 16 : 1];
  float bh_next = 0.0f;
  if (BIAS == 1) {
    // table rows = this workgroup's 128 queries: [bias_h[q][0..S) | bias_w[q][0..S)] * log2(e)
    const int q0 = blockIdx.x * ATT_BQ;
    const int per_q = 2 * p.S;
    for (int i = tid; i < ATT_BQ * per_q; i += 256) {
      const int ql = i / per_q, c = i - ql * per_q;
      int qg = q0 + ql;
      if (qg >= p.Nq) qg = p.Nq - 1;
      const long long rb = ((long long)bh * p.Nq + qg) * p.S;
      bt[ql * bt_ld + c] = (c < p.S ? p.bias_h[rb + c] : p.bias_w[rb + c - p.S]) * LOG2E;
    }
  }
  if (BIAS == 2) {
    // C-layout row of register r = 4g + j is 8g + 4*half + j: four consecutive keys -> one 16-byte load
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.bias_w + brow + 32 * t + 8 * g + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; ++j) bwr[BIAS == 2 ? t : 0][BIAS == 2 ? 4 * g + j : 0] = w4[j] * LOG2E;
      }
    bh_next = p.bias_h[brow] * LOG2E;
  }

  f16x8 kreg[NKLD], vreg[NVLD];
  auto load_kv = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NKLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / KCH, d0 = (c - row * KCH) * 8;
      const int key = kt * ATT_BK + row;
      if (c < ATT_BK * KCH && key < p.Nk && d0 < D) kreg[i] = ea_ld8(kp + (long long)key * p.k_sn + d0);
      else kreg[i] = ea_zero8();
    }
#pragma unroll
    for (int i = 0; i < NVLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / VCH, d0 = (c - row * VCH) * 8;
      const int key = kt * ATT_BK + row;
      if (c < ATT_BK * VCH && key < p.Nk) vreg[i] = ea_ld8(vp + (long long)key * p.v_sn + d0);
      else vreg[i] = ea_zero8();
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int i = 0; i < NKLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / KCH, cc = c - row * KCH;
      if (c < ATT_BK * KCH) *reinterpret_cast<f16x8*>(ks + row * KROW + cc * 16) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < NVLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / VCH, d0 = (c - row * VCH) * 8;
      if (c < ATT_BK * VCH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f16*>(vs + (d0 + j) * VROW + row * 2) = vreg[i][j];
      }
    }
  };

  const int nkt = (p.Nk + ATT_BK - 1) / ATT_BK;
  load_kv(0);
  store_kv();
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_kv(kt + 1);
    const float bh_cur = bh_next;
    if (BIAS == 2 && kt + 1 < nkt) bh_next = p.bias_h[brow + kt + 1] * LOG2E;

    // ---- S^T = K Q^T : two 32-key tiles
    f32x16 sacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[t][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(ks + (32 * t + l31) * KROW + (16 * s + 8 * half) * 2);
        sacc[t] = ea_mfma_32x32x16(a, qf[s], sacc[t]);
      }
    }
    // ---- scale, bias, mask, running max
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * ATT_BK + 32 * t + ea_mfma_row(r, lane);
        float sv = sacc[t][r] * sc2;
        if (BIAS == 1) {
          if (key < p.Nk) {
            const int kh = (int)(((unsigned)key * p.magic) >> 22), kw = key - kh * p.S;
            sv += btq[kh] + btq[p.S + kw];
          }
        }
        if (BIAS == 2) sv += bh_cur + bwr[BIAS == 2 ? t : 0][BIAS == 2 ? r : 0];
        if (key >= p.Nk) sv = -INFINITY;
        sacc[t][r] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, ea_shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;
    const float alpha = exp2f(m_run - m_use);
    m_run = m_new;
    float psum = 0.0f;
    f16x8 pb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2f(sacc[t][r] - m_use);
        psum += pv;
        pb[t][r >> 3][r & 7] = (f16)pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int e = 0; e < NDT; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[e][r] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int e = 0; e < NDT; ++e)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const char* base = vs + (32 * e + l31) * VROW + (32 * t + 16 * u + 4 * half) * 2;
          const f16x4 lo = *reinterpret_cast<const f16x4*>(base);
          const f16x4 hi = *reinterpret_cast<const f16x4*>(base + 16);
          f16x8 a;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
          oacc[e] = ea_mfma_32x32x16(a, pb[t][u], oacc[e]);
        }

    __syncthreads();
    if (kt + 1 < nkt) {
      store_kv();
      __syncthreads();
    }
  }

  // ---- normalise and store: lane holds O[q][32e + 8g + 4*half + 0..3]
  const float l_tot = l_run + ea_shfl_xor(l_run, 32);
  const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (q_ok) {
    f16* op = p.o + b * p.o_sb + (long long)q_row * p.o_sn + (long long)h * D;
#pragma unroll
    for (int e = 0; e < NDT; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * e + 8 * g + 4 * half;
        if (d0 < D) {
          f16x4 o4;
#pragma unroll
          for (int j = 0; j < 4; ++j) o4[j] = (f16)(oacc[e][4 * g + j] * inv);
          *reinterpret_cast<f16x4*>(op + d0) = o4;
        }
      }
  }
}

template <int D, int BIAS>
static int launch_attn(const AttnParams& p, void* stream) {
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int NDT = (D + 31) / 32;
  const int smem = ATT_BK * (DQK * 2 + 16) + NDT * 32 * (ATT_BK * 2 + 8) + (BIAS == 1 ? ATT_BQ * (2 * p.S + 1) * 4 : 0);
  auto kfn = ea_attn_kernel<D, BIAS>;
  ea_allow_big_lds(kfn, smem);
  dim3 grid((p.Nq + ATT_BQ - 1) / ATT_BQ, p.B * p.H, 1);
  EA_LAUNCH(kfn, grid, dim3(256), smem, stream, p);
  return ea_launch_status();
}

// rel-pos tables: one thread per (bh, q, k<2S): dot over D channels.
struct RelposParams {
  const f16* q; int B, H, S, D;
  long long q_sb, q_sn;
  const f16* rel_h; const f16* rel_w;
  float* bias_h; float* bias_w;
};

__global__ __launch_bounds__(256) void ea_relpos_kernel(RelposParams p) {
  const int N = p.S * p.S;
  const long long total = (long long)p.B * p.H * N * 2 * p.S;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int kk = (int)(idx % (2 * p.S));
    const long long rest = idx / (2 * p.S);
    const int qi = (int)(rest % N);
    const int bh = (int)(rest / N);
    const int b = bh / p.H, h = bh % p.H;
    const int qh = qi / p.S, qw = qi % p.S;
    const bool is_w = kk >= p.S;
    const int kpos = is_w ? kk - p.S : kk;
    const int rel = (is_w ? qw : qh) - kpos + p.S - 1;
    const f16* qv = p.q + b * p.q_sb + (long long)qi * p.q_sn + (long long)h * p.D;
    const f16* rv = (is_w ? p.rel_w : p.rel_h) + (long long)rel * p.D;
    float acc = 0.0f;
    for (int c = 0; c < p.D; c += 8) {
      f16x8 a = ea_ld8(qv + c), r8 = ea_ld8(rv + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)r8[j];
    }
    float* dst = (is_w ? p.bias_w : p.bias_h) + ((long long)bh * N + qi) * p.S + kpos;
    *dst = acc;
  }
}

}  // namespace

extern "C" int ea_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq,
                                int Nk, int D, long long q_sb, long long q_sn, long long k_sb, long long k_sn,
                                long long v_sb, long long v_sn, long long o_sb, long long o_sn, float scale,
                                const float* bias_h, const float* bias_w, int S, void* stream) {
  if (!q || !k || !v || !out) return EA_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return EA_ERR_BAD_SHAPE;
  if ((q_sn & 7) || (k_sn & 7) || (v_sn & 7) || (o_sn & 3) || (q_sb & 7) || (k_sb & 7) || (v_sb & 7) || (o_sb & 3))
    return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 7)) return EA_ERR_BAD_ARG;
  if (S != 0 && (!bias_h || !bias_w)) return EA_ERR_BAD_ARG;
  if (S != 0 && Nk != S * S) return EA_ERR_BAD_SHAPE;
  AttnParams p;
  p.q = (const f16*)q; p.k = (const f16*)k; p.v = (const f16*)v; p.o = (f16*)out;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_sb = q_sb; p.q_sn = q_sn; p.k_sb = k_sb; p.k_sn = k_sn; p.v_sb = v_sb; p.v_sn = v_sn;
  p.o_sb = o_sb; p.o_sn = o_sn;
  p.scale = scale; p.bias_h = bias_h; p.bias_w = bias_w;
  p.S = S;
  p.magic = S > 0 ? (unsigned)(((1u << 22) + S - 1) / S) : 0u;
  if (S == 0) {
    switch (D) {
      case 40: return launch_attn<40, 0>(p, stream);
      case 64: return launch_attn<64, 0>(p, stream);
      case 80: return launch_attn<80, 0>(p, stream);
      case 160: return launch_attn<160, 0>(p, stream);
      default: return EA_ERR_UNSUPPORTED;
    }
  }
  if (S == ATT_BK) {   // SAM global attention (64 x 64 token grid); bias rows are 256-B aligned fp32
    if (((uintptr_t)bias_h & 15) || ((uintptr_t)bias_w & 15)) return EA_ERR_BAD_ARG;
    switch (D) {
      case 64: return launch_attn<64, 2>(p, stream);
      case 80: return launch_attn<80, 2>(p, stream);
      default: return EA_ERR_UNSUPPORTED;
    }
  }
  if (S > 32) return EA_ERR_UNSUPPORTED;   // window sizes: the per-workgroup bias table must fit next to the K/V tiles
  switch (D) {
    case 64: return launch_attn<64, 1>(p, stream);
    case 80: return launch_attn<80, 1>(p, stream);
    default: return EA_ERR_UNSUPPORTED;
  }
}

extern "C" int ea_relpos_tables_f16(const void* q, int B, int H, int S, int D, long long q_sb, long long q_sn,
                                    const void* rel_h, const void* rel_w, float* bias_h, float* bias_w,
                                    void* stream) {
  if (!q || !rel_h || !rel_w || !bias_h || !bias_w) return EA_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || S <= 0 || D <= 0 || (D & 7) || (q_sn & 7) || (q_sb & 7)) return EA_ERR_BAD_SHAPE;
  RelposParams p;
  p.q = (const f16*)q; p.B = B; p.H = H; p.S = S; p.D = D; p.q_sb = q_sb; p.q_sn = q_sn;
  p.rel_h = (const f16*)rel_h; p.rel_w = (const f16*)rel_w; p.bias_h = bias_h; p.bias_w = bias_w;
  const long long total = (long long)B * H * S * S * 2 * S;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  auto kfn = ea_relpos_kernel;
  EA_LAUNCH(kfn, dim3((unsigned)nb), dim3(256), 0, stream, p);
  return ea_launch_status();
}
