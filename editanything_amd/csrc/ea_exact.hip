// ea_exact.hip -- the fp32-ACCURATE SAM mode (editanything_amd/sam_exact.py) on the fp16 matrix cores.
//
// The reference never halves SAM (sam2image.py:69-70, editany_lora.py:87-94: `sam.to(device)` only), so the id map that
// conditions the ControlNet comes from fp32 arithmetic.  gfx950 has no fast fp32-input matrix path (the f32 MFMA runs at the
// vector rate), so fp32-accurate products are built from fp16 MFMAs on SPLIT operands:
//     v = hi + 2^-11 lo,   hi = fp16(v),  lo = fp16(2^11 (v - hi))          (22 mantissa bits; lo scaled out of the denormals)
//     a b ~= a_hi b_hi + 2^-11 (a_hi b_lo + a_lo b_hi)                      (products exact, fp32 accumulate; lo*lo ~ 2^-22 dropped)
// Three kernels:
//   ea_split3_f32            x (fp32, optional exact GELU first) -> [hi | lo | hi] fp16 rows of 3K: the A operand of ONE
//                            K-concatenated contraction against [W_lo | W_hi | W_hi] whose accumulator is multiplied by 2^-11
//                            after the first 2K columns (ea_epilogue.acc_scale_k) -- an exact Linear in one launch
//   ea_layernorm_split3_f32  LayerNorm (fp32 statistics and affine) fused in front of that split, with an optional output
//                            row map (SAM's window_partition layout; unmapped pad rows stay zero)
//   ea_attention_exact_f32   softmax(scale q k^T + rel-pos bias) v on fp32 q / k / v with split-operand MFMAs for BOTH
//                            products (P is split too), fp32 online softmax -- one pass over K / V, no score matrix in HBM
#include "ea_platform.h"
#include "../../include/editanything_hip.h"

namespace {

constexpr float EX_LO = 2048.0f;           // 2^11
constexpr float EX_ILO = 1.0f / 2048.0f;

__device__ __forceinline__ int ex_min(int a, int b) { return a < b ? a : b; }

__device__ __forceinline__ void ex_split(float v, f16& hi, f16& lo) {
  hi = (f16)v;
  lo = (f16)((v - (float)hi) * EX_LO);
}

__device__ __forceinline__ float ex_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ------------------------------------------------------------------------------------------------ split3
__global__ __launch_bounds__(256) void ea_split3_kernel(const float* x, f16* out, long long M, int K, int act) {
  const int k4 = K >> 2;
  const long long total = M * k4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / k4;
    const int c = (int)(i - m * k4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + m * K + c);
    f16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = act == EA_ACT_GELU ? ex_gelu(v[j]) : v[j];
      f16 a, b;
      ex_split(t, a, b);
      hi[j] = a;
      lo[j] = b;
    }
    f16* o = out + m * 3 * K + c;
    *reinterpret_cast<f16x4*>(o) = hi;
    *reinterpret_cast<f16x4*>(o + K) = lo;
    *reinterpret_cast<f16x4*>(o + 2 * K) = hi;
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm + split3
// one wave per row; the row stays in registers between the two statistics passes and the output (C <= 4096)
constexpr int LNS_MAXV = 16;
__global__ __launch_bounds__(256) void ea_ln_split3_kernel(const float* x, const float* gamma, const float* beta, float eps, f16* out,
                                                           int M, int C, const int* out_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int c4 = C >> 2;
  const float* xr = x + (long long)row * C;
  f32x4 v[LNS_MAXV];
  float s = 0.0f;
#pragma unroll
  for (int u = 0; u < LNS_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < c4) {
      v[u] = *reinterpret_cast<const f32x4*>(xr + 4 * i);
      s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
  }
  const float mean = ea_wave_sum(s) / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int u = 0; u < LNS_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < c4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[u][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(ea_wave_sum(q) / (float)C + eps);
  const int orow = out_rows ? out_rows[row] : row;
  if (orow < 0) return;
  f16* o = out + (long long)orow * 3 * C;
#pragma unroll
  for (int u = 0; u < LNS_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < c4) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * i), b = *reinterpret_cast<const f32x4*>(beta + 4 * i);
      f16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16 a, bb;
        ex_split((v[u][j] - mean) * rstd * g[j] + b[j], a, bb);
        hi[j] = a;
        lo[j] = bb;
      }
      *reinterpret_cast<f16x4*>(o + 4 * i) = hi;
      *reinterpret_cast<f16x4*>(o + C + 4 * i) = lo;
      *reinterpret_cast<f16x4*>(o + 2 * C + 4 * i) = hi;
    }
  }
}

// ------------------------------------------------------------------------------------------------ exact attention
struct ExAttnParams {
  const float* q; const float* k; const float* v; float* out;
  int B, H, N;
  long long s_b, s_n, o_sb, o_sn;     // element (b, i, h, d) of q / k / v at b * s_b + i * s_n + h * D + d (floats); out likewise
  float scale;
  const float* bias_h; const float* bias_w;   // [B * H][N][S] fp32 each (decomposed rel-pos: key j -> (j / S, j % S)), or NULL
  int S;
};

// Workgroup = 4 waves x 32 queries; key tiles of 32.  Everything is computed TRANSPOSED so that a lane owns ONE query:
// S^T = K Q^T (v_mfma_f32_32x32x16_f16: lane (q = lane % 32, half = lane / 32) holds the 16 keys
// key(r) = (r & 3) + 8 (r >> 2) + 4 half of its query -- row max and row sum are 16 in-lane values + one cross-half shuffle),
// O^T = V^T P^T with the keys of a k-step taken in exactly that order on both operands (a sum over keys does not care).
// A K / V tile is split into hi / lo ONCE per workgroup, on its way from registers (the next tile's global loads are in
// flight while the current one is computed) into LDS: K as [key][d] rows, V transposed as [d][key position] with the keys
// in the order the P^T operand holds them, so every MFMA fragment of the loop is one 16-byte LDS read.
template <int D>
__global__ __launch_bounds__(256, 2) void ea_attn_exact_kernel(ExAttnParams p) {
  constexpr int KS = D / 16;          // k-steps of S^T
  constexpr int DT = (D + 31) / 32;   // 32-row tiles of O^T (value dimension, zero-padded)
  constexpr int LDK = D + 8;          // K rows: halfs per row (16-byte aligned rows, skewed banks)
  constexpr int LDV = 40;             // V^T rows: 32 key positions + 8 halfs
  constexpr int C4 = D / 4;           // float4 pieces per row
  constexpr int NLD = (32 * C4 + 255) / 256;   // pieces per thread
  static_assert(D % 16 == 0, "head dimension");
  EA_SMEM(smem);
  f16* KsH = reinterpret_cast<f16*>(smem);
  f16* KsL = KsH + 32 * LDK;
  f16* VtH = KsL + 32 * LDK;          // [32 DT][LDV]
  f16* VtL = VtH + 32 * DT * LDV;
  float* Tw = reinterpret_cast<float*>(VtL + 32 * DT * LDV);   // [128][2 (S + 1)]: bias_h | bias_w rows of this workgroup's queries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int N = p.N, S = p.S;
  const int qw = blockIdx.x * 128;                    // first query of the workgroup
  const int qi = ex_min(qw + wave * 32 + l31, N - 1); // this lane's query (clamped: rows past N are computed and dropped)
  const float* qb = p.q + (long long)b * p.s_b + (long long)h * D;
  const float* kb = p.k + (long long)b * p.s_b + (long long)h * D;
  const float* vb = p.v + (long long)b * p.s_b + (long long)h * D;

  f16x8 qhi[KS], qlo[KS];
  {
    const float* qr = qb + (long long)qi * p.s_n;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qr + 16 * ks + 8 * half);
      const f32x4 c = *reinterpret_cast<const f32x4*>(qr + 16 * ks + 8 * half + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16 x, y;
        ex_split(a[j], x, y); qhi[ks][j] = x; qlo[ks][j] = y;
        ex_split(c[j], x, y); qhi[ks][4 + j] = x; qlo[ks][4 + j] = y;
      }
    }
  }
  // S % 32 == 0 (the global 64 x 64 grid): the 32 keys of a tile lie in ONE key row, so bias_h is one value per (query,
  // tile) -- read from global (the lane's own row, L1-resident) and only bias_w needs LDS: two workgroups fit a CU
  const bool hsame = S > 0 && (S & 31) == 0;
  const int TS = hsame ? S + 1 : 2 * (S + 1) + 1;     // odd row stride: the lanes of a wave read their own rows conflict-free
  const int WOFF = hsame ? 0 : S + 1;
  const float* bhrow = p.bias_h ? p.bias_h + ((long long)bh * N + qi) * S : nullptr;
  if (p.bias_h) {
    for (int idx = tid; idx < 128 * 2 * S; idx += 256) {
      const int ql = idx / (2 * S), c = idx - ql * 2 * S;
      const long long row = ((long long)bh * N + ex_min(qw + ql, N - 1)) * S;
      if (c >= S) Tw[ql * TS + WOFF + (c - S)] = p.bias_w[row + (c - S)];
      else if (!hsame) Tw[ql * TS + c] = p.bias_h[row + c];
    }
  }
  // zero the value rows past D once (D = 80: rows 80..95 of the third O^T tile)
  for (int idx = tid; idx < (32 * DT - D) * LDV; idx += 256) { VtH[D * LDV + idx] = (f16)0.0f; VtL[D * LDV + idx] = (f16)0.0f; }
  f32x16 om[DT], oc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { om[t][r] = 0.0f; oc[t][r] = 0.0f; }
  float m_run = -1.0e30f, l_run = 0.0f;
  const float* tw = Tw + (wave * 32 + l31) * TS;
  // uniform part of the (key / S, key % S) walk: key(r) = kt + (r & 3) + 8 (r >> 2) + 4 half
  const int q32 = S ? 32 / S : 0, r32 = S ? 32 - q32 * S : 0;
  int khb = 0, kwb = 0;                               // kt / S, kt % S

  f32x4 kreg[NLD], vreg[NLD];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 256 * u;
      if (idx < 32 * C4) {
        const int r = idx / C4, c4 = idx - r * C4;
        const long long off = (long long)ex_min(kt + r, N - 1) * p.s_n + 4 * c4;
        kreg[u] = *reinterpret_cast<const f32x4*>(kb + off);
        vreg[u] = *reinterpret_cast<const f32x4*>(vb + off);
      }
    }
  };
  fetch(0);
  for (int kt = 0; kt < N; kt += 32) {
    __syncthreads();                                   // the previous tile's readers are done (first pass: Tw / the zero rows are written)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 256 * u;
      if (idx < 32 * C4) {
        const int r = idx / C4, c4 = idx - r * C4;
        // position of key r in the P^T operand order: r = (j & 3) + 8 (rr >> 2) + 4 half with rr = 8 t + j  ->  (2 t + half) * 8 + j
        const int hv = (r >> 2) & 1, rr = (r & 3) + 4 * (r >> 3), pos = (2 * (rr >> 3) + hv) * 8 + (rr & 7);
        f16x4 kh4, kl4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f16 x, y;
          ex_split(kreg[u][j], x, y); kh4[j] = x; kl4[j] = y;
          ex_split(vreg[u][j], x, y);
          VtH[(4 * c4 + j) * LDV + pos] = x;
          VtL[(4 * c4 + j) * LDV + pos] = y;
        }
        *reinterpret_cast<f16x4*>(KsH + r * LDK + 4 * c4) = kh4;
        *reinterpret_cast<f16x4*>(KsL + r * LDK + 4 * c4) = kl4;
      }
    }
    __syncthreads();
    if (kt + 32 < N) fetch(kt + 32);                   // in flight under this tile's MFMAs
    // ---- S^T = K Q^T on split operands
    f32x16 sm, sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sm[r] = 0.0f; sc[r] = 0.0f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f16x8 khi = *reinterpret_cast<const f16x8*>(KsH + l31 * LDK + 16 * ks + 8 * half);
      const f16x8 klo = *reinterpret_cast<const f16x8*>(KsL + l31 * LDK + 16 * ks + 8 * half);
      sm = ea_mfma_32x32x16(khi, qhi[ks], sm);
      sc = ea_mfma_32x32x16(khi, qlo[ks], sc);
      sc = ea_mfma_32x32x16(klo, qhi[ks], sc);
    }
    // ---- scores of this lane's query against its 16 keys of the tile: scale, bias, mask, online softmax (fp32)
    float s[16];
    float tmax = -1.0e30f;
    const float bh_tile = hsame ? bhrow[ex_min(khb, S - 1)] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int koff = (r & 3) + 8 * (r >> 2) + 4 * half;
      float x = (sm[r] + sc[r] * EX_ILO) * p.scale;
      if (p.bias_h) {
        if (hsame) {
          x += bh_tile + tw[kwb + koff];                       // kwb in {0, 32, ...}, koff < 32: no wrap
        } else {
          int kwu = kwb + (r & 3) + 8 * (r >> 2), khu = khb;   // wave-uniform part of (key / S, key % S): scalar arithmetic
          while (kwu >= S) { kwu -= S; ++khu; }
          int kw = kwu + 4 * half, kh = khu;                   // S > 4: one more wrap at most
          if (kw >= S) { kw -= S; ++kh; }
          x += tw[ex_min(kh, S - 1)] + tw[WOFF + kw];
        }
      }
      if (kt + koff >= N) x = -1.0e30f;
      s[r] = x;
      tmax = fmaxf(tmax, x);
    }
    khb += q32; kwb += r32;
    if (kwb >= S && S) { kwb -= S; ++khb; }
    tmax = fmaxf(tmax, ea_shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { om[t][r] *= alpha; oc[t][r] *= alpha; }
    f16x8 pbh[2], pbl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = s[r] <= -1.0e29f ? 0.0f : expf(s[r] - m_new);
      l_run += pr;
      f16 x, y;
      ex_split(pr, x, y);
      pbh[r >> 3][r & 7] = x;
      pbl[r >> 3][r & 7] = y;
    }
    m_run = m_new;
    // ---- O^T += V^T P^T: k-step t takes the keys in the order they sit in this lane's score registers
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const f16x8 vhi = *reinterpret_cast<const f16x8*>(VtH + (32 * dt + l31) * LDV + (2 * t + half) * 8);
        const f16x8 vlo = *reinterpret_cast<const f16x8*>(VtL + (32 * dt + l31) * LDV + (2 * t + half) * 8);
        om[dt] = ea_mfma_32x32x16(vhi, pbh[t], om[dt]);
        oc[dt] = ea_mfma_32x32x16(vhi, pbl[t], oc[dt]);
        oc[dt] = ea_mfma_32x32x16(vlo, pbh[t], oc[dt]);
      }
  }
  const float l = l_run + ea_shfl_xor(l_run, 32);
  const float inv = 1.0f / l;
  if (qw + wave * 32 + l31 < N) {
    float* orow = p.out + (long long)b * p.o_sb + (long long)qi * p.o_sn + (long long)h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv0 = 32 * dt + 8 * g + 4 * half;
        if (dv0 < D) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (om[dt][4 * g + j] + oc[dt][4 * g + j] * EX_ILO) * inv;
          *reinterpret_cast<f32x4*>(orow + dv0) = o;
        }
      }
  }
}

template <int D>
static int launch_exact(const ExAttnParams& p, void* stream) {
  const int smem = (2 * 32 * (D + 8) + 2 * 32 * ((D + 31) / 32) * 40) * 2 + (p.bias_h ? 128 * ((p.S & 31) == 0 ? p.S + 1 : 2 * (p.S + 1) + 1) * 4 : 0);
  auto kfn = ea_attn_exact_kernel<D>;
  ea_allow_big_lds(kfn, smem);
  EA_LAUNCH(kfn, dim3((p.N + 127) / 128, p.B * p.H, 1), dim3(256), smem, stream, p);
  return ea_launch_status();
}

}  // namespace

extern "C" int ea_split3_f32(const float* x, void* out, long long M, int K, int act, void* stream) {
  if (!x || !out) return EA_ERR_BAD_ARG;
  if (M <= 0 || K <= 0 || (K & 3)) return EA_ERR_BAD_SHAPE;
  if (act != EA_ACT_NONE && act != EA_ACT_GELU) return EA_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)out) & 7)) return EA_ERR_BAD_ARG;
  long long nb = (M * (K >> 2) + 255) / 256;
  if (nb > 8192) nb = 8192;
  auto kfn = ea_split3_kernel;
  EA_LAUNCH(kfn, dim3((unsigned)nb), dim3(256), 0, stream, x, (f16*)out, M, K, act);
  return ea_launch_status();
}

extern "C" int ea_layernorm_split3_f32(const float* x, const float* gamma, const float* beta, float eps, void* out, int M, int C,
                                       const int* out_rows, void* stream) {
  if (!x || !gamma || !beta || !out) return EA_ERR_BAD_ARG;
  if (M <= 0 || C <= 0 || (C & 3) || (C >> 2) > 64 * LNS_MAXV) return EA_ERR_BAD_SHAPE;
  if ((((uintptr_t)x) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) return EA_ERR_BAD_ARG;
  if (((uintptr_t)out) & 7) return EA_ERR_BAD_ARG;
  auto kfn = ea_ln_split3_kernel;
  EA_LAUNCH(kfn, dim3((M + 3) / 4), dim3(256), 0, stream, x, gamma, beta, eps, (f16*)out, M, C, out_rows);
  return ea_launch_status();
}

extern "C" int ea_attention_exact_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int N, int D,
                                      long long s_b, long long s_n, long long o_sb, long long o_sn, float scale,
                                      const float* bias_h, const float* bias_w, int S, void* stream) {
  if (!q || !k || !v || !out) return EA_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || N <= 0) return EA_ERR_BAD_SHAPE;
  if ((s_n & 3) || (s_b & 3) || (o_sn & 3) || (o_sb & 3)) return EA_ERR_BAD_SHAPE;
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) return EA_ERR_BAD_ARG;
  if ((bias_h == nullptr) != (bias_w == nullptr)) return EA_ERR_BAD_ARG;
  if (bias_h && (S <= 4 || S > 64 || (long long)S * S < N)) return EA_ERR_BAD_SHAPE;
  ExAttnParams p{q, k, v, out, B, H, N, s_b, s_n, o_sb, o_sn, scale, bias_h, bias_w, bias_h ? S : 0};
  if (D == 64) return launch_exact<64>(p, stream);
  if (D == 80) return launch_exact<80>(p, stream);
  return EA_ERR_UNSUPPORTED;
}
