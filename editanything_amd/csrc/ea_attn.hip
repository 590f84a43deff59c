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
#include "ea_prims.h"
#include "../../include/editanything_hip.h"
#include <string.h>
#include <type_traits>

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

#ifndef EA_ATTN_PRIO
#define EA_ATTN_PRIO 0   // measured null on this kernel (270 vs 268 us at N = 4096, d = 64)
#endif
constexpr int ATT_BQ = 128;  // queries per workgroup
constexpr int ATT_BK = 64;   // keys per tile
// Deferred rescale (guide T13): the running max only moves (and O / l are only rescaled) when some lane's tile max
// exceeds it by more than this many log2 units; until then P = exp2(s - m_run) <= 2^8, exact in fp16's range, l and O
// accumulate in fp32 -- the result differs from the always-rescale form only by fp32 rounding.
constexpr float ATT_DEFER = 8.0f;
#ifndef EA_ATTN_EXP
#define EA_ATTN_EXP 0
#endif
#ifdef EA_EMU
constexpr long long ATT_G2_MIN_WG = 2;      // host emulation: small test shapes must reach the two-group kernel
#else
constexpr long long ATT_G2_MIN_WG = 256;    // CUs of an MI355X
#endif

// LDS hand-off between the lanes of one wave (the LDS pipeline is in order per wave)
__device__ __forceinline__ void ea_wave_lds_sync_() {
#ifdef EA_EMU
  ea_emu::wave_sync();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#endif
}
// raise this wave's issue priority around its MFMA clusters (guide T5): the co-resident waves' VALU softmax work then
// fills the gaps instead of delaying the matrix instructions
template <int P>
__device__ __forceinline__ void ea_setprio() {
#ifndef EA_EMU
  __builtin_amdgcn_s_setprio(P);
#endif
}
__device__ __forceinline__ float ea_exp2(float x) {
#ifdef EA_EMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);   // bare v_exp_f32 (inputs here are <= 8, denormal results flush to 0)
#endif
}
// max(a, b, c) as ONE v_max3_f32: written through fmaxf the compiler first canonicalises MFMA outputs (a v_pk_mul_f32
// by 1.0 per pair -- 16 extra VALU issues per tile in a loop that is VALU-issue bound)
__device__ __forceinline__ float ea_max3(float a, float b, float c) {
#ifdef EA_EMU
  return fmaxf(a, fmaxf(b, c));
#else
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#endif
}
// c + a.x + a.y (fp32 accumulate)
__device__ __forceinline__ float ea_dot2_ones(f16x2 a, float c) {
#ifdef EA_EMU
  return c + (float)a[0] + (float)a[1];
#else
  f16x2 one;
  one[0] = (f16)1.0f;
  one[1] = (f16)1.0f;
  return __builtin_amdgcn_fdot2(a, one, c, false);
#endif
}
__device__ __forceinline__ bool ea_wave_any(bool v) {
#ifdef EA_EMU
  char* sc = ea_emu::wave_scratch();
  const int l = ea_emu::lane_id();
  sc[l * 64] = v ? 1 : 0;
  ea_emu::wave_sync();
  bool r = false;
  for (int i = 0; i < ea_emu::wave_lanes(); ++i) r = r || sc[i * 64];
  ea_emu::wave_sync();
  return r;
#else
  return __any(v);
#endif
}

// Workgroup -> (query block, batch*head).  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with
// its own L2; taken as they come, every XCD would see every head's K/V (tens of MB against a 4-MiB L2, so each of a
// head's query blocks re-streams its 2 * Nk * D * 2 bytes from the fabric).  Re-indexing so that XCD x walks the
// contiguous slice x of the (head, query block) space keeps a head's K/V resident in the one L2 that uses it.
__device__ __forceinline__ void ea_attn_block(int& qb, int& bh) {
  const unsigned gx = gridDim.x, total = gx * gridDim.y;
  unsigned L = blockIdx.y * gx + blockIdx.x;
  const unsigned main = total & ~7u;
  if (L < main) L = (L & 7u) * (main >> 3) + (L >> 3);
  bh = (int)(L / gx);
  qb = (int)(L - (unsigned)bh * gx);
}

// BIAS: 0 none; 1 decomposed rel-pos bias through a per-workgroup LDS table (any S <= 32: SAM's 14x14 windows);
// 2 the S == ATT_BK == 64 case (SAM global attention): a key tile is exactly one key row, so bias_h is ONE value per
// query per tile and bias_w is the same 32 values per lane for every tile -> registers, no per-score memory access.
template <int D, int BIAS>
__global__ __launch_bounds__(256, (D <= 64 && BIAS == 0 ? 3 : (D <= 80 && BIAS != 1 ? 2 : 1))) void ea_attn_kernel(AttnParams p) {
  constexpr int DQK = (D + 15) / 16 * 16;  // QK^T contraction length (zero padded)
  constexpr int NKS = DQK / 16;
  constexpr int NDT = (D + 31) / 32;       // 32-wide tiles of the head dim for O^T
  constexpr int DV = NDT * 32;             // V row width in LDS (columns >= D stay zero)
  constexpr int KROW = DQK * 2 + 16;       // bytes per K row in LDS (pad -> conflict-free b128 reads)
  // V row stride: the transpose read of a 32-lane half touches 4 consecutive rows x 64 bytes; a stride of 64 or 192
  // mod 256 puts those four segments on disjoint banks
  constexpr int VROW = ((DV * 2) % 256 == 64 || (DV * 2) % 256 == 192) ? DV * 2 : DV * 2 + 64;
  static_assert(VROW % 256 == 64 || VROW % 256 == 192, "V row stride");
  constexpr int KCH = DQK / 8;             // 16-B chunks per K row
  constexpr int VCH = D / 8;
  constexpr int NKLD = (ATT_BK * KCH + 255) / 256;
  constexpr int NVLD = (ATT_BK * VCH + 255) / 256;
  constexpr int STAGE = ATT_BK * KROW + ATT_BK * VROW;   // one K/V tile; two of them form the ring
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr bool PRIO = EA_ATTN_PRIO;
  EA_SMEM(smem);
  float* bt = reinterpret_cast<float*>(smem + 2 * STAGE);   // BIAS == 1: [128][2S + 1] fp32

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int qblk, bh;
  ea_attn_block(qblk, bh);
  const int b = bh / p.H, h = bh % p.H;
  const int q_row = qblk * ATT_BQ + wave * 32 + l31;
  const bool q_ok = q_row < p.Nq;
  const int q_ld = q_ok ? q_row : p.Nq - 1;

  const f16* qp = p.q + b * p.q_sb + (long long)h * D;
  const f16* kp = p.k + b * p.k_sb + (long long)h * D;
  const f16* vp = p.v + b * p.v_sb + (long long)h * D;

  // zero the V columns that pad D up to DV in both stages (never rewritten afterwards)
  if (DV > D) {
    constexpr int PADC = DV - D;   // multiple of 8
    for (int i = tid; i < 2 * ATT_BK * (PADC / 8); i += 256) {
      const int st = i / (ATT_BK * (PADC / 8)), rem = i - st * (ATT_BK * (PADC / 8));
      const int row = rem / (PADC / 8), c = rem - row * (PADC / 8);
      *reinterpret_cast<f16x8*>(smem + st * STAGE + ATT_BK * KROW + row * VROW + (D + 8 * c) * 2) = ea_zero8();
    }
  }

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
  float m_run = -INFINITY, l_run = 0.0f;   // m_run in the scaled log2 domain
  const float sc2 = p.scale * LOG2E;

  const long long brow = ((long long)bh * p.Nq + q_ld) * p.S;
  const int bt_ld = 2 * p.S + 1;
  const float* btq = bt + (wave * 32 + l31) * bt_ld;
  float bwr[BIAS == 2 ? 2 : 1][BIAS == 2 ? 16 : 1];
  float bh_next = 0.0f;
  if (BIAS == 1) {
    // table rows = this workgroup's 128 queries: [bias_h[q][0..S) | bias_w[q][0..S)] * log2(e)
    const int q0 = qblk * ATT_BQ;
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

  // K/V staging (guide T14): global -> registers early, registers -> LDS late; both row-major 16-byte stores
  f16x8 kreg[NKLD], vreg[NVLD];
  // lean staging (every chunk of the tile is a real one: no per-thread predicates): wave-uniform tile base + a
  // loop-invariant 32-bit element offset per thread, so a load is one global_load with an SGPR base -- the generic
  // form below recomputes a 64-bit row * stride product per load (quarter-rate integer multiplies)
  constexpr bool LEAN = (ATT_BK * KCH) % 256 == 0 && KCH * 8 == D && (ATT_BK * VCH) % 256 == 0;
  unsigned koff[NKLD], voff[NVLD];
  if (LEAN) {
#pragma unroll
    for (int i = 0; i < NKLD; ++i) {
      const int c = tid + 256 * i, row = c / KCH;
      koff[i] = (unsigned)(row * p.k_sn + (c - row * KCH) * 8);
    }
#pragma unroll
    for (int i = 0; i < NVLD; ++i) {
      const int c = tid + 256 * i, row = c / VCH;
      voff[i] = (unsigned)(row * p.v_sn + (c - row * VCH) * 8);
    }
  }
  auto load_kv = [&](int kt) {
    if (LEAN) {
      const int left = p.Nk - kt * ATT_BK;      // keys in this tile (wave-uniform)
      const f16* kb = kp + (long long)kt * ATT_BK * p.k_sn;
      const f16* vb = vp + (long long)kt * ATT_BK * p.v_sn;
      if (left >= ATT_BK) {
#pragma unroll
        for (int i = 0; i < NKLD; ++i) kreg[i] = ea_ld8(kb + koff[i]);
#pragma unroll
        for (int i = 0; i < NVLD; ++i) vreg[i] = ea_ld8(vb + voff[i]);
      } else {
        // ragged last tile: rows past Nk re-read the last key (their scores are masked to -inf, so their
        // probabilities are exactly 0 and the duplicated V rows contribute nothing)
#pragma unroll
        for (int i = 0; i < NKLD; ++i) {
          const int c = tid + 256 * i;
          int row = c / KCH;
          const int d0 = (c - row * KCH) * 8;
          row = row < left ? row : left - 1;
          kreg[i] = ea_ld8(kb + (unsigned)(row * p.k_sn + d0));
        }
#pragma unroll
        for (int i = 0; i < NVLD; ++i) {
          const int c = tid + 256 * i;
          int row = c / VCH;
          const int d0 = (c - row * VCH) * 8;
          row = row < left ? row : left - 1;
          vreg[i] = ea_ld8(vb + (unsigned)(row * p.v_sn + d0));
        }
      }
      return;
    }
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
  auto store_kv = [&](int buf) {
    char* ks = smem + buf * STAGE;
    char* vs = ks + ATT_BK * KROW;
#pragma unroll
    for (int i = 0; i < NKLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / KCH, cc = c - row * KCH;
      if (LEAN || c < ATT_BK * KCH) *reinterpret_cast<f16x8*>(ks + row * KROW + cc * 16) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < NVLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / VCH, cc = c - row * VCH;
      if (LEAN || c < ATT_BK * VCH) *reinterpret_cast<f16x8*>(vs + row * VROW + cc * 16) = vreg[i];
    }
  };

  // per-lane part of the V^T fragment address: 16-lane group g -> key half (g >> 1), d block (g & 1); lane i of the
  // group supplies key row (i >> 2), columns 4*(i & 3) of the [4 keys][16 d] block
  const int vt_off = (4 * half + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  const int nkt = (p.Nk + ATT_BK - 1) / ATT_BK;
  load_kv(0);
  store_kv(0);
  if (nkt > 1) load_kv(1);
  __syncthreads();

  // One K/V tile.  MASKED is the ragged last tile (keys >= Nk get -inf): a separate instantiation, so the full tiles
  // carry no per-score compare/select work.
  auto tile = [&](int kt, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const char* ks = smem + (kt & 1) * STAGE;
    const char* vs = ks + ATT_BK * KROW;
    const float bh_cur = bh_next;
    if (BIAS == 2 && kt + 1 < nkt) bh_next = p.bias_h[brow + kt + 1] * LOG2E;

    // ---- S^T = K Q^T : two 32-key tiles
    f32x16 sacc[2];
    if (PRIO) ea_setprio<1>();
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
    if (PRIO) ea_setprio<0>();
    // ---- tile max.  BIAS == 0: on the raw scores (the positive scale is folded into the exponent's FMA);
    // with a bias the scores are first moved to the scaled log2 domain.
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = sacc[t][r];
        if (BIAS != 0) sv *= sc2;
        if (BIAS == 1) {
          const int key = kt * ATT_BK + 32 * t + ea_mfma_row(r, lane);
          if (!MASKED || key < p.Nk) {
            const int kh = (int)(((unsigned)key * p.magic) >> 22), kw = key - kh * p.S;
            sv += btq[kh] + btq[p.S + kw];
          }
        }
        if (BIAS == 2) sv += bh_cur + bwr[BIAS == 2 ? t : 0][BIAS == 2 ? r : 0];
        if (MASKED) {
          const int key = kt * ATT_BK + 32 * t + ea_mfma_row(r, lane);
          if (key >= p.Nk) sv = -INFINITY;
        }
        sacc[t][r] = sv;
        mx = fmaxf(mx, sv);
      }
    if (BIAS == 0) mx *= sc2;
    mx = fmaxf(mx, ea_shfl_xor(mx, 32));
    // ---- deferred rescale: move the running max only when some lane outgrew it by more than ATT_DEFER
    if (ea_wave_any(mx > m_run + ATT_DEFER)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == -INFINITY) ? 1.0f : ea_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int e = 0; e < NDT; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[e][r] *= alpha;
#ifndef EA_EMU
      asm volatile("" ::: "memory");   // a real (wave-uniform) branch: the common no-rescale path pays nothing
#endif
    }
    const float m_use = (m_run == -INFINITY) ? 0.0f : m_run;
    // the row sum is taken over the fp16-rounded probabilities -- the values the PV MFMA actually uses -- two per
    // v_dot2_f32_f16 (half the VALU instructions of an fp32 add chain; this loop is VALU-bound at D = 64)
    float psum = 0.0f;
    f16x8 pb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = (BIAS == 0) ? ea_exp2(fmaf(sacc[t][r], sc2, -m_use)) : ea_exp2(sacc[t][r] - m_use);
        const float p1 = (BIAS == 0) ? ea_exp2(fmaf(sacc[t][r + 1], sc2, -m_use)) : ea_exp2(sacc[t][r + 1] - m_use);
        f16x2 pp;
        pp[0] = (f16)p0;
        pp[1] = (f16)p1;
        psum = ea_dot2_ones(pp, psum);
        pb[t][r >> 3][r & 7] = pp[0];
        pb[t][r >> 3][(r & 7) + 1] = pp[1];
      }
    l_run += psum;

    // ---- O^T += V^T P^T.  P^T (B operand) comes straight from the S^T accumulator registers: MFMA k index
    // 8*half + j  <->  key 32t + 16u + 4*half + (j & 3) + 8*(j >> 2); the V^T fragment is gathered to match by two
    // transpose reads of 4 keys each (keys +0..3 and +8..11).
    const char* vbase = vs + vt_off;
    ea_static_for<NDT>([&](auto e_tag) {
      constexpr int e = decltype(e_tag)::value;
      f16x4 vlo[4], vhi[4];
      ea_static_for<4>([&](auto tu_tag) {
        constexpr int tu = decltype(tu_tag)::value;
        vlo[tu] = ea_lds_read_tr16<(16 * tu) * VROW + 64 * e>(vbase);
        vhi[tu] = ea_lds_read_tr16<(16 * tu + 8) * VROW + 64 * e>(vbase);
      });
      ea_lds_tr_wait();
      if (PRIO) ea_setprio<1>();
#pragma unroll
      for (int tu = 0; tu < 4; ++tu) {
        f16x8 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = vlo[tu][j]; a[4 + j] = vhi[tu][j]; }
        oacc[e] = ea_mfma_32x32x16(a, pb[tu >> 1][tu & 1], oacc[e]);
      }
      if (PRIO) ea_setprio<0>();
    });

    // ---- stage the next tile into the other buffer (last read during iteration kt - 1, i.e. before the barrier
    // every wave passed at the end of that iteration), then fetch the tile after it into the freed registers
    if (kt + 1 < nkt) {
      store_kv((kt + 1) & 1);
      if (kt + 2 < nkt) load_kv(kt + 2);
    }
    __syncthreads();
  };
  const int nfull = p.Nk / ATT_BK;   // tiles with no key mask
  for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
  if (nfull < nkt) tile(nfull, std::true_type{});

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


// ------------------------------------------------------------------------------------------------------------------
// Software-pipelined flash loop with LDS-DMA staging for the UNet's self-attention (d = 64, no bias, >= 4 key tiles).
// ea_attn_kernel runs a tile as QK^T -> softmax -> PV in one wave, in order.  Here each iteration issues the NEXT tile's
// score MFMAs ahead of this tile's softmax and this tile's PV MFMAs ahead of the next tile's max reduction, so the K
// ring runs one tile ahead of the V ring (iteration j reads K(j+1) and V(j)).  Everything else -- S^T = K Q^T operand
// swap, deferred rescale, fp16 probabilities with an fp32 dot2 row sum, V^T by transpose reads -- is ea_attn_kernel's
// arithmetic, so the results are bit-identical to it.
// Staging: a K / V tile goes global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB = 8 rows of 128 bytes per wave
// instruction): no staging registers, no ds_write, no vmcnt(0) in front of one.  (Round 2 first shipped this loop with
// global -> VGPR -> LDS staging; the counters showed it WAITING on that chain -- 40 % of a wave's life in s_waitcnt /
// s_barrier, profiles/r02_attention_counters.md -- and the DMA form measured +2.5..3.4 % in the same call, bit-identical.)
// The rings are three deep, so a tile is requested TWO iterations before its first read and the end-of-iteration wait is
// a counted vmcnt: the newest requests stay in flight across the barrier.
// The DMA writes lane-linear, so rows are 128 bytes unpadded and the bank spread comes from an XOR of the 16-byte chunk
// index applied on the SOURCE address (which chunk a lane fetches) and again on the fragment reads:
//   K (ds_read_b128, 32 consecutive rows per half wave):   chunk ^ ((row >> 1) & 7)   -- conflict-free per 16 lanes
//   V (ds_read_b64_tr_b16, 8 rows x 64 bytes per read):    chunk ^ (((row >> 1) & 1) << 2)
// The V form only flips the 64-byte half, so it commutes with the row immediates (multiples of 8) the transpose reads
// use; the two 32-channel blocks get one address register each.  Rows past Nk in the ragged last tile are fetched as
// zeros (offset past the descriptor's range); their scores are masked to -inf, their probabilities are exactly 0.
#ifdef EA_EMU
struct AttnRsrc { const char* base; unsigned limit; };
__device__ __forceinline__ AttnRsrc attn_make_rsrc(const void* p, unsigned bytes) { return AttnRsrc{(const char*)p, bytes}; }
__device__ __forceinline__ void attn_dma16(AttnRsrc r, unsigned voff, unsigned soff, char* lds_base) {
  char* dst = lds_base + 16 * ea_emu::lane_id();
  if (voff >= r.limit) memset(dst, 0, 16);
  else memcpy(dst, r.base + voff + soff, 16);
}
template <int N> __device__ __forceinline__ void attn_wait_dma() {}
__device__ __forceinline__ void attn_barrier() { __syncthreads(); }
#else
typedef __amdgpu_buffer_rsrc_t AttnRsrc;
__device__ __forceinline__ AttnRsrc attn_make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void attn_dma16(AttnRsrc r, unsigned voff, unsigned soff, char* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void attn_wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// never __syncthreads() in this loop: its fence drains vmcnt to 0
__device__ __forceinline__ void attn_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#endif
constexpr unsigned ATT_OOB = 0x80000000u;     // a byte offset past every buffer's num_records: the lane gets zeros

// G = 2 (round 3, EXPERIMENT -- slower, see launch_attn): a wave owns TWO 32-row query groups.  Every K / V fragment read from LDS feeds two MFMAs (the loop at G = 1
// asks the LDS for 32 bytes per MFMA cycle per wave: 128 B/clk/CU, all it has), the staging per query row halves, and -- the
// point -- the wave carries two independent softmax chains: group 0's exponentials issue under group 1's matrix work and the
// other way round, in ONE instruction stream.  (A SIMD overlaps a wave's VALU with that wave's own MFMAs, not with another
// wave's: tools/probe_overlap2, profiles/r02_attention_counters.md -- two waves per SIMD add their VALU and MFMA times.)
// One workgroup per CU, one wave per SIMD, ~330 registers.  Same arithmetic per row as G = 1: bit-identical results.
template <int NW, int G>
__global__ __launch_bounds__(64 * NW, G == 1 ? 2 : 1) void ea_attn_dma_kernel(AttnParams p) {
  constexpr int D = 64, NKS = D / 16, NDT = D / 32;
  constexpr int ROWB = 128;                          // LDS bytes per K / V row (unpadded: the DMA writes lane-linear)
  constexpr int STAGE = ATT_BK * ROWB;               // 8 KiB per tile and operand
  constexpr int RING = 3;
  constexpr int NINS = STAGE / 1024 / NW;            // DMA instructions per wave, tile and operand
  static_assert(STAGE % (1024 * NW) == 0, "whole 1-KiB pieces per wave");
  constexpr float LOG2E = 1.4426950408889634f;
  EA_SMEM(smem);
  char* const kring = smem;
  char* const vring = smem + RING * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef EA_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int half = lane >> 5, l31 = lane & 31;
  int qblk, bh;
  ea_attn_block(qblk, bh);
  const int b = bh / p.H, h = bh % p.H;
  int q_row[G];
  bool q_ok[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    q_row[g] = qblk * (32 * NW * G) + (wave * G + g) * 32 + l31;
    q_ok[g] = q_row[g] < p.Nq;
  }
  const f16* qp = p.q + b * p.q_sb + (long long)h * D;
  const f16* kp = p.k + b * p.k_sb + (long long)h * D;
  const f16* vp = p.v + b * p.v_sb + (long long)h * D;

  f16x8 qf[G][NKS];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int s = 0; s < NKS; ++s) qf[g][s] = ea_ld8(qp + (long long)(q_ok[g] ? q_row[g] : p.Nq - 1) * p.q_sn + 16 * s + 8 * half);

  f32x16 oacc[G][NDT];
  float m_run[G], l_run[G], m_use[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int e = 0; e < NDT; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[g][e][r] = 0.0f;
    m_run[g] = -INFINITY; l_run[g] = 0.0f; m_use[g] = 0.0f;
  }
  const float sc2 = p.scale * LOG2E;
  const int nkt = (p.Nk + ATT_BK - 1) / ATT_BK, nfull = p.Nk / ATT_BK;

  // staging: piece g = wave * NINS + i covers rows [8g, 8g + 8); lane -> row 8g + (lane >> 3), LDS chunk slot lane & 7,
  // which holds the row's global chunk slot ^ swizzle(row)
  const AttnRsrc rk = attn_make_rsrc(kp, (unsigned)(((long long)(p.Nk - 1) * p.k_sn + D) * 2));
  const AttnRsrc rv = attn_make_rsrc(vp, (unsigned)(((long long)(p.Nk - 1) * p.v_sn + D) * 2));
  unsigned koff[NINS], voff[NINS];
  int srow[NINS];
#pragma unroll
  for (int i = 0; i < NINS; ++i) {
    const int row = (wave * NINS + i) * 8 + (lane >> 3), slot = lane & 7;
    srow[i] = row;
    koff[i] = (unsigned)(row * p.k_sn + ((slot ^ ((row >> 1) & 7)) * 8)) * 2u;
    voff[i] = (unsigned)(row * p.v_sn + ((slot ^ (((row >> 1) & 1) << 2)) * 8)) * 2u;
  }
  auto k_issue = [&](int t, int slot) {
    char* dst = kring + slot * STAGE + wave * NINS * 1024;
    const unsigned soff = (unsigned)((long long)t * ATT_BK * p.k_sn * 2);
    const int left = p.Nk - t * ATT_BK;
#pragma unroll
    for (int i = 0; i < NINS; ++i) attn_dma16(rk, (left >= ATT_BK || srow[i] < left) ? koff[i] : ATT_OOB, soff, dst + i * 1024);
  };
  auto v_issue = [&](int t, int slot) {
    char* dst = vring + slot * STAGE + wave * NINS * 1024;
    const unsigned soff = (unsigned)((long long)t * ATT_BK * p.v_sn * 2);
    const int left = p.Nk - t * ATT_BK;
#pragma unroll
    for (int i = 0; i < NINS; ++i) attn_dma16(rv, (left >= ATT_BK || srow[i] < left) ? voff[i] : ATT_OOB, soff, dst + i * 1024);
  };

  // fragment addresses inside a stage.  K: row 32u + l31 (u adds an immediate), chunk 2s + half; the swizzle term of a
  // row does not depend on u.  V^T: row 4 * half + ((lane & 15) >> 2) (+ 16 tu, + 8: immediates), chunk 4e + cl.
  int kfo[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) kfo[s] = l31 * ROWB + (((2 * s + half) ^ ((l31 >> 1) & 7)) << 4);
  const int vrow0 = 4 * half + ((lane & 15) >> 2);
  const int vcl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  int vfo[NDT];
#pragma unroll
  for (int e = 0; e < NDT; ++e) vfo[e] = vrow0 * ROWB + (((4 * e + vcl) ^ (((vrow0 >> 1) & 1) << 2)) << 4) + 8 * (lane & 1);

  auto qk = [&](int slot, f32x16 (&sc)[G][2]) {
    const char* ks = kring + slot * STAGE;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[g][u][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(ks + kfo[s] + 32 * u * ROWB);
#pragma unroll
        for (int g = 0; g < G; ++g) sc[g][u] = ea_mfma_32x32x16(a, qf[g][s], sc[g][u]);
      }
  };
  auto advance_max1 = [&](int t, f32x16 (&sc)[2], float& m_run, float& l_run, float& m_use, f32x16 (&oacc)[NDT], auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASKED) {
          const int key = t * ATT_BK + 32 * u + ea_mfma_row(r, lane);
          if (key >= p.Nk) sc[u][r] = -INFINITY;
        }
        mx = fmaxf(mx, sc[u][r]);
      }
    mx *= sc2;
    mx = fmaxf(mx, ea_shfl_xor(mx, 32));
    if (ea_wave_any(mx > m_run + ATT_DEFER)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == -INFINITY) ? 1.0f : ea_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int e = 0; e < NDT; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[e][r] *= alpha;
#ifndef EA_EMU
      asm volatile("" ::: "memory");
#endif
    }
    m_use = (m_run == -INFINITY) ? 0.0f : m_run;
  };
  auto advance_max = [&](int t, f32x16 (&sc)[G][2], auto masked_tag) {
#pragma unroll
    for (int g = 0; g < G; ++g) advance_max1(t, sc[g], m_run[g], l_run[g], m_use[g], oacc[g], masked_tag);
  };

  // iteration j: requests K(j+3) / V(j+2), reads K(j+1) (ring slot s1) and V(j) (slot s0); s0 = j % 3, s1 = (j+1) % 3,
  // s2 = (j+2) % 3.  K(j+3) goes to K slot s0 (K(j), last read in iteration j-1), V(j+2) to V slot s2 (V(j-1), ditto).
  auto iter = [&](int j, int s0, int s1, int s2, f32x16 (&cur)[G][2], f32x16 (&nxt)[G][2], auto next_tag) {
    constexpr int NEXT = decltype(next_tag)::value;
    const bool k_req = j + 3 < nkt, v_req = j + 2 < nkt;
    if (k_req) k_issue(j + 3, s0);
    if (v_req) v_issue(j + 2, s2);
    const char* vb = vring + s0 * STAGE;
    f16x4 vlo[NDT][4], vhi[NDT][4];
    ea_static_for<NDT>([&](auto e_tag) {
      constexpr int e = decltype(e_tag)::value;
      const char* ve = vb + vfo[e];
      ea_static_for<4>([&](auto tu_tag) {
        constexpr int tu = decltype(tu_tag)::value;
        vlo[e][tu] = ea_lds_read_tr16<(16 * tu) * ROWB>(ve);
        vhi[e][tu] = ea_lds_read_tr16<(16 * tu + 8) * ROWB>(ve);
      });
    });
    if (NEXT) qk(s1, nxt);
    // per group: probabilities (VALU), then its PV MFMAs -- group 1's exponentials issue under group 0's PV
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float psum = 0.0f;
      f16x8 pb[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f16x2 pp;
          pp[0] = (f16)ea_exp2(fmaf(cur[g][u][r], sc2, -m_use[g]));
          pp[1] = (f16)ea_exp2(fmaf(cur[g][u][r + 1], sc2, -m_use[g]));
          psum = ea_dot2_ones(pp, psum);
          pb[u][r >> 3][r & 7] = pp[0];
          pb[u][r >> 3][(r & 7) + 1] = pp[1];
        }
      l_run[g] += psum;
      if (g == 0) ea_lds_tr_wait();
#pragma unroll
      for (int tu = 0; tu < 4; ++tu)
#pragma unroll
        for (int e = 0; e < NDT; ++e) {
          f16x8 a;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { a[jj] = vlo[e][tu][jj]; a[4 + jj] = vhi[e][tu][jj]; }
          oacc[g][e] = ea_mfma_32x32x16(a, pb[tu >> 1][tu & 1], oacc[g][e]);
        }
    }
    if (NEXT) {
      advance_max(j + 1, nxt, std::integral_constant<bool, NEXT == 2>{});
      // the next iteration reads K(j+2) and V(j+1), requested one iteration ago; this iteration's requests stay in flight
      if (k_req) attn_wait_dma<2 * NINS>();
      else if (v_req) attn_wait_dma<NINS>();
      else attn_wait_dma<0>();
      attn_barrier();
    }
  };

  f32x16 sA[G][2], sB[G][2];
  k_issue(0, 0);
  v_issue(0, 0);
  if (nkt > 1) k_issue(1, 1);
  if (nkt > 2) k_issue(2, 2);
  if (nkt > 1) v_issue(1, 1);
  attn_wait_dma<0>();
  attn_barrier();
  qk(0, sA);
  // Iteration 0 requests K(3) into K slot 0, which EVERY wave has just read here: all of them must be past these reads before
  // any of them issues that request.  (Round 4: without this barrier a wave that ran ahead could overwrite K(0) under a slower
  // wave's reads -- never seen alone, where the four waves leave the barrier above in step and the DMA takes longer than the
  // eight fragment reads, but a few launches in 10^4 once another stream's waves shared the SIMDs: the "results of the captured
  // loop change beside a busy second stream" of profiles/r04_pipelined_race.jsonl, found by tools/diag_kernel_race.py.)
  if (nkt > 3) attn_barrier();
  if (nfull == 0) advance_max(0, sA, std::true_type{});
  else advance_max(0, sA, std::false_type{});

  auto take = [&]() {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[g][u][r] = sB[g][u][r];
  };
  int j = 0, s0 = 0, s1 = 1, s2 = 2;
  auto rot = [&]() { const int t = s0; s0 = s1; s1 = s2; s2 = t; };
  const int nfn = nfull - 1;     // iterations whose next tile is a full one
  for (; j + 1 < nfn; j += 2) {
    iter(j, s0, s1, s2, sA, sB, std::integral_constant<int, 1>{});
    rot();
    iter(j + 1, s0, s1, s2, sB, sA, std::integral_constant<int, 1>{});
    rot();
  }
  if (j < nfn) {
    iter(j, s0, s1, s2, sA, sB, std::integral_constant<int, 1>{});
    rot();
    take();
    ++j;
  }
  if (nfull < nkt && nfull >= 1) {
    iter(j, s0, s1, s2, sA, sB, std::integral_constant<int, 2>{});
    rot();
    take();
    ++j;
  }
  iter(j, s0, s1, s2, sA, sB, std::integral_constant<int, 0>{});

#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float l_tot = l_run[g] + ea_shfl_xor(l_run[g], 32);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    if (q_ok[g]) {
      f16* op = p.o + b * p.o_sb + (long long)q_row[g] * p.o_sn + (long long)h * D;
#pragma unroll
      for (int e = 0; e < NDT; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f16x4 o4;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) o4[jj] = (f16)(oacc[g][e][4 * c + jj] * inv);
          *reinterpret_cast<f16x4*>(op + 32 * e + 8 * c + 4 * half) = o4;
        }
    }
  }
}

template <int D, int BIAS>
static int launch_attn(const AttnParams& p, void* stream) {
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int DV = (D + 31) / 32 * 32;
  constexpr int VROW = ((DV * 2) % 256 == 64 || (DV * 2) % 256 == 192) ? DV * 2 : DV * 2 + 64;
  const int smem = 2 * ATT_BK * ((DQK * 2 + 16) + VROW) + (BIAS == 1 ? ATT_BQ * (2 * p.S + 1) * 4 : 0);
  dim3 grid((p.Nq + ATT_BQ - 1) / ATT_BQ, p.B * p.H, 1);
  if constexpr (BIAS == 0 && D == 64) {
    // the pipelined loop pays a longer prologue: it wins from 4 key tiles up (self-attention), the in-order kernel
    // keeps the 77-token cross-attention (20.0 vs 22.1 us at Nq = 4096, 11.8 vs 14.2 us at Nq = 1024).  Its buffer
    // offsets are 32-bit: K / V of one (batch, head) must span < 2 GiB (always true for an fp16 projection output).
    // (NW = 8 -- 256 queries per workgroup, half the staging per wave -- measured +1..3 % where the workgroup count
    // divides the chip evenly and -11 % at the level-0 shape, 640 workgroups on 256 CUs: profiles/r02_attention_counters.md)
    const long long span = (long long)p.Nk * (p.k_sn > p.v_sn ? p.k_sn : p.v_sn) * 2;
    if (p.Nk >= 4 * ATT_BK && span < (1ll << 31)) {
      const int dsmem = 2 * 3 * ATT_BK * 128;        // K ring + V ring, three 8-KiB stages each
#if (EA_ATTN_EXP & 2)
      // EXPERIMENT (side builds and the emulation build only): two query groups per wave (256 queries per workgroup, one
      // workgroup per CU) once the launch has a workgroup for every CU in that form.  Measured on the MI355X against the
      // shipped form in the same call (profiles/r03_attention_two_groups_per_wave.jsonl): 531 vs 772 TF/s at the level-0
      // shape (B8 H5 4096^2), 352 vs 531 at 1024^2 -- with ONE wave per SIMD nothing covers the LDS and MFMA result
      // latencies between the dependent steps of a tile, and that costs more than sharing the fragments saves
      const long long wg2 = (long long)p.B * p.H * ((p.Nq + 255) / 256);
      if (wg2 >= ATT_G2_MIN_WG) {
        auto dfn2 = ea_attn_dma_kernel<4, 2>;
        ea_allow_big_lds(dfn2, dsmem);
        EA_LAUNCH(dfn2, dim3((p.Nq + 255) / 256, p.B * p.H, 1), dim3(256), dsmem, stream, p);
        return ea_launch_status();
      }
#endif
      auto dfn = ea_attn_dma_kernel<4, 1>;
      ea_allow_big_lds(dfn, dsmem);
      EA_LAUNCH(dfn, grid, dim3(256), dsmem, stream, p);
      return ea_launch_status();
    }
  }
  auto kfn = ea_attn_kernel<D, BIAS>;
  ea_allow_big_lds(kfn, smem);
  EA_LAUNCH(kfn, grid, dim3(256), smem, stream, p);
  return ea_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// SAM windowed attention, one workgroup per (window, head): the whole window (N = S*S <= 256 tokens, 14x14 = 196 in
// SAM) lives in LDS, so there is ONE global-load latency per workgroup instead of one per key tile, no barrier in the
// key loop, and the decomposed relative-position bias is computed in the kernel (no bias tables through HBM):
//   attn[q][k] = scale * q.k + q.Rh[qh - kh + S - 1] + q.Rw[qw - kw + S - 1]        (unscaled q in the bias terms)
// 8 waves; wave w owns the 32 queries [32w, 32w + 32).  Per wave: T^T = R Q^T (two 32-row MFMA tiles: the 2S - 1 <= 31
// relative offsets of each axis) is scattered into a private LDS table T[q][kh | kw]; then the flash loop of
// ea_attn_kernel over 32-key tiles read straight from the resident K / V images.
struct WinParams {
  const f16* q; const f16* k; const f16* v; f16* o;
  const f16* rel_h; const f16* rel_w;   // [2S - 1][D]
  int nbh, H, S, N;
  long long q_sb, q_sn, k_sb, k_sn, v_sb, v_sn, o_sb, o_sn;
  float scale;
};

// The bias is folded into the score MFMA chain: bias[q][key] = T[q][kh(key)] + T[q][16 + kw(key)] = (E^T T^T)[key][q]
// with E the one-hot selector of a key's (kh, kw) -- so every K row in LDS carries 32 extra one-hot fp16 columns and
// the Q^T operand 32 extra rows (the wave's bias table T, pre-divided by the softmax scale), and
// S^T = [K | E^T] [Q ; T]^T comes out of NKS + 2 MFMAs per 32-key tile with no per-score integer or LDS work.
template <int D>
__global__ __launch_bounds__(512, 2) void ea_attn_window_kernel(WinParams p) {
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int NKS = DQK / 16;
  constexpr int NDT = (D + 31) / 32;
  constexpr int DV = NDT * 32;
  constexpr int KROW = DQK * 2 + 64 + 16;    // K row: DQK channels | 32 one-hot bias selectors | pad (odd multiple of 16 B)
  static_assert((KROW / 16) % 2 == 1, "K row stride must be an odd number of 16-byte slots");
  constexpr int RROW = DQK * 2 + 16;         // rel-pos table rows
  constexpr int VROW = ((DV * 2) % 256 == 64 || (DV * 2) % 256 == 192) ? DV * 2 : DV * 2 + 64;
  constexpr int KCH = DQK / 8 + 4, VCH = DV / 8, RCH = DQK / 8;
  constexpr int NPMAX = 256;                 // padded token count limit
  constexpr int TLD = 33;                    // fp32 words per query row of the bias table: [kh 0..15 | kw 0..15] + pad
  constexpr float LOG2E = 1.4426950408889634f;
  EA_SMEM(smem);
  const int NP = (p.N + 31) & ~31;
  char* ks = smem;
  char* vs = ks + NPMAX * KROW;
  char* rs = vs + NPMAX * VROW;                                  // rel_h rows 0..31, rel_w rows 32..63
  float* tb = reinterpret_cast<float*>(rs + 64 * RROW);          // [8 waves][32][TLD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int bh = blockIdx.x;
  const int b = bh / p.H, h = bh % p.H;
  const f16* qp = p.q + b * p.q_sb + (long long)h * D;
  const f16* kp = p.k + b * p.k_sb + (long long)h * D;
  const f16* vp = p.v + b * p.v_sb + (long long)h * D;
  const int R = 2 * p.S - 1;

  // ---- stage K (+ selectors), V (zero beyond N rows / D columns) and the two relative-position tables.  All global
  // loads of a thread are issued before its first LDS store (fully unrolled): one memory latency per workgroup.
  constexpr int KIT = (NPMAX * KCH + 511) / 512, VIT = (NPMAX * VCH + 511) / 512, RIT = (64 * RCH + 511) / 512;
  f16x8 kb[KIT], vb[VIT], rb[RIT];
#pragma unroll
  for (int i = 0; i < KIT; ++i) {
    const int c = tid + 512 * i;
    const int row = c / KCH, cc = c - row * KCH;
    f16x8 x = ea_zero8();
    if (c < NP * KCH && row < p.N) {
      if (cc < DQK / 8) {
        if (cc * 8 < D) x = ea_ld8(kp + (long long)row * p.k_sn + cc * 8);
      } else {
        const int kh = row / p.S, kw = row - kh * p.S;
        const int j0 = (cc - DQK / 8) * 8;        // selector columns j0 .. j0 + 7 of [kh 0..15 | kw 0..15]
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (j0 + j == kh || j0 + j == 16 + kw) ? (f16)1.0f : (f16)0.0f;
      }
    }
    kb[i] = x;
  }
#pragma unroll
  for (int i = 0; i < VIT; ++i) {
    const int c = tid + 512 * i;
    const int row = c / VCH, d0 = (c - row * VCH) * 8;
    vb[i] = (c < NP * VCH && row < p.N && d0 < D) ? ea_ld8(vp + (long long)row * p.v_sn + d0) : ea_zero8();
  }
#pragma unroll
  for (int i = 0; i < RIT; ++i) {
    const int c = tid + 512 * i;
    const int row = c / RCH, d0 = (c - row * RCH) * 8;
    const int rr = row & 31;
    rb[i] = (c < 64 * RCH && rr < R && d0 < D) ? ea_ld8((row < 32 ? p.rel_h : p.rel_w) + (long long)rr * D + d0) : ea_zero8();
  }

  const int q_row = wave * 32 + l31;
  const bool q_ok = q_row < p.N;
  const int q_ld = q_ok ? q_row : p.N - 1;
  f16x8 qf[NKS + 2];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const int d0 = 16 * s + 8 * half;
    if (wave * 32 < p.N && d0 < D) qf[s] = ea_ld8(qp + (long long)q_ld * p.q_sn + d0);
    else qf[s] = ea_zero8();
  }
#pragma unroll
  for (int i = 0; i < KIT; ++i) {
    const int c = tid + 512 * i;
    const int row = c / KCH, cc = c - row * KCH;
    if (c < NP * KCH) *reinterpret_cast<f16x8*>(ks + row * KROW + cc * 16) = kb[i];
  }
#pragma unroll
  for (int i = 0; i < VIT; ++i) {
    const int c = tid + 512 * i;
    const int row = c / VCH, d0 = (c - row * VCH) * 8;
    if (c < NP * VCH) *reinterpret_cast<f16x8*>(vs + row * VROW + d0 * 2) = vb[i];
  }
#pragma unroll
  for (int i = 0; i < RIT; ++i) {
    const int c = tid + 512 * i;
    const int row = c / RCH, d0 = (c - row * RCH) * 8;
    if (c < 64 * RCH) *reinterpret_cast<f16x8*>(rs + row * RROW + d0 * 2) = rb[i];
  }
  __syncthreads();
  if (wave * 32 >= p.N) return;     // no workgroup barrier below

  // ---- bias table of this wave's 32 queries: T[q][kh] = q.Rh[qh - kh + S - 1], T[q][16 + kw] likewise, stored
  // divided by the softmax scale (the score accumulator is scaled as a whole afterwards); unused entries stay 0
  float* tq = tb + (wave * 32 + l31) * TLD;
  const float sc2 = p.scale * LOG2E;
  const float tscale = 1.0f / p.scale;
#pragma unroll
  for (int j = 0; j < 16; ++j) tq[16 * half + j] = 0.0f;
  const int qh = q_ld / p.S, qw = q_ld - qh * p.S;
#pragma unroll
  for (int ax = 0; ax < 2; ++ax) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(rs + (32 * ax + l31) * RROW + (16 * s + 8 * half) * 2);
      acc = ea_mfma_32x32x16(a, qf[s], acc);
    }
    const int qa = ax == 0 ? qh : qw;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rho = ea_mfma_row(r, lane);          // relative offset index held in register r
      const int kk = qa + p.S - 1 - rho;             // key coordinate this offset belongs to
      if (kk >= 0 && kk < p.S) tq[16 * ax + kk] = acc[r] * tscale;
    }
  }
  ea_wave_lds_sync_();
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[NKS + s2][j] = (f16)tq[16 * s2 + 8 * half + j];

  f32x16 oacc[NDT];
#pragma unroll
  for (int e = 0; e < NDT; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[e][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const int vt_off = (4 * half + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int nkt = NP / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NKS + 2; ++s) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(ks + (32 * kt + l31) * KROW + (16 * s + 8 * half) * 2);
      sacc = ea_mfma_32x32x16(a, qf[s], sacc);
    }
    float mx = -INFINITY;
    if (32 * kt + 32 > p.N) {        // ragged last tile (wave-uniform branch): keys >= N are -inf
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + ea_mfma_row(r, lane);
        const float sv = key < p.N ? sacc[r] * sc2 : -INFINITY;
        sacc[r] = sv;
        mx = fmaxf(mx, sv);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] *= sc2;
        mx = fmaxf(mx, sacc[r]);
      }
    }
    mx = fmaxf(mx, ea_shfl_xor(mx, 32));
    if (ea_wave_any(mx > m_run + ATT_DEFER)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == -INFINITY) ? 1.0f : ea_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int e = 0; e < NDT; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[e][r] *= alpha;
    }
    const float m_use = (m_run == -INFINITY) ? 0.0f : m_run;
    float psum = 0.0f;
    f16x8 pb[2];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f16x2 pp;
      pp[0] = (f16)ea_exp2(sacc[r] - m_use);
      pp[1] = (f16)ea_exp2(sacc[r + 1] - m_use);
      psum = ea_dot2_ones(pp, psum);
      pb[r >> 3][r & 7] = pp[0];
      pb[r >> 3][(r & 7) + 1] = pp[1];
    }
    l_run += psum;
    const char* vbase = vs + (32 * kt) * VROW + vt_off;
    ea_static_for<NDT>([&](auto e_tag) {
      constexpr int e = decltype(e_tag)::value;
      f16x4 vlo[2], vhi[2];
      ea_static_for<2>([&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        vlo[u] = ea_lds_read_tr16<(16 * u) * VROW + 64 * e>(vbase);
        vhi[u] = ea_lds_read_tr16<(16 * u + 8) * VROW + 64 * e>(vbase);
      });
      ea_lds_tr_wait();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = vlo[u][j]; a[4 + j] = vhi[u][j]; }
        oacc[e] = ea_mfma_32x32x16(a, pb[u], oacc[e]);
      }
    });
  }
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

template <int D>
static int launch_attn_window(const WinParams& p, void* stream) {
  constexpr int DQK = (D + 15) / 16 * 16;
  constexpr int DV = (D + 31) / 32 * 32;
  constexpr int KROW = DQK * 2 + 64 + 16, RROW = DQK * 2 + 16;
  constexpr int VROW = ((DV * 2) % 256 == 64 || (DV * 2) % 256 == 192) ? DV * 2 : DV * 2 + 64;
  const int smem = 256 * KROW + 256 * VROW + 64 * RROW + 8 * 32 * 33 * 4;
  auto kfn = ea_attn_window_kernel<D>;
  ea_allow_big_lds(kfn, smem);
  EA_LAUNCH(kfn, dim3(p.nbh), dim3(512), smem, stream, p);
  return ea_launch_status();
}

// rel-pos tables: one thread per (bh, q, k<2S): dot over D channels.
struct RelposParams {
  const f16* q; int B, H, S, D;
  long long q_sb, q_sn;
  const f16* rel_h; const f16* rel_w;
  float* bias_h; float* bias_w;
};

// grid = (ceil(N * 2S / 256), B*H): 32-bit index math only (the first version's 64-bit div/mod per output dominated its
// run time); the 2S outputs of one query are consecutive threads, so the query row is a broadcast load and the table rows
// (2S-1 x D fp16, L1-resident) are read back to back.
__global__ __launch_bounds__(256) void ea_relpos_kernel(RelposParams p) {
  const int N = p.S * p.S;
  const int twoS = 2 * p.S;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * twoS) return;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int qi = idx / twoS, kk = idx - qi * twoS;
  const int qh = qi / p.S, qw = qi - qh * p.S;
  const bool is_w = kk >= p.S;
  const int kpos = is_w ? kk - p.S : kk;
  const int rel = (is_w ? qw : qh) - kpos + p.S - 1;
  const f16* qv = p.q + b * p.q_sb + (long long)qi * p.q_sn + (long long)h * p.D;
  const f16* rv = (is_w ? p.rel_w : p.rel_h) + rel * p.D;
  float acc = 0.0f;
  for (int c = 0; c < p.D; c += 8) {
    f16x8 a = ea_ld8(qv + c), r8 = ea_ld8(rv + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)r8[j];
  }
  float* dst = (is_w ? p.bias_w : p.bias_h) + ((long long)bh * N + qi) * p.S + kpos;
  *dst = acc;
}

// The same tables on the matrix pipe for the 64 x 64 token grid (SAM's global-attention blocks).  For a fixed query row
// qh the outputs bias_h[(qh, qw)][kh] = q . Rh[qh - kh + 63] are a 64 x 64 x D product of that row's 64 queries with
// 64 consecutive (descending) rows of the table; likewise bias_w for a fixed query column.  One workgroup per (query row
// or column, axis, batch*head), one 32 x 32 output tile per wave, operands straight from global memory in MFMA fragment
// order (no LDS); computed as out^T = R Q^T so a lane's four accumulators per register quad are four consecutive key
// positions of ONE query: 16-byte stores into the [bh][q][S] tables.  (The one-thread-per-output kernel above took
// 482 us per layer at ViT-H size: 5.4 GFLOP of scalar dot products for 134 MB of tables.)
template <int D>
__global__ __launch_bounds__(256) void ea_relpos_mfma_kernel(RelposParams p) {
  constexpr int S = 64, NKS = D / 16;
  static_assert(D % 16 == 0, "head dim");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int fix = blockIdx.x;              // qh (axis h) / qw (axis w)
  const bool is_w = blockIdx.y != 0;
  const int bh = blockIdx.z;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tb = wave & 1, qb = wave >> 1;
  const int qi = is_w ? (32 * qb + l31) * S + fix : fix * S + 32 * qb + l31;     // this lane's query (B operand column)
  const f16* qv = p.q + b * p.q_sb + (long long)qi * p.q_sn + (long long)h * D;
  const f16* rv = (is_w ? p.rel_w : p.rel_h) + (fix - (32 * tb + l31) + S - 1) * D;   // this lane's table row (A operand row)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const f16x8 a = ea_ld8(rv + 16 * s + 8 * half);
    const f16x8 q8 = ea_ld8(qv + 16 * s + 8 * half);
    acc = ea_mfma_32x32x16(a, q8, acc);
  }
  float* dst = (is_w ? p.bias_w : p.bias_h) + ((long long)bh * (S * S) + qi) * S + 32 * tb + 4 * half;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(dst + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
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
  if (S > 256) return EA_ERR_BAD_SHAPE;          // N * 2S must fit 32 bits
  if (S == 64 && (D == 64 || D == 80) && (((uintptr_t)bias_h | (uintptr_t)bias_w) & 15) == 0 && (((uintptr_t)rel_h | (uintptr_t)rel_w | (uintptr_t)q) & 15) == 0) {
    auto mfn = D == 64 ? ea_relpos_mfma_kernel<64> : ea_relpos_mfma_kernel<80>;
    EA_LAUNCH(mfn, dim3(64, 2, (unsigned)(B * H)), dim3(256), 0, stream, p);
    return ea_launch_status();
  }
  const int per_bh = S * S * 2 * S;
  auto kfn = ea_relpos_kernel;
  EA_LAUNCH(kfn, dim3((unsigned)((per_bh + 255) / 256), (unsigned)(B * H)), dim3(256), 0, stream, p);
  return ea_launch_status();
}

extern "C" int ea_sam_window_attn_f16(const void* q, const void* k, const void* v, void* out, int nWin, int heads, int S,
                                      int D, long long q_sb, long long q_sn, long long k_sb, long long k_sn,
                                      long long v_sb, long long v_sn, long long o_sb, long long o_sn, float scale,
                                      const void* rel_h, const void* rel_w, void* stream) {
  if (!q || !k || !v || !out || !rel_h || !rel_w) return EA_ERR_BAD_ARG;
  if (nWin <= 0 || heads <= 0 || S <= 0 || S > 16) return EA_ERR_BAD_SHAPE;
  if ((q_sn & 7) || (k_sn & 7) || (v_sn & 7) || (o_sn & 3) || (q_sb & 7) || (k_sb & 7) || (v_sb & 7) || (o_sb & 3))
    return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 7) ||
      ((uintptr_t)rel_h & 15) || ((uintptr_t)rel_w & 15))
    return EA_ERR_BAD_ARG;
  WinParams p;
  p.q = (const f16*)q; p.k = (const f16*)k; p.v = (const f16*)v; p.o = (f16*)out;
  p.rel_h = (const f16*)rel_h; p.rel_w = (const f16*)rel_w;
  p.nbh = nWin * heads; p.H = heads; p.S = S; p.N = S * S;
  p.q_sb = q_sb; p.q_sn = q_sn; p.k_sb = k_sb; p.k_sn = k_sn; p.v_sb = v_sb; p.v_sn = v_sn; p.o_sb = o_sb; p.o_sn = o_sn;
  p.scale = scale;
  switch (D) {
    case 64: return launch_attn_window<64>(p, stream);
    case 80: return launch_attn_window<80>(p, stream);
    default: return EA_ERR_UNSUPPORTED;
  }
}
