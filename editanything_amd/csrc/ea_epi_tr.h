// ea_epi_tr.h -- the REGISTER-DIRECT epilogue of the MFMA contraction kernels (ea_gemm2.h, ea_gemm3.h).
//
// The kernels that use it run their MFMAs with swapped operands (D^T = W A^T), so a lane holds 4 CONSECUTIVE output
// columns of one output row per 16x16 tile:  acc[i][j][r] = C[rowbase + i*16 + lane % 16][colbase + j*16 + 4*(lane / 16) + r].
// One v_permlane16_swap per register pairs two tiles into 8 consecutive columns and the finished fp16 row segment is
// stored from registers with 16-byte stores -- no LDS slab, no scatter / gather passes (measured in round 2: the LDS-slab
// epilogues were 25 % of the contraction time of an evaluation).  What it fuses (reference call sites):
//   bias / per-sample time-embedding row vector / SiLU (ResBlock, openaimodel.py:254-274), GELU, scalar scale
//   (ControlNet conditioning scale, cldm/cldm.py:338), fp16 residual (skip / x + attn(x), attention.py:271-275),
//   GEGLU with 32-row packing (attention.py:54-56), the LayerNorm fold and the row / GroupNorm statistics of the
//   ROUNDED outputs for the next launch's norm, and the raw fp32 dump of a split-K slice.
// MI x NI: the 16x16 tiles this wave emits; WTN = NI * 16; TRX = 2 compiles the fold / statistics in.
// `scratch`: >= (waves) * 288 bytes of LDS no DMA targets (GroupNorm bins); SYNC = the caller needs a workgroup barrier
// before that LDS is free.
#pragma once
#include "ea_gemm.h"
#include "ea_prims.h"

// F32OUT (ea_gemm8.h only): compiles in the fp32-output form (p.epi_fast == 4: fp32 out, optional fp32 residual -- SAM's
// residual stream, sam.py / sam_exact.py) -- a lane's quad IS 16 bytes of one fp32 output row, no pairing needed.
template <int MI, int NI, int TRX, bool SYNC, bool F32OUT = false>
__device__ __forceinline__ void ea_tr_epilogue(const EaGemmParams& p, f32x4 (&acc)[MI][NI], const int rowbase, const int colbase,
                                               const int tile_m0, const int batch, const int bz, const float (&ln_mu)[MI],
                                               const float (&ln_rs)[MI], char* scratch, const int wave_slot) {
  constexpr int WTN = NI * 16;
  const EaEpilogue& e = p.epi;
  const int lane = ea_lane();
  static_assert(NI % 2 == 0 || MI % 2 == 0, "an odd column-tile count needs an even row-tile count to pair up");
  const int c16 = lane & 15, q4 = lane >> 4;
  if (p.splits > 1) {
    // split-K slice: the raw fp32 accumulators go to the [slice][M][N] partials straight from the registers -- a lane's
    // quad is 4 consecutive columns of one row = one 16-byte store (64 contiguous bytes per row per wave instruction);
    // ea_splitk_reduce_kernel sums the slices and applies the epilogue.  (The LDS-slab form of this dump cost the
    // 75 split launches of an evaluation 5-8 us each.)
    float* part = p.partial + (long long)bz * p.M * p.N;
#pragma unroll
    for (int ii = 0; ii < MI; ++ii) {
      const int m = rowbase + ii * 16 + c16;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = colbase + j * 16 + 4 * q4;
        if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(part + (long long)m * p.N + n) = acc[ii][j];
      }
    }
    
    return;
  }
  const int sel = q4 & 1, coff = 8 * (q4 >> 1);
    const long long cb0 = (long long)batch * p.strideC, rb0 = (long long)batch * p.strideR;
  f16* outp = (f16*)e.out + cb0;
  // Per-column terms.  Every global read of this epilogue is UNCONDITIONAL (clamped index, select afterwards) and sits
  // in a wave-uniform block per operand: written as `if (n < N) x = load` hipcc branches around each load and parks an
  // s_waitcnt vmcnt(0) behind it, which turned the NI row-vector reads + MI * NI / 2 residual reads of a tile into as
  // many serial L2 round trips (guide section 5 trap (c); the round-2 ISA had 6 of them back to back).
  f32x4 cb[NI], cs[NI];
  int ncl[NI];
  bool nok[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = colbase + j * 16 + 4 * q4;
    nok[j] = n < p.N;
    ncl[j] = nok[j] ? n : 0;
    cb[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    cs[j] = cb[j];
  }
  if (e.bias) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(e.bias + ncl[j]);
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[j][r] = nok[j] ? t[r] : 0.0f;
    }
  }
  if (TRX == 2 && e.ln_stats) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(e.ln_colsum + ncl[j]);
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[j][r] = nok[j] ? t[r] : 0.0f;
    }
  }
  auto ln_fold = [&](f32x4 x, int i, int j) {
    if (TRX == 2 && e.ln_stats) x = (x - ln_mu[i] * cs[j]) * ln_rs[i];
    return x;
  };
  // row-statistics accumulators (per row tile of this lane)
  float st1[MI], st2[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { st1[i] = 0.0f; st2[i] = 0.0f; }
  auto stats_flush = [&]() {
    if (!(TRX == 2 && e.row_stats_out)) return;
    const int part = colbase / WTN;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float a1 = st1[i], a2 = st2[i];
      a1 += ea_shfl_xor(a1, 16); a2 += ea_shfl_xor(a2, 16);
      a1 += ea_shfl_xor(a1, 32); a2 += ea_shfl_xor(a2, 32);
      const int m = rowbase + i * 16 + c16;
      if (q4 == 0 && m < p.M && colbase < e.N)
        *reinterpret_cast<f32x2*>(e.row_stats_out + ((long long)part * p.M + m) * 2) = f32x2{a1, a2};
    }
  };
  if (p.epi_fast == 3) {
    // ---- GEGLU, weight rows packed [16 value | 16 gate] per 32: column tile 2t holds the values, 2t+1 the gates of
    // outputs t*16 .. +15 in the SAME lanes and registers (attention.py:54-56: x, gate = proj(x).chunk(2); x * gelu(gate)).
    // Output tiles pair up along the columns like above: NI / 2 output tiles, (NI / 4) 16-byte vectors per row tile.
    // (only launched on the 128-wide tiles, NI = 4; the NI = 5 instantiations compile this branch but never take it)
    constexpr int OP = NI / 4;                 // output-tile pairs per row tile (NI = 4: one)
    const int obase = colbase >> 1;
#pragma unroll
    for (int ii = 0; ii < MI; ++ii)
#pragma unroll
      for (int op = 0; op < OP; ++op) {
        f32x4 o2[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int jv = 4 * op + 2 * h2, jg = jv + 1;
          const f32x4 v = ln_fold(acc[ii][jv < NI ? jv : 0], ii, jv < NI ? jv : 0) + cb[jv < NI ? jv : 0];
          f32x4 g = ln_fold(acc[ii][jg < NI ? jg : 0], ii, jg < NI ? jg : 0) + cb[jg < NI ? jg : 0];
#pragma unroll
          for (int r = 0; r < 4; ++r) g[r] = ea_gelu_erf(g[r]);
          o2[h2] = v * g * e.scale;
        }
        f32x4 a = o2[0], b = o2[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = a[r], y = b[r];
          ea_swap16(x, y);
          a[r] = x;
          b[r] = y;
        }
        const int m = rowbase + ii * 16 + c16, n = obase + (2 * op + sel) * 16 + coff;
        if (m < p.M && n < e.N) {
          f16x8 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) { h[r] = (f16)a[r]; h[4 + r] = (f16)b[r]; }
          ea_st8(outp + (long long)m * e.ldc + n, h);
        }
      }
    
    return;
  }
  const float* rvp = e.rowvec ? e.rowvec + (long long)(tile_m0 / e.rows_per_group) * e.rowvec_ld : nullptr;
  if constexpr (F32OUT) {
    if (p.epi_fast == 4) {
      // ---- fp32 output (+ fp32 residual): same per-output order as the general epilogue (bias / row vector, activation,
      // scale, residual), one 16-byte store per quad, 64 contiguous bytes per row per wave instruction.  Residual quads are
      // fetched one row tile ahead of their use.
      float* outf = (float*)e.out + cb0;
      const float* res32 = e.residual32 ? e.residual32 + rb0 : nullptr;
      if (rvp) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(rvp + ncl[j]);
#pragma unroll
          for (int r = 0; r < 4; ++r) cb[j][r] += nok[j] ? t[r] : 0.0f;
        }
      }
      f32x4 rq32[2][NI];
      auto res_load = [&](int ii, int slot) {
        int m = rowbase + ii * 16 + c16;
        m = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) rq32[slot][j] = *reinterpret_cast<const f32x4*>(res32 + (long long)m * e.ldr + ncl[j]);
      };
      if (res32) res_load(0, 0);
#pragma unroll
      for (int ii = 0; ii < MI; ++ii) {
        if (res32 && ii + 1 < MI) res_load(ii + 1, (ii + 1) & 1);
        const int m = rowbase + ii * 16 + c16;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x4 x = acc[ii][j] + cb[j];
          if (e.act == EA_ACT_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = ea_silu(x[r]);
          } else if (e.act == EA_ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = ea_gelu_erf(x[r]);
          }
          x = x * e.scale;
          if (res32) x += rq32[ii & 1][j];
          if (m < p.M && nok[j]) *reinterpret_cast<f32x4*>(outf + (long long)m * e.ldc + ncl[j]) = x;
        }
      }
      return;
    }
  }
  constexpr int JP = NI / 2;                    // column-tile pairs per row tile
  constexpr int IP = (NI & 1) ? MI / 2 : 0;     // row-tile pairs of the odd last column tile
  const f16* resp = e.residual ? e.residual + rb0 : nullptr;
  // where this lane's 8-column vectors go: column-pair vectors (ii, jp), then the odd tile's row-pair vectors (ip)
  int voff[MI * JP + IP + 1], roff[MI * JP + IP + 1];
#pragma unroll
  for (int ii = 0; ii < MI; ++ii)
#pragma unroll
    for (int jp = 0; jp < JP; ++jp) {
      const int m = rowbase + ii * 16 + c16, n = colbase + (2 * jp + sel) * 16 + coff;
      const bool ok = m < p.M && n < e.N;
      voff[ii * JP + jp] = ok ? m * e.ldc + n : -1;
      roff[ii * JP + jp] = ok ? m * e.ldr + n : -1;
    }
#pragma unroll
  for (int ip = 0; ip < IP; ++ip) {
    const int m = rowbase + (2 * ip + sel) * 16 + c16, n = colbase + (NI - 1) * 16 + coff;
    const bool ok = m < p.M && n < e.N;
    voff[MI * JP + ip] = ok ? m * e.ldc + n : -1;
    roff[MI * JP + ip] = ok ? m * e.ldr + n : -1;
  }
  // every global read of the epilogue is issued before the first use: residual vectors, then the per-column terms
  f16x8 rq[MI * JP + IP + 1];
  if (resp) {
#pragma unroll
    for (int v = 0; v < MI * JP + IP; ++v) rq[v] = ea_ld8(resp + (roff[v] >= 0 ? roff[v] : 0));   // used only where voff >= 0
  }
  if (rvp) {
    f32x4 rv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) rv[j] = *reinterpret_cast<const f32x4*>(rvp + ncl[j]);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[j][r] += nok[j] ? rv[j][r] : 0.0f;
  }
  // ---- GroupNorm statistics of the output (TRX == 2, e.gn_stats_out): per wave tile [MI * 16 rows of one sample] x [WTN
  // columns = whole groups], the (sum, sum of squares) of the ROUNDED fp16 outputs per group -- what the GroupNorm that
  // reads this tensor needs (openaimodel.py:254-274: conv -> GroupNorm32 -> SiLU -> conv), so that norm runs as one
  // streaming normalise pass with no statistics pass.  A lane's 8-column vector touches at most two groups (cpg >= 8):
  // column sums are kept per vector position over the row tiles, then binned into (first group, second group).
  const bool gn_on = TRX == 2 && e.gn_stats_out != nullptr;
  float gcs[8], gcq[8];          // column sums of the vector position being swept
  float gbin[3][4];              // per vector position (2 column pairs + the odd tile): s0, q0, s1, q1
  int gbin_g[3];                 // first group of each position (the second is + 1)
#pragma unroll
  for (int v = 0; v < 3; ++v) { gbin_g[v] = 0; gbin[v][0] = gbin[v][1] = gbin[v][2] = gbin[v][3] = 0.0f; }
  auto gn_reset = [&]() {
#pragma unroll
    for (int r = 0; r < 8; ++r) { gcs[r] = 0.0f; gcq[r] = 0.0f; }
  };
  auto gn_bin = [&](int slot, int ncol0) {     // ncol0: global column of the vector's first element
    const int g0 = ncol0 / e.gn_cpg;
    const int split = e.gn_cpg - (ncol0 - g0 * e.gn_cpg);   // columns of the vector that belong to group g0 (>= 8: all)
    float s0 = 0.0f, q0 = 0.0f, s1 = 0.0f, q1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool first = r < split;
      s0 += first ? gcs[r] : 0.0f; q0 += first ? gcq[r] : 0.0f;
      s1 += first ? 0.0f : gcs[r]; q1 += first ? 0.0f : gcq[r];
    }
    gbin_g[slot] = g0;
    gbin[slot][0] = s0; gbin[slot][1] = q0; gbin[slot][2] = s1; gbin[slot][3] = q1;
  };
  auto finish = [&](f32x4 x, int i, int j) {
    x = ln_fold(x, i, j) + cb[j];
    if (e.act == EA_ACT_SILU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = ea_silu(x[r]);
    } else if (e.act == EA_ACT_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = ea_gelu_erf(x[r]);
    }
    return x * e.scale;
  };
  // a, b: finished quads of the pair's first / second tile; i0 / i1: row tile of the result in even- / odd-q4 lanes
  auto emit = [&](f32x4 a, f32x4 b, int v, int i0, int i1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = a[r], y = b[r];
      ea_swap16(x, y);
      a[r] = x;
      b[r] = y;
    }
    const bool on = voff[v] >= 0;
    if (on && resp) {
      const f16x8 rr = rq[v];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] += (float)rr[r]; b[r] += (float)rr[4 + r]; }
    }
    f16x8 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) { h[r] = (f16)a[r]; h[4 + r] = (f16)b[r]; }
    if (TRX == 2 && e.row_stats_out) {
      // statistics of the ROUNDED fp16 outputs -- what the consumer's LayerNorm sees, and what the fallback producer
      // (ea_row_stats_kernel, split-K / generic launches) computes: both producers agree whatever path the planner picks
      float t1 = 0.0f, t2 = 0.0f;
      if (on) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float f = (float)h[r]; t1 += f; t2 += f * f; }
      }
      if (i0 == i1) { st1[i0] += t1; st2[i0] += t2; }
      else { st1[i0] += sel ? 0.0f : t1; st2[i0] += sel ? 0.0f : t2; st1[i1] += sel ? t1 : 0.0f; st2[i1] += sel ? t2 : 0.0f; }
    }
    if (!on) return;
    ea_st8(outp + voff[v], h);
    if (gn_on) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { const float f = (float)h[r]; gcs[r] += f; gcq[r] += f * f; }
    }
  };
  static_assert(JP <= 2, "GroupNorm statistics slots");
  // column pairs outermost: one vector position is swept over all its row tiles before the next (store order only)
#pragma unroll
  for (int jp = 0; jp < JP; ++jp) {
    gn_reset();
#pragma unroll
    for (int ii = 0; ii < MI; ++ii)
      emit(finish(acc[ii][2 * jp], ii, 2 * jp), finish(acc[ii][2 * jp + 1], ii, 2 * jp + 1),
           ii * JP + jp, ii, ii);
    if (gn_on) gn_bin(jp, colbase + (2 * jp + sel) * 16 + coff);
  }
  gn_reset();
#pragma unroll
  for (int ip = 0; ip < IP; ++ip)
    emit(finish(acc[2 * ip][NI - 1], 2 * ip, NI - 1),
         finish(acc[2 * ip + 1 < MI ? 2 * ip + 1 : 0][NI - 1], 2 * ip + 1 < MI ? 2 * ip + 1 : 0, NI - 1), MI * JP + ip, 2 * ip,
         2 * ip + 1 < MI ? 2 * ip + 1 : 0);
  if (gn_on && IP > 0) gn_bin(2, colbase + (NI - 1) * 16 + coff);
  stats_flush();
  if (gn_on) {
    // rows: the 16 lanes of a q4 group hold the same columns -> butterfly over lane bits 0..3; then the four q4
    // lanes' bins go through LDS and lane g < groups-per-wave-tile sums the ones of ITS group in a fixed order
    constexpr int NSLOT = (IP > 0) ? 3 : 2;
#pragma unroll
    for (int v = 0; v < NSLOT; ++v)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = gbin[v][k];
        x += ea_shfl_xor(x, 1); x += ea_shfl_xor(x, 2); x += ea_shfl_xor(x, 4); x += ea_shfl_xor(x, 8);
        gbin[v][k] = x;
      }
    if (SYNC) __syncthreads();   // every wave is past its last fragment read: the stage ring is free
    float* gl = reinterpret_cast<float*>(scratch) + wave_slot * (4 * NSLOT * 2 * 3);   // [q4][slot][half] x (group, s, q)
    if (c16 == 0) {
#pragma unroll
      for (int v = 0; v < NSLOT; ++v)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float* d = gl + ((q4 * NSLOT + v) * 2 + hf) * 3;
          d[0] = (float)(gbin_g[v] + hf);
          d[1] = gbin[v][2 * hf];
          d[2] = gbin[v][2 * hf + 1];
        }
    }
    ea_wave_lds_sync();
    const int ngw = WTN / e.gn_cpg;               // groups of this wave tile
    if (lane < ngw && colbase < e.N && rowbase < p.M) {
      const int g = colbase / e.gn_cpg + lane;
      float s1 = 0.0f, s2 = 0.0f;
      for (int t = 0; t < 4 * NSLOT * 2; ++t) {
        const bool mine = (int)gl[t * 3] == g;
        s1 += mine ? gl[t * 3 + 1] : 0.0f;
        s2 += mine ? gl[t * 3 + 2] : 0.0f;
      }
      const int b = rowbase / e.gn_hw, chunk = (rowbase - b * e.gn_hw) / (MI * 16);
      const int nchunk = e.gn_hw / (MI * 16), groups = e.N / e.gn_cpg;
      *reinterpret_cast<f32x2*>(e.gn_stats_out + (((long long)b * nchunk + chunk) * groups + g) * 2) = f32x2{s1, s2};
    }
  }
}
