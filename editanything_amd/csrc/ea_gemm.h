// ea_gemm.h -- the one MFMA contraction kernel of the hot path.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )        fp16 operands, fp32 accumulate
//
// A is either a dense row-major matrix (Linear layers; reference call sites
// ldm/modules/attention.py:54,152-160,316-339) or the *implicit im2col* of an
// NHWC fp16 activation for a 3x3 / 1x1 convolution (ResBlock convs
// ldm/modules/diffusionmodules/openaimodel.py:200-231, Up/Downsample :108-152,
// ControlNet zero-convs cldm/cldm.py:281-282, VAE convs
// ldm/modules/diffusionmodules/model.py:68-149), optionally the channel-concat
// of two tensors (decoder skip connections cldm/cldm.py:38-41, never
// materialised) and optionally through a fused nearest-2x upsample.
//
// MI355X design: 256-thread workgroups (4 waves, one per SIMD), each wave owns a
// 64x64 accumulator = 2x2 v_mfma_f32_32x32x16_f16 tiles (64 acc VGPRs).  K is
// walked in 64-wide tiles, register-staged (16-B global loads issued a tile
// ahead) into a double-buffered, XOR-swizzled LDS image so the ds_read_b128
// fragment reads are bank-conflict free; one barrier per K-tile.  The epilogue
// goes through LDS so every global store / residual load is a coalesced 16-B
// access.  Small-M problems (8x8 / 16x16 UNet levels) use split-K into an fp32
// workspace plus a reduce+epilogue kernel.  Tile ids are remapped so that the
// tiles sharing an A row-panel land on the same XCD (private L2).
#pragma once
#include "ea_platform.h"

#define EA_BK 64

enum { EA_ACT_NONE = 0, EA_ACT_SILU = 1, EA_ACT_GELU = 2, EA_ACT_GEGLU = 3 };

struct EaEpilogue {
  const float* bias;      // [N] (or [M] if bias_per_row), nullable
  int bias_per_row;
  const float* rowvec;    // [groups][rowvec_ld], added per (m / rows_per_group, n), nullable
  int rowvec_ld;
  int rows_per_group;
  int act;                // EA_ACT_*
  float scale;            // applied after act
  const float* row_scale; // [M] per-row multiplier (ControlNetModel2 scale map), nullable
  const f16* residual;    // [M][ldr] fp16, nullable (added after scale)
  const float* residual32;// [M][ldr] fp32, nullable
  int ldr;
  void* out;              // fp16 or fp32 [M][ldc]
  int ldc;
  int out_f32;
  int geglu_block;        // GEGLU packing granule (64, 80 or 32)
  int M, N;               // logical output extent (N = N_gemm/2 for GEGLU)
  // LayerNorm folded into the contraction (register-direct epilogue only; see include/editanything_hip.h)
  const float* ln_stats;  // [ln_parts][M][2] partial (sum, sum of squares) of A's rows; NULL = no fold
  int ln_parts;
  const float* ln_colsum; // [N_gemm] row sums of the gamma-folded fp16 weight
  float ln_eps;
  float* row_stats_out;   // [parts][M][2] partial (sum, sum of squares) of the OUTPUT rows, one part per wave-column block
  float* gn_stats_out;    // [B][gn_hw / WTM][N / gn_cpg][2] GroupNorm partials of the OUTPUT (register-direct epilogue only)
  int gn_hw, gn_cpg;
  f16* gn_next_out;       // the consuming GroupNorm applied by the split-K reduction (ea_splitk_reduce_gn_kernel); NULL = off
  const float* gn_next_gamma;
  const float* gn_next_beta;
  float gn_next_eps;
  int gn_next_silu;
};

struct EaGemmParams {
  // ---- A operand
  const f16* a1;
  int c1;  // channels of source 1 (conv) / unused (dense)
  const f16* a2;
  int c2;
  const f16* a2_add;  // optional addend on source 2 (skip + control)
  int lda;            // dense: row stride in elements
  int conv;           // 0 dense, 1 implicit conv
  int ksize;          // 1 or 3
  int Hin, Win, Hout, Wout;
  int stride, pad, ups;
  // ---- W operand [N][ldw]
  const f16* w;
  int ldw;
  int M, N, K;
  // ---- batching / split-K
  int batch;
  long long strideA, strideW, strideC, strideR;
  int splits;
  int ktiles_per_split;
  int debug;       // bench-only ablation (EA_GEMM2_DEBUG): 1 = skip the epilogue, 2 = skip the K loop
  int epi_fast;    // host-checked: the launch qualifies for ea_gemm2's streamlined epilogue (see launch_fast)
  int raster_gm;   // ea_gemm3: tile rows per group of the grouped (L2-aware) tile order
  unsigned long long* prof;   // ea_gemm3 -DEA_G3_PROF builds only (tools/g3_prof): per-wave phase cycle totals; NULL in the product
  float* partial;  // [batch*splits][M][N] fp32 when splits > 1
  int acc_scale_kt;   // K-concatenated split operands (ea_epilogue.acc_scale_k / 64): before K tile `acc_scale_kt` is
  float acc_scale;    // accumulated, the accumulators are multiplied by acc_scale.  0 = off
  EaEpilogue epi;
};

// XCD-aware bijective remap of a linear workgroup id (guide T1): workgroups are
// dealt round-robin to the 8 XCDs, so give each XCD a contiguous chunk.
__device__ __forceinline__ int ea_xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8;
  const int xcd = bid % 8, idx = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// GROUPED tile order (L2-aware rasterisation): tile t of a tiles_m x tiles_n grid -> rows are taken `gm` at a time and a
// group is walked column-major, so a run of consecutive tiles covers ~gm row panels of A x run/gm column panels of W
// instead of one row panel x `run` column panels.  With a wide N (SAM's MLP / qkv Linears, the GEGLU projections) the
// row-major order streams the whole weight matrix through an XCD's 4-MiB L2 once per tile row: measured 5.8-8.5x the
// algorithmic HBM-side traffic, 5.9 TB/s on the fabric (profiles/r02_pmc_traffic_tap_major.json).
// `id` may carry a split-K slice in front (slice-major): id = slice * tiles + t.
__device__ __forceinline__ void ea_grouped_item(int id, int tiles_m, int tiles_n, int gm, int& tm, int& tn, int& split) {
  const int tiles = tiles_m * tiles_n;
  split = id / tiles;
  const int t = id - split * tiles;
  const int per_group = gm * tiles_n;
  const int grp = t / per_group;
  const int first_m = grp * gm;
  const int gsz = (tiles_m - first_m) < gm ? (tiles_m - first_m) : gm;
  const int r = t - grp * per_group;
  tn = r / gsz;
  tm = first_m + (r - tn * gsz);
}

// Store 8 consecutive outputs (row m, cols n..n+7) with the full epilogue.
__device__ __forceinline__ void ea_epilogue_store8(const EaEpilogue& e, long long cbase, long long rbase,
                                                   int m, int n, float v[8], bool apply_act_bias) {
  if (m >= e.M || n >= e.N) return;
  const int nvalid = (e.N - n) < 8 ? (e.N - n) : 8;
  if (apply_act_bias) {
    if (e.bias) {
      if (e.bias_per_row) {
        const float b = e.bias[m];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += b;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nvalid) v[j] += e.bias[n + j];
      }
    }
    if (e.rowvec) {
      const float* rv = e.rowvec + (long long)(m / e.rows_per_group) * e.rowvec_ld + n;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nvalid) v[j] += rv[j];
    }
    if (e.act == EA_ACT_SILU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ea_silu(v[j]);
    } else if (e.act == EA_ACT_GELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ea_gelu_erf(v[j]);
    }
  }
  float sc = e.scale;
  if (e.row_scale) sc *= e.row_scale[m];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] *= sc;
  const long long roff = rbase + (long long)m * e.ldr + n;
  const long long coff = cbase + (long long)m * e.ldc + n;
  const bool vec = (nvalid == 8) && ((e.ldc & 7) == 0) && ((coff & 7) == 0);
  if (e.residual) {
    if (vec && ((e.ldr & 7) == 0) && ((roff & 7) == 0)) {
      f16x8 r = ea_ld8(e.residual + roff);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
    } else {
      for (int j = 0; j < nvalid; ++j) v[j] += (float)e.residual[roff + j];
    }
  }
  if (e.residual32) {
    for (int j = 0; j < nvalid; ++j) v[j] += e.residual32[roff + j];
  }
  if (e.out_f32) {
    float* o = (float*)e.out + coff;
    if (vec) {
      f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<f32x4*>(o) = lo;
      *reinterpret_cast<f32x4*>(o + 4) = hi;
    } else {
      for (int j = 0; j < nvalid; ++j) o[j] = v[j];
    }
  } else {
    f16* o = (f16*)e.out + coff;
    if (vec) {
      f16x8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (f16)v[j];
      ea_st8(o, h);
    } else {
      for (int j = 0; j < nvalid; ++j) o[j] = (f16)v[j];
    }
  }
}

// WGM x WGN waves, each 64x64.  (2,2) -> 128x128 tile; (4,1) -> 256x64 tile.
template <int WGM, int WGN>
__global__ __launch_bounds__(256) void ea_gemm_kernel(EaGemmParams p) {
  constexpr int BM = WGM * 64, BN = WGN * 64;
  constexpr int A_VEC = BM / 32, B_VEC = BN / 32;  // 16-B vectors per thread per K-tile
  constexpr int STAGE_BYTES = (BM + BN) * EA_BK * 2;
  constexpr int EPI_LD = BN + 4;                   // fp32 words per staged row
  EA_SMEM(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = tiles_m * tiles_n;
  const int tile = ea_xcd_remap(blockIdx.x, ntile);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int bz = blockIdx.z;
  const int batch = bz / p.splits, split = bz % p.splits;

  const int nk_total = (p.K + EA_BK - 1) / EA_BK;
  const int kt_begin = split * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nk_total) kt_end = nk_total;

  const f16* a1 = p.a1 + batch * p.strideA;
  const f16* a2 = p.a2 ? p.a2 + batch * p.strideA : nullptr;
  const f16* a2_add = p.a2_add ? p.a2_add + batch * p.strideA : nullptr;
  const f16* wp = p.w + batch * p.strideW;

  // ---- per-thread load coordinates
  const int chunk = tid & 7;   // which 8-element (16-B) K chunk of the 64-wide tile
  const int lrow = tid >> 3;   // 0..31
  const int ctot = p.c1 + p.c2;

  // A rows
  int a_b[A_VEC], a_y[A_VEC], a_x[A_VEC];
  bool a_ok[A_VEC];
#pragma unroll
  for (int i = 0; i < A_VEC; ++i) {
    const int m = m0 + lrow + 32 * i;
    a_ok[i] = m < p.M;
    if (p.conv) {
      const int hw = p.Hout * p.Wout;
      const int mm = a_ok[i] ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      const int oy = rem / p.Wout;
      a_b[i] = b;
      a_y[i] = oy * p.stride - p.pad;
      a_x[i] = (rem - oy * p.Wout) * p.stride - p.pad;
    } else {
      a_b[i] = m;
      a_y[i] = 0;
      a_x[i] = 0;
    }
  }
  // current K position of this thread's chunk: k = kt*64 + chunk*8 -> (tap, cin)
  int k_cur = kt_begin * EA_BK + chunk * 8;
  int tap = 0, cin = k_cur;
  if (p.conv) {
    tap = k_cur / ctot;
    cin = k_cur - tap * ctot;
  }

  f16x8 ra[A_VEC], rb[B_VEC];

  auto load_tile = [&]() {
    const bool kok = k_cur < p.K;
    if (p.conv) {
      const int ky = (p.ksize == 3) ? tap / 3 : 0;
      const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
      const f16* src;
      const f16* src_add = nullptr;
      int cs, coff;
      if (cin < p.c1) {
        src = a1; cs = p.c1; coff = cin;
      } else {
        src = a2; cs = p.c2; coff = cin - p.c1; src_add = a2_add;
      }
      const int hlim = p.ups ? 2 * p.Hin : p.Hin;
      const int wlim = p.ups ? 2 * p.Win : p.Win;
#pragma unroll
      for (int i = 0; i < A_VEC; ++i) {
        int iy = a_y[i] + ky, ix = a_x[i] + kx;
        const bool ok = kok && a_ok[i] && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
        if (ok) {
          if (p.ups) { iy >>= 1; ix >>= 1; }
          const long long off = (((long long)a_b[i] * p.Hin + iy) * p.Win + ix) * cs + coff;
          f16x8 v = ea_ld8(src + off);
          if (src_add) v = v + ea_ld8(src_add + off);
          ra[i] = v;
        } else {
          ra[i] = ea_zero8();
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_VEC; ++i) {
        if (kok && a_ok[i]) ra[i] = ea_ld8(a1 + (long long)a_b[i] * p.lda + k_cur);
        else ra[i] = ea_zero8();
      }
    }
#pragma unroll
    for (int i = 0; i < B_VEC; ++i) {
      const int n = n0 + lrow + 32 * i;
      if (kok && n < p.N) rb[i] = ea_ld8(wp + (long long)n * p.ldw + k_cur);
      else rb[i] = ea_zero8();
    }
    // advance to the next K tile
    k_cur += EA_BK;
    if (p.conv) {
      cin += EA_BK;
      while (cin >= ctot) { cin -= ctot; ++tap; }
    }
  };

  auto store_tile = [&](int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + BM * EA_BK * 2;
#pragma unroll
    for (int i = 0; i < A_VEC; ++i) {
      const int r = lrow + 32 * i;
      *reinterpret_cast<f16x8*>(sa + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_VEC; ++i) {
      const int r = lrow + 32 * i;
      *reinterpret_cast<f16x8*>(sb + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = kt_end - kt_begin;
  if (nk > 0) {
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_tile();
      const char* sa = smem + cur * STAGE_BYTES;
      const char* sb = sa + BM * EA_BK * 2;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 2 * ks + (lane >> 5);
        f16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wm * 64 + i * 32 + (lane & 31);
          fa[i] = *reinterpret_cast<const f16x8*>(sa + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = wn * 64 + j * 32 + (lane & 31);
          fb[j] = *reinterpret_cast<const f16x8*>(sb + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = ea_mfma_32x32x16(fa[i], fb[j], acc[i][j]);
      }
      if (kt + 1 < nk) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // ------------------------------------------------------------- epilogue
  // Phase 1: accumulators -> LDS (fp32, padded rows).  Phase 2: each thread
  // takes 8 consecutive columns of one row -> coalesced 16-B global accesses.
  float* stg = reinterpret_cast<float*>(smem);
  const EaEpilogue& e = p.epi;
  const bool raw = p.splits > 1;
  const bool geglu = (!raw) && e.act == EA_ACT_GEGLU;
  const int col = lane & 31;
  if (geglu) {
    // value columns live in MFMA tile j=0, gate columns in j=1 (weights are packed that way).
    const int nv = n0 + wn * 64 + col;
    const float bv = (e.bias && nv < p.N) ? e.bias[nv] : 0.0f;
    const float bg = (e.bias && nv + 32 < p.N) ? e.bias[nv + 32] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + ea_mfma_row(r, lane);
        const float val = acc[i][0][r] + bv;
        const float gate = acc[i][1][r] + bg;
        stg[row * EPI_LD + wn * 32 + col] = val * ea_gelu_erf(gate);
      }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 64 + i * 32 + ea_mfma_row(r, lane);
          stg[row * EPI_LD + wn * 64 + j * 32 + col] = acc[i][j][r];
        }
  }
  __syncthreads();
  const int tile_cols = geglu ? BN / 2 : BN;
  const int ncol0 = geglu ? n0 / 2 : n0;
  constexpr int VEC_PER_ROW_MAX = BN / 8;
  const int vec_per_row = tile_cols / 8;
  (void)VEC_PER_ROW_MAX;
  for (int idx = tid; idx < BM * vec_per_row; idx += 256) {
    const int row = idx / vec_per_row;
    const int cv = (idx - row * vec_per_row) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = stg[row * EPI_LD + cv + j];
    const int m = m0 + row, n = ncol0 + cv;
    if (raw) {
      if (m < p.M && n < p.N) {
        float* dst = p.partial + ((long long)bz * p.M + m) * p.N + n;
        const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
        for (int j = 0; j < nvalid; ++j) dst[j] = v[j];
      }
    } else {
      ea_epilogue_store8(e, batch * p.strideC, batch * p.strideR, m, n, v, !geglu);
    }
  }
}

// Reduce split-K partials and run the epilogue.  One thread per 8 outputs.
__device__ __forceinline__ void ea_splitk_reduce_body(const EaGemmParams& p) {
  const long long vec_per_row = (p.N + 7) / 8;
  const long long total = (long long)p.batch * p.M * vec_per_row;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int batch = (int)(idx / (p.M * vec_per_row));
    const long long rem = idx - (long long)batch * p.M * vec_per_row;
    const int m = (int)(rem / vec_per_row);
    const int n = (int)(rem - (long long)m * vec_per_row) * 8;
    const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    const long long slab = (long long)p.M * p.N;
    const float* src = p.partial + ((long long)(batch * p.splits) * p.M + m) * p.N + n;
    // The common launch (plain fp16 output, column bias / embedding row vector / fp16 residual, everything 16-byte aligned):
    // EVERY load of the thread -- partials and epilogue operands -- is issued before the first use.  Through
    // ea_epilogue_store8 the bias, row-vector and residual loads each start after the previous one returned: four
    // dependent round trips to a cold L2 (11 us per launch for 10 MB of partials, 92 launches per evaluation).
    const EaEpilogue& e = p.epi;
    const long long coff = (long long)batch * p.strideC + (long long)m * e.ldc + n;
    const long long roff = (long long)batch * p.strideR + (long long)m * e.ldr + n;
    if ((p.N & 7) == 0 && p.splits <= 8 && !e.bias_per_row && !e.row_scale && !e.residual32 && !e.out_f32 &&
        (e.act == EA_ACT_NONE || e.act == EA_ACT_SILU) && (e.ldc & 7) == 0 && (coff & 7) == 0 &&
        (!e.residual || ((e.ldr & 7) == 0 && (roff & 7) == 0)) && (!e.rowvec || (e.rowvec_ld & 3) == 0) &&
        ((((uintptr_t)e.out) | ((uintptr_t)e.residual) | ((uintptr_t)e.bias) | ((uintptr_t)e.rowvec)) & 15) == 0) {
      f32x4 t[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < p.splits) {
          t[u][0] = *reinterpret_cast<const f32x4*>(src + u * slab);
          t[u][1] = *reinterpret_cast<const f32x4*>(src + u * slab + 4);
        }
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, r0 = b0, r1 = b0;
      f16x8 res;
#pragma unroll
      for (int j = 0; j < 8; ++j) res[j] = (f16)0.0f;
      if (e.bias) { b0 = *reinterpret_cast<const f32x4*>(e.bias + n); b1 = *reinterpret_cast<const f32x4*>(e.bias + n + 4); }
      if (e.rowvec) {
        const float* rv = e.rowvec + (long long)(m / e.rows_per_group) * e.rowvec_ld + n;
        r0 = *reinterpret_cast<const f32x4*>(rv); r1 = *reinterpret_cast<const f32x4*>(rv + 4);
      }
      if (e.residual) res = ea_ld8(e.residual + roff);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < p.splits) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += t[u][0][j]; v[4 + j] += t[u][1][j]; }
        }
      // same order of operations as ea_epilogue_store8: + bias, + row vector, activation, * scale, + residual, round
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
      if (e.act == EA_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ea_silu(v[j]);
      }
      f16x8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = v[j] * e.scale;
        if (e.residual) x += (float)res[j];
        h[j] = (f16)x;
      }
      ea_st8((f16*)e.out + coff, h);
      continue;
    }
    if ((p.N & 7) == 0) {
      // 16-byte loads, four splits in flight per lane before the first add (the adds keep split order: results do
      // not depend on the unroll)
      int s = 0;
      for (; s + 4 <= p.splits; s += 4) {
        f32x4 t[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          t[u][0] = *reinterpret_cast<const f32x4*>(src + (s + u) * slab);
          t[u][1] = *reinterpret_cast<const f32x4*>(src + (s + u) * slab + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += t[u][0][j]; v[4 + j] += t[u][1][j]; }
      }
      for (; s < p.splits; ++s) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src + s * slab);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(src + s * slab + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += lo[j]; v[4 + j] += hi[j]; }
      }
    } else {
      for (int s = 0; s < p.splits; ++s)
        for (int j = 0; j < nvalid; ++j) v[j] += src[s * slab + j];
    }
    ea_epilogue_store8(p.epi, batch * p.strideC, batch * p.strideR, m, n, v, true);
  }
}

__global__ __launch_bounds__(256) void ea_splitk_reduce_kernel(EaGemmParams p) { ea_splitk_reduce_body(p); }
// the reductions of a twin launch (ea_gemm2_pair_kernel) as one grid: blockIdx.y = problem
__global__ __launch_bounds__(256) void ea_splitk_reduce_pair_kernel(EaGemmParams p0, EaGemmParams p1) {
  const EaGemmParams& p = blockIdx.y ? p1 : p0;
  ea_splitk_reduce_body(p);
}

// Split-K reduction + the GroupNorm (+ SiLU) that consumes the result, one workgroup per (sample, group): the slab
// [gn_hw rows x gn_cpg channels] is summed over the slices in 4-channel pieces (<= MAXQ per thread, kept in registers as the
// ROUNDED fp16 values the output holds), its statistics are reduced over the workgroup, and both the output and its
// normalised form are written.  Replaces reduce + statistics pass + normalise pass (3 launches, 2 extra trips over the
// tensor) at the levels where the contraction is split.  Same order of operations per element as ea_epilogue_store8.
// Footprint: a workgroup must not need a whole CU -- the ControlNet trunk and the UNet encoder run on two streams and their
// launches share CUs (that overlap is worth 19 % of the denoising loop); a 1024-thread x 128-register instantiation, which only
// fits an EMPTY CU, sent one graph replay in three into a 10 % slower mode (same kernel times, idle gaps).  Both forms take
// half a CU's registers: 1024 x 1 piece (<= 64 registers) and 512 x 5 pieces (<= 128).
constexpr int EA_RGN_MAXQ = 5;
template <int MAXQ>
__device__ __forceinline__ void ea_splitk_reduce_gn_body(const EaGemmParams& p) {
  constexpr int EA_RGN_THREADS = MAXQ == 1 ? 1024 : 512;
  EA_SMEM(smem);
  float (*red)[16] = reinterpret_cast<float (*)[16]>(smem);    // [2][16]
  const EaEpilogue& e = p.epi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int hw = e.gn_hw, cpg = e.gn_cpg, qpr = cpg >> 2;
  const int nq = hw * qpr;
  const long long slab = (long long)p.M * p.N;
  // ONE round: every load of the thread (its <= MAXQ pieces x all slices, bias, row vector, residual) is issued before the
  // first use -- a workgroup is on its own CU with nothing to overlap a second round trip with
  f32x4 acc[MAXQ], bv[MAXQ], rv[MAXQ];
  f16x4 res[MAXQ];
  int mrow[MAXQ], col[MAXQ];
  bool ok[MAXQ];
#pragma unroll
  for (int u = 0; u < MAXQ; ++u) {
    const int q = tid + EA_RGN_THREADS * u;
    ok[u] = q < nq;
    const int qq = ok[u] ? q : 0;
    const int row = qq / qpr;
    mrow[u] = b * hw + row;
    col[u] = g * cpg + (qq - row * qpr) * 4;
    acc[u] = *reinterpret_cast<const f32x4*>(p.partial + (long long)mrow[u] * p.N + col[u]);
    bv[u] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + col[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    rv[u] = e.rowvec ? *reinterpret_cast<const f32x4*>(e.rowvec + (long long)(mrow[u] / e.rows_per_group) * e.rowvec_ld + col[u])
                     : f32x4{0.f, 0.f, 0.f, 0.f};
    if (e.residual) res[u] = *reinterpret_cast<const f16x4*>(e.residual + (long long)mrow[u] * e.ldr + col[u]);
  }
  // the other slices, SCH at a time, all of a chunk's loads in flight together (the adds keep slice order)
  constexpr int SCH = MAXQ == 1 ? 7 : 2;
  for (int s0 = 1; s0 < p.splits; s0 += SCH) {
    f32x4 t[SCH][MAXQ];
#pragma unroll
    for (int c = 0; c < SCH; ++c)
#pragma unroll
      for (int u = 0; u < MAXQ; ++u)
        t[c][u] = (s0 + c < p.splits) ? *reinterpret_cast<const f32x4*>(p.partial + (s0 + c) * slab + (long long)mrow[u] * p.N + col[u])
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < SCH; ++c)
      if (s0 + c < p.splits) {
#pragma unroll
        for (int u = 0; u < MAXQ; ++u) acc[u] += t[c][u];
      }
  }
  // the consuming norm's scale / shift: requested now, used after the workgroup reduction
  f32x4 ga[MAXQ], be[MAXQ];
#pragma unroll
  for (int u = 0; u < MAXQ; ++u) {
    ga[u] = *reinterpret_cast<const f32x4*>(e.gn_next_gamma + col[u]);
    be[u] = *reinterpret_cast<const f32x4*>(e.gn_next_beta + col[u]);
  }
  float v[MAXQ][4];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int u = 0; u < MAXQ; ++u) {
    f16x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = acc[u][j] + bv[u][j];
      x += rv[u][j];
      if (e.act == EA_ACT_SILU) x = ea_silu(x);
      x *= e.scale;
      if (e.residual) x += (float)res[u][j];
      h[j] = (f16)x;
      const float r = ok[u] ? (float)h[j] : 0.0f;
      v[u][j] = r;
      s1 += r;
      s2 += r * r;
    }
    if (ok[u]) *reinterpret_cast<f16x4*>((f16*)e.out + (long long)mrow[u] * e.ldc + col[u]) = h;
  }
  s1 = ea_wave_sum(s1);
  s2 = ea_wave_sum(s2);
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  s1 = 0.0f; s2 = 0.0f;
#pragma unroll
  for (int w = 0; w < EA_RGN_THREADS / 64; ++w) { s1 += red[0][w]; s2 += red[1][w]; }
  const float inv_n = 1.0f / ((float)hw * (float)cpg);
  const float mean = s1 * inv_n;
  float var = s2 * inv_n - mean * mean;
  var = var > 0.0f ? var : 0.0f;
  const float rstd = 1.0f / sqrtf(var + e.gn_next_eps);
#pragma unroll
  for (int u = 0; u < MAXQ; ++u) {
    if (ok[u]) {
      f16x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = rstd * ga[u][j];                 // the normalise pass's form (ea_norm.hip gn_finish): x * a + sh
        float f = v[u][j] * a + (be[u][j] - mean * a);
        if (e.gn_next_silu) f = ea_silu(f);
        y[j] = (f16)f;
      }
      *reinterpret_cast<f16x4*>(e.gn_next_out + (long long)mrow[u] * e.ldc + col[u]) = y;
    }
  }
}

template <int MAXQ>
__global__ __launch_bounds__(MAXQ == 1 ? 1024 : 512) void ea_splitk_reduce_gn_kernel(EaGemmParams p) { ea_splitk_reduce_gn_body<MAXQ>(p); }
// twin form: blockIdx.z = problem
template <int MAXQ>
__global__ __launch_bounds__(MAXQ == 1 ? 1024 : 512) void ea_splitk_reduce_gn_pair_kernel(EaGemmParams p0, EaGemmParams p1) {
  const EaGemmParams& p = blockIdx.z ? p1 : p0;
  ea_splitk_reduce_gn_body<MAXQ>(p);
}

// Row statistics of a finished fp16 [M][ld] output, in the layout the register-direct epilogue writes
// ([parts][M][2] partial (sum, sum of squares)): the fallback producer for launches whose epilogue cannot emit them
// (split-K, generic kernel).  One wave per row; everything lands in part 0, the other parts are zeroed.
__global__ __launch_bounds__(256) void ea_row_stats_kernel(const f16* x, int ld, float* stats, int M, int C, int parts) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s1 = 0.0f, s2 = 0.0f;
  const f16* xr = x + (long long)row * ld;
  for (int c = lane * 8; c < C; c += 64 * 8) {
    if (c + 8 <= C) {
      const f16x8 v = ea_ld8(xr + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s1 += f; s2 += f * f; }
    } else {
      for (int j = 0; c + j < C; ++j) { const float f = (float)xr[c + j]; s1 += f; s2 += f * f; }
    }
  }
  s1 = ea_wave_sum(s1);
  s2 = ea_wave_sum(s2);
  if (lane == 0) *reinterpret_cast<f32x2*>(stats + (long long)row * 2) = f32x2{s1, s2};
  for (int pp = lane; pp < parts; pp += 64)   // every other part is zero, however many there are (N > 5120: more than 64)
    if (pp > 0) *reinterpret_cast<f32x2*>(stats + ((long long)pp * M + row) * 2) = f32x2{0.0f, 0.0f};
}
