// ea_gemm8.h -- the LARGE-TILE contraction kernel: 256 x 256 x 64 workgroup tiles, 8 waves (2 x 4), 128 x 64 per wave.
//
// Same contract as ea_gemm2.h (C = epilogue(A W^T); A dense or the implicit im2col of one / two NHWC sources; split-K raw
// dump; the register-direct epilogue of ea_epi_tr.h) for the launches whose M x N is large enough to fill the chip with
// 256 x 256 tiles: SAM's MLP / qkv Linears (segment_anything ImageEncoderViT blocks, reached from sam2image.py:117-120),
// the fp32-accurate encoder's K-concatenated split-operand launches (ea_exact.hip), the VAE's 512-channel convolutions
// (ldm/modules/diffusionmodules/model.py:619-652).  Everything else stays on ea_gemm2.h (the planner: ea_gemm.hip
// gemm8_shape_ok -- measured per class on the MI355X, profiles/r05_gemm8_*).
//
// Why another main loop (profiles/HISTORY.md 8e-1, 10-1): ea_gemm2's wave tile is 64 x 80 -- 9 fragment reads (ds_read_b128) per 20
// MFMAs, and its compute side tops out at ~1.29 PF/s on LDS read issue alone.  Here a wave owns 128 x 64 outputs (8 x 4
// MFMA tiles, 128 accumulator registers) and a K tile is cut into FOUR phases, one 64 x 32 accumulator quadrant each:
// 16 MFMAs (v_mfma_f32_16x16x32_f16) fed by 8 (A) or 4 (B) fragment reads; the B fragments of a quadrant column stay in
// registers across two phases.  That is 24 fragment reads per 64 MFMAs (ea_gemm2: 28.8).
//
// Schedule (MI355X guide section 5, "256^2 8-phase" template; prototype + measurements: tools/gemm8_probe.hip,
// profiles/r05_gemm8_probe_staggered.jsonl):
//  * LDS: two K tiles of 64 KiB.  A K tile is staged as FOUR half tiles cut by USE -- A-h0 = the rows both wave rows read
//    for their first quadrant, B-h0 = the 32 columns each wave column reads first, ... -- one half (2 LDS-DMA instructions
//    per wave) per phase, in the order the next tile needs them: A-h0, B-h0, B-h1, A-h1.
//  * every phase is  [load part | barrier | MFMA part | barrier]:  the load part issues the phase's fragment reads and
//    one half tile of the NEXT K tile, then waits with a COUNTED vmcnt(4) (the two youngest halves stay in flight -- the
//    queue never drains in the steady state); the MFMA part is a pure 16-MFMA cluster at raised priority.
//  * the two wave rows are STAGGERED by one barrier: on every SIMD one wave sits in its MFMA cluster while its partner
//    (the other wave row's wave of the same column pair) reads fragments and issues DMA.  The matrix pipe of a SIMD
//    alternates between its two waves and never waits for an LDS round trip or a VMEM issue burst of the wave feeding it.
//    (Lockstep -- all eight waves in the same part at once -- measured 1086 TF/s at 4096^3; staggered 1223-1235.)
//  * RAW: a half staged in load part L(q) is first read in L(q + 3) (L(q + 4) for A-h0), one barrier after the LAGGING
//    wave row's own vmcnt wait for it.  WAR: a half's last read (retired by the lgkmcnt(0) in front of that phase's
//    MFMAs) lies >= 2 barriers before the DMA that restages its buffer.
//  * im2col addressing as in ea_gemm2.h: K ordered (tap, cin), a 64-wide K tile never straddles a tap or a concat
//    source; the per-lane pixel offsets of the four A pieces are recomputed only when the tap / source changes.
#pragma once
#include "ea_gemm.h"
#include "ea_prims.h"
#include "ea_epi_tr.h"

#define EA_G8_BM 256
#define EA_G8_BN 256
#define EA_G8_KT_BYTES ((EA_G8_BM + EA_G8_BN) * 128)   // one K tile: 64 KiB
#define EA_G8_LDS_BYTES (2 * EA_G8_KT_BYTES)

__device__ __forceinline__ void ea_g8_barrier() {      // no lgkmcnt wait in front: fragment reads fly across the barrier
#ifdef EA_EMU
  ea_emu::block_sync();
#else
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}
__device__ __forceinline__ void ea_g8_wait_lds() {
#ifndef EA_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void ea_g8_prio(int on) {
#ifndef EA_EMU
  if (on) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
  (void)on;
#endif
}
// rows of half `h` of the A region, 8-row group g (0..15): wave row (g >> 3), 64-row quadrant h
__device__ __forceinline__ int ea_g8_a_row(int h, int g) { return (g >> 3) * 128 + h * 64 + (g & 7) * 8; }
// rows (= output columns) of half `h` of the B region, group g: wave column (g >> 2), 32-column quadrant h
__device__ __forceinline__ int ea_g8_b_row(int h, int g) { return (g >> 2) * 64 + h * 32 + (g & 3) * 8; }

template <int TRX>
__device__ __forceinline__ void ea_gemm8_tile(const EaGemmParams& p, const int wg_x, const int wg_z) {
  constexpr int BM = EA_G8_BM, BN = EA_G8_BN;
  EA_SMEM(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = ea_uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

#ifndef EA_EMU
  {   // one batch of scalar loads over the kernel-argument block's cache lines (see ea_gemm2.h)
    static_assert(sizeof(EaGemmParams) > 0x100, "the five 64-byte lines touched below must lie inside the kernel-argument block");
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned w0, w1, w2, w3, w4;
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\t"
                 "s_load_dword %3, %5, 0xc0\n\ts_load_dword %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3), "=&s"(w4) : "s"(ka) : "memory");   // early-clobber: no output may land on the ka pair (SMEM returns are asynchronous)
  }
#endif
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = ea_xcd_remap(wg_x, tiles_m * tiles_n);
  int tm = ea_uniform(tile / tiles_n), tn = tile - tm * tiles_n;
  if (p.raster_gm > 1) {
    int unused;
    ea_grouped_item(tile, tiles_m, tiles_n, p.raster_gm, tm, tn, unused);
    tm = ea_uniform(tm);
    tn = ea_uniform(tn);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int bz = wg_z;
  const int batch = ea_uniform(bz / p.splits), split = bz - batch * p.splits;
  const int nk_total = p.K / EA_BK;
  const int kt_begin = split * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nk_total) kt_end = nk_total;
  const int nkt = kt_end - kt_begin;

  const ea_rsrc rs_a1 = ea_make_rsrc(p.a1 + batch * p.strideA);
  const ea_rsrc rs_a2 = ea_make_rsrc(p.a2 ? p.a2 + batch * p.strideA : p.a1);
  const ea_rsrc rs_w = ea_make_rsrc(p.w + batch * p.strideW);

  // ---- per-lane DMA coordinates.  Piece (h, i) of this wave = 8-row group g = 2 * wave + i of half h; lane l fetches row
  // (l >> 3), 16-byte slot (l & 7) of the group (source chunk XOR-swizzled, the same involution as the fragment reads)
  const int lrow = lane >> 3, slot = lane & 7;
  // conv set-up: pixel -> (sample, row, column) by float-reciprocal division where the row count allows (ea_prims.h ea_div_small)
  const bool div_small = p.conv && p.M < EA_DIV_SMALL_MAX;
  const float rcp_hw = 1.0f / (float)(p.conv ? p.Hout * p.Wout : 1), rcp_w = 1.0f / (float)(p.conv ? p.Wout : 1);
  int a_y[2][2], a_x[2][2], a_base[2][2], a_lds[2][2], b_lds[2][2];
  unsigned a_chunk[2][2], a_voff[2][2], b_voff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wave + i;
      const int ar = ea_g8_a_row(h, g), br = ea_g8_b_row(h, g);
      a_lds[h][i] = ar * 128;
      b_lds[h][i] = BM * 128 + br * 128;
      const int r = ar + lrow;
      a_chunk[h][i] = (unsigned)((slot ^ ea_swz(r)) * 8);
      const int m = m0 + r;
      a_y[h][i] = a_x[h][i] = 0;
      a_base[h][i] = -1;
      a_voff[h][i] = EA_OOB;
      if (m < p.M) {
        if (p.conv) {
          const int hw = p.Hout * p.Wout;
          const int b = div_small ? ea_div_small(m, hw, rcp_hw) : m / hw;
          const int rem = m - b * hw;
          const int oy = div_small ? ea_div_small(rem, p.Wout, rcp_w) : rem / p.Wout;
          a_base[h][i] = b * p.Hin * p.Win;
          a_y[h][i] = oy * p.stride - p.pad;
          a_x[h][i] = (rem - oy * p.Wout) * p.stride - p.pad;
        } else {
          a_voff[h][i] = ((unsigned)m * (unsigned)p.lda + a_chunk[h][i]) * 2u;
        }
      }
      const int rb = br + lrow, n = n0 + rb;
      b_voff[h][i] = (n < p.N) ? ((unsigned)n * (unsigned)p.ldw + (unsigned)((slot ^ ea_swz(rb)) * 8)) * 2u : EA_OOB;
    }

  const int ctot = p.c1 + p.c2;
  int k_cur = kt_begin * EA_BK;   // first K element of the tile being staged
  int tap = 0, cin = 0;
  auto set_voff = [&]() {
    const int ky = (p.ksize == 3) ? tap / 3 : 0;
    const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
    const int hlim = p.ups ? 2 * p.Hin : p.Hin;
    const int wlim = p.ups ? 2 * p.Win : p.Win;
    const unsigned cs = (unsigned)(cin >= p.c1 ? p.c2 : p.c1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int iy = a_y[h][i] + ky, ix = a_x[h][i] + kx;
        const bool ok = a_base[h][i] >= 0 && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
        if (p.ups) { iy >>= 1; ix >>= 1; }
        a_voff[h][i] = ok ? ((unsigned)(a_base[h][i] + iy * p.Win + ix) * cs + a_chunk[h][i]) * 2u : EA_OOB;
      }
  };
  if (p.conv) {
    tap = ea_uniform(k_cur / ctot);
    cin = k_cur - tap * ctot;
    set_voff();
  }
  // scalar state of the K tile being staged (fixed by `begin_tile`, used by its four half-tile issues)
  ea_rsrc is_rs_a = rs_a1;
  unsigned is_soff_a = 0, is_soff_b = 0;
  char* is_buf = smem;
  auto begin_tile = [&](int t) {          // t: tile index within this slice
    is_buf = smem + (t & 1) * EA_G8_KT_BYTES;
    k_cur = ea_uniform(k_cur);
    cin = ea_uniform(cin);
    tap = ea_uniform(tap);
    const bool second = p.conv && cin >= p.c1;
    is_rs_a = second ? rs_a2 : rs_a1;
    is_soff_a = (unsigned)(p.conv ? (second ? cin - p.c1 : cin) : k_cur) * 2u;
    is_soff_b = (unsigned)k_cur * 2u;
  };
  auto advance_k = [&]() {                // after the LAST half of a tile is issued
    k_cur += EA_BK;
    if (p.conv) {
      cin += EA_BK;
      if (cin >= ctot) {
        cin = 0;
        ++tap;
        set_voff();
      } else if (cin == p.c1) {
        set_voff();
      }
    }
  };
  auto stage_a = [&](int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ea_dma16(is_rs_a, a_voff[h][i], is_soff_a, is_buf + a_lds[h][i]);
  };
  auto stage_b = [&](int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ea_dma16(rs_w, b_voff[h][i], is_soff_b, is_buf + b_lds[h][i]);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const int l15 = lane & 15, lq = lane >> 4;
  f16x8 fa[4][2], fb[2][2][2];
  auto read_a = [&](const char* buf, int qa) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wm * 128 + qa * 64 + i * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = *reinterpret_cast<const f16x8*>(buf + r * 128 + (((ks * 4 + lq) ^ ea_swz(r)) << 4));
    }
  };
  auto read_b = [&](const char* buf, int qb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wn * 64 + qb * 32 + j * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[qb][j][ks] = *reinterpret_cast<const f16x8*>(buf + BM * 128 + r * 128 + (((ks * 4 + lq) ^ ea_swz(r)) << 4));
    }
  };
  // D^T = W A^T (operands swapped): a lane ends up with 4 consecutive output columns of one output row (ea_epi_tr.h)
  auto mma = [&](int qa, int qb) {
    ea_g8_wait_lds();
    ea_g8_prio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[qa * 4 + i][qb * 2 + j] = ea_mfma_16x16x32(fb[qb][j][ks], fa[i][ks], acc[qa * 4 + i][qb * 2 + j]);
    ea_g8_prio(0);
  };

  if (nkt > 0) {
    // prologue: the four halves of the slice's first K tile, in use order
    begin_tile(0);
    stage_a(0);
    stage_b(0);
    stage_b(1);
    stage_a(1);
    advance_k();
    ea_wait_dma<4>();
    ea_g8_barrier();
    if (wm == 1) ea_g8_barrier();          // stagger: wave row 1 runs one barrier behind wave row 0
    auto tile_body = [&](int t, auto more_c) {
      constexpr bool more = decltype(more_c)::value;
      const char* buf = smem + (t & 1) * EA_G8_KT_BYTES;
      if (p.acc_scale_kt > 0 && kt_begin + t == p.acc_scale_kt) {
        // K-concatenated split operands (ea_epilogue.acc_scale_k): the correction products are in, scale them (exactly: a
        // power of two) before the hi x hi product is accumulated on top
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] *= p.acc_scale;
      }
      // phase 0: quadrant (A0, B0); stage A-h0 of the next tile
      read_a(buf, 0);
      read_b(buf, 0);
      if (more) { begin_tile(t + 1); stage_a(0); ea_wait_dma<4>(); } else ea_wait_dma<2>();
      ea_g8_barrier();
      mma(0, 0);
      ea_g8_barrier();
      // phase 1: (A0, B1); stage B-h0
      read_b(buf, 1);
      if (more) { stage_b(0); ea_wait_dma<4>(); } else ea_wait_dma<0>();
      ea_g8_barrier();
      mma(0, 1);
      ea_g8_barrier();
      // phase 2: (A1, B1); stage B-h1
      read_a(buf, 1);
      if (more) { stage_b(1); ea_wait_dma<4>(); }
      ea_g8_barrier();
      mma(1, 1);
      ea_g8_barrier();
      // phase 3: (A1, B0) -- B quadrant 0 is still in registers; stage A-h1
      if (more) { stage_a(1); advance_k(); ea_wait_dma<4>(); }
      ea_g8_barrier();
      mma(1, 0);
      ea_g8_barrier();
    };
    for (int t = 0; t + 1 < nkt; ++t) tile_body(t, std::true_type{});   // no "is there a next tile" test inside a phase
    tile_body(nkt - 1, std::false_type{});
    if (wm == 0) ea_g8_barrier();          // stagger: both wave rows execute the same number of barriers
  }

  float ln_mu[8], ln_rs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ln_mu[i] = 0.0f; ln_rs[i] = 1.0f; }
  ea_tr_epilogue<8, 4, TRX, true, true>(p, acc, m0 + wm * 128, n0 + wn * 64, m0, batch, bz, ln_mu, ln_rs, smem, wave);
}

// The grid may be NARROWER than the tile count: workgroup w then walks tiles w, w + gridDim.x, ... (a persistent launch; the host
// picks it from 8 rounds of tiles up, ea_gemm.hip).  A 256 x 256 workgroup owns a whole CU (128 KiB of LDS, 2 x 250 registers per
// SIMD lane), so with one workgroup per tile the next one is dispatched only after the slowest wave of the previous has retired;
// walking the tiles inside the launch removes that at the price of the dispatcher's load balancing.  tile -> (row, column) uses
// the same XCD-aware map either way (workgroup w and tile w + k * 256 sit on the same XCD).  Between two tiles nothing needs a
// barrier of its own: both wave rows leave a tile through the same final barrier (after which no wave reads the stage ring again)
// and the register-direct epilogue does not touch LDS.
template <int TRX>
__global__ __launch_bounds__(512, 2) void ea_gemm8_kernel(EaGemmParams p) {
  const int ntile = ((p.M + EA_G8_BM - 1) / EA_G8_BM) * ((p.N + EA_G8_BN - 1) / EA_G8_BN);
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) ea_gemm8_tile<TRX>(p, t, blockIdx.z);
}
