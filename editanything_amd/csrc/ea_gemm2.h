// ea_gemm2.h -- the fast MFMA contraction kernel (LDS-DMA staged, 160-wide tiles).
//
// Same contract as ea_gemm.h's kernel (C = epilogue(A W^T), A dense or the implicit
// im2col of an NHWC activation) for the aligned shapes the hot path is made of:
// K % 64 == 0 and, for convolutions, channel counts that are multiples of 64 -- i.e.
// every ResBlock / Up / Downsample conv (openaimodel.py:108-152,200-231), zero-conv
// (cldm/cldm.py:281-305) and transformer Linear (attention.py:54,152-160,316-339) of
// SD2.1/SD1.5, the SAM ViT Linears and the VAE convs.  ea_gemm.h stays as the generic
// path (ragged K, Cin = 4/8/16/96, x2_add).
//
// MI355X design (cdna_hip_programming.md section 5 / T2 / rule 21):
//  * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave
//    instruction, no VGPR round trip, no ds_write pass).  The LDS image of a wave
//    instruction is lane-linear, so the bank-conflict XOR swizzle is applied on the
//    per-lane SOURCE chunk and again on the fragment read (the same involution).
//    Buffer descriptors give the zero fill for free: convolution padding and ragged
//    M / N edges use an out-of-range per-lane offset, which the hardware returns as 0.
//  * the im2col address is tracked incrementally: K is ordered (tap, cin), a 64-wide
//    K tile never straddles a tap or a concat source, so per K tile only the scalar
//    offset moves (no vector ALU work at all in the steady state); per-lane pixel
//    offsets + validity are recomputed when the tap or the concat source changes.
//  * 160-wide N tiles: every channel count of the UNet/ControlNet (320/640/1280 and
//    their 3x/8x products) is a multiple of 160, so 128x160 tiles cover
//    [32768 x 320] with exactly 512 workgroups = 2 per CU and [8192 x 640] with 256.
//    v_mfma_f32_16x16x32_f16, 4 waves as 2x2, wave tile 64x80 (4x5 MFMA tiles, 80
//    accumulator VGPRs); 2-stage LDS ring (72 KiB) -> two workgroups per CU, whose
//    barriers / DMA waits overlap each other's MFMA phases.
//  * epilogue: each wave stages its own accumulator tile through a private fp32 LDS slab
//    (no workgroup barriers), gathers every global read of a slab before the first use
//    and writes 16-byte coalesced vectors; GEGLU weights are packed [40 value | 40 gate]
//    per 80 rows so both halves of an output land in the same wave's slab.
#pragma once
#include "ea_gemm.h"
#include "ea_prims.h"
#include "ea_epi_tr.h"

#ifndef EA_EXP
#define EA_EXP 0   // bit mask of compile-time experiments (tools/build_exp.sh builds side libraries; 0 in the product)
#endif
// EA_TOOLS = 1 (tools/build_exp.sh side libraries and the CPU emulation build of the tests; 0 in the product): compiles in
// the K-loop / epilogue ablation selectors (EaGemmParams::debug) and the opt-in instantiations (launch_fast kinds 2-8,
// 10-13).  The shipped library carries the two planned instantiations (128- / 64-row 2-stage tiles, each with the
// register-direct epilogue forms) and the persistent kernel of ea_gemm3.h, and no debug selector reaches its kernels.
#ifndef EA_TOOLS
#define EA_TOOLS 0
#endif
#define EA_DBG(n) (EA_TOOLS && p.debug == (n))
// Phase timestamps for tools/phase_times.py (EA_GEMM2_DEBUG=3): wave 0 / lane 0 of every workgroup stores the
// constant-rate wall clock (100 MHz) at a few program points into the (otherwise unused) split-K workspace.
#ifdef EA_EMU
#define EA_STAMP(i) do {} while (0)
#else
#define EA_STAMP(i)                                                                                   \
  do {                                                                                                \
    if (EA_DBG(3) && tid == 0)                                                                     \
      reinterpret_cast<unsigned long long*>(p.partial)[(wg_x + gridDim.x * wg_z) * 8 + (i)] = wall_clock64(); \
  } while (0)
#endif


// MT = 16: v_mfma_f32_16x16x32_f16 (wave tile WTM x WTN in 16x16 tiles); MT = 32: v_mfma_f32_32x32x16_f16.
// Register budget: two co-resident 4-wave workgroups per CU (the 2-stage 128-row tiles, <= 80 KiB LDS each) need
// <= 256 VGPR+AGPR per lane, i.e. 2 waves per SIMD; the 1-workgroup-per-CU instantiations may use the whole file.
constexpr int ea_gemm2_occ(int bm, int bn, int nwaves, int stages) {
  return (nwaves == 4 && stages * (bm + bn) * 128 <= 80 * 1024) ? 2 : 1;
}

// LDR = 1: wave specialisation.  The workgroup carries NW extra "loader" waves (one beside each MFMA wave on its SIMD)
// that do nothing but issue the LDS-DMA of the tiles ahead; the NW compute waves run a pure ds_read + MFMA stream.
// A wave's own DMA burst (~60 clk per 1-KiB piece, 9 pieces per tile) otherwise sits in front of its MFMAs every
// iteration -- for a workgroup alone on its CU the K-loop iteration is issue + compute (0.7 us) instead of
// max(issue, compute).  One barrier per K tile, 3-deep ring: at barrier kt the loaders have waited for tile kt
// (counted vmcnt, tile kt + 1 still in flight), the compute waves have retired their reads of tile kt - 1, whose buffer
// the loaders refill with tile kt + 2 right after.
// TR = 1 / 2: TRANSPOSED accumulators + register-direct epilogue (2 = with the LayerNorm fold and the row-statistics
// output compiled in; the plain launches run the instantiation without them).  The MFMA operands are swapped (D^T = W A^T), so a lane
// holds 4 CONSECUTIVE output columns of one output row per 16x16 tile (row = lane % 16, columns 4 * (lane / 16) ..+3)
// instead of 4 consecutive rows of one column: after one v_permlane16_swap per register two tiles give every lane 8
// consecutive columns, i.e. the finished fp16 row segment goes from registers to memory as a 16-byte store -- no LDS
// slab, no scatter / gather passes, no waits between slabs.  Measured with tools/gemm_bench --debug 0,1,2 (round 2): the
// LDS-slab epilogues cost 25 % of the contraction time of an evaluation (10-54 us per launch; the bare store stream of
// the same bytes takes 5-15 us, tools/probe_misc).  Same products and the same fp32 summation order as TR = 0.
// The body is a device function of (problem, workgroup x index, workgroup z index) so that ONE grid can carry the tiles
// of two problems (ea_gemm2_pair_kernel below).
// KS = 2 (round 6, EXPERIMENT -- tools builds, variants 31 / 32): INTRA-WORKGROUP split-K.  The workgroup carries KS wave groups of
// WM x WN waves; group g owns its own 2-stage ring and multiplies the K tiles g, g + KS, ... of the workgroup's K range into its own
// accumulators (the groups share every barrier: they run the same steps on different K tiles), then the groups' accumulators are
// summed through LDS and group 0 runs the register-direct epilogue.  What it is for: a launch whose tiles give every CU ONE 4-wave
// workgroup has a latency-bound K loop (0.6 - 0.7 us per 64-deep K tile whatever the tile does: DMA -> barrier -> fragment reads);
// two such workgroups per CU hide each other's latency but need split-K across workgroups -- fp32 partials through HBM and a
// reduce launch.  With KS = 2 one workgroup has the two K streams and the reduction stays on chip.
template <int BM, int BN, int WM, int WN, int STAGES, int MT, int ILV, int LDR = 0, int TR = 0, int KS = 1>
__device__ __forceinline__ void ea_gemm2_tile(const EaGemmParams& p, const int wg_x, const int wg_z) {
  constexpr int NW = WM * WN, NT = NW * 64 * (1 + LDR) * KS;
  static_assert(KS == 1 || (KS == 2 && (TR == 1 || TR == 2) && STAGES == 2 && MT == 16 && !LDR && !ILV), "intra-workgroup split-K: register-direct 2-stage tiles");
  static_assert(!TR || (MT == 16 && ILV == 0 && !LDR && (STAGES == 2 || STAGES == 3)), "register-direct epilogue: the 2- / 3-stage 16x16x32 tiles");
  static_assert(!LDR || !ILV, "loader waves replace the interleaved issue");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / MT, NI = WTN / MT;
  static_assert(MT == 16 || MT == 32, "MFMA tile");
  static_assert(ILV == 0 || STAGES == 3, "interleaved issue / ping-pong need the 3-deep ring");
  static_assert(WTM % MT == 0 && WTN % MT == 0, "wave tile must be a whole number of MFMA tiles");
  constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;          // 1-KiB LDS-DMA instructions per K tile
  constexpr int A_PW = (A_INSTR + NW - 1) / NW, B_PW = (B_INSTR + NW - 1) / NW;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");
  EA_SMEM(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = ea_uniform(tid >> 6);
  const bool is_loader = LDR && wave_all >= NW;
  const int grp = (KS > 1) ? ea_uniform(wave_all / NW) : 0;                                   // K-stream group (KS > 1)
  const int wave = (KS > 1) ? wave_all - grp * NW : (is_loader ? wave_all - NW : wave_all);   // index among the loaders / among the compute waves
  char* const smem_g = smem + grp * (STAGES * (BM + BN) * 128);                               // this group's stage ring
  const int wm = wave / WN, wn = wave % WN;

#if !(EA_EXP & 64) && !defined(EA_EMU)
  // touch every 64-byte line of the kernel-argument block with ONE batch of scalar loads at entry, so that the compiler's five
  // dependent batches of argument loads further down hit the scalar cache: -0.2 ... -0.4 us per launch on the K <= 1280 Linears
  // (profiles/r05_kernarg_warm_ab.jsonl; EA_EXP & 64 switches it off for A/B builds)
  {
    static_assert(sizeof(EaGemmParams) > 0x100, "the five 64-byte lines touched below must lie inside the kernel-argument block");
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned w0, w1, w2, w3, w4;
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\t"
                 "s_load_dword %3, %5, 0xc0\n\ts_load_dword %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3), "=&s"(w4) : "s"(ka) : "memory");   // early-clobber: no output may land on the ka pair (SMEM returns are asynchronous)
  }
#endif
  EA_STAMP(0);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = ((p.M + BM - 1) / BM) * tiles_n;
  const int tile = ea_xcd_remap(wg_x, ntile);
  // integer division runs on the vector ALU: mark the quotients wave-uniform so everything derived from them
  // (K position, descriptors, scalar offsets) stays in SGPRs -- otherwise every DMA is wrapped in a waterfall loop (T20)
  int tm = ea_uniform(tile / tiles_n), tn = tile - tm * tiles_n;
  if (p.raster_gm > 1) {   // grouped order: an XCD's contiguous chunk of tiles (ea_xcd_remap) covers few W column panels x raster_gm row tiles
    int unused;
    ea_grouped_item(tile, ntile / tiles_n, tiles_n, p.raster_gm, tm, tn, unused);
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
  const int nk = EA_DBG(2) ? 0 : kt_end - kt_begin;

  const ea_rsrc rs_a1 = ea_make_rsrc(p.a1 + batch * p.strideA);
  const ea_rsrc rs_a2 = ea_make_rsrc(p.a2 ? p.a2 + batch * p.strideA : p.a1);
  const ea_rsrc rs_w = ea_make_rsrc(p.w + batch * p.strideW);

  // ---- per-lane DMA coordinates: instruction j of this wave covers LDS rows (j*NW + wave)*8 .. +7
  const int lrow = lane >> 3, slot = lane & 7;
  // conv set-up: pixel -> (sample, row, column) by float-reciprocal division where the row count allows (ea_prims.h ea_div_small)
  const bool div_small = p.conv && p.M < EA_DIV_SMALL_MAX;
  const float rcp_hw = 1.0f / (float)(p.conv ? p.Hout * p.Wout : 1), rcp_w = 1.0f / (float)(p.conv ? p.Wout : 1);
  int a_y[A_PW], a_x[A_PW];   // conv: input-space origin of the row's pixel
  int a_base[A_PW];           // conv: b*Hin*Win (pixel index of the sample), -1 = row out of range
  unsigned a_chunk[A_PW];     // element offset of the (swizzled) 16-B chunk this lane fetches
  unsigned a_voff[A_PW];      // per-lane byte offset of the DMA (EA_OOB -> zeros)
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int r = (j * NW + wave) * 8 + lrow;
    a_chunk[j] = (unsigned)((slot ^ ea_swz(r)) * 8);
    const int m = m0 + r;
    const bool ok = (r < BM) && (m < p.M);
    a_y[j] = a_x[j] = 0;
    a_base[j] = -1;
    a_voff[j] = EA_OOB;
    if (ok) {
      if (p.conv) {
        const int hw = p.Hout * p.Wout;
        const int b = div_small ? ea_div_small(m, hw, rcp_hw) : m / hw;
        const int rem = m - b * hw;
        const int oy = div_small ? ea_div_small(rem, p.Wout, rcp_w) : rem / p.Wout;
        a_base[j] = b * p.Hin * p.Win;
        a_y[j] = oy * p.stride - p.pad;
        a_x[j] = (rem - oy * p.Wout) * p.stride - p.pad;
      } else {
        a_voff[j] = ((unsigned)m * (unsigned)p.lda + a_chunk[j]) * 2u;
      }
    }
  }
  unsigned b_voff[B_PW];
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int r = (j * NW + wave) * 8 + lrow;
    const int n = n0 + r;
    b_voff[j] = (r < BN && n < p.N) ? ((unsigned)n * (unsigned)p.ldw + (unsigned)((slot ^ ea_swz(r)) * 8)) * 2u : EA_OOB;
  }

  const int ctot = p.c1 + p.c2;
  int k_cur = (kt_begin + grp) * EA_BK;  // first K element of the next tile to stage (KS > 1: group g starts at its own tile)
  const int nkg = (KS > 1) ? (nk > grp ? (nk - grp + KS - 1) / KS : 0) : nk;   // K tiles of this group
  int tap = 0, cin = 0;
  // conv: per-lane offsets for the current (tap, concat source); runs when either changes (every >= c/64 K tiles)
  auto set_voff = [&]() {
    const int ky = (p.ksize == 3) ? tap / 3 : 0;
    const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
    const int hlim = p.ups ? 2 * p.Hin : p.Hin;
    const int wlim = p.ups ? 2 * p.Win : p.Win;
    const unsigned cs = (unsigned)(cin >= p.c1 ? p.c2 : p.c1);
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      int iy = a_y[j] + ky, ix = a_x[j] + kx;
      const bool ok = a_base[j] >= 0 && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
      if (p.ups) { iy >>= 1; ix >>= 1; }
      a_voff[j] = ok ? ((unsigned)(a_base[j] + iy * p.Win + ix) * cs + a_chunk[j]) * 2u : EA_OOB;
    }
  };
  if (p.conv) {
    tap = ea_uniform(k_cur / ctot);
    cin = k_cur - tap * ctot;
    set_voff();
  }

  // Staging of one K tile = PIECES 1-KiB DMA instructions per wave (A rows first, then W rows).  `begin_issue` fixes
  // the tile's scalar state, `issue_piece` launches one instruction, `end_issue` advances K (and, for convolutions,
  // the tap / concat-source state).
  constexpr int PIECES = A_PW + B_PW;
  ea_rsrc is_rs_a = rs_a1;
  unsigned is_soff_a = 0, is_soff_b = 0;
  char* is_sa = smem_g;
  char* is_sb = smem_g;
  auto begin_issue = [&](int buf) {
    is_sa = smem_g + buf * STAGE_BYTES;
    is_sb = is_sa + BM * 128;
    k_cur = ea_uniform(k_cur);   // loop-carried scalars: keep them provably wave-uniform (SGPR descriptors / offsets)
    cin = ea_uniform(cin);
    tap = ea_uniform(tap);
    const bool second = p.conv && cin >= p.c1;
    is_rs_a = second ? rs_a2 : rs_a1;
    is_soff_a = (unsigned)(p.conv ? (second ? cin - p.c1 : cin) : k_cur) * 2u;
    is_soff_b = (unsigned)k_cur * 2u;
  };
  auto issue_piece = [&](int pc) {
    if (pc < A_PW) {
      if (A_INSTR % NW == 0 || pc * NW + wave < A_INSTR) ea_dma16(is_rs_a, a_voff[pc], is_soff_a, is_sa + (pc * NW + wave) * 1024);
    } else {
      const int j = pc - A_PW;
      if (B_INSTR % NW == 0 || j * NW + wave < B_INSTR) ea_dma16(rs_w, b_voff[j], is_soff_b, is_sb + (j * NW + wave) * 1024);
    }
  };
  auto end_issue = [&]() {
    if (KS > 1) {   // stride of KS tiles: taps / concat sources may be stepped over
      k_cur += EA_BK * KS;
      if (p.conv) {
        const int tap0 = tap;
        const bool sec0 = cin >= p.c1;
        cin += EA_BK * KS;
        while (cin >= ctot) { cin -= ctot; ++tap; }
        if (tap != tap0 || (cin >= p.c1) != sec0) set_voff();
      }
      return;
    }
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
  auto issue_tile = [&](int buf) {
    begin_issue(buf);
#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) issue_piece(pc);
    end_issue();
  };

  // accumulators: MT == 16 -> f32x4 per tile, MT == 32 -> f32x16 per tile (the unused array is dead code)
  f32x4 acc[MT == 16 ? MI : 1][MT == 16 ? NI : 1];
  f32x16 acc32[MT == 32 ? MI : 1][MT == 32 ? NI : 1];
  if (MT == 16) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[MT == 16 ? i : 0][MT == 16 ? j : 0][r] = 0.0f;
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[MT == 32 ? i : 0][MT == 32 ? j : 0][r] = 0.0f;
  }

  // fragment read coordinates: row (lane % MT) of an MFMA tile; 16-B chunk (lane / MT) of a K step
  // (16x16x32: 4 chunks = 32 K per step, 2 steps per tile; 32x32x16: 2 chunks = 16 K per step, 4 steps per tile)
  const int frow = lane & (MT - 1), fq = lane / MT;
  // One K tile: KSTEPS MFMA steps.  Fragments are register double-buffered -- step s+1's ds_reads are issued before
  // step s's MFMAs, so the LDS latency of a step hides under the previous step's matrix work (the waits become
  // counted lgkmcnt(N), not drains).  Only the first step of a tile is exposed; the co-resident workgroup covers it.
  // `ilv_issue` (ILV instantiations): the next tile's DMA pieces are issued BETWEEN the MFMA groups of this tile, one
  // every few MFMAs, instead of as one burst in front of them.  A wave's DMA burst occupies the CU's single
  // address/texture path (64 B/clk: 576 clk for a 36-KiB tile) while the wave sits in VMEM issue, i.e. burst + MFMA
  // phases add up (measured: tile time = 576 + 640 clk); interleaved, the path drains while the matrix pipe works.
#if (EA_EXP & 4) && !defined(EA_EMU)
  constexpr int KSTEPS = (MT == 16) ? 2 : 4;
  constexpr int CH_PER_STEP = (MT == 16) ? 4 : 2;
  f16x8 fa[2][MI], fb[2][NI];
  auto load_frags = [&](int buf, int ks, int slot) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + BM * 128;
    const int ch = ks * CH_PER_STEP + fq;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * WTM + i * MT + frow;
      fa[slot][i] = *reinterpret_cast<const f16x8*>(sa + r * 128 + ((ch ^ ea_swz(r)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * WTN + j * MT + frow;
      fb[slot][j] = *reinterpret_cast<const f16x8*>(sb + r * 128 + ((ch ^ ea_swz(r)) << 4));
    }
  };
  // `prefetched`: the caller already issued load_frags(buf, 0, 0) (EA_EXP & 4: ahead of the next tile's DMA burst)
  auto compute_tile = [&](int buf, bool ilv_issue = false, bool prefetched = false) {
    if (!prefetched) load_frags(buf, 0, 0);
#define EA_LOADF(ks_, slot_) load_frags(buf, (ks_), (slot_))
#else
  auto compute_tile = [&](int buf, bool ilv_issue = false) {
    const char* sa = smem_g + buf * STAGE_BYTES;
    const char* sb = sa + BM * 128;
    constexpr int KSTEPS = (MT == 16) ? 2 : 4;
    constexpr int CH_PER_STEP = (MT == 16) ? 4 : 2;
    f16x8 fa[2][MI], fb[2][NI];
    auto load_frags = [&](int ks, int slot) {
      const int ch = ks * CH_PER_STEP + fq;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * WTM + i * MT + frow;
        fa[slot][i] = *reinterpret_cast<const f16x8*>(sa + r * 128 + ((ch ^ ea_swz(r)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int r = wn * WTN + j * MT + frow;
        fb[slot][j] = *reinterpret_cast<const f16x8*>(sb + r * 128 + ((ch ^ ea_swz(r)) << 4));
      }
    };
    load_frags(0, 0);
#define EA_LOADF(ks_, slot_) load_frags((ks_), (slot_))
#endif
#if (EA_EXP & 2) && !defined(EA_EMU)
    __builtin_amdgcn_s_setprio(1);   // experiment: the MFMA stream outranks the co-resident workgroup's issue phases
#endif
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 1 < KSTEPS) EA_LOADF(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (MT == 16 && TR) acc[MT == 16 ? i : 0][MT == 16 ? j : 0] = ea_mfma_16x16x32(fb[ks & 1][j], fa[ks & 1][i], acc[MT == 16 ? i : 0][MT == 16 ? j : 0]);
          else if (MT == 16) acc[MT == 16 ? i : 0][MT == 16 ? j : 0] = ea_mfma_16x16x32(fa[ks & 1][i], fb[ks & 1][j], acc[MT == 16 ? i : 0][MT == 16 ? j : 0]);
          else acc32[MT == 32 ? i : 0][MT == 32 ? j : 0] = ea_mfma_32x32x16(fa[ks & 1][i], fb[ks & 1][j], acc32[MT == 32 ? i : 0][MT == 32 ? j : 0]);
        }
        if (ILV == 1) {
          // MFMA group g of G: issue the pieces [g*PIECES/G, (g+1)*PIECES/G) of the next tile
          constexpr int G = KSTEPS * MI;
          const int g = ks * MI + i;
          if (ilv_issue) {
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc)
              if (pc >= (g * PIECES) / G && pc < ((g + 1) * PIECES) / G) issue_piece(pc);
          }
        }
      }
    }
#if (EA_EXP & 2) && !defined(EA_EMU)
    __builtin_amdgcn_s_setprio(0);
#endif
#if !(EA_EXP & 16) && !defined(EA_EMU)
    // Pin the fragment double-buffering (guide T19).  hipcc otherwise re-fuses the two K steps to save registers and
    // leaves 3-4 exposed `ds_read -> s_waitcnt lgkmcnt -> MFMA` round trips per K tile (seen in the ISA): step s+1's
    // reads are spread one per two MFMAs of step s; only the first step's reads stay exposed.  Same register count, no
    // change in arithmetic order (bit-identical results); +1..3 % on every 2-stage launch measured
    // (profiles/r01x_gemm_bench.jsonl, "exp1").  EA_EXP & 16 switches it off for A/B builds.
    if (!ILV && !LDR && (STAGES == 2 || (STAGES == 3 && TR))) {
      constexpr int RD = MI + NI, MF = MI * NI;
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
      for (int ks = 0; ks + 1 < KSTEPS; ++ks) {
#pragma unroll
        for (int r = 0; r < RD; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (MF >= 2 * RD) ? 2 : 1, 0);
        }
        if (MF > ((MF >= 2 * RD) ? 2 : 1) * RD) __builtin_amdgcn_sched_group_barrier(0x008, MF - ((MF >= 2 * RD) ? 2 : 1) * RD, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
    }
#endif
  };
#undef EA_LOADF

  // LayerNorm fold (register-direct epilogue): mean / rstd of this lane's MI output rows from the producer's row
  // partials.  Runs right AFTER the first tile's DMA is issued (below), every load in flight before the first add, so
  // the round trip rides under that tile's latency.  (Measured: inside the epilogue the dependent 8-byte loads cost each
  // workgroup ln_parts x ~0.5 us; in front of the first DMA issue still 3-4 us per workgroup round.)
  float ln_mu[MI], ln_rs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { ln_mu[i] = 0.0f; ln_rs[i] = 1.0f; }
  auto ln_prologue = [&]() {
    if (!(TR == 2 && p.epi.ln_stats)) return;
    const int c16p = lane & 15;
    constexpr int CH = 8;                     // parts per chunk: CH * MI 8-byte loads in flight
    float s1[MI], s2[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { s1[i] = 0.0f; s2[i] = 0.0f; }
    for (int pp0 = 0; pp0 < p.epi.ln_parts; pp0 += CH) {
      f32x2 t2[CH][MI];
      // UNCONDITIONAL loads (indices clamped, the surplus masked arithmetically): a per-load `if` makes hipcc branch
      // around every load and wait for it on its own (guide section 5, trap (c)) -- 32 dependent round trips, 5 us per
      // workgroup, measured
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          int m = m0 + wm * WTM + i * 16 + c16p;
          m = m < p.M ? m : p.M - 1;
          int pp = pp0 + u;
          pp = pp < p.epi.ln_parts ? pp : p.epi.ln_parts - 1;
          t2[u][i] = *reinterpret_cast<const f32x2*>(p.epi.ln_stats + ((long long)pp * p.M + m) * 2);
        }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float keep = (pp0 + u < p.epi.ln_parts) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) { s1[i] += keep * t2[u][i][0]; s2[i] += keep * t2[u][i][1]; }
      }
    }
    const float inv = 1.0f / (float)p.K;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float mu = s1[i] * inv;
      ln_mu[i] = mu;
      ln_rs[i] = 1.0f / sqrtf(fmaxf(s2[i] * inv - mu * mu, 0.0f) + p.epi.ln_eps);
    }
  };
  EA_STAMP(1);
  if (STAGES == 2 && LDR) {
    // loader waves + 2-deep ring: the tile after the one being multiplied is in flight during exactly one compute
    // phase, so a single workgroup is DMA-latency bound -- this instantiation is built for TWO 8-wave workgroups per CU
    // (64-row tiles: <= 128 registers, 56 KiB of LDS), whose compute phases fill each other's waits.
    if (is_loader && nk > 0) issue_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
      if (is_loader) ea_wait_dma<0>();
      ea_raw_barrier();
      if (is_loader) {
        if (kt + 1 < nk) issue_tile((kt + 1) & 1);
      } else {
        compute_tile(kt & 1);
      }
    }
  } else if (STAGES == 2) {
    if (nkg > 0) issue_tile(0);
    ln_prologue();
#ifndef EA_EMU
    // Two co-resident workgroups that start together run their DMA-issue and MFMA phases in lockstep (both contend
    // for the texture path, then both for the matrix pipe).  Workgroups b and b + 256 normally share a CU (dispatch
    // is round-robin over XCDs, then CUs): delaying the second one by ~half an iteration lets them alternate
    // (measured +8..12 % on the 512-tile 64x64-level convolutions; a speed heuristic only, never correctness).
    if (!EA_DBG(8) && ((wg_x >> 8) & 1)) __builtin_amdgcn_s_sleep(10);
#endif
    const int nk_steps = (KS > 1) ? (nk + KS - 1) / KS : nk;   // (KS > 1: every group walks the same number of barriers)
    for (int kt = 0; kt < nk_steps; ++kt) {
      // waits for this wave's own LDS-DMA (vmcnt(0), emitted by the fence) and then for everyone's: tile kt is complete
      // in LDS and every wave has finished reading the buffer tile kt+1 is about to overwrite.
      if (!EA_DBG(12)) __syncthreads();                               // debug 12: compute only, no barrier either
#if (EA_EXP & 4) && !defined(EA_EMU)
      // experiment: the first K step's fragment reads go out BEFORE the next tile's DMA burst (whose ~9 x 60 clk of
      // VMEM issue then covers their LDS latency) instead of after it
      load_frags(kt & 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) issue_tile((kt + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_tile(kt & 1, false, true);
#else
      if (kt + 1 < nkg && !EA_DBG(11) && !EA_DBG(12)) issue_tile((kt + 1) & 1);   // debug 11: no staging after the first tile
      if (MT == 16 && p.acc_scale_kt > 0 && kt_begin + kt == p.acc_scale_kt) {
        // K-concatenated split operands (ea_epilogue.acc_scale_k): the correction products are in, scale them (exactly:
        // a power of two) before the hi x hi product is accumulated on top
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[MT == 16 ? i : 0][MT == 16 ? j : 0][r] *= p.acc_scale;
      }
      if (!EA_DBG(10) && kt < nkg) compute_tile(kt & 1);             // debug 10: staging only
#endif
    }
  } else if (ILV == 2) {
    // ---- ping-pong (8 waves, 3-deep ring).  The waves form two groups, G0 = waves [0, NW/2) and G1 = the rest; the
    // hardware places wave i on SIMD i % 4, so every SIMD hosts one wave of each group.  A K tile takes two barrier
    // intervals ("slots"): in slot A group 0 runs its 40 MFMAs for tile kt purely from registers while group 1 is in
    // its load phase (all fragment reads of tile kt into registers + the DMA issue of its pieces of tile kt + 2); in
    // slot B the roles swap.  The matrix pipe of a SIMD therefore alternates between its two waves and never waits
    // for an LDS read or a VMEM issue burst of the wave that feeds it (the lock-step structure above pays DMA issue +
    // LDS latency + MFMA in sequence, all waves at once).
    //   G0, tile kt:  slot A  MFMA(kt); vmcnt(0) [its pieces of tile kt+1, issued a slot ago]; barrier
    //                 slot B  read frags(kt+1); issue its pieces of tile kt+2; lgkmcnt(0); barrier
    //   G1, tile kt:  slot A  read frags(kt);   issue its pieces of tile kt+2; lgkmcnt(0); barrier
    //                 slot B  MFMA(kt); vmcnt(0) [its pieces of tile kt+2]; barrier
    // RAW: every piece of tile t is waited for (by its issuing wave) before a barrier that precedes the first read of
    // tile t (G0 reads it in slot B(t-1), G1 in slot A(t)).  WAR: tile kt+2 reuses the buffer of tile kt-1, whose last
    // reads (G1 in slot A(kt-1), G0 in slot B(kt-2)) retired -- lgkmcnt(0) before a barrier -- at least one barrier
    // before the first piece of tile kt+2 is issued (G1 in slot A(kt)).
    static_assert(ILV != 2 || (NW == 8 && MT == 16 && !LDR), "ping-pong: 8 MFMA waves, 16x16x32");
    constexpr int B_EXTRA = B_INSTR % NW;
    static_assert(A_INSTR % NW == 0, "A rows must divide evenly over the waves");
    constexpr int PER_TILE_LO = A_PW + B_INSTR / NW;
    const int grp = wave / (NW / 2);
    // Fragments of one K tile: both K steps of B (2 x NI) and the FIRST step of A are read in the load phase; the second
    // step's A fragments are read inside the MFMA phase, each into the register of the first-step fragment it replaces
    // (row tile i is finished after NI MFMAs), 15+ MFMAs before their first use -- 56 fragment registers instead of 72
    // (at 72 the kernel spilled into the K loop: two waves per SIMD leave 256 registers per wave).
    f16x8 pfa[MI], pfb[2][NI];
    const char* pp_sa = smem;
    auto pp_load = [&](int buf) {
      const char* sa = smem + buf * STAGE_BYTES;
      const char* sb = sa + BM * 128;
      pp_sa = sa;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * WTM + i * MT + frow;
        pfa[i] = *reinterpret_cast<const f16x8*>(sa + r * 128 + ((fq ^ ea_swz(r)) << 4));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = ks * 4 + fq;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int r = wn * WTN + j * MT + frow;
          pfb[ks][j] = *reinterpret_cast<const f16x8*>(sb + r * 128 + ((ch ^ ea_swz(r)) << 4));
        }
      }
    };
    auto pp_mfma = [&]() {
#ifndef EA_EMU
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[MT == 16 ? i : 0][MT == 16 ? j : 0] = ea_mfma_16x16x32(pfa[i], pfb[ks][j], acc[MT == 16 ? i : 0][MT == 16 ? j : 0]);
          if (ks == 0) {
            const int r = wm * WTM + i * MT + frow;
            pfa[i] = *reinterpret_cast<const f16x8*>(pp_sa + r * 128 + (((4 + fq) ^ ea_swz(r)) << 4));
          }
        }
#ifndef EA_EMU
      // pin the source order (row tile i's NI MFMAs, then its second-step read): left alone, hipcc sinks two of the four
      // reads to just before their first use and stalls the matrix pipe on the LDS round trip
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
      __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto ring = [](int t) { return t % 3; };
    // prologue = the virtual tile -1: everyone stages tile 0, G1 also tile 1; G0 loads frags(0) and stages tile 1
    if (nk > 0) issue_tile(0);
    if (grp == 1 && nk > 1) issue_tile(1);
    if (grp == 1 && nk > 1) {
      if (B_EXTRA != 0 && wave < B_EXTRA) ea_wait_dma<PER_TILE_LO + 1>();
      else ea_wait_dma<PER_TILE_LO>();
    } else {
      ea_wait_dma<0>();
    }
    ea_raw_barrier();
    if (grp == 0) {
      if (nk > 0) pp_load(0);
      if (nk > 1) issue_tile(1);
    } else {
      ea_wait_dma<0>();
    }
    ea_raw_barrier();
    // one loop per group (no control-flow merges inside the K loop: a shared loop makes the fragment registers loop-
    // carried PHIs of both roles and hipcc spills ~36 registers into the loop); both execute two barriers per tile
    const bool do_mfma = !EA_DBG(10), do_dma = !EA_DBG(11);   // ablation knobs (tools/gemm_bench --debug): staging only / compute only
    if (grp == 0) {
      for (int kt = 0; kt < nk; ++kt) {
        if (do_mfma) pp_mfma();                                 // slot A
        ea_wait_dma<0>();
        ea_raw_barrier();
        if (kt + 1 < nk && do_mfma) pp_load(ring(kt + 1));      // slot B
        if (kt + 2 < nk && do_dma) issue_tile(ring(kt + 2));
        ea_raw_barrier();
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        if (do_mfma) pp_load(ring(kt));                         // slot A
        if (kt + 2 < nk && do_dma) issue_tile(ring(kt + 2));
        ea_raw_barrier();
        if (do_mfma) pp_mfma();                                 // slot B
        ea_wait_dma<0>();
        ea_raw_barrier();
      }
    }
  } else {
    // 3-deep ring, two tiles in flight: at iteration kt wait until only tile kt+1's DMA group is outstanding (counted
    // vmcnt), barrier (tile kt visible to all waves; all waves are past compute(kt-1), whose buffer tile kt+2 reuses),
    // issue tile kt+2, compute tile kt.
    constexpr int B_EXTRA = B_INSTR % NW;   // waves [0, B_EXTRA) issue one more B instruction per tile
    constexpr int A_EXTRA = A_INSTR % NW;
    static_assert(A_EXTRA == 0, "A rows must divide evenly over the waves");
    constexpr int PER_TILE_LO = A_PW + B_INSTR / NW;
    if (!LDR || is_loader) {
      if (nk > 0) issue_tile(0);
      if (nk > 1) issue_tile(1);
    }
    if (TR) ln_prologue();      // (its loads are waited for inside: both tiles land with them; the first counted wait below is then free)
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (!LDR || is_loader) {
        if (kt + 1 < nk) {
          if (B_EXTRA != 0 && wave < B_EXTRA) ea_wait_dma<PER_TILE_LO + 1>();
          else ea_wait_dma<PER_TILE_LO>();
        } else {
          ea_wait_dma<0>();
        }
      }
      ea_raw_barrier();
      if (LDR) {
        if (is_loader) {
          if (kt + 2 < nk) issue_tile(cur >= 1 ? cur - 1 : 2);   // (cur + 2) % 3
        } else {
          compute_tile(cur);
        }
      } else if (ILV == 1) {
        const bool more = kt + 2 < nk;
        if (more) begin_issue(cur >= 1 ? cur - 1 : 2);   // (cur + 2) % 3
        compute_tile(cur, more);
        if (more) end_issue();
      } else {
        if (kt + 2 < nk) issue_tile(cur >= 1 ? cur - 1 : 2);
        if (TR && MT == 16 && p.acc_scale_kt > 0 && kt_begin + kt == p.acc_scale_kt) {      // K-concatenated split operands: see the 2-stage loop
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[MT == 16 ? i : 0][MT == 16 ? j : 0][r] *= p.acc_scale;
        }
        compute_tile(cur);
      }
      cur = (cur == 2) ? 0 : cur + 1;
    }
  }

  // ------------------------------------------------------------- epilogue
  // Each wave turns its own WTM x WTN accumulator tile into coalesced 16-byte global accesses through a private fp32
  // LDS slab (SLAB rows at a time), with no workgroup barriers after the first one.  Per slab every lane first issues
  // ALL its global reads (residual / time-embedding row vector) and only then does the math and the stores, so the
  // memory latency of the epilogue is paid once per slab, not once per output vector (a per-vector load -> use -> store
  // chain measured 25 us per [32768 x 320] launch -- more than the whole K loop of the K = 320 linears).
  const EaEpilogue& e = p.epi;
  if (EA_DBG(1)) {   // ablation: keep the accumulators live, write (almost) nothing
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) sum += (MT == 16) ? acc[MT == 16 ? i : 0][MT == 16 ? j : 0][0] : acc32[MT == 32 ? i : 0][MT == 32 ? j : 0][0];
    if (sum == 123456.789f) ((f16*)e.out)[0] = (f16)sum;
    return;
  }
  if (TR) {
    // ---- register-direct epilogue (p.epi_fast == 1 launches: fp16 out, optional fp16 residual, bias / per-sample row
    // vector / SiLU / GELU / scalar scale, 16-byte aligned, no split-K; p.epi_fast == 3: GEGLU with 32-row packing;
    // checked on the host).
    // acc[i][j][r] = C[row i*16 + c16][col j*16 + 4*q4 + r].  Tiles are paired -- (j, j+1) along the columns, and when
    // NI is odd the last column tile along the rows, (i, i+1) -- and each register pair goes through ea_swap16: the
    // even-q4 lanes end up with columns 8*(q4/2) .. +7 of the pair's FIRST tile, the odd-q4 lanes with the same columns
    // of its SECOND tile.  Per wave instruction: 16 rows x 64 contiguous bytes (column pairs).
    // Optional, both in fp32 on the values about to be rounded to fp16:
    //  * LayerNorm FOLD (e.ln_stats): the A operand is the UN-normalised activation and W carries gamma, so
    //    LN(x) W^T + b = rstd_m * (acc - mean_m * colsum_n) + (W beta + b)_n with mean / rstd from the row partials the
    //    producing launch left behind (attention.py:271-275: norm -> to_q / GEGLU proj) -- no LayerNorm pass at all;
    //  * ROW STATISTICS out (e.row_stats_out): per output row the (sum, sum of squares) over this wave's WTN columns,
    //    part = (column of the wave tile) / WTN -- what the next launch's LayerNorm fold consumes.
    if constexpr (TR != 0) {
      EA_STAMP(2);
      if constexpr (KS > 1) {
        // the K streams meet: group 1's accumulators go through LDS (the stage rings are free after the barrier; one 16-byte
        // slot per lane and register quad, lane-linear: conflict-free), group 0 adds them in and owns the epilogue
        constexpr int RED_BYTES = MI * NI * NW * 64 * 16;
        static_assert(RED_BYTES + NW * 288 <= KS * STAGES * (BM + BN) * 128, "reduction slots + GroupNorm bins must fit the rings");
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        __syncthreads();
        if (grp == 1) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) red[((i * NI + j) * NW + wave) * 64 + lane] = acc[MT == 16 ? i : 0][MT == 16 ? j : 0];
        }
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[MT == 16 ? i : 0][MT == 16 ? j : 0] += red[((i * NI + j) * NW + wave) * 64 + lane];
        ea_tr_epilogue<MI, NI, TR, false>(p, acc, m0 + wm * WTM, n0 + wn * WTN, m0, batch, bz, ln_mu, ln_rs, smem + RED_BYTES, wave);
      } else {
        ea_tr_epilogue<MI, NI, TR, true>(p, acc, m0 + wm * WTM, n0 + wn * WTN, m0, batch, bz, ln_mu, ln_rs, smem, wave);
      }
      EA_STAMP(4);
    }
    return;
  }
  constexpr int SLAB = (WTN > 80) ? 16 : 32;   // rows per slab: keeps the per-lane gather depth at <= 6 vectors
  constexpr int SLD = WTN + 4;                 // fp32 words per slab row (pad: conflict-free accumulator scatter)
  constexpr int NSLAB = WTM / SLAB;
  constexpr int ITERS = 6, SUB = 3;             // vectors per lane per slab (upper bound), gathered SUB at a time
  static_assert(NW * SLAB * SLD * 4 <= STAGES * STAGE_BYTES, "epilogue slabs must fit in the stage ring");
  static_assert(WTM % SLAB == 0, "slab rows");
  float* wstg = reinterpret_cast<float*>(smem) + wave * (SLAB * SLD);
  const bool raw = p.splits > 1;
  const bool geglu = (!raw) && e.act == EA_ACT_GEGLU;
  const int out_w = geglu ? WTN / 2 : WTN;   // output columns this wave produces
  const int vpr = out_w / 8;                 // 8-wide vectors per output row
  const int rpp = 64 / vpr;                  // rows per pass of the wave
  const int lc = lane % vpr, lr = lane / vpr;
  const bool lane_on = lr < rpp;
  // staging column(s) and global column of this lane's vector.  GEGLU: weight rows are packed [40 value | 40 gate]
  // per 80 (geglu_block), so value and gate of one output live in the same wave's slab.
  int scol = lc * 8, gcol = 0;
  if (geglu) {
    const int q = lc * 8, g = q / 40;
    scol = g * 80 + (q - g * 40);
    gcol = scol + 40;
  }
  const int nbase = geglu ? (n0 + wn * WTN) / 2 : n0 + wn * WTN;
  const int n = nbase + lc * 8;
  const int Nout = raw ? p.N : e.N;
  const bool col_on = lane_on && n < Nout;
  const long long cbase = (long long)batch * p.strideC, rbase = (long long)batch * p.strideR;
  // fully vectorisable launch? (uniform)  otherwise every vector goes through the generic scalar-capable helper
  const bool fast = raw ? ((p.N & 7) == 0)
                        : ((e.N & 7) == 0 && (e.ldc & 7) == 0 && (cbase & 7) == 0 &&
                           (!(e.residual || e.residual32) || ((e.ldr & 7) == 0 && (rbase & 7) == 0)) &&
                           (!e.rowvec || ((e.rowvec_ld & 3) == 0 && (((uintptr_t)e.rowvec) & 15) == 0)));
  float bias8[8], biasg8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias8[j] = 0.0f; biasg8[j] = 0.0f; }
  if (!raw && e.bias && !e.bias_per_row && col_on) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (geglu) {
        bias8[j] = e.bias[n0 + wn * WTN + scol + j];
        biasg8[j] = e.bias[n0 + wn * WTN + gcol + j];
      } else if (n + j < Nout) {
        bias8[j] = e.bias[n + j];
      }
    }
  }

  EA_STAMP(2);
  __syncthreads();  // every wave is done reading the K-loop stages: the ring becomes slab memory
  if (is_loader) return;   // no workgroup barrier follows: the epilogue is wave-local
  EA_STAMP(3);
  // ---- streamlined epilogue for the common launch (fp16 out, optional fp16 residual, bias / per-sample row vector /
  // SiLU / GELU / scalar scale, everything 16-byte aligned, no split-K; checked on the host -> p.epi_fast).  The general
  // path below evaluates every option per output vector behind wave-uniform branches and waits on each vector's LDS
  // reads in turn (measured ~3 us per 32-row slab per wave, as much as a 5-tile K loop).  Here the per-COLUMN terms
  // (bias + row vector) and the activation are applied while the accumulators are scattered to the LDS slab, and the
  // gather side is a fully unrolled {all LDS reads + all residual loads} -> {add, convert, 16-byte store} sequence.
  // ---- streamlined GEGLU epilogue (p.epi_fast == 2; attention.py:54-56 `x, gate = proj(x).chunk(2); x * gelu(gate)`).
  // Weight rows are packed [40 value | 40 gate] per 80, so one wave's 64 x 80 accumulator tile holds value and gate of
  // the same 40 outputs.  bias + GELU(gate) are applied while the accumulators are scattered to the wave's LDS slab
  // (each element once, in the MFMA layout: the gate test is per lane, no divergence); the gather side reads a value
  // vector and its gate vector, multiplies and writes 16 bytes.  The general path below spends 74 us of a 131-us
  // [32768 x 2560 x 320] launch in its per-vector wait chains (tools/gemm_bench --debug 0,1).
  if (MT == 16 && WTN == 80 && p.epi_fast == 2) {
    constexpr int SLABG = 16;
    constexpr int NSLABG = WTM / SLABG;
    constexpr int SLDG = WTN + 4;
    constexpr int VPRG = 5;                                  // 16-byte output vectors per row (40 outputs)
    constexpr int NVG = (SLABG * VPRG + 63) / 64;            // 2
    static_assert(NW * SLABG * SLDG * 4 <= STAGES * STAGE_BYTES, "epilogue slabs must fit in the stage ring");
    float* wst = reinterpret_cast<float*>(smem) + wave * (SLABG * SLDG);
    const int colbase = n0 + wn * WTN;                       // packed weight row of this wave's first column
    const int obase = colbase >> 1;                          // first output column
    float cb[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = colbase + j * 16 + frow;
      cb[j] = (e.bias && col < p.N) ? e.bias[col] : 0.0f;
    }
    f16* outp = (f16*)e.out + (long long)batch * p.strideC;
#pragma unroll
    for (int slab = 0; slab < NSLABG; ++slab) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const bool gate = (j * 16 + frow) >= 40;             // compile-time for j != 2
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[MT == 16 ? (MI > slab ? slab : 0) : 0][MT == 16 ? j : 0][r] + cb[j];
          if (j >= 2) { const float g = ea_gelu_erf(x); x = gate ? g : x; }
          wst[(fq * 4 + r) * SLDG + j * 16 + frow] = x;
        }
      }
      ea_wave_lds_sync();
      const int mrow0 = m0 + wm * WTM + slab * SLABG;
#pragma unroll
      for (int v = 0; v < NVG; ++v) {
        const int id = lane + 64 * v;
        const int row = id / VPRG, c = id - row * VPRG;
        const int m = mrow0 + row, n = obase + c * 8;
        if (id < SLABG * VPRG && m < p.M && n < e.N) {
          const float* sp = wst + row * SLDG + c * 8;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(sp + 40), g1 = *reinterpret_cast<const f32x4*>(sp + 44);
          f16x8 h;
#pragma unroll
          for (int q = 0; q < 4; ++q) { h[q] = (f16)(v0[q] * g0[q] * e.scale); h[4 + q] = (f16)(v1[q] * g1[q] * e.scale); }
          ea_st8(outp + (long long)m * e.ldc + n, h);
        }
      }
      ea_wave_lds_sync();
    }
    EA_STAMP(4);
    return;
  }
  if (MT == 16 && p.epi_fast == 1) {
    constexpr int SLABF = 16;   // rows per slab: 3 output vectors per lane in flight (32 rows spill the 128x160 kernel)
    constexpr int NSLABF = WTM / SLABF;
    constexpr int SLDF = WTN + 4;
    constexpr int VPR = WTN / 8;                            // 16-byte output vectors per row
    constexpr int NV = (SLABF * VPR + 63) / 64;             // vectors per lane per slab
    static_assert(NW * SLABF * SLDF * 4 <= STAGES * STAGE_BYTES, "epilogue slabs must fit in the stage ring");
    float* wst = reinterpret_cast<float*>(smem) + wave * (SLABF * SLDF);
    const int colbase = n0 + wn * WTN;
    const float* rvp = e.rowvec ? e.rowvec + (long long)(m0 / e.rows_per_group) * e.rowvec_ld : nullptr;
    float cb[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = colbase + j * 16 + frow;
      cb[j] = 0.0f;
      if (col < e.N) cb[j] = (e.bias ? e.bias[col] : 0.0f) + (rvp ? rvp[col] : 0.0f);
    }
    const long long cb0 = (long long)batch * p.strideC, rb0 = (long long)batch * p.strideR;
    f16* outp = (f16*)e.out + cb0;
    const f16* resp = e.residual ? e.residual + rb0 : nullptr;
    // Residual rows are fetched up to RD slabs ahead: with the load inside the slab body every slab paid a full memory
    // round trip (all waves of the chip are in their epilogues at once, nothing else hides it) -- 4 slabs x ~1.5 us of
    // a 15-us epilogue on the [32768 x 320] residual launches.
    constexpr int RD = NSLABF < 4 ? NSLABF : 4;
    f16x8 rq[RD][NV];
    auto res_load = [&](int slab_, int slot_) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int id = lane + 64 * v;
        const int row = id / VPR, c = id - row * VPR;
        const int m = m0 + wm * WTM + slab_ * SLABF + row, n = colbase + c * 8;
        if (id < SLABF * VPR && m < p.M && n < e.N) rq[slot_][v] = ea_ld8(resp + (long long)m * e.ldr + n);
      }
    };
    if (resp) {
#pragma unroll
      for (int s_ = 0; s_ < RD; ++s_) res_load(s_, s_);
    }
    // fully unrolled over the slabs: with a runtime slab index the compiler hoists the (slab-invariant) bias /
    // activation arithmetic of ALL accumulators out of the loop and spills
#pragma unroll
    for (int slab = 0; slab < NSLABF; ++slab) {
      constexpr int TPS = SLABF / 16;
#pragma unroll
      for (int ii = 0; ii < (MT == 16 ? MI : 1); ++ii) {
        if (ii / TPS != slab) continue;
        const int il = ii % TPS;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[MT == 16 ? ii : 0][MT == 16 ? j : 0][r] + cb[j];
            if (e.act == EA_ACT_SILU) x = ea_silu(x);
            else if (e.act == EA_ACT_GELU) x = ea_gelu_erf(x);
            wst[(il * 16 + fq * 4 + r) * SLDF + j * 16 + frow] = x * e.scale;
          }
      }
      ea_wave_lds_sync();
      const int mrow0 = m0 + wm * WTM + slab * SLABF;
      f32x4 lo[NV], hi[NV];
      int off[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int id = lane + 64 * v;
        const int row = id / VPR, c = id - row * VPR;
        const int m = mrow0 + row, n = colbase + c * 8;
        const bool ok = id < SLABF * VPR && m < p.M && n < e.N;
        off[v] = ok ? m * e.ldc + n : -1;
        const float* sp = wst + (ok ? row * SLDF + c * 8 : 0);
        lo[v] = *reinterpret_cast<const f32x4*>(sp);
        hi[v] = *reinterpret_cast<const f32x4*>(sp + 4);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (off[v] < 0) continue;
        f16x8 h;
        if (resp) {
          const f16x8 rr = rq[slab % RD][v];
#pragma unroll
          for (int j = 0; j < 4; ++j) { h[j] = (f16)(lo[v][j] + (float)rr[j]); h[4 + j] = (f16)(hi[v][j] + (float)rr[4 + j]); }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { h[j] = (f16)lo[v][j]; h[4 + j] = (f16)hi[v][j]; }
        }
#if (EA_EXP & 8) && !defined(EA_EMU)
        __builtin_nontemporal_store(h, reinterpret_cast<f16x8*>(outp + off[v]));   // experiment: streaming output stores
#else
        ea_st8(outp + off[v], h);
#endif
      }
      if (resp && slab + RD < NSLABF) res_load(slab + RD, slab % RD);
      ea_wave_lds_sync();
    }
    EA_STAMP(4);
    return;
  }

#pragma unroll 1
  for (int slab = 0; slab < NSLAB; ++slab) {
    // ---- scatter this slab's accumulators (MFMA C layout) into the wave's LDS slab
    if (MT == 16) {
      constexpr int TPS = SLAB / 16;  // 16-row MFMA tiles per slab
#pragma unroll
      for (int ii = 0; ii < (MT == 16 ? MI : 1); ++ii) {
        if (ii / TPS != slab) continue;
        const int il = ii % TPS;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            wstg[(il * 16 + fq * 4 + r) * SLD + j * 16 + frow] = acc[MT == 16 ? ii : 0][MT == 16 ? j : 0][r];
      }
    } else {
      constexpr int SPT = 32 / SLAB;  // slabs per 32-row MFMA tile (1 or 2)
#pragma unroll
      for (int ii = 0; ii < (MT == 32 ? MI : 1); ++ii) {
        if (ii != slab / SPT) continue;
        const int part = slab % SPT;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (SPT == 2 && (r >> 3) != part) continue;
            const int row = (r & 3) + 8 * ((r >> 2) & (SPT == 2 ? 1 : 3)) + 4 * fq;
            wstg[row * SLD + j * 32 + frow] = acc32[MT == 32 ? ii : 0][MT == 32 ? j : 0][r];
          }
      }
    }
    ea_wave_lds_sync();
    if (slab == 0) EA_STAMP(5);
    const int mrow0 = m0 + wm * WTM + slab * SLAB;
    if (!fast) {
      // launches that cannot use 16-byte vectors (ragged N, odd strides): one vector at a time through the
      // scalar-capable helper; not unrolled, so it does not inflate the hot path's register allocation.
#pragma unroll 1
      for (int row = lr; row < SLAB; row += rpp) {
        const int m = mrow0 + row;
        if (!(col_on && m < p.M)) continue;
        const float* sp = wstg + row * SLD;
        float v1[8];
        if (geglu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v1[j] = (sp[scol + j] + bias8[j]) * ea_gelu_erf(sp[gcol + j] + biasg8[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v1[j] = sp[scol + j];
        }
        if (raw) {
          float* dst = p.partial + ((long long)bz * p.M + m) * p.N + n;
          const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
          for (int j = 0; j < nvalid; ++j) dst[j] = v1[j];
        } else {
          ea_epilogue_store8(e, cbase, rbase, m, n, v1, !geglu);
        }
      }
      ea_wave_lds_sync();
      continue;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < ITERS; k0 += SUB) {
      // ---- phase A: gather (slab reads + every global read this lane needs for SUB vectors), no dependent use yet
      float v[SUB][8];
      f16x8 r16[SUB];
      f32x4 aux[SUB][2];   // fp32 residual, else the row vector (both at once: the row vector is read in phase B)
      bool ok[SUB];
#pragma unroll
      for (int kk = 0; kk < SUB; ++kk) {
        const int row = lr + (k0 + kk) * rpp;
        const int m = mrow0 + row;
        ok[kk] = col_on && row < SLAB && m < p.M;
        if (!ok[kk]) continue;
        const float* sp = wstg + row * SLD;
        if (geglu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[kk][j] = (sp[scol + j] + bias8[j]) * ea_gelu_erf(sp[gcol + j] + biasg8[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[kk][j] = sp[scol + j];
        }
        if (!raw) {
          const long long roff = rbase + (long long)m * e.ldr + n;
          if (e.residual) r16[kk] = ea_ld8(e.residual + roff);
          if (e.residual32) {
            aux[kk][0] = *reinterpret_cast<const f32x4*>(e.residual32 + roff);
            aux[kk][1] = *reinterpret_cast<const f32x4*>(e.residual32 + roff + 4);
          } else if (e.rowvec) {
            const float* rp = e.rowvec + (long long)(m / e.rows_per_group) * e.rowvec_ld + n;
            aux[kk][0] = *reinterpret_cast<const f32x4*>(rp);
            aux[kk][1] = *reinterpret_cast<const f32x4*>(rp + 4);
          }
        }
      }
      // ---- phase B: math + stores
#pragma unroll
      for (int kk = 0; kk < SUB; ++kk) {
        if (!ok[kk]) continue;
        const int m = mrow0 + lr + (k0 + kk) * rpp;
        if (raw) {
          float* dst = p.partial + ((long long)bz * p.M + m) * p.N + n;
          f32x4 lo = {v[kk][0], v[kk][1], v[kk][2], v[kk][3]}, hi = {v[kk][4], v[kk][5], v[kk][6], v[kk][7]};
          *reinterpret_cast<f32x4*>(dst) = lo;
          *reinterpret_cast<f32x4*>(dst + 4) = hi;
          continue;
        }
        if (!geglu) {
          if (e.bias_per_row) {
            const float b = e.bias ? e.bias[m] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[kk][j] += b;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[kk][j] += bias8[j];
          }
          if (e.rowvec) {
            if (e.residual32) {   // rare: both fp32 residual and row vector -> the row vector was not prefetched
              const float* rp = e.rowvec + (long long)(m / e.rows_per_group) * e.rowvec_ld + n;
#pragma unroll
              for (int j = 0; j < 8; ++j) v[kk][j] += rp[j];
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[kk][j] += aux[kk][0][j]; v[kk][4 + j] += aux[kk][1][j]; }
            }
          }
          if (e.act == EA_ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[kk][j] = ea_silu(v[kk][j]);
          } else if (e.act == EA_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[kk][j] = ea_gelu_erf(v[kk][j]);
          }
        }
        float sc = e.scale;
        if (e.row_scale) sc *= e.row_scale[m];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[kk][j] *= sc;
        if (e.residual) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[kk][j] += (float)r16[kk][j];
        }
        if (e.residual32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[kk][j] += aux[kk][0][j]; v[kk][4 + j] += aux[kk][1][j]; }
        }
        const long long coff = cbase + (long long)m * e.ldc + n;
        if (e.out_f32) {
          float* o = (float*)e.out + coff;
          f32x4 lo = {v[kk][0], v[kk][1], v[kk][2], v[kk][3]}, hi = {v[kk][4], v[kk][5], v[kk][6], v[kk][7]};
          *reinterpret_cast<f32x4*>(o) = lo;
          *reinterpret_cast<f32x4*>(o + 4) = hi;
        } else {
          f16x8 h;
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = (f16)v[kk][j];
          ea_st8((f16*)e.out + coff, h);
        }
      }
    }
    ea_wave_lds_sync();  // slab reads retired before the next slab's scatter overwrites it
    if (slab == 0) EA_STAMP(6);
  }
  EA_STAMP(4);
}

template <int BM, int BN, int WM, int WN, int STAGES, int MT, int ILV, int LDR = 0, int TR = 0>
__global__ __launch_bounds__(WM* WN * 64 * (1 + LDR), (LDR ? (STAGES == 2 && BM == 64 ? 4 : 2) : ea_gemm2_occ(BM, BN, WM* WN, STAGES)))
void ea_gemm2_kernel(EaGemmParams p) {
  ea_gemm2_tile<BM, BN, WM, WN, STAGES, MT, ILV, LDR, TR>(p, blockIdx.x, blockIdx.z);
}

// intra-workgroup split-K (KS = 2, round-6 experiment): 8 waves, two stage rings, one workgroup per CU
template <int BM, int BN, int TR>
__global__ __launch_bounds__(512, 1)
void ea_gemm2_ks2_kernel(EaGemmParams p) {
  ea_gemm2_tile<BM, BN, 2, 2, 2, 16, 0, 0, TR, 2>(p, blockIdx.x, blockIdx.z);
}

// TWIN launch: two problems of ONE shape and plan (same M, N, K, tile plan, split-K factor, epilogue form -- the host
// checks, ea_gemm.hip launch_pair) in one grid, blockIdx.y = problem.  The ControlNet trunk is a copy of the UNet encoder
// (cldm/cldm.py:284-305 vs :22-45: identical layers on identical shapes), so every contraction of one has a twin in the
// other: as one grid the pair fills twice the workgroup slots exactly where M is smallest (16 x 16 / 8 x 8 latents: 80-320
// workgroups on 512 slots), deterministically, instead of two streams packing into each other when the timing allows.
// The problem is picked by a wave-uniform select between the two kernel-argument blocks: every field stays a scalar load.
template <int BM, int BN, int WM, int WN, int STAGES, int MT, int ILV, int LDR = 0, int TR = 0>
__global__ __launch_bounds__(WM* WN * 64 * (1 + LDR), (LDR ? (STAGES == 2 && BM == 64 ? 4 : 2) : ea_gemm2_occ(BM, BN, WM* WN, STAGES)))
void ea_gemm2_pair_kernel(EaGemmParams p0, EaGemmParams p1) {
  const EaGemmParams& p = blockIdx.y ? p1 : p0;
  ea_gemm2_tile<BM, BN, WM, WN, STAGES, MT, ILV, LDR, TR>(p, blockIdx.x, blockIdx.z);
}
