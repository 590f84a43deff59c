// ea_gemm3.h -- the PERSISTENT contraction kernel (round 3): one 8-wave workgroup per CU walks a list of output tiles,
// its LDS-DMA stream running up to three K tiles ahead of the MFMAs and straight across tile boundaries.
//
// Same contract as ea_gemm2.h (C = epilogue(A W^T), A dense or the implicit im2col of an NHWC activation, K % 64 == 0,
// channel counts multiples of 64), for the launches whose epilogue is the register-direct one (ea_epi_tr.h): the
// ResBlock / Up / Downsample convolutions (openaimodel.py:108-152,200-231,254-274), the transformer Linears
// (attention.py:54,152-160,263-275,316-339), the ControlNet zero-convs (cldm/cldm.py:281-305).
//
// Why a new kernel (measurements of rounds 1-2 in section 8c/8d):
//  * ea_gemm2's workgroups are one-shot: every launch pays the first DMA round trip (an HBM miss for weights: 1-2 us)
//    before its first MFMA and ends in a store burst with nothing in flight behind it -- prologue + epilogue were ~45 %
//    of the contraction time of an evaluation.  Here a workgroup's DMA stream is ONE sequence of K tiles over all the
//    output tiles it owns: while tile t's accumulators go through the epilogue, the first K tiles of tile t + 1 are
//    already in LDS or in flight.
//  * its 2-stage ring keeps ONE K tile in flight per workgroup and leaves the pipe empty between the wait and the next
//    issue; with an HBM-cold weight stream (the 16x16 / 8x8 levels) an iteration costs the memory round trip (1.2-1.4 us
//    per 64-deep K tile against 0.27 us of MFMA work).  Here the ring has four stages (144 KiB of the CU's 160 KiB) and
//    counted vmcnt: two K tiles are in flight behind the one being multiplied and the one landed ahead of it.
//  * 8 waves on ONE tile instead of 4 + 4 on two: the same LDS bytes per flop as ea_gemm2's wave tiles (KS = 2, below)
//    but one set of operands per CU -- twice the ring depth out of the same LDS.
//
// Wave roles (512 threads, 2 waves per SIMD, <= 256 registers each), tile 128 x BN (BN = 160 / 128), K tile 64:
//   KS = 2  "k-split": two groups of four waves (2 x 2, wave tile 64 x BN/2, like ea_gemm2); group g multiplies K step g
//           (32 of the 64) of every K tile.  Per K tile and wave: 9 ds_read_b128 + 20 MFMAs (16x16x32), fragments of the
//           NEXT K tile read under this one's MFMAs.  At the end of an output tile the groups swap half of their partial
//           accumulators through the ring slot just multiplied (group 0 ends up with rows 0..31 of each wave tile,
//           group 1 with rows 32..63) and all 8 waves run the register-direct epilogue on 32 x BN/2 each.
//   KS = 1  "m-split": 4 x 2 waves, wave tile 32 x BN/2, both K steps -- no exchange (4 barriers less per output tile);
//           more LDS reads per flop.  For short K (a handful of K tiles per output tile), where the exchange would cost
//           more than the reads.
// Tile order: work item = (split-K slice, tile); items are dealt to the workgroups round by round, each XCD getting a
// contiguous run of a GROUPED order (raster_gm rows of tiles x all columns, column-major inside a group) so that the
// tiles resident on one XCD at a time share A row panels and W column panels in its private L2 (guide T1).
#pragma once
#include "ea_prims.h"
#include "ea_epi_tr.h"

#define EA_G3_STAGES 4
// -DEA_G3_PROF=1 (tools/g3_prof, never in the product): per-wave cycle totals of the loop's phases (s_memtime) into the
// workspace -- 0 counted wait, 1 barrier, 2 early issue, 3 fragment reads + MFMAs, 4 late issue, 5 end of item, 6 opening.
#ifndef EA_G3_PROF
#define EA_G3_PROF 0
#endif
// -DEA_G3_ABL=mask (side builds for tools/g3_prof only; results are wrong by construction): 1 no MFMAs, 2 no fragment
// reads, 4 no per-K-tile barrier, 8 no DMA after the prologue
#ifndef EA_G3_ABL
#define EA_G3_ABL 0
#endif
#if EA_G3_PROF && !defined(EA_EMU)
#define G3T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); g3t[i] += t_ - g3last; g3last = t_; } while (0)
#else
#define G3T(i) do {} while (0)
#endif
#define EA_G3_SPARE 4096   // LDS behind the ring: GroupNorm-statistics bins of the epilogue (no DMA ever targets it)

constexpr int ea_gemm3_lds_bytes(int bn) { return EA_G3_STAGES * (128 + bn) * 128 + 2 * EA_G3_SPARE; }   // + the exchange overflow

template <int BN, int TRX, int KS>
__global__ __launch_bounds__(KS == 0 ? 768 : 512) void ea_gemm3_kernel(EaGemmParams p) {
  constexpr int BM = 128, NW = 8;                  // NW: waves that ISSUE the LDS-DMA (KS = 0: the eight loader waves)
  constexpr int WTN = BN / 2, NI = WTN / 16;
  constexpr int MI = (KS == 1) ? 2 : 4;            // accumulator row tiles per wave (m-split: 32 rows, else 64)
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;   // 1-KiB LDS-DMA instructions per K tile
  constexpr int A_PW = A_INSTR / NW;                   // 2
  constexpr int B_PW = (B_INSTR + NW - 1) / NW;        // 3 (BN = 160: waves 0..3 issue the third) / 2
  constexpr int B_EXTRA = B_INSTR % NW;                // waves [0, B_EXTRA) issue B_PW pieces, the rest B_PW - 1 (0: all B_PW)
  constexpr int P_HI = A_PW + B_PW, P_LO = A_PW + (B_EXTRA ? B_PW - 1 : B_PW);
  static_assert(A_INSTR % NW == 0, "A rows divide evenly over the waves");
  static_assert(KS >= 0 && KS <= 2, "wave roles");
  EA_SMEM(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = ea_uniform(tid >> 6);
  // KS = 0: waves 0..3 multiply (2 x 2, wave tile 64 x BN/2, both K steps), waves 4..11 only issue the LDS-DMA
  const bool is_loader = KS == 0 && wave_all >= 4;
  const int wave = is_loader ? wave_all - 4 : wave_all;        // index among the issuing waves / among the multiplying waves
  const int grp = (KS == 2) ? (wave >> 2) : 0;                 // k-split group
  const int wm = (KS == 1) ? (wave >> 1) : ((wave & 3) >> 1);  // wave row: 32-row (m-split) / 64-row units
  const int wn = wave & 1;
  const bool p_hi = B_EXTRA != 0 && wave < B_EXTRA;            // this (issuing) wave issues P_HI pieces per K tile

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nitems = tiles_m * tiles_n * p.splits;
  const int G = gridDim.x;
  const int nk_total = p.K / EA_BK;
  const int ctot = p.c1 + p.c2;
  const ea_rsrc rs_a1 = ea_make_rsrc(p.a1);
  const ea_rsrc rs_a2 = ea_make_rsrc(p.a2 ? p.a2 : p.a1);
  const ea_rsrc rs_w = ea_make_rsrc(p.w);

  // item of this workgroup in round r: ids [r*G, r*G + n_r) are dealt so that each XCD holds a contiguous run
  auto my_item = [&](int round, int& m0, int& n0, int& split, int& kt0, int& nk) -> bool {
    const int base = round * G;
    int n_r = nitems - base;
    if (n_r > G) n_r = G;
    if ((int)blockIdx.x >= n_r) return false;
    const int id = base + ea_xcd_remap(blockIdx.x, n_r);
    int tm, tn;
    ea_grouped_item(id, tiles_m, tiles_n, p.raster_gm, tm, tn, split);
    tm = ea_uniform(tm); tn = ea_uniform(tn); split = ea_uniform(split);
    m0 = tm * BM;
    n0 = tn * BN;
    kt0 = split * p.ktiles_per_split;
    int kt1 = kt0 + p.ktiles_per_split;
    if (kt1 > nk_total) kt1 = nk_total;
    nk = kt1 - kt0;
    return true;
  };
  const int nrounds = (nitems + G - 1) / G;

  // ------------------------------------------------------------------------------------------------ issue side
  // The DMA stream: K tile after K tile, item after item, into ring slot (job index) % 4.
  const int lrow = lane >> 3, slot8 = lane & 7;
  unsigned a_chunk[A_PW];
  int a_y[A_PW], a_x[A_PW], a_base[A_PW];
  unsigned a_voff[A_PW], b_voff[B_PW];
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int r = (j * NW + wave) * 8 + lrow;
    a_chunk[j] = (unsigned)((slot8 ^ ea_swz(r)) * 8);
    a_y[j] = a_x[j] = 0; a_base[j] = -1; a_voff[j] = EA_OOB;
  }
#pragma unroll
  for (int j = 0; j < B_PW; ++j) b_voff[j] = EA_OOB;
  int is_round = 0, is_left = 0;          // next round to open; K tiles left in the open item
  int is_kcur = 0, is_tap = 0, is_cin = 0;
  int issued = 0;                         // jobs issued so far (= index of the next job)
  auto set_voff = [&]() {
    const int ky = (p.ksize == 3) ? is_tap / 3 : 0;
    const int kx = (p.ksize == 3) ? is_tap - ky * 3 : 0;
    const int hlim = p.ups ? 2 * p.Hin : p.Hin;
    const int wlim = p.ups ? 2 * p.Win : p.Win;
    const unsigned cs = (unsigned)(is_cin >= p.c1 ? p.c2 : p.c1);
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      int iy = a_y[j] + ky, ix = a_x[j] + kx;
      const bool ok = a_base[j] >= 0 && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
      if (p.ups) { iy >>= 1; ix >>= 1; }
      a_voff[j] = ok ? ((unsigned)(a_base[j] + iy * p.Win + ix) * cs + a_chunk[j]) * 2u : EA_OOB;
    }
  };
  auto open_item = [&]() -> bool {        // per-lane coordinates of the next item; false = the stream is finished
    int m0, n0, split, kt0, nk;
    while (is_round < nrounds) {
      const bool ok = my_item(is_round, m0, n0, split, kt0, nk);
      ++is_round;
      if (!ok || nk <= 0) continue;
      is_left = nk;
      is_kcur = kt0 * EA_BK;
#pragma unroll
      for (int j = 0; j < A_PW; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        const int m = m0 + r;
        a_y[j] = a_x[j] = 0; a_base[j] = -1; a_voff[j] = EA_OOB;
        if (m < p.M) {
          if (p.conv) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.Wout;
            a_base[j] = b * p.Hin * p.Win;
            a_y[j] = oy * p.stride - p.pad;
            a_x[j] = (rem - oy * p.Wout) * p.stride - p.pad;
          } else {
            a_voff[j] = ((unsigned)m * (unsigned)p.lda + a_chunk[j]) * 2u;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < B_PW; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        const int n = n0 + r;
        b_voff[j] = (r < BN && n < p.N) ? ((unsigned)n * (unsigned)p.ldw + (unsigned)((slot8 ^ ea_swz(r)) * 8)) * 2u : EA_OOB;
      }
      if (p.conv) {
        is_tap = ea_uniform(is_kcur / ctot);
        is_cin = is_kcur - is_tap * ctot;
        set_voff();
      }
      return true;
    }
    return false;
  };
  auto issue_next = [&]() {
    if (is_left == 0 && !open_item()) return;
#if (EA_G3_ABL & 8) && !defined(EA_EMU)
    if (issued >= 3) { ++issued; --is_left; return; }
#endif
    char* sa = smem + (issued & (EA_G3_STAGES - 1)) * STAGE_BYTES;
    char* sb = sa + BM * 128;
    is_kcur = ea_uniform(is_kcur);     // loop-carried scalars: keep them provably wave-uniform (SGPR descriptors / offsets, T20)
    is_cin = ea_uniform(is_cin);
    is_tap = ea_uniform(is_tap);
    const bool second = p.conv && is_cin >= p.c1;
    const ea_rsrc rs_a = second ? rs_a2 : rs_a1;
    const unsigned soff_a = (unsigned)(p.conv ? (second ? is_cin - p.c1 : is_cin) : is_kcur) * 2u;
    const unsigned soff_b = (unsigned)is_kcur * 2u;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) ea_dma16(rs_a, a_voff[j], soff_a, sa + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < B_PW; ++j)
      if (B_EXTRA == 0 || j + 1 < B_PW || wave < B_EXTRA) ea_dma16(rs_w, b_voff[j], soff_b, sb + (j * NW + wave) * 1024);
    ++issued;
    --is_left;
    is_kcur += EA_BK;
    if (p.conv) {
      is_cin += EA_BK;
      if (is_cin >= ctot) {
        is_cin = 0;
        ++is_tap;
        if (is_left > 0) set_voff();
      } else if (is_cin == p.c1) {
        set_voff();
      }
    }
  };
  // wait until job `j` of this wave's DMA has landed: everything issued after it may stay in flight.  (Epilogue loads
  // and stores also sit in the vmcnt queue; counting only the DMA pieces makes the wait conservative, never early:
  // loads retire in order, and a count that ignores younger operations only waits for more.)
  auto wait_job = [&](int j) {
    const int younger = issued - 1 - j;
    if (younger >= 2) { if (p_hi) ea_wait_dma<2 * P_HI>(); else ea_wait_dma<2 * P_LO>(); }
    else if (younger == 1) { if (p_hi) ea_wait_dma<P_HI>(); else ea_wait_dma<P_LO>(); }
    else ea_wait_dma<0>();
  };

  // ---------------------------------------------------------------------------------------------- compute side
  const int frow = lane & 15, fq = lane >> 4;
#if (EA_G3_ABL & 2) && !defined(EA_EMU)
  int q_abl = -1;                         // ablation: fragment reads only before the first K tile
#endif
  f32x4 acc[MI][NI];
  f16x8 fa[2][MI], fb[2][NI];
  // fragments of K step `ks` of the K tile in ring slot `s` -> register set `set`
  auto read_frags = [&](int s, int ks, int set) {
#if (EA_G3_ABL & 2) && !defined(EA_EMU)
    if (q_abl >= 0) return;
#endif
    const char* sa = smem + s * STAGE_BYTES;
    const char* sb = sa + BM * 128;
    const int ch = ks * 4 + fq;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * (MI * 16) + i * 16 + frow;
      fa[set][i] = *reinterpret_cast<const f16x8*>(sa + r * 128 + ((ch ^ ea_swz(r)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * WTN + j * 16 + frow;
      fb[set][j] = *reinterpret_cast<const f16x8*>(sb + r * 128 + ((ch ^ ea_swz(r)) << 4));
    }
  };
  auto mfma_step = [&](int set) {
#if (EA_G3_ABL & 1) && !defined(EA_EMU)
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(fa[set][i]));
#pragma unroll
    for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(fb[set][j]));
#else
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ea_mfma_16x16x32(fb[set][j], fa[set][i], acc[i][j]);
#endif
  };
  // pin "one fragment read per MFMA (pair)" for a step whose reads (of the NEXT step's fragments, into the other register
  // set) and MFMAs share a basic block (guide T19).  Left alone hipcc sinks the reads BEHIND the MFMAs to reuse the
  // current fragments' registers -- and every K tile then pays the whole LDS round trip in front of the barrier, all 8
  // waves at once (first builds of this kernel: the loop WITHOUT any DMA ran at 36 % of the MFMA rate)
  auto pin_interleave = [&]() {
#ifndef EA_EMU
    constexpr int MF = MI * NI;
#pragma unroll
    for (int r = 0; r < MI + NI; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, (MF >= 2 * (MI + NI)) ? 2 : 1, 0);
    }
    if (MF > ((MF >= 2 * (MI + NI)) ? 2 : 1) * (MI + NI))
      __builtin_amdgcn_sched_group_barrier(0x008, MF - ((MF >= 2 * (MI + NI)) ? 2 : 1) * (MI + NI), 0);
#endif
  };

  // ------------------------------------------------------------------------------------------------- the stream
  // Jobs (K tiles) q = 0, 1, ... of this workgroup; iteration q multiplies job q and issues job q + 3:
  //   top of iteration q:  job q + 1 landed for this wave (counted vmcnt) -> barrier: landed for everyone, and every
  //                        wave has retired its fragment reads of job q - 1's slot (lgkmcnt(0) in the barrier)
  //   then:                job q + 3 is issued into that slot;  job q is multiplied, job q + 1's first fragments are read
  // Items are the outer loop so that the accumulators / fragment registers are provably dead across the epilogue (a flat
  // loop with "if (first K tile) zero / read" keeps 150 registers alive through it and spills).
  char* spare = smem + EA_G3_STAGES * STAGE_BYTES;
  if constexpr (KS == 0) {
    // ---------------------------------------------------------------------------------- loader-wave roles (KS = 0)
    // Measured on the MI355X (tools/probe_dma2/3, profiles/r03_probe_dma_*): ONE wave issues an LDS-DMA instruction
    // (1 KiB) every 75-125 cycles at best, whatever it has outstanding -- 8-13 B/clk per wave; a CU reaches 64 B/clk with
    // 8 waves issuing and 100-128 B/clk with 16.  A 128 x 160 x 64 K tile is 36 instructions: issued by the four waves
    // that also multiply it (ea_gemm2) it takes ~1000 cycles of each wave's time against 640 cycles of MFMA work, which
    // is where every earlier restructuring of that loop ended up.  Here eight EXTRA waves do nothing but issue (4-5
    // instructions each per K tile, ~450 cycles) and the four multiplying waves -- one per SIMD -- never touch the
    // memory pipe: barrier, fragment reads, 40 MFMAs.  One barrier per K tile for all twelve waves.
    bool any = false;
    {
      int m0_, n0_, sp_, kt0_, nk_;
      for (int r = 0; r < nrounds; ++r) any = any || (my_item(r, m0_, n0_, sp_, kt0_, nk_) && nk_ > 0);
    }
    if (!any) return;                       // (uniform over the workgroup)
    if (is_loader) {
#pragma unroll 1
      for (int i = 0; i < 3; ++i) issue_next();
      wait_job(0);
      ea_raw_barrier();
#pragma unroll 1
      for (int q = 0; q < issued; ++q) {
        wait_job(q + 1 < issued ? q + 1 : q);
        ea_raw_barrier();
        issue_next();
      }
      return;
    }
    ea_raw_barrier();                       // job 0 landed
    int q = 0;
#pragma unroll 1
    for (int round = 0; round < nrounds; ++round) {
      int m0, n0, split, kt0, nk;
      if (!my_item(round, m0, n0, split, kt0, nk) || nk <= 0) continue;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      read_frags(q & (EA_G3_STAGES - 1), 0, 0);
      const int emit_row0 = m0 + wm * 64;
      float ln_mu[MI], ln_rs[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) { ln_mu[i] = 0.0f; ln_rs[i] = 1.0f; }
      if (TRX == 2 && p.epi.ln_stats) {
        constexpr int CH = 8;
        float s1[MI], s2[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) { s1[i] = 0.0f; s2[i] = 0.0f; }
        for (int pp0 = 0; pp0 < p.epi.ln_parts; pp0 += CH) {
          f32x2 t2[CH][MI];
#pragma unroll
          for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              int m = emit_row0 + i * 16 + frow;
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
      }
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt, ++q) {
        ea_raw_barrier();                   // job q + 1 landed; every wave is past its reads of job q - 1's slot
        read_frags(q & (EA_G3_STAGES - 1), 1, 1);
        mfma_step(0);
        pin_interleave();
        read_frags((q + 1) & (EA_G3_STAGES - 1), 0, 0);
        mfma_step(1);
        pin_interleave();
      }
      ea_tr_epilogue<MI, NI, TRX, false>(p, acc, emit_row0, n0 + wn * WTN, m0, 0, split, ln_mu, ln_rs, spare, wave);
    }
    return;
  }
  constexpr int ME = 2;                   // row tiles a wave emits
#if EA_G3_PROF && !defined(EA_EMU)
  unsigned long long g3t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long g3last = __builtin_readcyclecounter();
  const unsigned long long g3start = g3last;
#endif
#pragma unroll 1
  for (int i = 0; i < 3; ++i) issue_next();
  if (issued == 0) return;                // nothing to do for this workgroup (uniform)
  wait_job(0);
  ea_raw_barrier();

  int q = 0;                              // job being multiplied
#if (EA_G3_ABL & 2) && !defined(EA_EMU)
  read_frags(0, 0, 0); read_frags(0, 1, 1); q_abl = 0;
#endif
#pragma unroll 1
  for (int round = 0; round < nrounds; ++round) {
    int m0, n0, split, kt0, nk;
    if (!my_item(round, m0, n0, split, kt0, nk) || nk <= 0) continue;
    // ---- open the item: accumulators, first fragments (the one exposed LDS round trip; job q was waited for and
    // barrier'd at the top of the previous iteration / in the prologue), LayerNorm-fold row terms
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    read_frags(q & (EA_G3_STAGES - 1), (KS == 2) ? grp : 0, 0);
    const int emit_row0 = m0 + ((KS == 2) ? wm * 64 + grp * 32 : wm * 32);
    float ln_mu[ME], ln_rs[ME];
#pragma unroll
    for (int i = 0; i < ME; ++i) { ln_mu[i] = 0.0f; ln_rs[i] = 1.0f; }
    if (TRX == 2 && p.epi.ln_stats) {
      // mean / rstd of the rows this wave will emit, from the producer's row partials (ea_epi_tr.h); unconditional
      // loads (clamped indices, arithmetic masks): a per-load `if` would serialise them (guide section 5 trap (c))
      constexpr int CH = 8;
      float s1[ME], s2[ME];
#pragma unroll
      for (int i = 0; i < ME; ++i) { s1[i] = 0.0f; s2[i] = 0.0f; }
      for (int pp0 = 0; pp0 < p.epi.ln_parts; pp0 += CH) {
        f32x2 t2[CH][ME];
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
          for (int i = 0; i < ME; ++i) {
            int m = emit_row0 + i * 16 + frow;
            m = m < p.M ? m : p.M - 1;
            int pp = pp0 + u;
            pp = pp < p.epi.ln_parts ? pp : p.epi.ln_parts - 1;
            t2[u][i] = *reinterpret_cast<const f32x2*>(p.epi.ln_stats + ((long long)pp * p.M + m) * 2);
          }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const float keep = (pp0 + u < p.epi.ln_parts) ? 1.0f : 0.0f;
#pragma unroll
          for (int i = 0; i < ME; ++i) { s1[i] += keep * t2[u][i][0]; s2[i] += keep * t2[u][i][1]; }
        }
      }
      const float inv = 1.0f / (float)p.K;
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const float mu = s1[i] * inv;
        ln_mu[i] = mu;
        ln_rs[i] = 1.0f / sqrtf(fmaxf(s2[i] * inv - mu * mu, 0.0f) + p.epi.ln_eps);
      }
    }

    G3T(6);
    // One K tile: top = job q + 1 landed for this wave (the stream's last job has no successor: wait for the job itself),
    // barrier, issue job q + 3; then the MFMAs of job q with the next fragments read under them.
    // k-split: this wave's K step of a job sits in ONE register set, the next job's goes into the other.  The set index
    // must be a compile-time constant (a runtime-indexed register array goes to scratch, guide rule 20) and the two
    // parities must not meet in a control-flow merge (hipcc then shuffles 72 fragment + 80 accumulator registers through
    // copies and spills): the K loop is unrolled by two, even K tiles on set 0, odd ones on set 1, an item always starts
    // on set 0.
    // The DMA issue of an iteration (~60 clk per 1-KiB piece on the CU's one address path) is STAGGERED between the two
    // halves of the workgroup: waves 0..3 issue before their MFMAs, waves 4..7 after theirs.  Every SIMD hosts one wave
    // of each half, so one half's issue burst runs beside the other half's MFMAs instead of all eight waves bursting and
    // then all multiplying (first GPU run of this kernel, all waves in phase: 30-55 % SLOWER than ea_gemm2, whose two
    // workgroups per CU de-phase by themselves).
    // The next fragments are read UNCONDITIONALLY (at the last K tile of the stream a stale slot: never used): a branch
    // around the reads splits the block and hipcc parks an s_waitcnt lgkmcnt(0) in front of the MFMAs -- the whole LDS
    // round trip exposed, every K tile (seen in the first build's ISA).
    const bool early = wave < 4;
    if constexpr (KS == 2) {
#pragma unroll 1
      for (int kt = 0; kt < nk; kt += 2) {
        wait_job(q + 1 < issued ? q + 1 : q);
        G3T(0);
        if (!(EA_G3_ABL & 4)) ea_raw_barrier();
        G3T(1);
        if (early) issue_next();
        G3T(2);
        read_frags((q + 1) & (EA_G3_STAGES - 1), grp, 1);
        mfma_step(0);
        pin_interleave();
        G3T(3);
        if (!early) issue_next();
        G3T(4);
        ++q;
        if (kt + 1 == nk) break;
        wait_job(q + 1 < issued ? q + 1 : q);
        G3T(0);
        if (!(EA_G3_ABL & 4)) ea_raw_barrier();
        G3T(1);
        if (early) issue_next();
        G3T(2);
        read_frags((q + 1) & (EA_G3_STAGES - 1), grp, 0);
        mfma_step(1);
        pin_interleave();
        G3T(3);
        if (!early) issue_next();
        G3T(4);
        ++q;
      }
    } else {
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt, ++q) {
        wait_job(q + 1 < issued ? q + 1 : q);
        G3T(0);
        if (!(EA_G3_ABL & 4)) ea_raw_barrier();
        G3T(1);
        if (early) issue_next();
        G3T(2);
        // K step 0 is in set 0: read step 1 under it; then the next job's step 0 under step 1
        read_frags(q & (EA_G3_STAGES - 1), 1, 1);
        mfma_step(0);
        pin_interleave();
        read_frags((q + 1) & (EA_G3_STAGES - 1), 0, 0);
        mfma_step(1);
        pin_interleave();
        G3T(3);
        if (!early) issue_next();
        G3T(4);
      }
    }

    // -------------------------------------------------------------------------------------------- end of the item
    const int colbase = n0 + wn * WTN;
    if constexpr (KS == 2) {
      // Exchange of partial accumulators between the two k-split groups through the ring slot of the item's last job
      // (q - 1 now: its fragments are in registers, and the job that reuses the slot is issued one iteration from now,
      // behind a barrier) plus 4 KiB behind the ring: round r moves row tile r of group 1 to group 0 (region X) and row
      // tile 2 + r of group 0 to group 1 (region Y); [column tile][wave][lane] x 16 bytes, conflict-free both ways.
      char* X = smem + ((q - 1) & (EA_G3_STAGES - 1)) * STAGE_BYTES;
      char* Y = X + NI * 4096;
      char* Y2 = spare + EA_G3_SPARE;       // what of Y does not fit behind X in the slot: the last column tile
      constexpr int YFIT = (STAGE_BYTES - NI * 4096) / 4096;   // column tiles of Y inside the slot (BN = 160: 4 of 5)
      static_assert(YFIT >= NI - 1, "exchange regions");
      const int loff = ((wave & 3) * 64 + lane) * 16;
      f32x4 fin[ME][NI];
      if (nk == 1) ea_raw_barrier();        // one-K-tile item: the slot's fragments were read after the last barrier
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const f32x4 give = grp ? acc[r][j] : acc[2 + r][j];
          char* dst = grp ? X + j * 4096 : (j < YFIT ? Y + j * 4096 : Y2 + (j - YFIT) * 4096);
          *reinterpret_cast<f32x4*>(dst + loff) = give;
        }
        ea_raw_barrier();
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const char* src = grp ? (j < YFIT ? Y + j * 4096 : Y2 + (j - YFIT) * 4096) : X + j * 4096;
          const f32x4 got = *reinterpret_cast<const f32x4*>(src + loff);
          fin[r][j] = (grp ? acc[2 + r][j] : acc[r][j]) + got;
        }
        ea_raw_barrier();                   // reads retired (lgkmcnt(0) inside) before the regions are rewritten / restaged
      }
      ea_tr_epilogue<ME, NI, TRX, false>(p, fin, emit_row0, colbase, m0, 0, split, ln_mu, ln_rs, spare, wave);
    } else if constexpr (KS == 1) {
      ea_tr_epilogue<ME, NI, TRX, false>(p, acc, emit_row0, colbase, m0, 0, split, ln_mu, ln_rs, spare, wave);
    }
    G3T(5);
  }
#if EA_G3_PROF && !defined(EA_EMU)
  if (lane == 0 && p.prof) {
    unsigned long long* o = p.prof + ((long long)blockIdx.x * 8 + wave) * 8;
    g3t[7] = g3last - g3start;
    for (int i = 0; i < 8; ++i) o[i] = g3t[i];
  }
#endif
}
