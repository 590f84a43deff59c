// ea_gemm.hip -- C-ABI launchers for the MFMA contraction kernel (ea_gemm.h).
#include "ea_gemm2.h"
#include "ea_gemm8.h"
#ifndef EA_TOOLS
#define EA_TOOLS 0
#endif
#ifndef EA_EXP
#define EA_EXP 0
#endif
#if EA_TOOLS
#include "../../tools/kernels/ea_gemm3.h"   // the persistent-kernel experiment of round 3 (measured slower; tools build only)
#endif
#include <stdlib.h>
#include "../../include/editanything_hip.h"

namespace {

struct TilePlan {
  int wide;    // 1: 128x128 tile, 0: 256x64 tile
  int bm, bn;
  int tiles;
  int splits;
  int ktiles_per_split;
};

static TilePlan plan_tiles(int M, int N, int K, int batch, int allow_split) {
  TilePlan t;
  const int pad128 = ((N + 127) / 128) * 128;
  const int pad64 = ((N + 63) / 64) * 64;
  t.wide = (pad64 < pad128) ? 0 : 1;
  t.bm = t.wide ? 128 : 256;
  t.bn = t.wide ? 128 : 64;
  t.tiles = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
  const int nk = (K + EA_BK - 1) / EA_BK;
  int splits = 1;
  const long long blocks = (long long)t.tiles * batch;
  if (allow_split && blocks < 256 && nk >= 8) {
    splits = (int)((512 + blocks - 1) / blocks);
    if (splits > nk / 4) splits = nk / 4;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
  }
  t.ktiles_per_split = (nk + splits - 1) / splits;
  t.splits = (nk + t.ktiles_per_split - 1) / t.ktiles_per_split;
  return t;
}

static int fill_epilogue(EaEpilogue& e, const ea_epilogue* epi, int M, int N) {
  if (!epi || !epi->out) return EA_ERR_BAD_ARG;
  e.bias = epi->bias;
  e.bias_per_row = epi->bias_per_row;
  e.rowvec = epi->rowvec;
  e.rowvec_ld = epi->rowvec_ld;
  e.rows_per_group = epi->rows_per_group > 0 ? epi->rows_per_group : 1;
  e.act = epi->act;
  e.scale = epi->scale;
  e.row_scale = epi->row_scale;
  e.residual = (const f16*)epi->residual;
  e.residual32 = epi->residual32;
  e.ldr = epi->ldr;
  e.out = epi->out;
  e.ldc = epi->ldc;
  e.out_f32 = epi->out_f32;
  e.geglu_block = epi->geglu_block > 0 ? epi->geglu_block : 64;
  e.ln_stats = epi->ln_stats;
  e.ln_parts = epi->ln_parts;
  e.ln_colsum = epi->ln_colsum;
  e.ln_eps = epi->ln_eps;
  e.row_stats_out = epi->row_stats_out;
  e.gn_stats_out = epi->gn_stats_out;
  e.gn_hw = epi->gn_rows_per_sample;
  e.gn_cpg = epi->gn_cpg;
  e.gn_next_out = (f16*)epi->gn_next_out;
  e.gn_next_gamma = epi->gn_next_gamma;
  e.gn_next_beta = epi->gn_next_beta;
  e.gn_next_eps = epi->gn_next_eps;
  e.gn_next_silu = epi->gn_next_silu;
  if (e.gn_next_out && (!e.gn_next_gamma || !e.gn_next_beta || e.gn_hw <= 0 || e.gn_cpg < 4 || (e.gn_cpg & 3) || e.gn_stats_out ||
                        ((((uintptr_t)e.gn_next_out) | ((uintptr_t)e.gn_next_gamma) | ((uintptr_t)e.gn_next_beta)) & 15)))
    return EA_ERR_BAD_ARG;
  if (e.gn_stats_out && (e.gn_hw <= 0 || e.gn_cpg < 8 || (((uintptr_t)e.gn_stats_out) & 7) || epi->act == EA_ACT_GEGLU)) return EA_ERR_BAD_ARG;
  if (e.ln_stats && (!e.ln_colsum || e.ln_parts <= 0)) return EA_ERR_BAD_ARG;
  if ((((uintptr_t)e.ln_stats) & 7) || (((uintptr_t)e.ln_colsum) & 15) || (((uintptr_t)e.row_stats_out) & 7)) return EA_ERR_BAD_ARG;
  e.M = M;
  e.N = (epi->act == EA_ACT_GEGLU) ? N / 2 : N;
  if (epi->act < 0 || epi->act > EA_ACT_GEGLU) return EA_ERR_BAD_ARG;
  if (epi->act == EA_ACT_GEGLU && e.geglu_block != 64 && e.geglu_block != 80 && e.geglu_block != 32) return EA_ERR_UNSUPPORTED;
  if (epi->act == EA_ACT_GEGLU && e.row_stats_out) return EA_ERR_UNSUPPORTED;
  if (epi->act == EA_ACT_GEGLU && (N % e.geglu_block) != 0) return EA_ERR_BAD_SHAPE;
  if (epi->act == EA_ACT_GEGLU && epi->bias_per_row) return EA_ERR_UNSUPPORTED;
  if (e.ldc < e.N) return EA_ERR_BAD_SHAPE;
  if ((e.residual || e.residual32) && e.ldr < e.N) return EA_ERR_BAD_SHAPE;
  return EA_OK;
}

// ---- fast path (ea_gemm2.h): plan + eligibility
struct Plan2 {
  int kind;       // kernel instantiation, see launch_fast
  int bm, bn;
  int tiles;
  int splits;
  int ktiles_per_split;
};

static bool fast_eligible(const EaGemmParams& p) {
  if (p.K % EA_BK) return false;
  // (GEGLU weights packed for the register-direct epilogue must run here whatever M is: the packing is fixed at load time)
  if (p.N < 64 || (p.M < 32 && !(p.epi.act == EA_ACT_GEGLU && p.epi.geglu_block == 32))) return false;
  // LDS-DMA goes through 2 GiB buffer descriptors with 32-bit per-lane byte offsets
  const long long lim = 0x7fffffffLL - 4096;
  if ((long long)p.N * p.ldw * 2 > lim) return false;
  if (p.conv) {
    if ((p.c1 % EA_BK) || (p.c2 % EA_BK) || p.a2_add) return false;
    const long long cmax = p.c1 > p.c2 ? p.c1 : p.c2;
    if ((long long)(p.M / (p.Hout * p.Wout) + 1) * p.Hin * p.Win * cmax * 2 > lim) return false;
  } else {
    if ((long long)p.M * p.lda * 2 > lim) return false;
  }
  if (p.epi.act == EA_ACT_GEGLU)
    return (p.epi.geglu_block == 80 && (p.N % 160) == 0) || (p.epi.geglu_block == 32 && (p.N % 128) == 0);
  return true;
}

// wave-column blocks (= row-statistics parts) of an N-wide output: 80 columns with 160-wide tiles, else 64
static int row_stat_parts(int N) { return (N % 160 == 0) ? N / 80 : (N + 63) / 64; }

// ---- ea_gemm8.h (256 x 256 tiles, one 8-wave workgroup per CU): which launches take it.  Measured per class against the
// 128 x 160 / 128 x 128 two-workgroups-per-CU tiles on one MI355X (profiles/r05_gemm8_probe_staggered.jsonl beside
// r05_gemm_bench_shipped_same_box.jsonl, then through this library: profiles/r05_gemm8_shipped.jsonl): it wins where the
// K loop is long enough to amortise a 64-KiB-per-tile prologue and a 128-KiB epilogue (K >= 1024) AND the tile count
// fills whole rounds of the 256 CUs (a 320-tile launch runs two rounds for 1.25 rounds of work and loses).
static bool gemm8_shape_ok(const EaGemmParams& p) {
  if (p.K < 1024 || p.M < 2048 || p.N < 256) return false;
  const long long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
  const double tiles = (double)tm * tn * p.batch;
  const double rounds = tiles / 256.0;
  if (rounds / ceil(rounds) < 0.9) return false;                              // partial last round
  if ((double)p.M * p.N / ((double)tm * tn * 65536.0) < 0.93) return false;   // ragged edge tiles
  return true;
}

// Tuning / A-B knobs (include/editanything_hip.h `ea_tuning`): set explicitly through ea_set_tuning() by tools and tests,
// per host thread, all zero in production.  Nothing on the launch path reads the environment.
//   force_generic   route everything to ea_gemm.h
//   variant         0 auto; k forces instantiation k of launch_fast:
//        1: 128 x bn, 4 waves 2x2 (wave tile 64x80), 2-stage, 16x16x32     2: same, 3-stage counted vmcnt
//        3: 256 x bn, 8 waves 4x2 (64x80), 3-stage                        4: 256 x bn, 4 waves 2x2 (128x80), 3-stage
//        5: 256 x bn, 4 waves 4x1 (64x160), 3-stage, 32x32x16
//        6 / 7 / 8: variants 3 / 2 / 5 with the next tile's DMA pieces interleaved between the MFMA groups
//        9: 64 x bn, 2-stage   10 / 11 / 12: loader-wave forms   13: 256 x bn, 8 waves 4x2, 3-deep ring, PING-PONG
//        30: ea_gemm8.h, 256 x 256, 8 waves 2x4 (wave tile 128x64), staggered 8-phase K tile (auto: gemm8_shape_ok)
//        33 / 34: variants 9 / 1 with a 3-stage ring under the register-direct epilogue (round 6; a launch that needs the
//                 LDS-slab epilogue, and a twin launch, run the 2-stage instantiation of the same tile)
//        1 forced also keeps the automatic policy off ea_gemm8 (A/B)
//   splits / bn     tuning sweeps: force the split-K factor / 128-wide column tiles
//   no_register_direct   keep the LDS-slab epilogue where the register-direct one applies (A/B)
//   debug           K-loop / epilogue ablations (ea_gemm2.h p.debug)
static thread_local ea_tuning g_tune = {0, 0, 0, 0, 0, 0};
#define g_force_generic (g_tune.force_generic)
#define g_variant (g_tune.variant)
#define g_force_splits (g_tune.splits)
#define g_force_bn (g_tune.bn)
#define g_no_tr (g_tune.no_register_direct)
static void read_env() {}

// Cost model (microseconds) that picks tile height (64 / 128 rows) and split-K factor.  Fitted to the forced
// (variant, splits) sweep of tools/sweep_splits.py on MI355X (profiles/r01_sweep_splits.json):
//  * a K-loop iteration of the 2-stage LDS-DMA pipeline is latency-bound, not MFMA-bound: ~0.7 us per 64-deep K tile
//    for a workgroup alone on its CU whatever the tile height, ~1.2 us (128 rows) / ~0.8 us (64 rows) when two
//    workgroups share the CU -- so two co-resident workgroups nearly double a CU's throughput, and 64-row tiles win
//    whenever 128-row tiles would leave CUs with fewer than two workgroups;
//  * fixed cost per launch (launch + prologue + epilogue): ~13 us for 128-row tiles, ~7 us for 64-row tiles;
//  * split-K adds the reduce launch (~3 us) and (splits + 1) passes over the fp32 output at ~8 TB/s (MALL-resident).
static double plan_cost(int bm, int bn, int M, int N, int K, int batch, int s, int conv, int* kps_out, int* s_eff_out) {
  const int nk = K / EA_BK;
  const double tiles = (double)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
  const int kps = (nk + s - 1) / s;
  const int s_eff = (nk + kps - 1) / kps;
  const double wgs = tiles * s_eff;
  const double shape = (bn == 160 ? 1.0 : 0.9) * (conv ? 1.07 : 1.0);
  const double t1 = (bm == 64 ? 0.62 : (bm == 128 ? 0.72 : 1.1)) * shape;   // alone on the CU
  const double t2 = (bm == 64 ? 0.72 : (bm == 128 ? 1.20 : 1.1)) * shape;   // two per CU (256-row tiles: one per CU)
  const double slots = (bm == 256) ? 256.0 : 512.0;
  double loop;
  if (wgs <= 256.0) loop = kps * t1;
  else if (wgs <= slots) loop = kps * (t1 + (t2 - t1) * (wgs - 256.0) / 256.0);
  else loop = ceil(2.0 * wgs / slots) * 0.5 * kps * t2;
  double cost = loop + (bm == 64 ? 7.0 : 13.0);
  if (s_eff > 1) cost += 3.0 + 0.15 * s_eff + (double)M * N * batch * 4.0 * (s_eff + 1.0) / 8.0e6;
  *kps_out = kps;
  *s_eff_out = s_eff;
  return cost;
}

static Plan2 plan_fast(int M, int N, int K, int batch, int allow_split, int conv, int geglu = 0) {
  read_env();
  Plan2 t;
  t.bn = (N % 160 == 0) ? 160 : 128;
  if (g_force_bn == 128 && !geglu) t.bn = 128;   // tuning sweeps only (EA_GEMM2_BN)
  if (geglu == 32) t.bn = 128;                    // [16 value | 16 gate] packing: 64-column wave tiles
  const int nk = K / EA_BK;
  double best = 1e30;
  t.bm = 128; t.splits = 1; t.ktiles_per_split = nk; t.kind = 1;
  const int smax = allow_split ? 16 : 1;
  // candidate instantiations: auto = {128-row, 64-row} 2-stage tiles; a forced variant restricts to its own height
  const int forced_bm = (g_variant == 0) ? 0 : (g_variant == 9 || g_variant == 11 || g_variant == 12 || g_variant == 32 || g_variant == 33) ? 64 : (g_variant <= 2 || g_variant == 7 || g_variant == 10 || g_variant == 31 || g_variant == 34) ? 128 : 256;   // 3..6, 8, 13: 256 rows; 31 / 32: intra-workgroup split-K (KS = 2) on 128- / 64-row tiles
  const int cand_bm[2] = {128, 64};
  for (int ci = 0; ci < (forced_bm ? 1 : 2); ++ci) {
    const int bm = forced_bm ? forced_bm : cand_bm[ci];
    for (int s = 1; s <= smax; ++s) {
      if (g_force_splits > 0 && s != g_force_splits && allow_split) continue;
      if (s > 1 && nk / s < 4 && g_force_splits == 0) break;
      // 64-row tiles re-read every weight tile twice as often: no deep split-K on them.  (Round 5 re-measured this cut-off:
      // ALONE, 64-row x 8 slices = 512 workgroups is ahead on the M = 512 convolutions since the XCD-partitioned tile order --
      // conv3x3 8x8 1280->1280 38.3 -> 32.8 us, the stride-2 convolutions 39 -> 33 us, profiles/r05_split_sweep.jsonl,
      // r05_plan_rule_ab.jsonl -- but in the STEP, where the ControlNet branch runs beside the UNet encoder, the launches that
      // take every CU slot give the neighbour stream nothing: same box A/B/A/B 11.99 / 12.02 / 11.93 / 12.06 images/s (new / old),
      // profiles/r05_plan_rule_bench_aba.jsonl.  The rule stays; EA_EXP & 128 builds the relaxed one (>= 10 K tiles per slice).)
      if (bm == 64 && s > 2 && (!(EA_EXP & 128) || nk / s < 10) && g_force_splits == 0) break;
      // ... and they only pay while every workgroup is resident at once (<= 2 per CU): with a second round the
      // 128-row tiles' better weight reuse wins again (M32768 x K1280: 44 us vs 38 us)
      if (bm == 64 && g_force_splits == 0 && !forced_bm &&
          (double)((M + 63) / 64) * ((N + t.bn - 1) / t.bn) * batch * s > 512.0) continue;
      int kps, s_eff;
      const double c = plan_cost(bm, t.bn, M, N, K, batch, s, conv, &kps, &s_eff);
      if (c < best - 1e-9) {
        best = c; t.bm = bm; t.splits = s_eff; t.ktiles_per_split = kps;
        t.kind = forced_bm ? g_variant : (bm == 64 ? 9 : 1);
      }
    }
  }
  // (The loader-wave instantiation, kind 11, used to be picked here for the 64x64-level convolutions on the strength of
  // a +4..9 % microbenchmark.  Re-measured with tools/gemm_bench in graph replay -- how the product runs -- it is 1.65x
  // SLOWER than the two-workgroup 128-row tiles on exactly those launches (117 vs 70 us for conv3x3 B8 64x64 320->320,
  // profiles/r01x_gemm_bench.jsonl): kind 11 stays reachable through EA_GEMM2_VARIANT only.)
#if EA_TOOLS && (EA_EXP & 256)
  // EXPERIMENT (round 6, tools builds with EA_EXP & 256): intra-workgroup split-K (kind 32: 64-row tiles, two K streams per
  // workgroup, ea_gemm2.h KS = 2) where the 64-row tiles give every CU at most ONE workgroup and a K stream has 8 .. 64 K tiles --
  // the classes where the forced sweep has it ahead of the plan above (profiles/r06_intra_wg_splitk_sweep.jsonl): M = 2048 / 512
  // linears at K = 1280, the 8 x 8 and stride-2 convolutions (with 2 - 4 K slices across workgroups on top), the 1 x 1 convolutions
  if (g_variant == 0 && g_force_splits == 0 && !geglu && batch == 1) {
    const long long tiles64 = (long long)((M + 63) / 64) * ((N + t.bn - 1) / t.bn);
    if (tiles64 <= 256) {
      int s = 1;
      while (allow_split && tiles64 * (s * 2) <= 256 && nk / (s * 2) >= 8 && s * 2 <= 16) s *= 2;
      if (nk / s >= 8 && nk / s <= 64) {
        t.kind = 32; t.bm = 64;
        t.ktiles_per_split = (nk + s - 1) / s;
        t.splits = (nk + t.ktiles_per_split - 1) / t.ktiles_per_split;
      }
    }
  }
#endif
#if (EA_EXP & 512)
  // EXPERIMENT (round 6, EA_EXP & 512): the 3-stage ring (kind 33) where the 64-row plan is unsplit, gives every CU at most ONE workgroup
  // and the K loop is short (<= 24 tiles) -- the [M <= 2048 x N x 1280] Linears / 1 x 1 convolutions of the 16 x 16 and 8 x 8 levels.
  // Their 2-stage loop waits out every tile's memory latency, and in a denoising step the weights come from HBM (2.5 GB per
  // evaluation): tools/gemm_bench --rotate-mb 700 has [2048 x 1280 x 1280] at 21.7 us (2-stage) against 19.9 (3-stage), with
  // cache-hot weights 16.1 against 19.8 (profiles/r06_three_stage_tr_forced_sweep.jsonl).
  if (g_variant == 0 && g_force_splits == 0 && !geglu && batch == 1 && t.kind == 9 && t.splits == 1 && nk >= 8 && nk <= 24 &&
      (long long)((M + 63) / 64) * ((N + t.bn - 1) / t.bn) <= 256)
    t.kind = 33;
#endif
  t.tiles = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
  return t;
}

// instantiations that carry the register-direct epilogue of the planned 2-stage tiles (fold / statistics / raw split-K dump)
static bool tr_plan_kind(int k) { return k == 1 || k == 9 || k == 33 || k == 34 || (EA_TOOLS && (k == 31 || k == 32)); }

// rows per GroupNorm-statistics chunk of plan t (its wave tile height), 0 = the epilogue cannot emit them: the wave
// tiles (bm/2 x bn/2) must hold whole groups and row ranges inside one sample
static int gn_stats_rows(const Plan2& t, int M, int N, int hw, int cpg) {
  if (t.splits != 1 || !tr_plan_kind(t.kind) || g_no_tr) return 0;
  const int wtm = t.bm / 2, wtn = t.bn / 2;
  if (hw <= 0 || cpg < 8 || (M % hw) || (hw % wtm) || (wtn % cpg) || (N % t.bn) || (N % cpg)) return 0;
  // a workgroup tile must not straddle two samples: the launches that ask for the partials carry a per-sample row
  // vector (ResBlock time embedding), and the register-direct epilogue that writes the partials needs all bm rows of a
  // tile in ONE row-vector group (launch_fast: rows_per_group % bm == 0).  hw = 8x8, 24x24, 40x40, 56x56 with 128-row tiles
  // used to pass this query and then fail the launch (round-2 advisor finding): now the query says 0 and the caller
  // runs the statistics pass.
  if (hw % t.bm) return 0;
  if (hw / wtm > 128) return 0;   // EA_GN_MAX_CHUNKS of the apply pass
  return wtm;
}

static int launch_reduce(EaGemmParams& p, void* stream) {
  const long long total = (long long)p.batch * p.M * ((p.N + 7) / 8);
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  auto rfn = ea_splitk_reduce_kernel;
  EA_LAUNCH(rfn, dim3((unsigned)nb), dim3(256), 0, stream, p);
  return ea_launch_status();
}

// ---- split-K reduction that also applies the consuming GroupNorm (ea_gemm.h ea_splitk_reduce_gn_kernel)
// shape side of the decision (ea_gemm_gn_next_ok + the launch): whole samples, 4-channel pieces, a slab one workgroup holds
static bool gn_next_shape_ok(int M, int N, int hw, int cpg) {
  if (hw <= 0 || cpg < 4 || (cpg & 3) || (N % cpg) || (M % hw) || (N & 3)) return false;
  return (long long)hw * (cpg >> 2) <= 512ll * EA_RGN_MAXQ;
}
static bool gn_next_epi_ok(const EaGemmParams& p) {
  const EaEpilogue& e = p.epi;
  return p.batch == 1 && !e.out_f32 && !e.bias_per_row && !e.row_scale && !e.residual32 && !e.row_stats_out && !e.ln_stats &&
         (e.act == EA_ACT_NONE || e.act == EA_ACT_SILU) && (e.ldc & 3) == 0 && (!e.residual || (e.ldr & 3) == 0) &&
         (!e.rowvec || (e.rowvec_ld & 3) == 0) &&
         ((((uintptr_t)e.out) | ((uintptr_t)e.residual)) & 7) == 0 && ((((uintptr_t)e.bias) | ((uintptr_t)e.rowvec)) & 15) == 0;
}
// fallback producer of the output row statistics (launches whose epilogue could not write them)
static int launch_row_stats(EaGemmParams& p, void* stream) {
  const EaEpilogue& e = p.epi;
  if (!e.row_stats_out) return EA_OK;
  if (e.out_f32 || p.batch != 1) return EA_ERR_UNSUPPORTED;
  auto kfn = ea_row_stats_kernel;
  EA_LAUNCH(kfn, dim3((p.M + 3) / 4), dim3(256), 0, stream, (const f16*)e.out, e.ldc, e.row_stats_out, p.M, e.N, row_stat_parts(e.N));
  return ea_launch_status();
}

// Number of CUs of the device (persistent launches: one resident workgroup per CU).
static int cu_count() {
#ifdef EA_EMU
  return 4;   // host emulation: a tiny "device", so the tests walk several rounds of the persistent tile loop
#else
  static const int n = []() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
#endif
}

#if EA_TOOLS
// ---- ea_gemm3.h: the persistent 8-wave kernel of round 3 (tools builds only).
struct Plan3 {
  int use;        // 0: stay on ea_gemm2
  int bn, splits, ktiles_per_split, ks, gm, trx, gn_rows;
  int epi_fast;
};

// Split-K factor of the persistent kernel: the launch runs ceil(tiles * s / CUs) rounds of (K tiles of a slice + a fixed
// per-item cost, in K-tile units); split launches pay the reduce kernel and the fp32 partial round trip.
static int plan3_splits(int tiles, int nk, long long MN, int allow_split) {
  const int ncu = cu_count();
  int best_s = 1;
  double best = 1e30;
  const int smax = allow_split ? 16 : 1;
  for (int s = 1; s <= smax; ++s) {
    if (g_force_splits > 0 && s != g_force_splits && allow_split) continue;
    if (s > 1 && nk / s < 4 && g_force_splits == 0) break;
    const int kps = (nk + s - 1) / s;
    const int s_eff = (nk + kps - 1) / kps;
    const double rounds = ceil((double)tiles * s_eff / ncu);
    double cost = rounds * (kps * 0.55 + 4.0);
    if (s_eff > 1) cost += 4.0 + 0.15 * s_eff + (double)MN * 4.0 * (s_eff + 1.0) / 6.0e6;
    if (cost < best - 1e-9) { best = cost; best_s = s_eff; }
  }
  return best_s;
}

// Eligibility + plan of a launch on ea_gemm3 (the register-direct epilogue's conditions, batch 1).  `want`: 0 = the
// automatic policy, 20 = forced with the automatic wave roles, 21 / 22 / 23 = forced m-split / k-split / loader waves.
static Plan3 plan3(const EaGemmParams& p, int want) {
  Plan3 t{};
  const EaEpilogue& e = p.epi;
  if (want == 0) return t;                      // never chosen automatically: opt-in through ea_set_tuning variants 20-23
  if (p.batch != 1 || g_no_tr) return t;
  const bool geglu = e.act == EA_ACT_GEGLU;
  if (geglu && e.geglu_block != 32) return t;
  t.bn = (p.N % 160 == 0 && !geglu) ? 160 : 128;
  if (g_force_bn == 128) t.bn = 128;
  const long long span = (long long)p.M * e.ldc + e.N;
  bool ok = !e.out_f32 && !e.residual32 && !e.row_scale && !e.bias_per_row && (e.N & 7) == 0 && (e.ldc & 7) == 0 &&
            (((uintptr_t)e.out) & 15) == 0 && span < 0x7fffffffLL && (((uintptr_t)e.bias) & 15) == 0;
  if (ok && e.residual) ok = (e.ldr & 7) == 0 && (((uintptr_t)e.residual) & 15) == 0 && (long long)p.M * e.ldr < 0x7fffffffLL;
  if (ok && e.rowvec) ok = e.rows_per_group > 0 && (e.rows_per_group % 128) == 0 && (((uintptr_t)e.rowvec) & 15) == 0 && (e.rowvec_ld & 3) == 0;
  if (ok && geglu) ok = !e.residual && !e.rowvec && (p.N % 128) == 0;
  if (!ok) return t;
  const int tiles = ((p.M + 127) / 128) * ((p.N + t.bn - 1) / t.bn);
  const int nk = p.K / EA_BK;
  const bool stats = e.ln_stats || e.gn_stats_out;
  t.splits = plan3_splits(tiles, nk, (long long)p.M * p.N, !geglu && (p.N & 3) == 0);
  if (stats && t.splits > 1) return t;          // the fold / GroupNorm partials exist in unsplit launches only (the queries say so)
  t.ktiles_per_split = (nk + t.splits - 1) / t.splits;
  t.splits = (nk + t.ktiles_per_split - 1) / t.ktiles_per_split;
  t.epi_fast = geglu ? 3 : 1;
  t.trx = ((e.ln_stats || e.row_stats_out || e.gn_stats_out) && t.splits == 1) ? 2 : 1;
  const int ks_ = (want == 21) ? 1 : (want == 22) ? 2 : (want == 23) ? 0 : (t.ktiles_per_split >= 12 ? 2 : 1);
  t.gn_rows = ks_ == 0 ? 64 : 32;               // rows a wave emits = rows per GroupNorm-statistics chunk
  if (e.gn_stats_out) {
    const int hw = e.gn_hw, cpg = e.gn_cpg;
    if (hw <= 0 || cpg < 8 || (p.M % hw) || (hw % 128) || ((t.bn / 2) % cpg) || (p.N % t.bn) || (p.N % cpg) || hw / t.gn_rows > 128) return t;
  }
  t.ks = (want == 21) ? 1 : (want == 22) ? 2 : (want == 23) ? 0 : (t.ktiles_per_split >= 12 ? 2 : 1);
  // grouped tile order: an XCD's run of (items per round) / 8 tiles should cover about as many A row panels as W column
  // panels; with few tile columns take them all
  const int tiles_m = (p.M + 127) / 128, tiles_n = (p.N + t.bn - 1) / t.bn;
  int run = (tiles < cu_count() ? tiles : cu_count()) / 8;
  if (run < 1) run = 1;
  int gm = 1;
  while (gm * gm < run) ++gm;                                  // ~ sqrt(run) rows x sqrt(run) columns
  if (tiles_n * gm < run) gm = (run + tiles_n - 1) / tiles_n;  // few tile columns: take them all, more rows
  if (gm > tiles_m) gm = tiles_m;
  t.gm = gm;
  t.use = 1;
  return t;
}

static int launch_fast3(EaGemmParams& p, const Plan3& t, void* workspace, size_t ws_bytes, void* stream) {
  p.splits = t.splits;
  p.ktiles_per_split = t.ktiles_per_split;
  p.partial = nullptr;
  p.debug = 0;
  p.epi_fast = t.epi_fast;
  p.raster_gm = t.gm;
#if EA_G3_PROF
  // profiling side build: the phase totals go behind the split-K partials in the caller's workspace
  p.prof = (workspace && ws_bytes >= ((size_t)t.splits * p.M * p.N * 4 + (1u << 20))) ? (unsigned long long*)((char*)workspace + (size_t)(t.splits > 1 ? t.splits : 0) * p.M * p.N * 4) : nullptr;
#endif
  if (t.splits > 1) {
    const size_t need = (size_t)t.splits * p.M * p.N * sizeof(float);
    if (!workspace || ws_bytes < need) return EA_ERR_WORKSPACE;
    p.partial = (float*)workspace;
  }
  if ((p.epi.ln_stats || p.epi.gn_stats_out) && t.splits > 1) return EA_ERR_UNSUPPORTED;
  if (p.epi.gn_next_out) return EA_ERR_UNSUPPORTED;
  const int items = ((p.M + 127) / 128) * ((p.N + t.bn - 1) / t.bn) * t.splits;
  const int ncu = cu_count();
  dim3 grid(items < ncu ? items : ncu, 1, 1);
#define EA_LAUNCH_G3(BN_, TRX_, KS_)                                                  \
  do {                                                                                \
    auto kfn = ea_gemm3_kernel<BN_, TRX_, KS_>;                                       \
    const int smem = ea_gemm3_lds_bytes(BN_);                                         \
    ea_allow_big_lds(kfn, smem);                                                      \
    EA_LAUNCH(kfn, grid, dim3(KS_ == 0 ? 768 : 512, 1, 1), smem, stream, p);          \
  } while (0)
#define EA_LAUNCH_G3K(BN_, TRX_)                                                      \
  do {                                                                                \
    if (t.ks == 2) EA_LAUNCH_G3(BN_, TRX_, 2);                                        \
    else if (t.ks == 1) EA_LAUNCH_G3(BN_, TRX_, 1);                                   \
    else EA_LAUNCH_G3(BN_, TRX_, 0);                                                  \
  } while (0)
  if (t.bn == 160) { if (t.trx == 2) EA_LAUNCH_G3K(160, 2); else EA_LAUNCH_G3K(160, 1); }
  else { if (t.trx == 2) EA_LAUNCH_G3K(128, 2); else EA_LAUNCH_G3K(128, 1); }
#undef EA_LAUNCH_G3K
#undef EA_LAUNCH_G3
  int st = ea_launch_status();
  if (st == EA_OK && t.splits > 1) {
    st = launch_reduce(p, stream);
    if (st == EA_OK) st = launch_row_stats(p, stream);
  }
  return st;
}

#endif  // EA_TOOLS

// ---- the LDS-DMA kernel (ea_gemm2.h): selection and issue are separate steps so that TWO problems that select the same
// launch (twins: ea_gemm_f16_pair / ea_conv2d_f16_pair) can be issued as one grid.
struct FastSel {
  Plan2 t;        // tile plan: instantiation kind, tile shape, tiles, split-K
  int tr;         // register-direct epilogue instantiation (0: the LDS-slab epilogue kernels)
  int tr_raw;     // split-K slices dumped raw by the register-direct epilogue
  int lnx;        // TR = 2: LayerNorm fold / row statistics / GroupNorm partials compiled in
  int reduce_gn;  // the split-K reduction applies the consuming GroupNorm
};

static bool same_launch(const FastSel& a, const FastSel& b) {
  return a.t.kind == b.t.kind && a.t.bm == b.t.bm && a.t.bn == b.t.bn && a.t.tiles == b.t.tiles && a.t.splits == b.t.splits &&
         a.t.ktiles_per_split == b.t.ktiles_per_split && a.tr == b.tr && a.tr_raw == b.tr_raw && a.lnx == b.lnx && a.reduce_gn == b.reduce_gn;
}

// Plans the launch of p: fills the plan-dependent fields of p (split-K, partial buffer at workspace + ws_off, epilogue
// form, tile order) and `s`.
static int fast_select_plan(EaGemmParams& p, Plan2 t, void* workspace, size_t ws_bytes, size_t ws_off, FastSel& s);

static int fast_select(EaGemmParams& p, void* workspace, size_t ws_bytes, size_t ws_off, FastSel& s) {
  // (an accumulator rescale at a K position -- acc_scale_kt -- needs the whole K range in one workgroup: no split-K)
  Plan2 t = plan_fast(p.M, p.N, p.K, p.batch, p.epi.act != EA_ACT_GEGLU && p.acc_scale_kt == 0, p.conv, p.epi.act == EA_ACT_GEGLU ? p.epi.geglu_block : 0);
  // ea_gemm8.h first where the shape policy (or a forced variant 30) asks for it; it exists with the register-direct epilogue
  // only (plain / raw split-K forms, no fold, no statistics), so a launch that needs anything else keeps the plan above
  const EaEpilogue& e0 = p.epi;
  if ((g_variant == 30 || (g_variant == 0 && g_force_splits == 0 && !g_no_tr && gemm8_shape_ok(p))) && e0.act != EA_ACT_GEGLU &&
      !e0.ln_stats && !e0.row_stats_out && !e0.gn_stats_out && !e0.gn_next_out && !(EA_TOOLS && g_tune.debug && g_tune.debug != 21 && g_tune.debug != 22)) {
    Plan2 t8 = t;
    t8.kind = 30; t8.bm = 256; t8.bn = 256;
    t8.tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (g_variant == 0) { t8.splits = 1; t8.ktiles_per_split = p.K / EA_BK; }
    FastSel s8;
    if (fast_select_plan(p, t8, workspace, ws_bytes, ws_off, s8) == EA_OK && s8.tr && !s8.lnx) { s = s8; return EA_OK; }
    if (g_variant == 30) return EA_ERR_UNSUPPORTED;   // forced, but not eligible: say so
  }
  return fast_select_plan(p, t, workspace, ws_bytes, ws_off, s);
}

static int fast_select_plan(EaGemmParams& p, Plan2 t, void* workspace, size_t ws_bytes, size_t ws_off, FastSel& s) {
  if (p.acc_scale_kt > 0 && (t.splits != 1 || (t.kind != 1 && t.kind != 9 && t.kind != 30))) return EA_ERR_UNSUPPORTED;
  p.splits = t.splits;
  p.ktiles_per_split = t.ktiles_per_split;
  p.partial = nullptr;
  p.debug = EA_TOOLS ? g_tune.debug : 0;
  if (t.splits > 1) {
    const size_t need = (size_t)p.batch * t.splits * p.M * p.N * sizeof(float);
    if (!workspace || ws_bytes < ws_off + need) return EA_ERR_WORKSPACE;
    p.partial = (float*)((char*)workspace + ws_off);
  } else if (p.debug == 3) {   // phase-timestamp dump (tools/phase_times.py): 8 x u64 per workgroup
    if (!workspace || ws_bytes < ws_off + (size_t)t.tiles * p.batch * 64) return EA_ERR_WORKSPACE;
    p.partial = (float*)((char*)workspace + ws_off);
  }
  // streamlined epilogue (ea_gemm2.h): every per-output option it does not implement must be off, offsets must fit
  // 32 bits, rows of a workgroup tile must share one row-vector group
  const EaEpilogue& e = p.epi;
  const long long span = (long long)p.M * e.ldc + e.N;
  {
    bool ok = t.splits == 1 && !e.out_f32 && !e.residual32 && !e.row_scale && !e.bias_per_row && e.act != EA_ACT_GEGLU &&
              (e.N & 7) == 0 && (e.ldc & 7) == 0 && (((uintptr_t)e.out) & 15) == 0 && (p.strideC & 7) == 0 &&
              span < 0x7fffffffLL && p.debug != 9;
    if (ok && e.residual)
      ok = (e.ldr & 7) == 0 && (((uintptr_t)e.residual) & 15) == 0 && (p.strideR & 7) == 0 && (long long)p.M * e.ldr < 0x7fffffffLL;
    if (ok && e.rowvec) ok = p.batch == 1 && e.rows_per_group > 0 && (e.rows_per_group % t.bm) == 0;
    p.epi_fast = ok ? 1 : 0;
    // streamlined GEGLU epilogue: 80-row packing on 160-wide tiles, plain fp16 output, nothing else per output
    if (e.act == EA_ACT_GEGLU && e.geglu_block == 80 && t.bn == 160 && t.splits == 1 && !e.out_f32 && !e.residual &&
        !e.residual32 && !e.row_scale && !e.rowvec && !e.bias_per_row && (e.N & 7) == 0 && (e.ldc & 7) == 0 &&
        (((uintptr_t)e.out) & 15) == 0 && (p.strideC & 7) == 0 && p.debug != 9)
      p.epi_fast = 2;
    // register-direct fp32 output (ea_gemm8.h only: ea_epi_tr.h F32OUT): fp32 out, optional fp32 residual
    if (t.kind == 30 && t.splits == 1 && e.out_f32 && !e.residual && !e.row_scale && !e.bias_per_row && e.act != EA_ACT_GEGLU &&
        (e.N & 3) == 0 && (e.ldc & 3) == 0 && (((uintptr_t)e.out) & 15) == 0 && (p.strideC & 3) == 0 && span < 0x7fffffffLL &&
        (!e.residual32 || ((e.ldr & 3) == 0 && (((uintptr_t)e.residual32) & 15) == 0 && (p.strideR & 3) == 0 && (long long)p.M * e.ldr < 0x7fffffffLL)) &&
        (!e.rowvec || (p.batch == 1 && e.rows_per_group > 0 && (e.rows_per_group % t.bm) == 0)) && p.debug != 9)
      p.epi_fast = 4;
    // register-direct GEGLU epilogue: 32-row packing on 128-wide tiles
    if (e.act == EA_ACT_GEGLU && e.geglu_block == 32) {
      if (!(t.bn == 128 && t.splits == 1 && (t.kind == 1 || t.kind == 9 || t.kind == 33 || t.kind == 34) && !e.out_f32 && !e.residual && !e.residual32 &&
            !e.row_scale && !e.rowvec && !e.bias_per_row && (e.N & 7) == 0 && (e.ldc & 7) == 0 && (((uintptr_t)e.out) & 15) == 0 &&
            (((uintptr_t)e.bias) & 15) == 0 && (p.strideC & 7) == 0 && span < 0x7fffffffLL && p.batch == 1))
        return EA_ERR_UNSUPPORTED;
      p.epi_fast = 3;
    }
  }
  // grouped tile order for wide outputs (ea_gemm.h ea_grouped_item): ~64 tiles of an XCD are resident at a time
  p.raster_gm = 0;
  {
    const int tiles_n = (p.N + t.bn - 1) / t.bn, tiles_m = t.tiles / tiles_n;
    // ... when the weight matrix does not stay in a 4-MiB L2 anyway (measured on the MI355X, same call, row-major vs
    // grouped: SAM MLP [16384 x 5120 x 1280] 305 -> 269 us, GEGLU [2048 x 10240 x 1280] 77 -> 63 us; the K = 320 GEGLU
    // projection, whose 1.6-MB weight is L2 resident, LOSES 6 % and keeps the row-major order)
    if (!(EA_TOOLS && g_tune.debug == 20) && tiles_n > 8 && tiles_m >= 16 && (long long)p.N * p.K * 2 > (3ll << 20)) p.raster_gm = 8;   // debug 20: row-major everywhere (A/B)
#if !(EA_EXP & 32)
    // XCD-PARTITIONED TILES for the weight-heavy launches of the 16 x 16 / 8 x 8 levels.  Row-major order deals every XCD a
    // few row tiles x ALL column panels: each of the 8 L2s pulls the whole weight matrix (5-10x the algorithmic fabric-side
    // bytes, profiles/r03_pmc_traffic.json).  Grouped order with gm row tiles per group makes an XCD's chunk gm rows x
    // (chunk / gm) column panels: a weight panel crosses the fabric tiles_m / gm times instead of 8, while the A rows of
    // the group (re-read once per tap by a 3x3 convolution) still fit its 4-MiB L2 -- gm = all rows (pure column-major)
    // at M = 512, 8 of 16 row tiles for conv3x3 16 x 16 1280 -> 1280 (a 2 x 4 XCD grid).
    else if (!(EA_TOOLS && g_tune.debug == 20) && tiles_n >= 8 && tiles_m > 1 && p.batch == 1) {
      const long long wbytes = (long long)p.N * p.K * 2;
      const long long abytes = p.conv ? (long long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * (p.c1 + p.c2) * 2 : (long long)p.M * p.K * 2;
      if (wbytes >= 2 * abytes) {
        const long long a_tile = abytes / tiles_m > 0 ? abytes / tiles_m : 1;
        long long gm = (5ll << 19) / a_tile;                 // ~2.5 MiB of A per XCD-resident row group
        if (gm > tiles_m) gm = tiles_m;
        if (gm > 1) p.raster_gm = (int)gm;
      }
    }
#endif
  }
  // register-direct epilogue (ea_gemm2.h TR = 1): the plain streamlined launches of the 2-stage 128- / 64-row tiles
  // split-K slices: the register-direct raw dump (N % 4 == 0 keeps the 16-byte stores aligned); the reduce kernel follows
  const bool tr_kind = t.kind == 1 || t.kind == 9 || t.kind == 30 || t.kind == 33 || t.kind == 34 || (EA_TOOLS && (t.kind == 24 || t.kind == 31 || t.kind == 32));
  const bool tr_raw = t.splits > 1 && !g_no_tr && tr_kind && (p.N & 3) == 0 && p.debug != 9;
  const bool tr = tr_raw || p.epi_fast == 3 ||
                  ((p.epi_fast == 1 || p.epi_fast == 4) && !g_no_tr && tr_kind && (((uintptr_t)p.epi.bias) & 15) == 0 &&
                   (!p.epi.rowvec || ((((uintptr_t)p.epi.rowvec) & 15) == 0 && (p.epi.rowvec_ld & 3) == 0)));
  // the LayerNorm fold exists in the register-direct epilogue only (callers ask ea_gemm_ln_fold_ok first)
  if (p.epi.ln_stats && (!tr || t.splits > 1)) return EA_ERR_UNSUPPORTED;
  // the consuming GroupNorm rides on the split-K reduction of the raw register-direct dump only (callers ask ea_gemm_gn_next_ok)
  if (p.epi.gn_next_out && (!tr_raw || !gn_next_shape_ok(p.M, p.N, p.epi.gn_hw, p.epi.gn_cpg) || !gn_next_epi_ok(p))) return EA_ERR_UNSUPPORTED;
  // ... and so do the GroupNorm partials (callers ask ea_gemm_gn_stats_chunk_rows first)
  if (p.epi.gn_stats_out && (!tr || t.splits > 1 || p.epi_fast != 1 || p.batch != 1 || !gn_stats_rows(t, p.M, p.N, p.epi.gn_hw, p.epi.gn_cpg)))
    return EA_ERR_UNSUPPORTED;
  // kinds 33 / 34 (the 3-stage ring of the 64- / 128-row tiles) exist with the register-direct epilogue only: a launch that
  // needs the LDS-slab epilogue runs the 2-stage instantiation of the same tile
  if ((t.kind == 33 || t.kind == 34) && !tr) t.kind = (t.kind == 33) ? 9 : 1;
  s.t = t;
  s.tr = tr ? 1 : 0;
  s.tr_raw = tr_raw ? 1 : 0;
  s.lnx = (tr && (p.epi.ln_stats || ((p.epi.row_stats_out || p.epi.gn_stats_out) && t.splits == 1))) ? 1 : 0;   // fold / statistics compiled in
  s.reduce_gn = p.epi.gn_next_out ? 1 : 0;
  return EA_OK;
}

// split-K reduction (+ the consuming GroupNorm) of p, or of the twins p and q as one grid
static int issue_reduce(const FastSel& s, EaGemmParams& p, EaGemmParams* q, void* stream) {
  if (s.reduce_gn) {
    const EaEpilogue& e = p.epi;
    const long long nq = (long long)e.gn_hw * (e.gn_cpg >> 2);
    dim3 grid((unsigned)(p.N / e.gn_cpg), (unsigned)(p.M / e.gn_hw), q ? 2 : 1);
    if (!q) {
      if (nq <= 1024) { auto k = ea_splitk_reduce_gn_kernel<1>; EA_LAUNCH(k, grid, dim3(1024), 128, stream, p); }
      else { auto k = ea_splitk_reduce_gn_kernel<EA_RGN_MAXQ>; EA_LAUNCH(k, grid, dim3(512), 128, stream, p); }
    } else {
      if (nq <= 1024) { auto k = ea_splitk_reduce_gn_pair_kernel<1>; EA_LAUNCH(k, grid, dim3(1024), 128, stream, p, *q); }
      else { auto k = ea_splitk_reduce_gn_pair_kernel<EA_RGN_MAXQ>; EA_LAUNCH(k, grid, dim3(512), 128, stream, p, *q); }
    }
    return ea_launch_status();
  }
  const long long total = (long long)p.batch * p.M * ((p.N + 7) / 8);
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (!q) { auto rfn = ea_splitk_reduce_kernel; EA_LAUNCH(rfn, dim3((unsigned)nb), dim3(256), 0, stream, p); }
  else { auto rfn = ea_splitk_reduce_pair_kernel; EA_LAUNCH(rfn, dim3((unsigned)nb, 2, 1), dim3(256), 0, stream, p, *q); }
  return ea_launch_status();
}

// Issues the launch fast_select planned for p -- and for its twin q (same FastSel: same instantiation, grid and LDS) in the
// same grid, blockIdx.y = problem.  Twins exist for the register-direct instantiations of the two planned tile heights.
static int fast_issue(const FastSel& s, EaGemmParams& p, EaGemmParams* q, void* stream) {
  const Plan2& t = s.t;
  dim3 grid(t.tiles, q ? 2 : 1, p.batch * t.splits);
#define EA_LAUNCH_G2L(BM_, BN_, WM_, WN_, ST_, MT_, IL_, LD_)                             \
  do {                                                                                \
    auto kfn = ea_gemm2_kernel<BM_, BN_, WM_, WN_, ST_, MT_, IL_, LD_>;               \
    const int smem = ST_ * (BM_ + BN_) * 128;                                         \
    ea_allow_big_lds(kfn, smem);                                                      \
    EA_LAUNCH(kfn, grid, dim3(WM_ * WN_ * 64 * (1 + LD_), 1, 1), smem, stream, p);    \
  } while (0)
#define EA_LAUNCH_G2(BM_, BN_, WM_, WN_, ST_, MT_, IL_) EA_LAUNCH_G2L(BM_, BN_, WM_, WN_, ST_, MT_, IL_, 0)
#define EA_LAUNCH_TR3(BM_, BN_, TR_)                                                  \
  do {                                                                                \
    auto kfn = ea_gemm2_kernel<BM_, BN_, 2, 2, 3, 16, 0, 0, TR_>;                     \
    const int smem = 3 * (BM_ + BN_) * 128;                                           \
    ea_allow_big_lds(kfn, smem);                                                      \
    EA_LAUNCH(kfn, grid, dim3(256, 1, 1), smem, stream, p);                           \
  } while (0)
#define EA_LAUNCH_TR(BM_, BN_, TR_)                                                   \
  do {                                                                                \
    const int smem = 2 * (BM_ + BN_) * 128;                                           \
    if (!q) {                                                                         \
      auto kfn = ea_gemm2_kernel<BM_, BN_, 2, 2, 2, 16, 0, 0, TR_>;                   \
      ea_allow_big_lds(kfn, smem);                                                    \
      EA_LAUNCH(kfn, grid, dim3(256, 1, 1), smem, stream, p);                         \
    } else {                                                                          \
      auto kfn = ea_gemm2_pair_kernel<BM_, BN_, 2, 2, 2, 16, 0, 0, TR_>;              \
      ea_allow_big_lds(kfn, smem);                                                    \
      EA_LAUNCH(kfn, grid, dim3(256, 1, 1), smem, stream, p, *q);                     \
    }                                                                                 \
  } while (0)
  if (s.tr) {
#if EA_TOOLS
    // kind 24 (experiment, round 3): ONE 8-wave workgroup per CU on a 256-row tile -- the two co-resident 128-row workgroups
    // merged, the weight panel fetched once for both halves (22 % fewer operand bytes from L2); same wave tiles, same
    // epilogue.  Measured 3-10 % SLOWER than the two independent workgroups on every level-0 / SAM class
    // (profiles/r03_kind24_merged_workgroups.jsonl): operand traffic is not what bounds the loop
    if (t.kind == 24) {
      if (s.lnx || q) return EA_ERR_UNSUPPORTED;
      if (t.bn == 160) {
        auto kfn = ea_gemm2_kernel<256, 160, 4, 2, 2, 16, 0, 0, 1>;
        const int smem = 2 * (256 + 160) * 128;
        ea_allow_big_lds(kfn, smem);
        EA_LAUNCH(kfn, grid, dim3(512, 1, 1), smem, stream, p);
      } else {
        auto kfn = ea_gemm2_kernel<256, 128, 4, 2, 2, 16, 0, 0, 1>;
        const int smem = 2 * (256 + 128) * 128;
        ea_allow_big_lds(kfn, smem);
        EA_LAUNCH(kfn, grid, dim3(512, 1, 1), smem, stream, p);
      }
    } else
#endif
#if EA_TOOLS
    // kinds 31 / 32 (experiment, round 6): intra-workgroup split-K -- 8 waves = two K streams with their own stage rings,
    // accumulators summed through LDS (ea_gemm2.h KS = 2); one workgroup per CU
    if (t.kind == 31 || t.kind == 32) {
      if (q || p.epi_fast == 3 || p.acc_scale_kt > 0) return EA_ERR_UNSUPPORTED;
#define EA_LAUNCH_KS2(BM_, BN_, TR_)                                                  \
  do {                                                                                \
    auto kfn = ea_gemm2_ks2_kernel<BM_, BN_, TR_>;                                    \
    const int smem = 2 * 2 * (BM_ + BN_) * 128;                                       \
    ea_allow_big_lds(kfn, smem);                                                      \
    EA_LAUNCH(kfn, grid, dim3(512, 1, 1), smem, stream, p);                           \
  } while (0)
      if (t.kind == 31) {
        if (t.bn == 160) { if (s.lnx) EA_LAUNCH_KS2(128, 160, 2); else EA_LAUNCH_KS2(128, 160, 1); }
        else { if (s.lnx) EA_LAUNCH_KS2(128, 128, 2); else EA_LAUNCH_KS2(128, 128, 1); }
      } else {
        if (t.bn == 160) { if (s.lnx) EA_LAUNCH_KS2(64, 160, 2); else EA_LAUNCH_KS2(64, 160, 1); }
        else { if (s.lnx) EA_LAUNCH_KS2(64, 128, 2); else EA_LAUNCH_KS2(64, 128, 1); }
      }
#undef EA_LAUNCH_KS2
    } else
#endif
    if (t.kind == 33 || t.kind == 34) {
      // 3-stage ring under the register-direct epilogue (round 6): one workgroup per CU, two K tiles in flight -- for the launches
      // whose tiles give a CU one workgroup anyway, where the 2-stage loop waits out every tile's memory latency and the
      // weights come from HBM
      if (q) return EA_ERR_UNSUPPORTED;
      if (t.kind == 34) {
        if (t.bn == 160) { if (s.lnx) EA_LAUNCH_TR3(128, 160, 2); else EA_LAUNCH_TR3(128, 160, 1); }
        else { if (s.lnx) EA_LAUNCH_TR3(128, 128, 2); else EA_LAUNCH_TR3(128, 128, 1); }
      } else {
        if (t.bn == 160) { if (s.lnx) EA_LAUNCH_TR3(64, 160, 2); else EA_LAUNCH_TR3(64, 160, 1); }
        else { if (s.lnx) EA_LAUNCH_TR3(64, 128, 2); else EA_LAUNCH_TR3(64, 128, 1); }
      }
    } else if (t.kind == 30) {
      if (s.lnx || q) return EA_ERR_UNSUPPORTED;
      auto kfn = ea_gemm8_kernel<1>;
      ea_allow_big_lds(kfn, EA_G8_LDS_BYTES);
      // persistent over the tiles (ea_gemm8.h: one workgroup per CU and K slice walks its tiles) from 8 rounds of tiles up.
      // Measured against one workgroup per tile (profiles/r05_gemm8_persistent_ab.jsonl): -12 ... -13 % at 8 / 16 rounds (the VAE's
      // up-sampling convolutions), -8 % at 8192^3, but +-0 ... +4 % WORSE at 2 - 5 rounds (SAM's Linears, the other VAE
      // convolutions): the static walk gives up the dispatcher's load balancing, which matters while a round is a large
      // share of the launch.  debug 21 / 22 (tools): never / always persistent (A/B)
      const bool persist = (EA_TOOLS && g_tune.debug == 21) ? false : (EA_TOOLS && g_tune.debug == 22) ? true : t.tiles >= 8 * cu_count();
      const int width = persist ? cu_count() : t.tiles;
      dim3 grid8(t.tiles < width ? t.tiles : width, 1, p.batch * t.splits);
      EA_LAUNCH(kfn, grid8, dim3(512, 1, 1), EA_G8_LDS_BYTES, stream, p);
    } else if (t.kind == 1) {
      if (t.bn == 160) { if (s.lnx) EA_LAUNCH_TR(128, 160, 2); else EA_LAUNCH_TR(128, 160, 1); }
      else { if (s.lnx) EA_LAUNCH_TR(128, 128, 2); else EA_LAUNCH_TR(128, 128, 1); }
    } else {
      if (t.bn == 160) { if (s.lnx) EA_LAUNCH_TR(64, 160, 2); else EA_LAUNCH_TR(64, 160, 1); }
      else { if (s.lnx) EA_LAUNCH_TR(64, 128, 2); else EA_LAUNCH_TR(64, 128, 1); }
    }
    int st_tr = ea_launch_status();
    if (st_tr == EA_OK && t.splits > 1) {
      st_tr = issue_reduce(s, p, q, stream);
      if (st_tr == EA_OK) st_tr = launch_row_stats(p, stream);
      if (st_tr == EA_OK && q) st_tr = launch_row_stats(*q, stream);
    }
    return st_tr;                  // (unsplit launches: row statistics, if asked for, were written by the epilogue)
  }
#undef EA_LAUNCH_TR
#undef EA_LAUNCH_TR3
  if (q) return EA_ERR_UNSUPPORTED;   // twins run on the register-direct instantiations only (launch_pair checks first)
  switch (t.kind) {
    case 1: if (t.bn == 160) EA_LAUNCH_G2(128, 160, 2, 2, 2, 16, 0); else EA_LAUNCH_G2(128, 128, 2, 2, 2, 16, 0); break;
#if EA_TOOLS
    case 2: if (t.bn == 160) EA_LAUNCH_G2(128, 160, 2, 2, 3, 16, 0); else EA_LAUNCH_G2(128, 128, 2, 2, 3, 16, 0); break;
    case 3: if (t.bn == 160) EA_LAUNCH_G2(256, 160, 4, 2, 3, 16, 0); else EA_LAUNCH_G2(256, 128, 4, 2, 3, 16, 0); break;
    case 4: if (t.bn == 160) EA_LAUNCH_G2(256, 160, 2, 2, 3, 16, 0); else EA_LAUNCH_G2(256, 128, 2, 2, 3, 16, 0); break;
    case 5: if (t.bn == 160) EA_LAUNCH_G2(256, 160, 4, 1, 3, 32, 0); else EA_LAUNCH_G2(256, 128, 4, 1, 3, 32, 0); break;
    case 6: if (t.bn == 160) EA_LAUNCH_G2(256, 160, 4, 2, 3, 16, 1); else EA_LAUNCH_G2(256, 128, 4, 2, 3, 16, 1); break;
    case 7: if (t.bn == 160) EA_LAUNCH_G2(128, 160, 2, 2, 3, 16, 1); else EA_LAUNCH_G2(128, 128, 2, 2, 3, 16, 1); break;
    case 8: if (t.bn == 160) EA_LAUNCH_G2(256, 160, 4, 1, 3, 32, 1); else EA_LAUNCH_G2(256, 128, 4, 1, 3, 32, 1); break;
#endif
    case 9: if (t.bn == 160) EA_LAUNCH_G2(64, 160, 2, 2, 2, 16, 0); else EA_LAUNCH_G2(64, 128, 2, 2, 2, 16, 0); break;
#if EA_TOOLS
    case 10: if (t.bn == 160) EA_LAUNCH_G2L(128, 160, 2, 2, 3, 16, 0, 1); else EA_LAUNCH_G2L(128, 128, 2, 2, 3, 16, 0, 1); break;
    case 12: if (t.bn == 160) EA_LAUNCH_G2L(64, 160, 2, 2, 2, 16, 0, 1); else EA_LAUNCH_G2L(64, 128, 2, 2, 2, 16, 0, 1); break;
    case 13: if (t.bn == 160) EA_LAUNCH_G2(256, 160, 4, 2, 3, 16, 2); else EA_LAUNCH_G2(256, 128, 4, 2, 3, 16, 2); break;   // ping-pong
    case 11: if (t.bn == 160) EA_LAUNCH_G2L(64, 160, 2, 2, 3, 16, 0, 1); else EA_LAUNCH_G2L(64, 128, 2, 2, 3, 16, 0, 1); break;
#endif
    default: return EA_ERR_UNSUPPORTED;
  }
#undef EA_LAUNCH_G2
#undef EA_LAUNCH_G2L
  int st = ea_launch_status();
  if (st != EA_OK) return st;
  if (t.splits > 1) st = issue_reduce(s, p, nullptr, stream);
  if (st == EA_OK) st = launch_row_stats(p, stream);
  return st;
}

static int launch_fast(EaGemmParams& p, void* workspace, size_t ws_bytes, void* stream) {
#if EA_TOOLS
  {   // the persistent kernel of tools/kernels/ea_gemm3.h: opt-in (variants 20-23), tools build only
    const Plan3 t3 = plan3(p, (g_variant >= 20 && g_variant <= 23) ? g_variant : 0);
    if (t3.use) return launch_fast3(p, t3, workspace, ws_bytes, stream);
    if (g_variant >= 20 && g_variant <= 23) return EA_ERR_UNSUPPORTED;   // forced, but not eligible: say so
  }
#endif
  FastSel s;
  const int st = fast_select(p, workspace, ws_bytes, 0, s);
  if (st != EA_OK) return st;
  return fast_issue(s, p, nullptr, stream);
}

static int launch_gemm(EaGemmParams& p, void* workspace, size_t ws_bytes, void* stream) {
  read_env();
  if (!g_force_generic && fast_eligible(p)) return launch_fast(p, workspace, ws_bytes, stream);
  if (p.acc_scale_kt > 0) return EA_ERR_UNSUPPORTED;     // the K-position accumulator rescale lives in the LDS-DMA kernel only
  if (p.epi.act == EA_ACT_GEGLU && p.epi.geglu_block != 64) return EA_ERR_UNSUPPORTED;
  if (p.epi.ln_stats || p.epi.gn_stats_out || p.epi.gn_next_out) return EA_ERR_UNSUPPORTED;
  const int allow_split = (p.epi.act != EA_ACT_GEGLU);
  TilePlan t = plan_tiles(p.M, p.N, p.K, p.batch, allow_split);
  p.splits = t.splits;
  p.ktiles_per_split = t.ktiles_per_split;
  p.partial = nullptr;
  if (t.splits > 1) {
    const size_t need = (size_t)p.batch * t.splits * p.M * p.N * sizeof(float);
    if (!workspace || ws_bytes < need) return EA_ERR_WORKSPACE;
    p.partial = (float*)workspace;
  }
  dim3 grid(t.tiles, 1, p.batch * t.splits);
  dim3 block(256, 1, 1);
  if (t.wide) {
    auto kfn = ea_gemm_kernel<2, 2>;
    const int main_b = 2 * (128 + 128) * EA_BK * 2, epi_b = 128 * (128 + 4) * 4;
    const int smem = main_b > epi_b ? main_b : epi_b;
    ea_allow_big_lds(kfn, smem);
    EA_LAUNCH(kfn, grid, block, smem, stream, p);
  } else {
    auto kfn = ea_gemm_kernel<4, 1>;
    const int main_b = 2 * (256 + 64) * EA_BK * 2, epi_b = 256 * (64 + 4) * 4;
    const int smem = main_b > epi_b ? main_b : epi_b;
    ea_allow_big_lds(kfn, smem);
    EA_LAUNCH(kfn, grid, block, smem, stream, p);
  }
  int st = ea_launch_status();
  if (st != EA_OK) return st;
  if (t.splits > 1) st = launch_reduce(p, stream);
  if (st == EA_OK) st = launch_row_stats(p, stream);
  return st;
}

// Twins: two problems of one shape.  One grid when both select the same register-direct launch (the ControlNet trunk beside
// the UNet encoder: always); otherwise -- other instantiations, tools-only variants, scratch too small for two sets of
// split-K partials -- the two launches go out one after the other on the same stream (and then share the scratch).
static int launch_pair(EaGemmParams& p, EaGemmParams& q, void* workspace, size_t ws_bytes, void* stream) {
  read_env();
  const bool same_shape = p.M == q.M && p.N == q.N && p.K == q.K && p.batch == q.batch && p.conv == q.conv;
  if (same_shape && !g_force_generic && !(g_variant >= 20 && g_variant <= 23) && fast_eligible(p) && fast_eligible(q)) {
    FastSel sp, sq;
    const size_t half = (ws_bytes / 2) & ~(size_t)255;
    int st = fast_select(p, workspace, half, 0, sp);
    if (st == EA_OK) st = fast_select(q, workspace, 2 * half, half, sq);
    if (st == EA_OK) {        // a twin grid carries twice the workgroups: the 2-stage instantiation of the planned tile
      if (sp.t.kind == 33 || sp.t.kind == 34) sp.t.kind = (sp.t.kind == 33) ? 9 : 1;
      if (sq.t.kind == 33 || sq.t.kind == 34) sq.t.kind = (sq.t.kind == 33) ? 9 : 1;
    }
    if (st == EA_OK && same_launch(sp, sq) && sp.tr && (sp.t.kind == 1 || sp.t.kind == 9) && p.debug == 0 && q.debug == 0)
      return fast_issue(sp, p, &q, stream);
    if (st != EA_OK && st != EA_ERR_WORKSPACE) return st;
  }
  int st = launch_gemm(p, workspace, ws_bytes, stream);
  if (st == EA_OK) st = launch_gemm(q, workspace, ws_bytes, stream);
  return st;
}

}  // namespace

extern "C" int ea_set_tuning(const ea_tuning* t) {
  if (t) g_tune = *t;
  else g_tune = ea_tuning{0, 0, 0, 0, 0, 0};
  return EA_OK;
}

extern "C" int ea_tools_build(void) { return EA_TOOLS; }

extern "C" int ea_row_stats_parts(int N) { return N > 0 ? row_stat_parts(N) : 0; }

// 1 when a LayerNorm-folded launch of this shape runs through the register-direct epilogue (fp16 output, 16-byte
// aligned operands assumed): the 2-stage LDS-DMA tiles with no split-K.  act: EA_ACT_* (GEGLU = the 32-row packing).
// a launch of this shape with every epilogue operand aligned, as the queries assume
static EaGemmParams query_params(int M, int N, int K, int conv, int geglu32) {
  EaGemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.N = N; p.K = K; p.batch = 1; p.conv = conv;
  p.epi.N = geglu32 ? N / 2 : N; p.epi.ldc = p.epi.N; p.epi.M = M;
  p.epi.act = geglu32 ? EA_ACT_GEGLU : EA_ACT_NONE;
  p.epi.geglu_block = geglu32 ? 32 : 64;
  return p;
}

extern "C" int ea_gemm_ln_fold_ok(int M, int N, int K) {
  if (M < 32 || N < 64 || K <= 0 || (K % EA_BK) || (N % 8)) return 0;
#if EA_TOOLS
  if (g_variant >= 20 && g_variant <= 23) {
    EaGemmParams q = query_params(M, N, K, 0, 0);
    static const float dummy = 0.0f;
    q.epi.ln_stats = &dummy;
    return plan3(q, g_variant).use;
  }
#endif
  Plan2 t = plan_fast(M, N, K, 1, 1, 0, 0);
  return (t.splits == 1 && tr_plan_kind(t.kind) && !g_no_tr) ? 1 : 0;
}

extern "C" int ea_gemm_gn_next_ok(int M, int N, int K, int conv, int rows_per_sample, int cpg) {
  if (M < 32 || N < 64 || K <= 0 || (K % EA_BK) || (N % 8)) return 0;
  read_env();
  if (g_force_generic || g_no_tr || (g_variant != 0 && g_variant != 1 && g_variant != 9 && g_variant != 33 && g_variant != 34)) return 0;
  if (!gn_next_shape_ok(M, N, rows_per_sample, cpg)) return 0;
  const Plan2 t = plan_fast(M, N, K, 1, 1, conv ? 1 : 0, 0);
  return (t.splits > 1 && tr_plan_kind(t.kind)) ? 1 : 0;
}

extern "C" int ea_gemm_gn_stats_chunk_rows(int M, int N, int K, int conv, int rows_per_sample, int cpg) {
  if (M < 32 || N < 64 || K <= 0 || (K % EA_BK) || (N % 8)) return 0;
#if EA_TOOLS
  if (g_variant >= 20 && g_variant <= 23) {
    EaGemmParams q = query_params(M, N, K, conv ? 1 : 0, 0);
    static float dummy = 0.0f;
    q.epi.gn_stats_out = &dummy;
    q.epi.gn_hw = rows_per_sample;
    q.epi.gn_cpg = cpg;
    const Plan3 t3 = plan3(q, g_variant);
    return t3.use ? t3.gn_rows : 0;
  }
#endif
  Plan2 t = plan_fast(M, N, K, 1, 1, conv ? 1 : 0, 0);
  return gn_stats_rows(t, M, N, rows_per_sample, cpg);
}

extern "C" size_t ea_gemm_workspace_bytes(int M, int N, int K, int batch) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
  TilePlan t = plan_tiles(M, N, K, batch, 1);
  int splits = t.splits;
  if (K % EA_BK == 0) {
    for (int conv = 0; conv < 2; ++conv) {
      Plan2 f = plan_fast(M, N, K, batch, 1, conv);
      if (f.splits > splits) splits = f.splits;
    }
  }
#if EA_TOOLS
  if (K % EA_BK == 0 && batch == 1 && (N & 3) == 0 && g_variant >= 20 && g_variant <= 23) {   // the persistent kernel's own split choice, when forced
    const int bn = (N % 160 == 0) ? 160 : 128;
    const int s3 = plan3_splits(((M + 127) / 128) * ((N + bn - 1) / bn), K / EA_BK, (long long)M * N, 1);
    if (s3 > splits) splits = s3;
  }
#endif
  if (splits <= 1) return 0;
  return (size_t)batch * splits * M * N * sizeof(float);
}

extern "C" int ea_gemm_f16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int batch,
                           long long strideA, long long strideW, long long strideC, long long strideR,
                           const ea_epilogue* epi, void* workspace, size_t ws_bytes, void* stream);

static int setup_gemm(EaGemmParams& p, const void* A, int lda, const void* W, int ldw, int M, int N, int K, int batch,
                      long long strideA, long long strideW, long long strideC, long long strideR, const ea_epilogue* epi) {
  if (!A || !W) return EA_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return EA_ERR_BAD_SHAPE;
  if ((K & 7) || (lda & 7) || (ldw & 7) || lda < K || ldw < K) return EA_ERR_BAD_SHAPE;
  if ((strideA & 7) || (strideW & 7)) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return EA_ERR_BAD_ARG;
  memset(&p, 0, sizeof(p));
  int st = fill_epilogue(p.epi, epi, M, N);
  if (st != EA_OK) return st;
  p.a1 = (const f16*)A;
  p.lda = lda;
  p.conv = 0;
  p.w = (const f16*)W;
  p.ldw = ldw;
  p.M = M; p.N = N; p.K = K;
  p.batch = batch;
  p.strideA = strideA; p.strideW = strideW; p.strideC = strideC; p.strideR = strideR;
  if (epi->acc_scale_k) {   // K-concatenated split operands
    if (epi->acc_scale_k < 0 || epi->acc_scale_k >= K || (epi->acc_scale_k % EA_BK)) return EA_ERR_BAD_SHAPE;
    p.acc_scale_kt = epi->acc_scale_k / EA_BK;
    p.acc_scale = epi->acc_scale;
  }
  return EA_OK;
}

extern "C" int ea_gemm_f16_pair(const void* A0, const void* A1, int lda, const void* W0, const void* W1, int ldw, int M, int N, int K,
                                const ea_epilogue* epi0, const ea_epilogue* epi1, void* workspace, size_t ws_bytes, void* stream) {
  EaGemmParams p, q;
  int st = setup_gemm(p, A0, lda, W0, ldw, M, N, K, 1, 0, 0, 0, 0, epi0);
  if (st == EA_OK) st = setup_gemm(q, A1, lda, W1, ldw, M, N, K, 1, 0, 0, 0, 0, epi1);
  if (st != EA_OK) return st;
  return launch_pair(p, q, workspace, ws_bytes, stream);
}

static int setup_conv(EaGemmParams& p, const ea_conv_src* s, const void* W, int Cout, const ea_epilogue* epi) {
  if (!s || !s->x1 || !W) return EA_ERR_BAD_ARG;
  if (s->ksize != 1 && s->ksize != 3) return EA_ERR_UNSUPPORTED;
  if (s->stride != 1 && s->stride != 2) return EA_ERR_UNSUPPORTED;
  if (s->B <= 0 || s->Hin <= 0 || s->Win <= 0 || s->Hout <= 0 || s->Wout <= 0 || Cout <= 0) return EA_ERR_BAD_SHAPE;
  if (s->c1 <= 0 || (s->c1 & 7) || (s->c2 & 7) || s->c2 < 0) return EA_ERR_BAD_SHAPE;
  if (s->c2 > 0 && !s->x2) return EA_ERR_BAD_ARG;
  if (((uintptr_t)s->x1 & 15) || ((uintptr_t)s->x2 & 15) || ((uintptr_t)s->x2_add & 15) || ((uintptr_t)W & 15))
    return EA_ERR_BAD_ARG;
  const long long M = (long long)s->B * s->Hout * s->Wout;
  if (M > 0x7fffffffLL) return EA_ERR_BAD_SHAPE;
  const int ctot = s->c1 + s->c2;
  memset(&p, 0, sizeof(p));
  int st = fill_epilogue(p.epi, epi, (int)M, Cout);
  if (st != EA_OK) return st;
  p.a1 = (const f16*)s->x1; p.c1 = s->c1;
  p.a2 = s->c2 > 0 ? (const f16*)s->x2 : nullptr; p.c2 = s->c2;
  p.a2_add = s->c2 > 0 ? (const f16*)s->x2_add : nullptr;
  p.conv = 1;
  p.ksize = s->ksize;
  p.Hin = s->Hin; p.Win = s->Win; p.Hout = s->Hout; p.Wout = s->Wout;
  p.stride = s->stride; p.pad = s->pad; p.ups = s->ups;
  p.w = (const f16*)W;
  p.K = s->ksize * s->ksize * ctot;
  p.ldw = p.K;
  p.M = (int)M; p.N = Cout;
  p.batch = 1;
  return EA_OK;
}

extern "C" int ea_gemm_f16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int batch,
                           long long strideA, long long strideW, long long strideC, long long strideR,
                           const ea_epilogue* epi, void* workspace, size_t ws_bytes, void* stream) {
  EaGemmParams p;
  int st = setup_gemm(p, A, lda, W, ldw, M, N, K, batch, strideA, strideW, strideC, strideR, epi);
  if (st != EA_OK) return st;
  return launch_gemm(p, workspace, ws_bytes, stream);
}

extern "C" int ea_conv2d_f16_pair(const ea_conv_src* src0, const ea_conv_src* src1, const void* W0, const void* W1, int Cout,
                                  const ea_epilogue* epi0, const ea_epilogue* epi1, void* workspace, size_t ws_bytes, void* stream) {
  EaGemmParams p, q;
  int st = setup_conv(p, src0, W0, Cout, epi0);
  if (st == EA_OK) st = setup_conv(q, src1, W1, Cout, epi1);
  if (st != EA_OK) return st;
  return launch_pair(p, q, workspace, ws_bytes, stream);
}

extern "C" int ea_conv2d_f16(const ea_conv_src* src, const void* W, int Cout, const ea_epilogue* epi,
                             void* workspace, size_t ws_bytes, void* stream) {
  EaGemmParams p;
  int st = setup_conv(p, src, W, Cout, epi);
  if (st != EA_OK) return st;
  return launch_gemm(p, workspace, ws_bytes, stream);
}
