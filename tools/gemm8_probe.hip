// gemm8_probe.hip -- stand-alone prototype of the NEXT contraction main loop: 256 x 256 x 64 tiles, 8 waves
// (2 x 4, 128 x 64 per wave), 128 KiB of LDS (two K tiles), the K tile cut into four phases (one 64 x 32 accumulator
// quadrant = 16 MFMAs each) with one LDS-DMA half tile staged per phase and counted vmcnt (never 0 in the steady state) --
// the structure MI355X guide section 5 "256^2 8-phase template" describes, written from that description.
//   C[M][N] (fp16) = A[M][K] W[N][K]^T, fp32 accumulation; M, N multiples of 256, K of 64.  No epilogue options, no im2col:
//   this measures the loop, it is not wired into the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm8_probe.hip -o tools/gemm8_probe
//   tools/gemm8_probe [M N K]...     (default: the shapes of DESIGN section 8)
//
// Half tiles are cut by USE, not by position: A-h0 = the rows both wave rows read for their first accumulator quadrant
// (rows [0,64) and [128,192)), A-h1 the rest; B-h0 = the 32 columns each of the four wave columns reads first.  Quadrant
// order (A0,B0) (A0,B1) (A1,B1) (A1,B0), so a K tile needs A-h0 + B-h0 at its first phase, B-h1 at the second, A-h1 at the
// third; staging order for the next tile is the same, one half per phase, and every phase boundary is `s_waitcnt vmcnt(4)`
// (the two youngest halves stay in flight) + one barrier.  A half is read only in phases after the barrier that follows
// the wait which retired it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int KT_BYTES = (BM + BN) * BK * 2;       // 64 KiB per K tile
constexpr unsigned BUF_BYTES = 0xFFFFFFFFu;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, char* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// rows of half `h` (0 / 1) of the A region, 8-row group g (0..15): wave row (g >> 3), 64-row quadrant h
__device__ __forceinline__ int a_half_row(int h, int g) { return (g >> 3) * 128 + h * 64 + (g & 7) * 8; }
// rows (= output columns) of half `h` of the B region, group g: wave column (g >> 2), 32-column quadrant h
__device__ __forceinline__ int b_half_row(int h, int g) { return (g >> 2) * 64 + h * 32 + (g & 3) * 8; }

template <int FLAGS>
__global__ __launch_bounds__(512, 1) void gemm8(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K,
                                                int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs; give each XCD a contiguous slice
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + orig / 8;
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = K / BK;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, BUF_BYTES, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, BUF_BYTES, 0x00020000);

  // staging: half h, instruction i -> 8-row group g = 2 * wave + i; lane l -> row l >> 3, 16-byte slot l & 7
  unsigned a_voff[2][2], b_voff[2][2];
  int a_lds[2][2], b_lds[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wave + i;
      const int ar = a_half_row(h, g), br = b_half_row(h, g);
      const int rl = lane >> 3, slot = lane & 7;
      a_voff[h][i] = ((unsigned)(m0 + ar + rl) * (unsigned)K + (unsigned)((slot ^ swz(ar + rl)) * 8)) * 2u;
      b_voff[h][i] = ((unsigned)(n0 + br + rl) * (unsigned)K + (unsigned)((slot ^ swz(br + rl)) * 8)) * 2u;
      a_lds[h][i] = ar * 128;                      // wave-uniform
      b_lds[h][i] = BM * 128 + br * 128;
    }
  auto stage_a = [&](int h, int t) {
    char* buf = smem + (t & 1) * KT_BYTES;
    const unsigned soff = (unsigned)t * BK * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(ra, a_voff[h][i], soff, buf + a_lds[h][i]);
  };
  auto stage_b = [&](int h, int t) {
    char* buf = smem + (t & 1) * KT_BYTES;
    const unsigned soff = (unsigned)t * BK * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(rw, b_voff[h][i], soff, buf + b_lds[h][i]);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;
  f16x8 fa[2][4][2], fb[2][2][2];
  constexpr bool PF = (FLAGS & 8) != 0;      // fragment registers for both quadrants; reads issued one phase early
  auto read_a = [&](const char* buf, int qa) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wm * 128 + qa * 64 + i * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[PF ? qa : 0][i][ks] = *reinterpret_cast<const f16x8*>(buf + r * 128 + (((ks * 4 + lq) ^ swz(r)) << 4));
    }
  };
  auto read_b = [&](const char* buf, int qb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wn * 64 + qb * 32 + j * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[PF ? qb : 0][j][ks] = *reinterpret_cast<const f16x8*>(buf + BM * 128 + r * 128 + (((ks * 4 + lq) ^ swz(r)) << 4));
    }
  };
  // D^T = W A^T: a lane ends up with 4 consecutive output columns of one output row
  auto mma = [&](int qa, int qb) {
    if (FLAGS & 1) barrier();                 // the template's first barrier: every wave enters its MFMA cluster together
    if (FLAGS & 4) __builtin_amdgcn_sched_barrier(0);
    if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[PF ? qb : 0][j][ks], fa[PF ? qa : 0][i][ks], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0);
    if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(0);
    if (FLAGS & 4) __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: the four halves of K tile 0, in use order
  stage_a(0, 0);
  stage_b(0, 0);
  stage_b(1, 0);
  stage_a(1, 0);
  if (PF) wait_dma<2>(); else wait_dma<4>();    // PF: phase 0 also reads B-h1
  barrier();

  if (!PF) {
  for (int t = 0; t < nkt; ++t) {
    const char* buf = smem + (t & 1) * KT_BYTES;
    const bool more = t + 1 < nkt;
    // phase 0: quadrant (A0, B0); stage A-h0 of the next tile; next phase needs B-h1 of this tile
    read_a(buf, 0);
    read_b(buf, 0);
    if (more) stage_a(0, t + 1);
    wait_lds();
    mma(0, 0);
    if (more) wait_dma<4>(); else wait_dma<2>();
    barrier();
    // phase 1: (A0, B1); stage B-h0; next phase needs A-h1
    read_b(buf, 1);
    if (more) stage_b(0, t + 1);
    wait_lds();
    mma(0, 1);
    if (more) wait_dma<4>(); else wait_dma<0>();
    barrier();
    // phase 2: (A1, B1); stage B-h1; next phase needs nothing new
    read_a(buf, 1);
    if (more) stage_b(1, t + 1);
    wait_lds();
    mma(1, 1);
    barrier();
    // phase 3: (A1, B0); stage A-h1; the next tile's first phase needs its A-h0 and B-h0
    read_b(buf, 0);
    if (more) stage_a(1, t + 1);
    wait_lds();
    mma(1, 0);
    if (more) wait_dma<4>();
    barrier();
  }
  } else {
  // Fragment reads one phase ahead of their MFMAs (both quadrants' registers live): the LDS round trip of phase p + 1
  // runs under phase p's MFMAs.  A half must therefore be retired one phase earlier than above: order of need is
  // A-h0, B-h0 (phase 3 of the previous tile, for phase 0), B-h1 (phase 0, for phase 1), A-h1 (phase 1, for phase 2).
  // Staging order stays A-h0, B-h0, B-h1, A-h1, one per phase, so at the wait of phase p the halves younger than the
  // one needed are: P3 -> [B-h1, A-h1](t+1)... see the counts at each wait.
  read_a(smem, 0);
  read_b(smem, 0);
  for (int t = 0; t < nkt; ++t) {
    const char* buf = smem + (t & 1) * KT_BYTES;
    const char* nbuf = smem + ((t + 1) & 1) * KT_BYTES;
    const bool more = t + 1 < nkt;
    // phase 0: MFMAs (A0, B0); reads B1 (needs B-h1(t): retired before this phase -- see phase 3 / prologue)
    read_b(buf, 1);
    if (more) stage_a(0, t + 1);
    mma(0, 0);
    // next phase reads A1: needs A-h1(t); younger in flight: A-h0(t+1)
    if (more) wait_dma<2>(); else wait_dma<0>();
    wait_lds();
    barrier();
    // phase 1: MFMAs (A0, B1); reads A1
    read_a(buf, 1);
    if (more) stage_b(0, t + 1);
    mma(0, 1);
    wait_lds();
    barrier();
    // phase 2: MFMAs (A1, B1); no reads
    if (more) stage_b(1, t + 1);
    mma(1, 1);
    // next phase reads the next tile's A0, B0: needs A-h0(t+1), B-h0(t+1); younger: B-h1(t+1)
    if (more) wait_dma<2>();
    barrier();
    // phase 3: MFMAs (A1, B0); reads the next tile's A0 (set 0 is free since phase 1) ... B0 is still in use: after the MFMAs
    if (more) read_a(nbuf, 0);
    if (more) stage_a(1, t + 1);
    mma(1, 0);
    if (more) read_b(nbuf, 0);
    // next phase reads B1(t+1): needs B-h1(t+1); younger: A-h1(t+1)
    if (more) wait_dma<2>();
    wait_lds();
    barrier();
  }
  }

  // epilogue: fp16 rows straight from the registers, 8 bytes per lane and tile
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + lq * 4;
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)acc[i][j][e];
      *reinterpret_cast<f16x4*>(C + (long long)m * N + n) = o;
    }
  }
}


// ---- round 5: the same tile and halves, scheduled as the guide's template actually runs it: TWO barriers per phase
// (load part | barrier | MFMA part | barrier) and the two wave rows STAGGERED by one barrier, so that on every SIMD one wave
// sits in its 16-MFMA cluster (s_setprio 1) while its partner reads fragments and issues the next half tile -- the round-2
// prototype above ran all eight waves through the same part at the same time (the matrix pipe idles during every load part).
// Group g = wave row.  Global phase q = 4 t + p.  Load part L(q): fragment reads for M(q), DMA of half s(p) of tile t + 1
// (s = A-h0, B-h0, B-h1, A-h1), then vmcnt(4): the half staged in L(q - 2) has landed for this wave.  A half staged in L(q)
// is first read in L(q + 3) (L(q + 4) for A-h0) -- one barrier after the last wave's wait (the lagging group's L(q + 2)).
// WAR: a half's last read (L(q - 4 + r), r <= 2, retired by the lgkmcnt(0) in front of that phase's MFMAs) lies >= 2 barriers
// before its restaging in L(q).  B quadrant 0 stays in registers from phase 0 to phase 3.
//   FL bit 0: no stagger (both groups in lockstep)   bit 1: no s_setprio   bit 2: sched_barrier(0) around the MFMA cluster
template <int FL>
__global__ __launch_bounds__(512, 1) void gemm8s(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K,
                                                 int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
  int tm, tn;
  if (FL & 8) {            // grouped order: 8 row tiles x all column tiles per group (an XCD's chunk shares few A row panels)
    const int tiles_m = nwg / tiles_n, gm = tiles_m >= 8 ? 8 : tiles_m, per = gm * tiles_n;
    const int grp = wg / per, in = wg % per;
    const int rows = (grp * gm + gm <= tiles_m) ? gm : tiles_m - grp * gm;
    tm = __builtin_amdgcn_readfirstlane(grp * gm + in % rows); tn = __builtin_amdgcn_readfirstlane(in / rows);
  } else { tm = wg / tiles_n; tn = wg % tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = K / BK;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, BUF_BYTES, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, BUF_BYTES, 0x00020000);
  unsigned a_voff[2][2], b_voff[2][2];
  int a_lds[2][2], b_lds[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wave + i;
      const int ar = a_half_row(h, g), br = b_half_row(h, g);
      const int rl = lane >> 3, slot = lane & 7;
      a_voff[h][i] = ((unsigned)(m0 + ar + rl) * (unsigned)K + (unsigned)((slot ^ swz(ar + rl)) * 8)) * 2u;
      b_voff[h][i] = ((unsigned)(n0 + br + rl) * (unsigned)K + (unsigned)((slot ^ swz(br + rl)) * 8)) * 2u;
      a_lds[h][i] = ar * 128;
      b_lds[h][i] = BM * 128 + br * 128;
    }
  auto stage_a = [&](int h, int t) {
    char* buf = smem + (t & 1) * KT_BYTES;
    const unsigned soff = (unsigned)t * BK * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(ra, a_voff[h][i], soff, buf + a_lds[h][i]);
  };
  auto stage_b = [&](int h, int t) {
    char* buf = smem + (t & 1) * KT_BYTES;
    const unsigned soff = (unsigned)t * BK * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(rw, b_voff[h][i], soff, buf + b_lds[h][i]);
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, lq = lane >> 4;
  f16x8 fa[4][2], fb[2][2][2];
  auto read_a = [&](const char* buf, int qa) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wm * 128 + qa * 64 + i * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = *reinterpret_cast<const f16x8*>(buf + r * 128 + (((ks * 4 + lq) ^ swz(r)) << 4));
    }
  };
  auto read_b = [&](const char* buf, int qb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wn * 64 + qb * 32 + j * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[qb][j][ks] = *reinterpret_cast<const f16x8*>(buf + BM * 128 + r * 128 + (((ks * 4 + lq) ^ swz(r)) << 4));
    }
  };
  auto mma = [&](int qa, int qb) {
    wait_lds();
    if (FL & 4) __builtin_amdgcn_sched_barrier(0);
    if (!(FL & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[qb][j][ks], fa[i][ks], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0);
    if (!(FL & 2)) __builtin_amdgcn_s_setprio(0);
    if (FL & 4) __builtin_amdgcn_sched_barrier(0);
  };
  stage_a(0, 0);
  stage_b(0, 0);
  stage_b(1, 0);
  stage_a(1, 0);
  wait_dma<4>();
  barrier();
  if (!(FL & 1) && wm == 1) barrier();
  auto tile = [&](int t, auto more_c) {
    constexpr bool more = decltype(more_c)::value;
    const char* buf = smem + (t & 1) * KT_BYTES;
    read_a(buf, 0);
    read_b(buf, 0);
    if (more) { stage_a(0, t + 1); wait_dma<4>(); } else wait_dma<2>();
    barrier();
    mma(0, 0);
    barrier();
    read_b(buf, 1);
    if (more) { stage_b(0, t + 1); wait_dma<4>(); } else wait_dma<0>();
    barrier();
    mma(0, 1);
    barrier();
    read_a(buf, 1);
    if (more) { stage_b(1, t + 1); wait_dma<4>(); }
    barrier();
    mma(1, 1);
    barrier();
    if (more) { stage_a(1, t + 1); wait_dma<4>(); }
    barrier();
    mma(1, 0);
    barrier();
  };
  for (int t = 0; t + 1 < nkt; ++t) tile(t, std::true_type{});     // no "is there a next tile" test inside a phase
  tile(nkt - 1, std::false_type{});
  if (!(FL & 1) && wm == 0) barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + lq * 4;
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)acc[i][j][e];
      *reinterpret_cast<f16x4*>(C + (long long)m * N + n) = o;
    }
  }
}


template <int FLAGS>
static double run_shape(int M, int N, int K, bool check) {
  std::vector<f16> ha((size_t)M * K), hw((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f * 2.0f - 1.0f; };
  for (auto& v : ha) v = (f16)rnd();
  for (auto& v : hw) v = (f16)(rnd() * 0.25f);
  f16 *da, *dw, *dc;
  CK(hipMalloc(&da, ha.size() * 2));
  CK(hipMalloc(&dw, hw.size() * 2));
  CK(hipMalloc(&dc, (size_t)M * N * 2));
  CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  const int tiles_m = M / BM, tiles_n = N / BN;
  const int smem = 2 * KT_BYTES;
  auto kfn = FLAGS >= 256 ? gemm8s<(FLAGS >= 256 ? FLAGS - 256 : 0)> : gemm8<(FLAGS >= 256 ? 0 : FLAGS)>;
  CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  auto launch = [&]() { kfn<<<tiles_m * tiles_n, 512, smem>>>(da, dw, dc, M, N, K, tiles_n); };
  launch();
  CK(hipDeviceSynchronize());
  double max_err = 0.0;
  if (check) {
    std::vector<f16> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost));
    unsigned s2 = 777u;
    for (int it = 0; it < 4096; ++it) {
      s2 = s2 * 1664525u + 1013904223u;
      const int m = (s2 >> 8) % M;
      s2 = s2 * 1664525u + 1013904223u;
      const int n = (s2 >> 8) % N;
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)m * K + k] * (double)hw[(size_t)n * K + k];
      const double err = fabs(ref - (double)hc[(size_t)m * N + n]) / (1.0 + fabs(ref));
      if (err > max_err) max_err = err;
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 5; ++w) launch();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 10 < best) best = ms / 10;
  }
  const double tf = 2.0 * M * N * K / (best * 1e-3) / 1e12;
  printf("{\"kernel\": \"gemm8_probe 256x256x64 8 waves\", \"flags\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"us\": %.2f, \"tflops\": %.1f, \"mfma_frac\": %.4f, \"max_rel_err_sampled\": %.3g}\n",
         FLAGS, M, N, K, best * 1e3, tf, tf / 2500.0, max_err);
  CK(hipFree(da));
  CK(hipFree(dw));
  CK(hipFree(dc));
  return tf;
}

template <int FLAGS>
static void sweep(int argc, char** argv) {
  if (argc >= 4) {
    for (int i = 1; i + 2 < argc; i += 3) run_shape<FLAGS>(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), true);
    return;
  }
  run_shape<FLAGS>(512, 512, 256, true);          // race / layout screen at small sizes first
  run_shape<FLAGS>(1024, 768, 320, true);
  run_shape<FLAGS>(4096, 4096, 4096, true);       // the guide's calibration shape
  run_shape<FLAGS>(8192, 8192, 8192, false);
  run_shape<FLAGS>(16384, 3840, 1280, true);      // SAM qkv
  run_shape<FLAGS>(16384, 1280, 5120, true);
  run_shape<FLAGS>(16384, 5120, 1280, true);      // SAM MLP in
  run_shape<FLAGS>(65536, 512, 4608, true);       // VAE conv3x3 128^2 512 -> 512 as a plain GEMM
  run_shape<FLAGS>(8192, 5120, 640, true);        // level-1 GEGLU projection
  run_shape<FLAGS>(32768, 256, 2880, true);       // a level-0 3x3 convolution's K
  run_shape<FLAGS>(32768, 2560, 320, true);       // level-0 GEGLU projection
}

int main(int argc, char** argv) {
  sweep<0>(argc, argv);         // round-2 prototype (one barrier per phase, lockstep)
  sweep<256>(argc, argv);       // round 5: two barriers per phase, staggered wave rows, s_setprio
  sweep<256 + 8>(argc, argv);   //   ... grouped tile order (8 row tiles x all column tiles per group)
  sweep<256 + 2>(argc, argv);   //   ... no s_setprio
  return 0;
}
