// gemm4w_probe.hip -- stand-alone prototype of a LARGE-WAVE-TILE main loop (round 5, after ea_gemm8): 256 x 256 x 64 tiles, FOUR waves
// (2 x 2, one per SIMD), 128 x 128 per wave = 8 x 8 MFMA tiles (v_mfma_f32_16x16x32_f16), 256 accumulator registers per lane.
//   C[M][N] (fp16) = A[M][K] W[N][K]^T, fp32 accumulation; M, N multiples of 256, K of 64.  No epilogue options: this measures the loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm4w_probe.hip -o tools/gemm4w_probe ;  tools/gemm4w_probe [M N K]...
// Why: ea_gemm8's 128 x 64 wave tiles read 24 KiB of fragments per wave and K tile -- 8 waves x 24 KiB + 64 KiB of staging = 256 KiB of
// LDS traffic per K tile against 2048 clocks of MFMA work: the LDS port (128 B/clk) is as busy as the matrix pipe.  128 x 128 wave
// tiles halve the B re-reads (4 x 32 KiB + 64 KiB = 192 KiB).  With ONE wave per SIMD nothing else covers a wave's stalls, so:
//  * operands are REGISTER-staged (buffer_load -> VGPR -> ds_write_b128): an LDS-DMA instruction holds the issuing wave ~100 clocks
//    per KiB, which a partner wave hides in ea_gemm8 and nothing hides here;
//  * one K tile = two K steps of 64 MFMAs; the next step's 16 fragment reads are issued between this step's MFMAs (register double
//    buffer), the next TILE's 16 global loads at the tile's start and its 16 LDS writes between the second step's MFMAs;
//  * one barrier per K tile (after the writes: the other stage is complete and this stage is free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int KT_BYTES = (BM + BN) * BK * 2;       // 64 KiB per K tile
constexpr unsigned BUF_BYTES = 0xFFFFFFFFu;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// FL bit 0: no sched_group_barrier pinning   bit 1: no s_setprio   bit 2: 32x32x16 MFMA instead of 16x16x32 (not implemented)
template <int FL>
__global__ __launch_bounds__(256, 1) void gemm4w(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K,
                                                 int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = K / BK;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, BUF_BYTES, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, BUF_BYTES, 0x00020000);

  // staging: thread -> 16-byte chunk `slot` of rows base + 32 i (i < 8) of the A and of the B region
  const int slot = tid & 7, rbase = tid >> 3;
  unsigned a_goff[8], b_goff[8];
  int s_lds[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = rbase + 32 * i;
    a_goff[i] = ((unsigned)(m0 + r) * (unsigned)K + (unsigned)(slot * 8)) * 2u;
    b_goff[i] = ((unsigned)(n0 + r) * (unsigned)K + (unsigned)(slot * 8)) * 2u;
    s_lds[i] = r * 128 + ((slot ^ swz(r)) << 4);
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 sa[8], sb[8];
  auto gload = [&](int t) {
    const unsigned soff = (unsigned)t * BK * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, a_goff[i], soff, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_goff[i], soff, 0);
  };
  auto lstore = [&](int t) {
    char* buf = smem + (t & 1) * KT_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(buf + s_lds[i]) = sa[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(buf + BM * 128 + s_lds[i]) = sb[i];
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, lq = lane >> 4;
  f16x8 fa[2][8], fb[2][8];
  // per-lane fragment row offsets (the swizzle of row r = base + 16 i + l15 does not depend on i when base % 16 == 0: (r >> 1) & 7)
  const int a_row0 = wm * 128 + l15, b_row0 = wn * 128 + l15;
  auto read_frags = [&](const char* buf, int ks, int set) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = a_row0 + 16 * i;
      fa[set][i] = *reinterpret_cast<const f16x8*>(buf + r * 128 + (((ks * 4 + lq) ^ swz(r)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = b_row0 + 16 * j;
      fb[set][j] = *reinterpret_cast<const f16x8*>(buf + BM * 128 + r * 128 + (((ks * 4 + lq) ^ swz(r)) << 4));
    }
  };
  auto mma = [&](int set) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
  };

  // prologue: tile 0 through the registers into stage 0
  gload(0);
  lstore(0);
  __syncthreads();
  read_frags(smem, 0, 0);
  for (int t = 0; t < nkt; ++t) {
    const char* buf = smem + (t & 1) * KT_BYTES;
    const bool more = t + 1 < nkt;
    if (more) gload(t + 1);
    // K step 0: its MFMAs with the 16 fragment reads of K step 1 between them
    read_frags(buf, 1, 1);
    if (!(FL & 2)) __builtin_amdgcn_s_setprio(1);
    mma(0);
    if (!(FL & 1)) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
      }
    }
    // K step 1: its MFMAs with the next tile's 16 LDS writes between them (the other stage: nobody reads it now)
    if (more) lstore(t + 1);
    mma(1);
    if (!(FL & 1)) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // 1 DS write
      }
    }
    if (!(FL & 2)) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (more) read_frags(smem + ((t + 1) & 1) * KT_BYTES, 0, 0);
  }

  // epilogue: fp16 rows straight from the registers, 8 bytes per lane and tile
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wn * 128 + j * 16 + lq * 4;
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
  auto kfn = gemm4w<FLAGS>;
  CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  auto launch = [&]() { kfn<<<tiles_m * tiles_n, 256, smem>>>(da, dw, dc, M, N, K, tiles_n); };
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
  printf("{\"kernel\": \"gemm4w_probe 256x256x64 4 waves (128x128 per wave)\", \"flags\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"us\": %.2f, \"tflops\": %.1f, \"mfma_frac\": %.4f, \"max_rel_err_sampled\": %.3g}\n",
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
  run_shape<FLAGS>(4096, 4096, 4096, true);
  run_shape<FLAGS>(8192, 8192, 8192, false);
  run_shape<FLAGS>(16384, 3840, 1280, true);      // SAM qkv
  run_shape<FLAGS>(16384, 5120, 1280, true);      // SAM MLP in
  run_shape<FLAGS>(65536, 512, 4608, true);       // VAE conv3x3 128^2 512 -> 512 as a plain GEMM
}

int main(int argc, char** argv) {
  sweep<0>(argc, argv);
  sweep<1>(argc, argv);         // compiler's own schedule
  sweep<2>(argc, argv);         // pinned, no s_setprio
  return 0;
}
