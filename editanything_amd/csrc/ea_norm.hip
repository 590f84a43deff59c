// ea_norm.hip -- HBM-bound normalisation kernels (GroupNorm+SiLU, LayerNorm, row softmax).
//
// GroupNorm32 of the reference computes in fp32 and casts back
// (ldm/modules/diffusionmodules/util.py:217-219); the SpatialTransformer / VAE
// norms use eps 1e-6 (ldm/modules/attention.py:88-89, model.py:46-47).  Here the
// activation is NHWC fp16, so one group's channels are a short contiguous run
// per pixel: workgroups stream whole pixel rows with 16-B loads (thread <-> a
// fixed channel octet), reduce per channel, then per group.  Pass 1 writes
// per-(sample, chunk, group) partial sums; pass 2 folds them into per-channel
// scale/shift, applies (+SiLU) and writes fp16.  The input may be the virtual
// channel-concat cat(x1, x2 [+ x2_add]) of the UNet decoder (cldm/cldm.py:38-41).
#include "ea_platform.h"
#include "../../include/editanything_hip.h"
#include <string.h>

#define EA_GN_MAX_CHUNKS 128

namespace {

struct GnParams {
  const f16* x1; int c1;
  const f16* x2; int c2;
  const f16* x2_add;
  const float* gamma; const float* beta;
  f16* out;
  float* partial;  // [B][nchunk][groups][2]
  int B, HW, C, groups, cpg;
  int V, R;        // channel octets per pixel, pixel rows per workgroup pass
  int nchunk, chunk_px;     // stats pass chunking (partials per sample)
  int anchunk, achunk_px;   // apply pass chunking
  float eps;
  int silu;
};

// One thread's view of the (possibly two-source) input: its channel octet lives in ONE source for every pixel, so the
// source choice is a per-thread pointer + stride fixed before the pixel loop, and a pixel's load is unconditional.
// (Written as a branch per load -- `if (c0 < c1) load x1 else load x2` -- hipcc parks an s_waitcnt vmcnt(0) behind every
// load and the "N loads in flight" of the loops below become N serial round trips: the round-2 profile had the
// single-pass kernel at 18 us for 160 KB per workgroup.)
template <bool ADD>
struct GnSrc {
  const f16* src; long long stride;
  const f16* add;    // ADD only: the addend of an x2-side thread; an x1-side thread re-reads its own vector and drops it
  bool second;
  __device__ __forceinline__ void init(const GnParams& p, int c0) {
    second = c0 >= p.c1;
    src = second ? p.x2 + (c0 - p.c1) : p.x1 + c0;
    stride = second ? p.c2 : p.c1;
    add = (ADD && second) ? p.x2_add + (c0 - p.c1) : src;
  }
  __device__ __forceinline__ f16x8 load(long long pix) const {
    f16x8 v = ea_ld8(src + pix * stride);
    if (ADD) {
      const f16x8 a = ea_ld8(add + pix * stride);
      const f16x8 sum = v + a;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = second ? sum[j] : v[j];
    }
    return v;
  }
};

// Pass 1: per-(sample, chunk, group) partial sums.  Four pixels in flight per thread before the first use.
template <bool ADD>
__global__ void ea_gn_stats_kernel(GnParams p) {
  EA_SMEM(smem);
  float* chs = reinterpret_cast<float*>(smem);  // [R][C]
  float* chq = chs + p.R * p.C;                 // [R][C]
  const int tid = threadIdx.x;
  const int v = tid % p.V, pr = tid / p.V;
  const int c0 = v * 8;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int p_begin = chunk * p.chunk_px;
  int p_end = p_begin + p.chunk_px;
  if (p_end > p.HW) p_end = p.HW;
  GnSrc<ADD> in;
  in.init(p, c0);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
  const long long pix0 = (long long)b * p.HW;
  // every trip loads 4 pixels unconditionally (rows past the chunk end re-read its last pixel) and masks the sums.
  // The trip count is the SAME for every thread (the bound does not involve `pr`; a thread whose pixels are all past the end runs
  // a fully masked trip: + 0 to every sum): the loop then carries no per-lane EXEC updates.  Round 4: with the per-thread bound
  // `px < p_end` the last instructions of an iteration were the sum-of-squares updates, directly followed by the EXEC update that
  // retires the lanes that are done -- and beside another stream's generic-kernel launches lanes 48..63 of a wave occasionally
  // lost exactly those last updates (sums intact, sums of squares a few terms short: profiles/r04_pipelined_race.jsonl).
#if defined(EA_GN_STATS_LOOP) && EA_GN_STATS_LOOP
#include "../../tools/kernels/ea_gn_stats_loops.h"   // reproducer builds only (tools/build_gn_repro.sh): the round-3 loop forms
#else
  for (int base = p_begin; base < p_end; base += 4 * p.R) {
    const int px = base + pr;
    f16x8 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = px + u * p.R;
      x[u] = in.load(pix0 + (pu < p_end ? pu : p_end - 1));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float m = (px + u * p.R < p_end) ? 1.0f : 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)x[u][j] * m;
        s[j] += f;
        q[j] += f * f;
      }
    }
  }
#endif
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    chs[pr * p.C + c0 + j] = s[j];
    chq[pr * p.C + c0 + j] = q[j];
  }
  __syncthreads();
  // fold rows: thread c sums column c over the R rows (conflict-free), then one thread per group sums cpg columns
  for (int c = tid; c < p.C; c += blockDim.x) {
    float cs = 0.0f, cq = 0.0f;
    for (int rr = 0; rr < p.R; ++rr) {
      cs += chs[rr * p.C + c];
      cq += chq[rr * p.C + c];
    }
    chs[c] = cs;   // row 0 is only read by this thread in the loop above
    chq[c] = cq;
  }
  __syncthreads();
  for (int g = tid; g < p.groups; g += blockDim.x) {
    float gs = 0.0f, gq = 0.0f;
    for (int c = g * p.cpg; c < (g + 1) * p.cpg; ++c) {
      gs += chs[c];
      gq += chq[c];
    }
    float* dst = p.partial + (((long long)b * p.nchunk + chunk) * p.groups + g) * 2;
    dst[0] = gs;
    dst[1] = gq;
  }
}

template <bool SILU>
__device__ __forceinline__ f16x8 gn_finish(const f16x8& x, const float (&a)[8], const float (&sh)[8]) {
  f16x8 y;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float f = (float)x[j] * a[j] + sh[j];
    if (SILU) f = ea_silu(f);
    y[j] = (f16)f;
  }
  return y;
}

// Pass 2: fold the partials (in parallel, fixed order), normalise (+SiLU), write fp16.  Its own, finer chunking.
template <bool ADD, bool SILU>
__global__ void ea_gn_apply_kernel(GnParams p) {
  EA_SMEM(smem);
  float* part = reinterpret_cast<float*>(smem);   // [nsub][groups][2]
  float* gst = part + 2 * p.groups * (blockDim.x / p.groups > 0 ? blockDim.x / p.groups : 1);   // [groups][2]: mean, rstd
  const int tid = threadIdx.x;
  const int v = tid % p.V, pr = tid / p.V;
  const int c0 = v * 8;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int p_begin = chunk * p.achunk_px;
  int p_end = p_begin + p.achunk_px;
  if (p_end > p.HW) p_end = p.HW;
  {
    int nsub = blockDim.x / p.groups;
    if (nsub < 1) nsub = 1;
    for (int t = tid; t < nsub * p.groups; t += blockDim.x) {
      const int g = t % p.groups, sub = t / p.groups;
      // chunks sub, sub + nsub, ... in that order, four loads in flight per trip (clamped index, masked sum); the trip count
      // does not depend on the thread (see ea_gn_stats_kernel: no per-lane EXEC update behind the accumulation)
      float gs = 0.0f, gq = 0.0f;
      for (int ch0 = 0; ch0 < p.nchunk; ch0 += 4 * nsub) {
        const int ch = ch0 + sub;
        float ps[4], pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cu = ch + u * nsub;
          const float* src = p.partial + (((long long)b * p.nchunk + (cu < p.nchunk ? cu : p.nchunk - 1)) * p.groups + g) * 2;
          ps[u] = src[0];
          pq[u] = src[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = ch + u * nsub < p.nchunk;
          gs += ok ? ps[u] : 0.0f;
          gq += ok ? pq[u] : 0.0f;
        }
      }
      part[(sub * p.groups + g) * 2] = gs;
      part[(sub * p.groups + g) * 2 + 1] = gq;
    }
    __syncthreads();
    const float inv_n = 1.0f / ((float)p.HW * (float)p.cpg);
    for (int g = tid; g < p.groups; g += blockDim.x) {
      float gs = 0.0f, gq = 0.0f;
      for (int sub = 0; sub < nsub; ++sub) {
        gs += part[(sub * p.groups + g) * 2];
        gq += part[(sub * p.groups + g) * 2 + 1];
      }
      const float mean = gs * inv_n;
      float var = gq * inv_n - mean * mean;
      var = var > 0.0f ? var : 0.0f;
      gst[g * 2] = mean;
      gst[g * 2 + 1] = 1.0f / sqrtf(var + p.eps);
    }
    __syncthreads();
  }
  float a[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    const int g = c / p.cpg;
    a[j] = gst[g * 2 + 1] * p.gamma[c];
    sh[j] = p.beta[c] - gst[g * 2] * a[j];
  }
  GnSrc<ADD> in;
  in.init(p, c0);
  const long long pix0 = (long long)b * p.HW;
  for (int px = p_begin + pr; px < p_end; px += 4 * p.R) {
    f16x8 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = px + u * p.R;
      x[u] = in.load(pix0 + (pu < p_end ? pu : p_end - 1));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = px + u * p.R;
      const f16x8 y = gn_finish<SILU>(x[u], a, sh);
      if (pu < p_end) ea_st8(p.out + (pix0 + pu) * p.C + c0, y);
    }
  }
}

// Single pass for activations whose (sample, channel slab) fits the register file of one workgroup: one read, one
// write, one launch.  A workgroup owns `SG` whole groups (slab = SG * cpg channels, a multiple of 8) of one sample for
// ALL pixels; thread <-> (channel octet of the slab, pixel row), up to GN_MAXIT pixels per thread held in registers.
constexpr int GN_MAXIT = 16;   // pixels per thread, upper bound (the kernel is instantiated for 2 / 4 / 8 / 16)
struct GnFusedParams {
  GnParams g;
  int slab_ch, slab_oct, sg;   // channels / octets / groups per slab
  int rows;                    // pixel rows per pass = threads / slab_oct
  int its;                     // pixels per thread actually needed: ceil(HW / rows) rounded up to 2 / 4 / 8 / 16
};

template <bool ADD, bool SILU, int MAXIT>
__global__ __launch_bounds__(512) void ea_gn_fused_kernel(GnFusedParams fp) {
  const GnParams& p = fp.g;
  EA_SMEM(smem);
  float* chs = reinterpret_cast<float*>(smem);      // [rows][slab_ch]
  float* chq = chs + fp.rows * fp.slab_ch;          // [rows][slab_ch]
  float* gst = chq + fp.rows * fp.slab_ch;          // [sg][2]
  const int tid = threadIdx.x;
  const int v = tid % fp.slab_oct, pr = tid / fp.slab_oct;
  const bool on = pr < fp.rows;
  const int slab = blockIdx.x, b = blockIdx.y;
  const int cl = v * 8;                   // channel within the slab
  const int c0 = slab * fp.slab_ch + cl;  // global channel
  const long long pix0 = (long long)b * p.HW;
  GnSrc<ADD> in;
  in.init(p, c0);
  // all MAXIT loads issue back to back: pixels past the end (and the spare threads of the last pixel row) re-read
  // the last pixel and are zeroed by a select afterwards
  f16x8 x[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int px = pr + it * fp.rows;
    x[it] = in.load(pix0 + (px < p.HW ? px : p.HW - 1));
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const bool ok = on && pr + it * fp.rows < p.HW;
    const f16x8 z = ea_zero8();
#pragma unroll
    for (int j = 0; j < 8; ++j) x[it][j] = ok ? x[it][j] : z[j];
  }
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)x[it][j];
      s[j] += f;
      q[j] += f * f;
    }
  if (on) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      chs[pr * fp.slab_ch + cl + j] = s[j];
      chq[pr * fp.slab_ch + cl + j] = q[j];
    }
  }
  __syncthreads();
  for (int c = tid; c < fp.slab_ch; c += blockDim.x) {
    float cs = 0.0f, cq = 0.0f;
    for (int rr = 0; rr < fp.rows; ++rr) {
      cs += chs[rr * fp.slab_ch + c];
      cq += chq[rr * fp.slab_ch + c];
    }
    chs[c] = cs;
    chq[c] = cq;
  }
  __syncthreads();
  if (tid < fp.sg) {
    float gs = 0.0f, gq = 0.0f;
    for (int c = tid * p.cpg; c < (tid + 1) * p.cpg; ++c) {
      gs += chs[c];
      gq += chq[c];
    }
    const float inv_n = 1.0f / ((float)p.HW * (float)p.cpg);
    const float mean = gs * inv_n;
    float var = gq * inv_n - mean * mean;
    var = var > 0.0f ? var : 0.0f;
    gst[tid * 2] = mean;
    gst[tid * 2 + 1] = 1.0f / sqrtf(var + p.eps);
  }
  __syncthreads();
  if (!on) return;
  float a[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int g = (cl + j) / p.cpg;
    a[j] = gst[g * 2 + 1] * p.gamma[c0 + j];
    sh[j] = p.beta[c0 + j] - gst[g * 2] * a[j];
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int px = pr + it * fp.rows;
    const f16x8 y = gn_finish<SILU>(x[it], a, sh);
    if (px < p.HW) ea_st8(p.out + (pix0 + px) * p.C + c0, y);
  }
}

// Slab of the single-pass kernel: the fewest whole groups whose channels are a multiple of 8 and span >= 64 bytes,
// never straddling the x1 | x2 boundary of a two-source input.  Returns 0 when the single pass does not apply.
static int gn_fused_plan(const GnParams& p, GnFusedParams& fp) {
#ifdef EA_GN_NO_FUSED
  return 0;
#endif
  int sg = 0;
  for (int k = 1; k <= p.groups; ++k) {
    const int ch = k * p.cpg;
    if ((ch & 7) == 0 && ch >= 32 && (p.groups % k) == 0) { sg = k; break; }
  }
  if (!sg) return 0;
  // from 32 x 32 latents up the two streaming passes win (B 8, 640 ch: 15.0 vs 17.1 us; 1280 ch concat: 20.8 vs 22.7):
  // one workgroup per (sample, slab) leaves half the CUs idle and its phases run back to back
  if (p.HW > 256) return 0;
  const int slab_ch = sg * p.cpg;
  if (p.c2 > 0 && (p.c1 % slab_ch) != 0) return 0;
  const int slab_oct = slab_ch / 8;
  const long long octets = (long long)p.HW * slab_oct;      // 16-byte vectors per (sample, slab)
  int threads = 256;
  while (threads < 512 && (long long)(threads / slab_oct) * slab_oct * GN_MAXIT < octets) threads *= 2;
  const int rows = threads / slab_oct;
  if (rows < 1 || (long long)rows * GN_MAXIT < p.HW) return 0;
  fp.g = p;
  fp.sg = sg; fp.slab_ch = slab_ch; fp.slab_oct = slab_oct; fp.rows = rows;
  const int need = (p.HW + rows - 1) / rows;
  fp.its = need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
  return threads;
}

static int gn_plan(GnParams& p) {
  if (p.C <= 0 || (p.C & 7) || p.groups <= 0 || (p.C % p.groups) || (p.c1 & 7) || (p.c2 & 7)) return EA_ERR_BAD_SHAPE;
  p.cpg = p.C / p.groups;
  p.V = p.C / 8;
  if (p.V > 1024) return EA_ERR_UNSUPPORTED;
  int r = 256 / p.V;
  if (r < 1) r = 1;
  if (r > 32) r = 32;
  if (r > p.HW) r = p.HW;
  p.R = r;
  int target = 2048 / (p.B > 0 ? p.B : 1);
  if (target < 1) target = 1;
  int nchunk = p.HW / (r * 4);
  if (nchunk > target) nchunk = target;
  if (nchunk > EA_GN_MAX_CHUNKS) nchunk = EA_GN_MAX_CHUNKS;
  if (nchunk < 1) nchunk = 1;
  p.chunk_px = (p.HW + nchunk - 1) / nchunk;
  p.nchunk = (p.HW + p.chunk_px - 1) / p.chunk_px;
  // apply pass: ~8 pixels per thread, at most 4096 workgroups per sample
  int achunk = r * 8;
  if ((p.HW + achunk - 1) / achunk > 4096) achunk = (p.HW + 4095) / 4096;
  p.achunk_px = achunk;
  p.anchunk = (p.HW + achunk - 1) / achunk;
  return EA_OK;
}

// ------------------------------------------------------------------ LayerNorm
struct LnParams {
  const void* x; int in_f32;
  const float* gamma; const float* beta;
  f16* out;
  int M, C;
  float eps;
  const int* out_rows;   // optional: output row of input row m (negative = drop the row); NULL = identity
};

// One wave per row, ROWS rows per wave in flight (all their loads are issued before the first reduction: the kernel
// is latency-bound, not bandwidth-bound, at one row per wave); two-pass in registers (mean, then centred variance),
// as torch does.  MAXV = 16-byte vectors per lane: C <= 512 * MAXV.
template <int MAXV, int ROWS>
__global__ __launch_bounds__(256) void ea_layernorm_kernel(LnParams p) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * ROWS;
  const int nv = p.C / 8;
  float vals[ROWS][MAXV][8];
  float sum[ROWS];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const int row = row0 + rr;
    const int r = row < p.M ? row : p.M - 1;
    sum[rr] = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
        if (p.in_f32) {
          const float* src = (const float*)p.x + (long long)r * p.C + v * 8;
          f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { vals[rr][i][j] = lo[j]; vals[rr][i][4 + j] = hi[j]; }
        } else {
          f16x8 h = ea_ld8((const f16*)p.x + (long long)r * p.C + v * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) vals[rr][i][j] = (float)h[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[rr][i][j] = 0.0f;
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum[rr] += vals[rr][i][j];
    sum[rr] = ea_wave_sum(sum[rr]);
  }
  float mean[ROWS], rstd[ROWS];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    mean[rr] = sum[rr] / (float)p.C;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = vals[rr][i][j] - mean[rr];
          sq += d * d;
        }
      }
    }
    sq = ea_wave_sum(sq);
    rstd[rr] = 1.0f / sqrtf(sq / (float)p.C + p.eps);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      float ga[8], be[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { ga[j] = p.gamma[v * 8 + j]; be[j] = p.beta[v * 8 + j]; }
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr) {
        const int row = row0 + rr;
        if (row < p.M) {
          const int orow = p.out_rows ? p.out_rows[row] : row;
          if (orow < 0) continue;
          f16x8 y;
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (f16)((vals[rr][i][j] - mean[rr]) * rstd[rr] * ga[j] + be[j]);
          ea_st8(p.out + (long long)orow * p.C + v * 8, y);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- row softmax
// One workgroup per row.  Rows of up to 8192 columns (a multiple of 4) are read ONCE with 16-byte loads and kept in
// registers between the max, the sum and the write (the SAM decoder's token -> image scores: 65536 rows x 4096 per image,
// 1 GB in / 0.5 GB out per call -- the three-pass scalar form ran at 2.3 TB/s); anything else takes the generic loop.
template <int NV>
__device__ __forceinline__ void ea_softmax_row_regs(const float* src, f16* dst, int cols, float scale, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 v[NV];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 256 + tid) * 4;
    const int cc = c < cols ? c : 0;                       // clamped, unconditional load (guide section 5 trap (c))
    v[i] = *reinterpret_cast<const f32x4*>(src + cc);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[i][r] = c < cols ? v[i][r] * scale : -INFINITY; m = fmaxf(m, v[i][r]); }
  }
  m = ea_wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[i][r] = ea_expf(v[i][r] - m); s += v[i][r]; }
  s = ea_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / s;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 256 + tid) * 4;
    if (c < cols) {
      f16x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (f16)(v[i][r] * inv);
      *reinterpret_cast<f16x4*>(dst + c) = h;
    }
  }
}

__global__ __launch_bounds__(256) void ea_softmax_rows_kernel(const float* x, f16* out, int rows, int cols, float scale) {
  EA_SMEM(smem);
  float* red = reinterpret_cast<float*>(smem);  // [8]
  const int row = blockIdx.x;
  if (row >= rows) return;
  const float* src = x + (long long)row * cols;
  f16* dst = out + (long long)row * cols;
  if ((cols & 3) == 0 && cols <= 8192 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0) {
    if (cols <= 1024) ea_softmax_row_regs<1>(src, dst, cols, scale, red);
    else if (cols <= 4096) ea_softmax_row_regs<4>(src, dst, cols, scale, red);
    else ea_softmax_row_regs<8>(src, dst, cols, scale, red);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -INFINITY;
  for (int c = tid; c < cols; c += 256) m = fmaxf(m, src[c] * scale);
  m = ea_wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
  for (int c = tid; c < cols; c += 256) s += ea_expf(src[c] * scale - m);
  s = ea_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / s;
  for (int c = tid; c < cols; c += 256) dst[c] = (f16)(ea_expf(src[c] * scale - m) * inv);
}

}  // namespace

namespace {
template <bool ADD, bool SILU>
int gn_fused_launch_its(const GnFusedParams& fp, dim3 grid, int threads, int smem, void* stream) {
  void (*kf)(GnFusedParams) = fp.its == 2 ? ea_gn_fused_kernel<ADD, SILU, 2> : fp.its == 4 ? ea_gn_fused_kernel<ADD, SILU, 4>
                            : fp.its == 8 ? ea_gn_fused_kernel<ADD, SILU, 8> : ea_gn_fused_kernel<ADD, SILU, 16>;
  ea_allow_big_lds(kf, smem);
  EA_LAUNCH(kf, grid, dim3(threads, 1, 1), smem, stream, fp);
  return ea_launch_status();
}
int gn_fused_launch(const GnFusedParams& fp, bool add, bool silu, dim3 grid, int threads, int smem, void* stream) {
  if (add) return silu ? gn_fused_launch_its<true, true>(fp, grid, threads, smem, stream) : gn_fused_launch_its<true, false>(fp, grid, threads, smem, stream);
  return silu ? gn_fused_launch_its<false, true>(fp, grid, threads, smem, stream) : gn_fused_launch_its<false, false>(fp, grid, threads, smem, stream);
}
}  // namespace

extern "C" size_t ea_groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
  if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
  return (size_t)B * EA_GN_MAX_CHUNKS * groups * 2 * sizeof(float);
}

extern "C" int ea_groupnorm_f16(const void* x1, int c1, const void* x2, int c2, const void* x2_add,
                                const float* gamma, const float* beta, void* out, int B, int HW, int groups,
                                float eps, int silu, void* workspace, size_t ws_bytes, void* stream) {
  if (!x1 || !gamma || !beta || !out || !workspace) return EA_ERR_BAD_ARG;
  if (c2 > 0 && !x2) return EA_ERR_BAD_ARG;
  if (B <= 0 || HW <= 0 || c1 <= 0 || c2 < 0) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)x1 & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)x2_add & 15) || ((uintptr_t)out & 15)) return EA_ERR_BAD_ARG;
  GnParams p;
  memset(&p, 0, sizeof(p));
  p.x1 = (const f16*)x1; p.c1 = c1;
  p.x2 = c2 > 0 ? (const f16*)x2 : nullptr; p.c2 = c2;
  p.x2_add = c2 > 0 ? (const f16*)x2_add : nullptr;
  p.gamma = gamma; p.beta = beta;
  p.out = (f16*)out;
  p.B = B; p.HW = HW; p.C = c1 + c2; p.groups = groups;
  p.eps = eps; p.silu = silu;
  int st = gn_plan(p);
  if (st != EA_OK) return st;
  if (ws_bytes < ea_groupnorm_workspace_bytes(B, HW, p.C, groups)) return EA_ERR_WORKSPACE;
  p.partial = (float*)workspace;
  const bool add = p.x2_add != nullptr;
  GnFusedParams fp;
  const int fthreads = gn_fused_plan(p, fp);
  if (fthreads > 0) {
    dim3 fgrid(p.C / fp.slab_ch, B, 1);
    const int fsmem = (2 * fp.rows * fp.slab_ch + 2 * fp.sg) * (int)sizeof(float);
    return gn_fused_launch(fp, add, silu != 0, fgrid, fthreads, fsmem, stream);
  }
  dim3 grid(p.nchunk, B, 1), block(p.V * p.R, 1, 1);
  const int smem = 2 * p.R * p.C * (int)sizeof(float);
  auto k1 = add ? ea_gn_stats_kernel<true> : ea_gn_stats_kernel<false>;
  ea_allow_big_lds(k1, smem);
  EA_LAUNCH(k1, grid, block, smem, stream, p);
  st = ea_launch_status();
  if (st != EA_OK) return st;
  auto k2 = add ? (silu ? ea_gn_apply_kernel<true, true> : ea_gn_apply_kernel<true, false>)
                : (silu ? ea_gn_apply_kernel<false, true> : ea_gn_apply_kernel<false, false>);
  int nsub = (p.V * p.R) / groups;
  if (nsub < 1) nsub = 1;
  const int smem2 = (2 * groups * nsub + 2 * groups) * (int)sizeof(float);
  EA_LAUNCH(k2, dim3(p.anchunk, B, 1), block, smem2, stream, p);
  return ea_launch_status();
}

extern "C" int ea_groupnorm_apply_f16(const void* x, int C, const float* gamma, const float* beta, void* out, int B,
                                      int HW, int groups, float eps, int silu, const float* partial, int nchunk,
                                      void* stream) {
  if (!x || !gamma || !beta || !out || !partial) return EA_ERR_BAD_ARG;
  if (B <= 0 || HW <= 0 || C <= 0 || nchunk <= 0 || nchunk > EA_GN_MAX_CHUNKS) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)partial & 7)) return EA_ERR_BAD_ARG;
  GnParams p;
  memset(&p, 0, sizeof(p));
  p.x1 = (const f16*)x; p.c1 = C;
  p.gamma = gamma; p.beta = beta;
  p.out = (f16*)out;
  p.B = B; p.HW = HW; p.C = C; p.groups = groups;
  p.eps = eps; p.silu = silu;
  int st = gn_plan(p);
  if (st != EA_OK) return st;
  p.partial = const_cast<float*>(partial);   // read only by the apply pass
  p.nchunk = nchunk;
  dim3 block(p.V * p.R, 1, 1);
  auto k2 = silu ? ea_gn_apply_kernel<false, true> : ea_gn_apply_kernel<false, false>;
  int nsub = (p.V * p.R) / groups;
  if (nsub < 1) nsub = 1;
  const int smem2 = (2 * groups * nsub + 2 * groups) * (int)sizeof(float);
  EA_LAUNCH(k2, dim3(p.anchunk, B, 1), block, smem2, stream, p);
  return ea_launch_status();
}

extern "C" int ea_layernorm_f16(const void* x, int in_f32, const float* gamma, const float* beta, void* out,
                                int M, int C, float eps, void* stream) {
  return ea_layernorm_rows_f16(x, in_f32, gamma, beta, out, M, C, eps, nullptr, stream);
}

extern "C" int ea_layernorm_rows_f16(const void* x, int in_f32, const float* gamma, const float* beta, void* out,
                                     int M, int C, float eps, const int* out_rows, void* stream) {
  if (!x || !gamma || !beta || !out) return EA_ERR_BAD_ARG;
  if (M <= 0 || C <= 0 || (C & 7) || C > 4096) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return EA_ERR_BAD_ARG;
  LnParams p;
  p.x = x; p.in_f32 = in_f32; p.gamma = gamma; p.beta = beta; p.out = (f16*)out;
  p.M = M; p.C = C; p.eps = eps; p.out_rows = out_rows;
  if (C <= 512) {
    auto kfn = ea_layernorm_kernel<1, 4>;
    EA_LAUNCH(kfn, dim3((M + 15) / 16), dim3(256), 0, stream, p);
  } else if (C <= 1536) {
    auto kfn = ea_layernorm_kernel<3, 2>;
    EA_LAUNCH(kfn, dim3((M + 7) / 8), dim3(256), 0, stream, p);
  } else {
    auto kfn = ea_layernorm_kernel<8, 1>;
    EA_LAUNCH(kfn, dim3((M + 3) / 4), dim3(256), 0, stream, p);
  }
  return ea_launch_status();
}

extern "C" int ea_softmax_rows_f32_f16(const float* x, void* out, int rows, int cols, float scale, void* stream) {
  if (!x || !out) return EA_ERR_BAD_ARG;
  if (rows <= 0 || cols <= 0) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_softmax_rows_kernel;
  EA_LAUNCH(kfn, dim3(rows), dim3(256), 64, stream, x, (f16*)out, rows, cols, scale);
  return ea_launch_status();
}
