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
  int nchunk, chunk_px;
  float eps;
  int silu;
};

__device__ __forceinline__ f16x8 gn_load8(const GnParams& p, long long pix, int c0) {
  if (c0 < p.c1) return ea_ld8(p.x1 + pix * p.c1 + c0);
  const long long off = pix * p.c2 + (c0 - p.c1);
  f16x8 v = ea_ld8(p.x2 + off);
  if (p.x2_add) v = v + ea_ld8(p.x2_add + off);
  return v;
}

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
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
  for (int px = p_begin + pr; px < p_end; px += p.R) {
    f16x8 x = gn_load8(p, (long long)b * p.HW + px, c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)x[j];
      s[j] += f;
      q[j] += f * f;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    chs[pr * p.C + c0 + j] = s[j];
    chq[pr * p.C + c0 + j] = q[j];
  }
  __syncthreads();
  for (int g = tid; g < p.groups; g += blockDim.x) {
    float gs = 0.0f, gq = 0.0f;
    for (int rr = 0; rr < p.R; ++rr)
      for (int c = g * p.cpg; c < (g + 1) * p.cpg; ++c) {
        gs += chs[rr * p.C + c];
        gq += chq[rr * p.C + c];
      }
    float* dst = p.partial + (((long long)b * p.nchunk + chunk) * p.groups + g) * 2;
    dst[0] = gs;
    dst[1] = gq;
  }
}

__global__ void ea_gn_apply_kernel(GnParams p) {
  const int tid = threadIdx.x;
  const int v = tid % p.V, pr = tid / p.V;
  const int c0 = v * 8;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int p_begin = chunk * p.chunk_px;
  int p_end = p_begin + p.chunk_px;
  if (p_end > p.HW) p_end = p.HW;
  // fold the partial sums of the groups this octet touches into scale/shift
  float a[8], sh[8];
  int g_prev = -1;
  float mean = 0.0f, rstd = 0.0f;
  const float inv_n = 1.0f / ((float)p.HW * (float)p.cpg);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    const int g = c / p.cpg;
    if (g != g_prev) {
      float gs = 0.0f, gq = 0.0f;
      for (int ch = 0; ch < p.nchunk; ++ch) {
        const float* src = p.partial + (((long long)b * p.nchunk + ch) * p.groups + g) * 2;
        gs += src[0];
        gq += src[1];
      }
      mean = gs * inv_n;
      float var = gq * inv_n - mean * mean;
      var = var > 0.0f ? var : 0.0f;
      rstd = 1.0f / sqrtf(var + p.eps);
      g_prev = g;
    }
    a[j] = rstd * p.gamma[c];
    sh[j] = p.beta[c] - mean * a[j];
  }
  for (int px = p_begin + pr; px < p_end; px += p.R) {
    const long long pix = (long long)b * p.HW + px;
    f16x8 x = gn_load8(p, pix, c0);
    f16x8 y;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)x[j] * a[j] + sh[j];
      if (p.silu) f = ea_silu(f);
      y[j] = (f16)f;
    }
    ea_st8(p.out + pix * p.C + c0, y);
  }
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
  int target = 1024 / (p.B > 0 ? p.B : 1);
  if (target < 1) target = 1;
  int nchunk = p.HW / (r * 2);
  if (nchunk > target) nchunk = target;
  if (nchunk > 64) nchunk = 64;
  if (nchunk < 1) nchunk = 1;
  p.chunk_px = (p.HW + nchunk - 1) / nchunk;
  p.nchunk = (p.HW + p.chunk_px - 1) / p.chunk_px;
  return EA_OK;
}

// ------------------------------------------------------------------ LayerNorm
struct LnParams {
  const void* x; int in_f32;
  const float* gamma; const float* beta;
  f16* out;
  int M, C;
  float eps;
};

// One wave per row; two-pass in registers (mean, then centred variance), as torch does.
__global__ __launch_bounds__(256) void ea_layernorm_kernel(LnParams p) {
  constexpr int MAXV = 8;  // up to 8 octets per lane -> C <= 4096
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const bool active = row < p.M;
  const int r = active ? row : 0;
  const int nv = p.C / 8;
  float vals[MAXV][8];
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      if (p.in_f32) {
        const float* src = (const float*)p.x + (long long)r * p.C + v * 8;
        f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { vals[i][j] = lo[j]; vals[i][4 + j] = hi[j]; }
      } else {
        f16x8 h = ea_ld8((const f16*)p.x + (long long)r * p.C + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[i][j] = (float)h[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += vals[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) vals[i][j] = 0.0f;
    }
  }
  sum = ea_wave_sum(sum);
  const float mean = sum / (float)p.C;
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = vals[i][j] - mean;
        sq += d * d;
      }
    }
  }
  sq = ea_wave_sum(sq);
  const float rstd = 1.0f / sqrtf(sq / (float)p.C + p.eps);
  if (!active) return;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      f16x8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        y[j] = (f16)((vals[i][j] - mean) * rstd * p.gamma[c] + p.beta[c]);
      }
      ea_st8(p.out + (long long)row * p.C + v * 8, y);
    }
  }
}

// ---------------------------------------------------------------- row softmax
__global__ __launch_bounds__(256) void ea_softmax_rows_kernel(const float* x, f16* out, int rows, int cols, float scale) {
  EA_SMEM(smem);
  float* red = reinterpret_cast<float*>(smem);  // [8]
  const int row = blockIdx.x;
  if (row >= rows) return;
  const float* src = x + (long long)row * cols;
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
  for (int c = tid; c < cols; c += 256) out[(long long)row * cols + c] = (f16)(ea_expf(src[c] * scale - m) * inv);
}

}  // namespace

extern "C" size_t ea_groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
  if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
  return (size_t)B * 64 * groups * 2 * sizeof(float);
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
  dim3 grid(p.nchunk, B, 1), block(p.V * p.R, 1, 1);
  const int smem = 2 * p.R * p.C * (int)sizeof(float);
  auto k1 = ea_gn_stats_kernel;
  ea_allow_big_lds(k1, smem);
  EA_LAUNCH(k1, grid, block, smem, stream, p);
  st = ea_launch_status();
  if (st != EA_OK) return st;
  auto k2 = ea_gn_apply_kernel;
  EA_LAUNCH(k2, grid, block, 0, stream, p);
  return ea_launch_status();
}

extern "C" int ea_layernorm_f16(const void* x, int in_f32, const float* gamma, const float* beta, void* out,
                                int M, int C, float eps, void* stream) {
  if (!x || !gamma || !beta || !out) return EA_ERR_BAD_ARG;
  if (M <= 0 || C <= 0 || (C & 7) || C > 4096) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return EA_ERR_BAD_ARG;
  LnParams p;
  p.x = x; p.in_f32 = in_f32; p.gamma = gamma; p.beta = beta; p.out = (f16*)out;
  p.M = M; p.C = C; p.eps = eps;
  auto kfn = ea_layernorm_kernel;
  EA_LAUNCH(kfn, dim3((M + 3) / 4), dim3(256), 0, stream, p);
  return ea_launch_status();
}

extern "C" int ea_softmax_rows_f32_f16(const float* x, void* out, int rows, int cols, float scale, void* stream) {
  if (!x || !out) return EA_ERR_BAD_ARG;
  if (rows <= 0 || cols <= 0) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_softmax_rows_kernel;
  EA_LAUNCH(kfn, dim3(rows), dim3(256), 64, stream, x, (f16*)out, rows, cols, scale);
  return ea_launch_status();
}
