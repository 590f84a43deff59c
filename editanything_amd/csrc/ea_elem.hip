// ea_elem.hip -- launch-count-bound elementwise pieces of the sampler loop and
// the layout plumbing at the NCHW(fp32, reference API) <-> NHWC(fp16, HBM) boundary.
#include "ea_platform.h"
#include "../../include/editanything_hip.h"
#include <string.h>

namespace {

// cldm/ddim_hacked.py:187-231 (p_sample_ddim) + the inpaint latent blend of
// utils/stable_diffusion_controlnet_inpaint.py:1647-1664, one pass over the latents.
__global__ __launch_bounds__(256) void ea_cfg_ddim_kernel(const float* x, const float* eps_c, const float* eps_u,
                                                          const float* noise, const float* coef, const float* mask,
                                                          const float* x_orig, const float* noise_orig,
                                                          float* x_prev, float* pred_x0, long long n) {
  const float a_t = coef[0], a_prev = coef[1], sigma = coef[2], g = coef[3];
  const bool vpred = coef[4] != 0.0f;
  const float sqrt_at = sqrtf(a_t), sqrt_1mat = sqrtf(1.0f - a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir = sqrtf(fmaxf(1.0f - a_prev - sigma * sigma, 0.0f));
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float xi = x[i];
    float mo = eps_c[i];
    if (eps_u) {
      const float u = eps_u[i];
      mo = u + g * (mo - u);
    }
    float e_t, x0;
    if (vpred) {
      e_t = sqrt_at * mo + sqrt_1mat * xi;  // ldm/models/diffusion/ddpm.py:296-302
      x0 = sqrt_at * xi - sqrt_1mat * mo;   // ddpm.py:290-294
    } else {
      e_t = mo;
      x0 = (xi - sqrt_1mat * e_t) / sqrt_at;
    }
    float xp = sqrt_aprev * x0 + dir * e_t;
    if (noise) xp += sigma * noise[i];
    if (mask) {
      const float keep = sqrt_aprev * x_orig[i] + sqrtf(1.0f - a_prev) * (noise_orig ? noise_orig[i] : 0.0f);
      const float mk = mask[i];
      xp = mk * xp + (1.0f - mk) * keep;
    }
    x_prev[i] = xp;
    if (pred_x0) pred_x0[i] = x0;
  }
}

__global__ __launch_bounds__(256) void ea_nchw2nhwc_kernel(const float* x, f16* out, int B, int C, int H, int W,
                                                          int Cpad, float mul, float add) {
  const long long total = (long long)B * H * W * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long long pix = i / Cpad;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int b = (int)(pix / ((long long)W * H));
    float v = 0.0f;
    if (c < C) v = x[(((long long)b * C + c) * H + h) * W + w] * mul + add;
    out[i] = (f16)v;
  }
}

__global__ __launch_bounds__(256) void ea_nhwc2nchw_kernel(const f16* x, float* out, int B, int C, int H, int W,
                                                          int Cstride, float mul, float add) {
  const long long total = (long long)B * C * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int c = (int)((i / ((long long)W * H)) % C);
    const int b = (int)(i / ((long long)W * H * C));
    out[i] = (float)x[(((long long)b * H + h) * W + w) * Cstride + c] * mul + add;
  }
}

__global__ __launch_bounds__(256) void ea_silu_kernel(const float* x, float* out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = ea_silu(x[i]);
}

__global__ __launch_bounds__(256) void ea_add_kernel(const f16* a, const f16* b, f16* out, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x)
    ea_st8(out + i * 8, ea_ld8(a + i * 8) + ea_ld8(b + i * 8));
}

// out = sum_k coef[k] * src[k]  (k < 5, NULL sources skipped), optionally blended through a mask with a second
// combination: out = mask * main + (1 - mask) * (coef[5] * alt0 + coef[6] * alt1).  The multistep sampler update
// (UniPC predictor / corrector: a linear combination of the sample and the stored x0 predictions with per-step scalar
// coefficients) and the inpaint re-noise blend; coefficients come from a device buffer so a captured step replays.
__global__ __launch_bounds__(256) void ea_lincomb_kernel(const float* s0, const float* s1, const float* s2, const float* s3,
                                                        const float* s4, const float* coef, const float* mask,
                                                        const float* alt0, const float* alt1, float* out, long long n) {
  const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4], e0 = coef[5], e1 = coef[6];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (s0) v += c0 * s0[i];
    if (s1) v += c1 * s1[i];
    if (s2) v += c2 * s2[i];
    if (s3) v += c3 * s3[i];
    if (s4) v += c4 * s4[i];
    if (mask) {
      float a = 0.0f;
      if (alt0) a += e0 * alt0[i];
      if (alt1) a += e1 * alt1[i];
      const float mk = mask[i];
      v = mk * v + (1.0f - mk) * a;
    }
    out[i] = v;
  }
}

// One step's inputs out of per-call tables, by a DEVICE step index: segment s copies row `*index` of its table (row_bytes bytes)
// dst_rows times into its destination, then the index advances.  ONE workgroup: the copies are a few hundred KB out of L2, and the
// increment needs no ordering against other workgroups' reads of the index.  (Replaces, in the captured denoising step, four
// index_select + four to eight copy nodes + an add -- a graph memcpy node costs ~10 us on this stack, profiles/r06_bench_kernel_stats.csv.)
#define EA_GATHER_MAX 8
struct EaGatherParams {
  const char* table[EA_GATHER_MAX];
  char* dst[EA_GATHER_MAX];
  long long row_bytes[EA_GATHER_MAX];
  int dst_rows[EA_GATHER_MAX];
  int nseg, increment;
  long long* index;
};
__global__ __launch_bounds__(1024) void ea_gather_rows_kernel(EaGatherParams p) {
  const long long idx = *p.index;
  for (int s = 0; s < p.nseg; ++s) {
    const long long rb = p.row_bytes[s];
    const char* src = p.table[s] + idx * rb;
    char* dst = p.dst[s];
    const bool v16 = ((rb | (long long)(uintptr_t)src | (long long)(uintptr_t)dst) & 15) == 0;
    for (int r = 0; r < p.dst_rows[s]; ++r) {
      char* d = dst + (long long)r * rb;
      if (v16) {
        for (long long i = threadIdx.x; i < (rb >> 4); i += blockDim.x) reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(src)[i];
      } else {
        for (long long i = threadIdx.x; i < (rb >> 2); i += blockDim.x) reinterpret_cast<unsigned*>(d)[i] = reinterpret_cast<const unsigned*>(src)[i];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && p.increment) *p.index = idx + p.increment;
}

static unsigned grid_for(long long n) {
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;  // 256 CUs x 8 workgroups, grid-stride the rest
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

}  // namespace

extern "C" int ea_version(void) { return 100; }

extern "C" int ea_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
#ifdef EA_EMU
  if (cu_count) *cu_count = 0;
  if (lds_bytes) *lds_bytes = 160 * 1024;
  if (arch && arch_len > 0) { strncpy(arch, "emu", arch_len - 1); arch[arch_len - 1] = 0; }
  return EA_OK;
#else
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return EA_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return EA_ERR_LAUNCH;
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  return EA_OK;
#endif
}

extern "C" int ea_cfg_ddim_step(const float* x, const float* eps_c, const float* eps_u, const float* noise,
                                const float* coef, const float* mask, const float* x_orig,
                                const float* noise_orig, float* x_prev, float* pred_x0, long long n,
                                void* stream) {
  if (!x || !eps_c || !coef || !x_prev) return EA_ERR_BAD_ARG;
  if (mask && !x_orig) return EA_ERR_BAD_ARG;
  if (n <= 0) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_cfg_ddim_kernel;
  EA_LAUNCH(kfn, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_c, eps_u, noise, coef, mask, x_orig, noise_orig,
            x_prev, pred_x0, n);
  return ea_launch_status();
}

extern "C" int ea_gather_rows(const void* const* tables, void* const* dsts, const long long* row_bytes, const int* dst_rows, int nseg,
                              long long* index, int increment, void* stream) {
  if (!tables || !dsts || !row_bytes || !dst_rows || !index) return EA_ERR_BAD_ARG;
  if (nseg <= 0 || nseg > EA_GATHER_MAX) return EA_ERR_BAD_SHAPE;
  EaGatherParams p{};
  for (int s = 0; s < nseg; ++s) {
    if (!tables[s] || !dsts[s]) return EA_ERR_BAD_ARG;
    if (row_bytes[s] <= 0 || (row_bytes[s] & 3) || dst_rows[s] <= 0) return EA_ERR_BAD_SHAPE;
    if ((((uintptr_t)tables[s]) | ((uintptr_t)dsts[s])) & 3) return EA_ERR_BAD_ARG;
    p.table[s] = (const char*)tables[s];
    p.dst[s] = (char*)dsts[s];
    p.row_bytes[s] = row_bytes[s];
    p.dst_rows[s] = dst_rows[s];
  }
  p.nseg = nseg;
  p.increment = increment;
  p.index = index;
  auto kfn = ea_gather_rows_kernel;
  EA_LAUNCH(kfn, dim3(1), dim3(1024), 0, stream, p);
  return ea_launch_status();
}

extern "C" int ea_lincomb_f32(const float* s0, const float* s1, const float* s2, const float* s3, const float* s4,
                              const float* coef, const float* mask, const float* alt0, const float* alt1, float* out,
                              long long n, void* stream) {
  if (!coef || !out) return EA_ERR_BAD_ARG;
  if (n <= 0) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_lincomb_kernel;
  EA_LAUNCH(kfn, dim3(grid_for(n)), dim3(256), 0, stream, s0, s1, s2, s3, s4, coef, mask, alt0, alt1, out, n);
  return ea_launch_status();
}

extern "C" int ea_nchw_f32_to_nhwc_f16(const float* x, void* out, int B, int C, int H, int W, int Cpad, float mul,
                                       float add, void* stream) {
  if (!x || !out) return EA_ERR_BAD_ARG;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_nchw2nhwc_kernel;
  EA_LAUNCH(kfn, dim3(grid_for((long long)B * H * W * Cpad)), dim3(256), 0, stream, x, (f16*)out, B, C, H, W, Cpad,
            mul, add);
  return ea_launch_status();
}

extern "C" int ea_nhwc_f16_to_nchw_f32(const void* x, float* out, int B, int C, int H, int W, int Cstride,
                                       float mul, float add, void* stream) {
  if (!x || !out) return EA_ERR_BAD_ARG;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cstride < C) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_nhwc2nchw_kernel;
  EA_LAUNCH(kfn, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, stream, (const f16*)x, out, B, C, H, W,
            Cstride, mul, add);
  return ea_launch_status();
}

extern "C" int ea_silu_f32(const float* x, float* out, long long n, void* stream) {
  if (!x || !out) return EA_ERR_BAD_ARG;
  if (n <= 0) return EA_ERR_BAD_SHAPE;
  auto kfn = ea_silu_kernel;
  EA_LAUNCH(kfn, dim3(grid_for(n)), dim3(256), 0, stream, x, out, n);
  return ea_launch_status();
}

extern "C" int ea_add_f16(const void* a, const void* b, void* out, long long n, void* stream) {
  if (!a || !b || !out) return EA_ERR_BAD_ARG;
  if (n <= 0 || (n & 7)) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)out & 15)) return EA_ERR_BAD_ARG;
  auto kfn = ea_add_kernel;
  EA_LAUNCH(kfn, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const f16*)a, (const f16*)b, (f16*)out, n / 8);
  return ea_launch_status();
}

// x[t][:] += src[rows[t]][:]  (fp32 residual stream += gathered fp16 rows): SAM window_unpartition + residual add
__global__ __launch_bounds__(256) void ea_gather_add_rows_kernel(float* x, const f16* src, const int* rows, int T, int C8) {
  const long long total = (long long)T * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / C8), c = (int)(i - (long long)t * C8) * 8;
    const int r = rows[t];
    if (r < 0) continue;
    const f16x8 s8 = ea_ld8(src + (long long)r * C8 * 8 + c);
    float* xp = x + (long long)t * C8 * 8 + c;
    f32x4 lo = *reinterpret_cast<f32x4*>(xp), hi = *reinterpret_cast<f32x4*>(xp + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { lo[j] += (float)s8[j]; hi[j] += (float)s8[4 + j]; }
    *reinterpret_cast<f32x4*>(xp) = lo;
    *reinterpret_cast<f32x4*>(xp + 4) = hi;
  }
}

extern "C" int ea_gather_add_rows_f32(float* x, const void* src, const int* rows, int T, int C, void* stream) {
  if (!x || !src || !rows) return EA_ERR_BAD_ARG;
  if (T <= 0 || C <= 0 || (C & 7)) return EA_ERR_BAD_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)src & 15)) return EA_ERR_BAD_ARG;
  auto kfn = ea_gather_add_rows_kernel;
  EA_LAUNCH(kfn, dim3(grid_for((long long)T * (C / 8))), dim3(256), 0, stream, x, (const f16*)src, rows, T, C / 8);
  return ea_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// SAM mask post-processing in ONE pass (segment_anything Sam.postprocess_masks + utils/amg.py calculate_stability_score
// + batched_mask_to_box, as SamAutomaticMaskGenerator._process_batch chains them):
//   logits = bilinear(bilinear(low -> S x S)[:in_h, :in_w] -> H x W)        (both align_corners=False, torch's formula)
//   mask = logits > thr;  inter = #(logits > thr + off);  union = #(logits > thr - off);  box = extent of mask
// The reference materialises the S x S (1024^2) fp32 upsampling of every candidate mask (805 MB per 64-point batch) and
// re-reads it four times; here each output pixel evaluates the two bilinear stages analytically from the (cache
// resident) low-res logits and only the 1-byte mask is written.  Integer atomics -> deterministic.
struct MaskPostParams {
  const float* low;
  const int* index;      // optional: mask slot -> index into `low`
  unsigned char* mask;
  int* stats;   // [Nm][6]: inter, union, xmin, ymin, xmax, ymax (caller-initialised to 0, 0, W, H, -1, -1)
  int lh, lw, S, in_h, in_w, H, W;
  float thr, off;
  int* idmap;   // ea_sam_id_map_kernel: int32 [H][W]
  int n, id_base, band;   // masks of the launch; id of slot 0 minus one; output rows per workgroup (tabled kernels)
};

__device__ __forceinline__ void ea_bilin_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;     // area_pixel_compute_source_index, align_corners = false
  if (src < 0.0f) src = 0.0f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__device__ __forceinline__ void ea_atomic_add_i(int* p, int v) {
#ifdef EA_EMU
  *p += v;
#else
  atomicAdd(p, v);
#endif
}
__device__ __forceinline__ void ea_atomic_min_i(int* p, int v) {
#ifdef EA_EMU
  if (v < *p) *p = v;
#else
  atomicMin(p, v);
#endif
}
__device__ __forceinline__ void ea_atomic_max_i(int* p, int v) {
#ifdef EA_EMU
  if (v > *p) *p = v;
#else
  atomicMax(p, v);
#endif
}

// The value of ONE pixel of one bilinear stage from its four taps (torch's upsample_bilinear2d form): shared by every
// kernel below, so the per-pixel kernel and the tabled one evaluate the same expression tree.
__device__ __forceinline__ float ea_bilerp(float my, float mx, float v00, float v01, float v10, float v11) {
  return (1.0f - my) * ((1.0f - mx) * v00 + mx * v01) + my * ((1.0f - mx) * v10 + mx * v11);
}

__device__ __forceinline__ int ea_shfl_xor_i(int v, int mask) {
#ifdef EA_EMU
  return ea_emu_shfl_xor<int>(v, mask);
#else
  return __shfl_xor(v, mask, 64);
#endif
}

// Block reduction of the six per-thread statistics (two sums, two minima, two maxima) + the integer atomics into stats[slot]:
// wave shuffles, then one LDS slot per wave (`red`: >= 4 * 6 ints).
__device__ __forceinline__ void ea_mask_stats_commit(int* red, int* stats_slot, int inter, int uni, int xmin, int ymin, int xmax, int ymax) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    inter += ea_shfl_xor_i(inter, m);
    uni += ea_shfl_xor_i(uni, m);
    const int a = ea_shfl_xor_i(xmin, m), b = ea_shfl_xor_i(ymin, m), c = ea_shfl_xor_i(xmax, m), d = ea_shfl_xor_i(ymax, m);
    xmin = a < xmin ? a : xmin;
    ymin = b < ymin ? b : ymin;
    xmax = c > xmax ? c : xmax;
    ymax = d > ymax ? d : ymax;
  }
  if ((tid & 63) == 0) {
    int* r = red + (tid >> 6) * 6;
    r[0] = inter; r[1] = uni; r[2] = xmin; r[3] = ymin; r[4] = xmax; r[5] = ymax;
  }
  __syncthreads();
  if (tid < 6) {
    int acc = red[tid];
    for (int w = 1; w < 4; ++w) {
      const int v = red[w * 6 + tid];
      if (tid < 2) acc += v;
      else if (tid < 4) acc = v < acc ? v : acc;
      else acc = v > acc ? v : acc;
    }
    int* dst = stats_slot + tid;
    if (tid < 2) ea_atomic_add_i(dst, acc);
    else if (tid < 4) ea_atomic_min_i(dst, acc);
    else ea_atomic_max_i(dst, acc);
  }
}

// One thread = one output pixel at a time (grid-stride over the mask): both bilinear resizes evaluated analytically from the
// low-resolution logits, 16 taps that hit L1.  (Round 3 tried staging the four low-resolution rows of an output row in LDS --
// two barriers per row for 512 pixels: 3x SLOWER than the scattered global reads it replaced; reverted.)  `index`
// (optional) selects the masks to process out of the low-resolution tensor, `mask` may be NULL (statistics only): the
// generator filters on the statistics and never copies logits or writes masks it will drop.  Since round 6 this is the
// fallback for rows wider than the tabled kernel's LDS tables hold (W > 2048) and the A/B reference of that kernel.
__global__ __launch_bounds__(256) void ea_mask_post_kernel(MaskPostParams p) {
  EA_SMEM(smem);
  int* red = reinterpret_cast<int*>(smem);   // [4][6]
  const int tid = threadIdx.x;
  const int slot = blockIdx.y;
  const int m = p.index ? p.index[slot] : slot;
  const float* low = p.low + (long long)m * p.lh * p.lw;
  unsigned char* out = p.mask ? p.mask + (long long)slot * p.H * p.W : nullptr;
  const float s1y = (float)p.lh / (float)p.S, s1x = (float)p.lw / (float)p.S;
  const float s2y = (float)p.in_h / (float)p.H, s2x = (float)p.in_w / (float)p.W;
  int inter = 0, uni = 0, xmin = p.W, ymin = p.H, xmax = -1, ymax = -1;
  const int npix = p.H * p.W;
  for (int i = blockIdx.x * 256 + tid; i < npix; i += gridDim.x * 256) {
    const int y = i / p.W, x = i - y * p.W;
    int Y[2], X[2];
    float ly, lx;
    ea_bilin_index(s2y, y, p.in_h, Y[0], Y[1], ly);
    ea_bilin_index(s2x, x, p.in_w, X[0], X[1], lx);
    float up[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int y0, y1;
      float my;
      ea_bilin_index(s1y, Y[a], p.lh, y0, y1, my);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int x0, x1;
        float mx;
        ea_bilin_index(s1x, X[b], p.lw, x0, x1, mx);
        up[a][b] = ea_bilerp(my, mx, low[y0 * p.lw + x0], low[y0 * p.lw + x1], low[y1 * p.lw + x0], low[y1 * p.lw + x1]);
      }
    }
    const float v = ea_bilerp(ly, lx, up[0][0], up[0][1], up[1][0], up[1][1]);
    const bool on = v > p.thr;
    if (out) out[i] = on ? 1 : 0;
    inter += (v > p.thr + p.off) ? 1 : 0;
    uni += (v > p.thr - p.off) ? 1 : 0;
    if (on) {
      xmin = x < xmin ? x : xmin;
      xmax = x > xmax ? x : xmax;
      ymin = y < ymin ? y : ymin;
      ymax = y > ymax ? y : ymax;
    }
  }
  ea_mask_stats_commit(red, p.stats + slot * 6, inter, uni, xmin, ymin, xmax, ymax);
}

// ---- the tabled form (round 6).  The per-pixel kernel spends most of its issue slots recomputing, for every pixel, index
// arithmetic that depends on the pixel's COLUMN alone (two stage-2 taps, their stage-1 taps and weights) or on its ROW alone,
// and reads 16 taps of which -- whenever both stage-2 taps of a pixel fall into one cell of the low-resolution grid, which is
// every pixel of an up-by-4 / down-by-2 chain like 256 -> 1024 -> 512 -- only 4 are distinct.  Here a workgroup owns a band of
// output rows of one mask, builds the column table (all W columns) and the band's row table in LDS ONCE with the very
// ea_bilin_index calls of the per-pixel kernel, notes whether every column / every row of the band has coinciding taps, and
// runs the matching one of four inner loops (4, 8, 8 or 16 loads per pixel); a pixel's value is the same ea_bilerp tree over the
// same taps, so masks and statistics equal the per-pixel kernel's bit for bit (tests/test_kernels.py).
struct MaskCol { int xa, xb; float mxa, mxb, lx; };   // xa = x0 | x1 << 16 of the first stage-2 tap, xb of the second
struct MaskRow { int y0a, y1a, y0b, y1b; float mya, myb, ly; };   // row offsets (elements) into the low-resolution mask
#define EA_MASK_MAXW 2048
#define EA_MASK_MAXR 32

__device__ __forceinline__ void ea_mask_tables(const MaskPostParams& p, int row0, int rows, int col0, int cols, int* flags, int* cxa, int* cxb,
                                               float* cmxa, float* cmxb, float* clx, MaskRow* rtab) {
  const int tid = threadIdx.x;
  const float s1y = (float)p.lh / (float)p.S, s1x = (float)p.lw / (float)p.S;
  const float s2y = (float)p.in_h / (float)p.H, s2x = (float)p.in_w / (float)p.W;
  if (tid < 2) flags[tid] = 1;
  __syncthreads();
  for (int x = tid; x < cols; x += 256) {
    int X0, X1, a0, a1, b0, b1;
    float lx, ma, mb;
    ea_bilin_index(s2x, col0 + x, p.in_w, X0, X1, lx);
    ea_bilin_index(s1x, X0, p.lw, a0, a1, ma);
    ea_bilin_index(s1x, X1, p.lw, b0, b1, mb);
    cxa[x] = a0 | (a1 << 16);
    cxb[x] = b0 | (b1 << 16);
    cmxa[x] = ma; cmxb[x] = mb; clx[x] = lx;
    if (a0 != b0 || a1 != b1) flags[0] = 0;
  }
  if (tid < rows) {
    int Y0, Y1, a0, a1, b0, b1;
    MaskRow r;
    ea_bilin_index(s2y, row0 + tid, p.in_h, Y0, Y1, r.ly);
    ea_bilin_index(s1y, Y0, p.lh, a0, a1, r.mya);
    ea_bilin_index(s1y, Y1, p.lh, b0, b1, r.myb);
    r.y0a = a0 * p.lw; r.y1a = a1 * p.lw; r.y0b = b0 * p.lw; r.y1b = b1 * p.lw;
    rtab[tid] = r;
    if (a0 != b0 || a1 != b1) flags[1] = 0;
  }
  __syncthreads();
}

// value of output pixel (row entry r, column x) of the low-resolution mask `low`
template <bool SX, bool SY>
__device__ __forceinline__ float ea_mask_value(const float* low, const MaskRow& r, int xa, int xb, float mxa, float mxb, float lx) {
  const int x0a = xa & 0xffff, x1a = xa >> 16;
  const float* ra0 = low + r.y0a;
  const float* ra1 = low + r.y1a;
  const float a00 = ra0[x0a], a01 = ra0[x1a], a10 = ra1[x0a], a11 = ra1[x1a];      // rows of tap a, columns of tap a
  float b00 = a00, b01 = a01, b10 = a10, b11 = a11;                                 // rows of tap a, columns of tap b
  float c00 = a00, c01 = a01, c10 = a10, c11 = a11;                                 // rows of tap b, columns of tap a
  float d00 = a00, d01 = a01, d10 = a10, d11 = a11;                                 // rows of tap b, columns of tap b
  if (!SX) {
    const int x0b = xb & 0xffff, x1b = xb >> 16;
    b00 = ra0[x0b]; b01 = ra0[x1b]; b10 = ra1[x0b]; b11 = ra1[x1b];
    d00 = b00; d01 = b01; d10 = b10; d11 = b11;
  }
  if (!SY) {
    const float* rb0 = low + r.y0b;
    const float* rb1 = low + r.y1b;
    c00 = rb0[x0a]; c01 = rb0[x1a]; c10 = rb1[x0a]; c11 = rb1[x1a];
    if (!SX) {
      const int x0b = xb & 0xffff, x1b = xb >> 16;
      d00 = rb0[x0b]; d01 = rb0[x1b]; d10 = rb1[x0b]; d11 = rb1[x1b];
    } else {
      d00 = c00; d01 = c01; d10 = c10; d11 = c11;
    }
  }
  const float u00 = ea_bilerp(r.mya, mxa, a00, a01, a10, a11);
  const float u01 = ea_bilerp(r.mya, mxb, b00, b01, b10, b11);
  const float u10 = ea_bilerp(r.myb, mxa, c00, c01, c10, c11);
  const float u11 = ea_bilerp(r.myb, mxb, d00, d01, d10, d11);
  return ea_bilerp(r.ly, lx, u00, u01, u10, u11);
}

template <bool SX, bool SY>
__device__ __forceinline__ void ea_mask_band(const MaskPostParams& p, const float* low, unsigned char* out, int row0, int rows, const int* cxa,
                                             const int* cxb, const float* cmxa, const float* cmxb, const float* clx, const MaskRow* rtab,
                                             int& inter, int& uni, int& xmin, int& ymin, int& xmax, int& ymax) {
  const float thr = p.thr, hi = p.thr + p.off, lo = p.thr - p.off;
  for (int x = threadIdx.x; x < p.W; x += 256) {
    const int xa = cxa[x], xb = cxb[x];
    const float mxa = cmxa[x], mxb = cmxb[x], lx = clx[x];
    bool any = false;
    for (int j = 0; j < rows; ++j) {
      const MaskRow r = rtab[j];
      const float v = ea_mask_value<SX, SY>(low, r, xa, xb, mxa, mxb, lx);
      const bool on = v > thr;
      const int y = row0 + j;
      if (out) out[(long long)y * p.W + x] = on ? 1 : 0;
      inter += (v > hi) ? 1 : 0;
      uni += (v > lo) ? 1 : 0;
      if (on) {
        any = true;
        ymin = y < ymin ? y : ymin;
        ymax = y > ymax ? y : ymax;
      }
    }
    if (any) {
      xmin = x < xmin ? x : xmin;
      xmax = x > xmax ? x : xmax;
    }
  }
}

__global__ __launch_bounds__(256) void ea_mask_post_tab_kernel(MaskPostParams p) {
  EA_SMEM(smem);
  int* red = reinterpret_cast<int*>(smem);                     // [4][6] + 2 flags
  int* flags = red + 24;
  MaskRow* rtab = reinterpret_cast<MaskRow*>(smem + 128);      // [EA_MASK_MAXR]
  int* cxa = reinterpret_cast<int*>(smem + 128 + EA_MASK_MAXR * (int)sizeof(MaskRow));
  int* cxb = cxa + p.W;
  float* cmxa = reinterpret_cast<float*>(cxb + p.W);
  float* cmxb = cmxa + p.W;
  float* clx = cmxb + p.W;
  const int slot = blockIdx.y;
  const int m = p.index ? p.index[slot] : slot;
  const float* low = p.low + (long long)m * p.lh * p.lw;
  unsigned char* out = p.mask ? p.mask + (long long)slot * p.H * p.W : nullptr;
  const int row0 = blockIdx.x * p.band;
  const int rows = p.H - row0 < p.band ? p.H - row0 : p.band;
  ea_mask_tables(p, row0, rows, 0, p.W, flags, cxa, cxb, cmxa, cmxb, clx, rtab);
  int inter = 0, uni = 0, xmin = p.W, ymin = p.H, xmax = -1, ymax = -1;
  const bool sx = flags[0] != 0, sy = flags[1] != 0;
  if (sx && sy) ea_mask_band<true, true>(p, low, out, row0, rows, cxa, cxb, cmxa, cmxb, clx, rtab, inter, uni, xmin, ymin, xmax, ymax);
  else if (sx) ea_mask_band<true, false>(p, low, out, row0, rows, cxa, cxb, cmxa, cmxb, clx, rtab, inter, uni, xmin, ymin, xmax, ymax);
  else if (sy) ea_mask_band<false, true>(p, low, out, row0, rows, cxa, cxb, cmxa, cmxb, clx, rtab, inter, uni, xmin, ymin, xmax, ymax);
  else ea_mask_band<false, false>(p, low, out, row0, rows, cxa, cxb, cmxa, cmxb, clx, rtab, inter, uni, xmin, ymin, xmax, ymax);
  __syncthreads();
  ea_mask_stats_commit(red, p.stats + slot * 6, inter, uni, xmin, ymin, xmax, ymax);
}

// ---- show_anns' id map straight from the records' low-resolution logits (sam2image.py:92-115 over the list
// SamAutomaticMaskGenerator.generate returns: record i paints i + 1 over its mask, later records over earlier ones, i.e. a pixel
// carries the LARGEST record number whose mask covers it).  One thread owns a pixel and walks the records from the last to
// the first until one covers it -- the tabled evaluation above, so the decision per (record, pixel) is the bit the mask kernel
// writes -- instead of writing n full-resolution masks and reducing them: no mask byte exists, nothing is atomic.
// idmap[pixel] = max(idmap[pixel], id_base + 1 + the largest covering slot).
template <bool SX, bool SY>
__device__ __forceinline__ int ea_id_walk(const MaskPostParams& p, const int* sidx, const MaskRow& r, int xa, int xb, float mxa, float mxb, float lx) {
  const long long lsz = (long long)p.lh * p.lw;
  int s = p.n - 1;
  // four records per trip: their 4 x (4 ... 16) taps are in flight together -- the walk is a chain of dependent loads otherwise
  for (; s >= 3; s -= 4) {
    const float v0 = ea_mask_value<SX, SY>(p.low + sidx[s] * lsz, r, xa, xb, mxa, mxb, lx);
    const float v1 = ea_mask_value<SX, SY>(p.low + sidx[s - 1] * lsz, r, xa, xb, mxa, mxb, lx);
    const float v2 = ea_mask_value<SX, SY>(p.low + sidx[s - 2] * lsz, r, xa, xb, mxa, mxb, lx);
    const float v3 = ea_mask_value<SX, SY>(p.low + sidx[s - 3] * lsz, r, xa, xb, mxa, mxb, lx);
    if (v0 > p.thr) return s + 1;
    if (v1 > p.thr) return s;
    if (v2 > p.thr) return s - 1;
    if (v3 > p.thr) return s - 2;
  }
  for (; s >= 0; --s)
    if (ea_mask_value<SX, SY>(p.low + sidx[s] * lsz, r, xa, xb, mxa, mxb, lx) > p.thr) return s + 1;
  return 0;
}

// grid (column blocks of 256, rows): one pixel per thread
__global__ __launch_bounds__(256) void ea_sam_id_map_kernel(MaskPostParams p) {
  EA_SMEM(smem);
  int* flags = reinterpret_cast<int*>(smem) + 24;
  MaskRow* rtab = reinterpret_cast<MaskRow*>(smem + 128);
  int* cxa = reinterpret_cast<int*>(smem + 128 + EA_MASK_MAXR * (int)sizeof(MaskRow));
  int* cxb = cxa + 256;
  float* cmxa = reinterpret_cast<float*>(cxb + 256);
  float* cmxb = cmxa + 256;
  float* clx = cmxb + 256;
  int* sidx = reinterpret_cast<int*>(clx + 256);            // [n]: the records' places in `low`
  const int tid = threadIdx.x;
  const int col0 = blockIdx.x * 256, y = blockIdx.y;
  const int cols = p.W - col0 < 256 ? p.W - col0 : 256;
  for (int s = tid; s < p.n; s += 256) sidx[s] = p.index ? p.index[s] : s;
  ea_mask_tables(p, y, 1, col0, cols, flags, cxa, cxb, cmxa, cmxb, clx, rtab);
  if (tid >= cols) return;
  const bool sx = flags[0] != 0, sy = flags[1] != 0;
  const MaskRow r = rtab[0];
  const int xa = cxa[tid], xb = cxb[tid];
  const float mxa = cmxa[tid], mxb = cmxb[tid], lx = clx[tid];
  int id;
  if (sx && sy) id = ea_id_walk<true, true>(p, sidx, r, xa, xb, mxa, mxb, lx);
  else if (sx) id = ea_id_walk<true, false>(p, sidx, r, xa, xb, mxa, mxb, lx);
  else if (sy) id = ea_id_walk<false, true>(p, sidx, r, xa, xb, mxa, mxb, lx);
  else id = ea_id_walk<false, false>(p, sidx, r, xa, xb, mxa, mxb, lx);
  if (id) {
    int* dst = p.idmap + (long long)y * p.W + col0 + tid;
    id += p.id_base;
    if (id > *dst) *dst = id;
  }
}

#define EA_ID_MAP_MAXN 8192
static int mask_tab_lds(int W) { return 128 + EA_MASK_MAXR * (int)sizeof(MaskRow) + W * 20; }

static int mask_post_fill(MaskPostParams& p, const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size, int in_h,
                          int in_w, int H, int W, float threshold, float offset) {
  if (n_masks <= 0 || lh <= 0 || lw <= 0 || img_size <= 0 || in_h <= 0 || in_w <= 0 || H <= 0 || W <= 0) return EA_ERR_BAD_SHAPE;
  if (in_h > img_size || in_w > img_size || (long long)H * W > 0x3fffffffLL || lw > 4096) return EA_ERR_BAD_SHAPE;
  p.low = low_res; p.index = index; p.mask = nullptr; p.stats = nullptr; p.idmap = nullptr;
  p.lh = lh; p.lw = lw; p.S = img_size; p.in_h = in_h; p.in_w = in_w; p.H = H; p.W = W;
  p.thr = threshold; p.off = offset;
  p.n = n_masks; p.id_base = 0; p.band = 1;
  return EA_OK;
}

extern "C" int ea_sam_mask_postprocess_ex(const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size, int in_h,
                                          int in_w, int H, int W, float threshold, float offset, unsigned char* mask, int* stats,
                                          int kernel, void* stream) {
  if (!low_res || !stats) return EA_ERR_BAD_ARG;   // mask == NULL: statistics only (no mask is written)
  if (kernel < 0 || kernel > 2) return EA_ERR_BAD_ARG;
  MaskPostParams p;
  const int st = mask_post_fill(p, low_res, index, n_masks, lh, lw, img_size, in_h, in_w, H, W, threshold, offset);
  if (st != EA_OK) return st;
  p.mask = mask; p.stats = stats;
  if (kernel == 2 && W > EA_MASK_MAXW) return EA_ERR_UNSUPPORTED;
  if (kernel == 1 || W > EA_MASK_MAXW) {
    int bx = (H * W + 256 * 8 - 1) / (256 * 8);      // ~8 pixels per thread
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    auto kfn = ea_mask_post_kernel;
    EA_LAUNCH(kfn, dim3((unsigned)bx, (unsigned)n_masks), dim3(256), 128, stream, p);
    return ea_launch_status();
  }
  // band height: ~16 pixels per thread, at least 1024 workgroups over the launch where the masks are few
  int band = (16 * 256 + W - 1) / W;
  if (band < 1) band = 1;
  if (band > EA_MASK_MAXR) band = EA_MASK_MAXR;
  while (band > 1 && (long long)((H + band - 1) / band) * n_masks < 1024) band = (band + 1) / 2;
  p.band = band;
  auto kfn = ea_mask_post_tab_kernel;
  EA_LAUNCH(kfn, dim3((unsigned)((H + band - 1) / band), (unsigned)n_masks), dim3(256), mask_tab_lds(W), stream, p);
  return ea_launch_status();
}

extern "C" int ea_sam_mask_postprocess_indexed(const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size,
                                               int in_h, int in_w, int H, int W, float threshold, float offset, unsigned char* mask,
                                               int* stats, void* stream) {
  return ea_sam_mask_postprocess_ex(low_res, index, n_masks, lh, lw, img_size, in_h, in_w, H, W, threshold, offset, mask, stats, 0, stream);
}

extern "C" int ea_sam_mask_postprocess(const float* low_res, int n_masks, int lh, int lw, int img_size, int in_h, int in_w,
                                       int H, int W, float threshold, float offset, unsigned char* mask, int* stats,
                                       void* stream) {
  return ea_sam_mask_postprocess_ex(low_res, nullptr, n_masks, lh, lw, img_size, in_h, in_w, H, W, threshold, offset, mask, stats, 0, stream);
}

extern "C" int ea_sam_id_map(const float* low_res, const int* index, int n_masks, int lh, int lw, int img_size, int in_h, int in_w,
                             int H, int W, float threshold, int id_base, int* idmap, void* stream) {
  if (!low_res || !idmap) return EA_ERR_BAD_ARG;
  if (id_base < 0) return EA_ERR_BAD_ARG;
  MaskPostParams p;
  const int st = mask_post_fill(p, low_res, index, n_masks, lh, lw, img_size, in_h, in_w, H, W, threshold, 0.0f);
  if (st != EA_OK) return st;
  if (W > EA_MASK_MAXW || H > 65535) return EA_ERR_UNSUPPORTED;
  if (!index && n_masks > EA_ID_MAP_MAXN) return EA_ERR_UNSUPPORTED;   // pieces need a selection to address
  p.idmap = idmap;
  auto kfn = ea_sam_id_map_kernel;
  // pieces of at most EA_ID_MAP_MAXN records (their places in `low` live in LDS); a piece raises the map over the pieces before
  for (int s0 = 0; s0 < n_masks; s0 += EA_ID_MAP_MAXN) {
    p.n = n_masks - s0 < EA_ID_MAP_MAXN ? n_masks - s0 : EA_ID_MAP_MAXN;
    p.index = index ? index + s0 : nullptr;
    p.id_base = id_base + s0;
    const int lds = 128 + EA_MASK_MAXR * (int)sizeof(MaskRow) + 256 * 20 + p.n * 4;
    EA_LAUNCH(kfn, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), lds, stream, p);
    const int ls = ea_launch_status();
    if (ls != EA_OK) return ls;
  }
  return EA_OK;
}

// ---- fused entry points (several launches on the caller's stream, one call) ----
extern "C" int ea_groupnorm_silu_conv3x3(const ea_conv_src* src, const float* gamma, const float* beta, int groups,
                                         float eps, void* norm_out, const void* W, int Cout, const ea_epilogue* epi,
                                         void* workspace, size_t ws_bytes, void* stream) {
  if (!src || !norm_out) return EA_ERR_BAD_ARG;
  const int C = src->c1 + src->c2;
  const size_t gn_ws = ea_groupnorm_workspace_bytes(src->B, src->Hin * src->Win, C, groups);
  if (ws_bytes < gn_ws) return EA_ERR_WORKSPACE;
  int st = ea_groupnorm_f16(src->x1, src->c1, src->x2, src->c2, src->x2_add, gamma, beta, norm_out, src->B,
                            src->Hin * src->Win, groups, eps, 1, workspace, gn_ws, stream);
  if (st != EA_OK) return st;
  ea_conv_src s2 = *src;
  s2.x1 = norm_out; s2.c1 = C; s2.x2 = nullptr; s2.c2 = 0; s2.x2_add = nullptr;
  // the GroupNorm partials are consumed by the apply kernel, stream-ordered before the conv: reuse the workspace
  return ea_conv2d_f16(&s2, W, Cout, epi, workspace, ws_bytes, stream);
}

extern "C" int ea_ln_gemm_f16(const void* x, int in_f32, const float* gamma, const float* beta, float eps,
                              void* ln_out, const void* W, int ldw, int M, int N, int K, const ea_epilogue* epi,
                              void* workspace, size_t ws_bytes, void* stream) {
  if (!ln_out) return EA_ERR_BAD_ARG;
  int st = ea_layernorm_f16(x, in_f32, gamma, beta, ln_out, M, K, eps, stream);
  if (st != EA_OK) return st;
  return ea_gemm_f16(ln_out, K, W, ldw, M, N, K, 1, 0, 0, 0, 0, epi, workspace, ws_bytes, stream);
}
