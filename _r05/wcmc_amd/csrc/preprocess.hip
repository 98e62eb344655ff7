// Per-image preprocessing of raw renderer samples (the step before the KPCN-Manifold path).
//
// Replaces the numpy code of the reference's support/datasets.py:
//   DenoiseDataset._preprocess_llpm :302-361   raw (h,w,s,C) -> (h,w,s,37) path descriptors (log / sqrt transforms)
//   DenoiseDataset._preprocess_kpcn :487-582   raw (h,w,s,C) -> (h,w,44) per-pixel statistics over the spp axis,
//                                              albedo factorisation, log transform, depth normalisation, gradients
//   DenoiseDataset._gradients       :286-300   backward differences with a zero first column / row
// Raw channel map: datasets.py:223-267 (C = 38 + 11 * (MAX_DEPTH + 1) = 104 at MAX_DEPTH = 5).
// All three are streaming, HBM-bound kernels: the raw buffer (416 B per sample) is read once per function.
#include "common.h"

namespace wcmc {

struct PPMap { int radiance, diffuse, bounce, albedo, normal, depth, pweight, rwow, light, thr, rough, d; };

static PPMap pp_map(int max_depth) {
  const int d = max_depth + 1;
  PPMap m;
  m.radiance = 2; m.diffuse = 5; m.bounce = 24 + d * 6; m.albedo = 24 + d * 7; m.normal = 27 + d * 7;
  m.depth = 30 + d * 7; m.pweight = 31 + d * 7; m.rwow = 32 + d * 7; m.light = 35 + d * 7;
  m.thr = 38 + d * 7; m.rough = 38 + d * 10; m.d = d;
  return m;
}

// generic form: one thread per (sample, output channel)
__device__ __forceinline__ float pp_llpm_value(const float* r, int c, const PPMap& m, int base) {
  // r: the sample's raw record shifted by `base` channels (0 for a global pointer, m.bounce for the LDS tile)
  if (c < 1) return logf(r[m.pweight - base] + 1e-6f) / 90.0f;
  if (c < 4) return logf(r[m.rwow + c - 1 - base] + 1e-6f) / 30.0f;
  if (c < 7) return logf(r[m.light + c - 4 - base] + 1e-8f) / 10.0f;
  if (c < 7 + 3 * m.d) return logf(r[m.thr + c - 7 - base] + 1e-6f) / 30.0f;
  if (c < 7 + 4 * m.d) return r[m.bounce + c - 7 - 3 * m.d - base] / 19.0f;
  return sqrtf(r[m.rough + c - 7 - 4 * m.d - base]);
}

__global__ __launch_bounds__(256) void pp_llpm_kernel(const float* __restrict__ raw, float* __restrict__ out, int64_t n,
                                                      int C, PPMap m) {
  const int OC = 7 + 5 * m.d;                       // 1 + 3 + 3 + 3d + d + d
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * OC;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / OC;
    out[idx] = pp_llpm_value(raw + i * C, (int)(idx - i * OC), m, 0);
  }
}

// Tiled form (16-byte aligned records): every channel the function reads lies in [bounce, C) -- 44 of the 104
// channels at MAX_DEPTH 5.  A block stages that range of 256 consecutive samples in LDS with 16-byte loads
// (176-byte runs per record; the first 240 bytes of a record are never requested), then writes the 256 x 37
// outputs as one contiguous run.
__global__ __launch_bounds__(256) void pp_llpm_tiled_kernel(const float* __restrict__ raw, float* __restrict__ out,
                                                            int64_t n, int C, PPMap m) {
  extern __shared__ __attribute__((aligned(16))) float pp_tile[];
  const int W4 = (C - m.bounce) / 4;                // float4 per record (11)
  const int LD = W4 * 4 + 1;                        // odd row pitch: conflict-free column reads
  const int OC = 7 + 5 * m.d;
  for (int64_t s0 = (int64_t)blockIdx.x * 256; s0 < n; s0 += (int64_t)gridDim.x * 256) {
    const int cnt = (int)min((int64_t)256, n - s0);
    for (int t = threadIdx.x; t < cnt * W4; t += 256) {
      const int j = t / W4, q = t - j * W4;
      const float4 v = *reinterpret_cast<const float4*>(raw + (s0 + j) * C + m.bounce + q * 4);
      float* d = pp_tile + j * LD + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * OC; t += 256) {
      const int j = t / OC, c = t - j * OC;
      out[s0 * OC + t] = pp_llpm_value(pp_tile + j * LD, c, m, m.bounce);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void pp_gradients_kernel(const float* __restrict__ buf, float* __restrict__ out, int h,
                                                           int w, int c) {
  const int64_t total = (int64_t)h * w * c;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % c);
    const int64_t p = idx / c;
    const int x = (int)(p % w), y = (int)(p / w);
    const float v = buf[idx];
    out[p * 2 * c + ch] = x > 0 ? v - buf[idx - c] : 0.f;
    out[p * 2 * c + c + ch] = y > 0 ? v - buf[idx - (int64_t)w * c] : 0.f;
  }
}

// mean and population variance over the s samples of one raw channel group (numpy .mean(2) / .var(2))
template <int NC, class F>
__device__ __forceinline__ void pp_mean_var(const float* __restrict__ px, int s, int C, F f, float* mean, float* var) {
  float sum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) sum[c] = 0.f;
  for (int k = 0; k < s; ++k)
#pragma unroll
    for (int c = 0; c < NC; ++c) sum[c] += f(px + (int64_t)k * C, c);
#pragma unroll
  for (int c = 0; c < NC; ++c) { mean[c] = sum[c] / (float)s; sum[c] = 0.f; }
  for (int k = 0; k < s; ++k)
#pragma unroll
    for (int c = 0; c < NC; ++c) { const float d = f(px + (int64_t)k * C, c) - mean[c]; sum[c] += d * d; }
#pragma unroll
  for (int c = 0; c < NC; ++c) var[c] = sum[c] / (float)s;
}

// output channel offsets of the 44-channel KPCN buffer
constexpr int KP_DIFF = 0, KP_SPEC = 10, KP_NORM = 20, KP_DEPTH = 30, KP_ALB = 34, KP_C = 44;

// pass 1: everything that needs only the pixel's own samples; depth stays raw in the workspace
__global__ __launch_bounds__(256) void pp_kpcn_stats_kernel(const float* __restrict__ raw, float* __restrict__ out,
                                                            float* __restrict__ ws, int64_t npix, int s, int C, PPMap m) {
  const float eps = 0.00316f;
  float bmax = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const float* px = raw + p * s * C;
    float* o = out + p * KP_C;
    float mean[3], var[3];
    const float spp = (float)s;
    pp_mean_var<3>(px, s, C, [&](const float* r, int c) { return r[m.normal + c]; }, mean, var);
    o[KP_NORM + 0] = mean[0]; o[KP_NORM + 1] = mean[1]; o[KP_NORM + 2] = mean[2];
    o[KP_NORM + 3] = ((var[0] + var[1] + var[2]) / 3.0f) / spp;
    float dm[1], dv[1];
    pp_mean_var<1>(px, s, C, [&](const float* r, int) { return r[m.depth]; }, dm, dv);
    ws[2 * p] = dm[0]; ws[2 * p + 1] = dv[0];
    bmax = fmaxf(bmax, dm[0]);
    float alb[3];
    pp_mean_var<3>(px, s, C, [&](const float* r, int c) { return r[m.albedo + c]; }, alb, var);
    o[KP_ALB + 0] = alb[0]; o[KP_ALB + 1] = alb[1]; o[KP_ALB + 2] = alb[2];
    o[KP_ALB + 3] = ((var[0] + var[1] + var[2]) / 3.0f) / spp;
    const float a0 = alb[0] + eps, a1 = alb[1] + eps, a2 = alb[2] + eps;
    const float albedo_sqr = (a0 * a0 + a1 * a1 + a2 * a2) / 3.0f;
    pp_mean_var<3>(px, s, C, [&](const float* r, int c) { return fmaxf(r[m.diffuse + c], 0.f); }, mean, var);
    const float diffuse_v = ((var[0] + var[1] + var[2]) / 3.0f) / spp;
    o[KP_DIFF + 0] = mean[0] / a0; o[KP_DIFF + 1] = mean[1] / a1; o[KP_DIFF + 2] = mean[2] / a2;
    o[KP_DIFF + 3] = diffuse_v / albedo_sqr;
    pp_mean_var<3>(px, s, C,
                   [&](const float* r, int c) {
                     return fmaxf(fmaxf(r[m.radiance + c], 0.f) - fmaxf(r[m.diffuse + c], 0.f), 0.f);
                   },
                   mean, var);
    const float specular_v = ((var[0] + var[1] + var[2]) / 3.0f) / spp;
    const float s0 = 1.0f + mean[0], s1 = 1.0f + mean[1], s2 = 1.0f + mean[2];
    const float specular_sqr = (s0 * s0 + s1 * s1 + s2 * s2) / 3.0f;
    o[KP_SPEC + 0] = logf(s0); o[KP_SPEC + 1] = logf(s1); o[KP_SPEC + 2] = logf(s2);
    o[KP_SPEC + 3] = specular_v / specular_sqr;
  }
  // image maximum of the mean depth (only its positive part matters: datasets.py:517-520 scales when max > 0)
  bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(ws + 2 * npix), __float_as_int(bmax));
}

// pass 1, one lane per (pixel, sample) for power-of-two spp <= 64: the s lanes of a pixel sit side by side, so a
// wave reads 64 consecutive raw records (26 KB of contiguous memory) and the statistics are xor-shuffle trees
// (the per-pixel form above walks each pixel's records from a single lane, 3.3 KB apart across the wave).
template <bool VEC>
__global__ __launch_bounds__(256) void pp_kpcn_stats_lanes_kernel(const float* __restrict__ raw, float* __restrict__ out,
                                                                  float* __restrict__ ws, int64_t npix, int s, int C,
                                                                  PPMap m) {
  const float eps = 0.00316f, spp = (float)s;
  const int ppw = 64 / s;                                       // pixels per wave
  const int lane = threadIdx.x & 63, k = lane & (s - 1), pl = lane / s;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float bmax = 0.f;
  for (int64_t p0 = wave * ppw; p0 < npix; p0 += nwaves * ppw) {
    const int64_t p = p0 + pl;
    const bool ok = p < npix;
    const float* r = raw + ((ok ? p : npix - 1) * s + k) * C;
    // v: normal(3) depth(1) albedo(3) diffuse+(3) specular+(3)
    float v[13], in[13];       // in: radiance(3) diffuse(3) albedo(3) normal(3) depth(1)
    if (VEC) {                 // 16-byte aligned records, albedo at an even channel with albedo+2 a multiple of 4
      const float2 a = *reinterpret_cast<const float2*>(r + 2), b = *reinterpret_cast<const float2*>(r + 4),
                   c2 = *reinterpret_cast<const float2*>(r + 6), d2 = *reinterpret_cast<const float2*>(r + m.albedo);
      const float4 e = *reinterpret_cast<const float4*>(r + m.albedo + 2);
      in[0] = a.x; in[1] = a.y; in[2] = b.x; in[3] = b.y; in[4] = c2.x; in[5] = c2.y;
      in[6] = d2.x; in[7] = d2.y; in[8] = e.x; in[9] = e.y; in[10] = e.z; in[11] = e.w; in[12] = r[m.depth];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        in[c] = r[m.radiance + c]; in[3 + c] = r[m.diffuse + c]; in[6 + c] = r[m.albedo + c]; in[9 + c] = r[m.normal + c];
      }
      in[12] = r[m.depth];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] = in[9 + c];
      v[4 + c] = in[6 + c];
      const float df = fmaxf(in[3 + c], 0.f);
      v[7 + c] = df;
      v[10 + c] = fmaxf(fmaxf(in[c], 0.f) - df, 0.f);
    }
    v[3] = in[12];
    float mean[13], var[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) {
      float a = v[c];
      for (int o = 1; o < s; o <<= 1) a += __shfl_xor(a, o, 64);
      mean[c] = a / spp;
      const float d = v[c] - mean[c];
      float q = d * d;
      for (int o = 1; o < s; o <<= 1) q += __shfl_xor(q, o, 64);
      var[c] = q / spp;
    }
    if (ok && k == 0) {
      float* o = out + p * KP_C;
      o[KP_NORM + 0] = mean[0]; o[KP_NORM + 1] = mean[1]; o[KP_NORM + 2] = mean[2];
      o[KP_NORM + 3] = ((var[0] + var[1] + var[2]) / 3.0f) / spp;
      ws[2 * p] = mean[3]; ws[2 * p + 1] = var[3];
      bmax = fmaxf(bmax, mean[3]);
      o[KP_ALB + 0] = mean[4]; o[KP_ALB + 1] = mean[5]; o[KP_ALB + 2] = mean[6];
      o[KP_ALB + 3] = ((var[4] + var[5] + var[6]) / 3.0f) / spp;
      const float a0 = mean[4] + eps, a1 = mean[5] + eps, a2 = mean[6] + eps;
      const float albedo_sqr = (a0 * a0 + a1 * a1 + a2 * a2) / 3.0f;
      o[KP_DIFF + 0] = mean[7] / a0; o[KP_DIFF + 1] = mean[8] / a1; o[KP_DIFF + 2] = mean[9] / a2;
      o[KP_DIFF + 3] = (((var[7] + var[8] + var[9]) / 3.0f) / spp) / albedo_sqr;
      const float s0 = 1.0f + mean[10], s1 = 1.0f + mean[11], s2 = 1.0f + mean[12];
      const float specular_sqr = (s0 * s0 + s1 * s1 + s2 * s2) / 3.0f;
      o[KP_SPEC + 0] = logf(s0); o[KP_SPEC + 1] = logf(s1); o[KP_SPEC + 2] = logf(s2);
      o[KP_SPEC + 3] = (((var[10] + var[11] + var[12]) / 3.0f) / spp) / specular_sqr;
    }
  }
  bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64));
  if (lane == 0) atomicMax(reinterpret_cast<int*>(ws + 2 * npix), __float_as_int(bmax));
}

// pass 2: depth normalisation + clip, and the backward differences of the five feature groups.  One thread per
// (pixel, output channel): a wave touches consecutive floats of the 176-byte pixel records.
__global__ __launch_bounds__(256) void pp_kpcn_finish_kernel(float* __restrict__ out, const float* __restrict__ ws, int h,
                                                             int w, int s) {
  const int64_t npix = (int64_t)h * w;
  const float maxd = ws[2 * npix];
  auto depth_of = [&](int64_t p) {
    float d = ws[2 * p];
    if (maxd > 0.f) d = d / maxd;
    return fminf(fmaxf(d, 0.f), 1.f);
  };
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < npix * KP_C;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / KP_C;
    const int c = (int)(idx - p * KP_C);
    const int x = (int)(p % w), y = (int)(p / w);
    if (c >= KP_DEPTH && c < KP_ALB) {
      float v;
      if (c == KP_DEPTH) v = depth_of(p);
      else if (c == KP_DEPTH + 1) { v = ws[2 * p + 1]; if (maxd > 0.f) v = v / (maxd * maxd * (float)s); }
      else if (c == KP_DEPTH + 2) v = x > 0 ? depth_of(p) - depth_of(p - 1) : 0.f;
      else v = y > 0 ? depth_of(p) - depth_of(p - w) : 0.f;
      out[idx] = v;
      continue;
    }
    const int g0 = c < KP_SPEC ? KP_DIFF : c < KP_NORM ? KP_SPEC : c < KP_DEPTH ? KP_NORM : KP_ALB;
    const int j = c - g0;
    if (j < 4) continue;                                  // values and variance: final since pass 1
    const int src = g0 + (j < 7 ? j - 4 : j - 7);
    const float v = out[p * KP_C + src];
    if (j < 7) out[idx] = x > 0 ? v - out[(p - 1) * KP_C + src] : 0.f;
    else out[idx] = y > 0 ? v - out[(p - w) * KP_C + src] : 0.f;
  }
}

static unsigned pp_grid(int64_t work) {
  const int64_t b = ceil_div64(work, 256);
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace wcmc

using namespace wcmc;

extern "C" int wcmc_preprocess_llpm(const float* raw, int64_t nsamples, int C, int max_depth, float* out, void* stream) {
  WCMC_REQUIRE(raw && out && nsamples > 0 && max_depth >= 0 && C >= 38 + 11 * (max_depth + 1), WCMC_ERR_BAD_ARG,
               "preprocess_llpm: bad argument (raw needs >= 38 + 11*(max_depth+1) channels)");
  const PPMap m = pp_map(max_depth);
  if (C % 4 == 0 && m.bounce % 4 == 0 && aligned16(raw)) {
    const size_t lds = (size_t)256 * (C - m.bounce + 1) * sizeof(float);
    const int64_t blocks = ceil_div64(nsamples, 256);
    hipLaunchKernelGGL(pp_llpm_tiled_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), lds,
                       (hipStream_t)stream, raw, out, nsamples, C, m);
  } else {
    hipLaunchKernelGGL(pp_llpm_kernel, dim3(pp_grid(nsamples * (7 + 5 * m.d))), dim3(256), 0, (hipStream_t)stream, raw,
                       out, nsamples, C, m);
  }
  return check_launch("preprocess_llpm");
}

extern "C" int wcmc_gradients(const float* buf, int h, int w, int c, float* out, void* stream) {
  WCMC_REQUIRE(buf && out && h > 0 && w > 0 && c > 0, WCMC_ERR_BAD_ARG, "gradients: bad argument");
  hipLaunchKernelGGL(pp_gradients_kernel, dim3(pp_grid((int64_t)h * w * c)), dim3(256), 0, (hipStream_t)stream, buf, out, h,
                     w, c);
  return check_launch("gradients");
}

extern "C" size_t wcmc_preprocess_kpcn_workspace_bytes(int h, int w) {
  if (h <= 0 || w <= 0) return 0;
  return ((size_t)2 * h * w + 4) * sizeof(float);
}

extern "C" int wcmc_preprocess_kpcn(const float* raw, int h, int w, int s, int C, int max_depth, float* out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  WCMC_REQUIRE(raw && out && workspace && h > 0 && w > 0 && s > 0 && max_depth >= 0 && C >= 38 + 11 * (max_depth + 1),
               WCMC_ERR_BAD_ARG, "preprocess_kpcn: bad argument");
  WCMC_REQUIRE(workspace_bytes >= wcmc_preprocess_kpcn_workspace_bytes(h, w), WCMC_ERR_WORKSPACE,
               "preprocess_kpcn: workspace too small");
  const PPMap m = pp_map(max_depth);
  const int64_t npix = (int64_t)h * w;
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws + 2 * npix, 0, sizeof(float), st) != hipSuccess) {
    set_error("preprocess_kpcn: memset failed");
    return WCMC_ERR_LAUNCH;
  }
  const bool vec = C % 4 == 0 && aligned16(raw) && m.radiance == 2 && m.diffuse == 5 && m.albedo % 2 == 0 &&
                   (m.albedo + 2) % 4 == 0 && m.normal == m.albedo + 3 && m.depth == m.albedo + 6;
  if (s <= 64 && (s & (s - 1)) == 0 && vec)
    hipLaunchKernelGGL(pp_kpcn_stats_lanes_kernel<true>, dim3(pp_grid(npix * s)), dim3(256), 0, st, raw, out, ws, npix, s, C, m);
  else if (s <= 64 && (s & (s - 1)) == 0)
    hipLaunchKernelGGL(pp_kpcn_stats_lanes_kernel<false>, dim3(pp_grid(npix * s)), dim3(256), 0, st, raw, out, ws, npix, s, C, m);
  else
    hipLaunchKernelGGL(pp_kpcn_stats_kernel, dim3(pp_grid(npix)), dim3(256), 0, st, raw, out, ws, npix, s, C, m);
  int rc = check_launch("preprocess_kpcn(stats)");
  if (rc) return rc;
  hipLaunchKernelGGL(pp_kpcn_finish_kernel, dim3(pp_grid(npix * KP_C)), dim3(256), 0, st, out, ws, h, w, s);
  return check_launch("preprocess_kpcn(finish)");
}

// ------------------------------------------------------------------ patch batch assembly (datasets.py:1026-1146)
// What DenoiseDataset.__getitem__ + _sample_patches + _transpose do per patch on the loader's CPU worker, for the
// KPCN base model: crop a P x P window out of the preprocessed per-image buffers and lay the batch dictionary's
// tensors out channel-first --
//   kpcn_diffuse_in  = [kpcn 0:10, kpcn 20:44 (, mean_s llpm[..., 0])]       34 (+1) channels      :1080,1099-1103
//   kpcn_specular_in = [kpcn 10:44 (, mean_s llpm[..., 0])]                  34 (+1)               :1081,1104-1107
//   kpcn_diffuse_buffer = kpcn 0:3, kpcn_specular_buffer = kpcn 10:13, kpcn_albedo = kpcn 34:37 + 0.00316  :1082-1084
//   paths = llpm[..., 1:37] as (S, 36, P, P)                                                       :1110
//   target_total = gt 0:3, target_diffuse = gt 3:6 / (gt 6:9 + 0.00316), target_specular = log(1 + total - diffuse)  :1117-1126
// One thread per (patch, y, x): every output plane is written as coalesced rows, the inputs are read once.
namespace wcmc {
struct PatchOut {
  float *din, *sin, *dbuf, *sbuf, *alb, *paths, *tdif, *tspec, *ttot;
};
__global__ __launch_bounds__(256) void pp_assemble_kpcn_kernel(const float* __restrict__ kpcn, const float* __restrict__ llpm,
                                                               const float* __restrict__ gt, const int* __restrict__ origins,
                                                               PatchOut o, int B, int H, int W, int S, int P) {
  const int64_t total = (int64_t)B * P * P;
  const int cin = llpm ? 35 : 34;
  const int64_t plane = (int64_t)P * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % P), y = (int)((i / P) % P), b = (int)(i / plane);
    const int r = origins[2 * b] + y, c = origins[2 * b + 1] + x;             // (row, column) of the image
    const int64_t pix = (int64_t)r * W + c;
    const float* k = kpcn + pix * 44;
    const int64_t po = (int64_t)y * P + x;
    float* din = o.din + (int64_t)b * cin * plane + po;
    float* sin = o.sin + (int64_t)b * cin * plane + po;
    for (int ch = 0; ch < 10; ++ch) din[ch * plane] = k[ch];
    for (int ch = 20; ch < 44; ++ch) din[(ch - 10) * plane] = k[ch];
    for (int ch = 10; ch < 44; ++ch) sin[(ch - 10) * plane] = k[ch];
    for (int ch = 0; ch < 3; ++ch) {
      o.dbuf[((int64_t)b * 3 + ch) * plane + po] = k[ch];
      o.sbuf[((int64_t)b * 3 + ch) * plane + po] = k[10 + ch];
      o.alb[((int64_t)b * 3 + ch) * plane + po] = k[34 + ch] + 0.00316f;
    }
    if (llpm) {
      const float* l = llpm + pix * S * 37;
      float pw = 0.f;
      for (int s = 0; s < S; ++s) {
        pw += l[s * 37];
        float* pp = o.paths + (((int64_t)b * S + s) * 36) * plane + po;
        for (int ch = 0; ch < 36; ++ch) pp[ch * plane] = l[s * 37 + 1 + ch];
      }
      pw /= (float)S;
      din[34 * plane] = pw;
      sin[34 * plane] = pw;
    }
    const float* g = gt + pix * 9;
    for (int ch = 0; ch < 3; ++ch) {
      const float tot = g[ch], dif = g[3 + ch], alb = g[6 + ch];
      o.ttot[((int64_t)b * 3 + ch) * plane + po] = tot;
      o.tdif[((int64_t)b * 3 + ch) * plane + po] = dif / (alb + 0.00316f);
      o.tspec[((int64_t)b * 3 + ch) * plane + po] = logf(1.f + tot - dif);
    }
  }
}
}  // namespace wcmc

extern "C" int wcmc_assemble_kpcn_patches(const float* kpcn, const float* llpm, const float* gt, const int* origins,
                                          int B, int H, int W, int S, int P, float* diffuse_in, float* specular_in,
                                          float* diffuse_buffer, float* specular_buffer, float* albedo, float* paths,
                                          float* target_diffuse, float* target_specular, float* target_total,
                                          void* stream) {
  WCMC_REQUIRE(kpcn && gt && origins && B > 0 && H > 0 && W > 0 && P > 0 && P <= H && P <= W && diffuse_in &&
                   specular_in && diffuse_buffer && specular_buffer && albedo && target_diffuse && target_specular &&
                   target_total && (!llpm || (paths && S > 0)),
               WCMC_ERR_BAD_ARG, "assemble_kpcn_patches: bad argument");
  wcmc::PatchOut o{diffuse_in, specular_in, diffuse_buffer, specular_buffer, albedo, paths, target_diffuse,
                   target_specular, target_total};
  const int64_t total = (int64_t)B * P * P;
  const unsigned grid = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
  hipLaunchKernelGGL(wcmc::pp_assemble_kpcn_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, kpcn, llpm, gt, origins, o,
                     B, H, W, S, P);
  return wcmc::check_launch("assemble_kpcn_patches");
}
