// HBM-bound glue kernels of the PathNet / U-Net / interface path (gfx950).
// Everything here moves 16 bytes per lane along the channel axis of an NHWC view.
//
//   wcmc_to_nhwc / wcmc_from_nhwc   batch tensors <-> NHWC views (LDS-tiled transpose)
//   wcmc_maxpool2_*, wcmc_upsample2_*   sbmc.modules.Autoencoder glue (support/networks.py:20-22)
//   wcmc_spp_reduce / wcmc_spp_broadcast  support/networks.py:35-36,39-40
//   wcmc_pbuffer_cat_*                 support/interfaces.py:165-176
#include "common.h"

namespace wcmc {

// ------------------------------------------------------------------ NCHW-style <-> NHWC
// Tile: 64 consecutive x of one image row times 32 channels through a padded LDS tile.
__global__ __launch_bounds__(256) void to_nhwc_kernel(const float* __restrict__ src, int64_t ssn, int64_t ssc,
                                                      int64_t ssh, int64_t ssw, float* __restrict__ dst,
                                                      int64_t dsn, int64_t dsh, int64_t dsw, int C, int H, int W) {
  __shared__ float tile[32][65];
  const int xt = blockIdx.x * 64, y = blockIdx.y % H, n = blockIdx.y / H;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int c0 = blockIdx.z * 32; c0 < C; c0 += gridDim.z * 32) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + grp * 8 + i, x = xt + lane;
      tile[grp * 8 + i][lane] =
          (c < C && x < W) ? src[(int64_t)n * ssn + (int64_t)c * ssc + (int64_t)y * ssh + (int64_t)x * ssw] : 0.f;
    }
    __syncthreads();
    // 8 threads write the 32 channels of one pixel as float4; 32 pixels per pass, 2 passes
    const int c4 = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int px = (threadIdx.x >> 3) + 32 * ps, x = xt + px;
      if (x < W && c0 + c4 < C) {
        float4 v = make_float4(tile[c4][px], tile[c4 + 1][px], tile[c4 + 2][px], tile[c4 + 3][px]);
        *reinterpret_cast<float4*>(dst + (int64_t)n * dsn + (int64_t)y * dsh + (int64_t)x * dsw + c0 + c4) = v;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void from_nhwc_kernel(const float* __restrict__ src, int64_t ssn, int64_t ssh,
                                                        int64_t ssw, float* __restrict__ dst, int64_t dsn,
                                                        int64_t dsc, int64_t dsh, int64_t dsw, int C, int H, int W) {
  __shared__ float tile[32][65];
  const int xt = blockIdx.x * 64, y = blockIdx.y % H, n = blockIdx.y / H;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int c0 = blockIdx.z * 32; c0 < C; c0 += gridDim.z * 32) {
    const int c4 = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int px = (threadIdx.x >> 3) + 32 * ps, x = xt + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x < W && c0 + c4 < C)
        v = *reinterpret_cast<const float4*>(src + (int64_t)n * ssn + (int64_t)y * ssh + (int64_t)x * ssw + c0 + c4);
      tile[c4][px] = v.x; tile[c4 + 1][px] = v.y; tile[c4 + 2][px] = v.z; tile[c4 + 3][px] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + grp * 8 + i, x = xt + lane;
      if (c < C && x < W)
        dst[(int64_t)n * dsn + (int64_t)c * dsc + (int64_t)y * dsh + (int64_t)x * dsw] = tile[grp * 8 + i][lane];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ generic float4-per-lane indexer
struct View { const float* p; int64_t sn, sh, sw; };
struct MView { float* p; int64_t sn, sh, sw; };
__device__ __forceinline__ float4 ld4(const View& v, int n, int y, int x, int c) {
  return *reinterpret_cast<const float4*>(v.p + n * v.sn + y * v.sh + x * v.sw + c);
}
__device__ __forceinline__ void st4(const MView& v, int n, int y, int x, int c, float4 a) {
  *reinterpret_cast<float4*>(v.p + n * v.sn + y * v.sh + x * v.sw + c) = a;
}
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 b) {
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// zero the lanes of a vector that lie at or beyond channel C (keeps pad channels of a buffer at 0)
__device__ __forceinline__ float4 f4_mask(float4 a, int c, int C) {
  if (c + 1 >= C) a.y = 0.f;
  if (c + 2 >= C) a.z = 0.f;
  if (c + 3 >= C) a.w = 0.f;
  return a;
}

#define WCMC_ITER_NHWC(N, H, W, C4)                                                                  \
  const int64_t total_ = (int64_t)(N) * (H) * (W) * (C4);                                            \
  for (int64_t idx_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx_ < total_;                 \
       idx_ += (int64_t)gridDim.x * blockDim.x)
#define WCMC_DECODE_NHWC(H, W, C4)                                              \
  const NhwvIndex ix_ = decode_nhwv(idx_, total_, (H), (W), (C4));              \
  const int c = ix_.v * 4, x = ix_.x, y = ix_.y, n = ix_.n;

__global__ void maxpool2_fwd_kernel(View in, MView out, int N, int Ho, int Wo, int C4, int C) {
  WCMC_ITER_NHWC(N, Ho, Wo, C4) {
    WCMC_DECODE_NHWC(Ho, Wo, C4)
    const float4 a = ld4(in, n, 2 * y, 2 * x, c), b = ld4(in, n, 2 * y, 2 * x + 1, c);
    const float4 d = ld4(in, n, 2 * y + 1, 2 * x, c), e = ld4(in, n, 2 * y + 1, 2 * x + 1, c);
    float4 m = make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y)),
                           fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w)));
    st4(out, n, y, x, c, f4_mask(m, c, C));
  }
}

// first maximum in window order (0,0),(0,1),(1,0),(1,1) takes the gradient, as ATen's max_pool2d does
__device__ __forceinline__ void route4(float a, float b, float d, float e, float g, float& ga, float& gb, float& gd,
                                       float& ge) {
  ga = gb = gd = ge = 0.f;
  float m = a; int k = 0;
  if (b > m) { m = b; k = 1; }
  if (d > m) { m = d; k = 2; }
  if (e > m) { m = e; k = 3; }
  if (k == 0) ga = g; else if (k == 1) gb = g; else if (k == 2) gd = g; else ge = g;
}
// add.p != null: dx = route(dy) + add -- the pooled tensor's input also feeds a skip connection (sbmc Autoencoder), whose gradient
// `add` autograd would otherwise sum in with one more pass over three tensors of this size
__global__ void maxpool2_bwd_kernel(View in, View dy, View add, MView dx, int N, int Ho, int Wo, int C4, int C) {
  WCMC_ITER_NHWC(N, Ho, Wo, C4) {
    WCMC_DECODE_NHWC(Ho, Wo, C4)
    const float4 a = ld4(in, n, 2 * y, 2 * x, c), b = ld4(in, n, 2 * y, 2 * x + 1, c);
    const float4 d = ld4(in, n, 2 * y + 1, 2 * x, c), e = ld4(in, n, 2 * y + 1, 2 * x + 1, c);
    const float4 g = f4_mask(ld4(dy, n, y, x, c), c, C);
    float4 ga, gb, gd, ge;
    route4(a.x, b.x, d.x, e.x, g.x, ga.x, gb.x, gd.x, ge.x);
    route4(a.y, b.y, d.y, e.y, g.y, ga.y, gb.y, gd.y, ge.y);
    route4(a.z, b.z, d.z, e.z, g.z, ga.z, gb.z, gd.z, ge.z);
    route4(a.w, b.w, d.w, e.w, g.w, ga.w, gb.w, gd.w, ge.w);
    if (add.p) {
      ga = f4_add(f4_mask(ld4(add, n, 2 * y, 2 * x, c), c, C), ga); gb = f4_add(f4_mask(ld4(add, n, 2 * y, 2 * x + 1, c), c, C), gb);
      gd = f4_add(f4_mask(ld4(add, n, 2 * y + 1, 2 * x, c), c, C), gd); ge = f4_add(f4_mask(ld4(add, n, 2 * y + 1, 2 * x + 1, c), c, C), ge);
    }
    st4(dx, n, 2 * y, 2 * x, c, ga); st4(dx, n, 2 * y, 2 * x + 1, c, gb);
    st4(dx, n, 2 * y + 1, 2 * x, c, gd); st4(dx, n, 2 * y + 1, 2 * x + 1, c, ge);
  }
}

// bilinear x2, align_corners=False: out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1]
// with the neighbour index clamped to the image.
__global__ void upsample2_fwd_kernel(View in, MView out, int N, int H, int W, int C4, int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  WCMC_ITER_NHWC(N, Ho, Wo, C4) {
    WCMC_DECODE_NHWC(Ho, Wo, C4)
    const int iy = y >> 1, ix = x >> 1;
    const int ny = (y & 1) ? min(iy + 1, H - 1) : max(iy - 1, 0);
    const int nx = (x & 1) ? min(ix + 1, W - 1) : max(ix - 1, 0);
    const float4 v00 = ld4(in, n, iy, ix, c), v01 = ld4(in, n, iy, nx, c);
    const float4 v10 = ld4(in, n, ny, ix, c), v11 = ld4(in, n, ny, nx, c);
    float4 r = f4_scale(0.5625f, v00);
    r = f4_fma(0.1875f, v01, r); r = f4_fma(0.1875f, v10, r); r = f4_fma(0.0625f, v11, r);
    st4(out, n, y, x, c, f4_mask(r, c, C));
  }
}
// dx[i] gathers from the <= 3x3 fine pixels it contributed to (per axis: weights of fine rows
// 2i-1 (.25), 2i (.75), 2i+1 (.75), 2i+2 (.25), plus the clamped contributions at the borders).
__device__ __forceinline__ int up_taps(int i, int L, int* fine, float* wt) {
  // fine index f receives from coarse i with weight: f=2i or 2i+1 -> .75 ; f=2i-1 or 2i+2 -> .25;
  // at the borders the clamped neighbour adds another .25 onto f=0 (i=0) and f=2L-1 (i=L-1).
  int k = 0;
  fine[k] = 2 * i; wt[k++] = (i == 0) ? 1.0f : 0.75f;
  fine[k] = 2 * i + 1; wt[k++] = (i == L - 1) ? 1.0f : 0.75f;
  if (i > 0) { fine[k] = 2 * i - 1; wt[k++] = 0.25f; }
  if (i < L - 1) { fine[k] = 2 * i + 2; wt[k++] = 0.25f; }
  return k;
}
__global__ void upsample2_bwd_kernel(View dy, MView dx, int N, int H, int W, int C4, int C) {
  WCMC_ITER_NHWC(N, H, W, C4) {
    WCMC_DECODE_NHWC(H, W, C4)
    int fy[4], fx[4]; float wy[4], wx[4];
    const int ky = up_taps(y, H, fy, wy), kx = up_taps(x, W, fx, wx);
    float4 acc = f4_zero();
    for (int a = 0; a < ky; ++a)
      for (int b = 0; b < kx; ++b) acc = f4_fma(wy[a] * wx[b], ld4(dy, n, fy[a], fx[b], c), acc);
    st4(dx, n, y, x, c, f4_mask(acc, c, C));
  }
}

__global__ void spp_reduce_kernel(View in, MView out, int B, int S, int H, int W, int C4, int C, float scale) {
  WCMC_ITER_NHWC(B, H, W, C4) {
    WCMC_DECODE_NHWC(H, W, C4)
    float4 acc = f4_zero();
    for (int s = 0; s < S; ++s) acc = f4_add(acc, ld4(in, n * S + s, y, x, c));
    st4(out, n, y, x, c, f4_mask(f4_scale(scale, acc), c, C));
  }
}
__global__ void spp_broadcast_kernel(View in, MView out, int B, int S, int H, int W, int C4, int C, float scale,
                                     int accumulate) {
  WCMC_ITER_NHWC(B, H, W, C4) {
    WCMC_DECODE_NHWC(H, W, C4)
    const float4 v = f4_mask(f4_scale(scale, ld4(in, n, y, x, c)), c, C);
    for (int s = 0; s < S; ++s) {
      float4 o = v;
      if (accumulate) {
        const View ov = {out.p, out.sn, out.sh, out.sw};
        o = f4_add(o, f4_mask(ld4(ov, n * S + s, y, x, c), c, C));
      }
      st4(out, n * S + s, y, x, c, o);
    }
  }
}

// ------------------------------------------------------------------ P-buffer statistics + concat
// One block = 64 consecutive x of one row.  Phase 1 (lane = pixel): base channels and the spp
// statistics go into a [64][CT+1] LDS tile; phase 2 writes 16-byte NHWC vectors.
__global__ __launch_bounds__(256) void pbuffer_cat_fwd_kernel(const float* __restrict__ base, int64_t bsn,
                                                              int64_t bsc, int64_t bsh, int64_t bsw,
                                                              const float* __restrict__ pb, int64_t psb, int64_t pss,
                                                              int64_t psc, int64_t psh, int64_t psw,
                                                              float* __restrict__ out, int64_t osn, int64_t osh,
                                                              int64_t osw, int S, int Cb, int Cp, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CT = Cb + Cp + 1, CT4 = (CT + 3) / 4 * 4, LD = CT4 + 1;
  const int xt = blockIdx.x * 64, y = blockIdx.y % H, b = blockIdx.y / H;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int x = xt + lane;
  for (int c = grp; c < Cb; c += 4)
    smem[lane * LD + c] = x < W ? base[(int64_t)b * bsn + (int64_t)c * bsc + (int64_t)y * bsh + (int64_t)x * bsw] : 0.f;
  if (grp == 0) {
    float varsum = 0.f;
    for (int c = 0; c < Cp; ++c) {
      float s1 = 0.f;
      const float* q = pb + (int64_t)b * psb + (int64_t)c * psc + (int64_t)y * psh + (int64_t)x * psw;
      if (x < W) for (int s = 0; s < S; ++s) s1 += q[(int64_t)s * pss];
      const float mean = s1 / (float)S;
      float s2 = 0.f;
      if (x < W) for (int s = 0; s < S; ++s) { const float d = q[(int64_t)s * pss] - mean; s2 += d * d; }
      smem[lane * LD + Cb + c] = mean;
      varsum += s2 / (float)(S - 1);            // unbiased, torch.var default (interfaces.py:165)
    }
    smem[lane * LD + Cb + Cp] = varsum / (float)Cp / (float)S;
    for (int c = CT; c < CT4; ++c) smem[lane * LD + c] = 0.f;
  }
  __syncthreads();
  const int nv = CT4 / 4;
  for (int i = threadIdx.x; i < 64 * nv; i += 256) {
    const int px = i / nv, c = (i - px * nv) * 4;
    if (xt + px < W) {
      const float* t = smem + px * LD + c;
      *reinterpret_cast<float4*>(out + (int64_t)b * osn + (int64_t)y * osh + (int64_t)(xt + px) * osw + c) =
          make_float4(t[0], t[1], t[2], t[3]);
    }
  }
}

__global__ void pbuffer_cat_bwd_kernel(const float* __restrict__ g, int64_t gsn, int64_t gsh, int64_t gsw,
                                       float* __restrict__ dp, int64_t psb, int64_t pss, int64_t psc, int64_t psh,
                                       int64_t psw, int B, int S, int Cb, int Cp, int H, int W) {
  const int64_t total = (int64_t)B * H * W;
  const float inv = 1.f / (float)S;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W); int64_t t = idx / W;
    const int y = (int)(t % H); const int b = (int)(t / H);
    const float* gp = g + (int64_t)b * gsn + (int64_t)y * gsh + (int64_t)x * gsw + Cb;
    for (int c = 0; c < Cp; ++c) {
      const float v = gp[c] * inv;
      float* q = dp + (int64_t)b * psb + (int64_t)c * psc + (int64_t)y * psh + (int64_t)x * psw;
      for (int s = 0; s < S; ++s) q[(int64_t)s * pss] = v;
    }
  }
}

// ------------------------------------------------------------------ KPCN recombination
// radiance = albedo * r_diffuse + exp(r_specular) - 1   (tail of sbmc.KPCN.forward; consumed at
// support/interfaces.py:207-211).  All tensors (N,C,H,W) with arbitrary element strides.
struct S4 { int64_t n, c, h, w; };
__global__ void recombine_fwd_kernel(const float* __restrict__ alb, S4 sa, const float* __restrict__ rd, S4 sd,
                                     const float* __restrict__ rs, S4 ss, float* __restrict__ out, int C, int H, int W,
                                     int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int n = (int)(t / C);
    const float a = alb[n * sa.n + c * sa.c + y * sa.h + x * sa.w];
    const float d = rd[n * sd.n + c * sd.c + y * sd.h + x * sd.w];
    const float s = rs[n * ss.n + c * ss.c + y * ss.h + x * ss.w];
    out[i] = a * d + expf(s) - 1.f;
  }
}
// g (contiguous) -> d r_diffuse = g * albedo ; d r_specular = g * exp(r_specular)   (contiguous outputs)
__global__ void recombine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ alb, S4 sa,
                                     const float* __restrict__ rs, S4 ss, float* __restrict__ dd,
                                     float* __restrict__ ds, int C, int H, int W, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int n = (int)(t / C);
    const float gv = g[i];
    dd[i] = gv * alb[n * sa.n + c * sa.c + y * sa.h + x * sa.w];
    ds[i] = gv * expf(rs[n * ss.n + c * ss.c + y * ss.h + x * ss.w]);
  }
}


// ------------------------------------------------------------------ image losses
// L1Loss (mean |x - ref|; train_kpcn.py:299-304, applied at interfaces.py:213-249) and RelativeMSE
// (0.5 * mean((x - ref)^2 / (ref^2 + eps)), losses.py:245-264) of one (N,C,H,W) pair in ONE pass: both sums are reduced
// together -- a block sums its grid-strided share in a fixed order (wave xor-shuffles, then the waves in order), the
// one-block finish launch adds the per-block partials in order -> bitwise reproducible, no atomics.
constexpr int IL_BLOCKS = 64;
__global__ __launch_bounds__(256) void image_loss_partial_kernel(const float* __restrict__ x, S4 sx, const float* __restrict__ r,
                                                                 S4 sr, float eps, float* __restrict__ partial, int C, int H,
                                                                 int W, int64_t total) {
  float a = 0.f, b = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int n = (int)(t / C);
    const float v = x[n * sx.n + c * sx.c + y * sx.h + xx * sx.w];
    const float q = r[n * sr.n + c * sr.c + y * sr.h + xx * sr.w];
    const float d = v - q;
    a += fabsf(d);
    b += (d * d) / (q * q + eps);
  }
  a = wave_sum(a); b = wave_sum(b);
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    partial[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}
__global__ __launch_bounds__(64) void image_loss_finish_kernel(const float* __restrict__ partial, int nblocks, float inv_total,
                                                               float* __restrict__ l1, float* __restrict__ relmse) {
  if (threadIdx.x != 0) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < nblocks; ++g) { a += partial[2 * g]; b += partial[2 * g + 1]; }
  if (l1) l1[0] = a * inv_total;
  if (relmse) relmse[0] = 0.5f * (b * inv_total);
}
// d L1 / dx = g * sign(x - ref) / total  (sign(0) = 0, as torch's L1Loss backward)
__global__ void l1_mean_bwd_kernel(const float* __restrict__ x, S4 sx, const float* __restrict__ r, S4 sr,
                                   const float* __restrict__ g, float inv_total, float* __restrict__ dx, int C, int H, int W,
                                   int64_t total) {
  const float gs = g[0] * inv_total;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int n = (int)(t / C);
    const float d = x[n * sx.n + c * sx.c + y * sx.h + xx * sx.w] - r[n * sr.n + c * sr.c + y * sr.h + xx * sr.w];
    dx[i] = d > 0.f ? gs : (d < 0.f ? -gs : (d == 0.f ? 0.f : d * gs));      // (NaN stays NaN)
  }
}

// ---- the sample-based interfaces' image losses (support/losses.py:267-320): SMAPE (LBMC), TonemappedMSE, TonemappedRelativeMSE
// (SBMC), forward in one pass + the one-block finish of image_loss_finish_kernel's kind, backward in one pass.
//   T(v) = max(v, 0) / (1 + max(v, 0))   (Reinhard, losses.py:234-242);  T'(v) = 1 / (1 + v)^2 for v >= 0 (torch.clamp passes the
//   gradient at the bound), 0 below.
//   kind 0  SMAPE:                 mean |x - r| / (eps + |x| + |r|), the denominator carries no gradient (losses.py:279-282)
//   kind 1  TonemappedMSE:         0.5 * mean (T(x) - T(r))^2
//   kind 2  TonemappedRelativeMSE: 0.5 * mean (T(x) - T(r))^2 / (T(r)^2 + eps)
__device__ __forceinline__ float reinhard(float v) { v = fmaxf(v, 0.f); return v / (1.f + v); }
template <int KIND>
__device__ __forceinline__ float loss2_term(float v, float q, float eps) {
  if (KIND == 0) return fabsf(v - q) / (eps + fabsf(v) + fabsf(q));
  const float tv = reinhard(v), tq = reinhard(q), d = tv - tq;
  return KIND == 1 ? d * d : (d * d) / (tq * tq + eps);
}
template <int KIND>
__device__ __forceinline__ float loss2_grad(float v, float q, float eps) {
  if (KIND == 0) {
    const float d = v - q, den = eps + fabsf(v) + fabsf(q);
    return d > 0.f ? 1.f / den : (d < 0.f ? -1.f / den : (d == 0.f ? 0.f : d));
  }
  const float tv = reinhard(v), tq = reinhard(q), d = tv - tq;
  const float dt = v >= 0.f ? 1.f / ((1.f + v) * (1.f + v)) : (v < 0.f ? 0.f : v);      // (NaN stays NaN)
  return KIND == 1 ? d * dt : d * dt / (tq * tq + eps);                                  // (the 0.5 and the 2 of the square cancel)
}
template <int KIND>
__global__ __launch_bounds__(256) void image_loss2_partial_kernel(const float* __restrict__ x, S4 sx, const float* __restrict__ r,
                                                                  S4 sr, float eps, float* __restrict__ partial, int C, int H,
                                                                  int W, int64_t total) {
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int n = (int)(t / C);
    a += loss2_term<KIND>(x[n * sx.n + c * sx.c + y * sx.h + xx * sx.w], r[n * sr.n + c * sr.c + y * sr.h + xx * sr.w], eps);
  }
  a = wave_sum(a);
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(64) void image_loss2_finish_kernel(const float* __restrict__ partial, int nblocks, float scale,
                                                                float* __restrict__ loss) {
  if (threadIdx.x != 0) return;
  float a = 0.f;
  for (int g = 0; g < nblocks; ++g) a += partial[g];
  loss[0] = a * scale;
}
template <int KIND>
__global__ void image_loss2_bwd_kernel(const float* __restrict__ x, S4 sx, const float* __restrict__ r, S4 sr, float eps,
                                       const float* __restrict__ g, float inv_total, float* __restrict__ dx, int C, int H, int W,
                                       int64_t total) {
  const float gs = g[0] * inv_total;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int n = (int)(t / C);
    dx[i] = gs * loss2_grad<KIND>(x[n * sx.n + c * sx.c + y * sx.h + xx * sx.w], r[n * sr.n + c * sr.c + y * sr.h + xx * sr.w], eps);
  }
}

// ---- clip_grad_norm_ over a model's gradient tensors (interfaces.py:454-458, 826-833): sums of squares per tensor chunk, the total
// norm and the clip factor by one block, then one scaling pass -- three launches for any number of tensors (<= GN_MAX per call group).
constexpr int GN_MAX = 96, GN_CHUNK = 256 * 16;
struct GNEntry { float* g; int64_t n; unsigned block0; };
struct GNTable { GNEntry e[GN_MAX]; int n; };
__device__ __forceinline__ const GNEntry& gn_find(const GNTable& t, unsigned b) {
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i)
    if (b >= t.e[i].block0) k = i;
  return t.e[k];
}
__global__ __launch_bounds__(256) void grad_sumsq_kernel(GNTable t, float* __restrict__ partial) {
  const GNEntry& q = gn_find(t, blockIdx.x);
  const int64_t i0 = (int64_t)(blockIdx.x - q.block0) * GN_CHUNK;
  float a = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < q.n && i < i0 + GN_CHUNK; i += 256) a += q.g[i] * q.g[i];
  a = wave_sum(a);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
// out[0] = total norm, out[1] = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_'s clamped coefficient)
__global__ __launch_bounds__(256) void grad_norm_finish_kernel(const float* __restrict__ partial, int nblocks, float max_norm,
                                                               float* __restrict__ out) {
  float a = 0.f;
  for (int g = threadIdx.x; g < nblocks; g += 256) a += partial[g];
  a = wave_sum(a);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(((red[0] + red[1]) + red[2]) + red[3]);
    const float coef = max_norm / (nrm + 1e-6f);
    out[0] = nrm;
    out[1] = coef < 1.f ? coef : 1.f;
  }
}
__global__ __launch_bounds__(256) void grad_scale_kernel(GNTable t, const float* __restrict__ coef) {
  const float c = coef[1];
  if (c >= 1.f) return;
  const GNEntry& q = gn_find(t, blockIdx.x);
  const int64_t i0 = (int64_t)(blockIdx.x - q.block0) * GN_CHUNK;
  for (int64_t i = i0 + threadIdx.x; i < q.n && i < i0 + GN_CHUNK; i += 256) q.g[i] *= c;
}

static unsigned grid_for(int64_t total) {
  const int64_t g = ceil_div64(total, 256);
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace wcmc

using namespace wcmc;

#define VIEW_OK(p, sn, sh, sw, C) nhwc_view_ok(p, sn, sh, sw, C)

extern "C" int wcmc_to_nhwc(const float* src, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw, float* dst,
                            int64_t dsn, int64_t dsh, int64_t dsw, int N, int C, int H, int W, void* stream) {
  WCMC_REQUIRE(src && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG, "to_nhwc: bad argument");
  WCMC_REQUIRE(VIEW_OK(dst, dsn, dsh, dsw, C), WCMC_ERR_ALIGNMENT, "to_nhwc: dst violates the NHWC-view contract");
  WCMC_REQUIRE((int64_t)N * H <= 65535, WCMC_ERR_BAD_ARG, "to_nhwc: N*H > 65535");
  const dim3 grid((unsigned)((W + 63) / 64), (unsigned)(N * H), (unsigned)((C + 31) / 32 < 4 ? (C + 31) / 32 : 4));
  hipLaunchKernelGGL(to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ssn, ssc, ssh, ssw, dst, dsn, dsh,
                     dsw, C, H, W);
  return check_launch("to_nhwc");
}

extern "C" int wcmc_from_nhwc(const float* src, int64_t ssn, int64_t ssh, int64_t ssw, float* dst, int64_t dsn,
                              int64_t dsc, int64_t dsh, int64_t dsw, int N, int C, int H, int W, void* stream) {
  WCMC_REQUIRE(dst && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG, "from_nhwc: bad argument");
  WCMC_REQUIRE(VIEW_OK(src, ssn, ssh, ssw, C), WCMC_ERR_ALIGNMENT, "from_nhwc: src violates the NHWC-view contract");
  WCMC_REQUIRE((int64_t)N * H <= 65535, WCMC_ERR_BAD_ARG, "from_nhwc: N*H > 65535");
  const dim3 grid((unsigned)((W + 63) / 64), (unsigned)(N * H), (unsigned)((C + 31) / 32 < 4 ? (C + 31) / 32 : 4));
  hipLaunchKernelGGL(from_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ssn, ssh, ssw, dst, dsn, dsc,
                     dsh, dsw, C, H, W);
  return check_launch("from_nhwc");
}

extern "C" int wcmc_maxpool2_fwd(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, float* y, int64_t ysn,
                                 int64_t ysh, int64_t ysw, int N, int H, int W, int C, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, WCMC_ERR_BAD_ARG,
               "maxpool2_fwd: bad shape (H,W must be even)");
  WCMC_REQUIRE(VIEW_OK(x, xsn, xsh, xsw, C) && VIEW_OK(y, ysn, ysh, ysw, C), WCMC_ERR_ALIGNMENT,
               "maxpool2_fwd: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * C4)), dim3(256), 0,
                     (hipStream_t)stream, View{x, xsn, xsh, xsw}, MView{y, ysn, ysh, ysw}, N, H / 2, W / 2, C4, C);
  return check_launch("maxpool2_fwd");
}

extern "C" int wcmc_maxpool2_bwd(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, const float* dy,
                                 int64_t dsn, int64_t dsh, int64_t dsw, float* dx, int64_t gsn, int64_t gsh,
                                 int64_t gsw, int N, int H, int W, int C, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, WCMC_ERR_BAD_ARG,
               "maxpool2_bwd: bad shape (H,W must be even)");
  WCMC_REQUIRE(VIEW_OK(x, xsn, xsh, xsw, C) && VIEW_OK(dy, dsn, dsh, dsw, C) && VIEW_OK(dx, gsn, gsh, gsw, C),
               WCMC_ERR_ALIGNMENT, "maxpool2_bwd: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * C4)), dim3(256), 0,
                     (hipStream_t)stream, View{x, xsn, xsh, xsw}, View{dy, dsn, dsh, dsw}, View{nullptr, 0, 0, 0}, MView{dx, gsn, gsh, gsw}, N,
                     H / 2, W / 2, C4, C);
  return check_launch("maxpool2_bwd");
}

extern "C" int wcmc_maxpool2_bwd_add(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, const float* dy, int64_t dsn, int64_t dsh,
                                     int64_t dsw, const float* add, int64_t asn, int64_t ash, int64_t asw, float* dx, int64_t gsn,
                                     int64_t gsh, int64_t gsw, int N, int H, int W, int C, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, WCMC_ERR_BAD_ARG,
               "maxpool2_bwd_add: bad shape (H,W must be even)");
  WCMC_REQUIRE(VIEW_OK(x, xsn, xsh, xsw, C) && VIEW_OK(dy, dsn, dsh, dsw, C) && VIEW_OK(add, asn, ash, asw, C) && VIEW_OK(dx, gsn, gsh, gsw, C),
               WCMC_ERR_ALIGNMENT, "maxpool2_bwd_add: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * C4)), dim3(256), 0,
                     (hipStream_t)stream, View{x, xsn, xsh, xsw}, View{dy, dsn, dsh, dsw}, View{add, asn, ash, asw}, MView{dx, gsn, gsh, gsw}, N,
                     H / 2, W / 2, C4, C);
  return check_launch("maxpool2_bwd_add");
}

extern "C" int wcmc_upsample2_fwd(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, float* y, int64_t ysn,
                                  int64_t ysh, int64_t ysw, int N, int H, int W, int C, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG, "upsample2_fwd: bad shape");
  WCMC_REQUIRE(VIEW_OK(x, xsn, xsh, xsw, C) && VIEW_OK(y, ysn, ysh, ysw, C), WCMC_ERR_ALIGNMENT,
               "upsample2_fwd: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(grid_for((int64_t)N * 4 * H * W * C4)), dim3(256), 0,
                     (hipStream_t)stream, View{x, xsn, xsh, xsw}, MView{y, ysn, ysh, ysw}, N, H, W, C4, C);
  return check_launch("upsample2_fwd");
}

extern "C" int wcmc_upsample2_bwd(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw, float* dx, int64_t xsn,
                                  int64_t xsh, int64_t xsw, int N, int H, int W, int C, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG, "upsample2_bwd: bad shape");
  WCMC_REQUIRE(VIEW_OK(dy, dsn, dsh, dsw, C) && VIEW_OK(dx, xsn, xsh, xsw, C), WCMC_ERR_ALIGNMENT,
               "upsample2_bwd: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C4)), dim3(256), 0, (hipStream_t)stream,
                     View{dy, dsn, dsh, dsw}, MView{dx, xsn, xsh, xsw}, N, H, W, C4, C);
  return check_launch("upsample2_bwd");
}

extern "C" int wcmc_spp_reduce(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, float* y, int64_t ysn,
                               int64_t ysh, int64_t ysw, int B, int S, int H, int W, int C, float scale,
                               void* stream) {
  WCMC_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG, "spp_reduce: bad shape");
  WCMC_REQUIRE(VIEW_OK(x, xsn, xsh, xsw, C) && VIEW_OK(y, ysn, ysh, ysw, C), WCMC_ERR_ALIGNMENT,
               "spp_reduce: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(spp_reduce_kernel, dim3(grid_for((int64_t)B * H * W * C4)), dim3(256), 0, (hipStream_t)stream,
                     View{x, xsn, xsh, xsw}, MView{y, ysn, ysh, ysw}, B, S, H, W, C4, C, scale);
  return check_launch("spp_reduce");
}

extern "C" int wcmc_spp_broadcast(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, float* y, int64_t ysn,
                                  int64_t ysh, int64_t ysw, int B, int S, int H, int W, int C, float scale,
                                  int accumulate, void* stream) {
  WCMC_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG, "spp_broadcast: bad shape");
  WCMC_REQUIRE(VIEW_OK(x, xsn, xsh, xsw, C) && VIEW_OK(y, ysn, ysh, ysw, C), WCMC_ERR_ALIGNMENT,
               "spp_broadcast: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(spp_broadcast_kernel, dim3(grid_for((int64_t)B * H * W * C4)), dim3(256), 0, (hipStream_t)stream,
                     View{x, xsn, xsh, xsw}, MView{y, ysn, ysh, ysw}, B, S, H, W, C4, C, scale, accumulate);
  return check_launch("spp_broadcast");
}

// ---- per-sample feature assembly of the sample-based denoisers (SBMCInterface / LBMCInterface,
// support/interfaces.py:394-403 and :797-806): features' = cat([features, P, repeat_S(var_S(P).mean_c / S)], 2).
// out is contiguous (B, S, C + Cp + 1, H, W); one thread per (b, y, x): consecutive lanes = consecutive x in every
// (s, c) plane, so all loads and stores are coalesced rows.
namespace wcmc {
__global__ __launch_bounds__(256) void sample_cat_kernel(const float* __restrict__ f, int64_t fsb, int64_t fss, int64_t fsc,
                                                         int64_t fsh, int64_t fsw, const float* __restrict__ p, int64_t psb,
                                                         int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                                                         float* __restrict__ out, int B, int S, int C, int Cp, int H, int W) {
  const int64_t total = (int64_t)B * H * W;
  const int CT = C + Cp + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int b = (int)(i / ((int64_t)W * H));
    const float* pb = p + b * psb + y * psh + x * psw;
    const float* fb = f + b * fsb + y * fsh + x * fsw;
    float* ob = out + (((int64_t)b * S * CT) * H + y) * W + x;
    const int64_t plane = (int64_t)H * W;
    float acc = 0.f;
    for (int c = 0; c < Cp; ++c) {                       // unbiased variance over the samples, two passes (torch.var)
      float m = 0.f;
      for (int s = 0; s < S; ++s) m += pb[s * pss + c * psc];
      m /= (float)S;
      float v = 0.f;
      for (int s = 0; s < S; ++s) { const float d = pb[s * pss + c * psc] - m; v += d * d; }
      acc += v / (float)(S - 1);
    }
    const float pvar = acc / (float)Cp / (float)S;
    for (int s = 0; s < S; ++s) {
      float* o = ob + (int64_t)s * CT * plane;
      for (int c = 0; c < C; ++c) o[c * plane] = fb[s * fss + c * fsc];
      for (int c = 0; c < Cp; ++c) o[(C + c) * plane] = pb[s * pss + c * psc];
      o[(C + Cp) * plane] = pvar;
    }
  }
}
}  // namespace wcmc

extern "C" int wcmc_sample_cat_fwd(const float* feat, int64_t fsb, int64_t fss, int64_t fsc, int64_t fsh, int64_t fsw,
                                   const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                                   float* out, int B, int S, int C, int Cp, int H, int W, void* stream) {
  WCMC_REQUIRE(feat && p && out && B > 0 && S > 1 && C > 0 && Cp > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "sample_cat_fwd: bad argument (S must be >= 2 for the unbiased variance)");
  hipLaunchKernelGGL(wcmc::sample_cat_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, feat,
                     fsb, fss, fsc, fsh, fsw, p, psb, pss, psc, psh, psw, out, B, S, C, Cp, H, W);
  return check_launch("sample_cat_fwd");
}

extern "C" int wcmc_pbuffer_cat_fwd(const float* base, int64_t bsn, int64_t bsc, int64_t bsh, int64_t bsw,
                                    const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                                    float* out, int64_t osn, int64_t osh, int64_t osw, int B, int S, int Cb, int Cp,
                                    int H, int W, void* stream) {
  WCMC_REQUIRE(base && p && B > 0 && S > 1 && Cb > 0 && Cp > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "pbuffer_cat_fwd: bad argument (S must be >= 2 for the unbiased variance)");
  const int CT = Cb + Cp + 1;
  WCMC_REQUIRE(VIEW_OK(out, osn, osh, osw, CT), WCMC_ERR_ALIGNMENT, "pbuffer_cat_fwd: out violates the NHWC-view contract");
  WCMC_REQUIRE((int64_t)B * H <= 65535, WCMC_ERR_BAD_ARG, "pbuffer_cat_fwd: B*H > 65535");
  const size_t lds = (size_t)64 * (round_up(CT, 4) + 1) * sizeof(float);
  hipLaunchKernelGGL(pbuffer_cat_fwd_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)(B * H)), dim3(256), lds,
                     (hipStream_t)stream, base, bsn, bsc, bsh, bsw, p, psb, pss, psc, psh, psw, out, osn, osh, osw, S,
                     Cb, Cp, H, W);
  return check_launch("pbuffer_cat_fwd");
}

extern "C" int wcmc_pbuffer_cat_bwd(const float* g, int64_t gsn, int64_t gsh, int64_t gsw, float* dp, int64_t psb,
                                    int64_t pss, int64_t psc, int64_t psh, int64_t psw, int B, int S, int Cb, int Cp,
                                    int H, int W, void* stream) {
  WCMC_REQUIRE(g && dp && B > 0 && S > 0 && Cb > 0 && Cp > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "pbuffer_cat_bwd: bad argument");
  hipLaunchKernelGGL(pbuffer_cat_bwd_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, g,
                     gsn, gsh, gsw, dp, psb, pss, psc, psh, psw, B, S, Cb, Cp, H, W);
  return check_launch("pbuffer_cat_bwd");
}

extern "C" int wcmc_recombine_fwd(const float* albedo, int64_t asn, int64_t asc, int64_t ash, int64_t asw,
                                  const float* r_diffuse, int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw,
                                  const float* r_specular, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw,
                                  float* out, int N, int C, int H, int W, void* stream) {
  WCMC_REQUIRE(albedo && r_diffuse && r_specular && out && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "recombine_fwd: bad argument");
  const int64_t total = (int64_t)N * C * H * W;
  hipLaunchKernelGGL(recombine_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, albedo,
                     S4{asn, asc, ash, asw}, r_diffuse, S4{dsn, dsc, dsh, dsw}, r_specular, S4{ssn, ssc, ssh, ssw}, out,
                     C, H, W, total);
  return check_launch("recombine_fwd");
}

extern "C" size_t wcmc_image_loss_workspace_bytes(void) { return (size_t)2 * IL_BLOCKS * sizeof(float); }

extern "C" int wcmc_image_loss_fwd(const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw, const float* ref,
                                   int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, float eps, float* l1_mean,
                                   float* relative_mse, void* workspace, size_t workspace_bytes, int N, int C, int H, int W,
                                   void* stream) {
  WCMC_REQUIRE(x && ref && (l1_mean || relative_mse) && workspace && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "image_loss_fwd: bad argument");
  WCMC_REQUIRE(workspace_bytes >= wcmc_image_loss_workspace_bytes(), WCMC_ERR_WORKSPACE, "image_loss_fwd: workspace too small");
  const int64_t total = (int64_t)N * C * H * W;
  const int64_t want = ceil_div64(total, 256);
  const int blocks = (int)(want < IL_BLOCKS ? want : IL_BLOCKS);
  hipLaunchKernelGGL(image_loss_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     S4{xsn, xsc, xsh, xsw}, ref, S4{rsn, rsc, rsh, rsw}, eps, (float*)workspace, C, H, W, total);
  hipLaunchKernelGGL(image_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, blocks,
                     (float)(1.0 / (double)total), l1_mean, relative_mse);
  return check_launch("image_loss_fwd");
}

extern "C" int wcmc_l1_mean_bwd(const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw, const float* ref,
                                int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, const float* grad_loss, float* dx, int N,
                                int C, int H, int W, void* stream) {
  WCMC_REQUIRE(x && ref && grad_loss && dx && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG, "l1_mean_bwd: bad argument");
  const int64_t total = (int64_t)N * C * H * W;
  hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, S4{xsn, xsc, xsh, xsw},
                     ref, S4{rsn, rsc, rsh, rsw}, grad_loss, (float)(1.0 / (double)total), dx, C, H, W, total);
  return check_launch("l1_mean_bwd");
}

extern "C" int wcmc_recombine_bwd(const float* grad_out, const float* albedo, int64_t asn, int64_t asc, int64_t ash,
                                  int64_t asw, const float* r_specular, int64_t ssn, int64_t ssc, int64_t ssh,
                                  int64_t ssw, float* d_diffuse, float* d_specular, int N, int C, int H, int W,
                                  void* stream) {
  WCMC_REQUIRE(grad_out && albedo && r_specular && d_diffuse && d_specular && N > 0 && C > 0 && H > 0 && W > 0,
               WCMC_ERR_BAD_ARG, "recombine_bwd: bad argument");
  const int64_t total = (int64_t)N * C * H * W;
  hipLaunchKernelGGL(recombine_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, grad_out, albedo,
                     S4{asn, asc, ash, asw}, r_specular, S4{ssn, ssc, ssh, ssw}, d_diffuse, d_specular, C, H, W, total);
  return check_launch("recombine_bwd");
}

extern "C" int wcmc_image_loss2_fwd(int kind, const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw, const float* ref,
                                    int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, float eps, float* loss, void* workspace,
                                    size_t workspace_bytes, int N, int C, int H, int W, void* stream) {
  WCMC_REQUIRE(kind >= 0 && kind <= 2 && x && ref && loss && workspace && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "image_loss2_fwd: bad argument");
  WCMC_REQUIRE(workspace_bytes >= wcmc_image_loss_workspace_bytes(), WCMC_ERR_WORKSPACE, "image_loss2_fwd: workspace too small");
  const int64_t total = (int64_t)N * C * H * W;
  const int64_t want = ceil_div64(total, 256);
  const int blocks = (int)(want < IL_BLOCKS ? want : IL_BLOCKS);
  const S4 sx{xsn, xsc, xsh, xsw}, sr{rsn, rsc, rsh, rsw};
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(image_loss2_partial_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, x, sx, ref, sr, eps, (float*)workspace, C, H, W, total);
  else if (kind == 1) hipLaunchKernelGGL(image_loss2_partial_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, sx, ref, sr, eps, (float*)workspace, C, H, W, total);
  else hipLaunchKernelGGL(image_loss2_partial_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, x, sx, ref, sr, eps, (float*)workspace, C, H, W, total);
  hipLaunchKernelGGL(image_loss2_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, blocks,
                     (float)((kind == 0 ? 1.0 : 0.5) / (double)total), loss);
  return check_launch("image_loss2_fwd");
}

extern "C" int wcmc_image_loss2_bwd(int kind, const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw, const float* ref,
                                    int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, float eps, const float* grad_loss, float* dx,
                                    int N, int C, int H, int W, void* stream) {
  WCMC_REQUIRE(kind >= 0 && kind <= 2 && x && ref && grad_loss && dx && N > 0 && C > 0 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "image_loss2_bwd: bad argument");
  const int64_t total = (int64_t)N * C * H * W;
  const S4 sx{xsn, xsc, xsh, xsw}, sr{rsn, rsc, rsh, rsw};
  hipStream_t st = (hipStream_t)stream;
  const float inv = (float)(1.0 / (double)total);
  if (kind == 0) hipLaunchKernelGGL(image_loss2_bwd_kernel<0>, dim3(grid_for(total)), dim3(256), 0, st, x, sx, ref, sr, eps, grad_loss, inv, dx, C, H, W, total);
  else if (kind == 1) hipLaunchKernelGGL(image_loss2_bwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, x, sx, ref, sr, eps, grad_loss, inv, dx, C, H, W, total);
  else hipLaunchKernelGGL(image_loss2_bwd_kernel<2>, dim3(grid_for(total)), dim3(256), 0, st, x, sx, ref, sr, eps, grad_loss, inv, dx, C, H, W, total);
  return check_launch("image_loss2_bwd");
}

extern "C" size_t wcmc_grad_norm_clip_workspace_bytes(int n_tensors, const int64_t* numel) {
  size_t blocks = 0;
  for (int i = 0; i < n_tensors; ++i) blocks += (size_t)ceil_div64(numel[i] > 0 ? numel[i] : 1, GN_CHUNK);
  return (blocks + 4) * sizeof(float);
}

extern "C" int wcmc_grad_norm_clip(int n_tensors, float* const* grads, const int64_t* numel, float max_norm, float* norm_and_coef,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  WCMC_REQUIRE(n_tensors > 0 && n_tensors <= GN_MAX && grads && numel && norm_and_coef && workspace && max_norm > 0.f, WCMC_ERR_BAD_ARG,
               "grad_norm_clip: bad argument (1..%d tensors)", GN_MAX);
  WCMC_REQUIRE(workspace_bytes >= wcmc_grad_norm_clip_workspace_bytes(n_tensors, numel), WCMC_ERR_WORKSPACE, "grad_norm_clip: workspace too small");
  GNTable t;
  t.n = n_tensors;
  unsigned blocks = 0;
  for (int i = 0; i < n_tensors; ++i) {
    WCMC_REQUIRE(grads[i] && numel[i] > 0, WCMC_ERR_BAD_ARG, "grad_norm_clip: bad tensor %d", i);
    t.e[i].g = grads[i]; t.e[i].n = numel[i]; t.e[i].block0 = blocks;
    blocks += (unsigned)ceil_div64(numel[i], GN_CHUNK);
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, st, t, (float*)workspace);
  hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)blocks, max_norm, norm_and_coef);
  hipLaunchKernelGGL(grad_scale_kernel, dim3(blocks), dim3(256), 0, st, t, (const float*)norm_and_coef);
  return check_launch("grad_norm_clip");
}
