// Split-bf16 ("bf16x3") convolution forward / data-gradient / weight-gradient for gfx950.
//
// Same GEMM views as conv.hip, but every fp32 operand x is carried as two bf16 planes
//   hi = bf16(x), lo = bf16(x - hi)          (x = hi + lo to ~2^-17 relative)
// and a product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation (the dropped lo*lo term is ~2^-18 relative).  bf16 x bf16 products are exact in
// fp32, so the only roundings are the operand split and the fp32 accumulate: 3 MFMAs at 16x the fp32
// MFMA rate = 5.3x the fp32-MFMA throughput at close to fp32 accuracy (measured in
// tests/test_gpu_ops.py).  Replaces the same reference expressions as conv.hip (torch.nn.Conv2d
// inside sbmc.modules.ConvChain; cuDNN with TF32 on the reference's hardware).
//
// Split tensor layout (chain-internal, dense): u16 [N][H][W][2][Cp], Cp = round_up(C, 8);
// plane 0 = hi, plane 1 = lo; pad channels are ZERO (producers guarantee it), so loaders need no
// channel masks.  Packed weights: u16 wp[Np][2][Kt], k = tap*Kp + c, Kp = round_up(kchan, 8),
// Kt = round_up(taps*Kp, 32), Np = round_up(rows, 16).
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace wcmc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 f2bf(float x) { return __builtin_bit_cast(u16, (__bf16)x); }
__device__ __forceinline__ float bf2f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ void split1(float x, u16& hi, u16& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// ------------------------------------------------------------------ fp32 NHWC view -> split
// Optional gate (post != nullptr): out = split(x * act'(post)), the output-activation backward of a chain
// (wcmc_act_backward) folded into the split of its upstream gradient -- one pass over dy instead of two.
__global__ void split_kernel(const float* __restrict__ x, int64_t xsn, int64_t xsh, int64_t xsw,
                             u16* __restrict__ out, int H, int W, int C, int Cp, int64_t total,
                             const float* __restrict__ post = nullptr, int64_t psn = 0, int64_t psh = 0, int64_t psw = 0,
                             int act = 0, float slope = 0.f) {
  const int V = Cp / 8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const NhwvIndex ix_ = decode_nhwv(idx, total, H, W, V);
    const int v = ix_.v, xx = ix_.x, y = ix_.y, n = ix_.n;
    const float* src = x + n * xsn + y * xsh + xx * xsw + v * 8;
    float f[8];
    const int c0 = v * 8;
    if (c0 + 8 <= C) {
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (c0 + e < C) ? src[e] : 0.f;
    }
    if (post) {
      const float* ps = post + n * psn + y * psh + xx * psw + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < C) f[e] *= act_gate(ps[e], act, slope);
    }
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1(f[e], hi[e], lo[e]);
    u16* o = out + (((int64_t)n * H + y) * W + xx) * 2 * Cp + c0;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
    *reinterpret_cast<uint4*>(o + Cp) = *reinterpret_cast<const uint4*>(lo);
  }
}

// ------------------------------------------------------------------ split -> one fp16 plane
// x = hi + lo of a split tensor, rounded once to fp16 (11 bits; saturating at +-65504): the A operand of the one-MFMA fp16 forward
// of an un-gated output layer ("bf16x321h" mode).  [N*H*W][Cp] halfs, Cp = round_up(C, 8).
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u16 f2h_sat(float v) {
  v = v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v);
  return __builtin_bit_cast(u16, (_Float16)v);
}
__global__ __launch_bounds__(256) void split_to_f16_kernel(const u16* __restrict__ xs, u16* __restrict__ out, int Cp, int64_t total) {
  const int V = Cp / 8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(idx % V); const int64_t px = idx / V;
    const uint4 h = *reinterpret_cast<const uint4*>(xs + px * 2 * Cp + v * 8), l = *reinterpret_cast<const uint4*>(xs + px * 2 * Cp + Cp + v * 8);
    const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
    u16 o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw[e] << 16);
      const float b = __builtin_bit_cast(float, hw[e] & 0xffff0000u) + __builtin_bit_cast(float, lw[e] & 0xffff0000u);
      o[2 * e] = f2h_sat(a); o[2 * e + 1] = f2h_sat(b);
    }
    *reinterpret_cast<uint4*>(out + px * Cp + v * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// ------------------------------------------------------------------ strided (N,C,H,W) -> split
// The per-sample path descriptors arrive channel-first (`paths` (B,S,36,H,W), support/networks.py:31-33) and are only
// ever read as the embedding chain's split input: transpose and split in one pass (64 pixels of one image row x all
// channels through LDS) instead of a channel-last fp32 copy that is then split (0.3 GB less traffic per step).
__global__ __launch_bounds__(256) void nchw_split_kernel(const float* __restrict__ src, int64_t ssn, int64_t ssc, int64_t ssh,
                                                         int64_t ssw, u16* __restrict__ out, int C, int Cp, int H, int W) {
  __shared__ float tile[64][65];                    // [channel][pixel]
  const int xt = blockIdx.x * 64, y = blockIdx.y % H, n = blockIdx.y / H;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int c = grp; c < Cp; c += 4) {
    const int x = xt + lane;
    tile[c][lane] = (c < C && x < W) ? src[(int64_t)n * ssn + (int64_t)c * ssc + (int64_t)y * ssh + (int64_t)x * ssw] : 0.f;
  }
  __syncthreads();
  const int V = Cp / 8;
  for (int i = threadIdx.x; i < 64 * V; i += 256) {
    const int px = i / V, v = i - px * V, x = xt + px;
    if (x >= W) continue;
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1(tile[v * 8 + e][px], hi[e], lo[e]);
    u16* o = out + (((int64_t)n * H + y) * W + x) * 2 * Cp + v * 8;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
    *reinterpret_cast<uint4*>(o + Cp) = *reinterpret_cast<const uint4*>(lo);
  }
}

// ------------------------------------------------------------------ cat([flat, repeat_S(prop)], 1) -> split
// support/networks.py:39-40 feeding the `final` ConvChain: the 128-channel concatenation of the per-sample
// embedding (B*S images) and the spp-broadcast U-Net output (B images) is written once, directly as the
// chain's split-bf16 input (separately: copy 268 MB + broadcast 268 MB into an fp32 tensor, then read its
// 537 MB and write 537 MB of split planes).  One thread = 8 channels of one pixel.
// up != 0 (U-Net skip concatenation, Autoencoder of sbmc.modules): `flat` is the level below at (H/2, W/2) and is
// upsampled on the fly -- bilinear x2, align_corners=False, the same four taps and the same fma chain as
// upsample2_fwd_kernel (elementwise.hip), so the result equals upsample + concatenation bit for bit.
__global__ void cat_broadcast_split_kernel(const float* __restrict__ flat, int64_t fsn, int64_t fsh, int64_t fsw,
                                           const float* __restrict__ prop, int64_t psn, int64_t psh, int64_t psw,
                                           u16* __restrict__ out, int S, int H, int W, int C1, int C2, int Cp,
                                           int64_t total, int up = 0) {
  const int V = Cp / 8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const NhwvIndex ix_ = decode_nhwv(idx, total, H, W, V);
    const int v = ix_.v, xx = ix_.x, y = ix_.y, n = ix_.n;
    const int c0 = v * 8;
    float f[8];
    if (c0 < C1 && up) {
      const int hh = H >> 1, wh = W >> 1, iy = y >> 1, ix = xx >> 1;
      const int ny = (y & 1) ? min(iy + 1, hh - 1) : max(iy - 1, 0);
      const int nx = (xx & 1) ? min(ix + 1, wh - 1) : max(ix - 1, 0);
      const float* b0 = flat + n * fsn + c0;
      const float* p00 = b0 + iy * fsh + ix * fsw;
      const float* p01 = b0 + iy * fsh + nx * fsw;
      const float* p10 = b0 + ny * fsh + ix * fsw;
      const float* p11 = b0 + ny * fsh + nx * fsw;
      __attribute__((aligned(16))) float t00[8], t01[8], t10[8], t11[8];      // (two 16-byte loads per tap; the same fma chain per element)
      *reinterpret_cast<float4*>(t00) = *reinterpret_cast<const float4*>(p00); *reinterpret_cast<float4*>(t00 + 4) = *reinterpret_cast<const float4*>(p00 + 4);
      *reinterpret_cast<float4*>(t01) = *reinterpret_cast<const float4*>(p01); *reinterpret_cast<float4*>(t01 + 4) = *reinterpret_cast<const float4*>(p01 + 4);
      *reinterpret_cast<float4*>(t10) = *reinterpret_cast<const float4*>(p10); *reinterpret_cast<float4*>(t10 + 4) = *reinterpret_cast<const float4*>(p10 + 4);
      *reinterpret_cast<float4*>(t11) = *reinterpret_cast<const float4*>(p11); *reinterpret_cast<float4*>(t11 + 4) = *reinterpret_cast<const float4*>(p11 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float r = 0.5625f * t00[e];
        r = fmaf(0.1875f, t01[e], r);
        r = fmaf(0.1875f, t10[e], r);
        r = fmaf(0.0625f, t11[e], r);
        f[e] = r;
      }
    } else if (c0 < C1) {                            // C1 % 8 == 0: a vector never straddles the two sources
      const float* src = flat + n * fsn + y * fsh + xx * fsw + c0;
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
      const float* src = prop + (n / S) * psn + y * psh + xx * psw + (c0 - C1);
      if (c0 - C1 + 8 <= C2) {
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (c0 - C1 + e < C2) ? src[e] : 0.f;
      }
    }
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1(f[e], hi[e], lo[e]);
    u16* o = out + (((int64_t)n * H + y) * W + xx) * 2 * Cp + c0;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
    *reinterpret_cast<uint4*>(o + Cp) = *reinterpret_cast<const uint4*>(lo);
  }
}

// ------------------------------------------------------------------ split(g + repeat_S(gm) * scale)
// Gradient of the embedding chain's output (support/networks.py:35-40): the per-sample gradient from the
// concatenation plus the spp-broadcast gradient of the mean, written once as the split dy of the chain's
// backward (separately: broadcast into 268 MB, an elementwise add over 3 x 268 MB, then the split pass).
// g may be null (only the mean path carries gradient).  One thread = 8 channels of one pixel.
__global__ void add_broadcast_split_kernel(const float* __restrict__ g, int64_t gsn, int64_t gsh, int64_t gsw,
                                           const float* __restrict__ gm, int64_t msn, int64_t msh, int64_t msw,
                                           float scale, u16* __restrict__ out, int S, int H, int W, int C, int Cp,
                                           int64_t total) {
  const int V = Cp / 8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const NhwvIndex ix_ = decode_nhwv(idx, total, H, W, V);
    const int v = ix_.v, xx = ix_.x, y = ix_.y, n = ix_.n;
    const int c0 = v * 8;
    float f[8];
    if (c0 + 8 <= C) {                               // whole vector: two 16-byte loads per operand
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
      if (g) {
        const float* q = g + n * gsn + y * gsh + xx * gsw + c0;
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
      }
      if (gm) {
        const float* q = gm + (n / S) * msn + y * msh + xx * msw + c0;
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        f[0] += a.x * scale; f[1] += a.y * scale; f[2] += a.z * scale; f[3] += a.w * scale;
        f[4] += b.x * scale; f[5] += b.y * scale; f[6] += b.z * scale; f[7] += b.w * scale;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = 0.f;
        if (c0 + e < C) {
          if (g) a = g[n * gsn + y * gsh + xx * gsw + c0 + e];
          if (gm) a += gm[(n / S) * msn + y * msh + xx * msw + c0 + e] * scale;
        }
        f[e] = a;
      }
    }
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1(f[e], hi[e], lo[e]);
    u16* o = out + (((int64_t)n * H + y) * W + xx) * 2 * Cp + c0;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
    *reinterpret_cast<uint4*>(o + Cp) = *reinterpret_cast<const uint4*>(lo);
  }
}

// ------------------------------------------------------------------ the gradient entering a chain's backward -> split + column sums
// out = split((dy [+ repeat_S(gm) * scale]) [* act'(post)]) -- what split_kernel / add_broadcast_split_kernel produce for the
// LAST layer of a chain -- and, in the same pass, the per-block column sums of the result in the layout the GEMM epilogues
// leave ([row][Np] + trailer word = rows written): the layer's bias gradient is then finished by its weight gradient's
// slab-reduction launch (wcmc_conv2d_wgrad_bf16x3, dy_colsum_partial) instead of a column-sum pass that re-reads the split
// tensor (268 MB for a PathNet embedding) plus a finish launch.  One thread = one 8-channel vector of every PL-th pixel of
// the block's range; fixed-order LDS tree over the pixel lanes: bitwise reproducible.
__global__ __launch_bounds__(256) void split_dy_colsum_kernel(const float* __restrict__ dy, int64_t dsn, int64_t dsh, int64_t dsw,
                                                              const float* __restrict__ post, int64_t psn, int64_t psh, int64_t psw,
                                                              int act, float slope, const float* __restrict__ gm, int64_t msn,
                                                              int64_t msh, int64_t msw, int S, float scale, u16* __restrict__ out,
                                                              int H, int W, int C, int Cp, int64_t M, int64_t per_block,
                                                              float* __restrict__ partial, int Np, int Gmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [PL][V][8]
  const int V = Cp / 8, PL = 256 / V;
  const int v = threadIdx.x % V, pl = threadIdx.x / V;
  const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = min(M, p0 + per_block);
  const int c0 = v * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < PL) {
    // pixel cursor (n, y, xx) of q, advanced by PL per iteration (the first version divided the 64-bit pixel index three
    // times per pixel: the kernel ran at 1.7 TB/s on its address arithmetic)
    int xx, y, n;
    { const int64_t q0 = p0 + pl; xx = (int)(q0 % W); const int64_t t = q0 / W; y = (int)(t % H); n = (int)(t / H); }
    for (int64_t q = p0 + pl; q < p1; q += PL, xx += PL) {
      while (xx >= W) { xx -= W; if (++y == H) { y = 0; ++n; } }
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
      const bool whole = c0 + 8 <= C;
      if (dy) {
        const float* src = dy + n * dsn + y * dsh + xx * dsw + c0;
        if (whole) {
          const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
          f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (c0 + e < C) f[e] = src[e];
        }
      }
      if (gm) {
        const float* src = gm + (n / S) * msn + y * msh + xx * msw + c0;
        if (whole) {
          const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
          f[0] += a.x * scale; f[1] += a.y * scale; f[2] += a.z * scale; f[3] += a.w * scale;
          f[4] += b.x * scale; f[5] += b.y * scale; f[6] += b.z * scale; f[7] += b.w * scale;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (c0 + e < C) f[e] += src[e] * scale;
        }
      }
      if (post) {
        const float* ps = post + n * psn + y * psh + xx * psw + c0;
        if (whole) {
          const float4 a = *reinterpret_cast<const float4*>(ps), b = *reinterpret_cast<const float4*>(ps + 4);
          const float g[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= act_gate(g[e], act, slope);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < C) f[e] *= act_gate(ps[e], act, slope);
        }
      }
      u16 hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        split1(f[e], hi[e], lo[e]);
        acc[e] += bf2f(hi[e]) + bf2f(lo[e]);          // (the sum of what the GEMMs will see, as colsum_split_kernel)
      }
      u16* o = out + q * 2 * Cp + c0;
      *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
      *reinterpret_cast<uint4*>(o + Cp) = *reinterpret_cast<const uint4*>(lo);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) smem[(pl * V + v) * 8 + e] = acc[e];
  }
  __syncthreads();
  if (pl == 0) {
    for (int q = 1; q < PL; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += smem[(q * V + v) * 8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < Np) partial[(int64_t)blockIdx.x * Np + c0 + e] = (c0 + e < C) ? acc[e] : 0.f;
  }
  // (Np - Cp can be 8: the columns past the last vector)
  if (threadIdx.x < Np - Cp) partial[(int64_t)blockIdx.x * Np + Cp + threadIdx.x] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(partial)[(int64_t)Gmax * Np] = (int)gridDim.x;
}

// K order of the packed weights: k = slab*Ks + tap*cs + cl, channel = slab*CS + cl (cs = CS, or CSl in the last slab).  The streaming
// kernel uses one slab of all (padded) channels (CS = Kp, Ks = Kt); the halo kernel cuts the channels into
// slabs of CS <= 64 that fit in LDS with their halo (x_plan_k below decides, from (kchan, ks) alone).
__global__ void pack_weight_split_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int Cin,
                                         int ks, int mode, int rows, int Np, int CS, int Ks, int Kt, int nslabs, int CSl, int f16) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Np * Kt) return;
  const int n = (int)(idx / Kt), k = (int)(idx - (int64_t)n * Kt);
  int slab = k / Ks;
  if (slab > nslabs - 1) slab = nslabs - 1;                 // (the last slab's Ks may be the smaller one)
  const int kk = k - slab * Ks;
  const int cs = slab == nslabs - 1 ? CSl : CS;
  const int tap = kk / cs, cl = kk - tap * cs;
  const int c = slab * CS + cl;
  const int taps = ks * ks;
  const int kchan = mode == 0 ? Cin : Cout;
  float v = 0.f;
  if (n < rows && tap < taps && c < kchan) {
    if (mode == 0) v = w[((int64_t)n * Cin + c) * taps + tap];
    else           v = w[((int64_t)c * Cin + n) * taps + (taps - 1 - tap)];
  }
  u16 hi, lo;
  split1(v, hi, lo);
  if (f16) { hi = f2h_sat(v); lo = 0; }          // (mode 4: ONE fp16 plane in the hi rows; the lo rows are never read)
  wp[((int64_t)n * 2) * Kt + k] = hi;
  wp[((int64_t)n * 2 + 1) * Kt + k] = lo;
}

static int x_env_on(const char* name) {           // switch is ON unless the variable starts with '0' (debug build only: ab_env)
  const char* e = ab_env(name);
  return (e && e[0] == '0') ? 0 : 1;
}
struct XKPlan { bool halo; int Kp, CS, nslabs, Ks, Kt, PXS, CSl, Ksl, ap; };     // CSl / Ksl: the last (narrower) slab
static int x_pick_nt(int tiles);
// conv_halo64_bf16x3_kernel, last slab of at most FOUR real channels (KPCN's 100 = 6 x 16 + 4 = 3 x 32 + 4): its k order is four
// channels per tap, EIGHT taps per 32-k stage -- 4 stages for the 25 taps instead of the 7 of an 8-channel slab (four taps per stage,
// half of every k-group zeros): 85 -> 82 stages in the forward, 82 -> 79 in the data gradient (-3.6 % MFMAs).  CSl = 4 is the K ORDER
// only: the halo still holds the slab as 8-channel units (one 16-byte DMA granule per plane); a lane's k-group is two 8-byte reads
// from two neighbouring taps.  (WCMC_HALO64_L4=0, debug build: the 8-channel order.)
static bool x_last4(int kchan, int CS, int nslabs) {
  const int left = kchan - (nslabs - 1) * CS;
  return nslabs >= 2 && left >= 1 && left <= 4 && x_env_on("WCMC_HALO64_L4");
}
// ap_req: planes of the A operand (the pixels) the caller wants multiplied -- 2 = hi + lo (three MFMAs per product), 1 = hi only
// (two: W_lo*A_hi + W_hi*A_hi; the data gradient of the "bf16x321" mode).  q.ap is what the plan grants: 1 only where a
// kernel instance for it exists (rows = the GEMM's output channels pick the instance), else the three-term plan.  The K
// order of the packed weights follows the plan, so packing and launch must ask with the same (kchan, ks, ap_req, rows).
static XKPlan x_plan_k(int kchan, int ks, int ap_req = 2, int rows = 0) {
  // (debug build: A/B switches are read per call, so that a script can flip them in one process)
  const int enable = x_env_on("WCMC_IGEMM_HALO");
  XKPlan q;
  q.ap = 2;
  q.Kp = round_up(kchan, 8);
  q.halo = enable && ks >= 3 && ks <= 5 && q.Kp >= 32;
  // (rows: seven cout tiles per block -- the KPCN layers -- or ONE: the first layer's data gradient restricted to the 8 input
  // channels whose gradient is read, ops.conv_chain)
  const int nt_rows = rows > 0 ? x_pick_nt(round_up(rows, 16) / 16) : 0;
  if (ap_req == 1 && q.halo && ks == 5 && x_env_on("WCMC_HALO64") && (nt_rows == 7 || nt_rows == 1) &&
      q.Kp % 32 != 24 && x_env_on("WCMC_DGRAD_AP1")) {
    // conv_halo64_bf16x3_kernel<7, 3, PT, 0, 80, 1>: the halo holds the hi plane only, so a pixel's 80 bytes carry 32 channels
    // instead of 16 -- half the slabs (104 channels = 32 + 32 + 32 + 8: K = 3 x 800 + 224 = 2624 of 2500 useful, three halo
    // reloads per tile instead of six), one tap per 32-k stage (four in the 8-channel slab)
    q.ap = 1;
    q.nslabs = (q.Kp + 31) / 32;
    q.CS = 32; q.CSl = q.Kp - (q.nslabs - 1) * 32;            // 8, 16 or 32
    if (x_last4(kchan, q.CS, q.nslabs)) q.CSl = 4;            // (see x_last4)
    q.PXS = 80;
    q.Ks = round_up(ks * ks * q.CS, 32); q.Ksl = round_up(ks * ks * q.CSl, 32);
    q.Kt = (q.nslabs - 1) * q.Ks + q.Ksl;
    return q;
  }
  if (ap_req == 1 && q.halo && ks == 3 && rows > 0 && x_pick_nt(round_up(rows, 16) / 16) >= 4 && x_env_on("WCMC_DGRAD_AP1")) {
    // conv_halo_bf16x3_kernel<4 | 7, .., AP = 1> (the U-Net's 3x3 data gradients): hi plane only, so a slab holds up to 128 channels
    // in the strides the two-plane plan uses for 64 -- half the halo reloads (and a 224-byte halo for the 64-channel layers)
    q.ap = 1;
    q.nslabs = (q.Kp + 127) / 128;
    q.CS = round_up((q.Kp + q.nslabs - 1) / q.nslabs, 8);
    q.PXS = q.CS <= 112 ? 224 : 288;          // 16 B x (14 or 2 mod 16): conflict-free b128 reads, as below
    q.Ks = round_up(ks * ks * q.CS, 32);
    q.CSl = q.Kp - (q.nslabs - 1) * q.CS;
    if (q.CSl < 32) q.CSl = q.CS;
    q.Ksl = round_up(ks * ks * q.CSl, 32);
    q.Kt = (q.nslabs - 1) * q.Ks + q.Ksl;
    return q;
  }
  if (q.halo && ks == 5 && x_env_on("WCMC_HALO64")) {
    // conv_halo64_bf16x3_kernel: slabs of 16 channels (the last one 8 or 16), halo pixel stride 80 B, two taps per stage
    // (5x5 only: on the U-Net's 3x3 layers it wins 4 % at 128^2 and loses 30-70 % on the 64^2 / 32^2 levels, whose 16x16
    // tilings leave most CUs with one workgroup -- scripts/time_unet_layers.py)
    if (q.Kp >= 256 && q.Kp % 32 == 0 && x_env_on("WCMC_HALO64_CS32")) {
      // many input channels (the 441-cout layer's data gradient: 448): 32-channel slabs, one tap per stage -- half the
      // halo reloads and no padded taps (14 x 800 k instead of 28 x 416); the 160-byte halo only fits the 12x16 tile
      // beside a second workgroup, with two weight stages (launch_xhalo64)
      q.nslabs = q.Kp / 32;
      q.CS = q.CSl = 32;
      q.PXS = 160;
      q.Ks = q.Ksl = round_up(ks * ks * 32, 32);
      q.Kt = q.nslabs * q.Ks;
      return q;
    }
    q.nslabs = (q.Kp + 15) / 16;
    q.CS = 16; q.CSl = q.Kp - (q.nslabs - 1) * 16;
    if (x_last4(kchan, q.CS, q.nslabs)) q.CSl = 4;
    // halo pixel stride 80 B (5 slots of 16 B: hi 0-1, lo 2-3, one of pad).  The ds_read_b128 of the pixel fragments are
    // 2-way bank conflicts with it (PMC: 23-26 % of the LDS cycles; the four 16-lane groups of a b128 read take k-groups 0
    // and 1 of different pixel columns together and 5 f, 5 f' + 1 meet mod 16); 96 B is conflict-free for the 16-channel
    // slabs and measured 0.7 % SLOWER (3.120 vs 3.097 ms per branch: the LDS pipe is not what the loop waits for, and the
    // halo grows by a fifth); the switch for that A/B is gone (round 3)
    q.PXS = 80;
    q.Ks = round_up(ks * ks * q.CS, 32); q.Ksl = round_up(ks * ks * q.CSl, 32);
    q.Kt = (q.nslabs - 1) * q.Ks + q.Ksl;
    return q;
  }
  const int th8 = x_env_on("WCMC_HALO_TH8_5X5");   // =0: A/B switch back to 16x16 tiles with 56/48-channel slabs
  if (q.halo && th8 && ks == 5 && (q.Kp % 32 == 0 || q.Kp % 32 == 8)) {
    // slabs of 32 channels, the last one 32 or 40: halo pixel stride 160 B (10 units = 2 mod 4), 38 KB for a 12x20 halo
    q.nslabs = q.Kp / 32;
    q.CS = 32; q.CSl = q.Kp - (q.nslabs - 1) * 32;
    q.PXS = 160;
    q.Ks = round_up(ks * ks * q.CS, 32); q.Ksl = round_up(ks * ks * q.CSl, 32);
    q.Kt = (q.nslabs - 1) * q.Ks + q.Ksl;
    return q;
  }
  if (q.halo) {
    q.nslabs = (q.Kp + 63) / 64;
    q.CS = round_up((q.Kp + q.nslabs - 1) / q.nslabs, 8);
    q.PXS = q.CS <= 56 ? 224 : 288;           // halo pixel stride: 16 B x (2 or 14 mod 16) -> conflict-free b128 reads
    q.Ks = round_up(ks * ks * q.CS, 32);
    // the last slab holds what is left (104 channels = 56 + 48: 1408 + 1216 k instead of 2 x 1408); the lane's
    // tap stepping assumes at most one wrap per 32-k stage, so a slab narrower than 32 channels is padded instead
    q.CSl = q.Kp - (q.nslabs - 1) * q.CS;
    if (q.CSl < 32) q.CSl = q.CS;
    q.Ksl = round_up(ks * ks * q.CSl, 32);
  } else {
    q.nslabs = 1; q.CS = q.Kp; q.PXS = 0;
    q.Ks = round_up(ks * ks * q.Kp, 32);
    q.CSl = q.CS; q.Ksl = q.Ks;
  }
  q.Kt = (q.nslabs - 1) * q.Ks + q.Ksl;
  return q;
}
// All weights of a chain, both orientations, in ONE launch: a table of up to 20 (layer, mode) entries by value; a block
// finds its entry by its block range and runs pack_weight_split_kernel's body on it.  (114 packing launches of ~4 us per
// step become 16.)
constexpr int XPACK_MAX = 32;      // (2.3 KB of kernel arguments: the fifteen U-Net layers of a PathNet, both orientations, in one launch)
struct XPackEntry { const float* w; u16* wp; int Cout, Cin, mode, rows, Np, CS, Ks, Kt, nslabs, CSl; unsigned block0; int f16; };
struct XPackTable { XPackEntry e[XPACK_MAX]; int n, ks; };
__global__ __launch_bounds__(256) void pack_weight_split_multi_kernel(XPackTable t) {
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i)
    if (blockIdx.x >= t.e[i].block0) k = i;
  const XPackEntry& q = t.e[k];
  const int64_t idx = (int64_t)(blockIdx.x - q.block0) * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)q.Np * q.Kt) return;
  const int n = (int)(idx / q.Kt), kk0 = (int)(idx - (int64_t)n * q.Kt);
  int slab = kk0 / q.Ks;
  if (slab > q.nslabs - 1) slab = q.nslabs - 1;
  const int kk = kk0 - slab * q.Ks;
  const int cs = slab == q.nslabs - 1 ? q.CSl : q.CS;
  const int tap = kk / cs, cl = kk - tap * cs;
  const int c = slab * q.CS + cl;
  const int taps = t.ks * t.ks;
  const int kchan = q.mode == 0 ? q.Cin : q.Cout;
  float v = 0.f;
  if (n < q.rows && tap < taps && c < kchan) {
    if (q.mode == 0) v = q.w[((int64_t)n * q.Cin + c) * taps + tap];
    else             v = q.w[((int64_t)c * q.Cin + n) * taps + (taps - 1 - tap)];
  }
  u16 hi, lo;
  split1(v, hi, lo);
  if (q.f16) { hi = f2h_sat(v); lo = 0; }
  q.wp[((int64_t)n * 2) * q.Kt + kk0] = hi;
  q.wp[((int64_t)n * 2 + 1) * q.Kt + kk0] = lo;
}

// rows of the per-tile column-sum buffer: enough for either kernel's tiling of (N, Ho, Wo)
static int x_colsum_rows(int N, int Ho, int Wo) {
  const int64_t gl = ceil_div64((int64_t)N * Ho * Wo, 128);
  const int64_t gh = (int64_t)N * ((Ho + 7) / 8) * ((Wo + 15) / 16);      // 8x16 halo tiles (16x16: fewer)
  return (int)(gl > gh ? gl : gh);
}

// ------------------------------------------------------------------ implicit GEMM (fwd + dgrad)
// PMC profile of the first version (profiles/): 97 % L2 hits, but 3.3 VALU + 1.5 SALU per MFMA, half of
// the LDS cycles bank conflicts, 37 % of wave time parked on vmcnt/barrier.  Hence:
//   * operand loads are buffer loads: out-of-image taps / rows past the tensor use an out-of-range
//     offset and the hardware returns zeros (no branches, no zero-fill moves, 32-bit offsets);
//   * LDS rows are 64 B (one 32-k stage of one plane) with the 16-byte slot XOR-swizzled by
//     (row >> 1) & 3 and the lo plane shifted by 64 B: ds_read_b128 and ds_write_b128 conflict-free;
//   * the register prefetch runs TWO stages ahead of the MFMAs.
constexpr int XBM = 128;   // pixels per block
constexpr int XKC = 32;    // k per LDS stage = one MFMA k-step
constexpr int XROW = 32;   // bf16 per LDS row
constexpr unsigned XOOB = 0x80000000u;   // byte offset beyond any buffer (num_records < 2 GiB)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// bit e = (bf16 number e of the vector is > 0): the activation-derivative predicate of act_gate on a hi plane.
// A bf16 is positive iff it is positive as an int16; per dword (two of them): max(., 0) of both halves in one packed instruction,
// "half != 0" as bit 15 of half + 0x7fff (no carry between the halves: a clamped half is at most 0x7fff) -- four vector
// instructions per pair where the test half by half took ten (the gate masks cost 3 of a U-Net layer's 37 us, profiles/r06_unet_halo3.txt).
typedef short xs16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned positive_pair(unsigned w) {          // bit 0: low half > 0, bit 16: high half > 0
  const xs16x2 z = {0, 0};
  const xs16x2 c = __builtin_elementwise_max(__builtin_bit_cast(xs16x2, w), z);
  return ((__builtin_bit_cast(unsigned, c) + 0x7fff7fffu) >> 15) & 0x00010001u;
}
__device__ __forceinline__ unsigned char positive_mask8(const u32x4 v) {
  // pairs e = 0..3 -> bits 2e (low half) and 2e + 1 (high half)
  const unsigned x = positive_pair(v[0]) | (positive_pair(v[1]) << 2) | (positive_pair(v[2]) << 4) | (positive_pair(v[3]) << 6);
  return (unsigned char)((x | (x >> 15)) & 0xffu);
}

// Epilogue of one accumulator quad (4 consecutive couts of one pixel): bias, activation, pixel validity, gate, split --
// as packed conversions and selects.  Couts past Cout need no test: their packed weights and their bias (out-of-range
// buffer load) are zeros, and every activation maps 0 to +0.
typedef __bf16 xbf16x2 __attribute__((ext_vector_type(2)));
typedef float xf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x_split2(float a, float b, unsigned& hi2, unsigned& lo2) {
  const xf32x2 v = {a, b};
  hi2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, xbf16x2));
  const xf32x2 back = {__builtin_bit_cast(float, hi2 << 16), __builtin_bit_cast(float, hi2 & 0xffff0000u)};
  lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, xbf16x2));
}
// activation as selects (bit-identical to act_apply, no branches inside an unrolled epilogue)
struct XAct { float ns; bool zero; };
__device__ __forceinline__ XAct x_act(int act, float slope) {
  return XAct{act == WCMC_ACT_LEAKY_RELU ? slope : 1.f, act == WCMC_ACT_RELU};
}
__device__ __forceinline__ float x_act_apply(float v, XAct a) { return v > 0.f ? v : (a.zero ? 0.f : v * a.ns); }

// gate kinds of x_epi_quad: 0 none, 1 split gate tensor (hi plane of 4 values in g2), 2 bit mask (byte in g2.x)
__device__ __forceinline__ void x_epi_quad(const f32x4 a4, const float (&b)[4], bool ok, XAct ak, int gkind, u32x2 g2, int co,
                                           float gate_off, float (&v)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t = a4[e] + b[e];
    const float neg = ak.zero ? 0.f : t * ak.ns;
    const float r = t > 0.f ? t : neg;
    v[e] = ok ? r : 0.f;
  }
  if (gkind == 1) {
    const unsigned g[4] = {g2.x << 16, g2.x & 0xffff0000u, g2.y << 16, g2.y & 0xffff0000u};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, g[e]) > 0.f ? v[e] : v[e] * gate_off;
  } else if (gkind == 2) {
    const unsigned bits = g2.x >> (co & 7);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ((bits >> e) & 1u) ? v[e] : v[e] * gate_off;
  }
}

struct XIgemmParams {
  const u16* x; int N, H, W, Cin, Cpi;
  const u16* wp; const float* bias;
  float* yf; int64_t ysn, ysh, ysw;       // fp32 NHWC view output (or null)
  u16* ys; int Cpo;                       // split dense output (or null)
  int Ho, Wo, Cout;
  const u16* gate; int gate_act; float gate_slope;   // split dense, geometry of y
  const unsigned char* gate_mask;         // alternative to gate: 1 bit per element, [pixel][Cpo/8] (what mask_out wrote)
  unsigned char* mask_out;                // optional with ys: bit = (hi plane of the result > 0), [pixel][Cpo/8]
  int ks, pad, act; float slope;
  int Kp, Kt, Np;
  int64_t M;
  unsigned x_bytes, wp_bytes;
  float* colsum;                          // optional [G][Np] per-tile column sums of the split output
  int G;                                  // rows of colsum (tiles past the kernel's own are zero-filled)
  int CS, nslabs, SPS, PXS, tilesX, tilesY;   // halo kernel: channel slab, stages per slab, halo pixel stride
  int CSl, SPSl;                              // ... of the last slab
  int rows16;                                 // conv_halo64, PT = 4 instance: tile rows [0, rows16) are 16 pixels high, the rest 12 (launch_xhalo64; set there)
  int stripX, stripY;                         // ... and a strip of stripX TRANSPOSED tile columns of 12 pixels (16 rows high, stripY of them) right of the tilesX columns of 16
  int ap;                                     // planes of x multiplied: 2 = hi + lo, 1 = hi only (two MFMAs per product)
  int wplanes;                                // planes of the weights multiplied: 2, or 1 with ap == 1 (ONE MFMA per product; conv_halo64 only)
  int f16;                                    // with ap == wplanes == 1: x is ONE fp16 plane [pixel][Cpi], the pack's hi rows are fp16
  unsigned y_bytes, m_bytes;                  // pointwise kernel: extents of the output and of the 1-bit masks
  // pointwise kernel, optional tail layer (a second 1x1 conv of <= 4 couts applied to the tile while it is in LDS)
  const u16* wp2; const float* bias2; float* y2; int64_t y2sn, y2sh, y2sw;
  int Cout2, act2, Kt2; float slope2; unsigned wp2_bytes, y2_bytes;
};

// DBUF: two LDS stage buffers and one barrier per stage (2 workgroups per CU), or one buffer and two
// barriers per stage (3 workgroups per CU = 3 waves per SIMD to cover the barriers and LDS latency).
// DBG (timing-only ablations, results are wrong): 1 = no MFMA, 2 = no LDS stores, 8 = no LDS fragment
// reads, 16 = no barriers.  DBG = 0 is the product kernel.
template <int NT, bool PADDED, bool DBUF, int DBG = 0>
__global__ __launch_bounds__(256, DBUF ? 2 : 3) void conv_igemm_bf16x3_kernel(XIgemmParams p) {
  constexpr int BN = NT * 16;
  constexpr int NJ = (BN + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  // per buffer (u16 units): A hi [XBM][32], A lo at +XBM*32+32 (64 B shift), then B hi / B lo likewise
  constexpr int A_LO = XBM * XROW + 32, A_ELEMS = 2 * XBM * XROW + 64;
  constexpr int B_LO = BN * XROW + 32, B_ELEMS = 2 * BN * XROW + 64;
  constexpr int BUF = A_ELEMS + B_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: wave-uniform tests and LDS-DMA destinations stay scalar code)
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private 4 MB L2 each), so
  // give each XCD one contiguous run of pixel tiles (speed only, any placement is correct).
  int tile;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int64_t m0 = (int64_t)tile * XBM;
  const int n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);

  // loader mapping: 8 consecutive threads = one row's 2 planes x 4 vectors of 8 bf16
  const int vq = tid & 3, pl = (tid >> 2) & 1, prow = tid >> 3;
  unsigned abase[4]; int aiy[4], aix[4];
  const int64_t HoWo = (int64_t)p.Ho * p.Wo;
  const int pixb = 4 * p.Cpi;                           // bytes per input pixel (2 planes of bf16)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = m0 + prow + 32 * j;
    if (m < p.M) {
      const int n = (int)(m / HoWo);
      const int r = (int)(m - (int64_t)n * HoWo);
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      aiy[j] = oy - p.pad; aix[j] = ox - p.pad;
      abase[j] = (unsigned)((((int64_t)n * p.H + aiy[j]) * p.W + aix[j]) * pixb + pl * 2 * p.Cpi);
    } else {
      aiy[j] = -(1 << 28); aix[j] = -(1 << 28); abase[j] = XOOB;      // stays out of range for every tap
    }
  }
  unsigned wbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int nrow = prow + 32 * j;
    wbase[j] = (nrow < BN && n0 + nrow < p.Np) ? (unsigned)((((n0 + nrow) * 2 + pl) * p.Kt + vq * 8) * 2) : XOOB;
  }
  int ci = vq * 8, tdy = 0, tdx = 0;
  while (ci >= p.Kp) { ci -= p.Kp; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
  const int nchunks = p.Kt / XKC;

  auto load_chunk = [&](int c, u32x4* ra, u32x4* rb) {
    // taps past ks*ks fall outside the tensor (or hit zero weights): no tap predicate needed.
    // Stages past the end (the K loop is run in pairs) load nothing: out-of-range offsets.
    const unsigned kill = c < nchunks ? 0u : XOOB;
    const unsigned toff = (unsigned)((tdy * p.W + tdx) * pixb + ci * 2) | kill;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned off = abase[j] + toff;
      if (PADDED) {
        const int iy = aiy[j] + tdy, ix = aix[j] + tdx;
        off = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? off : XOOB;
      }
      if (DBG & 4) ra[j] = u32x4{off, 0u, 0u, 0u};             // ablation: no A-operand load instruction at all
      else ra[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (DBG & 32) rb[j] = u32x4{wbase[j], 0u, 0u, 0u};       // ablation: no B-operand load instruction
      else rb[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, (wbase[j] + (unsigned)(c * XKC * 2)) | kill, 0, 0);
    }
    ci += XKC;
    while (ci >= p.Kp) { ci -= p.Kp; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
  };
  const int wslot = (vq ^ ((prow >> 1) & 3)) * 8;       // swizzled 16-byte slot of this thread's vector
  auto store_chunk = [&](int buf, const u32x4* ra, const u32x4* rb) {
    if (DBG & 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(ra[j]));
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(rb[j]));
      return;
    }
    u16* a = smem16 + buf * BUF + pl * A_LO + wslot;
    u16* b = smem16 + buf * BUF + A_ELEMS + pl * B_LO + wslot;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(a + (prow + 32 * j) * XROW) = ra[j];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nrow = prow + 32 * j;
      if (nrow < BN) *reinterpret_cast<u32x4*>(b + nrow * XROW) = rb[j];
    }
  };

  f32x4 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // DBG & 64: in-kernel stamps (s_memtime) accumulate per-phase cycles of every wave into p.colsum
  // reinterpreted as u64 [tile][wave][8] (diagnostic build: read the shares, not the run time).
  unsigned long long st_prev = 0, st_acc[6] = {0, 0, 0, 0, 0, 0}, st_rt[7] = {0, 0, 0, 0, 0, 0, 0};
  auto rstamp = [&](int i) {                   // (stamp builds) wall clock, 100 MHz: kernel entry / loop start / loop end / exit
    if (DBG & 64) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      st_rt[i] = t;
    }
  };
  rstamp(0);
  auto stamp = [&](int i) {
    if (DBG & 64) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 0) st_acc[i] += t - st_prev;
      st_prev = t;
    }
  };

  const int frow = lane & 15;
  const int fslot = ((lane >> 4) ^ ((frow >> 1) & 3)) * 8;   // MFMA 16x16x32: lane holds k = 8*(lane>>4) .. +7
  auto compute = [&](int buf) {
    const u16* a = smem16 + buf * BUF + (wave * 32 + frow) * XROW + fslot;
    const u16* b = smem16 + buf * BUF + A_ELEMS + frow * XROW + fslot;
    // every fragment read of the stage is issued before the first MFMA (hipcc otherwise emits
    // read -> lgkmcnt(0) -> 6 MFMAs per cout tile and exposes the LDS latency seven times per stage)
    bf16x8 ah[2], al[2], wh[NT], wl[NT];
    if (DBG & 8) {
      const bf16x8 z = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, 0u, 0u, 0u});
#pragma unroll
      for (int i = 0; i < 2; ++i) { ah[i] = z; al[i] = z; }
#pragma unroll
      for (int j = 0; j < NT; ++j) { wh[j] = z; wl[j] = z; }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(a + i * 16 * XROW);
        al[i] = *reinterpret_cast<const bf16x8*>(a + A_LO + i * 16 * XROW);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        wh[j] = *reinterpret_cast<const bf16x8*>(b + j * 16 * XROW);
        wl[j] = *reinterpret_cast<const bf16x8*>(b + B_LO + j * 16 * XROW);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);                                // fragment reads issued and returned
    if (DBG & 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { asm volatile("" ::"v"(ah[i])); asm volatile("" ::"v"(al[i])); }
#pragma unroll
      for (int j = 0; j < NT; ++j) { asm volatile("" ::"v"(wh[j])); asm volatile("" ::"v"(wl[j])); }
      return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah[i], acc[j][i], 0, 0, 0);   // small terms first
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al[i], acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[i], acc[j][i], 0, 0, 0);
      }
    }
  };

  // prologue: stage 0 in LDS, stage 1 in flight in the second register set
  u32x4 ra0[4], rb0[NJ], ra1[4], rb1[NJ];
  load_chunk(0, ra0, rb0);
  load_chunk(1, ra1, rb1);
  store_chunk(0, ra0, rb0);
  __syncthreads();
  if (DBUF) {
    // Straight-line body, two stages per trip (a stage past the end multiplies zeros): no branch
    // between a load and its use, so hipcc's vmcnt bookkeeping keeps both register sets in flight.
    stamp(-1);
    for (int c = 0; c < nchunks; c += 2) {
      load_chunk(c + 2, ra0, rb0);          // set 0 is free; set 1 holds stage c+1
      stamp(0);                             // global loads issued
      compute(0);                           // stage c from LDS buffer 0
      stamp(2);                             // MFMAs issued
      store_chunk(1, ra1, rb1);
      stamp(3);                             // vmcnt wait + LDS stores
      if (!(DBG & 16)) __syncthreads();
      stamp(4);                             // barrier
      load_chunk(c + 3, ra1, rb1);          // set 1 is free; set 0 holds stage c+2
      stamp(0);
      compute(1);                           // stage c+1 from LDS buffer 1
      stamp(2);
      store_chunk(0, ra0, rb0);
      stamp(3);
      if (!(DBG & 16)) __syncthreads();
      stamp(4);
    }
    if (DBG & 64) {
      if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum) + ((int64_t)tile * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = st_acc[i];
        o[5] = st_prev; o[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
      }
    }
  } else {
    for (int c = 0; c < nchunks; c += 2) {
      if (c + 2 < nchunks) load_chunk(c + 2, ra0, rb0);
      compute(0);
      __syncthreads();                                   // every wave has read stage c
      if (c + 1 >= nchunks) break;
      store_chunk(0, ra1, rb1);
      __syncthreads();
      if (c + 3 < nchunks) load_chunk(c + 3, ra1, rb1);
      compute(0);
      __syncthreads();
      if (c + 2 < nchunks) { store_chunk(0, ra0, rb0); __syncthreads(); }
    }
  }

  // ---- epilogue: lane holds couts n0 + j*16 + 4*(lane>>4) + {0..3} of pixel (lane&15).
  // Bias / activation / gate in registers, then the tile goes through LDS (free after the last
  // barrier) so that HBM sees whole 16-byte-per-lane contiguous rows instead of 8-byte fragments.
  const int fq = (lane >> 4) * 4;
  if (p.ys) {
    constexpr int OLD = 2 * BN + 8;                      // bf16 per LDS pixel row: [hi BN][lo BN] + pad
    u16* so = smem16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = wave * 32 + i * 16 + frow;
      const int64_t m = m0 + pr;
      const u16* gp = (p.gate && m < p.M) ? p.gate + (int64_t)m * 2 * p.Cpo : nullptr;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (co + e < p.Cout) {
            if (p.bias) v[e] += p.bias[co + e];
            v[e] = act_apply(v[e], p.act, p.slope);
          } else {
            v[e] = 0.f;
          }
        }
        if (gp && co < p.Cpo) {
          const uint2 g2 = *reinterpret_cast<const uint2*>(gp + co);
          const u16 g[4] = {(u16)(g2.x & 0xffff), (u16)(g2.x >> 16), (u16)(g2.y & 0xffff), (u16)(g2.y >> 16)};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= act_gate(bf2f(g[e]), p.gate_act, p.gate_slope);
        }
        else if (p.gate_mask && m < p.M && co < p.Cpo && p.gate_act != WCMC_ACT_LINEAR) {
          // the same predicate (hi plane > 0) from the bit mask the producing launch left: 1/16 of the bytes
          const unsigned bits = (unsigned)p.gate_mask[m * (p.Cpo >> 3) + (co >> 3)] >> (co & 7);
          const float off = p.gate_act == WCMC_ACT_LEAKY_RELU ? p.gate_slope : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= ((bits >> e) & 1u) ? 1.f : off;
        }
        u16 hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split1(v[e], hi[e], lo[e]);
        *reinterpret_cast<uint2*>(so + pr * OLD + j * 16 + fq) =
            make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
        *reinterpret_cast<uint2*>(so + pr * OLD + BN + j * 16 + fq) =
            make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
      }
    }
    __syncthreads();
    constexpr int VPP = BN / 8;                          // 16-byte vectors per plane per pixel
    for (int v = tid; v < XBM * 2 * VPP; v += 256) {
      const int pr = v / (2 * VPP), q = v - pr * (2 * VPP);
      const int plane = q >= VPP, vec = q - plane * VPP;
      const int64_t m = m0 + pr;
      const int co = n0 + vec * 8;
      if (m < p.M && co < p.Cpo) {
        const u32x4 hv = *reinterpret_cast<const u32x4*>(so + pr * OLD + plane * BN + vec * 8);
        *reinterpret_cast<u32x4*>(p.ys + (int64_t)m * 2 * p.Cpo + plane * p.Cpo + co) = hv;
        if (p.mask_out && plane == 0) p.mask_out[m * (p.Cpo >> 3) + (co >> 3)] = positive_mask8(hv);
      }
    }
    if (!(DBG & 64) && p.colsum) {
      // bias gradient of the consumer layer for free: column sums of this tile (hi + lo) while it is in LDS;
      // RG row groups per column, combined through LDS in a fixed order
      constexpr int CW = BN <= 16 ? 16 : BN <= 32 ? 32 : BN <= 64 ? 64 : 128, RG = 256 / CW;
      float* red = reinterpret_cast<float*>(so + XBM * OLD);
      const int c = tid % CW, rg = tid / CW;
      const int rows = (int)min((int64_t)XBM, p.M - m0);
      float a = 0.f;
      if (c < BN)
        for (int r = rg; r < rows; r += RG) a += bf2f(so[r * OLD + c]) + bf2f(so[r * OLD + BN + c]);
      if (rg > 0 && c < BN) red[(rg - 1) * BN + c] = a;
      __syncthreads();
      if (rg == 0 && c < BN && n0 + c < p.Np) {
        for (int q = 0; q < RG - 1; ++q) a += red[q * BN + c];
        p.colsum[(int64_t)tile * p.Np + n0 + c] = a;
        // trailer: the number of rows this launch wrote (the finish kernel reads no further)
        if (tile == 0 && n0 + c == 0) reinterpret_cast<int*>(p.colsum)[(int64_t)p.G * p.Np] = (int)gridDim.x;
      }
    }
  } else {
    constexpr int OLD = BN + 4;                          // floats per LDS pixel row
    float* so = reinterpret_cast<float*>(smem16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = wave * 32 + i * 16 + frow;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (co + e < p.Cout) {
            if (p.bias) v[e] += p.bias[co + e];
            v[e] = act_apply(v[e], p.act, p.slope);
          } else {
            v[e] = 0.f;
          }
        }
        *reinterpret_cast<float4*>(so + pr * OLD + j * 16 + fq) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
    constexpr int VPP = BN / 4;                          // float4 per pixel
    for (int v = tid; v < XBM * VPP; v += 256) {
      const int pr = v / VPP, vec = v - pr * VPP;
      const int64_t m = m0 + pr;
      const int co = n0 + vec * 4;
      if (m < p.M && co < p.Cpo) {                       // Cpo = round_up(Cout, 4) here
        const int n = (int)(m / HoWo);
        const int r = (int)(m - (int64_t)n * HoWo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        *reinterpret_cast<float4*>(p.yf + (int64_t)n * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw + co) =
            *reinterpret_cast<const float4*>(so + pr * OLD + vec * 4);
      }
    }
  }
}


// Workgroup barrier for kernels that keep LDS-DMA in flight across it: __syncthreads() carries a release fence,
// for which hipcc waits for EVERY outstanding LDS-DMA (vmcnt(0)); the ring below orders its DMA by explicit counts.
__device__ __forceinline__ void pw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------ implicit GEMM, halo-resident (ks 3..5)
// Stamps of the streaming kernel above (scripts/stamp_igemm.py): per 32-k stage a wave spends 830 cycles
// issuing its 8 buffer loads and 540 storing them to LDS, against 770 issuing MFMAs -- the L1/TA path and
// L2 bandwidth (23 B/clk/CU sustained), not the matrix pipe, set the pace, and 53 % of those bytes are
// the A operand re-read once per filter tap.  This kernel keeps the input pixels of a 16x16 output
// tile with their (ks-1) halo resident in LDS for one channel slab (CS <= 64 channels, both planes) and
// reads every tap's A fragments from there with shifted addresses; only the weights stream (14 KB per
// stage for 256 pixels instead of 30 KB for 128).  512 threads = 8 waves, each 32 pixels (two tile rows)
// x all NT*16 couts; one workgroup per CU (LDS: halo 90-115 KB + two weight stages).
// K order: slab-major (pack_weight_split_kernel); stages never straddle slabs (Ks % 32 == 0).
template <int NT, int TH, int TW, int DBG = 0, int NB = 3, int AP = 2>       // AP: see conv_halo64_bf16x3_kernel
__global__ __launch_bounds__(TH * TW * 2, (TH * TW <= 128 ? 2 : 1)) void conv_halo_bf16x3_kernel(XIgemmParams p) {
  constexpr int BN = NT * 16;
  constexpr int TPX = TH * TW, NTHR = TPX * 2, NWV = NTHR / 64;   // one wave per 32 pixels (two MFMA pixel tiles)
  constexpr int TPR = TW / 16;                 // MFMA pixel tiles per tile row
  static_assert(TPX % 32 == 0 && TW % 16 == 0, "a wave = 2 pixel tiles of 16");
  constexpr int STW = BN * 4 >= NWV * 14 * 8 ? NWV : 1;    // (stamp builds: waves with a record in the tile's colsum row)
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  constexpr int B_LO = BN * XROW + 32, B_ELEMS = 2 * BN * XROW + 64;
  const int HWd = TW + p.ks - 1, HHt = TH + p.ks - 1, HP = HWd * HHt;
  char* const halo = reinterpret_cast<char*>(smem16);
  u16* const bsm = smem16 + ((HP * p.PXS + 127) & ~127) / 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: wave-uniform tests and LDS-DMA destinations stay scalar code)
  int tile;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tpi = p.tilesX * p.tilesY;
  const int img = tile / tpi, trem = tile - img * tpi;
  const int oy0 = (trem / p.tilesX) * TH, ox0 = (trem % p.tilesX) * TW;
  const int n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  const int pixb = 4 * p.Cpi;

  // ---- halo: [pixel][hi CS][lo CS] at stride PXS; out-of-image pixels and channels >= Cpi read zeros.
  // Filled by LDS-DMA as one linear run of 16-byte vectors (PXS / 16 per pixel, the last ones pad): wave
  // instruction ii writes vectors [64 ii, 64 ii + 64), the per-lane source picks pixel / plane / channel.
  int cs_cur = p.nslabs == 1 ? p.CSl : p.CS;   // channels of the slab being multiplied (the last one may be narrower)
  int sps_cur = p.nslabs == 1 ? p.SPSl : p.SPS;
  const int VP = p.PXS / 16;                   // vectors per halo pixel with pad
  const int hvecs = HP * VP;
  const float invVP = 1.0f / (float)VP, invHW = 1.0f / (float)HWd;
  auto dma_halo = [&](int slab) {
    const int V = (slab == p.nslabs - 1 ? p.CSl : p.CS) / (AP == 1 ? 8 : 4);      // data vectors per halo pixel (AP planes x cs/8)
    for (int ii = wave; ii * 64 < hvecs; ii += NTHR / 64) {
      const int v = ii * 64 + lane;
      if (v < hvecs) {
        const int px = (int)(((float)v + 0.5f) * invVP), part = v - px * VP;     // exact: v < 2^13
        const int hy = (int)(((float)px + 0.5f) * invHW), hx = px - hy * HWd;
        const int iy = oy0 - p.pad + hy, ix = ox0 - p.pad + hx;
        const int plane = AP == 1 ? 0 : part >= (V >> 1), vec = part - plane * (V >> 1);
        const int ch = slab * p.CS + vec * 8;
        unsigned off = XOOB;
        if (part < V && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && ch < p.Cpi)
          off = (unsigned)(((img * p.H + iy) * p.W + ix) * pixb + plane * 2 * p.Cpi + ch * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(halo + ii * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  // ---- weights: LDS-DMA (buffer_load ... lds), no staging registers and no ds_write pass.  One wave
  // instruction fills 16 cout rows x 64 B of one plane (1 KB, lane-linear destination: row 16*wave + lane/4,
  // 16-byte slot lane%4); the XOR swizzle of the slot goes on the per-lane SOURCE column.
  const int nstages = p.Kt / XKC;
  // row group = 16 cout rows; wave w fills groups w, w + NWV, ... (one each with 8 waves; up to two with 4)
  constexpr int NGMAX = (NT + NWV - 1) / NWV;
  const int ngroups = wave < NT ? (NT - wave + NWV - 1) / NWV : 0;       // wave-uniform
  unsigned dbase[NGMAX], dbase2[NGMAX];
#pragma unroll
  for (int q = 0; q < NGMAX; ++q) {
    const int drow = 16 * (wave + q * NWV) + (lane >> 2);
    const int dvq = (lane & 3) ^ ((drow >> 1) & 3);
    dbase[q] = (q < ngroups && n0 + drow < p.Np) ? (unsigned)(((n0 + drow) * 2 * p.Kt + dvq * 8) * 2) : XOOB;
    dbase2[q] = dbase[q] >= XOOB ? XOOB : dbase[q] + (unsigned)(p.Kt * 2);
  }
  // one row group (hi + lo plane: two wave instructions) of stage g's weights; one addition per instruction (the stage's
  // byte offset is a scalar; stages past the end add 2^30: valid rows -- the packed weights are a few MB -- and invalid
  // ones (2^31) alike land beyond the buffer, without wrapping)
  auto dma_b_group = [&](int g, int buf, int q) {
    if (q < ngroups) {
      const unsigned sg = g < nstages ? (unsigned)(g * XKC * 2) : 0x40000000u;
      const unsigned off = dbase[q] + sg;
      const unsigned off2 = dbase2[q] + sg;
      u16* d = bsm + buf * B_ELEMS + 16 * (wave + q * NWV) * XROW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(d + B_LO), 16, off2, 0, 0, 0);
    }
  };
  auto dma_b = [&](int g, int buf) {
#pragma unroll
    for (int q = 0; q < NGMAX; ++q) dma_b_group(g, buf, q);
  };

  f32x4 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  unsigned long long st_prev = 0, st_acc[6] = {0, 0, 0, 0, 0, 0}, st_rt[7] = {0, 0, 0, 0, 0, 0, 0};
  auto rstamp = [&](int i) {                   // (stamp builds) wall clock, 100 MHz: kernel entry / loop start / loop end / exit
    if (DBG & 64) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      st_rt[i] = t;
    }
  };
  rstamp(0);
  auto stamp = [&](int i) {
    if (DBG & 64) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 0) st_acc[i] += t - st_prev;
      st_prev = t;
    }
  };

  // ---- fragments: lane = pixel (lane & 15) of a 16-pixel row segment, k group kg = lane >> 4 (8 k each)
  const int frow = lane & 15, kg = lane >> 4;
  const int fslot = (kg ^ ((frow >> 1) & 3)) * 8;
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pt = wave * 2 + i;
    abase[i] = ((pt / TPR) * HWd + (pt % TPR) * 16 + frow) * p.PXS;
  }
  int cl = kg * 8, tdx = 0, tdy = 0, aoff = cl * 2;      // this lane's (channel, tap) inside the slab
  int lo_off = cs_cur * 2;
  // Software pipeline inside every wave (stamps of the first version: all eight waves read fragments,
  // then all multiply -- 53 % MFMA issue occupancy; a two-group ping-pong did no better): the fragments of
  // stage g+1 are read WHILE the MFMAs of stage g issue, cout tile by cout tile into the registers the
  // tile's MFMAs have just consumed, so no wave ever waits for LDS with an idle matrix pipe.  NB weight
  // buffers: while stage g multiplies (its fragments are in registers), stage g+1 is read from its buffer and
  // the DMAs of stages g+2 .. g+NB-1 are in flight or landed (one stage of latency cover was not enough: stamps
  // showed 400 of 2340 cycles per stage waiting for the weights); each wave waits for its own share of stage
  // g+1 with a counted vmcnt before the stage barrier (no fence: a release fence would drain every DMA).
  bf16x8 ah[2], al[2], wh[NT], wl[NT];
  auto read_a = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + aoff);
      if (AP == 2) al[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + aoff + lo_off);
    }
    // the following stage's tap / channel of this lane (CS >= 32: at most one wrap); taps past ks*ks (slab
    // padding, zero weights) read the tile's first pixels
    cl += XKC;
    if (cl >= cs_cur) { cl -= cs_cur; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
    aoff = tdy < p.ks ? (int)__umul24(__umul24((unsigned)tdy, (unsigned)HWd) + (unsigned)tdx, (unsigned)p.PXS) + cl * 2 : 0;
  };
  const u16* const bfrag = bsm + frow * XROW + fslot;
  auto read_b = [&](int buf, int j) {
    wh[j] = *reinterpret_cast<const bf16x8*>(bfrag + buf * B_ELEMS + j * 16 * XROW);
    wl[j] = *reinterpret_cast<const bf16x8*>(bfrag + buf * B_ELEMS + B_LO + j * 16 * XROW);
  };

#pragma unroll
  for (int b = 0; b < NB; ++b) dma_b(b, b);
  dma_halo(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read_a();
#pragma unroll
  for (int j = 0; j < NT; ++j) read_b(0, j);
  int s_in = 0, slab = 0, bcur = 0;
  rstamp(1);
  stamp(-1);
  for (int g = 0; g < nstages; ++g) {
    const int b1 = bcur + 1 == NB ? 0 : bcur + 1;      // buffer of stage g+1; stage g's fragments are in registers
    // this wave's share of stage g+1 has landed; the NB-2 stages behind it (two DMA instructions each) stay in flight
    if (NGMAX == 1 || ngroups < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NB - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NB - 2)) : "memory");
    stamp(4);                                    // (stamp builds: slot 4 = the wait for this wave's own weight DMA)
    if (!(DBG & 16)) pw_barrier();               // ... everyone's; and everyone has read stage g's fragments
    stamp(0);
    if (!(DBG & 2)) dma_b(g + NB, bcur);           // (DBG & 2, timing only: no weight stream inside the loop)
    bcur = b1;
    stamp(5);                                    // weight DMA issued
    const bool last_of_slab = (s_in + 1 == sps_cur);
    // the fragments of a slab's last stage are in registers and the barrier above retired every read of the
    // halo: the next slab's halo lands while this stage multiplies
    if (last_of_slab && slab + 1 < p.nslabs && !(DBG & 4)) dma_halo(slab + 1);      // (DBG & 4, timing only: one halo per tile)
    stamp(1);
    // A fragments of stage g+1: with two workgroups per CU (8x16 tiles) they replace a pixel tile's registers as soon as its
    // last MFMAs of this stage have issued (LATE; reading them into a second register set during the first cout tile and
    // copying costs 8 v_mov_b64 per stage in a loop of 24 MFMAs that is bound by vector issue: 64 -> 64 at 128^2 43.3 -> 41.9
    // us); with ONE workgroup per CU (16x16 tiles, all eight waves in step) the early read hides the LDS latency that
    // nothing else covers there and stays (128 -> 128 at 64^2: 37 us early, 39-40 late).
    constexpr bool LATE = TH * TW <= 128;
    bf16x8 ahn[2], aln[2];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!(DBG & 1)) {
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah[i], acc[j][i], 0, 0, 0);   // small terms first
          if (AP == 2) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al[i], acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[i], acc[j][i], 0, 0, 0);
        }
        if (LATE && j == NT - 1 && !last_of_slab && !(DBG & 8)) {
          ah[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + aoff);
          if (AP == 2) al[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + aoff + lo_off);
        }
      }
      if (!(DBG & 8)) read_b(b1, j);             // stage g+1, same cout tile, into the registers just consumed
      if (!LATE && j == 0 && !last_of_slab && !(DBG & 8)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ahn[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + aoff);
          if (AP == 2) aln[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + aoff + lo_off);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(2);
    if (!last_of_slab) {
      if (!LATE) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = ahn[i]; if (AP == 2) al[i] = aln[i]; }
      }
      cl += XKC;
      if (cl >= cs_cur) { cl -= cs_cur; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
      aoff = tdy < p.ks ? (int)__umul24(__umul24((unsigned)tdy, (unsigned)HWd) + (unsigned)tdx, (unsigned)p.PXS) + cl * 2 : 0;
      ++s_in;
    } else {                                     // slab boundary: the next A fragments come from the next halo
      s_in = 0;
      ++slab;
      if (slab == p.nslabs - 1) { cs_cur = p.CSl; sps_cur = p.SPSl; lo_off = cs_cur * 2; }
      cl = kg * 8; tdx = 0; tdy = 0; aoff = cl * 2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the new halo (and of stage g+2)
      __syncthreads();
      read_a();
    }
    stamp(3);                                // tail of the stage (slab boundaries included: halo wait + barrier + re-read)
  }
  // ---- epilogue operands: the bias of this lane's couts and the gate of its (pixel, cout) quads, as ONE batch of
  // unconditional buffer loads (out of range -> 0) issued before the drain.  (The first version loaded them one by one
  // inside the per-element branches: 56 global loads, each with its own full wait -- 11-12 us of a ~105 us tile.)
  const int fq = kg * 4;
  auto pix_of = [&](int pr, int& oy, int& ox) {
    const int pt = pr >> 4;
    oy = oy0 + pt / TPR; ox = ox0 + (pt % TPR) * 16 + (pr & 15);
    return oy < p.Ho && ox < p.Wo;
  };
  float bv[NT][4];
  {
    const __amdgpu_buffer_rsrc_t brs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.wp), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bv[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (n0 + j * 16 + fq + e) * 4, 0, 0));
  }
  const bool use_gate = p.ys && p.gate, use_mask = p.ys && !p.gate && p.gate_mask && p.gate_act != WCMC_ACT_LINEAR;
  u32x2 gv[2][NT];                             // split gate: 4 hi-plane bf16 per quad; bit mask: one byte in .x
  bool okp[2]; int64_t mp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int oy, ox;
    okp[i] = pix_of(wave * 32 + i * 16 + frow, oy, ox);
    mp[i] = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
  }
  if (use_gate) {
    const int64_t gbytes = (int64_t)p.N * p.Ho * p.Wo * 4 * p.Cpo;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate, 0, (int)(gbytes < 0x7fffffff ? gbytes : 0x7fffffff), 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        gv[i][j] = __builtin_amdgcn_raw_buffer_load_b64(grs, (okp[i] && co < p.Cpo) ? (unsigned)((mp[i] * 2 * p.Cpo + co) * 2) : XOOB, 0, 0);
      }
  } else if (use_mask) {
    const int64_t mbytes = (int64_t)p.N * p.Ho * p.Wo * (p.Cpo >> 3);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate_mask, 0, (int)mbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        gv[i][j].x = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(mrs, (okp[i] && co < p.Cpo) ? (unsigned)(mp[i] * (p.Cpo >> 3) + (co >> 3)) : XOOB, 0, 0);
      }
  }
  const XAct ak = x_act(p.act, p.slope);
  const float gate_off = p.gate_act == WCMC_ACT_RELU ? 0.f : p.gate_act == WCMC_ACT_LEAKY_RELU ? p.gate_slope : 1.f;
  const int gkind = use_gate ? 1 : use_mask ? 2 : 0;
  rstamp(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the (zero) weight stages past the end have landed:
  __syncthreads();                                     // LDS is free for the epilogue staging
  rstamp(3);
  if ((DBG & 32) && !(DBG & 64)) {                     // timing only: no epilogue (one store keeps the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) keep += (acc[j][0][0] + acc[j][0][1] + acc[j][0][2] + acc[j][0][3]) +
                                         (acc[j][1][0] + acc[j][1][1] + acc[j][1][2] + acc[j][1][3]);
    if (keep == 12345.678f && p.ys) p.ys[0] = 1;
    return;
  }
  if (DBG & 64) {
    if (lane == 0 && wave < STW) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum) + ((int64_t)tile * STW + wave) * 14;
      for (int i = 0; i < 6; ++i) o[i] = st_acc[i];
      for (int i = 0; i < 3; ++i) o[6 + i] = st_rt[i];
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[13] = hw | ((unsigned long long)xcc << 32);
    }
  }

  // ---- epilogue (as the streaming kernel; pixels of the tile outside the image are written as zeros to LDS
  // and skipped on the way out).  Tile-local pixel pr = 16 * pixel-tile + column.
  if (p.ys) {
    constexpr int OLD = 2 * BN + 8;
    u16* so = smem16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = wave * 32 + i * 16 + frow;
      const bool ok = okp[i];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        float v[4];
        // (gate: the hi-plane predicate from the split tensor, or from the bit mask the producing launch left)
        x_epi_quad(acc[j][i], bv[j], ok, ak, gkind, gv[i][j], co, gate_off, v);
        unsigned h01, l01, h23, l23;
        x_split2(v[0], v[1], h01, l01);
        x_split2(v[2], v[3], h23, l23);
        *reinterpret_cast<uint2*>(so + pr * OLD + j * 16 + fq) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(so + pr * OLD + BN + j * 16 + fq) = make_uint2(l01, l23);
      }
    }
    __syncthreads();
    rstamp(4);
    constexpr int VPP = BN / 8;
    for (int v = tid; v < TPX * 2 * VPP; v += NTHR) {
      const int pr = v / (2 * VPP), q = v - pr * (2 * VPP);
      const int plane = q >= VPP, vec = q - plane * VPP;
      const int co = n0 + vec * 8;
      int oy, ox;
      if (pix_of(pr, oy, ox) && co < p.Cpo) {
        const int64_t m = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
        const u32x4 hv = *reinterpret_cast<const u32x4*>(so + pr * OLD + plane * BN + vec * 8);
        *reinterpret_cast<u32x4*>(p.ys + m * 2 * p.Cpo + plane * p.Cpo + co) = hv;
        if (p.mask_out && plane == 0) p.mask_out[m * (p.Cpo >> 3) + (co >> 3)] = positive_mask8(hv);
      }
    }
    rstamp(5);
    if (!(DBG & 64) && p.colsum) {
      constexpr int CW = BN <= 16 ? 16 : BN <= 32 ? 32 : BN <= 64 ? 64 : 128, RG = NTHR / CW;
      float* red = reinterpret_cast<float*>(so + TPX * OLD);
      const int c = tid % CW, rg = tid / CW;
      float a = 0.f;
      if (c < BN)
        for (int r = rg; r < TPX; r += RG) a += bf2f(so[r * OLD + c]) + bf2f(so[r * OLD + BN + c]);
      if (rg > 0 && c < BN) red[(rg - 1) * BN + c] = a;
      __syncthreads();
      if (rg == 0 && c < BN && n0 + c < p.Np) {
        for (int q = 0; q < RG - 1; ++q) a += red[q * BN + c];
        p.colsum[(int64_t)tile * p.Np + n0 + c] = a;
        // trailer: the number of rows this launch wrote (the finish kernel reads no further)
        if (tile == 0 && n0 + c == 0) reinterpret_cast<int*>(p.colsum)[(int64_t)p.G * p.Np] = (int)gridDim.x;
      }
    }
  } else {
    constexpr int OLD = BN + 4;
    float* so = reinterpret_cast<float*>(smem16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = wave * 32 + i * 16 + frow;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        float v[4];
        x_epi_quad(acc[j][i], bv[j], true, ak, 0, u32x2{0u, 0u}, co, 1.f, v);
        *reinterpret_cast<float4*>(so + pr * OLD + j * 16 + fq) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
    constexpr int VPP = BN / 4;
    for (int v = tid; v < TPX * VPP; v += NTHR) {
      const int pr = v / VPP, vec = v - pr * VPP;
      const int co = n0 + vec * 4;
      int oy, ox;
      if (pix_of(pr, oy, ox) && co < p.Cpo)
        *reinterpret_cast<float4*>(p.yf + (int64_t)img * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw + co) =
            *reinterpret_cast<const float4*>(so + pr * OLD + vec * 4);
    }
  }
  if (DBG & 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the stores have left)
    rstamp(6);
    if (lane == 0 && wave < STW) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum) + ((int64_t)tile * STW + wave) * 14;
      for (int i = 3; i < 7; ++i) o[6 + i] = st_rt[i];
    }
  }
}


// ------------------------------------------------------------------ implicit GEMM, halo-resident, 3x3, K split over two wave groups
// The PathNet U-Net's 3x3 layers (64 .. 384 channels, 128^2 .. 32^2 pixels; support/networks.py:20-22).  In the kernel above one
// wave walks a 32-k stage in ~1,000-1,500 cycles of which 384 are its 24 MFMAs: an in-order wave pays the stage barrier, its two
// weight-DMA instructions, the LDS latency of its last fragment reads and the tap arithmetic one after the other, and with LDS
// for two 128-pixel workgroups per CU only two waves share a SIMD to cover them (profiles/r02_halo_unet_timeline.txt: 15 us in the
// stage loop for 7 us of MFMAs; one wave per SIMD on the 64^2 / 32^2 levels).  Here a workgroup is EIGHT waves on the same 8x16
// tile: wave (pg, grp) owns the 32 pixels of tile rows 2 pg, 2 pg + 1 as before, and the two groups grp = 0 / 1 multiply the even
// / odd 32-k stages of the tile -- half the stages, barriers and DMA issues per wave, FOUR waves per SIMD at two workgroups per CU
// (<= 128 VGPRs), the same weight stream per workgroup.  The two groups' partial sums meet in LDS after the loop (grp 0 + grp 1,
// a fixed order) and each group finishes half of the cout tiles, so the epilogue per wave halves too.
//   * One barrier per PAIR of stages; four weight buffers (pair being read, pair landing).  They fit beside the second workgroup
//     because a halo pixel is a 256-byte record without pad (AP = 2: 64 channels hi | lo; AP = 1: 128 channels of the hi plane):
//     the 16-byte unit q of pixel p sits at slot q ^ (p & 15), MFMA row r of a pixel tile is pixel column {1,3,5,7, 0,2,..,14,
//     9,11,13,15}[r] and k-group kg reads unit swap01(kg) + 4 h: the two 8-lane halves of a ds_read_b128 lane group then hold
//     pixels of opposite parity and units that differ in bit 1, i.e. sixteen different slots for every filter tap (the 288-byte
//     stride of the kernel above buys the same with 32 bytes of pad per pixel, which is what did not fit).
//   * ks = 3 and the slab width are template constants and a slab's iterations are unrolled: a tap's halo offset is an immediate
//     addition on the lane's pixel index, the k-half inside a tap an XOR constant; no tap counters, no wraps.
// Stage s of a slab = tap s / SPT, 32-channel quarter s % SPT of the slab's CS = 32 SPT / AP ... channels (pack order k = tap CS + c,
// as x_plan_k and pack_weight_split_multi_kernel lay it out: the packs are those of the kernel above).
// DBG (debug library, timing only, WRONG results): 1 no MFMA, 2 no weight DMA in the loop, 8 no fragment reads, 16 no loop barrier, 32 no epilogue
template <int AP, int SPT, int KG = 2, int DBG = 0>
__global__ __launch_bounds__(256 * KG, 4) void conv_halo3_bf16x3_kernel(XIgemmParams p) {
  static_assert(KG == 2, "two K groups (h = hc | grp below)");
  constexpr int NT = 4, BN = NT * 16, TH = 8, TW = 16, KS = 3, HWd = TW + KS - 1, HHt = TH + KS - 1, HP = HWd * HHt;   // 18 x 10 halo
  constexpr int NTHR = 256 * KG, NWV = 4 * KG, TPX = TH * TW, PXB = 256;
  constexpr int B_LO = BN * XROW + 32, B_ELEMS = 2 * BN * XROW + 64;
  constexpr int ITS = KS * KS * SPT / KG;                 // iterations (stage pairs) per slab: 9 or 18
  static_assert((KS * KS * SPT) % KG == 0, "a slab is a whole number of stage pairs");
  constexpr int CSU = AP == 2 ? 8 : 4 * SPT, CS = CSU * 8;    // 16-byte units / channels of one plane of a slab
  static_assert(AP == 1 || SPT == 2, "two planes: 64-channel slabs");
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  char* const halo = reinterpret_cast<char*>(smem16);
  u16* const bsm = smem16 + HP * PXB / 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave & 3, grp = wave >> 2;               // pixel group (tile rows 2 pg, 2 pg + 1), K group
  int tile;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tpi = p.tilesX * p.tilesY;
  const int img = tile / tpi, trem = tile - img * tpi;
  const int oy0 = (trem / p.tilesX) * TH, ox0 = (trem % p.tilesX) * TW;
  const int n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  const int pixb = 4 * p.Cpi;

  // ---- halo: 180 records of 16 units, filled by LDS-DMA as 45 linear kilobytes; wave instruction ii covers pixels 4 ii .. 4 ii + 3,
  // the per-lane SOURCE picks the unit that belongs into the lane's slot.  The source offsets of slab 0 are worked out once; a
  // slab's fill is one addition per instruction.
  constexpr int NHI = (HP * 16 / 64 + NWV - 1) / NWV;      // 45 instructions over 8 waves: up to 6 each
  static_assert(HP * 16 % 64 == 0, "no tail instruction");
  static_assert(NHI == 6, "hoff");                        // (a literal bound: an array of dependent size captured by the lambda below loses the kernel's host stub, clang 22)
  unsigned hoff[6];
#pragma unroll
  for (int kq = 0; kq < NHI; ++kq) {
    const int ii = wave + NWV * kq;
    const int px = ii * 4 + (lane >> 4), slot = lane & 15;
    const int q = slot ^ (px & 15);
    const int w = (q & 12) | ((q & 1) << 1) | ((q >> 1) & 1);          // source unit: bits 0 and 1 swapped
    const int hy = (px * 3641) >> 16, hx = px - hy * HWd;               // px / 18, exact below 180
    const int iy = oy0 - p.pad + hy, ix = ox0 - p.pad + hx;
    const int plane = AP == 2 ? (w >> 3) : 0, chunk = AP == 2 ? (w & 7) : w;
    hoff[kq] = XOOB;
    if (ii * 64 < HP * 16 && chunk < CSU && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      hoff[kq] = (unsigned)(((img * p.H + iy) * p.W + ix) * pixb + plane * 2 * p.Cpi + chunk * 16);
  }
  auto dma_halo = [&](int slab) {
    const unsigned so = (unsigned)(slab * CS * 2);
#pragma unroll
    for (int kq = 0; kq < NHI; ++kq) {
      const int ii = wave + NWV * kq;
      if (ii * 64 < HP * 16)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(halo + ii * 1024), 16, hoff[kq] + so, 0, 0, 0);
    }
  };

  // ---- weights: LDS-DMA, 2 KG stage buffers.  A pair of stages is sixteen 1-KB pieces (stage, row group, plane); wave (pg, grp)
  // fetches both planes of row group pg of ITS group's stage.
  const int nstages = p.Kt / XKC;
  const int drow = 16 * pg + (lane >> 2);
  const int dvq = (lane & 3) ^ ((drow >> 1) & 3);
  const unsigned dbase = n0 + drow < p.Np ? (unsigned)(((n0 + drow) * 2 * p.Kt + dvq * 8) * 2) : XOOB;
  const unsigned dbase2 = dbase >= XOOB ? XOOB : dbase + (unsigned)(p.Kt * 2);
  auto dma_b = [&](int g, int buf) {                        // global stage g -> buffer buf (stages past the end: out of range, zeros)
    const unsigned sg = g < nstages ? (unsigned)(g * XKC * 2) : 0x40000000u;
    u16* d = bsm + buf * B_ELEMS + 16 * pg * XROW;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)d, 16, dbase + sg, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(d + B_LO), 16, dbase2 + sg, 0, 0, 0);
  };

  f32x4 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (DBG & 256) {                        // (debug build) 128 vector instructions that change nothing: the price of a VALU in the step
    int d = lane;
#pragma unroll
    for (int r = 0; r < 128; ++r) asm volatile("v_add_u32 %0, %0, %0" : "+v"(d));
    if (d == 0x12345 && p.ys) p.ys[0] = 1;
  }

  // ---- fragments
  const int frow = lane & 15, kg = lane >> 4;
  const int col = frow < 4 ? 2 * frow + 1 : frow < 12 ? 2 * (frow - 4) : 2 * (frow - 12) + 9;   // pixel column of MFMA row frow
  const int qsel = ((kg & 1) << 1) | (kg >> 1) | (grp << 2);        // unit of this lane's k-group in quarter h = hc | grp (hc below)
  int pl0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) pl0[i] = (2 * pg + i) * HWd + col;
  bf16x8 ah[2], al[2], wh[NT], wl[NT];
  // local iteration l of a slab: this group's stage s = KG l + grp -> tap s / SPT, quarter s % SPT = hc | grp
  auto read_a1 = [&](int i, int l) {
    const int tap = SPT == 2 ? l : (l >> 1), hc = SPT == 2 ? 0 : 2 * (l & 1);
    // (opaque copy: the five vector instructions of a tap's address are recomputed where they are used -- hoisted out of the slab
    // loop, the 36 addresses of a slab spilled to scratch memory, whose loads share the wave's vmcnt with the LDS-DMA)
    int pb = pl0[i];
    asm volatile("" : "+v"(pb));
    const int P = pb + (tap / KS) * HWd + (tap % KS);
    const int a = ((P << 8) | (((qsel ^ P) & 15) << 4)) ^ (hc << 6);
    ah[i] = *reinterpret_cast<const bf16x8*>(halo + a);
    if (AP == 2) al[i] = *reinterpret_cast<const bf16x8*>(halo + (a ^ 128));
  };
  const int fslot = (kg ^ ((frow >> 1) & 3)) * 8;
  const u16* bfr_c = bsm + frow * XROW + fslot + grp * B_ELEMS;           // this group's buffer of the pair being multiplied next ...
  const u16* bfr_n = bfr_c + KG * B_ELEMS;                                // ... and of the pair after it
  auto read_b = [&](const u16* b, int j) {
    wh[j] = *reinterpret_cast<const bf16x8*>(b + j * 16 * XROW);
    wl[j] = *reinterpret_cast<const bf16x8*>(b + B_LO + j * 16 * XROW);
  };

  dma_b(grp, grp);
  dma_b(KG + grp, KG + grp);
  dma_halo(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) read_a1(i, 0);
#pragma unroll
  for (int j = 0; j < NT; ++j) read_b(bfr_c, j);
  { const u16* t = bfr_c; bfr_c = bfr_n; bfr_n = t; }     // bfr_c: what is read DURING the iteration (the next pair)
  int gi = 0, dset = 0;                                    // global iteration; buffer set whose fragments are in registers
  for (int slab = 0; slab < p.nslabs; ++slab) {
#pragma unroll
    for (int l = 0; l < ITS; ++l) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the next pair (requested one iteration ago)
      if (!(DBG & 16)) pw_barrier();                       // ... everyone's; everyone has read the fragments of this pair
      if (!(DBG & 2)) dma_b(KG * (gi + 2) + grp, dset * KG + grp);
      const bool last = l == ITS - 1;
      if (last && slab + 1 < p.nslabs) dma_halo(slab + 1); // (every fragment of this slab is in registers)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (!(DBG & 1)) {
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah[i], acc[j][i], 0, 0, 0);   // small terms first
            if (AP == 2) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al[i], acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[i], acc[j][i], 0, 0, 0);
          }
          if (j == NT - 1 && !last && !(DBG & 8)) read_a1(i, l + 1);      // the pixel tile's next fragments replace it at once
        }
        if (!(DBG & 8)) read_b(bfr_c, j);                   // next pair, same cout tile, into the registers just consumed
        __builtin_amdgcn_sched_barrier(0);
      }
      { const u16* t = bfr_c; bfr_c = bfr_n; bfr_n = t; }
      dset ^= 1;
      ++gi;
      if (last && slab + 1 < p.nslabs) {                    // slab boundary: the next A fragments come from the next halo
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) read_a1(i, 0);
      }
    }
  }

  if (DBG & 32) {                                      // timing only: no epilogue (one store keeps the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) keep += (acc[j][0][0] + acc[j][0][1] + acc[j][0][2] + acc[j][0][3]) +
                                         (acc[j][1][0] + acc[j][1][1] + acc[j][1][2] + acc[j][1][3]);
    if (keep == 12345.678f && p.ys) p.ys[0] = 1;
    return;
  }
  // ---- epilogue: group grp finishes cout tiles 2 grp, 2 grp + 1 of its pixels
  const int fq = kg * 4;
  float bv[2][4];
  {
    const __amdgpu_buffer_rsrc_t brs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.wp), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bv[jl][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (n0 + (2 * grp + jl) * 16 + fq + e) * 4, 0, 0));
  }
  const bool use_gate = p.ys && p.gate, use_mask = p.ys && !p.gate && p.gate_mask && p.gate_act != WCMC_ACT_LINEAR;
  u32x2 gv[2][2];
  bool okp[2]; int64_t mp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = oy0 + 2 * pg + i, ox = ox0 + col;
    okp[i] = oy < p.Ho && ox < p.Wo;
    mp[i] = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
  }
  if (use_gate) {
    const int64_t gbytes = (int64_t)p.N * p.Ho * p.Wo * 4 * p.Cpo;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate, 0, (int)(gbytes < 0x7fffffff ? gbytes : 0x7fffffff), 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int co = n0 + (2 * grp + jl) * 16 + fq;
        gv[i][jl] = __builtin_amdgcn_raw_buffer_load_b64(grs, (okp[i] && co < p.Cpo) ? (unsigned)((mp[i] * 2 * p.Cpo + co) * 2) : XOOB, 0, 0);
      }
  } else if (use_mask) {
    const int64_t mbytes = (int64_t)p.N * p.Ho * p.Wo * (p.Cpo >> 3);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate_mask, 0, (int)mbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int co = n0 + (2 * grp + jl) * 16 + fq;
        gv[i][jl].x = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(mrs, (okp[i] && co < p.Cpo) ? (unsigned)(mp[i] * (p.Cpo >> 3) + (co >> 3)) : XOOB, 0, 0);
      }
  }
  const XAct ak = x_act(p.act, p.slope);
  const float gate_off = p.gate_act == WCMC_ACT_RELU ? 0.f : p.gate_act == WCMC_ACT_LEAKY_RELU ? p.gate_slope : 1.f;
  const int gkind = use_gate ? 1 : use_mask ? 2 : 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the (zero) weight stages past the end have landed:
  __syncthreads();                                     // LDS is free
  // exchange: a wave parks the two cout tiles its partner (same pixels, other group) finishes, then adds the partner's to its own --
  // always (group 0) + (group 1)
  constexpr int XOFF = 36864;                          // behind the staging tile (34,816 B) and its column-sum partials
  f32x4* const xch = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem16) + XOFF);
  f32x4 fin[2][2];
  if (grp == 0) {
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int i = 0; i < 2; ++i) xch[(wave * 4 + jl * 2 + i) * 64 + lane] = acc[2 + jl][i];
  } else {
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int i = 0; i < 2; ++i) xch[(wave * 4 + jl * 2 + i) * 64 + lane] = acc[jl][i];
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int i = 0; i < 2; ++i) fin[jl][i] = acc[jl][i] + xch[((wave ^ 4) * 4 + jl * 2 + i) * 64 + lane];
  } else {
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int i = 0; i < 2; ++i) fin[jl][i] = xch[((wave ^ 4) * 4 + jl * 2 + i) * 64 + lane] + acc[2 + jl][i];
  }

  auto pix_of = [&](int pr, int& oy, int& ox) {
    oy = oy0 + (pr >> 4); ox = ox0 + (pr & 15);
    return oy < p.Ho && ox < p.Wo;
  };
  if (p.ys) {
    constexpr int OLD = 2 * BN + 8;
    u16* so = smem16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = (2 * pg + i) * 16 + col;
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int j = 2 * grp + jl;
        const int co = n0 + j * 16 + fq;
        float v[4];
        x_epi_quad(fin[jl][i], bv[jl], okp[i], ak, gkind, gv[i][jl], co, gate_off, v);
        unsigned h01, l01, h23, l23;
        x_split2(v[0], v[1], h01, l01);
        x_split2(v[2], v[3], h23, l23);
        *reinterpret_cast<uint2*>(so + pr * OLD + j * 16 + fq) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(so + pr * OLD + BN + j * 16 + fq) = make_uint2(l01, l23);
      }
    }
    __syncthreads();
    constexpr int VPP = BN / 8;
    static_assert(TPX * 2 * VPP % NTHR == 0, "every thread takes part in every pass (the mask bytes meet by DPP below)");
    // The gate mask of the tile (128 pixels x 8 bytes) leaves through LDS as ONE 8-byte store per pixel: a pixel's eight hi-plane
    // vectors sit in lanes 16 n .. 16 n + 7, four neighbouring lanes put their mask bytes into one word by two quad permutations
    // and park it; two wave instructions then store the tile's kilobyte.  (Byte stores from the lanes that hold the vectors
    // were 1,024 per tile in four more store instructions per wave -- and it is the store INSTRUCTIONS the epilogue waits
    // for: the masks cost 3 of a 64 -> 64 layer's 37 us, profiles/r06_unet_halo3.txt.)  Mask rows that are not whole 8-byte
    // groups (channel counts off a multiple of 64) keep the byte stores.
    const bool mask_lds = p.mask_out && (p.Cpo & 63) == 0 && !(DBG & 128);
    unsigned* const mstage = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem16) + XOFF);      // (the exchange area is free again)
    for (int v = tid; v < TPX * 2 * VPP; v += NTHR) {
      const int pr = v / (2 * VPP), q = v - pr * (2 * VPP);
      const int plane = q >= VPP, vec = q - plane * VPP;
      const int co = n0 + vec * 8;
      int oy, ox;
      const bool ok = pix_of(pr, oy, ox) && co < p.Cpo;
      const int64_t m = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
      const u32x4 hv = *reinterpret_cast<const u32x4*>(so + pr * OLD + plane * BN + vec * 8);
      if (ok && !(DBG & 64)) *reinterpret_cast<u32x4*>(p.ys + m * 2 * p.Cpo + plane * p.Cpo + co) = hv;      // (DBG & 64 / 128, timing only: no result stores / no gate mask)
      if (mask_lds) {
        unsigned w = (unsigned)positive_mask8(hv) << (8 * (vec & 3));
        w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
        w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xf, 0xf, false);      // quad_perm [2, 3, 0, 1]
        if (plane == 0 && (vec & 3) == 0) mstage[pr * 2 + (vec >> 2)] = w;
      } else if (ok && p.mask_out && plane == 0 && !(DBG & 128)) {
        p.mask_out[m * (p.Cpo >> 3) + (co >> 3)] = positive_mask8(hv);
      }
    }
    if (mask_lds) {
      __syncthreads();
      int oy, ox;
      if (tid < TPX && pix_of(tid, oy, ox)) {
        const int64_t m = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
        *reinterpret_cast<uint2*>(p.mask_out + m * (p.Cpo >> 3) + (n0 >> 3)) = *reinterpret_cast<const uint2*>(mstage + tid * 2);
      }
    }
    if (p.colsum) {
      constexpr int CW = 64, RG = NTHR / CW;
      float* red = reinterpret_cast<float*>(so + TPX * OLD);
      const int c = tid % CW, rg = tid / CW;
      float a = 0.f;
      for (int r = rg; r < TPX; r += RG) a += bf2f(so[r * OLD + c]) + bf2f(so[r * OLD + BN + c]);
      if (rg > 0) red[(rg - 1) * BN + c] = a;
      __syncthreads();
      if (rg == 0 && n0 + c < p.Np) {
        for (int q = 0; q < RG - 1; ++q) a += red[q * BN + c];
        p.colsum[(int64_t)tile * p.Np + n0 + c] = a;
        // trailer: the number of rows this launch wrote (the finish kernel reads no further)
        if (tile == 0 && n0 + c == 0) reinterpret_cast<int*>(p.colsum)[(int64_t)p.G * p.Np] = (int)gridDim.x;
      }
    }
  } else {
    constexpr int OLD = BN + 4;
    float* so = reinterpret_cast<float*>(smem16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = (2 * pg + i) * 16 + col;
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int j = 2 * grp + jl;
        const int co = n0 + j * 16 + fq;
        float v[4];
        x_epi_quad(fin[jl][i], bv[jl], true, ak, 0, u32x2{0u, 0u}, co, 1.f, v);
        *reinterpret_cast<float4*>(so + pr * OLD + j * 16 + fq) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
    constexpr int VPP = BN / 4;
    for (int v = tid; v < TPX * VPP; v += NTHR) {
      const int pr = v / VPP, vec = v - pr * VPP;
      const int co = n0 + vec * 4;
      int oy, ox;
      if (pix_of(pr, oy, ox) && co < p.Cpo)
        *reinterpret_cast<float4*>(p.yf + (int64_t)img * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw + co) =
            *reinterpret_cast<const float4*>(so + pr * OLD + vec * 4);
    }
  }
}


// ------------------------------------------------------------------ implicit GEMM, halo-resident, 64 pixels per wave
// Interleaved timing ablations of the kernel above (scripts/time_halo_abl.py, 175 us): without the weight stream of the
// stage loop 156, without fragment reads 160, without both 135 -- per 32-k stage a workgroup of 128 pixels moves 86 KB
// through LDS (14 KB of weights in, the same 14 KB out again to EACH of its four waves, 16 KB of pixels) for 168 MFMAs
// and waits for a weight stage that was requested only one stage earlier.  This variant halves the LDS traffic and the
// weight stream per MFMA: a wave owns FOUR pixel tiles (64 pixels x all NT*16 couts: 22 KB of fragments for 12*NT MFMAs),
// a workgroup is four waves on a 16x16 tile, and two workgroups still share a CU (independent stage barriers) because
// the channel slab is thinner: 16 channels (the last one 8 or 16), pixel stride 80 B, 32 KB for the 20x20 halo.  A 32-k
// stage is then TWO filter taps x 16 channels (four taps x 8 in an 8-channel slab): k-group kg of the lanes reads tap
// 2s + (kg >> 1); taps past ks*ks are slab padding (zero weights) and read the tile's first pixel.
// Wave w owns tile rows w, w+4, w+8, w+12 (a tile that hangs over the image edge idles every wave equally).
// PXST: the halo pixel stride as a compile-time constant (80 or 160; 0 = p.PXS, the WCMC_HALO64_PXS experiments) -- with it the
// pixel tiles of a wave sit at immediate offsets of ONE address register per stage (the kernel is launched for ks == 5 only).
// AP: planes of the pixel operand that are multiplied -- 2: W_lo*A_hi + W_hi*A_lo + W_hi*A_hi; 1: the hi plane only (W_lo*A_hi +
// W_hi*A_hi: the data gradient of the "bf16x321" mode, whose A operand is dy) -- the halo then holds no lo plane and a
// pixel's PXS bytes carry twice the channels (x_plan_k).
// WP: planes of the WEIGHTS that are multiplied -- 2: both; 1 (with AP = 1 only): W_hi*A_hi alone, ONE bf16 MFMA per product -- the
// forward of an un-gated OUTPUT layer in the "bf16x321o" mode (the KPCN chains' 100 -> 441 logits: no ReLU behind it, so the
// rounding flips no gate; profiles/r04_forward_ladder.txt, table "last").  The lo plane of the pack is neither fetched nor read.
// F16 (with AP = 1, WP = 1): the operands are ONE fp16 plane each -- x as [pixel][Cpi] halfs (wcmc_split_to_f16), the weights' hi rows
// as fp16 (pack mode 4) -- multiplied by v_mfma_f32_16x16x32_f16: 11 bits per operand instead of bf16's 8 at the same MFMA count
// (the "bf16x321h" mode's output layers; same data movement as the bf16 one-term instance, half the halo bytes per channel pair).
template <int NT, int NB, int PT = 4, int DBG = 0, int PXST = 0, int AP = 2, int WP = 2, int F16 = 0>
__global__ __launch_bounds__(256, 2) void conv_halo64_bf16x3_kernel(XIgemmParams p0) {
  static_assert(WP == 2 || AP == 1, "one weight plane only together with one pixel plane");
  static_assert(!F16 || (AP == 1 && WP == 1), "fp16 operands: one plane each");
  XIgemmParams p = p0;
  if (PXST) { p.PXS = PXST; p.ks = 5; }
  constexpr int BN = NT * 16, TH = 4 * PT, TW = 16, NTHR = 256, NWV = 4;
  constexpr int NG = (PT + 1) / 2;             // epilogue groups of two pixel tiles per wave (128 pixels of staging)
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  constexpr int B_LO = BN * XROW + 32, B_ELEMS = 2 * BN * XROW + 64;
  const int HWd = TW + p.ks - 1, HHt = TH + p.ks - 1, HP = HWd * HHt;     // (a 12-row workgroup of the PT = 4 instance keeps the 16-row layout)
  char* const halo = reinterpret_cast<char*>(smem16);
  u16* const bsm = smem16 + ((HP * p.PXS + 127) & ~127) / 2;

  // (wave as a SCALAR: the weight ring's LDS destinations, the group tests and the wait counts become scalar code -- as a
  // vector value they cost ~10 vector instructions and 4 v_readfirstlane per stage in a loop bound by vector issue)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // DBG (debug library, timing only, WRONG results): 1 no MFMA, 2 no weight DMA in the stage loop, 4 one halo per tile,
  // 8 no fragment reads, 32 no epilogue; 64 = wall-clock stamps (scripts/timeline_halo.py)
  unsigned long long st_rt[7] = {0, 0, 0, 0, 0, 0, 0};
  auto rstamp = [&](int i) {
    if (DBG & 64) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      st_rt[i] = t;
    }
  };
  rstamp(0);
  int tile;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nmain = p.tilesX * p.tilesY;
  const int tpi = nmain + (PT == 4 ? p.stripX * p.stripY : 0);
  const int img = tile / tpi, trem0 = tile - img * tpi;
  // Transposed strip (PT = 4 instance): where 16 does not divide the output WIDTH either, the launcher covers it with tilesX columns
  // of 16 (these tiles) and stripX columns of 12 whose workgroups hold their halo TRANSPOSED in LDS -- LDS row = image column, LDS
  // column = image row: 16 image rows x 12 image columns look exactly like a 12-row tile to the stage loop (same pitch, same pixel
  // tile distance, same bank pattern); only the halo fill (which pixel goes where), the LDS offset of a filter tap (dx rows, dy
  // columns) and the epilogue's pixel coordinates know.  A pixel's products are summed in the same order in either orientation.
  const bool tr = PT == 4 && trem0 >= nmain;
  const int trem = tr ? trem0 - nmain : trem0;
  // Mixed tile heights (PT = 4 instance): the launcher covers Ho EXACTLY with rows16 tile rows of 16 pixels followed by tile rows of
  // 12 where it can (100 = 4 x 16 + 3 x 12: 100 rows of MFMAs instead of 108 or 112) -- a 12-row workgroup stages a 16-row halo and
  // skips its fourth pixel tile (ptc, wave-uniform).
  const int tcols = tr ? p.stripX : p.tilesX;
  const int trow = trem / tcols;
  const int ptc = tr ? 3 : (PT == 4 && trow >= p.rows16) ? 3 : PT;
  const int oy0 = tr ? trow * 16 : PT == 4 ? (trow < p.rows16 ? trow * 16 : p.rows16 * 16 + (trow - p.rows16) * 12) : trow * TH;
  const int ox0 = tr ? p.tilesX * TW + (trem - trow * tcols) * 12 : (trem - trow * tcols) * TW;
  const int n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  const int pixb = (F16 ? 2 : 4) * p.Cpi;                // bytes per pixel of x: two bf16 planes, or one fp16 plane

  // ---- halo: [pixel][hi cs][lo cs] at stride PXS, one linear run of 16-byte vectors filled by LDS-DMA
  const int VP = p.PXS / 16;
  const int hvecs = HWd * (4 * ptc + p.ks - 1) * VP;      // (the rows this workgroup's pixel tiles reach)
  const float invVP = 1.0f / (float)VP, invHW = 1.0f / (float)HWd;
  // With the stride a template constant the per-lane source offsets of the halo's 16-byte vectors are worked out ONCE, for a
  // regular slab and for the last (narrower) one; a slab's fill is then one addition per instruction (+ slab * CS * 2, a
  // scalar).  Decoding them again for every slab cost ~25 vector instructions per vector, ~800 cycles of vector issue per
  // wave in front of the MFMAs of each slab's last stage (the "six halo reloads per tile: 4 %" of the ablations).
  constexpr int NHV = PXST ? (((TH + 4) * (TW + 4) * (PXST / 16) + 63) / 64 + NWV - 1) / NWV : 0;
  const int CSlh = p.CSl < 8 ? 8 : p.CSl;      // channels per plane the halo holds of the last slab (CSl = 4: a K order, x_last4)
  unsigned hoff[NHV ? NHV : 1], hoffl[NHV ? NHV : 1];
  if (PXST) {
#pragma unroll
    for (int kq = 0; kq < NHV; ++kq) {
      const int v = (wave + NWV * kq) * 64 + lane;
      hoff[kq] = XOOB; hoffl[kq] = XOOB;
      if (v < hvecs) {
        const int px = (int)(((float)v + 0.5f) * invVP), part = v - px * VP;     // exact: v < 2^13
        const int hy = (int)(((float)px + 0.5f) * invHW), hx = px - hy * HWd;
        const int iy = oy0 - p.pad + (tr ? hx : hy), ix = ox0 - p.pad + (tr ? hy : hx);       // (transposed: LDS row = image column)
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          const unsigned pbase = (unsigned)(((img * p.H + iy) * p.W + ix) * pixb);
          if (AP == 1) {                             // hi plane only: part = 16-byte unit of the slab's channels
            if (part < p.CS / 8 && (p.nslabs - 2) * p.CS + part * 8 < p.Cpi) hoff[kq] = pbase + (unsigned)(part * 16);
            if (part < CSlh / 8 && (p.nslabs - 1) * p.CS + part * 8 < p.Cpi) hoffl[kq] = pbase + (unsigned)(part * 16);
          } else {
          {
            const int V = p.CS / 4, plane = part >= (V >> 1), vec = part - plane * (V >> 1);
            // (channel test against the widest regular slab, nslabs - 2: it then holds for every regular slab)
            if (part < V && (p.nslabs - 2) * p.CS + vec * 8 < p.Cpi) hoff[kq] = pbase + (unsigned)(plane * 2 * p.Cpi + vec * 16);
          }
          {
            const int V = CSlh / 4, plane = part >= (V >> 1), vec = part - plane * (V >> 1);
            if (part < V && (p.nslabs - 1) * p.CS + vec * 8 < p.Cpi) hoffl[kq] = pbase + (unsigned)(plane * 2 * p.Cpi + vec * 16);
          }
          }
        }
      }
    }
  }
  auto dma_halo = [&](int slab) {
    if (PXST) {
      const unsigned so = (unsigned)(slab * p.CS * 2);
      const bool lastslab = slab == p.nslabs - 1;
#pragma unroll
      for (int kq = 0; kq < NHV; ++kq) {
        const int ii = wave + NWV * kq;
        if (ii * 64 < hvecs) {
          const unsigned off = (lastslab ? hoffl[kq] : hoff[kq]) + so;     // (invalid: 2^31 + a few hundred: out of range)
          if (ii * 64 + lane < hvecs)                                      // (the tail of the last instruction would land in the weight ring)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(halo + ii * 1024), 16, off, 0, 0, 0);
        }
      }
      return;
    }
    const int V = (slab == p.nslabs - 1 ? CSlh : p.CS) / (AP == 1 ? 8 : 4);      // data vectors per halo pixel (AP planes x cs/8)
    for (int ii = wave; ii * 64 < hvecs; ii += NWV) {
      const int v = ii * 64 + lane;
      if (v < hvecs) {
        const int px = (int)(((float)v + 0.5f) * invVP), part = v - px * VP;     // exact: v < 2^13
        const int hy = (int)(((float)px + 0.5f) * invHW), hx = px - hy * HWd;
        const int iy = oy0 - p.pad + (tr ? hx : hy), ix = ox0 - p.pad + (tr ? hy : hx);
        const int plane = AP == 1 ? 0 : part >= (V >> 1), vec = part - plane * (V >> 1);
        const int ch = slab * p.CS + vec * 8;
        unsigned off = XOOB;
        if (part < V && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && ch < p.Cpi)
          off = (unsigned)(((img * p.H + iy) * p.W + ix) * pixb + plane * 2 * p.Cpi + ch * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(halo + ii * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  // ---- weights: LDS-DMA ring of NB stages, as in the kernel above (row group = 16 cout rows x 64 B per plane)
  const int nstages = p.Kt / XKC;
  constexpr int NGMAX = (NT + NWV - 1) / NWV;
  const int ngroups = wave < NT ? (NT - wave + NWV - 1) / NWV : 0;       // wave-uniform
  unsigned dbase[NGMAX], dbase2[NGMAX];                  // hi / lo plane of this lane's 16 bytes of a stage's row group
#pragma unroll
  for (int q = 0; q < NGMAX; ++q) {
    const int drow = 16 * (wave + q * NWV) + (lane >> 2);
    const int dvq = (lane & 3) ^ ((drow >> 1) & 3);
    dbase[q] = (q < ngroups && n0 + drow < p.Np) ? (unsigned)(((n0 + drow) * 2 * p.Kt + dvq * 8) * 2) : XOOB;
    dbase2[q] = dbase[q] >= XOOB ? XOOB : dbase[q] + (unsigned)(p.Kt * 2);
  }
  auto dma_b = [&](int g, int buf) {
    // one addition per instruction: the stage's byte offset is a scalar; stages past the end add 2^30 instead, which puts
    // valid rows (< 2^30: the packed weights are a few MB) and invalid ones (2^31) alike beyond the buffer without wrapping
    const unsigned sg = g < nstages ? (unsigned)(g * XKC * 2) : 0x40000000u;
#pragma unroll
    for (int q = 0; q < NGMAX; ++q) {
      if (q < ngroups) {
        const unsigned off = dbase[q] + sg;
        const unsigned off2 = dbase2[q] + sg;
        u16* d = bsm + buf * B_ELEMS + 16 * (wave + q * NWV) * XROW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
        if (WP == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(d + B_LO), 16, off2, 0, 0, 0);
      }
    }
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < PT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragments: lane = pixel column (lane & 15) of its four tile rows, k group kg = lane >> 4 (8 k each)
  const int frow = lane & 15, kg = lane >> 4;
  const int fslot = (kg ^ ((frow >> 1) & 3)) * 8;
  const int abase0 = (wave * HWd + frow) * p.PXS, dA = NWV * HWd * p.PXS;   // tile row i of the wave: + i * dA (a constant with PXST)
  // this lane's (tap, channel) of the stage whose A fragments are read next
  int cs_cur, sps_cur, lo_off, tps, coff, tdx, tdy;
  bool cs4 = false;                                      // (wave-uniform) the slab being read is in the four-channel K order (x_last4)
  // The LDS offset of the lane's tap is carried along instead of being worked out from (tdy, tdx) every stage: one filter column is
  // sMx bytes away, one filter row sMy -- a pixel and a halo row in a regular workgroup, the other way round in a transposed one (the
  // tap (dy, dx) is then dx LDS rows and dy LDS columns away) -- so the orientation costs the stage loop nothing.
  const int sMx = tr ? HWd * p.PXS : p.PXS, sMy = tr ? p.PXS : HWd * p.PXS;
  int toff = 0, dstep = 0;                               // toff: (tdy, tdx) as bytes + coff; dstep: one stage's taps in x
  const int wrapc = sMy - p.ks * sMx;                    // ... and what a wrap into the next filter row adds
  auto slab_begin = [&](int slab) {
    cs_cur = slab == p.nslabs - 1 ? p.CSl : p.CS;
    sps_cur = slab == p.nslabs - 1 ? p.SPSl : (p.SPS & 0xff);
    cs4 = cs_cur == 4;
    lo_off = (cs4 ? 8 : cs_cur) * 2;
    tps = 32 / cs_cur;                                   // taps per stage: 1 (32 channels), 2 (16), 4 (8) or 8 (4)
    coff = cs4 ? 0 : ((kg * 8) & (cs_cur - 1)) * 2;
    tdy = 0; tdx = cs_cur == 32 ? 0 : cs_cur == 16 ? (kg >> 1) : cs4 ? 2 * kg : kg;        // (cs4: the FIRST of the lane's two taps)
    if (tdx >= p.ks) { tdx -= p.ks; ++tdy; }             // (cs4, kg = 3: tap 6)
    toff = tdy * sMy + tdx * sMx + coff;
    dstep = tps * sMx;
  };
  bf16x8 ah[PT], al[PT], wh[NT], wl[NT];
  auto a_off = [&]() { return tdy < p.ks ? toff : coff; };          // (taps past ks * ks are slab padding: zero weights, any pixel)
  // (cs4) the lane's second tap: the next one in the filter's raster order
  auto a_off2 = [&]() {
    const bool wrap = tdx + 1 >= p.ks;
    const int y = tdy + (wrap ? 1 : 0);
    return y < p.ks ? toff + sMx + (wrap ? wrapc : 0) : 0;
  };
  auto a_advance = [&]() {
    tdx += tps; toff += dstep;
    if (tdx >= p.ks) { tdx -= p.ks; ++tdy; toff += wrapc; }
    if (cs4 && tdx >= p.ks) { tdx -= p.ks; ++tdy; toff += wrapc; }      // (eight taps ahead: up to two rows of five)
  };
  auto read_a1 = [&](int i, int aoff, int aoff2) {
    const char* pa = halo + abase0 + aoff;
    if (cs4) {                                            // two taps x four channels: 8 bytes each
      const char* pb = halo + abase0 + aoff2;
      const u32x2 a0 = *reinterpret_cast<const u32x2*>(pa + i * dA), a1 = *reinterpret_cast<const u32x2*>(pb + i * dA);
      ah[i] = __builtin_bit_cast(bf16x8, u32x4{a0.x, a0.y, a1.x, a1.y});
      if (AP == 2) {
        const u32x2 l0 = *reinterpret_cast<const u32x2*>(pa + lo_off + i * dA), l1 = *reinterpret_cast<const u32x2*>(pb + lo_off + i * dA);
        al[i] = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
      }
      return;
    }
    ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * dA);
    if (AP == 2) al[i] = *reinterpret_cast<const bf16x8*>(pa + lo_off + i * dA);
  };
  const u16* const bfrag = bsm + frow * XROW + fslot;
  auto read_b = [&](int buf, int j) {
    wh[j] = *reinterpret_cast<const bf16x8*>(bfrag + buf * B_ELEMS + j * 16 * XROW);
    if (WP == 2) wl[j] = *reinterpret_cast<const bf16x8*>(bfrag + buf * B_ELEMS + B_LO + j * 16 * XROW);
  };

#pragma unroll
  for (int b = 0; b < NB; ++b) dma_b(b, b);
  dma_halo(0);
  slab_begin(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int aoff = a_off(), aoff2 = cs4 ? a_off2() : 0;
#pragma unroll
    for (int i = 0; i < PT; ++i)
      if (i < ptc) read_a1(i, aoff, aoff2);
    a_advance();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) read_b(0, j);
  int s_in = 0, slab = 0, bcur = 0;
  const int wgpar = (blockIdx.x >> 8) & 1;
  const bool prio_on = p.SPS & 0x100;          // (set by the launcher)
  rstamp(1);
  for (int g = 0; g < nstages; ++g) {
    const int b1 = bcur + 1 == NB ? 0 : bcur + 1;      // buffer of stage g+1; stage g's fragments are in registers
    if (NGMAX == 1 || ngroups < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP * (NB - 2)) : "memory");       // (WP DMA instructions per row group)
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WP * (NB - 2)) : "memory");
    pw_barrier();                                // stage g+1 has landed for everyone; everyone has read stage g's fragments
    // The two workgroups of a CU are dispatched one after the other (local block indices l and l + 32 of an XCD) and the
    // instruction arbiter prefers the older wave: stamps showed the first one through its stage loop in 114 us and the
    // second in 158, the last 40 us alone on the CU.  Alternating the priority stage by stage shares the matrix pipe.
    if (prio_on) { if ((g ^ wgpar) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    if (!(DBG & 2)) dma_b(g + NB, bcur);
    bcur = b1;
    const bool last_of_slab = (s_in + 1 == sps_cur);
    if (last_of_slab && slab + 1 < p.nslabs && !(DBG & 4)) dma_halo(slab + 1);      // (no wave reads the halo during a slab's last stage)
    const int aoff = a_off(), aoff2 = cs4 ? a_off2() : 0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        if (i >= ptc) continue;                    // (PT = 4 instance on a 12-row tile: wave-uniform)
        if (!(DBG & 1)) {
          if (WP == 2) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah[i], acc[j][i], 0, 0, 0);   // small terms first
          if (AP == 2) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al[i], acc[j][i], 0, 0, 0);
          if (F16) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xf16x8, wh[j]), __builtin_bit_cast(xf16x8, ah[i]), acc[j][i], 0, 0, 0);
          else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[i], acc[j][i], 0, 0, 0);
        }
        // the pixel tile's fragments of stage g+1 replace it as soon as its last MFMAs of this stage have issued
        if (j == NT - 1 && !last_of_slab && !(DBG & 8)) read_a1(i, aoff, aoff2);
      }
      if (!(DBG & 8)) read_b(b1, j);             // stage g+1, same cout tile, into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!last_of_slab) {
      a_advance();
      ++s_in;
    } else {                                     // slab boundary: the next A fragments come from the next halo
      s_in = 0;
      ++slab;
      slab_begin(slab < p.nslabs ? slab : p.nslabs - 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the new halo (and of the weight ring)
      __syncthreads();
      const int a2 = a_off(), a22 = cs4 ? a_off2() : 0;
#pragma unroll
      for (int i = 0; i < PT; ++i)
        if (i < ptc) read_a1(i, a2, a22);
      a_advance();
    }
  }
  __builtin_amdgcn_s_setprio(0);
  rstamp(2);
  if ((DBG & 32) && !(DBG & 64)) {                     // timing only: no epilogue (one store keeps the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < PT; ++i) keep += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    if (keep == 12345.678f && p.ys) p.ys[0] = 1;
    return;
  }
  // ---- epilogue operands (one batch of unconditional buffer loads, issued before the drain)
  const int fq = kg * 4;
  float bv[NT][4];
  {
    const __amdgpu_buffer_rsrc_t brs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.wp), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bv[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (n0 + j * 16 + fq + e) * 4, 0, 0));
  }
  const bool use_gate = p.ys && p.gate, use_mask = p.ys && !p.gate && p.gate_mask && p.gate_act != WCMC_ACT_LINEAR;
  u32x2 gv[PT][NT];                            // split gate: 4 hi-plane bf16 per quad; bit mask: one byte in .x
  bool okp[PT]; int64_t mp[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int oy = tr ? oy0 + frow : oy0 + wave + NWV * i, ox = tr ? ox0 + wave + NWV * i : ox0 + frow;
    okp[i] = i < ptc && oy < p.Ho && ox < p.Wo;
    mp[i] = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
  }
  if (use_gate) {
    const int64_t gbytes = (int64_t)p.N * p.Ho * p.Wo * 4 * p.Cpo;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate, 0, (int)(gbytes < 0x7fffffff ? gbytes : 0x7fffffff), 0x00020000);
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        gv[i][j] = __builtin_amdgcn_raw_buffer_load_b64(grs, (okp[i] && co < p.Cpo) ? (unsigned)((mp[i] * 2 * p.Cpo + co) * 2) : XOOB, 0, 0);
      }
  } else if (use_mask) {
    const int64_t mbytes = (int64_t)p.N * p.Ho * p.Wo * (p.Cpo >> 3);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate_mask, 0, (int)mbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + fq;
        gv[i][j].x = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(mrs, (okp[i] && co < p.Cpo) ? (unsigned)(mp[i] * (p.Cpo >> 3) + (co >> 3)) : XOOB, 0, 0);
      }
  }
  const XAct ak = x_act(p.act, p.slope);
  const float gate_off = p.gate_act == WCMC_ACT_RELU ? 0.f : p.gate_act == WCMC_ACT_LEAKY_RELU ? p.gate_slope : 1.f;
  const int gkind = use_gate ? 1 : use_mask ? 2 : 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the (zero) weight stages past the end have landed:
  __syncthreads();                                     // LDS is free for the epilogue staging
  rstamp(3);

  // ---- epilogue in two halves of 128 pixels (the staging tile of 256 split pixels would not fit beside a second
  // workgroup): half h = pixel tiles 2h, 2h+1 of every wave; staging row pr = 32 * wave + 16 * (i & 1) + column
  auto pix_of = [&](int h, int pr, int& oy, int& ox) {
    const int i = 2 * h + ((pr >> 4) & 1);             // (PT odd: the last group holds one pixel tile)
    if (tr) { oy = oy0 + (pr & 15); ox = ox0 + (pr >> 5) + NWV * i; }
    else { oy = oy0 + (pr >> 5) + NWV * i; ox = ox0 + (pr & 15); }
    return i < ptc && oy < p.Ho && ox < p.Wo;
  };
  if (p.ys) {
    constexpr int OLD = 2 * BN + 8;
    u16* so = smem16;
    constexpr int CW = BN <= 16 ? 16 : BN <= 32 ? 32 : BN <= 64 ? 64 : 128, RG = NTHR / CW;
    const int cc = tid % CW, rg = tid / CW;
    float csum = 0.f;
#pragma unroll
    for (int h = 0; h < NG; ++h) {
      if (h) __syncthreads();                            // the first half has left the staging tile
#pragma unroll
      for (int il = 0; il < 2; ++il) {
        const int pr = wave * 32 + il * 16 + frow;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int co = n0 + j * 16 + fq;
          unsigned h01 = 0, l01 = 0, h23 = 0, l23 = 0;
          if (2 * h + il < PT) {                           // (else: zeros, the column sums run over all 128 rows)
            constexpr int dummy = 0; (void)dummy;
            float v[4];
            const int ti = 2 * h + il < PT ? 2 * h + il : 0;
            x_epi_quad(acc[j][ti], bv[j], okp[ti], ak, gkind, gv[ti][j], co, gate_off, v);
            x_split2(v[0], v[1], h01, l01);
            x_split2(v[2], v[3], h23, l23);
          }
          *reinterpret_cast<uint2*>(so + pr * OLD + j * 16 + fq) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(so + pr * OLD + BN + j * 16 + fq) = make_uint2(l01, l23);
        }
      }
      __syncthreads();
      if (h == 0) rstamp(4);
      constexpr int VPP = BN / 8;
      for (int v = tid; v < 128 * 2 * VPP; v += NTHR) {
        const int pr = v / (2 * VPP), q = v - pr * (2 * VPP);
        const int plane = q >= VPP, vec = q - plane * VPP;
        const int co = n0 + vec * 8;
        int oy, ox;
        if (pix_of(h, pr, oy, ox) && co < p.Cpo) {
          const int64_t m = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
          const u32x4 hv = *reinterpret_cast<const u32x4*>(so + pr * OLD + plane * BN + vec * 8);
          *reinterpret_cast<u32x4*>(p.ys + m * 2 * p.Cpo + plane * p.Cpo + co) = hv;
          if (p.mask_out && plane == 0) p.mask_out[m * (p.Cpo >> 3) + (co >> 3)] = positive_mask8(hv);
        }
      }
      if (h == 0) rstamp(5);
      if (p.colsum && cc < BN)
        for (int r = rg; r < 128; r += RG) csum += bf2f(so[r * OLD + cc]) + bf2f(so[r * OLD + BN + cc]);
    }
    if (p.colsum && !(DBG & 64)) {
      __syncthreads();
      float* red = reinterpret_cast<float*>(so);
      if (rg > 0 && cc < BN) red[(rg - 1) * BN + cc] = csum;
      __syncthreads();
      if (rg == 0 && cc < BN && n0 + cc < p.Np) {
        for (int q = 0; q < RG - 1; ++q) csum += red[q * BN + cc];
        p.colsum[(int64_t)tile * p.Np + n0 + cc] = csum;
        // trailer: the number of rows this launch wrote (the finish kernel reads no further)
        if (tile == 0 && n0 + cc == 0) reinterpret_cast<int*>(p.colsum)[(int64_t)p.G * p.Np] = (int)gridDim.x;
      }
    }
  } else {
    constexpr int OLD = BN + 4;
    float* so = reinterpret_cast<float*>(smem16);
#pragma unroll
    for (int h = 0; h < NG; ++h) {
      if (h) __syncthreads();
#pragma unroll
      for (int il = 0; il < 2; ++il) {
        const int pr = wave * 32 + il * 16 + frow;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int co = n0 + j * 16 + fq;
          const f32x4 a4 = acc[j][2 * h + il < PT ? 2 * h + il : 0];
          float v[4];
          x_epi_quad(a4, bv[j], true, ak, 0, u32x2{0u, 0u}, co, 1.f, v);
          *reinterpret_cast<float4*>(so + pr * OLD + j * 16 + fq) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      __syncthreads();
      constexpr int VPP = BN / 4;
      for (int v = tid; v < 128 * VPP; v += NTHR) {
        const int pr = v / VPP, vec = v - pr * VPP;
        const int co = n0 + vec * 4;
        int oy, ox;
        if (pix_of(h, pr, oy, ox) && co < p.Cpo)
          *reinterpret_cast<float4*>(p.yf + (int64_t)img * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw + co) =
              *reinterpret_cast<const float4*>(so + pr * OLD + vec * 4);
      }
    }
  }
  if (DBG & 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the stores have left)
    rstamp(6);
    if (lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum) + ((int64_t)tile * NWV + wave) * 14;
      for (int i = 0; i < 7; ++i) o[6 + i] = st_rt[i];
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[13] = hw | ((unsigned long long)xcc << 32);
    }
  }
}

// ------------------------------------------------------------------ pointwise (1x1) GEMM, persistent
// The PathNet chains are 1x1 convolutions over B*S*H*W = 1 M pixels with 36..128 channels: 0.5-1 GB of HBM
// traffic and a few hundred MFMAs per 64 pixels -- pure streaming.  The tiled kernel above reaches 3 TB/s on
// them (a workgroup loads, multiplies, then writes; two per CU cannot keep ~40 KB per CU in flight) and
// re-reads the weights (and, for 128 couts, the input) once per tile.  Here a workgroup stays on its CU
// and walks pixel tiles (64 pixels, grid-stride): the tile's split input lands in a 3-stage LDS ring by
// LDS-DMA two tiles ahead (a pixel tile is one contiguous run of bytes: the copy is linear, with the
// 16-byte units of a pixel XOR-swizzled on the SOURCE side where the pixel stride would otherwise put all
// rows of a fragment read on the same banks); every wave owns one 16-cout tile and holds its weight fragments
// in registers for the whole launch; results go through an LDS staging tile and leave as whole 16-byte
// vectors.  All global traffic of the loop is counted buffer instructions (out-of-range = dropped), so a
// wave waits with an exact vmcnt for the tile it is about to read and never for the tiles behind it.
// U = 16-byte units per input pixel (2 planes x Cpi / 8).
// LDS stores the compiler does not see as such: behind an LDS-DMA it orders every ds_write it knows of with
// vmcnt(0) (write-after-write on LDS it cannot disambiguate).  The staging tile never overlaps the ring.
__device__ __forceinline__ void pw_lds_store_b64(unsigned addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void pw_lds_store_b128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int NTW, int U, bool SPLIT, int TAIL = 0>
__global__ __launch_bounds__(NTW * 64, (U >= 32 ? 1 : 2)) void conv_pw_bf16x3_kernel(XIgemmParams p) {
  constexpr int NW = NTW, NTHR = NW * 64, TP = 64, RT = TP / 16, BN = NTW * 16;
  constexpr int KC = U > 16 ? 4 : (U > 8 ? 2 : 1);     // 32-k steps: Kt = 128 / 64 / 32
  constexpr int HALF = U / 2;                          // units per plane
  constexpr bool SWZ = (U & 3) == 0;                   // pixel stride = 0 mod 64 B: swizzle (else u = 2 mod 4: conflict-free as is)
  constexpr int D = (U + NW - 1) / NW;                 // tile DMA instructions per wave (1 KB each)
  constexpr int AREG = D * NW * 1024;                  // tile region of a stage (data, then zeros)
  constexpr int STAGE = AREG + (SPLIT ? NW * 256 : 0); // + one gate-mask slot per wave
  constexpr int NS = 3;
  constexpr int DM = D + (SPLIT ? 1 : 0);              // vector-memory instructions per wave: fill of one stage,
  constexpr int SI = (SPLIT ? 8 : 4) + (TAIL == 1 ? 1 : TAIL == 2 ? 4 : 0);   // ... stores of one tile
  static_assert(!TAIL || (SPLIT && NTW >= 4), "the tail layer reads the split staging tile, one wave per 16 pixels");
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  char* const ring = reinterpret_cast<char*>(smem16);
  char* const stg = ring + NS * STAGE;
  const unsigned stg_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)stg);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = (int)gridDim.x, bx = (int)blockIdx.x;
  const int ntiles = (int)((p.M + TP - 1) / TP);
  const int nk = bx < ntiles ? (ntiles - bx + nb - 1) / nb : 0;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr =
      __builtin_amdgcn_make_buffer_rsrc(SPLIT ? (void*)p.ys : (void*)p.yf, 0, (int)p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mor = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask_out, 0, p.mask_out ? (int)p.m_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t gmr = __builtin_amdgcn_make_buffer_rsrc((void*)p.gate_mask, 0, p.gate_mask ? (int)p.m_bytes : 0, 0x00020000);

  // stage fill: LDS unit L = 64 * (d * NW + wave) + lane holds unit (L % U) ^ swizzle of pixel L / U
  unsigned rel[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int L = (d * NW + wave) * 64 + lane;
    const int px = L / U, pos = L - px * U;
    rel[d] = L < TP * U ? (unsigned)((px * U + (SWZ ? (pos ^ (px & 7)) : pos)) * 16) : XOOB;
  }
  int st_fill = 0;
  auto fill = [&](int k) {
    const bool live = k < nk;
    const int tile = bx + k * nb;
    const unsigned base = (unsigned)tile * (unsigned)(TP * U * 16), kill = live ? 0u : XOOB;
    char* dst = ring + st_fill * STAGE + wave * 1024;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // (a plain `unsigned`: with the type-dependent rel[d] in the argument list the host pass checks the 16-byte
      // LDS-DMA builtin at instantiation time, against the host's feature set, and silently drops the kernel stub)
      const unsigned off = (rel[d] + base) | kill;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(dst + d * NW * 1024), 16, off, 0, 0, 0);
    }
    if (SPLIT) {    // 4 mask bytes per pixel around this wave's two (couts 16 wave .. +15), pixel = lane
      const unsigned moff = (unsigned)(((int64_t)tile * TP + lane) * (BN / 8) + ((2 * wave) & ~3));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gmr, (__attribute__((address_space(3))) void*)(ring + st_fill * STAGE + AREG + wave * 256), 4,
                                               moff | kill, 0, 0, 0);
    }
    st_fill = st_fill + 1 == NS ? 0 : st_fill + 1;
  };

  // this wave's weight fragments and bias: registers for the whole launch
  const int fr = lane & 15, q = lane >> 4;
  bf16x8 wh[KC], wl[KC];
  {
    const int wrow = wave * 16 + fr;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const unsigned o = (unsigned)(((wrow * 2) * p.Kt + c * 32 + q * 8) * 2);
      wh[c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, o, 0, 0));
      wl[c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, o + (unsigned)(p.Kt * 2), 0, 0));
    }
  }
  float bs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int co = wave * 16 + q * 4 + e;
    bs[e] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
  }
  // tail layer: y2 = act2(W2 * tile + b2), a second 1x1 layer applied to the split tile while it is in LDS -- the
  // hidden activation is written once and never re-read by a second launch.  TAIL == 1: <= 4 couts (the 128 -> 3
  // output layer of PathNet.final): wave i < 4 multiplies pixel tile i by W2's only cout tile.  TAIL == 2: as many
  // couts as the first layer (64 -> 64 of the embedding chain; the 128 -> 128 data gradient behind 3 -> 128):
  // every wave multiplies the four pixel tiles by ITS cout tile, the fp32 result takes the staging tile's place.
  constexpr int KC2 = TAIL ? BN / 32 : 1;
  bf16x8 w2h[KC2], w2l[KC2];
  float bs2[4] = {0.f, 0.f, 0.f, 0.f};
  __amdgpu_buffer_rsrc_t y2r = yr;
  if (TAIL) {
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp2, 0, (int)p.wp2_bytes, 0x00020000);
    y2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.y2, 0, (int)p.y2_bytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < KC2; ++c) {
      const unsigned o = (unsigned)((((TAIL == 2 ? wave * 16 : 0) + fr) * 2 * p.Kt2 + c * 32 + q * 8) * 2);
      w2h[c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2r, o, 0, 0));
      w2l[c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2r, o + (unsigned)(p.Kt2 * 2), 0, 0));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co2 = (TAIL == 2 ? wave * 16 + q * 4 : 0) + e;
      bs2[e] = (p.bias2 && (TAIL == 2 || q == 0) && co2 < p.Cout2) ? p.bias2[co2] : 0.f;
    }
  }
  const bool is_relu = p.act == WCMC_ACT_RELU;
  const float nslope = p.act == WCMC_ACT_LEAKY_RELU ? p.slope : 1.f;
  auto actf = [&](float v) { const float neg = v * nslope; return v > 0.f ? v : (is_relu ? 0.f : neg); };   // act_apply without branches
  const bool gated = SPLIT && p.gate_mask && p.gate_act != WCMC_ACT_LINEAR;
  const float goff = p.gate_act == WCMC_ACT_LEAKY_RELU ? p.gate_slope : 0.f;
  const int axor = SWZ ? (lane & 7) : 0;
  constexpr int VPP = BN / 8;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of this thread's (plane, 8 couts) over its rows
  const int64_t HoWo = (int64_t)p.Ho * p.Wo;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight loads are not part of the counted stream
  fill(0);
  fill(1);
  int st_cur = 0;
  for (int it = 0; it < nk; ++it) {
    fill(it + 2);
    // behind tile `it`'s fill: fill(it+1), the stores of tile it-1, fill(it+2)
    if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DM) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DM + SI) : "memory");
    pw_barrier();                                   // everyone's share of the tile; staging tile is free
    const char* A = ring + st_cur * STAGE;
    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const char* a = A + (16 * i + fr) * (U * 16);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a + ((c * 4 + q) ^ axor) * 16);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(a + ((HALF + c * 4 + q) ^ axor) * 16);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[c], ah, acc[i], 0, 0, 0);   // small terms first
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c], al, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c], ah, acc[i], 0, 0, 0);
      }
    }
    const int64_t m0 = (int64_t)(bx + it * nb) * TP;
    // lane holds couts 16 wave + 4 q + {0..3} of pixel 16 i + fr
    if (SPLIT) {
      constexpr int OLD = 2 * BN + 8;
      u16* so = reinterpret_cast<u16*>(stg);
      const unsigned char* ms = reinterpret_cast<const unsigned char*>(A + AREG + wave * 256) + ((2 * wave) & 3) + (q >> 1);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int pr = 16 * i + fr;
        const bool ok = m0 + pr < p.M;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? actf(acc[i][e] + bs[e]) : 0.f;
        if (gated) {
          const unsigned bits = (unsigned)ms[pr * 4] >> (4 * (q & 1));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= ((bits >> e) & 1u) ? 1.f : goff;
        }
        u16 hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split1(v[e], hi[e], lo[e]);
        const unsigned sa = stg_lds + (unsigned)((pr * OLD + wave * 16 + q * 4) * 2);
        pw_lds_store_b64(sa, u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)});
        pw_lds_store_b64(sa + BN * 2, u32x2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)});
      }
      pw_barrier();
      // the tile's output is one contiguous run: vector v of the tile = (pixel v / 2VPP, plane, 8 couts)
      const unsigned ybase = (unsigned)(m0 * (4 * BN)), mbase = (unsigned)(m0 * VPP);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = tid + k * NTHR;
        const int pr = v / (2 * VPP), qv = v - pr * (2 * VPP);
        const int plane = qv >= VPP, vec = qv - plane * VPP;
        const u32x4 hv = *reinterpret_cast<const u32x4*>(so + pr * OLD + plane * BN + vec * 8);
        __builtin_amdgcn_raw_buffer_store_b128(hv, yr, ybase + (unsigned)(v * 16), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8(positive_mask8(hv), mor, plane ? XOOB : mbase + (unsigned)(pr * VPP + vec), 0, 0);
        if (p.colsum) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cs[2 * e] += bf2f((u16)(hv[e] & 0xffffu));
            cs[2 * e + 1] += bf2f((u16)(hv[e] >> 16));
          }
        }
      }
      if (TAIL == 1) {
        f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
        const int pr = 16 * (wave & 3) + fr;
        if (wave < 4) {
#pragma unroll
          for (int c = 0; c < KC2; ++c) {
            const bf16x8 th = *reinterpret_cast<const bf16x8*>(so + pr * OLD + c * 32 + q * 8);
            const bf16x8 tl = *reinterpret_cast<const bf16x8*>(so + pr * OLD + BN + c * 32 + q * 8);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2l[c], th, a2, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[c], tl, a2, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[c], th, a2, 0, 0, 0);
          }
        }
        // lanes q == 0 hold couts 0..3 of pixel pr: one 16-byte store per pixel (Cout2 <= 4, the view's channel pad is 4)
        const bool relu2 = p.act2 == WCMC_ACT_RELU;
        const float ns2 = p.act2 == WCMC_ACT_LEAKY_RELU ? p.slope2 : 1.f;
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = a2[e] + bs2[e], neg = t * ns2;
          const float r = e < p.Cout2 ? (t > 0.f ? t : (relu2 ? 0.f : neg)) : 0.f;
          ov[e] = __builtin_bit_cast(unsigned, r);
        }
        const int64_t m = m0 + pr;
        const int n2 = (int)(m / HoWo);
        const int r2 = (int)(m - (int64_t)n2 * HoWo);
        const int oy2 = r2 / p.Wo, ox2 = r2 - oy2 * p.Wo;
        const int64_t off2 = ((int64_t)n2 * p.y2sn + (int64_t)oy2 * p.y2sh + (int64_t)ox2 * p.y2sw) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(ov, y2r, (wave < 4 && q == 0 && m < p.M) ? (unsigned)off2 : XOOB, 0, 0);
      }
      if (TAIL == 2) {
        f32x4 a2[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          a2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          const u16* t0 = so + (16 * i + fr) * OLD + q * 8;
#pragma unroll
          for (int c = 0; c < KC2; ++c) {
            const bf16x8 th = *reinterpret_cast<const bf16x8*>(t0 + c * 32);
            const bf16x8 tl = *reinterpret_cast<const bf16x8*>(t0 + BN + c * 32);
            a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2l[c], th, a2[i], 0, 0, 0);
            a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[c], tl, a2[i], 0, 0, 0);
            a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[c], th, a2[i], 0, 0, 0);
          }
        }
        pw_barrier();                                    // everyone is done with the split tile (stores and fragments)
        constexpr int OLF = BN + 4;                      // floats per pixel row: same bytes as the split tile
        const bool relu2 = p.act2 == WCMC_ACT_RELU;
        const float ns2 = p.act2 == WCMC_ACT_LEAKY_RELU ? p.slope2 : 1.f;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          u32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = a2[i][e] + bs2[e], neg = t * ns2;
            ov[e] = __builtin_bit_cast(unsigned, t > 0.f ? t : (relu2 ? 0.f : neg));
          }
          pw_lds_store_b128(stg_lds + (unsigned)(((16 * i + fr) * OLF + wave * 16 + q * 4) * 4), ov);
        }
        pw_barrier();
        const float* sf = reinterpret_cast<const float*>(stg);
        constexpr int VF = BN / 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = tid + k * NTHR;
          const int pr = v / VF, vec = v - pr * VF;
          const int64_t m = m0 + pr;
          const int n2 = (int)(m / HoWo);
          const int r2 = (int)(m - (int64_t)n2 * HoWo);
          const int oy2 = r2 / p.Wo, ox2 = r2 - oy2 * p.Wo;
          const int64_t off2 = ((int64_t)n2 * p.y2sn + (int64_t)oy2 * p.y2sh + (int64_t)ox2 * p.y2sw + vec * 4) * 4;
          const u32x4 hv = *reinterpret_cast<const u32x4*>(sf + pr * OLF + vec * 4);
          __builtin_amdgcn_raw_buffer_store_b128(hv, y2r, m < p.M ? (unsigned)off2 : XOOB, 0, 0);
        }
      }
    } else {
      constexpr int OLD = BN + 4;
      float* so = reinterpret_cast<float*>(stg);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int pr = 16 * i + fr;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = actf(acc[i][e] + bs[e]);
        pw_lds_store_b128(stg_lds + (unsigned)((pr * OLD + wave * 16 + q * 4) * 4),
                          u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])});
      }
      pw_barrier();
      constexpr int VF = BN / 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = tid + k * NTHR;
        const int pr = v / VF, vec = v - pr * VF;
        const int64_t m = m0 + pr;
        const int n = (int)(m / HoWo);
        const int r = (int)(m - (int64_t)n * HoWo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        const int64_t off = ((int64_t)n * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw + vec * 4) * 4;
        const u32x4 hv = *reinterpret_cast<const u32x4*>(so + pr * OLD + vec * 4);
        __builtin_amdgcn_raw_buffer_store_b128(hv, yr, m < p.M ? (unsigned)off : XOOB, 0, 0);
      }
    }
    st_cur = st_cur + 1 == NS ? 0 : st_cur + 1;
  }

  if (SPLIT && p.colsum) {
    // one row of partial column sums per workgroup (hi + lo planes, 16 row groups combined in a fixed order)
    pw_barrier();
    float* red = reinterpret_cast<float*>(stg);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];      // tid = group * 2VPP + (plane * VPP + vec)
    pw_barrier();
    if (tid < BN) {
      const int vec = tid >> 3, e = tid & 7;
      float a = 0.f;
      for (int g = 0; g < NTHR / (2 * VPP); ++g)
        a += red[(g * 2 * VPP + vec) * 8 + e] + red[(g * 2 * VPP + VPP + vec) * 8 + e];
      p.colsum[(int64_t)bx * p.Np + tid] = a;
      if (bx == 0 && tid == 0) reinterpret_cast<int*>(p.colsum)[(int64_t)p.G * p.Np] = nb;   // trailer: rows written
    }
  }
}

template <int NTW, int U, bool SPLIT, int TAIL = 0>
static int launch_xpw2(const XIgemmParams& p, hipStream_t stream) {
  constexpr int NW = NTW, BN = NTW * 16, D = (U + NW - 1) / NW;
  constexpr size_t stage = (size_t)D * NW * 1024 + (SPLIT ? NW * 256 : 0);
  constexpr size_t stg = SPLIT ? (size_t)64 * (2 * BN + 8) * sizeof(u16) : (size_t)64 * (BN + 4) * sizeof(float);
  constexpr size_t red = SPLIT ? (size_t)NW * 64 * 8 * sizeof(float) : 0;
  constexpr size_t lds = 3 * stage + (stg > red ? stg : red);
  static_assert(lds <= 160 * 1024, "LDS");
  static int cus = 0;
  static LdsAttr attr_set;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_pw_bf16x3_kernel<NTW, U, SPLIT, TAIL>), lds, attr_set) != hipSuccess) return WCMC_ERR_LAUNCH;
  if (cus == 0) {                     // (one node holds one kind of GPU: the CU count is read once)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int64_t ntiles = ceil_div64(p.M, 64);
  int64_t nb = (int64_t)cus * (U >= 32 ? 1 : 2);
  if (nb > ntiles) nb = ntiles;
  if (p.colsum && nb > p.G) nb = p.G;
  hipLaunchKernelGGL((conv_pw_bf16x3_kernel<NTW, U, SPLIT, TAIL>), dim3((unsigned)nb), dim3(NW * 64), lds, stream, p);
  return check_launch("conv2d_igemm_bf16x3(pointwise)");
}

// 1x1, no padding, the channel counts of the PathNet chains; anything else stays on the tiled kernel
static bool x_plan_pw(const XIgemmParams& p, int* ntw, int* u) {
  if (p.ks != 1 || p.pad != 0 || p.gate || p.PXS) return false;
  const char* e = ab_env("WCMC_IGEMM_PW");      // read per call: the parity tests switch kernels inside one process
  if (e && e[0] == '0') return false;
  const int U = p.Cpi / 4;
  if (p.Np == 64 && (U == 16 || U == 10)) *ntw = 4;
  else if (p.Np == 128 && (U == 32 || U == 2)) *ntw = 8;
  else return false;
  *u = U;
  if (p.Cout != p.Np || p.Kt != (U > 16 ? 128 : U > 8 ? 64 : 32)) return false;
  if (p.M * 4 * p.Np >= 0x7ff00000LL) return false;
  if (p.yf) {
    const int64_t ext = ((int64_t)(p.N - 1) * p.ysn + (int64_t)(p.Ho - 1) * p.ysh + (int64_t)(p.Wo - 1) * p.ysw + p.Cpo) * 4;
    if (p.ysn < 0 || p.ysh < 0 || p.ysw < 0 || ext >= 0x7ff00000LL) return false;
  }
  return true;
}
static int launch_xpw(XIgemmParams& p, int ntw, int u, hipStream_t st) {
  if (p.ys) {
    p.y_bytes = (unsigned)(p.M * 4 * p.Np);
    p.m_bytes = (unsigned)(p.M * (p.Np / 8));
    if (ntw == 4) return u == 16 ? launch_xpw2<4, 16, true>(p, st) : launch_xpw2<4, 10, true>(p, st);
    return u == 32 ? launch_xpw2<8, 32, true>(p, st) : launch_xpw2<8, 2, true>(p, st);
  }
  p.y_bytes = (unsigned)(((int64_t)(p.N - 1) * p.ysn + (int64_t)(p.Ho - 1) * p.ysh + (int64_t)(p.Wo - 1) * p.ysw + p.Cpo) * 4);
  p.m_bytes = 0;
  if (ntw == 4) return u == 16 ? launch_xpw2<4, 16, false>(p, st) : launch_xpw2<4, 10, false>(p, st);
  return u == 32 ? launch_xpw2<8, 32, false>(p, st) : launch_xpw2<8, 2, false>(p, st);
}


// ------------------------------------------------------------------ weight gradient
// D[co][ci] (per tap) = sum_pix dy[pix][co] * x[pix+tap][ci]; both operands are read with the
// transposing LDS load (ds_read_b64_tr_b16): the tiles sit in LDS as [pixel][channel] exactly as
// they come from HBM, and a lane receives 4 consecutive PIXELS (= MFMA k) of its channel column.
// Block = 64-pixel stage x (TM*16 couts) x 64 cins; waves: 2 (pixel halves = MFMA k-steps) x 2 (cin halves).
// PMC profile of the first version: 36 % L2 hit rate and 2.7 GB fetched per launch -- the 50 blocks
// that share a pixel range (25 taps x 2 cin blocks) ran on different XCDs at different times.  The
// 1-D grid is therefore remapped so that one XCD runs the (tap, tile) blocks of a pixel split back to
// back (speed only), rows of the LDS tiles are an odd multiple of 32 B and the k -> pixel assignment
// of the transposing reads is {4g..4g+3, 16+4g..16+4g+3} (conflict-free, identical for both operands).
struct XWgradParams {
  const u16* x; int N, H, W, Cin, Cpi;
  const u16* dy; int Ho, Wo, Cout, Cpo;
  int ks, pad;
  float* slabs; int S; int64_t M, pix_per_split;
  int Np, Cq, coBlocks, ciBlocks;
  unsigned x_bytes, dy_bytes;
  int xps, yps;                     // pixel stride (bytes) of x / dy: 4 * Cp for a split tensor, 2 * Cp for a single bf16 plane
};

constexpr int xw_stride(int ch) { return ((ch / 16) & 1) ? ch : ch + 16; }   // bf16 elements; bytes = odd * 32

// PL = planes multiplied: 2 = hi + lo of both operands, three MFMAs per product (yl*xh + yh*xl + yh*xh); 1 = the hi planes
// only, ONE MFMA per product (the round-3 precision ladder, profiles/r03_precision_ladder.txt: rounding dy and x to bf16 is
// independent from pixel to pixel and averages out over the pixel sum -- the gradients of the benchmarked step move from
// 1.09e-3 to 1.14e-3 of the fp32 oracle's in relative L2).  Half the stage bytes, half the fragment reads, a third of the MFMAs.
template <int TM, int PL = 2>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16x3_kernel(XWgradParams p) {
  constexpr int PK = 64;
  constexpr int YC = TM * 16, XC = 64;
  constexpr int SA = xw_stride(YC), SB = xw_stride(XC);
  constexpr int YV = YC / 8, XV = XC / 8;          // 16-byte vectors per plane per pixel
  constexpr int TOTV = PL * YV + PL * XV;
  constexpr int NV = (TOTV + 3) / 4;               // vectors per thread (4 threads share a pixel)
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  u16* Ys = smem16;                        // [PL][PK][SA]
  u16* Xs = smem16 + PL * PK * SA;         // [PL][PK][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: wave-uniform tests and LDS-DMA destinations stay scalar code)
  // block -> (split, tap, tile): XCD x (= blockIdx & 7) owns splits s = x, x+8, ...; its consecutive
  // blocks sweep the taps and tiles of one split.
  const int taps = p.ks * p.ks;
  const int per_split = taps * p.coBlocks * p.ciBlocks;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int s = (local / per_split) * 8 + xcd;
  if (s >= p.S) return;
  const int within = local - (local / per_split) * per_split;
  const int tap = within % taps, tileid = within / taps;
  const int cob = tileid / p.ciBlocks, cib = tileid - cob * p.ciBlocks;
  const int co0 = cob * YC, ci0 = cib * XC;
  const int tdy = tap / p.ks - p.pad, tdx = tap % p.ks - p.pad;
  // waves: 2 (pixel halves = MFMA k-steps) x 2 (cout halves); every wave covers the 4 cin tiles, so a
  // stage costs it 8 + 2*MT transposing fragment loads for 12*MT MFMAs (was 36 for 42).
  constexpr int MT = (TM + 1) / 2;                 // cout tiles per wave (the second half may hold one less)
  const int wk = wave >> 1, wm = wave & 1;
  const int tm_valid = min(MT, max(0, min(TM, (p.Np - co0) / 16) - wm * MT));
  const int tn_valid = min(4, max(0, (p.Cq - ci0) / 16));
  const int64_t pstart = (int64_t)s * p.pix_per_split;
  const int64_t pend = min(p.M, pstart + p.pix_per_split);
  const int nstages = (int)((pend - pstart + PK - 1) / PK);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);

  // loader: thread -> pixel tid/4 of the stage, vectors (tid&3) + 4*j; per-vector constant parts
  const int lpx = tid >> 2, lv0 = tid & 3;
  unsigned voff[NV]; int lds_off[NV]; bool isy[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lv0 + 4 * j;
    if (v < PL * YV) {
      const int plane = v >= YV, vec = v - plane * YV;
      const int co = co0 + vec * 8;
      isy[j] = true;
      voff[j] = co < p.Cpo ? (unsigned)((plane * p.Cpo + co) * 2) : XOOB;
      lds_off[j] = (plane * PK + lpx) * SA + vec * 8;
    } else if (v < TOTV) {
      const int u = v - PL * YV;
      const int plane = u >= XV, vec = u - plane * XV;
      const int ci = ci0 + vec * 8;
      isy[j] = false;
      voff[j] = ci < p.Cpi ? (unsigned)((plane * p.Cpi + ci) * 2) : XOOB;
      lds_off[j] = PL * PK * SA + (plane * PK + lpx) * SB + vec * 8;
    } else {                               // (TOTV not a multiple of 4: this thread has one vector less)
      isy[j] = true; voff[j] = XOOB; lds_off[j] = -1;
    }
  }
  int cn, coy, cox; int64_t cp = pstart + lpx;
  {
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    cn = (int)(cp / hw);
    const int r = (int)(cp - (int64_t)cn * hw);
    coy = r / p.Wo; cox = r - coy * p.Wo;
  }
  u32x4 rv[NV];
  auto load_stage = [&]() {
    const bool pv = cp < pend;
    const int iy = coy + tdy, ix = cox + tdx;
    const bool xv = pv && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const unsigned yb = pv ? (unsigned)((((int64_t)cn * p.Ho + coy) * p.Wo + cox) * p.yps) : XOOB;
    const unsigned xb = xv ? (unsigned)((((int64_t)cn * p.H + iy) * p.W + ix) * p.xps) : XOOB;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      rv[j] = isy[j] ? __builtin_amdgcn_raw_buffer_load_b128(yr, (yb | voff[j]) >= XOOB ? XOOB : yb + voff[j], 0, 0)
                     : __builtin_amdgcn_raw_buffer_load_b128(xr, (xb | voff[j]) >= XOOB ? XOOB : xb + voff[j], 0, 0);
    cp += PK; cox += PK;
    while (cox >= p.Wo) { cox -= p.Wo; if (++coy == p.Ho) { coy = 0; ++cn; } }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (TOTV % 4 == 0 || lds_off[j] >= 0) *reinterpret_cast<u32x4*>(smem16 + lds_off[j]) = rv[j];
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing read: lane (group g = lane>>4, i = lane&15, q = i>>2, pp = i&3) addresses pixel row
  // 4g + q (first read) / 16 + 4g + q (second read) and channels 4pp..4pp+3 of a 16-channel tile;
  // it receives channel i of those 4 pixels.  Both MFMA operands use the same pixel order.
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int prow0 = wk * 32 + 4 * g + tq;
  auto tr_read = [&](const u16* base, int stride, int col0, bf16x8& out) {
    const u16* a0 = base + prow0 * stride + col0 + 4 * tp;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(a0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(a0 + 16 * stride));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 cat = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    out = __builtin_bit_cast(bf16x8, cat);
  };

  if (nstages > 0) load_stage();
  for (int st = 0; st < nstages; ++st) {
    __syncthreads();                 // every wave is done reading the previous stage
    store_stage();
    __syncthreads();
    if (st + 1 < nstages) load_stage();
    bf16x8 xh[4], xl[PL == 2 ? 4 : 1], yh[MT], yl[PL == 2 ? MT : 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tr_read(Xs, SB, j * 16, xh[j]);
      if constexpr (PL == 2) tr_read(Xs + PK * SB, SB, j * 16, xl[j]);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      tr_read(Ys, SA, (wm * MT + i) * 16, yh[i]);            // (a tile past TM reads the X region: unused)
      if constexpr (PL == 2) tr_read(Ys + PK * SA, SA, (wm * MT + i) * 16, yl[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < tm_valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < tn_valid) {
            if constexpr (PL == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl[i], xh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh[i], xl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh[i], xh[j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- ordered sum of the two k-halves through LDS, then one coalesced slab write
  constexpr int RS = XC + 4;
  float* red = reinterpret_cast<float*>(smem16);         // [YC][RS] floats
  const int fcol = lane & 15, fq = (lane >> 4) * 4;
  for (int h = 0; h < 2; ++h) {
    if (wk == h) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (wm * MT + i < TM) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* q = red + ((wm * MT + i) * 16 + fq + r) * RS + j * 16 + fcol;
              *q = (h == 0 ? 0.f : *q) + acc[i][j][r];
            }
        }
      }
    }
    __syncthreads();
  }
  float* slab = p.slabs + ((int64_t)s * taps + tap) * p.Np * p.Cq;
  for (int idx = tid; idx < YC * (XC / 4); idx += 256) {
    const int r = idx / (XC / 4), c = (idx - r * (XC / 4)) * 4;
    if (co0 + r < p.Np && ci0 + c < p.Cq)
      *reinterpret_cast<float4*>(slab + (int64_t)(co0 + r) * p.Cq + ci0 + c) =
          *reinterpret_cast<const float4*>(red + r * RS + c);
  }
}


// ------------------------------------------------------------------ weight gradient, one filter row per block
// The kernel above runs one tap per block: every 64-pixel stage (44 KB of dy and x) feeds 168 MFMAs, i.e.
// 65 B per clock and CU from L2 -- the load path, not the matrix pipe, sets its pace (120-150 TF/s).
// Here a block owns a whole filter ROW (KS taps) of one 112-cout block and keeps all KS x 7 x 7
// accumulator tiles in registers (wave w = cin tile w: KS x 7 tiles = 140 VGPRs at KS = 5): a stage is
// 64 pixels of one output row, dy [64][112] and the x row segment [64 + KS - 1][112] (both planes),
// and feeds 2 x KS x 49 x 3 = 1470 MFMAs -- 8.6 B per clock.  The KS taps read the same x rows at shifted
// pixel offsets (the transposing LDS read addresses pixel rows per lane, so any shift is free), dy
// fragments are shared by all taps.  Stages are filled by LDS-DMA (buffer_load ... lds, no staging
// registers) into two buffers; one barrier per stage of ~3400 MFMA cycles per wave.
struct XWRowsParams {
  const u16* x; int N, H, W, Cpi;
  const u16* dy; int Ho, Wo, Cpo;
  int pad;
  float* slabs; int S, rps, R;
  float* dbg;                       // clock-probe build only
  int prio;                         // rows8: iteration (of 14 per stage) at which waves 0-3 hand the priority to waves 4-7; 0 = off
  int Np, Cq, coBlocks, ciBlocks;
  unsigned x_bytes, dy_bytes;
  int xps, yps;                     // pixel stride (bytes) of x / dy: 4 * Cp for a split tensor, 2 * Cp for a single bf16 plane
};

// LDS row stride (u16) of a CH-channel tile: bytes = odd multiple of 32 (conflict-free transposing reads);
// the pad vectors of a row are filled by DMA lanes with an out-of-range source (zeros).
constexpr int xwr_stride(int ch) { return ((ch / 16) | 1) * 16; }

// KS = filter size, TM = cout tiles (16) per block, NW = waves = cin tiles per block.
// Transposing LDS reads the compiler does not see as LDS reads.  Behind an LDS-DMA hipcc orders every LDS read it knows of
// with s_waitcnt vmcnt(0) (it cannot tell the stage being filled from the stage being read inside one dynamic array): the
// first version of the kernel below therefore waited for stage st+1 to LAND before it multiplied stage st -- no overlap of
// the fill with the MFMAs at all.  The pair (rows prow, prow + 16 of one 16-channel tile) is issued without a wait;
// xwr_frag() orders it (lgkmcnt) and assembles the MFMA operand -- any register copy the compiler adds sits behind the wait.
struct XwrRaw { u32x2 a, b; };
template <int OFF2>
__device__ __forceinline__ void xwr_tr_issue(unsigned addr, XwrRaw& r) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(r.a), "=&v"(r.b) : "v"(addr), "n"(OFF2));
}
__device__ __forceinline__ bf16x8 xwr_cat(const XwrRaw& r) {
  const u32x4 c = {r.a[0], r.a[1], r.b[0], r.b[1]};
  return __builtin_bit_cast(bf16x8, c);
}

template <int OFF1, int OFF2>
__device__ __forceinline__ void xwr_tr_issue_at(unsigned addr, XwrRaw& r) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(r.a), "=&v"(r.b) : "v"(addr), "n"(OFF1), "n"(OFF2));
}

// loops whose index must be a constant expression (immediate offsets of the transposing reads: an address that is a register
// plus a constant costs a vector addition per read as a plain unrolled loop, and these kernels are bound by vector issue)
template <class F, int... I>
__device__ __forceinline__ void xstatic_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void xstatic_for(F&& f) { xstatic_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// PL: planes multiplied (see conv_wgrad_bf16x3_kernel): 2 = [Yh | Yl | Xh | Xl] stages, three MFMAs per product; 1 = [Yh | Xh], one.
template <int KS, int TM, int NW, int DBG = 0, int PL = 2>
__global__ __launch_bounds__(NW * 64, (NW <= 4 ? 2 : 1)) void conv_wgrad_rows_bf16x3_kernel(XWRowsParams p) {
  constexpr int CHY = TM * 16, CHX = NW * 16, PK = 64, XR = PK + KS - 1;
  constexpr int SY = xwr_stride(CHY), SX = xwr_stride(CHX);
  constexpr int VY = SY / 8, VX = SX / 8;                   // 16-byte vectors per row and plane (with pad)
  constexpr int YV = PK * VY, XV = XR * VX;
  constexpr int NVEC = PL * YV + PL * XV;
  constexpr int NI = (NVEC + NW * 64 - 1) / (NW * 64);      // LDS-DMA instructions per wave and stage
  constexpr int BUF = NI * NW * 64 * 8;                     // u16 per buffer (whole instructions)
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // unit = (split, cout block, cin block); its KS filter-row blocks run side by side on one XCD
  // (blockIdx & 7) and share the unit's dy rows and x rows in that XCD's L2.  The plan keeps the units of
  // an XCD within its 32 CUs: one round.
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int unit = (local / KS) * 8 + xcd;
  const int upb = p.coBlocks * p.ciBlocks;
  if (unit >= p.S * upb) return;
  const int trow = local % KS;
  const int s = unit / upb, ub = unit - s * upb;
  const int cob = ub / p.ciBlocks, cib = ub - cob * p.ciBlocks;
  const int co0 = cob * CHY, ci0 = cib * CHX;
  const int r0 = s * p.rps, r1 = min(p.R, r0 + p.rps);
  const int nch = (p.Wo + PK - 1) / PK;
  const int nrows = r1 - r0;
  const int nst = nrows * nch;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);

  // ---- stage fill: the buffer is one linear run of 16-byte vectors [Yh | Yl | Xh | Xl];
  // instruction i of wave w writes vectors (i*NW + w)*64 + lane (lane-linear destination), the per-lane
  // SOURCE picks the pixel / plane / channel; invalid sources use an out-of-range offset and land as zeros.
  // What a lane fetches for instruction i is the same in every stage up to the stage's base address and
  // edge tests: one packed word per instruction -- bits 0..19 byte offset / 2 relative to the stage's first
  // pixel, 20..26 pixel row of the tile, 27 operand (1 = x), 28 never valid (row pad, tail of the buffer).
  // What a lane fetches for instruction i is the same in every stage up to the stage's base address and its edge tests:
  // relv = byte offset relative to the stage's first pixel, rowv = pixel row of the tile (127: never valid -- row pad,
  // tail of the buffer, channel past the tensor).
  unsigned relv[NI]; int rowv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int v = (i * NW + wave) * 64 + lane;
    unsigned rel = 0; int rw = 127;
    if (v < PL * YV) {
      const int plane = v >= YV, vv = v - plane * YV;
      const int row = vv / VY, vec = vv - row * VY;
      const int co = co0 + vec * 8;
      if (vec * 8 < CHY && co < p.Cpo) { rel = (unsigned)(row * p.yps + plane * 2 * p.Cpo + co * 2); rw = row; }
    } else if (v < NVEC) {
      const int u = v - PL * YV;
      const int plane = u >= XV, uu = u - plane * XV;
      const int row = uu / VX, vec = uu - row * VX;
      const int ci = ci0 + vec * 8;
      if (vec * 8 < CHX && ci < p.Cpi) { rel = (unsigned)(row * p.xps + plane * 2 * p.Cpi + ci * 2); rw = row; }
    }
    relv[i] = rel; rowv[i] = rw;
  }
  // per-stage scalars of the fill (issue_prep) and one DMA instruction of it (issue_one, four vector instructions and no
  // branch): the instructions of stage st+1 are spread over the MFMA stream of stage st (stamps of the first version,
  // which issued them in one burst after the barrier: 1500-1950 of 10500 cycles per stage, the matrix pipe idle)
  unsigned f_ybase = 0, f_xbase = 0, f_yn = 0, f_xn = 0; int f_xlo = 0, f_buf = 0;
  // Row order skewed by the filter row: at step j the KS blocks of a unit read the SAME x row r0 + j and dy rows one
  // step apart -- stage st is chunk st % nch of row r0 + (st / nch - trow) mod nrows.  The stages are prepared in order,
  // so the cursor advances by increments (the divisions of the first version cost 500-850 cycles per stage).
  int f_c = 0, f_rs = nrows > 0 ? (nrows - trow % nrows) % nrows : 0, f_n, f_oy;
  const int f_n0 = r0 / p.Ho, f_oy0 = r0 - f_n0 * p.Ho;
  { const int r = r0 + f_rs; f_n = r / p.Ho; f_oy = r - f_n * p.Ho; }
  auto issue_prep = [&](int buf) {
    const int ox0 = f_c * PK;
    const int iy = f_oy + trow - p.pad;
    const bool rowok = (unsigned)iy < (unsigned)p.H;
    f_ybase = (unsigned)(((f_n * p.Ho + f_oy) * p.Wo + ox0) * p.yps);
    f_xbase = (unsigned)(((f_n * p.H + iy) * p.W + ox0 - p.pad) * p.xps);       // may wrap: only used when valid
    f_yn = (unsigned)max(0, p.Wo - ox0);               // dy rows [0, yn) exist
    f_xlo = p.pad - ox0;                               // x rows [xlo, xlo + xn) are inside the image
    f_xn = rowok ? (unsigned)p.W : 0u;
    f_buf = buf;
    if (++f_c == nch) {                                // the cursor of the following stage
      f_c = 0;
      if (++f_rs == nrows) { f_rs = 0; f_n = f_n0; f_oy = f_oy0; }
      else if (++f_oy == p.Ho) { f_oy = 0; ++f_n; }
    }
  };
  auto issue_one = [&](int i) {
    if ((i + 1) * NW * 64 > NVEC && (i * NW + wave) * 64 >= NVEC) return;   // (the tail of the last instruction row: nothing to fetch)
    // PL*YV is a multiple of 64: a wave-instruction is all dy or all x (wave-uniform choice of descriptor and base)
    const bool isx = (PL * VY) % NW == 0 ? i >= (PL * VY) / NW : (i * NW + wave) * 64 >= PL * YV;
    const unsigned base = isx ? f_xbase : f_ybase, cnt = isx ? f_xn : f_yn;
    const int lo = isx ? f_xlo : 0;
    const unsigned off = (unsigned)(rowv[i] - lo) < cnt ? base + relv[i] : XOOB;
    __attribute__((address_space(3))) void* dst =
        (__attribute__((address_space(3))) void*)(smem16 + f_buf * BUF + (i * NW + wave) * 512);
    if (!isx) __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, dst, 16, off, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, off, 0, 0, 0);
  };
  auto issue = [&](int buf) {
    issue_prep(buf);
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_one(i);
  };

  f32x4 acc[KS][TM];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing read (see conv_wgrad_bf16x3_kernel): lane addresses pixel row 4g + q (+16) and channels
  // 4pp..4pp+3 of a 16-channel tile and receives channel (lane & 15) of pixels {4g..4g+3, 16+4g..16+4g+3}
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) u16*)smem16);

  unsigned long long tc0 = 0, tr0 = 0;
  if (DBG & 4) { tc0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
  // DBG & 16 (scripts/timeline_wgrad.py): wall-clock stamps (100 MHz) of entry / loop start / loop end / exit and the
  // shader-clock cycles of the stage loop spent waiting (DMA + barrier), issuing the next stage and multiplying
  unsigned long long rt[4] = {0, 0, 0, 0}, cyc[3] = {0, 0, 0}, tprev = 0;
  auto rts = [&](int i) {
    if (DBG & 16) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      rt[i] = t;
    }
  };
  auto cst = [&](int i) {
    if (DBG & 16) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 0) cyc[i] += t - tprev;
      tprev = t;
    }
  };
  rts(0);
  if (nst > 0) issue(0);
  rts(1);
  cst(-1);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of stage st has landed
    __syncthreads();                                      // ... everyone's; and everyone is done with stage st-1
    cst(0);
    const bool fill = st + 1 < nst && !((DBG & 2) && st > 0);
    cst(1);
    // byte addresses of this lane's first fragment row in the four planes of the stage
    const int prow0 = 4 * g + tq;
    const unsigned aYh = lds0 + (unsigned)(((st & 1) * BUF + prow0 * SY + 4 * tp) * 2);
    const unsigned aXh = lds0 + (unsigned)(((st & 1) * BUF + PL * PK * SY + prow0 * SX + wave * 16 + 4 * tp) * 2);
    const int c = st % nch;
    const int nk = (min(PK, p.Wo - c * PK) + 31) / 32;    // 32-pixel MFMA k-steps with any valid pixel (1 or 2)
    // Software pipeline inside the wave (the first version read a k-step's fragments, waited, multiplied: a wave alone on
    // its SIMD kept the matrix pipe 61 % busy): the x fragments of all KS taps stay in registers for a k-step and are
    // replaced tap by tap during its last cout tile; the dy fragments are double-buffered one cout tile ahead.  The
    // order of the MFMAs on every accumulator is unchanged (bit-identical results).
    XwrRaw rxh[KS], rxl[KS], ryh[2], ryl[2];
    constexpr int XLOB = XR * SX * 2, YLOB = PK * SY * 2;   // lo planes; every read below = aXh / aYh + an immediate
    xstatic_for<KS>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      xwr_tr_issue_at<t * SX * 2, t * SX * 2 + 16 * SX * 2>(aXh, rxh[t]);
      if constexpr (PL == 2) xwr_tr_issue_at<XLOB + t * SX * 2, XLOB + t * SX * 2 + 16 * SX * 2>(aXh, rxl[t]);
    });
    xwr_tr_issue_at<0, 16 * SY * 2>(aYh, ryh[0]);
    if constexpr (PL == 2) xwr_tr_issue_at<YLOB, YLOB + 16 * SY * 2>(aYh, ryl[0]);
    bf16x8 xh[KS], xl[KS];
    xstatic_for<2>([&](auto K_) {
      constexpr int kk = decltype(K_)::value;
      if (kk < nk) {
        xstatic_for<TM>([&](auto I_) {
          constexpr int i = decltype(I_)::value;
          constexpr int cur = (kk * TM + i) & 1, nxt = cur ^ 1;
          // everything issued so far has landed (the reads of this iteration were issued one iteration ago)
          if constexpr (PL == 1) {                     // (the lo registers do not exist in this instance)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ryh[cur].a), "+v"(ryh[cur].b));
            if (i == 0) {
#pragma unroll
              for (int t = 0; t < KS; ++t) {
                asm volatile("" : "+v"(rxh[t].a), "+v"(rxh[t].b));
                xh[t] = xwr_cat(rxh[t]);
              }
            }
          } else if (DBG & 8) {                        // (timing only: no wait for the fragments)
            asm volatile("" : "+v"(ryh[cur].a), "+v"(ryh[cur].b), "+v"(ryl[cur].a), "+v"(ryl[cur].b));
            if (i == 0) {
#pragma unroll
              for (int t = 0; t < KS; ++t) { xh[t] = xwr_cat(rxh[t]); xl[t] = xwr_cat(rxl[t]); }
            }
          } else if (i == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ryh[cur].a), "+v"(ryh[cur].b), "+v"(ryl[cur].a), "+v"(ryl[cur].b));
#pragma unroll
            for (int t = 0; t < KS; ++t) {
              asm volatile("" : "+v"(rxh[t].a), "+v"(rxh[t].b), "+v"(rxl[t].a), "+v"(rxl[t].b));
              xh[t] = xwr_cat(rxh[t]); xl[t] = xwr_cat(rxl[t]);
            }
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ryh[cur].a), "+v"(ryh[cur].b), "+v"(ryl[cur].a), "+v"(ryl[cur].b));
          }
          bf16x8 yh = xwr_cat(ryh[cur]), yl = yh;
          if constexpr (PL == 2) yl = xwr_cat(ryl[cur]);
          if constexpr (i + 1 < TM) {
            constexpr int O = (kk * 32 * SY + (i + 1) * 16) * 2;
            xwr_tr_issue_at<O, O + 16 * SY * 2>(aYh, ryh[nxt]);
            if constexpr (PL == 2) xwr_tr_issue_at<YLOB + O, YLOB + O + 16 * SY * 2>(aYh, ryl[nxt]);
          } else if (kk + 1 < nk) {
            constexpr int O = (kk + 1) * 32 * SY * 2;
            xwr_tr_issue_at<O, O + 16 * SY * 2>(aYh, ryh[nxt]);
            if constexpr (PL == 2) xwr_tr_issue_at<YLOB + O, YLOB + O + 16 * SY * 2>(aYh, ryl[nxt]);
          }
          __builtin_amdgcn_sched_barrier(0);             // (the prefetch leaves before the MFMAs, not among them)
          xstatic_for<KS>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            if (DBG & 1) { asm volatile("" ::"v"(yl), "v"(yh), "v"(xh[t]), "v"(xl[t])); }
            else {
              if constexpr (PL == 2) {
                acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl, xh[t], acc[t][i], 0, 0, 0);
                acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xl[t], acc[t][i], 0, 0, 0);
              }
              acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xh[t], acc[t][i], 0, 0, 0);
            }
            if (i == TM - 1 && kk + 1 < nk) {             // this tap's fragments of the next k-step
              constexpr int O = ((kk + 1) * 32 + t) * SX * 2;
              xwr_tr_issue_at<O, O + 16 * SX * 2>(aXh, rxh[t]);
              if constexpr (PL == 2) xwr_tr_issue_at<XLOB + O, XLOB + O + 16 * SX * 2>(aXh, rxl[t]);
            }
          });
          // (the next stage's scalars are worked out behind the first MFMAs of the stage, not at the barrier where all
          // waves of the block would do it at the same moment with the matrix pipe empty)
          if (fill && kk * TM + i == 0) issue_prep((st + 1) & 1);
          if (fill && kk * TM + i < NI) issue_one(kk * TM + i);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
    if (fill) {                                          // what the MFMA stream had no slot for (one k-step, or NI > 2 TM)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (i >= nk * TM) issue_one(i);
    }
    cst(2);
  }
  rts(2);

  if (DBG & 4) {     // clock probe: shader-clock ticks and 100 MHz ticks over the main loop
    const unsigned long long tc1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.dbg) + (int64_t)blockIdx.x * 4;
      o[0] = tc1 - tc0; o[1] = tr1 - tr0; o[2] = (unsigned long long)nst;
    }
  }
  // ---- slab write: lane holds D[co = 4*(lane>>4) + r][ci = 16*wave + (lane & 15)] of each tile.  The tile
  // goes through LDS and leaves as whole 16-byte vectors, CHX*4-byte row segments (direct stores are
  // 64-byte fragments of 128-byte lines: 0.4 TB/s measured).
  __syncthreads();
  constexpr int RS = CHX + 4;
  float* red = reinterpret_cast<float*>(smem16);           // [CHY][RS]
  const int fcol = lane & 15, fq = (lane >> 4) * 4;
#pragma unroll
  for (int t = 0; t < KS; ++t) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(i * 16 + fq + r) * RS + wave * 16 + fcol] = acc[t][i][r];
    __syncthreads();
    float* slab = p.slabs + (((int64_t)s * KS * KS + trow * KS + t) * p.Np + co0) * p.Cq + ci0;
    for (int idx = tid; idx < CHY * (CHX / 4); idx += NW * 64) {
      const int row = idx / (CHX / 4), v = idx - row * (CHX / 4);
      if (co0 + row < p.Np && ci0 + v * 4 < p.Cq)
        *reinterpret_cast<float4*>(slab + (int64_t)row * p.Cq + v * 4) = *reinterpret_cast<const float4*>(red + row * RS + v * 4);
    }
    __syncthreads();
  }
  if (DBG & 16) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rts(3);
    if (lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.dbg) + ((int64_t)blockIdx.x * NW + wave) * 8;
      for (int i = 0; i < 4; ++i) o[i] = rt[i];
      for (int i = 0; i < 3; ++i) o[4 + i] = cyc[i];
      o[7] = (unsigned long long)nst;
    }
  }
}

// The KPCN instance (5x5, 7 x 7 channel tiles) of the filter-row kernel on EIGHT waves.  conv_wgrad_rows_bf16x3_kernel
// <5, 7, 7> gives wave w the input-channel tile w: seven waves on four SIMDs, 105 MFMAs per wave and k-step -- three SIMDs
// carry two waves (210 MFMAs per k-step), the fourth one (scripts/timeline_wgrad.py: waves 0-3 wait 3100 of 9060 cycles
// per stage for waves 4-6).  Here the 245 accumulator tiles (5 taps x 7 cin tiles x 7 cout tiles) are dealt evenly:
// (tap, cin tile) pair q = 7 tap + ci, wave w owns pairs 4w .. 4w+3 with all seven cout tiles (28 tiles) and, of the
// three pairs left over (tap 4, cin tiles 4..6), the cout tile w (wave 7 multiplies wave 0's again and drops it: no
// branch in the MFMA stream) -- 93 MFMAs per wave and k-step, 186 per SIMD.  Stage layout, fill, slab layout and the
// order of the MFMAs on every accumulator are those of the seven-wave kernel: the slabs are bit-identical.
template <int DBG = 0, int XE = 1, int PL = 2>
__global__ __launch_bounds__(512, 1) void conv_wgrad_rows8_bf16x3_kernel(XWRowsParams p) {
#define XWR8_READ(O1, O2, ADDR, REG) do { if (DBG & 32) { asm volatile("" : "+v"((REG).a), "+v"((REG).b)); } else xwr_tr_issue_at<O1, O2>(ADDR, REG); } while (0)
  constexpr int KS = 5, TM = 7, NCI = 7, NW = 8, NS = 4, NE = 3;
  constexpr int CHY = TM * 16, CHX = NCI * 16, PK = 64, XR = PK + KS - 1;
  constexpr int SY = xwr_stride(CHY), SX = xwr_stride(CHX);
  constexpr int VY = SY / 8, VX = SX / 8;
  constexpr int YV = PK * VY, XV = XR * VX;
  constexpr int NVEC = PL * YV + PL * XV;                   // PL = 1: [Yh | Xh] stages, one MFMA per product (see conv_wgrad_bf16x3_kernel)
  constexpr int NI = (NVEC + NW * 64 - 1) / (NW * 64);
  constexpr int BUF = NI * NW * 64 * 8;
  constexpr int XLO = XR * SX * 2;                          // byte offset of the lo plane of x (and below: of dy)
  constexpr int YLO = PK * SY * 2;
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int unit = (local / KS) * 8 + xcd;
  const int upb = p.coBlocks * p.ciBlocks;
  if (unit >= p.S * upb) return;
  const int trow = local % KS;
  const int s = unit / upb, ub = unit - s * upb;
  const int cob = ub / p.ciBlocks, cib = ub - cob * p.ciBlocks;
  const int co0 = cob * CHY, ci0 = cib * CHX;
  const int r0 = s * p.rps, r1 = min(p.R, r0 + p.rps);
  const int nch = (p.Wo + PK - 1) / PK;
  const int nrows = r1 - r0;
  const int nst = nrows * nch;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);

  // ---- stage fill: as conv_wgrad_rows_bf16x3_kernel (one linear run of 16-byte vectors [Yh | Yl | Xh | Xl])
  unsigned relv[NI]; int rowv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int v = (i * NW + wave) * 64 + lane;
    unsigned rel = 0; int rw = 127;
    if (v < PL * YV) {
      const int plane = v >= YV, vv = v - plane * YV;
      const int row = vv / VY, vec = vv - row * VY;
      const int co = co0 + vec * 8;
      if (vec * 8 < CHY && co < p.Cpo) { rel = (unsigned)(row * p.yps + plane * 2 * p.Cpo + co * 2); rw = row; }
    } else if (v < NVEC) {
      const int u = v - PL * YV;
      const int plane = u >= XV, uu = u - plane * XV;
      const int row = uu / VX, vec = uu - row * VX;
      const int ci = ci0 + vec * 8;
      if (vec * 8 < CHX && ci < p.Cpi) { rel = (unsigned)(row * p.xps + plane * 2 * p.Cpi + ci * 2); rw = row; }
    }
    relv[i] = rel; rowv[i] = rw;
  }
  unsigned f_ybase = 0, f_xbase = 0, f_yn = 0, f_xn = 0; int f_xlo = 0, f_buf = 0;
  int f_c = 0, f_rs = nrows > 0 ? (nrows - trow % nrows) % nrows : 0, f_n, f_oy;
  const int f_n0 = r0 / p.Ho, f_oy0 = r0 - f_n0 * p.Ho;
  { const int r = r0 + f_rs; f_n = r / p.Ho; f_oy = r - f_n * p.Ho; }
  auto issue_prep = [&](int buf) {
    const int ox0 = f_c * PK;
    const int iy = f_oy + trow - p.pad;
    const bool rowok = (unsigned)iy < (unsigned)p.H;
    f_ybase = (unsigned)(((f_n * p.Ho + f_oy) * p.Wo + ox0) * p.yps);
    f_xbase = (unsigned)(((f_n * p.H + iy) * p.W + ox0 - p.pad) * p.xps);       // may wrap: only used when valid
    f_yn = (unsigned)max(0, p.Wo - ox0);
    f_xlo = p.pad - ox0;
    f_xn = rowok ? (unsigned)p.W : 0u;
    f_buf = buf;
    if (++f_c == nch) {
      f_c = 0;
      if (++f_rs == nrows) { f_rs = 0; f_n = f_n0; f_oy = f_oy0; }
      else if (++f_oy == p.Ho) { f_oy = 0; ++f_n; }
    }
  };
  auto issue_one = [&](int i) {
    // the last instruction row is mostly past the stage's 3696 vectors: six of the eight waves have nothing to fetch there
    // (an LDS-DMA instruction holds the SIMD's vector issue for 60-100 cycles whether or not its lanes are in range)
    if ((i + 1) * NW * 64 > NVEC && (i * NW + wave) * 64 >= NVEC) return;
    // 2*YV is a multiple of 64: a wave-instruction is all dy or all x; only one instruction row straddles the two (written
    // out so that the others are compile-time choices and not wave-uniform masks kept in spilled scalar registers)
    const bool isx = (i * NW + NW - 1) * 64 < PL * YV ? false : i * NW * 64 >= PL * YV ? true : (i * NW + wave) * 64 >= PL * YV;
    const unsigned base = isx ? f_xbase : f_ybase, cnt = isx ? f_xn : f_yn;
    const int lo = isx ? f_xlo : 0;
    const unsigned off = (unsigned)(rowv[i] - lo) < cnt ? base + relv[i] : XOOB;
    __attribute__((address_space(3))) void* dst =
        (__attribute__((address_space(3))) void*)(smem16 + f_buf * BUF + (i * NW + wave) * 512);
    if (!isx) __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, dst, 16, off, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, off, 0, 0, 0);
  };

  // this wave's pairs: byte offset of the pair's fragment column (tap row + cin tile) inside an x plane, and the
  // cout tile of loop slot i (rotated by the wave: slot 0 is the tile of the wave's three extra accumulators)
  int xoff[NS], ycol[TM];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int pr = NS * wave + q, pt = pr / NCI, pc = pr - pt * NCI;
    xoff[q] = (pt * SX + pc * 16) * 2;
  }
  // XE = 1 (shipped; WCMC_WGRAD_ROWS8_XE=0 for the A/B): the 21 left-over tiles are dealt as ONE pair per wave x 2..4
  // consecutive cout tiles -- pair 32: waves 0-2 (cout tiles {0,1}, {2,3}, {4,5,6}), pair 33: waves 3-5 alike, pair 34: waves
  // 6, 7 ({0,1,2}, {3,4,5,6}); 4 / 5 / 6 / 6 extra tiles per SIMD (waves w, w + 4) -- so that a wave reads ONE extra x
  // fragment per k-step instead of three (48 instead of 56 transposing reads per 90-96 MFMAs; the kernel is bound by the
  // issue of its non-MFMA instructions: profiles/HISTORY.md 6.1).  The extras sit in loop slots 0 .. nex-1 (slots 2, 3 behind a
  // wave-uniform test); XE = 0: three pairs x cout tile `wave` in slot 0, wave 7 multiplies wave 0's again and drops them.
  const int er = wave % 3;
  const int epair = XE ? (wave < 6 ? wave / 3 : 2) : 0;
  const int ebase = XE ? (wave < 6 ? 2 * er : wave == 6 ? 0 : 3) : wave;
  const int nex = XE ? (wave < 6 ? (er == 2 ? 3 : 2) : wave == 6 ? 3 : 4) : 1;
#pragma unroll
  for (int i = 0; i < TM; ++i) ycol[i] = (ebase + i) % TM;
  constexpr int ETAP = KS - 1, ECI0 = NCI - NE;              // the left-over pairs: tap 4, cin tiles 4..6
  constexpr int NA = XE ? 4 : NE, NF = XE ? 1 : NE;          // extra accumulators / extra x fragments per wave
  const int exoff = (ETAP * SX + (ECI0 + epair) * 16) * 2;   // (XE) byte offset of the wave's extra pair inside an x plane

  f32x4 acc[NS][TM], ace[NA];
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < NA; ++e) ace[e] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) u16*)smem16);

  unsigned long long rt[4] = {0, 0, 0, 0}, cyc[3] = {0, 0, 0}, tprev = 0;
  auto rts = [&](int i) {
    if (DBG & 16) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      rt[i] = t;
    }
  };
  auto cst = [&](int i) {
    if (DBG & 16) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 0) cyc[i] += t - tprev;
      tprev = t;
    }
  };
  rts(0);
  if (nst > 0) {
    issue_prep(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_one(i);
  }
  rts(1);
  cst(-1);
  // The stage loop is unrolled by two so that the buffer of a stage is a compile-time choice: this lane's fragment
  // addresses in either buffer (7 dy cout tiles, 4 + 1 x slots) are worked out ONCE and every transposing read is an
  // address register plus an immediate -- the loop had ~30 address additions per stage and wave, and it is bound by the
  // issue of exactly such instructions (profiles/HISTORY.md 6.1).
  unsigned ayv[2][TM], axv[2][NS], aEv[2], aXv[2];
  {
    const int prow0 = 4 * g + tq;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const unsigned by = lds0 + (unsigned)((bb * BUF + prow0 * SY + 4 * tp) * 2);
      const unsigned bx = lds0 + (unsigned)((bb * BUF + PL * PK * SY + prow0 * SX + 4 * tp) * 2);
#pragma unroll
      for (int i = 0; i < TM; ++i) { ayv[bb][i] = by + (unsigned)(ycol[i] * 32); asm volatile("" : "+v"(ayv[bb][i])); }
#pragma unroll
      for (int q = 0; q < NS; ++q) { axv[bb][q] = bx + (unsigned)xoff[q]; asm volatile("" : "+v"(axv[bb][q])); }
      aEv[bb] = bx + (unsigned)exoff; asm volatile("" : "+v"(aEv[bb]));
      aXv[bb] = bx; asm volatile("" : "+v"(aXv[bb]));
    }
  }
  auto stage = [&](const int st, auto PAR) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cst(0);
    const bool fill = st + 1 < nst && !((DBG & 2) && st > 0);   // (DBG: timing-only ablations, wrong results -- 1 no MFMA, 2 no fills after the first, 8 no fragment waits, 32 no fragment reads)
    cst(1);
    // Two waves share a SIMD (w and w + 4) and of two ready waves the older one issues: waves 0-3 ran ahead and then
    // waited ~2800 of 8200 cycles per stage at the barrier while waves 4-7 finished alone, a lone wave keeping the matrix
    // pipe ~60 % busy against ~86 % for a pair (scripts/timeline_wgrad.py).  Waves 0-3 take priority 2 for the first
    // p.prio iterations of the stage and 0 afterwards, waves 4-7 stay at 1: both reach the barrier together.
    if (p.prio) { if (wave < 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
    const unsigned aX = aXv[par];
    const unsigned (&ax)[NS] = axv[par];
    const unsigned (&ayp)[TM] = ayv[par];
    const int c = st % nch;
    const int nk = (min(PK, p.Wo - c * PK) + 31) / 32;
    XwrRaw rxh[NS], rxl[NS], reh[NF], rel_[NF], ryh[2], ryl[2];
    const unsigned aE = aEv[par];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      XWR8_READ(0, 16 * SX * 2, ax[q], rxh[q]);
      if constexpr (PL == 2) XWR8_READ(XLO, XLO + 16 * SX * 2, ax[q], rxl[q]);
    }
    if (XE) {
      XWR8_READ(0, 16 * SX * 2, aE, reh[0]);
      if constexpr (PL == 2) XWR8_READ(XLO, XLO + 16 * SX * 2, aE, rel_[0]);
    } else {
#pragma unroll
      for (int e = 0; e < NF; ++e) {
        XWR8_READ(0, 16 * SX * 2, aX + (unsigned)((ETAP * SX + (ECI0 + e) * 16) * 2), reh[e]);
        if constexpr (PL == 2) XWR8_READ(XLO, XLO + 16 * SX * 2, aX + (unsigned)((ETAP * SX + (ECI0 + e) * 16) * 2), rel_[e]);
      }
    }
    XWR8_READ(0, 16 * SY * 2, ayp[0], ryh[0]);
    if constexpr (PL == 2) XWR8_READ(YLO, YLO + 16 * SY * 2, ayp[0], ryl[0]);
    bf16x8 xh[NS], xl[NS], eh[NF], el[NF];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk < nk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int cur = (kk * TM + i) & 1, nxt = cur ^ 1;
          if (kk * TM + i > 0 && p.prio == kk * TM + i && wave < 4) __builtin_amdgcn_s_setprio(0);
          if constexpr (PL == 1) {                        // (the lo registers do not exist in this instance)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ryh[cur].a), "+v"(ryh[cur].b));
            if (i == 0) {
#pragma unroll
              for (int q = 0; q < NS; ++q) { asm volatile("" : "+v"(rxh[q].a), "+v"(rxh[q].b)); xh[q] = xwr_cat(rxh[q]); }
#pragma unroll
              for (int e = 0; e < NF; ++e) { asm volatile("" : "+v"(reh[e].a), "+v"(reh[e].b)); eh[e] = xwr_cat(reh[e]); }
            }
          } else {
          if (DBG & 8) { asm volatile("" : "+v"(ryh[cur].a), "+v"(ryh[cur].b), "+v"(ryl[cur].a), "+v"(ryl[cur].b)); }
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ryh[cur].a), "+v"(ryh[cur].b), "+v"(ryl[cur].a), "+v"(ryl[cur].b));
          if (i == 0) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
              asm volatile("" : "+v"(rxh[q].a), "+v"(rxh[q].b), "+v"(rxl[q].a), "+v"(rxl[q].b));
              xh[q] = xwr_cat(rxh[q]); xl[q] = xwr_cat(rxl[q]);
            }
#pragma unroll
            for (int e = 0; e < NF; ++e) {
              asm volatile("" : "+v"(reh[e].a), "+v"(reh[e].b), "+v"(rel_[e].a), "+v"(rel_[e].b));
              eh[e] = xwr_cat(reh[e]); el[e] = xwr_cat(rel_[e]);
            }
          }
          }
          bf16x8 yh = xwr_cat(ryh[cur]), yl = yh;
          if constexpr (PL == 2) yl = xwr_cat(ryl[cur]);
          constexpr int KY = 32 * SY * 2;                 // the second k-step of the dy planes
          if (i + 1 < TM) {
            if (kk == 0) {
              XWR8_READ(0, 16 * SY * 2, ayp[i + 1 < TM ? i + 1 : 0], ryh[nxt]);
              if constexpr (PL == 2) XWR8_READ(YLO, YLO + 16 * SY * 2, ayp[i + 1 < TM ? i + 1 : 0], ryl[nxt]);
            } else {
              XWR8_READ(KY, KY + 16 * SY * 2, ayp[i + 1 < TM ? i + 1 : 0], ryh[nxt]);
              if constexpr (PL == 2) XWR8_READ(KY + YLO, KY + YLO + 16 * SY * 2, ayp[i + 1 < TM ? i + 1 : 0], ryl[nxt]);
            }
          } else if (kk + 1 < nk) {
            XWR8_READ(KY, KY + 16 * SY * 2, ayp[0], ryh[nxt]);
            if constexpr (PL == 2) XWR8_READ(KY + YLO, KY + YLO + 16 * SY * 2, ayp[0], ryl[nxt]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (XE) {                                       // the left-over pair of this wave: cout tiles ycol[0 .. nex-1]
            if (i < 4) {
              if (i < 2 || i < nex) {
                if (DBG & 1) { asm volatile("" ::"v"(yl), "v"(yh), "v"(eh[0]), "v"(el[0])); }
                else {
                  if constexpr (PL == 2) {
                    ace[i < NA ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl, eh[0], ace[i < NA ? i : 0], 0, 0, 0);
                    ace[i < NA ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, el[0], ace[i < NA ? i : 0], 0, 0, 0);
                  }
                  ace[i < NA ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, eh[0], ace[i < NA ? i : 0], 0, 0, 0);
                }
              }
              if (i == 3 && kk + 1 < nk) {
                constexpr int K1 = 32 * SX * 2;
                XWR8_READ(K1, K1 + 16 * SX * 2, aE, reh[0]);
                if constexpr (PL == 2) XWR8_READ(K1 + XLO, K1 + XLO + 16 * SX * 2, aE, rel_[0]);
              }
            }
          } else if (i == 0) {                            // the left-over pairs: cout tile ycol[0] = wave
#pragma unroll
            for (int e = 0; e < NF; ++e) {
              if (DBG & 1) { asm volatile("" ::"v"(yl), "v"(yh), "v"(eh[e]), "v"(el[e])); }
              else {
                if constexpr (PL == 2) {
                  ace[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl, eh[e], ace[e], 0, 0, 0);
                  ace[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, el[e], ace[e], 0, 0, 0);
                }
                ace[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, eh[e], ace[e], 0, 0, 0);
              }
              if (kk + 1 < nk) {
                const unsigned ae = aX + (unsigned)((((kk + 1) * 32 + ETAP) * SX + (ECI0 + e) * 16) * 2);
                XWR8_READ(0, 16 * SX * 2, ae, reh[e]);
                if constexpr (PL == 2) XWR8_READ(XLO, XLO + 16 * SX * 2, ae, rel_[e]);
              }
            }
          }
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            if (DBG & 1) { asm volatile("" ::"v"(yl), "v"(yh), "v"(xh[q]), "v"(xl[q])); }
            else {
              if constexpr (PL == 2) {
                acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl, xh[q], acc[q][i], 0, 0, 0);
                acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xl[q], acc[q][i], 0, 0, 0);
              }
              acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xh[q], acc[q][i], 0, 0, 0);
            }
            if (i == TM - 1 && kk + 1 < nk) {
              constexpr int K1 = 32 * SX * 2;              // (kk + 1 < nk <= 2: the second k-step)
              XWR8_READ(K1, K1 + 16 * SX * 2, ax[q], rxh[q]);
              if constexpr (PL == 2) XWR8_READ(K1 + XLO, K1 + XLO + 16 * SX * 2, ax[q], rxl[q]);
            }
          }
          // (the next stage's scalars are worked out here, behind the first MFMAs of the stage, not at the barrier where
          // both waves of every SIMD would do it at the same moment with the matrix pipe empty)
          if (fill && kk * TM + i == 0) issue_prep(par ^ 1);
          if (fill && kk * TM + i < NI) issue_one(kk * TM + i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (fill) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (i >= nk * TM) issue_one(i);
    }
    cst(2);
  };
  for (int st = 0; st < nst; st += 2) {
    stage(st, std::integral_constant<int, 0>{});
    if (st + 1 < nst) stage(st + 1, std::integral_constant<int, 1>{});
  }
  rts(2);

  // ---- slab write, tap by tap through LDS (as the seven-wave kernel): the wave stages the tiles of its pairs of this tap
  __syncthreads();
  constexpr int RS = CHX + 4;
  float* red = reinterpret_cast<float*>(smem16);           // [CHY][RS]
  const int fcol = lane & 15, fq = (lane >> 4) * 4;
  int etap[NS], eci[NS];                                   // (recomputed: not kept live through the stage loop)
#pragma unroll
  for (int q = 0; q < NS; ++q) { const int pr = NS * wave + q; etap[q] = pr / NCI; eci[q] = pr - etap[q] * NCI; }
#pragma unroll
  for (int t = 0; t < KS; ++t) {
#pragma unroll
    for (int q = 0; q < NS; ++q)
      if (etap[q] == t) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(ycol[i] * 16 + fq + r) * RS + eci[q] * 16 + fcol] = acc[q][i][r];
      }
    if (XE) {
      if (t == ETAP) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
          if (j < nex) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(ycol[j] * 16 + fq + r) * RS + (ECI0 + epair) * 16 + fcol] = ace[j][r];
          }
      }
    } else if (t == ETAP && wave < TM) {
#pragma unroll
      for (int e = 0; e < NF; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + fq + r) * RS + (ECI0 + e) * 16 + fcol] = ace[e][r];
    }
    __syncthreads();
    float* slab = p.slabs + (((int64_t)s * KS * KS + trow * KS + t) * p.Np + co0) * p.Cq + ci0;
    for (int idx = tid; idx < CHY * (CHX / 4); idx += NW * 64) {
      const int row = idx / (CHX / 4), v = idx - row * (CHX / 4);
      if (co0 + row < p.Np && ci0 + v * 4 < p.Cq)
        *reinterpret_cast<float4*>(slab + (int64_t)row * p.Cq + v * 4) = *reinterpret_cast<const float4*>(red + row * RS + v * 4);
    }
    __syncthreads();
  }
  if (DBG & 16) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rts(3);
    if (lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.dbg) + ((int64_t)blockIdx.x * NW + wave) * 8;
      for (int i = 0; i < 4; ++i) o[i] = rt[i];
      for (int i = 0; i < 3; ++i) o[4 + i] = cyc[i];
      o[7] = (unsigned long long)nst;
    }
  }
}
#undef XWR8_READ

template <int KS, int TM, int NW, int PL = 2>
static constexpr size_t xwr_lds_bytes() {
  constexpr int NVEC = PL * 64 * (xwr_stride(TM * 16) / 8) + PL * (64 + KS - 1) * (xwr_stride(NW * 16) / 8);
  constexpr int NI = (NVEC + NW * 64 - 1) / (NW * 64);
  constexpr size_t stage = (size_t)2 * NI * NW * 64 * 16;
  constexpr size_t red = (size_t)TM * 16 * (NW * 16 + 4) * sizeof(float);
  return stage > red ? stage : red;
}
static size_t xwr_lds_bytes_rt(int ks, int tm, int nw, int pl = 2) {
  const int nvec = pl * 64 * (xwr_stride(tm * 16) / 8) + pl * (64 + ks - 1) * (xwr_stride(nw * 16) / 8);
  const int ni = (nvec + nw * 64 - 1) / (nw * 64);
  const size_t stage = (size_t)2 * ni * nw * 64 * 16, red = (size_t)tm * 16 * (nw * 16 + 4) * sizeof(float);
  return stage > red ? stage : red;
}

template <int KS, int TM, int NW, int PL = 2>
static int launch_xwgrad_rows(const XWRowsParams& q, hipStream_t st) {
  constexpr size_t lds = xwr_lds_bytes<KS, TM, NW, PL>();
  const dim3 grid((unsigned)(((q.S * q.coBlocks * q.ciBlocks + 7) / 8) * 8 * KS));
  // The eight-wave kernel for the two-plane (three-term) launches, the seven-wave one for the one-plane launches of the default
  // mode: there the seven waves are faster alone (0.311 against 0.295 of the bf16 peak in the eager profile) and beside the other
  // half of the step (+0.9 % per step, round 4).  WCMC_WGRAD_ROWS8=1 / 0: eight / seven waves for both.
  const char* r8e = ab_env("WCMC_WGRAD_ROWS8");
  const bool rows8 = r8e ? r8e[0] != '0' : PL == 2;
  if (KS == 5 && TM == 7 && NW == 7 && rows8) {
    // two stages of NI = 8 (PL = 1: 4) instructions x 8 waves x 1 KB (> the 52 KB staging tile of the slab write)
    constexpr size_t lds8 = (size_t)2 * ((PL * (64 * 14 + 68 * 14) + 511) / 512) * 512 * 16;
    if (PL == 1) {
      static LdsAttr attr81_set;
      if (set_max_lds(reinterpret_cast<const void*>(&conv_wgrad_rows8_bf16x3_kernel<0, 1, 1>), (size_t)lds8, attr81_set) != hipSuccess) return WCMC_ERR_LAUNCH;
      hipLaunchKernelGGL((conv_wgrad_rows8_bf16x3_kernel<0, 1, 1>), grid, dim3(512), lds8, st, q);
      return check_launch("conv2d_wgrad_bf16x3(rows8, one plane)");
    }
    static LdsAttr attr8_set, attr80_set;
#ifdef WCMC_DEBUG_BUILD
    { const char* e = ab_env("WCMC_DEBUG_ABLATE");
      const int ab = e ? atoi(e) : 0;
      auto kfn = ab == 16 ? &conv_wgrad_rows8_bf16x3_kernel<16> : ab == 1 ? &conv_wgrad_rows8_bf16x3_kernel<1> : ab == 2 ? &conv_wgrad_rows8_bf16x3_kernel<2>
                 : ab == 3 ? &conv_wgrad_rows8_bf16x3_kernel<3> : ab == 8 ? &conv_wgrad_rows8_bf16x3_kernel<8> : ab == 32 ? &conv_wgrad_rows8_bf16x3_kernel<32>
                 : ab == 34 ? &conv_wgrad_rows8_bf16x3_kernel<34> : ab == 35 ? &conv_wgrad_rows8_bf16x3_kernel<35> : nullptr;
      if (kfn) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
        hipLaunchKernelGGL(kfn, grid, dim3(512), lds8, st, q);
        return check_launch("conv2d_wgrad_bf16x3(rows8 ablation / stamps)");
      } }
#endif
    if (set_max_lds(reinterpret_cast<const void*>(&conv_wgrad_rows8_bf16x3_kernel<0, 1>), (size_t)lds8, attr8_set) != hipSuccess) return WCMC_ERR_LAUNCH;
    if (set_max_lds(reinterpret_cast<const void*>(&conv_wgrad_rows8_bf16x3_kernel<0, 0>), (size_t)lds8, attr80_set) != hipSuccess) return WCMC_ERR_LAUNCH;
    if (x_env_on("WCMC_WGRAD_ROWS8_XE")) hipLaunchKernelGGL((conv_wgrad_rows8_bf16x3_kernel<0, 1>), grid, dim3(512), lds8, st, q);
    else hipLaunchKernelGGL((conv_wgrad_rows8_bf16x3_kernel<0, 0>), grid, dim3(512), lds8, st, q);
    return check_launch("conv2d_wgrad_bf16x3(rows8)");
  }
#ifdef WCMC_DEBUG_BUILD        // `make debug` only: timing-only instances that compute WRONG results are not in the release library
  if (KS == 5 && TM == 7 && NW == 7 && PL == 2) {
    int ab;                             // WCMC_DEBUG_ABLATE: timing-only builds (1 = no MFMA, 2 = no stage fills, 4 = clock probe)
    { const char* e = ab_env("WCMC_DEBUG_ABLATE"); ab = e ? atoi(e) : 0; }
    if (ab == 1 || ab == 2 || ab == 3 || ab == 4 || ab == 8 || ab == 16) {
      auto kfn = ab == 16 ? &conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 16> : ab == 1 ? &conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 1> : ab == 2 ? &conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 2>
                 : ab == 3 ? &conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 3> : ab == 4 ? &conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 4>
                 : &conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 8>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, grid, dim3(448), lds, st, q);
      return check_launch("conv2d_wgrad_bf16x3(rows ablation)");
    }
  }
#endif
  static LdsAttr attr_set;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_wgrad_rows_bf16x3_kernel<KS, TM, NW, 0, PL>), (size_t)lds, attr_set) != hipSuccess) return WCMC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv_wgrad_rows_bf16x3_kernel<KS, TM, NW, 0, PL>), grid, dim3(NW * 64), lds, st, q);
  return check_launch("conv2d_wgrad_bf16x3(rows)");
}

// bias gradient from a split tensor: partial[g][c] = sum over the block's pixels of hi + lo.
// One thread = 8 channels (two 16-byte loads per pixel), 256/V pixel lanes, LDS tree across them.
__global__ __launch_bounds__(256) void colsum_split_kernel(const u16* __restrict__ dy, int Cp, int C, int64_t M,
                                                            int64_t per_block, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [PL][V][8]
  const int V = Cp / 8;                      // <= 256 (C <= 2048)
  const int PL = 256 / V;
  const int v = threadIdx.x % V, pl = threadIdx.x / V;
  const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = min(M, p0 + per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < PL) {
    for (int64_t q = p0 + pl; q < p1; q += PL) {
      const u16* r = dy + q * 2 * Cp + v * 8;
      const uint4 h = *reinterpret_cast<const uint4*>(r), l = *reinterpret_cast<const uint4*>(r + Cp);
      const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw[e] << 16);
        acc[2 * e + 1] += __builtin_bit_cast(float, hw[e] & 0xffff0000u) + __builtin_bit_cast(float, lw[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) smem[(pl * V + v) * 8 + e] = acc[e];
  }
  __syncthreads();
  if (pl == 0) {
    for (int q = 1; q < PL; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += smem[(q * V + v) * 8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (v * 8 + e < C) partial[(int64_t)blockIdx.x * C + v * 8 + e] = acc[e];
  }
}

static int x_pick_nt(int tiles) {
  const int cand[4] = {7, 4, 2, 1};
  int best = 1, best_cost = 1 << 30;
  for (int i = 0; i < 4; ++i) {
    const int nt = cand[i];
    const int cost = ((tiles + nt - 1) / nt) * (nt + 2);
    if (cost < best_cost) { best_cost = cost; best = nt; }
  }
  return best;
}

struct XWgradPlan { int rows, rps, R, rTM, rNW; int TM, coBlocks, ciBlocks, S, Np, Cq, G; int64_t pix_per_split, per_block; size_t slab_elems, bytes; };
// terms: bf16 MFMAs per product of the launch (3 | 1).  The one-plane instances need half the LDS per stage, so two of the
// four-wave filter-row blocks share a CU where the two-plane ones run alone: twice the splits for those layers (the
// U-Net's 64-channel levels: 28 -> 23 us, 192 -> 64: 72 -> 47 us; scripts/time_wgrad_unet.py) -- one wave per SIMD cannot hide
// its own DMA issue and barriers.  The slab layout follows the plan: the launch and its reduction ask with the same `terms`.
static XWgradPlan x_plan_wgrad(int N, int Ho, int Wo, int Cout, int Cin, int ks, int terms = 3) {
  XWgradPlan pl;
  pl.Np = round_up(Cout, 16); pl.Cq = round_up(Cin, 16);
  const int coT = pl.Np / 16, ciT = pl.Cq / 16;
  pl.TM = (coT % 7 == 0) ? 7 : 4;
  pl.coBlocks = (coT + pl.TM - 1) / pl.TM;
  pl.ciBlocks = (ciT + 3) / 4;
  const int64_t M = (int64_t)N * Ho * Wo;
  const int taps = ks * ks;
  const int rows_on = x_env_on("WCMC_WGRAD_ROWS");   // =0: A/B switch back to the one-tap-per-block kernel
  // filter-row kernel: (KS, TM, NW) instances below; TM / NW must divide the tile counts
  pl.R = N * Ho; pl.rps = 0; pl.rows = 0; pl.rTM = pl.rNW = 0;
  if (rows_on && (ks == 5 || ks == 3 || (ks == 1 && x_env_on("WCMC_WGRAD_ROWS_1X1"))) && (int64_t)N * Ho >= 64) {
    // measured against the one-tap kernel (scripts/profile_layers.py): the 5x5 layers gain 1.9-2.7x; of the
    // 3x3 U-Net layers only those with >= 256 input channels gain (a filter row is 3 taps of reuse, not 5)
    int tm = 0, nw = 0;
    if (ks == 5) { tm = coT % 7 == 0 ? 7 : 0; nw = ciT % 7 == 0 ? 7 : ciT == 3 ? 3 : 0; }
    else if (ks == 1) {
      // the PathNet 1x1 layers: pure streaming (two operands read once); the stage ring of this kernel fills by LDS-DMA
      // while the previous stage multiplies, the one-tap kernel stages through registers between two barriers
      // (measured, scripts/time_wgrad_1x1.py: 128 -> 128 gains, 220 -> 184 us = 5.8 TB/s; 64 -> 64 is even and the narrow
      // layers 36 -> 64 and 128 -> 3 lose 10-14 %: they stay on the one-tap kernel, which already streams them at 5.6-5.9 TB/s)
      if (coT == 8 && ciT == 8) { tm = 8; nw = 8; }
    }
    else if (ks == 3) {
      // (with the fill overlapped -- see the kernel -- the filter-row kernel also wins on the 128-channel levels: 128 -> 128 at
      // 64^2 57 -> 36 us, 128 -> 256 at 32^2 32 -> 26; the 64-channel layers are even; WCMC_WGRAD_ROWS_3X3=0: A/B switch back
      // to >= 256 input channels only -- scripts/time_wgrad_unet.py)
      const int wide = x_env_on("WCMC_WGRAD_ROWS_3X3");
      const char* e44 = ab_env("WCMC_WGRAD_44");          // (debug build) 1: 4 x 4 channel tiles per block everywhere
      if (coT % 8 == 0 && ciT % 8 == 0 && (ciT >= 16 || wide) && !(e44 && e44[0] == '1')) { tm = 8; nw = 8; }
      else if (wide && coT % 4 == 0 && ciT % 4 == 0) { tm = 4; nw = 4; }
    }
    if (tm && nw) {
      pl.rows = 1; pl.rTM = tm; pl.rNW = nw;
      pl.coBlocks = coT / tm; pl.ciBlocks = ciT / nw;
      // one block per (unit, filter row); the ks blocks of a unit share an XCD (32 CUs x resident blocks
      // per CU): at most that many per XCD keeps the launch to one round
      const size_t lds = xwr_lds_bytes_rt(ks, tm, nw, terms == 1 ? 1 : 2);
      int wpc = (int)((160 * 1024) / lds);
      int wcap = nw <= 4 ? 2 : 1;                     // as the kernel's __launch_bounds__
      { const char* e = ab_env("WCMC_WGRAD_WCAP"); if (e && nw <= 4) wcap = atoi(e); }      // (debug build: scripts/time_wgrad_unet.py)
      int sdiv = 1;
      { const char* e = ab_env("WCMC_WGRAD_SDIV"); if (e) sdiv = atoi(e); }                 // (debug build: fewer, longer splits)
      if (wpc > wcap) wpc = wcap;
      if (wpc < 1) wpc = 1;
      int S = 8 * ((32 * wpc) / ks) / (pl.coBlocks * pl.ciBlocks) / sdiv;
      if (S > pl.R / 4) S = pl.R / 4;                 // at least 4 rows per block
      if (S < 1) S = 1;
      pl.rps = (pl.R + S - 1) / S;
      pl.S = (pl.R + pl.rps - 1) / pl.rps;
      pl.pix_per_split = 0;
    }
  }
  const int64_t tiles = (int64_t)taps * pl.coBlocks * pl.ciBlocks;
  // ~2 waves of 512 co-resident blocks for the multi-tap convs; one wave for the HBM-bound 1x1 layers,
  // whose slab traffic (S x Np x Cq floats, written and re-read) otherwise rivals the operand stream
  int64_t S = (ks == 1 ? 512 : 1024) / tiles;
  const int64_t maxS = M / 512 > 0 ? M / 512 : 1;     // >= 8 stages of 64 pixels per block
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  if (!pl.rows) {
    pl.pix_per_split = ceil_div64(ceil_div64(M, S), 64) * 64;
    pl.S = (int)ceil_div64(M, pl.pix_per_split);
  }
  pl.slab_elems = (size_t)pl.S * taps * pl.Np * pl.Cq;
  pl.G = (int)(M / 64 > 0 ? (M / 64 < 1024 ? M / 64 : 1024) : 1);
  pl.per_block = ceil_div64(M, pl.G);
  pl.G = (int)ceil_div64(M, pl.per_block);
  pl.bytes = (pl.slab_elems + (size_t)pl.G * Cout) * sizeof(float);
  return pl;
}

}  // namespace wcmc

using namespace wcmc;

extern "C" size_t wcmc_split_elems(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  return (size_t)N * H * W * 2 * round_up(C, 8);
}

extern "C" int wcmc_split_bf16(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, void* out, int N, int H, int W,
                               int C, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && out, WCMC_ERR_BAD_ARG, "split_bf16: bad argument");
  WCMC_REQUIRE(nhwc_view_ok(x, xsn, xsh, xsw, C) && aligned16(out), WCMC_ERR_ALIGNMENT,
               "split_bf16: x violates the NHWC-view contract (or out unaligned)");
  const int Cp = round_up(C, 8);
  const int64_t total = (int64_t)N * H * W * (Cp / 8);
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                     x, xsn, xsh, xsw, (u16*)out, H, W, C, Cp, total, (const float*)nullptr, (int64_t)0, (int64_t)0,
                     (int64_t)0, 0, 0.f);
  return check_launch("split_bf16");
}

extern "C" int wcmc_split_gated_bf16(const float* dy, int64_t xsn, int64_t xsh, int64_t xsw, const float* post, int64_t psn,
                                     int64_t psh, int64_t psw, int act, float slope, void* out, int N, int H, int W, int C,
                                     void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && out, WCMC_ERR_BAD_ARG, "split_gated_bf16: bad argument");
  WCMC_REQUIRE(nhwc_view_ok(dy, xsn, xsh, xsw, C) && nhwc_view_ok(post, psn, psh, psw, C) && aligned16(out),
               WCMC_ERR_ALIGNMENT, "split_gated_bf16: dy / post violate the NHWC-view contract (or out unaligned)");
  const int Cp = round_up(C, 8);
  const int64_t total = (int64_t)N * H * W * (Cp / 8);
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                     dy, xsn, xsh, xsw, (u16*)out, H, W, C, Cp, total, post, psn, psh, psw, act, slope);
  return check_launch("split_gated_bf16");
}

extern "C" int wcmc_split_from_nchw(const float* src, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw, void* out_split,
                                    int N, int C, int H, int W, void* stream) {
  WCMC_REQUIRE(src && out_split && N > 0 && C > 0 && C <= 64 && H > 0 && W > 0, WCMC_ERR_BAD_ARG,
               "split_from_nchw: bad argument (at most 64 channels)");
  WCMC_REQUIRE(aligned16(out_split), WCMC_ERR_ALIGNMENT, "split_from_nchw: out must be 16-byte aligned");
  WCMC_REQUIRE((int64_t)N * H <= 65535, WCMC_ERR_BAD_ARG, "split_from_nchw: N*H > 65535");
  hipLaunchKernelGGL(nchw_split_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)(N * H)), dim3(256), 0,
                     (hipStream_t)stream, src, ssn, ssc, ssh, ssw, (u16*)out_split, C, round_up(C, 8), H, W);
  return check_launch("split_from_nchw");
}

extern "C" int wcmc_cat_broadcast_split(const float* flat, int64_t fsn, int64_t fsh, int64_t fsw, const float* prop,
                                       int64_t psn, int64_t psh, int64_t psw, void* out_split, int B, int S, int H,
                                       int W, int C1, int C2, void* stream) {
  WCMC_REQUIRE(flat && prop && out_split && B > 0 && S > 0 && H > 0 && W > 0 && C1 > 0 && C2 > 0, WCMC_ERR_BAD_ARG,
               "cat_broadcast_split: bad argument");
  WCMC_REQUIRE(C1 % 8 == 0, WCMC_ERR_BAD_ARG, "cat_broadcast_split: the first operand needs a multiple of 8 channels");
  WCMC_REQUIRE(nhwc_view_ok(flat, fsn, fsh, fsw, C1) && nhwc_view_ok(prop, psn, psh, psw, C2) && aligned16(out_split),
               WCMC_ERR_ALIGNMENT, "cat_broadcast_split: a view violates the NHWC-view contract");
  const int Cp = round_up(C1 + C2, 8);
  const int64_t total = (int64_t)B * S * H * W * (Cp / 8);
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(cat_broadcast_split_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, flat, fsn, fsh, fsw, prop, psn, psh, psw, (u16*)out_split, S, H, W, C1, C2, Cp,
                     total, 0);
  return check_launch("cat_broadcast_split");
}

extern "C" int wcmc_cat_upsample_split(const float* deep, int64_t dsn, int64_t dsh, int64_t dsw, const float* skip,
                                       int64_t ssn, int64_t ssh, int64_t ssw, void* out_split, int N, int H, int W, int C1,
                                       int C2, void* stream) {
  WCMC_REQUIRE(deep && skip && out_split && N > 0 && H > 1 && W > 1 && (H % 2) == 0 && (W % 2) == 0 && C1 > 0 && C2 > 0,
               WCMC_ERR_BAD_ARG, "cat_upsample_split: bad argument (H and W are the fine, even, geometry)");
  WCMC_REQUIRE(C1 % 8 == 0, WCMC_ERR_BAD_ARG, "cat_upsample_split: the upsampled operand needs a multiple of 8 channels");
  WCMC_REQUIRE(nhwc_view_ok(deep, dsn, dsh, dsw, C1) && nhwc_view_ok(skip, ssn, ssh, ssw, C2) && aligned16(out_split),
               WCMC_ERR_ALIGNMENT, "cat_upsample_split: a view violates the NHWC-view contract");
  const int Cp = round_up(C1 + C2, 8);
  const int64_t total = (int64_t)N * H * W * (Cp / 8);
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(cat_broadcast_split_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, deep, dsn, dsh, dsw, skip, ssn, ssh, ssw, (u16*)out_split, 1, H, W, C1, C2, Cp,
                     total, 1);
  return check_launch("cat_upsample_split");
}

extern "C" int wcmc_add_broadcast_split(const float* g, int64_t gsn, int64_t gsh, int64_t gsw, const float* gm,
                                       int64_t msn, int64_t msh, int64_t msw, float scale, void* out_split, int B,
                                       int S, int H, int W, int C, void* stream) {
  WCMC_REQUIRE((g || gm) && out_split && B > 0 && S > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG,
               "add_broadcast_split: bad argument");
  WCMC_REQUIRE((!g || nhwc_view_ok(g, gsn, gsh, gsw, C)) && (!gm || nhwc_view_ok(gm, msn, msh, msw, C)) &&
                   aligned16(out_split),
               WCMC_ERR_ALIGNMENT, "add_broadcast_split: a view violates the NHWC-view contract");
  const int Cp = round_up(C, 8);
  const int64_t total = (int64_t)B * S * H * W * (Cp / 8);
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(add_broadcast_split_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, g, gsn, gsh, gsw, gm, msn, msh, msw, scale, (u16*)out_split, S, H, W, C, Cp,
                     total);
  return check_launch("add_broadcast_split");
}

extern "C" int wcmc_split_dy_colsum_bf16(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw, const float* post, int64_t psn,
                                         int64_t psh, int64_t psw, int act, float slope, const float* gm, int64_t msn,
                                         int64_t msh, int64_t msw, int S, float scale, void* out_split, float* colsum_partial,
                                         int N, int H, int W, int C, void* stream) {
  WCMC_REQUIRE((dy || gm) && out_split && colsum_partial && N > 0 && S > 0 && H > 0 && W > 0 && C > 0 && C <= 2048,
               WCMC_ERR_BAD_ARG, "split_dy_colsum_bf16: bad argument");
  WCMC_REQUIRE(!gm || N % S == 0, WCMC_ERR_BAD_ARG, "split_dy_colsum_bf16: N must be a multiple of S");
  WCMC_REQUIRE((!dy || nhwc_view_ok(dy, dsn, dsh, dsw, C)) && (!post || nhwc_view_ok(post, psn, psh, psw, C)) &&
                   (!gm || nhwc_view_ok(gm, msn, msh, msw, C)) && aligned16(out_split),
               WCMC_ERR_ALIGNMENT, "split_dy_colsum_bf16: a view violates the NHWC-view contract (or out unaligned)");
  const int Cp = round_up(C, 8), Np = round_up(C, 16);
  const int64_t M = (int64_t)N * H * W;
  const int Gmax = x_colsum_rows(N, H, W);
  int blocks = Gmax < 1024 ? Gmax : 1024;
  const int64_t per_block = ceil_div64(M, blocks);
  blocks = (int)ceil_div64(M, per_block);
  hipLaunchKernelGGL(split_dy_colsum_kernel, dim3((unsigned)blocks), dim3(256), (size_t)256 * 8 * sizeof(float),
                     (hipStream_t)stream, dy, dsn, dsh, dsw, post, psn, psh, psw, act, slope, gm, msn, msh, msw, S, scale,
                     (u16*)out_split, H, W, C, Cp, M, per_block, colsum_partial, Np, Gmax);
  return check_launch("split_dy_colsum_bf16");
}

// mode of a packed weight: 0 = forward orientation, 1 = data-gradient orientation (flipped taps, channels swapped), 2 = the
// data-gradient orientation in the K order of a TWO-term launch (terms = 2 of wcmc_conv2d_igemm_bf16x3: x hi plane only)
// 3 = the FORWARD orientation in the K order of a two- / one-term launch (terms <= 2 of a forward launch: the un-gated output layers
// of the "bf16x321o" mode)
// 4 = mode 3 with the weights rounded ONCE to fp16 in the hi rows (wcmc_conv2d_out_f16; the lo rows are zero and never read)
static inline int x_mode_ap(int mode) { return mode >= 2 ? 1 : 2; }
static inline bool x_mode_fwd(int mode) { return mode == 0 || mode == 3 || mode == 4; }
extern "C" size_t wcmc_conv2d_packed_elems_bf16x3(int rows, int kchan, int ks, int mode) {
  if (rows <= 0 || kchan <= 0 || ks <= 0 || mode < 0 || mode > 4) return 0;
  return (size_t)round_up(rows, 16) * 2 * x_plan_k(kchan, ks, x_mode_ap(mode), rows).Kt;
}

extern "C" int wcmc_conv2d_pack_weight_bf16x3(const float* w, void* wp, int Cout, int Cin, int ks, int mode,
                                              void* stream) {
  WCMC_REQUIRE(w && wp && Cout > 0 && Cin > 0 && ks > 0 && mode >= 0 && mode <= 4, WCMC_ERR_BAD_ARG,
               "conv2d_pack_weight_bf16x3: bad argument");
  const int rows = x_mode_fwd(mode) ? Cout : Cin, kchan = x_mode_fwd(mode) ? Cin : Cout;
  const int Np = round_up(rows, 16);
  const XKPlan q = x_plan_k(kchan, ks, x_mode_ap(mode), rows);
  const int64_t total = (int64_t)Np * q.Kt;
  hipLaunchKernelGGL(pack_weight_split_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, (u16*)wp, Cout, Cin, ks, x_mode_fwd(mode) ? 0 : 1, rows, Np, q.CS, q.Ks, q.Kt, q.nslabs, q.CSl, mode == 4 ? 1 : 0);
  return check_launch("conv2d_pack_weight_bf16x3");
}

extern "C" int wcmc_conv2d_pack_chain_bf16x3(int n_entries, const float* const* w, void* const* wp, const int* Cout,
                                             const int* Cin, const int* mode, int ks, void* stream) {
  WCMC_REQUIRE(n_entries > 0 && n_entries <= XPACK_MAX && w && wp && Cout && Cin && mode && ks > 0, WCMC_ERR_BAD_ARG,
               "conv2d_pack_chain_bf16x3: bad argument (at most %d entries)", XPACK_MAX);
  XPackTable t;
  t.n = n_entries; t.ks = ks;
  unsigned blocks = 0;
  for (int i = 0; i < n_entries; ++i) {
    WCMC_REQUIRE(w[i] && wp[i] && Cout[i] > 0 && Cin[i] > 0 && mode[i] >= 0 && mode[i] <= 4, WCMC_ERR_BAD_ARG,
                 "conv2d_pack_chain_bf16x3: bad entry %d", i);
    XPackEntry& e = t.e[i];
    e.w = w[i]; e.wp = (u16*)wp[i]; e.Cout = Cout[i]; e.Cin = Cin[i]; e.mode = x_mode_fwd(mode[i]) ? 0 : 1;      // (the kernel knows orientations only)
    e.f16 = mode[i] == 4 ? 1 : 0;
    e.rows = x_mode_fwd(mode[i]) ? Cout[i] : Cin[i];
    const int kchan = x_mode_fwd(mode[i]) ? Cin[i] : Cout[i];
    e.Np = round_up(e.rows, 16);
    const XKPlan q = x_plan_k(kchan, ks, x_mode_ap(mode[i]), e.rows);
    e.CS = q.CS; e.Ks = q.Ks; e.Kt = q.Kt; e.nslabs = q.nslabs; e.CSl = q.CSl;
    e.block0 = blocks;
    blocks += (unsigned)ceil_div64((int64_t)e.Np * q.Kt, 256);
  }
  hipLaunchKernelGGL(pack_weight_split_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
  return check_launch("conv2d_pack_chain_bf16x3");
}

extern "C" int wcmc_conv2d_wgrad_reduce_multi(int n, void* const* workspace, float* const* dw, float* const* db,
                                              const float* const* dy_colsum_partial, const int* N, const int* Ho, const int* Wo,
                                              const int* Cout, const int* Cin, const int* ks, int terms, void* stream) {
  WCMC_REQUIRE(n > 0 && n <= WRM_MAX && workspace && dw && db && dy_colsum_partial && N && Ho && Wo && Cout && Cin && ks &&
               (terms == 1 || terms == 3), WCMC_ERR_BAD_ARG, "conv2d_wgrad_reduce_multi: bad argument (1..%d layers)", WRM_MAX);
  WRMTable t;
  t.n = n;
  unsigned blocks = 0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    WCMC_REQUIRE(workspace[i] && dw[i] && N[i] > 0 && Ho[i] > 0 && Wo[i] > 0 && Cout[i] > 0 && Cin[i] > 0 && ks[i] > 0 && ks[i] <= 7,
                 WCMC_ERR_BAD_ARG, "conv2d_wgrad_reduce_multi: bad layer %d", i);
    WCMC_REQUIRE(!db[i] || dy_colsum_partial[i], WCMC_ERR_BAD_ARG,
                 "conv2d_wgrad_reduce_multi: layer %d wants a bias gradient without the column sums of dy", i);
    const XWgradPlan pl = x_plan_wgrad(N[i], Ho[i], Wo[i], Cout[i], Cin[i], ks[i], terms);
    WRMEntry& e = t.e[i];
    const bool fuse_db = db[i] != nullptr;
    e.slabs = (const float*)workspace[i]; e.dw = dw[i]; e.cs_partial = fuse_db ? dy_colsum_partial[i] : nullptr; e.db = db[i];
    e.S = pl.S; e.taps = ks[i] * ks[i]; e.Cout = Cout[i]; e.Cin = Cin[i]; e.Np = pl.Np; e.Cq = pl.Cq;
    e.cs_gmax = x_colsum_rows(N[i], Ho[i], Wo[i]); e.cs_ld = round_up(Cout[i], 16);
    e.gx = (Cin[i] + WR_CI - 1) / WR_CI;
    e.block0 = blocks;
    blocks += (unsigned)e.gx * (unsigned)(Cout[i] + (fuse_db ? (Cout[i] + 63) / 64 : 0));
    size_t l = (size_t)WR_CI * (e.taps + 1) * sizeof(float) + 256 * sizeof(float);
    if (fuse_db && l < (size_t)16 * 64 * sizeof(float)) l = (size_t)16 * 64 * sizeof(float);
    if (l > lds) lds = l;
  }
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, t);
  return check_launch("conv2d_wgrad_reduce_multi");
}

static int g_xigemm_dbuf = -1;      // WCMC_IGEMM_DBUF=0/1 (A/B switch); default: double buffer
template <int NT, bool PADDED, bool DBUF>
static int launch_xigemm3(const XIgemmParams& p, hipStream_t stream) {
  const size_t lds_stage = (size_t)(DBUF ? 2 : 1) * (2 * XBM * XROW + 64 + 2 * NT * 16 * XROW + 64) * sizeof(u16);
  const size_t lds_out = (size_t)XBM * (2 * NT * 16 + 8) * sizeof(u16) + (size_t)16 * NT * 16 * sizeof(float);   // epilogue staging tile + column-sum partials
  const size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
  static LdsAttr attr_set;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_igemm_bf16x3_kernel<NT, PADDED, DBUF>), (size_t)lds, attr_set) != hipSuccess) return WCMC_ERR_LAUNCH;
  const dim3 grid((unsigned)ceil_div64(p.M, XBM), (unsigned)((p.Np / 16 + NT - 1) / NT));
  hipLaunchKernelGGL((conv_igemm_bf16x3_kernel<NT, PADDED, DBUF>), grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_igemm_bf16x3");
}
#ifdef WCMC_DEBUG_BUILD
template <int DBG>
static int launch_xigemm_dbg(const XIgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)2 * (2 * XBM * XROW + 64 + 2 * 7 * 16 * XROW + 64) * sizeof(u16);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3_kernel<7, false, true, DBG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 grid((unsigned)ceil_div64(p.M, XBM), (unsigned)((p.Np / 16 + 6) / 7));
  hipLaunchKernelGGL((conv_igemm_bf16x3_kernel<7, false, true, DBG>), grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_igemm_bf16x3(ablation)");
}
#endif
template <int NT, bool PADDED>
static int launch_xigemm2(const XIgemmParams& p, hipStream_t stream) {
#ifdef WCMC_DEBUG_BUILD
  if (NT == 7 && !PADDED) {       // WCMC_DEBUG_ABLATE=<mask>: timing-only ablation builds of the 5x5 forward GEMM
    static int ab = -1;
    if (ab < 0) { const char* e = ab_env("WCMC_DEBUG_ABLATE"); ab = e ? atoi(e) : 0; }
    switch (ab) {
      case 1: return launch_xigemm_dbg<1>(p, stream);
      case 2: return launch_xigemm_dbg<2>(p, stream);
      case 8: return launch_xigemm_dbg<8>(p, stream);
      case 16: return launch_xigemm_dbg<16>(p, stream);
      case 10: return launch_xigemm_dbg<10>(p, stream);
      case 26: return launch_xigemm_dbg<26>(p, stream);
      case 4: return launch_xigemm_dbg<4>(p, stream);
      case 32: return launch_xigemm_dbg<32>(p, stream);
      case 36: return launch_xigemm_dbg<36>(p, stream);
      case 62: return launch_xigemm_dbg<62>(p, stream);
      case 64: return launch_xigemm_dbg<64>(p, stream);
      default: break;
    }
  }
#endif
  g_xigemm_dbuf = x_env_on("WCMC_IGEMM_DBUF");
  return g_xigemm_dbuf ? launch_xigemm3<NT, PADDED, true>(p, stream) : launch_xigemm3<NT, PADDED, false>(p, stream);
}
template <int NT, int NB, int AP = 2>
static int launch_xhalo2(const XIgemmParams& p, size_t lds, hipStream_t stream) {
  constexpr int TH = 16, TW = 16;
  static LdsAttr attr;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_halo_bf16x3_kernel<NT, TH, TW, 0, NB, AP>), lds, attr) != hipSuccess) return WCMC_ERR_LAUNCH;
  const dim3 grid((unsigned)(p.N * p.tilesX * p.tilesY), (unsigned)((p.Np / 16 + NT - 1) / NT));
  hipLaunchKernelGGL((conv_halo_bf16x3_kernel<NT, TH, TW, 0, NB, AP>), grid, dim3(512), lds, stream, p);
  return check_launch("conv2d_igemm_bf16x3(halo)");
}
template <int NT, int NB, int PT, int PXST, int AP = 2, int WP = 2, int F16 = 0>
static int launch_xhalo64c(const XIgemmParams& p, size_t lds, hipStream_t stream) {
  static LdsAttr attr;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_halo64_bf16x3_kernel<NT, NB, PT, 0, PXST, AP, WP, F16>), lds, attr) != hipSuccess) return WCMC_ERR_LAUNCH;
  const dim3 grid((unsigned)(p.N * (p.tilesX * p.tilesY + (PT == 4 ? p.stripX * p.stripY : 0))), (unsigned)((p.Np / 16 + NT - 1) / NT));
  hipLaunchKernelGGL((conv_halo64_bf16x3_kernel<NT, NB, PT, 0, PXST, AP, WP, F16>), grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_igemm_bf16x3(halo, 64 pixels per wave)");
}
template <int NT, int NB, int PT>
static int launch_xhalo64b(const XIgemmParams& p, size_t lds, hipStream_t stream) {
  // the halo pixel stride as a template constant for the two shipped values (NB = 3: 16-channel slabs, 80 B; NB = 2 with
  // 12x16 tiles: 32-channel slabs, 160 B); anything else (WCMC_HALO64_PXS, WCMC_HALO_NB experiments) reads it from the params
  if constexpr (NT == 7 && NB == 3) {
    if (p.ap == 1 && p.wplanes == 1 && p.f16) return launch_xhalo64c<NT, NB, PT, 80, 1, 1, 1>(p, lds, stream);      // one fp16 MFMA per product ("bf16x321h" output layers)
    if (p.ap == 1 && p.wplanes == 1) return launch_xhalo64c<NT, NB, PT, 80, 1, 1>(p, lds, stream);      // one MFMA per product ("bf16x321o" output layers)
  }
  if constexpr ((NT == 7 || NT == 1) && NB == 3) {
    if (p.ap == 1) return launch_xhalo64c<NT, NB, PT, 80, 1>(p, lds, stream);          // (x_plan_k grants ap = 1 with PXS = 80, ks = 5 only)
  }
  if (p.ks == 5 && p.PXS == 80 && NB == 3) return launch_xhalo64c<NT, NB, PT, NB == 3 ? 80 : 0>(p, lds, stream);
  if (p.ks == 5 && p.PXS == 160 && NB == 2 && PT == 3) return launch_xhalo64c<NT, NB, PT, (NB == 2 && PT == 3) ? 160 : 0>(p, lds, stream);
  return launch_xhalo64c<NT, NB, PT, 0>(p, lds, stream);
}
template <int NT>
static int launch_xhalo64(const XIgemmParams& p0, hipStream_t stream) {
  XIgemmParams p = p0;
  p.SPS |= 0x100;     // alternate the priority of a CU's two workgroups stage by stage (measured neutral on the launch time, kept: it evens the two workgroups' finish times; its A/B switch is gone)
  // Tile height 16 (four pixel tiles per wave) or 12 (three): 512 workgroups are resident (two per CU), a launch takes
  // ceil(workgroups / 512) rounds of a time proportional to the tile height.  The KPCN layers of 100..108 output rows
  // are 392 tiles of 16x16 (one round, a quarter of the slots empty) but 504 of 12x16 (one round of 3/4 the length).
  const int gy = (p.Np / 16 + NT - 1) / NT;
  auto rounds = [&](int th) { return ((int64_t)p.N * p.tilesX * ((p.Ho + th - 1) / th) * gy + 511) / 512 * th; };
  bool pt3 = p.PXS == 160 || (x_env_on("WCMC_HALO64_PT3") && rounds(12) < rounds(16));     // (32-channel slabs: 12x16 only)
  // 16 does not divide Ho (KPCN: 124, 120, 116, 108, 104, 100, 92 rows): a tile rows of 16 followed by b of 12 cover it EXACTLY -- the
  // 16-row instance, whose workgroups of the last b tile rows skip their fourth pixel tile (p.rows16).  Smallest b: as many workgroups
  // as the 16-row tiling and no padded rows (3-7 % fewer MFMAs than the better pure tiling where 12 does not divide Ho either; where
  // it does -- 120, 108 -- the same MFMAs in fewer, taller workgroups: less weight streaming per pixel).  ALONE such a launch is slower
  // (one round of 16-row workgroups where the 12-row tiling ran a shorter round: +2 % per branch); the captured two-stream step is
  // faster by 1.8 %, every one of the seven heights contributing (profiles/r06_step_ab.txt).  Where 16 divides Ho
  // the rule above stands (96 rows: 288 workgroups of 16 rows are slower than 384 of 12, in the step too).
  // Debug build, WCMC_HALO64_MIX: 0 = pure tilings, 1 = mix only where 12 does not divide Ho either; WCMC_HALO64_NOMIX=<Ho>: one
  // height keeps its pure tiling (the per-height A/B).
  p.rows16 = 1 << 20; p.stripX = p.stripY = 0;
  const char* mixe = ab_env("WCMC_HALO64_MIX");
  const bool mix12 = !(mixe && mixe[0] == '1');
  const char* nomix = ab_env("WCMC_HALO64_NOMIX");
  if (p.PXS != 160 && p.Ho % 16 != 0 && (p.Ho % 12 != 0 || mix12) && x_env_on("WCMC_HALO64_MIX") && !(nomix && atoi(nomix) == p.Ho)) {
    for (int b = 1; 12 * b < p.Ho; ++b)
      if ((p.Ho - 12 * b) % 16 == 0) { p.rows16 = (p.Ho - 12 * b) / 16; pt3 = false; break; }
  }
  const bool mixed = p.rows16 != (1 << 20);
  const int th = pt3 ? 12 : 16;
  p.tilesY = mixed ? p.rows16 + (p.Ho - 16 * p.rows16) / 12 : (p.Ho + th - 1) / th;
  const int HP = (th + p.ks - 1) * (16 + p.ks - 1);
  const size_t halo = (size_t)((HP * p.PXS + 127) & ~127), bstage = (size_t)(2 * NT * 16 * XROW + 64) * sizeof(u16);
  const size_t out = p.ys ? (size_t)128 * (2 * NT * 16 + 8) * sizeof(u16) : (size_t)128 * (NT * 16 + 4) * sizeof(float);
#ifdef WCMC_DEBUG_BUILD
  if (NT == 7 && !pt3 && p.PXS == 80 && p.ks == 5) {
    const char* e = ab_env("WCMC_DEBUG_ABLATE");
    const int ab = e ? atoi(e) : 0;
    if (ab) {
      auto kfn = ab == 1 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 1, 80> : ab == 2 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 2, 80>
                 : ab == 4 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 4, 80> : ab == 8 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 8, 80>
                 : ab == 10 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 10, 80> : ab == 14 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 14, 80>
                 : ab == 32 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 32, 80> : ab == 46 ? &conv_halo64_bf16x3_kernel<7, 3, 4, 46, 80>
                 : &conv_halo64_bf16x3_kernel<7, 3, 4, 64, 80>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      const dim3 grid((unsigned)(p.N * p.tilesX * p.tilesY), (unsigned)((p.Np / 16 + NT - 1) / NT));
      hipLaunchKernelGGL(kfn, grid, dim3(256), halo + 3 * bstage, stream, p);
      return check_launch("conv2d_igemm_bf16x3(halo64, debug)");
    }
  }
#endif
  // three weight stages where two workgroups still fit a CU (80 KB each), else two
  const char* nbe = ab_env("WCMC_HALO_NB");
  const int nb = (p.ap == 1 || (!(nbe && nbe[0] == '2') && halo + 3 * bstage <= 80 * 1024)) ? 3 : 2;
  // The width likewise: where 16 does not divide Wo, a tile columns of 16 + b columns of 12 cover it exactly; the b columns are a
  // strip of workgroups that hold their halo transposed (see the kernel: 16 image rows x 12 image columns each, 16-row tile rows) --
  // the 16-row instance only.  (WCMC_HALO64_STRIP=0, debug build: 16-wide tiles throughout.)
  if (!pt3 && p.ks == 5 && p.Wo % 16 != 0 && p.Wo % 4 == 0 && x_env_on("WCMC_HALO64_STRIP")) {
    for (int b = 1; b <= 3; ++b)
      if (12 * b < p.Wo && (p.Wo - 12 * b) % 16 == 0) { p.stripX = b; p.tilesX = (p.Wo - 12 * b) / 16; p.stripY = (p.Ho + 15) / 16; break; }
  }
  const size_t main_ = halo + nb * bstage;
  const size_t lds = main_ > out ? main_ : out;
  if (pt3) return nb == 3 ? launch_xhalo64b<NT, 3, 3>(p, lds, stream) : launch_xhalo64b<NT, 2, 3>(p, lds, stream);
  return nb == 3 ? launch_xhalo64b<NT, 3, 4>(p, lds, stream) : launch_xhalo64b<NT, 2, 4>(p, lds, stream);
}
template <int NT>
static int launch_xhalo(const XIgemmParams& p, hipStream_t stream) {
  if ((p.CS == 16 || p.ap == 1 ||
       (p.CS == 32 && p.PXS == 160 && p.CSl == 32 && p.Kp >= 256 && x_env_on("WCMC_HALO64") && x_env_on("WCMC_HALO64_CS32"))) &&
      p.ks == 5)
    return launch_xhalo64<NT>(p, stream);
  constexpr int TH = 16, TW = 16;
  const int HP = (TH + p.ks - 1) * (TW + p.ks - 1);
  const size_t halo = (size_t)((HP * p.PXS + 127) & ~127), bstage = (size_t)(2 * NT * 16 * XROW + 64) * sizeof(u16);
  const size_t lds_out = p.ys ? (size_t)256 * (2 * NT * 16 + 8) * sizeof(u16) + (size_t)32 * NT * 16 * sizeof(float)
                              : (size_t)256 * (NT * 16 + 4) * sizeof(float);
  // three weight stages (two stages of DMA latency cover) where LDS allows, else two
  const char* nbe = ab_env("WCMC_HALO_NB");
  const int nbmax = (nbe && nbe[0] == '2') ? 2 : 3;
  const int nb = (nbmax >= 3 && halo + 3 * bstage <= 160 * 1024) ? 3 : 2;
  const size_t lds_main = halo + nb * bstage;
  const size_t lds = lds_main > lds_out ? lds_main : lds_out;
  WCMC_REQUIRE(lds <= 160 * 1024, WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: halo tile does not fit in LDS");
#ifdef WCMC_DEBUG_BUILD
  if (NT == 7 && p.PXS == 160 && p.ks == 5) {
    int ab;                             // (read per call: scripts interleave the modes inside one process)
    { const char* e = ab_env("WCMC_DEBUG_ABLATE"); ab = e ? atoi(e) : 0; }
    if (ab == 1 || ab == 2 || ab == 4 || ab == 8 || ab == 16 || ab == 10 || ab == 26 || ab == 18 || ab == 27 || ab == 31 || ab == 59 || ab == 63 || ab == 32) {
      // timing only (WRONG results): 1 = no MFMA, 2 = no weight DMA in the stage loop, 8 = no fragment reads, 16 = no stage
      // barrier, 4 = one halo per tile (no slab reloads), 32 = no epilogue; sums combine (27 = empty stage loop)
      constexpr int TH8 = 8;
      XIgemmParams q = p;
      q.tilesY = (p.Ho + TH8 - 1) / TH8;
      const size_t halo8 = (size_t)(((TH8 + p.ks - 1) * (TW + p.ks - 1) * p.PXS + 127) & ~127);
      const dim3 grid((unsigned)(q.N * q.tilesX * q.tilesY), (unsigned)((q.Np / 16 + NT - 1) / NT));
      auto kfn = ab == 4 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 4, 2> : ab == 1 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 1, 2> : ab == 2 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 2, 2>
                 : ab == 8 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 8, 2> : ab == 16 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 16, 2>
                 : ab == 10 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 10, 2> : ab == 18 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 18, 2>
                 : ab == 27 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 27, 2> : ab == 31 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 31, 2>
                 : ab == 59 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 59, 2> : ab == 63 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 63, 2>
                 : ab == 32 ? &conv_halo_bf16x3_kernel<7, TH8, TW, 32, 2>
                 : &conv_halo_bf16x3_kernel<7, TH8, TW, 26, 2>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      hipLaunchKernelGGL(kfn, grid, dim3(TH8 * TW * 2), halo8 + 2 * bstage, stream, q);
      return check_launch("conv2d_igemm_bf16x3(halo 8x16, ablation)");
    }
    if (ab == 64) {      // stamp build of the shipped 8x16 tiling (scripts/stamp_igemm.py)
      constexpr int TH8 = 8;
      XIgemmParams q = p;
      q.tilesY = (p.Ho + TH8 - 1) / TH8;
      const size_t halo8 = (size_t)(((TH8 + p.ks - 1) * (TW + p.ks - 1) * p.PXS + 127) & ~127);
      const dim3 grid((unsigned)(q.N * q.tilesX * q.tilesY), (unsigned)((q.Np / 16 + NT - 1) / NT));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_bf16x3_kernel<7, TH8, TW, 64, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      hipLaunchKernelGGL((conv_halo_bf16x3_kernel<7, TH8, TW, 64, 2>), grid, dim3(TH8 * TW * 2), halo8 + 2 * bstage, stream, q);
      return check_launch("conv2d_igemm_bf16x3(halo 8x16, stamps)");
    }
  }
#endif
#ifdef WCMC_DEBUG_BUILD
  if (NT == 4 && p.ks == 3) {        // wall-clock stamps of the U-Net 3x3 launches, 8x16 tiling (scripts/timeline_halo.py --unet)
    const char* e = ab_env("WCMC_DEBUG_ABLATE");
    if (e && atoi(e) == 64) {
      constexpr int TH8 = 8;
      XIgemmParams q = p;
      q.tilesY = (p.Ho + TH8 - 1) / TH8;
      const size_t halo8 = (size_t)(((TH8 + p.ks - 1) * (TW + p.ks - 1) * p.PXS + 127) & ~127);
      const size_t out8 = (size_t)TH8 * TW * (2 * NT * 16 + 8) * sizeof(u16) + (size_t)32 * NT * 16 * sizeof(float);
      const size_t main8 = halo8 + 2 * bstage;
      const dim3 grid((unsigned)(q.N * q.tilesX * q.tilesY), (unsigned)((q.Np / 16 + NT - 1) / NT));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_bf16x3_kernel<4, TH8, TW, 64, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      hipLaunchKernelGGL((conv_halo_bf16x3_kernel<4, TH8, TW, 64, 2>), grid, dim3(TH8 * TW * 2), main8 > out8 ? main8 : out8, stream, q);
      return check_launch("conv2d_igemm_bf16x3(halo 8x16 3x3, stamps)");
    }
  }
#endif
  // ... and for any launch whose 16x16 tiling has fewer workgroups than the chip has CUs (the deepest U-Net level: 32 tiles
  // x 4 cout blocks), where half-size tiles simply fill the machine (<= 4 cout tiles: one wave per weight row group pair)
  const int64_t blocks16 = (int64_t)p.N * p.tilesX * p.tilesY * ((p.Np / 16 + NT - 1) / NT);
  // ... and, measured (scripts/time_unet_layers.py), where the 16x16 tiling is two or more rounds (the 128^2 level: 512
  // tiles): two 128-pixel workgroups per CU with their own stage barriers instead of one of 256 -- 46.8 -> 43.5 us
  const bool underfilled = NT <= 4 && (blocks16 < 256 || blocks16 >= 512) && p.Ho >= 16;
  if ((p.PXS == 160 && p.ks == 5) || underfilled) {
    // WCMC_HALO_TH8_5X5 plan (32-channel slabs): 8x16-pixel tiles, four waves, TWO workgroups per CU -- their stage
    // barriers are independent, so the non-MFMA phases of one hide behind the MFMAs of the other
    constexpr int TH8 = 8;
    XIgemmParams q = p;
    q.tilesY = (p.Ho + TH8 - 1) / TH8;
    const size_t halo8 = (size_t)(((TH8 + p.ks - 1) * (TW + p.ks - 1) * p.PXS + 127) & ~127);
    const size_t out8 = p.ys ? (size_t)TH8 * TW * (2 * NT * 16 + 8) * sizeof(u16) + (size_t)32 * NT * 16 * sizeof(float)
                             : (size_t)TH8 * TW * (NT * 16 + 4) * sizeof(float);
    // (three weight stages, which still fit beside the second workgroup for <= 4 cout tiles, measured no faster on the
    // U-Net's 3x3 layers, nor five in the 16x16 tiling: wall-clock stamps show 15 us in the stage loop of 64 -> 64 at
    // 128^2 for 7 us of MFMAs, but the DMA is not what the stages wait for -- scripts/timeline_halo.py --unet)
    const size_t main8 = halo8 + 2 * bstage;
    const size_t lds8 = main8 > out8 ? main8 : out8;
    const dim3 grid((unsigned)(q.N * q.tilesX * q.tilesY), (unsigned)((q.Np / 16 + NT - 1) / NT));
    if constexpr (NT == 4 || NT == 7) {
      if (p.ap == 1) {                           // (x_plan_k grants ap = 1 to this kernel for ks = 3 and NT = 4 or 7 only)
        static LdsAttr attr81;
        if (set_max_lds(reinterpret_cast<const void*>(&conv_halo_bf16x3_kernel<NT, TH8, TW, 0, 2, 1>), lds8, attr81) != hipSuccess) return WCMC_ERR_LAUNCH;
        hipLaunchKernelGGL((conv_halo_bf16x3_kernel<NT, TH8, TW, 0, 2, 1>), grid, dim3(TH8 * TW * 2), lds8, stream, q);
        return check_launch("conv2d_igemm_bf16x3(halo, 8x16, x hi plane)");
      }
    }
    static LdsAttr attr8;
    if (set_max_lds(reinterpret_cast<const void*>(&conv_halo_bf16x3_kernel<NT, TH8, TW, 0, 2>), lds8, attr8) != hipSuccess) return WCMC_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_halo_bf16x3_kernel<NT, TH8, TW, 0, 2>), grid, dim3(TH8 * TW * 2), lds8, stream, q);
    return check_launch("conv2d_igemm_bf16x3(halo, 8x16)");
  }
  if constexpr (NT == 4 || NT == 7) {
    if (p.ap == 1) return nb == 3 ? launch_xhalo2<NT, 3, 1>(p, lds, stream) : launch_xhalo2<NT, 2, 1>(p, lds, stream);
  }
  return nb == 3 ? launch_xhalo2<NT, 3>(p, lds, stream) : launch_xhalo2<NT, 2>(p, lds, stream);
}
// conv_halo3_bf16x3_kernel (3x3, K split over two wave groups): 64-cout blocks, slabs of exactly 64 channels (two planes) or
// 64 / 128 channels (hi plane only) -- every U-Net layer of support/networks.py:20-22 in both directions
static bool x_halo3_ok(const XIgemmParams& p) {
  if (p.ks != 3 || !p.PXS || p.Np % 64 != 0 || !x_env_on("WCMC_HALO3")) return false;
  if (p.ap == 2) return p.CS == 64 && p.CSl == 64 && p.SPS == 18 && p.SPSl == 18;
  return (p.CS == 64 && p.nslabs == 1 && p.SPSl == 18) || (p.CS == 128 && p.CSl == 128 && p.SPS == 36 && p.SPSl == 36);
}
template <int AP, int SPT>
static int launch_xhalo3b(const XIgemmParams& q, hipStream_t stream) {
  constexpr int KG = 2;
  const size_t lds = (size_t)180 * 256 + (size_t)2 * KG * (2 * 64 * XROW + 64) * sizeof(u16);      // 79,360 B: two workgroups per CU
  static LdsAttr attr;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_halo3_bf16x3_kernel<AP, SPT, KG>), lds, attr) != hipSuccess) return WCMC_ERR_LAUNCH;
  const dim3 grid((unsigned)(q.N * q.tilesX * q.tilesY), (unsigned)(q.Np / 64));
#ifdef WCMC_DEBUG_BUILD
  {                                       // WCMC_HALO3_VALU=1: CORRECT results, 128 more vector instructions per wave (what is a VALU worth in the step?)
    const char* e = ab_env("WCMC_HALO3_VALU");
    if (e && e[0] == '1') {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo3_bf16x3_kernel<AP, SPT, KG, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((conv_halo3_bf16x3_kernel<AP, SPT, KG, 256>), grid, dim3(256 * KG), lds, stream, q);
      return check_launch("conv2d_igemm_bf16x3(halo 3x3, +128 VALU)");
    }
  }
  if (AP == 2) {                          // WCMC_DEBUG_ABLATE=<mask>: timing-only ablations of the forward instance (scripts/time_unet_abl.py)
    const char* e = ab_env("WCMC_DEBUG_ABLATE");
    const int ab = e ? atoi(e) : 0;
    if (ab) {
      auto kfn = ab == 1 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 1> : ab == 2 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 2> : ab == 8 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 8>
                 : ab == 16 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 16> : ab == 32 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 32> : ab == 10 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 10>
                 : ab == 26 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 26> : ab == 27 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 27> : ab == 59 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 59>
                 : ab == 33 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 33> : ab == 18 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 18> : ab == 64 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 64>
                 : ab == 128 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 128> : ab == 192 ? &conv_halo3_bf16x3_kernel<2, 2, 2, 192> : &conv_halo3_bf16x3_kernel<2, 2, 2, 9>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, grid, dim3(256 * KG), lds, stream, q);
      return check_launch("conv2d_igemm_bf16x3(halo 3x3, ablation)");
    }
  }
#endif
  hipLaunchKernelGGL((conv_halo3_bf16x3_kernel<AP, SPT, KG>), grid, dim3(256 * KG), lds, stream, q);
  return check_launch("conv2d_igemm_bf16x3(halo 3x3, K groups)");
}
static int launch_xhalo3(const XIgemmParams& p, hipStream_t stream) {
  XIgemmParams q = p;
  q.tilesY = (p.Ho + 7) / 8;
  q.PXS = 256;
  if (p.ap == 2) return launch_xhalo3b<2, 2>(q, stream);
  return p.CS == 128 ? launch_xhalo3b<1, 4>(q, stream) : launch_xhalo3b<1, 2>(q, stream);
}
template <int NT>
static int launch_xigemm(const XIgemmParams& p, hipStream_t stream) {
  if (p.PXS) return launch_xhalo<NT>(p, stream);
  return p.pad > 0 ? launch_xigemm2<NT, true>(p, stream) : launch_xigemm2<NT, false>(p, stream);
}

extern "C" int wcmc_conv2d_igemm_bf16x3(const void* x_split, int N, int H, int W, int Cin, const void* wp,
                                        const float* bias, float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                                        void* y_split, int Cout, int ks, int pad, int act, float slope,
                                        const void* gate_split, int gate_act, float gate_slope,
                                        float* colsum_partial, const void* gate_mask, void* mask_out,
                                        int terms, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks > 0 && pad >= 0 && x_split && wp,
               WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: bad argument");
  WCMC_REQUIRE(terms >= 1 && terms <= 3, WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: terms must be 3, 2 (x hi plane only; wp packed with mode 2 / 3) or 1 (hi planes of x and W only)");
  WCMC_REQUIRE(!colsum_partial || y_split, WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: column sums are produced with the split output only");
  WCMC_REQUIRE((y != nullptr) != (y_split != nullptr), WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: exactly one of y (fp32 view) and y_split must be given");
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: empty output");
  WCMC_REQUIRE(aligned16(x_split) && aligned16(wp) && (!y_split || aligned16(y_split)) &&
                   (!gate_split || aligned16(gate_split)),
               WCMC_ERR_ALIGNMENT, "conv2d_igemm_bf16x3: split buffers must be 16-byte aligned");
  WCMC_REQUIRE(!y || nhwc_view_ok(y, ysn, ysh, ysw, Cout), WCMC_ERR_ALIGNMENT,
               "conv2d_igemm_bf16x3: y violates the NHWC-view contract");
  WCMC_REQUIRE((!gate_split && !gate_mask && !mask_out) || y_split, WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: a gate / a mask requires the split output geometry");
  WCMC_REQUIRE(!(gate_split && gate_mask), WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: gate_split and gate_mask are exclusive");
  XIgemmParams p;
  p.x = (const u16*)x_split; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cpi = round_up(Cin, 8);
  p.wp = (const u16*)wp; p.bias = bias;
  p.yf = y; p.ysn = ysn; p.ysh = ysh; p.ysw = ysw;
  p.ys = (u16*)y_split; p.Cpo = y_split ? round_up(Cout, 8) : round_up(Cout, 4);
  p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.gate = (const u16*)gate_split; p.gate_act = gate_act; p.gate_slope = gate_slope;
  p.gate_mask = (const unsigned char*)gate_mask; p.mask_out = (unsigned char*)mask_out;
  p.ks = ks; p.pad = pad; p.act = act; p.slope = slope;
  const XKPlan q = x_plan_k(Cin, ks, terms <= 2 ? 1 : 2, Cout);
  p.ap = q.ap;
  // one term: where the plan grants the hi-plane instance of the 64-pixel 5x5 kernel (the only one with a one-plane weight path);
  // anywhere else the launch multiplies what the plan's instance multiplies (two or three terms) -- more exact, never less
  p.wplanes = (terms == 1 && q.ap == 1 && ks == 5 && q.PXS == 80 && x_pick_nt(round_up(Cout, 16) / 16) == 7) ? 1 : 2;
  p.f16 = 0;
  p.Kp = p.Cpi; p.Kt = q.Kt; p.Np = round_up(Cout, 16);
  p.CS = q.CS; p.nslabs = q.nslabs; p.SPS = q.Ks / 32; p.PXS = q.halo ? q.PXS : 0;
  p.CSl = q.CSl; p.SPSl = q.Ksl / 32;
  p.tilesY = (Ho + 15) / 16; p.tilesX = (Wo + 15) / 16;
  p.G = x_colsum_rows(N, Ho, Wo);
  p.M = (int64_t)N * Ho * Wo;
  const size_t xb = wcmc_split_elems(N, H, W, Cin) * sizeof(u16), wb = (size_t)p.Np * 2 * p.Kt * sizeof(u16);
  WCMC_REQUIRE(xb < 0x7ff00000u && wb < 0x7ff00000u, WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: operand larger than 2 GiB (split the batch)");
  // (the halo kernels push the weight-DMA offsets of stages past the end out of range by adding 2^30: see dma_b)
  WCMC_REQUIRE(!q.halo || wb < 0x40000000u, WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: packed weights of 1 GiB or more");
  p.x_bytes = (unsigned)xb; p.wp_bytes = (unsigned)wb;
  p.colsum = colsum_partial;
#ifdef WCMC_DEBUG_BUILD
  {  // timing-only experiments (guide section 7: zero-record descriptors drop one operand's traffic)
    static int dbg = -1;
    if (dbg < 0) { const char* e = ab_env("WCMC_DEBUG_DROP"); dbg = e ? atoi(e) : 0; }
    if (dbg & 1) p.x_bytes = 0;
    if (dbg & 2) p.wp_bytes = 0;
  }
#endif
  hipStream_t st = (hipStream_t)stream;
  p.y_bytes = 0; p.m_bytes = 0;
  p.wp2 = nullptr; p.bias2 = nullptr; p.y2 = nullptr; p.y2sn = p.y2sh = p.y2sw = 0; p.Cout2 = 0; p.act2 = 0; p.Kt2 = 0;
  p.slope2 = 0.f; p.wp2_bytes = p.y2_bytes = 0;
  {
    int ntw = 0, u = 0;
    if (x_plan_pw(p, &ntw, &u)) return launch_xpw(p, ntw, u, st);
  }
  if (x_halo3_ok(p)) return launch_xhalo3(p, st);
  switch (x_pick_nt(p.Np / 16)) {
    case 7: return launch_xigemm<7>(p, st);
    case 4: return launch_xigemm<4>(p, st);
    case 2: return launch_xigemm<2>(p, st);
    default: return launch_xigemm<1>(p, st);
  }
}

// ---- the fp16 one-MFMA forward of an un-gated 5x5 output layer ("bf16x321h" mode; profiles/r04_forward_ladder.txt, table "last", rung E)
static bool x_out_f16_plan(int Cin, int Cout, int ks, XKPlan* q) {
  if (ks != 5 || Cin <= 0 || Cout <= 0) return false;
  *q = x_plan_k(Cin, ks, 1, Cout);
  return q->ap == 1 && q->halo && q->PXS == 80 && x_pick_nt(round_up(Cout, 16) / 16) == 7;
}
extern "C" int wcmc_conv2d_out_f16_supported(int Cin, int Cout, int ks) {
  XKPlan q;
  return x_out_f16_plan(Cin, Cout, ks, &q) ? 1 : 0;
}
extern "C" size_t wcmc_split_to_f16_elems(int N, int H, int W, int C) {
  return (N > 0 && H > 0 && W > 0 && C > 0) ? (size_t)N * H * W * round_up(C, 8) : 0;
}
extern "C" int wcmc_split_to_f16(const void* x_split, int N, int H, int W, int C, void* out_f16, void* stream) {
  WCMC_REQUIRE(x_split && out_f16 && N > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG, "split_to_f16: bad argument");
  WCMC_REQUIRE(aligned16(x_split) && aligned16(out_f16), WCMC_ERR_ALIGNMENT, "split_to_f16: buffers must be 16-byte aligned");
  const int Cp = round_up(C, 8);
  const int64_t total = (int64_t)N * H * W * (Cp / 8);
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(split_to_f16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)x_split, (u16*)out_f16, Cp, total);
  return check_launch("split_to_f16");
}
extern "C" int wcmc_conv2d_out_f16(const void* x_f16, int N, int H, int W, int Cin, const void* wp_f16, const float* bias, float* y,
                                   int64_t ysn, int64_t ysh, int64_t ysw, int Cout, int ks, int pad, void* stream) {
  WCMC_REQUIRE(x_f16 && wp_f16 && y && N > 0 && H > 0 && W > 0 && pad >= 0, WCMC_ERR_BAD_ARG, "conv2d_out_f16: bad argument");
  XKPlan q;
  WCMC_REQUIRE(x_out_f16_plan(Cin, Cout, ks, &q), WCMC_ERR_BAD_ARG,
               "conv2d_out_f16: no fp16 instance for this shape (ask wcmc_conv2d_out_f16_supported; 5x5, cout blocks of seven tiles)");
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_out_f16: empty output");
  WCMC_REQUIRE(aligned16(x_f16) && aligned16(wp_f16) && nhwc_view_ok(y, ysn, ysh, ysw, Cout), WCMC_ERR_ALIGNMENT,
               "conv2d_out_f16: unaligned operand or y violates the NHWC-view contract");
  XIgemmParams p = {};
  p.x = (const u16*)x_f16; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cpi = round_up(Cin, 8);
  p.wp = (const u16*)wp_f16; p.bias = bias;
  p.yf = y; p.ysn = ysn; p.ysh = ysh; p.ysw = ysw; p.ys = nullptr; p.Cpo = round_up(Cout, 4);
  p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.gate = nullptr; p.gate_act = WCMC_ACT_LINEAR; p.gate_slope = 0.f; p.gate_mask = nullptr; p.mask_out = nullptr;
  p.ks = ks; p.pad = pad; p.act = WCMC_ACT_LINEAR; p.slope = 0.f;
  p.ap = 1; p.wplanes = 1; p.f16 = 1;
  p.Kp = p.Cpi; p.Kt = q.Kt; p.Np = round_up(Cout, 16);
  p.CS = q.CS; p.nslabs = q.nslabs; p.SPS = q.Ks / 32; p.PXS = q.PXS; p.CSl = q.CSl; p.SPSl = q.Ksl / 32;
  p.tilesY = (Ho + 15) / 16; p.tilesX = (Wo + 15) / 16;
  p.G = x_colsum_rows(N, Ho, Wo); p.M = (int64_t)N * Ho * Wo;
  const size_t xb = (size_t)N * H * W * p.Cpi * sizeof(u16), wb = (size_t)p.Np * 2 * p.Kt * sizeof(u16);
  WCMC_REQUIRE(xb < 0x7ff00000u && wb < 0x40000000u, WCMC_ERR_BAD_ARG, "conv2d_out_f16: operand too large (split the batch)");
  p.x_bytes = (unsigned)xb; p.wp_bytes = (unsigned)wb; p.colsum = nullptr;
  return launch_xigemm<7>(p, (hipStream_t)stream);
}

static bool x_pair_enabled() {
  const char* e = ab_env("WCMC_IGEMM_PW");
  const char* t = ab_env("WCMC_PW_TAIL");       // WCMC_PW_TAIL=0: A/B switch back to two launches
  return !(e && e[0] == '0') && !(t && t[0] == '0');
}
// fused instances: (input units, couts of the first layer, tail kind)
static int x_pair_kind(int Cin, int Cout1, int Cout2) {
  const int cpi = round_up(Cin, 8);
  if (cpi == 128 && Cout1 == 128 && Cout2 >= 1 && Cout2 <= 4) return 1;     // PathNet.final forward: 128 -> 128 -> 3
  if (cpi == 8 && Cout1 == 128 && Cout2 == 128) return 2;                    // its data gradient: 3 -> 128 -> 128
  if (cpi == 64 && Cout1 == 64 && Cout2 == 64) return 3;                     // PathNet.embedding forward: 64 -> 64 -> 64
  return 0;
}

extern "C" int wcmc_conv1x1_pair_supported(int Cin, int Cout1, int Cout2) {
  return x_pair_enabled() && x_pair_kind(Cin, Cout1, Cout2) != 0;
}

extern "C" int wcmc_conv1x1_pair_bf16x3(const void* x_split, int N, int H, int W, int Cin, const void* wp1,
                                        const float* bias1, int Cout1, int act1, float slope1, void* y1_split,
                                        void* mask1, const void* gate_mask1, int gate_act1, float gate_slope1,
                                        float* colsum1, const void* wp2, const float* bias2, int Cout2, int act2,
                                        float slope2, float* y2, int64_t y2sn, int64_t y2sh, int64_t y2sw, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && x_split && wp1 && wp2 && y1_split && y2, WCMC_ERR_BAD_ARG,
               "conv1x1_pair_bf16x3: bad argument");
  const int kind = x_pair_enabled() ? x_pair_kind(Cin, Cout1, Cout2) : 0;
  WCMC_REQUIRE(kind != 0, WCMC_ERR_BAD_ARG, "conv1x1_pair_bf16x3: no fused instance for %d -> %d -> %d channels", Cin,
               Cout1, Cout2);
  WCMC_REQUIRE(aligned16(x_split) && aligned16(wp1) && aligned16(wp2) && aligned16(y1_split), WCMC_ERR_ALIGNMENT,
               "conv1x1_pair_bf16x3: split buffers must be 16-byte aligned");
  WCMC_REQUIRE(nhwc_view_ok(y2, y2sn, y2sh, y2sw, Cout2), WCMC_ERR_ALIGNMENT, "conv1x1_pair_bf16x3: y2 violates the NHWC-view contract");
  XIgemmParams p;
  p.x = (const u16*)x_split; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cpi = round_up(Cin, 8);
  p.wp = (const u16*)wp1; p.bias = bias1;
  p.yf = nullptr; p.ysn = p.ysh = p.ysw = 0;
  p.ys = (u16*)y1_split; p.Cpo = round_up(Cout1, 8);
  p.Ho = H; p.Wo = W; p.Cout = Cout1;
  p.gate = nullptr; p.gate_act = gate_act1; p.gate_slope = gate_slope1; p.gate_mask = (const unsigned char*)gate_mask1;
  p.mask_out = (unsigned char*)mask1;
  p.ks = 1; p.pad = 0; p.act = act1; p.slope = slope1;
  p.Kp = p.Cpi; p.Kt = round_up(p.Cpi, 32); p.Np = round_up(Cout1, 16);
  p.CS = p.Kp; p.nslabs = 1; p.SPS = p.Kt / 32; p.PXS = 0; p.CSl = p.CS; p.SPSl = p.SPS; p.ap = 2; p.wplanes = 2; p.f16 = 0;
  p.tilesY = p.tilesX = 0; p.G = x_colsum_rows(N, H, W); p.colsum = colsum1;
  p.M = (int64_t)N * H * W;
  const int cp2 = round_up(Cout2, 4);
  const int64_t xb = p.M * 4 * p.Cpi, yb = p.M * 4 * p.Np;
  const int64_t y2b = ((int64_t)(N - 1) * y2sn + (int64_t)(H - 1) * y2sh + (int64_t)(W - 1) * y2sw + cp2) * 4;
  WCMC_REQUIRE(xb < 0x7ff00000LL && yb < 0x7ff00000LL && y2b < 0x7ff00000LL && y2sn >= 0 && y2sh >= 0 && y2sw >= cp2,
               WCMC_ERR_BAD_ARG, "conv1x1_pair_bf16x3: operand larger than 2 GiB (split the batch)");
  p.x_bytes = (unsigned)xb; p.wp_bytes = (unsigned)((size_t)p.Np * 2 * p.Kt * sizeof(u16));
  p.y_bytes = (unsigned)yb; p.m_bytes = (unsigned)(p.M * (p.Np / 8));
  p.wp2 = (const u16*)wp2; p.bias2 = bias2; p.y2 = y2; p.y2sn = y2sn; p.y2sh = y2sh; p.y2sw = y2sw;
  p.Cout2 = Cout2; p.act2 = act2; p.slope2 = slope2; p.Kt2 = p.Np;
  p.wp2_bytes = (unsigned)((size_t)round_up(Cout2, 16) * 2 * p.Kt2 * sizeof(u16)); p.y2_bytes = (unsigned)y2b;
  hipStream_t st = (hipStream_t)stream;
  if (kind == 1) return launch_xpw2<8, 32, true, 1>(p, st);
  if (kind == 2) return launch_xpw2<8, 2, true, 2>(p, st);
  return launch_xpw2<4, 16, true, 2>(p, st);
}

extern "C" size_t wcmc_conv2d_igemm_colsum_elems(int N, int Ho, int Wo, int Cout) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return 0;
  return (size_t)x_colsum_rows(N, Ho, Wo) * round_up(Cout, 16) + 4;      // + trailer: rows the producing launch wrote
}

extern "C" int wcmc_colsum_finish(const float* partial, int N, int Ho, int Wo, int Cout, float* db, void* stream) {
  WCMC_REQUIRE(partial && db && N > 0 && Ho > 0 && Wo > 0 && Cout > 0, WCMC_ERR_BAD_ARG, "colsum_finish: bad argument");
  const int G = x_colsum_rows(N, Ho, Wo), Np = round_up(Cout, 16);
  // partial rows are Np wide: reduce the first Cout columns of each
  hipLaunchKernelGGL(colsum_final_strided_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(1024), 0,
                     (hipStream_t)stream, partial, G, Np, Cout, db);
  return check_launch("colsum_finish");
}

extern "C" size_t wcmc_conv2d_wgrad_bf16x3_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int ks) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin <= 0 || ks <= 0) return 0;
  const size_t b3 = x_plan_wgrad(N, Ho, Wo, Cout, Cin, ks, 3).bytes, b1 = x_plan_wgrad(N, Ho, Wo, Cout, Cin, ks, 1).bytes;
  return b3 > b1 ? b3 : b1;                              // (enough for either number of terms)
}

template <int TM, int PL = 2>
static int launch_xwgrad(const XWgradParams& p, hipStream_t stream) {
  constexpr size_t lds_stage = (size_t)PL * 64 * (xw_stride(TM * 16) + xw_stride(64)) * sizeof(u16);
  constexpr size_t lds_red = (size_t)TM * 16 * (64 + 4) * sizeof(float);
  constexpr size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
  const int per_split = p.ks * p.ks * p.coBlocks * p.ciBlocks;
  const dim3 grid((unsigned)(((p.S + 7) / 8) * 8 * per_split));
  hipLaunchKernelGGL((conv_wgrad_bf16x3_kernel<TM, PL>), grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_wgrad_bf16x3");
}

extern "C" int wcmc_conv2d_wgrad_bf16x3(const void* x_split, int N, int H, int W, int Cin, const void* dy_split,
                                        int Cout, int ks, int pad, float* dw, float* db, void* workspace,
                                        size_t workspace_bytes, int phase, const float* dy_colsum_partial, int terms,
                                        void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks > 0 && pad >= 0 && dw && workspace &&
                   x_split && dy_split,
               WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: bad argument");
  WCMC_REQUIRE(terms == 3 || terms == 1, WCMC_ERR_BAD_ARG,
               "conv2d_wgrad_bf16x3: terms must be 3 (hi*hi + hi*lo + lo*hi) or 1 (the hi planes only)");
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: empty output");
  WCMC_REQUIRE(aligned16(x_split) && aligned16(dy_split), WCMC_ERR_ALIGNMENT,
               "conv2d_wgrad_bf16x3: split buffers must be 16-byte aligned");
  const XWgradPlan pl = x_plan_wgrad(N, Ho, Wo, Cout, Cin, ks, terms);
  WCMC_REQUIRE(workspace_bytes >= pl.bytes && aligned16(workspace), WCMC_ERR_WORKSPACE,
               "conv2d_wgrad_bf16x3: workspace %zu < %zu bytes (or unaligned)", workspace_bytes, pl.bytes);
  hipStream_t st = (hipStream_t)stream;
  XWgradParams p;
  p.x = (const u16*)x_split; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cpi = round_up(Cin, 8);
  p.dy = (const u16*)dy_split; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.Cpo = round_up(Cout, 8);
  p.ks = ks; p.pad = pad; p.slabs = (float*)workspace; p.S = pl.S; p.M = (int64_t)N * Ho * Wo;
  p.pix_per_split = pl.pix_per_split; p.Np = pl.Np; p.Cq = pl.Cq; p.coBlocks = pl.coBlocks; p.ciBlocks = pl.ciBlocks;
  const size_t xb = wcmc_split_elems(N, H, W, Cin) * sizeof(u16), yb = wcmc_split_elems(N, Ho, Wo, Cout) * sizeof(u16);
  WCMC_REQUIRE(xb < 0x7ff00000u && yb < 0x7ff00000u, WCMC_ERR_BAD_ARG,
               "conv2d_wgrad_bf16x3: operand larger than 2 GiB (split the batch)");
  p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
  p.xps = 4 * p.Cpi; p.yps = 4 * p.Cpo;
  WCMC_REQUIRE(phase >= 0 && phase <= 2, WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: phase must be 0, 1 or 2");
  int rc = 0;
  if (phase != 2 && pl.rows) {
    XWRowsParams q;
    q.x = p.x; q.N = N; q.H = H; q.W = W; q.Cpi = p.Cpi; q.dy = p.dy; q.Ho = Ho; q.Wo = Wo; q.Cpo = p.Cpo;
    q.dbg = (float*)workspace + pl.slab_elems;
    { const char* e = ab_env("WCMC_WGRAD_ROWS8_PRIO"); q.prio = e ? atoi(e) : 8; if (q.prio < 0 || q.prio > 13) q.prio = 0; }   // (scripts/time_wgrad_rows8.py: 6-8 of 14 best)
    q.pad = pad; q.slabs = p.slabs; q.S = pl.S; q.rps = pl.rps; q.R = pl.R; q.Np = pl.Np; q.Cq = pl.Cq;
    q.coBlocks = pl.coBlocks; q.ciBlocks = pl.ciBlocks; q.x_bytes = p.x_bytes; q.dy_bytes = p.dy_bytes;
    q.xps = p.xps; q.yps = p.yps;
    const int key = (terms == 1 ? 1000 : 0) + ks * 100 + pl.rTM * 10 + pl.rNW;
    switch (key) {
      case 577: rc = launch_xwgrad_rows<5, 7, 7>(q, st); break;
      case 573: rc = launch_xwgrad_rows<5, 7, 3>(q, st); break;
      case 388: rc = launch_xwgrad_rows<3, 8, 8>(q, st); break;
      case 344: rc = launch_xwgrad_rows<3, 4, 4>(q, st); break;
      case 188: rc = launch_xwgrad_rows<1, 8, 8>(q, st); break;
      case 1577: rc = launch_xwgrad_rows<5, 7, 7, 1>(q, st); break;
      case 1573: rc = launch_xwgrad_rows<5, 7, 3, 1>(q, st); break;
      case 1388: rc = launch_xwgrad_rows<3, 8, 8, 1>(q, st); break;
      case 1344: rc = launch_xwgrad_rows<3, 4, 4, 1>(q, st); break;
      case 1188: rc = launch_xwgrad_rows<1, 8, 8, 1>(q, st); break;
      default: WCMC_REQUIRE(false, WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: no filter-row instance for the plan");
    }
  } else if (phase != 2) {
    if (terms == 1) rc = pl.TM == 7 ? launch_xwgrad<7, 1>(p, st) : launch_xwgrad<4, 1>(p, st);
    else rc = pl.TM == 7 ? launch_xwgrad<7>(p, st) : launch_xwgrad<4>(p, st);
  }
  if (rc || phase == 1) return rc;
  // the slab reduction; with the column sums of dy at hand its launch also finishes the bias gradient (extra grid rows)
  const bool fuse_db = db && dy_colsum_partial;
  const int cs_rows = fuse_db ? (Cout + 63) / 64 : 0;
  size_t red_lds = (size_t)WR_CI * (ks * ks + 1) * sizeof(float) + 256 * sizeof(float);   // (+ the group sums of the few-tap path)
  if (fuse_db && red_lds < (size_t)16 * 64 * sizeof(float)) red_lds = (size_t)16 * 64 * sizeof(float);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cin + WR_CI - 1) / WR_CI), (unsigned)(Cout + cs_rows)), dim3(256),
                     red_lds, st, p.slabs, dw, pl.S, ks * ks, Cout, Cin, pl.Np, pl.Cq, fuse_db ? dy_colsum_partial : nullptr,
                     x_colsum_rows(N, Ho, Wo), round_up(Cout, 16), db);
  rc = check_launch("conv2d_wgrad_bf16x3_reduce");
  if (rc || !db || fuse_db) return rc;
  float* partial = (float*)workspace + pl.slab_elems;
  WCMC_REQUIRE(p.Cpo / 8 <= 256, WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: Cout > 2048 unsupported");
  hipLaunchKernelGGL(colsum_split_kernel, dim3((unsigned)pl.G), dim3(256), (size_t)256 * 8 * sizeof(float), st, p.dy,
                     p.Cpo, Cout, p.M, pl.per_block, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(1024), 0, st, partial, pl.G, Cout,
                     db);
  return check_launch("conv2d_bias_grad_bf16x3");
}
