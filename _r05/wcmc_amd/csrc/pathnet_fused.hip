// PathNet.embedding as ONE launch per direction.
//
//   support/networks.py:33-36:  y = ConvChain(36 -> 64 -> 64 -> 64, ksize 1, ReLU, ReLU, linear)(paths.view(B*S, 36, H, W))
//
// The layer-by-layer path moves the hidden activations through HBM four times per direction (268 MB each at the benchmark
// shape: written split, re-read by the next layer, re-read by its data gradient and by its weight gradient) -- 2.5 GB per
// backbone and backward for 0.17 GB of input.  Here a workgroup owns 64-pixel tiles and keeps everything between the
// input and the output of the chain in LDS:
//
//   forward   x tile (split, 160 B / pixel) -> h0 -> h1 -> y (fp32, 256 B / pixel); nothing else is written.  Same MFMA
//             sequence per output as the layer-by-layer kernels (three bf16 MFMAs per product, small terms first; bias,
//             ReLU and the hi / lo split between the layers as conv_pw_bf16x3_kernel does them): y is BIT-IDENTICAL.
//   backward  RECOMPUTES h0, h1 from the x tile (two GEMMs: cheaper than 2 x 268 MB of reads), forms
//             dy = g_y + repeat_S(g_mean) / S on the fly, then dh1 = (W2^T dy) . [h1 > 0], dh0 = (W1^T dh1) . [h0 > 0] (two
//             MFMAs per product: dy_hi x (W_hi + W_lo), the data-gradient rung of the default mode) and accumulates the
//             three weight gradients (one MFMA per product: hi x hi, pixels on the k axis through the transposing LDS
//             read) and the three bias gradients in REGISTERS across all the tiles of the workgroup; per-workgroup
//             partials leave once, a fixed-order finish kernel sums them.  The gradient with respect to x is never formed
//             (paths is data).  Reads x + g_y (+ g_mean): 0.47 GB instead of ~2.5 GB.
//
// Both directions are HBM-bound by construction (~100 MFMAs per 64-pixel tile and wave against 27 KB of traffic).
#include <stdlib.h>

#include "common.h"

namespace wcmc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned E3_OOB = 0x80000000u;
constexpr int E3_TP = 64;                 // pixels per tile
constexpr int E3_C = 64;                  // channels of every layer of the chain
constexpr int E3_RS = 80;                 // LDS row stride in bf16: 160 B = 5 x 32 (conflict-free transposing reads)
constexpr int E3_TILE = E3_TP * E3_RS;    // u16 per tile plane

__device__ __forceinline__ u16 e3_bf(float x) { return __builtin_bit_cast(u16, (__bf16)x); }
__device__ __forceinline__ float e3_f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// (by-value helper on purpose: __builtin_bit_cast applied directly to a vector-element lvalue reads element 0 -- ROCm 7.2)
__device__ __forceinline__ float e3_u2f(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned e3_f2u(float v) { return __builtin_bit_cast(unsigned, v); }

struct E3Params {
  const u16* x; int64_t M; int Cp0, Kt0;              // split input [M][2][Cp0]; k extent of layer 0's pack (32 or 64)
  const u16* wp0; const u16* wp1; const u16* wp2;     // forward packs [64][2][64]
  const float* b0; const float* b1; const float* b2;
  float* y;                                           // forward: fp32 [M][64]
  float* ym; unsigned ym_bytes;                       // forward, optional: the spp mean of y, fp32 [M / S][64] (S, HW, gm_scale = 1 / S)
  // backward
  const u16* wt1; const u16* wt2;                     // data-gradient packs of W1, W2: [64][2][64]
  const float* gy; const float* gm; int S; int64_t HW; float gm_scale;
  int gy_ps, gm_ps;                                   // pixel stride (floats) of g_y / g_mean: >= 64 (a channel slice of a wider tensor)
  float* ws;                                          // per-workgroup partials
  unsigned x_bytes, y_bytes, gy_bytes, gm_bytes;
};
constexpr int E3_WS_PER_BLOCK = 3 * E3_C * E3_C + 3 * E3_C;

// this wave's 16 weight rows of one layer: hi / lo fragments of the two 32-k steps
struct E3W { bf16x8 h[2], l[2]; };
// (Kt: k extent of the pack -- 64, or 32 for a first layer of <= 32 input channels, whose second k-step is then all zeros:
// its MFMAs add exact zeros, the results stay those of the one-step layer-by-layer kernel bit for bit)
__device__ __forceinline__ E3W e3_load_w(const u16* wp, int row, int q, int Kt = E3_C) {
  E3W w;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const u16* a = wp + (int64_t)(row * 2) * Kt + c * 32 + q * 8;
    const u32x4 z = {0u, 0u, 0u, 0u};
    w.h[c] = c * 32 < Kt ? *reinterpret_cast<const bf16x8*>(a) : __builtin_bit_cast(bf16x8, z);
    w.l[c] = c * 32 < Kt ? *reinterpret_cast<const bf16x8*>(a + Kt) : __builtin_bit_cast(bf16x8, z);
  }
  return w;
}

// acc[i] (pixel tile pt0 + i) += W x T over the 64 channels of tile T; TERMS = 3: W_lo*T_hi + W_hi*T_lo + W_hi*T_hi, 2: no T_lo term
template <int TERMS, int NPT>
__device__ __forceinline__ void e3_gemm(f32x4 (&acc)[NPT], const E3W& w, const u16* th, const u16* tl, int pt0, int fr, int q) {
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int o = (16 * (pt0 + i) + fr) * E3_RS + c * 32 + q * 8;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(th + o);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l[c], ah, acc[i], 0, 0, 0);      // small terms first
      if (TERMS == 3) {
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(tl + o);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[c], al, acc[i], 0, 0, 0);
      }
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[c], ah, acc[i], 0, 0, 0);
    }
  }
}

// relu(acc + bias) of this lane's (pixel 16 (pt0 + i) + fr, couts 16 ct + 4 q ..) as split planes into th / tl (tl may be null)
template <int NPT>
__device__ __forceinline__ void e3_store_relu_split(const f32x4 (&acc)[NPT], const float (&b)[4], u16* th, u16* tl, int ct, int pt0,
                                                    int fr, int q) {
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    u16 hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = acc[i][e] + b[e];
      const float v = t > 0.f ? t : 0.f;
      hi[e] = e3_bf(v);
      lo[e] = e3_bf(v - e3_f(hi[e]));
    }
    const int o = (16 * (pt0 + i) + fr) * E3_RS + 16 * ct + 4 * q;
    *reinterpret_cast<u32x2*>(th + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    if (tl) *reinterpret_cast<u32x2*>(tl + o) = u32x2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
  }
}

// the x tile: 16 16-byte units per pixel in LDS (8 per plane: Cp0 / 8 of data, zeros behind them); thread -> units tid + NT k
template <int NT>
struct E3XPre { u32x4 v[1024 / NT]; };
// (m0 is wave-uniform: the 64-bit products stay scalar; per thread everything is 32-bit -- these kernels run ~100 MFMAs per
// tile and wave, a 64-bit multiply or divide per thread and vector would cost as much as the GEMMs)
template <int NT>
__device__ __forceinline__ E3XPre<NT> e3_load_x(const __amdgpu_buffer_rsrc_t xr, int64_t m0, int64_t M, int Cp0, int tid) {
  E3XPre<NT> r;
  const int dv = Cp0 >> 3;                                    // data units per plane
  const unsigned base = (unsigned)(m0 * (4 * Cp0));
  const int left = (int)(M - m0 < E3_TP ? M - m0 : E3_TP);    // pixels of this tile that exist
#pragma unroll
  for (int k = 0; k < 1024 / NT; ++k) {
    const int v = tid + NT * k, px = v >> 4, u = v & 15, plane = u >> 3, vec = u & 7;
    const unsigned off = (vec < dv && px < left) ? base + (unsigned)(px * (4 * Cp0) + plane * 2 * Cp0 + vec * 16) : E3_OOB;
    r.v[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
  }
  return r;
}
template <int NT>
__device__ __forceinline__ void e3_store_x(const E3XPre<NT>& r, u16* xh, u16* xl, int tid) {
#pragma unroll
  for (int k = 0; k < 1024 / NT; ++k) {
    const int v = tid + NT * k, px = v >> 4, u = v & 15, plane = u >> 3, vec = u & 7;
    *reinterpret_cast<u32x4*>((plane ? xl : xh) + px * E3_RS + vec * 8) = r.v[k];
  }
}

// ------------------------------------------------------------------ forward
// MEAN: the spp mean of y (support/networks.py:35-36) leaves with it.  A workgroup then walks SUPER-tiles -- the same 64 pixels
// of an image's S samples, one after the other -- and keeps the running sum of its y fragments in registers: s ascending, fp32
// adds, one multiply by 1 / S at the end, i.e. the sums wcmc_spp_reduce forms from the stored y (bit-identical), without
// reading y (268 MB at the benchmark shape) again.  Needs HW % 64 == 0.
template <bool MEAN>
__global__ __launch_bounds__(256, 3) void embed3_fwd_kernel(E3Params p) {
  __shared__ __attribute__((aligned(16))) u16 lds[4 * E3_TILE];
  u16* const XH = lds; u16* const XL = lds + E3_TILE; u16* const AH = lds + 2 * E3_TILE; u16* const AL = lds + 3 * E3_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void*)(MEAN ? p.ym : p.y), 0, (int)(MEAN ? p.ym_bytes : 0u), 0x00020000);
  const E3W w0 = e3_load_w(p.wp0, 16 * wave + fr, q, p.Kt0), w1 = e3_load_w(p.wp1, 16 * wave + fr, q), w2 = e3_load_w(p.wp2, 16 * wave + fr, q);
  float b0[4], b1[4], b2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int co = 16 * wave + 4 * q + e; b0[e] = p.b0[co]; b1[e] = p.b1[co]; b2[e] = p.b2[co]; }
  // a unit of work: one 64-pixel tile, or (MEAN) the S tiles of one super-tile
  const int per = MEAN ? p.S : 1;
  const int64_t tpi = MEAN ? p.HW / E3_TP : 1;                                   // tiles per image
  const int64_t nunits = MEAN ? (p.M / ((int64_t)p.S * p.HW)) * tpi : (p.M + E3_TP - 1) / E3_TP;
  auto first_pixel = [&](int64_t u, int sidx) -> int64_t {
    if (!MEAN) return u * E3_TP;
    const int64_t b = u / tpi;
    return (b * p.S + sidx) * p.HW + (u - b * tpi) * E3_TP;
  };
  int64_t u = blockIdx.x;
  E3XPre<256> pre;
  if (u < nunits) pre = e3_load_x<256>(xr, first_pixel(u, 0), p.M, p.Cp0, tid);
  for (; u < nunits; u += gridDim.x) {
    f32x4 msum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) msum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sidx = 0; sidx < per; ++sidx) {
      e3_store_x<256>(pre, XH, XL, tid);
      {                                                                          // next tile's loads fly under this tile's GEMMs
        const bool more = sidx + 1 < per;
        const int64_t un = more ? u : u + gridDim.x;
        if (un < nunits) pre = e3_load_x<256>(xr, first_pixel(un, more ? sidx + 1 : 0), p.M, p.Cp0, tid);
      }
      __syncthreads();
      f32x4 acc[4];
      e3_gemm<3, 4>(acc, w0, XH, XL, 0, fr, q);
      e3_store_relu_split<4>(acc, b0, AH, AL, wave, 0, fr, q);
      __syncthreads();
      e3_gemm<3, 4>(acc, w1, AH, AL, 0, fr, q);
      e3_store_relu_split<4>(acc, b1, XH, XL, wave, 0, fr, q);                       // h1 takes the x tile's place
      __syncthreads();
      e3_gemm<3, 4>(acc, w2, XH, XL, 0, fr, q);
      // y tile through LDS (fp32 [64][68] over the h0 tiles) so that it leaves as whole 256-byte rows
      float* stg = reinterpret_cast<float*>(AH);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 yv = make_float4(acc[i][0] + b2[0], acc[i][1] + b2[1], acc[i][2] + b2[2], acc[i][3] + b2[3]);
        *reinterpret_cast<float4*>(stg + (16 * i + fr) * 68 + 16 * wave + 4 * q) = yv;
        if (MEAN) { msum[i][0] += yv.x; msum[i][1] += yv.y; msum[i][2] += yv.z; msum[i][3] += yv.w; }
      }
      __syncthreads();
      const int64_t m0 = first_pixel(u, sidx);
      const unsigned ybase = (unsigned)(m0 * 256);
      const int left = (int)(p.M - m0 < E3_TP ? p.M - m0 : E3_TP);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = tid + 256 * k, px = v >> 4, c4 = (v & 15) * 4;
        const float4 o = *reinterpret_cast<const float4*>(stg + px * 68 + c4);
        const u32x4 ov = {__builtin_bit_cast(unsigned, o.x), __builtin_bit_cast(unsigned, o.y), __builtin_bit_cast(unsigned, o.z),
                          __builtin_bit_cast(unsigned, o.w)};
        __builtin_amdgcn_raw_buffer_store_b128(ov, yr, px < left ? ybase + (unsigned)(px * 256 + c4 * 4) : E3_OOB, 0, 0);
      }
      // (the next iteration's x stores touch XH / XL only, which every wave finished reading before the barrier above;
      // the staging tile is read here and first written again behind the next tile's first barrier)
    }
    if (MEAN) {                                                                  // the super-tile's mean, the same way
      float* stg = reinterpret_cast<float*>(AH);
      __syncthreads();                                                           // (the last y tile has left the staging tile)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(stg + (16 * i + fr) * 68 + 16 * wave + 4 * q) =
            make_float4(p.gm_scale * msum[i][0], p.gm_scale * msum[i][1], p.gm_scale * msum[i][2], p.gm_scale * msum[i][3]);
      __syncthreads();
      const int64_t b = u / tpi;
      const unsigned mbase = (unsigned)((b * p.HW + (u - b * tpi) * E3_TP) * 256);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = tid + 256 * k, px = v >> 4, c4 = (v & 15) * 4;
        const float4 o = *reinterpret_cast<const float4*>(stg + px * 68 + c4);
        const u32x4 ov = {__builtin_bit_cast(unsigned, o.x), __builtin_bit_cast(unsigned, o.y), __builtin_bit_cast(unsigned, o.z),
                          __builtin_bit_cast(unsigned, o.w)};
        __builtin_amdgcn_raw_buffer_store_b128(ov, mr, mbase + (unsigned)(px * 256 + c4 * 4), 0, 0);
      }
    }
  }
}

// ------------------------------------------------------------------ backward
// transposing read of a [pixel][channel] tile: lane (g = lane >> 4, tq, tp) addresses pixel rows 4 g + tq (+ 16) of k-step kk
// and channels 16 ct + 4 tp ..; it receives channel 16 ct + (lane & 15) of pixels {4g..4g+3, 16+4g..16+4g+3} -- the same k order
// for both MFMA operands
__device__ __forceinline__ bf16x8 e3_tr(const u16* tile, int kk, int ct, int lane) {
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const u16* a0 = tile + (kk * 32 + 4 * g + tq) * E3_RS + ct * 16 + 4 * tp;
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 16 * E3_RS));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 cat = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(bf16x8, cat);
}

// One workgroup of eight waves per CU: wave (ct = wave & 3, ph = wave >> 2) owns cout tile ct of every GEMM for the pixel tiles
// 2 ph, 2 ph + 1, and the weight-gradient tiles (cout tile ct) x (cin tiles 2 ph, 2 ph + 1) of the three layers -- half the
// accumulators and half the prefetch registers of a four-wave layout, which spilled (40 VGPRs to scratch inside the loop: the
// kernel ran at 1.2 TB/s).
__global__ __launch_bounds__(512, 1) void embed3_bwd_kernel(E3Params p) {
  __shared__ __attribute__((aligned(16))) u16 lds[7 * E3_TILE];
  __shared__ float red[3][32][64];
  u16* const XH = lds; u16* const XL = lds + E3_TILE; u16* const H0H = lds + 2 * E3_TILE; u16* const H0L = lds + 3 * E3_TILE;
  u16* const H1H = lds + 4 * E3_TILE; u16* const DYH = lds + 5 * E3_TILE;
  u16* const DH1H = lds + 6 * E3_TILE;     // (its own plane: it is written while other waves still read h0's lo plane)
  u16* const DH0H = XL;                    // x's lo plane is dead once h0 is recomputed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, q = lane >> 4;
  const int ct = wave & 3, ph = wave >> 2, pt0 = 2 * ph;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gyr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gy ? (const void*)p.gy : (const void*)p.x), 0,
                                                                       p.gy ? (int)p.gy_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t gmr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gm ? (const void*)p.gm : (const void*)p.x), 0,
                                                                       p.gm ? (int)p.gm_bytes : 0, 0x00020000);
  const E3W w0 = e3_load_w(p.wp0, 16 * ct + fr, q, p.Kt0), w1 = e3_load_w(p.wp1, 16 * ct + fr, q);
  const E3W t2 = e3_load_w(p.wt2, 16 * ct + fr, q), t1 = e3_load_w(p.wt1, 16 * ct + fr, q);
  float b0[4], b1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int co = 16 * ct + 4 * q + e; b0[e] = p.b0[co]; b1[e] = p.b1[co]; }
  // weight-gradient accumulators of this wave: cout tile ct x the cin tiles 2 ph, 2 ph + 1 of dW2, dW1, dW0
  f32x4 g2[2], g1[2], g0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { g2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; g1[j] = g2[j]; g0[j] = g2[j]; }
  const int nci0 = (p.Cp0 + 15) >> 4;       // cin tiles of layer 0
  float sb2[4] = {0.f, 0.f, 0.f, 0.f}, sb1[4] = {0.f, 0.f, 0.f, 0.f}, sb0[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (p.M + E3_TP - 1) / E3_TP;
  const int64_t SHW = (int64_t)p.S * p.HW;

  struct DyPre { u32x4 g[2], m[2]; };
  // (HW % 64 == 0: a tile lies inside one image, its (image, pixel) of the mean is wave-uniform; otherwise per thread, in 32
  // bits -- M < 2^23 pixels, checked by the host)
  const bool mean_uniform = p.HW % E3_TP == 0;
  auto load_dy = [&](int64_t m0) {
    DyPre r;
    const unsigned gbase = (unsigned)(m0 * p.gy_ps * 4);
    const int left = (int)(p.M - m0 < E3_TP ? p.M - m0 : E3_TP);
    const int64_t b = m0 / SHW, hw0 = m0 % p.HW;
    const unsigned mbase = (unsigned)((b * p.HW + hw0) * p.gm_ps * 4);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = tid + 512 * k, px = v >> 4, c4 = (v & 15) * 4;
      const bool ok = px < left;
      r.g[k] = __builtin_amdgcn_raw_buffer_load_b128(gyr, ok ? gbase + (unsigned)((px * p.gy_ps + c4) * 4) : E3_OOB, 0, 0);
      unsigned moff;
      if (mean_uniform) {
        moff = mbase + (unsigned)((px * p.gm_ps + c4) * 4);
      } else {
        const unsigned m = (unsigned)m0 + (unsigned)px, bi = m / (unsigned)SHW, hw = m % (unsigned)p.HW;
        moff = (bi * (unsigned)p.HW + hw) * (unsigned)(p.gm_ps * 4) + (unsigned)(c4 * 4);
      }
      r.m[k] = __builtin_amdgcn_raw_buffer_load_b128(gmr, (ok && p.gm) ? moff : E3_OOB, 0, 0);
    }
    return r;
  };
  int64_t t = blockIdx.x;
  E3XPre<512> pre;
  DyPre dpre;
  if (t < ntiles) { pre = e3_load_x<512>(xr, t * E3_TP, p.M, p.Cp0, tid); dpre = load_dy(t * E3_TP); }
  for (; t < ntiles; t += gridDim.x) {
    e3_store_x<512>(pre, XH, XL, tid);
    // dy = g_y + g_mean / S: hi plane into its tile, exact column sums (the last layer's bias gradient) on the side
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = tid + 512 * k, px = v >> 4, c4 = (v & 15) * 4;
      u16 hi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = e3_u2f(dpre.g[k][e]) + e3_u2f(dpre.m[k][e]) * p.gm_scale;
        sb2[e] += d;
        hi[e] = e3_bf(d);
      }
      *reinterpret_cast<u32x2*>(DYH + px * E3_RS + c4) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    }
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) { pre = e3_load_x<512>(xr, tn * E3_TP, p.M, p.Cp0, tid); dpre = load_dy(tn * E3_TP); }
    __syncthreads();
    // Five barriers per tile: the two GEMMs that depend on nothing but the staged tile run in ONE interval (h0 = W0 x and
    // a = W2^T dy), and the ReLU gates are applied from registers -- the wave that multiplies cout tile ct of a layer for its
    // pixel tiles is the wave that holds the same channels and pixels of the gradient (no gate read from LDS, no phase of
    // its own for dh1).
    f32x4 acc[2], acca[2];
    unsigned m0bits = 0;                     // [h0 > 0] of this lane's 2 x 4 hidden units (bit 4 i + e)
    // ---- h0 = relu(W0 x + b0) (the forward's arithmetic), a = W2^T dy
    e3_gemm<3, 2>(acc, w0, XH, XL, pt0, fr, q);
    e3_gemm<2, 2>(acca, t2, DYH, nullptr, pt0, fr, q);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u16 hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t_ = acc[i][e] + b0[e];
        const float v = t_ > 0.f ? t_ : 0.f;
        hi[e] = e3_bf(v);
        lo[e] = e3_bf(v - e3_f(hi[e]));
        m0bits |= (e3_f(hi[e]) > 0.f ? 1u : 0u) << (4 * i + e);
      }
      const int o = (16 * (pt0 + i) + fr) * E3_RS + 16 * ct + 4 * q;
      *reinterpret_cast<u32x2*>(H0H + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
      *reinterpret_cast<u32x2*>(H0L + o) = u32x2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
    }
    __syncthreads();
    // ---- h1 = relu(W1 h0 + b1) (hi plane: the weight gradient's operand), dh1 = a . [h1 > 0]
    e3_gemm<3, 2>(acc, w1, H0H, H0L, pt0, fr, q);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u16 hi[4], dh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t_ = acc[i][e] + b1[e];
        hi[e] = e3_bf(t_ > 0.f ? t_ : 0.f);
        const float d = e3_f(hi[e]) > 0.f ? acca[i][e] : 0.f;
        sb1[e] += d;
        dh[e] = e3_bf(d);
      }
      const int o = (16 * (pt0 + i) + fr) * E3_RS + 16 * ct + 4 * q;
      *reinterpret_cast<u32x2*>(H1H + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
      *reinterpret_cast<u32x2*>(DH1H + o) = u32x2{(unsigned)dh[0] | ((unsigned)dh[1] << 16), (unsigned)dh[2] | ((unsigned)dh[3] << 16)};
    }
    __syncthreads();
    // ---- dh0 = (W1^T dh1) . [h0 > 0]
    e3_gemm<2, 2>(acc, t1, DH1H, nullptr, pt0, fr, q);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u16 dh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = ((m0bits >> (4 * i + e)) & 1u) ? acc[i][e] : 0.f;
        sb0[e] += d;
        dh[e] = e3_bf(d);
      }
      const int o = (16 * (pt0 + i) + fr) * E3_RS + 16 * ct + 4 * q;
      *reinterpret_cast<u32x2*>(DH0H + o) = u32x2{(unsigned)dh[0] | ((unsigned)dh[1] << 16), (unsigned)dh[2] | ((unsigned)dh[3] << 16)};
    }
    __syncthreads();
    // ---- weight gradients: D[co][ci] += sum over the tile's pixels; one MFMA per product (hi x hi)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 a2 = e3_tr(DYH, kk, ct, lane), a1 = e3_tr(DH1H, kk, ct, lane), a0 = e3_tr(DH0H, kk, ct, lane);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = pt0 + jj;
        const bf16x8 bh1 = e3_tr(H1H, kk, j, lane), bh0 = e3_tr(H0H, kk, j, lane);
        g2[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bh1, g2[jj], 0, 0, 0);
        g1[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh0, g1[jj], 0, 0, 0);
        if (j < nci0) {
          const bf16x8 bx = e3_tr(XH, kk, j, lane);
          g0[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bx, g0[jj], 0, 0, 0);
        }
      }
    }
    __syncthreads();                         // every tile is free for the next iteration's stores
  }
  // ---- per-workgroup partials: [layer 2 | 1 | 0][co][ci], then the bias sums [2 | 1 | 0][64]
  float* ws = p.ws + (int64_t)blockIdx.x * E3_WS_PER_BLOCK;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = 16 * ct + 4 * q + e, j = pt0 + jj, ci = 16 * j + fr;
      ws[(0 * E3_C + co) * E3_C + ci] = g2[jj][e];
      ws[(1 * E3_C + co) * E3_C + ci] = g1[jj][e];
      ws[(2 * E3_C + co) * E3_C + ci] = j < nci0 ? g0[jj][e] : 0.f;
    }
  // bias sums: sb2 per (thread: channels 4 (tid & 15) .., pixels of its rows) -> 32 threads share a channel quad;
  // sb1 / sb0 per lane (channels 16 ct + 4 q .., pixel columns fr of its two pixel tiles) -> 32 lanes share them
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][tid >> 4][(tid & 15) * 4 + e] = sb2[e];
    red[1][16 * ph + fr][16 * ct + 4 * q + e] = sb1[e];
    red[2][16 * ph + fr][16 * ct + 4 * q + e] = sb0[e];
  }
  __syncthreads();
  if (tid < 3 * E3_C) {
    const int l = tid / E3_C, c = tid - l * E3_C;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[l][r][c];
    ws[3 * E3_C * E3_C + l * E3_C + c] = s;
  }
}

// dW / db = sum over the workgroups' partials, in workgroup order (fixed -> bitwise reproducible); OIHW with ks = 1
__global__ __launch_bounds__(256) void embed3_bwd_finish_kernel(const float* __restrict__ ws, int nblk, int Cin0, float* __restrict__ dw0,
                                                                float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1,
                                                                float* __restrict__ dw2, float* __restrict__ db2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= E3_WS_PER_BLOCK) return;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += ws[(int64_t)(b + u) * E3_WS_PER_BLOCK + i];
  }
  for (; b < nblk; ++b) a[0] += ws[(int64_t)b * E3_WS_PER_BLOCK + i];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  if (i < 3 * E3_C * E3_C) {
    const int l = i / (E3_C * E3_C), r = i - l * E3_C * E3_C, co = r / E3_C, ci = r - co * E3_C;
    if (l == 0) dw2[co * E3_C + ci] = s;
    else if (l == 1) dw1[co * E3_C + ci] = s;
    else if (ci < Cin0) dw0[co * Cin0 + ci] = s;
  } else {
    const int r = i - 3 * E3_C * E3_C, l = r / E3_C, c = r - l * E3_C;
    (l == 0 ? db2 : l == 1 ? db1 : db0)[c] = s;
  }
}

// ================================================================== PathNet.final, fused
//   support/networks.py:39-42:  out = ConvChain(128 -> 128 -> outc <= 8, ksize 1, ReLU, ReLU)(cat([y, repeat_S(prop)], 1))
// (outc = 3 is the default P-buffer; the reference's m10r01 / m11r01 runs use --pnet_out_size 6, train_kpcn.py:209-212: round 4 widened the
// output to eight channels -- the lanes q = 0, 1 of the output tile hold couts 0..3, 4..7; out / gout are [M][round_up(outc, 4)] fp32)
// The layer-by-layer path writes the 128-channel concatenation (537 MB, split) and the 128-channel hidden activation
// (537 MB + mask) per backbone and direction, re-reads both in the backward, and materialises the 128-channel input
// gradient (537 MB fp32) only to slice it into d_y and sum the other half over the samples.  Here a workgroup (eight
// waves, one cout tile of the 128 each) owns "super-tiles": 64 pixels of one image x its S samples.  prop's 64 pixels
// are split into the tile once per super-tile; per sample the y tile arrives, the concatenation and the hidden
// activation live in LDS only; the backward recomputes them, gates with them, writes d_y, keeps d_prop in registers over
// the S samples (written once) and accumulates all weight / bias gradients in registers across the workgroup's tiles.
constexpr int F2_C = 128;                  // concatenation and hidden width
constexpr int F2_RS = 144;                 // LDS row stride in bf16: 288 B = 9 x 32 (conflict-free transposing reads)
constexpr int F2_TILE = E3_TP * F2_RS;
constexpr int F2_ORS = 48;                 // row stride of the d_out tile (32 channels + pad: 96 B = 3 x 32)
constexpr int F2_W1RS = 2 * 128 + 8;       // row stride of the output layer's pack in LDS (hi | lo | pad: 16 rows spread over the banks)

struct F2Params {
  const float* y; int y_ps;                // fp32, pixel stride y_ps floats, M = B*S*HW pixels
  const float* prop; int p_ps;             // fp32, B*HW pixels
  int B, S; int64_t HW;
  const u16* wp0; const u16* wp1;          // forward packs: [128][2][128], [16][2][128]
  const float* b0; const float* b1; int outc;
  float* out; int os;                      // fp32 [M][os], os = round_up(outc, 4) = 4 or 8
  // backward
  const u16* wt0; const u16* wt1;          // data-gradient packs: [128][2][128] (rows = concat channels), [128][2][32] (rows = hidden channels)
  const float* gout;                       // fp32 [M][os]
  float* dy; float* dprop;                 // fp32 [M][64], [B*HW][64]
  float* ws;
  unsigned y_bytes, p_bytes, o_bytes, dy_bytes, dp_bytes;
};
constexpr int F2_WS_PER_BLOCK = F2_C * F2_C + 16 * F2_C + F2_C + 16;      // dW0, dW1 (16 rows), db0, db1 (16)

struct F2W { bf16x8 h[4], l[4]; };
__device__ __forceinline__ F2W f2_load_w(const u16* wp, int row, int q) {
  F2W w;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const u16* a = wp + (int64_t)(row * 2) * F2_C + c * 32 + q * 8;
    w.h[c] = *reinterpret_cast<const bf16x8*>(a);
    w.l[c] = *reinterpret_cast<const bf16x8*>(a + F2_C);
  }
  return w;
}
template <int TERMS, int NPT>
__device__ __forceinline__ void f2_gemm(f32x4 (&acc)[NPT], const F2W& w, const u16* th, const u16* tl, int pt0, int fr, int q) {
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = (16 * (pt0 + i) + fr) * F2_RS + c * 32 + q * 8;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(th + o);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l[c], ah, acc[i], 0, 0, 0);
      if (TERMS == 3) {
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(tl + o);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[c], al, acc[i], 0, 0, 0);
      }
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[c], ah, acc[i], 0, 0, 0);
    }
  }
}
__device__ __forceinline__ bf16x8 f2_tr(const u16* tile, int rs, int kk, int ct, int lane) {
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const u16* a0 = tile + (kk * 32 + 4 * g + tq) * rs + ct * 16 + 4 * tp;
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 16 * rs));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 cat = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(bf16x8, cat);
}
// 64 pixels x 64 fp32 channels (two float4 per thread of 512) -> hi / lo planes of channels [c0, c0 + 64) of a concat tile
struct F2Pre { u32x4 v[2]; };
__device__ __forceinline__ F2Pre f2_load64(const __amdgpu_buffer_rsrc_t r, int64_t m0, int ps, int tid) {
  F2Pre o;
  const unsigned base = (unsigned)(m0 * ps * 4);               // (m0 is wave-uniform: scalar arithmetic)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = tid + 512 * k, px = v >> 4, c4 = (v & 15) * 4;
    o.v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, base + (unsigned)((px * ps + c4) * 4), 0, 0);
  }
  return o;
}
__device__ __forceinline__ void f2_store64_split(const F2Pre& o, u16* th, u16* tl, int c0, int tid) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = tid + 512 * k, px = v >> 4, c4 = (v & 15) * 4;
    u16 hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = e3_u2f(o.v[k][e]);
      hi[e] = e3_bf(f);
      lo[e] = e3_bf(f - e3_f(hi[e]));
    }
    const int a = px * F2_RS + c0 + c4;
    *reinterpret_cast<u32x2*>(th + a) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    *reinterpret_cast<u32x2*>(tl + a) = u32x2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
  }
}
// h = relu(acc + bias): split planes of this lane's (pixel 16 i + fr, couts 16 wave + 4 q ..) into th / tl
__device__ __forceinline__ void f2_store_relu_split(const f32x4 (&acc)[4], const float (&b)[4], u16* th, u16* tl, int wave, int fr, int q) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u16 hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = acc[i][e] + b[e];
      const float v = t > 0.f ? t : 0.f;
      hi[e] = e3_bf(v);
      lo[e] = e3_bf(v - e3_f(hi[e]));
    }
    const int o = (16 * i + fr) * F2_RS + 16 * wave + 4 * q;
    *reinterpret_cast<u32x2*>(th + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    *reinterpret_cast<u32x2*>(tl + o) = u32x2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
  }
}

template <bool BWD>
__global__ __launch_bounds__(512, 1) void final2_kernel(F2Params p) {
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  u16* const CH = lds; u16* const CL = lds + F2_TILE; u16* const HH = lds + 2 * F2_TILE; u16* const HL = lds + 3 * F2_TILE;
  u16* const DHH = lds + 4 * F2_TILE;                            // (backward) dh, hi plane
  u16* const DOH = lds + 5 * F2_TILE;                            // (backward) gated d_out, hi plane, [64][F2_ORS]
  float* const stg = reinterpret_cast<float*>(HL);               // (backward) d_y tile [64][68] over h's lo plane
  float* const red = reinterpret_cast<float*>(lds + 5 * F2_TILE + E3_TP * F2_ORS);    // (backward) bias sums [16][128] + [64][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)p.prop, 0, (int)p.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)(BWD ? (void*)p.gout : (void*)p.out), 0, (int)p.o_bytes, 0x00020000);
  u16* const W1S = reinterpret_cast<u16*>(red + 16 * F2_C + 64 * 8);                  // (backward) the output layer's pack, [16][F2_W1RS]
  const F2W w0 = f2_load_w(p.wp0, 16 * wave + fr, q);
  // the output layer's one cout tile (rows >= outc are zero): registers in the forward; the backward, which has none to spare
  // (it spilled with them), keeps the 8 KB pack in LDS and reads the fragments where waves 0..3 multiply
  F2W w1;
  if (!BWD) w1 = f2_load_w(p.wp1, fr, q);
  float b0[4], b1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { b0[e] = p.b0[16 * wave + 4 * q + e]; b1[e] = (4 * q + e < p.outc) ? p.b1[4 * q + e] : 0.f; }
  const int oq = p.os >> 2;                                      // lanes q < oq hold (and move) output channels 4 q .. 4 q + 3
  const int64_t tpi = p.HW / E3_TP;                              // 64-pixel tiles per image (HW % 64 == 0: checked by the host)
  const int64_t nsuper = (int64_t)p.B * tpi;

  // backward state
  F2W t0;                                                        // W0^T rows of this wave's concat-channel tile
  bf16x8 t1h, t1l;                                               // W1^T rows of this wave's hidden-channel tile (k = 32: one step)
  f32x4 gw0[8], gw1;                                             // dW0[co tile = wave][8 ci tiles], dW1[16][ci tile = wave]
  float sb0[4] = {0.f, 0.f, 0.f, 0.f}, sb1[4] = {0.f, 0.f, 0.f, 0.f};
  __amdgpu_buffer_rsrc_t dyr = yr, dpr = yr;
  if (BWD) {
    t0 = f2_load_w(p.wt0, 16 * wave + fr, q);
    const u16* a = p.wt1 + (int64_t)((16 * wave + fr) * 2) * 32 + q * 8;
    t1h = *reinterpret_cast<const bf16x8*>(a); t1l = *reinterpret_cast<const bf16x8*>(a + 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) gw0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    gw1 = f32x4{0.f, 0.f, 0.f, 0.f};
    dyr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    dpr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dprop, 0, (int)p.dp_bytes, 0x00020000);
    for (int i = tid; i < E3_TP * F2_ORS / 2; i += 512) reinterpret_cast<unsigned*>(DOH)[i] = 0u;     // channels >= os stay zero
    *reinterpret_cast<u32x4*>(W1S + (tid >> 5) * F2_W1RS + (tid & 31) * 8) = *reinterpret_cast<const u32x4*>(p.wp1 + tid * 8);
  }

  for (int64_t st = blockIdx.x; st < nsuper; st += gridDim.x) {
    const int64_t b = st / tpi, hw0 = (st - b * tpi) * E3_TP;
    // the spp-broadcast half of the concatenation: once per super-tile
    F2Pre pp = f2_load64(pr, b * p.HW + hw0, p.p_ps, tid);
    F2Pre yp = f2_load64(yr, (b * p.S) * p.HW + hw0, p.y_ps, tid);
    __syncthreads();                                             // (the previous super-tile's last readers of the tiles are done)
    f2_store64_split(pp, CH, CL, 64, tid);
    f32x4 dpacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dpacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.S; ++s) {
      const int64_t m0 = (b * p.S + s) * p.HW + hw0;
      f2_store64_split(yp, CH, CL, 0, tid);
      u32x4 go = {0u, 0u, 0u, 0u};
      // (backward) d_out of pixel 16 wave + fr, for the lanes that will hold out of that pixel: waves 0..3, lanes q < oq
      if (BWD && wave < 4 && q < oq)
        go = __builtin_amdgcn_raw_buffer_load_b128(orr, (unsigned)(m0 * p.os * 4) + (unsigned)((16 * wave + fr) * p.os * 4 + q * 16), 0, 0);
      if (s + 1 < p.S) yp = f2_load64(yr, (b * p.S + s + 1) * p.HW + hw0, p.y_ps, tid);      // next sample's tile flies under the GEMMs
      __syncthreads();
      // ---- h = relu(W0 c + b0)
      f32x4 acc[4];
      f2_gemm<3, 4>(acc, w0, CH, CL, 0, fr, q);
      f2_store_relu_split(acc, b0, HH, HL, wave, fr, q);
      __syncthreads();
      // ---- out = relu(W1 h + b1): wave w < 4 multiplies pixel tile w; lanes q hold couts 4 q .. 4 q + 3 of pixel 16 w + fr
      float ov[4] = {0.f, 0.f, 0.f, 0.f};
      if (wave < 4) {
        f32x4 a2[1];
        if (BWD) {
          a2[0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int o = (16 * wave + fr) * F2_RS + c * 32 + q * 8;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(HH + o), al = *reinterpret_cast<const bf16x8*>(HL + o);
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(W1S + fr * F2_W1RS + c * 32 + q * 8);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(W1S + fr * F2_W1RS + F2_C + c * 32 + q * 8);
            a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, a2[0], 0, 0, 0);
            a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, a2[0], 0, 0, 0);
            a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, a2[0], 0, 0, 0);
          }
        } else {
          f2_gemm<3, 1>(a2, w1, HH, HL, wave, fr, q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = a2[0][e] + b1[e]; ov[e] = (4 * q + e < p.outc && t > 0.f) ? t : 0.f; }
      }
      if (!BWD) {
        if (wave < 4 && q < oq) {
          const u32x4 o4 = {__builtin_bit_cast(unsigned, ov[0]), __builtin_bit_cast(unsigned, ov[1]), __builtin_bit_cast(unsigned, ov[2]),
                            __builtin_bit_cast(unsigned, ov[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o4, orr, (unsigned)(m0 * p.os * 4) + (unsigned)((16 * wave + fr) * p.os * 4 + q * 16), 0, 0);
        }
        __syncthreads();                                         // h is consumed: the next sample may overwrite the tiles
        continue;
      }
      // ---- backward.  d_o = d_out . [out > 0] (hi plane into its tile; exact sums = the output layer's bias gradient), by the
      // lanes that hold out (no second phase, no gate bits through LDS)
      if (wave < 4 && q < oq) {
        const int px = 16 * wave + fr;
        u16 hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = ov[e] > 0.f ? e3_u2f(go[e]) : 0.f;
          sb1[e] += d;
          hi[e] = e3_bf(d);
        }
        *reinterpret_cast<u32x2*>(DOH + px * F2_ORS + 4 * q) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
      }
      __syncthreads();
      // ---- dh = (W1^T d_o) . [h > 0]: this wave's hidden-channel tile, k = the 32-wide step whose channels >= 4 are zero
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 dof = *reinterpret_cast<const bf16x8*>(DOH + (16 * i + fr) * F2_ORS + q * 8);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1l, dof, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1h, dof, a, 0, 0, 0);
        const int o = (16 * i + fr) * F2_RS + 16 * wave + 4 * q;
        const u32x2 hm = *reinterpret_cast<const u32x2*>(HH + o);
        u16 hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const u16 hv = (u16)((e < 2 ? hm[0] : hm[1]) >> (16 * (e & 1)));
          const float d = e3_f(hv) > 0.f ? a[e] : 0.f;
          sb0[e] += d;
          hi[e] = e3_bf(d);
        }
        *reinterpret_cast<u32x2*>(DHH + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
      }
      __syncthreads();
      // ---- dc = W0^T dh: waves 0..3 hold d_y (concat channels 0..63), waves 4..7 the d_prop contribution of this sample
      f2_gemm<2, 4>(acc, t0, DHH, nullptr, 0, fr, q);
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(stg + (16 * i + fr) * 68 + 16 * wave + 4 * q) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dpacc[i] += acc[i];
      }
      // ---- weight gradients (hi x hi, pixels on k): dW0[co tile = wave][ci tile j] += dh^T c;  dW1[16][ci tile = wave] += d_o^T h
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 a0 = f2_tr(DHH, F2_RS, kk, wave, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) gw0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, f2_tr(CH, F2_RS, kk, j, lane), gw0[j], 0, 0, 0);
        gw1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2_tr(DOH, F2_ORS, kk, 0, lane), f2_tr(HH, F2_RS, kk, wave, lane), gw1, 0, 0, 0);
      }
      __syncthreads();
      // d_y tile: whole 256-byte rows
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = tid + 512 * k, px = v >> 4, c4 = (v & 15) * 4;
        const float4 o = *reinterpret_cast<const float4*>(stg + px * 68 + c4);
        const u32x4 o4 = {__builtin_bit_cast(unsigned, o.x), __builtin_bit_cast(unsigned, o.y), __builtin_bit_cast(unsigned, o.z),
                          __builtin_bit_cast(unsigned, o.w)};
        __builtin_amdgcn_raw_buffer_store_b128(o4, dyr, (unsigned)(m0 * 256) + (unsigned)(px * 256 + c4 * 4), 0, 0);
      }
      // (no barrier here: the next sample's y tile goes into c's planes, which nobody reads any more, and the barrier behind
      // it orders these reads of the staging tile before h's lo plane is written again)
    }
    if (BWD && wave >= 4) {                                      // d_prop of the super-tile: the sum over its S samples
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 o4 = {e3_f2u(dpacc[i][0]), e3_f2u(dpacc[i][1]), e3_f2u(dpacc[i][2]), e3_f2u(dpacc[i][3])};
        __builtin_amdgcn_raw_buffer_store_b128(o4, dpr, (unsigned)((b * p.HW + hw0) * 256) + (unsigned)((16 * i + fr) * 256 + (16 * (wave - 4) + 4 * q) * 4), 0, 0);
      }
    }
  }
  if (!BWD) return;
  // ---- per-workgroup partials: dW0 [128][128] | dW1 [16][128] | db0 [128] | db1 [16]
  float* ws = p.ws + (int64_t)blockIdx.x * F2_WS_PER_BLOCK;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) ws[(16 * wave + 4 * q + e) * F2_C + 16 * j + fr] = gw0[j][e];
#pragma unroll
  for (int e = 0; e < 4; ++e) ws[F2_C * F2_C + (4 * q + e) * F2_C + 16 * wave + fr] = gw1[e];
  __syncthreads();
  float* redb = red;                                             // [16 pixel columns][128 channels], then [64 threads][4]
#pragma unroll
  for (int e = 0; e < 4; ++e) redb[fr * F2_C + 16 * wave + 4 * q + e] = sb0[e];
  if (wave < 4 && q < 2) {                                       // (lanes q = 1 hold zeros when os == 4)
#pragma unroll
    for (int e = 0; e < 4; ++e) redb[16 * F2_C + (16 * wave + fr) * 8 + 4 * q + e] = sb1[e];
  }
  __syncthreads();
  if (tid < F2_C) {
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += redb[r * F2_C + tid];
    ws[F2_C * F2_C + 16 * F2_C + tid] = sacc;
  } else if (tid < F2_C + 16) {
    const int c = tid - F2_C;
    float sacc = 0.f;
    if (c < 8)
      for (int r = 0; r < 64; ++r) sacc += redb[16 * F2_C + r * 8 + c];
    ws[F2_C * F2_C + 16 * F2_C + F2_C + c] = sacc;
  }
}

__global__ __launch_bounds__(256) void final2_bwd_finish_kernel(const float* __restrict__ ws, int nblk, int outc, float* __restrict__ dw0,
                                                                float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F2_WS_PER_BLOCK) return;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += ws[(int64_t)(b + u) * F2_WS_PER_BLOCK + i];
  }
  for (; b < nblk; ++b) a[0] += ws[(int64_t)b * F2_WS_PER_BLOCK + i];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  if (i < F2_C * F2_C) dw0[i] = s;
  else if (i < F2_C * F2_C + 16 * F2_C) {
    const int r = i - F2_C * F2_C, co = r / F2_C, ci = r - co * F2_C;
    if (co < outc) dw1[co * F2_C + ci] = s;
  } else if (i < F2_C * F2_C + 16 * F2_C + F2_C) db0[i - F2_C * F2_C - 16 * F2_C] = s;
  else {
    const int c = i - F2_C * F2_C - 16 * F2_C - F2_C;
    if (c < outc) db1[c] = s;
  }
}

static int f2_grid() { return 256; }
constexpr size_t F2_LDS_FWD = (size_t)4 * F2_TILE * sizeof(u16);
constexpr size_t F2_LDS_BWD = (size_t)5 * F2_TILE * sizeof(u16) + (size_t)E3_TP * F2_ORS * sizeof(u16) + (size_t)(16 * F2_C + 64 * 8) * sizeof(float) +
                              (size_t)16 * F2_W1RS * sizeof(u16);

static int e3_grid_bwd() { return 256; }

}  // namespace wcmc

using namespace wcmc;

extern "C" int wcmc_embed3_supported(int Cin, int C1, int C2, int C3) {
  return Cin >= 1 && Cin <= 64 && C1 == 64 && C2 == 64 && C3 == 64;
}

extern "C" size_t wcmc_embed3_bwd_workspace_bytes(void) { return (size_t)e3_grid_bwd() * E3_WS_PER_BLOCK * sizeof(float); }

static int e3_fill(E3Params& p, const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                   const float* b1) {
  WCMC_REQUIRE(x_split && wp0 && wp1 && b0 && b1 && M > 0 && Cin >= 1 && Cin <= 64, WCMC_ERR_BAD_ARG, "embed3: bad argument");
  WCMC_REQUIRE(aligned16(x_split) && aligned16(wp0) && aligned16(wp1), WCMC_ERR_ALIGNMENT, "embed3: buffers must be 16-byte aligned");
  p.x = (const u16*)x_split; p.M = M; p.Cp0 = round_up(Cin, 8); p.Kt0 = round_up(p.Cp0, 32);
  p.wp0 = (const u16*)wp0; p.wp1 = (const u16*)wp1; p.b0 = b0; p.b1 = b1;
  const int64_t xb = M * 4 * p.Cp0, yb = M * 256;
  WCMC_REQUIRE(xb < 0x7ff00000ll && yb < 0x7ff00000ll, WCMC_ERR_BAD_ARG, "embed3: more than 2 GiB per tensor (split the batch)");
  p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
  return 0;
}

extern "C" int wcmc_embed3_fwd(const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                               const float* b1, const void* wp2, const float* b2, float* y, void* stream) {
  E3Params p = {};
  if (int rc = e3_fill(p, x_split, M, Cin, wp0, b0, wp1, b1)) return rc;
  WCMC_REQUIRE(wp2 && b2 && y && aligned16(wp2) && aligned16(y), WCMC_ERR_BAD_ARG, "embed3_fwd: bad argument");
  p.wp2 = (const u16*)wp2; p.b2 = b2; p.y = y;
  const int64_t ntiles = (M + E3_TP - 1) / E3_TP;
  const unsigned grid = (unsigned)(ntiles < 768 ? ntiles : 768);
  hipLaunchKernelGGL(embed3_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("embed3_fwd");
}

extern "C" int wcmc_embed3_mean_supported(int S, int64_t HW) { return S >= 1 && HW > 0 && HW % E3_TP == 0; }

extern "C" int wcmc_embed3_mean_fwd(const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                                    const float* b1, const void* wp2, const float* b2, float* y, float* y_mean, int S, int64_t HW,
                                    void* stream) {
  E3Params p = {};
  if (int rc = e3_fill(p, x_split, M, Cin, wp0, b0, wp1, b1)) return rc;
  WCMC_REQUIRE(wp2 && b2 && y && y_mean && aligned16(wp2) && aligned16(y) && aligned16(y_mean), WCMC_ERR_BAD_ARG, "embed3_mean_fwd: bad argument");
  WCMC_REQUIRE(wcmc_embed3_mean_supported(S, HW) && M % ((int64_t)S * HW) == 0, WCMC_ERR_BAD_ARG,
               "embed3_mean_fwd: M must be B * S * HW with HW a multiple of 64 (ask wcmc_embed3_mean_supported)");
  p.wp2 = (const u16*)wp2; p.b2 = b2; p.y = y; p.ym = y_mean; p.ym_bytes = (unsigned)((M / S) * 256);
  p.S = S; p.HW = HW; p.gm_scale = 1.0f / (float)S;
  const int64_t nsuper = M / ((int64_t)S * E3_TP);
  const unsigned grid = (unsigned)(nsuper < 768 ? nsuper : 768);
  hipLaunchKernelGGL(embed3_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("embed3_mean_fwd");
}

extern "C" int wcmc_embed3_bwd(const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                               const float* b1, const void* wt1, const void* wt2, const float* gy, int gy_pixel_stride,
                               const float* gm, int gm_pixel_stride, int S, int64_t HW, float gm_scale, float* dw0, float* db0, float* dw1, float* db1, float* dw2, float* db2,
                               void* workspace, size_t workspace_bytes, void* stream) {
  E3Params p = {};
  if (int rc = e3_fill(p, x_split, M, Cin, wp0, b0, wp1, b1)) return rc;
  WCMC_REQUIRE(wt1 && wt2 && (gy || gm) && dw0 && db0 && dw1 && db1 && dw2 && db2 && workspace && aligned16(wt1) && aligned16(wt2) &&
                   (!gy || aligned16(gy)) && (!gm || aligned16(gm)) && aligned16(workspace),
               WCMC_ERR_BAD_ARG, "embed3_bwd: bad argument");
  WCMC_REQUIRE(workspace_bytes >= wcmc_embed3_bwd_workspace_bytes(), WCMC_ERR_WORKSPACE, "embed3_bwd: workspace too small");
  WCMC_REQUIRE(!gm || (S >= 1 && HW >= 1 && M % ((int64_t)S * HW) == 0), WCMC_ERR_BAD_ARG,
               "embed3_bwd: with the gradient of the spp mean M must be B * S * HW");
  p.wt1 = (const u16*)wt1; p.wt2 = (const u16*)wt2; p.gy = gy; p.gm = gm; p.S = S > 0 ? S : 1; p.HW = HW > 0 ? HW : M;
  p.gm_scale = gm_scale; p.ws = (float*)workspace;
  WCMC_REQUIRE((!gy || (gy_pixel_stride >= 64 && gy_pixel_stride % 4 == 0)) && (!gm || (gm_pixel_stride >= 64 && gm_pixel_stride % 4 == 0)),
               WCMC_ERR_ALIGNMENT, "embed3_bwd: pixel strides must be multiples of 4 floats and >= 64");
  p.gy_ps = gy ? gy_pixel_stride : 64; p.gm_ps = gm ? gm_pixel_stride : 64;
  const int64_t gyb = gy ? ((M - 1) * p.gy_ps + 64) * 4 : 0, gmb = gm ? ((M / p.S - 1) * p.gm_ps + 64) * 4 : 0;
  WCMC_REQUIRE(gyb < 0x7ff00000ll && gmb < 0x7ff00000ll, WCMC_ERR_BAD_ARG, "embed3_bwd: gradient view spans more than 2 GiB");
  p.gy_bytes = (unsigned)gyb; p.gm_bytes = (unsigned)gmb;
  const int nblk = e3_grid_bwd();
  hipLaunchKernelGGL(embed3_bwd_kernel, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, p);
  if (int rc = check_launch("embed3_bwd")) return rc;
  hipLaunchKernelGGL(embed3_bwd_finish_kernel, dim3((unsigned)((E3_WS_PER_BLOCK + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nblk, Cin, dw0, db0, dw1, db1, dw2, db2);
  return check_launch("embed3_bwd_finish");
}


// ---------------------------------------------------------------- PathNet.final, fused
extern "C" int wcmc_final2_supported(int C1, int C2, int Chid, int outc, int64_t HW) {
  return C1 == 64 && C2 == 64 && Chid == 128 && outc >= 1 && outc <= 8 && HW > 0 && HW % 64 == 0;
}
extern "C" size_t wcmc_final2_bwd_workspace_bytes(void) { return (size_t)f2_grid() * F2_WS_PER_BLOCK * sizeof(float); }

static int f2_fill(F2Params& p, const float* y, int y_ps, const float* prop, int p_ps, int B, int S, int64_t HW, const void* wp0,
                   const float* b0, const void* wp1, const float* b1, int outc) {
  WCMC_REQUIRE(y && prop && wp0 && wp1 && b0 && b1 && B > 0 && S > 0 && wcmc_final2_supported(64, 64, 128, outc, HW), WCMC_ERR_BAD_ARG,
               "final2: bad argument (64 + 64 -> 128 -> <= 8 channels, H*W a multiple of 64)");
  WCMC_REQUIRE(aligned16(y) && aligned16(prop) && aligned16(wp0) && aligned16(wp1) && y_ps >= 64 && y_ps % 4 == 0 && p_ps >= 64 && p_ps % 4 == 0,
               WCMC_ERR_ALIGNMENT, "final2: views must be 16-byte aligned with pixel strides that are multiples of 4 floats");
  p.y = y; p.y_ps = y_ps; p.prop = prop; p.p_ps = p_ps; p.B = B; p.S = S; p.HW = HW;
  p.wp0 = (const u16*)wp0; p.wp1 = (const u16*)wp1; p.b0 = b0; p.b1 = b1; p.outc = outc; p.os = outc <= 4 ? 4 : 8;
  const int64_t M = (int64_t)B * S * HW;
  const int64_t yb = ((M - 1) * y_ps + 64) * 4, pb = (((int64_t)B * HW - 1) * p_ps + 64) * 4, ob = M * p.os * 4;
  WCMC_REQUIRE(yb < 0x7ff00000ll && pb < 0x7ff00000ll && M * 256 < 0x7ff00000ll, WCMC_ERR_BAD_ARG, "final2: more than 2 GiB per tensor");
  p.y_bytes = (unsigned)yb; p.p_bytes = (unsigned)pb; p.o_bytes = (unsigned)ob;
  return 0;
}

extern "C" int wcmc_final2_fwd(const float* y, int y_pixel_stride, const float* prop, int prop_pixel_stride, int B, int S, int64_t HW,
                               const void* wp0, const float* b0, const void* wp1, const float* b1, int outc, float* out, void* stream) {
  F2Params p = {};
  if (int rc = f2_fill(p, y, y_pixel_stride, prop, prop_pixel_stride, B, S, HW, wp0, b0, wp1, b1, outc)) return rc;
  WCMC_REQUIRE(out && aligned16(out), WCMC_ERR_BAD_ARG, "final2_fwd: bad output");
  p.out = out;
  static LdsAttr attr;
  if (set_max_lds(reinterpret_cast<const void*>(&final2_kernel<false>), (size_t)F2_LDS_FWD, attr) != hipSuccess) return WCMC_ERR_LAUNCH;
  const int64_t nsuper = (int64_t)B * (HW / 64);
  hipLaunchKernelGGL(final2_kernel<false>, dim3((unsigned)(nsuper < f2_grid() ? nsuper : f2_grid())), dim3(512), F2_LDS_FWD, (hipStream_t)stream, p);
  return check_launch("final2_fwd");
}

extern "C" int wcmc_final2_bwd(const float* y, int y_pixel_stride, const float* prop, int prop_pixel_stride, int B, int S, int64_t HW,
                               const void* wp0, const float* b0, const void* wp1, const float* b1, int outc, const void* wt0,
                               const void* wt1, const float* gout, float* dy, float* dprop, float* dw0, float* db0, float* dw1,
                               float* db1, void* workspace, size_t workspace_bytes, void* stream) {
  F2Params p = {};
  if (int rc = f2_fill(p, y, y_pixel_stride, prop, prop_pixel_stride, B, S, HW, wp0, b0, wp1, b1, outc)) return rc;
  WCMC_REQUIRE(wt0 && wt1 && gout && dy && dprop && dw0 && db0 && dw1 && db1 && workspace && aligned16(wt0) && aligned16(wt1) &&
                   aligned16(gout) && aligned16(dy) && aligned16(dprop) && aligned16(workspace),
               WCMC_ERR_BAD_ARG, "final2_bwd: bad argument");
  WCMC_REQUIRE(workspace_bytes >= wcmc_final2_bwd_workspace_bytes(), WCMC_ERR_WORKSPACE, "final2_bwd: workspace too small");
  p.wt0 = (const u16*)wt0; p.wt1 = (const u16*)wt1; p.gout = gout; p.dy = dy; p.dprop = dprop; p.ws = (float*)workspace;
  const int64_t M = (int64_t)B * S * HW;
  p.dy_bytes = (unsigned)(M * 256); p.dp_bytes = (unsigned)((int64_t)B * HW * 256);
  static LdsAttr attr;
  if (set_max_lds(reinterpret_cast<const void*>(&final2_kernel<true>), (size_t)F2_LDS_BWD, attr) != hipSuccess) return WCMC_ERR_LAUNCH;
  const int nblk = f2_grid();
  hipLaunchKernelGGL(final2_kernel<true>, dim3((unsigned)nblk), dim3(512), F2_LDS_BWD, (hipStream_t)stream, p);
  if (int rc = check_launch("final2_bwd")) return rc;
  hipLaunchKernelGGL(final2_bwd_finish_kernel, dim3((unsigned)((F2_WS_PER_BLOCK + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nblk, outc, dw0, db0, dw1, db1);
  return check_launch("final2_bwd_finish");
}
