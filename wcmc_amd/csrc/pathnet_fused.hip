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

struct E3Params {
  const u16* x; int64_t M; int Cp0, Kt0;              // split input [M][2][Cp0]; k extent of layer 0's pack (32 or 64)
  const u16* wp0; const u16* wp1; const u16* wp2;     // forward packs [64][2][64]
  const float* b0; const float* b1; const float* b2;
  float* y;                                           // forward: fp32 [M][64]
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

// acc[i] (pixel tile i) += W x T over the 64 channels of tile T; TERMS = 3: W_lo*T_hi + W_hi*T_lo + W_hi*T_hi, 2: no T_lo term
template <int TERMS>
__device__ __forceinline__ void e3_gemm(f32x4 (&acc)[4], const E3W& w, const u16* th, const u16* tl, int fr, int q) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int o = (16 * i + fr) * E3_RS + c * 32 + q * 8;
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

// relu(acc + bias) of this lane's (pixel 16 i + fr, couts 16 wave + 4 q ..) as split planes into th / tl (tl may be null)
__device__ __forceinline__ void e3_store_relu_split(const f32x4 (&acc)[4], const float (&b)[4], u16* th, u16* tl, int wave,
                                                    int fr, int q) {
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
    const int o = (16 * i + fr) * E3_RS + 16 * wave + 4 * q;
    *reinterpret_cast<u32x2*>(th + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    if (tl) *reinterpret_cast<u32x2*>(tl + o) = u32x2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
  }
}

// the x tile: 16 16-byte units per pixel in LDS (8 per plane: Cp0 / 8 of data, zeros behind them); thread -> units tid + 256 k
struct E3XPre { u32x4 v[4]; };
__device__ __forceinline__ E3XPre e3_load_x(const __amdgpu_buffer_rsrc_t xr, int64_t m0, int64_t M, int Cp0, int tid) {
  E3XPre r;
  const int dv = Cp0 >> 3;                                    // data units per plane
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = tid + 256 * k, px = v >> 4, u = v & 15, plane = u >> 3, vec = u & 7;
    const int64_t m = m0 + px;
    const unsigned off = (vec < dv && m < M) ? (unsigned)(m * (4 * Cp0) + plane * 2 * Cp0 + vec * 16) : E3_OOB;
    r.v[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
  }
  return r;
}
__device__ __forceinline__ void e3_store_x(const E3XPre& r, u16* xh, u16* xl, int tid) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = tid + 256 * k, px = v >> 4, u = v & 15, plane = u >> 3, vec = u & 7;
    *reinterpret_cast<u32x4*>((plane ? xl : xh) + px * E3_RS + vec * 8) = r.v[k];
  }
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 3) void embed3_fwd_kernel(E3Params p) {
  __shared__ __attribute__((aligned(16))) u16 lds[4 * E3_TILE];
  u16* const XH = lds; u16* const XL = lds + E3_TILE; u16* const AH = lds + 2 * E3_TILE; u16* const AL = lds + 3 * E3_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
  const E3W w0 = e3_load_w(p.wp0, 16 * wave + fr, q, p.Kt0), w1 = e3_load_w(p.wp1, 16 * wave + fr, q), w2 = e3_load_w(p.wp2, 16 * wave + fr, q);
  float b0[4], b1[4], b2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int co = 16 * wave + 4 * q + e; b0[e] = p.b0[co]; b1[e] = p.b1[co]; b2[e] = p.b2[co]; }
  const int64_t ntiles = (p.M + E3_TP - 1) / E3_TP;
  int64_t t = blockIdx.x;
  E3XPre pre;
  if (t < ntiles) pre = e3_load_x(xr, t * E3_TP, p.M, p.Cp0, tid);
  for (; t < ntiles; t += gridDim.x) {
    e3_store_x(pre, XH, XL, tid);
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) pre = e3_load_x(xr, tn * E3_TP, p.M, p.Cp0, tid);       // next tile's loads fly under this tile's GEMMs
    __syncthreads();
    f32x4 acc[4];
    e3_gemm<3>(acc, w0, XH, XL, fr, q);
    e3_store_relu_split(acc, b0, AH, AL, wave, fr, q);
    __syncthreads();
    e3_gemm<3>(acc, w1, AH, AL, fr, q);
    e3_store_relu_split(acc, b1, XH, XL, wave, fr, q);                         // h1 takes the x tile's place
    __syncthreads();
    e3_gemm<3>(acc, w2, XH, XL, fr, q);
    // y tile through LDS (fp32 [64][68] over the h0 tiles) so that it leaves as whole 256-byte rows
    float* stg = reinterpret_cast<float*>(AH);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(stg + (16 * i + fr) * 68 + 16 * wave + 4 * q) =
          make_float4(acc[i][0] + b2[0], acc[i][1] + b2[1], acc[i][2] + b2[2], acc[i][3] + b2[3]);
    __syncthreads();
    const int64_t m0 = t * E3_TP;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = tid + 256 * k, px = v >> 4, c4 = (v & 15) * 4;
      const float4 o = *reinterpret_cast<const float4*>(stg + px * 68 + c4);
      const u32x4 ov = {__builtin_bit_cast(unsigned, o.x), __builtin_bit_cast(unsigned, o.y), __builtin_bit_cast(unsigned, o.z),
                        __builtin_bit_cast(unsigned, o.w)};
      __builtin_amdgcn_raw_buffer_store_b128(ov, yr, m0 + px < p.M ? (unsigned)((m0 + px) * 256 + c4 * 4) : E3_OOB, 0, 0);
    }
    // (the next iteration's x stores touch XH / XL only, which every wave finished reading before the barrier above;
    // the staging tile is read here and first written again behind the next tile's first barrier)
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

__global__ __launch_bounds__(256, 2) void embed3_bwd_kernel(E3Params p) {
  __shared__ __attribute__((aligned(16))) u16 lds[6 * E3_TILE];
  __shared__ float red[3][16][64];
  u16* const XH = lds; u16* const XL = lds + E3_TILE; u16* const H0H = lds + 2 * E3_TILE; u16* const H0L = lds + 3 * E3_TILE;
  u16* const H1H = lds + 4 * E3_TILE; u16* const DYH = lds + 5 * E3_TILE;
  u16* const DH1H = H0L;                   // h0's lo plane is dead once h1 is recomputed
  u16* const DH0H = XL;                    // x's lo plane is dead once h0 is recomputed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gyr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gy ? (const void*)p.gy : (const void*)p.x), 0,
                                                                       p.gy ? (int)p.gy_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t gmr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gm ? (const void*)p.gm : (const void*)p.x), 0,
                                                                       p.gm ? (int)p.gm_bytes : 0, 0x00020000);
  const E3W w0 = e3_load_w(p.wp0, 16 * wave + fr, q, p.Kt0), w1 = e3_load_w(p.wp1, 16 * wave + fr, q);
  const E3W t2 = e3_load_w(p.wt2, 16 * wave + fr, q), t1 = e3_load_w(p.wt1, 16 * wave + fr, q);
  float b0[4], b1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int co = 16 * wave + 4 * q + e; b0[e] = p.b0[co]; b1[e] = p.b1[co]; }
  // weight-gradient accumulators of this wave: cout tile `wave` x the cin tiles of dW2 (4), dW1 (4), dW0 (<= 4)
  f32x4 g2[4], g1[4], g0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { g2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; g1[j] = g2[j]; g0[j] = g2[j]; }
  const int nci0 = (p.Cp0 + 15) >> 4;       // cin tiles of layer 0
  float sb2[4] = {0.f, 0.f, 0.f, 0.f}, sb1[4] = {0.f, 0.f, 0.f, 0.f}, sb0[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (p.M + E3_TP - 1) / E3_TP;
  const int64_t SHW = (int64_t)p.S * p.HW;

  struct DyPre { u32x4 g[4], m[4]; };
  auto load_dy = [&](int64_t m0) {
    DyPre r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = tid + 256 * k, px = v >> 4, c4 = (v & 15) * 4;
      const int64_t m = m0 + px;
      const bool ok = m < p.M;
      r.g[k] = __builtin_amdgcn_raw_buffer_load_b128(gyr, ok ? (unsigned)((m * p.gy_ps + c4) * 4) : E3_OOB, 0, 0);
      unsigned moff = E3_OOB;
      if (ok && p.gm) {
        const int64_t b = m / SHW, hw = m % p.HW;
        moff = (unsigned)(((b * p.HW + hw) * p.gm_ps + c4) * 4);
      }
      r.m[k] = __builtin_amdgcn_raw_buffer_load_b128(gmr, moff, 0, 0);
    }
    return r;
  };

  int64_t t = blockIdx.x;
  E3XPre pre;
  DyPre dpre;
  if (t < ntiles) { pre = e3_load_x(xr, t * E3_TP, p.M, p.Cp0, tid); dpre = load_dy(t * E3_TP); }
  for (; t < ntiles; t += gridDim.x) {
    e3_store_x(pre, XH, XL, tid);
    // dy = g_y + g_mean / S: hi plane into its tile, exact column sums (the last layer's bias gradient) on the side
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = tid + 256 * k, px = v >> 4, c4 = (v & 15) * 4;
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
    if (tn < ntiles) { pre = e3_load_x(xr, tn * E3_TP, p.M, p.Cp0, tid); dpre = load_dy(tn * E3_TP); }
    __syncthreads();
    f32x4 acc[4];
    // ---- recompute h0, h1 (the forward's arithmetic)
    e3_gemm<3>(acc, w0, XH, XL, fr, q);
    e3_store_relu_split(acc, b0, H0H, H0L, wave, fr, q);
    __syncthreads();
    e3_gemm<3>(acc, w1, H0H, H0L, fr, q);
    e3_store_relu_split(acc, b1, H1H, nullptr, wave, fr, q);
    __syncthreads();
    // ---- dh1 = (W2^T dy) . [h1 > 0]   (this lane: channels 16 wave + 4 q .. of pixel 16 i + fr)
    e3_gemm<2>(acc, t2, DYH, nullptr, fr, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = (16 * i + fr) * E3_RS + 16 * wave + 4 * q;
      const u32x2 hm = *reinterpret_cast<const u32x2*>(H1H + o);
      u16 hi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u16 h = (u16)((e < 2 ? hm[0] : hm[1]) >> (16 * (e & 1)));
        const float d = e3_f(h) > 0.f ? acc[i][e] : 0.f;
        sb1[e] += d;
        hi[e] = e3_bf(d);
      }
      *reinterpret_cast<u32x2*>(DH1H + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    }
    __syncthreads();
    // ---- dh0 = (W1^T dh1) . [h0 > 0]
    e3_gemm<2>(acc, t1, DH1H, nullptr, fr, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = (16 * i + fr) * E3_RS + 16 * wave + 4 * q;
      const u32x2 hm = *reinterpret_cast<const u32x2*>(H0H + o);
      u16 hi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u16 h = (u16)((e < 2 ? hm[0] : hm[1]) >> (16 * (e & 1)));
        const float d = e3_f(h) > 0.f ? acc[i][e] : 0.f;
        sb0[e] += d;
        hi[e] = e3_bf(d);
      }
      *reinterpret_cast<u32x2*>(DH0H + o) = u32x2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
    }
    __syncthreads();
    // ---- weight gradients: D[co][ci] += sum over the tile's pixels; cout tile = wave, one MFMA per product (hi x hi)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 a2 = e3_tr(DYH, kk, wave, lane), a1 = e3_tr(DH1H, kk, wave, lane), a0 = e3_tr(DH0H, kk, wave, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bh1 = e3_tr(H1H, kk, j, lane), bh0 = e3_tr(H0H, kk, j, lane);
        g2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bh1, g2[j], 0, 0, 0);
        g1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh0, g1[j], 0, 0, 0);
        if (j < nci0) {
          const bf16x8 bx = e3_tr(XH, kk, j, lane);
          g0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bx, g0[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                         // every tile is free for the next iteration's stores
  }
  // ---- per-workgroup partials: [layer 2 | 1 | 0][co][ci], then the bias sums [2 | 1 | 0][64]
  float* ws = p.ws + (int64_t)blockIdx.x * E3_WS_PER_BLOCK;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = 16 * wave + 4 * q + e, ci = 16 * j + fr;
      ws[(0 * E3_C + co) * E3_C + ci] = g2[j][e];
      ws[(1 * E3_C + co) * E3_C + ci] = g1[j][e];
      ws[(2 * E3_C + co) * E3_C + ci] = j < nci0 ? g0[j][e] : 0.f;
    }
  // bias sums: sb2 per (thread: channels 4 (tid & 15) .., pixels of its rows) -> 16 threads share a channel quad;
  // sb1 / sb0 per lane (channels 16 wave + 4 q .., pixel column fr) -> the 16 lanes of a q-group share them
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][tid >> 4][(tid & 15) * 4 + e] = sb2[e];
    red[1][fr][16 * wave + 4 * q + e] = sb1[e];
    red[2][fr][16 * wave + 4 * q + e] = sb0[e];
  }
  __syncthreads();
  if (tid < 3 * E3_C) {
    const int l = tid / E3_C, c = tid - l * E3_C;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[l][r][c];
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

static int e3_grid_bwd() { return 512; }

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
  hipLaunchKernelGGL(embed3_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("embed3_fwd");
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
               "embed3_bwd: M must be B * S * HW when the gradient of the spp mean is given");
  p.wt1 = (const u16*)wt1; p.wt2 = (const u16*)wt2; p.gy = gy; p.gm = gm; p.S = S > 0 ? S : 1; p.HW = HW > 0 ? HW : M;
  p.gm_scale = gm_scale; p.ws = (float*)workspace;
  WCMC_REQUIRE((!gy || (gy_pixel_stride >= 64 && gy_pixel_stride % 4 == 0)) && (!gm || (gm_pixel_stride >= 64 && gm_pixel_stride % 4 == 0)),
               WCMC_ERR_ALIGNMENT, "embed3_bwd: pixel strides must be multiples of 4 floats and >= 64");
  p.gy_ps = gy ? gy_pixel_stride : 64; p.gm_ps = gm ? gm_pixel_stride : 64;
  const int64_t gyb = gy ? ((M - 1) * p.gy_ps + 64) * 4 : 0, gmb = gm ? ((M / p.S - 1) * p.gm_ps + 64) * 4 : 0;
  WCMC_REQUIRE(gyb < 0x7ff00000ll && gmb < 0x7ff00000ll, WCMC_ERR_BAD_ARG, "embed3_bwd: gradient view spans more than 2 GiB");
  p.gy_bytes = (unsigned)gyb; p.gm_bytes = (unsigned)gmb;
  const int nblk = e3_grid_bwd();
  hipLaunchKernelGGL(embed3_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  if (int rc = check_launch("embed3_bwd")) return rc;
  hipLaunchKernelGGL(embed3_bwd_finish_kernel, dim3((unsigned)((E3_WS_PER_BLOCK + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nblk, Cin, dw0, db0, dw1, db1, dw2, db2);
  return check_launch("embed3_bwd_finish");
}
