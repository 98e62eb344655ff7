// Kernels shared by the fp32 (conv.hip) and split-bf16 (conv_bf16x3.hip) convolution paths.
// `static`: each translation unit gets its own device copy (the library is built without -fgpu-rdc).
#pragma once
#include "common.h"

namespace wcmc {

// dW[co][ci][tap] = sum_s slab[s][tap][co][ci].  One block = one cout x 32 cins x all taps:
// slab reads are coalesced along ci (128 B per half wave), the OIHW write is contiguous ((ci, tap)
// row-major) after an LDS transpose.  The s-loop runs in a fixed order -> bitwise reproducible.
constexpr int WR_CI = 32;
//
// Rows blockIdx.y >= Cout of the grid (present when cs_partial is given; blockIdx.x == 0 only) finish the BIAS gradient
// from the per-tile column sums of dy that the launch producing dy left: 64 channels per block, the 16 row groups of
// colsum_final_strided_kernel (four per quarter of the block), same order of additions -- the bias gradient costs no
// launch of its own.
// (bx, by) = the block's position in the launch's (Cin / 32, Cout + bias rows) grid: wgrad_reduce_kernel passes blockIdx,
// wgrad_reduce_multi_kernel -- the reductions of several layers in one launch -- a position inside its entry's share of a flat grid.
static __device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slabs, float* __restrict__ dw,
                                                         int S, int taps, int Cout, int Cin, int Np, int Cq,
                                                         const float* __restrict__ cs_partial, int cs_gmax,
                                                         int cs_ld, float* __restrict__ db, int bx, int by, float* smem) {
  if (by >= Cout) {
    if (bx != 0) return;
    float (*red)[64] = reinterpret_cast<float (*)[64]>(smem);
    const int cl = threadIdx.x & 63, q4 = threadIdx.x >> 6;
    const int c = (by - Cout) * 64 + cl;
    const int G = min(cs_gmax, reinterpret_cast<const int*>(cs_partial)[(int64_t)cs_gmax * cs_ld]);
    for (int gg = q4; gg < 16; gg += 4) {
      float acc = 0.f;
      if (c < Cout) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int g = gg;
        for (; g + 7 * 16 < G; g += 8 * 16) {
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] += cs_partial[(int64_t)(g + u * 16) * cs_ld + c];
        }
        for (; g < G; g += 16) a[0] += cs_partial[(int64_t)g * cs_ld + c];
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
      red[gg][cl] = acc;
    }
    __syncthreads();
    if (q4 == 0 && c < Cout) {
      float t = red[0][cl];
#pragma unroll
      for (int q = 1; q < 16; ++q) t += red[q][cl];
      db[c] = t;
    }
    return;
  }
  const int co = by, ci0 = bx * WR_CI;
  const int LD = taps + 1;
  const int64_t sstride = (int64_t)taps * Np * Cq;
  if (taps * WR_CI * 2 <= 256) {
    // few taps (the 1x1 layers: 32 sums of S = 512 slabs per block, i.e. 43 dependent round trips for 32 of the 256
    // threads -- 80-130 us per layer, as long as the GEMM itself): the slab range is cut into SG contiguous groups, one
    // per (taps * 32)-thread slice of the block, and the group sums are added in group order (fixed order -> reproducible)
    const int TE = taps * WR_CI, SG = 256 / TE;
    const int e = threadIdx.x % TE, sg = threadIdx.x / TE;
    const int tap = e / WR_CI, cl = e - tap * WR_CI;
    float* part = smem + WR_CI * LD;                    // [SG][TE]
    float acc = 0.f;
    if (sg < SG && ci0 + cl < Cin) {
      const int chunk = (S + SG - 1) / SG, s0 = sg * chunk, s1 = min(S, s0 + chunk);
      const float* q = slabs + ((int64_t)tap * Np + co) * Cq + ci0 + cl;
      int s = s0;
      for (; s + 12 <= s1; s += 12) {
        float v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = q[(s + u) * sstride];
#pragma unroll
        for (int u = 0; u < 12; ++u) acc += v[u];
      }
      for (; s < s1; ++s) acc += q[s * sstride];
    }
    if (sg < SG) part[sg * TE + e] = acc;
    __syncthreads();
    if (sg == 0) {
      float t = part[e];
      for (int g = 1; g < SG; ++g) t += part[g * TE + e];
      smem[cl * LD + tap] = t;
    }
  } else
  // one thread = four neighbouring input channels of one tap (16-byte loads: a (tap, cout) row of the block is 128 B),
  // 12 slabs in flight, every element summed in slab order as before (the scalar version of this loop moved the
  // 60 MB of a KPCN layer's slabs at 1.3 TB/s: 47 us, a quarter of the GEMM that wrote them)
  for (int e = threadIdx.x; e < taps * (WR_CI / 4); e += 256) {
    const int tap = e / (WR_CI / 4), cl = (e - tap * (WR_CI / 4)) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ci0 + cl < Cq) {
      const float* q = slabs + ((int64_t)tap * Np + co) * Cq + ci0 + cl;
      int s = 0;
      for (; s + 12 <= S; s += 12) {
        float4 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = *reinterpret_cast<const float4*>(q + (s + u) * sstride);
#pragma unroll
        for (int u = 0; u < 12; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; s + 4 <= S; s += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(q + (s + u) * sstride);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(q + s * sstride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    smem[(cl + 0) * LD + tap] = acc.x; smem[(cl + 1) * LD + tap] = acc.y;
    smem[(cl + 2) * LD + tap] = acc.z; smem[(cl + 3) * LD + tap] = acc.w;
  }
  __syncthreads();
  const int ncl = min(WR_CI, Cin - ci0);
  float* out = dw + ((int64_t)co * Cin + ci0) * taps;
  for (int e = threadIdx.x; e < ncl * taps; e += 256) {
    const int cl = e / taps, tap = e - cl * taps;
    out[e] = smem[cl * LD + tap];
  }
}

static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                  int S, int taps, int Cout, int Cin, int Np, int Cq,
                                                                  const float* __restrict__ cs_partial, int cs_gmax,
                                                                  int cs_ld, float* __restrict__ db) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [WR_CI][taps + 1] (>= 16 x 64 floats with cs_partial)
  wgrad_reduce_body(slabs, dw, S, taps, Cout, Cin, Np, Cq, cs_partial, cs_gmax, cs_ld, db, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// The slab reductions (and bias-gradient finishes) of up to WRM_MAX layers in ONE launch: a U-Net level's layers have a few
// hundred reduction blocks of 5-15 us each -- fifteen launches per PathNet whose kernels do not fill the chip and whose
// boundaries cost as much as they do.  Same arithmetic per block as wgrad_reduce_kernel (bit-identical results).
constexpr int WRM_MAX = 32;
struct WRMEntry { const float* slabs; float* dw; const float* cs_partial; float* db;
                  int S, taps, Cout, Cin, Np, Cq, cs_gmax, cs_ld, gx; unsigned block0; };
struct WRMTable { WRMEntry e[WRM_MAX]; int n; };
static __global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WRMTable t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i)
    if (blockIdx.x >= t.e[i].block0) k = i;
  const WRMEntry& q = t.e[k];
  const int local = (int)(blockIdx.x - q.block0);
  wgrad_reduce_body(q.slabs, q.dw, q.S, q.taps, q.Cout, q.Cin, q.Np, q.Cq, q.cs_partial, q.cs_gmax, q.cs_ld, q.db,
                    local % q.gx, local / q.gx, smem);
}

// out[c] = sum_g partial[g][c]; 16 g-groups x 64 channels per 1024-thread block, fixed order.
static __global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int G, int C,
                                                                   float* __restrict__ out) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, gg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc = 0.f;
  if (c < C)
    for (int g = gg; g < G; g += 16) acc += partial[(int64_t)g * C + c];
  red[gg][cl] = acc;
  __syncthreads();
  if (gg == 0 && c < C) {
    float t = red[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][cl];
    out[c] = t;
  }
}

// same with a row pitch (partial[g][ld], first C columns); the producing launch left the number of rows it wrote
// in the trailer word partial[Gmax * ld] (one row per workgroup: 512 for the persistent 1x1 kernel on 8192 tiles)
static __global__ __launch_bounds__(1024) void colsum_final_strided_kernel(const float* __restrict__ partial, int Gmax,
                                                                           int ld, int C, float* __restrict__ out) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, gg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int G = min(Gmax, reinterpret_cast<const int*>(partial)[(int64_t)Gmax * ld]);
  float acc = 0.f;
  if (c < C) {
    // eight independent loads in flight per thread (one block reduces up to 8192 tile rows: latency-bound)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int g = gg;
    for (; g + 7 * 16 < G; g += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += partial[(int64_t)(g + u * 16) * ld + c];
    }
    for (; g < G; g += 16) a[0] += partial[(int64_t)g * ld + c];
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[gg][cl] = acc;
  __syncthreads();
  if (gg == 0 && c < C) {
    float t = red[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][cl];
    out[c] = t;
  }
}

}  // namespace wcmc
