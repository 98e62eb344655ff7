// Kernels shared by the fp32 (conv.hip) and split-bf16 (conv_bf16x3.hip) convolution paths.
// `static`: each translation unit gets its own device copy (the library is built without -fgpu-rdc).
#pragma once
#include "common.h"

namespace wcmc {

// dW[co][ci][tap] = sum_s slab[s][tap][co][ci].  One block = one cout x 64 cins x all taps:
// slab reads are coalesced along ci, the OIHW write is contiguous ((ci, tap) row-major) after an
// LDS transpose.  The s-loop runs in a fixed order -> bitwise reproducible.
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                           int S, int taps, int Cout, int Cin, int Np, int Cq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [64][taps + 1]
  const int co = blockIdx.y, ci0 = blockIdx.x * 64;
  const int LD = taps + 1;
  const int64_t sstride = (int64_t)taps * Np * Cq;
  for (int e = threadIdx.x; e < taps * 64; e += 256) {
    const int tap = e >> 6, cl = e & 63;
    float acc = 0.f;
    if (ci0 + cl < Cin) {
      const float* q = slabs + ((int64_t)tap * Np + co) * Cq + ci0 + cl;
      for (int s = 0; s < S; ++s) acc += q[s * sstride];
    }
    smem[cl * LD + tap] = acc;
  }
  __syncthreads();
  const int ncl = min(64, Cin - ci0);
  float* out = dw + ((int64_t)co * Cin + ci0) * taps;
  for (int e = threadIdx.x; e < ncl * taps; e += 256) {
    const int cl = e / taps, tap = e - cl * taps;
    out[e] = smem[cl * LD + tap];
  }
}

static __global__ void colsum_final_kernel(const float* __restrict__ partial, int G, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int g = 0; g < G; ++g) acc += partial[(int64_t)g * C + c];
  out[c] = acc;
}


}  // namespace wcmc
