// Weight normalisation of every conv layer of a model in ONE launch per direction.
//
// sbmc.modules.ConvChain wraps its nn.Conv2d layers in torch.nn.utils.weight_norm unless told otherwise, and
// support/networks.py:18-24 (PathNet) does not tell it otherwise: each layer's weight is  w = g * v / ||v||  with the norm
// over (in, kh, kw) per output channel.  PathNet has 20 such layers (2,243 output channels, 2.9 M weights); the step forms
// all of them with one launch before the model's first chain and turns all 20 weight gradients into (dg, dv) with one launch
// behind the model's last weight-gradient GEMM.
//
// HBM-bound byte work (forward 2 x 4 B per weight, backward 3 x 4 B; 23 / 35 MB per PathNet): one wavefront per output
// channel, float4 accesses along the contiguous (in, kh, kw) run, the two reductions as 64-lane butterflies.  Rows are short
// (36 .. 3,456 floats), so the second pass over a row in either kernel is served by the L1 / L2 the first pass filled.
#include "common.h"

namespace wcmc {

constexpr int WN_MAX = 32;
struct WnEntry {
  const float* v; const float* g; const float* dw;   // dw: backward only
  float* w; float* norm;                              // forward outputs (backward: norm is an input)
  float* dv; float* dg;
  int rows, len;
  unsigned row0;                                      // first global row of this layer
};
struct WnTable { WnEntry e[WN_MAX]; int n; unsigned total_rows; };

__device__ __forceinline__ const WnEntry& wn_find(const WnTable& t, unsigned row, int& k) {
  k = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i)
    if (row >= t.e[i].row0) k = i;
  return t.e[k];
}

// w = v * (g / ||v||) -- the association torch._weight_norm uses -- and ||v|| kept for the backward.
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(WnTable t) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= t.total_rows) return;
  const int lane = threadIdx.x & 63;
  int k;
  const WnEntry& q = wn_find(t, row, k);
  const int r = (int)(row - q.row0), len = q.len;
  const float* v = q.v + (int64_t)r * len;
  float* w = q.w + (int64_t)r * len;
  float ss = 0.f;
  const bool vec = (len & 3) == 0;
  if (vec) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    for (int i = lane; i < len / 4; i += 64) {
      const f32x4 a = v4[i];
      ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
  } else {
    for (int i = lane; i < len; i += 64) ss += v[i] * v[i];
  }
  const float nrm = sqrtf(wave_sum(ss));
  const float s = q.g[r] / nrm;
  if (lane == 0) q.norm[r] = nrm;
  if (vec) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    f32x4* w4 = reinterpret_cast<f32x4*>(w);
    for (int i = lane; i < len / 4; i += 64) w4[i] = v4[i] * s;
  } else {
    for (int i = lane; i < len; i += 64) w[i] = v[i] * s;
  }
}

// dg = <dw, v> / ||v||;  dv = (g / ||v||) * (dw - v * <dw, v> / ||v||^2)   (torch._weight_norm_interface_backward)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(WnTable t) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= t.total_rows) return;
  const int lane = threadIdx.x & 63;
  int k;
  const WnEntry& q = wn_find(t, row, k);
  const int r = (int)(row - q.row0), len = q.len;
  const float* v = q.v + (int64_t)r * len;
  const float* dw = q.dw + (int64_t)r * len;
  float* dv = q.dv + (int64_t)r * len;
  float dot = 0.f;
  const bool vec = (len & 3) == 0;
  if (vec) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dw);
    for (int i = lane; i < len / 4; i += 64) {
      const f32x4 a = v4[i], b = d4[i];
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (int i = lane; i < len; i += 64) dot += v[i] * dw[i];
  }
  dot = wave_sum(dot);
  const float nrm = q.norm[r];
  const float s = q.g[r] / nrm, c = dot / (nrm * nrm);
  if (lane == 0) q.dg[r] = dot / nrm;
  if (vec) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dw);
    f32x4* o4 = reinterpret_cast<f32x4*>(dv);
    for (int i = lane; i < len / 4; i += 64) o4[i] = (d4[i] - v4[i] * c) * s;
  } else {
    for (int i = lane; i < len; i += 64) dv[i] = (dw[i] - v[i] * c) * s;
  }
}

static int wn_table(WnTable& t, const char* what, int n_layers, const int* rows, const int* row_len) {
  WCMC_REQUIRE(n_layers > 0 && n_layers <= WN_MAX && rows && row_len, WCMC_ERR_BAD_ARG, "%s: bad argument (1..%d layers)", what, WN_MAX);
  t.n = n_layers;
  unsigned total = 0;
  for (int i = 0; i < n_layers; ++i) {
    WCMC_REQUIRE(rows[i] > 0 && row_len[i] > 0, WCMC_ERR_BAD_ARG, "%s: layer %d has no rows", what, i);
    t.e[i].rows = rows[i]; t.e[i].len = row_len[i]; t.e[i].row0 = total;
    total += (unsigned)rows[i];
  }
  t.total_rows = total;
  return WCMC_OK;
}

}  // namespace wcmc
using namespace wcmc;

extern "C" int wcmc_weight_norm_fwd(int n_layers, const float* const* v, const float* const* g, float* const* w,
                                    float* const* norm, const int* rows, const int* row_len, void* stream) {
  WnTable t;
  if (int rc = wn_table(t, "weight_norm_fwd", n_layers, rows, row_len)) return rc;
  WCMC_REQUIRE(v && g && w && norm, WCMC_ERR_BAD_ARG, "weight_norm_fwd: null pointer table");
  for (int i = 0; i < n_layers; ++i) {
    WCMC_REQUIRE(v[i] && g[i] && w[i] && norm[i], WCMC_ERR_BAD_ARG, "weight_norm_fwd: null pointer in layer %d", i);
    WCMC_REQUIRE(row_len[i] % 4 != 0 || (aligned16(v[i]) && aligned16(w[i])), WCMC_ERR_ALIGNMENT,
                 "weight_norm_fwd: layer %d is not 16-byte aligned", i);
    t.e[i].v = v[i]; t.e[i].g = g[i]; t.e[i].w = w[i]; t.e[i].norm = norm[i];
    t.e[i].dw = nullptr; t.e[i].dv = nullptr; t.e[i].dg = nullptr;
  }
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((t.total_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, t);
  return check_launch("weight_norm_fwd");
}

extern "C" int wcmc_weight_norm_bwd(int n_layers, const float* const* dw, const float* const* v, const float* const* g,
                                    const float* const* norm, float* const* dv, float* const* dg, const int* rows,
                                    const int* row_len, void* stream) {
  WnTable t;
  if (int rc = wn_table(t, "weight_norm_bwd", n_layers, rows, row_len)) return rc;
  WCMC_REQUIRE(dw && v && g && norm && dv && dg, WCMC_ERR_BAD_ARG, "weight_norm_bwd: null pointer table");
  for (int i = 0; i < n_layers; ++i) {
    WCMC_REQUIRE(dw[i] && v[i] && g[i] && norm[i] && dv[i] && dg[i], WCMC_ERR_BAD_ARG, "weight_norm_bwd: null pointer in layer %d", i);
    WCMC_REQUIRE(row_len[i] % 4 != 0 || (aligned16(v[i]) && aligned16(dw[i]) && aligned16(dv[i])), WCMC_ERR_ALIGNMENT,
                 "weight_norm_bwd: layer %d is not 16-byte aligned", i);
    t.e[i].v = v[i]; t.e[i].g = g[i]; t.e[i].dw = dw[i]; t.e[i].norm = const_cast<float*>(norm[i]);
    t.e[i].dv = dv[i]; t.e[i].dg = dg[i]; t.e[i].w = nullptr;
  }
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((t.total_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, t);
  return check_launch("weight_norm_bwd");
}
