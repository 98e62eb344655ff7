// Per-pixel softmax(k*k logits) x zero-extended k x k gather of C-channel radiance.
//
// Replaces sbmc.modules.KernelApply(softmax=True, splat=False) inside sbmc.KPCN
// (call site support/interfaces.py:203-204; upstream: Halide kernel_weighting op).
//
// HBM-bound: per pixel the forward streams k*k*4 B of logits (1764 B for 21x21) against
// 12 B of radiance in / 12 B out; the backward reads the logits again and writes as many
// bytes of d_logits.  Layout: logits are pixel-major (NHWC view, taps contiguous), which
// is what the conv epilogue writes.  16 lanes own one pixel: each lane loads 16-byte
// vectors j, j+16, ... of the pixel's tap row (so a wave instruction reads 4 x 256
// contiguous bytes), keeps its <= 28 logits in registers for the two softmax passes, and
// gathers radiance from an LDS-resident (tile + 2r)^2 halo of float4 pixels.  The three
// reductions (max, sum, weighted rgb) are 4-step xor-shuffles inside the 16-lane group.
#include "common.h"

namespace wcmc {

constexpr int KA_TILE = 8;          // 8x8 output pixels per 256-thread block
constexpr int KA_MAXV = 7;          // float4 per lane: 16*7*4 = 448 >= 441 taps

struct KAParams {
  const float* logits; int64_t lsn, lsh, lsw;
  const float* data; int64_t dsn, dsc, dsh, dsw;
  const float* out; int64_t osn, osc, osh, osw;           // fwd: written, bwd: read
  const float* gout; int64_t gsn, gsc, gsh, gsw;          // bwd only
  float* lse;                                             // fwd: written (may be null), bwd: read
  float* dlogits; int64_t qsn, qsh, qsw;                  // bwd only
  unsigned short* dlsplit; int Cps;                       // bwd only: d_logits as a dense split tensor [pixel][hi Cps][lo Cps] instead
  float* ddata;                                           // bwd only, may be null
  int N, C, h, w, k, r, taps, nvec, halo;
};

// Reductions over the 16-lane group that owns a pixel: a DPP row is 16 lanes, and four row
// rotations (8, 4, 2, 1) leave the full sum / max in every lane -- one VALU op per step, no LDS.
template <int CTRL>
__device__ __forceinline__ float dpp_ror(float v) {
  const int iv = __builtin_bit_cast(int, v);      // old = src: every lane is enabled, no zero-init mov
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, dpp_ror<0x128>(v));
  v = fmaxf(v, dpp_ror<0x124>(v));
  v = fmaxf(v, dpp_ror<0x122>(v));
  v = fmaxf(v, dpp_ror<0x121>(v));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_ror<0x128>(v);
  v += dpp_ror<0x124>(v);
  v += dpp_ror<0x122>(v);
  v += dpp_ror<0x121>(v);
  return v;
}

// Stage the zero-extended radiance halo of this tile as one float4 per pixel:
//   C <= 3:  {c0, c1, c2, 1}  -- the constant 1 makes sum_t w_t fall out of the same packed FMAs that
//            accumulate the weighted radiance (it is 1 outside the image too: the softmax runs over
//            all k*k taps, only the radiance is zero-extended);
//   C == 4:  {c0, c1, c2, c3}.
template <bool C4>
__device__ __forceinline__ void ka_load_halo(const KAParams& p, float4* halo, int n, int ty0, int tx0) {
  const int hs = p.halo;
  for (int i = threadIdx.x; i < hs * hs; i += blockDim.x) {
    const int hy = i / hs, hx = i - hy * hs;
    const int y = ty0 + hy - p.r, x = tx0 + hx - p.r;
    float v[4] = {0.f, 0.f, 0.f, C4 ? 0.f : 1.f};
    if ((unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w) {
      const float* d = p.data + (int64_t)n * p.dsn + (int64_t)y * p.dsh + (int64_t)x * p.dsw;
      for (int c = 0; c < p.C; ++c) v[c] = d[(int64_t)c * p.dsc];
    }
    halo[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// KS > 0: compile-time kernel size (21 on the KPCN path: tap -> (dy,dx) divisions fold to multiplies);
// KS == 0: runtime p.k.  128 threads = 2 waves per 8x8 tile, 8 pixel quads per wave.
constexpr float KA_NEG = -1.0e30f;    // logit of a tap slot beyond k*k: exp() of it is exactly 0

template <bool BWD, int KS, bool C4>
__global__ __launch_bounds__(128) void kernel_apply_kernel(KAParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* halo = reinterpret_cast<float4*>(smem);
  float* dacc = smem + 4 * p.halo * p.halo;       // BWD with d_data: per-halo-pixel float4 accumulators

  const int tiles_x = (p.w + KA_TILE - 1) / KA_TILE;
  const int n = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * KA_TILE, tx0 = (blockIdx.x % tiles_x) * KA_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int kk = KS > 0 ? KS : p.k;
  const int taps = KS > 0 ? KS * KS : p.taps;
  const int nvec = (taps + 3) / 4;
  constexpr int NIT = 8;                          // 64 pixels / 2 waves / 4 pixels per quad
  constexpr float LOG2E = 1.4426950408889634f;

  // The logits stream is the whole cost: put the first pixel quad's loads in flight before anything
  // else, and keep one quad ahead of the arithmetic after that.  Slots beyond k*k get KA_NEG so that
  // the arithmetic below needs no per-tap predicate.
  auto load_quad = [&](int it, float4* lv) {
    const int pi = wave * 32 + it * 4 + q;
    const int y = ty0 + (pi >> 3), x = tx0 + (pi & 7);
    const bool valid = y < p.h && x < p.w;
    const float* lrow = p.logits + (int64_t)n * p.lsn + (int64_t)y * p.lsh + (int64_t)x * p.lsw;
#pragma unroll
    for (int i = 0; i < KA_MAXV; ++i) {
      const int vi = j + 16 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && vi < nvec) v = *reinterpret_cast<const float4*>(lrow + 4 * vi);
      v.x *= LOG2E; v.y *= LOG2E; v.z *= LOG2E; v.w *= LOG2E;      // softmax in base 2 from here on
      if (4 * (15 + 16 * i) + 3 >= taps) {        // only the last vector(s) can run past k*k
        const int t = 4 * vi;
        if (t + 0 >= taps) v.x = KA_NEG;
        if (t + 1 >= taps) v.y = KA_NEG;
        if (t + 2 >= taps) v.z = KA_NEG;
        if (t + 3 >= taps) v.w = KA_NEG;
      }
      lv[i] = v;
    }
  };
  float4 lvA[KA_MAXV], lvB[KA_MAXV];
  load_quad(0, lvA);

  ka_load_halo<C4>(p, halo, n, ty0, tx0);
  if (BWD && p.ddata)
    for (int i = threadIdx.x; i < 4 * p.halo * p.halo; i += blockDim.x) dacc[i] = 0.f;

  // byte offset into the halo of each of this lane's taps (slot beyond k*k -> tap 0: weight is 0)
  int toff[KA_MAXV][4];
#pragma unroll
  for (int i = 0; i < KA_MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = 4 * (j + 16 * i) + e;
      const int dy = t / kk, dx = t - dy * kk;
      toff[i][e] = t < taps ? (dy * p.halo + dx) * 16 : 0;
    }
  __syncthreads();
  const char* halo_b = reinterpret_cast<const char*>(halo);

#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float4* lv = (it & 1) ? lvB : lvA;
    if (it + 1 < NIT) load_quad(it + 1, (it & 1) ? lvA : lvB);
    const int pi = wave * 32 + it * 4 + q;            // pixel inside the 8x8 tile
    const int ty = pi >> 3, tx = pi & 7;
    const int y = ty0 + ty, x = tx0 + tx;
    const bool valid = y < p.h && x < p.w;            // uniform across the 16-lane group
    const int64_t pix = ((int64_t)n * p.h + y) * p.w + x;
    const char* hbase = halo_b + (ty * p.halo + tx) * 16;   // tap (dy,dx) reads hbase + toff

    if (!BWD) {
      float m = KA_NEG;
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) m = fmaxf(fmaxf(m, fmaxf(lv[i].x, lv[i].y)), fmaxf(lv[i].z, lv[i].w));
      m = group16_max(m);                               // max of the base-2 logits
      float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) {
        const float* l = reinterpret_cast<const float*>(&lv[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = __builtin_amdgcn_exp2f(l[e] - m);
          const float4 d = *reinterpret_cast<const float4*>(hbase + toff[i][e]);
          if (C4) {
            s += ex; a0 = fmaf(ex, d.x, a0); a1 = fmaf(ex, d.y, a1); a2 = fmaf(ex, d.z, a2); a3 = fmaf(ex, d.w, a3);
          } else {                                  // d = {r, g, b, 1}: two packed FMAs per tap
            a0 = fmaf(ex, d.x, a0); a1 = fmaf(ex, d.y, a1); a2 = fmaf(ex, d.z, a2); s = fmaf(ex, d.w, s);
          }
        }
      }
      s = group16_sum(s); a0 = group16_sum(a0); a1 = group16_sum(a1); a2 = group16_sum(a2);
      if (C4) a3 = group16_sum(a3);
      if (valid && j == 0) {
        const float inv = 1.f / s;
        float* o = const_cast<float*>(p.out) + (int64_t)n * p.osn + (int64_t)y * p.osh + (int64_t)x * p.osw;
        const float av[4] = {a0 * inv, a1 * inv, a2 * inv, a3 * inv};
        for (int c = 0; c < p.C; ++c) o[(int64_t)c * p.osc] = av[c];
        if (p.lse) p.lse[pix] = (m + __builtin_amdgcn_logf(s)) * 0.6931471805599453f;   // natural-log LSE
      }
    } else {
      // w_t = exp(l_t - lse);  d l_t = w_t * (g . data_t - g . out)
      float g[4] = {0.f, 0.f, 0.f, 0.f};
      float go = 0.f, lse = 0.f;
      if (valid) {
        const float* gp = p.gout + (int64_t)n * p.gsn + (int64_t)y * p.gsh + (int64_t)x * p.gsw;
        const float* op = p.out + (int64_t)n * p.osn + (int64_t)y * p.osh + (int64_t)x * p.osw;
        for (int c = 0; c < p.C; ++c) {
          g[c] = gp[(int64_t)c * p.gsc];
          go += g[c] * op[(int64_t)c * p.osc];
        }
        lse = p.lse[pix];
      }
      const float lb = lse * LOG2E;
      float* qrow = p.dlogits + (int64_t)n * p.qsn + (int64_t)y * p.qsh + (int64_t)x * p.qsw;
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) {
        const float* l = reinterpret_cast<const float*>(&lv[i]);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float wt = __builtin_amdgcn_exp2f(l[e] - lb);
          const float4 d = *reinterpret_cast<const float4*>(hbase + toff[i][e]);
          const float gd = C4 ? fmaf(g[0], d.x, fmaf(g[1], d.y, fmaf(g[2], d.z, g[3] * d.w)))
                              : fmaf(g[0], d.x, fmaf(g[1], d.y, g[2] * d.z));
          o[e] = wt * (gd - go);
          if (p.ddata && valid && 4 * (j + 16 * i) + e < taps) {
            float* da = dacc + ((ty * p.halo + tx) * 16 + toff[i][e]) / 4;     // channel-major like the gradient
            atomicAdd(da + 0, wt * g[0]); atomicAdd(da + 1, wt * g[1]);
            atomicAdd(da + 2, wt * g[2]); atomicAdd(da + 3, wt * g[3]);
          }
        }
        const int vi = j + 16 * i;
        if (p.dlsplit) {
          // straight into the conv chain's split-bf16 gradient layout (hi = bf16(v), lo = bf16(v - hi)); slots past
          // k*k are exact zeros (their weight is exp2(-1e30)): the pad channels of the split tensor
          if (valid && 4 * vi < p.Cps) {
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              hi[e] = __builtin_bit_cast(unsigned short, (__bf16)o[e]);
              const float hf = __builtin_bit_cast(float, (unsigned)hi[e] << 16);
              lo[e] = __builtin_bit_cast(unsigned short, (__bf16)(o[e] - hf));
            }
            unsigned short* q = p.dlsplit + pix * 2 * p.Cps + 4 * vi;
            *reinterpret_cast<uint2*>(q) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
            *reinterpret_cast<uint2*>(q + p.Cps) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
          }
        } else if (valid && vi < nvec) {
          *reinterpret_cast<float4*>(qrow + 4 * vi) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }

  if (BWD && p.ddata) {
    __syncthreads();
    const int hs = p.halo;
    for (int i = threadIdx.x; i < hs * hs; i += blockDim.x) {
      const int hy = i / hs, hx = i - hy * hs;
      const int y = ty0 + hy - p.r, x = tx0 + hx - p.r;
      if ((unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w)
        for (int c = 0; c < p.C; ++c) {
          const float v = dacc[4 * i + c];
          if (v != 0.f) atomicAdd(p.ddata + (((int64_t)n * p.C + c) * p.h + y) * p.w + x, v);
        }
    }
  }
}

static int ka_fill(KAParams& p, int N, int C, int h, int w, int k) {
  WCMC_REQUIRE(N > 0 && h > 0 && w > 0 && C >= 1 && C <= 4 && k >= 1 && (k & 1) && k * k <= 16 * KA_MAXV * 4,
               WCMC_ERR_BAD_ARG, "kernel_apply: unsupported shape (N=%d C=%d h=%d w=%d k=%d)", N, C, h, w, k);
  p.N = N; p.C = C; p.h = h; p.w = w; p.k = k; p.r = k / 2; p.taps = k * k; p.nvec = (k * k + 3) / 4;
  p.halo = KA_TILE + k - 1;
  return 0;
}

}  // namespace wcmc

using namespace wcmc;

extern "C" int wcmc_kernel_apply_fwd(const float* logits, int64_t lsn, int64_t lsh, int64_t lsw, const float* data,
                                     int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw, float* out, int64_t osn,
                                     int64_t osc, int64_t osh, int64_t osw, float* lse, int N, int C, int h, int w,
                                     int k, void* stream) {
  KAParams p = {};
  if (int rc = ka_fill(p, N, C, h, w, k)) return rc;
  WCMC_REQUIRE(data && out, WCMC_ERR_BAD_ARG, "kernel_apply_fwd: null pointer");
  WCMC_REQUIRE(nhwc_view_ok(logits, lsn, lsh, lsw, k * k), WCMC_ERR_ALIGNMENT,
               "kernel_apply_fwd: logits violate the NHWC-view contract");
  p.logits = logits; p.lsn = lsn; p.lsh = lsh; p.lsw = lsw;
  p.data = data; p.dsn = dsn; p.dsc = dsc; p.dsh = dsh; p.dsw = dsw;
  p.out = out; p.osn = osn; p.osc = osc; p.osh = osh; p.osw = osw; p.lse = lse;
  const dim3 grid((unsigned)(((h + KA_TILE - 1) / KA_TILE) * ((w + KA_TILE - 1) / KA_TILE)), (unsigned)N);
  const size_t lds = (size_t)p.halo * p.halo * sizeof(float4);
  if (k == 21 && C <= 3) hipLaunchKernelGGL((kernel_apply_kernel<false, 21, false>), grid, dim3(128), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((kernel_apply_kernel<false, 0, true>), grid, dim3(128), lds, (hipStream_t)stream, p);
  return check_launch("kernel_apply_fwd");
}

static int ka_bwd_launch(KAParams& p, int N, int C, int h, int w, int k, void* stream) {
  const dim3 grid((unsigned)(((h + KA_TILE - 1) / KA_TILE) * ((w + KA_TILE - 1) / KA_TILE)), (unsigned)N);
  const size_t lds = (size_t)p.halo * p.halo * sizeof(float4) * (p.ddata ? 2 : 1);
  if (k == 21 && C <= 3) hipLaunchKernelGGL((kernel_apply_kernel<true, 21, false>), grid, dim3(128), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((kernel_apply_kernel<true, 0, true>), grid, dim3(128), lds, (hipStream_t)stream, p);
  return check_launch("kernel_apply_bwd");
}

extern "C" int wcmc_kernel_apply_bwd_split(const float* logits, int64_t lsn, int64_t lsh, int64_t lsw, const float* data,
                                           int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw, const float* out,
                                           int64_t osn, int64_t osc, int64_t osh, int64_t osw, const float* grad_out,
                                           int64_t gsn, int64_t gsc, int64_t gsh, int64_t gsw, const float* lse,
                                           void* d_logits_split, int N, int C, int h, int w, int k, void* stream) {
  KAParams p = {};
  if (int rc = ka_fill(p, N, C, h, w, k)) return rc;
  WCMC_REQUIRE(data && out && grad_out && lse && d_logits_split, WCMC_ERR_BAD_ARG, "kernel_apply_bwd_split: null pointer");
  WCMC_REQUIRE(nhwc_view_ok(logits, lsn, lsh, lsw, k * k) && aligned16(d_logits_split), WCMC_ERR_ALIGNMENT,
               "kernel_apply_bwd_split: logits violate the NHWC-view contract (or the split output is unaligned)");
  WCMC_REQUIRE(round_up(k * k, 8) <= 64 * KA_MAXV, WCMC_ERR_BAD_ARG, "kernel_apply_bwd_split: k*k too large");
  p.logits = logits; p.lsn = lsn; p.lsh = lsh; p.lsw = lsw;
  p.data = data; p.dsn = dsn; p.dsc = dsc; p.dsh = dsh; p.dsw = dsw;
  p.out = out; p.osn = osn; p.osc = osc; p.osh = osh; p.osw = osw;
  p.gout = grad_out; p.gsn = gsn; p.gsc = gsc; p.gsh = gsh; p.gsw = gsw;
  p.lse = const_cast<float*>(lse);
  p.dlogits = nullptr; p.ddata = nullptr;
  p.dlsplit = (unsigned short*)d_logits_split; p.Cps = round_up(k * k, 8);
  return ka_bwd_launch(p, N, C, h, w, k, stream);
}

extern "C" int wcmc_kernel_apply_bwd(const float* logits, int64_t lsn, int64_t lsh, int64_t lsw, const float* data,
                                     int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw, const float* out,
                                     int64_t osn, int64_t osc, int64_t osh, int64_t osw, const float* grad_out,
                                     int64_t gsn, int64_t gsc, int64_t gsh, int64_t gsw, const float* lse,
                                     float* d_logits, int64_t qsn, int64_t qsh, int64_t qsw, float* d_data, int N,
                                     int C, int h, int w, int k, void* stream) {
  KAParams p = {};
  if (int rc = ka_fill(p, N, C, h, w, k)) return rc;
  WCMC_REQUIRE(data && out && grad_out && lse, WCMC_ERR_BAD_ARG, "kernel_apply_bwd: null pointer");
  WCMC_REQUIRE(nhwc_view_ok(logits, lsn, lsh, lsw, k * k) && nhwc_view_ok(d_logits, qsn, qsh, qsw, k * k),
               WCMC_ERR_ALIGNMENT, "kernel_apply_bwd: logits/d_logits violate the NHWC-view contract");
  p.logits = logits; p.lsn = lsn; p.lsh = lsh; p.lsw = lsw;
  p.data = data; p.dsn = dsn; p.dsc = dsc; p.dsh = dsh; p.dsw = dsw;
  p.out = out; p.osn = osn; p.osc = osc; p.osh = osh; p.osw = osw;
  p.gout = grad_out; p.gsn = gsn; p.gsc = gsc; p.gsh = gsh; p.gsw = gsw;
  p.lse = const_cast<float*>(lse);
  p.dlogits = d_logits; p.qsn = qsn; p.qsh = qsh; p.qsw = qsw; p.ddata = d_data;
  const dim3 grid((unsigned)(((h + KA_TILE - 1) / KA_TILE) * ((w + KA_TILE - 1) / KA_TILE)), (unsigned)N);
  const size_t lds = (size_t)p.halo * p.halo * sizeof(float4) * (d_data ? 2 : 1);
  if (k == 21 && C <= 3) hipLaunchKernelGGL((kernel_apply_kernel<true, 21, false>), grid, dim3(128), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((kernel_apply_kernel<true, 0, true>), grid, dim3(128), lds, (hipStream_t)stream, p);
  return check_launch("kernel_apply_bwd");
}
