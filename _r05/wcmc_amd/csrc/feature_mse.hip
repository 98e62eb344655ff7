// FeatureMSE: the contrastive path-disentangling loss (support/losses.py:33-61,63-65,82-113).
//
//   rows i = (b, s, y, x) of P in R^C and of the tonemapped reference R in R^3 (shared by all s)
//   d_i    = 1/2 |P_i - P_pi(i)|^2 - 1/2 |R_i - R_pi(i)|^2
//   loss   = 1/2 mean_i d_i^2   summed over the intra-patch pairing (pi over S*h*w, shared by all b)
//            and the intra-batch pairing (pi over B*S*h*w); non_local=False counts the first twice.
//
// HBM/latency-bound permuted gather: one thread per row, the partner row is a random 4..32 byte
// read.  Partial sums go through a wavefront xor-shuffle tree, one slot per block, and a
// fixed-order final pass -- no float atomics, so the scalar is bitwise reproducible.
// The backward needs the pairing in both directions; the inverse permutations and the
// per-row displacements are left in the workspace by the forward.
#include "common.h"

namespace wcmc {

constexpr int FM_BLOCKS = 1024;
constexpr int FM_MAXC = 8;

struct FMParams {
  const float* p; int64_t psb, pss, psc, psh, psw;
  const float* ref; int64_t rsb, rsc, rsh, rsw;
  const int64_t* idx_patch; const int64_t* idx_batch;
  float* rt;            // [B][h][w][4] tonemapped reference
  float* d_patch; float* d_batch;       // [N]
  int64_t* inv_patch; int64_t* inv_batch;
  float* partial;       // [FM_BLOCKS][4]: sum d_patch^2, sum d_batch^2, max |d|, -
  float* grs;           // GRS scratch: [0] = log-sum-exp, [1] = alpha, [4+blk] = per-block exp sums
  int B, S, C, h, w; int64_t SHW, N;
};

struct FMLayout { size_t rt, d_patch, d_batch, inv_patch, inv_batch, partial, grs, bytes; };
static FMLayout fm_layout(int B, int S, int h, int w) {
  FMLayout L; size_t o = 0;
  const size_t N = (size_t)B * S * h * w, SHW = (size_t)S * h * w;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  L.rt = take((size_t)B * h * w * 4 * sizeof(float));
  L.d_patch = take(N * sizeof(float));
  L.d_batch = take(N * sizeof(float));
  L.inv_patch = take(SHW * sizeof(int64_t));
  L.inv_batch = take(N * sizeof(int64_t));
  L.partial = take((size_t)FM_BLOCKS * 4 * sizeof(float));
  L.grs = take((size_t)(FM_BLOCKS + 4) * sizeof(float));     // [0]=lse, [1]=alpha, [4..] per-block exp sums
  L.bytes = o;
  return L;
}

// (img / (1 + img)) ** 0.454545 on clamp(img, 0)   (losses.py:63-65)
__global__ void fm_tonemap_kernel(FMParams q) {
  const int64_t total = (int64_t)q.B * q.h * q.w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % q.w); int64_t t = i / q.w;
    const int y = (int)(t % q.h); const int b = (int)(t / q.h);
    const float* r = q.ref + (int64_t)b * q.rsb + (int64_t)y * q.rsh + (int64_t)x * q.rsw;
    float v[3];
    for (int c = 0; c < 3; ++c) {
      const float a = fmaxf(r[(int64_t)c * q.rsc], 0.f);
      v[c] = powf(a / (1.f + a), 0.454545f);
    }
    reinterpret_cast<float4*>(q.rt)[i] = make_float4(v[0], v[1], v[2], 0.f);
  }
}

__global__ void fm_inverse_kernel(FMParams q) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.N; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < q.SHW) q.inv_patch[q.idx_patch[i]] = i;
    if (q.idx_batch) q.inv_batch[q.idx_batch[i]] = i;
  }
}

struct RowRef { int b; int64_t poff; int64_t roff; };
__device__ __forceinline__ RowRef fm_row(const FMParams& q, int64_t i) {
  RowRef r;
  const int x = (int)(i % q.w); int64_t t = i / q.w;
  const int y = (int)(t % q.h); t /= q.h;
  const int s = (int)(t % q.S); r.b = (int)(t / q.S);
  r.poff = (int64_t)r.b * q.psb + (int64_t)s * q.pss + (int64_t)y * q.psh + (int64_t)x * q.psw;
  r.roff = (((int64_t)r.b * q.h + y) * q.w + x);
  return r;
}
__device__ __forceinline__ void fm_load_p(const FMParams& q, int64_t poff, float* v) {
#pragma unroll
  for (int c = 0; c < FM_MAXC; ++c) v[c] = c < q.C ? q.p[poff + (int64_t)c * q.psc] : 0.f;
}
__device__ __forceinline__ float fm_disp(const float* pi, float4 ri, const float* pj, float4 rj) {
  float dp = 0.f;
#pragma unroll
  for (int c = 0; c < FM_MAXC; ++c) { const float d = pi[c] - pj[c]; dp += d * d; }
  const float dx = ri.x - rj.x, dy = ri.y - rj.y, dz = ri.z - rj.z;
  return 0.5f * dp - 0.5f * (dx * dx + dy * dy + dz * dz);
}

__global__ __launch_bounds__(256) void fm_fwd_kernel(FMParams q) {
  __shared__ float red[3][4];
  float s1 = 0.f, s2 = 0.f, mx = 0.f;
  const float4* rt = reinterpret_cast<const float4*>(q.rt);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.N; i += (int64_t)gridDim.x * blockDim.x) {
    const RowRef ri = fm_row(q, i);
    float pi[FM_MAXC], pj[FM_MAXC];
    fm_load_p(q, ri.poff, pi);
    const float4 r_i = rt[ri.roff];
    const int64_t j1 = (int64_t)ri.b * q.SHW + q.idx_patch[i - (int64_t)ri.b * q.SHW];
    const RowRef rj = fm_row(q, j1);
    fm_load_p(q, rj.poff, pj);
    const float d1 = fm_disp(pi, r_i, pj, rt[rj.roff]);
    q.d_patch[i] = d1; s1 += d1 * d1; mx = fmaxf(mx, fabsf(d1));
    if (q.idx_batch) {
      const RowRef rk = fm_row(q, q.idx_batch[i]);
      fm_load_p(q, rk.poff, pj);
      const float d2 = fm_disp(pi, r_i, pj, rt[rk.roff]);
      q.d_batch[i] = d2; s2 += d2 * d2; mx = fmaxf(mx, fabsf(d2));
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; red[2][wave] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    q.partial[4 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q.partial[4 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    q.partial[4 * blockIdx.x + 2] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
  }
}

__global__ __launch_bounds__(256) void fm_final_kernel(const float* partial, int nblk, double inv_n, int non_local,
                                                       float* loss) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { s1 += partial[4 * i]; s2 += partial[4 * i + 1]; }
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * 0.5 * inv_n;
    const double b = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * 0.5 * inv_n;
    *loss = (float)(non_local ? a + b : a + a);
  }
}

// dL/dP_i = (gs/N) * sum over pairings [ d_i (P_i - P_pi(i)) + d_i' (P_i - P_i') ],  pi(i') = i
// GRS (losses.py:116-211): with e = alpha*[d_p, d_b, -d_p, -d_b, 0],
//   loss = (logsumexp(e) - log(1 + 4N)) / sqrt(alpha);  dloss/dd_i = sqrt(alpha) * (exp(a d_i - lse) - exp(-a d_i - lse)).
__global__ __launch_bounds__(256) void grs_sum_kernel(FMParams q, int nblk_fwd, float alpha) {
  __shared__ float red[4];
  float mx = 0.f;
  for (int i = threadIdx.x; i < nblk_fwd; i += 256) mx = fmaxf(mx, q.partial[4 * i + 2]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  const float M = alpha * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));     // max of every exponent (>= 0)
  __syncthreads();
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.N; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = alpha * q.d_patch[i], b = alpha * q.d_batch[i];
    acc += __expf(a - M) + __expf(-a - M) + __expf(b - M) + __expf(-b - M);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    q.grs[4 + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0) { q.grs[2] = M; q.grs[1] = alpha; }
  }
}
__global__ __launch_bounds__(256) void grs_final_kernel(FMParams q, int nblk, float alpha, float* loss) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) acc += q.grs[4 + i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double M = q.grs[2];
    const double S = red[0] + red[1] + red[2] + red[3] + exp(-M);        // the appended zero exponent
    const double lse = M + log(S);
    q.grs[0] = (float)lse;
    *loss = (float)((lse - log(1.0 + 4.0 * (double)q.N)) / sqrt((double)alpha));
  }
}

template <bool GRS>
__global__ __launch_bounds__(256) void fm_bwd_kernel(FMParams q, const float* grad_scale, float* dp) {
  const float lse = GRS ? q.grs[0] : 0.f, alpha = GRS ? q.grs[1] : 0.f, sa = GRS ? sqrtf(alpha) : 0.f;
  auto coef = [&](float d) { return GRS ? sa * (__expf(alpha * d - lse) - __expf(-alpha * d - lse)) : d; };
  const float gs = GRS ? grad_scale[0] : grad_scale[0] / (float)q.N * (q.idx_batch ? 1.f : 2.f);
  const int64_t hw = (int64_t)q.h * q.w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.N; i += (int64_t)gridDim.x * blockDim.x) {
    const RowRef ri = fm_row(q, i);
    float pi[FM_MAXC], pj[FM_MAXC], g[FM_MAXC];
    fm_load_p(q, ri.poff, pi);
#pragma unroll
    for (int c = 0; c < FM_MAXC; ++c) g[c] = 0.f;
    const int64_t boff = (int64_t)ri.b * q.SHW, r = i - boff;
    {
      const float d = coef(q.d_patch[i]);
      fm_load_p(q, fm_row(q, boff + q.idx_patch[r]).poff, pj);
#pragma unroll
      for (int c = 0; c < FM_MAXC; ++c) g[c] += d * (pi[c] - pj[c]);
      const int64_t ip = boff + q.inv_patch[r];
      const float d2 = coef(q.d_patch[ip]);
      fm_load_p(q, fm_row(q, ip).poff, pj);
#pragma unroll
      for (int c = 0; c < FM_MAXC; ++c) g[c] += d2 * (pi[c] - pj[c]);
    }
    if (q.idx_batch) {
      const float d = coef(q.d_batch[i]);
      fm_load_p(q, fm_row(q, q.idx_batch[i]).poff, pj);
#pragma unroll
      for (int c = 0; c < FM_MAXC; ++c) g[c] += d * (pi[c] - pj[c]);
      const int64_t ip = q.inv_batch[i];
      const float d2 = coef(q.d_batch[ip]);
      fm_load_p(q, fm_row(q, ip).poff, pj);
#pragma unroll
      for (int c = 0; c < FM_MAXC; ++c) g[c] += d2 * (pi[c] - pj[c]);
    }
    // contiguous (B,S,C,h,w): row i = (b*S+s)*hw + y*w + x
    const int64_t bs = i / hw, yx = i - bs * hw;
    float* o = dp + bs * q.C * hw + yx;
#pragma unroll
    for (int c = 0; c < FM_MAXC; ++c)
      if (c < q.C) o[(int64_t)c * hw] = gs * g[c];
  }
}

static int fm_fill(FMParams& q, const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                   const int64_t* idx_patch, const int64_t* idx_batch, void* ws, size_t ws_bytes, int B, int S, int C,
                   int h, int w) {
  WCMC_REQUIRE(p && idx_patch && ws && B > 0 && S > 0 && C > 0 && C <= FM_MAXC && h > 0 && w > 0, WCMC_ERR_BAD_ARG,
               "feature_mse: bad argument (B=%d S=%d C=%d h=%d w=%d; C <= %d)", B, S, C, h, w, FM_MAXC);
  const FMLayout L = fm_layout(B, S, h, w);
  WCMC_REQUIRE(ws_bytes >= L.bytes && aligned16(ws), WCMC_ERR_WORKSPACE, "feature_mse: workspace %zu < %zu bytes",
               ws_bytes, L.bytes);
  char* base = (char*)ws;
  q.p = p; q.psb = psb; q.pss = pss; q.psc = psc; q.psh = psh; q.psw = psw;
  q.idx_patch = idx_patch; q.idx_batch = idx_batch;
  q.rt = (float*)(base + L.rt); q.d_patch = (float*)(base + L.d_patch); q.d_batch = (float*)(base + L.d_batch);
  q.inv_patch = (int64_t*)(base + L.inv_patch); q.inv_batch = (int64_t*)(base + L.inv_batch);
  q.partial = (float*)(base + L.partial); q.grs = (float*)(base + L.grs);
  q.B = B; q.S = S; q.C = C; q.h = h; q.w = w; q.SHW = (int64_t)S * h * w; q.N = (int64_t)B * q.SHW;
  return 0;
}

}  // namespace wcmc

using namespace wcmc;

extern "C" size_t wcmc_feature_mse_workspace_bytes(int B, int S, int C, int h, int w) {
  (void)C;
  if (B <= 0 || S <= 0 || h <= 0 || w <= 0) return 0;
  return fm_layout(B, S, h, w).bytes;
}

extern "C" int wcmc_feature_mse_fwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                                    const float* ref, int64_t rsb, int64_t rsc, int64_t rsh, int64_t rsw,
                                    const int64_t* idx_patch, const int64_t* idx_batch, float* loss, void* workspace,
                                    size_t workspace_bytes, int B, int S, int C, int h, int w, void* stream) {
  FMParams q = {};
  if (int rc = fm_fill(q, p, psb, pss, psc, psh, psw, idx_patch, idx_batch, workspace, workspace_bytes, B, S, C, h, w))
    return rc;
  WCMC_REQUIRE(ref && loss, WCMC_ERR_BAD_ARG, "feature_mse_fwd: null pointer");
  q.ref = ref; q.rsb = rsb; q.rsc = rsc; q.rsh = rsh; q.rsw = rsw;
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * h * w;
  hipLaunchKernelGGL(fm_tonemap_kernel, dim3((unsigned)(ceil_div64(npix, 256) < 2048 ? ceil_div64(npix, 256) : 2048)),
                     dim3(256), 0, st, q);
  const unsigned gb = (unsigned)(ceil_div64(q.N, 256) < FM_BLOCKS ? ceil_div64(q.N, 256) : FM_BLOCKS);
  hipLaunchKernelGGL(fm_inverse_kernel, dim3(gb), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fm_fwd_kernel, dim3(gb), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fm_final_kernel, dim3(1), dim3(256), 0, st, q.partial, (int)gb, 1.0 / (double)q.N,
                     idx_batch ? 1 : 0, loss);
  return check_launch("feature_mse_fwd");
}

extern "C" int wcmc_feature_mse_bwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                                    const int64_t* idx_patch, const int64_t* idx_batch, const float* grad_scale,
                                    float* dp, void* workspace, size_t workspace_bytes, int B, int S, int C, int h,
                                    int w, void* stream) {
  FMParams q = {};
  if (int rc = fm_fill(q, p, psb, pss, psc, psh, psw, idx_patch, idx_batch, workspace, workspace_bytes, B, S, C, h, w))
    return rc;
  WCMC_REQUIRE(grad_scale && dp, WCMC_ERR_BAD_ARG, "feature_mse_bwd: null pointer");
  const unsigned gb = (unsigned)(ceil_div64(q.N, 256) < 4096 ? ceil_div64(q.N, 256) : 4096);
  hipLaunchKernelGGL(fm_bwd_kernel<false>, dim3(gb), dim3(256), 0, (hipStream_t)stream, q, grad_scale, dp);
  return check_launch("feature_mse_bwd");
}

extern "C" int wcmc_grs_fwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                            const float* ref, int64_t rsb, int64_t rsc, int64_t rsh, int64_t rsw,
                            const int64_t* idx_patch, const int64_t* idx_batch, float alpha, float* loss,
                            void* workspace, size_t workspace_bytes, int B, int S, int C, int h, int w,
                            void* stream) {
  FMParams q = {};
  WCMC_REQUIRE(idx_batch && alpha > 0.f, WCMC_ERR_BAD_ARG, "grs_fwd: needs both pairings and alpha > 0");
  if (int rc = fm_fill(q, p, psb, pss, psc, psh, psw, idx_patch, idx_batch, workspace, workspace_bytes, B, S, C, h, w))
    return rc;
  WCMC_REQUIRE(ref && loss, WCMC_ERR_BAD_ARG, "grs_fwd: null pointer");
  q.ref = ref; q.rsb = rsb; q.rsc = rsc; q.rsh = rsh; q.rsw = rsw;
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * h * w;
  hipLaunchKernelGGL(fm_tonemap_kernel, dim3((unsigned)(ceil_div64(npix, 256) < 2048 ? ceil_div64(npix, 256) : 2048)),
                     dim3(256), 0, st, q);
  const unsigned gb = (unsigned)(ceil_div64(q.N, 256) < FM_BLOCKS ? ceil_div64(q.N, 256) : FM_BLOCKS);
  hipLaunchKernelGGL(fm_inverse_kernel, dim3(gb), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fm_fwd_kernel, dim3(gb), dim3(256), 0, st, q);
  hipLaunchKernelGGL(grs_sum_kernel, dim3(gb), dim3(256), 0, st, q, (int)gb, alpha);
  hipLaunchKernelGGL(grs_final_kernel, dim3(1), dim3(256), 0, st, q, (int)gb, alpha, loss);
  return check_launch("grs_fwd");
}

extern "C" int wcmc_grs_bwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                            const int64_t* idx_patch, const int64_t* idx_batch, const float* grad_scale, float* dp,
                            void* workspace, size_t workspace_bytes, int B, int S, int C, int h, int w, void* stream) {
  FMParams q = {};
  WCMC_REQUIRE(idx_batch, WCMC_ERR_BAD_ARG, "grs_bwd: needs both pairings");
  if (int rc = fm_fill(q, p, psb, pss, psc, psh, psw, idx_patch, idx_batch, workspace, workspace_bytes, B, S, C, h, w))
    return rc;
  WCMC_REQUIRE(grad_scale && dp, WCMC_ERR_BAD_ARG, "grs_bwd: null pointer");
  const unsigned gb = (unsigned)(ceil_div64(q.N, 256) < 4096 ? ceil_div64(q.N, 256) : 4096);
  hipLaunchKernelGGL(fm_bwd_kernel<true>, dim3(gb), dim3(256), 0, (hipStream_t)stream, q, grad_scale, dp);
  return check_launch("grs_bwd");
}

// ---------------------------------------------------------------- pseudo-random permutation of [0, n)
// For the `rng='device'` pairings of FeatureMSE / GRS (wcmc_amd/support/losses.py): torch.randperm on the device
// is four sort passes per call (0.12 ms for 541,696 rows; four calls per step).  A keyed bijection needs no sort:
// a 6-round balanced Feistel network on 2h bits (2^(2h) >= n), cycle-walked back into [0, n).  Every i < n maps
// to a distinct value < n; which permutation is drawn depends on the 64-bit seed only.
namespace wcmc {
__device__ __forceinline__ unsigned perm_mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void random_permutation_kernel(int64_t* __restrict__ out, int64_t n, int h,
                                                                unsigned k0, unsigned k1) {
  const unsigned mask = (1u << h) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i;
    do {
      unsigned l = (unsigned)(x >> h) & mask, r = (unsigned)x & mask;
#pragma unroll
      for (int rd = 0; rd < 6; ++rd) {
        const unsigned f = perm_mix(r ^ (rd & 1 ? k1 : k0) ^ (0x9e3779b9u * (unsigned)(rd + 1))) & mask;
        const unsigned t = l ^ f;
        l = r; r = t;
      }
      x = ((uint64_t)l << h) | r;
    } while (x >= (uint64_t)n);
    out[i] = (int64_t)x;
  }
}
// The same bijection with its key formed on the device from state = {seed, step counter} and a slot number: a launch captured
// into the step's hipGraph draws another permutation at every replay (the by-value key of wcmc_random_permutation is frozen at
// capture).  key = splitmix64(seed + GOLDEN * (8 * counter + slot + 1)); wcmc_permutation_key is the same arithmetic on the host.
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t permutation_key(uint64_t seed, uint64_t counter, int slot) {
  return splitmix64(seed + 0x632be59bd9b4e019ull * (8ull * counter + (uint64_t)slot + 1ull));
}
__global__ __launch_bounds__(256) void random_permutation_dev_kernel(int64_t* __restrict__ out, int64_t n, int h,
                                                                    const uint64_t* __restrict__ state, int slot) {
  const uint64_t key = permutation_key(state[0], state[1], slot);
  const unsigned k0 = (unsigned)(key & 0xffffffffu), k1 = (unsigned)(key >> 32);
  const unsigned mask = (1u << h) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i;
    do {
      unsigned l = (unsigned)(x >> h) & mask, r = (unsigned)x & mask;
#pragma unroll
      for (int rd = 0; rd < 6; ++rd) {
        const unsigned f = perm_mix(r ^ (rd & 1 ? k1 : k0) ^ (0x9e3779b9u * (unsigned)(rd + 1))) & mask;
        const unsigned t = l ^ f;
        l = r; r = t;
      }
      x = ((uint64_t)l << h) | r;
    } while (x >= (uint64_t)n);
    out[i] = (int64_t)x;
  }
}
__global__ void step_counter_advance_kernel(uint64_t* state) { state[1] += 1ull; }
}  // namespace wcmc

static int perm_half_bits(int64_t n) {
  int bits = 1;
  while (((int64_t)1 << bits) < n) ++bits;
  return (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
}

extern "C" uint64_t wcmc_permutation_key(uint64_t seed, uint64_t counter, int slot) { return wcmc::permutation_key(seed, counter, slot); }

extern "C" int wcmc_random_permutation_dev(int64_t* out, int64_t n, const uint64_t* state, int slot, void* stream) {
  WCMC_REQUIRE(out && state && n > 0 && n < ((int64_t)1 << 62) && slot >= 0 && slot < 8, WCMC_ERR_BAD_ARG, "random_permutation_dev: bad argument");
  const int h = perm_half_bits(n);
  WCMC_REQUIRE(h <= 31, WCMC_ERR_BAD_ARG, "random_permutation_dev: n too large");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(wcmc::random_permutation_dev_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, out, n, h, state, slot);
  return wcmc::check_launch("random_permutation_dev");
}

extern "C" int wcmc_step_counter_advance(uint64_t* state, void* stream) {
  WCMC_REQUIRE(state, WCMC_ERR_BAD_ARG, "step_counter_advance: null state");
  hipLaunchKernelGGL(wcmc::step_counter_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  return wcmc::check_launch("step_counter_advance");
}

extern "C" int wcmc_random_permutation(int64_t* out, int64_t n, uint64_t seed, void* stream) {
  WCMC_REQUIRE(out && n > 0 && n < ((int64_t)1 << 62), WCMC_ERR_BAD_ARG, "random_permutation: bad argument");
  int bits = 1;
  while (((int64_t)1 << bits) < n) ++bits;
  const int h = (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
  WCMC_REQUIRE(h <= 31, WCMC_ERR_BAD_ARG, "random_permutation: n too large");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(wcmc::random_permutation_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, out, n, h, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32));
  return wcmc::check_launch("random_permutation");
}

