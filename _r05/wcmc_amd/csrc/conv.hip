// Convolution forward / data-gradient (one implicit-GEMM kernel) and weight-gradient
// for gfx950 on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Replaces torch.nn.Conv2d fwd/bwd (cuDNN, train_kpcn.py:349) inside
// sbmc.modules.ConvChain (support/networks.py:18-24, train_kpcn.py:213).
//
// GEMM view of the forward:  D[co][m] = sum_k Wp[co][k] * A[m][k]
//   m  = flat output pixel (n, oy, ox)               -- MFMA column (lane & 15)
//   co = output channel                              -- MFMA row
//   k  = tap*Kp + ci, tap = ky*ks+kx                 -- reduction, flattened across taps
//   A[m][k] = x[n, oy+ky-pad, ox+kx-pad, ci] (0 outside the image)
// With the couts on the MFMA row each lane ends up holding 4 CONSECUTIVE output
// channels of one pixel, i.e. one 16-byte NHWC store.
// The data gradient is the same kernel run on dy with the flipped/transposed packing
// (mode 1 of wcmc_conv2d_pack_weight), pad' = ks-1-pad, and the ReLU mask of the
// producing layer fused as an epilogue gate.
//
// Roofline: MFMA-bound.  fp32 MFMA does 256 FLOP/clk/CU; one 128-pixel x 112-cout block
// step (32 k) needs 30 KB of L2->LDS traffic per 0.92 MFLOP, ~18 GB/s per CU at full
// MFMA rate -- far below the L2 port, so plain register-staged double buffering is enough.
#include "common.h"
#include "conv_common.h"

namespace wcmc {

constexpr int BM = 128;   // pixels per block
constexpr int KC = 32;    // k per LDS stage
constexpr int LDK = 36;   // LDS row stride in floats (16-byte aligned rows, 2-way conflicts at worst)

struct IgemmParams {
  const float* x; int64_t xsn, xsh, xsw; int N, H, W, Cin;
  const float* wp; const float* bias;
  float* y; int64_t ysn, ysh, ysw; int Ho, Wo, Cout;
  const float* gate; int64_t gsn, gsh, gsw; int gate_act; float gate_slope;
  int ks, pad, act; float slope;
  int Kp, Kt, Np;
  int64_t M;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(IgemmParams p) {
  constexpr int BN = NT * 16;
  constexpr int NJ = (BN + 31) / 32;   // B rows per loader thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;        // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- loader mapping: 8 threads cover one 32-k row (8 x float4)
  const int kq = tid & 7, prow = tid >> 3;
  int64_t abase[4]; int aiy[4], aix[4];
  const int64_t HoWo = (int64_t)p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = m0 + prow + 32 * j;
    if (m < p.M) {
      const int n = (int)(m / HoWo);
      const int r = (int)(m - (int64_t)n * HoWo);
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      aiy[j] = oy - p.pad; aix[j] = ox - p.pad;
      abase[j] = (int64_t)n * p.xsn + (int64_t)aiy[j] * p.xsh + (int64_t)aix[j] * p.xsw;
    } else {
      aiy[j] = -(1 << 28); aix[j] = -(1 << 28); abase[j] = 0;   // every bounds test fails
    }
  }
  // running (tap, ci) of this thread's float4 inside the flattened k axis
  int ci = kq * 4, tap = 0, tdy = 0, tdx = 0;
  while (ci >= p.Kp) { ci -= p.Kp; ++tap; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
  const int ntaps = p.ks * p.ks;
  const int nchunks = p.Kt / KC;

  float4 ra[4], rb[NJ];
  auto load_chunk = [&](int c) {
    const bool tap_ok = tap < ntaps;
    const int64_t toff = (int64_t)tdy * p.xsh + (int64_t)tdx * p.xsw + ci;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = aiy[j] + tdy, ix = aix[j] + tdx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        v = *reinterpret_cast<const float4*>(p.x + abase[j] + toff);
        if (ci + 4 > p.Cin) {            // partial last vector of a pixel: kill pad channels
          if (ci + 1 >= p.Cin) v.y = 0.f;
          if (ci + 2 >= p.Cin) v.z = 0.f;
          if (ci + 3 >= p.Cin) v.w = 0.f;
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nrow = prow + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nrow < BN && n0 + nrow < p.Np)
        v = *reinterpret_cast<const float4*>(p.wp + (int64_t)(n0 + nrow) * p.Kt + (int64_t)c * KC + kq * 4);
      rb[j] = v;
    }
    // advance to the next chunk
    ci += KC;
    while (ci >= p.Kp) { ci -= p.Kp; ++tap; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BM * LDK;
    float* b = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(a + (prow + 32 * j) * LDK + kq * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nrow = prow + 32 * j;
      if (nrow < BN) *reinterpret_cast<float4*>(b + nrow * LDK + kq * 4) = rb[j];
    }
  };

  f32x4 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int frow = lane & 15, fk = (lane >> 4) * 4;
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    const float* a = As + buf * BM * LDK + (wave * 32 + frow) * LDK + fk;
    const float* b = Bs + buf * BN * LDK + frow * LDK + fk;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 a0 = *reinterpret_cast<const float4*>(a + g * 16);
      const float4 a1 = *reinterpret_cast<const float4*>(a + 16 * LDK + g * 16);
      float4 bw[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bw[j] = *reinterpret_cast<const float4*>(b + j * 16 * LDK + g * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].x, a0.x, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].x, a1.x, acc[j][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].y, a0.y, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].y, a1.y, acc[j][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].z, a0.z, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].z, a1.z, acc[j][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].w, a0.w, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j].w, a1.w, acc[j][1], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds couts n0 + j*16 + 4*(lane>>4) + {0..3} of pixel (lane&15)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + wave * 32 + i * 16 + frow;
    if (m >= p.M) continue;
    const int n = (int)(m / HoWo);
    const int r = (int)(m - (int64_t)n * HoWo);
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    float* yp = p.y + (int64_t)n * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw;
    const float* gp = p.gate ? p.gate + (int64_t)n * p.gsn + (int64_t)oy * p.gsh + (int64_t)ox * p.gsw : nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + j * 16 + fk;
      if (co >= p.Cout) continue;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      const bool full = co + 4 <= p.Cout;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < p.Cout) {
          if (p.bias) v[e] += p.bias[co + e];
          v[e] = act_apply(v[e], p.act, p.slope);
        } else {
          v[e] = 0.f;
        }
      }
      if (gp) {
        if (full) {
          const float4 g4 = *reinterpret_cast<const float4*>(gp + co);
          v[0] *= act_gate(g4.x, p.gate_act, p.gate_slope);
          v[1] *= act_gate(g4.y, p.gate_act, p.gate_slope);
          v[2] *= act_gate(g4.z, p.gate_act, p.gate_slope);
          v[3] *= act_gate(g4.w, p.gate_act, p.gate_slope);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) v[e] *= act_gate(gp[co + e], p.gate_act, p.gate_slope);
        }
      }
      // sw >= round_up(Cout,4): the pad lanes of the last vector are inside the pixel and get 0
      *reinterpret_cast<float4*>(yp + co) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ------------------------------------------------------------------ weight packing
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                   int ks, int mode, int rows, int Np, int Kp, int Kt) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Np * Kt) return;
  const int n = (int)(idx / Kt), k = (int)(idx - (int64_t)n * Kt);
  const int tap = k / Kp, c = k - tap * Kp;
  const int taps = ks * ks;
  const int kchan = mode == 0 ? Cin : Cout;
  float v = 0.f;
  if (n < rows && tap < taps && c < kchan) {
    if (mode == 0) v = w[((int64_t)n * Cin + c) * taps + tap];                 // n = co, c = ci
    else           v = w[((int64_t)c * Cin + n) * taps + (taps - 1 - tap)];    // n = ci, c = co, flipped
  }
  wp[idx] = v;
}

// ------------------------------------------------------------------ weight gradient
// D[co][ci] (per tap) = sum_pix dy[pix][co] * x[pix+tap][ci];  pix on the MFMA k axis.
// grid = (splits, taps, coBlocks*ciBlocks); the 4 waves of a block split each 32-pixel stage
// 8/8/8/8 and are summed through LDS in wave order, so a slab is bitwise reproducible.
struct WgradParams {
  const float* x; int64_t xsn, xsh, xsw; int N, H, W, Cin;
  const float* dy; int64_t dsn, dsh, dsw; int Ho, Wo, Cout;
  int ks, pad;
  float* slabs; int S; int64_t M, pix_per_split;
  int Np, Cq, coBlocks, ciBlocks;
};

struct PixCursor {   // flat pixel -> (n, oy, ox), advanced by a fixed stride per stage
  int n, oy, ox; int64_t p;
  __device__ __forceinline__ void init(int64_t p0, int Ho, int Wo) {
    p = p0;
    const int64_t hw = (int64_t)Ho * Wo;
    n = (int)(p0 / hw);
    const int r = (int)(p0 - (int64_t)n * hw);
    oy = r / Wo; ox = r - oy * Wo;
  }
  __device__ __forceinline__ void advance(int d, int Ho, int Wo) {
    p += d; ox += d;
    while (ox >= Wo) { ox -= Wo; if (++oy == Ho) { oy = 0; ++n; } }
  }
};

constexpr int wg_stride(int t) { return (t * 16) % 32 == 16 ? t * 16 : t * 16 + 16; }

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradParams p) {
  constexpr int PK = 32;
  constexpr int SA = wg_stride(TM), SB = wg_stride(TN);
  constexpr int NA = (PK * TM * 4 + 255) / 256;   // dY float4 per thread per stage
  constexpr int NB = (PK * TN * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ys = smem;                    // [2][PK][SA]
  float* Xs = smem + 2 * PK * SA;      // [2][PK][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, tap = blockIdx.y;
  const int cob = blockIdx.z / p.ciBlocks, cib = blockIdx.z - cob * p.ciBlocks;
  const int co0 = cob * TM * 16, ci0 = cib * TN * 16;
  const int tdy = tap / p.ks - p.pad, tdx = tap % p.ks - p.pad;
  const int tm_valid = min(TM, (p.Np - co0) / 16), tn_valid = min(TN, (p.Cq - ci0) / 16);

  const int64_t pstart = (int64_t)s * p.pix_per_split;
  const int64_t pend = min(p.M, pstart + p.pix_per_split);
  const int nstages = (int)((pend - pstart + PK - 1) / PK);

  PixCursor ca[NA], cb[NB];
  int a_px[NA], a_c[NA], b_px[NB], b_c[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int idx = tid + 256 * j;
    a_px[j] = idx / (TM * 4); a_c[j] = (idx - a_px[j] * (TM * 4)) * 4;
    ca[j].init(pstart + min(a_px[j], PK - 1), p.Ho, p.Wo);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int idx = tid + 256 * j;
    b_px[j] = idx / (TN * 4); b_c[j] = (idx - b_px[j] * (TN * 4)) * 4;
    cb[j].init(pstart + min(b_px[j], PK - 1), p.Ho, p.Wo);
  }

  float4 ra[NA], rb[NB];
  auto load_stage = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int co = co0 + a_c[j];
      if (a_px[j] < PK && ca[j].p < pend && co < p.Cout) {
        v = *reinterpret_cast<const float4*>(p.dy + (int64_t)ca[j].n * p.dsn + (int64_t)ca[j].oy * p.dsh +
                                             (int64_t)ca[j].ox * p.dsw + co);
        if (co + 4 > p.Cout) {
          if (co + 1 >= p.Cout) v.y = 0.f;
          if (co + 2 >= p.Cout) v.z = 0.f;
          if (co + 3 >= p.Cout) v.w = 0.f;
        }
      }
      ra[j] = v;
      ca[j].advance(PK, p.Ho, p.Wo);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ci = ci0 + b_c[j];
      const int iy = cb[j].oy + tdy, ix = cb[j].ox + tdx;
      if (b_px[j] < PK && cb[j].p < pend && ci < p.Cin && (unsigned)iy < (unsigned)p.H &&
          (unsigned)ix < (unsigned)p.W) {
        v = *reinterpret_cast<const float4*>(p.x + (int64_t)cb[j].n * p.xsn + (int64_t)iy * p.xsh +
                                             (int64_t)ix * p.xsw + ci);
        if (ci + 4 > p.Cin) {
          if (ci + 1 >= p.Cin) v.y = 0.f;
          if (ci + 2 >= p.Cin) v.z = 0.f;
          if (ci + 3 >= p.Cin) v.w = 0.f;
        }
      }
      rb[j] = v;
      cb[j].advance(PK, p.Ho, p.Wo);
    }
  };
  auto store_stage = [&](int buf) {
    float* a = Ys + buf * PK * SA;
    float* b = Xs + buf * PK * SB;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (a_px[j] < PK) *reinterpret_cast<float4*>(a + a_px[j] * SA + a_c[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (b_px[j] < PK) *reinterpret_cast<float4*>(b + b_px[j] * SB + b_c[j]) = rb[j];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nstages > 0) {
    load_stage();
    store_stage(0);
  }
  __syncthreads();
  const int fcol = lane & 15, fk = lane >> 4;
  for (int st = 0; st < nstages; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstages) load_stage();
    const float* a = Ys + buf * PK * SA + (wave * 8 + fk) * SA + fcol;
    const float* b = Xs + buf * PK * SB + (wave * 8 + fk) * SB + fcol;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = a[kk * 4 * SA + i * 16];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = b[kk * 4 * SB + j * 16];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i < tm_valid) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (j < tn_valid) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (st + 1 < nstages) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- ordered cross-wave sum in LDS, then one coalesced slab write
  constexpr int RS = TN * 16 + 4;
  float* red = smem;                   // [TM*16][RS]  (fits: TM*16*RS <= 2*PK*(SA+SB))
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = red + (i * 16 + fk * 4 + r) * RS + j * 16 + fcol;
            *q = (wv == 0 ? 0.f : *q) + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  const int taps = p.ks * p.ks;
  float* slab = p.slabs + ((int64_t)s * taps + tap) * p.Np * p.Cq;
  for (int idx = tid; idx < TM * 16 * TN * 4; idx += 256) {
    const int r = idx / (TN * 4), c = (idx - r * (TN * 4)) * 4;
    if (co0 + r < p.Np && ci0 + c < p.Cq)
      *reinterpret_cast<float4*>(slab + (int64_t)(co0 + r) * p.Cq + ci0 + c) =
          *reinterpret_cast<const float4*>(red + r * RS + c);
  }
}

// column sums of an NHWC view: stage 1 partial[g][c], stage 2 out[c]
__global__ void colsum_partial_kernel(const float* __restrict__ dy, int64_t dsn, int64_t dsh, int64_t dsw, int Ho,
                                      int Wo, int C, int64_t M, int64_t per_block, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = min(M, p0 + per_block);
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    float acc = 0.f;
    if (c < C) {
      PixCursor cur;
      if (p0 + pg < p1) {
        cur.init(p0 + pg, Ho, Wo);
        for (int64_t q = p0 + pg; q < p1; q += 4) {
          acc += dy[(int64_t)cur.n * dsn + (int64_t)cur.oy * dsh + (int64_t)cur.ox * dsw + c];
          cur.advance(4, Ho, Wo);
        }
      }
    }
    red[pg][cl] = acc;
    __syncthreads();
    if (pg == 0 && c < C) partial[(int64_t)blockIdx.x * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    __syncthreads();
  }
}
// Fast path for pixel-regular views (dsh == Wo*dsw, dsn == Ho*dsh: every buffer this library
// allocates, and channel slices of them): pixel p lives at p*dsw.  16-byte loads along channels,
// 256/C4 pixel lanes per block, LDS tree across the pixel lanes.
__global__ __launch_bounds__(256) void colsum_partial_flat_kernel(const float* __restrict__ dy, int64_t dsw, int C,
                                                                   int64_t M, int64_t per_block,
                                                                   float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* red = reinterpret_cast<float4*>(smem);
  const int C4 = (C + 3) / 4;
  const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = min(M, p0 + per_block);
  for (int cb = 0; cb < C4; cb += 256) {           // C4 > 256 never happens on this path, kept for safety
    const int cw = min(256, C4 - cb);
    const int PL = 256 / cw;
    const int c4 = threadIdx.x % cw, pl = threadIdx.x / cw;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < PL) {
      const int c = (cb + c4) * 4;
      for (int64_t p = p0 + pl; p < p1; p += PL) {
        float4 v = *reinterpret_cast<const float4*>(dy + p * dsw + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      red[pl * cw + c4] = acc;
    }
    __syncthreads();
    if (pl == 0) {
      for (int q = 1; q < PL; ++q) {
        const float4 v = red[q * cw + c4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const int c = (cb + c4) * 4;
      float* o = partial + (int64_t)blockIdx.x * C + c;
      o[0] = acc.x;
      if (c + 1 < C) o[1] = acc.y;
      if (c + 2 < C) o[2] = acc.z;
      if (c + 3 < C) o[3] = acc.w;
    }
    __syncthreads();
  }
}
__global__ void act_backward_kernel(const float* __restrict__ dy, int64_t dsn, int64_t dsh, int64_t dsw,
                                    const float* __restrict__ y, int64_t ysn, int64_t ysh, int64_t ysw,
                                    float* __restrict__ dx, int64_t xsn, int64_t xsh, int64_t xsw, int H, int W,
                                    int C4, int C, int64_t total, int act, float slope) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C4) * 4;
    int64_t t = idx / C4;
    const int x = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int n = (int)(t / H);
    const float4 g = *reinterpret_cast<const float4*>(dy + n * dsn + yy * dsh + x * dsw + c);
    const float4 v = *reinterpret_cast<const float4*>(y + n * ysn + yy * ysh + x * ysw + c);
    float4 o;
    o.x = g.x * act_gate(v.x, act, slope);
    o.y = c + 1 < C ? g.y * act_gate(v.y, act, slope) : 0.f;
    o.z = c + 2 < C ? g.z * act_gate(v.z, act, slope) : 0.f;
    o.w = c + 3 < C ? g.w * act_gate(v.w, act, slope) : 0.f;
    *reinterpret_cast<float4*>(dx + n * xsn + yy * xsh + x * xsw + c) = o;
  }
}

static int pick_nt(int tiles) {
  const int cand[4] = {7, 4, 2, 1};
  int best = 1, best_cost = 1 << 30;
  for (int i = 0; i < 4; ++i) {
    const int nt = cand[i];
    const int cost = ((tiles + nt - 1) / nt) * (nt + 2);
    if (cost < best_cost) { best_cost = cost; best = nt; }
  }
  return best;
}

struct WgradPlan { int TM, coBlocks, ciBlocks, S, Np, Cq, G; int64_t pix_per_split, per_block; size_t slab_elems, bytes; };
static WgradPlan plan_wgrad(int N, int Ho, int Wo, int Cout, int Cin, int ks) {
  WgradPlan pl;
  pl.Np = round_up(Cout, 16); pl.Cq = round_up(Cin, 16);
  const int coT = pl.Np / 16, ciT = pl.Cq / 16;
  pl.TM = (coT % 7 == 0) ? 7 : 4;
  pl.coBlocks = (coT + pl.TM - 1) / pl.TM;
  pl.ciBlocks = (ciT + 3) / 4;
  const int64_t M = (int64_t)N * Ho * Wo;
  const int taps = ks * ks;
  const int64_t tiles = (int64_t)taps * pl.coBlocks * pl.ciBlocks;
  // 2 blocks/CU x 256 CUs = 512 co-resident blocks; land just under a whole number of rounds (3)
  int64_t S = 1536 / tiles;
  const int64_t maxS = M / 256 > 0 ? M / 256 : 1;     // >= 8 stages of 32 pixels per block
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  pl.pix_per_split = ceil_div64(ceil_div64(M, S), 32) * 32;
  pl.S = (int)ceil_div64(M, pl.pix_per_split);
  pl.slab_elems = (size_t)pl.S * taps * pl.Np * pl.Cq;
  pl.G = (int)(M / 256 > 0 ? (M / 256 < 256 ? M / 256 : 256) : 1);
  pl.per_block = ceil_div64(M, pl.G);
  pl.G = (int)ceil_div64(M, pl.per_block);
  pl.bytes = (pl.slab_elems + (size_t)pl.G * Cout) * sizeof(float);
  return pl;
}

}  // namespace wcmc

using namespace wcmc;

extern "C" size_t wcmc_conv2d_packed_elems(int rows, int kchan, int ks) {
  if (rows <= 0 || kchan <= 0 || ks <= 0) return 0;
  return (size_t)round_up(rows, 16) * round_up(ks * ks * round_up(kchan, 4), 32);
}

extern "C" int wcmc_conv2d_pack_weight(const float* w, float* wp, int Cout, int Cin, int ks, int mode,
                                       void* stream) {
  WCMC_REQUIRE(w && wp && Cout > 0 && Cin > 0 && ks > 0 && (mode == 0 || mode == 1), WCMC_ERR_BAD_ARG,
               "conv2d_pack_weight: bad argument (Cout=%d Cin=%d ks=%d mode=%d)", Cout, Cin, ks, mode);
  const int rows = mode == 0 ? Cout : Cin, kchan = mode == 0 ? Cin : Cout;
  const int Np = round_up(rows, 16), Kp = round_up(kchan, 4), Kt = round_up(ks * ks * Kp, 32);
  const int64_t total = (int64_t)Np * Kt;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     w, wp, Cout, Cin, ks, mode, rows, Np, Kp, Kt);
  return check_launch("conv2d_pack_weight");
}

template <int NT>
static int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)2 * (BM + NT * 16) * LDK * sizeof(float);
  static LdsAttr attr_set;
  if (set_max_lds(reinterpret_cast<const void*>(&conv_igemm_kernel<NT>), (size_t)lds, attr_set) != hipSuccess) return WCMC_ERR_LAUNCH;
  const dim3 grid((unsigned)ceil_div64(p.M, BM), (unsigned)((p.Np / 16 + NT - 1) / NT));
  hipLaunchKernelGGL(conv_igemm_kernel<NT>, grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_igemm");
}

extern "C" int wcmc_conv2d_igemm(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, int N, int H, int W,
                                 int Cin, const float* wp, const float* bias, float* y, int64_t ysn, int64_t ysh,
                                 int64_t ysw, int Cout, int ks, int pad, int act, float slope, const float* gate,
                                 int64_t gsn, int64_t gsh, int64_t gsw, int gate_act, float gate_slope,
                                 void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks > 0 && pad >= 0 && wp, WCMC_ERR_BAD_ARG,
               "conv2d_igemm: bad argument (N=%d H=%d W=%d Cin=%d Cout=%d ks=%d pad=%d)", N, H, W, Cin, Cout, ks,
               pad);
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_igemm: empty output (%dx%d)", Ho, Wo);
  WCMC_REQUIRE(nhwc_view_ok(x, xsn, xsh, xsw, Cin), WCMC_ERR_ALIGNMENT, "conv2d_igemm: x violates the NHWC-view contract");
  WCMC_REQUIRE(nhwc_view_ok(y, ysn, ysh, ysw, Cout), WCMC_ERR_ALIGNMENT, "conv2d_igemm: y violates the NHWC-view contract");
  WCMC_REQUIRE(aligned16(wp), WCMC_ERR_ALIGNMENT, "conv2d_igemm: wp not 16-byte aligned");
  WCMC_REQUIRE(!gate || nhwc_view_ok(gate, gsn, gsh, gsw, Cout), WCMC_ERR_ALIGNMENT,
               "conv2d_igemm: gate violates the NHWC-view contract");
  IgemmParams p;
  p.x = x; p.xsn = xsn; p.xsh = xsh; p.xsw = xsw; p.N = N; p.H = H; p.W = W; p.Cin = Cin;
  p.wp = wp; p.bias = bias;
  p.y = y; p.ysn = ysn; p.ysh = ysh; p.ysw = ysw; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.gate = gate; p.gsn = gsn; p.gsh = gsh; p.gsw = gsw; p.gate_act = gate_act; p.gate_slope = gate_slope;
  p.ks = ks; p.pad = pad; p.act = act; p.slope = slope;
  p.Kp = round_up(Cin, 4); p.Kt = round_up(ks * ks * p.Kp, 32); p.Np = round_up(Cout, 16);
  p.M = (int64_t)N * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  switch (pick_nt(p.Np / 16)) {
    case 7: return launch_igemm<7>(p, st);
    case 4: return launch_igemm<4>(p, st);
    case 2: return launch_igemm<2>(p, st);
    default: return launch_igemm<1>(p, st);
  }
}

extern "C" size_t wcmc_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int ks) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin <= 0 || ks <= 0) return 0;
  return plan_wgrad(N, Ho, Wo, Cout, Cin, ks).bytes;
}

template <int TM>
static int launch_wgrad(const WgradParams& p, hipStream_t stream) {
  constexpr int TN = 4;
  constexpr int SA = wg_stride(TM), SB = wg_stride(TN);
  constexpr size_t lds_stage = (size_t)2 * 32 * (SA + SB) * sizeof(float);
  constexpr size_t lds_red = (size_t)TM * 16 * (TN * 16 + 4) * sizeof(float);
  constexpr size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
  const dim3 grid((unsigned)p.S, (unsigned)(p.ks * p.ks), (unsigned)(p.coBlocks * p.ciBlocks));
  hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN>), grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_wgrad");
}

extern "C" int wcmc_conv2d_wgrad(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, int N, int H, int W,
                                 int Cin, const float* dy, int64_t dsn, int64_t dsh, int64_t dsw, int Cout, int ks,
                                 int pad, float* dw, float* db, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks > 0 && pad >= 0 && dw && workspace,
               WCMC_ERR_BAD_ARG, "conv2d_wgrad: bad argument");
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_wgrad: empty output");
  WCMC_REQUIRE(nhwc_view_ok(x, xsn, xsh, xsw, Cin), WCMC_ERR_ALIGNMENT, "conv2d_wgrad: x violates the NHWC-view contract");
  WCMC_REQUIRE(nhwc_view_ok(dy, dsn, dsh, dsw, Cout), WCMC_ERR_ALIGNMENT, "conv2d_wgrad: dy violates the NHWC-view contract");
  const WgradPlan pl = plan_wgrad(N, Ho, Wo, Cout, Cin, ks);
  WCMC_REQUIRE(workspace_bytes >= pl.bytes && aligned16(workspace), WCMC_ERR_WORKSPACE,
               "conv2d_wgrad: workspace %zu < %zu bytes (or unaligned)", workspace_bytes, pl.bytes);
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  p.x = x; p.xsn = xsn; p.xsh = xsh; p.xsw = xsw; p.N = N; p.H = H; p.W = W; p.Cin = Cin;
  p.dy = dy; p.dsn = dsn; p.dsh = dsh; p.dsw = dsw; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.ks = ks; p.pad = pad; p.slabs = (float*)workspace; p.S = pl.S; p.M = (int64_t)N * Ho * Wo;
  p.pix_per_split = pl.pix_per_split; p.Np = pl.Np; p.Cq = pl.Cq; p.coBlocks = pl.coBlocks; p.ciBlocks = pl.ciBlocks;
  int rc = pl.TM == 7 ? launch_wgrad<7>(p, st) : launch_wgrad<4>(p, st);
  if (rc) return rc;
  const int64_t total = (int64_t)ks * ks * Cout * Cin;
  (void)total;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cin + WR_CI - 1) / WR_CI), (unsigned)Cout), dim3(256),
                     (size_t)WR_CI * (ks * ks + 1) * sizeof(float) + 256 * sizeof(float), st, p.slabs, dw, pl.S, ks * ks, Cout, Cin, pl.Np, pl.Cq, nullptr, 0, 0, nullptr);
  rc = check_launch("conv2d_wgrad_reduce");
  if (rc || !db) return rc;
  float* partial = (float*)workspace + pl.slab_elems;
  if (dsh == (int64_t)Wo * dsw && dsn == (int64_t)Ho * dsh && (Cout + 3) / 4 <= 256)
    hipLaunchKernelGGL(colsum_partial_flat_kernel, dim3((unsigned)pl.G), dim3(256), 256 * sizeof(float4), st, dy, dsw,
                       Cout, p.M, pl.per_block, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)pl.G), dim3(256), 0, st, dy, dsn, dsh, dsw, Ho, Wo, Cout,
                       p.M, pl.per_block, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(1024), 0, st, partial, pl.G, Cout,
                     db);
  return check_launch("conv2d_bias_grad");
}

extern "C" int wcmc_act_backward(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw, const float* y,
                                 int64_t ysn, int64_t ysh, int64_t ysw, float* dx, int64_t xsn, int64_t xsh,
                                 int64_t xsw, int N, int H, int W, int C, int act, float slope, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, WCMC_ERR_BAD_ARG, "act_backward: bad shape");
  WCMC_REQUIRE(nhwc_view_ok(dy, dsn, dsh, dsw, C) && nhwc_view_ok(y, ysn, ysh, ysw, C) &&
                   nhwc_view_ok(dx, xsn, xsh, xsw, C),
               WCMC_ERR_ALIGNMENT, "act_backward: a view violates the NHWC-view contract");
  const int C4 = (C + 3) / 4;
  const int64_t total = (int64_t)N * H * W * C4;
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
  hipLaunchKernelGGL(act_backward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dsn, dsh, dsw, y, ysn,
                     ysh, ysw, dx, xsn, xsh, xsw, H, W, C4, C, total, act, slope);
  return check_launch("act_backward");
}
