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
__global__ void split_kernel(const float* __restrict__ x, int64_t xsn, int64_t xsh, int64_t xsw,
                             u16* __restrict__ out, int H, int W, int C, int Cp, int64_t total) {
  const int V = Cp / 8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(idx % V); int64_t t = idx / V;
    const int xx = (int)(t % W); t /= W;
    const int y = (int)(t % H); const int n = (int)(t / H);
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
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1(f[e], hi[e], lo[e]);
    u16* o = out + (((int64_t)n * H + y) * W + xx) * 2 * Cp + c0;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
    *reinterpret_cast<uint4*>(o + Cp) = *reinterpret_cast<const uint4*>(lo);
  }
}

__global__ void pack_weight_split_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int Cin,
                                         int ks, int mode, int rows, int Np, int Kp, int Kt) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Np * Kt) return;
  const int n = (int)(idx / Kt), k = (int)(idx - (int64_t)n * Kt);
  const int tap = k / Kp, c = k - tap * Kp;
  const int taps = ks * ks;
  const int kchan = mode == 0 ? Cin : Cout;
  float v = 0.f;
  if (n < rows && tap < taps && c < kchan) {
    if (mode == 0) v = w[((int64_t)n * Cin + c) * taps + tap];
    else           v = w[((int64_t)c * Cin + n) * taps + (taps - 1 - tap)];
  }
  u16 hi, lo;
  split1(v, hi, lo);
  wp[((int64_t)n * 2) * Kt + k] = hi;
  wp[((int64_t)n * 2 + 1) * Kt + k] = lo;
}

// ------------------------------------------------------------------ implicit GEMM (fwd + dgrad)
constexpr int XBM = 128;   // pixels per block
constexpr int XKC = 32;    // k per LDS stage = one MFMA k-step
constexpr int XLD = 40;    // LDS row stride in bf16 (80 B: 16-byte aligned, spreads the b128 reads)

struct XIgemmParams {
  const u16* x; int N, H, W, Cin, Cpi;
  const u16* wp; const float* bias;
  float* yf; int64_t ysn, ysh, ysw;       // fp32 NHWC view output (or null)
  u16* ys; int Cpo;                       // split dense output (or null)
  int Ho, Wo, Cout;
  const u16* gate; int gate_act; float gate_slope;   // split dense, geometry of y
  int ks, pad, act; float slope;
  int Kp, Kt, Np;
  int64_t M;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16x3_kernel(XIgemmParams p) {
  constexpr int BN = NT * 16;
  constexpr int NJ = (BN + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  // per buffer: A[2 planes][XBM][XLD], B[2 planes][BN][XLD]
  constexpr int A_ELEMS = 2 * XBM * XLD, B_ELEMS = 2 * BN * XLD, BUF = A_ELEMS + B_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private 4 MB L2 each), so
  // give each XCD one contiguous run of pixel tiles -- vertically adjacent tiles then share their 5x5
  // halo rows through the same L2 instead of each XCD streaming the whole image (speed only).
  int tile;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int64_t m0 = (int64_t)tile * XBM;
  const int n0 = blockIdx.y * BN;

  // loader mapping: 8 consecutive threads = one row's 2 planes x 4 vectors of 8 bf16
  const int vq = tid & 3, pl = (tid >> 2) & 1, prow = tid >> 3;
  int64_t abase[4]; int aiy[4], aix[4];
  const int64_t HoWo = (int64_t)p.Ho * p.Wo;
  const int64_t pixs = 2 * p.Cpi;                       // u16 per input pixel
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = m0 + prow + 32 * j;
    if (m < p.M) {
      const int n = (int)(m / HoWo);
      const int r = (int)(m - (int64_t)n * HoWo);
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      aiy[j] = oy - p.pad; aix[j] = ox - p.pad;
      abase[j] = (((int64_t)n * p.H + aiy[j]) * p.W + aix[j]) * pixs + (int64_t)pl * p.Cpi;
    } else {
      aiy[j] = -(1 << 28); aix[j] = -(1 << 28); abase[j] = 0;
    }
  }
  int ci = vq * 8, tap = 0, tdy = 0, tdx = 0;
  while (ci >= p.Kp) { ci -= p.Kp; ++tap; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
  const int ntaps = p.ks * p.ks;
  const int nchunks = p.Kt / XKC;

  uint4 ra[4], rb[NJ];
  auto load_chunk = [&](int c) {
    const bool tap_ok = tap < ntaps;
    const int64_t toff = ((int64_t)tdy * p.W + tdx) * pixs + ci;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = aiy[j] + tdy, ix = aix[j] + tdx;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const uint4*>(p.x + abase[j] + toff);
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nrow = prow + 32 * j;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (nrow < BN && n0 + nrow < p.Np)
        v = *reinterpret_cast<const uint4*>(p.wp + ((int64_t)(n0 + nrow) * 2 + pl) * p.Kt + (int64_t)c * XKC + vq * 8);
      rb[j] = v;
    }
    ci += XKC;
    while (ci >= p.Kp) { ci -= p.Kp; ++tap; if (++tdx == p.ks) { tdx = 0; ++tdy; } }
  };
  auto store_chunk = [&](int buf) {
    u16* a = smem16 + buf * BUF + pl * (XBM * XLD);
    u16* b = smem16 + buf * BUF + A_ELEMS + pl * (BN * XLD);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4*>(a + (prow + 32 * j) * XLD + vq * 8) = ra[j];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nrow = prow + 32 * j;
      if (nrow < BN) *reinterpret_cast<uint4*>(b + nrow * XLD + vq * 8) = rb[j];
    }
  };

  f32x4 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int frow = lane & 15, fk = (lane >> 4) * 8;     // MFMA 16x16x32: lane holds k = 8*(lane>>4) .. +7
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    const u16* a = smem16 + buf * BUF + (wave * 32 + frow) * XLD + fk;
    const u16* b = smem16 + buf * BUF + A_ELEMS + frow * XLD + fk;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(a + i * 16 * XLD);
      al[i] = *reinterpret_cast<const bf16x8*>(a + XBM * XLD + i * 16 * XLD);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(b + j * 16 * XLD);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(b + BN * XLD + j * 16 * XLD);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[i], acc[j][i], 0, 0, 0);   // small terms first
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[i], acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[i], acc[j][i], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds couts n0 + j*16 + 4*(lane>>4) + {0..3} of pixel (lane&15)
  const int fq = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + wave * 32 + i * 16 + frow;
    if (m >= p.M) continue;
    const int n = (int)(m / HoWo);
    const int r = (int)(m - (int64_t)n * HoWo);
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    float* yp = p.yf ? p.yf + (int64_t)n * p.ysn + (int64_t)oy * p.ysh + (int64_t)ox * p.ysw : nullptr;
    u16* sp = p.ys ? p.ys + (int64_t)m * 2 * p.Cpo : nullptr;
    const u16* gp = p.gate ? p.gate + (int64_t)m * 2 * p.Cpo : nullptr;   // gate shares y's geometry (hi plane)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + j * 16 + fq;
      if (co >= p.Cpo) continue;                       // Cpo = round_up(Cout, 8) (fp32 output: round_up(Cout, 4))
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
      if (gp) {
        const uint2 g2 = *reinterpret_cast<const uint2*>(gp + co);
        const u16 g[4] = {(u16)(g2.x & 0xffff), (u16)(g2.x >> 16), (u16)(g2.y & 0xffff), (u16)(g2.y >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= act_gate(bf2f(g[e]), p.gate_act, p.gate_slope);
      }
      if (yp) *reinterpret_cast<float4*>(yp + co) = make_float4(v[0], v[1], v[2], v[3]);
      if (sp) {
        u16 hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split1(v[e], hi[e], lo[e]);
        *reinterpret_cast<uint2*>(sp + co) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
        *reinterpret_cast<uint2*>(sp + p.Cpo + co) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
      }
    }
  }
}

// ------------------------------------------------------------------ weight gradient
// D[co][ci] (per tap) = sum_pix dy[pix][co] * x[pix+tap][ci]; both operands are read with the
// transposing LDS load (ds_read_b64_tr_b16): the tiles sit in LDS as [pixel][channel] exactly as
// they come from HBM, and a lane receives 4 consecutive PIXELS (= MFMA k) of its channel column.
// Block = 64-pixel stage x (TM*16 couts) x 64 cins; waves: 2 (pixel halves = MFMA k-steps) x 2 (cin halves).
struct XWgradParams {
  const u16* x; int N, H, W, Cin, Cpi;
  const u16* dy; int Ho, Wo, Cout, Cpo;
  int ks, pad;
  float* slabs; int S; int64_t M, pix_per_split;
  int Np, Cq, coBlocks, ciBlocks;
};

template <int TM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16x3_kernel(XWgradParams p) {
  constexpr int PK = 64;
  constexpr int YC = TM * 16, XC = 64;
  constexpr int SA = YC + 8, SB = XC + 8;          // bf16 row strides (16-byte aligned, odd multiple of 16 B)
  constexpr int YV = YC / 8, XV = XC / 8;          // 16-byte vectors per plane per pixel
  constexpr int NV = (2 * YV + 2 * XV) / 4;        // vectors per thread (4 threads share a pixel)
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];
  u16* Ys = smem16;                        // [2][PK][SA]
  u16* Xs = smem16 + 2 * PK * SA;          // [2][PK][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, tap = blockIdx.y;
  const int cob = blockIdx.z / p.ciBlocks, cib = blockIdx.z - cob * p.ciBlocks;
  const int co0 = cob * YC, ci0 = cib * XC;
  const int tdy = tap / p.ks - p.pad, tdx = tap % p.ks - p.pad;
  const int tm_valid = min(TM, (p.Np - co0) / 16);
  const int wk = wave >> 1, wn = wave & 1;         // k-step (pixel half) and cin half of this wave
  const int tn_valid = min(2, max(0, (p.Cq - ci0) / 16 - wn * 2));

  const int64_t pstart = (int64_t)s * p.pix_per_split;
  const int64_t pend = min(p.M, pstart + p.pix_per_split);
  const int nstages = (int)((pend - pstart + PK - 1) / PK);

  // loader: thread -> pixel tid/4 of the stage, vectors (tid&3) + 4*j
  const int lpx = tid >> 2, lv0 = tid & 3;
  int cn, coy, cox; int64_t cp = pstart + lpx;
  {
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    cn = (int)(cp / hw);
    const int r = (int)(cp - (int64_t)cn * hw);
    coy = r / p.Wo; cox = r - coy * p.Wo;
  }
  uint4 rv[NV];
  auto load_stage = [&]() {
    const bool pv = cp < pend;
    const int iy = coy + tdy, ix = cox + tdx;
    const bool xv = pv && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const u16* yb = p.dy + (((int64_t)cn * p.Ho + coy) * p.Wo + cox) * 2 * p.Cpo;
    const u16* xb = p.x + (((int64_t)cn * p.H + iy) * p.W + ix) * 2 * p.Cpi;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = lv0 + 4 * j;
      uint4 val = make_uint4(0u, 0u, 0u, 0u);
      if (v < 2 * YV) {
        const int plane = v >= YV, vec = v - plane * YV;
        const int co = co0 + vec * 8;
        if (pv && co < p.Cpo) val = *reinterpret_cast<const uint4*>(yb + plane * p.Cpo + co);
      } else {
        const int u = v - 2 * YV;
        const int plane = u >= XV, vec = u - plane * XV;
        const int ci = ci0 + vec * 8;
        if (xv && ci < p.Cpi) val = *reinterpret_cast<const uint4*>(xb + plane * p.Cpi + ci);
      }
      rv[j] = val;
    }
    cp += PK; cox += PK;
    while (cox >= p.Wo) { cox -= p.Wo; if (++coy == p.Ho) { coy = 0; ++cn; } }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = lv0 + 4 * j;
      if (v < 2 * YV) {
        const int plane = v >= YV, vec = v - plane * YV;
        *reinterpret_cast<uint4*>(Ys + (plane * PK + lpx) * SA + vec * 8) = rv[j];
      } else {
        const int u = v - 2 * YV;
        const int plane = u >= XV, vec = u - plane * XV;
        *reinterpret_cast<uint4*>(Xs + (plane * PK + lpx) * SB + vec * 8) = rv[j];
      }
    }
  };

  f32x4 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // transposing read: lane (group g = lane>>4, i = lane&15, q = i>>2, pp = i&3) addresses row
  // (pixel) 8g + q (+4 for the second half) and columns 4pp..4pp+3 of the 16-channel tile; it
  // receives column i of those 4 rows.
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int prow0 = wk * 32 + 8 * g + tq;
  auto tr_read = [&](const u16* base, int stride, int col0, bf16x8& out) {
    const u16* a0 = base + prow0 * stride + col0 + 4 * tp;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(a0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(a0 + 4 * stride));
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
    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      tr_read(Xs, SB, (wn * 2 + j) * 16, xh[j]);
      tr_read(Xs + PK * SB, SB, (wn * 2 + j) * 16, xl[j]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i < tm_valid) {
        bf16x8 yh, yl;
        tr_read(Ys, SA, i * 16, yh);
        tr_read(Ys + PK * SA, SA, i * 16, yl);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j < tn_valid) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl, xh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xh[j], acc[i][j], 0, 0, 0);
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
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = red + (i * 16 + fq + r) * RS + (wn * 2 + j) * 16 + fcol;
            *q = (h == 0 ? 0.f : *q) + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  const int taps = p.ks * p.ks;
  float* slab = p.slabs + ((int64_t)s * taps + tap) * p.Np * p.Cq;
  for (int idx = tid; idx < YC * (XC / 4); idx += 256) {
    const int r = idx / (XC / 4), c = (idx - r * (XC / 4)) * 4;
    if (co0 + r < p.Np && ci0 + c < p.Cq)
      *reinterpret_cast<float4*>(slab + (int64_t)(co0 + r) * p.Cq + ci0 + c) =
          *reinterpret_cast<const float4*>(red + r * RS + c);
  }
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

struct XWgradPlan { int TM, coBlocks, ciBlocks, S, Np, Cq, G; int64_t pix_per_split, per_block; size_t slab_elems, bytes; };
static XWgradPlan x_plan_wgrad(int N, int Ho, int Wo, int Cout, int Cin, int ks) {
  XWgradPlan pl;
  pl.Np = round_up(Cout, 16); pl.Cq = round_up(Cin, 16);
  const int coT = pl.Np / 16, ciT = pl.Cq / 16;
  pl.TM = (coT % 7 == 0) ? 7 : 4;
  pl.coBlocks = (coT + pl.TM - 1) / pl.TM;
  pl.ciBlocks = (ciT + 3) / 4;
  const int64_t M = (int64_t)N * Ho * Wo;
  const int taps = ks * ks;
  const int64_t tiles = (int64_t)taps * pl.coBlocks * pl.ciBlocks;
  int64_t S = 1536 / tiles;
  const int64_t maxS = M / 512 > 0 ? M / 512 : 1;     // >= 8 stages of 64 pixels per block
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  pl.pix_per_split = ceil_div64(ceil_div64(M, S), 64) * 64;
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
                     x, xsn, xsh, xsw, (u16*)out, H, W, C, Cp, total);
  return check_launch("split_bf16");
}

extern "C" size_t wcmc_conv2d_packed_elems_bf16x3(int rows, int kchan, int ks) {
  if (rows <= 0 || kchan <= 0 || ks <= 0) return 0;
  return (size_t)round_up(rows, 16) * 2 * round_up(ks * ks * round_up(kchan, 8), 32);
}

extern "C" int wcmc_conv2d_pack_weight_bf16x3(const float* w, void* wp, int Cout, int Cin, int ks, int mode,
                                              void* stream) {
  WCMC_REQUIRE(w && wp && Cout > 0 && Cin > 0 && ks > 0 && (mode == 0 || mode == 1), WCMC_ERR_BAD_ARG,
               "conv2d_pack_weight_bf16x3: bad argument");
  const int rows = mode == 0 ? Cout : Cin, kchan = mode == 0 ? Cin : Cout;
  const int Np = round_up(rows, 16), Kp = round_up(kchan, 8), Kt = round_up(ks * ks * Kp, 32);
  const int64_t total = (int64_t)Np * Kt;
  hipLaunchKernelGGL(pack_weight_split_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, (u16*)wp, Cout, Cin, ks, mode, rows, Np, Kp, Kt);
  return check_launch("conv2d_pack_weight_bf16x3");
}

template <int NT>
static int launch_xigemm(const XIgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)2 * 2 * (XBM + NT * 16) * XLD * sizeof(u16);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3_kernel<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((unsigned)ceil_div64(p.M, XBM), (unsigned)((p.Np / 16 + NT - 1) / NT));
  hipLaunchKernelGGL(conv_igemm_bf16x3_kernel<NT>, grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_igemm_bf16x3");
}

extern "C" int wcmc_conv2d_igemm_bf16x3(const void* x_split, int N, int H, int W, int Cin, const void* wp,
                                        const float* bias, float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                                        void* y_split, int Cout, int ks, int pad, int act, float slope,
                                        const void* gate_split, int gate_act, float gate_slope, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks > 0 && pad >= 0 && x_split && wp,
               WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: bad argument");
  WCMC_REQUIRE((y != nullptr) != (y_split != nullptr), WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: exactly one of y (fp32 view) and y_split must be given");
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_igemm_bf16x3: empty output");
  WCMC_REQUIRE(aligned16(x_split) && aligned16(wp) && (!y_split || aligned16(y_split)) &&
                   (!gate_split || aligned16(gate_split)),
               WCMC_ERR_ALIGNMENT, "conv2d_igemm_bf16x3: split buffers must be 16-byte aligned");
  WCMC_REQUIRE(!y || nhwc_view_ok(y, ysn, ysh, ysw, Cout), WCMC_ERR_ALIGNMENT,
               "conv2d_igemm_bf16x3: y violates the NHWC-view contract");
  WCMC_REQUIRE(!gate_split || y_split, WCMC_ERR_BAD_ARG,
               "conv2d_igemm_bf16x3: a gate requires the split output geometry");
  XIgemmParams p;
  p.x = (const u16*)x_split; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cpi = round_up(Cin, 8);
  p.wp = (const u16*)wp; p.bias = bias;
  p.yf = y; p.ysn = ysn; p.ysh = ysh; p.ysw = ysw;
  p.ys = (u16*)y_split; p.Cpo = y_split ? round_up(Cout, 8) : round_up(Cout, 4);
  p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.gate = (const u16*)gate_split; p.gate_act = gate_act; p.gate_slope = gate_slope;
  p.ks = ks; p.pad = pad; p.act = act; p.slope = slope;
  p.Kp = p.Cpi; p.Kt = round_up(ks * ks * p.Kp, 32); p.Np = round_up(Cout, 16);
  p.M = (int64_t)N * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  switch (x_pick_nt(p.Np / 16)) {
    case 7: return launch_xigemm<7>(p, st);
    case 4: return launch_xigemm<4>(p, st);
    case 2: return launch_xigemm<2>(p, st);
    default: return launch_xigemm<1>(p, st);
  }
}

extern "C" size_t wcmc_conv2d_wgrad_bf16x3_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int ks) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin <= 0 || ks <= 0) return 0;
  return x_plan_wgrad(N, Ho, Wo, Cout, Cin, ks).bytes;
}

template <int TM>
static int launch_xwgrad(const XWgradParams& p, hipStream_t stream) {
  constexpr size_t lds_stage = (size_t)2 * 64 * ((TM * 16 + 8) + (64 + 8)) * sizeof(u16);
  constexpr size_t lds_red = (size_t)TM * 16 * (64 + 4) * sizeof(float);
  constexpr size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
  const dim3 grid((unsigned)p.S, (unsigned)(p.ks * p.ks), (unsigned)(p.coBlocks * p.ciBlocks));
  hipLaunchKernelGGL(conv_wgrad_bf16x3_kernel<TM>, grid, dim3(256), lds, stream, p);
  return check_launch("conv2d_wgrad_bf16x3");
}

extern "C" int wcmc_conv2d_wgrad_bf16x3(const void* x_split, int N, int H, int W, int Cin, const void* dy_split,
                                        int Cout, int ks, int pad, float* dw, float* db, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  WCMC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks > 0 && pad >= 0 && dw && workspace &&
                   x_split && dy_split,
               WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: bad argument");
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  WCMC_REQUIRE(Ho > 0 && Wo > 0, WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: empty output");
  WCMC_REQUIRE(aligned16(x_split) && aligned16(dy_split), WCMC_ERR_ALIGNMENT,
               "conv2d_wgrad_bf16x3: split buffers must be 16-byte aligned");
  const XWgradPlan pl = x_plan_wgrad(N, Ho, Wo, Cout, Cin, ks);
  WCMC_REQUIRE(workspace_bytes >= pl.bytes && aligned16(workspace), WCMC_ERR_WORKSPACE,
               "conv2d_wgrad_bf16x3: workspace %zu < %zu bytes (or unaligned)", workspace_bytes, pl.bytes);
  hipStream_t st = (hipStream_t)stream;
  XWgradParams p;
  p.x = (const u16*)x_split; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cpi = round_up(Cin, 8);
  p.dy = (const u16*)dy_split; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.Cpo = round_up(Cout, 8);
  p.ks = ks; p.pad = pad; p.slabs = (float*)workspace; p.S = pl.S; p.M = (int64_t)N * Ho * Wo;
  p.pix_per_split = pl.pix_per_split; p.Np = pl.Np; p.Cq = pl.Cq; p.coBlocks = pl.coBlocks; p.ciBlocks = pl.ciBlocks;
  int rc = pl.TM == 7 ? launch_xwgrad<7>(p, st) : launch_xwgrad<4>(p, st);
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cin + WR_CI - 1) / WR_CI), (unsigned)Cout), dim3(256),
                     (size_t)WR_CI * (ks * ks + 1) * sizeof(float), st, p.slabs, dw, pl.S, ks * ks, Cout, Cin, pl.Np, pl.Cq);
  rc = check_launch("conv2d_wgrad_bf16x3_reduce");
  if (rc || !db) return rc;
  float* partial = (float*)workspace + pl.slab_elems;
  WCMC_REQUIRE(p.Cpo / 8 <= 256, WCMC_ERR_BAD_ARG, "conv2d_wgrad_bf16x3: Cout > 2048 unsupported");
  hipLaunchKernelGGL(colsum_split_kernel, dim3((unsigned)pl.G), dim3(256), (size_t)256 * 8 * sizeof(float), st, p.dy,
                     p.Cpo, Cout, p.M, pl.per_block, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(256), 0, st, partial, pl.G, Cout,
                     db);
  return check_launch("conv2d_bias_grad_bf16x3");
}
