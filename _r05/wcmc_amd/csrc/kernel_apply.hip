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
#include <stdlib.h>

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
  float* ddata;                                           // bwd only, may be null
  int N, C, h, w, k, r, taps, nvec, halo;
};

// Reductions over the 16-lane group that owns a pixel: a DPP row is 16 lanes, and four row
// rotations (8, 4, 2, 1) leave the full sum / max in every lane -- one VALU op per step, no LDS.
// DPP-fused: `v_max_f32_dpp v, v, v row_ror:n` is max(v, ror_n(v)) in ONE instruction (the compiler's own lowering of
// update_dpp + fmaxf is v_mov_dpp + canonicalising max + max + mov).  Hazard: a VALU write of a VGPR needs 2 wait states
// before a DPP read of it; inline asm is invisible to the compiler's hazard pass, so the s_nop is in the asm.
__device__ __forceinline__ float group16_max(float v) {
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  return v;
}
// four sums at once: the four independent chains interleave, so the three other instructions between a write of a
// register and the DPP read of it are the wait states (one s_nop for the producers of the inputs)
__device__ __forceinline__ void group16_sum4(float& a, float& b, float& c, float& d) {
#define WCMC_DPP4(N)                                                                                   \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                             \
  "v_add_f32_dpp %1, %1, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                             \
  "v_add_f32_dpp %2, %2, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                             \
  "v_add_f32_dpp %3, %3, %3 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t" WCMC_DPP4(8) WCMC_DPP4(4) WCMC_DPP4(2) WCMC_DPP4(1) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef WCMC_DPP4
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
      m = group16_max(m) * LOG2E;                       // softmax in base 2: exp2(l * log2(e) - m), one fma per tap
      float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) {
        const float* l = reinterpret_cast<const float*>(&lv[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = __builtin_amdgcn_exp2f(fmaf(l[e], LOG2E, -m));
          const float4 d = *reinterpret_cast<const float4*>(hbase + toff[i][e]);
          if (C4) {
            s += ex; a0 = fmaf(ex, d.x, a0); a1 = fmaf(ex, d.y, a1); a2 = fmaf(ex, d.z, a2); a3 = fmaf(ex, d.w, a3);
          } else {                                  // d = {r, g, b, 1}: two packed FMAs per tap
            a0 = fmaf(ex, d.x, a0); a1 = fmaf(ex, d.y, a1); a2 = fmaf(ex, d.z, a2); s = fmaf(ex, d.w, s);
          }
        }
      }
      group16_sum4(s, a0, a1, a2);
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
          const float wt = __builtin_amdgcn_exp2f(fmaf(l[e], LOG2E, -lb));
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
        if (valid && vi < nvec) {
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

// ------------------------------------------------------------------ strip kernel (k = 21, C <= 3, no d_data)
// The tile kernel above streams at 2.6-3.1 TB/s from a cold Infinity Cache (bench.py's rotating-buffer probe): 1,152 short-
// lived blocks, each of which starts with a dependent halo gather and then keeps only ONE pixel quad of logits (7 KB per
// wave) in flight ahead of its arithmetic.  This kernel is persistent instead: every block walks DOWN a 16-pixel-wide strip
// of the image, one row per step (16 pixels = 28 KB of contiguous logits, 4 pixels = 7 KB per wave), across strips and
// images, and every wave streams the rows of ITS four pixels through a wave-private ring of D slots in LDS by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: no staging registers, D rows in flight while a row is multiplied, counted `vmcnt`
// waits, no barrier: the ring is private to the wave).  The radiance halo of all the rows a block owns in a strip
// ((rows + 20) x 36 pixels) is staged once per strip.  Every global access of the steady-state loop is a buffer
// instruction whose masked-off lanes use an out-of-range offset, so the instruction count per row is static and
// `vmcnt(N)` can be exact.  Lane / tap assignment, arithmetic and reduction order are those of the tile kernel: results
// are bit-identical to it (tests/test_gpu_ops.py::test_kernel_apply_strip_equals_tile_kernel).
constexpr int KS_TX = 16;                     // pixels per strip row: 4 waves x 4 pixels
constexpr int KS_HW = KS_TX + 20 + 1;         // halo row stride in pixels (36 used; odd, so rows shift the LDS banks)
constexpr int KS_PXB = 111 * 16;              // bytes of one pixel's taps in the ring: 111 vectors (441 taps + 3 pad)
constexpr int KS_ROWB = 7 * 1024;             // ring slot: 4 pixels = 444 vectors, filled by 7 wave instructions of 1 KB
constexpr unsigned KS_OOB = 0x80000000u;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ks_f32(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned ks_u32(float f) { return __builtin_bit_cast(unsigned, f); }
struct KSParams { KAParams a; int strips, nchunks; unsigned logit_bytes, out_bytes, lse_bytes, dl_bytes, data_bytes, gout_bytes; };

// ABL (timing-only, `make debug` builds): 1 = no arithmetic (stream only), 2 = no logits stream (arithmetic only)
// workgroup barrier without the release fence of __syncthreads(), for which hipcc drains every outstanding LDS-DMA
__device__ __forceinline__ void pw_barrier_ks() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool BWD, int D, int MAXROWS, int ABL = 0>
__global__ __launch_bounds__(256, D <= 2 ? 2 : 1) void kernel_apply_strip_kernel(KSParams ps) {
  const KAParams& p = ps.a;
  constexpr int KS = 21, TAPS = KS * KS, NVEC = (TAPS + 3) / 4, R = KS / 2;
  constexpr int S = BWD ? KA_MAXV : 0;        // global stores per row and wave (static; the forward parks its results in LDS)
  constexpr int NWAIT = D * S + 7 * (D - 1);  // vector-memory instructions younger than the DMA of the row being consumed
  static_assert(NWAIT < 64, "vmcnt is a 6-bit counter");
  static_assert(D <= 4, "the start-up waits below are written out for up to four ring slots");
  constexpr float LOG2E = 1.4426950408889634f;
  // SEPARATE static arrays on purpose: hipcc (ROCm 7.2) makes every LDS read it cannot prove disjoint from an outstanding
  // LDS-DMA wait for vmcnt(0); reads of another __shared__ variable carry the alias scopes that prove it (one dynamic
  // array does not).  The ring itself is read through inline asm below, behind the counted wait.
  __shared__ __attribute__((aligned(16))) char ring_all[4 * D * KS_ROWB];
  __shared__ __attribute__((aligned(16))) float4 halo[(MAXROWS + 2 * R) * KS_HW];
  __shared__ __attribute__((aligned(16))) float4 gq[BWD ? MAXROWS * KS_TX : 1];       // {g0, g1, g2, g . out} per pixel
  __shared__ float lse2[BWD ? MAXROWS * KS_TX : 1];                                   // log-sum-exp in base 2
  // FWD: {result rgb, log-sum-exp} of every pixel of the block's rows in the current strip, written to global memory when
  // the block leaves the strip.  (Twelve-byte strided stores per pixel inside the loop would sit in the in-order vmcnt
  // queue in front of the next rows' DMA: the stream-only ablation ran at 3.8 TB/s with them.)
  __shared__ __attribute__((aligned(16))) float4 outs[BWD ? 1 : MAXROWS * KS_TX];
  char* const ring = ring_all + (threadIdx.x >> 6) * (D * KS_ROWB);                   // this wave's D slots

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int c0 = (int)((int64_t)blockIdx.x * ps.nchunks / gridDim.x);
  const int c1 = (int)((int64_t)(blockIdx.x + 1) * ps.nchunks / gridDim.x);
  const int px = 4 * wave + q;                    // this lane group's pixel inside the strip row
  const int per_img = ps.strips * p.h;

  auto decode = [&](int c, int& img, int& strip, int& y) {
    img = c / per_img;
    const int rem = c - img * per_img;
    strip = rem / p.h;
    y = rem - strip * p.h;
  };

  // ---- the logits stream.  DMA instruction k of a row moves vectors [64k, 64k + 64) of the wave's 444-vector image
  // (pixel v / 111, vector v % 111); the per-lane source offset relative to the wave's first pixel is loop-invariant.
  const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc((void*)p.logits, 0, (int)ps.logit_bytes, 0x00020000);
  unsigned rel[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int v = 64 * k + lane, pq = v / 111, vec = v - pq * 111;
    rel[k] = pq < 4 ? (unsigned)(pq * p.lsw * 4 + vec * 16) | ((unsigned)pq << 28) : KS_OOB;   // pixel index rides in bits 28-29
  }
  // (img, strip, y) of consecutive rows by stepping, not by two divisions per row
  auto advance = [&](int& img, int& strip, int& y) {
    if (++y == p.h) {
      y = 0;
      if (++strip == ps.strips) { strip = 0; ++img; }
    }
  };
  auto dma_row = [&](int c, int img, int strip, int y, int slot) {
    const int xw = strip * KS_TX + 4 * wave;                    // first pixel of this wave
    const bool row_ok = c < c1;
    const unsigned base = (unsigned)(((int64_t)img * p.lsn + (int64_t)y * p.lsh + (int64_t)xw * p.lsw) * 4);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const unsigned r = rel[k];
      const int pq = (int)((r >> 28) & 3u);
      const unsigned off = (ABL != 2 && row_ok && r < KS_OOB && xw + pq < p.w) ? base + (r & 0x0fffffffu) : KS_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (__attribute__((address_space(3))) void*)(ring + slot * KS_ROWB + k * 1024),
                                               16, off, 0, 0, 0);
    }
  };
  int pimg, pstrip, py;                                         // the row the next DMA fetches
  decode(c0, pimg, pstrip, py);
  int img = pimg, strip = pstrip, y = py;                       // the row being multiplied

  int toff[KA_MAXV][4];
#pragma unroll
  for (int i = 0; i < KA_MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = 4 * (j + 16 * i) + e;
      const int dy = t / KS, dx = t - dy * KS;
      toff[i][e] = t < TAPS ? (dy * KS_HW + dx) * 16 : 0;
    }
  const char* halo_b = reinterpret_cast<const char*>(halo);
  const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dlogits, 0, (int)ps.dl_bytes, 0x00020000);

  int seg_rows = 0, seg_img, seg_strip, seg_y0;
  // this wave's pixels of the finished rows of the strip: lane = 16 * (row & 3) + 4 * pixel + component
  auto flush_outs = [&]() {
    if (BWD) return;
    for (int r0 = 0; r0 < seg_rows; r0 += 4) {
      const int r = r0 + (lane >> 4), pq = (lane >> 2) & 3, ch = lane & 3;
      const int y = seg_y0 + r, x = seg_strip * KS_TX + 4 * wave + pq;
      if (r < seg_rows && x < p.w) {
        const float v = reinterpret_cast<const float*>(&outs[r * KS_TX + 4 * wave + pq])[ch];
        if (ch < 3) {
          if (ch < p.C) const_cast<float*>(p.out)[(int64_t)seg_img * p.osn + (int64_t)ch * p.osc + (int64_t)y * p.osh + (int64_t)x * p.osw] = v;
        } else if (p.lse) {
          p.lse[((int64_t)seg_img * p.h + y) * p.w + x] = v;
        }
      }
    }
  };
  // ---- staging of a strip segment (the rows [y, y + rows) of one strip that this block owns): the radiance halo, and for
  // the backward {g, g . out, lse} per pixel.  Loads and LDS writes are separate steps with a STATIC number of buffer
  // loads (masked-off elements use an out-of-range offset), so that at the start of the block the logits DMA of the
  // first D rows can be issued between them and a counted wait lets it stay in flight.
  constexpr int HCOLS = KS_TX + 2 * R, HIT = ((MAXROWS + 2 * R) * HCOLS + 255) / 256;
  static_assert(MAXROWS * KS_TX <= 256 || !BWD, "one pixel of the segment per thread");
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)p.data, 0, (int)ps.data_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)p.gout, 0, (int)ps.gout_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)ps.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)p.lse, 0, (int)ps.lse_bytes, 0x00020000);
  float hv[HIT][3], gv[7];
  auto stage_loads = [&](int simg, int sstrip, int sy, int rows) {
    const int hrows = rows + 2 * R, sx0 = sstrip * KS_TX;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int i = threadIdx.x + 256 * it;
      const int hy = i / HCOLS, hx = i - hy * HCOLS;
      const int yy = sy + hy - R, xx = sx0 + hx - R;
      const bool ok = hy < hrows && (unsigned)yy < (unsigned)p.h && (unsigned)xx < (unsigned)p.w;
      const unsigned base = (unsigned)(((int64_t)simg * p.dsn + (int64_t)yy * p.dsh + (int64_t)xx * p.dsw) * 4);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        hv[it][ch] = ks_f32(__builtin_amdgcn_raw_buffer_load_b32(dr, (ok && ch < p.C) ? base + (unsigned)(ch * p.dsc * 4) : KS_OOB, 0, 0));
    }
    if (BWD) {
      const int i = threadIdx.x, ry = i / KS_TX, rx = i - ry * KS_TX;
      const int yy = sy + ry, xx = sx0 + rx;
      const bool ok = ry < rows && xx < p.w;
      const unsigned gb = (unsigned)(((int64_t)simg * p.gsn + (int64_t)yy * p.gsh + (int64_t)xx * p.gsw) * 4);
      const unsigned ob = (unsigned)(((int64_t)simg * p.osn + (int64_t)yy * p.osh + (int64_t)xx * p.osw) * 4);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        gv[ch] = ks_f32(__builtin_amdgcn_raw_buffer_load_b32(gr, (ok && ch < p.C) ? gb + (unsigned)(ch * p.gsc * 4) : KS_OOB, 0, 0));
        gv[3 + ch] = ks_f32(__builtin_amdgcn_raw_buffer_load_b32(orr, (ok && ch < p.C) ? ob + (unsigned)(ch * p.osc * 4) : KS_OOB, 0, 0));
      }
      gv[6] = ks_f32(__builtin_amdgcn_raw_buffer_load_b32(sr, ok ? (unsigned)((((int64_t)simg * p.h + yy) * p.w + xx) * 4) : KS_OOB, 0, 0));
    }
  };
  auto stage_writes = [&](int rows) {
    const int hrows = rows + 2 * R;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int i = threadIdx.x + 256 * it;
      const int hy = i / HCOLS, hx = i - hy * HCOLS;
      if (hy < hrows) halo[hy * KS_HW + hx] = make_float4(hv[it][0], hv[it][1], hv[it][2], 1.f);
    }
    if (BWD) {
      const int i = threadIdx.x;
      if (i < rows * KS_TX) {
        float go = 0.f;                                     // same order as the tile kernel: sum over the channels
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) go += gv[ch] * gv[3 + ch];
        gq[i] = make_float4(gv[0], gv[1], gv[2], go);
        lse2[i] = gv[6] * LOG2E;
      }
    }
  };

  // ---- block start: halo loads, then the DMA of the first D rows, then wait for the halo loads ONLY
  seg_img = img; seg_strip = strip; seg_y0 = y;
  {
    const int rows = min(p.h - y, c1 - c0);
    stage_loads(img, strip, y, rows);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dma_row(c0 + d, pimg, pstrip, py, d);
      advance(pimg, pstrip, py);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * D) : "memory");
    stage_writes(rows);
    pw_barrier_ks();
  }
  // Outer loop: the strip segments of this block (one, seldom two or three); inner loop: the rows of a segment.  The
  // inner loop contains no vector-memory instruction that returns into registers (only LDS-DMA and, in the backward,
  // stores), so the compiler's own wait counts stay out of it.
  int slot = 0;
  int nrow = 0;                                             // rows multiplied since the block started (counted up to D)
  for (int c = c0; c < c1;) {
    const int seg_n = min(p.h - y, c1 - c);                 // rows of this strip the block owns (<= MAXROWS)
    if (c != c0) {                                          // the block enters another strip (at most twice per block)
      flush_outs();
      seg_rows = 0;
      // drain this wave's DMA: after it the first D - 1 rows need no wait at all and the exact count below holds again
      // from row c + D on; the barrier keeps a fast wave off the halo a slow one still reads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pw_barrier_ks();
      stage_loads(img, strip, y, seg_n);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stage_writes(seg_n);
      pw_barrier_ks();
      seg_img = img; seg_strip = strip; seg_y0 = y;
    }
   for (int sr_ = 0; sr_ < seg_n; ++sr_, ++c, ++y) {
    const int x0 = strip * KS_TX;
    // this wave's row c has landed; the D - 1 rows behind it (and the stores of the last D rows) stay in flight.
    // NWAIT is the steady state.  The first D rows of a block have fewer stores behind them: row k of the block is
    // followed by the DMA of the D - 1 rows after it and by the S stores of each of the k rows already multiplied
    // (after a strip change everything issued before it was drained, so the steady-state count is never too weak there).
    if (S != 0 && nrow < D) {
      if (nrow == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (D - 1)) : "memory");
      else if (nrow == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (D - 1) + S) : "memory");
      else if (nrow == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (D - 1) + 2 * S) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (D - 1) + 3 * S) : "memory");
      ++nrow;
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
    }
    float4 l4[KA_MAXV];
    {
      // (vector 111 of a pixel -- lane 15, i = 6 -- is the next pixel's first vector or DMA zero-fill: all four of its
      // slots are past k*k and become -1e30 below)
      const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(ring + slot * KS_ROWB + q * KS_PXB + j * 16);
      u32x4 rv[KA_MAXV];
      asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:256\n\tds_read_b128 %2, %7 offset:512\n\t"
                   "ds_read_b128 %3, %7 offset:768\n\tds_read_b128 %4, %7 offset:1024\n\tds_read_b128 %5, %7 offset:1280\n\t"
                   "ds_read_b128 %6, %7 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3]), "=&v"(rv[4]), "=&v"(rv[5]), "=&v"(rv[6])
                   : "v"(ra)
                   : "memory");
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) {
        const int vi = j + 16 * i;
        // (by-value helper on purpose: __builtin_bit_cast applied directly to a vector-element lvalue reads element 0)
        float4 v = make_float4(ks_f32(rv[i][0]), ks_f32(rv[i][1]), ks_f32(rv[i][2]), ks_f32(rv[i][3]));
        if (4 * (15 + 16 * i) + 3 >= TAPS) {
          const int t = 4 * vi;
          if (t + 0 >= TAPS) v.x = KA_NEG;
          if (t + 1 >= TAPS) v.y = KA_NEG;
          if (t + 2 >= TAPS) v.z = KA_NEG;
          if (t + 3 >= TAPS) v.w = KA_NEG;
        }
        l4[i] = v;
      }
    }
    dma_row(c + D, pimg, pstrip, py, slot);                 // the slot is in registers: refill it, D rows ahead
    advance(pimg, pstrip, py);
    slot = slot + 1 == D ? 0 : slot + 1;

    const int x = x0 + px;
    const bool valid = x < p.w;
    const int ry = y - seg_y0;
    const char* hbase = halo_b + (ry * KS_HW + px) * 16;

    if (!BWD) {
      float m = KA_NEG;
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) m = fmaxf(fmaxf(m, fmaxf(l4[i].x, l4[i].y)), fmaxf(l4[i].z, l4[i].w));
      m = group16_max(m) * LOG2E;
      float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int i = 0; i < (ABL == 1 ? 1 : KA_MAXV); ++i) {
        const float* l = reinterpret_cast<const float*>(&l4[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = __builtin_amdgcn_exp2f(fmaf(l[e], LOG2E, -m));
          const float4 dv = *reinterpret_cast<const float4*>(hbase + toff[i][e]);
          a0 = fmaf(ex, dv.x, a0); a1 = fmaf(ex, dv.y, a1); a2 = fmaf(ex, dv.z, a2); s = fmaf(ex, dv.w, s);
        }
      }
      group16_sum4(s, a0, a1, a2);
      const float inv = 1.f / s;
      const float lsev = (m + __builtin_amdgcn_logf(s)) * 0.6931471805599453f;     // natural-log LSE
      if (j == 0) outs[ry * KS_TX + px] = make_float4(a0 * inv, a1 * inv, a2 * inv, lsev);
      seg_rows = ry + 1;
    } else {
      const float4 gg = gq[ry * KS_TX + px];
      const float lb = lse2[ry * KS_TX + px];
      const unsigned qb = (unsigned)(((int64_t)img * p.qsn + (int64_t)y * p.qsh + (int64_t)x * p.qsw) * 4);
#pragma unroll
      for (int i = 0; i < KA_MAXV; ++i) {
        const float* l = reinterpret_cast<const float*>(&l4[i]);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float wt = __builtin_amdgcn_exp2f(fmaf(l[e], LOG2E, -lb));
          const float4 dv = *reinterpret_cast<const float4*>(hbase + toff[i][e]);
          const float gd = fmaf(gg.x, dv.x, fmaf(gg.y, dv.y, gg.z * dv.z));
          o[e] = wt * (gd - gg.w);
        }
        const int vi = j + 16 * i;
        const u32x4 ov = {ks_u32(o[0]), ks_u32(o[1]), ks_u32(o[2]), ks_u32(o[3])};
        __builtin_amdgcn_raw_buffer_store_b128(ov, qr, (valid && vi < NVEC) ? qb + 16u * (unsigned)vi : KS_OOB, 0, 0);
      }
    }
   }
    if (y == p.h) {                                         // on to the next strip / image
      y = 0;
      if (++strip == ps.strips) { strip = 0; ++img; }
    }
  }
  flush_outs();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // DMAs of the (masked-off) rows past the end
}

template <bool BWD, int D, int MAXROWS>
static int ks_launch2(const KSParams& ps, int nb, hipStream_t st) {
#ifdef WCMC_DEBUG_BUILD
  {
    const char* e = ab_env("WCMC_DEBUG_ABLATE");
    const int ab = e ? atoi(e) : 0;
    if (!BWD && D == 2 && (ab == 1 || ab == 2)) {
      if (ab == 1) hipLaunchKernelGGL((kernel_apply_strip_kernel<false, 2, MAXROWS, 1>), dim3((unsigned)nb), dim3(256), 0, st, ps);
      else hipLaunchKernelGGL((kernel_apply_strip_kernel<false, 2, MAXROWS, 2>), dim3((unsigned)nb), dim3(256), 0, st, ps);
      return check_launch("kernel_apply_fwd(strip, ablation)");
    }
  }
#endif
  hipLaunchKernelGGL((kernel_apply_strip_kernel<BWD, D, MAXROWS>), dim3((unsigned)nb), dim3(256), 0, st, ps);
  return check_launch(BWD ? "kernel_apply_bwd(strip)" : "kernel_apply_fwd(strip)");
}

static int64_t ks_span(int64_t n, int64_t sn, int64_t h, int64_t sh, int64_t w, int64_t sw, int64_t c, int64_t sc) {
  return ((n - 1) * sn + (h - 1) * sh + (w - 1) * sw + (c - 1) * sc + 1) * 4;
}
static int ks_launch(KAParams& a, bool bwd, void* stream) {
  KSParams ps;
  ps.a = a;
  ps.strips = (a.w + KS_TX - 1) / KS_TX;
  ps.nchunks = a.N * ps.strips * a.h;
  ps.logit_bytes = (unsigned)ks_span(a.N, a.lsn, a.h, a.lsh, a.w, a.lsw, a.taps, 1);
  ps.out_bytes = (unsigned)ks_span(a.N, a.osn, a.h, a.osh, a.w, a.osw, a.C, a.osc);
  ps.lse_bytes = (unsigned)((int64_t)a.N * a.h * a.w * 4);
  ps.dl_bytes = bwd ? (unsigned)ks_span(a.N, a.qsn, a.h, a.qsh, a.w, a.qsw, a.taps, 1) : 0u;
  ps.data_bytes = (unsigned)ks_span(a.N, a.dsn, a.h, a.dsh, a.w, a.dsw, a.C, a.dsc);
  ps.gout_bytes = bwd ? (unsigned)ks_span(a.N, a.gsn, a.h, a.gsh, a.w, a.gsw, a.C, a.gsc) : 0u;
  // two blocks per CU (2-slot rings: 57 KB of LDS per block); one block with 3 or 4 slots per wave measured 12-20 % slower
  constexpr int MAXROWS = 10;
  int nb = 512;
  if (nb > ps.nchunks) nb = ps.nchunks;
  const int need = (ps.nchunks + MAXROWS - 1) / MAXROWS;              // a block's share of one strip fits the halo buffer
  if (nb < need) nb = need;
  hipStream_t st = (hipStream_t)stream;
  return bwd ? ks_launch2<true, 2, MAXROWS>(ps, nb, st) : ks_launch2<false, 2, MAXROWS>(ps, nb, st);
}
// 32-bit buffer offsets: every view must span less than 2 GiB and have non-negative strides
static bool ks_ok(const KAParams& a, bool bwd) {
  if (a.k != 21 || a.C > 3 || (int64_t)a.N * ((a.w + KS_TX - 1) / KS_TX) * a.h >= (1ll << 30)) return false;
  const int64_t lim = 0x7ff00000ll;
  if (a.lsn < 0 || a.lsh < 0 || a.lsw < 0 || ks_span(a.N, a.lsn, a.h, a.lsh, a.w, a.lsw, a.taps, 1) >= lim) return false;
  if ((int64_t)a.lsw * 16 >= (1ll << 28)) return false;              // (rel[] packs the pixel index above bit 28)
  if (a.osn < 0 || a.osc < 0 || a.osh < 0 || a.osw < 0 || ks_span(a.N, a.osn, a.h, a.osh, a.w, a.osw, a.C, a.osc) >= lim) return false;
  if (bwd && (a.qsn < 0 || a.qsh < 0 || a.qsw < 0 || ks_span(a.N, a.qsn, a.h, a.qsh, a.w, a.qsw, a.taps, 1) >= lim)) return false;
  if (a.dsn < 0 || a.dsc < 0 || a.dsh < 0 || a.dsw < 0 || ks_span(a.N, a.dsn, a.h, a.dsh, a.w, a.dsw, a.C, a.dsc) >= lim) return false;
  if (bwd && (a.gsn < 0 || a.gsc < 0 || a.gsh < 0 || a.gsw < 0 || ks_span(a.N, a.gsn, a.h, a.gsh, a.w, a.gsw, a.C, a.gsc) >= lim)) return false;
  return (int64_t)a.N * a.h * a.w * 4 < lim;
}

// WCMC_KA_TILE=1: A/B switch back to the tile kernel (read per call: the parity test compares the two in one process)
static bool ka_force_tile() {
  const char* e = ab_env("WCMC_KA_TILE");
  return e && e[0] == '1';
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
  if (ks_ok(p, false) && !ka_force_tile()) return ks_launch(p, false, stream);
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
  if (!d_data && ks_ok(p, true) && !ka_force_tile()) return ks_launch(p, true, stream);
  const dim3 grid((unsigned)(((h + KA_TILE - 1) / KA_TILE) * ((w + KA_TILE - 1) / KA_TILE)), (unsigned)N);
  const size_t lds = (size_t)p.halo * p.halo * sizeof(float4) * (d_data ? 2 : 1);
  if (k == 21 && C <= 3) hipLaunchKernelGGL((kernel_apply_kernel<true, 21, false>), grid, dim3(128), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((kernel_apply_kernel<true, 0, true>), grid, dim3(128), lds, (hipStream_t)stream, p);
  return check_launch("kernel_apply_bwd");
}
