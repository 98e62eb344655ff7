// Fused clip_grad_value_ + Adam step over one flat parameter buffer
// (support/interfaces.py:260-261,269-271; optimiser built at train_kpcn.py:274-277 with
// torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
// HBM-bound: 16 B read + 12 B written per parameter... 4 streams in, 4 out, 16 bytes per lane.
#include <math.h>

#include "common.h"

namespace wcmc {

// torch.clamp (what clip_grad_value_ calls) propagates NaN; fminf / fmaxf return the non-NaN operand and would turn a
// NaN gradient into exactly -clip.  The reference lets the NaN reach the parameters, so that the next loss is
// non-finite and training stops (interfaces.py:254-257); so does this.
__device__ __forceinline__ float clamp_nan(float g, float clip) {
  return g != g ? g : fminf(fmaxf(g, -clip), clip);
}

// hyper (optional, device): {step_size, beta1, beta2, 1 - beta1, 1 - beta2, eps, 1 / sqrt(1 - beta2^t)} -- the same seven
// floats the host passes by value otherwise.  A launch captured into the step's hipGraph bakes its by-value arguments in;
// step_size and the bias correction change every step (and lr whenever the caller's schedule says so), so the captured
// launch reads them from a buffer the host refreshes before each replay.
__global__ void clip_adam_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, float clip, float step_size, float beta1,
                                 float beta2, float omb1, float omb2, float eps, float inv_bc2_sqrt,
                                 float grad_scale, const float* __restrict__ guard, const float* __restrict__ hyper) {
  if (guard && guard[0] == 0.f) return;        // non-finite loss upstream: leave parameters and moments untouched
  if (hyper) {
    step_size = hyper[0]; beta1 = hyper[1]; beta2 = hyper[2]; omb1 = hyper[3]; omb2 = hyper[4]; eps = hyper[5];
    inv_bc2_sqrt = hyper[6];
  }
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 g = reinterpret_cast<float4*>(grad)[i];
    float4 p = reinterpret_cast<float4*>(param)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* gp = reinterpret_cast<float*>(&g); float* pp = reinterpret_cast<float*>(&p);
    float* mp = reinterpret_cast<float*>(&mm); float* vp = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gc = clamp_nan(gp[e] * grad_scale, clip);
      gp[e] = gc;
      mp[e] = beta1 * mp[e] + omb1 * gc;
      vp[e] = beta2 * vp[e] + omb2 * gc * gc;
      pp[e] -= step_size * mp[e] / (sqrtf(vp[e]) * inv_bc2_sqrt + eps);
    }
    reinterpret_cast<float4*>(grad)[i] = g;
    reinterpret_cast<float4*>(param)[i] = p;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  const int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float gc = clamp_nan(grad[t] * grad_scale, clip);
    grad[t] = gc;
    const float m1 = beta1 * m[t] + omb1 * gc;
    const float v1 = beta2 * v[t] + omb2 * gc * gc;
    m[t] = m1; v[t] = v1;
    param[t] -= step_size * m1 / (sqrtf(v1) * inv_bc2_sqrt + eps);
  }
}

// The head of the captured optimiser tail in ONE launch (round 4; it replaced ~10 five-microsecond torch kernels: stack,
// isfinite, all, cast, multiply, copy, where, add, cat): the reference's non-finite check of loss_dict (interfaces.py:254-257),
// the device guard of this step's update, and the running loss sums of interfaces.py:263-267.
//   flags[i] = isfinite(loss_i), flags[n] = guard = (all finite) * ok;  ok <- guard (a failed step poisons the steps enqueued
//   behind it until the host has raised);  sums[i] += guard ? loss_i : 0.
constexpr int GUARD_MAX = 16;
struct GuardArgs { const float* loss[GUARD_MAX]; int n; };
__global__ __launch_bounds__(64) void step_guard_kernel(GuardArgs a, float* __restrict__ ok, float* __restrict__ sums, float* __restrict__ flags) {
  const int t = threadIdx.x;
  const float v = t < a.n ? a.loss[t][0] : 0.f;
  const bool fin = !(v != v) && fabsf(v) != INFINITY;
  const unsigned long long bad = __ballot(t < a.n && !fin);
  const float guard = (bad == 0ull) ? ok[0] : 0.f;             // (ok is 1 or 0)
  if (t < a.n) {
    flags[t] = fin ? 1.f : 0.f;
    if (guard != 0.f) sums[t] += v;
  }
  __syncthreads();                                             // (every lane has read ok[0])
  if (t == 0) { flags[a.n] = guard; ok[0] = guard; }
}

// The same head for the MULTI-RANK tail, in its two halves (wcmc_amd/graph.py: graph A | eager RCCL all-reduces | graph B):
//   local   flags[i] = isfinite(loss_i); *slot = 1 - (all finite) * ok -- this rank's entry of the flag slot that travels with the first
//           gradient bucket (after the SUM over the ranks the slot holds the number of ranks that must not update)
//   global  guard = (*slot == 0): flags[n] = guard, ok <- guard, sums[i] += guard ? loss_i : 0
__global__ __launch_bounds__(64) void step_guard_local_kernel(GuardArgs a, const float* __restrict__ ok, float* __restrict__ flags, float* __restrict__ slot) {
  const int t = threadIdx.x;
  const float v = t < a.n ? a.loss[t][0] : 0.f;
  const bool fin = !(v != v) && fabsf(v) != INFINITY;
  const unsigned long long bad = __ballot(t < a.n && !fin);
  if (t < a.n) flags[t] = fin ? 1.f : 0.f;
  if (t == 0) slot[0] = 1.f - ((bad == 0ull) ? ok[0] : 0.f);
}
__global__ __launch_bounds__(64) void step_guard_global_kernel(GuardArgs a, const float* __restrict__ slot, float* __restrict__ ok, float* __restrict__ sums,
                                                               float* __restrict__ flags) {
  const int t = threadIdx.x;
  const float guard = slot[0] == 0.f ? 1.f : 0.f;
  if (t < a.n && guard != 0.f) sums[t] += a.loss[t][0];
  if (t == 0) { flags[a.n] = guard; ok[0] = guard; }
}

}  // namespace wcmc

using namespace wcmc;

static int guard_args(GuardArgs& a, const float* const* losses, int n) {
  a.n = n;
  for (int i = 0; i < GUARD_MAX; ++i) a.loss[i] = i < n ? losses[i] : nullptr;
  for (int i = 0; i < n; ++i)
    if (!a.loss[i]) return -1;
  return 0;
}
extern "C" int wcmc_step_guard_local(const float* const* losses, int n, const float* ok, float* flags, float* flag_slot, void* stream) {
  WCMC_REQUIRE(losses && ok && flags && flag_slot && n >= 1 && n <= GUARD_MAX, WCMC_ERR_BAD_ARG, "step_guard_local: bad argument (1 <= n <= %d)", GUARD_MAX);
  GuardArgs a;
  WCMC_REQUIRE(guard_args(a, losses, n) == 0, WCMC_ERR_BAD_ARG, "step_guard_local: null loss pointer");
  hipLaunchKernelGGL(step_guard_local_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, ok, flags, flag_slot);
  return check_launch("step_guard_local");
}
extern "C" int wcmc_step_guard_global(const float* const* losses, int n, const float* flag_slot, float* ok, float* sums, float* flags, void* stream) {
  WCMC_REQUIRE(losses && ok && sums && flags && flag_slot && n >= 1 && n <= GUARD_MAX, WCMC_ERR_BAD_ARG, "step_guard_global: bad argument (1 <= n <= %d)", GUARD_MAX);
  GuardArgs a;
  WCMC_REQUIRE(guard_args(a, losses, n) == 0, WCMC_ERR_BAD_ARG, "step_guard_global: null loss pointer");
  hipLaunchKernelGGL(step_guard_global_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, flag_slot, ok, sums, flags);
  return check_launch("step_guard_global");
}

extern "C" int wcmc_step_guard(const float* const* losses, int n, float* ok, float* sums, float* flags, void* stream) {
  WCMC_REQUIRE(losses && ok && sums && flags && n >= 1 && n <= GUARD_MAX, WCMC_ERR_BAD_ARG, "step_guard: bad argument (1 <= n <= %d)", GUARD_MAX);
  GuardArgs a;
  a.n = n;
  for (int i = 0; i < GUARD_MAX; ++i) a.loss[i] = i < n ? losses[i] : nullptr;
  for (int i = 0; i < n; ++i) WCMC_REQUIRE(a.loss[i], WCMC_ERR_BAD_ARG, "step_guard: null loss pointer %d", i);
  hipLaunchKernelGGL(step_guard_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, ok, sums, flags);
  return check_launch("step_guard");
}

extern "C" int wcmc_clip_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float clip,
                              double lr, double beta1, double beta2, double eps, int step, float grad_scale,
                              const float* guard, void* stream) {
  WCMC_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, WCMC_ERR_BAD_ARG,
               "clip_adam: bad argument (n=%lld step=%d)", (long long)n, step);
  WCMC_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), WCMC_ERR_ALIGNMENT,
               "clip_adam: buffers must be 16-byte aligned");
  // same arithmetic as torch.optim.Adam (single-tensor path): step_size = lr / (1 - beta1^t),
  // denom = sqrt(v) / sqrt(1 - beta2^t) + eps
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const int64_t blocks = ceil_div64(n / 4 > 0 ? n / 4 : 1, 256);
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, clip, step_size, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, inv_bc2_sqrt, grad_scale, guard,
                     (const float*)nullptr);
  return check_launch("clip_adam");
}

extern "C" void wcmc_clip_adam_hyper(double lr, double beta1, double beta2, double eps, int step, float* out7) {
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  out7[0] = (float)(lr / bc1); out7[1] = (float)beta1; out7[2] = (float)beta2; out7[3] = (float)(1.0 - beta1);
  out7[4] = (float)(1.0 - beta2); out7[5] = (float)eps; out7[6] = (float)(1.0 / sqrt(bc2));
}

extern "C" int wcmc_clip_adam_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float clip,
                                  float grad_scale, const float* hyper7, const float* guard, void* stream) {
  WCMC_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && hyper7, WCMC_ERR_BAD_ARG,
               "clip_adam_dev: bad argument (n=%lld)", (long long)n);
  WCMC_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), WCMC_ERR_ALIGNMENT,
               "clip_adam_dev: buffers must be 16-byte aligned");
  const int64_t blocks = ceil_div64(n / 4 > 0 ? n / 4 : 1, 256);
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, clip, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                     grad_scale, guard, hyper7);
  return check_launch("clip_adam_dev");
}
