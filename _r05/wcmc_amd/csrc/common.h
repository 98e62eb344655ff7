// Shared helpers for libwcmc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/wcmc_hip.h"

namespace wcmc {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// A/B switches -- WCMC_* environment variables that select a non-default kernel, tiling or schedule, every one of them measured
// and decided (scripts/sweep_switches.sh, profiles/) -- exist in the DEBUG build only (`make debug`, loaded with WCMC_DEBUG_LIB=1):
// the release library reads no environment variable, its launch plans are the defaults.
#ifdef WCMC_DEBUG_BUILD
inline const char* ab_env(const char* name) { return getenv(name); }
#else
inline const char* ab_env(const char*) { return nullptr; }
#endif

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// An NHWC view obeys the contract in wcmc_hip.h.
inline bool nhwc_view_ok(const void* p, int64_t sn, int64_t sh, int64_t sw, int C) {
  return p != nullptr && aligned16(p) && (sn % 4 == 0) && (sh % 4 == 0) && (sw % 4 == 0) &&
         sw >= round_up(C, 4);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  if (act == WCMC_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == WCMC_ACT_LEAKY_RELU) return v > 0.f ? v : v * slope;
  return v;
}
// derivative factor from the POST-activation value
__device__ __forceinline__ float act_gate(float post, int act, float slope) {
  if (act == WCMC_ACT_RELU) return post > 0.f ? 1.f : 0.f;
  if (act == WCMC_ACT_LEAKY_RELU) return post > 0.f ? 1.f : slope;
  return 1.f;
}

// (n, y, x, v) of the idx-th unit of an [N][H][W][V] index space, in 32-bit arithmetic where the count fits: a 64-bit divide
// costs ~100 instructions on gfx950 and the glue kernels that decode an index this way move 16-32 bytes per thread (three of
// them per element held wcmc_cat_upsample_split at 1.3-2.3 TB/s)
struct NhwvIndex { int n, y, x, v; };
__device__ __forceinline__ NhwvIndex decode_nhwv(int64_t idx, int64_t total, int H, int W, int V) {
  NhwvIndex r;
  if (total <= 0x7fffffffll) {
    unsigned t = (unsigned)idx;
    r.v = (int)(t % (unsigned)V); t /= (unsigned)V;
    r.x = (int)(t % (unsigned)W); t /= (unsigned)W;
    r.y = (int)(t % (unsigned)H); r.n = (int)(t / (unsigned)H);
  } else {
    int64_t t = idx;
    r.v = (int)(t % V); t /= V;
    r.x = (int)(t % W); t /= W;
    r.y = (int)(t % H); r.n = (int)(t / H);
  }
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, DEVICE): a process-wide "already set" flag leaves the kernel at
// the 64 KB default on every other GPU of the process.  One LdsAttr per launch site remembers the largest value set
// per device ordinal; ordinals beyond the table set the attribute on every launch.
struct LdsAttr { std::atomic<size_t> v[16]; LdsAttr() { for (auto& x : v) x.store(0); } };
// Returns hipSuccess, or the error of hipFuncSetAttribute (the launch that follows would fail with a less telling one).  The table
// entry is published only AFTER the attribute has been set: a second host thread that sees it may launch with that much dynamic LDS.
static inline hipError_t set_max_lds(const void* fn, size_t lds, LdsAttr& a) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < 16;
  if (tracked && lds <= a.v[dev].load(std::memory_order_acquire)) return hipSuccess;
  const hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (rc != hipSuccess) {
    set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed: %s", lds, hipGetErrorString(rc));
    return rc;
  }
  if (tracked) {
    size_t cur = a.v[dev].load(std::memory_order_relaxed);
    while (cur < lds && !a.v[dev].compare_exchange_weak(cur, lds, std::memory_order_release, std::memory_order_relaxed)) {}
  }
  return hipSuccess;
}

}  // namespace wcmc

#define WCMC_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      wcmc::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)
