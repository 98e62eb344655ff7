// Cold-cache read ceiling for a 121 MB buffer on this box: plain grid-stride float4 loads vs LDS-DMA, several depths.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void rd_plain(const float4* __restrict__ g, float* o, size_t n4) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = g[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) { float4 v = g[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) o[0] = acc;
}
// each block streams a contiguous chunk through an LDS ring of R slots of 1 KB per wave
template <int R>
__global__ __launch_bounds__(256) void rd_dma(const float* g, float* o, unsigned nbytes, unsigned per_wave) {
  __shared__ __attribute__((aligned(16))) char ring[4 * R * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)nbytes, 0x00020000);
  const unsigned w0 = (blockIdx.x * 4 + wave) * per_wave;
  const int n = per_wave / 1024;
  char* my = ring + wave * R * 1024;
#pragma unroll
  for (int s = 0; s < R; ++s)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + s * 1024), 16, s < n ? w0 + s * 1024 + lane * 16 : 0x80000000u, 0, 0, 0);
  int slot = 0;
  for (int i = 0; i < n; ++i) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + slot * 1024), 16, i + R < n ? w0 + (i + R) * 1024 + lane * 16 : 0x80000000u, 0, 0, 0);
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (per_wave == 7) o[0] = ring[lane];
}
// the strip kernel's walk: block = 4 waves x 7104 contiguous bytes per row (one 16-pixel strip row), rows `rstride` bytes
// apart, D rows in flight per wave, wait for a whole row
template <int D>
__global__ __launch_bounds__(256, 2) void rd_rows(const float* g, float* o, unsigned nbytes, int nchunks, int h, int strips,
                                                  unsigned rstride, unsigned sstride, unsigned istride, int aligned) {
  __shared__ __attribute__((aligned(16))) char ring[4 * D * 7168];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)nbytes, 0x00020000);
  const int c0 = (int)((long long)blockIdx.x * nchunks / gridDim.x), c1 = (int)((long long)(blockIdx.x + 1) * nchunks / gridDim.x);
  char* my = ring + wave * D * 7168;
  auto issue = [&](int c, int slot) {
    const int img = c / (strips * h), rem = c - img * strips * h, strip = rem / h, y = rem - strip * h;
    unsigned base = img * istride + y * rstride + strip * sstride + wave * 7104;
    if (aligned) base &= ~127u;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int v = 64 * k + lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + slot * 7168 + k * 1024), 16,
                                               (c < c1 && v < 444) ? base + v * 16 : 0x80000000u, 0, 0, 0);
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(c0 + d, d);
  int slot = 0;
  for (int c = c0; c < c1; ++c) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (D - 1)) : "memory");
    issue(c + D, slot);
    slot = slot + 1 == D ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (nchunks == 7) o[0] = ring[lane];
}
int main() {
  const size_t bytes = 121069056 / 16 * 16;   // (8,441,92,92)-sized
  const int NB = 8;
  std::vector<float*> bufs(NB);
  for (auto& b : bufs) { (void)hipMalloc(&b, bytes + 4096); (void)hipMemset(b, 1, bytes); }
  float* o; (void)hipMalloc(&o, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < NB; ++i) launch(bufs[i]);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int it = 48;
    for (int i = 0; i < it; ++i) launch(bufs[i % NB]);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.1f us  %5.2f TB/s\n", name, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
  };
  const size_t n4 = bytes / 16;
  for (int blocks : {512, 1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "plain U=4 blocks=%d", blocks);
    timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_plain<4>, dim3(blocks), dim3(256), 0, 0, (const float4*)b, o, n4); });
    snprintf(nm, 64, "plain U=8 blocks=%d", blocks);
    timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_plain<8>, dim3(blocks), dim3(256), 0, 0, (const float4*)b, o, n4); });
  }
  for (int blocks : {256, 512, 1024}) {
    const unsigned per_wave = (unsigned)(bytes / (blocks * 4) / 1024 * 1024);
    char nm[64];
    snprintf(nm, 64, "lds-dma R=8 blocks=%d", blocks);
    timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_dma<8>, dim3(blocks), dim3(256), 0, 0, b, o, (unsigned)bytes, per_wave); });
    snprintf(nm, 64, "lds-dma R=16 blocks=%d", blocks);
    timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_dma<16>, dim3(blocks), dim3(256), 0, 0, b, o, (unsigned)bytes, per_wave); });
    snprintf(nm, 64, "lds-dma R=32 blocks=%d", blocks);
    timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_dma<32>, dim3(blocks), dim3(256), 0, 0, b, o, (unsigned)bytes, per_wave); });
  }
  {
    const int N = 8, H = 92, W = 92, strips = 6, nchunks = N * strips * H;
    const unsigned rstride = W * 1776, sstride = 16 * 1776, istride = H * W * 1776;
    for (int aligned : {0, 1})
      for (int blocks : {512, 1024, 2048}) {
        char nm[64];
        snprintf(nm, 64, "rows D=2 blocks=%d aligned=%d", blocks, aligned);
        timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_rows<2>, dim3(blocks), dim3(256), 0, 0, b, o, (unsigned)bytes, nchunks, H, strips, rstride, sstride, istride, aligned); });
        snprintf(nm, 64, "rows D=4 blocks=%d aligned=%d", blocks, aligned);
        timeit(nm, [&](float* b) { hipLaunchKernelGGL(rd_rows<4>, dim3(blocks), dim3(256), 0, 0, b, o, (unsigned)bytes, nchunks, H, strips, rstride, sstride, istride, aligned); });
      }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
