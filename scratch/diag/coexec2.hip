// Cost of VMEM / LDS-DMA / LDS-write / barrier instructions issued by OTHER waves of the SIMD on an MFMA-bound wave.
// MFMA waves time themselves with wall_clock64 (100 MHz) so slower noise waves do not pollute the number.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int KIND, int R>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters, const float* gin, float* gout) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  if (wave < 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    const long long t1 = wall_clock64();
    if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
    if (s == 123.456f) out[0] = s;
  } else {
    const int w4 = __builtin_amdgcn_readfirstlane(wave - 4);
    const float* src = gin + (((blockIdx.x * 4 + w4) * 64 + lane) & 16383) * 4;
    float* dst = gout + ((blockIdx.x * 4 + w4) * 64 + lane) * 4;
    f32x4 v = (f32x4){1.f, 2.f, 3.f, 4.f};
    // R operations per 8 MFMAs of the partner wave; batches of 8 iterations then one wait, so latency is amortised
    for (int it = 0; it < iters / 8; ++it) {
#pragma unroll
      for (int q = 0; q < 8 * R; ++q) {
        if (KIND == 1) __builtin_amdgcn_global_load_lds(src + (q & 7) * 4096, (lds_ptr_t)(lds + 1024 + w4 * 256), 16, 0, 0);
        if (KIND == 2) { f32x4 t = *(const volatile f32x4*)(src + (q & 7) * 4096); asm volatile("" :: "v"(t)); }
        if (KIND == 3) *(volatile f32x4*)(dst + (q & 7) * 65536) = v;
        if (KIND == 4) *(volatile f32x4*)(lds + 2048 + (threadIdx.x & 255) * 4) = v;
      }
      if (KIND == 5) { for (int q = 0; q < 8 * R; ++q) asm volatile("s_nop 7\n\ts_sleep 1"); }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    }
    if (v.x == 123.456f) out[1] = v.x;
  }
}
static float *gin, *gout;
template <int KIND, int R>
static void run(const char* name, int iters) {
  float* out; long long* clk; hipMalloc(&out, 8); hipMalloc(&clk, 256 * 4 * 8);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<KIND, R>), dim3(256), dim3(512), 0, 0, out, clk, iters, gin, gout);
    hipDeviceSynchronize();
    std::vector<long long> h(1024); hipMemcpy(h.data(), clk, 1024 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= 1024;
    best = std::min(best, avg);
  }
  const double ns = best * 10.0;     // 100 MHz
  printf("%-36s %d per 8 MFMA: MFMA waves %.3f ms -> %.1f cycles/MFMA (2.4 GHz)  (+%.1f cycles per op)\n", name, R, ns * 1e-6, ns * 2.4 / (iters * 8.0),
         (ns * 2.4 / (iters * 8.0) - 32.3) * 8.0 / R);
  hipFree(out); hipFree(clk);
}
int main() {
  hipMalloc(&gin, 16384 * 16 + 8 * 4096 * 4 + 65536); hipMemset(gin, 0, 16384 * 16 + 8 * 4096 * 4 + 65536);
  hipMalloc(&gout, (size_t)8 * 65536 * 4 + 256 * 4 * 64 * 16 + 65536);
  const int it = 16000;
  { float* o; long long* c; hipMalloc(&o, 8); hipMalloc(&c, 8192); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(512), 0, 0, o, c, 40000, gin, gout); hipDeviceSynchronize(); }
  run<0, 1>("mfma only", it);
  run<1, 1>("other-wave global_load_lds x4", it); run<1, 2>("other-wave global_load_lds x4", it); run<1, 4>("other-wave global_load_lds x4", it);
  run<2, 1>("other-wave global_load_dwordx4", it); run<2, 2>("other-wave global_load_dwordx4", it); run<2, 4>("other-wave global_load_dwordx4", it);
  run<3, 1>("other-wave global_store_dwordx4", it); run<3, 2>("other-wave global_store_dwordx4", it);
  run<4, 1>("other-wave ds_write_b128", it); run<4, 4>("other-wave ds_write_b128", it);
  return 0;
}
