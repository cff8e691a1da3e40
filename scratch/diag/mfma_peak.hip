// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate and the shader clock it runs at (clock64 vs wall_clock64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int NACC>
static void run(int blocks, int iters) {
  float* out; long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 4 * NACC * 2048.0;
    double cyc = 0, wal = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wal += h[2 * i + 1]; }
    cyc /= blocks; wal /= blocks;
    int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
    printf("nacc %d blocks %d iters %d: %.3f ms  %.1f TF/s  clock64/wall = %.3f  wallrate %d kHz -> shader %.0f MHz ; mfma issue cycles/instr %.2f\n",
           NACC, blocks, iters, ms, flops / ms * 1e-9, cyc / wal, wcr, cyc / wal * wcr * 1e-3,
           cyc / ((double)iters * 4 * NACC * (blocks >= 256 ? (blocks / 256) : 1)));
  }
}
// same loop with per-lane random operands in distinct registers (data-dependent switching power)
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_rand(float* out, const float* in, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u * 2 + i) & 7], b[(u + i * 3) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 123.456f) out[0] = s;
}
static void run_rand(int blocks, int iters, float scale) {
  float* out; float* in;
  hipMalloc(&out, 4); hipMalloc(&in, 4096 * 4);
  std::vector<float> h(4096);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) - (1 << 23)) * scale / (1 << 23); }
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop_rand<8>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 4 * 8 * 2048.0;
    printf("rand scale %g blocks %d iters %d: %.3f ms  %.1f TF/s\n", scale, blocks, iters, ms, flops / ms * 1e-9);
  }
}
int main() {
  run_rand(512, 10000, 1.0f); run_rand(512, 10000, 1e-3f); run_rand(512, 100000, 1.0f);
  run<4>(256, 20000); run<8>(256, 10000); run<8>(512, 10000); run<8>(1024, 5000); run<8>(256 * 8, 20000); run<8>(64, 10000);
  return 0;
}
