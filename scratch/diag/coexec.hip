// Does ordinary VALU / LDS / SALU work issued by OTHER waves on the same SIMD slow down an MFMA-bound wave?
// Workgroup = 512 threads: waves 0..3 (one per SIMD) run an MFMA chain loop; waves 4..7 (one per SIMD) run a
// "noise" loop of the selected kind.  Also: MFMA + VALU interleaved inside the same wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NVALU>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* gin) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.f;
  __syncthreads();
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  if (wave < 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v0 = a, v1 = b, v2 = a + 1.f, v3 = b + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (KIND == 10) {           // same-wave independent VALU between MFMAs
#pragma unroll
          for (int q = 0; q < NVALU; ++q) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v2) : "v"(v3)); }
        }
      }
    }
    float s = v0 + v2;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[0] = s;
  } else {
    float v0 = a, v1 = b, v2 = a + 1.f, v3 = b + 1.f;
    int addr = (threadIdx.x & 255) * 16;
    f32x4 l = (f32x4){0.f, 0.f, 0.f, 0.f};
    // noise waves run a fixed multiple of the MFMA wave's iteration count (sized to last about as long)
    for (int it = 0; it < iters; ++it) {
      if (KIND == 1) {              // VALU noise: 8 x NVALU x 2 adds per MFMA-wave iteration
#pragma unroll
        for (int q = 0; q < 8 * NVALU; ++q) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v2) : "v"(v3)); }
      } else if (KIND == 2) {       // LDS noise: NVALU ds_read_b128 per MFMA-wave iteration (8 MFMAs), one wait per batch
        f32x4 t[NVALU];
#pragma unroll
        for (int q = 0; q < NVALU; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[q]) : "v"(addr), "n"(q * 16));
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int q = 0; q < NVALU; ++q) asm volatile("" :: "v"(t[q]));
      } else if (KIND == 4) {       // global loads (L2-resident), NVALU dwordx4 loads per iteration
        f32x4 t[NVALU];
#pragma unroll
        for (int q = 0; q < NVALU; ++q) t[q] = *(const volatile f32x4*)(gin + ((threadIdx.x + q * 512) & 4095) * 4);
#pragma unroll
        for (int q = 0; q < NVALU; ++q) asm volatile("" :: "v"(t[q]));
      } else if (KIND == 3) {       // SALU noise
#pragma unroll
        for (int q = 0; q < 8 * NVALU * 2; ++q) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
      } else if (KIND == 0) {
        break;
      }
    }
    float s = v0 + v2 + l.x;
    if (s == 123.456f) out[1] = s;
  }
}
static float* gbuf;
template <int KIND, int NVALU>
static void run(const char* name, int iters) {
  float* out; hipMalloc(&out, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NVALU>), dim3(256), dim3(512), 0, 0, out, iters, gbuf);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double mf = (double)256 * 4 * iters * 8;
  printf("%-34s nvalu/mfma %2d: %.3f ms  %.1f TF/s (MFMA waves only)  cycles/MFMA at 2.4 GHz %.1f\n", name, NVALU * 2, best, mf * 2048 / best * 1e-9, best * 1e-3 * 2.4e9 / (iters * 8.0));
  hipFree(out);
}
int main() {
  hipMalloc(&gbuf, 4096 * 16 + 65536); hipMemset(gbuf, 0, 4096 * 16 + 65536);
  { float* o; hipMalloc(&o, 8); hipLaunchKernelGGL((k<0, 1>), dim3(2048), dim3(512), 0, 0, o, 20000, gbuf); hipDeviceSynchronize(); }
  const int it = 20000;
  run<0, 1>("mfma only (1 wave/SIMD)", it);
  run<1, 1>("other-wave VALU", it); run<1, 2>("other-wave VALU", it); run<1, 4>("other-wave VALU", it); run<1, 8>("other-wave VALU", it);
  run<2, 1>("other-wave ds_read_b128 x1 /8mfma", it); run<2, 2>("other-wave ds_read_b128 x2 /8mfma", it); run<2, 4>("other-wave ds_read_b128 x4 /8mfma", it); run<2, 8>("other-wave ds_read_b128 x8 /8mfma", it);
  run<4, 1>("other-wave global_load x1 /8mfma", it); run<4, 2>("other-wave global_load x2 /8mfma", it); run<4, 4>("other-wave global_load x4 /8mfma", it);
  run<3, 4>("other-wave SALU", it);
  run<10, 1>("same-wave VALU", it); run<10, 2>("same-wave VALU", it); run<10, 3>("same-wave VALU", it); run<10, 4>("same-wave VALU", it);
  return 0;
}
