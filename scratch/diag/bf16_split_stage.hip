// VERDICT round 4, item 6 (experiment, report only): issue-rate microbenchmark of ONE Winograd stage of a (group, 16-channel block) unit over
// 32 input channels -- 16 transform positions x [16 tiles] x [16 out channels] x [K = 32] -- in two forms, operands read from LDS with the
// access pattern of conv_wino_vs (lane-contiguous ds_read_b128):
//   f32   : 16 positions x 8 v_mfma_f32_16x16x4_f32            = 128 MFMAs (the shipped arithmetic),  2 x b128 per 4 MFMAs
//   bf16xP: 16 positions x P v_mfma_f32_16x16x32_bf16, P = 9 / 6 / 3 piece products of the three-way bf16 split (weights split at pack
//           time, V split once per group by the transforming wave: neither is part of this loop), 2 x P/3.. b128 per position
// Prints cycles per stage per SIMD at 1, 2, 3 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -o bf16_split_stage bf16_split_stage.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int P>   // P = 0: f32
__global__ __launch_bounds__(768) void stage_kernel(float* out, long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* base = lds + (wv & 3) * 2048 + lane * 4;
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (P == 0) {
#pragma unroll
      for (int cq = 0; cq < 4; ++cq)                 // four 8-channel chunks: per chunk 8 position pairs x (U b128, V b128) -> 4 MFMAs
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
          const f32x4 u = *(const f32x4*)(base + ((cq * 8 + pp) & 15) * 256), v = *(const f32x4*)(base + 8192 + ((cq * 8 + pp) & 7) * 256);
          acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v.x, acc[2 * pp], 0, 0, 0);
          acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v.z, acc[2 * pp + 1], 0, 0, 0);
          acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v.y, acc[2 * pp], 0, 0, 0);
          acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v.w, acc[2 * pp + 1], 0, 0, 0);
        }
    } else {
      constexpr int NP = P == 9 ? 3 : (P == 6 ? 3 : 2);            // pieces of each operand that are read
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        bf16x8 a[3], b[3];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          a[i] = *(const bf16x8*)(base + ((p * 3 + i) & 15) * 256);
          b[i] = *(const bf16x8*)(base + 8192 + ((p * 3 + i) & 7) * 256);
        }
#pragma unroll
        for (int i = NP - 1; i >= 0; --i)
#pragma unroll
          for (int j = NP - 1; j >= 0; --j) {
            if (P == 6 && i + j > 2) continue;       // drop a2b2, a2b1, a1b2
            if (P == 3 && i + j > 1) continue;       // keep a0b0, a0b1, a1b0
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[p], 0, 0, 0);
          }
      }
    }
  }
  const long long c1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < 16; ++p) s += acc[p].x + acc[p].y + acc[p].z + acc[p].w;
  if (s == 123.456f) out[0] = s;
  if (lane == 0) clk[blockIdx.x * 12 + wv] = c1 - c0;
}

template <int P>
static void run(const char* name, int waves) {
  float* out; long long* clk;
  const int blocks = 256, iters = 2000;
  hipMalloc(&out, 4); hipMalloc(&clk, blocks * 12 * 8);
  auto k = stage_kernel<P>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 64 * 1024, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(blocks * 12);
  hipMemcpy(h.data(), clk, blocks * 12 * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) cyc += h[b * 12 + w];
  cyc /= (double)blocks * waves;
  const double per_simd = waves / 4.0;                              // waves per SIMD
  // a stage of the f32 form = 128 x 32.3 cycles = 4134 MFMA cycles per wave
  printf("%-10s %2d waves/CU (%.2g per SIMD): %8.1f cycles per stage per wave, %8.1f per stage per SIMD-slot, %.3f ms\n", name, waves, per_simd,
         cyc / iters, cyc / iters / per_simd, ms);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int waves : {4, 8, 12}) {
    run<0>("f32", waves); run<9>("bf16 x 9", waves); run<6>("bf16 x 6", waves); run<3>("bf16 x 3", waves);
  }
  return 0;
}
