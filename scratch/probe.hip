#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// D[16x16] = A[16xK] * B[Kx16], K multiple of 4; A row-major [16][K], B row-major [K][16]
__global__ void mfma16_probe(const float* A, const float* B, float* D, int K) {
  int l = threadIdx.x; int r = l & 15, g = l >> 4;
  f32x4 acc = {0,0,0,0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    float a = A[r * K + k0 + g];
    float b = B[(k0 + g) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  // C/D: col = lane&15, row = (lane>>4)*4 + reg
  for (int j = 0; j < 4; ++j) D[(g * 4 + j) * 16 + r] = acc[j];
}
extern "C" int probe_mfma(const float* A, const float* B, float* D, int K, void* stream) {
  hipLaunchKernelGGL(mfma16_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D, K);
  return (int)hipGetLastError();
}
extern "C" int probe_rtver() { int v = 0; hipRuntimeGetVersion(&v); return v; }
