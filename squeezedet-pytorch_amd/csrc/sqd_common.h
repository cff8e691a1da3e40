// Shared device/host helpers for the SqueezeDet gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Status codes returned by every C-ABI entry point (0 = ok).  Never abort, never sync.
#define SQD_OK 0
#define SQD_ERR_BAD_ARG 1
#define SQD_ERR_UNSUPPORTED 2
#define SQD_ERR_LAUNCH 3

#define SQD_CHECK_ARG(cond) do { if (!(cond)) return SQD_ERR_BAD_ARG; } while (0)

static inline int sqd_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SQD_OK : SQD_ERR_LAUNCH;
}

static inline int sqd_cdiv(int a, int b) { return (a + b - 1) / b; }

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 (k-ordered fma chain).
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; acc reg r holds D[4*(l>>4)+r][l&15].
// Workgroup id -> position in a persistent kernel's tile walk such that the workgroups of ONE XCD (ids w, w + 8, w + 16, ... share
// an XCD under the round-robin dispatch: MI355X_MICROARCH.md, "Workgroup dispatch") own a CONTIGUOUS run of positions: tiles that are
// neighbours in the walk -- and share halo rows / partial cache lines -- are then fetched through the same 4 MB L2 at about the
// same time instead of once per XCD.  A permutation of [0, G) for any grid size G; placement is a speed matter only.
__device__ __forceinline__ int sqd_xcd_contiguous(int w, int G) {
  const int x = w & 7, q = G >> 3, r = G & 7;
  return x * q + (x < r ? x : r) + (w >> 3);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
