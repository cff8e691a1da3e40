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

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the device that is current at the call: kernels that need more than 64 KB of
// LDS set it once PER DEVICE (a process may launch on several: reference-style DataParallel, tests that switch devices).  Idempotent,
// so two threads racing through the first launch on a device is harmless.
struct SqdDevOnce { unsigned long long done = 0; };
static inline int sqd_max_lds_once(SqdDevOnce& st, const void* kern, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SQD_ERR_LAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&st.done, __ATOMIC_ACQUIRE) & bit) return SQD_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return SQD_ERR_LAUNCH;
  __atomic_fetch_or(&st.done, bit, __ATOMIC_RELEASE);
  return SQD_OK;
}

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

// ---------------------------------------------------------------------------------------------------------------------
// Counter-based dropout in front of ConvDet (reference: nn.Dropout(p, inplace=True), src/model/squeezedet.py:71-72,81-82).
// The keep decision of element e of the dropped tensor is a pure function of (seed, step, e): any kernel that produces the
// element -- the last Fire's expand1x1 and expand3x3 launches, whatever their tiling -- or a stand-alone mask kernel evaluates
// the same bits, and the host can reproduce them (ops.dropout_mask_reference).  One 64-bit hash per group of four consecutive
// elements (an f32x4 of the NHWC tensor), four 16-bit fields, keep where field < keep16 = round((1 - p) * 65536); kept values
// are scaled by 1 / (1 - p).  state = {seed, step} in device memory: `step` is advanced on the device once per forward (the
// captured training step draws a fresh mask every replay).  The backward needs no mask: an element of the dropped ReLU output
// is > 0 exactly where it was kept AND the ReLU was active, so the ConvDet data gradient masks by it and scales by a constant.
// ---------------------------------------------------------------------------------------------------------------------
struct SqdDrop { unsigned k0, k1, keep16; float scale; };

__host__ __device__ __forceinline__ unsigned sqd_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ SqdDrop sqd_drop_key(unsigned long long seed, unsigned long long step, int keep16, float scale) {
  SqdDrop d;
  const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32), t0 = (unsigned)step, t1 = (unsigned)(step >> 32);
  d.k0 = sqd_mix32(s0 ^ sqd_mix32(t0 + 0x9e3779b9u));
  d.k1 = sqd_mix32(s1 ^ sqd_mix32(t0 ^ 0x85ebca6bu) ^ (t1 * 0xc2b2ae35u));
  d.keep16 = (unsigned)keep16; d.scale = scale;
  return d;
}

// multipliers (scale or 0) of the four elements e4 * 4 .. e4 * 4 + 3
__host__ __device__ __forceinline__ f32x4 sqd_drop_mul4(unsigned long long e4, SqdDrop d) {
  const unsigned lo = (unsigned)e4 ^ ((unsigned)(e4 >> 32) * 0x9e3779b9u);
  const unsigned h1 = sqd_mix32(lo ^ d.k0), h2 = sqd_mix32(h1 ^ d.k1);
  f32x4 m;
  m.x = (h1 & 0xffffu) < d.keep16 ? d.scale : 0.f; m.y = (h1 >> 16) < d.keep16 ? d.scale : 0.f;
  m.z = (h2 & 0xffffu) < d.keep16 ? d.scale : 0.f; m.w = (h2 >> 16) < d.keep16 ? d.scale : 0.f;
  return m;
}
