// Stand-alone forms of the counter-based dropout of sqd_common.h (reference: nn.Dropout in front of ConvDet,
// src/model/squeezedet.py:71-72,81-82): the scaled keep mask as a tensor -- for layer configurations whose kernels have no fused
// dropout epilogue, and as the device-side witness the fused epilogues are tested against -- and the per-forward advance of the
// step counter when no kernel of the forward carries it.
#include "sqd_common.h"

__global__ __launch_bounds__(256) void dropout_mask_kernel(const unsigned long long* __restrict__ state, int keep16, float scale,
                                                          f32x4* __restrict__ mask, long long n4) {
  const SqdDrop d = sqd_drop_key(state[0], state[1], keep16, scale);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    mask[i] = sqd_drop_mul4((unsigned long long)i, d);
}

__global__ void dropout_advance_kernel(unsigned long long* state) { state[1] += 1ull; }

// mask [4 * n4] fp32 = scale where element e is kept else 0, for the mask of (state[0] = seed, state[1] = step).
extern "C" int sqd_dropout_mask_fwd(const unsigned long long* state, int keep16, float scale, float* mask, long long n4, void* stream) {
  SQD_CHECK_ARG(state && mask && n4 > 0 && keep16 >= 0 && keep16 <= 65536 && ((uintptr_t)mask & 15) == 0);
  const long long blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, state, keep16, scale,
                     (f32x4*)mask, n4);
  return sqd_launch_status();
}

// state[1] += 1 (one forward consumed its mask).
extern "C" int sqd_dropout_advance(unsigned long long* state, void* stream) {
  SQD_CHECK_ARG(state);
  hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  return sqd_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// A kernel that does nothing for `us` microseconds (one wave polling the constant-rate wall clock).  Used by the lane executor
// (lanes.py) to find out, once, which of its HIP streams share a hardware queue: work on two streams that alias one queue is
// serialised by the queue's in-order barrier packets, which costs the inference driver (reference: src/engine/detector.py:52-85)
// its copy / compute overlap.  Every wave leaves after at most 20 ms.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int sqd_spin_us(int us, void* stream) {
  SQD_CHECK_ARG(us >= 0 && us <= 20000);
  int dev = 0, khz = 100000;                                         // wall_clock64 ticks at a constant 100 MHz on gfx9
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, dev) == hipSuccess && v > 0) khz = v;
  }
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * khz / 1000);
  return sqd_launch_status();
}
