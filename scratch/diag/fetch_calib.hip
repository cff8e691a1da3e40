// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this library uses (MI355X_MICROARCH.md says the
// counter reads 1/2 for 16 B/lane streams and that "other access widths are uncalibrated").  Each kernel reads a known
// byte count once (1 GiB: four times the Infinity Cache) -- run under `rocprofv3 --pmc FETCH_SIZE` and compare.
//   k16_global : 16 B/lane global_load_dwordx4
//   k4_global  : 4 B/lane global_load_dword
//   k16_dma    : 16 B/lane buffer_load_dwordx4 ... lds   (conv / Winograd staging)
//   k4_dma     : 4 B/lane buffer_load_dword ... lds      (the fused stem's NCHW patch staging)
//   k16_reread : 16 B/lane, every workgroup reads the SAME 64 MiB window 16 times (L2 / Infinity Cache served)
// build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(256) void k16_global(const f32x4* __restrict__ p, long long n16, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k4_global(const float* __restrict__ p, long long n4, float* out) {
  float acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) acc += p[i];
  if (acc == 123.456f) out[0] = acc;
}
template <int BYTES>
__global__ __launch_bounds__(256) void k_dma(const float* __restrict__ p, long long nbytes, float* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float buf[256 * 4];
  const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffff0, 0x00020000);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long per_wg = 256ll * BYTES;
  float acc = 0;
  // the resource covers < 2 GiB: rebase every 1 GiB is not needed for this 1 GiB test
  for (long long off = (long long)blockIdx.x * per_wg; off < nbytes; off += (long long)gridDim.x * per_wg) {
    if constexpr (BYTES == 16)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr_t)(buf + wave * 64 * 4), 16, lane * 16, (int)(off + wave * 64 * 16), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr_t)(buf + wave * 64), 4, lane * 4, (int)(off + wave * 64 * 4), 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += buf[(wave * 64 + lane) * (BYTES / 4)];
  }
  if (acc == 123.456f) out[0] = acc;
#endif
}
__global__ __launch_bounds__(256) void k16_reread(const f32x4* __restrict__ p, long long n16_window, int reps, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16_window; i += (long long)gridDim.x * 256) acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

int main() {
  const long long nbytes = 1ll << 30;
  float* p; float* out;
  if (hipMalloc(&p, nbytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(p, 0, nbytes);
  hipDeviceSynchronize();
  const int grid = 256 * 8;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k16_global, dim3(grid), dim3(256), 0, 0, (const f32x4*)p, nbytes / 16, out);
    hipLaunchKernelGGL(k4_global, dim3(grid), dim3(256), 0, 0, p, nbytes / 4, out);
    hipLaunchKernelGGL(k_dma<16>, dim3(grid), dim3(256), 0, 0, p, nbytes, out);
    hipLaunchKernelGGL(k_dma<4>, dim3(grid), dim3(256), 0, 0, p, nbytes, out);
    hipLaunchKernelGGL(k16_reread, dim3(grid), dim3(256), 0, 0, (const f32x4*)p, (64ll << 20) / 16, 16, out);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("fetch_calib: every kernel reads %lld bytes once (k16_reread: 16 x 64 MiB = the same bytes from a cache-resident window); status %d\n", nbytes, (int)e);
  return e == hipSuccess ? 0 : 2;
}
