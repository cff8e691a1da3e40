// What staging P x 1 KB per stage costs the ISSUING wave of an MFMA loop (round 4: the Winograd kernels' stages are 64 MFMAs + 8 LDS-DMA
// pieces per wave): per iteration M MFMAs (independent accumulators) and P staged pieces, the pieces issued between the MFMAs (one
// piece every M / P MFMAs), retired one iteration later.  Forms: 0 = MFMAs only; 1 = LDS-DMA (buffer_load_dwordx4 ... lds);
// 2 = buffer_load_dwordx4 into registers + ds_write_b128 one iteration later; 3 = like 1 but all P pieces in one burst at the top.
// WGS workgroups of 4 waves per CU (1 or 2), source = an L2-resident 8 MB window, every wave its own 1 KB pieces.
// build: hipcc -O3 --offload-arch=gfx950 stage_issue.hip -o stage_issue ; run: ./stage_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int FORM, int M, int P, int PAT = 0, int NLDS = 0>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* clk, int iters, const float* gin, int window_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* const my = lds + wave * (2 * P * 256);                 // two stages of P x 1 KB
  const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc((void*)gin, 0, 0x7ffffff0, 0x00020000);
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 st[FORM == 2 ? P : 1];
  for (int i = 0; i < (FORM == 2 ? P : 1); ++i) st[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned base = ((blockIdx.x * 4 + wave) * 64 * 1024) % (unsigned)(window_floats * 4 - 64 * 1024 * 4);
  // PAT 0: 1 KB contiguous per piece; PAT 1: four 256-byte runs 3 KB apart (a 64-channel window of a 768-channel NHWC tensor);
  // PAT 2: sixteen 64-byte runs 512 B apart (a 16-channel window of a 128-channel tensor)
  const int voff = PAT == 0 ? lane * 16 : (PAT == 1 ? (lane >> 4) * 3072 + (lane & 15) * 16 : (lane >> 2) * 512 + (lane & 3) * 16);
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    const unsigned soff = base + (unsigned)((it * P * 1024) % (48 * 1024));
    if (FORM == 1 || FORM == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // last iteration's pieces have landed
    if (FORM == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < P; ++p) *(f32x4*)(my + (buf * P + p) * 256 + lane * 4) = st[p];
    }
    if (FORM == 3) {
#pragma unroll
      for (int p = 0; p < P; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr_t)(my + (buf * P + p) * 256), 16, voff, (int)(soff + p * 1024), 0, 0);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (m % (M / P) == 0) {
        const int p = m / (M / P);
        if (FORM == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr_t)(my + (buf * P + p) * 256), 16, voff, (int)(soff + p * 1024), 0, 0);
        if (FORM == 2) st[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, voff, (int)(soff + p * 1024), 0));
      }
      float bb = (FORM != 0 && (m & 15) == 0) ? my[((buf ^ 1) * P) * 256 + lane] : b;      // (the staged data is consumed: one LDS read per 16 MFMAs)
      if (NLDS > 0 && (m % (M / NLDS)) == 1) bb += my[((buf ^ 1) * P) * 256 + ((lane + m) & 255)];      // extra ds_read_b32 traffic: NLDS per iteration
      acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc[m & 15], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < (FORM == 2 ? P : 1); ++i) s += st[i].x;
  if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
  if (s == 123.456f) out[0] = s;
}

static float* gin; static const int WINDOW = 2 * 1024 * 1024;   // floats (8 MB)
template <int FORM, int M, int P, int PAT = 0, int NLDS = 0>
static double run(const char* name, int wgs_per_cu, int iters) {
  float* out; long long* clk; hipMalloc(&out, 8); const int nwg = 256 * wgs_per_cu; hipMalloc(&clk, nwg * 4 * 8);
  const size_t lds = (size_t)4 * 2 * P * 1024;
  auto kern = k<FORM, M, P, PAT, NLDS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  double best = 1e30; long long lo = 0, hi = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, out, clk, iters, gin, WINDOW);
    hipDeviceSynchronize();
    std::vector<long long> h(nwg * 4); hipMemcpy(h.data(), clk, nwg * 4 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= h.size();
    if (avg < best) { best = avg; lo = *std::min_element(h.begin(), h.end()); hi = *std::max_element(h.begin(), h.end()); }
  }
  const double ns_it = best * 10.0 / iters;     // wall_clock64: 100 MHz
  printf("%-44s M=%3d P=%d wgs/CU=%d: %8.1f ns per iteration = %6.0f cycles at 2.4 GHz (MFMA-only ideal %d x %d = %d)  [per wave min %.1f max %.1f ns]\n", name, M, P, wgs_per_cu, ns_it,
         ns_it * 2.4, M, 32 * wgs_per_cu, M * 32 * wgs_per_cu, lo * 10.0 / iters, hi * 10.0 / iters);
  hipFree(out); hipFree(clk);
  return ns_it;
}
int main() {
  hipMalloc(&gin, (size_t)WINDOW * 4 + (1 << 20)); hipMemset(gin, 0, (size_t)WINDOW * 4 + (1 << 20));
  const int it = 4000;
  for (int w = 1; w <= 2; ++w) {
    run<0, 64, 8>("MFMAs only", w, it);
    run<1, 64, 8>("LDS-DMA pieces spread over the MFMAs", w, it);
    run<3, 64, 8>("LDS-DMA pieces in one burst", w, it);
    run<2, 64, 8>("buffer_load to registers + ds_write_b128", w, it);
    run<1, 32, 4>("LDS-DMA spread, short stage", w, it);
    run<2, 32, 4>("register staging, short stage", w, it);
    run<1, 32, 8>("LDS-DMA spread, 8 pieces per 32 MFMAs", w, it);
    run<2, 32, 8>("register staging, 8 pieces per 32 MFMAs", w, it);
    run<1, 64, 8, 1>("LDS-DMA spread, 4 x 256 B runs per piece", w, it);
    run<1, 64, 8, 2>("LDS-DMA spread, 16 x 64 B runs per piece", w, it);
    run<1, 64, 8, 0, 32>("LDS-DMA spread + 32 ds_read_b32", w, it);
    run<1, 64, 8, 1, 64>("LDS-DMA 4 x 256 B + 64 ds_read_b32", w, it);
    run<3, 64, 8, 1, 64>("LDS-DMA burst 4 x 256 B + 64 ds_read_b32", w, it);
  }
  return 0;
}
