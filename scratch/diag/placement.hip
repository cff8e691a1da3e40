// Where do the workgroups of a "two per CU" persistent launch actually run?  Every wave records (XCC id, HW_ID, start, end) around an
// MFMA loop; the host counts, per (XCC, SE, CU), how many workgroups were resident at the same time.
// build: hipcc -O3 --offload-arch=gfx950 placement.hip -o placement ; run: ./placement [lds_kb] [wgs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(float* out, long long* rec, int iters) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds[threadIdx.x] = 1.f;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x * 1e-3f, b = lds[lane];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i].x;
  const long long t1 = wall_clock64();
  if (lane == 0) { long long* r = rec + ((long long)blockIdx.x * 4 + wave) * 4; r[0] = xcc; r[1] = hw; r[2] = t0; r[3] = t1; }
  if (s == 123.456f) out[0] = s;
}
int main(int argc, char** argv) {
  const int lds_kb = argc > 1 ? atoi(argv[1]) : 80, nwg = argc > 2 ? atoi(argv[2]) : 512, iters = 20000;
  float* out; long long* rec; hipMalloc(&out, 8); hipMalloc(&rec, (size_t)nwg * 4 * 4 * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(nwg), dim3(256), (size_t)lds_kb * 1024, 0, out, rec, iters); hipDeviceSynchronize(); }
  std::vector<long long> h((size_t)nwg * 16); hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
  // per CU key = (xcc, se, sh, cu): list of (t0, t1) of wave 0 of every workgroup
  std::map<long long, std::vector<std::pair<long long, long long>>> cus;
  long long tmin = h[2], tmax = h[3];
  double dur = 0;
  for (int w = 0; w < nwg; ++w) {
    const long long xcc = h[(size_t)w * 16 + 0] & 0xf, hw = h[(size_t)w * 16 + 1];
    const long long cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back({h[(size_t)w * 16 + 2], h[(size_t)w * 16 + 3]});
    tmin = std::min(tmin, h[(size_t)w * 16 + 2]); tmax = std::max(tmax, h[(size_t)w * 16 + 3]);
    dur += (h[(size_t)w * 16 + 3] - h[(size_t)w * 16 + 2]) * 0.01;
  }
  std::map<int, int> hist, conc;
  for (auto& kv : cus) {
    hist[(int)kv.second.size()]++;
    // maximum number of workgroups resident together on this CU
    int best = 0;
    for (auto& a : kv.second) { int c = 0; for (auto& b : kv.second) if (b.first <= a.first && a.first < b.second) ++c; best = std::max(best, c); }
    conc[best]++;
  }
  printf("LDS %d KB per workgroup, %d workgroups of 4 waves: %zu distinct CUs used; launch span %.1f us, mean workgroup time %.1f us\n", lds_kb, nwg, cus.size(),
         (tmax - tmin) * 0.01, dur / nwg);
  for (auto& kv : hist) printf("   CUs that ran %d workgroups over the launch: %d\n", kv.first, kv.second);
  for (auto& kv : conc) printf("   CUs whose maximum number of co-resident workgroups was %d: %d\n", kv.first, kv.second);
  return 0;
}
