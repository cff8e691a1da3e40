// Gradient clipping + SGD with momentum for ALL parameter tensors in one launch (reference: src/engine/trainer.py:47-50,
// `clip_grad_norm_(params, cfg.grad_norm)` + `optimizer.step()` with torch.optim.SGD(lr, momentum, weight_decay)).
// torch runs these as ~10 foreach / elementwise launches over the 64 parameter tensors (about 0.15 ms per step, nothing else
// on the GPU meanwhile); here a descriptor table {param, grad, momentum buffer, elements} per tensor feeds one grid.
// Arithmetic in torch's order, element for element:
//   coef = min(1, max_norm / (total_norm + 1e-6));  g = grad * coef;  g = g + wd * p;  buf = mom * buf + g;  p = p - lr * buf
// (a zero-initialised buffer reproduces torch's first step, where buf = g).  total_norm is read from device memory (the L2
// norm of the flat gradient buffer, computed by the caller), so the step stays capturable in a hipGraph.
#include "sqd_common.h"

struct SgdDesc { float* p; long long g; float* m; long long n; };      // g: element offset into g_base, or an address when g_base is null

__global__ __launch_bounds__(256) void sgd_clip_batched_kernel(const SgdDesc* __restrict__ descs, const float* __restrict__ g_base,
                                                               const float* __restrict__ total_norm, float max_norm, float lr, float momentum,
                                                               float wd) {
  const SgdDesc d = descs[blockIdx.y];
  const float* __restrict__ dg = g_base ? g_base + d.g : (const float*)d.g;
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (*total_norm + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  auto one = [&](float p, float gr, float m, float& pn, float& mn) {
    float g = gr * coef;
    g = g + wd * p;
    mn = momentum * m + g;
    pn = p - lr * mn;
  };
  // 16-byte accesses where the three pointers allow it (they do for every tensor of the model), scalar tail / fallback
  const bool vec = ((((uintptr_t)d.p) | ((uintptr_t)dg) | ((uintptr_t)d.m)) & 15) == 0;
  const long long n4 = vec ? d.n >> 2 : 0;
  for (long long i = tid; i < n4; i += stride) {
    const f32x4 p = ((const f32x4*)d.p)[i], gr = ((const f32x4*)dg)[i], m = ((const f32x4*)d.m)[i];
    float pn[4], mn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) one(p[e], gr[e], m[e], pn[e], mn[e]);
    ((f32x4*)d.m)[i] = (f32x4){mn[0], mn[1], mn[2], mn[3]}; ((f32x4*)d.p)[i] = (f32x4){pn[0], pn[1], pn[2], pn[3]};
  }
  for (long long i = 4 * n4 + tid; i < d.n; i += stride) {
    float pn, mn;
    one(d.p[i], dg[i], d.m[i], pn, mn);
    d.m[i] = mn; d.p[i] = pn;
  }
}

// descs_dev: device array of n records of 4 int64 {param ptr, grad, momentum ptr, elements}; grad = element offset into grad_base (the
// flat gradient buffer the backward writes: the table then never changes, only this one pointer does) or, with grad_base NULL, the
// gradient's address; total_norm: device float (may be NULL when max_norm <= 0 = no clipping).
extern "C" int sqd_sgd_clip_step(const void* descs_dev, int n, const float* grad_base, const float* total_norm, float max_norm, float lr,
                                 float momentum, float weight_decay, int blocks_per_desc, void* stream) {
  SQD_CHECK_ARG(descs_dev && n > 0 && n <= 65535 && blocks_per_desc > 0 && blocks_per_desc <= 4096 && (total_norm || max_norm <= 0.f));
  hipLaunchKernelGGL(sgd_clip_batched_kernel, dim3((unsigned)blocks_per_desc, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     (const SgdDesc*)descs_dev, grad_base, total_norm, max_norm, lr, momentum, weight_decay);
  return sqd_launch_status();
}
