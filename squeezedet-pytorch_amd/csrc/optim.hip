// Gradient clipping + SGD with momentum for ALL parameter tensors in one launch (reference: src/engine/trainer.py:47-50,
// `clip_grad_norm_(params, cfg.grad_norm)` + `optimizer.step()` with torch.optim.SGD(lr, momentum, weight_decay)).
// torch runs these as ~10 foreach / elementwise launches over the 64 parameter tensors (about 0.15 ms per step, nothing else
// on the GPU meanwhile); here a descriptor table {param, grad, momentum buffer, elements} per tensor feeds one grid.
// Arithmetic in torch's order, element for element:
//   coef = min(1, max_norm / (total_norm + 1e-6));  g = grad * coef;  g = g + wd * p;  buf = mom * buf + g;  p = p - lr * buf
// (a zero-initialised buffer reproduces torch's first step, where buf = g).  total_norm is read from device memory (the L2
// norm of the flat gradient buffer, computed by the caller), so the step stays capturable in a hipGraph.
#include "sqd_common.h"

#define SQD_NORM_PARTS 256      // partial sums of squares of the flat gradient (sqd_grad_sumsq), summed in index order by every workgroup of the step

struct SgdDesc { float* p; long long g; float* m; long long n; };      // g: element offset into g_base, or an address when g_base is null

__global__ __launch_bounds__(256) void sgd_clip_batched_kernel(const SgdDesc* __restrict__ descs, const float* __restrict__ g_base,
                                                               const float* __restrict__ total_norm, float max_norm, float lr, float momentum,
                                                               float wd, int norm_parts, float* __restrict__ norm_out) {
  const SgdDesc d = descs[blockIdx.y];
  const float* __restrict__ dg = g_base ? g_base + d.g : (const float*)d.g;
  float coef = 1.f;
  if (max_norm > 0.f) {
    float tn;
    if (norm_parts > 0) {                    // total_norm = the partial sums of squares of sqd_grad_sumsq: fixed tree, then the root
      const int lane = threadIdx.x & 63;     // (every wave evaluates it for itself: lane l adds parts l, l + 64, ..., then a shuffle tree)
      float ssum = 0.f;
      for (int i = lane; i < norm_parts; i += 64) ssum += total_norm[i];
      for (int off = 32; off >= 1; off >>= 1) ssum += __shfl_xor(ssum, off);
      tn = sqrtf(ssum);
      if (norm_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *norm_out = tn;
    } else {
      tn = *total_norm;
    }
    coef = max_norm / (tn + 1e-6f);
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

// The same step with the work dealt in equal CHUNKS instead of a fixed number of workgroups per tensor (the 64 tensors of the model span
// 16 to 500 k elements: 64 workgroups each left the large ones 8 serial rounds and the small ones 63 idle workgroups -- 47 us for 42 MB).
// chunks: device array of [nchunks][2] int64 = {tensor, first element}; a workgroup handles SQD_SGD_CHUNK elements of one tensor.
#define SQD_SGD_CHUNK 4096
__global__ __launch_bounds__(256) void sgd_clip_chunked_kernel(const SgdDesc* __restrict__ descs, const long long* __restrict__ chunks,
                                                               const float* __restrict__ g_base, const float* __restrict__ parts, float max_norm,
                                                               float lr, float momentum, float wd, float* __restrict__ norm_out) {
  const SgdDesc d = descs[chunks[2 * blockIdx.x]];
  const long long e0 = chunks[2 * blockIdx.x + 1];
  const float* __restrict__ dg = g_base ? g_base + d.g : (const float*)d.g;
  float coef = 1.f;
  if (max_norm > 0.f) {
    const int lane = threadIdx.x & 63;
    float ssum = 0.f;
    for (int i = lane; i < SQD_NORM_PARTS; i += 64) ssum += parts[i];
    for (int off = 32; off >= 1; off >>= 1) ssum += __shfl_xor(ssum, off);
    const float tn = sqrtf(ssum);
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = tn;
    coef = max_norm / (tn + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  auto one = [&](float p, float gr, float m, float& pn, float& mn) {
    float g = gr * coef;
    g = g + wd * p;
    mn = momentum * m + g;
    pn = p - lr * mn;
  };
  const long long e1 = (e0 + SQD_SGD_CHUNK < d.n) ? e0 + SQD_SGD_CHUNK : d.n;
  const bool vec = ((((uintptr_t)d.p) | ((uintptr_t)dg) | ((uintptr_t)d.m)) & 15) == 0;       // (chunk origins are multiples of 4 elements)
  const long long q0 = e0 >> 2, q1 = vec ? (e1 >> 2) : q0;
  for (long long i = q0 + threadIdx.x; i < q1; i += 256) {
    const f32x4 p = ((const f32x4*)d.p)[i], gr = ((const f32x4*)dg)[i], m = ((const f32x4*)d.m)[i];
    float pn[4], mn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) one(p[e], gr[e], m[e], pn[e], mn[e]);
    ((f32x4*)d.m)[i] = (f32x4){mn[0], mn[1], mn[2], mn[3]}; ((f32x4*)d.p)[i] = (f32x4){pn[0], pn[1], pn[2], pn[3]};
  }
  for (long long i = 4 * q1 + threadIdx.x; i < e1; i += 256) {
    float pn, mn;
    one(d.p[i], dg[i], d.m[i], pn, mn);
    d.m[i] = mn; d.p[i] = pn;
  }
}

extern "C" int sqd_sgd_chunk_elems(void) { return SQD_SGD_CHUNK; }

// descs_dev as for sqd_sgd_clip_step; chunks_dev: [nchunks][2] int64 {tensor index, first element (a multiple of sqd_sgd_chunk_elems())}
// covering every tensor; sumsq_parts / norm_out as for sqd_sgd_clip_step_parts (sumsq_parts may be NULL when max_norm <= 0).
extern "C" int sqd_sgd_clip_step_chunked(const void* descs_dev, const void* chunks_dev, int nchunks, const float* grad_base,
                                         const float* sumsq_parts, float* norm_out, float max_norm, float lr, float momentum,
                                         float weight_decay, void* stream) {
  SQD_CHECK_ARG(descs_dev && chunks_dev && nchunks > 0 && (sumsq_parts || max_norm <= 0.f));
  hipLaunchKernelGGL(sgd_clip_chunked_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const SgdDesc*)descs_dev,
                     (const long long*)chunks_dev, grad_base, sumsq_parts, max_norm, lr, momentum, weight_decay, norm_out);
  return sqd_launch_status();
}

// descs_dev: device array of n records of 4 int64 {param ptr, grad, momentum ptr, elements}; grad = element offset into grad_base (the
// flat gradient buffer the backward writes: the table then never changes, only this one pointer does) or, with grad_base NULL, the
// gradient's address; total_norm: device float (may be NULL when max_norm <= 0 = no clipping).
extern "C" int sqd_sgd_clip_step(const void* descs_dev, int n, const float* grad_base, const float* total_norm, float max_norm, float lr,
                                 float momentum, float weight_decay, int blocks_per_desc, void* stream) {
  SQD_CHECK_ARG(descs_dev && n > 0 && n <= 65535 && blocks_per_desc > 0 && blocks_per_desc <= 4096 && (total_norm || max_norm <= 0.f));
  hipLaunchKernelGGL(sgd_clip_batched_kernel, dim3((unsigned)blocks_per_desc, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     (const SgdDesc*)descs_dev, grad_base, total_norm, max_norm, lr, momentum, weight_decay, 0, (float*)nullptr);
  return sqd_launch_status();
}

// ---- the gradient norm of clip_grad_norm_ (src/engine/trainer.py:49) without a torch reduction kernel ----
// sqd_grad_sumsq: parts[i] = sum of squares of block i of the flat gradient (SQD_NORM_PARTS equal blocks; within a block a fixed
// tree: bitwise reproducible).  sqd_sgd_clip_step_parts: the step above with total_norm = sqrt(sum of those partials in index order),
// evaluated redundantly by every workgroup; norm_out (or NULL) receives the norm for logging.
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ parts) {
  // blocks of whole 16-byte quads (the flat buffer is 16-byte aligned: checked by the launcher); the tail rides in the last block
  const long long nq = n >> 2;
  const long long per = (nq + SQD_NORM_PARTS - 1) / SQD_NORM_PARTS;
  const long long lo = (long long)blockIdx.x * per, hi = (lo + per < nq) ? lo + per : nq;
  float acc = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const f32x4 v = ((const f32x4*)g)[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == SQD_NORM_PARTS - 1 && threadIdx.x < (n & 3)) { const float v = g[4 * nq + threadIdx.x]; acc += v * v; }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}

extern "C" int sqd_grad_sumsq(const float* grad_flat, long long n, float* parts, void* stream) {
  SQD_CHECK_ARG(grad_flat && parts && n > 0 && ((uintptr_t)grad_flat & 15) == 0);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(SQD_NORM_PARTS), dim3(256), 0, (hipStream_t)stream, grad_flat, n, parts);
  return sqd_launch_status();
}

extern "C" int sqd_grad_sumsq_parts(void) { return SQD_NORM_PARTS; }

extern "C" int sqd_sgd_clip_step_parts(const void* descs_dev, int n, const float* grad_base, const float* sumsq_parts, float* norm_out,
                                       float max_norm, float lr, float momentum, float weight_decay, int blocks_per_desc, void* stream) {
  SQD_CHECK_ARG(descs_dev && n > 0 && n <= 65535 && blocks_per_desc > 0 && blocks_per_desc <= 4096 && sumsq_parts && max_norm > 0.f);
  hipLaunchKernelGGL(sgd_clip_batched_kernel, dim3((unsigned)blocks_per_desc, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     (const SgdDesc*)descs_dev, grad_base, sumsq_parts, max_norm, lr, momentum, weight_decay, SQD_NORM_PARTS, norm_out);
  return sqd_launch_status();
}

// ---- the element-wise steps of the data-parallel gradient exchange (trainer.GradientExchange) without a torch kernel ----
// The reference's DataParallel (src/utils/data_parallel.py:93-113) gathers the per-sample losses and differentiates their mean
// (src/engine/trainer.py:43,47); one process per GPU gets the same gradient as sum_r(B_r * g_r) / sum_r(B_r).  g[0..n) = g * mul /
// (*div_by if given); fill_ptr (or NULL) receives fill_value: the rank's image count that rides through the same all-reduce.
__global__ __launch_bounds__(256) void grad_scale_kernel(float* __restrict__ g, long long n, float mul, const float* __restrict__ div_by,
                                                         float* __restrict__ fill_ptr, float fill_value) {
  if (fill_ptr && blockIdx.x == 0 && threadIdx.x == 0) *fill_ptr = fill_value;
  const float d = div_by ? *div_by : 1.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = g[i] * mul;
    g[i] = div_by ? v / d : v;                 // a true division: the same rounding as torch's div_ by the summed count
  }
}

extern "C" int sqd_grad_scale(float* g, long long n, float mul, const float* div_by, float* fill_ptr, float fill_value, void* stream) {
  SQD_CHECK_ARG(n >= 0 && (g || n == 0) && (n > 0 || fill_ptr));
  SQD_CHECK_ARG(!fill_ptr || !g || fill_ptr < g || fill_ptr >= g + n);          // the slot is not part of the scaled range
  long long blocks = (n + 1023) / 1024;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, mul, div_by, fill_ptr, fill_value);
  return sqd_launch_status();
}
