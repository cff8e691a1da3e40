// Multi-task detection loss, forward and analytic backward.
//
// Reference: Loss.forward src/model/squeezedet.py:133-174, compute_overlaps src/model/modules.py:48-63,
// PredictionResolver (log_softmax branch) src/model/squeezedet.py:109-120, deltas_to_boxes
// src/model/modules.py:27-45.  The reference runs ~40 elementwise/reduction launches forward and an
// autograd graph backward; here: one reduction kernel (+ a tiny finalise) forward, one elementwise
// kernel backward.
//
//   class  = sum_a  w_c * mask * sum_c onehot_c * (-log_softmax_c)        / n_obj
//   pos    = sum_a  w_p * mask     * (iou - sigmoid(conf))^2              / n_obj
//   neg    = sum_a  w_n * (1-mask) * (iou - sigmoid(conf))^2              / (A - n_obj)
//   bbox   = sum_a  w_b * mask * sum_j (delta_j - gt_delta_j)^2           / n_obj
//   iou    = IoU(gt_box, decode(delta)) * mask, IoU = inter / (union + 1e-10)   -- NOT detached:
// the positive-score term back-propagates through iou -> predicted box -> clamp -> exp -> deltas
// (SURVEY.md section 8a row L).  All per-image ([B] vectors); n_obj = 0 gives NaN like the reference.
#include "sqd_common.h"
#include <math.h>

#define LOSS_MAX_CLASSES 16
#define LOSS_NPART 16          // partial-sum blocks per image (deterministic two-stage reduction)
#define LOSS_THREADS 256
#define LOSS_EPS 1e-10f

struct LossArgs {
  const float* pred; const float* gt; const float* anchors;
  int B, A, C;
  float wmax, hmax;
  float w_class, w_pos, w_neg, w_bbox;
};

struct AnchorTerms {          // everything both passes need about one anchor
  float mask, conf, iou_raw, e;          // e = iou*mask - conf
  float ce;                              // sum_c onehot_c * (-logp_c)
  float bb;                              // sum_j (delta_j - gt_j)^2
  float prob[LOSS_MAX_CLASSES];
  float onehot_sum;
  // box decode intermediates for the backward
  float w, h, aw, ah;
  float x1u, y1u, x2u, y2u;              // unclamped
  float px1, py1, px2, py2;              // clamped
  float gx1, gy1, gx2, gy2;
  float lr_raw, tb_raw, inter, uni;
};

__device__ __forceinline__ void anchor_terms(const LossArgs& a, const float* __restrict__ p, const float* __restrict__ g,
                                             const float* __restrict__ anc, AnchorTerms& t) {
  const int C = a.C;
  t.mask = g[0];
  t.gx1 = g[1]; t.gy1 = g[2]; t.gx2 = g[3]; t.gy2 = g[4];
  // log-softmax over the class logits
  float m = p[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, p[c]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) { t.prob[c] = expf(p[c] - m); sum += t.prob[c]; }
  const float lse = logf(sum);
  float ce = 0.f, ohs = 0.f;
  for (int c = 0; c < C; ++c) {
    const float oh = g[9 + c];
    ce += oh * (-((p[c] - m) - lse));
    ohs += oh;
    t.prob[c] = t.prob[c] / sum;
  }
  t.ce = ce; t.onehot_sum = ohs;
  t.conf = 1.f / (1.f + expf(-p[C]));
  // box decode (deltas_to_boxes)
  const float* d = p + C + 1;
  const float ax = anc[0], ay = anc[1];
  t.aw = anc[2]; t.ah = anc[3];
  const float cx = ax + t.aw * d[0], cy = ay + t.ah * d[1];
  t.w = t.aw * expf(d[2]); t.h = t.ah * expf(d[3]);
  t.x1u = cx - 0.5f * (t.w - 1.f); t.y1u = cy - 0.5f * (t.h - 1.f);
  t.x2u = cx + 0.5f * (t.w - 1.f); t.y2u = cy + 0.5f * (t.h - 1.f);
  t.px1 = fminf(fmaxf(t.x1u, 0.f), a.wmax); t.py1 = fminf(fmaxf(t.y1u, 0.f), a.hmax);
  t.px2 = fminf(fmaxf(t.x2u, 0.f), a.wmax); t.py2 = fminf(fmaxf(t.y2u, 0.f), a.hmax);
  // IoU(gt, pred)
  t.lr_raw = fminf(t.gx2, t.px2) - fmaxf(t.gx1, t.px1);
  t.tb_raw = fminf(t.gy2, t.py2) - fmaxf(t.gy1, t.py1);
  const float lr = fmaxf(t.lr_raw, 0.f), tb = fmaxf(t.tb_raw, 0.f);
  t.inter = lr * tb;
  t.uni = (t.gx2 - t.gx1) * (t.gy2 - t.gy1) + (t.px2 - t.px1) * (t.py2 - t.py1) - t.inter;
  t.iou_raw = t.inter / (t.uni + LOSS_EPS);
  t.e = t.iou_raw * t.mask - t.conf;
  float bb = 0.f;
  for (int j = 0; j < 4; ++j) { const float df = d[j] - g[5 + j]; bb += df * df; }
  t.bb = bb;
}

// partial[b][blk][5] = (n_obj, S_class, S_pos, S_neg, S_bbox) over the block's anchors
__global__ __launch_bounds__(LOSS_THREADS) void loss_partial_kernel(LossArgs a, float* __restrict__ partial) {
  const int b = blockIdx.y, blk = blockIdx.x;
  const int per = (a.A + LOSS_NPART - 1) / LOSS_NPART;
  const int lo = blk * per, hi = min(a.A, lo + per);
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = lo + threadIdx.x; i < hi; i += LOSS_THREADS) {
    const long long row = (long long)b * a.A + i;
    AnchorTerms t;
    anchor_terms(a, a.pred + row * (a.C + 5), a.gt + row * (a.C + 9), a.anchors + 4 * i, t);
    s[0] += t.mask;
    s[1] += t.mask * t.ce;
    s[2] += t.mask * (t.e * t.e);
    s[3] += (1.f - t.mask) * (t.e * t.e);
    s[4] += t.mask * t.bb;
  }
  __shared__ float red[5][LOSS_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float v = s[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float v = 0.f;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) v += red[threadIdx.x][w];
    partial[((long long)b * LOSS_NPART + blk) * 5 + threadIdx.x] = v;
  }
}

// losses[4][B] = (class, score = pos+neg, bbox, total); nobj[B].  mean4 (or null): the four batch means, loss.mean() of
// src/engine/trainer.py:43 without a torch reduction kernel -- ONE block then walks the images (thread t takes b = t, t + 64, ...:
// a fixed order) and the 64 partial sums meet in a fixed shuffle tree.
__global__ void loss_finalize_kernel(const float* __restrict__ partial, float* __restrict__ losses, float* __restrict__ nobj,
                                     int B, int A, float w_class, float w_pos, float w_neg, float w_bbox, float* __restrict__ mean4) {
  if (mean4) {
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += 64) {
      float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < LOSS_NPART; ++k)
        for (int j = 0; j < 5; ++j) s[j] += partial[((long long)b * LOSS_NPART + k) * 5 + j];
      const float n = s[0];
      const float cls = w_class * s[1] / n, pos = w_pos * s[2] / n, neg = w_neg * s[3] / ((float)A - n), bbx = w_bbox * s[4] / n;
      const float v[4] = {cls, pos + neg, bbx, cls + pos + neg + bbx};
      for (int j = 0; j < 4; ++j) { losses[j * B + b] = v[j]; m[j] += v[j]; }
      nobj[b] = n;
    }
    for (int j = 0; j < 4; ++j) {
      float v = m[j];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
      if (threadIdx.x == 0) mean4[j] = v / (float)B;
    }
    return;
  }
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < LOSS_NPART; ++k)
    for (int j = 0; j < 5; ++j) s[j] += partial[((long long)b * LOSS_NPART + k) * 5 + j];
  const float n = s[0];
  const float cls = w_class * s[1] / n;
  const float pos = w_pos * s[2] / n;
  const float neg = w_neg * s[3] / ((float)A - n);
  const float bbx = w_bbox * s[4] / n;
  losses[0 * B + b] = cls;
  losses[1 * B + b] = pos + neg;
  losses[2 * B + b] = bbx;
  losses[3 * B + b] = cls + pos + neg + bbx;     // same association as the reference (:166)
  nobj[b] = n;
}

// dpred[b][a][:] = u_class[b]*d(class_b) + u_score[b]*d(score_b) + u_bbox[b]*d(bbox_b), coef[3][B]
__global__ __launch_bounds__(LOSS_THREADS) void loss_bwd_kernel(LossArgs a, const float* __restrict__ nobj,
                                                                const float* __restrict__ coef, float* __restrict__ dpred,
                                                                const float* __restrict__ gmean) {
  const long long total = (long long)a.B * a.A;
  const int C = a.C;
  for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < total; row += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(row / a.A), i = (int)(row - (long long)b * a.A);
    const float* p = a.pred + row * (C + 5);
    const float* g = a.gt + row * (C + 9);
    AnchorTerms t;
    anchor_terms(a, p, g, a.anchors + 4 * i, t);
    const float n = nobj[b];
    // upstream gradients: per image and component (coef [3][B]), or (gmean: d / d mean(total)) the same gmean[0] / B for all
    const float gm = gmean ? gmean[0] / (float)a.B : 0.f;
    const float uc = gmean ? gm : coef[0 * a.B + b], us = gmean ? gm : coef[1 * a.B + b], ub = gmean ? gm : coef[2 * a.B + b];
    float* o = dpred + row * (C + 5);
    // class logits: w_c*mask/n * (sum(onehot)*softmax_j - onehot_j)
    const float kc = uc * a.w_class * t.mask / n;
    for (int c = 0; c < C; ++c) o[c] = kc * (t.onehot_sum * t.prob[c] - g[9 + c]);
    // score terms: k*(iou*mask - conf)^2, k = w_p*mask/n + w_n*(1-mask)/(A-n)
    const float k = us * (a.w_pos * t.mask / n + a.w_neg * (1.f - t.mask) / ((float)a.A - n));
    const float dL_de = 2.f * k * t.e;
    o[C] = -dL_de * t.conf * (1.f - t.conf);
    // IoU path: d e / d iou_raw = mask
    const float dL_dov = dL_de * t.mask;
    float gd[4] = {0.f, 0.f, 0.f, 0.f};
    if (dL_dov != 0.f) {
      const float den = t.uni + LOSS_EPS;
      const float dov_dinter = 1.f / den + t.inter / (den * den);     // union contains -inter
      const float dov_dap = -t.inter / (den * den);                   // pred-box area
      const float lr = fmaxf(t.lr_raw, 0.f), tb = fmaxf(t.tb_raw, 0.f);
      const float dlr = (t.lr_raw >= 0.f) ? dL_dov * dov_dinter * tb : 0.f;
      const float dtb = (t.tb_raw >= 0.f) ? dL_dov * dov_dinter * lr : 0.f;
      // min/max sub-gradients: ties split evenly (torch.minimum/maximum backward)
      auto wmin = [](float mine, float other) { return mine < other ? 1.f : (mine == other ? 0.5f : 0.f); };
      auto wmax = [](float mine, float other) { return mine > other ? 1.f : (mine == other ? 0.5f : 0.f); };
      const float pw = t.px2 - t.px1, ph = t.py2 - t.py1;
      const float dap = dL_dov * dov_dap;
      float dpx2 = dlr * wmin(t.px2, t.gx2) + dap * ph;
      float dpx1 = -dlr * wmax(t.px1, t.gx1) - dap * ph;
      float dpy2 = dtb * wmin(t.py2, t.gy2) + dap * pw;
      float dpy1 = -dtb * wmax(t.py1, t.gy1) - dap * pw;
      // clamp backward: passes where the unclamped value is inside [0, max] (inclusive)
      if (!(t.x1u >= 0.f && t.x1u <= a.wmax)) dpx1 = 0.f;
      if (!(t.x2u >= 0.f && t.x2u <= a.wmax)) dpx2 = 0.f;
      if (!(t.y1u >= 0.f && t.y1u <= a.hmax)) dpy1 = 0.f;
      if (!(t.y2u >= 0.f && t.y2u <= a.hmax)) dpy2 = 0.f;
      gd[0] = (dpx1 + dpx2) * t.aw;                    // d x{1,2}u / d dx = aw
      gd[1] = (dpy1 + dpy2) * t.ah;
      gd[2] = (dpx2 - dpx1) * 0.5f * t.w;              // d x2u/d dw = +w/2, d x1u/d dw = -w/2
      gd[3] = (dpy2 - dpy1) * 0.5f * t.h;
    }
    const float kb = ub * a.w_bbox * t.mask / n * 2.f;
    const float* d = p + C + 1;
    for (int j = 0; j < 4; ++j) o[C + 1 + j] = gd[j] + kb * (d[j] - g[5 + j]);
  }
}

static int fill_args(LossArgs& a, const float* pred, const float* gt, const float* anchors, int B, int A, int C,
                     int input_h, int input_w, float w_class, float w_pos, float w_neg, float w_bbox) {
  SQD_CHECK_ARG(pred && gt && anchors && B > 0 && A > 0 && C >= 1 && C <= LOSS_MAX_CLASSES);
  a.pred = pred; a.gt = gt; a.anchors = anchors; a.B = B; a.A = A; a.C = C;
  a.wmax = (float)(input_w - 1); a.hmax = (float)(input_h - 1);
  a.w_class = w_class; a.w_pos = w_pos; a.w_neg = w_neg; a.w_bbox = w_bbox;
  return SQD_OK;
}

// workspace: float[B * 16 * 5]; losses: float[4][B] = (class, score, bbox, total); nobj: float[B]
extern "C" int sqd_loss_fwd(const float* pred, const float* gt, const float* anchors, float* workspace, float* losses,
                            float* nobj, int B, int A, int num_classes, int input_h, int input_w, float w_class,
                            float w_pos, float w_neg, float w_bbox, void* stream) {
  LossArgs a;
  if (int rc = fill_args(a, pred, gt, anchors, B, A, num_classes, input_h, input_w, w_class, w_pos, w_neg, w_bbox)) return rc;
  SQD_CHECK_ARG(workspace && losses && nobj);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_NPART, (unsigned)B), dim3(LOSS_THREADS), 0, s, a, workspace);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3((unsigned)sqd_cdiv(B, 64)), dim3(64), 0, s, workspace, losses, nobj, B, A,
                     w_class, w_pos, w_neg, w_bbox, (float*)nullptr);
  return sqd_launch_status();
}

// The same + mean4 [4] = the batch means of (class, score, bbox, total): `loss.mean()` of the training step
// (src/engine/trainer.py:43) computed by the finalize launch itself.
extern "C" int sqd_loss_mean_fwd(const float* pred, const float* gt, const float* anchors, float* workspace, float* losses,
                                 float* nobj, float* mean4, int B, int A, int num_classes, int input_h, int input_w, float w_class,
                                 float w_pos, float w_neg, float w_bbox, void* stream) {
  LossArgs a;
  if (int rc = fill_args(a, pred, gt, anchors, B, A, num_classes, input_h, input_w, w_class, w_pos, w_neg, w_bbox)) return rc;
  SQD_CHECK_ARG(workspace && losses && nobj && mean4);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_NPART, (unsigned)B), dim3(LOSS_THREADS), 0, s, a, workspace);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, workspace, losses, nobj, B, A, w_class, w_pos, w_neg, w_bbox, mean4);
  return sqd_launch_status();
}

// coef: float[3][B] upstream gradients of (class, score, bbox) per image (total's gradient already added to each)
extern "C" int sqd_loss_bwd(const float* pred, const float* gt, const float* anchors, const float* nobj, const float* coef,
                            float* dpred, int B, int A, int num_classes, int input_h, int input_w, float w_class,
                            float w_pos, float w_neg, float w_bbox, void* stream) {
  LossArgs a;
  if (int rc = fill_args(a, pred, gt, anchors, B, A, num_classes, input_h, input_w, w_class, w_pos, w_neg, w_bbox)) return rc;
  SQD_CHECK_ARG(nobj && coef && dpred);
  const long long total = (long long)B * A;
  const int blocks = (int)((total + LOSS_THREADS - 1) / LOSS_THREADS);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a, nobj, coef, dpred, (const float*)nullptr);
  return sqd_launch_status();
}

// Backward of mean(total): gmean = DEVICE float, the gradient arriving at the mean (1 for `loss.mean().backward()`); every image's
// three components get gmean / B.
extern "C" int sqd_loss_mean_bwd(const float* pred, const float* gt, const float* anchors, const float* nobj, const float* gmean,
                                 float* dpred, int B, int A, int num_classes, int input_h, int input_w, float w_class,
                                 float w_pos, float w_neg, float w_bbox, void* stream) {
  LossArgs a;
  if (int rc = fill_args(a, pred, gt, anchors, B, A, num_classes, input_h, input_w, w_class, w_pos, w_neg, w_bbox)) return rc;
  SQD_CHECK_ARG(nobj && gmean && dpred);
  const long long total = (long long)B * A;
  const int blocks = (int)((total + LOSS_THREADS - 1) / LOSS_THREADS);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a, nobj, (const float*)nullptr, dpred, gmean);
  return sqd_launch_status();
}
