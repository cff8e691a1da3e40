// On-device GT encoding: greedy unique anchor assignment + regression targets +
// the dense [A][C+9] gt tensor, one workgroup per image.
//
// Reference (CPU, Python double loop with an argsort over all anchors per box, inside DataLoader workers):
// compute_deltas src/utils/boxes.py:84-135 (compute_overlaps :70-81, xyxy_to_xywh :12-23, xywh_to_xyxy :26-34),
// BaseDataset.prepare_annotations src/datasets/base.py:61-76.
//
// Arithmetic mirrors numpy's promotion in the reference exactly: boxes are float32 (KITTI.load_annotations +
// resize, float32 in place), anchors float64 (generate_anchors).  So
//   boxes_xywh   = float32 arithmetic ((x1+x2)/2, x2-x1+1)                                   boxes.py:17-22
//   anchors_xyxy = float64                                                                    boxes.py:29-34
//   overlaps     = float64, except the box area (box[2]-box[0])*(box[3]-box[1]) which is a float32 scalar product
//   dist         = float64 sum over (cx,cy,w,h) of squares, left to right (numpy's order for 4 addends)
//   deltas       = float64 expression, rounded to float32 when stored (np.array(..., dtype=float32))
// Built with -ffp-contract=off, so no fma contraction changes a result bit.
//
// Selection rule: for box i (in input order) the free anchor with the largest overlap if that overlap is > 0, else
// the free anchor with the smallest distance; an anchor is taken once.  Ties -> lowest anchor index (the reference
// leaves ties to numpy's unstable argsort; see tests/golden/make_golden_gt.py).
#include "sqd_common.h"

struct GtArgs {
  const float* boxes;        // [total][4] xyxy float32
  const int* class_ids;      // [total]
  const int* box_offsets;    // [B+1]
  const double* anchors;     // [A][4] cx,cy,w,h float64
  float* gt;                 // [B][A][C+9] or null
  int* anchor_idx;           // [total] or null
  float* deltas;             // [total][4] or null
  int B, A, C;
  const void* cand;          // GtCand [total] from gt_candidates_kernel, or null (every box scans with the taken mask)
};

struct Cand { double ov; double dist; int ov_idx; int dist_idx; };

__device__ __forceinline__ void cand_merge(Cand& c, double ov, int oi, double d, int di) {
  if (ov > c.ov || (ov == c.ov && oi < c.ov_idx)) { c.ov = ov; c.ov_idx = oi; }
  if (d < c.dist || (d == c.dist && di < c.dist_idx)) { c.dist = d; c.dist_idx = di; }
}

constexpr int GT_THREADS = 1024;

// First-choice pass (fully parallel, one workgroup per box): best overlap / nearest anchor over ALL anchors, ignoring the
// taken set.  The serial pass below accepts a box's first choice whenever that anchor is still free (masked arg-max =
// unmasked arg-max then) and only re-scans with the mask on a conflict, which is rare in real annotations.
struct GtCand { double ov; int ov_idx; int dist_idx; };

__global__ __launch_bounds__(256) void gt_candidates_kernel(const float* __restrict__ boxes, const double* __restrict__ anchors,
                                                            GtCand* __restrict__ cand, int A) {
  __shared__ Cand wave_c[4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float b0 = boxes[4 * i], b1 = boxes[4 * i + 1], b2 = boxes[4 * i + 2], b3 = boxes[4 * i + 3];
  const float bcx = (b0 + b2) / 2.f, bcy = (b1 + b3) / 2.f, bw = b2 - b0 + 1.f, bh = b3 - b1 + 1.f;
  const double barea = (double)((b2 - b0) * (b3 - b1));
  Cand c; c.ov = -1.0; c.ov_idx = 0x7fffffff; c.dist = __builtin_inf(); c.dist_idx = 0x7fffffff;
  for (int j = tid; j < A; j += 256) {
    const double ax = anchors[4 * j], ay = anchors[4 * j + 1], aw = anchors[4 * j + 2], ah = anchors[4 * j + 3];
    const double x0 = ax - 0.5 * (aw - 1.0), y0 = ay - 0.5 * (ah - 1.0);
    const double x1 = ax + 0.5 * (aw - 1.0), y1 = ay + 0.5 * (ah - 1.0);
    const double lr = fmax(fmin(x1, (double)b2) - fmax(x0, (double)b0), 0.0);
    const double tb = fmax(fmin(y1, (double)b3) - fmax(y0, (double)b1), 0.0);
    const double inter = lr * tb;
    const double uni = (x1 - x0) * (y1 - y0) + barea - inter;
    const double ov = inter / (uni + 1e-10);
    const double d0 = (double)bcx - ax, d1 = (double)bcy - ay, d2 = (double)bw - aw, d3 = (double)bh - ah;
    const double dist = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
    cand_merge(c, ov, j, dist, j);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(c.ov, off), d = __shfl_xor(c.dist, off);
    const int oi = __shfl_xor(c.ov_idx, off), di = __shfl_xor(c.dist_idx, off);
    cand_merge(c, ov, oi, d, di);
  }
  if (lane == 0) wave_c[wv] = c;
  __syncthreads();
  if (tid == 0) {
    Cand r = wave_c[0];
    for (int w = 1; w < 4; ++w) cand_merge(r, wave_c[w].ov, wave_c[w].ov_idx, wave_c[w].dist, wave_c[w].dist_idx);
    GtCand o; o.ov = r.ov; o.ov_idx = r.ov_idx; o.dist_idx = r.dist_idx;
    cand[i] = o;
  }
}

__global__ __launch_bounds__(GT_THREADS) void encode_gt_kernel(GtArgs a) {
  extern __shared__ unsigned taken[];                  // A bits
  __shared__ Cand wave_c[GT_THREADS / 64];
  constexpr int MAXB = 256;                            // boxes / first choices of the image staged in LDS
  __shared__ float s_box[MAXB][4];
  __shared__ GtCand s_cand[MAXB];
  __shared__ int s_next;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = a.C + 9;
  const int nwords = (a.A + 31) >> 5;
  const int beg = a.box_offsets[b], end = a.box_offsets[b + 1];
  for (int i = tid; i < nwords; i += GT_THREADS) taken[i] = 0u;
  for (int i = tid; i < MAXB && beg + i < end; i += GT_THREADS) {
    s_box[i][0] = a.boxes[4 * (beg + i)]; s_box[i][1] = a.boxes[4 * (beg + i) + 1];
    s_box[i][2] = a.boxes[4 * (beg + i) + 2]; s_box[i][3] = a.boxes[4 * (beg + i) + 3];
    if (a.cand) s_cand[i] = ((const GtCand*)a.cand)[beg + i];
  }
  if (a.gt) {                                          // zero the image's dense gt (16-B stores)
    float* g = a.gt + (long long)b * a.A * row;
    const long long n = (long long)a.A * row;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
    for (long long i = tid; i < n4; i += GT_THREADS) reinterpret_cast<f32x4*>(g)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long long i = n4 * 4 + tid; i < n; i += GT_THREADS) g[i] = 0.f;
  }
  __syncthreads();
  auto box_of = [&](int i, float (&bx)[4]) {
    if (i - beg < MAXB) { bx[0] = s_box[i - beg][0]; bx[1] = s_box[i - beg][1]; bx[2] = s_box[i - beg][2]; bx[3] = s_box[i - beg][3]; }
    else { bx[0] = a.boxes[4 * i]; bx[1] = a.boxes[4 * i + 1]; bx[2] = a.boxes[4 * i + 2]; bx[3] = a.boxes[4 * i + 3]; }
  };
  // thread 0 only: anchor `pick` goes to box i -- taken bit, regression targets, dense row, index
  auto commit = [&](int i, int pick, const float (&bx)[4]) {
    if (pick >= 0) {
      taken[pick >> 5] |= 1u << (pick & 31);
      const float bcx = (bx[0] + bx[2]) / 2.f, bcy = (bx[1] + bx[3]) / 2.f, bw = bx[2] - bx[0] + 1.f, bh = bx[3] - bx[1] + 1.f;
      const double ax = a.anchors[4 * pick], ay = a.anchors[4 * pick + 1], aw = a.anchors[4 * pick + 2], ah = a.anchors[4 * pick + 3];
      const float dx = (float)(((double)bcx - ax) / aw), dy = (float)(((double)bcy - ay) / ah);
      const float dw = (float)log((double)bw / aw), dh = (float)log((double)bh / ah);
      if (a.deltas) { float* d = a.deltas + 4 * (long long)i; d[0] = dx; d[1] = dy; d[2] = dw; d[3] = dh; }
      if (a.gt) {
        float* g = a.gt + ((long long)b * a.A + pick) * row;
        g[0] = 1.f; g[1] = bx[0]; g[2] = bx[1]; g[3] = bx[2]; g[4] = bx[3]; g[5] = dx; g[6] = dy; g[7] = dw; g[8] = dh;
        const int cls = a.class_ids[i];
        if (cls >= 0 && cls < a.C) g[9 + cls] = 1.f;
      }
    }
    if (a.anchor_idx) a.anchor_idx[i] = pick >= 0 ? pick : a.A;     // reference's "unassigned" value is num_anchors
  };
  int i = beg;
  while (i < end) {
    if (a.cand) {
      // thread 0 walks the boxes whose first choice is still free (no scan, no barrier per box) up to the first conflict
      if (tid == 0) {
        int k = i;
        for (; k < end; ++k) {
          const GtCand fc = (k - beg < MAXB) ? s_cand[k - beg] : ((const GtCand*)a.cand)[k];
          const int want = fc.ov > 0.0 ? fc.ov_idx : fc.dist_idx;
          if (want < 0 || want >= a.A || (taken[want >> 5] & (1u << (want & 31)))) break;
          float bx[4]; box_of(k, bx);
          commit(k, want, bx);
        }
        s_next = k;
      }
      __syncthreads();
      i = s_next;
      if (i >= end) break;
    }
    // ---- box i: scan all anchors with the taken mask (every box when there is no first-choice pass, else a conflict) ----
    float bx[4]; box_of(i, bx);
    const float b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3];
    const float bcx = (b0 + b2) / 2.f, bcy = (b1 + b3) / 2.f, bw = b2 - b0 + 1.f, bh = b3 - b1 + 1.f;
    const double barea = (double)((b2 - b0) * (b3 - b1));
    Cand c; c.ov = -1.0; c.ov_idx = 0x7fffffff; c.dist = __builtin_inf(); c.dist_idx = 0x7fffffff;
    for (int j = tid; j < a.A; j += GT_THREADS) {
      if (taken[j >> 5] & (1u << (j & 31))) continue;
      const double ax = a.anchors[4 * j], ay = a.anchors[4 * j + 1], aw = a.anchors[4 * j + 2], ah = a.anchors[4 * j + 3];
      const double x0 = ax - 0.5 * (aw - 1.0), y0 = ay - 0.5 * (ah - 1.0);
      const double x1 = ax + 0.5 * (aw - 1.0), y1 = ay + 0.5 * (ah - 1.0);
      const double lr = fmax(fmin(x1, (double)b2) - fmax(x0, (double)b0), 0.0);
      const double tb = fmax(fmin(y1, (double)b3) - fmax(y0, (double)b1), 0.0);
      const double inter = lr * tb;
      const double uni = (x1 - x0) * (y1 - y0) + barea - inter;
      const double ov = inter / (uni + 1e-10);
      const double d0 = (double)bcx - ax, d1 = (double)bcy - ay, d2 = (double)bw - aw, d3 = (double)bh - ah;
      const double dist = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
      cand_merge(c, ov, j, dist, j);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(c.ov, off), d = __shfl_xor(c.dist, off);
      const int oi = __shfl_xor(c.ov_idx, off), di = __shfl_xor(c.dist_idx, off);
      cand_merge(c, ov, oi, d, di);
    }
    if (lane == 0) wave_c[wv] = c;
    __syncthreads();
    if (tid == 0) {
      Cand r = wave_c[0];
      for (int w = 1; w < GT_THREADS / 64; ++w) cand_merge(r, wave_c[w].ov, wave_c[w].ov_idx, wave_c[w].dist, wave_c[w].dist_idx);
      int pick = (r.ov > 0.0) ? r.ov_idx : r.dist_idx;
      if (pick == 0x7fffffff) pick = -1;               // more boxes than anchors: nothing free
      commit(i, pick, bx);
    }
    __syncthreads();
    ++i;
  }
}

// boxes [total][4] xyxy fp32 (network-input coordinates), class_ids [total] int32, box_offsets [B+1] int32 (image b
// owns boxes box_offsets[b] .. box_offsets[b+1]-1, in the order the reference would iterate them), anchors [A][4]
// float64 (cx,cy,w,h).  Outputs (each may be NULL): gt [B][A][C+9] dense (fully overwritten), anchor_idx [total]
// int32, deltas [total][4] fp32.  All pointers are device pointers.
// workspace: 16 * total_boxes bytes of device memory (total_boxes given by the caller) or NULL; with it a fully parallel
// first-choice pass runs first and the serial pass only re-scans on conflicts (same results, several times faster).
extern "C" int sqd_encode_gt_fwd(const float* boxes, const int* class_ids, const int* box_offsets, const double* anchors,
                                 float* gt, int* anchor_idx, float* deltas, void* workspace, int total_boxes, int B, int A,
                                 int num_classes, void* stream) {
  SQD_CHECK_ARG(box_offsets && anchors && B > 0 && A > 0 && A <= (1 << 20) && num_classes > 0);
  SQD_CHECK_ARG(gt || anchor_idx || deltas);
  SQD_CHECK_ARG(!gt || class_ids);
  GtArgs a;
  a.boxes = boxes; a.class_ids = class_ids; a.box_offsets = box_offsets; a.anchors = anchors;
  a.gt = gt; a.anchor_idx = anchor_idx; a.deltas = deltas; a.B = B; a.A = A; a.C = num_classes;
  a.cand = nullptr;
  if (workspace && total_boxes > 0 && boxes) {
    SQD_CHECK_ARG(((uintptr_t)workspace & 7) == 0 && total_boxes <= (1 << 24));
    hipLaunchKernelGGL(gt_candidates_kernel, dim3((unsigned)total_boxes), dim3(256), 0, (hipStream_t)stream, boxes, anchors,
                       (GtCand*)workspace, A);
    a.cand = workspace;
  }
  const size_t lds = (size_t)((A + 31) / 32) * sizeof(unsigned);
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)encode_gt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SQD_ERR_LAUNCH;
  hipLaunchKernelGGL(encode_gt_kernel, dim3((unsigned)B), dim3(GT_THREADS), lds, (hipStream_t)stream, a);
  return sqd_launch_status();
}
