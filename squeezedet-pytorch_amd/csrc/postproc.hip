// Post-processing: per-anchor decode (softmax / sigmoid / anchor box transform) and the fused
// detection kernel (decode -> top-k by score -> class-wise NMS -> score threshold -> compact).
//
// Reference: PredictionResolver.forward src/model/squeezedet.py:109-120, safe_softmax
// src/model/modules.py:66-68, deltas_to_boxes :27-45, xywh_to_xyxy :17-24, SqueezeDet.forward
// src/model/squeezedet.py:199-206, Detector.filter src/engine/detector.py:87-122 (+ torchvision nms),
// boxes_postprocess src/utils/boxes.py:145-147 (the eval-time scale division).
//
// pred is [B][A][C+5] fp32 (anchor-major: C class logits, 1 confidence logit, 4 deltas), anchors
// [A][4] (cx,cy,w,h) fp32.  The reference runs ~15 elementwise launches plus a Python loop with
// >=10 host syncs per image; here ONE launch (detect_kernel, one workgroup per image) does everything on the device:
//   1. 16 waves score the image's anchors into LDS: score = max_c softmax_c * sigmoid(conf) per anchor; key = fp32 score
//      bits (non-negative floats order like uint32) if score > score_thresh else 0; then 4 waves carry on:
//   2. stable compaction of the non-zero keys into LDS, 4-pass 8-bit radix select of the K-th largest; ties at that
//      key taken in ascending anchor order (the build's documented tie rule),
//   3. one wave ranks the <=64 candidates (score desc, anchor index asc), decodes their boxes,
//      builds the 64x64 suppression bit matrix (one 64-bit row per lane) and runs greedy NMS with
//      a wave-uniform alive mask; survivors are compacted in class order 0..C-1, each class in
//      descending score, then thresholded -- the exact output order of Detector.filter.
// Arithmetic mirrors the oracle op for op in fp32 (built with -ffp-contract=off, IEEE division), so
// that, given identical pred, the kept anchor indices are bit-exact.
#include "sqd_common.h"
#include <math.h>

#define SQD_MAX_CLASSES 16

struct BoxF { float x1, y1, x2, y2; };

// score / class of one anchor row.  p points at C+5 floats.
__device__ __forceinline__ void anchor_score(const float* __restrict__ p, int C, float& score, int& cls) {
  float l[SQD_MAX_CLASSES];
  float conf_logit;
  if (C == 3) {
    const f32x4 v = *(const f32x4*)p;
    l[0] = v.x; l[1] = v.y; l[2] = v.z; conf_logit = v.w;
  } else {
    for (int c = 0; c < C; ++c) l[c] = p[c];
    conf_logit = p[C];
  }
  float m = l[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) { l[c] = expf(l[c] - m); sum += l[c]; }
  const float conf = 1.f / (1.f + expf(-conf_logit));
  float best = -1.f; int bc = 0;
  for (int c = 0; c < C; ++c) {
    const float v = (l[c] / sum) * conf;
    if (v > best) { best = v; bc = c; }
  }
  score = best; cls = bc;
}

__device__ __forceinline__ BoxF anchor_box(const float* __restrict__ d, const float* __restrict__ anc, float wmax, float hmax) {
  const float ax = anc[0], ay = anc[1], aw = anc[2], ah = anc[3];
  const float cx = ax + aw * d[0];
  const float cy = ay + ah * d[1];
  const float w = aw * expf(d[2]);
  const float h = ah * expf(d[3]);
  BoxF b;
  b.x1 = fminf(fmaxf(cx - 0.5f * (w - 1.f), 0.f), wmax);
  b.y1 = fminf(fmaxf(cy - 0.5f * (h - 1.f), 0.f), hmax);
  b.x2 = fminf(fmaxf(cx + 0.5f * (w - 1.f), 0.f), wmax);
  b.y2 = fminf(fmaxf(cy + 0.5f * (h - 1.f), 0.f), hmax);
  return b;
}

// ---- full decode (module surface of SqueezeDet.forward: dense class_ids / scores / boxes) ----
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ pred, const float* __restrict__ anchors,
                                                     long long* __restrict__ class_ids, float* __restrict__ scores,
                                                     float* __restrict__ boxes, int B, int A, int C, float wmax, float hmax) {
  const long long total = (long long)B * A;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const float* p = pred + i * (C + 5);
    float s; int c;
    anchor_score(p, C, s, c);
    const BoxF b = anchor_box(p + C + 1, anchors + 4 * a, wmax, hmax);
    class_ids[i] = c; scores[i] = s;
    *(f32x4*)(boxes + 4 * i) = (f32x4){b.x1, b.y1, b.x2, b.y2};
  }
}

extern "C" int sqd_decode_fwd(const float* pred, const float* anchors, long long* class_ids, float* scores,
                              float* boxes, int B, int A, int num_classes, int input_h, int input_w, void* stream) {
  SQD_CHECK_ARG(pred && anchors && class_ids && scores && boxes && B > 0 && A > 0);
  SQD_CHECK_ARG(num_classes >= 1 && num_classes <= SQD_MAX_CLASSES);
  SQD_CHECK_ARG(((uintptr_t)pred & 15) == 0 && ((uintptr_t)boxes & 15) == 0);
  const long long total = (long long)B * A;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, anchors, class_ids,
                     scores, boxes, B, A, num_classes, (float)(input_w - 1), (float)(input_h - 1));
  return sqd_launch_status();
}

// ---- PredictionResolver.forward proper: the reference's five dense outputs (src/model/squeezedet.py:109-120) ----
__global__ __launch_bounds__(256) void resolve_kernel(const float* __restrict__ pred, const float* __restrict__ anchors,
                                                      float* __restrict__ probs, float* __restrict__ logp, float* __restrict__ scores,
                                                      float* __restrict__ deltas, float* __restrict__ boxes, int B, int A, int C,
                                                      float wmax, float hmax) {
  const long long total = (long long)B * A;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const float* p = pred + i * (C + 5);
    float l[SQD_MAX_CLASSES];
    float m = p[0];
    for (int c = 0; c < C; ++c) { l[c] = p[c]; m = fmaxf(m, l[c]); }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(l[c] - m);
    const float lse = logf(sum);
    for (int c = 0; c < C; ++c) {
      probs[i * C + c] = expf(l[c] - m) / sum;                 // safe_softmax (modules.py:66-68)
      if (logp) logp[i * C + c] = (l[c] - m) - lse;            // torch.log_softmax
    }
    scores[i] = 1.f / (1.f + expf(-p[C]));
    const float* d = p + C + 1;
    *(f32x4*)(deltas + 4 * i) = (f32x4){d[0], d[1], d[2], d[3]};
    const BoxF b = anchor_box(d, anchors + 4 * a, wmax, hmax);
    *(f32x4*)(boxes + 4 * i) = (f32x4){b.x1, b.y1, b.x2, b.y2};
  }
}

// probs [B][A][C], logp [B][A][C] or NULL, scores [B][A] (the reference's [B,A,1]), deltas [B][A][4], boxes [B][A][4]
extern "C" int sqd_resolve_fwd(const float* pred, const float* anchors, float* probs, float* logp, float* scores,
                               float* deltas, float* boxes, int B, int A, int num_classes, int input_h, int input_w,
                               void* stream) {
  SQD_CHECK_ARG(pred && anchors && probs && scores && deltas && boxes && B > 0 && A > 0);
  SQD_CHECK_ARG(num_classes >= 1 && num_classes <= SQD_MAX_CLASSES);
  SQD_CHECK_ARG(((uintptr_t)deltas & 15) == 0 && ((uintptr_t)boxes & 15) == 0);
  const long long total = (long long)B * A;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, anchors, probs, logp, scores,
                     deltas, boxes, B, A, num_classes, (float)(input_w - 1), (float)(input_h - 1));
  return sqd_launch_status();
}

// ---- fused detection ----
// One launch, one workgroup per image of 1024 threads.  Phase 1 scores the anchors into LDS keys -- the fp32 score bits if
// score > score_thresh, else 0; phase 2 compacts the non-zero keys (stable, ascending anchor index) and selects the top K
// (radix select; the 256 histogram bins belong to the first 256 threads, the candidate sweeps run on all 1024), ranks them,
// runs class-wise NMS on one wave and writes the compact result.  (Tried and dropped in
// round 2: spreading phase 1 over the whole chip and handing the keys to the last-arriving workgroup of each image
// through global memory -- the device-scope release / acquire pair across the XCDs' L2s cost 12-100 us, more than the
// 20 idle-chip microseconds it saved; profiles/r02_detect_variants.log.)
// Why pre-filtering by the threshold is exact: Detector.filter thresholds AFTER NMS, but a box at or below the
// threshold can only suppress boxes with lower scores, which are dropped by the same threshold -- so the kept set
// is that of NMS over the top-K of the above-threshold anchors.
#define DET_THREADS 256
#define DET_K 64

struct DetArgs {
  const float* pred; const float* anchors; const float* scales;   // scales [B][2] = (sy, sx) or null
  const float* shifts;                                            // [B][2] = (dy, dx) added after the scale division, or null
  const long long* in_class; const float* in_score; const float* in_box;   // dense inputs (filter mode) or null
  unsigned* keys;                                                 // workspace [B][ceil4(A)] keys + [B] arrival counters, or null (one workgroup per image)
  int S, per;
  int* det_count; long long* det_class; float* det_score; float* det_box; int* det_anchor;
  int B, A, C, K;
  float wmax, hmax, nms_thresh, score_thresh;
};

#define DET_SCORE_THREADS 1024      // threads per workgroup (16 waves)
#define DET_SPLIT 8                 // workgroups scoring one image when a key workspace is given

__global__ __launch_bounds__(DET_SCORE_THREADS) void detect_kernel(DetArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned* keysL = (unsigned*)smem_raw;                      // [A4] all keys of this image (A rounded up to 4)
  unsigned short* cidx = (unsigned short*)(smem_raw + (size_t)((a.A + 3) & ~3) * 4);   // [A] candidate anchor indices
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need, s_cnt, s_nlive;
  __shared__ unsigned wave_tot[DET_SCORE_THREADS / 64];
  __shared__ unsigned cand_key[DET_K];
  __shared__ int cand_idx[DET_K];
  __shared__ int sorted_pos[DET_K];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S;                                          // workgroups scoring one image (1: everything in this workgroup)
  const int b = (int)blockIdx.x / S, seg = (int)blockIdx.x - b * S;
  const int A = a.A, C = a.C, K = a.K;
  const float* pred = a.pred ? a.pred + (long long)b * A * (C + 5) : nullptr;
  const bool dense = a.in_score != nullptr;
  const int lo_a = seg * a.per, hi_a = min(A, lo_a + a.per);  // this workgroup's anchors (a.per is a multiple of 4)

  // 1. score the anchors straight into LDS.  Two passes keep the exp / divide work dense:
  //    1a. confidence only: score = max_c softmax_c * conf <= conf (softmax_c <= 1 and the products round monotonically), so
  //        conf <= threshold already decides key = 0 -- exactly; the others are listed in LDS (any order);
  //    1b. the listed anchors (typically ~15 %) get the full score.
  //    With S > 1 the image's anchors are split over S workgroups (one CU reads a 539 KB `pred` slice in ~20 us; eight read it in
  //    ~3): each writes its keys to the global workspace (write-through), drains and draws a ticket; the last arriver of the image
  //    loads all keys back into its LDS and carries on alone -- selection and NMS never leave that workgroup (in-launch hand-off of
  //    cdna_hip_programming.md Guideline 16: sc1 payload + drained stores + agent-scope counter, no release fence, so the hundreds
  //    of megabytes ConvDet's launch left dirty in the L2s are not written back for it; that is what made round 2's version of this
  //    cost 61-156 us).
  if (tid == 0) s_nlive = 0u;
  __syncthreads();
  for (int i0 = lo_a; i0 < hi_a; i0 += DET_SCORE_THREADS) {
    const int i = i0 + tid;
    bool live = false;
    if (i < hi_a) {
      if (dense) {
        const float s = a.in_score[(long long)b * A + i];
        keysL[i] = (s > a.score_thresh) ? __float_as_uint(s) : 0u;
      } else {
        const float conf = 1.f / (1.f + expf(-pred[(long long)i * (C + 5) + C]));
        live = conf > a.score_thresh;
        if (!live) keysL[i] = 0u;
      }
    }
    const unsigned long long m = __ballot(live);
    if (m) {
      unsigned base = 0u;
      if (lane == 0) base = atomicAdd(&s_nlive, (unsigned)__popcll(m));
      base = __shfl(base, 0);
      if (live) cidx[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
    }
  }
  __syncthreads();
  {
    const int nlive = (int)s_nlive;
    for (int j = tid; j < nlive; j += DET_SCORE_THREADS) {
      const int i = cidx[j];
      float s; int c;
      anchor_score(pred + (long long)i * (C + 5), C, s, c);
      keysL[i] = (s > a.score_thresh) ? __float_as_uint(s) : 0u;   // scores are >= 0: bit patterns order like the floats
    }
  }
  __syncthreads();
  if (S > 1) {
    typedef unsigned int u32x4_d __attribute__((ext_vector_type(4)));
    const int A4 = (A + 3) & ~3;
    const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.keys + (long long)b * A4), 0, 0x7ffffff0, 0x00020000);
    for (int q = lo_a / 4 + tid; q < (hi_a + 3) / 4; q += DET_SCORE_THREADS) {       // (keys past A inside the last quad: never read back)
      u32x4_d v = *(const u32x4_d*)(keysL + 4 * q);
      __builtin_amdgcn_raw_buffer_store_b128(v, kres, q * 16, 0, 16);               // sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its stores ...
    __syncthreads();                                         // ... before ONE lane counts the workgroup in
    unsigned* const tick = a.keys + (long long)a.B * A4 + b;
    if (tid == 0) s_cnt = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if ((int)s_cnt != S - 1) return;                         // not the last arriver of this image: done
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the counter returns to zero for the next launch
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int q = tid; q < A4 / 4; q += DET_SCORE_THREADS)
      *(u32x4_d*)(keysL + 4 * q) = __builtin_amdgcn_raw_buffer_load_b128(kres, q * 16, 0, 16);   // sc1 loads: every key of the image
    __syncthreads();
  }

  if (tid < DET_K) { cand_key[tid] = 0u; cand_idx[tid] = 0x7fffffff - DET_K + tid; }   // distinct sentinels: ranks stay a permutation
  if (tid == 0) { s_prefix = 0u; s_need = (unsigned)K; s_cnt = 0u; }

  // block-wide exclusive scan helper over one value per thread (fixed order -> deterministic)
  auto block_excl_scan = [&](unsigned c, unsigned& total) {
    unsigned incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned woff = 0u, tot = 0u;
    for (int w = 0; w < DET_SCORE_THREADS / 64; ++w) { if (w < wave) woff += wave_tot[w]; tot += wave_tot[w]; }
    total = tot;
    return incl - c + woff;
  };

  // 1. stable compaction of the above-threshold anchors: contiguous index range per thread
  const int chunk = (A + DET_SCORE_THREADS - 1) / DET_SCORE_THREADS;
  const int lo = tid * chunk, hi = min(A, lo + chunk);
  __syncthreads();
  unsigned c = 0u;
  for (int i = lo; i < hi; ++i) c += keysL[i] != 0u ? 1u : 0u;
  unsigned M;
  unsigned off = block_excl_scan(c, M);
  for (int i = lo; i < hi; ++i)
    if (keysL[i] != 0u) { cidx[off] = (unsigned short)i; ++off; }
  __syncthreads();

  int ncand;
  if ((int)M <= K) {
    for (int i = tid; i < (int)M; i += DET_SCORE_THREADS) { cand_key[i] = keysL[cidx[i]]; cand_idx[i] = cidx[i]; }
    ncand = (int)M;
    __syncthreads();
  } else {
    // 2. radix select of the K-th largest key among the M candidates
    unsigned mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < DET_THREADS) hist[tid] = 0u;                   // DET_THREADS == 256 bins
      __syncthreads();
      const unsigned prefix = s_prefix;
      // scores cluster in 2-3 exponent bins: in the leading pass aggregate equal bins inside a wave (one
      // atomic per distinct bin) instead of serialising up to 64 same-address LDS atomics
      for (int i0 = 0; i0 < (int)M; i0 += DET_SCORE_THREADS) {
        const int i = i0 + tid;
        const unsigned k = (i < (int)M) ? keysL[cidx[i]] : 0u;
        const bool act = (i < (int)M) && ((k & mask) == prefix);
        const unsigned bin = (k >> shift) & 255u;
        if (pass == 0) {
          unsigned long long todo = __ballot(act);
          while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned lb = __shfl(bin, leader);
            const unsigned long long same = __ballot(act && bin == lb);
            if (lane == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
            todo &= ~same;
          }
        } else if (act) {
          atomicAdd(&hist[bin], 1u);
        }
      }
      __syncthreads();
      {
        // bucket holding the need-th largest key: thread t owns bin 255-t; an exclusive block scan gives the
        // number of keys in higher bins (256 threads in parallel instead of a serial 256-step walk)
        const unsigned need = s_need;
        const unsigned h = (tid < DET_THREADS) ? hist[255 - tid] : 0u;
        unsigned tot;
        const unsigned above = block_excl_scan(h, tot);
        if (tid < DET_THREADS && above < need && above + h >= need) {          // exactly one thread
          s_need = need - above;                          // how many keys to take from this bucket (and below-level ties)
          s_prefix = prefix | ((unsigned)(255 - tid) << shift);
        }
      }
      mask |= 255u << shift;
      __syncthreads();
    }
    const unsigned tkey = s_prefix;       // K-th largest key
    const unsigned need = s_need;         // number of keys == tkey to take (>= 1)
    const unsigned n_gt = (unsigned)K - need;
    // 3a. keys strictly above the threshold key (unordered; ranked later)
    for (int i = tid; i < (int)M; i += DET_SCORE_THREADS) {
      const unsigned k = keysL[cidx[i]];
      if (k > tkey) { const unsigned pos = atomicAdd(&s_cnt, 1u); cand_key[pos] = k; cand_idx[pos] = cidx[i]; }
    }
    // 3b. ties at the threshold key in ascending anchor order (the compacted list is index-ordered)
    const int chunk2 = ((int)M + DET_SCORE_THREADS - 1) / DET_SCORE_THREADS;
    const int lo2 = tid * chunk2, hi2 = min((int)M, lo2 + chunk2);
    unsigned c2 = 0u;
    for (int i = lo2; i < hi2; ++i) c2 += (keysL[cidx[i]] == tkey) ? 1u : 0u;
    unsigned tot2;
    unsigned r = block_excl_scan(c2, tot2);
    for (int i = lo2; i < hi2 && r < need; ++i)
      if (keysL[cidx[i]] == tkey) { cand_key[n_gt + r] = tkey; cand_idx[n_gt + r] = cidx[i]; ++r; }
    ncand = K;
    __syncthreads();
  }

  // 4. rank the candidates: score desc, anchor index asc (sentinels: key 0, idx ~INT_MAX -> last)
  if (tid < DET_K) {
    const unsigned mykey = cand_key[tid];
    const int myidx = cand_idx[tid];
    int rank = 0;
    for (int j = 0; j < DET_K; ++j) {
      const unsigned kj = cand_key[j]; const int ij = cand_idx[j];
      rank += (kj > mykey || (kj == mykey && ij < myidx)) ? 1 : 0;
    }
    sorted_pos[rank] = tid;             // ranks are a permutation (indices are distinct)
  }
  __syncthreads();
  if (wave != 0) return;                // 5..7: one wave
  const int src = sorted_pos[lane];
  const int idx = cand_idx[src];
  const bool ok = src < ncand && lane < K;
  float score = 0.f; int cls = -1;
  BoxF bx = {0.f, 0.f, 0.f, 0.f};
  if (ok && dense) {
    const long long o = (long long)b * A + idx;
    score = a.in_score[o]; cls = (int)a.in_class[o];
    const f32x4 v = *(const f32x4*)(a.in_box + 4 * o);
    bx.x1 = v.x; bx.y1 = v.y; bx.x2 = v.z; bx.y2 = v.w;
  } else if (ok) {
    const float* p = pred + (long long)idx * (C + 5);
    anchor_score(p, C, score, cls);
    bx = anchor_box(p + C + 1, a.anchors + 4 * idx, a.wmax, a.hmax);
  }
  // suppression row: bit j set if j ranks after me, same class, IoU > thresh
  const float area = (bx.x2 - bx.x1) * (bx.y2 - bx.y1);
  unsigned long long row = 0ull;
  for (int j = 0; j < DET_K; ++j) {
    const float jx1 = __shfl(bx.x1, j), jy1 = __shfl(bx.y1, j), jx2 = __shfl(bx.x2, j), jy2 = __shfl(bx.y2, j);
    const float jarea = __shfl(area, j);
    const int jcls = __shfl(cls, j);
    const float w = fmaxf(0.f, fminf(bx.x2, jx2) - fmaxf(bx.x1, jx1));
    const float h = fmaxf(0.f, fminf(bx.y2, jy2) - fmaxf(bx.y1, jy1));
    const float inter = w * h;
    const float ovr = inter / ((area + jarea) - inter);
    if (j > lane && jcls == cls && cls >= 0 && ovr > a.nms_thresh) row |= 1ull << j;
  }
  unsigned long long alive = __ballot(ok);
  for (int i = 0; i < DET_K; ++i) {
    const unsigned lo32 = __shfl((unsigned)(row & 0xffffffffull), i);
    const unsigned hi32 = __shfl((unsigned)(row >> 32), i);
    if ((alive >> i) & 1ull) alive &= ~(((unsigned long long)hi32 << 32) | lo32);
  }
  const bool keep = ((alive >> lane) & 1ull) && score > a.score_thresh;
  int pos = 0, total = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int c = 0; c < C; ++c) {
    const unsigned long long m = __ballot(keep && cls == c);
    if (c < cls) pos += __popcll(m);
    else if (c == cls) pos += __popcll(m & below);
    total += __popcll(m);
  }
  if (lane == 0) a.det_count[b] = total;
  if (keep) {
    float sx = 1.f, sy = 1.f;
    if (a.scales) { sy = a.scales[2 * b]; sx = a.scales[2 * b + 1]; }
    const long long o = (long long)b * K + pos;
    a.det_class[o] = cls;
    a.det_score[o] = score;
    a.det_anchor[o] = idx;
    f32x4 ob = (f32x4){bx.x1 / sx, bx.y1 / sy, bx.x2 / sx, bx.y2 / sy};
    // boxes_postprocess' padding / crops terms (src/utils/boxes.py:149-155): per axis only one of the two is non-zero, so
    // (b - padding) + crops == b + (crops - padding) bit for bit
    if (a.shifts) { const float dy = a.shifts[2 * b], dx = a.shifts[2 * b + 1]; ob.x += dx; ob.y += dy; ob.z += dx; ob.w += dy; }
    *(f32x4*)(a.det_box + 4 * o) = ob;
  }
}

static int launch_detect(DetArgs a, hipStream_t stream) {
  if (a.K < 1 || a.K > DET_K) return SQD_ERR_UNSUPPORTED;                 // one wave holds the NMS bit matrix
  const size_t lds = (size_t)((a.A + 3) & ~3) * 4 + (size_t)a.A * 2 + 16;  // keys + uint16 candidate indices
  if (a.A > 65535 || lds > 150 * 1024) return SQD_ERR_UNSUPPORTED;        // A <= 25600 anchors per image
  static SqdDevOnce lds_once;                                              // raise the dynamic-LDS cap once per device
  if (lds > 48 * 1024 && sqd_max_lds_once(lds_once, (const void*)detect_kernel, 150 * 1024) != SQD_OK) return SQD_ERR_LAUNCH;
  // pred mode with a workspace: eight workgroups score an image, its last arriver selects and suppresses (the workspace holds B x
  // ceil4(A) keys + B counters that are zero between launches)
  a.S = (a.pred && a.keys) ? DET_SPLIT : 1;
  a.per = (((a.A + 3) / 4 + a.S - 1) / a.S) * 4;
  hipLaunchKernelGGL(detect_kernel, dim3((unsigned)(a.B * a.S)), dim3(DET_SCORE_THREADS), lds, stream, a);
  return sqd_launch_status();
}

// Fused decode + Detector.filter for a batch, straight from pred.  sqd_detect_fwd's keys_ws: unused (may be NULL; earlier versions kept
// the per-anchor keys in global memory).  Outputs are fixed-capacity
// [B][K]; rows >= det_count[b] are left untouched.  scales ([B][2] = (sy,sx), may be null) folds
// boxes_postprocess' division into the store.
// sqd_detect_shift_fwd: the same with shifts ([B][2] = (dy, dx), may be null) added to the boxes after the scale division --
// boxes_postprocess' padding / crops terms of the reference's forbid_resize branch (src/utils/boxes.py:149-155) -- and, when keys_ws
// is given (B * ceil4(A) + B uint32, the last B zero before the first launch; every launch leaves them zero), the scoring of an
// image spread over eight workgroups whose last arriver selects and suppresses.  keys_ws NULL: one workgroup per image.
extern "C" int sqd_detect_shift_fwd(const float* pred, const float* anchors, const float* scales, const float* shifts, unsigned* keys_ws,
                                    int* det_count, long long* det_class, float* det_score, float* det_box, int* det_anchor, int B, int A,
                                    int num_classes, int input_h, int input_w, int keep_top_k, float nms_thresh,
                                    float score_thresh, void* stream) {
  SQD_CHECK_ARG(pred && anchors && det_count && det_class && det_score && det_box && det_anchor);
  SQD_CHECK_ARG(B > 0 && A > 0 && num_classes >= 1 && num_classes <= SQD_MAX_CLASSES);
  SQD_CHECK_ARG(((uintptr_t)pred & 15) == 0 && ((uintptr_t)det_box & 15) == 0);
  DetArgs a;
  a.shifts = shifts;
  a.pred = pred; a.anchors = anchors; a.scales = scales; a.in_class = nullptr; a.in_score = nullptr; a.in_box = nullptr; a.keys = keys_ws;
  a.det_count = det_count; a.det_class = det_class; a.det_score = det_score; a.det_box = det_box; a.det_anchor = det_anchor;
  a.B = B; a.A = A; a.C = num_classes; a.K = keep_top_k;
  a.wmax = (float)(input_w - 1); a.hmax = (float)(input_h - 1);
  a.nms_thresh = nms_thresh; a.score_thresh = score_thresh;
  return launch_detect(a, (hipStream_t)stream);
}

extern "C" int sqd_detect_fwd(const float* pred, const float* anchors, const float* scales, unsigned* keys_ws, int* det_count,
                              long long* det_class, float* det_score, float* det_box, int* det_anchor, int B, int A,
                              int num_classes, int input_h, int input_w, int keep_top_k, float nms_thresh,
                              float score_thresh, void* stream) {
  (void)keys_ws;                      // (this entry point never required a sized workspace: one workgroup per image)
  return sqd_detect_shift_fwd(pred, anchors, scales, nullptr, nullptr, det_count, det_class, det_score, det_box, det_anchor, B, A,
                              num_classes, input_h, input_w, keep_top_k, nms_thresh, score_thresh, stream);
}

// Detector.filter(det) proper: the same kernel fed with already decoded dense tensors
// (class_ids int64 [B][A], scores [B][A], boxes [B][A][4]).
extern "C" int sqd_filter_fwd(const long long* class_ids, const float* scores, const float* boxes, unsigned* keys_ws, int* det_count,
                              long long* det_class, float* det_score, float* det_box, int* det_anchor, int B, int A,
                              int num_classes, int keep_top_k, float nms_thresh, float score_thresh, void* stream) {
  SQD_CHECK_ARG(class_ids && scores && boxes && det_count && det_class && det_score && det_box && det_anchor);
  SQD_CHECK_ARG(B > 0 && A > 0 && num_classes >= 1 && num_classes <= SQD_MAX_CLASSES);
  SQD_CHECK_ARG(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)det_box & 15) == 0);
  DetArgs a;
  a.shifts = nullptr;
  a.pred = nullptr; a.anchors = nullptr; a.scales = nullptr; a.in_class = class_ids; a.in_score = scores; a.in_box = boxes; a.keys = keys_ws;
  a.det_count = det_count; a.det_class = det_class; a.det_score = det_score; a.det_box = det_box; a.det_anchor = det_anchor;
  a.B = B; a.A = A; a.C = num_classes; a.K = keep_top_k;
  a.wmax = 0.f; a.hmax = 0.f; a.nms_thresh = nms_thresh; a.score_thresh = score_thresh;
  return launch_detect(a, (hipStream_t)stream);
}
