// GPU-side input pipeline: uint8 HWC RGB images of arbitrary sizes ->
// whiten -> bilinear resize to the network input -> fp32 NCHW, one launch for the whole batch.
//
// Reference (CPU, per image inside DataLoader workers): DataWrapper.__getitem__ src/engine/detector.py:132-142,
// BaseDataset.preprocess src/datasets/base.py:43-59 (eval: drift/flip inactive), whiten src/utils/image.py:9-19
// ((image - mean) / std on float32 HWC), resize :77-88 (cv2.resize(image, (W, H)), default INTER_LINEAR;
// scales = [H/H0, W/W0] float32), transpose(2,0,1) detector.py:140.  KITTI mean/std: src/datasets/kitti.py:17-18.
//
// cv2.resize INTER_LINEAR on float32 data (OpenCV, third party, not under /root/reference -- published algorithm
// restated): for destination x, fx = (x + 0.5) * (W0 / W) - 0.5; sx = floor(fx); fx -= sx; if sx < 0 -> (sx, fx) =
// (0, 0); if sx >= W0 - 1 -> (sx, fx) = (W0 - 1, 0); likewise y; value = (1-fy) * ((1-fx) * s00 + fx * s01) +
// fy * ((1-fx) * s10 + fx * s11), horizontal pass first, all in float32.  Uploading uint8 instead of the
// reference's fp32 cuts host->device bytes 4x (5.75 MB -> ~1.4 MB per KITTI image).
#include "sqd_common.h"

struct PreArgs {
  const unsigned char* src;      // packed images, image b at src + offsets[b], HWC uint8, 3 channels
  const long long* offsets;      // [B]
  const int* sizes;              // [B][2] = (H0, W0)
  float* out;                    // [B][3][H][W]
  float* scales;                 // [B][2] = (H / H0, W / W0)  (may be null)
  float mean[3], stdv[3];
  int B, H, W;
};

// Both kernels: a workgroup produces 256 consecutive pixels of one output row.  The source bytes it needs are one or two CONTIGUOUS
// row segments, so they are fetched once as aligned dwords (coalesced) into LDS and the per-pixel 3-byte gathers read LDS -- a byte
// gather from global memory is one texture-addresser instruction per byte and wave (12 per pixel for the bilinear taps: ~50 us of
// address processing for a 20-image batch, against 25 us of HBM time for its 143 MB).  Whitening is a 256-entry table per channel,
// ((float)v - mean) / std evaluated once per workgroup with the same float32 subtract and IEEE divide the per-pixel form used: bit for bit
// the same values.  A segment that does not fit the LDS buffer (down-scaling by more than ~10x) takes the direct path.
constexpr int PRE_ROWB = 8192;                       // bytes of LDS per staged row segment

constexpr int PRE_ROWS = 4;                          // output rows per workgroup (the whitening table is built once for all of them)

__device__ __forceinline__ void pre_build_lut(float* lut, float m0, float m1, float m2, float s0, float s1, float s2) {
  for (int i = threadIdx.x; i < 768; i += 256) {
    const int c = i >> 8, v = i & 255;
    const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    lut[i] = ((float)v - m) / sd;
  }
}

// bytes [begin, begin + len) of the image at ``img`` (``img_end`` = one past its last byte) -> dst[shift ...]; returns shift (0..3).
// Aligned dword loads; a dword that would reach past the image's last byte is assembled from byte loads (the packed buffer may end there).
__device__ __forceinline__ int pre_stage(unsigned* dst, const unsigned char* img, const unsigned char* img_end, long long begin, int len) {
  const unsigned char* a0 = img + begin;
  const int shift = (int)((uintptr_t)a0 & 3);
  const unsigned* base = (const unsigned*)(a0 - shift);
  const int ndw = (shift + len + 3) >> 2;
  for (int i = threadIdx.x; i < ndw; i += 256) {
    const unsigned char* q = (const unsigned char*)(base + i);
    unsigned v;
    if (q + 4 <= img_end) v = base[i];
    else {
      v = 0;
      for (int k = 0; k < 4; ++k) if (q + k < img_end) v |= (unsigned)q[k] << (8 * k);
    }
    dst[i] = v;
  }
  return shift;
}

__global__ __launch_bounds__(256) void preprocess_kernel(PreArgs a) {
  __shared__ float lut[768];
  __shared__ unsigned rows[2][PRE_ROWB / 4];
  const int b = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int H0 = a.sizes[2 * b], W0 = a.sizes[2 * b + 1];
  if (x == 0 && blockIdx.y == 0 && a.scales) {
    a.scales[2 * b] = (float)a.H / (float)H0;       // np.array([H/H0, W/W0], dtype=float32): float64 division then cast;
    a.scales[2 * b + 1] = (float)a.W / (float)W0;   // identical for these integer ratios up to float32 rounding
  }
  const unsigned char* img = a.src + a.offsets[b];
  const unsigned char* img_end = img + (long long)H0 * W0 * 3;
  // source coordinates (double scale like OpenCV, then float weights)
  const double sclx = (double)W0 / (double)a.W, scly = (double)H0 / (double)a.H;
  auto src_x = [&](int xx, float& f) {
    f = (float)((xx + 0.5) * sclx - 0.5);
    int sx = (int)floorf(f); f -= (float)sx;
    if (sx < 0) { sx = 0; f = 0.f; }
    if (sx >= W0 - 1) { sx = W0 - 1; f = 0.f; }
    return sx;
  };
  // the workgroup's source column range [lo, hi] (sx is monotone in x): one segment per source row
  float fdummy;
  const int xa = blockIdx.x * 256, xb = min(a.W, xa + 256) - 1;
  const int lo = src_x(xa, fdummy), hi = min(src_x(xb, fdummy) + 1, W0 - 1);
  const int len = (hi - lo + 1) * 3;
  const bool staged = len + 8 <= PRE_ROWB;
  float fx;
  const int sx = src_x(min(x, a.W - 1), fx);
  const int sx1 = min(sx + 1, W0 - 1);
  const float ax0 = 1.f - fx, ax1 = fx;
  const long long plane = (long long)a.H * a.W;
  pre_build_lut(lut, a.mean[0], a.mean[1], a.mean[2], a.stdv[0], a.stdv[1], a.stdv[2]);
  for (int r = 0; r < PRE_ROWS; ++r) {
    const int y = blockIdx.y * PRE_ROWS + r;
    if (y >= a.H) break;                             // (uniform)
    float fy = (float)((y + 0.5) * scly - 0.5);
    int sy = (int)floorf(fy); fy -= (float)sy;
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= H0 - 1) { sy = H0 - 1; fy = 0.f; }
    const int sy1 = min(sy + 1, H0 - 1);
    int sh0 = 0, sh1 = 0;
    if (r) __syncthreads();                          // the previous row's readers are done with the segments
    if (staged) {
      sh0 = pre_stage(rows[0], img, img_end, ((long long)sy * W0 + lo) * 3, len);
      sh1 = pre_stage(rows[1], img, img_end, ((long long)sy1 * W0 + lo) * 3, len);
    }
    __syncthreads();
    if (x >= a.W) continue;
    const float ay0 = 1.f - fy, ay1 = fy;
    float* o = a.out + (long long)b * 3 * plane + (long long)y * a.W + x;
    const unsigned char *p00, *p01, *p10, *p11;
    if (staged) {
      const unsigned char* r0 = (const unsigned char*)rows[0] + sh0;
      const unsigned char* r1 = (const unsigned char*)rows[1] + sh1;
      p00 = r0 + (sx - lo) * 3; p01 = r0 + (sx1 - lo) * 3; p10 = r1 + (sx - lo) * 3; p11 = r1 + (sx1 - lo) * 3;
    } else {
      p00 = img + ((long long)sy * W0 + sx) * 3; p01 = img + ((long long)sy * W0 + sx1) * 3;
      p10 = img + ((long long)sy1 * W0 + sx) * 3; p11 = img + ((long long)sy1 * W0 + sx1) * 3;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* l = lut + 256 * c;                                    // whiten (image.py:17) in float32: the table holds (v - mean) / std
      const float v00 = l[p00[c]], v01 = l[p01[c]], v10 = l[p10[c]], v11 = l[p11[c]];
      const float r0 = v00 * ax0 + v01 * ax1;
      const float r1 = v10 * ax0 + v11 * ax1;
      o[c * plane] = r0 * ay0 + r1 * ay1;
    }
  }
}

// src: device buffer holding the B images back to back (HWC uint8 RGB); offsets [B] byte offsets; sizes [B][2] =
// (H0, W0) int32; out: NCHW fp32 [B][3][H][W]; scales: [B][2] fp32 or NULL; mean/std: 3 floats each (host).
extern "C" int sqd_preprocess_u8_fwd(const unsigned char* src, const long long* offsets, const int* sizes, float* out,
                                     float* scales, const float* mean3, const float* std3, int B, int H, int W,
                                     void* stream) {
  SQD_CHECK_ARG(src && offsets && sizes && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535);
  PreArgs a;
  a.src = src; a.offsets = offsets; a.sizes = sizes; a.out = out; a.scales = scales; a.B = B; a.H = H; a.W = W;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; SQD_CHECK_ARG(std3[c] != 0.f); }
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)sqd_cdiv(W, 256), (unsigned)sqd_cdiv(H, PRE_ROWS), (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  return sqd_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// The reference's OTHER pre-processing branch, cfg.forbid_resize (src/datasets/base.py:53-54): whiten (src/utils/image.py:9-19),
// then crop_or_pad (:91-124) -- per axis, a smaller image is zero-padded to the target (floor half in front, np.pad constant 0
// AFTER whitening: padded pixels are 0.0), a larger one is centre-cropped (floor half cut off in front) -- then HWC -> CHW.
// Pure index arithmetic plus the one float32 subtract and divide of whiten: bit-exact against the reference.
// Also writes what boxes_postprocess (src/utils/boxes.py:149-155) needs to map detections back: padding / crops (top, bottom,
// left, right) and the box shift (dy, dx) = (crops[0] - padding[0], crops[2] - padding[2]) the fused detect kernel adds.
// ---------------------------------------------------------------------------------------------------------------------
struct PadCropArgs {
  const unsigned char* src; const long long* offsets; const int* sizes;
  float* out; float* shifts; int* padcrop;
  float mean[3], stdv[3];
  int B, H, W;
};

__global__ __launch_bounds__(256) void preprocess_padcrop_kernel(PadCropArgs a) {
  __shared__ float lut[768];
  __shared__ unsigned row[PRE_ROWB / 4];
  const int b = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int H0 = a.sizes[2 * b], W0 = a.sizes[2 * b + 1];
  // (target - size) // 2 in front when padding, (size - target) // 2 cut in front when cropping
  const int pt = H0 < a.H ? (a.H - H0) / 2 : 0, ct = H0 > a.H ? (H0 - a.H) / 2 : 0;
  const int pl = W0 < a.W ? (a.W - W0) / 2 : 0, cl = W0 > a.W ? (W0 - a.W) / 2 : 0;
  if (x == 0 && blockIdx.y == 0) {
    if (a.padcrop) {
      int* pc = a.padcrop + 8 * b;
      pc[0] = pt; pc[1] = H0 < a.H ? (a.H - H0) - pt : 0; pc[2] = pl; pc[3] = W0 < a.W ? (a.W - W0) - pl : 0;
      pc[4] = ct; pc[5] = H0 > a.H ? (H0 - a.H) - ct : 0; pc[6] = cl; pc[7] = W0 > a.W ? (W0 - a.W) - cl : 0;
    }
    if (a.shifts) { a.shifts[2 * b] = (float)(ct - pt); a.shifts[2 * b + 1] = (float)(cl - pl); }
  }
  const int sx = x - pl + cl;
  const unsigned char* img = a.src + a.offsets[b];
  // the workgroup's source columns: [lo, hi] of the source row, clipped to the image (256 pixels = 768 bytes: always fits)
  const int xa = blockIdx.x * 256;
  const int lo = max(xa - pl + cl, 0), hi = min(min(a.W, xa + 256) - 1 - pl + cl, W0 - 1);
  const long long plane = (long long)a.H * a.W;
  pre_build_lut(lut, a.mean[0], a.mean[1], a.mean[2], a.stdv[0], a.stdv[1], a.stdv[2]);
  for (int r = 0; r < PRE_ROWS; ++r) {
    const int y = blockIdx.y * PRE_ROWS + r;
    if (y >= a.H) break;                             // (uniform)
    const int sy = y - pt + ct;
    const bool row_in = sy >= 0 && sy < H0 && hi >= lo;
    int sh = 0;
    if (r) __syncthreads();
    if (row_in) sh = pre_stage(row, img, img + (long long)H0 * W0 * 3, ((long long)sy * W0 + lo) * 3, (hi - lo + 1) * 3);
    __syncthreads();
    if (x >= a.W) continue;
    const bool inside = row_in && sx >= 0 && sx < W0;
    float* o = a.out + (long long)b * 3 * plane + (long long)y * a.W + x;
    const unsigned char* p = (const unsigned char*)row + sh + (inside ? (sx - lo) * 3 : 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * plane] = inside ? lut[256 * c + p[c]] : 0.f;
  }
}

// Arguments as sqd_preprocess_u8_fwd; shifts: [B][2] fp32 (dy, dx) or NULL; padcrop: [B][8] int32 = padding (top, bottom, left,
// right) then crops (top, bottom, left, right), or NULL.
extern "C" int sqd_preprocess_u8_padcrop_fwd(const unsigned char* src, const long long* offsets, const int* sizes, float* out,
                                             float* shifts, int* padcrop, const float* mean3, const float* std3, int B, int H, int W,
                                             void* stream) {
  SQD_CHECK_ARG(src && offsets && sizes && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535);
  PadCropArgs a;
  a.src = src; a.offsets = offsets; a.sizes = sizes; a.out = out; a.shifts = shifts; a.padcrop = padcrop; a.B = B; a.H = H; a.W = W;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; SQD_CHECK_ARG(std3[c] != 0.f); }
  hipLaunchKernelGGL(preprocess_padcrop_kernel, dim3((unsigned)sqd_cdiv(W, 256), (unsigned)sqd_cdiv(H, PRE_ROWS), (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  return sqd_launch_status();
}
