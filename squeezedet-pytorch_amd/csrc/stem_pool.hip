// Stem convolution (stride 2, 3 input channels, NCHW image -> NHWC features, bias + ReLU fused)
// and the ceil-mode 3x3/stride-2 max pool, forward and backward.
//
// Reference: src/model/squeezedet.py:34-36 (squeezedet: Conv2d(3,64,3,s2,p1)+ReLU+MaxPool(3,2,ceil)),
// :52-54 (squeezedetplus: Conv2d(3,96,7,s2,p3)), :39,:42 (the two later pools).
//
// The stem is an implicit GEMM with a tiny K (27 or 147): the NCHW input patch of a tile is staged
// planar in LDS, each lane gathers its im2col element with a precomputed per-lane offset table,
// weights ([N][K], the checkpoint's own OIHW order) sit in LDS with an odd row pitch.  MFMA operand
// A = weights, operand B = pixels, so a lane ends with 4 consecutive output channels of one pixel
// and stores 16 bytes (the stem output is the largest tensor of the network: 30.7 MB per image).
#include "sqd_common.h"

struct StemArgs {
  const float* x;     // [B][3][Hin][Win]
  const float* w;     // [N][3*KS*KS]
  const float* bias;  // [N]
  float* y;           // [B][Ho][Wo][N]
  int B, Hin, Win, Ho, Wo, N;
  int tiles_x, tiles_y;
};

template <int KS, int PAD, int NT>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemArgs a) {
  constexpr int MT = 2, TH = 8;
  constexpr int K = 3 * KS * KS;
  constexpr int KSTEPS = (K + 3) / 4;
  constexpr int KW = KSTEPS * 4 + 1;          // weight row pitch (odd)
  constexpr int IH = 2 * (TH - 1) + KS;
  constexpr int IW = 2 * 15 + KS;
  constexpr int IWP = IW | 1;                 // odd pitch
  constexpr int NIN = 3 * IH * IWP;
  constexpr int BN = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* inT = smem;                          // [3][IH][IWP] then one zero slot
  float* wT = smem + NIN + 1;                 // [BN][KW]

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y; const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;

  for (int idx = tid; idx < 3 * IH * IW; idx += 256) {
    const int c = idx % IW; int r = idx / IW; const int ci = r / IH; r -= ci * IH;
    const int iy = 2 * y0 - PAD + r, ix = 2 * x0 - PAD + c;
    float v = 0.f;
    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
      v = a.x[(((long long)b * 3 + ci) * a.Hin + iy) * a.Win + ix];
    inT[(ci * IH + r) * IWP + c] = v;
  }
  if (tid == 0) inT[NIN] = 0.f;
  for (int idx = tid; idx < BN * (KW - 1); idx += 256) {
    const int n = idx / (KW - 1), k = idx - n * (KW - 1);
    wT[n * KW + k] = (n < a.N && k < K) ? a.w[(long long)n * K + k] : 0.f;
  }
  __syncthreads();

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int pbase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) pbase[i] = (2 * (wm * MT + i)) * IWP + 2 * lr;

#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int k = 4 * s + g;
    const int ci = k / (KS * KS), rem = k - ci * (KS * KS), ky = rem / KS, kx = rem - ky * KS;
    const bool kok = k < K;
    const int koff = (ci * IH + ky) * IWP + kx;
    float bf[MT], af[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) bf[i] = inT[kok ? pbase[i] + koff : NIN];
#pragma unroll
    for (int j = 0; j < NT; ++j) af[j] = wT[(j * 16 + lr) * KW + k];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j], bf[i], acc[i][j]);
  }

#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int oy = y0 + wm * MT + i, ox = x0 + lr;
    if (oy >= a.Ho || ox >= a.Wo) continue;
    float* dst = a.y + (((long long)b * a.Ho + oy) * a.Wo + ox) * a.N;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = j * 16 + 4 * g;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if (a.bias) v += *(const f32x4*)(a.bias + n);
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      *(f32x4*)(dst + n) = v;
    }
  }
}

template <int KS, int PAD, int NT>
static int launch_stem(StemArgs a, hipStream_t s) {
  constexpr int K = 3 * KS * KS, KSTEPS = (K + 3) / 4, KW = KSTEPS * 4 + 1;
  constexpr int IH = 2 * 7 + KS, IW = 30 + KS, IWP = IW | 1;
  constexpr size_t lds = (size_t)(3 * IH * IWP + 1 + 16 * NT * KW) * sizeof(float);
  a.tiles_x = sqd_cdiv(a.Wo, 16); a.tiles_y = sqd_cdiv(a.Ho, 8);
  auto kern = stem_conv_kernel<KS, PAD, NT>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SQD_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), lds, s, a);
  return sqd_launch_status();
}

// x: NCHW [B,3,Hin,Win]; w: OIHW [N,3,KS,KS]; y: NHWC [B,Ho,Wo,N], Ho = (Hin + 2*pad - KS)/2 + 1.
extern "C" int sqd_stem_conv_relu_fwd(const float* x, const float* w, const float* bias, float* y,
                                      int B, int Hin, int Win, int N, int ksize, void* stream) {
  SQD_CHECK_ARG(x && w && y && B > 0 && Hin > 0 && Win > 0);
  SQD_CHECK_ARG(((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0));
  StemArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.B = B; a.Hin = Hin; a.Win = Win; a.N = N;
  hipStream_t s = (hipStream_t)stream;
  if (ksize == 3 && N == 64) {
    a.Ho = (Hin + 2 - 3) / 2 + 1; a.Wo = (Win + 2 - 3) / 2 + 1;
    return launch_stem<3, 1, 4>(a, s);
  }
  if (ksize == 7 && N == 96) {
    a.Ho = (Hin + 6 - 7) / 2 + 1; a.Wo = (Win + 6 - 7) / 2 + 1;
    return launch_stem<7, 3, 6>(a, s);
  }
  return SQD_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// Fused stem: conv(3->N, k, s2) + bias + ReLU + MaxPool(3, 2, ceil) in ONE persistent kernel.  The
// stem output (30.7 MB/image, the largest tensor of the network) never reaches HBM: a workgroup
// computes the (2*PH+1) x (2*PW+1) conv patch feeding a PH x PW pooled tile into LDS and pools it
// there.  Workgroups are persistent (weights stay in LDS; the next input patch is prefetched into
// registers under the current tile's MFMAs).  ReLU outputs are >= 0, so conv positions outside the
// feature map (ceil-mode clipped windows) are stored as 0 without changing any maximum.
// Optional argmax (uint8 0..8, first maximum in window scan order) for the backward pass.
// ---------------------------------------------------------------------------------------------
struct StemPoolArgs {
  const float* x; const float* w; const float* bias; float* y; uint8_t* amax;
  int B, Hin, Win, Ho, Wo, Hp, Wp, N;
  int tiles_x, tiles_y, ntiles;
};

template <int KS, int PAD, int NT>
__global__ __launch_bounds__(256) void stem_pool_kernel(StemPoolArgs a) {
  constexpr int PH = 4, PW = 8;                       // pooled tile
  constexpr int CH = 2 * PH + 1, CW = 2 * PW + 1;     // conv patch 9 x 17
  constexpr int NCP = CH * CW;                        // 153 conv pixels
  constexpr int NSUB = (NCP + 15) / 16;               // 10 subtiles of 16
  constexpr int MT = (NSUB + 3) / 4;                  // subtiles per wave (3)
  constexpr int K = 3 * KS * KS, KSTEPS = (K + 3) / 4, KW = KSTEPS * 4 + 1;
  constexpr int IH = 2 * (CH - 1) + KS, IW = 2 * (CW - 1) + KS, IWP = IW | 1;
  constexpr int NIN = 3 * IH * IWP;
  constexpr int BN = 16 * NT;
  constexpr int CP = BN + 4;                          // conv tile pitch (floats)
  constexpr int IN_IT = (3 * IH * IW + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* convT = smem;                                // [NSUB*16][CP]
  float* inT = convT + NSUB * 16 * CP;                // [3][IH][IWP] + zero slot
  float* wT = inT + NIN + 1;                          // [BN][KW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;

  for (int idx = tid; idx < BN * (KW - 1); idx += 256) {
    const int n = idx / (KW - 1), k = idx - n * (KW - 1);
    wT[n * KW + k] = (n < a.N && k < K) ? a.w[(long long)n * K + k] : 0.f;
  }
  if (tid == 0) inT[NIN] = 0.f;

  // per-lane im2col bases of this wave's subtiles (conv pixel p = sub*16 + lr -> (r, c) of the patch)
  int pbase[MT], prow[MT], pcol[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = (wave + 4 * i) * 16 + lr;
    if (p >= NCP) p = NCP - 1;
    prow[i] = p / CW; pcol[i] = p - prow[i] * CW;
    pbase[i] = (2 * prow[i]) * IWP + 2 * pcol[i];
  }

  float rin[IN_IT];
  auto load_in = [&](int t) {
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; const int b = t / a.tiles_y;
    const int iy0 = 2 * (2 * ty * PH) - PAD, ix0 = 2 * (2 * tx * PW) - PAD;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int idx = tid + it * 256;
      const int c = idx % IW; int r = idx / IW; const int ci = r / IH; r -= ci * IH;
      const int iy = iy0 + r, ix = ix0 + c;
      float v = 0.f;
      if (idx < 3 * IH * IW && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
        v = a.x[(((long long)b * 3 + ci) * a.Hin + iy) * a.Win + ix];
      rin[it] = v;
    }
  };
  auto store_in = [&]() {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int idx = tid + it * 256;
      const int c = idx % IW; int r = idx / IW; const int ci = r / IH; r -= ci * IH;
      if (idx < 3 * IH * IW) inT[(ci * IH + r) * IWP + c] = rin[it];
    }
  };

  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  load_in(tile);
  store_in();
  __syncthreads();

  for (;;) {
    const int ntile = tile + (int)gridDim.x;
    const bool has_next = ntile < a.ntiles;
    if (has_next) load_in(ntile);

    int t = tile;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; const int b = t / a.tiles_y;
    const int cy0 = 2 * ty * PH, cx0 = 2 * tx * PW;       // conv coords of the patch origin

    // ---- conv patch on the matrix cores ----
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int k = 4 * s + g;
      const int ci = k / (KS * KS), rem = k - ci * (KS * KS), ky = rem / KS, kx = rem - ky * KS;
      const bool kok = k < K;
      const int koff = (ci * IH + ky) * IWP + kx;
      float bf[MT], af[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) bf[i] = inT[kok ? pbase[i] + koff : NIN];
#pragma unroll
      for (int j = 0; j < NT; ++j) af[j] = wT[(j * 16 + lr) * KW + k];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j], bf[i], acc[i][j]);
    }
    // bias + ReLU -> LDS; positions outside the conv feature map become 0 (neutral for max of ReLU outputs)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int sub = wave + 4 * i;
      if (sub >= NSUB) continue;
      const int p = sub * 16 + lr;
      const bool inside = (p < NCP) && (cy0 + prow[i] < a.Ho) && (cx0 + pcol[i] < a.Wo);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + 4 * g;
        f32x4 v = acc[i][j];
        if (a.bias && n < a.N) v += *(const f32x4*)(a.bias + n);
        v.x = inside ? fmaxf(v.x, 0.f) : 0.f; v.y = inside ? fmaxf(v.y, 0.f) : 0.f;
        v.z = inside ? fmaxf(v.z, 0.f) : 0.f; v.w = inside ? fmaxf(v.w, 0.f) : 0.f;
        *(f32x4*)(convT + p * CP + n) = v;
      }
    }
    __syncthreads();
    // ---- 3x3 / stride-2 max pool out of LDS ----
    for (int idx = tid; idx < PH * PW * (BN / 4); idx += 256) {
      const int cq = idx % (BN / 4); const int pp = idx / (BN / 4);
      const int pr = pp / PW, pc = pp - pr * PW;
      const int py = ty * PH + pr, px = tx * PW + pc;
      if (py >= a.Hp || px >= a.Wp || 4 * cq >= a.N) continue;
      f32x4 m = (f32x4){-1.f, -1.f, -1.f, -1.f};
      int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          if (2 * py + dy >= a.Ho || 2 * px + dx >= a.Wo) continue;      // clipped window: not a candidate
          const f32x4 v = *(const f32x4*)(convT + ((2 * pr + dy) * CW + 2 * pc + dx) * CP + 4 * cq);
          const int tt = dy * 3 + dx;
          if (v.x > m.x || v.x != v.x) { m.x = v.x; ax = tt; }
          if (v.y > m.y || v.y != v.y) { m.y = v.y; ay = tt; }
          if (v.z > m.z || v.z != v.z) { m.z = v.z; az = tt; }
          if (v.w > m.w || v.w != v.w) { m.w = v.w; aw = tt; }
        }
      const long long o = (((long long)b * a.Hp + py) * a.Wp + px) * a.N + 4 * cq;
      *(f32x4*)(a.y + o) = m;
      if (a.amax) *(uint32_t*)(a.amax + o) = (uint32_t)ax | ((uint32_t)ay << 8) | ((uint32_t)az << 16) | ((uint32_t)aw << 24);
    }
    __syncthreads();                 // pooling done reading convT; everyone done reading inT
    if (has_next) store_in();
    __syncthreads();
    tile = ntile;
    if (!has_next) break;
  }
}

template <int KS, int PAD, int NT>
static int launch_stem_pool(StemPoolArgs a, hipStream_t s) {
  constexpr int CH = 9, CW = 17, NSUB = (CH * CW + 15) / 16;
  constexpr int K = 3 * KS * KS, KSTEPS = (K + 3) / 4, KW = KSTEPS * 4 + 1;
  constexpr int IH = 2 * (CH - 1) + KS, IW = 2 * (CW - 1) + KS, IWP = IW | 1;
  constexpr size_t lds = (size_t)(NSUB * 16 * (16 * NT + 4) + 3 * IH * IWP + 1 + 16 * NT * KW) * sizeof(float);
  static_assert(lds <= 160 * 1024, "stem_pool LDS budget");
  a.tiles_x = sqd_cdiv(a.Wp, 8); a.tiles_y = sqd_cdiv(a.Hp, 4);
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
  auto kern = stem_pool_kernel<KS, PAD, NT>;
  static int wgs_per_cu = 0;
  if (wgs_per_cu == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SQD_ERR_LAUNCH;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    wgs_per_cu = nb > 4 ? 4 : nb;
  }
  int dev = 0, cus = 256; hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const int slots = cus * wgs_per_cu;
  const int per_wg = sqd_cdiv(a.ntiles, slots);
  const int gx = sqd_cdiv(a.ntiles, per_wg);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, s, a);
  return sqd_launch_status();
}

// x NCHW [B,3,Hin,Win] -> y NHWC [B,Hp,Wp,N] = MaxPool(3,2,ceil)(ReLU(conv(x))); argmax may be NULL.
extern "C" int sqd_stem_conv_relu_pool_fwd(const float* x, const float* w, const float* bias, float* y,
                                           unsigned char* argmax, int B, int Hin, int Win, int N, int ksize,
                                           void* stream) {
  SQD_CHECK_ARG(x && w && y && B > 0 && Hin > 0 && Win > 0);
  SQD_CHECK_ARG(((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) && ((uintptr_t)argmax & 3) == 0);
  StemPoolArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.amax = argmax; a.B = B; a.Hin = Hin; a.Win = Win; a.N = N;
  const int pad = ksize == 3 ? 1 : 3;
  a.Ho = (Hin + 2 * pad - ksize) / 2 + 1; a.Wo = (Win + 2 * pad - ksize) / 2 + 1;
  SQD_CHECK_ARG(a.Ho >= 3 && a.Wo >= 3);
  a.Hp = (a.Ho - 3 + 1) / 2 + 1; a.Wp = (a.Wo - 3 + 1) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  if (ksize == 3 && N == 64) return launch_stem_pool<3, 1, 4>(a, s);
  if (ksize == 7 && N == 96) return launch_stem_pool<7, 3, 6>(a, s);
  return SQD_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, ceil_mode=True), NHWC.  Ho = ceil((H-3)/2)+1 (PyTorch additionally
// drops a last window that would start outside the input; with pad 0 that never happens for H>=3).
// Windows at the bottom/right edge are clipped to the input.  The argmax (0..8 = dy*3+dx, first
// maximum in row-major window order, as PyTorch's CPU kernel scans it) is optionally recorded
// for the backward pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ amax, int B, int H, int W, int C,
                                                          int Ho, int Wo) {
  const int cv = C >> 2;
  const long long total = (long long)B * Ho * Wo * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % cv); long long p = idx / cv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    f32x4 m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy + dy;
      if (iy >= H) break;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox + dx;
        if (ix >= W) break;
        const f32x4 v = *(const f32x4*)(x + (((long long)b * H + iy) * W + ix) * C + 4 * c4);
        const int t = dy * 3 + dx;
        // NaN propagates like PyTorch: (v > m) || isnan(v)
        if (v.x > m.x || v.x != v.x) { m.x = v.x; ax = t; }
        if (v.y > m.y || v.y != v.y) { m.y = v.y; ay = t; }
        if (v.z > m.z || v.z != v.z) { m.z = v.z; az = t; }
        if (v.w > m.w || v.w != v.w) { m.w = v.w; aw = t; }
      }
    }
    const long long o = (((long long)b * Ho + oy) * Wo + ox) * C + 4 * c4;
    *(f32x4*)(y + o) = m;
    if (amax) *(uint32_t*)(amax + o) = (uint32_t)ax | ((uint32_t)ay << 8) | ((uint32_t)az << 16) | ((uint32_t)aw << 24);
  }
}

extern "C" int sqd_maxpool3x3s2_ceil_fwd(const float* x, float* y, unsigned char* argmax, int B, int H, int W,
                                         int C, void* stream) {
  SQD_CHECK_ARG(x && y && B > 0 && H >= 3 && W >= 3 && C > 0 && (C & 3) == 0);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)argmax & 3) == 0);
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
  const long long total = (long long)B * Ho * Wo * (C >> 2);
  const int blocks = (int)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, argmax, B, H, W, C, Ho, Wo);
  return sqd_launch_status();
}

// Backward: dx[b,iy,ix,c] = sum over the (at most 4) windows covering (iy,ix) whose argmax is this
// element.  Gather form (one thread per input element, no atomics, deterministic).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                          float* __restrict__ dx, const float* __restrict__ relu_src,
                                                          int B, int H, int W, int C, int Ho, int Wo) {
  const int cv = C >> 2;
  const long long total = (long long)B * H * W * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % cv); long long p = idx / cv;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H); const int b = (int)(p / H);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // windows oy with 2*oy <= iy <= 2*oy+2  ->  oy in [ceil((iy-2)/2), floor(iy/2)]
    const int oy_lo = iy >= 2 ? (iy - 1) / 2 : 0, oy_hi = min(iy / 2, Ho - 1);
    const int ox_lo = ix >= 2 ? (ix - 1) / 2 : 0, ox_hi = min(ix / 2, Wo - 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const int t = (iy - 2 * oy) * 3 + (ix - 2 * ox);
        const long long o = (((long long)b * Ho + oy) * Wo + ox) * C + 4 * c4;
        const uint32_t am = *(const uint32_t*)(amax + o);
        const f32x4 g = *(const f32x4*)(dy + o);
        if ((int)(am & 255) == t) acc.x += g.x;
        if ((int)((am >> 8) & 255) == t) acc.y += g.y;
        if ((int)((am >> 16) & 255) == t) acc.z += g.z;
        if ((int)(am >> 24) == t) acc.w += g.w;
      }
    const long long xo = (((long long)b * H + iy) * W + ix) * C + 4 * c4;
    if (relu_src) {      // the pooled tensor was a ReLU output: fold its backward mask into this store
      const f32x4 m = *(const f32x4*)(relu_src + xo);
      acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f; acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
    }
    *(f32x4*)(dx + xo) = acc;
  }
}

extern "C" int sqd_maxpool3x3s2_ceil_bwd(const float* dy, const unsigned char* argmax, float* dx, const float* relu_src,
                                         int B, int H, int W, int C, void* stream) {
  SQD_CHECK_ARG(dy && argmax && dx && B > 0 && H >= 3 && W >= 3 && C > 0 && (C & 3) == 0);
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
  const long long total = (long long)B * H * W * (C >> 2);
  const int blocks = (int)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, relu_src, B, H, W, C, Ho, Wo);
  return sqd_launch_status();
}
