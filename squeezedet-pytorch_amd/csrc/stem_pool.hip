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
#include <type_traits>
#include <stdlib.h>

struct StemArgs {
  const float* x;     // [B][3][Hin][Win]
  const float* w;     // [N][3*KS*KS]
  const float* bias;  // [N]
  float* y;           // [B][Ho][Wo][N]
  int B, Hin, Win, Ho, Wo, N;
  int tiles_x, tiles_y;
  int relu;           // 0: plain Conv2d (the stand-alone features[0] of the module surface)
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
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
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
extern "C" int sqd_stem_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                                 int B, int Hin, int Win, int N, int ksize, int relu, void* stream) {
  SQD_CHECK_ARG(x && w && y && B > 0 && Hin > 0 && Win > 0);
  SQD_CHECK_ARG(((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0));
  StemArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.B = B; a.Hin = Hin; a.Win = Win; a.N = N; a.relu = relu;
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

extern "C" int sqd_stem_conv_relu_fwd(const float* x, const float* w, const float* bias, float* y,
                                      int B, int Hin, int Win, int N, int ksize, void* stream) {
  return sqd_stem_conv_fwd(x, w, bias, y, B, Hin, Win, N, ksize, 1, stream);
}

// ---------------------------------------------------------------------------------------------
// Fused stem: conv(3->N, k, s2) + bias + ReLU + MaxPool(3, 2, ceil) in ONE persistent kernel.  The
// stem output (30.7 MB/image, the largest tensor of the network) never reaches HBM: a workgroup
// computes the (2*PH+1) x (2*PW+1) conv patch feeding a PH x PW pooled tile into LDS and pools it
// there.  ReLU outputs are >= 0, so conv positions outside the feature map (ceil-mode clipped windows)
// are stored as 0 without changing any maximum.
//
// Per tile: the fp32 NCHW input patch [3][IH][IW] is moved global -> LDS by 4-byte LDS-DMA into a
// double buffer (no staging registers, no index arithmetic: uniform tile origin + a per-lane element
// offset computed once; only border tiles redirect out-of-image taps to a zero page), the weights
// stay in LDS, im2col happens in the LDS read addresses of the MFMA B operand.  PH = 3 makes the conv
// patch 7 x 17 = 119 pixels = 8 MFMA row tiles, two per wave (balanced).  Two barriers per tile.
// ARGMAX = false (inference): pooling is v_max3 over the window.  ARGMAX = true (training): uint8 argmax
// 0..8, first maximum in window scan order, NaN taken like PyTorch's CPU kernel.
// ---------------------------------------------------------------------------------------------
#ifndef SQD_STEM_WAVES
#define SQD_STEM_WAVES 8
#endif
struct StemPoolArgs {
  const float* x; const float* w; const float* bias; float* y; uint8_t* amax;
  int B, Hin, Win, Ho, Wo, Hp, Wp, N;
  int tiles_x, tiles_y, ntiles;
  unsigned tiles_x_m, tiles_y_m;      // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y): tile index split by scalar multiply-highs
};

__device__ __attribute__((aligned(16))) float sqd_stem_zero[4] = {0.f, 0.f, 0.f, 0.f};
typedef __attribute__((address_space(3))) void* stem_lds_ptr_t;

template <int KS, int PAD, int NT, bool ARGMAX, int WM>
__global__ __launch_bounds__(WM * 64, (WM == 4 && KS == 3) ? 3 : ((WM == 8 && KS == 3) ? 4 : 1)) void stem_pool_kernel(StemPoolArgs a) {
  constexpr int NTHR = WM * 64;                       // 4 or 8 waves; with 8, two waves per SIMD work on one tile
  constexpr int PH = 3, PW = 8;                       // pooled tile
  constexpr int CH = 2 * PH + 1, CW = 2 * PW + 1;     // conv patch 7 x 17
  constexpr int NCP = CH * CW;                        // 119 conv pixels
  constexpr int NSUB = (NCP + 15) / 16;               // 8 row tiles of 16
  constexpr int MT = NSUB / WM;                       // row tiles per wave
  static_assert(NSUB % WM == 0, "conv patch must split evenly over the waves");
  constexpr int K = 3 * KS * KS, KSTEPS = (K + 3) / 4, KW = KSTEPS * 4 + 1;
  constexpr int IH = 2 * (CH - 1) + KS, IW = 2 * (CW - 1) + KS;
  constexpr int NIN = 3 * IH * IW;
  // Wave roles (WM = 8): waves 0..3 are the only ones that issue input DMA, waves 4..7 the only ones that issue global
  // stores (they do the pooling).  A wave's vmcnt then counts one kind of operation: the DMA waves' vmcnt(0) at the
  // tile barrier waits for loads issued a whole tile earlier, and nobody ever waits for a store acknowledgement
  // (with mixed roles every tile barrier also waited out the HBM write latency of the previous tile's outputs).
  constexpr int NPROD = (WM == 8) ? 256 : NTHR;       // DMA threads
  constexpr int NPOOL = (WM == 8) ? 256 : NTHR;       // pooling / storing threads
  constexpr int IN_IT = (NIN + NPROD - 1) / NPROD, INSLOTS = IN_IT * NPROD;
  constexpr bool WREG = KSTEPS * NT <= 32;            // weights + im2col offsets live in registers (3x3 stem)
  constexpr int BN = 16 * NT;
  constexpr int CP = BN + 4;                          // conv tile pitch (floats): conflict-free b128 rows
  constexpr int NITEM = PH * PW * (BN / 4), P_IT = (NITEM + NPOOL - 1) / NPOOL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const convT = smem;                          // [NSUB*16][CP]
  float* const inB = convT + NSUB * 16 * CP;          // [2][INSLOTS] flat [ci][r][c] images + 4 zero floats
  float* const zeroL = inB + 2 * INSLOTS;
  float* const wT = zeroL + 4;                        // [BN][KW] (absent when the weights live in registers)
  float* const biasL = wT + (WREG ? 0 : BN * KW);     // [BN]
  int* const koffL = (int*)(biasL + BN);              // [KSTEPS*4] im2col offset of k = 4s+g inside the input image

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  if (!WREG)
    for (int idx = tid; idx < BN * (KW - 1); idx += NTHR) {
      const int n = idx / (KW - 1), k = idx - n * (KW - 1);
      wT[n * KW + k] = (n < a.N && k < K) ? a.w[(long long)n * K + k] : 0.f;
    }
  if (tid < BN) biasL[tid] = (a.bias && tid < a.N) ? a.bias[tid] : 0.f;
  if (tid < 4) zeroL[tid] = 0.f;
  if (tid < KSTEPS * 4) {
    const int k = tid < K ? tid : 0;
    const int ci = k / (KS * KS), rem = k - ci * (KS * KS), ky = rem / KS, kx = rem - ky * KS;
    koffL[tid] = (ci * IH + ky) * IW + kx;
  }

  // per-lane im2col bases of this wave's row tiles (conv pixel p = sub*16 + lr -> (r, c) of the patch)
  int pbase[MT], pkey[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = (wave + WM * i) * 16 + lr;
    const bool real = p < NCP;
    if (!real) p = NCP - 1;
    const int r = p / CW, c = p - r * CW;
    pbase[i] = (2 * r) * IW + 2 * c;
    pkey[i] = real ? (r << 8 | c) : -1;
  }
  // input DMA slots: element offset from the tile's origin pixel (channel 0, iy0, ix0); padding slots fetch the origin
  int in_off[IN_IT], in_key[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = (tid & (NPROD - 1)) + it * NPROD;
    const int c = idx % IW; int r = idx / IW; const int ci = r / IH; r -= ci * IH;
    const bool real = idx < NIN;
    in_off[it] = real ? (ci * a.Hin + r) * a.Win + c : 0;
    in_key[it] = real ? (r << 8 | c) : -1;
  }
  // pooling items: (pooled pixel, channel quad)
  int pl_lds[P_IT], pl_out[P_IT], pl_key[P_IT], pl_ch[P_IT];
#pragma unroll
  for (int it = 0; it < P_IT; ++it) {
    const int idx = (tid & (NPOOL - 1)) + it * NPOOL;
    const int cq = idx % (BN / 4); const int pp = idx / (BN / 4);
    const int pr = pp / PW, pc = pp - pr * PW;
    const bool real = idx < NITEM && 4 * cq < a.N;
    pl_lds[it] = ((2 * pr) * CW + 2 * pc) * CP + 4 * cq;
    pl_out[it] = (pr * a.Wp + pc) * a.N + 4 * cq;
    pl_key[it] = real ? (pr << 8 | pc) : -1;
    pl_ch[it] = 4 * cq;
  }

  struct Tile { int ty, tx, inner; const float* xorg; long long obase; };
  auto tile_at = [&](int t) {
    Tile q;
    const int t1 = a.tiles_x_m ? (int)__umulhi((unsigned)t, a.tiles_x_m) : t;      // t / tiles_x (exact: t * tiles_x < 2^32, host-checked; 0 = divisor 1)
    q.tx = t - t1 * a.tiles_x;
    const int b = a.tiles_y_m ? (int)__umulhi((unsigned)t1, a.tiles_y_m) : t1;
    q.ty = t1 - b * a.tiles_y;
    const int iy0 = 2 * (2 * q.ty * PH) - PAD, ix0 = 2 * (2 * q.tx * PW) - PAD;
    q.xorg = a.x + ((long long)b * 3 * a.Hin + iy0) * a.Win + ix0;      // dereferenced only where the pixel exists
    q.inner = iy0 >= 0 && iy0 + IH <= a.Hin && ix0 >= 0 && ix0 + IW <= a.Win;
    q.obase = (((long long)b * a.Hp + q.ty * PH) * a.Wp + q.tx * PW) * a.N;
    return q;
  };
  auto dma_in = [&](const Tile q, int buf) {
    const int iy0 = 2 * (2 * q.ty * PH) - PAD, ix0 = 2 * (2 * q.tx * PW) - PAD;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const float* src = q.xorg + in_off[it];
      if (!q.inner) {                                                     // uniform: border tile
        const int key = in_key[it];
        const bool ok = key >= 0 && (unsigned)(iy0 + (key >> 8)) < (unsigned)a.Hin && (unsigned)(ix0 + (key & 255)) < (unsigned)a.Win;
        src = ok ? src : sqd_stem_zero;
      }
      __builtin_amdgcn_global_load_lds(src, (stem_lds_ptr_t)(inB + buf * INSLOTS + it * NPROD + (wave_s & 3) * 64), 4, 0, 0);
    }
  };

  float wreg[WREG ? KSTEPS : 1][WREG ? NT : 1];
  int kreg[WREG ? KSTEPS : 1];
  if (WREG) {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int kk = 4 * s + g;
      const int k = kk < K ? kk : 0;
      const int ci = k / (KS * KS), rem = k - ci * (KS * KS), ky = rem / KS, kx = rem - ky * KS;
      kreg[s] = (ci * IH + ky) * IW + kx;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + lr;
        wreg[s][j] = (n < a.N && kk < K) ? a.w[(long long)n * K + kk] : 0.f;
      }
    }
  }

  // (WREG) byte address of every k-step's B operand inside input buffer 0: patch pixel base + im2col offset of k = 4s + g
  const char* bofs[WREG ? MT : 1][WREG ? KSTEPS : 1];
  if (WREG) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) bofs[i][s] = (const char*)(inB + pbase[i] + kreg[s]);
  }
  const bool is_prod = (WM != 8) || wave_s < 4;       // uniform
  const bool is_pool = (WM != 8) || wave_s >= 4;
  // XCD-contiguous walk: horizontally adjacent tiles share two of the three 128-byte lines every 140-byte patch row touches and
  // vertically adjacent ones 3 of 15 rows; with the plain id walk the eight neighbours of a run sat on eight XCDs and every
  // L2 fetched its own copy (375 MB per batch of 20 for a 115 MB image, profiles/traffic.json of round 2)
  int tile = sqd_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  if (tile >= a.ntiles) return;
  Tile cur = tile_at(tile);
  if (is_prod) dma_in(cur, 0);
  int buf = 0;

  for (;;) {
    const int ntile = tile + (int)gridDim.x;
    const bool has_next = ntile < a.ntiles;
    const Tile nxt = tile_at(has_next ? ntile : tile);
    // tile barrier: this tile's input landed (DMA waves: vmcnt(0)); previous tile's pooling left convT (LDS reads done)
    if (WM == 8) {
      if (is_prod) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      __syncthreads();
    }
    if (has_next && is_prod) dma_in(nxt, buf ^ 1);
    const float* inT = inB + buf * INSLOTS;
    const int cy0 = 2 * cur.ty * PH, cx0 = 2 * cur.tx * PW;       // conv coords of the patch origin

    // ---- conv patch on the matrix cores ----
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (WREG) {
      // 3x3 stem: the im2col addresses of all k-steps are per-lane constants (bofs, computed once before the tile loop); the
      // input buffer is a compile-time offset inside each of the two copies of this phase, so every operand read is
      // ds_read_b32 base + immediate, all of them are issued before the first MFMA, and the phase holds NO vector ALU
      // instruction (it had an add + add3 per k-step and waited for each read in turn; every VALU instruction delays the
      // matrix pipe, DESIGN.md cost model)
      auto mfma_phase = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        float bf[KSTEPS][MT];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const float v = *(const float*)(bofs[i][s] + BUF * INSLOTS * 4);
            bf[s][i] = (4 * s + 3 < K) ? v : ((4 * s + g < K) ? v : 0.f);     // padded k of the last step: weights are 0 there too
          }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(wreg[s][j], bf[s][i], acc[i][j]);
      };
      if (buf) mfma_phase(std::integral_constant<int, 1>{}); else mfma_phase(std::integral_constant<int, 0>{});
    } else {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int koff = koffL[4 * s + g];
      float bf[MT], af[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (4 * s + 3 < K) bf[i] = inT[pbase[i] + koff];
        else bf[i] = (4 * s + g < K) ? inT[pbase[i] + koff] : 0.f;        // padded k of the last step: weights are 0 there too
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) af[j] = wT[(j * 16 + lr) * KW + 4 * s + g];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j], bf[i], acc[i][j]);
    }
    }
    // bias + ReLU -> LDS; positions outside the conv feature map become 0 (neutral for max of ReLU outputs)
    const bool conv_inner = cy0 + CH <= a.Ho && cx0 + CW <= a.Wo;       // uniform
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int p = (wave + WM * i) * 16 + lr;
      bool inside = true;
      if (!conv_inner) inside = pkey[i] >= 0 && (cy0 + (pkey[i] >> 8) < a.Ho) && (cx0 + (pkey[i] & 255) < a.Wo);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + 4 * g;
        f32x4 v;
        if constexpr (ARGMAX) {
          v = acc[i][j] + *(const f32x4*)(biasL + n);
          // ReLU as ONE v_max per element (fmaxf costs a canonicalising max in front; every VALU op delays the MFMAs)
          asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3"
                       : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
          if (!conv_inner) {                                            // uniform: only border tiles mask per lane
            asm volatile("" ::: "memory");
            if (!inside) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        } else {
          // inference: bias and ReLU move BEHIND the pool -- max_i fl(x_i + b) == fl(max_i x_i + b) (rounding is monotone) and
          // relu(max) == max(relu), bit for bit -- so they run on the 24 pooled pixels of a tile instead of its 119 conv
          // pixels (every VALU instruction delays the matrix pipe); positions outside the conv map are -inf here
          v = acc[i][j];
          if (!conv_inner) {
            asm volatile("" ::: "memory");                                // keep the border path a real scalar branch: as a select
                                                                          // it cost 16 v_cndmask per tile on EVERY tile
            if (!inside) v = (f32x4){-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
          }
        }
        *(f32x4*)(convT + p * CP + n) = v;
      }
    }
    // publish convT: wait for this wave's LDS writes only -- __syncthreads() would also wait (vmcnt(0)) for the next
    // tile's input DMA issued above, exposing its latency once per tile
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- 3x3 / stride-2 max pool out of LDS ----
    const int py0 = cur.ty * PH, px0 = cur.tx * PW;
    const bool pool_inner = py0 + PH <= a.Hp && px0 + PW <= a.Wp;       // uniform
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
      const int key = is_pool ? pl_key[it] : -1;
      if (key < 0) continue;
      const int pr = key >> 8, pc = key & 255;
      if (!pool_inner && (py0 + pr >= a.Hp || px0 + pc >= a.Wp)) continue;
      const float* cv = convT + pl_lds[it];
      const long long o = cur.obase + pl_out[it];
      if (!ARGMAX) {
        f32x4 w[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) w[dy * 3 + dx] = *(const f32x4*)(cv + (dy * CW + dx) * CP);
        f32x4 m;
        m.x = fmaxf(fmaxf(fmaxf(fmaxf(w[0].x, w[1].x), w[2].x), fmaxf(fmaxf(w[3].x, w[4].x), w[5].x)), fmaxf(fmaxf(w[6].x, w[7].x), w[8].x));
        m.y = fmaxf(fmaxf(fmaxf(fmaxf(w[0].y, w[1].y), w[2].y), fmaxf(fmaxf(w[3].y, w[4].y), w[5].y)), fmaxf(fmaxf(w[6].y, w[7].y), w[8].y));
        m.z = fmaxf(fmaxf(fmaxf(fmaxf(w[0].z, w[1].z), w[2].z), fmaxf(fmaxf(w[3].z, w[4].z), w[5].z)), fmaxf(fmaxf(w[6].z, w[7].z), w[8].z));
        m.w = fmaxf(fmaxf(fmaxf(fmaxf(w[0].w, w[1].w), w[2].w), fmaxf(fmaxf(w[3].w, w[4].w), w[5].w)), fmaxf(fmaxf(w[6].w, w[7].w), w[8].w));
        m += *(const f32x4*)(biasL + pl_ch[it]);
        asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3"
                     : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
        *(f32x4*)(a.y + o) = m;
      } else {
        // max by v_max3, then the FIRST window position holding it (scan t = 8..0, the smallest match written last).
        // Taps outside the conv map hold 0 (epilogue) and sit after tap 0 in scan order, and a maximum of 0 means every
        // tap is 0, so a clipped tap is never reported.  NaN cannot occur here: the epilogue's ReLU maps it to 0.
        f32x4 w[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) w[dy * 3 + dx] = *(const f32x4*)(cv + (dy * CW + dx) * CP);
        f32x4 m;
        m.x = fmaxf(fmaxf(fmaxf(fmaxf(w[0].x, w[1].x), w[2].x), fmaxf(fmaxf(w[3].x, w[4].x), w[5].x)), fmaxf(fmaxf(w[6].x, w[7].x), w[8].x));
        m.y = fmaxf(fmaxf(fmaxf(fmaxf(w[0].y, w[1].y), w[2].y), fmaxf(fmaxf(w[3].y, w[4].y), w[5].y)), fmaxf(fmaxf(w[6].y, w[7].y), w[8].y));
        m.z = fmaxf(fmaxf(fmaxf(fmaxf(w[0].z, w[1].z), w[2].z), fmaxf(fmaxf(w[3].z, w[4].z), w[5].z)), fmaxf(fmaxf(w[6].z, w[7].z), w[8].z));
        m.w = fmaxf(fmaxf(fmaxf(fmaxf(w[0].w, w[1].w), w[2].w), fmaxf(fmaxf(w[3].w, w[4].w), w[5].w)), fmaxf(fmaxf(w[6].w, w[7].w), w[8].w));
        int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
        for (int tt = 8; tt >= 1; --tt) {
          ax = (w[tt].x == m.x) ? tt : ax; ay = (w[tt].y == m.y) ? tt : ay;
          az = (w[tt].z == m.z) ? tt : az; aw = (w[tt].w == m.w) ? tt : aw;
        }
        ax = (w[0].x == m.x) ? 0 : ax; ay = (w[0].y == m.y) ? 0 : ay; az = (w[0].z == m.z) ? 0 : az; aw = (w[0].w == m.w) ? 0 : aw;
        // ... and the ReLU mask rides in the code: 15 where the pooled value is 0 (every tap of the window was <= 0 before the
        // ReLU), so the backward never needs the pooled tensor for its sign (see maxpool_fwd_kernel<.., RELUMASK>)
        ax = m.x > 0.f ? ax : 15; ay = m.y > 0.f ? ay : 15; az = m.z > 0.f ? az : 15; aw = m.w > 0.f ? aw : 15;
        *(f32x4*)(a.y + o) = m;
        *(uint32_t*)(a.amax + o) = (uint32_t)ax | ((uint32_t)ay << 8) | ((uint32_t)az << 16) | ((uint32_t)aw << 24);
      }
    }
    if (!has_next) break;
    tile = ntile; cur = nxt; buf ^= 1;
  }
}

template <int KS, int PAD, int NT, bool ARGMAX, int WM>
static int launch_stem_pool_t(StemPoolArgs a, hipStream_t s) {
  constexpr int PH = 3, PW = 8, CH = 2 * PH + 1, CW = 2 * PW + 1, NSUB = (CH * CW + 15) / 16;
  constexpr int K = 3 * KS * KS, KSTEPS = (K + 3) / 4, KW = KSTEPS * 4 + 1, NTHR = WM * 64;
  constexpr int IH = 2 * (CH - 1) + KS, IW = 2 * (CW - 1) + KS;
  constexpr int NPROD = (WM == 8) ? 256 : NTHR;
  constexpr int INSLOTS = (3 * IH * IW + NPROD - 1) / NPROD * NPROD;
  constexpr bool WREG = KSTEPS * NT <= 32;
  constexpr size_t lds = (size_t)(NSUB * 16 * (16 * NT + 4) + 2 * INSLOTS + 4 + (WREG ? 0 : 16 * NT * KW) + 16 * NT + KSTEPS * 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "stem_pool LDS budget");
  static_assert(IH < 256 && IW < 256, "slot keys pack row/col in 8 bits");
  a.tiles_x = sqd_cdiv(a.Wp, PW); a.tiles_y = sqd_cdiv(a.Hp, PH);
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
  if ((long long)a.ntiles * (a.tiles_x > a.tiles_y ? a.tiles_x : a.tiles_y) >= (1ll << 32)) return SQD_ERR_UNSUPPORTED;
  a.tiles_x_m = a.tiles_x > 1 ? (unsigned)(((1ull << 32) + a.tiles_x - 1) / a.tiles_x) : 0u;
  a.tiles_y_m = a.tiles_y > 1 ? (unsigned)(((1ull << 32) + a.tiles_y - 1) / a.tiles_y) : 0u;
  auto kern = stem_pool_kernel<KS, PAD, NT, ARGMAX, WM>;
  static int wgs_per_cu = 0;
  static SqdDevOnce lds_once;
  if (lds > 64 * 1024 && sqd_max_lds_once(lds_once, (const void*)kern, (int)lds) != SQD_OK) return SQD_ERR_LAUNCH;
  if (wgs_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NTHR, lds) != hipSuccess || nb < 1) nb = 1;
    wgs_per_cu = nb > 4 ? 4 : nb;
  }
  int dev = 0, cus = 256; hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const int slots = cus * wgs_per_cu;
  const int per_wg = sqd_cdiv(a.ntiles, slots);
  const int gx = sqd_cdiv(a.ntiles, per_wg);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(NTHR), lds, s, a);
  return sqd_launch_status();
}

template <int KS, int PAD, int NT>
static int launch_stem_pool(StemPoolArgs a, hipStream_t s) {
  constexpr int WM = SQD_STEM_WAVES;
  return a.amax ? launch_stem_pool_t<KS, PAD, NT, true, WM>(a, s) : launch_stem_pool_t<KS, PAD, NT, false, WM>(a, s);
}

// ---------------------------------------------------------------------------------------------
// Wave-autonomous fused stem (round 3, the 3x3 / 64-channel stem of src/model/squeezedet.py:34-36 at inference): same
// arithmetic as stem_pool_kernel<3,1,4,false,8> -- a k-ordered fp32 MFMA chain from 0, max over the window, THEN bias and
// ReLU -- so the two kernels agree bit for bit, but no workgroup-level cooperation at all:
//   * a WAVE owns a pooled tile of PH rows x 7 columns = a conv patch of (2 PH + 1) rows x 15 columns: one MFMA column
//     block (16 conv columns, the 16th unused) per conv row, 4 channel blocks -> 4 (2 PH + 1) accumulators;
//   * its NCHW input patch ((4 PH + 3) rows x 36 floats x 3 planes, 16-byte slots: the patch starts 4 floats left of the
//     first tap so that every row segment is 16-byte aligned in a 4-float-aligned image row) arrives by buffer-resource
//     LDS-DMA in a wave-private double buffer, retired by a counted s_waitcnt vmcnt; slots outside the image carry an
//     out-of-range offset and are zero-filled by the hardware (the convolution's padding);
//   * im2col = the LDS read address of the B operand (per-lane base + immediates), weights in registers;
//   * the 3x3 / stride-2 pool runs in REGISTERS: vertical max = v_max3 across the accumulators of three conv rows (same
//     lane), horizontal max = two v_max_f32 with DPP row_shl:1 / row_shl:2 operands (lane = conv column), bias + ReLU on
//     the pooled values, one 16-byte store per (pooled row, channel block) from the 7 x 4 lanes that hold a pooled pixel.
// The workgroup kernel above spends 7700 cycles per tile of which 1792 issue MFMAs: two workgroup barriers per tile and
// the conv-patch round trip through LDS into the pool; here nothing waits for another wave.
// Needs Win % 4 == 0 and a 16-byte aligned image (else the launcher keeps the workgroup kernel).
// ---------------------------------------------------------------------------------------------
struct StemWaveArgs {
  const float* x; const float* w; const float* bias; float* y; uint8_t* amax;
  const float* wsq; const float* bsq;       // SQ variant: the next Fire's squeeze (1x1, 64 -> SQ channels, OIHW + bias); y is ITS output
  float* ysq;                               // ARGMAX + SQ (training): y and amax as without SQ, the squeeze output goes HERE
  int B, Hin, Win, Ho, Wo, Hp, Wp;
  int tiles_x, tiles_y, ntiles;
  unsigned tiles_x_m, tiles_y_m;
};

// CB = 1: one 16-column block per conv row, lane = conv column, 7 pooled columns per tile (lanes 0, 2, .., 12 end with a pooled
// pixel).  CB = 2: TWO column blocks per conv row, split by column PARITY -- block E holds conv columns 2m, block O columns 2m + 1
// (im2col is only an LDS read address, so the split is free) -- so that lane m's window is E[m], O[m], E[m + 1]: one plain max,
// one DPP max, and 15 of 16 lanes end with a pooled pixel (2.3x fewer vector instructions per pooled pixel than CB = 1; every
// vector instruction delays the SIMD's matrix pipe).
// ARGMAX (training forward, CB = 2 only): additionally writes the uint8 code of the FIRST window position (row-major) holding the
// pooled value, or 15 where the pooled value is not > 0 -- the codes of stem_pool_kernel<.., ARGMAX = true> / maxpool_fwd_kernel<true,
// true>.  Positions are compared AFTER the bias add (rounding can create ties that the raw sums do not have); the ReLU needs no
// compare of its own (a window whose maximum is <= 0 is code 15 whatever its arg-max).  Column E / O of the window: first row holding
// the column maximum by two compares, candidates 3 dy + dx of the three columns (the third arrives by DPP from lane + 1) merged by
// one v_min3.
// SQ > 0 (inference, CB = 2): the first Fire's squeeze (1x1 conv 64 -> SQ = 16 channels + bias + ReLU, src/model/squeezedet.py:17-18)
// is applied to the pooled pixels before they leave the registers, and ONLY its output is written (y: NHWC [B][Hp][Wp][SQ]): a
// pooled lane holds channels 16j + 4g + t of its pixel = the B operand (k = g, column = pixel) of v_mfma_f32_16x16x4_f32 for input
// channel 16j + 4g + t, so the squeeze is 16 MFMAs per pooled row against squeeze weights pre-arranged as A operands in LDS
// (sqA[j][lane] = W[n = lane & 15][16j + 4 (lane >> 4) + 0..3]) -- the Fire bridges' trick (wino_bridge.h).  The 153 MB pooled tensor is
// neither written nor read back by a separate squeeze launch.
// ARGMAX and SQ together (training forward): the pooled tensor and its codes are stored as in the ARGMAX form (the backward needs them)
// and the squeeze output goes to a.ysq; the squeeze launch that would re-read the 153 MB pooled tensor is gone.
template <int PH, int CB, bool ARGMAX = false, int SQ = 0>
__global__ __launch_bounds__(256, 2) void stem_wave_kernel(StemWaveArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PW = CB == 1 ? 7 : 15, CH = 2 * PH + 1, IH = 2 * CH + 1;
  constexpr int SL = CB == 1 ? 9 : 17, RP = 4 * SL;                                  // 16-byte slots / floats per patch row
  constexpr int NSLOT = 3 * IH * SL, N_IT = (NSLOT + 63) / 64, BUFF = N_IT * 64 * 4; // floats per buffer
  constexpr int K = 27, KSTEPS = 7, NT = 4, N = 64;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NST = ARGMAX ? PH * NT * 2 + (SQ ? PH : 0) : (SQ ? PH : PH * NT);    // stores per tile (always issued)
  static_assert(!ARGMAX || CB == 2, "the arg-max epilogue is written for the parity-split layout");
  static_assert(SQ == 0 || (SQ == 16 && CB == 2), "fused squeeze: 16 channels, parity-split layout");
  constexpr int NO = (SQ && !ARGMAX) ? SQ : N;                                       // channels per pixel of the tensor behind a.y
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const bufW = smem + wave_s * (2 * BUFF + N);  // [2][BUFF] patch images + the wave's own copy of the bias
  float* const biasW = bufW + 2 * BUFF;
  float* const sqA = smem + 4 * (2 * BUFF + N);        // (SQ) [4 channel blocks][64 lanes] f32x4 A operands + [SQ] squeeze bias: one copy per workgroup
  if constexpr (SQ > 0) {
    {
      const int j = tid >> 6;                          // 256 threads = 4 blocks x 64 lanes
      const int n = lane & 15, ch = 16 * j + 4 * (lane >> 4);
      *(f32x4*)(sqA + tid * 4) = *(const f32x4*)(a.wsq + n * N + ch);      // OIHW 1x1: W[n][ch .. ch + 3] contiguous
    }
    if (tid < SQ) sqA[1024 + tid] = a.bsq ? a.bsq[tid] : 0.f;
    __syncthreads();                                   // the only workgroup barrier (before any wave may leave)
  }
  typedef __attribute__((address_space(3))) const char* lds_cptr_t;   // 32-bit LDS addresses (a generic pointer costs two registers)

  // ---- per-lane constants ----
  int d_off[N_IT];                                     // byte offset of the lane's 16-byte slot from the patch origin
#pragma unroll
  for (int it = 0; it < N_IT; ++it) {
    const int slot = it * 64 + lane;
    const bool real = slot < NSLOT;
    const int ci = slot / (IH * SL), rem = slot - ci * (IH * SL), row = rem / SL, k4 = rem - row * SL;
    d_off[it] = real ? ((ci * a.Hin + row) * a.Win + 4 * k4) * 4 : (int)OOB;
  }
  float wreg[KSTEPS][NT];
  lds_cptr_t bp[KSTEPS];                               // B operand of k-step s, conv row 0, (even) column block, buffer 0
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int kk = 4 * s + g;
    const int k = kk < K ? kk : 0;                     // the padded k multiplies a real (finite) tap with a zero weight
    const int ci = k / 9, r9 = k - ci * 9, dy = r9 / 3, dx = r9 - dy * 3;
    bp[s] = (lds_cptr_t)(bufW + (ci * IH + dy) * RP + 2 * CB * c + dx + 3);
#pragma unroll
    for (int j = 0; j < NT; ++j) wreg[s][j] = kk < K ? a.w[(j * 16 + c) * K + kk] : 0.f;
  }
  biasW[lane] = a.bias ? a.bias[lane] : 0.f;           // read back by this wave only (epilogue): no barrier
  const f32x4* const biasL = (const f32x4*)(biasW + 4 * g);
  // the patch origin (input row 4 PH ty - 1, column 4 PW tx - 4) is never negative relative to this base
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - (a.Win + 4)), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc((void*)((ARGMAX && SQ) ? a.ysq : a.y), 0, 0x7ffffff0, 0x00020000);
  // codes: one byte per element; idle lanes carry 2^29 (a quarter of the fp32 streams' idle offset), past this resource's range
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(ARGMAX ? (void*)a.amax : (void*)a.y), 0,
                                                                        ARGMAX ? a.B * a.Hp * a.Wp * N : 0x7ffffff0, 0x00020000);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  auto store16 = [&](f32x4 v, int voff, int soff) {    // (MUBUF store + SGPR soffset write-after-read hazard: see conv_wino.hip)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yres, voff, soff, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // lanes that end up with a pooled pixel: conv columns 0, 2, .., 12 (CB = 1) / pooled columns 0..14 (CB = 2)
  const bool out_lane = CB == 1 ? ((c & 1) == 0 && c < 2 * PW) : (c < PW);
  const int pcol = CB == 1 ? (c >> 1) : c;
  const int o_voff = out_lane ? (pcol * NO + 4 * g) * 4 : (int)OOB;
  const int q_voff = out_lane ? (pcol * SQ + 4 * g) * 4 : (int)OOB;      // (ARGMAX + SQ) the lane's slot in a row of the squeeze output

  struct Tile { int ty, tx, inner; unsigned soff, osoff; };
  auto tile_at = [&](int t) {
    Tile q;
    const int t1 = a.tiles_x_m ? (int)__umulhi((unsigned)t, a.tiles_x_m) : t;
    q.tx = t - t1 * a.tiles_x;
    const int b = a.tiles_y_m ? (int)__umulhi((unsigned)t1, a.tiles_y_m) : t1;
    q.ty = t1 - b * a.tiles_y;
    const int iy0 = 4 * PH * q.ty - 1, ix0 = 4 * PW * q.tx - 4;
    q.soff = (unsigned)(((b * 3 * a.Hin + 4 * PH * q.ty) * a.Win + 4 * PW * q.tx) * 4);
    q.inner = iy0 >= 0 && iy0 + IH <= a.Hin && ix0 >= 0 && ix0 + RP <= a.Win;
    q.osoff = (unsigned)((((b * a.Hp + q.ty * PH) * a.Wp + q.tx * PW) * NO) * 4);
    return q;
  };
  auto dma_in = [&](const Tile q, int buf) {
    const int iy0 = 4 * PH * q.ty - 1, ix0 = 4 * PW * q.tx - 4;
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      int off = d_off[it];
      if (!q.inner) {                                  // uniform: border tile (the slot's row / column are recomputed here only)
        const int slot = it * 64 + lane;
        const int ci = slot / (IH * SL), rem = slot - ci * (IH * SL), row = rem / SL, k4 = rem - row * SL;
        const bool ok = slot < NSLOT && (unsigned)(iy0 + row) < (unsigned)a.Hin && (unsigned)(ix0 + 4 * k4) < (unsigned)a.Win;
        off = ok ? off : (int)OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (stem_lds_ptr_t)(bufW + buf * BUFF + it * 256), 16, off, (int)q.soff, 0, 0);
    }
  };

  // tile walk: the four waves of a workgroup take four neighbouring tiles, the workgroups of one XCD a contiguous run
  const int tstride = (int)gridDim.x * 4;
  int tile = sqd_xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * 4 + wave_s;
  if (tile >= a.ntiles) return;                        // (no barrier anywhere in this kernel)
  Tile cur = tile_at(tile);
  dma_in(cur, 0);
  int buf = 0;
  bool first = true;
  for (;;) {
    const int ntile = tile + tstride;
    const bool has_next = ntile < a.ntiles;
    const Tile nxt = tile_at(has_next ? ntile : tile);
    // this tile's patch has landed: everything but the previous tile's NST stores (issued after the patch request) is done
    __builtin_amdgcn_sched_barrier(0);
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    first = false;
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) dma_in(nxt, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);

    // ---- conv patch on the matrix cores: conv row r = CB 16-column blocks x 4 channel blocks ----
    f32x4 acc[CH][CB][NT];
    auto mfma_phase = [&](auto bufc) {
      constexpr int BO = decltype(bufc)::value * BUFF * 4;
      float bf[2][CB][KSTEPS];
      auto load_row = [&](int r) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) bf[r & 1][cb][s] = *(__attribute__((address_space(3))) const float*)(bp[s] + BO + r * (2 * RP * 4) + cb * 8);
      };
      load_row(0);
#pragma unroll
      for (int r = 0; r < CH; ++r) {
        if (r + 1 < CH) load_row(r + 1);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[r][cb][j] = mfma16(wreg[s][j], bf[r & 1][cb][s], s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[r][cb][j]);
      }
    };
    if (buf) mfma_phase(std::integral_constant<int, 1>{}); else mfma_phase(std::integral_constant<int, 0>{});

    // ---- conv positions outside the feature map are -inf for the pool (border tiles only) ----
    const int cy0 = 2 * PH * cur.ty, cx0 = 2 * PW * cur.tx;
    const int py0 = PH * cur.ty, px0 = PW * cur.tx;
    int voff = o_voff, qvoff = q_voff;
    if (!(cy0 + CH <= a.Ho && cx0 + 16 * CB <= a.Wo && px0 + PW <= a.Wp)) {      // uniform
      const float ninf = -__builtin_inff();
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const bool col_out = cx0 + (CB == 1 ? c : 2 * c + cb) >= a.Wo;
#pragma unroll
        for (int r = 0; r < CH; ++r) {
          const bool out = col_out || cy0 + r >= a.Ho;
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (out) acc[r][cb][j] = (f32x4){ninf, ninf, ninf, ninf};
        }
      }
      voff = (out_lane && px0 + pcol < a.Wp) ? o_voff : (int)OOB;
      qvoff = (out_lane && px0 + pcol < a.Wp) ? q_voff : (int)OOB;
    }
    // ---- pool in registers, bias + ReLU behind it, stores (always issued: rows / lanes without a pixel go out of range) ----
    if constexpr (ARGMAX) {
      // bias first: the arg-max is taken over the values the reference's pool sees (fl(conv + bias); the ReLU cannot reorder them)
#pragma unroll
      for (int r = 0; r < CH; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[r][cb][j] += biasL[j * 4];
    }
    auto vmax3 = [&](const f32x4 r0, const f32x4 r1, const f32x4 r2) {
      f32x4 v;
      asm volatile("v_max3_f32 %0, %4, %8, %12\n\tv_max3_f32 %1, %5, %9, %13\n\tv_max3_f32 %2, %6, %10, %14\n\tv_max3_f32 %3, %7, %11, %15"
                   : "=&v"(v.x), "=&v"(v.y), "=&v"(v.z), "=&v"(v.w)
                   : "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r0.w), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w),
                     "v"(r2.x), "v"(r2.y), "v"(r2.z), "v"(r2.w));
      return v;
    };
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      const int vrow = (py0 + i < a.Hp) ? voff : (int)OOB;
      const int soff = (int)(cur.osoff + (unsigned)(i * a.Wp * NO * 4));
      f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};          // (SQ) the pooled row's squeeze output: rows 4g .. 4g + 3 of pixel column c
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 h;
        // (row_shl:n reads lane + n of the 16-lane row; lanes past its end read 0 and only feed lanes that hold no pooled pixel.
        //  s_nop 1: a DPP operand may not be read in the two wait states behind the VALU write of that register.)
        if constexpr (ARGMAX) {
          const f32x4 e0 = acc[2 * i][0][j], e1 = acc[2 * i + 1][0][j], e2 = acc[2 * i + 2][0][j];
          const f32x4 o0 = acc[2 * i][1][j], o1 = acc[2 * i + 1][1][j], o2 = acc[2 * i + 2][1][j];
          const f32x4 ce = vmax3(e0, e1, e2), co = vmax3(o0, o1, o2);
          unsigned codes = 0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const unsigned ke = e0[t] == ce[t] ? 0u : (e1[t] == ce[t] ? 3u : 6u);          // 3 dy of the first row holding the column maximum
            const unsigned ko = o0[t] == co[t] ? 1u : (o1[t] == co[t] ? 4u : 7u);
            // lane + 1's column maximum and its row code, as ISA: with __builtin_amdgcn_update_dpp the compiler (ROCm 7.2) kept ONE
            // shifted copy of element 0's maximum and reused it for elements 1..3 of the quad (wrong pooled values in 3 of 4
            // channels; found by the bit-equality check against the workgroup kernel)
            float ce2; unsigned ke2;
            asm volatile("s_nop 1\n\t"
                         "v_mov_b32_dpp %0, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %1, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(ce2), "=&v"(ke2) : "v"(ce[t]), "v"(ke));
            ke2 += 2u;
            const float m = __builtin_fmaxf(__builtin_fmaxf(ce[t], co[t]), ce2);
            const unsigned k0 = ce[t] == m ? ke : 9u, k1 = co[t] == m ? ko : 9u, k2 = ce2 == m ? ke2 : 9u;
            unsigned code = k0 < k1 ? k0 : k1;
            code = code < k2 ? code : k2;
            code = m > 0.f ? code : 15u;
            codes |= code << (8 * t);
            h[t] = m;
            __builtin_amdgcn_sched_barrier(0);       // one element at a time: interleaving the four chains costs ~30 registers (spills)
          }
          asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3"
                       : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w));
          if constexpr (SQ > 0) {
            const f32x4 aw = *(const f32x4*)(sqA + (j * 64 + lane) * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc = mfma16(aw[t], h[t], sacc);
          }
          store16(h, vrow + j * 64, soff);
          // (vrow >> 2: byte offset of the lane's four codes; an idle lane's 2^31 becomes 2^29, past the code resource's range)
          __builtin_amdgcn_raw_buffer_store_b32(codes, ares, (int)((unsigned)vrow >> 2) + j * 16, (int)((unsigned)soff >> 2), 0);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 1" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        if constexpr (CB == 1) {
          const f32x4 v = vmax3(acc[2 * i][0][j], acc[2 * i + 1][0][j], acc[2 * i + 2][0][j]);
          asm volatile("s_nop 1\n\t"
                       "v_max_f32_dpp %0, %4, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %1, %5, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %2, %6, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %3, %7, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %0, %4, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %1, %5, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %2, %6, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %3, %7, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                       : "=&v"(h.x), "=&v"(h.y), "=&v"(h.z), "=&v"(h.w)
                       : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        } else {
          const f32x4 ve = vmax3(acc[2 * i][0][j], acc[2 * i + 1][0][j], acc[2 * i + 2][0][j]);
          const f32x4 vo = vmax3(acc[2 * i][CB - 1][j], acc[2 * i + 1][CB - 1][j], acc[2 * i + 2][CB - 1][j]);
          asm volatile("v_max_f32 %0, %4, %8\n\tv_max_f32 %1, %5, %9\n\tv_max_f32 %2, %6, %10\n\tv_max_f32 %3, %7, %11\n\t"
                       "v_max_f32_dpp %0, %4, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %1, %5, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %2, %6, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_max_f32_dpp %3, %7, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                       : "=&v"(h.x), "=&v"(h.y), "=&v"(h.z), "=&v"(h.w)
                       : "v"(ve.x), "v"(ve.y), "v"(ve.z), "v"(ve.w), "v"(vo.x), "v"(vo.y), "v"(vo.z), "v"(vo.w));
        }
        h += biasL[j * 4];
        asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3"
                     : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w));
        if constexpr (SQ > 0) {
          const f32x4 aw = *(const f32x4*)(sqA + (j * 64 + lane) * 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) sacc = mfma16(aw[t], h[t], sacc);
        } else {
          store16(h, vrow + j * 64, soff);
        }
      }
      if constexpr (SQ > 0) {
        sacc += *(const f32x4*)(sqA + 1024 + 4 * g);
        asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3"
                     : "+v"(sacc.x), "+v"(sacc.y), "+v"(sacc.z), "+v"(sacc.w));
        if constexpr (ARGMAX) {
          // the squeeze output is [B][Hp][Wp][SQ]: a quarter of the pooled tensor's offsets (N = 4 SQ)
          const int qrow = (py0 + i < a.Hp) ? qvoff : (int)OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, sacc), qres, qrow, (int)((unsigned)soff >> 2), 0);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 1" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        } else {
          store16(sacc, vrow, soff);
        }
      }
    }
    if (!has_next) break;
    tile = ntile; cur = nxt; buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <int PH, int CB, bool ARGMAX = false, int SQ = 0>
static int launch_stem_wave(StemWaveArgs a, hipStream_t s) {
  constexpr int PW = CB == 1 ? 7 : 15, CH = 2 * PH + 1, IH = 2 * CH + 1, SL = CB == 1 ? 9 : 17;
  constexpr int NSLOT = 3 * IH * SL, N_IT = (NSLOT + 63) / 64;
  constexpr size_t lds = (size_t)4 * (2 * N_IT * 64 * 16 + 64 * 4) + (SQ ? (1024 + 16) * 4 : 0);
  constexpr int OCC = 2;                                          // workgroups per CU (= waves per SIMD) the kernel is built for
  static_assert(OCC * lds <= 160 * 1024, "OCC workgroups per CU");
  a.tiles_x = sqd_cdiv(a.Wp, PW); a.tiles_y = sqd_cdiv(a.Hp, PH);
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
  a.tiles_x_m = a.tiles_x > 1 ? (unsigned)(((1ull << 32) + a.tiles_x - 1) / a.tiles_x) : 0u;
  a.tiles_y_m = a.tiles_y > 1 ? (unsigned)(((1ull << 32) + a.tiles_y - 1) / a.tiles_y) : 0u;
  auto kern = stem_wave_kernel<PH, CB, ARGMAX, SQ>;
  static SqdDevOnce attr_once;                 // (per device: ADVICE round 4)
  if (int rc_attr = lds <= 64 * 1024 ? SQD_OK : sqd_max_lds_once(attr_once, (const void*)kern, (int)lds)) return rc_attr;
  int dev = 0, cus = 256; hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const int wtiles = sqd_cdiv(a.ntiles, 4);                      // workgroup-sized runs of 4 tiles
  const int per_wg = sqd_cdiv(wtiles, OCC * cus);
  const int gx = sqd_cdiv(wtiles, per_wg);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, s, a);
  return sqd_launch_status();
}

// Which kernel runs the inference-mode 3x3 stem: 2 = stem_wave_kernel<2, 2> (default; in the bs=20 step 110.9 us against 154.6 us for
// the workgroup kernel, gpurun_out/r03z), 3 / 4 = stem_wave_kernel<PH, 1> (117.4 / 113.5 us), 0 = the workgroup kernel
// stem_pool_kernel.  SQD_STEM_WAVE in the environment overrides it (read per call: the parity tests switch it at run time).
static int stem_wave_variant() {
  const char* e = getenv("SQD_STEM_WAVE");
  return e ? atoi(e) : 2;
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
  const int variant = stem_wave_variant();
  if (ksize == 3 && N == 64 && variant && (Win & 3) == 0 && ((uintptr_t)x & 15) == 0 &&
      (long long)B * 3 * Hin * Win * 4 < (1ll << 31) && (long long)B * a.Hp * a.Wp * N * 4 < (1ll << 31)) {
    StemWaveArgs wa;
    wa.x = x; wa.w = w; wa.bias = bias; wa.y = y; wa.amax = argmax; wa.wsq = nullptr; wa.bsq = nullptr; wa.ysq = nullptr; wa.B = B; wa.Hin = Hin; wa.Win = Win; wa.Ho = a.Ho; wa.Wo = a.Wo; wa.Hp = a.Hp; wa.Wp = a.Wp;
    // training forward: one pooled row per tile -- with two (variant 5) the arg-max epilogue's temporaries spill 42 registers, and
    // every reload queues behind the next patch's DMA in vmcnt order
    if (argmax) return variant == 5 ? launch_stem_wave<2, 2, true>(wa, s) : launch_stem_wave<1, 2, true>(wa, s);
    return variant == 4 ? launch_stem_wave<4, 1>(wa, s) : (variant == 3 ? launch_stem_wave<3, 1>(wa, s) : launch_stem_wave<2, 2>(wa, s));
  }
  if (ksize == 3 && N == 64) return launch_stem_pool<3, 1, 4>(a, s);
  if (ksize == 7 && N == 96) return launch_stem_pool<7, 3, 6>(a, s);
  return SQD_ERR_UNSUPPORTED;
}

// features[0..2] + the first Fire's squeeze in one launch (inference): x NCHW [B,3,Hin,Win] -> y NHWC [B,Hp,Wp,nsq] =
// ReLU(squeeze(MaxPool(3,2,ceil)(ReLU(conv(x))))).  Only the 3x3 / 64-channel stem with a 16-channel squeeze, Win % 4 == 0 and a
// 16-byte aligned image (SQD_ERR_UNSUPPORTED otherwise: the caller keeps the two launches).
extern "C" int sqd_stem_pool_squeeze_fwd(const float* x, const float* w, const float* bias, const float* wsq, const float* bsq,
                                         float* y, int B, int Hin, int Win, int N, int ksize, int nsq, void* stream) {
  SQD_CHECK_ARG(x && w && wsq && y && B > 0 && Hin > 0 && Win > 0);
  SQD_CHECK_ARG(((uintptr_t)y & 15) == 0 && ((uintptr_t)wsq & 15) == 0);
  if (!(ksize == 3 && N == 64 && nsq == 16 && (Win & 3) == 0 && ((uintptr_t)x & 15) == 0)) return SQD_ERR_UNSUPPORTED;
  StemWaveArgs wa;
  wa.x = x; wa.w = w; wa.bias = bias; wa.y = y; wa.amax = nullptr; wa.wsq = wsq; wa.bsq = bsq; wa.ysq = nullptr; wa.B = B; wa.Hin = Hin; wa.Win = Win;
  wa.Ho = (Hin + 2 - 3) / 2 + 1; wa.Wo = (Win + 2 - 3) / 2 + 1;
  SQD_CHECK_ARG(wa.Ho >= 3 && wa.Wo >= 3);
  wa.Hp = (wa.Ho - 3 + 1) / 2 + 1; wa.Wp = (wa.Wo - 3 + 1) / 2 + 1;
  if ((long long)B * 3 * Hin * Win * 4 >= (1ll << 31) || (long long)B * wa.Hp * wa.Wp * N * 4 >= (1ll << 31)) return SQD_ERR_UNSUPPORTED;
  return launch_stem_wave<2, 2, false, 16>(wa, (hipStream_t)stream);
}

// Training form of sqd_stem_pool_squeeze_fwd: the pooled tensor y_pooled [B,Hp,Wp,64] and its arg-max / ReLU codes (argmax, one byte
// per element) are stored exactly as by sqd_stem_conv_relu_pool_fwd(argmax != NULL) -- the backward reads both -- and the first
// Fire's squeeze output y_sq [B,Hp,Wp,16] comes out of the same launch.  Same shape limits; SQD_ERR_UNSUPPORTED otherwise.
extern "C" int sqd_stem_pool_squeeze_train_fwd(const float* x, const float* w, const float* bias, const float* wsq, const float* bsq,
                                               float* y_pooled, unsigned char* argmax, float* y_sq, int B, int Hin, int Win, int N,
                                               int ksize, int nsq, void* stream) {
  SQD_CHECK_ARG(x && w && wsq && y_pooled && argmax && y_sq && B > 0 && Hin > 0 && Win > 0);
  SQD_CHECK_ARG(((uintptr_t)y_pooled & 15) == 0 && ((uintptr_t)y_sq & 15) == 0 && ((uintptr_t)wsq & 15) == 0 && ((uintptr_t)argmax & 3) == 0);
  SQD_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0);
  if (!(ksize == 3 && N == 64 && nsq == 16 && (Win & 3) == 0 && ((uintptr_t)x & 15) == 0)) return SQD_ERR_UNSUPPORTED;
  StemWaveArgs wa;
  wa.x = x; wa.w = w; wa.bias = bias; wa.y = y_pooled; wa.amax = argmax; wa.wsq = wsq; wa.bsq = bsq; wa.ysq = y_sq;
  wa.B = B; wa.Hin = Hin; wa.Win = Win;
  wa.Ho = (Hin + 2 - 3) / 2 + 1; wa.Wo = (Win + 2 - 3) / 2 + 1;
  SQD_CHECK_ARG(wa.Ho >= 3 && wa.Wo >= 3);
  wa.Hp = (wa.Ho - 3 + 1) / 2 + 1; wa.Wp = (wa.Wo - 3 + 1) / 2 + 1;
  if ((long long)B * 3 * Hin * Win * 4 >= (1ll << 31) || (long long)B * wa.Hp * wa.Wp * N * 4 >= (1ll << 31)) return SQD_ERR_UNSUPPORTED;
  return launch_stem_wave<1, 2, true, 16>(wa, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, ceil_mode=True), NHWC.  Ho = ceil((H-3)/2)+1 (PyTorch additionally
// drops a last window that would start outside the input; with pad 0 that never happens for H>=3).
// Windows at the bottom/right edge are clipped to the input.  The argmax (0..8 = dy*3+dx, first
// maximum in row-major window order, as PyTorch's CPU kernel scans it) is optionally recorded
// for the backward pass.
// ---------------------------------------------------------------------------------------------
// RELUMASK (training, pool input = a ReLU output): a pooled value that is not > 0 records code 15 instead of its window position.
// The backward routes dy to the window position equal to the code, so such a window passes nothing on -- which IS the ReLU
// backward of the pool's input at the arg-max element (its value is the pooled value; an all-zero window sends its gradient
// to an element whose ReLU mask is 0).  The backward then needs no mask tensor: it used to re-read the whole pool input
// (306 MB at 96x312x128, batch 20) for its sign.
template <bool ARGMAX, bool RELUMASK = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ amax, int B, int H, int W, int C,
                                                          int Ho, int Wo) {
  // One thread = TWO horizontally adjacent output pixels x 4 channels: the windows share a column, so 15 loads serve 2
  // outputs (the kernel is bound by the L2 read amplification of the window gather, 9 loads per output otherwise).
  // blockIdx.y = (b, oy): no 64-bit divisions per element.  Clipped border windows (ceil mode) clamp their tap offsets
  // to the last valid row / column for the value; the argmax only ever considers in-range taps.
  const int cv = C >> 2, Wp = (Wo + 1) >> 1;
  // XCD-contiguous block order: output rows oy and oy + 1 share an input row, and the plain (x, y) order dealt vertically
  // adjacent blocks to different XCDs, so each L2 fetched the shared row for itself (428 MB of fabric traffic per launch against
  // 287 MB algorithmic in the training step, profiles/traffic.json of round 3)
  const int gxn = (int)gridDim.x, nblk = gxn * (int)gridDim.y;
  const int lin = sqd_xcd_contiguous((int)blockIdx.y * gxn + (int)blockIdx.x, nblk);
  const int by = lin / gxn, bx = lin - by * gxn;
  const int t = bx * 256 + threadIdx.x;                    // index inside the output row: oxp * cv + c4
  if (t >= Wp * cv) return;
  const int oxp = t / cv, c4 = t - oxp * cv;
  const int oy = by % Ho, b = by / Ho;
  const int ox0 = 2 * oxp, ix0 = 4 * oxp;
  const float* row = x + ((long long)b * H + 2 * oy) * W * C + (long long)ix0 * C + 4 * c4;
  const int ny = (2 * oy + 3 <= H) ? 3 : H - 2 * oy;                               // rows of the (clipped) windows
  const int ncol = (ix0 + 5 <= W) ? 5 : W - ix0;                                    // columns available to the pair
  f32x4 v[3][5];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int ry = dy < ny ? dy : ny - 1, cx = dx < ncol ? dx : ncol - 1;
      v[dy][dx] = *(const f32x4*)(row + ((long long)ry * W + cx) * C);
    }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ox = ox0 + u;
    if (ox >= Wo) break;
    const int nx = (2 * ox + 3 <= W) ? 3 : W - 2 * ox;
    f32x4 m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (ARGMAX && (dy >= ny || dx >= nx)) continue;                              // clipped tap: not a candidate
        const f32x4 w = v[dy][2 * u + dx];
        if (ARGMAX) {
          const int tt = dy * 3 + dx;
          // NaN propagates like PyTorch: (v > m) || isnan(v)
          if (w.x > m.x || w.x != w.x) { m.x = w.x; ax = tt; }
          if (w.y > m.y || w.y != w.y) { m.y = w.y; ay = tt; }
          if (w.z > m.z || w.z != w.z) { m.z = w.z; az = tt; }
          if (w.w > m.w || w.w != w.w) { m.w = w.w; aw = tt; }
        } else {
          m.x = fmaxf(m.x, w.x); m.y = fmaxf(m.y, w.y); m.z = fmaxf(m.z, w.z); m.w = fmaxf(m.w, w.w);
        }
      }
    const long long o = (((long long)b * Ho + oy) * Wo + ox) * C + 4 * c4;
    *(f32x4*)(y + o) = m;
    if (RELUMASK) { ax = m.x > 0.f ? ax : 15; ay = m.y > 0.f ? ay : 15; az = m.z > 0.f ? az : 15; aw = m.w > 0.f ? aw : 15; }
    if (ARGMAX) *(uint32_t*)(amax + o) = (uint32_t)ax | ((uint32_t)ay << 8) | ((uint32_t)az << 16) | ((uint32_t)aw << 24);
  }
}

extern "C" int sqd_maxpool3x3s2_ceil_fwd(const float* x, float* y, unsigned char* argmax, int B, int H, int W,
                                         int C, void* stream) {
  SQD_CHECK_ARG(x && y && B > 0 && H >= 3 && W >= 3 && C > 0 && (C & 3) == 0);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)argmax & 3) == 0);
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
  SQD_CHECK_ARG((long long)B * Ho <= 65535);
  const dim3 grid((unsigned)sqd_cdiv(((Wo + 1) / 2) * (C >> 2), 256), (unsigned)(B * Ho));
  if (argmax) hipLaunchKernelGGL(maxpool_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, y, argmax, B, H, W, C, Ho, Wo);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, y, argmax, B, H, W, C, Ho, Wo);
  return sqd_launch_status();
}

// The same pool for a ReLU-output input in training: argmax (required) carries the input's ReLU mask (code 15 where the pooled
// value is not > 0), so sqd_maxpool3x3s2_ceil_bwd is called with relu_src = NULL and reads nothing but dy and the codes.
extern "C" int sqd_maxpool3x3s2_ceil_fwd_relu(const float* x, float* y, unsigned char* argmax, int B, int H, int W,
                                              int C, void* stream) {
  SQD_CHECK_ARG(x && y && argmax && B > 0 && H >= 3 && W >= 3 && C > 0 && (C & 3) == 0);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)argmax & 3) == 0);
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
  SQD_CHECK_ARG((long long)B * Ho <= 65535);
  const dim3 grid((unsigned)sqd_cdiv(((Wo + 1) / 2) * (C >> 2), 256), (unsigned)(B * Ho));
  hipLaunchKernelGGL((maxpool_fwd_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, x, y, argmax, B, H, W, C, Ho, Wo);
  return sqd_launch_status();
}

// Backward: dx[b,iy,ix,c] = sum over the (at most 4) windows covering (iy,ix) whose argmax is this
// element.  Gather form (one thread per input element, no atomics, deterministic).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                          float* __restrict__ dx, const float* __restrict__ relu_src,
                                                          int B, int H, int W, int C, int Ho, int Wo) {
  // One thread = a 2x2 quad of input pixels (rows 2r, 2r+1; columns 2q, 2q+1) x 4 channels.  The quad is covered by exactly the
  // four windows (r-1..r, q-1..q): 4 (code word, dy) loads feed 4 outputs, where a thread per pixel issued 9 for the same four
  // (the kernel is bound by those gather loads, not by its stores).  blockIdx.y = (b, r), blockIdx.x covers q * cv + c4:
  // 32-bit index arithmetic only.  Blocks are renumbered XCD-contiguously: vertically adjacent quads share two windows.
  const int cv = C >> 2, Wq = (W + 1) >> 1, Hq = (H + 1) >> 1;
  const int gxn = (int)gridDim.x, nblk = gxn * (int)gridDim.y;
  const int lin = sqd_xcd_contiguous((int)blockIdx.y * gxn + (int)blockIdx.x, nblk);
  const int by = lin / gxn, bx = lin - by * gxn;
  const int t = bx * 256 + (int)threadIdx.x;
  if (t >= Wq * cv) return;
  const int q = t / cv, c4 = t - q * cv;
  const int r = by % Hq, b = by / Hq;
  // window (oy, ox) contributes to pixel (iy, ix) at tap (iy - 2 oy) * 3 + (ix - 2 ox)
  f32x4 o00 = (f32x4){0.f, 0.f, 0.f, 0.f}, o01 = o00, o10 = o00, o11 = o00;
  auto add = [](f32x4& acc, uint32_t am, const f32x4& g, int tap) {
    if ((int)(am & 255) == tap) acc.x += g.x;
    if ((int)((am >> 8) & 255) == tap) acc.y += g.y;
    if ((int)((am >> 16) & 255) == tap) acc.z += g.z;
    if ((int)(am >> 24) == tap) acc.w += g.w;
  };
  // same order of additions per output as the one-pixel gather (oy ascending, then ox ascending): bitwise the same results
#pragma unroll
  for (int dr = -1; dr <= 0; ++dr)
#pragma unroll
    for (int dq = -1; dq <= 0; ++dq) {
      const int oy = r + dr, ox = q + dq;
      if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
      const long long o = (((long long)b * Ho + oy) * Wo + ox) * C + 4 * c4;
      const uint32_t am = *(const uint32_t*)(amax + o);
      const f32x4 g = *(const f32x4*)(dy + o);
      // taps of this window that fall on the quad: rows 2r - 2oy = -2dr (+0, +1), columns -2dq (+0, +1)
      const int ty = -2 * dr, tx = -2 * dq;
      add(o00, am, g, ty * 3 + tx);
      if (tx + 1 < 3) add(o01, am, g, ty * 3 + tx + 1);
      if (ty + 1 < 3) add(o10, am, g, (ty + 1) * 3 + tx);
      if (ty + 1 < 3 && tx + 1 < 3) add(o11, am, g, (ty + 1) * 3 + tx + 1);
    }
  const int iy = 2 * r, ix = 2 * q;
  auto put = [&](f32x4 acc, int y, int x) {
    if (y >= H || x >= W) return;
    const long long xo = (((long long)b * H + y) * W + x) * C + 4 * c4;
    if (relu_src) {      // the pooled tensor was a ReLU output: fold its backward mask into this store
      const f32x4 m = *(const f32x4*)(relu_src + xo);
      acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f; acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
    }
    *(f32x4*)(dx + xo) = acc;
  };
  put(o00, iy, ix); put(o01, iy, ix + 1); put(o10, iy + 1, ix); put(o11, iy + 1, ix + 1);
}

extern "C" int sqd_maxpool3x3s2_ceil_bwd(const float* dy, const unsigned char* argmax, float* dx, const float* relu_src,
                                         int B, int H, int W, int C, void* stream) {
  SQD_CHECK_ARG(dy && argmax && dx && B > 0 && H >= 3 && W >= 3 && C > 0 && (C & 3) == 0);
  SQD_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)argmax & 3) == 0 && ((uintptr_t)relu_src & 15) == 0);
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
  SQD_CHECK_ARG((long long)B * ((H + 1) / 2) <= 65535 && (long long)W * (C >> 2) < (1ll << 30));
  const dim3 grid((unsigned)sqd_cdiv(((W + 1) / 2) * (C >> 2), 256), (unsigned)(B * ((H + 1) / 2)));
  hipLaunchKernelGGL(maxpool_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, relu_src, B, H, W, C, Ho, Wo);
  return sqd_launch_status();
}
