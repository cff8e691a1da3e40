// Implicit-GEMM convolution (1x1 and 3x3/pad 1, stride 1) on the gfx950 fp32 matrix cores.
//
// Serves, for SqueezeDet (reference: src/model/squeezedet.py:12-14,18-22,73-75,83):
//   * Fire squeeze 1x1 + ReLU, Fire expand1x1 + ReLU, Fire expand3x3 + ReLU (each expand writes its
//     own channel range of one NHWC buffer, so torch.cat at :19-22 never materialises),
//   * ConvDet 3x3 (no activation; NHWC output makes the permute at :85 free),
//   * the data-gradient of all of the above (a dgrad is the same convolution with transposed /
//     tap-flipped weights; the ReLU mask of the forward output is applied while staging dY).
//
// Layout: activations NHWC fp32 ([B][H][W][pitch], a layer reads/writes a channel window
// [coff, coff+C) of a buffer whose pixel pitch may be larger: that is how concat is eliminated).
// Weights are pre-packed by the host as [C/KC][TAPS][Npad][KC] (zero padded) so that one K-chunk
// of one output-channel slice is a contiguous run.
//
// Work decomposition: one 256-thread workgroup (4 waves, one per SIMD) computes a tile of
// TH x 16 pixels (3x3: a TH-row x 16-column patch of one image with a 1-pixel halo; 1x1: TH*16
// consecutive pixels of the flattened B*H*W axis) times BN = 16*NT output channels.  The K loop
// walks channel chunks of KC; per chunk the activation tile and the weight slice for all taps are
// staged in LDS, then every wave issues v_mfma_f32_16x16x4_f32 over (tap, k).  MFMA operand A is
// the weight tile (row = output channel), operand B the activation tile (column = pixel), so each
// lane ends up holding 4 consecutive output channels of one pixel: the epilogue is one 16-byte
// store per lane and tile.  LDS rows are padded to KC+4 floats (4*odd) which makes the
// ds_read_b64 operand fetches bank-conflict free (16 rows x 2 k-pairs per 32-lane half).
//
// Numerics: fp32 operands, fp32 accumulate; the MFMA is bit-for-bit a k-ordered fmaf chain, so
// results differ from the reference's MKL-DNN/cuDNN summation order only by fp32 rounding
// (parity tolerance 1e-4, tests/test_conv_gpu.py).
#include "sqd_common.h"

struct ConvArgs {
  const float* x; const float* w; const float* bias; float* y; const float* xmask;
  int B, H, W;
  int C, x_pitch, x_coff;
  int N, Npad, y_pitch, y_coff;
  int relu, accumulate;
  int tiles_x, tiles_y;
  int xmask_pitch, xmask_coff;
  const float* ymask; const float* ymul;      // epilogue: zero where ymask <= 0 (ReLU backward), multiply by ymul (dropout)
  int ymask_pitch, ymask_coff, ymul_pitch, ymul_coff;
  long long total_px;
};

template <int TAPS, int KC, int MT, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int WM = 4;                 // waves along pixels; every wave covers all BN channels
  constexpr int TH = MT * WM;           // tile rows of 16 pixels
  constexpr int BN = 16 * NT;
  constexpr int KP = KC + 4;            // LDS row pitch in floats = 4 * odd
  constexpr int NPIX = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  constexpr int KV = KC / 4;
  static_assert(((KP / 4) & 1) == 1, "LDS pitch must be 4*odd floats");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* actT = smem;                   // [NPIX][KP]
  float* wT = smem + NPIX * KP;         // [TAPS][BN][KP]

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * BN;

  int b = 0, y0 = 0, x0 = 0;
  long long p0 = 0;
  if (TAPS == 9) {
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; b = t / a.tiles_y;
    y0 = ty * TH; x0 = tx * 16;
  } else {
    p0 = (long long)blockIdx.x * (TH * 16);
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (a.C + KC - 1) / KC;
  for (int cc = 0; cc < nchunks; ++cc) {
    const int c0 = cc * KC;
    if (cc) __syncthreads();
    // ---- stage the activation tile (zero outside the image / beyond C) ----
    for (int idx = tid; idx < NPIX * KV; idx += 256) {
      const int pix = idx / KV, v = idx - pix * KV;
      const int c = c0 + 4 * v;
      bool ok = c < a.C;
      long long gp;
      if (TAPS == 9) {
        const int r = pix / 18, col = pix - r * 18;
        const int iy = y0 + r - 1, ix = x0 + col - 1;
        ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        gp = ((long long)b * a.H + iy) * a.W + ix;
      } else {
        gp = p0 + pix;
        ok = ok && gp < a.total_px;
      }
      f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ok) {
        val = *(const f32x4*)(a.x + gp * a.x_pitch + a.x_coff + c);
        if (a.xmask) {
          const f32x4 m = *(const f32x4*)(a.xmask + gp * a.xmask_pitch + a.xmask_coff + c);
          val.x = m.x > 0.f ? val.x : 0.f; val.y = m.y > 0.f ? val.y : 0.f;
          val.z = m.z > 0.f ? val.z : 0.f; val.w = m.w > 0.f ? val.w : 0.f;
        }
      }
      *(f32x4*)(actT + pix * KP + 4 * v) = val;
    }
    // ---- stage the weight slice [TAPS][BN][KC] of this chunk ----
    const float* wc = a.w + ((long long)cc * TAPS * a.Npad + n0) * KC;
    for (int idx = tid; idx < TAPS * BN * KV; idx += 256) {
      const int tn = idx / KV, v = idx - tn * KV;
      const int tap = tn / BN, n = tn - tap * BN;
      *(f32x4*)(wT + tn * KP + 4 * v) = *(const f32x4*)(wc + ((long long)tap * a.Npad + n) * KC + 4 * v);
    }
    __syncthreads();
    // ---- MFMA over (tap, k) ----
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
      for (int k8 = 0; k8 < KC / 8; ++k8) {
        f32x2 bf[MT], af[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = (TAPS == 9) ? ((wm * MT + i) + dy) * 18 + lr + dx : (wm * MT + i) * 16 + lr;
          bf[i] = *(const f32x2*)(actT + row * KP + k8 * 8 + 2 * g);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
          af[j] = *(const f32x2*)(wT + (tap * BN + j * 16 + lr) * KP + k8 * 8 + 2 * g);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j].x, bf[i].x, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j].y, bf[i].y, acc[i][j]);
      }
    }
  }

  // ---- epilogue: lane holds channels n0 + 16j + 4g .. +3 of pixel (row i, column lr) ----
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    long long gp; bool valid;
    if (TAPS == 9) {
      const int iy = y0 + wm * MT + i, ix = x0 + lr;
      valid = iy < a.H && ix < a.W;
      gp = ((long long)b * a.H + iy) * a.W + ix;
    } else {
      gp = p0 + (wm * MT + i) * 16 + lr;
      valid = gp < a.total_px;
    }
    if (!valid) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 16 + 4 * g;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if (a.bias) v += *(const f32x4*)(a.bias + n);
      float* dst = a.y + gp * a.y_pitch + a.y_coff + n;
      if (a.accumulate) v += *(const f32x4*)dst;
      if (a.ymul) v *= *(const f32x4*)(a.ymul + gp * a.ymul_pitch + a.ymul_coff + n);
      if (a.ymask) {
        const f32x4 m = *(const f32x4*)(a.ymask + gp * a.ymask_pitch + a.ymask_coff + n);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
      }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *(f32x4*)dst = v;
    }
  }
}

template <int TAPS, int KC, int MT, int NT>
static int launch_conv(ConvArgs a, hipStream_t stream) {
  constexpr int TH = MT * 4, BN = 16 * NT, KP = KC + 4;
  constexpr int NPIX = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  constexpr size_t lds = (size_t)(NPIX + TAPS * BN) * KP * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  dim3 grid;
  if (TAPS == 9) {
    a.tiles_x = sqd_cdiv(a.W, 16); a.tiles_y = sqd_cdiv(a.H, TH);
    grid.x = (unsigned)(a.B * a.tiles_x * a.tiles_y);
  } else {
    a.tiles_x = a.tiles_y = 0;
    grid.x = (unsigned)((a.total_px + TH * 16 - 1) / (TH * 16));
  }
  grid.y = (unsigned)sqd_cdiv(a.N, BN);
  if ((int)grid.y * BN > a.Npad) return SQD_ERR_BAD_ARG;   // packed weights too short for this slice width
  auto kern = conv_igemm_kernel<TAPS, KC, MT, NT>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SQD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  return sqd_launch_status();
}

// Tile configurations (cfg_id) -> template instance.  The host picks per layer; any config is
// correct for any shape (edges are masked, partial K chunks zero-filled).
struct ConvCfg { int taps, kc, mt, nt; };
static const ConvCfg kConvCfgs[] = {
    {1, 16, 2, 4},  // 0  1x1, small C (expand1x1), 128 px x 64 ch
    {1, 32, 2, 1},  // 1  1x1 squeeze N<=16
    {1, 32, 2, 2},  // 2  N<=32
    {1, 32, 2, 3},  // 3  N<=48
    {1, 32, 2, 4},  // 4  N<=64
    {1, 32, 2, 6},  // 5  N<=96
    {1, 32, 1, 3},  // 6  64-px tiles (late, small layers)
    {1, 32, 1, 4},  // 7
    {1, 32, 1, 6},  // 8
    {1, 16, 1, 4},  // 9  1x1 small C, 64 px
    {9, 16, 2, 4},  // 10 3x3, 8x16 px x 64 ch
    {9, 16, 2, 5},  // 11 3x3, 8x16 px x 80 ch (ConvDet N=72)
    {9, 16, 1, 4},  // 12 3x3, 4x16 px x 64 ch
    {9, 16, 1, 5},  // 13 3x3, 4x16 px x 80 ch
    {9, 16, 2, 2},  // 14 3x3, 8x16 px x 32 ch
    {9, 16, 2, 6},  // 15 3x3, 8x16 px x 96 ch
    {9, 16, 1, 6},  // 16 3x3, 4x16 px x 96 ch
    {9, 16, 2, 3},  // 17 3x3, 8x16 px x 48 ch
    {9, 16, 2, 1},  // 18 3x3, 8x16 px x 16 ch (dgrad into a 16-channel squeeze)
};
static const int kNumConvCfgs = (int)(sizeof(kConvCfgs) / sizeof(kConvCfgs[0]));

extern "C" int sqd_conv_num_cfgs() { return kNumConvCfgs; }

extern "C" int sqd_conv_cfg_info(int cfg_id, int* taps, int* kc, int* tile_px, int* bn) {
  SQD_CHECK_ARG(cfg_id >= 0 && cfg_id < kNumConvCfgs);
  const ConvCfg& c = kConvCfgs[cfg_id];
  if (taps) *taps = c.taps;
  if (kc) *kc = c.kc;
  if (tile_px) *tile_px = c.mt * 4 * 16;
  if (bn) *bn = 16 * c.nt;
  return SQD_OK;
}

extern "C" int sqd_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                            const float* xmask, const float* ymask, const float* ymul, int B, int H, int W, int C,
                            int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu,
                            int accumulate, int xmask_pitch, int xmask_coff, int ymask_pitch, int ymask_coff,
                            int ymul_pitch, int ymul_coff, int cfg_id, void* stream) {
  SQD_CHECK_ARG(x && w_packed && y);
  SQD_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && N > 0);
  SQD_CHECK_ARG(cfg_id >= 0 && cfg_id < kNumConvCfgs);
  SQD_CHECK_ARG((C & 3) == 0 && (N & 3) == 0);
  SQD_CHECK_ARG((x_pitch & 3) == 0 && (x_coff & 3) == 0 && (y_pitch & 3) == 0 && (y_coff & 3) == 0);
  SQD_CHECK_ARG(x_coff + C <= x_pitch && y_coff + N <= y_pitch);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)w_packed & 15) == 0);
  SQD_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0);
  if (xmask) SQD_CHECK_ARG((xmask_pitch & 3) == 0 && (xmask_coff & 3) == 0 && xmask_coff + C <= xmask_pitch && ((uintptr_t)xmask & 15) == 0);
  if (ymask) SQD_CHECK_ARG((ymask_pitch & 3) == 0 && (ymask_coff & 3) == 0 && ymask_coff + N <= ymask_pitch && ((uintptr_t)ymask & 15) == 0);
  if (ymul) SQD_CHECK_ARG((ymul_pitch & 3) == 0 && (ymul_coff & 3) == 0 && ymul_coff + N <= ymul_pitch && ((uintptr_t)ymul & 15) == 0);
  ConvArgs a;
  a.ymask = ymask; a.ymul = ymul; a.ymask_pitch = ymask_pitch; a.ymask_coff = ymask_coff; a.ymul_pitch = ymul_pitch; a.ymul_coff = ymul_coff;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.xmask = xmask;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff;
  a.relu = relu; a.accumulate = accumulate; a.tiles_x = a.tiles_y = 0;
  a.xmask_pitch = xmask_pitch; a.xmask_coff = xmask_coff;
  a.total_px = (long long)B * H * W;
  hipStream_t s = (hipStream_t)stream;
  const ConvCfg& c = kConvCfgs[cfg_id];
#define SQD_CONV_CASE(T, K, M, Nn) \
  if (c.taps == T && c.kc == K && c.mt == M && c.nt == Nn) return launch_conv<T, K, M, Nn>(a, s);
  SQD_CONV_CASE(1, 16, 2, 4)
  SQD_CONV_CASE(1, 32, 2, 1)
  SQD_CONV_CASE(1, 32, 2, 2)
  SQD_CONV_CASE(1, 32, 2, 3)
  SQD_CONV_CASE(1, 32, 2, 4)
  SQD_CONV_CASE(1, 32, 2, 6)
  SQD_CONV_CASE(1, 32, 1, 3)
  SQD_CONV_CASE(1, 32, 1, 4)
  SQD_CONV_CASE(1, 32, 1, 6)
  SQD_CONV_CASE(1, 16, 1, 4)
  SQD_CONV_CASE(9, 16, 2, 4)
  SQD_CONV_CASE(9, 16, 2, 5)
  SQD_CONV_CASE(9, 16, 1, 4)
  SQD_CONV_CASE(9, 16, 1, 5)
  SQD_CONV_CASE(9, 16, 2, 2)
  SQD_CONV_CASE(9, 16, 2, 6)
  SQD_CONV_CASE(9, 16, 1, 6)
  SQD_CONV_CASE(9, 16, 2, 3)
  SQD_CONV_CASE(9, 16, 2, 1)
#undef SQD_CONV_CASE
  return SQD_ERR_UNSUPPORTED;
}
