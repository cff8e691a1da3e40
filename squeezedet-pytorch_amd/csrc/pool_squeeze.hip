// Fused MaxPool2d(3, 2, ceil_mode) + Fire squeeze 1x1 + ReLU (inference forward).
//
// Reference: nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True) between Fire groups (src/model/squeezedet.py:39,42)
// followed by the next Fire's squeeze conv + ReLU (:12,18).  Unfused, the pooled tensor (76 MB / 38 MB at bs=20) is
// written by the pool kernel and read again by the squeeze kernel, and both launches are memory-bound.  Here a workgroup
// pools a tile of 64 output pixels straight into the LDS operand image of the 1x1 convolution (k-quad-major, 128
// channels at a time), multiplies it with the LDS-resident squeeze weights on the matrix cores and writes only the
// small squeeze output.  The pooled tensor never exists; the unpooled input is read once.
// Pooling = plain max over the (clipped) window, NaN-agnostic like the inference stem; the training path keeps the
// separate kernels (it needs the pooled activations and the argmax).
#include "sqd_common.h"

struct PoolSqArgs {
  const float* x; const float* w; const float* bias; float* y;
  int B, H, W, C, x_pitch, x_coff;       // unpooled input window
  int Ho, Wo;
  int N, Npad, y_pitch, y_coff;
  long long total_px;                    // B * Ho * Wo
  int ntiles;
};

// w: packed [C/4 (+ padding planes)][Npad][4] (what sqd_pack_conv_weight emits for a 1x1 configuration with Npad a
// multiple of 16 and C a multiple of its KC).
template <int NT>
__global__ __launch_bounds__(256) void pool_squeeze_kernel(PoolSqArgs a) {
  constexpr int TP = 64;                       // pooled pixels per tile (16 per wave)
  constexpr int KCH = 128;                     // channels pooled per K chunk (LDS: 32 KB)
  constexpr int QCH = KCH / 4;                 // 16-byte quads per pixel and chunk
  constexpr int A_IT = TP * QCH / 256;         // (pixel, quad) slots per thread = 8
  constexpr int TPP = TP + 1;                  // padded pixel pitch of the LDS image: lanes that differ in the quad index
                                               // (stride TPP * 16 B = 1040 B) land in different banks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const actT = smem;                    // [QCH][TPP][4]
  float* const wT = smem + QCH * TPP * 4;      // [C/4][Npad][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int Npad = 16 * NT;
  const int cq_total = (a.C + 3) >> 2;
  for (int i = tid; i < cq_total * Npad; i += 256) ((f32x4*)wT)[i] = ((const f32x4*)a.w)[i];
  f32x4 biasr[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 16 + 4 * g;
    biasr[j] = (a.bias && n < a.N) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int nchunks = (a.C + KCH - 1) / KCH;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this thread's pooled pixels: slot = it * 256 + tid -> pixel = slot / QCH, quad = slot % QCH (quad fastest: 32
    // consecutive lanes read 512 contiguous bytes of one input pixel)
    // Clipped border windows (ceil mode) and pixels beyond the tensor are handled by CLAMPING the tap offsets to the last
    // valid row / column (a duplicated tap does not change a maximum), so all nine loads of a slot are unconditional and
    // the compiler can keep 18 of them in flight per thread.
    const float* src[A_IT]; int ro1[A_IT], ro2[A_IT], co1[A_IT], co2[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int slot = it * 256 + tid;
      const int pl = slot / QCH, q = slot - pl * QCH;
      long long gp = (long long)tile * TP + pl;
      if (gp >= a.total_px) gp = a.total_px - 1;                     // its result lands in an LDS row nobody stores
      const int ox = (int)(gp % a.Wo); const long long t = gp / a.Wo;
      const int oy = (int)(t % a.Ho); const int b = (int)(t / a.Ho);
      src[it] = a.x + (((long long)b * a.H + 2 * oy) * a.W + 2 * ox) * a.x_pitch + a.x_coff + 4 * q;
      const int rs = a.W * a.x_pitch, cs = a.x_pitch;
      ro1[it] = (2 * oy + 1 < a.H) ? rs : 0; ro2[it] = (2 * oy + 2 < a.H) ? 2 * rs : ro1[it];
      co1[it] = (2 * ox + 1 < a.W) ? cs : 0; co2[it] = (2 * ox + 2 < a.W) ? 2 * cs : co1[it];
    }
    for (int cc = 0; cc < nchunks; ++cc) {
      if (cc > 0 || tile != (int)blockIdx.x) __syncthreads();             // previous MFMA phase left actT
#pragma unroll
      for (int it0 = 0; it0 < A_IT; it0 += 2) {
        f32x4 v[2][9];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = it0 + u;
          const int slot = it * 256 + tid;
          const int q = slot % QCH;
          // channels beyond C in the last chunk: read the quad at channel 0 instead (finite, multiplied by zero weights)
          const float* p = src[it] + ((cc * KCH + 4 * q < a.C) ? cc * KCH : -4 * q);
          const int ro[3] = {0, ro1[it], ro2[it]}, co[3] = {0, co1[it], co2[it]};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[u][dy * 3 + dx] = *(const f32x4*)(p + ro[dy] + co[dx]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int slot = (it0 + u) * 256 + tid;
          const int pl = slot / QCH, q = slot - pl * QCH;
          f32x4 m;
          m.x = fmaxf(fmaxf(fmaxf(fmaxf(v[u][0].x, v[u][1].x), v[u][2].x), fmaxf(fmaxf(v[u][3].x, v[u][4].x), v[u][5].x)), fmaxf(fmaxf(v[u][6].x, v[u][7].x), v[u][8].x));
          m.y = fmaxf(fmaxf(fmaxf(fmaxf(v[u][0].y, v[u][1].y), v[u][2].y), fmaxf(fmaxf(v[u][3].y, v[u][4].y), v[u][5].y)), fmaxf(fmaxf(v[u][6].y, v[u][7].y), v[u][8].y));
          m.z = fmaxf(fmaxf(fmaxf(fmaxf(v[u][0].z, v[u][1].z), v[u][2].z), fmaxf(fmaxf(v[u][3].z, v[u][4].z), v[u][5].z)), fmaxf(fmaxf(v[u][6].z, v[u][7].z), v[u][8].z));
          m.w = fmaxf(fmaxf(fmaxf(fmaxf(v[u][0].w, v[u][1].w), v[u][2].w), fmaxf(fmaxf(v[u][3].w, v[u][4].w), v[u][5].w)), fmaxf(fmaxf(v[u][6].w, v[u][7].w), v[u][8].w));
          *(f32x4*)(actT + (q * TPP + pl) * 4) = m;
        }
      }
      __syncthreads();
      // ---- 1x1 conv of the pooled chunk: wave w owns pixels 16w..16w+15, all NT channel tiles ----
      const int qbase = cc * QCH;
#pragma unroll
      for (int s = 0; s < QCH / 4; ++s) {
        const int kq = 4 * s + g;
        if (4 * (qbase + 4 * s) >= a.C) break;                             // uniform: chunk tail beyond C
        const f32x4 bf = *(const f32x4*)(actT + (kq * TPP + wave * 16 + lr) * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 af = (qbase + kq < cq_total) ? *(const f32x4*)(wT + ((qbase + kq) * Npad + j * 16 + lr) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[j] = mfma16(af[t], bf[t], acc[j]);
        }
      }
    }
    // ---- bias + ReLU, lane holds channels 16j+4g..+3 of pixel (tile, 16*wave + lr) ----
    const long long gp = (long long)tile * TP + wave * 16 + lr;
    if (gp < a.total_px) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + 4 * g;
        if (n >= a.N) continue;
        f32x4 v = acc[j] + biasr[j];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *(f32x4*)(a.y + gp * a.y_pitch + a.y_coff + n) = v;
      }
    }
  }
}

template <int NT>
static int launch_pool_squeeze(PoolSqArgs a, hipStream_t s) {
  const size_t lds = (size_t)(32 * 65 * 4 + ((a.C + 3) / 4) * 16 * NT * 4) * sizeof(float);
  if (lds > 160 * 1024) return SQD_ERR_UNSUPPORTED;
  auto kern = pool_squeeze_kernel<NT>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SQD_ERR_LAUNCH;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
  if (nb > 4) nb = 4;
  int dev = 0, cus = 256; hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const int slots = cus * nb;
  const int per_wg = sqd_cdiv(a.ntiles, slots);
  const int gx = sqd_cdiv(a.ntiles, per_wg);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, s, a);
  return sqd_launch_status();
}

// y[..., y_coff : y_coff+N] = ReLU(conv1x1(MaxPool(3,2,ceil)(x[..., x_coff : x_coff+C])) + bias).  x: NHWC [B][H][W][x_pitch];
// y: NHWC [B][Ho][Wo][y_pitch] with Ho = ceil((H-3)/2)+1, Wo likewise; w_packed: sqd_pack_conv_weight output for a 1x1
// configuration whose KC divides C, Npad = N rounded up to 16 (<= 96).
extern "C" int sqd_pool_squeeze_fwd(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W,
                                    int C, int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, void* stream) {
  SQD_CHECK_ARG(x && w_packed && y && B > 0 && H >= 3 && W >= 3 && C > 0 && N > 0);
  SQD_CHECK_ARG((C & 3) == 0 && (N & 3) == 0 && (Npad & 15) == 0 && Npad >= N && Npad <= 96);
  SQD_CHECK_ARG((x_pitch & 3) == 0 && (x_coff & 3) == 0 && (y_pitch & 3) == 0 && (y_coff & 3) == 0);
  SQD_CHECK_ARG(x_coff + C <= x_pitch && y_coff + N <= y_pitch);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0));
  PoolSqArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.Ho = (H - 3 + 1) / 2 + 1; a.Wo = (W - 3 + 1) / 2 + 1;
  a.N = N; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff;
  a.total_px = (long long)B * a.Ho * a.Wo;
  a.ntiles = (int)((a.total_px + 63) / 64);
  hipStream_t s = (hipStream_t)stream;
  switch (Npad / 16) {
    case 1: return launch_pool_squeeze<1>(a, s);
    case 2: return launch_pool_squeeze<2>(a, s);
    case 3: return launch_pool_squeeze<3>(a, s);
    case 4: return launch_pool_squeeze<4>(a, s);
    case 5: return launch_pool_squeeze<5>(a, s);
    case 6: return launch_pool_squeeze<6>(a, s);
  }
  return SQD_ERR_UNSUPPORTED;
}
