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
// Weights are pre-packed (pack_weight_kernel) as [C/KC][KC/4][TAPS][Npad][4] (zero padded): the LDS
// image of one K-chunk of one output-channel slice is a set of contiguous runs.
//
// Work decomposition: one 256-thread workgroup (4 waves, one per SIMD) computes a tile of
// TH x 16 pixels (3x3: a TH-row x 16-column patch of one image with a 1-pixel halo; 1x1: TH*16
// consecutive pixels of the flattened B*H*W axis) times BN = 16*NT output channels.  The K loop
// walks channel chunks of KC; per chunk the activation tile and the weight slice for all taps are
// staged in LDS, then every wave issues v_mfma_f32_16x16x4_f32 over (tap, k).  Workgroups are
// PERSISTENT: the grid is one resident wave of workgroups (CUs x occupancy), each walks a strided
// list of pixel tiles, and the global loads of the next stage (next K chunk or next tile) are issued
// into registers before the MFMAs of the current stage and written to LDS after them, so HBM/L2
// latency hides under compute; when the whole K fits one chunk the weight slice stays in LDS.  MFMA operand A is
// the weight tile (row = output channel), operand B the activation tile (column = pixel), so each
// lane ends up holding 4 consecutive output channels of one pixel: the epilogue is one 16-byte
// store per lane and tile.
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
  int tiles_x, tiles_y, ntiles;
  int nslices, gx;                           // persistent grid: gx tile streams x nslices channel slices (1-D launch)
  int wg_cap;                                // host only: cap on resident workgroups per CU (0 = occupancy)
  int fuse_e;                                // fused Fire expand: E = channels of EACH half (expand1x1 | expand3x3), else 0
  int xmask_pitch, xmask_coff;
  const float* ymask; const float* ymul;      // epilogue: zero where ymask <= 0 (ReLU backward), multiply by ymul (dropout)
  int ymask_pitch, ymask_coff, ymul_pitch, ymul_coff;
  // epilogue (conv_ws family, sqd_conv_drop_fwd): counter-based dropout of the output, sqd_common.h; drop_state = {seed, step} or null
  const unsigned long long* drop_state; int drop_keep; float drop_scale;
  long long total_px;
};

// LDS tiles are "k-quad major": [KC/4][rows][4 floats].  A lane's MFMA operands for 4 consecutive
// k (one ds_read_b128) sit in plane 4*s + (lane>>4); with the row count a multiple of 16 the four
// 16-lane groups of a ds_read_b128 each cover 16 distinct 16-byte slots of the 256-byte bank row
// (MI355X_MICROARCH.md LDS table), i.e. conflict-free at the full 256 B/clk.  Staging stores walk rows
// fastest (8 consecutive lanes = 8 consecutive 16-byte slots of one plane): conflict-free too.
template <int TAPS, int KC, int MT, int NT, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_igemm_kernel(ConvArgs a) {
  constexpr int WM = 4;                 // waves along pixels; every wave covers all BN channels
  constexpr int TH = MT * WM;           // tile rows of 16 pixels
  constexpr int BN = 16 * NT;
  constexpr int NPIX = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  constexpr int NPIXP = (NPIX + 15) & ~15;
  constexpr int WROWS = TAPS * BN;      // multiple of 16
  constexpr int KV = KC / 4;            // planes
  constexpr int A_IT = (NPIX * KV + 255) / 256;          // float4 prefetch registers per thread
  constexpr int W_IT = (WROWS * KV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* actT = smem;                   // [KV][NPIXP][4]
  float* wT = smem + KV * NPIXP * 4;    // [KV][WROWS][4]

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  // XCD-aware mapping (workgroups b and b+8 share an XCD and its L2): the channel slices of one tile stream get
  // consecutive ids on the SAME XCD, so they run together and re-read their activation tiles from that L2
  const int wgq = (int)blockIdx.x >> 3;
  const int n0 = (wgq % a.nslices) * BN;
  const int tstride = a.gx;
  const int nchunks = (a.C + KC - 1) / KC;
  const bool w_stationary = (nchunks == 1);   // whole K fits one chunk: weights stay in LDS across tiles
  const int ntiles = a.ntiles;
  int tile = (wgq / a.nslices) * 8 + ((int)blockIdx.x & 7);
  if (tile >= ntiles) return;

  f32x4 ra[A_IT], rw[W_IT];

  // ---- global -> registers (zero outside the image / beyond C) ----
  auto load_act = [&](int t, int cc) {
    int b = 0, y0 = 0, x0 = 0;
    long long p0 = 0;
    if (TAPS == 9) {
      const int tx = t % a.tiles_x; t /= a.tiles_x;
      const int ty = t % a.tiles_y; b = t / a.tiles_y;
      y0 = ty * TH; x0 = tx * 16;
    } else {
      p0 = (long long)t * (TH * 16);
    }
    const int c0 = cc * KC;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int idx = tid + it * 256;
      const int v = idx / NPIX, pix = idx - v * NPIX;
      const int c = c0 + 4 * v;
      bool ok = (idx < NPIX * KV) && c < a.C;
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
      ra[it] = val;
    }
  };
  // packed weights: [chunk][plane v][tap][Npad][4]
  auto load_w = [&](int cc) {
    const float* wc = a.w + (long long)cc * KV * TAPS * a.Npad * 4;
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * 256;
      const int v = idx / WROWS, tn = idx - v * WROWS;
      const int tap = tn / BN, n = tn - tap * BN;
      f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (idx < WROWS * KV) val = *(const f32x4*)(wc + (((long long)v * TAPS + tap) * a.Npad + n0 + n) * 4);
      rw[it] = val;
    }
  };
  // ---- registers -> LDS ----
  auto store_act = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int idx = tid + it * 256;
      const int v = idx / NPIX, pix = idx - v * NPIX;
      if (idx < NPIX * KV) *(f32x4*)(actT + (v * NPIXP + pix) * 4) = ra[it];
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < WROWS * KV) *(f32x4*)(wT + idx * 4) = rw[it];      // idx = v*WROWS + tn
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: first stage
  load_act(tile, 0);
  load_w(0);
  store_act();
  store_w();
  __syncthreads();

  for (;;) {
    for (int cc = 0; cc < nchunks; ++cc) {
      // ---- prefetch the next stage (next K chunk, or chunk 0 of this workgroup's next tile) ----
      int ncc = cc + 1, ntile = tile;
      if (ncc == nchunks) { ncc = 0; ntile = tile + tstride; }
      const bool has_next = ntile < ntiles;
      const bool next_w = has_next && !w_stationary;
      if (has_next) load_act(ntile, ncc);
      if (next_w) load_w(ncc);

      // ---- MFMA over (tap, k) of the staged chunk ----
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
        for (int s = 0; s < KC / 16; ++s) {
          f32x4 bf[MT], af[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int row = (TAPS == 9) ? ((wm * MT + i) + dy) * 18 + lr + dx : (wm * MT + i) * 16 + lr;
            bf[i] = *(const f32x4*)(actT + ((4 * s + g) * NPIXP + row) * 4);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j)
            af[j] = *(const f32x4*)(wT + ((4 * s + g) * WROWS + tap * BN + j * 16 + lr) * 4);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j][t], bf[i][t], acc[i][j]);
        }
      }

      // ---- epilogue after the last chunk: lane holds channels n0+16j+4g..+3 of pixel (row i, col lr) ----
      if (cc == nchunks - 1) {
        int b = 0, y0 = 0, x0 = 0;
        long long p0 = 0;
        if (TAPS == 9) {
          int t = tile;
          const int tx = t % a.tiles_x; t /= a.tiles_x;
          const int ty = t % a.tiles_y; b = t / a.tiles_y;
          y0 = ty * TH; x0 = tx * 16;
        } else {
          p0 = (long long)tile * (TH * 16);
        }
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
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 16 + 4 * g;
            f32x4 v = acc[i][j];
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!valid || n >= a.N) continue;
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

      // ---- hand the prefetched stage over through LDS ----
      __syncthreads();                       // every wave is done reading the current stage
      if (has_next) store_act();
      if (next_w) store_w();
      __syncthreads();
    }
    tile += tstride;
    if (tile >= ntiles) break;
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: the same tiling and MFMA loop, but stages are moved global -> LDS by
// global_load_lds_dwordx4 (no staging registers) into a DOUBLE-buffered LDS image: the loads of stage
// s+1 are in flight during the whole MFMA phase of stage s and there is ONE barrier per stage (its
// vmcnt(0) retires the DMA, the barrier publishes it and frees the other buffer).  The DMA writes
// 64 consecutive 16-byte slots per wave instruction (wave-uniform base + lane*16), which is exactly the
// k-quad-major image when the slot index space is [plane][padded row]; out-of-image halo pixels and
// padded rows are zero-filled by the buffer range check (below) so the image needs no conditional writes.
// ---------------------------------------------------------------------------------------------

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ReLU as ONE v_max_f32 per element.  fmaxf(v, 0) compiles to two (a canonicalising max in front of the real one);
// on gfx950 every VALU instruction delays the fp32 MFMA stream of the SIMD (DESIGN.md cost model), so it matters.
__device__ __forceinline__ f32x4 sqd_relu4(f32x4 v, float lo) {      // lo (wave-uniform): 0 = ReLU, -inf = identity
  asm volatile("v_max_f32 %0, %4, %0\n\tv_max_f32 %1, %4, %1\n\tv_max_f32 %2, %4, %2\n\tv_max_f32 %3, %4, %3"
               : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "s"(lo));
  return v;
}

// FUSE = true is the fused Fire expand (reference: Fire.forward, src/model/squeezedet.py:18-22 -- expand1x1 and expand3x3
// both read the squeeze output and are concatenated): the packed weights hold 2E output channels in 16-channel groups
// that ALTERNATE between the two convolutions (group 2i = expand1x1 channels 16i.., as a 3x3 whose only non-zero tap
// is the centre; group 2i+1 = expand3x3 channels 16i..).  With NT even every workgroup slice carries the same mix, the
// 1x1 groups simply skip the MFMAs (and operand reads) of the 8 outer taps, and the epilogue writes group 2i to
// channel window [0, E) and group 2i+1 to [E, 2E) of the output -- the concat.  One launch, one staging of the
// squeeze tile, and the 1x1 outputs' stores drain under the 3x3's matrix work.
template <int TAPS, int KC, int MT, int NT, int WM, int MINW, bool FUSE, bool WSTAT>
__global__ __launch_bounds__(WM * 64, MINW) void conv_dma_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)    // the host pass only needs the launch stub (the buffer-resource builtins are device-only)
  // WSTAT: the whole K fits one chunk (C <= KC), the weight slice is loaded once and stays in LDS (launch-time property
  // made a template flag so that the tap loop below contains no branch at all)
  static_assert(!FUSE || (TAPS == 9 && (NT % 2) == 0), "fused expand: 3x3 tiles with an even number of channel groups");
  constexpr int NTHR = WM * 64;              // 4 or 8 waves; with 8, two waves per SIMD share one staged tile
  constexpr int TH = MT * WM;
  constexpr int BN = 16 * NT;
  constexpr int NPIX = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  constexpr int NPIXP = (NPIX + 15) & ~15;
  constexpr int WROWS = TAPS * BN;
  constexpr int KV = KC / 4;
  // 16-byte slots, rounded to whole workgroup passes (4 waves x 64 lanes) so the DMA issue is branch-free
  constexpr int ASLOTS = (KV * NPIXP + NTHR - 1) / NTHR * NTHR;
  constexpr int WSLOTS = (KV * WROWS + NTHR - 1) / NTHR * NTHR;
  constexpr int A_IT = ASLOTS / NTHR;
  constexpr int W_IT = WSLOTS / NTHR;
  constexpr int STEPS = TAPS * (KC / 16);               // MFMA groups per stage; DMA issue is spread over them
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: act[0], act[1], w[0], (w[1] unless the weights are stationary)
  float* const actB = smem;
  float* const wB = smem + 2 * ASLOTS * 4;

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  // XCD-aware mapping: see conv_igemm_kernel
  const int wgq = (int)blockIdx.x >> 3;
  const int n0 = (wgq % a.nslices) * BN;
  const int tstride = a.gx;
  const int nchunks = (a.C + KC - 1) / KC;
  constexpr bool w_stationary = WSTAT;
  const int ntiles = a.ntiles;
  int tile = (wgq / a.nslices) * 8 + ((int)blockIdx.x & 7);
  if (tile >= ntiles) return;

  const int wm_s = __builtin_amdgcn_readfirstlane(wm);  // wave index as a scalar: the LDS-DMA destinations stay in SGPRs

  // Both DMA streams go through buffer resources: address = resource base + wave-uniform SGPR byte offset (tile origin
  // + K chunk) + per-lane 32-bit byte offset, so issuing a stage's DMA costs NO vector instruction (every VALU
  // instruction stalls the fp32 MFMA stream of its SIMD, DESIGN.md cost model).  A lane whose slot does not exist (halo
  // outside the image, pixels past the end, channels past C in a partial last chunk) carries an offset beyond the
  // resource's range: the hardware range check -- which looks at the per-lane offset only, not at the SGPR offset --
  // then returns zeros without touching memory: the zero padding.  a_offB is the lane's byte offset from the tile's
  // origin (3x3: halo pixel (y0-1, x0-1); 1x1: first pixel); padding slots (never read by the MFMA loop) carry 0.
  constexpr unsigned OOB = 0x80000000u;
  int a_offB[A_IT], a_key[A_IT];                        // a_key = plane << 16 | tile row << 8 | tile col (1x1: flat pixel), -1 = padding
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int slot = it * NTHR + tid;
    const int v = slot / NPIXP, pix = slot - v * NPIXP;
    const bool real = v < KV && pix < NPIX;
    const int r = (TAPS == 9) ? pix / 18 : 0, c = (TAPS == 9) ? pix - r * 18 : pix;
    a_key[it] = real ? (v << 16 | ((TAPS == 9) ? (r << 8 | c) : c)) : -1;
    a_offB[it] = real ? ((r * a.W + c) * a.x_pitch + 4 * v) * 4 : 0;
  }
  int w_offB[W_IT];                                     // byte offset inside one chunk's packed weights (padding slots: 0)
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int slot = it * NTHR + tid;
    const int v = slot / WROWS, tn = slot - v * WROWS;
    const int tap = tn / BN, n = tn - tap * BN;
    w_offB[it] = (v < KV) ? ((v * TAPS + tap) * a.Npad + n0 + n) * 16 : 0;
  }
  const unsigned w_chunkB = (unsigned)(KV * TAPS * 16) * (unsigned)a.Npad;
  // activation resource: for 3x3 based one halo row + one halo column BEFORE the window's first element, so that every
  // tile's SGPR offset (p0 * pitch * 4; the host checks it stays below 3 GiB) is non-negative
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - ((TAPS == 9) ? (long long)(a.W + 1) * a.x_pitch : 0ll)), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);

  // All fields are wave-uniform (SGPRs).  p0 = flat index of the tile's first output pixel.
  struct TilePos { int y0, x0, inner; long long p0; unsigned soff; };     // soff: byte offset of the tile origin from the resource base
  auto tile_pos = [&](int t) {
    TilePos tp;
    if (TAPS == 9) {
      const int tx = t % a.tiles_x; t /= a.tiles_x;
      const int ty = t % a.tiles_y; const int b = t / a.tiles_y;
      tp.y0 = ty * TH; tp.x0 = tx * 16;
      tp.p0 = ((long long)b * a.H + tp.y0) * a.W + tp.x0;
      tp.soff = (unsigned)(tp.p0 * a.x_pitch * 4);
      // sign-bit arithmetic keeps the flag a plain SGPR integer (y0 >= 1, y0 + TH + 1 <= H, x0 >= 1, x0 + 17 <= W)
      tp.inner = (int)(((unsigned)(-tp.y0) & (unsigned)(tp.y0 + TH - a.H) & (unsigned)(-tp.x0) & (unsigned)(tp.x0 + 16 - a.W)) >> 31);
    } else {
      tp.y0 = 0; tp.x0 = 0;
      tp.p0 = (long long)t * (TH * 16);
      tp.soff = (unsigned)(tp.p0 * a.x_pitch * 4);
      tp.inner = (int)((unsigned long long)(tp.p0 + TH * 16 - a.total_px - 1) >> 63);      // p0 + TH*16 <= total_px
    }
    return tp;
  };
  // Per-lane offsets of the stage being fetched, evaluated once per (tile, mask situation) -- not per stage: interior
  // tiles take the plain offsets; border tiles and the partial last K chunk send non-existent slots out of range.
  const int has_partial = (a.C % KC) != 0;
  int a_offT[A_IT];
  auto stage_offsets = [&](const TilePos tp, int cc) {
    const bool part = has_partial && (cc + 1) * KC > a.C;
    if (tp.inner && !part) {                                              // uniform
#pragma unroll
      for (int it = 0; it < A_IT; ++it) a_offT[it] = a_offB[it];
      return;
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int key = a_key[it];
      bool ok = key >= 0;
      if (TAPS == 9) ok = ok && (unsigned)(tp.y0 + ((key >> 8) & 255) - 1) < (unsigned)a.H && (unsigned)(tp.x0 + (key & 255) - 1) < (unsigned)a.W;
      else ok = ok && tp.p0 + (key & 0xffff) < a.total_px;
      if (part) ok = ok && cc * KC + 4 * (key >> 16) < a.C;
      a_offT[it] = ok ? a_offB[it] : (int)OOB;
    }
  };
  // Straight-line DMA issue (the tap loop must stay ONE basic block so the compiler can hoist the next tap's LDS reads
  // above this tap's MFMAs)
  auto dma_act_one = [&](int it, unsigned soff, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_t)(actB + (buf * ASLOTS + it * NTHR + wm_s * 64) * 4), 16, a_offT[it], (int)soff, 0, 0);
  };
  auto dma_w_one = [&](int it, int cc, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_ptr_t)(wB + (buf * WSLOTS + it * NTHR + wm_s * 64) * 4), 16, w_offB[it],
                                             (int)((unsigned)cc * w_chunkB), 0, 0);
  };

  f32x4 acc[MT][NT], outv[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // epilogue constants: the slice's bias in LDS (zeros without bias / beyond N) and the lane's element offset inside
  // a tile's output window
  float* const biasL = wB + (w_stationary ? 1 : 2) * WSLOTS * 4;
  if (tid < BN) biasL[tid] = (a.bias && n0 + tid < a.N) ? a.bias[n0 + tid] : 0.f;   // published by the first stage barrier
  int o_off[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
    o_off[i] = ((TAPS == 9) ? (wm * MT + i) * a.W + lr : (wm * MT + i) * 16 + lr) * a.y_pitch + 4 * g;
  // check-free epilogue: whole tiles whose mask / scale tensors (if any) share the output's geometry, so one per-lane
  // offset addresses all of them (forward layers: bias + ReLU; dgrads: accumulate, ReLU-backward mask, dropout scale)
  const bool plain_epi = (!a.ymul || (a.ymul_pitch == a.y_pitch && a.ymul_coff == a.y_coff)) &&
                         (!a.ymask || (a.ymask_pitch == a.y_pitch && a.ymask_coff == a.y_coff));
  const int acc_i = a.accumulate, has_mul = a.ymul != nullptr, has_mask = a.ymask != nullptr;
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();             // branch-free ReLU switch (one v_max per element)

  TilePos cur = tile_pos(tile);
  stage_offsets(cur, 0);
  // prologue: stage 0 into buffer 0
#pragma unroll
  for (int it = 0; it < A_IT; ++it) dma_act_one(it, cur.soff, 0);
#pragma unroll
  for (int it = 0; it < W_IT; ++it) dma_w_one(it, 0, 0);
  int sbuf = 0, wbuf = 0;              // buffers holding the stage about to be computed
  bool pending = false;                // finished accumulators wait in outv for their store
  TilePos ptp = cur;                   // ... of this tile

  // store a finished tile (bias / accumulate / masks / ReLU fused); called right AFTER a stage barrier so the
  // stores drain under the next stage's MFMAs instead of in front of the barrier's vmcnt(0)
  auto flush = [&](const TilePos tp) {
    // channel of the slice's first group: plain conv n0; fused expand: groups alternate 1x1 / 3x3, both halves advance
    // by 16 channels per PAIR of groups (n0 is a multiple of 32 there)
    const int ch0 = FUSE ? (n0 >> 1) : n0;
    float* ybase = a.y + tp.p0 * a.y_pitch + a.y_coff + ch0;             // uniform
    const float* mulbase = a.ymul + tp.p0 * a.y_pitch + a.y_coff + ch0;  // (same geometry as y on the fast path)
    const float* maskbase = a.ymask + tp.p0 * a.y_pitch + a.y_coff + ch0;
    const bool whole = ((TAPS == 9) ? (tp.y0 + TH <= a.H && tp.x0 + 16 <= a.W) : (tp.p0 + TH * 16 <= a.total_px)) && n0 + BN <= a.N;
    if (plain_epi && whole) {                                             // uniform fast path: no bounds checks, no 64-bit math
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x4 v = outv[i][j];
          v += *(const f32x4*)(biasL + j * 16 + 4 * g);
          const int off = o_off[i] + (FUSE ? ((j >> 1) * 16 + ((j & 1) ? a.fuse_e : 0)) : j * 16);
          if (acc_i) v += *(const f32x4*)(ybase + off);
          if (has_mul) v *= *(const f32x4*)(mulbase + off);
          if (has_mask) {
            const f32x4 m = *(const f32x4*)(maskbase + off);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
          }
          v = sqd_relu4(v, relu_lo);
          *(f32x4*)(ybase + off) = v;
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      bool valid;
      long long gp;
      if (TAPS == 9) {
        valid = tp.y0 + wm * MT + i < a.H && tp.x0 + lr < a.W;
        gp = tp.p0 + (long long)(wm * MT + i) * a.W + lr;
      } else {
        gp = tp.p0 + (wm * MT + i) * 16 + lr;
        valid = gp < a.total_px;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int np = n0 + j * 16 + 4 * g;                                     // channel in packed (interleaved) order
        if (!valid || np >= a.N) continue;
        const int n = FUSE ? (((j & 1) ? a.fuse_e : 0) + ((n0 >> 1) + (j >> 1) * 16 + 4 * g)) : np;
        f32x4 v = outv[i][j];
        v += *(const f32x4*)(biasL + j * 16 + 4 * g);
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
  };


  for (;;) {
    const int more_i = (int)((unsigned)(tile + tstride - ntiles) >> 31);     // tile + tstride < ntiles
    const bool more = more_i != 0;
    const TilePos nxt = tile_pos(more ? tile + tstride : tile);
    for (int cc = 0; cc < nchunks; ++cc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the stage's DMA has landed (explicit: the compiler
                                                        // only tracks the issuing wave's own reads of DMA-written LDS)
      __syncthreads();                 // ... and is published; all waves left the previous stage
      if (pending) { flush(ptp); pending = false; }
      const int last_i = 1 - (int)((unsigned)(cc + 1 - nchunks) >> 31);      // cc == nchunks - 1
      const bool last = last_i != 0;
      const int ncc = last ? 0 : cc + 1;
      // plain SALU integers (sign-bit arithmetic): as i1 values the compiler round-trips them through VGPRs
      // The next stage's DMA is issued unconditionally: after the very last stage it re-fetches this workgroup's last tile
      // into the idle buffer (never read; retired by the vmcnt(0) before the kernel ends) -- cheaper than a branch per tap.
      // Its per-lane offsets change only at a tile switch and in front of a partial last chunk (uniform branches, outside
      // the tap loop).
      if (last) stage_offsets(nxt, 0);
      else if (has_partial && ncc == nchunks - 1) stage_offsets(cur, ncc);
      const unsigned nsoff = (last ? nxt.soff : cur.soff) + (unsigned)(ncc * KC * 4);
      // per-lane LDS bases of this stage (one VALU add each per stage); every read below is base + immediate
      const float* actL = actB + sbuf * ASLOTS * 4 + (g * NPIXP + ((TAPS == 9) ? wm * MT * 18 : wm * MT * 16) + lr) * 4;
      const float* wL = wB + wbuf * WSLOTS * 4 + (g * WROWS + lr) * 4;

      // ---- MFMA over (tap, k), software-pipelined over two operand register sets (the loop is fully unrolled and
      // branch-free); one DMA instruction of the next stage is issued per step ----
      auto load_ops = [&](int step, f32x4 (&bfr)[MT], f32x4 (&afr)[NT]) {
        const int tap = step / (KC / 16), sk = step - tap * (KC / 16);
        const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = (TAPS == 9) ? (i + dy) * 18 + dx : i * 16;                 // compile-time
          bfr[i] = *(const f32x4*)(actL + (4 * sk * NPIXP + row) * 4);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (FUSE && !(j & 1) && tap != 4) continue;                              // 1x1 group: centre tap only
          afr[j] = *(const f32x4*)(wL + (4 * sk * WROWS + tap * BN + j * 16) * 4);
        }
      };
      auto mfma_half = [&](int step, const f32x4 (&bfr)[MT], const f32x4 (&afr)[NT], int half) {
        const int tap = step / (KC / 16);
#pragma unroll
        for (int t = 2 * half; t < 2 * half + 2; ++t)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if (FUSE && !(j & 1) && tap != 4) continue;
              acc[i][j] = mfma16(afr[j][t], bfr[i][t], acc[i][j]);
            }
      };
      f32x4 bf0[MT], af0[NT], bf1[MT], af1[NT];
      load_ops(0, bf0, af0);
#pragma unroll
      for (int step = 0; step < STEPS; ++step) {
        // spread A_IT + W_IT DMA instructions over STEPS steps (the first steps take the remainder)
#pragma unroll
        for (int q = 0; q < A_IT + W_IT; ++q) {
          if (q * STEPS / (A_IT + W_IT) != step) continue;
          if (q < A_IT) dma_act_one(q < A_IT ? q : 0, nsoff, sbuf ^ 1);
          else if (!WSTAT) dma_w_one(q - A_IT, ncc, wbuf ^ 1);
        }
        // The next step's reads go out in the MIDDLE of this step's MFMAs: when the next step starts (and the compiler's
        // wait -- always lgkmcnt(0), it does not count LDS returns past an LDS-DMA -- is reached) they were issued half a
        // step of MFMAs ago and nothing newer is outstanding, so the wait costs nothing.
        if (step & 1) {
          mfma_half(step, bf1, af1, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (step + 1 < STEPS) load_ops(step + 1, bf0, af0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(step, bf1, af1, 1);
        } else {
          mfma_half(step, bf0, af0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (step + 1 < STEPS) load_ops(step + 1, bf1, af1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(step, bf0, af0, 1);
        }
      }

      if (last) {                      // tile finished: park the result, store it after the next barrier
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) { outv[i][j] = acc[i][j]; acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        pending = true; ptp = cur;
      }
      sbuf ^= 1;
      if (!w_stationary) wbuf ^= 1;
    }
    if (!more) break;
    tile += tstride;
    cur = nxt;
  }
  if (pending) flush(ptp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may still be in flight when the LDS is released
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-stationary, barrier-free 1x1 convolution (tile configurations with dma = 3: conv_ws<NT, WAVES, D>).
//
// Why (round 2 measurements, DESIGN.md "1x1 family"): on the 24x78 / 48x156 layers conv_dma_kernel<1,...> is bound neither
// by HBM nor by the matrix cores but by how a workgroup is fed: one stage (8-12 KB) in flight per workgroup per memory
// round trip, a workgroup barrier per stage, and whole workgroups (4 waves x 16 pixels) as the unit of load balance on
// a layer that only has 2340 16-pixel MFMA columns for 1024 SIMDs.  Here the slice's WHOLE weight matrix (C x 16 NT
// floats, k-quad-major) is fetched into LDS once per workgroup and stays; a wave then owns 16-pixel tiles on its own:
// its activation stages (16 pixels x 32 channels = 2 KB) arrive by LDS-DMA in a wave-private ring D stages deep, retired
// by counted s_waitcnt vmcnt(N) -- no workgroup barrier after the weight load, every wave runs free, and the unit of
// load balance is one wave x one 16-pixel tile.  Same arithmetic as the other kernels (k-ordered fp32 MFMA chain).
// Weights: the packed layout of a KC = 32 plan, [C/32][8][Npad][4] = [k-quad][row][4].
template <int NT, int WV, int D>
__global__ __launch_bounds__(WV * 64) void conv_ws_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NTHR = WV * 64, BN = 16 * NT, KC = 32;
  constexpr int NDMA = 2;                               // LDS-DMA instructions per stage and wave (128 slots of 16 B)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nchunks = (a.C + KC - 1) / KC, nplanes = nchunks * 8;
  const int wslots = (nplanes * BN + NTHR - 1) / NTHR * NTHR;
  float* const wS = smem;                               // [nplanes][BN][4]
  float* const ringB = smem + wslots * 4;               // [WV][D][8 planes][16 px][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wgq = (int)blockIdx.x >> 3;
  const int n0 = (wgq % a.nslices) * BN;
  const int stream = (wgq / a.nslices) * 8 + ((int)blockIdx.x & 7);
  const int ntasks = a.ntiles;                          // 16-pixel tiles
  const int tstride = a.gx * WV;                        // wave streams in flight
  int tile = stream * WV + wave_s;
  constexpr unsigned OOB = 0x80000000u;

  // ---- the slice's weights, once ----
  {
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);
    for (int s0 = 0; s0 < wslots; s0 += NTHR) {
      const int slot = s0 + tid;
      const int plane = slot / BN, row = slot - plane * BN;
      const int off = (plane < nplanes) ? (int)(((unsigned)plane * a.Npad + n0 + row) * 16u) : (int)OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_ptr_t)(wS + (s0 + wave_s * 64) * 4), 16, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // the only workgroup barrier
  }
  if (tile >= ntasks) return;

  // ---- per-lane constants ----
  int a_offB[NDMA], a_key[NDMA];                        // byte offset of the lane's slot inside a stage; key = plane << 8 | px
#pragma unroll
  for (int it = 0; it < NDMA; ++it) {
    const int slot = it * 64 + lane;
    const int v = slot >> 4, px = slot & 15;
    a_offB[it] = (px * a.x_pitch + 4 * v) * 4;
    a_key[it] = v << 8 | px;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + a.x_coff), 0, 0x7ffffff0, 0x00020000);
  float* const ringW = ringB + wave_s * (D * 128 * 4);
  const int full_c = (a.C % KC) == 0;
  auto issue = [&](int ptile, int pcc, int slotr) {     // request stage (ptile, pcc) into ring slot slotr
    const unsigned soff = ((unsigned)ptile * 16u * (unsigned)a.x_pitch + (unsigned)pcc * KC) * 4u;
    const bool full = ((long long)ptile * 16 + 16 <= a.total_px) && (full_c || (pcc + 1) * KC <= a.C);     // uniform
#pragma unroll
    for (int it = 0; it < NDMA; ++it) {
      int off = a_offB[it];
      if (!full) {
        const int key = a_key[it];
        const bool ok = (long long)ptile * 16 + (key & 255) < a.total_px && pcc * KC + 4 * (key >> 8) < a.C;
        off = ok ? off : (int)OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_t)(ringW + (slotr * 128 + it * 64) * 4), 16, off, (int)soff, 0, 0);
    }
  };
  f32x4 biasv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + 4 * g;
    biasv[j] = (a.bias && n < a.N) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int o_offB = (lr * a.y_pitch + 4 * g) * 4;
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + a.y_coff + n0), 0, 0x7ffffff0, 0x00020000);
  const bool same_geom = (!a.ymul || (a.ymul_pitch == a.y_pitch)) && (!a.ymask || (a.ymask_pitch == a.y_pitch));
  const __amdgpu_buffer_rsrc_t mulres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ymul ? a.ymul + a.ymul_coff + n0 : a.y), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t maskres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ymask ? a.ymask + a.ymask_coff + n0 : a.y), 0, 0x7ffffff0, 0x00020000);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  auto store16 = [&](f32x4 v, int voff, int soff) {     // (MUBUF store + SGPR soffset write-after-read hazard: see conv_wino.hip)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yres, voff, soff, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();
  const int acc_i = a.accumulate, has_mul = a.ymul != nullptr, has_mask = a.ymask != nullptr, has_drop = a.drop_state != nullptr;
  const bool plain_epi = !acc_i && !has_mul && !has_mask && !has_drop;
  const SqdDrop dropk = has_drop ? sqd_drop_key(a.drop_state[0], a.drop_state[1], a.drop_keep, a.drop_scale) : SqdDrop{0u, 0u, 0u, 0.f};
  const float* const wL = wS + (g * BN + lr) * 4;       // + (plane group * 4 * BN + j * 16) * 4: per-lane base of the A operands
  const float* const bL0 = ringW + (g * 16 + lr) * 4;   // + (slot * 128 + sk * 64) * 4

  // ---- prefetch cursor: D - 1 stages ahead; clamps on this wave's last tile ----
  int ptile = tile, pcc = 0;
  auto advance = [&]() {
    ++pcc;
    if (pcc == nchunks) { pcc = 0; ptile = (ptile + tstride < ntasks) ? ptile + tstride : ptile; }
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) { issue(ptile, pcc, s); advance(); }
  int rs = 0;                            // ring slot of the stage being computed
  int cc = 0;                            // its K chunk
  int stores_behind = 0;
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // operands of one stage: activations b[sk] (16 pixels x 16 k) from ring slot r, weights a[sk][j] of chunk c
  struct Ops { f32x4 b0, b1, a0[NT], a1[NT]; };
  auto load_ops = [&](Ops& o, int r, int c) {
    const float* const bL = bL0 + r * 128 * 4;
    const float* const wC = wL + c * (8 * BN * 4);
    o.b0 = *(const f32x4*)(bL); o.b1 = *(const f32x4*)(bL + 64 * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) o.a0[j] = *(const f32x4*)(wC + j * 64);
#pragma unroll
    for (int j = 0; j < NT; ++j) o.a1[j] = *(const f32x4*)(wC + (4 * BN + j * 16) * 4);
  };
  // One stage: the matrix work on `cur`, with the NEXT stage's operands fetched into `nxt` in the middle of it (the next
  // stage's activations were requested D - 1 stages ago: a counted wait retires them first) and the request for the stage
  // D - 1 ahead issued at the top -- a wave alone on its SIMD keeps the matrix pipe fed.  Returns false after the wave's
  // last tile.
  auto stage = [&](const Ops& cur, Ops& nxt) -> bool {
    issue(ptile, pcc, (rs == 0) ? D - 1 : rs - 1);           // into the slot consumed by the previous stage
    advance();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mfma16(cur.a0[j][t], cur.b0[t], acc[j]);
    __builtin_amdgcn_sched_barrier(0);
    // stage k + 1 has landed: all but the 2 (D - 2) youngest requests (+ a just-finished tile's stores) are done
    if (__builtin_amdgcn_readfirstlane(stores_behind)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (D - 2) + NT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (D - 2)) : "memory");
    stores_behind = 0;
    const int rsn = (rs == D - 1) ? 0 : rs + 1;
    const int ncc = (cc + 1 == nchunks) ? 0 : cc + 1;
    load_ops(nxt, rsn, ncc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mfma16(cur.a1[j][t], cur.b1[t], acc[j]);
    bool more = true;
    if (ncc == 0) {
      // ---- tile finished: bias, (accumulate, dropout scale, ReLU-backward mask), ReLU, one 16-byte store per block ----
      const int ysoff = (int)((unsigned)tile * 16u * (unsigned)a.y_pitch * 4u);
      const bool whole = (long long)tile * 16 + 16 <= a.total_px && n0 + BN <= a.N;
      if (whole && plain_epi) {
#pragma unroll
        for (int j = 0; j < NT; ++j) store16(sqd_relu4(acc[j] + biasv[j], relu_lo), o_offB + j * 64, ysoff);
        stores_behind = 1;
      } else {
        const bool pvalid = (long long)tile * 16 + lr < a.total_px;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (!pvalid || n0 + j * 16 + 4 * g >= a.N) continue;
          const int off = o_offB + j * 64;
          f32x4 v = acc[j] + biasv[j];
          if (acc_i) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, off, ysoff, 0));
          if (has_mul) v *= __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mulres, (lr * a.ymul_pitch + 4 * g) * 4 + j * 64, (int)((unsigned)tile * 16u * (unsigned)a.ymul_pitch * 4u), 0));
          if (has_drop)        // element index in the output BUFFER (pixel * pitch + channel): the same element gets the same bits from any kernel
            v *= sqd_drop_mul4((unsigned long long)(((long long)tile * 16 + lr) * a.y_pitch + a.y_coff + n0 + j * 16 + 4 * g) >> 2, dropk);
          if (has_mask) {
            const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(maskres, (lr * a.ymask_pitch + 4 * g) * 4 + j * 64, (int)((unsigned)tile * 16u * (unsigned)a.ymask_pitch * 4u), 0));
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
          }
          store16(sqd_relu4(v, relu_lo), off, ysoff);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      more = tile + tstride < ntasks;
      tile += tstride;
    }
    rs = rsn; cc = ncc;
    return more;
  };
  Ops opA, opB;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (D - 2)) : "memory");      // stage 0 has landed
  load_ops(opA, 0, 0);
  for (;;) {
    if (!stage(opA, opB)) break;
    if (!stage(opB, opA)) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may still be in flight when the LDS is released
#endif
}

static int sqd_num_cus() {
  static int cus = 0;                        // immutable per-process cache
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int NT, int WV>
static int launch_conv_ws(ConvArgs a, hipStream_t stream) {
  constexpr int BN = 16 * NT, NTHR = WV * 64, KC = 32;
  if (a.xmask) return SQD_ERR_UNSUPPORTED;
  if (a.total_px * a.x_pitch * 4 >= (3ll << 30) || a.total_px * a.y_pitch * 4 >= (3ll << 30)) return SQD_ERR_UNSUPPORTED;
  if (a.ymul && a.total_px * a.ymul_pitch * 4 >= (3ll << 30)) return SQD_ERR_UNSUPPORTED;
  if (a.ymask && a.total_px * a.ymask_pitch * 4 >= (3ll << 30)) return SQD_ERR_UNSUPPORTED;
  const int nchunks = sqd_cdiv(a.C, KC), nplanes = nchunks * 8;
  const int wslots = sqd_cdiv(nplanes * BN, NTHR) * NTHR;
  // ring depth from what the weights leave of the LDS (one workgroup per CU when they are large): 2, 3, 4, 6 or 8 stages
  const size_t wbytes = (size_t)wslots * 16, stage = (size_t)WV * 2048;
  int D = 0;
  for (int d : {8, 6, 4, 3}) if (wbytes + d * stage <= 160 * 1024 && (d <= 4 || wbytes + d * stage <= 80 * 1024)) { D = d; break; }
  if (D == 0) return SQD_ERR_UNSUPPORTED;
  const size_t lds = wbytes + D * stage;
  a.ntiles = (int)((a.total_px + 15) / 16);
  const int nslices = sqd_cdiv(a.N, BN);
  if (nslices * BN > a.Npad) return SQD_ERR_BAD_ARG;
  auto go = [&](auto kern) {
    static SqdDevOnce attr_once;                 // (per device: ADVICE round 4)
    if (int rc_attr = sqd_max_lds_once(attr_once, (const void*)kern, 160 * 1024)) return (int)rc_attr;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NTHR, lds) != hipSuccess || nb < 1) nb = 1;
    int wgs_per_cu = nb > 4 ? 4 : nb;
    if (a.wg_cap > 0 && a.wg_cap < wgs_per_cu) wgs_per_cu = a.wg_cap;
    const int slots = sqd_num_cus() * wgs_per_cu;
    int gx_max = slots / nslices; if (gx_max < 1) gx_max = 1;
    const int wave_tiles = sqd_cdiv(a.ntiles, WV);            // tiles per wave stream if there were one stream
    const int per_wg = sqd_cdiv(wave_tiles, gx_max);
    const int gx = (sqd_cdiv(wave_tiles, per_wg) + 7) & ~7;
    a.nslices = nslices; a.gx = gx;
    hipLaunchKernelGGL(kern, dim3((unsigned)(gx * nslices)), dim3(NTHR), lds, stream, a);
    return sqd_launch_status();
  };
  switch (D) {
    case 8: return go(conv_ws_kernel<NT, WV, 8>);
    case 6: return go(conv_ws_kernel<NT, WV, 6>);
    case 4: return go(conv_ws_kernel<NT, WV, 4>);
    default: return go(conv_ws_kernel<NT, WV, 3>);
  }
}

template <int TAPS, int KC, int MT, int NT>
static int launch_conv(ConvArgs a, hipStream_t stream) {
  constexpr int TH = MT * 4, BN = 16 * NT;
  constexpr int NPIX = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  constexpr int NPIXP = (NPIX + 15) & ~15;
  constexpr size_t lds = (size_t)(NPIXP + TAPS * BN) * KC * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  // waves/SIMD the register allocator must leave room for (= workgroups per CU): capped by what LDS admits
  // and by what each instantiation reaches without spilling (checked with -Rpass-analysis)
  constexpr int LDSW = (int)((160 * 1024) / lds);
  constexpr int REGW = (TAPS == 9) ? ((MT * NT <= 8 && NT <= 4) ? 2 : 1)
                                   : ((MT * NT <= 2) ? 4 : ((MT * NT <= 4 && KC == 32) || (KC == 16 && MT * NT <= 8) ? 3 : 2));
  constexpr int MINW = LDSW < REGW ? (LDSW < 1 ? 1 : LDSW) : REGW;
  auto kern = conv_igemm_kernel<TAPS, KC, MT, NT, MINW>;
  static int wgs_per_cu = 0;                 // occupancy of this instantiation (immutable once computed)
  static SqdDevOnce lds_once;                // (the LDS attribute is per device)
  if (lds > 64 * 1024 && sqd_max_lds_once(lds_once, (const void*)kern, (int)lds) != SQD_OK) return SQD_ERR_LAUNCH;
  if (wgs_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    wgs_per_cu = nb > 4 ? 4 : nb;
  }
  if (TAPS == 9) {
    a.tiles_x = sqd_cdiv(a.W, 16); a.tiles_y = sqd_cdiv(a.H, TH);
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
  } else {
    a.tiles_x = a.tiles_y = 0;
    a.ntiles = (int)((a.total_px + TH * 16 - 1) / (TH * 16));
  }
  const int nslices = sqd_cdiv(a.N, BN);
  if (nslices * BN > a.Npad) return SQD_ERR_BAD_ARG;     // packed weights too short for this slice width
  // persistent grid: at most one resident wave of workgroups, pixel tiles dealt evenly
  const int slots = sqd_num_cus() * ((a.wg_cap > 0 && a.wg_cap < wgs_per_cu) ? a.wg_cap : wgs_per_cu);
  int gx_max = slots / nslices; if (gx_max < 1) gx_max = 1;
  const int per_wg = sqd_cdiv(a.ntiles, gx_max);
  const int gx = (sqd_cdiv(a.ntiles, per_wg) + 7) & ~7;        // tile streams, a multiple of 8 (one per XCD lane)
  a.nslices = nslices; a.gx = gx;
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx * nslices)), dim3(256), lds, stream, a);
  return sqd_launch_status();
}

template <int TAPS, int KC, int MT, int NT, int WM, bool FUSE = false>
static int launch_conv_dma(ConvArgs a, hipStream_t stream) {
  constexpr int TH = MT * WM, BN = 16 * NT, NTHR = WM * 64;
  constexpr int NPIX = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  constexpr int NPIXP = (NPIX + 15) & ~15;
  constexpr int KV = KC / 4;
  constexpr int ASLOTS = (KV * NPIXP + NTHR - 1) / NTHR * NTHR, WSLOTS = (KV * TAPS * BN + NTHR - 1) / NTHR * NTHR;
  constexpr size_t lds_max = (size_t)(2 * ASLOTS + 2 * WSLOTS) * 16 + BN * sizeof(float);
  static_assert(lds_max <= 160 * 1024, "LDS budget");
  if (a.xmask) return SQD_ERR_UNSUPPORTED;               // input-side mask needs register staging (v3 path)
  // 32-bit SGPR byte offset of a tile origin / per-lane byte offsets inside a tile (buffer-resource addressing)
  if (a.total_px * a.x_pitch * 4 >= (3ll << 30) || (long long)a.W * (TH + 2) * a.x_pitch * 4 >= (1ll << 30)) return SQD_ERR_UNSUPPORTED;
  const int stationary = (a.C <= KC) ? 1 : 0;            // one K chunk: a single weight buffer suffices
  const size_t lds = (size_t)(2 * ASLOTS + (stationary ? 1 : 2) * WSLOTS) * 16 + BN * sizeof(float);
  // waves per SIMD the register allocator must leave room for: workgroups per CU (by LDS) x waves per SIMD of one
  // (the software-pipelined operand sets cost (MT + NT) * 4 more VGPRs than the accumulators alone)
  constexpr int REGW = (MT * NT <= 2) ? 4 : ((MT * NT <= 4) ? 3 : ((MT * NT <= 8) ? 2 : 1));
  constexpr int LDSW = (int)((160 * 1024) / ((size_t)(2 * ASLOTS + WSLOTS) * 16 + BN * sizeof(float))) * (WM / 4);
  constexpr int MINW0 = LDSW < REGW ? (LDSW < 1 ? 1 : LDSW) : REGW;
  constexpr int MINW = (MINW0 < WM / 4) ? WM / 4 : MINW0;
  auto kern = stationary ? conv_dma_kernel<TAPS, KC, MT, NT, WM, MINW, FUSE, true> : conv_dma_kernel<TAPS, KC, MT, NT, WM, MINW, FUSE, false>;
  static int wgs_per_cu[2] = {0, 0};
  static SqdDevOnce lds_once[2];
  if (lds_max > 64 * 1024 && sqd_max_lds_once(lds_once[stationary], (const void*)kern, (int)lds_max) != SQD_OK) return SQD_ERR_LAUNCH;
  if (wgs_per_cu[stationary] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NTHR, lds) != hipSuccess || nb < 1) nb = 1;
    wgs_per_cu[stationary] = nb > 6 ? 6 : nb;
  }
  if (TAPS == 9) {
    a.tiles_x = sqd_cdiv(a.W, 16); a.tiles_y = sqd_cdiv(a.H, TH);
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
  } else {
    a.tiles_x = a.tiles_y = 0;
    a.ntiles = (int)((a.total_px + TH * 16 - 1) / (TH * 16));
  }
  const int nslices = sqd_cdiv(a.N, BN);
  if (nslices * BN > a.Npad) return SQD_ERR_BAD_ARG;
  const int slots = sqd_num_cus() * ((a.wg_cap > 0 && a.wg_cap < wgs_per_cu[stationary]) ? a.wg_cap : wgs_per_cu[stationary]);
  int gx_max = slots / nslices; if (gx_max < 1) gx_max = 1;
  const int per_wg = sqd_cdiv(a.ntiles, gx_max);
  const int gx = (sqd_cdiv(a.ntiles, per_wg) + 7) & ~7;        // tile streams, a multiple of 8 (one per XCD lane)
  a.nslices = nslices; a.gx = gx;
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx * nslices)), dim3(NTHR), lds, stream, a);
  return sqd_launch_status();
}

// Tile configurations (cfg_id) -> template instance.  The host picks per layer; any config is
// correct for any shape (edges are masked, partial K chunks zero-filled).
struct ConvCfg { int taps, kc, mt, nt, dma; };   // dma: 0 = register-staged 4 waves, 1 = LDS-DMA 4 waves, 2 = LDS-DMA 8 waves
static const ConvCfg kConvCfgs[] = {
    {1, 16, 2, 4, 0},  // 0  1x1, small C (expand1x1), 128 px x 64 ch
    {1, 32, 2, 1, 0},  // 1  1x1 squeeze N<=16
    {1, 32, 2, 2, 0},  // 2  N<=32
    {1, 32, 2, 3, 0},  // 3  N<=48
    {1, 32, 2, 4, 0},  // 4  N<=64
    {1, 32, 2, 6, 0},  // 5  N<=96
    {1, 32, 1, 3, 0},  // 6  64-px tiles (late, small layers)
    {1, 32, 1, 4, 0},  // 7
    {1, 32, 1, 6, 0},  // 8
    {1, 16, 1, 4, 0},  // 9  1x1 small C, 64 px
    {9, 16, 2, 4, 0},  // 10 3x3, 8x16 px x 64 ch
    {9, 16, 2, 5, 0},  // 11 3x3, 8x16 px x 80 ch (ConvDet N=72)
    {9, 16, 1, 4, 0},  // 12 3x3, 4x16 px x 64 ch
    {9, 16, 1, 5, 0},  // 13 3x3, 4x16 px x 80 ch
    {9, 16, 2, 2, 0},  // 14 3x3, 8x16 px x 32 ch
    {9, 16, 2, 6, 0},  // 15 3x3, 8x16 px x 96 ch
    {9, 16, 1, 6, 0},  // 16 3x3, 4x16 px x 96 ch
    {9, 16, 2, 3, 0},  // 17 3x3, 8x16 px x 48 ch
    {9, 16, 2, 1, 0},  // 18 3x3, 8x16 px x 16 ch (dgrad into a 16-channel squeeze)
    {9, 16, 1, 1, 0},  // 19 3x3, 4x16 px x 16 ch
    {9, 16, 1, 2, 0},  // 20 3x3, 4x16 px x 32 ch
    {9, 16, 1, 3, 0},  // 21 3x3, 4x16 px x 48 ch
    {9, 16, 4, 1, 0},  // 22 3x3, 16x16 px x 16 ch
    {9, 16, 4, 2, 0},  // 23 3x3, 16x16 px x 32 ch
    {1, 16, 1, 2, 0},  // 24 1x1 small C, 64 px x 32 ch
    {1, 16, 2, 2, 0},  // 25 1x1 small C, 128 px x 32 ch
    {1, 16, 4, 2, 0},  // 26 1x1 small C, 256 px x 32 ch
    {1, 16, 4, 4, 0},  // 27 1x1 small C, 256 px x 64 ch
    {1, 32, 1, 1, 0},  // 28 1x1, 64 px x 16 ch
    {1, 32, 1, 2, 0},  // 29 1x1, 64 px x 32 ch
    {1, 32, 4, 1, 0},  // 30 1x1, 256 px x 16 ch
    {1, 32, 4, 2, 0},  // 31 1x1, 256 px x 32 ch
    {1, 64, 1, 2, 0},  // 32 1x1, KC=64, 64 px x 32 ch
    {1, 64, 1, 3, 0},  // 33 1x1, KC=64, 64 px x 48 ch
    {1, 64, 1, 4, 0},  // 34 1x1, KC=64, 64 px x 64 ch
    {1, 64, 2, 1, 0},  // 35 1x1, KC=64, 128 px x 16 ch
    {1, 64, 2, 2, 0},  // 36 1x1, KC=64, 128 px x 32 ch
    // ---- LDS-DMA double-buffered variants (dma = 1) ----
    {9, 16, 1, 1, 1},  // 37
    {9, 16, 1, 2, 1},  // 38
    {9, 16, 1, 3, 1},  // 39
    {9, 16, 1, 4, 1},  // 40
    {9, 16, 2, 1, 1},  // 41
    {9, 16, 2, 2, 1},  // 42
    {9, 16, 2, 3, 1},  // 43
    {9, 16, 2, 4, 1},  // 44
    {9, 16, 4, 1, 1},  // 45
    {9, 16, 4, 2, 1},  // 46
    {1, 16, 1, 2, 1},  // 47
    {1, 16, 1, 4, 1},  // 48
    {1, 16, 2, 4, 1},  // 49
    {1, 16, 4, 4, 1},  // 50
    {1, 32, 1, 1, 1},  // 51
    {1, 32, 1, 2, 1},  // 52
    {1, 32, 1, 3, 1},  // 53
    {1, 32, 1, 4, 1},  // 54
    {1, 32, 1, 6, 1},  // 55
    {1, 32, 2, 1, 1},  // 56
    {1, 32, 2, 2, 1},  // 57
    {1, 32, 4, 1, 1},  // 58
    {1, 64, 1, 2, 1},  // 59
    {1, 64, 1, 3, 1},  // 60
    {1, 64, 1, 4, 1},  // 61
    {1, 64, 1, 6, 1},  // 62
    {1, 64, 2, 1, 1},  // 63
    {1, 64, 2, 2, 1},  // 64
    // ---- LDS-DMA, 8-wave workgroups (dma = 2): tile = mt*8 rows ----
    {9, 16, 1, 2, 2},  // 65  8x16 px x 32 ch
    {9, 16, 1, 3, 2},  // 66
    {9, 16, 1, 4, 2},  // 67  8x16 px x 64 ch
    {9, 16, 2, 1, 2},  // 68  16x16 px x 16 ch
    {9, 16, 2, 2, 2},  // 69  16x16 px x 32 ch
    {9, 16, 2, 3, 2},  // 70
    {9, 16, 2, 4, 2},  // 71  16x16 px x 64 ch
    {9, 16, 1, 5, 2},  // 72  8x16 px x 80 ch
    {1, 32, 1, 2, 2},  // 73
    {1, 32, 1, 4, 2},  // 74
    {1, 32, 2, 2, 2},  // 75
    {1, 16, 1, 4, 2},  // 76
    {1, 16, 2, 4, 2},  // 77
    // ---- weight-stationary, barrier-free 1x1 (dma = 3: 4 waves, dma = 4: 8 waves); KC = 32 packing, 16-pixel wave tiles ----
    {1, 32, 1, 1, 3},  // 78
    {1, 32, 1, 2, 3},  // 79
    {1, 32, 1, 3, 3},  // 80
    {1, 32, 1, 4, 3},  // 81
    {1, 32, 1, 6, 3},  // 82
    {1, 32, 1, 2, 4},  // 83
    {1, 32, 1, 3, 4},  // 84
    {1, 32, 1, 4, 4},  // 85
    {1, 32, 1, 6, 4},  // 86
};
static const int kNumConvCfgs = (int)(sizeof(kConvCfgs) / sizeof(kConvCfgs[0]));

extern "C" int sqd_conv_num_cfgs() { return kNumConvCfgs; }

// 0: register-staged (4 waves); 1: LDS-DMA, 4 waves; 2: LDS-DMA, 8 waves; 3 / 4: weight-stationary barrier-free 1x1, 4 / 8 waves
// (no xmask support when != 0); -1: bad id
extern "C" int sqd_conv_cfg_is_dma(int cfg_id) {
  if (cfg_id < 0 || cfg_id >= kNumConvCfgs) return -1;
  return kConvCfgs[cfg_id].dma;
}

extern "C" int sqd_conv_cfg_info(int cfg_id, int* taps, int* kc, int* tile_px, int* bn) {
  SQD_CHECK_ARG(cfg_id >= 0 && cfg_id < kNumConvCfgs);
  const ConvCfg& c = kConvCfgs[cfg_id];
  if (taps) *taps = c.taps;
  if (kc) *kc = c.kc;
  if (tile_px) *tile_px = (c.dma >= 3) ? 16 * (c.dma == 4 ? 8 : 4) : c.mt * (c.dma == 2 ? 8 : 4) * 16;
  if (bn) *bn = 16 * c.nt;
  return SQD_OK;
}

static int conv_fwd_impl(const float* x, const float* w_packed, const float* bias, float* y,
                         const float* xmask, const float* ymask, const float* ymul, int B, int H, int W, int C,
                         int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu,
                         int accumulate, int xmask_pitch, int xmask_coff, int ymask_pitch, int ymask_coff,
                         int ymul_pitch, int ymul_coff, int cfg_id, void* stream, const unsigned long long* drop_state, int drop_keep,
                         float drop_scale) {
  SQD_CHECK_ARG(x && w_packed && y);
  SQD_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && N > 0);
  // cfg_id = tile configuration + 1000 * k: k > 0 caps the persistent grid at k workgroups per CU (fewer, longer tile
  // streams balance better on small layers; the tuner measures it), k = 0 fills the occupancy
  const int wg_cap = cfg_id >= 0 ? cfg_id / 1000 : 0;
  if (cfg_id >= 0) cfg_id %= 1000;
  SQD_CHECK_ARG(cfg_id >= 0 && cfg_id < kNumConvCfgs && wg_cap <= 8);
  SQD_CHECK_ARG((C & 3) == 0 && (N & 3) == 0);
  SQD_CHECK_ARG((x_pitch & 3) == 0 && (x_coff & 3) == 0 && (y_pitch & 3) == 0 && (y_coff & 3) == 0);
  SQD_CHECK_ARG(x_coff + C <= x_pitch && y_coff + N <= y_pitch);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)w_packed & 15) == 0);
  SQD_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0);
  if (xmask) SQD_CHECK_ARG((xmask_pitch & 3) == 0 && (xmask_coff & 3) == 0 && xmask_coff + C <= xmask_pitch && ((uintptr_t)xmask & 15) == 0);
  if (ymask) SQD_CHECK_ARG((ymask_pitch & 3) == 0 && (ymask_coff & 3) == 0 && ymask_coff + N <= ymask_pitch && ((uintptr_t)ymask & 15) == 0);
  if (ymul) SQD_CHECK_ARG((ymul_pitch & 3) == 0 && (ymul_coff & 3) == 0 && ymul_coff + N <= ymul_pitch && ((uintptr_t)ymul & 15) == 0);
  ConvArgs a = {};
  a.ymask = ymask; a.ymul = ymul; a.ymask_pitch = ymask_pitch; a.ymask_coff = ymask_coff; a.ymul_pitch = ymul_pitch; a.ymul_coff = ymul_coff;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.xmask = xmask;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff;
  a.relu = relu; a.accumulate = accumulate; a.tiles_x = a.tiles_y = 0; a.ntiles = 0; a.nslices = 1; a.gx = 8; a.wg_cap = wg_cap; a.fuse_e = 0;
  a.xmask_pitch = xmask_pitch; a.xmask_coff = xmask_coff;
  a.total_px = (long long)B * H * W;
  a.drop_state = drop_state; a.drop_keep = drop_keep; a.drop_scale = drop_scale;
  hipStream_t s = (hipStream_t)stream;
  const ConvCfg& c = kConvCfgs[cfg_id];
  if (drop_state && c.dma < 3) return SQD_ERR_UNSUPPORTED;      // only the weight-stationary 1x1 family carries the dropout epilogue
#define SQD_WS_CASE(Nn) \
  if (c.dma >= 3 && c.nt == Nn) return (c.dma == 4) ? launch_conv_ws<Nn, 8>(a, s) : launch_conv_ws<Nn, 4>(a, s);
  SQD_WS_CASE(1) SQD_WS_CASE(2) SQD_WS_CASE(3) SQD_WS_CASE(4) SQD_WS_CASE(6)
#undef SQD_WS_CASE
  if (c.dma >= 3) return SQD_ERR_UNSUPPORTED;
#define SQD_DMA8_CASE(T, K, M, Nn) \
  if (c.dma == 2 && c.taps == T && c.kc == K && c.mt == M && c.nt == Nn) return launch_conv_dma<T, K, M, Nn, 8>(a, s);
  SQD_DMA8_CASE(9, 16, 1, 2) SQD_DMA8_CASE(9, 16, 1, 3) SQD_DMA8_CASE(9, 16, 1, 4) SQD_DMA8_CASE(9, 16, 2, 1)
  SQD_DMA8_CASE(9, 16, 2, 2) SQD_DMA8_CASE(9, 16, 2, 3) SQD_DMA8_CASE(9, 16, 2, 4) SQD_DMA8_CASE(9, 16, 1, 5)
  SQD_DMA8_CASE(1, 32, 1, 2) SQD_DMA8_CASE(1, 32, 1, 4) SQD_DMA8_CASE(1, 32, 2, 2) SQD_DMA8_CASE(1, 16, 1, 4)
  SQD_DMA8_CASE(1, 16, 2, 4)
#undef SQD_DMA8_CASE
  if (c.dma == 2) return SQD_ERR_UNSUPPORTED;
#define SQD_DMA_CASE(T, K, M, Nn) \
  if (c.dma && c.taps == T && c.kc == K && c.mt == M && c.nt == Nn) return launch_conv_dma<T, K, M, Nn, 4>(a, s);
  SQD_DMA_CASE(9, 16, 1, 1) SQD_DMA_CASE(9, 16, 1, 2) SQD_DMA_CASE(9, 16, 1, 3) SQD_DMA_CASE(9, 16, 1, 4)
  SQD_DMA_CASE(9, 16, 2, 1) SQD_DMA_CASE(9, 16, 2, 2) SQD_DMA_CASE(9, 16, 2, 3) SQD_DMA_CASE(9, 16, 2, 4)
  SQD_DMA_CASE(9, 16, 4, 1) SQD_DMA_CASE(9, 16, 4, 2)
  SQD_DMA_CASE(1, 16, 1, 2) SQD_DMA_CASE(1, 16, 1, 4) SQD_DMA_CASE(1, 16, 2, 4) SQD_DMA_CASE(1, 16, 4, 4)
  SQD_DMA_CASE(1, 32, 1, 1) SQD_DMA_CASE(1, 32, 1, 2) SQD_DMA_CASE(1, 32, 1, 3) SQD_DMA_CASE(1, 32, 1, 4)
  SQD_DMA_CASE(1, 32, 1, 6) SQD_DMA_CASE(1, 32, 2, 1) SQD_DMA_CASE(1, 32, 2, 2) SQD_DMA_CASE(1, 32, 4, 1)
  SQD_DMA_CASE(1, 64, 1, 2) SQD_DMA_CASE(1, 64, 1, 3) SQD_DMA_CASE(1, 64, 1, 4) SQD_DMA_CASE(1, 64, 1, 6)
  SQD_DMA_CASE(1, 64, 2, 1) SQD_DMA_CASE(1, 64, 2, 2)
#undef SQD_DMA_CASE
  if (c.dma) return SQD_ERR_UNSUPPORTED;
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
  SQD_CONV_CASE(9, 16, 1, 1)
  SQD_CONV_CASE(9, 16, 1, 2)
  SQD_CONV_CASE(9, 16, 1, 3)
  SQD_CONV_CASE(9, 16, 4, 1)
  SQD_CONV_CASE(9, 16, 4, 2)
  SQD_CONV_CASE(1, 16, 1, 2)
  SQD_CONV_CASE(1, 16, 2, 2)
  SQD_CONV_CASE(1, 16, 4, 2)
  SQD_CONV_CASE(1, 16, 4, 4)
  SQD_CONV_CASE(1, 32, 1, 1)
  SQD_CONV_CASE(1, 32, 1, 2)
  SQD_CONV_CASE(1, 32, 4, 1)
  SQD_CONV_CASE(1, 32, 4, 2)
  SQD_CONV_CASE(1, 64, 1, 2)
  SQD_CONV_CASE(1, 64, 1, 3)
  SQD_CONV_CASE(1, 64, 1, 4)
  SQD_CONV_CASE(1, 64, 2, 1)
  SQD_CONV_CASE(1, 64, 2, 2)
#undef SQD_CONV_CASE
  return SQD_ERR_UNSUPPORTED;
}

extern "C" int sqd_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                            const float* xmask, const float* ymask, const float* ymul, int B, int H, int W, int C,
                            int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu,
                            int accumulate, int xmask_pitch, int xmask_coff, int ymask_pitch, int ymask_coff,
                            int ymul_pitch, int ymul_coff, int cfg_id, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, xmask, ymask, ymul, B, H, W, C, x_pitch, x_coff, N, Npad, y_pitch, y_coff, relu, accumulate,
                       xmask_pitch, xmask_coff, ymask_pitch, ymask_coff, ymul_pitch, ymul_coff, cfg_id, stream, nullptr, 0, 0.f);
}

// Forward convolution + bias (+ ReLU) + counter-based dropout of the output (sqd_common.h): the last Fire's expand1x1 in training
// mode (reference: Fire.forward + nn.Dropout, src/model/squeezedet.py:18-22,81-82; relu and dropout commute, the scale being > 0).
// drop_state: DEVICE {seed, step} (uint64 x 2); keep16 = round((1 - p) * 65536); scale = 1 / (1 - p).  cfg_id must be a
// weight-stationary 1x1 configuration (sqd_conv_cfg_is_dma >= 3), else SQD_ERR_UNSUPPORTED.
extern "C" int sqd_conv_drop_fwd(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W, int C,
                                 int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu,
                                 const unsigned long long* drop_state, int keep16, float scale, int cfg_id, void* stream) {
  SQD_CHECK_ARG(drop_state && keep16 >= 0 && keep16 <= 65536);
  return conv_fwd_impl(x, w_packed, bias, y, nullptr, nullptr, nullptr, B, H, W, C, x_pitch, x_coff, N, Npad, y_pitch, y_coff, relu, 0,
                       0, 0, 0, 0, 0, 0, cfg_id, stream, drop_state, keep16, scale);
}

// Fused Fire expand: y[..., y_coff : y_coff+E] = ReLU(conv1x1(x)), y[..., y_coff+E : y_coff+2E] = ReLU(conv3x3(x)) in one
// launch (reference: Fire.forward src/model/squeezedet.py:18-22).  w_packed / bias hold the 2E channels in the
// alternating 16-channel-group order described at conv_dma_kernel (built by the host from the two modules' weights:
// group 2i = expand1x1[16i:16i+16] as centre-tap-only 3x3, group 2i+1 = expand3x3[16i:16i+16]), packed with
// sqd_pack_conv_weight for a 3x3 LDS-DMA configuration with an even number of channel groups per slice.
// E must be a multiple of 16 * (NT / 2) ... i.e. 2E a multiple of the slice width; Npad >= 2E rounded up to it.
extern "C" int sqd_fire_expand_fwd(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W,
                                   int C, int x_pitch, int x_coff, int E, int Npad, int y_pitch, int y_coff, int cfg_id,
                                   void* stream) {
  SQD_CHECK_ARG(x && w_packed && y && B > 0 && H > 0 && W > 0 && C > 0 && E > 0);
  const int wg_cap = cfg_id >= 0 ? cfg_id / 1000 : 0;
  if (cfg_id >= 0) cfg_id %= 1000;
  SQD_CHECK_ARG(cfg_id >= 0 && cfg_id < kNumConvCfgs && wg_cap <= 8);
  SQD_CHECK_ARG((C & 3) == 0 && (E & 15) == 0);
  SQD_CHECK_ARG((x_pitch & 3) == 0 && (x_coff & 3) == 0 && (y_pitch & 3) == 0 && (y_coff & 3) == 0);
  SQD_CHECK_ARG(x_coff + C <= x_pitch && y_coff + 2 * E <= y_pitch);
  SQD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)w_packed & 15) == 0);
  SQD_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0);
  const ConvCfg& c = kConvCfgs[cfg_id];
  SQD_CHECK_ARG(c.taps == 9 && c.dma != 0 && (c.nt & 1) == 0);
  SQD_CHECK_ARG((2 * E) % (16 * c.nt) == 0);               // whole slices: every slice carries nt/2 groups of each half
  ConvArgs a = {};
  a.ymask = nullptr; a.ymul = nullptr; a.ymask_pitch = a.ymask_coff = a.ymul_pitch = a.ymul_coff = 0;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.xmask = nullptr;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = 2 * E; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff;
  a.relu = 1; a.accumulate = 0; a.tiles_x = a.tiles_y = 0; a.ntiles = 0; a.nslices = 1; a.gx = 8; a.wg_cap = wg_cap;
  a.fuse_e = E;
  a.xmask_pitch = a.xmask_coff = 0;
  a.total_px = (long long)B * H * W;
  hipStream_t s = (hipStream_t)stream;
#define SQD_FUSE_CASE(M, Nn, Wv) \
  if (c.dma == ((Wv) == 8 ? 2 : 1) && c.kc == 16 && c.mt == M && c.nt == Nn) return launch_conv_dma<9, 16, M, Nn, Wv, true>(a, s);
  SQD_FUSE_CASE(1, 2, 4) SQD_FUSE_CASE(1, 4, 4) SQD_FUSE_CASE(2, 2, 4) SQD_FUSE_CASE(2, 4, 4) SQD_FUSE_CASE(4, 2, 4)
  SQD_FUSE_CASE(1, 2, 8) SQD_FUSE_CASE(1, 4, 8) SQD_FUSE_CASE(2, 2, 8) SQD_FUSE_CASE(2, 4, 8)
#undef SQD_FUSE_CASE
  return SQD_ERR_UNSUPPORTED;
}
