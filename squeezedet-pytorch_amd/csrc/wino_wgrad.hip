// Winograd F(2x2,3x3) weight gradient (see the block comment below); split-K slabs and their reduction are shared with
// the direct kernels of wgrad.hip.  Built without the SI load/store optimiser (Makefile): it would pair the transform's
// ds_read_b32 by address adjacency, not by the (tile row 0, tile row 1) register pairs the packed arithmetic works on.
#include "sqd_common.h"
#include <utility>
#include <type_traits>

extern "C" int sqd_wgrad_reduce_launch(const float* slab, float* dw, float* db, int S, long long slab_stride, int N, int C, int taps,
                                       void* stream);

// ---------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) weight gradient of a 3x3 / pad 1 convolution.
//   dW = sum over 2x2 output tiles of  G^T [ (A dY A^T) .* (B^T d B) ] G
// i.e. per transform position (xi, nu) a GEMM dU[n][c] = sum_tiles dM[tile][n] * V[tile][c] with the TILE axis as K --
// 2.25x fewer multiply-adds than the direct form (conv_wgrad_kernel above).  Same split-K slab scheme and slab layout,
// so the (batched) slab reduction is shared.
// A workgroup (4 waves) owns TN*16 output channels x TC*16 input channels and every S-th 4x16-pixel group (16 tiles);
// wave xi owns the four positions (xi, 0..3).  Per group the dY tile (64 px) and the X patch (6x18 px) arrive by LDS-DMA
// (buffer resources, zero fill by the range check) in a double buffer; both operands are transformed IN REGISTERS straight
// into the MFMA layout: lane (lr, g) holds channel lr of a 16-channel block for the 2x2 tile quad g, so its four tiles
// are the four k-steps of one v_mfma_f32_16x16x4_f32 operand.  Row transforms are wave-uniform two-term combinations
// (coefficients 0 / +-1: exact), column transforms are fixed; everything is written on register pairs so that it
// compiles to packed fp32 instructions.  After the last group dg = G^T dU G: the nu sum inside the wave, the xi sum
// across the waves through LDS in a fixed order (bitwise reproducible); the bias gradient is the tile sum, which IS
// position (1,1) of A dY A^T, accumulated by wave 1 against a constant 1 operand.
// ---------------------------------------------------------------------------------------------
struct WwArgs {
  const float* dy; const float* x; float* slab;
  int B, H, W;
  int N, dy_pitch, dy_coff;
  int C, x_pitch, x_coff;
  int gxn, gyn, ngroups;
  int S, ncg;
  long long slab_stride;
};
typedef __attribute__((address_space(3))) void* lds_ptr_ww_t;

template <int I, int N, class F>
__device__ __forceinline__ void ww_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ww_static_for<I + 1, N>(f); }
}

// Diagnostic build only (-DSQD_WW_STAMP, scratch/diag/ww_stamp.sh; never in libsqdhip.so): s_memtime stamps around the segments of the
// group loop, summed per wave in scalar registers and stored once to a debug buffer of their own (cdna_hip_programming.md, "In-kernel
// stamps"): [workgroup][wave][8] = {wait + barrier, DMA issue, V transform, dM transforms + MFMAs, epilogue, whole kernel, groups, 0}.
#ifdef SQD_WW_STAMP
__device__ unsigned long long* sqd_ww_dbg = nullptr;
extern "C" int sqd_ww_set_debug(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(sqd_ww_dbg), &p, sizeof(p)) == hipSuccess ? SQD_OK : SQD_ERR_LAUNCH;
}
#define WW_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WW_STAMP(t) do { } while (0)
#endif

// Residency: two workgroups per CU.  The stamped timeline (scratch/diag/run_ww_stamp.py, profiles/r04t_ww_stamp.log) shows a wave spending
// ~40 % of a group's period outside its MFMAs: 10-22 % issuing the next group's eight LDS-DMA pieces in one burst behind the barrier
// (~230 cycles per piece when the four waves issue together; 38 cycles per piece when spread over MFMAs in isolation:
// profiles/r04zz_stage_issue_microbench.log), 7-11 % in the V transform, 7-10 % at the wait + barrier; the epilogue is 8-16 % of a wave's
// time where a workgroup only sees ~10 groups.  A THIRD workgroup per CU for the 64 x 16 form (<= 168 registers, 48 KB of LDS: it fits)
// was measured neutral (profiles/r04u_ww_occupancy3.log); an eight-wave form (two positions per wave, half the accumulators, FOUR waves per
// SIMD at 126 registers for the 64 x 16 tile) was 5-8 % SLOWER than this kernel and spills for the wider tiles
// (profiles/r04_wino_wgrad_eight_waves.log; the code is in the history at commit 'Final evidence set r04z'): more waves do not help.
// The body serves two launch forms: one layer per launch (wino_wgrad_kernel) and several layers of one backward stage in ONE launch
// (wino_wgrad_group_kernel below).  ``pos`` is the workgroup's position among the layer's ``nwg`` workgroups (XCD-contiguous order).
template <int TN, int TC>
__device__ __forceinline__ void ww_body(const WwArgs& a, const int pos, const int nwg) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(TN >= 4 && (TC == 1 || TC == 2), "the bank swizzle swaps 16-channel blocks pairwise inside the first four");
  constexpr int CHD = TN * 16, CHX = TC * 16;
  constexpr int DQ = TN * 4, XQ = TC * 4;                  // 16-byte slots per pixel
  constexpr int DSLOTS = 64 * DQ, D_IT = DSLOTS / 256;
  constexpr int XREAL = 108 * XQ, X_IT = (XREAL + 255) / 256, XSLOTS = X_IT * 256;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const dyB = smem;                                  // [2][DSLOTS][4]: dY tile, pixel-major [64 px][CHD]
  float* const xB = smem + 2 * DSLOTS * 4;                  // [2][XSLOTS][4]: X patch [108 px][CHX]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int xi = __builtin_amdgcn_readfirstlane(wv);
  // Which (split, channel-block pair) this workgroup owns.  All block pairs of ONE split read the same pixel groups (the dY tile is
  // re-read by every input-channel block, the X patch by every output-channel block), so they should share an XCD's L2: positions are
  // split-major and the workgroups of an XCD own a contiguous run of positions (sqd_xcd_contiguous) -- each pixel group then
  // crosses the fabric into one L2 instead of into several, and the re-reads come back at L2-hit latency.
  const int ntile = nwg / a.S;
  const int s = pos / ntile, bg = pos - s * ntile;
  const int ng = bg / a.ncg, cg = bg - ng * a.ncg;
  const int n0 = ng * CHD, c0 = cg * CHX;

  int d_offB[D_IT], d_key[D_IT];
#pragma unroll
  for (int it = 0; it < D_IT; ++it) {
    const int slot = it * 256 + tid;
    const int px = slot / DQ, chs = slot - px * DQ;
    const int row = px >> 4, col = px & 15;
    // bank swizzle: a lane quad g reads pixel columns 4g .. 4g + 3, and four columns are a multiple of 32 words apart, so the two
    // quads of a 32-lane LDS group would hit the same 16 banks (2-way conflict on every ds_read_b32 of the transform).  Pixels of odd
    // column quads therefore hold their first four 16-channel blocks pairwise swapped (slot chs holds channel quad chs ^ 4)
    const int chq = (chs < 16) ? (chs ^ (((col >> 2) & 1) << 2)) : chs;
    d_key[it] = (n0 + 4 * chq < a.N) ? (row << 8 | col) : -1;        // channels past N (partial last block) stay zero
    d_offB[it] = ((row * a.W + col) * a.dy_pitch + 4 * chq) * 4;
  }
  int x_offB[X_IT], x_key[X_IT];
#pragma unroll
  for (int it = 0; it < X_IT; ++it) {
    const int slot = it * 256 + tid;
    const int px = slot / XQ, chs = slot - px * XQ;
    const int r = px / 18, c = px - r * 18;
    const int chq = (TC == 2) ? (chs ^ (((c >> 2) & 1) << 2)) : chs;      // (the same swizzle on the patch: its two 16-channel blocks)
    const bool real = px < 108 && c0 + 4 * chq < a.C;       // channels past C (partial last block) stay zero
    x_key[it] = real ? (r << 8 | c) : -1;
    x_offB[it] = real ? ((r * a.W + c) * a.x_pitch + 4 * chq) * 4 : 0;
  }
  // interior fast path: X offsets with the tile-independent invalid slots (patch padding, channels past C) already out of
  // range; it applies when the dY channel block is full as well
  int x_offI[X_IT];
#pragma unroll
  for (int it = 0; it < X_IT; ++it) x_offI[it] = x_key[it] >= 0 ? x_offB[it] : (int)OOB;
  const bool full_blocks = n0 + CHD <= a.N;
  const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + a.dy_coff + n0), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff + c0 - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const int wv_s = xi;

  // DMA of a group into a buffer, one 1 KB piece (a wave instruction) at a time: pieces 0 .. D_IT - 1 are the dY tile, D_IT .. NPIECE - 1
  // the X patch.  Everything but the lane offsets is wave-uniform.  The pieces of the NEXT group are issued between the MFMA blocks of
  // the current one (the other buffer is free from the barrier on): issued as one burst behind the barrier they cost the wave ~230
  // cycles each (profiles/r04t_ww_stamp.log), spread over matrix work ~40 (profiles/r04zz_stage_issue_microbench.log).
  constexpr int NPIECE = D_IT + X_IT;
  struct GroupDma { int y0, x0; unsigned soffD, soffX; bool inner; };
  auto group_dma = [&](int q) {
    GroupDma gd;
    const int gxi = q % a.gxn; int t = q / a.gxn;
    const int gyi = t % a.gyn; const int b = t / a.gyn;
    gd.y0 = gyi * 4; gd.x0 = gxi * 16;
    const long long p0 = ((long long)b * a.H + gd.y0) * a.W + gd.x0;
    gd.soffD = (unsigned)(p0 * a.dy_pitch * 4); gd.soffX = (unsigned)(p0 * a.x_pitch * 4);
    // interior groups (the whole 6x18 patch inside the image) and full channel blocks: the precomputed offsets as they
    // are -- no per-lane validity arithmetic (uniform branch)
    gd.inner = gd.y0 >= 1 && gd.y0 + 5 <= a.H && gd.x0 >= 1 && gd.x0 + 17 <= a.W && full_blocks;
    return gd;
  };
  auto issue_piece = [&](const GroupDma& gd, int buf, auto piece_c) {
    constexpr int PIECE = decltype(piece_c)::value;
    if constexpr (PIECE < D_IT) {
      constexpr int it = PIECE;
      int off = d_offB[it];
      if (!gd.inner) {
        const int key = d_key[it];
        const bool ok = key >= 0 && gd.y0 + (key >> 8) < a.H && gd.x0 + (key & 255) < a.W;
        off = ok ? off : (int)OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dres, (lds_ptr_ww_t)(dyB + (buf * DSLOTS + it * 256 + wv_s * 64) * 4), 16, off, (int)gd.soffD, 0, 0);
    } else if constexpr (PIECE < NPIECE) {
      constexpr int it = PIECE - D_IT;
      int off = x_offI[it];
      if (!gd.inner) {
        const int key = x_key[it];
        const bool ok = key >= 0 && (unsigned)(gd.y0 + (key >> 8) - 1) < (unsigned)a.H && (unsigned)(gd.x0 + (key & 255) - 1) < (unsigned)a.W;
        off = ok ? x_offB[it] : (int)OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_ww_t)(xB + (buf * XSLOTS + it * 256 + wv_s * 64) * 4), 16, off, (int)gd.soffX, 0, 0);
    }
  };
  auto issue_all = [&](const GroupDma& gd, int buf) {        // (the prologue: the first group of the workgroup)
    ww_static_for<0, NPIECE>([&](auto pc) { issue_piece(gd, buf, pc); });
  };

  // Wave-uniform row transforms, each ONE packed fma per register pair (every VALU instruction stalls the MFMA stream):
  //   V row xi of B^T d B   = sgx * (d[i1] + sx * d[i2]):  xi0: d0 - d2, xi1: d1 + d2, xi2: -(d1 - d2), xi3: d1 - d3
  //   dM row xi of A dY A^T = sgd * (y[r0] + sd * y[1]):   xi0: y0 (sd = 0), xi1: y0 + y1, xi2: y0 - y1, xi3: -(y1 + 0 * y1)
  // The overall signs sgx * sgd multiply the whole dU row: applied once, in the epilogue.  Column nu = 3 of dM is -r1:
  // that sign moves into V's column 3 (t3 - t1 instead of t1 - t3).
  const int i1 = (xi == 0) ? 0 : 1, i2 = (xi == 3) ? 3 : 2;
  const float sx = (xi == 1) ? 1.f : -1.f;
  const float sd = (xi == 0 || xi == 3) ? 0.f : ((xi == 1) ? 1.f : -1.f);
  const int r0 = (xi == 3) ? 1 : 0;                           // xi = 3: "y0" is read from row 1 as well (r = y1 + 0 * y1)
  const float row_sign = ((xi == 2) ? -1.f : 1.f) * ((xi == 3) ? -1.f : 1.f);
  const f32x2 sx2 = {sx, sx}, sd2 = {sd, sd};
  // lane bases (floats): dY element (px, ch) at px*CHD + ch, X element at px*CHX + ch; the quad's 4 pixel columns 4g..
  const int dL0 = (r0 * 16 + 4 * g) * CHD + lr, dL1 = (1 * 16 + 4 * g) * CHD + lr;
  const int xL1 = (i1 * 18 + 4 * g) * CHX + lr, xL2 = (i2 * 18 + 4 * g) * CHX + lr;
  const int gsw = (g & 1) * 16;                               // the lane quad's swizzle (floats): block b of its pixels sits at (16 b) ^ gsw

  f32x4 acc[4][TN][TC], accb[TN];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) acc[nu][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TN; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int buf = 0;
#ifdef SQD_WW_STAMP
  unsigned long long ta = 0, tb = 0, tc = 0, td = 0, te = 0, t_begin = 0, sum_wait = 0, sum_issue = 0, sum_v = 0, sum_mm = 0, ngr = 0;
  WW_STAMP(t_begin);
#endif
  issue_all(group_dma(s), 0);
  for (int q = s; q < a.ngroups; q += a.S) {
    WW_STAMP(ta);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's share of the group's DMA has landed ...
    __syncthreads();                                         // ... and is published; the other buffer is free again
    WW_STAMP(tb);
    const bool more = q + a.S < a.ngroups;
    const GroupDma nxt = group_dma(more ? q + a.S : q);
    WW_STAMP(tc);
    const float* const dP0 = dyB + buf * DSLOTS * 4 + dL0;
    const float* const dP1 = dyB + buf * DSLOTS * 4 + dL1;
    const float* const xP1 = xB + buf * XSLOTS * 4 + xL1;
    const float* const xP2 = xB + buf * XSLOTS * 4 + xL2;

    // ---- V = B^T d B, row xi, for every input-channel block: bfr[cb][nu] = 4 tiles (k-steps) of channel lr ----
    // tile member t = 2*txl + ty (tile column 2g+txl, tile row ty): the (ty=0, ty=1) pair sits in adjacent registers
    f32x4 bfr[TC][4];
#pragma unroll
    for (int cbk = 0; cbk < TC; ++cbk) {
      f32x2 tt[6];                                           // row-transformed patch columns 0..5 of the quad, pair over ty
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        // (patch column 4g + jj: columns 4, 5 of the quad belong to the next column quad)
        const int xo = (TC == 2) ? ((cbk * 16) ^ gsw ^ ((jj >> 2) << 4)) : cbk * 16;
        const f32x2 d1 = {xP1[(0 * 18 + jj) * CHX + xo], xP1[(2 * 18 + jj) * CHX + xo]};
        const f32x2 d2 = {xP2[(0 * 18 + jj) * CHX + xo], xP2[(2 * 18 + jj) * CHX + xo]};
        tt[jj] = __builtin_elementwise_fma(sx2, d2, d1);
      }
#pragma unroll
      for (int txl = 0; txl < 2; ++txl) {
        const int j0 = 2 * txl;
        const f32x2 v0 = tt[j0] - tt[j0 + 2], v1 = tt[j0 + 1] + tt[j0 + 2], v2 = tt[j0 + 2] - tt[j0 + 1], v3 = tt[j0 + 3] - tt[j0 + 1];   // (sign of dM column 3)
        if (txl == 0) { bfr[cbk][0].lo = v0; bfr[cbk][1].lo = v1; bfr[cbk][2].lo = v2; bfr[cbk][3].lo = v3; }
        else          { bfr[cbk][0].hi = v0; bfr[cbk][1].hi = v1; bfr[cbk][2].hi = v2; bfr[cbk][3].hi = v3; }
      }
    }
    WW_STAMP(td);
    // ---- per output-channel block: dM = A dY A^T, row xi, then the MFMAs over (nu, cb, k-step) ----
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      f32x4 afr[4];
#pragma unroll
      for (int txl = 0; txl < 2; ++txl) {
        f32x2 rp[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int yo = (nb < 4) ? ((nb * 16) ^ gsw) : nb * 16;
          const f32x2 y0 = {dP0[(2 * txl + j) * CHD + yo], dP0[(2 * 16 + 2 * txl + j) * CHD + yo]};
          const f32x2 y1 = {dP1[(2 * txl + j) * CHD + yo], dP1[(2 * 16 + 2 * txl + j) * CHD + yo]};
          rp[j] = __builtin_elementwise_fma(sd2, y1, y0);
        }
        const f32x2 m0 = rp[0], m1 = rp[0] + rp[1], m2 = rp[0] - rp[1], m3 = rp[1];
        if (txl == 0) { afr[0].lo = m0; afr[1].lo = m1; afr[2].lo = m2; afr[3].lo = m3; }
        else          { afr[0].hi = m0; afr[1].hi = m1; afr[2].hi = m2; afr[3].hi = m3; }
      }
      // the next group's DMA pieces ride between the MFMA blocks: piece p in front of block floor(p * 4 TN / NPIECE) of the 4 TN blocks
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        // (nb, nu are fully unrolled: the comparison folds to a constant per piece)
        ww_static_for<0, NPIECE>([&](auto pc) {
          constexpr int PP = decltype(pc)::value;
          if ((PP * 4 * TN) / NPIECE == nb * 4 + nu && more) issue_piece(nxt, buf ^ 1, pc);
        });
#pragma unroll
        for (int cbk = 0; cbk < TC; ++cbk)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[nu][nb][cbk] = mfma16(afr[nu][t], bfr[cbk][nu][t], acc[nu][nb][cbk]);
      }
      if (xi == 1) {                                         // bias gradient: position (1,1) of A dY A^T is the tile sum
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[nb] = mfma16(afr[1][t], 1.0f, accb[nb]);
      }
    }
    WW_STAMP(te);
#ifdef SQD_WW_STAMP
    sum_wait += tb - ta; sum_issue += tc - tb; sum_v += td - tc; sum_mm += te - td; ngr += 1;
#endif
    buf ^= 1;
  }
#ifdef SQD_WW_STAMP
  unsigned long long t_loop_end = 0;
  WW_STAMP(t_loop_end);
#endif

  // ---- dg = G^T dU G: nu sum in registers, xi sum across the waves through LDS (fixed order) ----
  float* const sl = a.slab + (long long)s * a.slab_stride;
  f32x4* const wL = (f32x4*)smem;                            // [4 waves][3][TC][64 lanes] (+ bias row)
  f32x4* const bL = wL + 4 * 3 * TC * 64;                    // [64 lanes]
  const long long nw = (long long)a.N * 9 * a.C;
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    __syncthreads();
#pragma unroll
    for (int cbk = 0; cbk < TC; ++cbk) {
      const f32x4 u0 = row_sign * acc[0][nb][cbk], u1 = row_sign * acc[1][nb][cbk], u2 = row_sign * acc[2][nb][cbk], u3 = row_sign * acc[3][nb][cbk];
      const f32x4 h = 0.5f * (u1 + u2);
      wL[((xi * 3 + 0) * TC + cbk) * 64 + lane] = u0 + h;
      wL[((xi * 3 + 1) * TC + cbk) * 64 + lane] = 0.5f * (u1 - u2);
      wL[((xi * 3 + 2) * TC + cbk) * 64 + lane] = h + u3;
    }
    if (xi == 1) bL[lane] = accb[nb];
    __syncthreads();
    if (xi < 3) {                                            // wave r = xi finishes kernel row r
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int cbk = 0; cbk < TC; ++cbk) {
          const f32x4 w0 = wL[((0 * 3 + s3) * TC + cbk) * 64 + lane], w1 = wL[((1 * 3 + s3) * TC + cbk) * 64 + lane];
          const f32x4 w2 = wL[((2 * 3 + s3) * TC + cbk) * 64 + lane], w3 = wL[((3 * 3 + s3) * TC + cbk) * 64 + lane];
          f32x4 v;
          if (xi == 0) v = w0 + 0.5f * (w1 + w2);
          else if (xi == 1) v = 0.5f * (w1 - w2);
          else v = 0.5f * (w1 + w2) + w3;
          const int c = c0 + cbk * 16 + lr, n = n0 + nb * 16 + 4 * g;
          if (c < a.C && n < a.N) {                           // (N, C multiples of 4: a quad of channels is in or out as a whole)
            const int tap = xi * 3 + s3;
#pragma unroll
            for (int i = 0; i < 4; ++i) sl[((long long)(n + i) * 9 + tap) * a.C + c] = v[i];
          }
        }
    } else if (cg == 0 && lr == 0 && n0 + nb * 16 + 4 * g < a.N) {   // wave 3: bias gradient (every column of the 1-operand product is the sum)
      const f32x4 v = bL[lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) sl[nw + n0 + nb * 16 + 4 * g + i] = v[i];
    }
  }
#ifdef SQD_WW_STAMP
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t_end = 0;
    WW_STAMP(t_end);
    if (sqd_ww_dbg && lane == 0) {
      unsigned long long* d = sqd_ww_dbg + ((long long)blockIdx.x * 4 + xi) * 8;
      d[0] = sum_wait; d[1] = sum_issue; d[2] = sum_v; d[3] = sum_mm; d[4] = t_end - t_loop_end; d[5] = t_end - t_begin; d[6] = ngr; d[7] = t_begin;
    }
  }
#endif
#endif
}

template <int TN, int TC>
__global__ __launch_bounds__(256, 2) void wino_wgrad_kernel(WwArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  ww_body<TN, TC>(a, sqd_xcd_contiguous((int)blockIdx.x, (int)gridDim.x), (int)gridDim.x);
#endif
}

// Several layers of one backward stage (same B, H, W and tile form) in ONE launch: every layer gets the same number of splits S, so all
// workgroups do the same work, and S is what fills the chip once over ALL the layers -- a fraction of what each layer would need alone.
// A workgroup's K loop gets that much longer (the epilogue -- LDS exchange + 36-73 KB of slab stores -- is 8-16 % of a wave's time at ~10
// groups per workgroup, profiles/r04t_ww_stamp.log) and the slab bytes shrink by the same factor.  Layers are contiguous runs of
// positions, so an XCD still works on one or two layers' pixel groups at a time.
#define WW_MAX_GROUP 6
struct WwGroupArgs { WwArgs l[WW_MAX_GROUP]; int wg0[WW_MAX_GROUP + 1]; int n; };

template <int TN, int TC>
__global__ __launch_bounds__(256, 2) void wino_wgrad_group_kernel(WwGroupArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int pos = sqd_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  int i = 0;
#pragma unroll
  for (int k = 1; k < WW_MAX_GROUP; ++k) i = (k < ga.n && pos >= ga.wg0[k]) ? k : i;
  i = __builtin_amdgcn_readfirstlane(i);
  const WwArgs a = ga.l[i];
  ww_body<TN, TC>(a, pos - ga.wg0[i], ga.wg0[i + 1] - ga.wg0[i]);
#endif
}

template <int TN, int TC>
static int launch_wino_wgrad_group(WwGroupArgs& ga, int S, hipStream_t stream) {
  constexpr int DSLOTS = 64 * TN * 4, XSLOTS = (108 * TC * 4 + 255) / 256 * 256;
  constexpr size_t lds = (size_t)2 * (DSLOTS + XSLOTS) * 16;
  auto kern = wino_wgrad_group_kernel<TN, TC>;
  static SqdDevOnce once;
  if (lds > 64 * 1024 && sqd_max_lds_once(once, (const void*)kern, (int)lds) != SQD_OK) return SQD_ERR_LAUNCH;
  int wg = 0;
  for (int i = 0; i < ga.n; ++i) {
    WwArgs& a = ga.l[i];
    a.S = S; a.ncg = sqd_cdiv(a.C, TC * 16);
    ga.wg0[i] = wg;
    wg += sqd_cdiv(a.N, TN * 16) * a.ncg * S;
  }
  for (int i = ga.n; i <= WW_MAX_GROUP; ++i) ga.wg0[i] = wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)wg), dim3(256), lds, stream, ga);
  return sqd_launch_status();
}

template <int TN, int TC>
static int launch_wino_wgrad(WwArgs a, hipStream_t stream) {
  constexpr int DSLOTS = 64 * TN * 4, XSLOTS = (108 * TC * 4 + 255) / 256 * 256;
  constexpr size_t lds = (size_t)2 * (DSLOTS + XSLOTS) * 16;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  static_assert((size_t)(4 * 3 * TC + 1) * 64 * 16 <= lds, "epilogue exchange fits the staging buffers");
  auto kern = wino_wgrad_kernel<TN, TC>;
  static SqdDevOnce once;
  if (lds > 64 * 1024 && sqd_max_lds_once(once, (const void*)kern, (int)lds) != SQD_OK) return SQD_ERR_LAUNCH;
  a.ncg = sqd_cdiv(a.C, TC * 16);
  const int groups = sqd_cdiv(a.N, TN * 16) * a.ncg;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * a.S)), dim3(256), lds, stream, a);
  return sqd_launch_status();
}

// Winograd form of sqd_conv_wgrad for 3x3 layers (same arguments, slab layout and dw == NULL convention); supported:
// N % 64 == 0 or N <= 80, C % 4 == 0, S <= number of 4x16-pixel groups.  Returns SQD_ERR_UNSUPPORTED otherwise.
extern "C" int sqd_conv_wgrad_wino(const float* dy, const float* x, float* slab, float* dw, float* db, int B, int H, int W,
                                   int N, int dy_pitch, int dy_coff, int C, int x_pitch, int x_coff, int S, int tc, void* stream) {
  SQD_CHECK_ARG(dy && x && slab && B > 0 && H > 0 && W > 0 && N > 0 && C > 0 && S > 0 && S <= 65535 && (tc == 1 || tc == 2));
  SQD_CHECK_ARG((N & 3) == 0 && (C & 3) == 0 && (dy_pitch & 3) == 0 && (dy_coff & 3) == 0 && (x_pitch & 3) == 0 && (x_coff & 3) == 0);
  SQD_CHECK_ARG(dy_coff + N <= dy_pitch && x_coff + C <= x_pitch);
  SQD_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0);
  WwArgs a;
  a.dy = dy; a.x = x; a.slab = slab; a.B = B; a.H = H; a.W = W;
  a.N = N; a.dy_pitch = dy_pitch; a.dy_coff = dy_coff; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.gxn = sqd_cdiv(W, 16); a.gyn = sqd_cdiv(H, 4); a.ngroups = B * a.gxn * a.gyn;
  a.S = S; a.slab_stride = (long long)N * 9 * C + N;
  if ((N % 64 && N > 80) || S > a.ngroups) return SQD_ERR_UNSUPPORTED;
  const long long px = (long long)B * H * W;
  if (px * dy_pitch * 4 >= (3ll << 30) || px * x_pitch * 4 >= (3ll << 30)) return SQD_ERR_UNSUPPORTED;   // 32-bit SGPR byte offsets
  hipStream_t s = (hipStream_t)stream;
  // N <= 80 (ConvDet: 72): one 5-block output-channel group x 16 input channels (tc is ignored: the 32-channel form spilled and made the
  // slab reduction slower by more than it gained, profiles/r04*); else 64-channel groups x 16 or 32 input channels
  const int rc = (N % 64) ? launch_wino_wgrad<5, 1>(a, s)
                          : ((tc <= 1 || C <= 16) ? launch_wino_wgrad<4, 1>(a, s) : launch_wino_wgrad<4, 2>(a, s));
  if (rc != SQD_OK || !dw) return rc;
  return sqd_wgrad_reduce_launch(slab, dw, db, S, a.slab_stride, N, C, 9, stream);
}


// Winograd weight-gradient slabs of up to WW_MAX_GROUP 3x3 layers that share B, H, W (one backward stage) in ONE launch; every layer is
// cut into the same S splits.  ``layers``: n records of 9 64-bit words {dy, x, slab, N, dy_pitch, dy_coff, C, x_pitch, x_coff} (pointers
// and element counts; host memory).  All layers must select the same tile form: N % 64 == 0 for all of them, and C < 32 or C % 32 == 16
// for all (tc = 1) or for none (tc = 2).  Slabs only (the caller reduces them: sqd_wgrad_reduce_batched).
extern "C" int sqd_conv_wgrad_wino_group(const long long* layers, int n, int B, int H, int W, int S, int tc, void* stream) {
  SQD_CHECK_ARG(layers && n >= 1 && n <= WW_MAX_GROUP && B > 0 && H > 0 && W > 0 && S > 0 && S <= 65535 && (tc == 1 || tc == 2));
  WwGroupArgs ga;
  ga.n = n;
  const long long px = (long long)B * H * W;
  for (int i = 0; i < n; ++i) {
    const long long* r = layers + 9 * i;
    WwArgs& a = ga.l[i];
    a.dy = (const float*)(uintptr_t)r[0]; a.x = (const float*)(uintptr_t)r[1]; a.slab = (float*)(uintptr_t)r[2];
    a.N = (int)r[3]; a.dy_pitch = (int)r[4]; a.dy_coff = (int)r[5]; a.C = (int)r[6]; a.x_pitch = (int)r[7]; a.x_coff = (int)r[8];
    a.B = B; a.H = H; a.W = W;
    SQD_CHECK_ARG(a.dy && a.x && a.slab && a.N > 0 && a.C > 0);
    SQD_CHECK_ARG((a.N & 3) == 0 && (a.C & 3) == 0 && (a.dy_pitch & 3) == 0 && (a.dy_coff & 3) == 0 && (a.x_pitch & 3) == 0 && (a.x_coff & 3) == 0);
    SQD_CHECK_ARG(a.dy_coff + a.N <= a.dy_pitch && a.x_coff + a.C <= a.x_pitch);
    SQD_CHECK_ARG(((uintptr_t)a.dy & 15) == 0 && ((uintptr_t)a.x & 15) == 0);
    a.gxn = sqd_cdiv(W, 16); a.gyn = sqd_cdiv(H, 4); a.ngroups = B * a.gxn * a.gyn;
    a.slab_stride = (long long)a.N * 9 * a.C + a.N;
    if (a.N % 64 || S > a.ngroups) return SQD_ERR_UNSUPPORTED;
    if ((tc == 2) != !(a.C < 32 || a.C % 32 == 16)) return SQD_ERR_UNSUPPORTED;      // every layer must want this tile form
    if (px * a.dy_pitch * 4 >= (3ll << 30) || px * a.x_pitch * 4 >= (3ll << 30)) return SQD_ERR_UNSUPPORTED;
  }
  for (int i = n; i < WW_MAX_GROUP; ++i) ga.l[i] = ga.l[0];
  hipStream_t s = (hipStream_t)stream;
  return tc == 1 ? launch_wino_wgrad_group<4, 1>(ga, S, s) : launch_wino_wgrad_group<4, 2>(ga, S, s);
}
