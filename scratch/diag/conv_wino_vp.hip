// EXPERIMENT (round 5, not part of libsqdhip.so): the V-shared Winograd kernel of csrc/conv_wino_vs.hip in a PERSISTENT form for wide
// outputs with short reductions -- Fire expand3x3 C96 -> N384 (fire13 / fire14, src/model/squeezedet.py:14,20-22) -- as VERDICT round 4
// item 2 asked ("same form for C96 -> N384").  Correct (bit-identical to conv_wino_kernel<2,4>: scratch/vp_check.py) and measured:
// 131 us isolated against 135-141 for <2,4>, but 131 against 117 us INSIDE the bs=20 step (profiles/r05_conv_wino_vp.log), so no table
// row selects it and it is not shipped.  Ablation in the same log: of 137 us (back-to-back loop) the weight loads straight from the L2
// cost 25 (every wave fetches its block's 8 KB per stage: 4x the L2 -> CU bytes per MFMA of the LDS-shared slices of <2,4>), the stage
// barrier 22, the input transform 13 (split over four waves, one per SIMD), the set epilogue's stores 12; 92 without all of them.
// build: scratch/diag/vp_diag.sh -> scratch/libvpdiag.so (sqd_vp_diag_<mask>; mask 0 = the full kernel)
#include "sqd_common.h"
#include <type_traits>
#ifndef SQD_VP_DIAG
#define SQD_VP_DIAG 0             /* ablation builds of conv_wino_vp_kernel: bit 0 = no input transform, 1 = no epilogue stores, 2 = no epilogue
                                     at all (but the last set's), 3 = no stage barrier, 4 = U operands not loaded in the loop */
#endif
#ifndef SQD_VP_ENTRY
#define SQD_VP_ENTRY sqd_conv_wino_vp_fwd
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_vs_t;
constexpr int VS_WV = 12;                     // waves per workgroup = units per workgroup
constexpr int VS_VFL = 8 * 256;               // floats of one chunk's V of a group: [8 position pairs][64 lanes][4]
constexpr int VS_RFL = 256 * 4;               // floats of a raw patch image (226 of 256 16-byte slots used)

// ---------------------------------------------------------------------------------------------------------------------
// The same idea for WIDE outputs and short reductions -- Fire expand3x3 C96 -> N384 (fire13 / fire14, src/model/squeezedet.py:14,20-22):
// N is a multiple of 192, so a workgroup's twelve waves are twelve consecutive 16-channel blocks of ONE group (no group is cut), the
// transform of a chunk is done once per twelve blocks (conv_wino_kernel<2,4>: once per two), and the workgroup is PERSISTENT: it walks
// the unit sets s = wgpos, wgpos + nwg, ... (set = (group, twelve-block part)) as ONE stream of stages t = (set, chunk) -- the duty of
// stage t goes to wave t mod 12, which requests the patch of stage t + 12 (the next set's group: its per-lane offsets are computed once
// per set) right behind its transform, the U loads run one stage ahead across the seam (a wave keeps its block, so the weight addresses
// are periodic in the chunk), and a set's epilogue (inverse transform, bias, ReLU, four 16-byte stores) sits between its last stage and
// the next set's first.  C / 8 >= 12 chunks (the patch request distance of twelve stages then reaches at most into the next set).
// ---------------------------------------------------------------------------------------------------------------------
struct WinoVpArgs {
  const float* x; const float* u; const float* bias; float* y;
  int B, H, W;
  int C, x_pitch, x_coff;
  int N, Npad, y_pitch, y_coff;
  int relu;
  int gxn, gyn, ngroups, parts, nsets, nwg;      // parts = N / 192 twelve-block parts per group; nsets = ngroups * parts
};

__global__ __launch_bounds__(VS_WV * 64, 3) void conv_wino_vp_kernel(WinoVpArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RP = 113;
  constexpr int RAW_IT = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const VB = smem;                                   // [2][VS_VFL]
  float* const RB = VB + 2 * VS_VFL;                        // [VS_WV][VS_RFL]

  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunks = a.C >> 3;
  const int wgpos = sqd_xcd_contiguous((int)blockIdx.x, a.nwg);
  if (wgpos >= a.nsets) return;
  const int nmine = (a.nsets - wgpos + a.nwg - 1) / a.nwg;            // unit sets of this workgroup
  const int nstages = nmine * nchunks;
  const int nblk16 = a.Npad >> 4;                                      // 16-channel blocks of the packed weights

  // group geometry of a set (wave-uniform)
  struct GPos { int y0, x0; long long p0; unsigned soff; int blk; };
  auto set_pos = [&](int k) {
    GPos gp;
    const int s = wgpos + k * a.nwg;
    const int grp = s / a.parts, part = s - grp * a.parts;
    const int q1 = grp / a.gxn, gxi = grp - q1 * a.gxn;
    const int b = q1 / a.gyn, gyi = q1 - b * a.gyn;
    gp.y0 = gyi * 4; gp.x0 = gxi * 16;
    gp.p0 = ((long long)b * a.H + gp.y0) * a.W + gp.x0;
    gp.soff = (unsigned)(gp.p0 * a.x_pitch * 4);
    gp.blk = part * VS_WV + wv;
    return gp;
  };
  constexpr unsigned OOB = 0x80000000u;
  // per-lane patch slots: a set-independent byte offset per DMA instruction and, per set, ONE register of validity bits (bit `it` = the
  // slot of instruction `it` lies inside the image); an invalid slot is requested beyond the resource's range and arrives as zeros
  auto set_mask = [&](const GPos gp) {
    int m = 0;
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
      const int s = it * 64 + lane;
      const int kq = s / RP, pix = s - kq * RP;
      const int r = pix / 18, c = pix - r * 18;
      const bool ok = kq < 2 && pix < 108 && (unsigned)(gp.y0 + r - 1) < (unsigned)a.H && (unsigned)(gp.x0 + c - 1) < (unsigned)a.W;
      m |= ok ? (1 << it) : 0;
    }
    return m;
  };
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 0x7ffffff0, 0x00020000);
  float* const rawS = RB + wv * VS_RFL;
  // patch of (validity bits `mask`, origin `soff`, chunk cc) -> this wave's raw image
  int r_base[RAW_IT];                                       // byte offsets of this lane's four patch slots (set-independent)
#pragma unroll
  for (int it = 0; it < RAW_IT; ++it) {
    const int s = it * 64 + lane;
    const int kq = s / RP, pix = s - kq * RP;
    const int r = pix / 18, c = pix - r * 18;
    r_base[it] = ((r * a.W + c) * a.x_pitch + 4 * kq) * 4;
  }
  auto dma_raw = [&](int mask, unsigned soff, int cc) {
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_vs_t)(rawS + it * 64 * 4), 16, ((mask >> it) & 1) ? r_base[it] : (int)OOB,
                                               (int)(soff + (unsigned)cc * 32u), 0, 0);
  };
  const float* const rawL = rawS + (((g >> 1) * RP + (2 * (lr >> 3)) * 18 + 2 * (lr & 7)) * 4 + 2 * (g & 1));

  GPos cur = set_pos(0), nxt = set_pos(nmine > 1 ? 1 : 0);
  int off_cur = set_mask(cur), off_nxt = set_mask(nxt);
  // a wave keeps its block within a part; the part (and with it the block) may change from set to set
  auto u_voff_of = [&](int blk) { return (blk * 256 + lane * 4) * 4; };
  const unsigned u_chunkB = (unsigned)(8 * nblk16 * 256 * 4);
  auto load_u = [&](int voff, int cc, int pp) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ures, voff + pp * (nblk16 * 1024), (int)((unsigned)cc * u_chunkB), 0));
  };

  f32x4 acc[16];
  auto bias_of = [&](int blk) {
    const int n = blk * 16 + 4 * g;
    return (a.bias && n < a.N) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = z4;
  acc[5] = bias_of(cur.blk);                                // position (1,1) enters all four outputs with weight +1

  // The input transform of a stage is split over FOUR waves, one per SIMD: waves 4 j .. 4 j + 3 (j = stage mod 3) each hold their own copy
  // of the stage's patch and compute ONE row of B^T d B (8 LDS reads, 8 packed adds, two 16-byte V stores).  One wave doing all of it
  // made its SIMD the straggler of every stage barrier (ablation: 18 us of 137 for the transform, 33 for the barrier).
  const int trow = wv & 3, tgrp = wv >> 2;
  auto transform = [&](int vbuf) {
    if (SQD_VP_DIAG & 1) return;
    f32x4* const vdst = (f32x4*)(VB + vbuf * VS_VFL) + lane;
    // row i of t = column transform of patch rows (ra, rb): t[0] = d0 - d2, t[1] = d1 + d2, t[2] = d2 - d1, t[3] = d1 - d3
    const int ra = trow == 0 ? 0 : (trow == 2 ? 2 : 1), rb = trow == 0 ? 2 : (trow == 1 ? 2 : (trow == 2 ? 1 : 3));
    const float* const pa = rawL + ra * 72, * const pb = rawL + rb * 72;
    f32x2 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 da = *(const f32x2*)(pa + j * 4), db = *(const f32x2*)(pb + j * 4);
      t[j] = trow == 1 ? da + db : da - db;
    }
    const f32x2 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
    f32x4 w0, w1;
    w0.lo = v0; w0.hi = v1; w1.lo = v2; w1.hi = v3;
    vdst[(2 * trow) * 64] = w0;
    vdst[(2 * trow + 1) * 64] = w1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // the patch request that follows this wave's share of the transform of stage (.., chunk c): stage + 3, i.e. chunk c + 3 of the current
  // set or chunk c + 3 - nchunks of the next (cur / nxt are those of the stage that was just transformed)
  auto request_after = [&](int t, int c, int mask0, unsigned soff0, int mask1, unsigned soff1) {
    if (t + 3 >= nstages) return;
    if (c + 3 < nchunks) dma_raw(mask0, soff0, c + 3);
    else dma_raw(mask1, soff1, c + 3 - nchunks);
  };

  // ---- prologue: this wave's first patch (stage tgrp: set 0, chunk tgrp), chunk 0's U, T(0) by waves 0..3 ----
  if (tgrp < nstages) dma_raw(off_cur, cur.soff, tgrp);
  int u_voff = u_voff_of(cur.blk);
  f32x4 uq[8];
#pragma unroll
  for (int pp = 0; pp < 8; ++pp) uq[pp] = load_u(u_voff, 0, pp);
  if (tgrp == 0) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // (the patch request is older than the eight U loads)
    transform(0);
    __builtin_amdgcn_sched_barrier(0);
    request_after(0, 0, off_cur, cur.soff, off_nxt, nxt.soff);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const float relu_lo = a.relu ? 0.f : -__builtin_inff();
  const int ty = lr >> 3, tx = lr & 7;
  int duty = 1;                                             // (t + 1) mod 3 for the interval of stage t
  int t = 0;
  for (int k = 0; k < nmine; ++k) {
    for (int c = 0; c < nchunks; ++c, ++t) {
      const int buf = t & 1;
      const bool has1 = t + 1 < nstages;
      const bool seam = c + 1 == nchunks;                   // the next stage is chunk 0 of the next set
      const int nc = seam ? 0 : c + 1;
      const int nvoff = (seam && has1) ? u_voff_of(nxt.blk) : u_voff;
      const float* const vR = VB + buf * VS_VFL + lane * 4;
      f32x4 vq[2];                                          // (two register sets: a third spills -- 168 registers at three waves per SIMD)
      vq[0] = *(const f32x4*)vR;
      if (has1 && duty == tgrp) {
        // this wave's patch of stage t + 1 was requested three stages ago: older than the (at most eight) U loads in flight (and than a
        // seam's four stores)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        transform(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // stage t + 1 = (k, c + 1), or (k + 1, 0) at the seam: the request three stages on is relative to THAT stage's set
        if (!seam) request_after(t + 1, c + 1, off_cur, cur.soff, off_nxt, nxt.soff);
        else if (t + 4 < nstages) dma_raw(off_nxt, nxt.soff, 3);     // (nchunks >= 12: chunk 3 of the next set exists)
      }
      if (has1) duty = duty == 2 ? 0 : duty + 1;
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) {
        const f32x4 vc = vq[pp & 1];
        const f32x4 uc = uq[pp];
        acc[2 * pp] = mfma16(uc.x, vc.x, acc[2 * pp]);
        __builtin_amdgcn_sched_barrier(0);
        if (pp < 7) vq[(pp + 1) & 1] = *(const f32x4*)(vR + (pp + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
        acc[2 * pp + 1] = mfma16(uc.z, vc.z, acc[2 * pp + 1]);
        acc[2 * pp] = mfma16(uc.y, vc.y, acc[2 * pp]);
        acc[2 * pp + 1] = mfma16(uc.w, vc.w, acc[2 * pp + 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (!(SQD_VP_DIAG & 16)) uq[pp] = load_u(nvoff, nc, pp);                    // unconditional (see conv_wino_vs_kernel)
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(SQD_VP_DIAG & 8)) __builtin_amdgcn_s_barrier();
    }
    // ---- the set's epilogue: inverse transform (bias inside), ReLU, store; accumulators restart for the next set ----
    if (!(SQD_VP_DIAG & 4) || k + 1 == nmine) {
      f32x4 ov[4];
      auto inv = [&](auto half, auto put) {
        f32x2 s2[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          const f32x2 m0 = half(acc[xi * 4 + 0]), m1 = half(acc[xi * 4 + 1]), m2 = half(acc[xi * 4 + 2]), m3 = half(acc[xi * 4 + 3]);
          s2[xi][0] = m0 + m1 + m2;
          s2[xi][1] = m1 - (m2 + m3);
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          put(0 * 2 + bb, s2[0][bb] + s2[1][bb] + s2[2][bb]);
          put(1 * 2 + bb, s2[1][bb] - (s2[2][bb] + s2[3][bb]));
        }
      };
      inv([](const f32x4& v) { return (f32x2)v.lo; }, [&](int px, f32x2 yv) { ov[px].lo = yv; });
      inv([](const f32x4& v) { return (f32x2)v.hi; }, [&](int px, f32x2 yv) { ov[px].hi = yv; });
      const int n = cur.blk * 16 + 4 * g;
      if (n < a.N && (!(SQD_VP_DIAG & 2) || k + 1 == nmine)) {
        float* const ybase = a.y + cur.p0 * a.y_pitch + a.y_coff + n;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const int yy = 2 * ty + (px >> 1), xx = 2 * tx + (px & 1);
          if (cur.y0 + yy >= a.H || cur.x0 + xx >= a.W) continue;
          f32x4 v = ov[px];
          v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
          *(f32x4*)(ybase + ((long long)yy * a.W + xx) * a.y_pitch) = v;
        }
      }
    }
    if (k + 1 < nmine) {
      cur = nxt;
      off_cur = off_nxt;
      u_voff = u_voff_of(cur.blk);
      nxt = set_pos(k + 2 < nmine ? k + 2 : k + 1);
      off_nxt = set_mask(nxt);
#pragma unroll
      for (int p = 0; p < 16; ++p) acc[p] = z4;
      acc[5] = bias_of(cur.blk);
    }
  }
#endif
}

extern "C" int SQD_VP_ENTRY(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int C,
                                    int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu, void* stream) {
  SQD_CHECK_ARG(x && u_packed && y && B > 0 && H > 0 && W > 0 && C > 0 && N > 0);
  SQD_CHECK_ARG(C % 8 == 0 && C >= 8 * VS_WV && N % (16 * VS_WV) == 0 && Npad == N);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + N <= y_pitch);
  SQD_CHECK_ARG((long long)W * 6 * x_pitch * 4 < (1ll << 30));
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad * 8 * 4 < (1ll << 31));
  WinoVpArgs a{};
  a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = relu;
  a.gxn = sqd_cdiv(W, 16); a.gyn = sqd_cdiv(H, 4);
  const long long ngroups = (long long)B * a.gxn * a.gyn;
  a.parts = N / (16 * VS_WV);
  SQD_CHECK_ARG(ngroups * a.parts < (1ll << 30));
  a.ngroups = (int)ngroups; a.nsets = a.ngroups * a.parts;
  int cus = 256;
  { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount; }
  const int rounds = sqd_cdiv(a.nsets, cus);
  a.nwg = sqd_cdiv(a.nsets, rounds);                       // every workgroup walks `rounds` sets (the last ones one fewer)
  constexpr size_t lds = (size_t)(2 * VS_VFL + VS_WV * VS_RFL) * sizeof(float);
  hipLaunchKernelGGL(conv_wino_vp_kernel, dim3((unsigned)a.nwg), dim3(VS_WV * 64), lds, (hipStream_t)stream, a);
  return sqd_launch_status();
}

