// Winograd F(2x2, 3x3) convolution for NARROW outputs (N <= 80) and long reductions: ConvDet (reference: Conv2d(768 -> 72, 3, pad 1),
// src/model/squeezedet.py:73-75,83; squeezedetplus 512 -> 72) -- "V shared, N split across the waves".
//
// conv_wino_kernel<2,4> (conv_wino.hip) gives a wave one 4x16-pixel group and a 32-channel slice: N = 72 runs as three slices
// = 96 executed channels, the 6x18 input patch of a group is fetched and transformed once PER SLICE (three times), and 1800
// wave units on 2048 wave slots leave every SIMD with two units or one (profiles/r04_conv_wino_workgroup_times.log).  Here
//   * the unit of work is (group, 16-channel block): N = 72 is five blocks = 80 executed channels, 3000 units at bs = 20;
//   * a workgroup is TWELVE waves = twelve consecutive units of the flattened (group, block) list = three per SIMD, one workgroup
//     per CU: 250 workgroups on 256 CUs, every SIMD runs exactly 3 x C/8 x 32 MFMAs;
//   * the waves of a group share its transformed input V through LDS: per K chunk of 8 channels ONE wave of the group (the duty
//     rotates) transforms the patch and writes V (8 ds_write_b128), all of them read it back as MFMA B operands (8 ds_read_b128,
//     lane-contiguous: conflict-free).  A workgroup spans 3-4 groups; a group cut by a workgroup boundary is transformed by both
//     (1.4 transforms per group and chunk instead of 3);
//   * the weights never touch the LDS: a wave's A operands of a chunk -- its block's U, 16 bytes per lane and position pair, a
//     contiguous 1 KB per wave-instruction in the packed layout -- are loaded straight into registers, each position pair ONE
//     INTERVAL AHEAD of its use (eight loads in flight per wave, retired by counted s_waitcnt), so no wave ever waits for another
//     wave's weight fetch.  (The first version staged U through a double-buffered 2 x 40 KB LDS image: every wave then met the
//     slowest LDS-DMA of the workgroup at every stage barrier -- ablation: 195 us, 161 without the input transform, 170 without
//     the DMA, 139 without both; profiles/r05_conv_wino_vs_ablation.log.)
//   * the raw patch image is private to the wave that will transform it (one per wave: 48 KB), requested right after that wave's
//     previous transform has read it: its latency budget is the group's rotation period (1-5 intervals), and nobody but the
//     issuing wave waits for it;
//   * one raw s_barrier per chunk publishes V (double-buffered per group slot, 64 KB): interval c runs the MFMAs of chunk c next to
//     the transform of chunk c + 1.
// Arithmetic per output element is that of conv_wino_kernel (same transforms, same k order): bit-identical results.
#include "sqd_common.h"
#include <type_traits>
#ifndef SQD_VS_DIAG
#define SQD_VS_DIAG 0             /* ablation builds only (scratch/diag/vs_diag.sh; never in libsqdhip.so): bit 0 = no input transform,
                                     1 = no patch DMA inside the chunk loop, 2 = no stage barrier, 3 = U operands not loaded in the loop */
#endif
/* (s_setprio for the duty wave behind its transform measured slower, 186 vs 174 us, and is gone.) */
#ifndef SQD_VS_ENTRY
#define SQD_VS_ENTRY sqd_conv_wino_vs_fwd
#endif

struct WinoVsArgs {
  const float* x; const float* u; const float* bias; float* y;
  int B, H, W;
  int C, x_pitch, x_coff;
  int N, y_pitch, y_coff;
  int relu;
  int gxn, gyn, ngroups, nunits, nwg;
};

typedef __attribute__((address_space(3))) void* lds_ptr_vs_t;
typedef unsigned u32x4_vs __attribute__((ext_vector_type(4)));

constexpr int VS_NB = 5;                      // 16-channel blocks of the packed weights (Npad = 80)
constexpr int VS_WV = 12;                     // waves per workgroup = units per workgroup
constexpr int VS_SLOTS = 4;                   // groups a workgroup can touch (12 consecutive units of 5-unit groups)
constexpr int VS_UFL = 8 * VS_NB * 256;       // floats of one chunk's U: [8 position pairs][5 blocks][4 cp][16 n][parity][2 ch]
constexpr int VS_VFL = 8 * 256;               // floats of one chunk's V of a group: [8 position pairs][64 lanes][4]
constexpr int VS_RFL = 256 * 4;               // floats of a raw patch image (226 of 256 16-byte slots used)
constexpr size_t VS_LDS = (size_t)(VS_SLOTS * 2 * VS_VFL + VS_WV * VS_RFL) * sizeof(float);
static_assert(VS_LDS == 112 * 1024, "64 KB of V + 48 KB of wave-private patches: one workgroup per CU");

__global__ __launch_bounds__(VS_WV * 64, 3) void conv_wino_vs_kernel(WinoVsArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RP = 113;                     // slots per k-quad plane of the raw patch (as conv_wino_kernel: planes interleave in the bank row)
  constexpr int RAW_IT = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const VB = smem;                                   // [VS_SLOTS][2][VS_VFL]
  float* const RB = VB + VS_SLOTS * 2 * VS_VFL;             // [VS_WV][VS_RFL]

  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunks = a.C >> 3;
  // workgroups of one XCD own a contiguous run of units: the two workgroups that share a cut group fetch its patch through one L2
  const int wgpos = sqd_xcd_contiguous((int)blockIdx.x, a.nwg);
  const int u0 = wgpos * VS_WV;
  const int g_first = u0 / VS_NB;
  const int unit_raw = u0 + wv;
  const int valid = unit_raw < a.nunits;
  const int unit = valid ? unit_raw : a.nunits - 1;         // idle waves of the last workgroup redo its last unit (not stored)
  const int grp = unit / VS_NB, blk = unit - grp * VS_NB;
  const int slot = grp - g_first;                           // 0 .. 3
  // the waves of this workgroup that work on `grp`: [w_lo, w_hi); the transform duty of chunk c goes to wave w_lo + c mod (w_hi - w_lo)
  int w_lo = grp * VS_NB - u0; w_lo = w_lo < 0 ? 0 : w_lo;
  int w_hi = (grp == a.ngroups - 1) ? VS_WV : (grp + 1) * VS_NB - u0; w_hi = w_hi > VS_WV ? VS_WV : w_hi;
  const int nw = w_hi - w_lo, rank = wv - w_lo;

  // ---- group geometry (wave-uniform, once) ----
  const int q1 = grp / a.gxn, gxi = grp - q1 * a.gxn;
  const int b = q1 / a.gyn, gyi = q1 - b * a.gyn;
  const int y0 = gyi * 4, x0 = gxi * 16;
  const long long p0 = ((long long)b * a.H + y0) * a.W + x0;
  const unsigned soff0 = (unsigned)(p0 * a.x_pitch * 4);    // byte offset of the patch origin from the resource base (host-checked < 3 GiB)

  // ---- per-lane DMA slots of the patch (fixed for the kernel: the wave's group never changes) ----
  constexpr unsigned OOB = 0x80000000u;
  int r_off[RAW_IT];
#pragma unroll
  for (int it = 0; it < RAW_IT; ++it) {
    const int s = it * 64 + lane;
    const int kq = s / RP, pix = s - kq * RP;
    const bool real = kq < 2 && pix < 108;
    const int r = pix / 18, c = pix - r * 18;
    const bool ok = real && (unsigned)(y0 + r - 1) < (unsigned)a.H && (unsigned)(x0 + c - 1) < (unsigned)a.W;
    r_off[it] = ok ? ((r * a.W + c) * a.x_pitch + 4 * kq) * 4 : (int)OOB;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 0x7ffffff0, 0x00020000);
  float* const rawS = RB + wv * VS_RFL;                     // this wave's own patch image
  float* const vS = VB + slot * 2 * VS_VFL;
  auto dma_raw = [&](int cc) {                               // the patch of chunk cc -> this wave's raw image
    if ((SQD_VS_DIAG & 2) && cc >= 2 * nw) return;
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_vs_t)(rawS + it * 64 * 4), 16, r_off[it], (int)(soff0 + (unsigned)cc * 32u), 0, 0);
  };
  // A operands: U[chunk][pp][blk][cp = g][n = lr][parity][2 ch] -- 16 bytes per lane, lane-contiguous
  const int u_voff = (blk * 256 + lane * 4) * 4;
  auto load_u = [&](int cc, int pp) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ures, u_voff + pp * (VS_NB * 1024),
                                                                           (int)((unsigned)cc * (unsigned)(VS_UFL * 4)), 0));
  };

  // ---- transform unit of a lane = its MFMA B-operand role: tile lr, channel pair g of the chunk ----
  const float* const rawL = rawS + (((g >> 1) * RP + (2 * (lr >> 3)) * 18 + 2 * (lr & 7)) * 4 + 2 * (g & 1));
  // raw image -> V[vbuf] of this group (whole wave); the image is re-requested for chunk `next_cc` (< 0: not) as soon as it has been read
  auto transform = [&](int vbuf, int next_cc) {
    f32x2 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 d0 = *(const f32x2*)(rawL + (0 * 18 + j) * 4), d1 = *(const f32x2*)(rawL + (1 * 18 + j) * 4);
      const f32x2 d2 = *(const f32x2*)(rawL + (2 * 18 + j) * 4), d3 = *(const f32x2*)(rawL + (3 * 18 + j) * 4);
      t[0][j] = d0 - d2; t[1][j] = d1 + d2; t[2][j] = d2 - d1; t[3][j] = d1 - d3;
    }
    // the raw image is refilled by LDS-DMA right behind this: its reads must have returned, and nothing may move across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (next_cc >= 0) dma_raw(next_cc);
    __builtin_amdgcn_sched_barrier(0);
    if (SQD_VS_DIAG & 1) return;
    f32x4* const vdst = (f32x4*)(vS + vbuf * VS_VFL) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 v0 = t[i][0] - t[i][2], v1 = t[i][1] + t[i][2], v2 = t[i][2] - t[i][1], v3 = t[i][1] - t[i][3];
      f32x4 w0, w1;
      w0.lo = v0; w0.hi = v1; w1.lo = v2; w1.hi = v3;       // position pair 2i: positions 4i, 4i+1; pair 2i+1: positions 4i+2, 4i+3
      vdst[(2 * i) * 64] = w0;
      vdst[(2 * i + 1) * 64] = w1;
    }
  };

  f32x4 acc[16];
  f32x4 biasv = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const int n = blk * 16 + 4 * g;
    if (a.bias && n < a.N) biasv = *(const f32x4*)(a.bias + n);
  }
  const float* const vL = vS + lane * 4;                    // + buf * VS_VFL + pp * 256

  // ---- prologue: this wave's first patch (chunk `rank`: the first chunk it transforms), chunk 0's U, T(0) ----
  if (rank < nchunks) dma_raw(rank);
  f32x4 uq[8];
#pragma unroll
  for (int pp = 0; pp < 8; ++pp) uq[pp] = load_u(0, pp);
  int duty = 0;                                             // c mod nw of the chunk whose transform is next
  if (duty == rank) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // the patch (older than the eight U loads) has landed
    transform(0, nw < nchunks ? nw : -1);
  }
  duty = duty + 1 == nw ? 0 : duty + 1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  // one interval: M(cc) from registers (U) and LDS (V), T(cc + 1) by the duty wave, the U loads of chunk cc + 1 behind each step.
  // Chunk 0 is peeled so that the steady-state loop body is one straight-line variant.
  auto interval = [&](int cc, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    const int buf = cc & 1;
    const bool has1 = cc + 1 < nchunks;
    const int nc = has1 ? cc + 1 : cc;
    const float* const vR = vL + buf * VS_VFL;
    // B operands: V of step pp + 2 is requested while step pp runs (three register sets): a wave that has fallen behind -- the duty
    // wave, once its transform is done -- then runs its steps back to back without waiting for the LDS at each one
    f32x4 vq[3];
    vq[0] = *(const f32x4*)vR;
    vq[1] = *(const f32x4*)(vR + 256);
    if (has1 && duty == rank) {
      // this wave's patch of chunk cc + 1 was requested nw intervals ago, before everything but the (at most eight) U loads that
      // are still in flight: vector memory operations retire in order
      // (interval 0: a wave that transforms EVERY chunk requested chunk 1's patch behind the prologue's U loads -- wait for all)
      if (FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      transform(buf ^ 1, cc + 1 + nw < nchunks ? cc + 1 + nw : -1);
    }
    if (has1) duty = duty + 1 == nw ? 0 : duty + 1;
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      const f32x4 vc = vq[pp % 3];
      const f32x4 uc = uq[pp];
      acc[2 * pp] = mfma16(uc.x, vc.x, FIRST ? z4 : acc[2 * pp]);
      __builtin_amdgcn_sched_barrier(0);
      if (pp < 6) vq[(pp + 2) % 3] = *(const f32x4*)(vR + (pp + 2) * 256);
      __builtin_amdgcn_sched_barrier(0);
      acc[2 * pp + 1] = mfma16(uc.z, vc.z, FIRST ? ((2 * pp + 1 == 5) ? biasv : z4) : acc[2 * pp + 1]);
      acc[2 * pp] = mfma16(uc.y, vc.y, acc[2 * pp]);
      acc[2 * pp + 1] = mfma16(uc.w, vc.w, acc[2 * pp + 1]);
      __builtin_amdgcn_sched_barrier(0);
      // the same position pair of the next chunk: one interval of latency budget.  Issued UNCONDITIONALLY (the last interval re-reads
      // its own chunk): behind a branch the compiler's wait-count bookkeeping assumes the loads absent and drains them at every step
      if (!(SQD_VS_DIAG & 8)) uq[pp] = load_u(nc, pp);
      __builtin_amdgcn_sched_barrier(0);
    }
    // V(cc + 1) written (duty wave) and V(cc) read (everyone) before the barrier publishes one and frees the other; the U loads and
    // patch requests in flight stay in flight (a raw s_barrier: __syncthreads() would drain the vector memory queue)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(SQD_VS_DIAG & 4)) __builtin_amdgcn_s_barrier();
  };
  interval(0, std::true_type{});
  for (int cc = 1; cc < nchunks; ++cc) interval(cc, std::false_type{});

  // ---- inverse transform Y = A^T M A (register pairs), bias is inside (position (1,1) started from it), ReLU, store ----
  if (!valid) return;
  const int ty = lr >> 3, tx = lr & 7;
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();
  f32x4 ov[4];
  auto inv = [&](auto half, auto put) {
    f32x2 s[4][2];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const f32x2 m0 = half(acc[xi * 4 + 0]), m1 = half(acc[xi * 4 + 1]), m2 = half(acc[xi * 4 + 2]), m3 = half(acc[xi * 4 + 3]);
      s[xi][0] = m0 + m1 + m2;
      s[xi][1] = m1 - (m2 + m3);
    }
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      put(0 * 2 + bb, s[0][bb] + s[1][bb] + s[2][bb]);
      put(1 * 2 + bb, s[1][bb] - (s[2][bb] + s[3][bb]));
    }
  };
  inv([](const f32x4& v) { return (f32x2)v.lo; }, [&](int px, f32x2 yv) { ov[px].lo = yv; });
  inv([](const f32x4& v) { return (f32x2)v.hi; }, [&](int px, f32x2 yv) { ov[px].hi = yv; });
  const int n = blk * 16 + 4 * g;
  if (n >= a.N) return;
  float* const ybase = a.y + p0 * a.y_pitch + a.y_coff + n;
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    const int yy = 2 * ty + (px >> 1), xx = 2 * tx + (px & 1);
    if (y0 + yy >= a.H || x0 + xx >= a.W) continue;
    f32x4 v = ov[px];
    v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
    *(f32x4*)(ybase + ((long long)yy * a.W + xx) * a.y_pitch) = v;
  }
#endif
}

// y[..., y_coff : y_coff + N] = (ReLU)(conv3x3(x[..., x_coff : x_coff + C]) + bias), N <= 80, u_packed = sqd_pack_wino_weight output with
// Npad = 80 (C/8 * 16 * 80 * 8 floats).  Same results, bit for bit, as sqd_conv_wino_fwd.
extern "C" int SQD_VS_ENTRY(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int C,
                                    int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu, void* stream) {
  SQD_CHECK_ARG(x && u_packed && y && B > 0 && H > 0 && W > 0 && C > 0 && N > 0);
  SQD_CHECK_ARG(C % 8 == 0 && N % 4 == 0 && N <= 16 * VS_NB && Npad == 16 * VS_NB);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + N <= y_pitch);
  SQD_CHECK_ARG((long long)W * 6 * x_pitch * 4 < (1ll << 30));                                    // per-lane byte offsets inside a patch
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));                  // 32-bit SGPR byte offset of a group origin
  SQD_CHECK_ARG((long long)(C >> 3) * VS_UFL * 4 < (1ll << 32));
  WinoVsArgs a{};
  a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = relu;
  a.gxn = sqd_cdiv(W, 16); a.gyn = sqd_cdiv(H, 4);
  const long long ngroups = (long long)B * a.gxn * a.gyn;
  SQD_CHECK_ARG(ngroups * VS_NB < (1ll << 30));
  a.ngroups = (int)ngroups; a.nunits = a.ngroups * VS_NB; a.nwg = sqd_cdiv(a.nunits, VS_WV);
  static SqdDevOnce once;
  if (int rc = sqd_max_lds_once(once, (const void*)conv_wino_vs_kernel, (int)VS_LDS)) return rc;
  hipLaunchKernelGGL(conv_wino_vs_kernel, dim3((unsigned)a.nwg), dim3(VS_WV * 64), VS_LDS, (hipStream_t)stream, a);
  return sqd_launch_status();
}
