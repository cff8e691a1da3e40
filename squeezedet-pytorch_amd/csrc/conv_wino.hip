// Winograd F(2x2, 3x3) convolution (3x3 / pad 1 / stride 1) on the gfx950 fp32 matrix cores.
//
// Serves the same layers as the 3x3 implicit-GEMM kernel of conv_igemm.hip (reference: Fire expand3x3,
// src/model/squeezedet.py:14,20-22, and ConvDet, :73-75,83) where the host's measured table says it is faster: the
// 3x3 convolution over a 2x2 output tile is evaluated as 16 element-wise products in the transformed domain,
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A ,
// i.e. 16 independent [tiles x C] x [C x N] GEMMs -- 2.25x fewer multiply-adds than the direct form.  fp32 throughout
// (the transforms only add, subtract and halve: measured error vs an fp64 convolution is the same ~2e-7 relative as the
// direct kernel's, scratch/wino_numerics.py), so the 1e-4 parity bound is untouched.
//
// Decomposition: a GROUP is 4 rows x 16 columns of output = 16 Winograd tiles (one MFMA column block); each WAVE owns
// one group and all 16 transform positions for a slice of 16*NT output channels (16*NT f32x4 accumulators).  A
// workgroup is WV waves working on WV consecutive groups of the flattened (image, row-group, column-group) list --
// they need not be adjacent, so the only padding is the 16-column / 4-row granularity (78 -> 80 columns) -- and shares
// the transformed weights U of the slice through LDS.  Per K chunk of 8 channels:
//   * the wave's 6x18-pixel input patch arrives by buffer-resource LDS-DMA into a wave-private raw buffer (slots outside
//     the image are zero-filled by the hardware range check),
//   * every lane transforms one (tile, channel pair): 16 ds_read_b64, 32 packed adds, 8 ds_write_b128 into the
//     wave-private V image [8 position pairs][4 channel pairs][16 tiles][parity][2 channels],
//   * 16 positions x 2 k-steps x NT MFMAs read V and U, one conflict-free ds_read_b128 per position pair and operand,
//   * the next chunk's patch and U slice are fetched by LDS-DMA during the MFMA phase (U double-buffered; the patch
//     buffer is wave-private and dead once the wave has transformed it, so it needs no second copy).
// Workgroups are persistent (strided super-group list); a tile's first chunk starts its accumulators from 0 / the bias
// (position (1,1) enters all four outputs with weight +1), the inverse transform runs on register pairs after the last
// chunk and the stores drain behind the next stage's barrier.  4-wave workgroups use exactly 80 KB of LDS: two per CU.
//
// Packed weights: U[C/8][8 position pairs][Npad/16][4 channel pairs][16 n][position parity][2 channels] -- the LDS image
// of a slice is a set of contiguous runs (sqd_pack_wino_weight; host: ops.WinoPlan).
#include "sqd_common.h"
#include <type_traits>
#ifndef SQD_WINO_DIAG
#define SQD_WINO_DIAG 0           /* ablation builds only (scratch/diag/build_wino_diag.sh): bit 0 = no MFMA, 1 = no input transform,
                                     2 = no DMA inside the chunk loop, 3 = no output stores.  0 in the product library. */
#endif

struct WinoArgs {
  const float* x; const float* u; const float* bias; float* y;
  int B, H, W;
  int C, x_pitch, x_coff;
  int N, Npad, y_pitch, y_coff;
  int relu, accumulate;         // epilogue: y += result
  const float* ymask; const float* ymul;   // epilogue (same pitch / channel offset as y): multiply by ymul (dropout), zero where ymask <= 0
  int gxn, gyn;                 // column / row groups per image
  unsigned gxn_m, gyn_m;        // ceil(2^32 / gxn), ceil(2^32 / gyn): the group index is split with two scalar multiply-highs
                                // (a division of wave-uniform integers otherwise goes through ~10 vector instructions + readfirstlane,
                                // twice per tile, and every vector instruction delays the SIMD's MFMA stream)
  int ngroups, ntiles;          // B*gyn*gxn groups; ntiles = super-groups of WV groups
  int nslices, gx;              // persistent grid: gx tile streams x nslices channel slices
  int wg_cap;
  // fused Fire expand (sqd_fire_wino_fwd): slices [0, nslices3) are expand3x3 slices of 32 channels (N, y_coff, bias as above);
  // slices [nslices3, nslices) are expand1x1 slices of 128 channels (N1, y_coff1, bias1) riding in the same launch
  int nslices3, N1, y_coff1;
  const float* bias1;
  // Fire -> Fire bridge (sqd_fire_bridge_fwd, wino_bridge.h): the next squeeze as MFMA operands, bias tables, its width
  const float* br_w; const float* br_bias; const float* br_sqb;
  int br_nsq;
  // ... through a 3x3 / stride-2 max pool (sqd_fire_pool_bridge_fwd, wino_poolbridge.h): pooled size, strips of 7 pooled
  // columns, segments of pb_gseg group rows (4 pixel rows each) out of pb_ng
  int pb_hp, pb_wp, pb_ns, pb_nseg, pb_gseg, pb_ng;
  // training forms of the two bridges: the tensor the backward needs is stored as well.  Fire -> Fire: sv = the concatenated expand
  // output (windows sv_coff = expand3x3, sv_coff1 = expand1x1).  Through the pool: sv = the POOLED expand output (same two windows)
  // and sv_codes = the pool's arg-max / ReLU codes (one byte per pooled element, maxpool_fwd_kernel<true, true>'s codes).
  float* sv; int sv_pitch, sv_coff, sv_coff1;
  unsigned char* sv_codes;
};

typedef __attribute__((address_space(3))) void* lds_ptr_w_t;

// Diagnostic build only (-DSQD_WINO_STAMP, scratch/diag/wino_stamp.sh; never in libsqdhip.so): every workgroup of conv_wino_kernel
// records {HW_ID, XCC_ID, start, end (100 MHz clock), tiles done} -- which workgroups of a two-per-CU launch finish early.
#ifdef SQD_WINO_STAMP
__device__ long long* sqd_wino_dbg = nullptr;
extern "C" int sqd_wino_set_debug(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(sqd_wino_dbg), &p, sizeof(p)) == hipSuccess ? SQD_OK : SQD_ERR_LAUNCH;
}
#endif

__device__ __forceinline__ f32x4 wino_relu4(f32x4 v, float lo) {
  asm volatile("v_max_f32 %0, %4, %0\n\tv_max_f32 %1, %4, %1\n\tv_max_f32 %2, %4, %2\n\tv_max_f32 %3, %4, %3"
               : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "s"(lo));
  return v;
}

// Residency: 4 waves x 32-channel slices = 80 KB of LDS and 253 VGPRs -> two workgroups per CU (two waves per SIMD).  The
// 16-channel slice (NT = 1) needs half the accumulators and half the parking area: 48 KB and <= 168 VGPRs -> THREE workgroups per
// CU, three waves per SIMD.  That is the ConvDet configuration of round 3: N = 72 runs as 5 slices of 16 (80 channels) instead of
// 3 of 32 (96), i.e. 3000 wave-units of 3072 MFMAs on 3072 wave slots -- 9216 MFMAs on every SIMD -- where the 32-channel
// slicing put 1800 units of 6144 on 2048 slots (12288 on the busiest SIMDs).
template <int NT, int WV>
__global__ __launch_bounds__(WV * 64, (WV == 4 && NT == 2) ? 2 : ((WV == 4 && NT == 1) ? 3 : 1)) void conv_wino_kernel(WinoArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)    // the host pass only needs the launch stub (the buffer-resource builtins are device-only)
  constexpr int NTHR = WV * 64;
  constexpr int BN = 16 * NT;
  constexpr int RP = 113;                     // slots per k-quad plane of the raw patch (108 used): 113*16 B = 16 mod 256, so the
                                              // two planes interleave in the LDS bank row and the transform's reads do not collide
  constexpr int RAW_IT = 4;                   // 256 slots per wave (226 used)
  constexpr int USLOTS = 32 * BN;             // 16 pos x BN channels x 2 halves of 16 B
  static_assert(USLOTS % NTHR == 0, "U slice must be whole workgroup passes");
  constexpr int U_IT = USLOTS / NTHR;
  constexpr int NDMA = RAW_IT + U_IT;
  constexpr int NSTEP = 8;                    // MFMA steps of 2 positions
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const rawB = smem;                                 // [WV][256][4]
  float* const VB = rawB + WV * 256 * 4;                    // [WV][4 px][NT][64 lanes] f32x4: a finished tile's outputs, parked
  float* const UB = VB + WV * 1024 * NT;                    // [2][USLOTS][4]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int wgq = (int)blockIdx.x >> 3;
  const int n0 = (wgq % a.nslices) * BN;
  const int tstride = a.gx;
  const int nchunks = a.C >> 3;
  const int ntiles = a.ntiles;
  // tile stream of this workgroup: the streams of ONE XCD (workgroup ids equal mod 8) cover a CONTIGUOUS run of super-groups --
  // row-groups that are vertical neighbours share 2 of the 6 patch rows, and with the streams dealt round-robin (stream = 8 q + xcd,
  // rounds 1 and 2) every XCD's L2 fetched those rows for itself: 125.7 MB of fabric traffic per launch against 70 MB algorithmic.
  // The channel slices of a stream stay XCD-adjacent as before (consecutive wgq).
  int tile = ((int)blockIdx.x & 7) * (a.gx >> 3) + wgq / a.nslices;
  if (tile >= ntiles) return;
  const int wv_s = __builtin_amdgcn_readfirstlane(wv);
#ifdef SQD_WINO_STAMP
  const long long stamp_t0 = wall_clock64();
  long long stamp_tiles = 0;
#endif

  // ---- per-lane DMA slots ----
  // Both DMA streams go through buffer resources: address = resource base + wave-uniform SGPR offset (group origin /
  // K chunk) + per-lane 32-bit offset, so a stage's DMA issue costs no vector instruction at all, and a lane whose slot
  // lies outside the image carries an offset beyond the resource's range -- the hardware range check then returns zeros
  // (the zero padding of the convolution) without touching memory.  Only the per-lane offset is range-checked, the SGPR
  // offset is not (MI355X_MICROARCH.md buffer addressing), so the range only has to exceed any in-patch offset.
  constexpr unsigned OOB = 0x80000000u;
  int r_offB[RAW_IT], r_key[RAW_IT];                        // byte offset inside the patch (padding slots: 0), row << 8 | col (-1 = padding)
#pragma unroll
  for (int it = 0; it < RAW_IT; ++it) {
    const int slot = it * 64 + lane;
    const int kq = slot / RP, pix = slot - kq * RP;
    const bool real = kq < 2 && pix < 108;
    const int r = pix / 18, c = pix - r * 18;
    r_key[it] = real ? (r << 8 | c) : -1;
    r_offB[it] = real ? ((r * a.W + c) * a.x_pitch + 4 * kq) * 4 : 0;
  }
  int u_offB[U_IT];
#pragma unroll
  for (int it = 0; it < U_IT; ++it) {
    const int slot = it * NTHR + tid;
    const int pp = slot / (4 * BN), rem = slot - pp * (4 * BN);     // position pair, 16-byte slot inside the slice's run
    u_offB[it] = ((pp * a.Npad + n0) * 16 + rem * 4) * 4;
  }
  const unsigned u_chunkB = 16u * a.Npad * 8u * 4u;
  // patch resource: based one halo row + one halo column BEFORE the window's first element, so every group's SGPR offset
  // (p0 * pitch * 4, host-checked < 4 GiB) is non-negative
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 0x7ffffff0, 0x00020000);

  struct GPos { int y0, x0, inner, valid; long long p0; unsigned soff; };
  auto group_pos = [&](int t) {                              // this wave's group of super-group t (all wave-uniform)
    GPos gp;
    int q = t * WV + wv_s;
    gp.valid = (int)((unsigned)(q - a.ngroups) >> 31);       // q < ngroups
    q = gp.valid ? q : a.ngroups - 1;                        // idle waves of the last super-group redo the last group (not stored)
    const int q1 = a.gxn_m ? (int)__umulhi((unsigned)q, a.gxn_m) : q;        // q / gxn (exact: q * gxn < 2^32, host-checked; magic 0 = divisor 1)
    const int gxi = q - q1 * a.gxn;
    const int b = a.gyn_m ? (int)__umulhi((unsigned)q1, a.gyn_m) : q1;       // q1 / gyn
    const int gyi = q1 - b * a.gyn;
    gp.y0 = gyi * 4; gp.x0 = gxi * 16;
    gp.p0 = ((long long)b * a.H + gp.y0) * a.W + gp.x0;
    gp.soff = (unsigned)(gp.p0 * a.x_pitch * 4);             // byte offset of the patch origin from the resource base
    gp.inner = (int)(((unsigned)(-gp.y0) & (unsigned)(gp.y0 + 4 - a.H) & (unsigned)(-gp.x0) & (unsigned)(gp.x0 + 16 - a.W)) >> 31);
    return gp;
  };
  // per-lane patch offsets of the group whose stages are being fetched: evaluated once per group (interior groups: the
  // plain offsets), out-of-image slots -> OOB
  int r_offG[RAW_IT];
  auto group_offsets = [&](const GPos gp) {
    if (gp.inner) {
#pragma unroll
      for (int it = 0; it < RAW_IT; ++it) r_offG[it] = r_offB[it];
      return;
    }
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
      const int key = r_key[it];
      const bool ok = key >= 0 && (unsigned)(gp.y0 + (key >> 8) - 1) < (unsigned)a.H && (unsigned)(gp.x0 + (key & 255) - 1) < (unsigned)a.W;
      r_offG[it] = ok ? r_offB[it] : (int)OOB;
    }
  };
  float* const rawW = rawB + wv_s * 256 * 4;                 // this wave's raw patch / V image
  float* const VW = VB + wv_s * 1024 * NT;
  auto dma_raw_one = [&](int it, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_w_t)(rawW + it * 64 * 4), 16, r_offG[it], (int)soff, 0, 0);
  };
  auto dma_u_one = [&](int it, int cc, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ures, (lds_ptr_w_t)(UB + (buf * USLOTS + it * NTHR + wv_s * 64) * 4), 16, u_offB[it],
                                             (int)(cc * u_chunkB), 0, 0);
  };

  f32x4 acc[16][NT];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the slice's bias for this lane's 4 channels of every block (registers: with 4 waves the LDS is exactly two workgroups per CU)
  f32x4 biasv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + 4 * g;
    biasv[j] = (a.bias && n < a.N) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int ty = lr >> 3, tx = lr & 7;
  int o_off[4];
#pragma unroll
  for (int px = 0; px < 4; ++px) o_off[px] = ((2 * ty + (px >> 1)) * a.W + 2 * tx + (px & 1)) * a.y_pitch + 4 * g;
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();

  // transform unit of this lane: tile tt, channel pair cp of the chunk
  // (lanes 0-31 = tile row 0, lanes 32-63 = tile row 1; within a half: 8 tiles x (k-quad, channel half) -- the 32 lanes of
  // a ds_read_b64 half then cover 32 distinct 8-byte units of one 256-byte bank row: conflict-free)
  // = the lane's role as MFMA B operand (column = tile lr, k = channel pair g): the transformed values V[pos][tile][2 ch]
  // are consumed by the lane that computed them and never leave its registers (the patch reads are 2-way bank
  // conflicted in this order: 16 ds_read_b64 per chunk, cheap next to a V round trip through LDS)
  const int tt = lr, cp = g;
  const float* const rawL = rawW + (((cp >> 1) * RP + (2 * (tt >> 3)) * 18 + 2 * (tt & 7)) * 4 + 2 * (cp & 1));
  // U image per 16 channels: [8 position pairs][4 channel pairs][16 n][pos parity][2 channels] (k-quad-major: conflict-free)
  const float* const uR0 = UB + g * 64 + lr * 4;
  // the former V image now parks a finished tile's outputs until the next barrier: [4 px][NT][64 lanes] f32x4
  f32x4* const parkW = (f32x4*)VW + lane;

  GPos cur = group_pos(tile);
  group_offsets(cur);
#pragma unroll
  for (int it = 0; it < RAW_IT; ++it) dma_raw_one(it, cur.soff);
#pragma unroll
  for (int it = 0; it < U_IT; ++it) dma_u_one(it, 0, 0);
  int ubuf = 0;
  bool pending = false;
  GPos ptp = cur;

  const int acc_i = a.accumulate, has_mul = a.ymul != nullptr, has_mask = a.ymask != nullptr;
  auto epi = [&](f32x4 v, int j, float* dst, const float* mul, const float* mask) {
    if (acc_i) v += *(const f32x4*)dst;                       // (the bias is already inside: accumulator (1,1) started from it)
    if (has_mul) v *= *(const f32x4*)mul;
    if (has_mask) {
      const f32x4 m = *(const f32x4*)mask;
      v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    }
#if SQD_WINO_DIAG & 8
    if (v.x == 123.456f)
#endif
    *(f32x4*)dst = wino_relu4(v, relu_lo);
  };
  auto flush = [&](const GPos gp) {
    float* ybase = a.y + gp.p0 * a.y_pitch + a.y_coff + n0;
    const float* mulbase = a.ymul + gp.p0 * a.y_pitch + a.y_coff + n0;     // dereferenced only when present
    const float* maskbase = a.ymask + gp.p0 * a.y_pitch + a.y_coff + n0;
    const bool whole = gp.y0 + 4 <= a.H && gp.x0 + 16 <= a.W && n0 + BN <= a.N;
    if (!gp.valid) return;
    if (whole) {
#pragma unroll
      for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int off = o_off[px] + j * 16;
          epi(parkW[(px * NT + j) * 64], j, ybase + off, mulbase + off, maskbase + off);
        }
      return;
    }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const bool valid = gp.y0 + 2 * ty + (px >> 1) < a.H && gp.x0 + 2 * tx + (px & 1) < a.W;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!valid || n0 + j * 16 + 4 * g >= a.N) continue;
        const int off = o_off[px] + j * 16;
        epi(parkW[(px * NT + j) * 64], j, ybase + off, mulbase + off, maskbase + off);
      }
    }
  };

  for (;;) {
    const int more_i = (int)((unsigned)(tile + tstride - ntiles) >> 31);
    const bool more = more_i != 0;
    const GPos nxt = group_pos(more ? tile + tstride : tile);
    for (int cc = 0; cc < nchunks; ++cc) {
      // this wave's share of the stage's DMA (patch + U slots) must have LANDED before the barrier publishes it to the other
      // waves: the compiler only waits on vmcnt where the issuing wave itself reads DMA-written LDS, which says nothing
      // about the slots other waves fetched
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // all waves left the previous U buffer
      if (pending) { flush(ptp); pending = false; }
      const int last_i = 1 - (int)((unsigned)(cc + 1 - nchunks) >> 31);
      const bool last = last_i != 0;
      const int ncc = last ? 0 : cc + 1;
      // the stage fetched during this chunk: the next K chunk of this group, or (last chunk) chunk 0 of the next group,
      // whose per-lane offsets replace this group's now (its own last stage is already in LDS)
      if (last) group_offsets(nxt);
      const unsigned nsoff = (last ? nxt.soff : cur.soff) + (unsigned)ncc * 32u;

      // ---- input transform: V = B^T d B for (tile tt, channels 2cp, 2cp+1) ----
      f32x2 vv[16];                          // V[pos] for (tile lr, channels 2g, 2g+1): this lane's B operands of the chunk
#if SQD_WINO_DIAG & 2
#pragma unroll
      for (int i = 0; i < 16; ++i) vv[i] = (f32x2){(float)(lane + i + cc), 1.f};
#else
      {
        f32x2 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 d0 = *(const f32x2*)(rawL + (0 * 18 + j) * 4), d1 = *(const f32x2*)(rawL + (1 * 18 + j) * 4);
          const f32x2 d2 = *(const f32x2*)(rawL + (2 * 18 + j) * 4), d3 = *(const f32x2*)(rawL + (3 * 18 + j) * 4);
          t[0][j] = d0 - d2; t[1][j] = d1 + d2; t[2][j] = d2 - d1; t[3][j] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vv[i * 4 + 0] = t[i][0] - t[i][2]; vv[i * 4 + 1] = t[i][1] + t[i][2];
          vv[i * 4 + 2] = t[i][2] - t[i][1]; vv[i * 4 + 3] = t[i][1] - t[i][3];
        }
      }
#endif
      // The patch buffer is refilled (LDS-DMA, below) for the next chunk: its reads above must have returned, and neither
      // the compiler nor the machine scheduler may move a DMA issue across this point (the DMA's LDS side is invisible to
      // them).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const float* const uR = uR0 + ubuf * USLOTS * 4;

      // ---- 16 positions x 2 k-steps x NT MFMAs, software-pipelined over two operand sets (steps of 2 positions) ----
      auto load_ops = [&](int step, f32x4 (&afr)[NT]) {                       // step = position pair
#pragma unroll
        for (int j = 0; j < NT; ++j) afr[j] = *(const f32x4*)(uR + (step * NT + j) * 256);
      };
      // FIRST (chunk 0 of a tile): the first MFMA of every accumulator takes C = 0 instead of the accumulator -- no zeroing
      // pass after the previous tile -- and position (1,1) starts from the bias: m11 enters all four outputs of the inverse
      // transform with weight +1, so the bias add of the epilogue comes for free.
      auto mfma_phase = [&](auto first_c) {
      constexpr bool FIRST = decltype(first_c)::value;
      auto mfma_pos = [&](int step, const f32x4 (&afr)[NT], int h) {
        const int p = 2 * step + h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const f32x4 c0v = (FIRST && t == 0) ? ((p == 5) ? biasv[j] : (f32x4){0.f, 0.f, 0.f, 0.f}) : acc[p][j];
#if SQD_WINO_DIAG & 1
            asm volatile("" ::"v"(afr[j][2 * h + t]), "v"(vv[p][t]));      // operands materialised, no matrix instruction
            acc[p][j] = c0v;
#else
            acc[p][j] = mfma16(afr[j][2 * h + t], vv[p][t], c0v);
#endif
          }
      };
      f32x4 af0[NT], af1[NT];
      load_ops(0, af0);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
          if (q * NSTEP / NDMA != step) continue;      // the next stage's DMA instructions are spread over the 8 MFMA steps
          if (SQD_WINO_DIAG & 4) continue;
          if (q < RAW_IT) dma_raw_one(q < RAW_IT ? q : 0, nsoff);
          else dma_u_one(q - RAW_IT, ncc, ubuf ^ 1);
        }
        if (step & 1) {
          mfma_pos(step, af1, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (step + 1 < NSTEP) load_ops(step + 1, af0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_pos(step, af1, 1);
        } else {
          mfma_pos(step, af0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (step + 1 < NSTEP) load_ops(step + 1, af1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_pos(step, af0, 1);
        }
      }
      };
      if (cc == 0) mfma_phase(std::true_type{}); else mfma_phase(std::false_type{});

      if (last) {                            // inverse transform Y = A^T M A in registers; stored after the next barrier
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          // on register pairs (f32x2): adds AND subtracts then compile to one v_pk_add_f32 per pair (a 4-wide subtract
          // would be four scalar v_sub_f32)
          auto inv = [&](auto half, auto put) {             // half(v): the register pair to work on; put(px, y): store it
            f32x2 s[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
              const f32x2 m0 = half(acc[xi * 4 + 0][j]), m1 = half(acc[xi * 4 + 1][j]);
              const f32x2 m2 = half(acc[xi * 4 + 2][j]), m3 = half(acc[xi * 4 + 3][j]);
              s[xi][0] = m0 + m1 + m2;
              s[xi][1] = m1 - (m2 + m3);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              put(0 * 2 + b, s[0][b] + s[1][b] + s[2][b]);
              put(1 * 2 + b, s[1][b] - (s[2][b] + s[3][b]));
            }
          };
          f32x4 ov[4];
          inv([](const f32x4& v) { return (f32x2)v.lo; }, [&](int px, f32x2 y) { ov[px].lo = y; });
          inv([](const f32x4& v) { return (f32x2)v.hi; }, [&](int px, f32x2 y) { ov[px].hi = y; });
#pragma unroll
          for (int px = 0; px < 4; ++px) parkW[(px * NT + j) * 64] = ov[px];     // lane-contiguous 16-byte slots: conflict-free
        }
        pending = true; ptp = cur;
      }
      ubuf ^= 1;
    }
#ifdef SQD_WINO_STAMP
    stamp_tiles += 1;
#endif
    if (!more) break;
    tile += tstride;
    cur = nxt;
  }
  if (pending) flush(ptp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SQD_WINO_STAMP
  if (sqd_wino_dbg && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* d = sqd_wino_dbg + (long long)blockIdx.x * 8;
    d[0] = hw; d[1] = xcc; d[2] = stamp_t0; d[3] = wall_clock64(); d[4] = stamp_tiles; d[5] = n0;
  }
#endif
#endif
}

static int wino_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// ---------------------------------------------------------------------------------------------------------------------
// Deep-prefetch variant (cfg ids 4..7): the same decomposition and arithmetic (bit-identical results), but the staging
// runs TWO K chunks ahead of the matrix work instead of one.  Measured on the kernel above (scratch/diag/run_wino_diag.py,
// profiles/r02c_wino_ablation.log): with the MFMAs compiled out a chunk still takes 1.3-2.2 us -- one memory round trip
// of the LDS-DMA under load -- against 0.85-1.7 us of matrix work per chunk, so with a prefetch distance of one chunk
// every stage barrier waits for memory.  Here:
//   * U lives in a ring of THREE buffers, the wave-private raw patch in a ring of TWO; chunk k+2 is requested while
//     chunk k is multiplied: its U slice right behind the stage barrier (U[(k+2)%3] = U[(k-1)%3] is free from there on),
//     its patch as soon as the transform has read chunk k's (same buffer);
//   * the stage wait is a COUNTED s_waitcnt vmcnt(N) that leaves the younger chunk's DMA (and a finished tile's stores)
//     in flight -- vector-memory operations retire in issue order (MI355X_MICROARCH.md) -- and the stage barrier is a raw
//     s_barrier (a __syncthreads() would drain the DMA queue with vmcnt(0));
//   * the LDS for the extra buffers comes from dropping the output parking area: a finished tile is inverse-transformed
//     and stored straight from registers behind its last chunk; with counted waits those stores never block a stage.
// 4 waves x (2 x 4 KB patch) + 3 x 16 KB U = 80 KB: still two workgroups per CU.
// USTAT = true (cfg ids 8..11, layers with C <= 64): the slice's WHOLE transformed weight set (C/8 chunks) is fetched into LDS
// once per workgroup and stays there, so the chunk loop needs neither U requests nor a workgroup barrier -- the patch ring
// is wave-private -- and the waves of a workgroup run free of each other: the transform of one overlaps the matrix work
// of its SIMD neighbour instead of all waves meeting at a barrier 2..8 times per tile.
// E1 = true: the workgroup is an expand1x1 slice of the fused Fire launch (sqd_fire_wino_fwd).  A 1x1 convolution in the
// Winograd domain only touches the four inner positions (1,1), (1,2), (2,1), (2,2) (U = +-0.25 w there, 0 elsewhere), so such a
// slice carries 128 channels x 4 positions instead of 32 channels x 16 positions: the same 32 accumulators, the same 64
// MFMAs and the same 16 KB of U per chunk -- "virtual" position p' = 4 q + r, block j stands for position q, channel block
// 2 r + j -- and only the operand selection, the bias injection and the inverse transform differ.
template <int NT, int WV, bool USTAT, bool E1>
__device__ __forceinline__ void wino_pipe_body(const WinoArgs& a) {
  static_assert(!E1 || NT == 2, "expand1x1 slices are built on the 32-channel tiling");
  constexpr int NTHR = WV * 64;
  constexpr int BN = 16 * NT;
  constexpr int RP = 113;
  constexpr int RAW_IT = 4;
  constexpr int USLOTS = 32 * BN;
  static_assert(USLOTS % NTHR == 0, "U slice must be whole workgroup passes");
  constexpr int U_IT = USLOTS / NTHR;
  constexpr int NDMA = RAW_IT + (USTAT ? 0 : U_IT);      // DMA instructions per wave and chunk
  constexpr int NST = 4 * NT;                 // store instructions of a whole tile's plain epilogue
  constexpr int NSTEP = 8;
  constexpr int RAW_STEPS = 4;                // the patch refill is spread over the first MFMA steps
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const rawB = smem;                                 // [2][WV][256][4]
  float* const UB = rawB + 2 * WV * 256 * 4;                // [3][USLOTS][4] ring, or (USTAT) [C/8][USLOTS][4] resident

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int wgq = (int)blockIdx.x >> 3;
  const int n0 = (wgq % a.nslices) * BN;
  const int tstride = a.gx;
  const int nchunks = a.C >> 3;
  const int ntiles = a.ntiles;
  // tile stream of this workgroup: the streams of ONE XCD (workgroup ids equal mod 8) cover a CONTIGUOUS run of super-groups --
  // row-groups that are vertical neighbours share 2 of the 6 patch rows, and with the streams dealt round-robin (stream = 8 q + xcd,
  // rounds 1 and 2) every XCD's L2 fetched those rows for itself: 125.7 MB of fabric traffic per launch against 70 MB algorithmic.
  // The channel slices of a stream stay XCD-adjacent as before (consecutive wgq).
  int tile = ((int)blockIdx.x & 7) * (a.gx >> 3) + wgq / a.nslices;
  if (tile >= ntiles) return;
  const int wv_s = __builtin_amdgcn_readfirstlane(wv);

  constexpr unsigned OOB = 0x80000000u;
  int r_offB[RAW_IT], r_key[RAW_IT];
#pragma unroll
  for (int it = 0; it < RAW_IT; ++it) {
    const int slot = it * 64 + lane;
    const int kq = slot / RP, pix = slot - kq * RP;
    const bool real = kq < 2 && pix < 108;
    const int r = pix / 18, c = pix - r * 18;
    r_key[it] = real ? (r << 8 | c) : -1;
    r_offB[it] = real ? ((r * a.W + c) * a.x_pitch + 4 * kq) * 4 : 0;
  }
  int u_offB[U_IT];
#pragma unroll
  for (int it = 0; it < U_IT; ++it) {
    const int slot = it * NTHR + tid;
    const int pp = slot / (4 * BN), rem = slot - pp * (4 * BN);
    u_offB[it] = ((pp * a.Npad + n0) * 16 + rem * 4) * 4;
  }
  const unsigned u_chunkB = 16u * a.Npad * 8u * 4u;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 0x7ffffff0, 0x00020000);

  struct GPos { int y0, x0, inner, valid; long long p0; unsigned soff; };
  auto group_pos = [&](int t) {
    GPos gp;
    int q = t * WV + wv_s;
    gp.valid = (int)((unsigned)(q - a.ngroups) >> 31);
    q = gp.valid ? q : a.ngroups - 1;
    const int q1 = a.gxn_m ? (int)__umulhi((unsigned)q, a.gxn_m) : q;        // q / gxn (exact: q * gxn < 2^32, host-checked; magic 0 = divisor 1)
    const int gxi = q - q1 * a.gxn;
    const int b = a.gyn_m ? (int)__umulhi((unsigned)q1, a.gyn_m) : q1;       // q1 / gyn
    const int gyi = q1 - b * a.gyn;
    gp.y0 = gyi * 4; gp.x0 = gxi * 16;
    gp.p0 = ((long long)b * a.H + gp.y0) * a.W + gp.x0;
    gp.soff = (unsigned)(gp.p0 * a.x_pitch * 4);
    gp.inner = (int)(((unsigned)(-gp.y0) & (unsigned)(gp.y0 + 4 - a.H) & (unsigned)(-gp.x0) & (unsigned)(gp.x0 + 16 - a.W)) >> 31);
    return gp;
  };
  int r_offG[RAW_IT];                                        // per-lane patch offsets of the group at the PREFETCH cursor
  auto group_offsets = [&](int y0, int x0, int inner) {
    if (inner) {
#pragma unroll
      for (int it = 0; it < RAW_IT; ++it) r_offG[it] = r_offB[it];
      return;
    }
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
      const int key = r_key[it];
      const bool ok = key >= 0 && (unsigned)(y0 + (key >> 8) - 1) < (unsigned)a.H && (unsigned)(x0 + (key & 255) - 1) < (unsigned)a.W;
      r_offG[it] = ok ? r_offB[it] : (int)OOB;
    }
  };
  float* const rawW = rawB + wv_s * 256 * 4;                 // this wave's patch buffer 0 (buffer 1: + WV*256*4 floats)
  auto dma_raw_one = [&](int it, unsigned soff, int rb) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_w_t)(rawW + (rb * WV * 256 + it * 64) * 4), 16, r_offG[it], (int)soff, 0, 0);
  };
  auto dma_u_one = [&](int it, int cc, int ub) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ures, (lds_ptr_w_t)(UB + (ub * USLOTS + it * NTHR + wv_s * 64) * 4), 16, u_offB[it],
                                             (int)(cc * u_chunkB), 0, 0);
  };

  f32x4 acc[16][NT];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 biasv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + 4 * g;
    biasv[j] = (!E1 && a.bias && n < a.N) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // expand1x1 slice: its first channel, and the bias as MFMA operands -- one extra rank-1 product per accumulator of
  // position (1,1) at the start of a tile, A[n][k] = bias[n] for k = 0, B[k][tile] = 1 for k = 0: nine registers instead of
  // the 32 a per-lane copy of the slice's 128 biases would take
  const int e1_c0 = E1 ? (n0 - a.nslices3 * BN) * 4 : 0;
  float e1_biasA[E1 ? 8 : 1];
  const float e1_oneB = (g == 0) ? 1.f : 0.f;
  if constexpr (E1) {
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      const int ch = e1_c0 + blk * 16 + lr;
      e1_biasA[blk] = (g == 0 && a.bias1 && ch < a.N1) ? a.bias1[ch] : 0.f;
    }
  }
  const int ty = lr >> 3, tx = lr & 7;
  // output addressing through buffer resources too (base = the slice's first channel of y; wave-uniform SGPR byte offset
  // of the tile + the lane's 32-bit byte offset + an immediate): no 64-bit per-lane address arithmetic is kept alive
  // across the chunk loop (eight hoisted 64-bit offsets spilled the first version into scratch)
  int o_offB[4];
#pragma unroll
  for (int px = 0; px < 4; ++px) o_offB[px] = (((2 * ty + (px >> 1)) * a.W + 2 * tx + (px & 1)) * a.y_pitch + 4 * g) * 4;
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)(E1 ? a.y + a.y_coff1 + e1_c0 : a.y + a.y_coff + n0), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mulres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ymul ? a.ymul + a.y_coff + n0 : a.y), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t maskres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ymask ? a.ymask + a.y_coff + n0 : a.y), 0, 0x7ffffff0, 0x00020000);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  // A 16-byte MUBUF store whose soffset is an SGPR, directly followed by a VALU write of its first data register, stored
  // the NEW value in the last four lanes of every 16-lane row on this chip (found as rare wrong outputs, always component
  // 0 of tile columns 12..15; the compiler only pads this write-after-read hazard when soffset is NOT a register): two
  // wait states behind every such store, pinned in place.
  auto store16 = [&](f32x4 v, __amdgpu_buffer_rsrc_t res, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), res, voff, soff, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();
  const int tt = lr, cp = g;
  const int rawL_off = (((cp >> 1) * RP + (2 * (tt >> 3)) * 18 + 2 * (tt & 7)) * 4 + 2 * (cp & 1));
  const float* const uR0 = UB + g * 64 + lr * 4;
  const int acc_i = a.accumulate, has_mul = a.ymul != nullptr, has_mask = a.ymask != nullptr;
  const bool plain_epi = !acc_i && !has_mul && !has_mask;

  // ---- prefetch cursor: two chunks ahead of the compute cursor; clamps on this workgroup's last tile (the two stages
  // requested past the end re-read its chunks 0 / 1 into idle buffers and are retired by the final vmcnt(0)) ----
  GPos cur = group_pos(tile);
  int ptile = tile, pcc = 0;
  unsigned psoff = cur.soff;
  group_offsets(cur.y0, cur.x0, cur.inner);
  auto advance = [&]() {
    ++pcc;
    if (pcc == nchunks) {
      pcc = 0;
      ptile = (ptile + tstride < ntiles) ? ptile + tstride : ptile;
      const GPos pf = group_pos(ptile);
      psoff = pf.soff;
      group_offsets(pf.y0, pf.x0, pf.inner);
    }
  };
  if (USTAT) {                            // the whole slice of U, once; published by the only barrier of the kernel
    for (int c = 0; c < nchunks; ++c)
#pragma unroll
      for (int it = 0; it < U_IT; ++it) dma_u_one(it, c, c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // prologue: chunks 0 and 1
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (!USTAT) {
#pragma unroll
      for (int it = 0; it < U_IT; ++it) dma_u_one(it, pcc, s);
    }
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) dma_raw_one(it, psoff + (unsigned)pcc * 32u, s);
    advance();
  }
  int ub = 0, rb = 0;                    // ring slots of the chunk about to be computed
  int stores_behind = 0;                 // store instructions the previous chunk issued behind its DMA requests (0 = unknown)

  for (;;) {
    const int more_i = (int)((unsigned)(tile + tstride - ntiles) >> 31);
    const bool more = more_i != 0;
    for (int cc = 0; cc < nchunks; ++cc) {
      // chunk k's DMA (this wave's share) has landed; chunk k+1's requests -- and a just-finished tile's stores, all younger
      // -- stay in flight.  Vector-memory operations retire in issue order, so "all but the N youngest" is exact; when
      // the store count of the previous chunk is not known (border tiles, read-modify-write epilogues) the smaller N is
      // merely conservative.
      {
        const int sb = __builtin_amdgcn_readfirstlane(stores_behind);
        if (sb == NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NST) : "memory");
        else if (E1 && sb == 32) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + 32) : "memory");
        else if (E1 && sb == 16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + 16) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");      // unknown / other counts: the smaller N is merely conservative
      }
      if (!USTAT) __builtin_amdgcn_s_barrier();      // every wave's share of U[ub] is in LDS; every wave has left U[(ub+2)%3]
      asm volatile("" ::: "memory");
      stores_behind = 0;
      const int last_i = 1 - (int)((unsigned)(cc + 1 - nchunks) >> 31);
      const bool last = last_i != 0;
      const int ub2 = (ub >= 1) ? ub - 1 : 2;                   // (ub + 2) % 3
      // U slice of chunk k+2: its ring slot is free from the barrier on
      if (!USTAT) {
#pragma unroll
        for (int it = 0; it < U_IT; ++it) dma_u_one(it, pcc, ub2);
      }
      const unsigned nsoff = psoff + (unsigned)pcc * 32u;

      // ---- input transform of chunk k from patch buffer rb ----
      f32x2 vv[16];
      if constexpr (E1) {
        // only the four inner positions are needed: rows 1, 2 x columns 1, 2 of the tile's 4x4 patch
        const float* const rawL = rawW + rb * WV * 256 * 4 + rawL_off;
        const f32x2 d11 = *(const f32x2*)(rawL + (1 * 18 + 1) * 4), d12 = *(const f32x2*)(rawL + (1 * 18 + 2) * 4);
        const f32x2 d21 = *(const f32x2*)(rawL + (2 * 18 + 1) * 4), d22 = *(const f32x2*)(rawL + (2 * 18 + 2) * 4);
        const f32x2 t11 = d11 + d21, t12 = d12 + d22, t21 = d21 - d11, t22 = d22 - d12;      // column transform rows 1, 2
        vv[5] = t11 + t12; vv[6] = t12 - t11; vv[9] = t21 + t22; vv[10] = t22 - t21;
      } else {
        const float* const rawL = rawW + rb * WV * 256 * 4 + rawL_off;
        f32x2 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 d0 = *(const f32x2*)(rawL + (0 * 18 + j) * 4), d1 = *(const f32x2*)(rawL + (1 * 18 + j) * 4);
          const f32x2 d2 = *(const f32x2*)(rawL + (2 * 18 + j) * 4), d3 = *(const f32x2*)(rawL + (3 * 18 + j) * 4);
          t[0][j] = d0 - d2; t[1][j] = d1 + d2; t[2][j] = d2 - d1; t[3][j] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vv[i * 4 + 0] = t[i][0] - t[i][2]; vv[i * 4 + 1] = t[i][1] + t[i][2];
          vv[i * 4 + 2] = t[i][2] - t[i][1]; vv[i * 4 + 3] = t[i][1] - t[i][3];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the patch reads have returned: buffer rb may be refilled
      __builtin_amdgcn_sched_barrier(0);
      const float* const uR = uR0 + (USTAT ? cc : ub) * USLOTS * 4;

      auto load_ops = [&](int step, f32x4 (&afr)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) afr[j] = *(const f32x4*)(uR + (step * NT + j) * 256);
      };
      auto mfma_phase = [&](auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        auto mfma_pos = [&](int step, const f32x4 (&afr)[NT], int h) {
          const int p = 2 * step + h;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              // expand1x1 slice: virtual position p = 4 q + r uses the transformed input of inner position q
              const int pv = E1 ? ((p >> 2) == 0 ? 5 : (p >> 2) == 1 ? 6 : (p >> 2) == 2 ? 9 : 10) : p;
              f32x4 c0v;
              if (FIRST && t == 0) {
                if constexpr (E1) {
                  c0v = (f32x4){0.f, 0.f, 0.f, 0.f};
                  if (p < 4) c0v = mfma16(e1_biasA[(p & 3) * 2 + j], e1_oneB, c0v);      // bias enters all four outputs through m11
                } else {
                  c0v = (p == 5) ? biasv[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
              } else {
                c0v = acc[p][j];
              }
              acc[p][j] = mfma16(afr[j][2 * h + t], vv[pv][t], c0v);
            }
        };
        f32x4 af0[NT], af1[NT];
        load_ops(0, af0);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
          for (int q = 0; q < RAW_IT; ++q)
            if (q * RAW_STEPS / RAW_IT == step) dma_raw_one(q, nsoff, rb);       // patch of chunk k+2 into the buffer just read
          if (step & 1) {
            mfma_pos(step, af1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < NSTEP) load_ops(step + 1, af0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_pos(step, af1, 1);
          } else {
            mfma_pos(step, af0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < NSTEP) load_ops(step + 1, af1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_pos(step, af0, 1);
          }
        }
      };
      if (cc == 0) mfma_phase(std::true_type{}); else mfma_phase(std::false_type{});
      advance();                             // the prefetch cursor moves on (per-lane offsets change only at a tile switch)

      if (last && cur.valid) {               // inverse transform Y = A^T M A on register pairs, stored straight away
        const int ysoff = (int)(unsigned)(cur.p0 * a.y_pitch * 4);            // host-checked < 4 GiB
        if constexpr (E1) {
          // M is non-zero only at the inner 2x2: y00 = m11+m12+m21+m22, y01 = m11-m12+m21-m22, y10 = m11+m12-m21-m22,
          // y11 = m11-m12-m21+m22; block blk = 2 r + j lives in accumulators [4 q + r][j], q = 0..3
          const bool wholexy = cur.y0 + 4 <= a.H && cur.x0 + 16 <= a.W;
          int nst = 0;
#pragma unroll
          for (int blk = 0; blk < 8; ++blk) {
            if (e1_c0 + blk * 16 >= a.N1) continue;                          // (uniform) partial last slice
            const f32x4 m0 = acc[0 + (blk >> 1)][blk & 1], m1 = acc[4 + (blk >> 1)][blk & 1];
            const f32x4 m2 = acc[8 + (blk >> 1)][blk & 1], m3 = acc[12 + (blk >> 1)][blk & 1];
            const f32x4 s01 = m0 + m1, d01 = m0 - m1, s23 = m2 + m3, d23 = m2 - m3;
            const f32x4 ov[4] = {s01 + s23, d01 + d23, s01 - s23, d01 - d23};
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              const bool valid = wholexy || (cur.y0 + 2 * ty + (px >> 1) < a.H && cur.x0 + 2 * tx + (px & 1) < a.W);
              if (!valid || e1_c0 + blk * 16 + 4 * g >= a.N1) continue;
              store16(wino_relu4(ov[px], relu_lo), yres, o_offB[px] + blk * 64, ysoff);
            }
            nst += 4;
          }
          stores_behind = (wholexy && (a.N1 & 15) == 0) ? nst : 0;
        } else {
        const bool whole = cur.y0 + 4 <= a.H && cur.x0 + 16 <= a.W && n0 + BN <= a.N;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          auto inv = [&](auto half, auto put) {
            f32x2 sx[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
              const f32x2 m0 = half(acc[xi * 4 + 0][j]), m1 = half(acc[xi * 4 + 1][j]);
              const f32x2 m2 = half(acc[xi * 4 + 2][j]), m3 = half(acc[xi * 4 + 3][j]);
              sx[xi][0] = m0 + m1 + m2;
              sx[xi][1] = m1 - (m2 + m3);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              put(0 * 2 + b, sx[0][b] + sx[1][b] + sx[2][b]);
              put(1 * 2 + b, sx[1][b] - (sx[2][b] + sx[3][b]));
            }
          };
          f32x4 ov[4];
          inv([](const f32x4& v) { return (f32x2)v.lo; }, [&](int px, f32x2 y) { ov[px].lo = y; });
          inv([](const f32x4& v) { return (f32x2)v.hi; }, [&](int px, f32x2 y) { ov[px].hi = y; });
          if (whole && plain_epi) {
#pragma unroll
            for (int px = 0; px < 4; ++px)
              store16(wino_relu4(ov[px], relu_lo), yres, o_offB[px] + j * 64, ysoff);
          } else {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              const bool valid = cur.y0 + 2 * ty + (px >> 1) < a.H && cur.x0 + 2 * tx + (px & 1) < a.W;
              if (!valid || n0 + j * 16 + 4 * g >= a.N) continue;
              const int off = o_offB[px] + j * 64;
              f32x4 v = ov[px];
              if (acc_i) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, off, ysoff, 0));
              if (has_mul) v *= __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mulres, off, ysoff, 0));
              if (has_mask) {
                const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(maskres, off, ysoff, 0));
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
              }
              store16(wino_relu4(v, relu_lo), yres, off, ysoff);
            }
          }
        }
        stores_behind = (whole && plain_epi) ? NST : 0;
        }
      }
      ub = (ub == 2) ? 0 : ub + 1;
      rb ^= 1;
    }
    if (!more) break;
    tile += tstride;
    cur = group_pos(tile);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may still be in flight when the LDS is released
}

template <int NT, int WV, bool USTAT, bool FIRE>
__global__ __launch_bounds__(WV * 64, (WV == 4 && NT <= 2) ? 2 : 1) void conv_wino_pipe_kernel(WinoArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (FIRE) {
    if ((((int)blockIdx.x >> 3) % a.nslices) >= a.nslices3) { wino_pipe_body<NT, WV, USTAT, true>(a); return; }
  }
  wino_pipe_body<NT, WV, USTAT, false>(a);
#endif
}

template <int NT, int WV, bool USTAT = false, bool FIRE = false>
static int launch_wino_pipe(WinoArgs a, hipStream_t stream) {
  constexpr int BN = 16 * NT, NTHR = WV * 64;
  constexpr size_t lds_ring = (size_t)(2 * WV * 256 * 4 + 3 * 32 * BN * 4) * sizeof(float);
  static_assert(lds_ring <= 160 * 1024, "LDS budget");
  // U-stationary: the patch ring + C/8 chunks of the slice's U
  const size_t lds = USTAT ? (size_t)(2 * WV * 256 * 4 + (a.C >> 3) * 32 * BN * 4) * sizeof(float) : lds_ring;
  if (lds > 160 * 1024) return SQD_ERR_UNSUPPORTED;
  auto kern = conv_wino_pipe_kernel<NT, WV, USTAT, FIRE>;
  // 32-bit SGPR byte offset of a tile's output origin / per-lane byte offsets inside a group (buffer-resource stores)
  if ((long long)a.B * a.H * a.W * a.y_pitch * 4 >= (1ll << 32) - (1ll << 30)) return SQD_ERR_UNSUPPORTED;
  static SqdDevOnce attr_once;                 // (per device: ADVICE round 4)
  if (int rc_attr = sqd_max_lds_once(attr_once, (const void*)kern, 160 * 1024)) return rc_attr;
  int nb = 0;              // occupancy depends on the (C-dependent) LDS size of the stationary variant: ask per launch
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NTHR, lds) != hipSuccess || nb < 1) nb = 1;
  const int wgs_per_cu = nb > 4 ? 4 : nb;
  a.gxn = sqd_cdiv(a.W, 16); a.gyn = sqd_cdiv(a.H, 4);
  a.ngroups = a.B * a.gxn * a.gyn;
  if ((long long)(a.ngroups + 8) * (a.gxn > a.gyn ? a.gxn : a.gyn) >= (1ll << 32)) return SQD_ERR_UNSUPPORTED;
  a.gxn_m = a.gxn > 1 ? (unsigned)(((1ull << 32) + a.gxn - 1) / a.gxn) : 0u; a.gyn_m = a.gyn > 1 ? (unsigned)(((1ull << 32) + a.gyn - 1) / a.gyn) : 0u;
  a.ntiles = sqd_cdiv(a.ngroups, WV);
  // fused Fire launch: every 32-wide slice of the packed (virtual) channel axis is a workgroup stream -- first the
  // expand3x3 slices, then the expand1x1 slices (128 real channels each)
  const int nslices = FIRE ? a.Npad / BN : sqd_cdiv(a.N, BN);
  if (nslices * BN > a.Npad) return SQD_ERR_BAD_ARG;
  if (!FIRE) a.nslices3 = nslices;
  const int slots = wino_num_cus() * ((a.wg_cap > 0 && a.wg_cap < wgs_per_cu) ? a.wg_cap : wgs_per_cu);
  int gx_max = slots / nslices; if (gx_max < 1) gx_max = 1;
  const int per_wg = sqd_cdiv(a.ntiles, gx_max);
  const int gx = (sqd_cdiv(a.ntiles, per_wg) + 7) & ~7;
  a.nslices = nslices; a.gx = gx;
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx * nslices)), dim3(NTHR), lds, stream, a);
  return sqd_launch_status();
}

#include "wino_bridge.h"
#include "wino_poolbridge.h"

template <int NT, int WV>
static int launch_wino(WinoArgs a, hipStream_t stream) {
  constexpr int BN = 16 * NT, NTHR = WV * 64;
  constexpr size_t lds = (size_t)(WV * 256 * 4 + WV * 1024 * NT + 2 * 32 * BN * 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv_wino_kernel<NT, WV>;
  static int wgs_per_cu = 0;
  static SqdDevOnce lds_once;
  if (lds > 64 * 1024 && sqd_max_lds_once(lds_once, (const void*)kern, (int)lds) != SQD_OK) return SQD_ERR_LAUNCH;
  if (wgs_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NTHR, lds) != hipSuccess || nb < 1) nb = 1;
    wgs_per_cu = nb > 4 ? 4 : nb;
  }
  a.gxn = sqd_cdiv(a.W, 16); a.gyn = sqd_cdiv(a.H, 4);
  a.ngroups = a.B * a.gxn * a.gyn;
  if ((long long)(a.ngroups + 8) * (a.gxn > a.gyn ? a.gxn : a.gyn) >= (1ll << 32)) return SQD_ERR_UNSUPPORTED;
  a.gxn_m = a.gxn > 1 ? (unsigned)(((1ull << 32) + a.gxn - 1) / a.gxn) : 0u; a.gyn_m = a.gyn > 1 ? (unsigned)(((1ull << 32) + a.gyn - 1) / a.gyn) : 0u;
  a.ntiles = sqd_cdiv(a.ngroups, WV);
  const int nslices = sqd_cdiv(a.N, BN);
  if (nslices * BN > a.Npad) return SQD_ERR_BAD_ARG;
  const int slots = wino_num_cus() * ((a.wg_cap > 0 && a.wg_cap < wgs_per_cu) ? a.wg_cap : wgs_per_cu);
  int gx_max = slots / nslices; if (gx_max < 1) gx_max = 1;
  const int per_wg = sqd_cdiv(a.ntiles, gx_max);
  const int gx = (sqd_cdiv(a.ntiles, per_wg) + 7) & ~7;
  a.nslices = nslices; a.gx = gx;
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx * nslices)), dim3(NTHR), lds, stream, a);
  return sqd_launch_status();
}

// U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: one thread per (chunk, n, channel) writes its 16 positions
__device__ __forceinline__ void wino_pack_one(const float* __restrict__ w, float* __restrict__ u, int No, int Ci, int Npad, int dgrad,
                                              long long idx) {
  const int N = dgrad ? Ci : No;
  const int c8 = (int)(idx & 7);
  const int n = (int)((idx >> 3) % Npad);
  const int chunk = (int)((idx >> 3) / Npad);
  const int c = chunk * 8 + c8;
  float gk[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      gk[r][q] = (n >= N) ? 0.f : (dgrad ? w[(((long long)c * Ci + n) * 3 + (2 - r)) * 3 + (2 - q)] : w[(((long long)n * Ci + c) * 3 + r) * 3 + q]);
  float t[4][3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    t[0][q] = gk[0][q];
    t[1][q] = 0.5f * (gk[0][q] + gk[1][q] + gk[2][q]);
    t[2][q] = 0.5f * (gk[0][q] - gk[1][q] + gk[2][q]);
    t[3][q] = gk[2][q];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float o0 = t[i][0], o1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), o2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), o3 = t[i][2];
    // [chunk][position pair][Npad/16][channel pair][16 n][position parity][2 channels]
    float* dst = u + (((((long long)chunk * 8 + i * 2) * (Npad >> 4) + (n >> 4)) * 4 + (c8 >> 1)) * 16 + (n & 15)) * 4 + (c8 & 1);
    const long long pps = (long long)Npad * 16;                  // floats per position pair
    dst[0] = o0; dst[2] = o1; dst[pps] = o2; dst[pps + 2] = o3;
  }
}

__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ u, int No, int Ci, int Npad, int dgrad) {
  const int C = dgrad ? No : Ci;
  const long long total = (long long)(C >> 3) * Npad * 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) wino_pack_one(w, u, No, Ci, Npad, dgrad, idx);
}

// Batched re-transform after an optimizer step: one launch for every Winograd plan.  descs: device array of n records
// of 7 int64 {w ptr, out ptr, No, Ci, Npad, dgrad, total threads = C/8 * Npad * 8}.
struct WinoPackDesc { const float* w; float* out; long long No, Ci, Npad, dgrad, total; };

__global__ __launch_bounds__(256) void pack_wino_batched_kernel(const WinoPackDesc* __restrict__ descs) {
  const WinoPackDesc d = descs[blockIdx.y];
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < d.total; idx += (long long)gridDim.x * blockDim.x)
    wino_pack_one(d.w, d.out, (int)d.No, (int)d.Ci, (int)d.Npad, (int)d.dgrad, idx);
}

extern "C" int sqd_pack_wino_weights_batched(const void* descs_dev, int n, int blocks_per_desc, void* stream) {
  SQD_CHECK_ARG(descs_dev && n > 0 && n <= 65535 && blocks_per_desc > 0 && blocks_per_desc <= 4096);
  hipLaunchKernelGGL(pack_wino_batched_kernel, dim3((unsigned)blocks_per_desc, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     (const WinoPackDesc*)descs_dev);
  return sqd_launch_status();
}

extern "C" int sqd_pack_wino_weight(const float* w_oihw, float* u_packed, int No, int Ci, int Npad, int dgrad, void* stream) {
  SQD_CHECK_ARG(w_oihw && u_packed && No > 0 && Ci > 0);
  const int N = dgrad ? Ci : No, C = dgrad ? No : Ci;
  SQD_CHECK_ARG(C % 8 == 0 && Npad >= N && Npad % 16 == 0);
  const long long total = (long long)(C >> 3) * Npad * 8;
  hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, u_packed, No, Ci, Npad, dgrad);
  return sqd_launch_status();
}

// ---- fused Fire expand in Winograd form (sqd_fire_wino_fwd) ----
// Packed weights: the expand3x3 set as above in channels [0, Npad3) of a tensor Npad_total = Npad3 + Npad1v channels
// wide, followed by the expand1x1 weights as "virtual" channels: slice s (32 virtual = 128 real channels), virtual position
// p' = 4 q + r, virtual block j, row n  <->  inner position q in ((1,1), (1,2), (2,1), (2,2)), real channel
// 128 s + (2 r + j) 16 + n, value G g G^T at that position = +0.25 w, -0.25 w, -0.25 w, +0.25 w.
__global__ void pack_wino_e1_kernel(const float* __restrict__ w1, float* __restrict__ u, int N1, int C, int Npad3, int Npad_total) {
  const int nv = Npad_total - Npad3;                         // virtual channels of the expand1x1 part
  const long long total = (long long)(C >> 3) * nv * 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx & 7);
  const int nvi = (int)((idx >> 3) % nv);
  const int chunk = (int)((idx >> 3) / nv);
  const int c = chunk * 8 + c8;
  const int s = nvi >> 5, j = (nvi >> 4) & 1, n = nvi & 15;
  const int nvirt = Npad3 + nvi;
  const long long pps = (long long)Npad_total * 16;            // floats per position pair
#pragma unroll
  for (int pv = 0; pv < 16; ++pv) {
    const int q = pv >> 2, r = pv & 3;
    const int ch = 128 * s + (2 * r + j) * 16 + n;
    const float sgn = (q == 0 || q == 3) ? 0.25f : -0.25f;
    const float val = (ch < N1) ? sgn * w1[(long long)ch * C + c] : 0.f;
    float* dst = u + (((((long long)chunk * 8 + (pv >> 1)) * (Npad_total >> 4) + (nvirt >> 4)) * 4 + (c8 >> 1)) * 16 + (nvirt & 15)) * 4 + (c8 & 1) + 2 * (pv & 1);
    (void)pps;
    *dst = val;
  }
}

// w3 [N3][C][3][3], w1 [N1][C][1][1] (the two checkpoint tensors of a Fire's expand pair) -> u [C/8][8][Npad_total/16][4][16][2][2]
// with Npad_total = ceil32(N3) + 32 * ceil(N1 / 128).
extern "C" int sqd_pack_wino_fire(const float* w3_oihw, const float* w1_oihw, float* u_packed, int N3, int N1, int C, int Npad_total,
                                  void* stream) {
  SQD_CHECK_ARG(w3_oihw && w1_oihw && u_packed && N3 > 0 && N1 > 0 && C > 0 && C % 8 == 0);
  const int Npad3 = sqd_cdiv(N3, 32) * 32, nv = sqd_cdiv(N1, 128) * 32;
  SQD_CHECK_ARG(Npad_total == Npad3 + nv);
  int rc = sqd_pack_wino_weight(w3_oihw, u_packed, N3, C, Npad_total, 0, stream);      // (zero-fills every channel >= N3)
  if (rc != SQD_OK) return rc;
  const long long total = (long long)(C >> 3) * nv * 8;
  hipLaunchKernelGGL(pack_wino_e1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1_oihw, u_packed,
                     N1, C, Npad3, Npad_total);
  return sqd_launch_status();
}

// Batched "scaled gather" re-pack: every operand of the Fire bridges besides the expand3x3 transform is a copy of a parameter element
// (the squeeze weights as MFMA A operands, the bias tables) or +-0.25 times one (the expand1x1 weights at the four inner transform
// positions), at a position that only depends on the layer's shape.  The host derives the index map once (plans.FireBridgePlan);
// after an optimizer step ONE launch refreshes the operands of every bridge in place.  descs: n records of 7 int64 {dst, idx, count,
// src0, src1, src2, src3}; idx[i]: -1 -> dst[i] = 0, -2 -> dst[i] left alone, else bits 0..25 = element of source (bits 26..27),
// bits 28..29 = scale (0: 1, 1: +0.25, 2: -0.25).
struct GatherPackDesc { float* dst; const int* idx; long long n; const float* src[4]; };

__global__ __launch_bounds__(256) void gather_pack_batched_kernel(const GatherPackDesc* __restrict__ descs) {
  const GatherPackDesc d = descs[blockIdx.y];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (long long)gridDim.x * blockDim.x) {
    const int code = d.idx[i];
    if (code == -2) continue;
    float v = 0.f;
    if (code >= 0) {
      const int sc = (code >> 28) & 3;
      v = d.src[(code >> 26) & 3][code & 0x3ffffff];
      v = sc == 1 ? 0.25f * v : (sc == 2 ? -0.25f * v : v);
    }
    d.dst[i] = v;
  }
}

extern "C" int sqd_gather_pack_batched(const void* descs_dev, int n, int blocks_per_desc, void* stream) {
  SQD_CHECK_ARG(descs_dev && n > 0 && n <= 65535 && blocks_per_desc > 0 && blocks_per_desc <= 4096);
  hipLaunchKernelGGL(gather_pack_batched_kernel, dim3((unsigned)blocks_per_desc, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     (const GatherPackDesc*)descs_dev);
  return sqd_launch_status();
}

// Winograd configurations: cfg_id -> (channel blocks per slice NT, waves per workgroup WV)
struct WinoCfg { int nt, wv; };
// ids 4..7: the same tilings on the deep-prefetch kernel (conv_wino_pipe_kernel); 8..11: its U-stationary, barrier-free
// form (the slice's whole U must fit the LDS next to the patch ring: C <= 64 at 32-channel slices, else "unsupported")
static const WinoCfg kWinoCfgs[] = {{2, 8}, {1, 8}, {2, 4}, {1, 4}, {2, 8}, {1, 8}, {2, 4}, {1, 4}, {2, 8}, {1, 8}, {2, 4}, {1, 4}};
static const int kNumWinoCfgs = (int)(sizeof(kWinoCfgs) / sizeof(kWinoCfgs[0]));

extern "C" int sqd_wino_num_cfgs() { return kNumWinoCfgs; }

extern "C" int sqd_wino_cfg_info(int cfg_id, int* bn, int* waves) {
  if (cfg_id < 0 || cfg_id >= kNumWinoCfgs) return SQD_ERR_BAD_ARG;
  if (bn) *bn = 16 * kWinoCfgs[cfg_id].nt;
  if (waves) *waves = kWinoCfgs[cfg_id].wv;
  return SQD_OK;
}

extern "C" int sqd_conv_wino_fwd(const float* x, const float* u_packed, const float* bias, float* y, const float* ymask,
                                 const float* ymul, int B, int H, int W, int C, int x_pitch, int x_coff, int N, int Npad,
                                 int y_pitch, int y_coff, int relu, int accumulate, int cfg_id, void* stream) {
  SQD_CHECK_ARG(x && u_packed && y);
  SQD_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && N > 0);
  SQD_CHECK_ARG(C % 8 == 0 && N % 4 == 0 && Npad >= N);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + N <= y_pitch);
  SQD_CHECK_ARG((long long)W * 6 * (x_pitch > y_pitch ? x_pitch : y_pitch) * 4 < (1ll << 30));   // per-lane byte offsets inside a group
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));                   // 32-bit SGPR byte offset of a group origin
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad * 8 * 4 < (1ll << 32));
  const int cap = cfg_id / 1000; cfg_id %= 1000;
  SQD_CHECK_ARG(cfg_id >= 0 && cfg_id < kNumWinoCfgs);
  WinoArgs a{};
  a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = relu; a.wg_cap = cap;
  a.accumulate = accumulate; a.ymask = ymask; a.ymul = ymul;
  hipStream_t s = (hipStream_t)stream;
  switch (cfg_id) {
    case 0: return launch_wino<2, 8>(a, s);
    case 1: return launch_wino<1, 8>(a, s);
    case 2: return launch_wino<2, 4>(a, s);
    case 3: return launch_wino<1, 4>(a, s);
    // (ids 4..7, the deep-prefetch streamed-U forms, are retired: no measured row ever selected them -- profiles/r02d, r02f)
    case 8: return launch_wino_pipe<2, 8, true>(a, s);
    case 9: return launch_wino_pipe<1, 8, true>(a, s);
    case 10: return launch_wino_pipe<2, 4, true>(a, s);
    case 11: return launch_wino_pipe<1, 4, true>(a, s);
  }
  return SQD_ERR_UNSUPPORTED;
}

// Fire.forward's two expand convolutions + torch.cat (src/model/squeezedet.py:18-22) in ONE Winograd launch: y[..., y_coff3 :
// y_coff3 + N3] = ReLU(conv3x3(x) + b3), y[..., y_coff1 : y_coff1 + N1] = ReLU(conv1x1(x) + b1).  u_packed from
// sqd_pack_wino_fire; cfg_id: a 32-channel-slice id of the deep-prefetch family (4, 6: streamed U; 8, 10: U-stationary, C <= 64
// at 4 waves / C <= 32 at 8; 12: C <= 16, the group's transformed input stays in registers, U resident, eight waves),
// + 1000 k = workgroups-per-CU cap.  Plain epilogue: bias + ReLU.
extern "C" int sqd_fire_wino_fwd(const float* x, const float* u_packed, const float* bias3, const float* bias1, float* y, int B, int H,
                                 int W, int C, int x_pitch, int x_coff, int N3, int y_coff3, int N1, int y_coff1, int Npad_total,
                                 int y_pitch, int cfg_id, void* stream) {
  SQD_CHECK_ARG(x && u_packed && y && B > 0 && H > 0 && W > 0 && C > 0 && N3 > 0 && N1 > 0);
  SQD_CHECK_ARG(C % 8 == 0 && N3 % 4 == 0 && N1 % 16 == 0);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff3 % 4 == 0 && y_coff1 % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff3 >= 0 && y_coff3 + N3 <= y_pitch && y_coff1 >= 0 && y_coff1 + N1 <= y_pitch);
  SQD_CHECK_ARG(Npad_total == sqd_cdiv(N3, 32) * 32 + sqd_cdiv(N1, 128) * 32);
  SQD_CHECK_ARG((long long)W * 6 * (x_pitch > y_pitch ? x_pitch : y_pitch) * 4 < (1ll << 30));
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad_total * 8 * 4 < (1ll << 32));
  const int cap = cfg_id / 1000; cfg_id %= 1000;
  WinoArgs a{};
  a.x = x; a.u = u_packed; a.bias = bias3; a.bias1 = bias1; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N3; a.Npad = Npad_total; a.y_pitch = y_pitch; a.y_coff = y_coff3; a.relu = 1; a.wg_cap = cap;
  a.N1 = N1; a.y_coff1 = y_coff1; a.nslices3 = sqd_cdiv(N3, 32);
  hipStream_t s = (hipStream_t)stream;
  switch (cfg_id) {
    case 8: return launch_wino_pipe<2, 8, true, true>(a, s);
    case 10: return launch_wino_pipe<2, 4, true, true>(a, s);
    case 12: return launch_wino_bridge16<1, false>(a, s);      // C <= 16: transformed input in registers, 16-wide passes (wino_bridge.h)
  }
  return SQD_ERR_UNSUPPORTED;
}

// Fire k's two expand convolutions + torch.cat + Fire k+1's squeeze convolution (src/model/squeezedet.py:18-22, twice) in ONE
// launch: y[..., y_coff : y_coff + Nsq] = ReLU(Wsq . cat(ReLU(conv1x1(x) + b1), ReLU(conv3x3(x) + b3)) + bsq); the
// concatenated expand output is never written (inference forward).  u_packed from sqd_pack_wino_fire (Npad_total wide);
// bias_tab [Npad_total/32][8][16]: per pass (32-wide slice of the packed axis) the biases of its channel blocks (expand3x3
// slice: blocks 0, 1; expand1x1 slice: 8 blocks of its 128 channels; 0 where padded); sq_ops [blocks][4][ceil(Nsq/16)][64]:
// the squeeze weights as MFMA A operands, blocks in pass order, value for lane (lr, g) at (block, t, q) =
// Wsq[16 q + lr][channel(block) + 4 g + t] (0 where padded).  cfg_id 10: U resident in LDS (small C), 6: streamed through the
// ring; four waves per workgroup, one workgroup per CU (the wave owns 512 registers: 128 + 16..32 accumulators).
// cfg_id 12 (C <= 16): 16-wide passes, the group's transformed input stays in registers, eight waves; its bias_tab is
// [passes][4][16] and its sq_ops blocks follow ITS pass order: one block per expand3x3 pass (2 ceil(N3/32) passes of 16
// channels), then four per expand1x1 pass s1 (channels 128 (s1 >> 1) + (2 r + (s1 & 1)) 16 + n, r = 0..3).
extern "C" int sqd_fire_bridge_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                                   float* y, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1, int Npad_total,
                                   int Nsq, int y_pitch, int y_coff, int cfg_id, void* stream) {
  SQD_CHECK_ARG(x && u_packed && bias_tab && sq_ops && y && B > 0 && H > 0 && W > 0 && C > 0 && N3 > 0 && N1 > 0 && Nsq > 0);
  SQD_CHECK_ARG(C % 8 == 0 && N3 % 4 == 0 && N1 % 16 == 0 && Nsq % 4 == 0 && Nsq <= 32);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + Nsq <= y_pitch);
  SQD_CHECK_ARG(Npad_total == sqd_cdiv(N3, 32) * 32 + sqd_cdiv(N1, 128) * 32);
  SQD_CHECK_ARG((long long)W * 6 * (x_pitch > y_pitch ? x_pitch : y_pitch) * 4 < (1ll << 30));
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad_total * 8 * 4 < (1ll << 32));
  const int cap = cfg_id / 1000; cfg_id %= 1000;
  WinoArgs a{};
  a.x = x; a.u = u_packed; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N3; a.Npad = Npad_total; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = 1; a.wg_cap = cap;
  a.N1 = N1; a.nslices3 = sqd_cdiv(N3, 32);
  a.br_w = sq_ops; a.br_bias = bias_tab; a.br_sqb = sq_bias; a.br_nsq = Nsq;
  hipStream_t s = (hipStream_t)stream;
  const bool one = Nsq <= 16;
  switch (cfg_id) {
    case 10: return one ? launch_wino_bridge<4, true, 1>(a, s) : launch_wino_bridge<4, true, 2>(a, s);
    case 12: return one ? launch_wino_bridge16<1>(a, s) : launch_wino_bridge16<2>(a, s);
  }
  return SQD_ERR_UNSUPPORTED;
}

// Training form of sqd_fire_bridge_fwd (cfg 12 only: C <= 16): additionally stores the concatenated expand output -- `save`
// [B][H][W][save_pitch], expand1x1 at save_coff1, expand3x3 at save_coff3 -- which the backward reads (input of the next squeeze's
// weight gradient, ReLU masks).  The next Fire's squeeze then has no launch of its own and the 128-channel tensor is not read back.
extern "C" int sqd_fire_bridge_save_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                                        float* y, float* save, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1,
                                        int Npad_total, int Nsq, int y_pitch, int y_coff, int save_pitch, int save_coff3, int save_coff1,
                                        void* stream) {
  SQD_CHECK_ARG(x && u_packed && bias_tab && sq_ops && y && save && B > 0 && H > 0 && W > 0 && C > 0 && N3 > 0 && N1 > 0 && Nsq > 0);
  SQD_CHECK_ARG(C % 8 == 0 && C <= 16 && N3 % 4 == 0 && N1 % 16 == 0 && Nsq % 4 == 0 && Nsq <= 32);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(save_pitch % 4 == 0 && save_coff3 % 4 == 0 && save_coff1 % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + Nsq <= y_pitch);
  SQD_CHECK_ARG(save_coff3 >= 0 && save_coff3 + N3 <= save_pitch && save_coff1 >= 0 && save_coff1 + N1 <= save_pitch);
  SQD_CHECK_ARG(save_coff3 + N3 <= save_coff1 || save_coff1 + N1 <= save_coff3);
  SQD_CHECK_ARG(Npad_total == sqd_cdiv(N3, 32) * 32 + sqd_cdiv(N1, 128) * 32);
  const int widest = x_pitch > y_pitch ? (x_pitch > save_pitch ? x_pitch : save_pitch) : (y_pitch > save_pitch ? y_pitch : save_pitch);
  SQD_CHECK_ARG((long long)W * 6 * widest * 4 < (1ll << 30));
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad_total * 8 * 4 < (1ll << 32));
  WinoArgs a{};
  a.x = x; a.u = u_packed; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N3; a.Npad = Npad_total; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = 1;
  a.N1 = N1; a.nslices3 = sqd_cdiv(N3, 32);
  a.br_w = sq_ops; a.br_bias = bias_tab; a.br_sqb = sq_bias; a.br_nsq = Nsq;
  a.sv = save; a.sv_pitch = save_pitch; a.sv_coff = save_coff3; a.sv_coff1 = save_coff1;
  hipStream_t s = (hipStream_t)stream;
  return Nsq <= 16 ? launch_wino_bridge16<1, 2>(a, s) : launch_wino_bridge16<2, 2>(a, s);
}

// Fire k's expand pair + torch.cat + MaxPool2d(3, 2, ceil_mode=True) + Fire k+1's squeeze (src/model/squeezedet.py:18-22, 47-52) in
// ONE launch: y [B][Hp][Wp][y_pitch] window [y_coff, +Nsq) = ReLU(Wsq . maxpool(cat(ReLU(conv1x1(x) + b1), ReLU(conv3x3(x) + b3))) + bsq).
// Operands as for sqd_fire_bridge_fwd cfg 12 (16-wide passes) except that an expand1x1 pass contributes two channel blocks
// (N1 <= 64, N3 <= 64, C <= 16, Nsq <= 32).  Hp x Wp must be the pool's output size for H x W.  nseg: segments a column
// strip is cut into (parallelism vs. one recomputed group row per segment).
static int fire_pool_bridge_impl(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                                 float* y, float* save, unsigned char* codes, int save_pitch, int save_coff3, int save_coff1, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1, int Npad_total,
                                        int Nsq, int Hp, int Wp, int y_pitch, int y_coff, int nseg, void* stream) {
  SQD_CHECK_ARG(x && u_packed && bias_tab && sq_ops && y && B > 0 && H >= 3 && W >= 3 && C > 0 && N3 > 0 && N1 > 0 && Nsq > 0);
  SQD_CHECK_ARG(C % 8 == 0 && N3 % 4 == 0 && N1 % 16 == 0 && Nsq % 4 == 0 && Nsq <= 32);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + Nsq <= y_pitch);
  SQD_CHECK_ARG(Npad_total == sqd_cdiv(N3, 32) * 32 + sqd_cdiv(N1, 128) * 32);
  int hp = (H - 2) / 2 + 1, wp = (W - 2) / 2 + 1;            // ceil((n - 3) / 2) + 1, last window must start inside the map
  if ((hp - 1) * 2 >= H) --hp;
  if ((wp - 1) * 2 >= W) --wp;
  SQD_CHECK_ARG(Hp == hp && Wp == wp);
  SQD_CHECK_ARG((long long)W * 6 * x_pitch * 4 < (1ll << 30) && (long long)(Wp + 8) * 2 * y_pitch * 4 < (1ll << 30));
  SQD_CHECK_ARG((long long)B * (H + 8) * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)B * (Hp + 2) * Wp * y_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad_total * 8 * 4 < (1ll << 32));
  WinoArgs a{};
  a.x = x; a.u = u_packed; a.y = y;
  a.B = B; a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N3; a.Npad = Npad_total; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = 1;
  a.N1 = N1;
  a.br_w = sq_ops; a.br_bias = bias_tab; a.br_sqb = sq_bias; a.br_nsq = Nsq;
  a.pb_hp = Hp; a.pb_wp = Wp;
  hipStream_t s = (hipStream_t)stream;
  if (save || codes) {
    SQD_CHECK_ARG(save && codes && ((uintptr_t)codes & 3) == 0);
    SQD_CHECK_ARG(save_pitch % 4 == 0 && save_coff3 % 4 == 0 && save_coff1 % 4 == 0);
    SQD_CHECK_ARG(save_coff3 >= 0 && save_coff3 + N3 <= save_pitch && save_coff1 >= 0 && save_coff1 + N1 <= save_pitch);
    SQD_CHECK_ARG(save_coff3 + N3 <= save_coff1 || save_coff1 + N1 <= save_coff3);
    SQD_CHECK_ARG((long long)(Wp + 8) * 2 * save_pitch * 4 < (1ll << 30));
    SQD_CHECK_ARG((long long)B * (Hp + 2) * Wp * save_pitch * 4 < (1ll << 32) - (1ll << 30));
    a.sv = save; a.sv_codes = codes; a.sv_pitch = save_pitch; a.sv_coff = save_coff3; a.sv_coff1 = save_coff1;
    return Nsq <= 16 ? launch_wino_poolbridge16<1, true>(a, nseg, s) : launch_wino_poolbridge16<2, true>(a, nseg, s);
  }
  return Nsq <= 16 ? launch_wino_poolbridge16<1>(a, nseg, s) : launch_wino_poolbridge16<2>(a, nseg, s);
}

extern "C" int sqd_fire_pool_bridge_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                                        float* y, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1, int Npad_total,
                                        int Nsq, int Hp, int Wp, int y_pitch, int y_coff, int nseg, void* stream) {
  return fire_pool_bridge_impl(x, u_packed, bias_tab, sq_ops, sq_bias, y, nullptr, nullptr, 0, 0, 0, B, H, W, C, x_pitch, x_coff, N3, N1,
                               Npad_total, Nsq, Hp, Wp, y_pitch, y_coff, nseg, stream);
}

// Training form: additionally stores the POOLED expand output (save [B][Hp][Wp][save_pitch], expand1x1 at save_coff1, expand3x3 at
// save_coff3) and the pool's arg-max / ReLU codes (codes: one byte per element of the same geometry, first window position 3 dy + dx
// holding the pooled value, 15 where it is not > 0) -- everything the backward reads of this stage; the unpooled expand output, the
// max-pool launch and the next squeeze's launch are gone.
extern "C" int sqd_fire_pool_bridge_save_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops,
                                             const float* sq_bias, float* y, float* save, unsigned char* codes, int B, int H, int W, int C,
                                             int x_pitch, int x_coff, int N3, int N1, int Npad_total, int Nsq, int Hp, int Wp, int y_pitch,
                                             int y_coff, int save_pitch, int save_coff3, int save_coff1, int nseg, void* stream) {
  SQD_CHECK_ARG(save && codes);
  return fire_pool_bridge_impl(x, u_packed, bias_tab, sq_ops, sq_bias, y, save, codes, save_pitch, save_coff3, save_coff1, B, H, W, C, x_pitch,
                               x_coff, N3, N1, Npad_total, Nsq, Hp, Wp, y_pitch, y_coff, nseg, stream);
}
