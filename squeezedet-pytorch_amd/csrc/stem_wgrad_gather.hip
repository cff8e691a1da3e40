// Gather form of the stem weight gradient behind the fused conv + ReLU + max-pool forward (reference: the backward of
// src/model/squeezedet.py:34-36 as autograd derives it).  Its own translation unit because it is built WITHOUT the SLP vectoriser
// (Makefile): left on, it pairs the scalar tap-0 accumulators into 64-bit register tuples, which on gfx950 must be even-aligned,
// and the kernel goes from 171 registers to 256 + 62 spilled.
#include "stem_wgrad.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* wg_lds_ptr_t;

// ---------------------------------------------------------------------------------------------
// Stem weight gradient behind the fused forward, third generation (round 3): a GATHER on the vector ALU instead of a
// dense GEMM on the matrix cores.  The gradient of the conv output is dPool routed through the arg-max: of the 4 conv
// pixels a pooled pixel stands for, ONE per channel is non-zero.  The dense form (stem_wgrad_pooled_kernel above) rebuilds
// that 75 %-zero tensor in LDS (a zero fill, four scatter phases, six workgroup barriers per tile) and multiplies all of
// it: 242 us per batch of 20, a third of its MFMAs for the bias gradient.  Here every (pooled pixel, channel) item is
// ONE 27-tap dot-product update:
//     dW[n][:] += dPool[p][n] * patch(arg-max position of (p, n))[:],   db[n] += dPool[p][n]
//   * lane = (channel pair cp, pixel sub-index): it owns channels 2cp, 2cp+1 for the whole kernel -- 2 x 27 + 2
//     accumulators in registers, no reduction before the end -- and walks its wave's pooled tiles (2 x 8 pixels, two
//     pixels per step across the lane's sub-index; four channels per lane would need 112 accumulators: spills);
//   * the wave's NCHW image patch arrives by 16-byte buffer-resource LDS-DMA in a wave-private double buffer (same patch
//     geometry and zero-filled borders as stem_wave_kernel, stem_pool.hip); the arg-max codes (2 channels per load) and
//     dPool pairs of the NEXT tile are requested into the registers of the pixel group that has just been decoded;
//   * per channel: code -> window origin in the patch (three integer instructions), then per (plane, tap row) one 4-byte
//     and one 8-byte LDS read (taps 1, 2 are 8-byte aligned by construction) feeding one v_fma + one v_pk_fma;
//     code 15 (ReLU mask, or any code > 8) multiplies with 0 from a clamped, valid address;
//   * no workgroup barrier in the tile loop; at the end the two pixel sub-lanes are summed by an xor-shuffle, the four
//     waves through LDS in a fixed order (bitwise reproducible) and the workgroup writes one slab in the layout the
//     shared slab reduction expects.
// 2 flop per (item, tap) = 2.1 GFLOP per batch instead of 8.3 (+ 50 % for the bias tiles); bound by the LDS reads
// (72 per pixel group of a wave) and by the 306 MB of dPool / codes / image it streams.
// Needs Win % 4 == 0 and a 16-byte aligned image (else the launcher keeps the dense kernel).
// ---------------------------------------------------------------------------------------------
// OCC = waves per SIMD the register budget is held to (2: 171 registers; 3: 168 with a few spilled)
template <bool HAS_PL, int OCC>
__global__ __launch_bounds__(256, OCC) void stem_wgrad_gather_kernel(StemWgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PH = 2, PW = 8, N = 64, K = 27;
  // patch rows, 16-byte slots / floats per row.  One slot more than the 10 the taps need: with a 44-float pitch the three window rows
  // of a pixel (2 RP dy floats apart) start 0 / 24 / 16 banks apart for the 4-byte reads (32 banks) and 0 / 24 / 48 for the 8-byte
  // reads (64 banks) -- with 40 floats rows 0 and 2 met on the same banks (2-way conflicts of the 4-byte tap-0 reads)
  constexpr int IH = 4 * PH + 3, SL = PW + 3, RP = 4 * SL;
  constexpr int NSLOT = 3 * IH * SL, N_IT = (NSLOT + 63) / 64, BUFF = N_IT * 64 * 4; // floats per buffer
  constexpr int NG = PH * PW / 2;                                                    // pixel groups (2 pixels) per tile
  constexpr unsigned OOB = 0x80000000u;
  static_assert(PW == 8 && NG == 8, "group -> (row, column) split and the item enumeration below assume a 2 x 8 tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) const char* lds_cptr_t;
  typedef __attribute__((address_space(3))) const float* lds_f1_t;
  typedef __attribute__((address_space(3))) const f32x2* lds_f2_t;
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int cp = lane & 31, sub = lane >> 5;           // channels 2 cp, 2 cp + 1; pixel sub-index inside a group
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const bufW = smem + wave_s * (2 * BUFF);

  int d_off[N_IT];                                     // byte offset of the lane's 16-byte slot from the patch origin
#pragma unroll
  for (int it = 0; it < N_IT; ++it) {
    const int slot = it * 64 + lane;
    const int ci = slot / (IH * SL), rem = slot - ci * (IH * SL), row = rem / SL, k4 = rem - row * SL;
    d_off[it] = slot < NSLOT ? ((ci * a.Hin + row) * a.Win + 4 * k4) * 4 : (int)OOB;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.img - (a.Win + 4)), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t cres = __builtin_amdgcn_make_buffer_rsrc((void*)a.amax, 0, a.B * a.Hp * a.Wp * N, 0x00020000);
  const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_PL ? a.pooled : a.dy), 0, 0x7ffffff0, 0x00020000);
  // pixel of group gi for this lane: row gi / 4, column 2 (gi & 3) + sub; element offset of its channel pair from the tile's first pixel
  const int e_lane = sub * N + 2 * cp;                 // + ((gi / 4) * Wp + 2 * (gi & 3)) * N: wave-uniform, goes into the scalar offset
  // window origin of pooled pixel (row 0, column sub) in the patch: tap (0, 0) of window position (0, 0)
  const lds_cptr_t pL = (lds_cptr_t)(bufW + 4 * sub + 3);

  struct Tile { int ty, tx, inner; unsigned soff, eoff; };
  auto tile_at = [&](int t) __attribute__((always_inline)) {
    Tile z;
    const int t1 = t / a.tiles_x;
    z.tx = t - t1 * a.tiles_x;
    const int b = t1 / a.tiles_y;
    z.ty = t1 - b * a.tiles_y;
    const int iy0 = 4 * PH * z.ty - 1, ix0 = 4 * PW * z.tx - 4;
    z.soff = (unsigned)(((b * 3 * a.Hin + 4 * PH * z.ty) * a.Win + 4 * PW * z.tx) * 4);
    z.inner = iy0 >= 0 && iy0 + IH <= a.Hin && ix0 >= 0 && ix0 + RP <= a.Win;
    z.eoff = (unsigned)(((b * a.Hp + z.ty * PH) * a.Wp + z.tx * PW) * N);       // element index of the tile's first pooled pixel
    return z;
  };
  auto dma_in = [&](const Tile z, int buf) __attribute__((always_inline)) {
    const int iy0 = 4 * PH * z.ty - 1, ix0 = 4 * PW * z.tx - 4;
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      int off = d_off[it];
      if (!z.inner) {                                  // uniform: border tile
        const int slot = it * 64 + lane;
        const int ci = slot / (IH * SL), rem = slot - ci * (IH * SL), row = rem / SL, k4 = rem - row * SL;
        const bool ok = slot < NSLOT && (unsigned)(iy0 + row) < (unsigned)a.Hin && (unsigned)(ix0 + 4 * k4) < (unsigned)a.Win;
        off = ok ? off : (int)OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (wg_lds_ptr_t)(bufW + buf * BUFF + it * 256), 16, off, (int)z.soff, 0, 0);
    }
  };
  struct Items { unsigned cw[NG]; f32x2 dp[NG]; f32x2 pl[HAS_PL ? NG : 1]; };
  auto fetch_group = [&](const Tile z, Items& it, auto gic, bool want) __attribute__((always_inline)) {
    constexpr int gi = decltype(gic)::value;
    // branch-free: pixels past the pooled map (partial tiles) and requests behind the last tile go out of range and read zeros
    // (code 0, gradient 0)
    const bool ok = want && z.ty * PH + (gi >> 2) < a.Hp && z.tx * PW + 2 * (gi & 3) + sub < a.Wp;
    // ONE select for both streams (a second one tips the register allocator into ~600 spilled registers): the idle offset 2^29 is
    // past the code resource's range (= the tensor's size, < 2^29 bytes, host-checked) and, times 4, past every range
    const int ec = ok ? e_lane : (int)(OOB >> 2), ef = ec * 4;           // byte offsets into the uint8 codes / the fp32 tensors
    const unsigned ge = z.eoff + (unsigned)(((gi >> 2) * a.Wp + 2 * (gi & 3)) * N);
    it.cw[gi] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(cres, ec, (int)ge, 0);
    it.dp[gi] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, ef, (int)(ge * 4u), 0));
    if constexpr (HAS_PL) it.pl[gi] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(pres, ef, (int)(ge * 4u), 0));
  };

  float acc0[2][9];                                    // [channel][plane * 3 + tap row]: tap column 0
  f32x2 acc12[2][9];                                   //                                  tap columns 1, 2
  float bacc[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    bacc[e] = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) { acc0[e][r] = 0.f; acc12[e][r] = (f32x2){0.f, 0.f}; }
  }

  // One item = (pixel group gi, channel e of the lane's pair): 9 x (4-byte + 8-byte) LDS reads, then 9 x (fma + packed fma).
  // The reads of item i + 1 are issued before the arithmetic of item i (two tap sets); the scheduling barriers keep the
  // compiler from hoisting more reads than that (left alone it front-loads the reads of a whole tile and spills).  The item
  // registers of a pixel group are refilled with the NEXT tile's group as soon as its second channel has been decoded.
  struct Taps { float t0[9]; f32x2 t12[9]; float v; };
  auto consume = [&](Items& it, auto bufc, const Tile nxt, bool has_next) __attribute__((always_inline)) {
    constexpr int BO = decltype(bufc)::value * BUFF * 4;
    auto load_item = [&](Taps& T, auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value, gi = i >> 1, e = i & 1;
      const unsigned code = (it.cw[gi] >> (8 * e)) & 255u;
      bool live = code < 9u;
      if constexpr (HAS_PL) live = live && it.pl[gi][e] > 0.f;
      T.v = live ? it.dp[gi][e] : 0.f;
      const unsigned cc = code < 8u ? code : 8u;                      // masked items read a valid address and multiply by 0
      const unsigned dyw = (cc * 11u) >> 5;                           // cc / 3 for 0..8
      const unsigned rel = (2u * cc + dyw * (2u * RP - 6u)) * 4u;     // window position (dyw, cc - 3 dyw): 2 input rows / columns each
      const lds_cptr_t wp = pL + rel;
      constexpr int gimm = BO + ((4 * (gi >> 2)) * RP + 8 * (gi & 3)) * 4;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int imm = gimm + ((r / 3) * IH + (r % 3)) * RP * 4;
        T.t0[r] = *(lds_f1_t)(wp + imm);
        T.t12[r] = *(lds_f2_t)(wp + imm + 4);
      }
    };
    auto fma_item = [&](const Taps& T, auto ic) __attribute__((always_inline)) {
      constexpr int e = decltype(ic)::value & 1;
      const f32x2 vv = (f32x2){T.v, T.v};
      bacc[e] += T.v;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        acc0[e][r] = __builtin_fmaf(T.v, T.t0[r], acc0[e][r]);
        acc12[e][r] = __builtin_elementwise_fma(vv, T.t12[r], acc12[e][r]);
      }
    };
    Taps TA, TB;
    load_item(TA, std::integral_constant<int, 0>{});
    auto pair = [&](auto ic) __attribute__((always_inline)) {       // the two channels of pixel group i / 2
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_sched_barrier(0);
      load_item(TB, std::integral_constant<int, i + 1>{});
      fetch_group(nxt, it, std::integral_constant<int, (i >> 1)>{}, has_next);
      __builtin_amdgcn_sched_barrier(0);
      fma_item(TA, std::integral_constant<int, i>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i + 2 < 2 * NG) load_item(TA, std::integral_constant<int, i + 2>{});
      __builtin_amdgcn_sched_barrier(0);
      fma_item(TB, std::integral_constant<int, i + 1>{});
    };
    pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 2>{}); pair(std::integral_constant<int, 4>{});
    pair(std::integral_constant<int, 6>{}); pair(std::integral_constant<int, 8>{}); pair(std::integral_constant<int, 10>{});
    pair(std::integral_constant<int, 12>{}); pair(std::integral_constant<int, 14>{});
    __builtin_amdgcn_sched_barrier(0);
  };

  const int ntiles = a.nblocks;
  const int tstride = (int)gridDim.x * 4;
  int tile = sqd_xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * 4 + wave_s;
  if (tile < ntiles) {
    Items itA;
    Tile cur = tile_at(tile);
    dma_in(cur, 0);
    fetch_group(cur, itA, std::integral_constant<int, 0>{}, true); fetch_group(cur, itA, std::integral_constant<int, 1>{}, true);
    fetch_group(cur, itA, std::integral_constant<int, 2>{}, true); fetch_group(cur, itA, std::integral_constant<int, 3>{}, true);
    fetch_group(cur, itA, std::integral_constant<int, 4>{}, true); fetch_group(cur, itA, std::integral_constant<int, 5>{}, true);
    fetch_group(cur, itA, std::integral_constant<int, 6>{}, true); fetch_group(cur, itA, std::integral_constant<int, 7>{}, true);
    auto step = [&](auto bufc) __attribute__((always_inline)) -> bool {     // one tile out of patch buffer bufc; returns false after the last
      constexpr int BUF = decltype(bufc)::value;
      const int ntile = tile + tstride;
      const bool has_next = ntile < ntiles;
      const Tile nxt = tile_at(has_next ? ntile : tile);
      __builtin_amdgcn_sched_barrier(0);
      // this tile's patch (requested one tile ago) has landed; the item loads issued after it may stay in flight -- the compiler
      // waits for each of them (in issue order) where its registers are first read
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * (HAS_PL ? 3 : 2)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) dma_in(nxt, BUF ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      consume(itA, bufc, nxt, has_next);
      tile = ntile; cur = nxt;
      return has_next;
    };
    for (;;) {
      if (!step(std::integral_constant<int, 0>{})) break;
      if (!step(std::integral_constant<int, 1>{})) break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- reduction: the two pixel sub-lanes (xor 32), then the four waves through LDS in a fixed order ----
  __syncthreads();                                      // every wave has left its patch buffers
  float* const red = smem;                              // [4 waves][56][32]
  constexpr int NACC = 2 * 27 + 2;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      float s0 = acc0[e][r], s1 = acc12[e][r].x, s2 = acc12[e][r].y;
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (sub == 0) {
        float* const d = red + (wave_s * NACC + e * 27 + r * 3) * 32 + cp;
        d[0] = s0; d[32] = s1; d[64] = s2;
      }
    }
    float sb = bacc[e];
    sb += __shfl_xor(sb, 32);
    if (sub == 0) red[(wave_s * NACC + 54 + e) * 32 + cp] = sb;
  }
  __syncthreads();
  float* const slab = a.slab + (long long)blockIdx.x * a.slab_stride;
  for (int o = tid; o < N * K + N; o += 256) {
    int n, idx;
    if (o < N * K) { n = o / K; idx = (n & 1) * 27 + (o - n * K); } else { n = o - N * K; idx = 54 + (n & 1); }
    const float* const src = red + idx * 32 + (n >> 1);
    slab[o] = ((src[0] + src[NACC * 32]) + src[2 * NACC * 32]) + src[3 * NACC * 32];
  }
#endif
}

static int stem_wgrad_gather_occ() { return 2; }      // workgroups per CU the launch is sized for (3 measured no faster)

static int stem_wgrad_gather_slabs(const StemWgradArgs& a, int S, int occ) {
  int dev = 0, cus = 256; hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const int ntiles = a.B * sqd_cdiv(a.Hp, 2) * sqd_cdiv(a.Wp, 8);
  int g = sqd_cdiv(ntiles, 4);
  if (g > occ * cus) g = occ * cus;
  return g < S ? g : S;
}

// returns the number of slabs written (= workgroups), or a negative status
int launch_stem_wgrad_gather(StemWgradArgs a, int S, hipStream_t s) {
  constexpr int IH = 11, SL = 11, N_IT = (3 * IH * SL + 63) / 64;
  constexpr size_t lds_patch = (size_t)4 * 2 * N_IT * 64 * 16, lds_red = (size_t)4 * 56 * 32 * 4;
  constexpr size_t lds = lds_patch > lds_red ? lds_patch : lds_red;
  a.tiles_x = sqd_cdiv(a.Wp, 8); a.tiles_y = sqd_cdiv(a.Hp, 2);
  a.nblocks = a.B * a.tiles_x * a.tiles_y;
  const int occ = stem_wgrad_gather_occ();
  const int g = stem_wgrad_gather_slabs(a, S, occ);
  if (a.pooled) hipLaunchKernelGGL((stem_wgrad_gather_kernel<true, 2>), dim3((unsigned)g), dim3(256), lds, s, a);
  else if (occ == 3) hipLaunchKernelGGL((stem_wgrad_gather_kernel<false, 3>), dim3((unsigned)g), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((stem_wgrad_gather_kernel<false, 2>), dim3((unsigned)g), dim3(256), lds, s, a);
  return sqd_launch_status() == SQD_OK ? g : -1;
}

