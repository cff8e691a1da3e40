// Winograd F(2x2,3x3) convolution with a BALANCED (stream-K) work split -- sqd_conv_wino_sk_fwd.
//
// Same arithmetic, operand packing (sqd_pack_wino_weight) and per-stage structure as conv_wino_kernel<2,4> (conv_wino.hip: four
// waves sharing the transformed weights U of one channel slice through LDS, one K chunk of 8 channels per stage, the 6x18-pixel
// patch of a 4x16-pixel group by wave-private LDS-DMA, input transform in registers, 64 MFMAs per wave and stage).  What differs
// is WHICH stages a workgroup runs.  conv_wino_kernel hands out whole (super-group, slice) units: at 24x78x20 pixels that is
// 450 units (ConvDet, reference src/model/squeezedet.py:73-75,83: N = 72 as three 32-wide slices) or 900 / 1800 units (Fire
// expand3x3, :14,20-22) for 512 resident workgroups -- the busiest SIMD carries 12-33 % more than the average -- and the third
// ConvDet slice multiplies 24 zero channels.  Here:
//   * two workgroup CLASSES with the same stage weight (64 MFMAs per wave, 128 accumulator registers, 80 KB of LDS):
//       F: a wave owns ONE group x 32 channels (two 16-channel blocks)   -- every full 32-channel slice;
//       H: a wave owns TWO groups x 16 channels (one block)              -- the last slice when only its lower 16 channels
//          exist (N = 72: channels 64..79); the two groups are transformed and multiplied one after the other from one U stage.
//     N = 72 therefore executes 80 channels' worth of MFMAs instead of 96;
//   * the flattened (unit, K chunk) list of each class is cut into one contiguous, equally long range per workgroup of that
//     class (host: sqd_wino_sk_schedule; the classes share the grid in proportion to their stage counts);
//   * a unit cut by a range boundary is finished by several workgroups: every part inverse-transforms its partial accumulators
//     (the transform is linear; the bias rides in the part that owns K chunk 0), stores them to a workspace slab write-through
//     (sc1), drains, and draws a ticket from the unit's per-wave counter; the wave that draws the last ticket adds the slabs IN
//     PART ORDER (bitwise reproducible whatever the arrival order), applies the epilogue and stores the tile
//     (cdna_hip_programming.md section 5 "in-launch split-K reduction" / Guideline 16; no wave ever waits for another workgroup,
//     so the kernel cannot deadlock whatever the residency).  The last arriver returns the counter to zero.
#include "sqd_common.h"
#include <type_traits>
#include <algorithm>
#include <stdlib.h>

struct SkSeg { int tile, n0, c0, c1, cls, nparts, part, slab0; };      // 32 bytes; host layout = int32[8]

struct WinoSkArgs {
  const float* x; const float* u; const float* bias; float* y;
  int H, W;
  int C, x_pitch, x_coff;
  int N, Npad, y_pitch, y_coff;
  int relu, accumulate;
  const float* ymask; const float* ymul;     // epilogue (same pitch / channel offset as y): multiply by ymul, zero where ymask <= 0
  float yscale;                              // epilogue: multiply by a constant (1 = off)
  const unsigned long long* drop_state;      // epilogue: counter-based dropout of the output (sqd_common.h), {seed, step} or null
  int drop_keep; float drop_scale;
  unsigned long long* drop_advance;          // != null: one lane of the launch adds 1 to drop_advance[1] (the forward's mask is consumed)
  int gxn, gyn;
  unsigned gxn_m, gyn_m;
  int ngroups;
  const int* seg_off;                        // [grid + 1]
  const SkSeg* segs;
  float* ws;                                 // partial slabs: [slab][4 waves][2048 floats]
  unsigned* cnt;                             // [slab][4 waves] arrival counters (zero between launches)
};

typedef __attribute__((address_space(3))) void* lds_ptr_sk_t;
typedef unsigned int u32x4_sk __attribute__((ext_vector_type(4)));

// A segment record, read into SGPRs: the kernel stores to global memory between these loads, so the compiler would otherwise keep the
// (wave-uniform) values -- and every group origin / DMA offset derived from them -- in vector registers.
__device__ __forceinline__ SkSeg sk_load_seg(const SkSeg* p) {
  const int* q = (const int*)p;
  SkSeg s;
  s.tile = __builtin_amdgcn_readfirstlane(q[0]); s.n0 = __builtin_amdgcn_readfirstlane(q[1]);
  s.c0 = __builtin_amdgcn_readfirstlane(q[2]); s.c1 = __builtin_amdgcn_readfirstlane(q[3]);
  s.cls = __builtin_amdgcn_readfirstlane(q[4]); s.nparts = __builtin_amdgcn_readfirstlane(q[5]);
  s.part = __builtin_amdgcn_readfirstlane(q[6]); s.slab0 = __builtin_amdgcn_readfirstlane(q[7]);
  return s;
}

__device__ __forceinline__ f32x4 sk_relu4(f32x4 v, float lo) {
  asm volatile("v_max_f32 %0, %4, %0\n\tv_max_f32 %1, %4, %1\n\tv_max_f32 %2, %4, %2\n\tv_max_f32 %3, %4, %3"
               : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "s"(lo));
  return v;
}

// NT 16-channel blocks per slice, GW groups per wave (NT * GW == 2).  FULL: the epilogue options of the backward and of training
// (accumulate, ymul, yscale, dropout, ymask); the plain instantiation (bias + optional ReLU: every inference launch) keeps them out
// of its register budget.
template <int NT, int GW, bool FULL>
__device__ __forceinline__ void wino_sk_body(const WinoSkArgs& a, int sidx, const int send) {
  static_assert(NT * GW == 2, "128 accumulator registers per wave");
  constexpr int WV = 4;
  constexpr int NTHR = WV * 64;
  constexpr int BN = 16 * NT;
  constexpr int RP = 113;                     // slots per k-quad plane of a raw patch (108 used)
  constexpr int RAW_IT = 4;                   // 256 slots per patch
  constexpr int USLOTS = 32 * BN;
  constexpr int U_IT = USLOTS / NTHR;         // 4 (F) / 2 (H)
  constexpr int U_PH = U_IT / GW;             // U requests per group phase: 4 / 1
  constexpr int NSTEP = 8;                    // MFMA steps of 2 positions per group phase
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const rawB = smem;                                 // [WV][GW][256][4]
  float* const VB = rawB + WV * GW * 256 * 4;               // [WV][GW][4 px][NT][64 lanes] f32x4: finished tiles, parked (8 KB per wave)
  float* const UB = VB + WV * 2048;                         // [2][USLOTS][4]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
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
  int u_offB[U_IT];                                           // (slice origin n0 and K chunk ride in the SGPR offset)
#pragma unroll
  for (int it = 0; it < U_IT; ++it) {
    const int slot = it * NTHR + tid;
    const int pp = slot / (4 * BN), rem = slot - pp * (4 * BN);
    u_offB[it] = (pp * a.Npad * 16 + rem * 4) * 4;
  }
  const unsigned u_chunkB = 16u * a.Npad * 8u * 4u;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 0x7ffffff0, 0x00020000);

  struct GPos { int y0, x0, inner, valid; long long p0; unsigned soff; };
  auto group_pos = [&](int t, int gw) {                      // group gw of this wave in super-group t (all wave-uniform)
    GPos gp;
    int q = (t * WV + wv_s) * GW + gw;
    gp.valid = (int)((unsigned)(q - a.ngroups) >> 31);
    q = gp.valid ? q : a.ngroups - 1;                        // idle slots of the last super-group redo the last group (never stored)
    const int q1 = a.gxn_m ? (int)__umulhi((unsigned)q, a.gxn_m) : q;
    const int gxi = q - q1 * a.gxn;
    const int b = a.gyn_m ? (int)__umulhi((unsigned)q1, a.gyn_m) : q1;
    const int gyi = q1 - b * a.gyn;
    gp.y0 = gyi * 4; gp.x0 = gxi * 16;
    gp.p0 = ((long long)b * a.H + gp.y0) * a.W + gp.x0;
    gp.soff = (unsigned)(gp.p0 * a.x_pitch * 4);
    gp.inner = (int)(((unsigned)(-gp.y0) & (unsigned)(gp.y0 + 4 - a.H) & (unsigned)(-gp.x0) & (unsigned)(gp.x0 + 16 - a.W)) >> 31);
    return gp;
  };
  int r_offG[GW][RAW_IT];                                     // per-lane patch offsets of the groups whose stages are being fetched
  auto group_offsets = [&](int gw, const GPos gp) {
    if (gp.inner) {
#pragma unroll
      for (int it = 0; it < RAW_IT; ++it) r_offG[gw][it] = r_offB[it];
      return;
    }
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
      const int key = r_key[it];
      const bool ok = key >= 0 && (unsigned)(gp.y0 + (key >> 8) - 1) < (unsigned)a.H && (unsigned)(gp.x0 + (key & 255) - 1) < (unsigned)a.W;
      r_offG[gw][it] = ok ? r_offB[it] : (int)OOB;
    }
  };
  float* const rawW = rawB + wv_s * GW * 256 * 4;            // this wave's raw patches
  auto dma_raw_one = [&](int gw, int it, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_sk_t)(rawW + (gw * 256 + it * 64) * 4), 16, r_offG[gw][it], (int)soff, 0, 0);
  };
  auto dma_u_one = [&](int it, unsigned usoff, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ures, (lds_ptr_sk_t)(UB + (buf * USLOTS + it * NTHR + wv_s * 64) * 4), 16, u_offB[it],
                                             (int)usoff, 0, 0);
  };

  f32x4 acc[GW][16][NT];
#pragma unroll
  for (int gw = 0; gw < GW; ++gw)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[gw][p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ty = lr >> 3, tx = lr & 7;
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();
  const int tt = lr, cp = g;
  const float* const rawL = rawW + (((cp >> 1) * RP + (2 * (tt >> 3)) * 18 + 2 * (tt & 7)) * 4 + 2 * (cp & 1));
  const float* const uR0 = UB + g * 64 + lr * 4;
  f32x4* const parkW = (f32x4*)(VB + wv_s * 2048) + lane;    // slot k = (gw * 4 + px) * NT + j

  const int acc_i = FULL && a.accumulate, has_mul = FULL && a.ymul != nullptr, has_mask = FULL && a.ymask != nullptr;
  const int has_drop = FULL && a.drop_state != nullptr;
  const float yscale = FULL ? a.yscale : 1.0f;
  const int has_scale = yscale != 1.0f;
  const SqdDrop dropk = has_drop ? sqd_drop_key(a.drop_state[0], a.drop_state[1], a.drop_keep, a.drop_scale) : SqdDrop{0u, 0u, 0u, 0.f};
  auto epi = [&](f32x4 v, float* dst, const float* mul, const float* mask) {
    if constexpr (FULL) {
      if (acc_i) v += *(const f32x4*)dst;                     // (the bias is already inside: accumulator (1,1) started from it)
      if (has_mul) v *= *(const f32x4*)mul;
      if (has_scale) v *= yscale;
      if (has_drop) v *= sqd_drop_mul4((unsigned long long)(dst - a.y) >> 2, dropk);      // element index in the output buffer
      if (has_mask) {
        const f32x4 m = *(const f32x4*)mask;
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
      }
    }
    *(f32x4*)dst = sk_relu4(v, relu_lo);
  };
  // partial slabs: buffer resource; stores and loads carry sc1 (aux 16): write-through, L1-bypassing
  const __amdgpu_buffer_rsrc_t wsres = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, 0, 0x7ffffff0, 0x00020000);

  // The finished segment's tile(s) sit in the parking area (inverse-transformed).  A whole unit is stored straight away; a unit
  // cut into parts goes through three steps, one per stage top, so that no step waits for memory on the stage's critical path:
  //   A: the partial slab is stored (write-through);  B (the stage-top vmcnt(0) has drained those stores): the arrival ticket is
  //   drawn;  C (the ticket has returned): the last arriver adds every part's slab in part order and stores the tile.
  auto store_tile = [&](const SkSeg sg) {                    // epilogue: parking area -> y
    const int n0 = sg.n0;
#pragma unroll
    for (int gw = 0; gw < GW; ++gw) {
      const GPos gp = group_pos(sg.tile, gw);
      if (!gp.valid) continue;
      float* ybase = a.y + gp.p0 * a.y_pitch + a.y_coff + n0;
      const float* mulbase = a.ymul + gp.p0 * a.y_pitch + a.y_coff + n0;     // dereferenced only when present
      const float* maskbase = a.ymask + gp.p0 * a.y_pitch + a.y_coff + n0;
      const bool whole = gp.y0 + 4 <= a.H && gp.x0 + 16 <= a.W && n0 + BN <= a.N;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const bool valid = whole || (gp.y0 + 2 * ty + (px >> 1) < a.H && gp.x0 + 2 * tx + (px & 1) < a.W);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (!valid || (!whole && n0 + j * 16 + 4 * g >= a.N)) continue;
          const int off = ((2 * ty + (px >> 1)) * a.W + 2 * tx + (px & 1)) * a.y_pitch + 4 * g + j * 16;
          epi(parkW[((gw * 4 + px) * NT + j) * 64], ybase + off, mulbase + off, maskbase + off);
        }
      }
    }
  };
  auto step_a = [&](int ps) {                                // -> true: a cut unit, its slab is on the way
    const SkSeg sg = sk_load_seg(a.segs + ps);
    if (sg.nparts <= 1) { store_tile(sg); return false; }
    const unsigned slabB = (unsigned)((sg.slab0 + sg.part) * WV + wv_s) * 8192u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_sk, parkW[k * 64]), wsres, lane * 16 + k * 1024, (int)slabB, 16);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1" ::: "memory");                  // (SGPR-soffset 16-byte store followed by a write of its data register: conv_wino.hip)
      __builtin_amdgcn_sched_barrier(0);
    }
    return true;
  };
  auto step_b = [&](int ps) {                                // this wave's slab has left the chip's caches (vmcnt(0) since): count it
    const int slab0 = __builtin_amdgcn_readfirstlane(((const int*)(a.segs + ps))[7]);
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(a.cnt + (slab0 * WV + wv_s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  auto step_c = [&](int ps, unsigned t) {
    const SkSeg sg = sk_load_seg(a.segs + ps);
    if (__builtin_amdgcn_readfirstlane((int)t) != sg.nparts - 1) return;
    // last arriver of this (unit, wave): every part's slab is complete.  Add them in part order, through the parking area.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane == 0) __hip_atomic_store(a.cnt + (sg.slab0 * WV + wv_s), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int pt = 0; pt < sg.nparts; ++pt) {                 // one slab (8 loads) in flight: the running segment's accumulators are live here
      const unsigned sb = (unsigned)((sg.slab0 + pt) * WV + wv_s) * 8192u;
      f32x4 va[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) va[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsres, lane * 16 + k * 1024, (int)sb, 16));
#pragma unroll
      for (int k = 0; k < 8; ++k) parkW[k * 64] = (pt == 0) ? va[k] : parkW[k * 64] + va[k];
      __builtin_amdgcn_sched_barrier(0);
    }
    store_tile(sg);
  };

  SkSeg seg = sk_load_seg(a.segs + sidx);
  unsigned csoff[GW];                                         // byte offset of the current groups' patch origins
#pragma unroll
  for (int gw = 0; gw < GW; ++gw) {
    const GPos gp = group_pos(seg.tile, gw);
    group_offsets(gw, gp);
    csoff[gw] = gp.soff;
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) dma_raw_one(gw, it, gp.soff + (unsigned)seg.c0 * 32u);
  }
#pragma unroll
  for (int it = 0; it < U_IT; ++it) dma_u_one(it, (unsigned)seg.c0 * u_chunkB + (unsigned)seg.n0 * 64u, 0);
  int ubuf = 0;
  int pend_a = -1, pend_b = -1, pend_c = -1;                 // segments waiting for step A (tile parked) / B (slab stored) / C (ticket drawn)
  unsigned ticket = 0;

  for (;;) {
    // the slice's bias for this lane's channels, entering through accumulator (1,1) of the part that owns K chunk 0
    f32x4 biasv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = seg.n0 + j * 16 + 4 * g;
      biasv[j] = (a.bias && seg.c0 == 0 && n < a.N) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int has_next = (int)((unsigned)(sidx + 1 - send) >> 31);
    for (int cc = seg.c0; cc < seg.c1; ++cc) {
      // this wave's share of the stage's DMA must have LANDED before the barrier publishes it to the other waves
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // all waves left the previous U buffer
      if (pend_c >= 0) { step_c(pend_c, ticket); pend_c = -1; }
      if (pend_b >= 0) { ticket = step_b(pend_b); pend_c = pend_b; pend_b = -1; }
      if (pend_a >= 0) { if (step_a(pend_a)) pend_b = pend_a; pend_a = -1; }
      const int last_i = 1 - (int)((unsigned)(cc + 1 - seg.c1) >> 31);
      const bool last = last_i != 0;
      // the stage fetched during this one: the next K chunk of this segment, or (last chunk) the first stage of the next segment
      // (no next segment: this one's first stage once more, into idle buffers)
      unsigned nsoff[GW], nuoff;
      if (last) {
        const SkSeg ns = sk_load_seg(a.segs + (has_next ? sidx + 1 : sidx));
#pragma unroll
        for (int gw = 0; gw < GW; ++gw) {
          const GPos gp = group_pos(ns.tile, gw);
          group_offsets(gw, gp);
          csoff[gw] = gp.soff;
          nsoff[gw] = gp.soff + (unsigned)ns.c0 * 32u;
        }
        nuoff = (unsigned)ns.c0 * u_chunkB + (unsigned)ns.n0 * 64u;
      } else {
#pragma unroll
        for (int gw = 0; gw < GW; ++gw) nsoff[gw] = csoff[gw] + (unsigned)(cc + 1) * 32u;
        nuoff = (unsigned)(cc + 1) * u_chunkB + (unsigned)seg.n0 * 64u;
      }
      const float* const uR = uR0 + ubuf * USLOTS * 4;
      const bool first = cc == seg.c0;

#pragma unroll
      for (int gw = 0; gw < GW; ++gw) {
        // ---- input transform: V = B^T d B for (tile tt, channels 2cp, 2cp+1) of group gw ----
        f32x2 vv[16];
        {
          const float* const rl = rawL + gw * 256 * 4;
          f32x2 t[4][4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2 d0 = *(const f32x2*)(rl + (0 * 18 + j) * 4), d1 = *(const f32x2*)(rl + (1 * 18 + j) * 4);
            const f32x2 d2 = *(const f32x2*)(rl + (2 * 18 + j) * 4), d3 = *(const f32x2*)(rl + (3 * 18 + j) * 4);
            t[0][j] = d0 - d2; t[1][j] = d1 + d2; t[2][j] = d2 - d1; t[3][j] = d1 - d3;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            vv[i * 4 + 0] = t[i][0] - t[i][2]; vv[i * 4 + 1] = t[i][1] + t[i][2];
            vv[i * 4 + 2] = t[i][2] - t[i][1]; vv[i * 4 + 3] = t[i][1] - t[i][3];
          }
        }
        // The patch buffer is refilled (LDS-DMA, below) for the next stage: its reads above must have returned, and neither
        // the compiler nor the machine scheduler may move a DMA issue across this point.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        // ---- 16 positions x 2 k-steps x NT MFMAs, software-pipelined over two operand sets (steps of 2 positions) ----
        auto mfma_phase = [&](auto first_c) {
          constexpr bool FIRST = decltype(first_c)::value;
          auto load_ops = [&](int step, f32x4 (&afr)[NT]) {
#pragma unroll
            for (int j = 0; j < NT; ++j) afr[j] = *(const f32x4*)(uR + (step * NT + j) * 256);
          };
          auto mfma_pos = [&](int step, const f32x4 (&afr)[NT], int h) {
            if constexpr (NT == 1) {
              // one block per position: call h covers k-step h of BOTH positions of the step, so that two consecutive MFMAs never
              // chain on the same accumulator (the operand quad holds [position parity][k-step])
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const int p = 2 * step + hh;
                const f32x4 c0v = (FIRST && h == 0) ? ((p == 5) ? biasv[0] : (f32x4){0.f, 0.f, 0.f, 0.f}) : acc[gw][p][0];
                acc[gw][p][0] = mfma16(afr[0][2 * hh + h], vv[p][h], c0v);
              }
            } else {
              const int p = 2 * step + h;
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                  const f32x4 c0v = (FIRST && t == 0) ? ((p == 5) ? biasv[j] : (f32x4){0.f, 0.f, 0.f, 0.f}) : acc[gw][p][j];
                  acc[gw][p][j] = mfma16(afr[j][2 * h + t], vv[p][t], c0v);
                }
            }
          };
          f32x4 af0[NT], af1[NT];
          load_ops(0, af0);
#pragma unroll
          for (int step = 0; step < NSTEP; ++step) {
            if (step < RAW_IT) dma_raw_one(gw, step < RAW_IT ? step : 0, nsoff[gw]);
            else if (step - RAW_IT < U_PH) dma_u_one(gw * U_PH + (step - RAW_IT < U_PH ? step - RAW_IT : 0), nuoff, ubuf ^ 1);
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
        if (first) mfma_phase(std::true_type{}); else mfma_phase(std::false_type{});
      }

      if (last) {                            // inverse transform Y = A^T M A in registers; handed on after the next barrier
        if (pend_b >= 0 || pend_c >= 0) {      // (a segment shorter than the three hand-over steps: finish them now,
          if (pend_c >= 0) { step_c(pend_c, ticket); pend_c = -1; }       //  step C adds through the parking area that is about to be rewritten)
          if (pend_b >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ticket = step_b(pend_b); step_c(pend_b, ticket); pend_b = -1;
          }
        }
#pragma unroll
        for (int gw = 0; gw < GW; ++gw)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            auto inv = [&](auto half, auto put) {             // on register pairs: packed adds
              f32x2 s[4][2];
#pragma unroll
              for (int xi = 0; xi < 4; ++xi) {
                const f32x2 m0 = half(acc[gw][xi * 4 + 0][j]), m1 = half(acc[gw][xi * 4 + 1][j]);
                const f32x2 m2 = half(acc[gw][xi * 4 + 2][j]), m3 = half(acc[gw][xi * 4 + 3][j]);
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
            for (int px = 0; px < 4; ++px) parkW[((gw * 4 + px) * NT + j) * 64] = ov[px];     // lane-contiguous 16-byte slots: conflict-free
          }
        pend_a = sidx;
      }
      ubuf ^= 1;
    }
    if (!has_next) break;
    ++sidx;
    seg = sk_load_seg(a.segs + sidx);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the idle refetch: no LDS-DMA may be in flight when the LDS is released)
  if (pend_c >= 0) step_c(pend_c, ticket);
  if (pend_b >= 0) { ticket = step_b(pend_b); step_c(pend_b, ticket); }
  if (pend_a >= 0 && step_a(pend_a)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ticket = step_b(pend_a); step_c(pend_a, ticket);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FULL>
__global__ __launch_bounds__(256, 2) void conv_wino_sk_kernel(WinoSkArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int rk = sqd_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);    // ranges of ONE XCD are neighbours in the unit list
  const int sidx = __builtin_amdgcn_readfirstlane(a.seg_off[rk]);
  const int send = __builtin_amdgcn_readfirstlane(a.seg_off[rk + 1]);
  if (a.drop_advance && blockIdx.x == 0 && threadIdx.x == 0) a.drop_advance[1] += 1ull;     // (nothing in this launch reads it)
  if (sidx >= send) return;
  if (__builtin_amdgcn_readfirstlane(((const int*)(a.segs + sidx))[4])) wino_sk_body<1, 2, FULL>(a, sidx, send);
  else wino_sk_body<2, 1, FULL>(a, sidx, send);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Host: the balanced schedule.  Class F units u = tile4 * nfull + slice over super-groups of 4 groups (the slices of a super-group
// are neighbours in the list: they read the same patches, and neighbouring ranges sit on one XCD); class H units = super-groups of
// 8 groups, present when the last slice has only its lower 16 channels.  Every stage of either class is 64 MFMAs per wave, so a
// class's share of the G workgroups is its share of the stages (h_bias_pm: per-mille correction of the H share, 1000 = none).
// H workgroups are spread evenly over the grid ranks.  Two ways of cutting a class's stages into per-workgroup runs:
//   ksplit == 0  "stream-K": workgroup i of the class owns the contiguous stages [i S / G_c, (i + 1) S / G_c); a cut closer than
//                `minseg` stages to a unit's edge moves to the edge;
//   ksplit >= 1  "aligned split-K": every unit is cut at the SAME K boundaries into ksplit parts; the items (part, unit), part-major,
//                are dealt round-robin to the class's workgroups -- the workgroups of a round then walk the same K chunks at
//                the same time, so the transformed weights they stage are shared through the L2 (what the contiguous cut gives up).
// Pure function of its arguments.  segs: records of 8 ints {tile, n0, c0, c1, class (0 F / 1 H), nparts, part, slab0}.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int sqd_wino_sk_schedule(int ngroups, int N, int C, int G, int minseg, int h_bias_pm, int ksplit, int* seg_off, int* segs,
                                    int max_segs, int* nsegs_out, int* nslabs_out) {
  SQD_CHECK_ARG(ngroups > 0 && N > 0 && N % 4 == 0 && C > 0 && C % 8 == 0 && G > 0 && minseg >= 1 && ksplit >= 0 && ksplit <= 64);
  SQD_CHECK_ARG(seg_off && segs && max_segs > 0 && nsegs_out && nslabs_out);
  const int nchunks = C / 8, nslices = sqd_cdiv(N, 32);
  const bool half_last = (N - 32 * (nslices - 1)) <= 16;
  const int nfull = half_last ? nslices - 1 : nslices;
  const long long tilesF = sqd_cdiv(ngroups, 4), tilesH = half_last ? sqd_cdiv(ngroups, 8) : 0;
  const long long SF = tilesF * nfull * nchunks, SH = tilesH * nchunks;
  int GH = 0;
  if (SH > 0 && SF > 0 && G < 2) return SQD_ERR_BAD_ARG;     // a workgroup runs ONE class
  if (SH > 0) {
    const double share = (double)SH / (double)(SF + SH) * (h_bias_pm > 0 ? h_bias_pm / 1000.0 : 1.0);
    GH = (int)(share * G + 0.5);
    if (GH < 1) GH = 1;
    if (SF > 0 && GH > G - 1) GH = G - 1;
  }
  const int GF = G - GH;
  if (SF > 0 && GF < 1) return SQD_ERR_BAD_ARG;
  if (ksplit > nchunks) ksplit = nchunks;
  struct Cut { long long u; int c; };
  auto cut_at = [&](long long S, long long U, int Gc, int i) {
    if (i >= Gc) return Cut{U, 0};
    const long long target = (S * i + Gc - 1) / Gc;
    long long u = target / nchunks;
    int c = (int)(target - u * nchunks);
    if (c < minseg) c = 0;
    else if (nchunks - c < minseg) { c = 0; ++u; }
    return Cut{u, c};
  };
  int ns = 0, iF = 0, iH = 0;
  for (int r = 0; r < G; ++r) {
    seg_off[r] = ns;
    // rank r is an H workgroup when the running H count steps here (even spread)
    const bool isH = GH > 0 && ((long long)(r + 1) * GH / G) > ((long long)r * GH / G);
    const long long S = isH ? SH : SF, U = isH ? tilesH : tilesF * nfull;
    const int Gc = isH ? GH : GF, i = isH ? iH++ : iF++;
    if (S == 0) continue;
    auto emit = [&](long long u, int c0, int c1) {
      if (ns >= max_segs) return false;
      int* sg = segs + 8 * ns++;
      if (isH) { sg[0] = (int)u; sg[1] = 32 * nfull; }
      else { sg[0] = (int)(u / nfull); sg[1] = (int)(u % nfull) * 32; }
      sg[2] = c0; sg[3] = c1; sg[4] = isH ? 1 : 0; sg[5] = 1; sg[6] = 0; sg[7] = -1;
      return true;
    };
    if (ksplit >= 1) {
      for (long long it = i; it < U * ksplit; it += Gc) {
        const int part = (int)(it / U);
        const long long u = it - (long long)part * U;
        const int c0 = (int)((long long)part * nchunks / ksplit), c1 = (int)((long long)(part + 1) * nchunks / ksplit);
        if (!emit(u, c0, c1)) return SQD_ERR_BAD_ARG;
      }
      continue;
    }
    const Cut b0 = cut_at(S, U, Gc, i), b1 = cut_at(S, U, Gc, i + 1);
    long long u = b0.u; int c = b0.c;
    while (u < b1.u || (u == b1.u && c < b1.c)) {
      if (!emit(u, c, (u == b1.u) ? b1.c : nchunks)) return SQD_ERR_BAD_ARG;
      ++u; c = 0;
    }
  }
  seg_off[G] = ns;
  // parts of cut units: group the records by (class, tile, n0), ordered by first stage
  int* order = (int*)malloc(sizeof(int) * (size_t)(ns > 0 ? ns : 1));
  if (!order) return SQD_ERR_LAUNCH;
  for (int i = 0; i < ns; ++i) order[i] = i;
  auto key_less = [&](int x, int y) {
    const int* p = segs + 8 * x; const int* q = segs + 8 * y;
    if (p[4] != q[4]) return p[4] < q[4];
    if (p[0] != q[0]) return p[0] < q[0];
    if (p[1] != q[1]) return p[1] < q[1];
    return p[2] < q[2];
  };
  std::sort(order, order + ns, key_less);
  int nslabs = 0;
  for (int i = 0; i < ns;) {
    int j = i + 1;
    while (j < ns && segs[8 * order[j] + 4] == segs[8 * order[i] + 4] && segs[8 * order[j]] == segs[8 * order[i]] &&
           segs[8 * order[j] + 1] == segs[8 * order[i] + 1]) ++j;
    const int np = j - i;
    if (np > 1) {
      for (int k = i; k < j; ++k) { int* sg = segs + 8 * order[k]; sg[5] = np; sg[6] = k - i; sg[7] = nslabs; }
      nslabs += np;
    }
    i = j;
  }
  free(order);
  *nsegs_out = ns; *nslabs_out = nslabs;
  return SQD_OK;
}

static int sk_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// Workgroups of one launch: every resident slot of the device (two 80 KB workgroups per CU); the schedule is built for it.
extern "C" int sqd_wino_sk_grid(void) { return 2 * sk_num_cus(); }

// y[..., y_coff : y_coff + N] (=|+=) conv3x3(x[..., x_coff : x_coff + C]) (+ bias) (* ymul) (* yscale) (zero where ymask <= 0) (ReLU).
// u_packed / Npad: sqd_pack_wino_weight with 32-channel padding.  seg_off [G + 1] / segs [nsegs][8]: device copies of what
// sqd_wino_sk_schedule wrote for (B * ceil(H/4) * ceil(W/16), N, C, G); ws: nslabs * 4 * 2048 floats; cnt: nslabs * 4 unsigned,
// zero before the first launch (every launch leaves them zero).  drop_state (DEVICE {seed, step} or NULL) / drop_keep16 / drop_scale:
// counter-based dropout of the output (sqd_common.h; the last Fire's expand3x3 in training mode, reference src/model/squeezedet.py:
// 81-82); drop_advance (or NULL): the launch adds 1 to drop_advance[1] (ConvDet's forward: the mask of this step is consumed).
extern "C" int sqd_conv_wino_sk_fwd(const float* x, const float* u_packed, const float* bias, float* y, const float* ymask, const float* ymul,
                                    float yscale, int B, int H, int W, int C, int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff,
                                    int relu, int accumulate, const int* seg_off, const int* segs, int G, int nslabs, float* ws, unsigned* cnt,
                                    const unsigned long long* drop_state, int drop_keep16, float drop_scale, unsigned long long* drop_advance,
                                    void* stream) {
  SQD_CHECK_ARG(x && u_packed && y && seg_off && segs && G > 0 && nslabs >= 0 && (nslabs == 0 || (ws && cnt)));
  SQD_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && N > 0);
  SQD_CHECK_ARG(C % 8 == 0 && N % 4 == 0 && Npad >= N && Npad % 32 == 0);
  SQD_CHECK_ARG(x_pitch % 4 == 0 && x_coff % 4 == 0 && y_pitch % 4 == 0 && y_coff % 4 == 0);
  SQD_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_pitch && y_coff >= 0 && y_coff + N <= y_pitch);
  SQD_CHECK_ARG((long long)W * 6 * (x_pitch > y_pitch ? x_pitch : y_pitch) * 4 < (1ll << 30));
  SQD_CHECK_ARG((long long)B * H * W * x_pitch * 4 < (1ll << 32) - (1ll << 30));
  SQD_CHECK_ARG((long long)(C >> 3) * 16 * Npad * 8 * 4 < (1ll << 32));
  SQD_CHECK_ARG((long long)nslabs * 4 * 8192 < (1ll << 32) - (1ll << 30));
  WinoSkArgs a{};
  a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
  a.H = H; a.W = W; a.C = C; a.x_pitch = x_pitch; a.x_coff = x_coff;
  a.N = N; a.Npad = Npad; a.y_pitch = y_pitch; a.y_coff = y_coff; a.relu = relu; a.accumulate = accumulate;
  a.ymask = ymask; a.ymul = ymul; a.yscale = yscale;
  SQD_CHECK_ARG(!drop_state || (drop_keep16 >= 0 && drop_keep16 <= 65536));
  a.drop_state = drop_state; a.drop_keep = drop_keep16; a.drop_scale = drop_scale; a.drop_advance = drop_advance;
  a.gxn = sqd_cdiv(W, 16); a.gyn = sqd_cdiv(H, 4);
  a.ngroups = B * a.gxn * a.gyn;
  if ((long long)(a.ngroups + 16) * (a.gxn > a.gyn ? a.gxn : a.gyn) >= (1ll << 32)) return SQD_ERR_UNSUPPORTED;
  a.gxn_m = a.gxn > 1 ? (unsigned)(((1ull << 32) + a.gxn - 1) / a.gxn) : 0u; a.gyn_m = a.gyn > 1 ? (unsigned)(((1ull << 32) + a.gyn - 1) / a.gyn) : 0u;
  a.seg_off = seg_off; a.segs = (const SkSeg*)segs; a.ws = ws; a.cnt = cnt;
  constexpr size_t lds = (size_t)80 * 1024;
  static SqdDevOnce once_plain, once_full;          // the attribute belongs to the device that is current at the call: once per device
  if (int rc_attr = sqd_max_lds_once(once_plain, (const void*)conv_wino_sk_kernel<false>, (int)lds)) return rc_attr;
  if (int rc_attr = sqd_max_lds_once(once_full, (const void*)conv_wino_sk_kernel<true>, (int)lds)) return rc_attr;
  const bool full = accumulate || ymask || ymul || yscale != 1.0f || drop_state;
  if (full) hipLaunchKernelGGL(conv_wino_sk_kernel<true>, dim3((unsigned)G), dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv_wino_sk_kernel<false>, dim3((unsigned)G), dim3(256), lds, (hipStream_t)stream, a);
  return sqd_launch_status();
}
