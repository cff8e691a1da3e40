// Fire -> MaxPool -> Fire bridge for a small squeeze width (included by conv_wino.hip): Fire k's expand pair, the concat, the
// 3x3 / stride-2 / ceil-mode max pool behind it AND Fire k+1's squeeze in one launch (reference: src/model/squeezedet.py:18-22,
// 47-52: features = ..., Fire(128, 16, 64, 64), MaxPool2d(3, 2, ceil_mode=True), Fire(128, 32, 128, 128), ...).  Neither the
// concatenated expand output (306 MB at 1248x384 bs=20) nor the pooled tensor (77 MB) reaches HBM; the launch reads the
// 16-channel squeeze output and writes the next 32-channel squeeze output.  Inference only.
//
// Same machinery as wino_bridge16_body (the group's transformed input in registers, 16-wide channel passes, U resident in LDS,
// eight free-running waves).  What is new is the pool between the inverse transform and the squeeze product:
//   * a Winograd tile IS a 2x2 block of the pool's stride grid: pooled (i, j) = max(tile (i, j), left column of tile (i, j+1),
//     top row of tile (i+1, j), corner pixel of tile (i+1, j+1)).  A group holds 2 x 8 tiles, one per lane of a 16-lane row, so
//     the right neighbour is one DPP row shift away and the tile below eight;
//   * columns: a group only emits the pooled columns of its first seven tile columns -- groups step 7 tiles (14 pixels), the
//     eighth column is recomputed by the neighbour (1/7 more matrix work instead of an exchange between waves);
//   * rows: a wave walks DOWN a strip of groups, so the missing row of tiles below its second tile row arrives one iteration
//     later in its own registers: the partial maxima of that row are carried (4 registers per 16-channel block) and finished
//     by the next group's first tile row.  A strip is cut into segments for parallelism; each segment runs one extra group
//     at its lower end for that row only;
//   * the expand ReLU runs behind the pool (it commutes with max), which also makes 0 the neutral element everywhere (DPP
//     shifts with zero fill, pixels outside the map);
//   * the pooled values leave the max in exactly the lane layout the squeeze product wants (lane = pooled pixel, 4 channels).
// SAVE (training forward): the pooled tensor (a.sv: expand1x1 window at sv_coff1, expand3x3 at sv_coff) and the pool's arg-max /
// ReLU codes (a.sv_codes, one byte per pooled element: the FIRST window position 3 dy + dx holding the pooled value, 15 where it is
// not > 0 -- maxpool_fwd_kernel<true, true>'s codes) are stored as well: they are all the backward reads of this stage (the
// unpooled expand output is not needed: the codes carry its ReLU mask).  The code of a pooled element is assembled the way its
// value is: first position among the tile's own four pixels and the right neighbour's left column holding THEIR maximum (compares
// against that partial maximum), first among the row below (+ its right neighbour's corner), then whichever side holds the window
// maximum, the upper side on ties (its positions come first).  The partial code of the group's second tile row is carried with its
// partial maximum (four codes packed in one register per channel block).
template <int NSQ, int NCH, bool SAVE = false>
__device__ __forceinline__ void wino_poolbridge16_body(const WinoArgs& a) {
  constexpr int WV = 8, NTHR = WV * 64, RP = 113, RAW_IT = 4, RSLOTS = 224;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int P3 = a.nslices3, P = a.nslices;            // 16-wide passes: expand3x3, then expand1x1
  constexpr int nchunks = NCH;                         // C / 8 (1 or 2)
  const bool e1_half = a.N1 <= 64;
  const int e1_stage = e1_half ? 1024 : 2048;
  const int rb1 = e1_half ? 2 : 4;                     // channel blocks of an expand1x1 pass
  float* const rawB = smem;                            // [2 chunks][WV][RSLOTS][4]
  float* const UB = rawB + 2 * WV * RSLOTS * 4;
  float* const U1B = UB + P3 * nchunks * 2048;
  float* const sqAL = U1B + (P - P3) * nchunks * e1_stage;      // [P3 + rb1 (P - P3) blocks][4 t][NSQ][64 lanes]
  const int nblk = P3 + rb1 * (P - P3);
  float* const biasL = sqAL + nblk * 4 * NSQ * 64;     // [P][4][16], then the squeeze bias [NSQ * 16]
  float* const sqbL = biasL + P * 64;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int ntasks = a.ngroups;                        // (image, strip, segment) triples
  const int wv_s = __builtin_amdgcn_readfirstlane(wv);
  int task = (int)blockIdx.x * WV + wv_s;
  const int tstride = a.gx * WV;

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
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + a.x_coff - (long long)(a.W + 1) * a.x_pitch), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 0x7ffffff0, 0x00020000);
  // output rows are addressed from one pooled row ABOVE the group's first (the carried row), so every lane offset is >= 0
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.y + a.y_coff - (long long)a.pb_wp * a.y_pitch), 0, 0x7ffffff0, 0x00020000);
  // (SAVE) the pooled tensor and its codes, addressed the same way; the codes are one byte per element of the same geometry
  const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((SAVE ? a.sv : a.y) - (SAVE ? (long long)a.pb_wp * a.sv_pitch : 0)), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t cres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(SAVE ? (void*)(a.sv_codes - (long long)a.pb_wp * a.sv_pitch) : (void*)a.y), 0, 0x7ffffff0, 0x00020000);

  // ---- operands into LDS, once ----
  {
    const int u_lane = (g * 16 + lr) * 16;
    const unsigned u_chunkB = 16u * a.Npad * 8u * 4u;
    for (int ps = 0; ps < P; ++ps)
      for (int c = 0; c < nchunks; ++c) {
        const bool is1 = ps >= P3;
        if (is1 && e1_half && (wv_s & 1)) continue;
        float* const dst = is1 ? U1B + ((ps - P3) * nchunks + c) * e1_stage + (e1_half ? (wv_s >> 1) : wv_s) * 256
                               : UB + (ps * nchunks + c) * 2048 + wv_s * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ures, (lds_ptr_w_t)dst, 16, u_lane + wv_s * a.Npad * 64,
                                                 (int)(c * u_chunkB + (unsigned)ps * 1024u), 0, 0);
      }
    for (int i = tid; i < nblk * 4 * NSQ * 64; i += NTHR) sqAL[i] = a.br_w[i];
    for (int i = tid; i < P * 64; i += NTHR) biasL[i] = a.br_bias[i];
    for (int i = tid; i < NSQ * 16; i += NTHR) sqbL[i] = (a.br_sqb && i < a.br_nsq) ? a.br_sqb[i] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (task >= ntasks) return;                          // (after the only barrier)

  struct IterPos { int b, s, r, y0, x0, inner; unsigned soff; };
  // iteration `r` (group row) of task (b, s, .): the 4x16-pixel group at rows 4 r, columns 14 s
  auto iter_pos = [&](int b, int s, int r) {
    IterPos ip;
    ip.b = b; ip.s = s; ip.r = r;
    ip.y0 = 4 * r; ip.x0 = 14 * s;
    const long long p0 = ((long long)b * a.H + ip.y0) * a.W + ip.x0;
    ip.soff = (unsigned)(p0 * a.x_pitch * 4);
    ip.inner = (int)(((unsigned)(-ip.y0) & (unsigned)(ip.y0 + 4 - a.H) & (unsigned)(-ip.x0) & (unsigned)(ip.x0 + 16 - a.W)) >> 31);
    return ip;
  };
  float* const rawW = rawB + wv_s * RSLOTS * 4;
  auto dma_group = [&](const IterPos& ip) {
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
      const int key = r_key[it];
      const bool ok = ip.inner ? true : (key >= 0 && (unsigned)(ip.y0 + (key >> 8) - 1) < (unsigned)a.H && (unsigned)(ip.x0 + (key & 255) - 1) < (unsigned)a.W);
      const int off = ok ? r_offB[it] : (int)OOB;
      // the ring slot holds 224 positions: the last request only has 32 lanes (positions 192..223; 221.. are padding)
      if (it < RAW_IT - 1 || lane < RSLOTS - (RAW_IT - 1) * 64) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_w_t)(rawW + (it * 64) * 4), 16, off, (int)ip.soff, 0, 0);
        if (nchunks > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_ptr_w_t)(rawW + (WV * RSLOTS + it * 64) * 4), 16, off, (int)(ip.soff + 32u), 0, 0);
      }
    }
  };
  auto task_of = [&](int q, int& b, int& s, int& g0, int& g1) {
    const int v = q % a.pb_nseg; const int q1 = q / a.pb_nseg;
    s = q1 % a.pb_ns; b = q1 / a.pb_ns;
    g0 = v * a.pb_gseg; g1 = g0 + a.pb_gseg < a.pb_ng ? g0 + a.pb_gseg : a.pb_ng;
  };

  f32x4 acc[16];
  f32x4 acc_sq[NSQ];
  f32x4 carry[8];                                      // partial pooled maxima of the group's second tile row, per channel block
  unsigned carry_code[SAVE ? 8 : 1];                   // (SAVE) ... and the window positions holding them, four bytes per block
#pragma unroll
  for (int k = 0; k < 8; ++k) carry[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (SAVE ? 8 : 1); ++k) carry_code[k] = 0u;
  const float oneB = (g == 0) ? 1.f : 0.f;
  const int ty = lr >> 3, tx = lr & 7;
  // this lane's pooled pixel: row (2 r - 1) + (1 - ty) ... i.e. ty = 0 -> pooled row 2 r, ty = 1 -> the carried row 2 r - 1
  const int o_off = (((1 - ty) * a.pb_wp + tx) * a.y_pitch + 4 * g) * 4;
  const int p_off = SAVE ? (((1 - ty) * a.pb_wp + tx) * a.sv_pitch + 4 * g) * 4 : 0;       // pooled tensor (bytes); codes: a quarter
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const int rawL_off = (((g >> 1) * RP + (2 * (lr >> 3)) * 18 + 2 * (lr & 7)) * 4 + 2 * (g & 1));
  const int u_ln = g * 64 + lr * 4;
  auto max4 = [](f32x4 p, f32x4 q) {
    f32x4 r;
    r.x = __builtin_fmaxf(p.x, q.x); r.y = __builtin_fmaxf(p.y, q.y); r.z = __builtin_fmaxf(p.z, q.z); r.w = __builtin_fmaxf(p.w, q.w);
    return r;
  };
  // value of lane (lr + n) / (lr - n) of the same 16-lane row, 0 where that lane does not exist
  auto shl = [](float v, auto n_c) {
    constexpr int N = decltype(n_c)::value;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
  };
  auto shr = [](float v, auto n_c) {
    constexpr int N = decltype(n_c)::value;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
  };

  int tb, ts, g0, g1;
  task_of(task, tb, ts, g0, g1);
  int r = g0;
  IterPos cur = iter_pos(tb, ts, r);
  dma_group(cur);
  bool stored = false;

  for (;;) {
    // where the wave goes next: the next group row of this task (one past the segment for the carried row, unless the
    // segment ends at the bottom of the image), else the first group of its next task
    // (at the bottom of the image that extra group lies outside the map: all its pixels are masked to 0 and it only
    // releases the carried last pooled row)
    const int rlast = (g1 < a.pb_ng || 2 * a.pb_ng - 1 < a.pb_hp) ? g1 : g1 - 1;
    int nb = tb, ns = ts, ng0 = g0, ng1 = g1, nr = r + 1, ntask = task;
    bool more = true;
    if (r >= rlast) {
      ntask = task + tstride;
      more = ntask < ntasks;
      if (more) { task_of(ntask, nb, ns, ng0, ng1); nr = ng0; } else { nr = r; }
    }
    // (SAVE: two more stores per channel block; the count is exact for the eight-block shape, anything else waits for everything)
    if (stored && (!SAVE || nblk == 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSQ + (SAVE ? 16 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    // ---- input transform of the whole group (both chunks) ----
    f32x2 vv[NCH][16];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float* const rawL = rawW + c * WV * RSLOTS * 4 + rawL_off;
      f32x2 t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 d0 = *(const f32x2*)(rawL + (0 * 18 + j) * 4), d1 = *(const f32x2*)(rawL + (1 * 18 + j) * 4);
        const f32x2 d2 = *(const f32x2*)(rawL + (2 * 18 + j) * 4), d3 = *(const f32x2*)(rawL + (3 * 18 + j) * 4);
        t[0][j] = d0 - d2; t[1][j] = d1 + d2; t[2][j] = d2 - d1; t[3][j] = d1 - d3;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vv[c][i * 4 + 0] = t[i][0] - t[i][2]; vv[c][i * 4 + 1] = t[i][1] + t[i][2];
        vv[c][i * 4 + 2] = t[i][2] - t[i][1]; vv[c][i * 4 + 3] = t[i][1] - t[i][3];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const IterPos nxt = iter_pos(nb, ns, nr);
    dma_group(nxt);

#pragma unroll
    for (int q = 0; q < NSQ; ++q) acc_sq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool wholexy = cur.y0 + 4 <= a.H && cur.x0 + 16 <= a.W;
    // the pooled pixel this lane finishes in this iteration (see the squeeze store below) and where it goes in the saved tensors
    const bool emit_px = tx < 7 && 7 * ts + tx < a.pb_wp && 2 * r - ty < a.pb_hp && (ty ? (r > g0) : (r < g1));
    const unsigned psoff = SAVE ? (unsigned)((((long long)tb * a.pb_hp + 2 * r) * a.pb_wp + 7 * ts) * a.sv_pitch * 4) : 0u;
    // one 16-channel block of the expand output for the tile's four pixels -> pooled -> into the squeeze
    auto pool_in = [&](int bi, auto kslot_c, f32x4 (&ov)[4], int cbase, int climit) {
      constexpr int KS = decltype(kslot_c)::value;
      if (!wholexy) {
#pragma unroll
        for (int px = 0; px < 4; ++px)
          if (!(cur.y0 + 2 * ty + (px >> 1) < a.H && cur.x0 + 2 * tx + (px & 1) < a.W)) ov[px] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const f32x4 m_top = max4(ov[0], ov[1]), m_left = max4(ov[0], ov[2]);
      const f32x4 m_all = max4(m_top, max4(ov[2], ov[3]));
      // neighbour lanes through DPP operands of the max itself (row_shl:1 = lane + 1 = the tile to the right, row_shl:8 / row_shr:8
      // = the tile below / above; lanes without that neighbour read 0).  Written as ISA: the compiler leaves the shifts as
      // separate v_mov_dpp (16 per channel block) and every VALU instruction delays the matrix pipe.  (s_nop 1: a DPP operand
      // may not be read in the two wait states behind the VALU write of that register.)
      f32x4 A, Bt, pooled;
      const f32x4 ov0 = ov[0];
      asm volatile("s_nop 1\n\t"
                   "v_max_f32_dpp %0, %8, %12 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      // own tile + the right neighbour's left column
                   "v_max_f32_dpp %1, %9, %13 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %2, %10, %14 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %3, %11, %15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %4, %16, %20 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"     // own top row + the right neighbour's corner
                   "v_max_f32_dpp %5, %17, %21 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %6, %18, %22 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %7, %19, %23 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                   : "=&v"(A.x), "=&v"(A.y), "=&v"(A.z), "=&v"(A.w), "=&v"(Bt.x), "=&v"(Bt.y), "=&v"(Bt.z), "=&v"(Bt.w)
                   : "v"(m_left.x), "v"(m_left.y), "v"(m_left.z), "v"(m_left.w), "v"(m_all.x), "v"(m_all.y), "v"(m_all.z), "v"(m_all.w),
                     "v"(ov0.x), "v"(ov0.y), "v"(ov0.z), "v"(ov0.w), "v"(m_top.x), "v"(m_top.y), "v"(m_top.z), "v"(m_top.w));
      // first tile row: its own window + the row below (lane + 8); second tile row: the CARRIED window + the new first row (lane - 8)
      f32x4 base;
#pragma unroll
      for (int e = 0; e < 4; ++e) base[e] = ty ? carry[KS][e] : A[e];
      asm volatile("s_nop 1\n\t"
                   "v_max_f32_dpp %0, %4, %8 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %1, %5, %9 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %2, %6, %10 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %3, %7, %11 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %0, %4, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %1, %5, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %2, %6, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_max_f32_dpp %3, %7, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                   : "=&v"(pooled.x), "=&v"(pooled.y), "=&v"(pooled.z), "=&v"(pooled.w)
                   : "v"(Bt.x), "v"(Bt.y), "v"(Bt.z), "v"(Bt.w), "v"(base.x), "v"(base.y), "v"(base.z), "v"(base.w));
      // the expand ReLU is applied HERE, behind the pool (relu o max == max o relu; zero fills and masked pixels are neutral
      // under it): 4 instead of 16 maxima per channel block
      if constexpr (SAVE) {
        unsigned codes = 0u, cApack = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o0 = ov[0][e], o1 = ov[1][e], o2 = ov[2][e], o3 = ov[3][e];
          float r0;                                            // the right neighbour's top-left pixel (position 2; its lower-left is position 5)
          asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r0) : "v"(o0));
          const float Av = A[e], Bv = Bt[e];
          unsigned cA = 5u;
          cA = (o3 == Av) ? 4u : cA; cA = (o2 == Av) ? 3u : cA; cA = (r0 == Av) ? 2u : cA; cA = (o1 == Av) ? 1u : cA; cA = (o0 == Av) ? 0u : cA;
          const unsigned cB = (o0 == Bv) ? 6u : ((o1 == Bv) ? 7u : 8u);
          // the row neighbour's cB: lanes of the first tile row read lane + 8, of the second lane - 8 (bank_mask: who is written)
          unsigned cBn;
          asm volatile("s_nop 1\n\t"
                       "v_mov_b32_dpp %0, %1 row_shl:8 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
                       "v_mov_b32_dpp %0, %1 row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:1" : "=&v"(cBn) : "v"(cB));
          const unsigned cbase_e = ty ? ((carry_code[KS] >> (8 * e)) & 0xffu) : cA;
          unsigned code = (base[e] == pooled[e]) ? cbase_e : cBn;
          code = pooled[e] > 0.f ? code : 15u;
          codes |= code << (8 * e);
          cApack |= cA << (8 * e);
          __builtin_amdgcn_sched_barrier(0);                   // one element at a time (register pressure)
        }
        carry_code[KS] = cApack;
        pooled = wino_relu4(pooled, 0.f);
        // always issued (the counted wait relies on it): lanes without a pixel or beyond the channel count go out of range
        const bool st = emit_px && cbase + 4 * g < climit;
        const int pv = st ? p_off + cbase * 4 : (int)OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pooled), pres, pv, (int)psoff, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b32(codes, cres, st ? (p_off >> 2) + cbase : (int)OOB, (int)(psoff >> 2), 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      } else {
        pooled = wino_relu4(pooled, 0.f);
      }
      carry[KS] = A;
      const float* const sA = sqAL + bi * (4 * NSQ * 64) + lane;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < NSQ; ++q) acc_sq[q] = mfma16(sA[(t * NSQ + q) * 64], pooled[t], acc_sq[q]);
    };

    {
      auto run_pass = [&](auto kind_c, auto kslot_c, int pass) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int KS0 = decltype(kslot_c)::value;      // carry slot of the pass's first channel block
        constexpr bool E1 = KIND != 0;
        constexpr int SSTEP = (KIND == 2) ? 2 : 1, NS = 8 / SSTEP;
        const float* const bL = biasL + pass * 64 + lr;
        float bA[E1 ? 4 : 1];
#pragma unroll
        for (int k = 0; k < (E1 ? (KIND == 2 ? 2 : 4) : 1); ++k) bA[k] = bL[k * 16];
        {
          auto chunk = [&](auto first_c, auto cc_c) {
            constexpr int cc = decltype(cc_c)::value;
            const float* const uR = (E1 ? U1B + ((pass - P3) * nchunks + cc) * e1_stage : UB + (pass * nchunks + cc) * 2048) + u_ln;
            constexpr bool FIRST = decltype(first_c)::value;
            constexpr int CC = decltype(cc_c)::value;
            f32x4 af0 = *(const f32x4*)uR, af1;
#pragma unroll
            for (int si = 0; si < NS; ++si) {
              const int step = si * SSTEP;
              __builtin_amdgcn_sched_barrier(0);
              if (si + 1 < NS) { if (si & 1) af0 = *(const f32x4*)(uR + (si + 1) * 256); else af1 = *(const f32x4*)(uR + (si + 1) * 256); }
              __builtin_amdgcn_sched_barrier(0);
              const f32x4 af = (si & 1) ? af1 : af0;
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const int p = 2 * step + h;
                  const int pv = E1 ? ((p >> 2) == 0 ? 5 : (p >> 2) == 1 ? 6 : (p >> 2) == 2 ? 9 : 10) : p;
                  f32x4 c0v;
                  if (FIRST && t == 0) {
                    c0v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (E1 ? (p < 4) : (p == 5)) c0v = mfma16(bA[E1 ? (p & 3) : 0], oneB, c0v);
                  } else {
                    c0v = acc[p];
                  }
                  acc[p] = mfma16(af[2 * h + t], vv[CC][pv][t], c0v);
                }
            }
          };
          chunk(std::true_type{}, std::integral_constant<int, 0>{});
          if constexpr (NCH > 1) chunk(std::false_type{}, std::integral_constant<int, 1>{});
        }
        if constexpr (E1) {
          const int bi0 = P3 + rb1 * (pass - P3);
#pragma unroll
          for (int rr = 0; rr < (KIND == 2 ? 2 : 4); ++rr) {
            f32x4 ov[4];
            auto inv1 = [&](auto half, auto put) {
              const f32x2 m0 = half(acc[0 + rr]), m1 = half(acc[4 + rr]), m2 = half(acc[8 + rr]), m3 = half(acc[12 + rr]);
              const f32x2 s01 = m0 + m1, d01 = m0 - m1, s23 = m2 + m3, d23 = m2 - m3;
              put(0, s01 + s23); put(1, d01 + d23); put(2, s01 - s23); put(3, d01 - d23);
            };
            inv1([](const f32x4& v) { return (f32x2)v.lo; }, [&](int px, f32x2 y) { ov[px].lo = y; });
            inv1([](const f32x4& v) { return (f32x2)v.hi; }, [&](int px, f32x2 y) { ov[px].hi = y; });
            // (SAVE) channel of the block in the expand1x1 window of the saved tensor
            const int s1 = pass - P3, c1 = a.sv_coff1 + 128 * (s1 >> 1) + (2 * rr + (s1 & 1)) * 16, l1 = a.sv_coff1 + a.N1;
            if (rr == 0) pool_in(bi0 + rr, std::integral_constant<int, KS0>{}, ov, c1, l1);
            else if (rr == 1) pool_in(bi0 + rr, std::integral_constant<int, KS0 + 1>{}, ov, c1, l1);
            else if (rr == 2) pool_in(bi0 + rr, std::integral_constant<int, (KS0 + 2) & 7>{}, ov, c1, l1);
            else pool_in(bi0 + rr, std::integral_constant<int, (KS0 + 3) & 7>{}, ov, c1, l1);
          }
        } else {
          auto inv = [&](auto half, auto put) {
            f32x2 sx[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
              const f32x2 m0 = half(acc[xi * 4 + 0]), m1 = half(acc[xi * 4 + 1]);
              const f32x2 m2 = half(acc[xi * 4 + 2]), m3 = half(acc[xi * 4 + 3]);
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
          pool_in(pass, std::integral_constant<int, KS0>{}, ov, a.sv_coff + 16 * pass, a.sv_coff + a.N);
        }
      };
      // the carry slots are compile-time register indices, so the (at most 8) channel blocks are enumerated statically:
      // expand3x3 passes 0..3 -> slots 0..3, expand1x1 passes -> slots 4.. (host-checked: P3 <= 4, rb1 (P - P3) <= 4)
      if (P3 > 0) run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
      if (P3 > 1) run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 1);
      if (P3 > 2) run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, 2);
      if (P3 > 3) run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, 3);
      if (e1_half) {
        if (P > P3) run_pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{}, P3);
        if (P > P3 + 1) run_pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{}, P3 + 1);
      }
    }
    // ---- the pooled pixels this iteration finished: first tile row -> pooled row 2 r (not in the extra iteration past the
    // segment), second tile row -> the carried row 2 r - 1 (not in the segment's first iteration); columns 7 s .. 7 s + 6 ----
    {
      const int prow = 2 * r - ty, pcol = 7 * ts + tx;
      const bool emit = tx < 7 && pcol < a.pb_wp && prow < a.pb_hp && (ty ? (r > g0) : (r < g1));
      const unsigned ysoff = (unsigned)((((long long)tb * a.pb_hp + 2 * r) * a.pb_wp + 7 * ts) * a.y_pitch * 4);
#pragma unroll
      for (int q = 0; q < NSQ; ++q) {
        const f32x4 v = wino_relu4(acc_sq[q] + *(const f32x4*)(sqbL + q * 16 + 4 * g), 0.f);
        // every wave issues exactly NSQ stores per iteration (the counted wait above relies on it): lanes without a
        // pixel, or beyond the squeeze width, point outside the buffer and are dropped by the range check
        const int voff = (emit && q * 16 + 4 * g < a.br_nsq) ? o_off + q * 64 : (int)OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yres, voff, (int)ysoff, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      stored = true;
    }
    if (!more) break;
    task = ntask; tb = nb; ts = ns; g0 = ng0; g1 = ng1; r = nr;
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NSQ, int NCH, bool SAVE>
__global__ __launch_bounds__(512, 1) void fire_poolbridge16_kernel(WinoArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  wino_poolbridge16_body<NSQ, NCH, SAVE>(a);
#endif
}

template <int NSQ, int NCH, bool SAVE>
static int launch_wino_poolbridge16_t(WinoArgs a, int nseg, hipStream_t stream) {
  constexpr int WV = 8, NTHR = WV * 64, RSLOTS = 224;
  const int nchunks = a.C >> 3;
  if (nchunks != NCH) return SQD_ERR_UNSUPPORTED;
  const int P1 = 2 * sqd_cdiv(a.N1, 128);
  if (a.Npad != sqd_cdiv(a.N, 32) * 32 + sqd_cdiv(a.N1, 128) * 32) return SQD_ERR_BAD_ARG;
  const int first1 = sqd_cdiv(a.N, 32) * 2;
  const int e1_stage = a.N1 <= 64 ? 1024 : 2048, rb1 = a.N1 <= 64 ? 2 : 4;
  // carry registers: at most 8 sixteen-channel blocks (4 expand3x3 passes + 2 expand1x1 passes of 2 blocks)
  if (first1 > 4 || a.N1 > 64 || P1 > 2) return SQD_ERR_UNSUPPORTED;
  const int nblk = first1 + rb1 * P1;
  const size_t lds = (size_t)(2 * WV * RSLOTS * 4 + first1 * nchunks * 2048 + P1 * nchunks * e1_stage + nblk * 4 * NSQ * 64 + (first1 + P1) * 64 + NSQ * 16) * sizeof(float);
  if (lds > 160 * 1024) return SQD_ERR_UNSUPPORTED;
  if (SAVE && (!a.sv || !a.sv_codes)) return SQD_ERR_BAD_ARG;
  auto kern = fire_poolbridge16_kernel<NSQ, NCH, SAVE>;
  static SqdDevOnce attr_once;                 // (per device: ADVICE round 4)
  if (int rc_attr = sqd_max_lds_once(attr_once, (const void*)kern, 160 * 1024)) return rc_attr;
  a.pb_ng = sqd_cdiv(a.H, 4);
  a.pb_ns = sqd_cdiv(a.pb_wp, 7);
  if (nseg < 1) nseg = 1;
  if (nseg > a.pb_ng) nseg = a.pb_ng;
  a.pb_gseg = sqd_cdiv(a.pb_ng, nseg);
  a.pb_nseg = sqd_cdiv(a.pb_ng, a.pb_gseg);
  a.ngroups = a.B * a.pb_ns * a.pb_nseg;              // tasks
  const int wgs = sqd_cdiv(a.ngroups, WV);
  const int slots = wino_num_cus();
  const int per_wg = sqd_cdiv(wgs, slots);
  const int gx = sqd_cdiv(wgs, per_wg);
  a.nslices3 = first1; a.nslices = first1 + P1; a.gx = gx;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(NTHR), lds, stream, a);
  return sqd_launch_status();
}

template <int NSQ, bool SAVE = false>
static int launch_wino_poolbridge16(WinoArgs a, int nseg, hipStream_t stream) {
  if ((a.C >> 3) == 1) return launch_wino_poolbridge16_t<NSQ, 1, SAVE>(a, nseg, stream);
  if ((a.C >> 3) == 2) return launch_wino_poolbridge16_t<NSQ, 2, SAVE>(a, nseg, stream);
  return SQD_ERR_UNSUPPORTED;
}
