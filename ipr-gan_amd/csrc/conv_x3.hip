// Math mode "fp32x3" on tensors that LIVE as three bf16 planes (storage kind 2 of include/iprgan.h: x = h + m + l exactly,
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); plane-major, each plane an ordinary bf16 NHWC tensor).
//
// Round 3 split every fp32 operand element while it was staged into LDS (gconv_kernel<.., SPLIT>, conv_igemm.hip): the
// element is split again by every block that loads it (a 3x3 256 -> 256 layer loads an input element 18 times), the
// staging path global -> registers -> split -> LDS could not overlap the MFMAs, and the kernel stopped at 0.39 of its
// matrix-pipe bound.  Here the split happens ONCE, where a tensor is produced (conv / norm / activation epilogues, weight
// prep), and the K loop of this file has no vector-ALU work at all:
//   * the three planes of both operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), a wave-instruction
//     moving 16 tile rows x 64 bytes (32 channels of one plane) - no staging registers, no ds_write pass;
//   * a K step is 32 channels of one tap: per 16-deep sub-step a wave reads its 3 + 3 fragments per block row / column
//     once and multiplies SIX plane pairs per 32x32 block (l h', h l', m m', m h', h m', h h': smallest first, fp32
//     accumulator) - 6 MFMAs per pair of 16-byte LDS reads where the bf16 mode has one, so LDS bandwidth and the DMA
//     rate per MFMA are a third of the bf16 tiles';
//   * stages form a ring with counted `s_waitcnt vmcnt(N)` across ONE raw s_barrier per step; the refill pieces of the
//     next stage are woven into the MFMA stream (conv_pipe.hip measured DMA bursts and MFMA phases ADDING otherwise).
// LDS image of a stage: [A planes h, m, l][B planes h, m, l], each [rows][64 bytes]; 4 rows per 256-byte bank line, so
// the 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3): the four 16-lane groups of a ds_read_b128 then hit
// 16 distinct slots (rows {0-3, 12-15, 20-27}: r >> 2 = 0, 3, 5, 6 -> XOR 0, 3, 1, 2).  The DMA destination is
// lane-linear, so the swizzle is applied to the per-lane SOURCE chunk.
// Roofline: 6 MFMAs of 32 cycles per 32x32x16 block = 2.5 PFLOP/s / 6 = 416.7 TFLOP/s fp32-equivalent (bench.PEAK_X3_MFMA).
// Reference layers: networks/sn_discriminator.py:9-18, conv_generator.py:8,21, sr_resnet.py:22, discriminator_96.py:7-21,
// resnet_generator.py:7-34 (ReflectionPad2d folded into the DMA offsets), vgg.py:33.
#include "conv_pipe_shared.h"

// K-loop probes (A/B builds only: -DX3P_PROBE=1 no MFMAs, 2 no refills, 4 no epilogue; results are wrong, only the time matters)
#ifndef X3P_PROBE
#define X3P_PROBE 0
#endif

namespace iprgan {

template <int WGM, int WGN, int WM, int WN, int NSTAGE, bool STATS, bool PREF>
__global__ __launch_bounds__(WGM * WGN * 64) void gconv_x3p_kernel(const GConvArgs a) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, NW = WGM * WGN;
  constexpr bool FDB = NW <= 4;                  // one wave per SIMD: 512 registers
  constexpr int RSA = BM / 16 / NW, RSB = BN / 16 / NW;        // 16-row pieces per plane this wave stages
  constexpr int L = 3 * (RSA + RSB);                           // LDS-DMA instructions per wave and stage
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, A_BYTES = 3 * A_PLANE, STAGE_BYTES = 3 * (BM + BN) * 64;
  static_assert(RSA >= 1 && RSB >= 1 && BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "every wave stages whole 16-row pieces");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4 && L * (NSTAGE - 2) <= 63, "ring depth / vmcnt field");
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

  // logical tile order as in gconv_kernel: n tiles fastest, then the sub-pixel phases, then m tiles
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const unsigned lq = lt / gridDim.y;
  const int zi = (int)(lq % gridDim.z);
  const int pz = a.ksplit > 1 ? 0 : zi;          // blockIdx.z: sub-pixel phase, or K split of a single-phase geometry
  const int pM = a.ph[pz].M;
  const int m0 = (int)(lq / gridDim.z) * BM, n0 = (int)(lt % gridDim.y) * BN;
  if (m0 >= pM) {
    if (STATS) {
      for (int c = threadIdx.x; c < BN; c += NW * 64)
        if (n0 + c < a.Ns) { a.stat_part[((size_t)lq * 2) * a.Ns + n0 + c] = 0.f; a.stat_part[((size_t)lq * 2 + 1) * a.Ns + n0 + c] = 0.f; }
    }
    return;
  }
  const int p_tw = a.ph[pz].tw, p_th = a.ph[pz].th;
  int nt = a.ph[pz].steps, s_begin = 0;              // 32-deep K steps (Cs % 32 == 0: a step lies inside one tap)
  if (a.ksplit > 1) {                                // this block's share of the K walk; its partial tile goes to slab zi
    const int per = (nt + a.ksplit - 1) / a.ksplit;
    s_begin = zi * per;
    const int s_end = s_begin + per < nt ? s_begin + per : nt;
    nt = s_end > s_begin ? s_end - s_begin : 0;
  }
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;
  const int p_dy0 = a.ph[pz].dy0, p_dx0 = a.ph[pz].dx0, p_dys = a.ph[pz].dys, p_dxs = a.ph[pz].dxs;
  const int p_wbase = a.ph[pz].wbase, p_wsy = a.ph[pz].wsy, p_wsx = a.ph[pz].wsx;
  const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
  const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;
  const unsigned in_ps = a.in_ps, wt_ps = a.wt_ps;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // provably wave-uniform (LDS-DMA base, M0)
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 2;                                          // row of a 16-row piece this lane copies
  const unsigned sc = (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;   // its source chunk: position ^ swz(row)

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

  int aiy[RSA], aix[RSA];
  unsigned arow[RSA], wrow[RSB];
#pragma unroll
  for (int i = 0; i < RSA; ++i) {
    const int r = (i * NW + wave) * 16 + lrow;
    const int m = m0 + r;
    if (m < pM) {
      const int b = fdiv(m, d_plane);
      const int rem = m - b * plane;
      const int y = fdiv(rem, d_owg);
      const int x = rem - y * p_owg;
      aiy[i] = y * a.isy;
      aix[i] = x * a.isx;
      arow[i] = (unsigned)(((b * IH + aiy[i]) * IW + aix[i]) * Cs) * 2u + sc;
    } else {
      aiy[i] = ROW_INVALID; aix[i] = 0; arow[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < RSB; ++i) {
    const int r = (i * NW + wave) * 16 + lrow;
    wrow[i] = (unsigned)((n0 + r) * a.Kp) * 2u + sc;
  }

  // TWO accumulators per block: `acc` takes h h' only, `accs` the five small terms (<= 2^-7 of the product each).  The bf16
  // MFMA aligns its 16 products to the exponent of the accumulator input and truncates: every MFMA into a LARGE accumulator
  // costs about an ulp of it, however small its products (measured on K = 1152: one accumulator, six MFMAs per 16 k: rms
  // error 5.1e-7 of the result, against 3.0e-7 for the fp32 MFMA).  With the small terms summed among themselves the large
  // accumulator sees one MFMA per 16 k, the small one's roundings are 2^-7 of that, and the two meet once, in fp32, below.
  f32x16 acc[WM][WN], accs[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = accs[i][j][r] = 0.f;

  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  // wave-uniform walk over (channel chunk, tap): taps inside a 32-channel chunk, so that consecutive K steps re-read the
  // same pixels, shifted (they stay in L2)
  // K-walk order (A/B: IPRGAN_X3P_KORDER).  bit 0: channel chunks inside a tap - consecutive steps read the 64-byte pieces of
  // the SAME 128-byte lines of a pixel's channel vector, one after the other.  bit 1: taps of a stride-2 gather grouped by
  // parity - (ty, tx), (ty, tx + 2), (ty + 2, tx), (ty + 2, tx + 2) are the four taps through which ONE input pixel meets the
  // four output positions that use it; walked back to back, the re-reads of a pixel are 1-3 steps apart instead of 2 / 8 / 10
  // (an XCD's 4 MiB L2 is turned over by its 32 CUs' stages every ~2 steps: only near re-reads are L2 hits)
  const bool chunk_fast = (a.korder & 1) != 0;
  const int p_ntap = p_th * p_tw;
  const bool parity = (a.korder & 2) != 0 && a.isy == 2 && a.isx == 2 && p_th >= 2 && p_tw >= 2;
  // taps of parity class (cy, cx): ty = cy, cy + 2, ... < th (k4: two per class and axis; k3: two even, one odd)
  int u_c = 0, u_ty = 0, u_tx = 0;
  int q_cls = 0, q_dy = 0, q_dx = 0;          // parity walk: class (ty & 1, tx & 1), then (ty >> 1, tx >> 1) inside it
  auto tap_next = [&]() -> bool {             // next tap of the walk (counters only: no division in the K loop); true = wrapped
    bool wrapped = false;
    if (parity) {
      if (++q_dx == ((p_tw - (q_cls & 1) + 1) >> 1)) {
        q_dx = 0;
        if (++q_dy == ((p_th - (q_cls >> 1) + 1) >> 1)) { q_dy = 0; if (++q_cls == 4) { q_cls = 0; wrapped = true; } }
      }
      u_ty = (q_cls >> 1) + 2 * q_dy; u_tx = (q_cls & 1) + 2 * q_dx;
    } else if (++u_tx == p_tw) { u_tx = 0; if (++u_ty == p_th) { u_ty = 0; wrapped = true; } }
    return wrapped;
  };
  if (s_begin) {                              // (split K: the walk starts at step s_begin)
    const int nck = Cs / 32;
    const int ck = chunk_fast ? s_begin % nck : s_begin / p_ntap;
    const int j = chunk_fast ? s_begin / nck : s_begin - ck * p_ntap;
    u_c = ck * 32;
    if (parity) {
      for (int i = 0; i < j; ++i) tap_next();             // (once per block, at most th * tw - 1 steps)
    } else {
      u_ty = j / p_tw; u_tx = j - u_ty * p_tw;
    }
  }
  int w_dy = 0, w_dx = 0, w_tapoff = 0;
  unsigned w_wk = 0, w_sbase = 0;
  auto walk_begin = [&](int buf) {            // address pieces of the K step the walk points at, into stage `buf`
    w_dy = p_dy0 + u_ty * p_dys; w_dx = p_dx0 + u_tx * p_dxs;
    w_tapoff = ((w_dy * IW + w_dx) * Cs + u_c) * 2;
    w_wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * 2u;
    w_sbase = lds_base + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
  };
  auto walk_next = [&]() {
    if (chunk_fast) {
      u_c += 32;
      if (u_c == Cs) { u_c = 0; tap_next(); }
    } else if (tap_next()) u_c += 32;
  };
  // piece q of the stage: q < 3 * RSA: plane q / RSA of the activation rows of row set q % RSA; then the weight rows
  auto piece = [&](int q) {
    if (q < 3 * RSA) {
      const int p = q / RSA, i = q % RSA;
      const int iy = aiy[i] + w_dy, ix = aix[i] + w_dx;
      bool ok;
      unsigned off;
      if (reflect) {              // wave-uniform branch: the mirrored pixel instead of a zero
        ok = aiy[i] != ROW_INVALID;
        const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
        off = arow[i] + (unsigned)((((ry - aiy[i]) * IW + (rx - aix[i])) * Cs + u_c) * 2);
      } else {
        ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        off = arow[i] + (unsigned)w_tapoff;
      }
      dma16(rs_in, w_sbase + (unsigned)p * A_PLANE + (unsigned)(i * NW) * 1024u, ok ? off + (unsigned)p * in_ps : OOB_OFFSET);
    } else {
      const int p = (q - 3 * RSA) / RSB, i = (q - 3 * RSA) % RSB;
      dma16(rs_wt, w_sbase + A_BYTES + (unsigned)p * B_PLANE + (unsigned)(i * NW) * 1024u, wrow[i] + w_wk + (unsigned)p * wt_ps);
    }
  };
  auto issue = [&](int buf) {
    walk_begin(buf);
#pragma unroll
    for (int q = 0; q < L; ++q) piece(q);
    walk_next();
  };

  // fragment read offsets inside a plane image: row (..) * 32 + l31, chunk (2 kk + half) ^ swz(l31)
  const int half = lane >> 5, l31 = lane & 31;
  unsigned foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = (unsigned)l31 * 64u + (unsigned)((2 * kk + half) ^ ((l31 >> 2) & 3)) * 16u;
  const unsigned a_wave = (unsigned)(wm * WM) * 2048u, b_wave = A_BYTES + (unsigned)(wn * WN) * 2048u;
  const char* ldsc = (const char*)lds;

  // one K step on stage `cb`; ISS: the refill of stage `nb` is woven into the MFMA stream (one LDS-DMA instruction after
  // every (SPREAD / L)-th MFMA), so that a wave waits for one slot of the CU's address path at a time while its (and its
  // SIMD partner's) MFMAs run.  The fragments of sub-step 1 are read while sub-step 0 is multiplied.
  auto step = [&](int cb, int nb, auto ISS) {
    constexpr bool iss = decltype(ISS)::value;
    constexpr int NMF = 6 * WM * WN;                                   // MFMAs per 16-deep sub-step
    constexpr int SPREAD = NSTAGE >= 3 ? 2 * NMF : (3 * NMF) / 2;      // two-stage rings: the rest of the step is landing time
    const char* sb = ldsc + cb * STAGE_BYTES;
    if constexpr (iss) walk_begin(nb);
    constexpr int NFB = FDB ? 2 : 1;           // fragment buffers: two (sub-step 1 read while 0 is multiplied) when registers allow
    bf16x8 af[NFB][3][WM], bf[NFB][3][WN];
    auto frags = [&](int kk) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[kk % NFB][p][i] = *(const bf16x8*)(sb + a_wave + p * A_PLANE + i * 2048 + foff[kk]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[kk % NFB][p][j] = *(const bf16x8*)(sb + b_wave + p * B_PLANE + j * 2048 + foff[kk]);
      }
    };
    frags(0);
    int q = 0, mi = 0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (FDB ? kk == 0 : kk == 1) frags(1);
      // (A term, B term): l h', h l', m m', m h', h m' into the small accumulator, then h h' into the large one;
      // consecutive MFMAs go to different accumulators
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
        const int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if constexpr (X3P_PROBE & 1) acc[i][j][0] += (float)af[kk % NFB][pa][i][0] * (float)bf[kk % NFB][pb][j][0];
            else if (t < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk % NFB][pa][i], bf[kk % NFB][pb][j], accs[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk % NFB][pa][i], bf[kk % NFB][pb][j], acc[i][j], 0, 0, 0);
            ++mi;
            if constexpr (iss) {
              if (q < L && q * SPREAD < mi * L) {
                if constexpr (!(X3P_PROBE & 2)) piece(q);
                ++q;
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
      }
    }
    if constexpr (iss) {
#pragma unroll
      for (; q < L; ++q) { if constexpr (!(X3P_PROBE & 2)) piece(q); }
      walk_next();
    }
  };

  // PREF: the fused-derivative operand of this thread's stores, loaded FIRST: vmcnt retires in order, so the wait for
  // stage 0 covers these loads (the same latency, once per tile) and every later counted wait is unaffected
  using G = EpiGeom<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES>;
  u32x4 auxpf[PREF ? G::NIT : 1];
  if constexpr (PREF) pipe_aux_load<G>(a, pz, m0, n0, 0, auxpf);

  // ---- the ring
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nt) issue(s);
  int cur = 0, nxt = NSTAGE - 1;                 // stage read at step t, stage refilled at step t (= read at t - 1)
  int t = 0;
  for (; t < nt - (NSTAGE - 1); ++t) {           // steps that refill a stage
    wait_stages<L>(NSTAGE - 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of step t - 1 are complete
    __builtin_amdgcn_s_barrier();
    step(cur, nxt, std::true_type{});
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  for (; t < nt; ++t) {                          // the last NSTAGE - 1 steps drain the ring
    const int rem = nt - 1 - t;
    wait_stages<L>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    step(cur, nxt, std::false_type{});
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // the epilogue reuses the ring
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] += accs[i][j];

  const unsigned slab_off = a.ksplit > 1 ? (unsigned)zi * (unsigned)pM * (unsigned)a.Ns : 0u;      // elements (< 2^31: checked)
  if constexpr (X3P_PROBE & 4) {                 // (the accumulators stay live: one conditional store nobody takes)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e30f) ((float*)a.out)[0] = t;
    return;
  }
  pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf, nullptr, slab_off);
}

// ---- the ring with DEDICATED LOADER WAVES (round 5) -----------------------------------------------------------------------
// conv_pipe.hip's K-loop probes measured the DMA-only and the MFMA-only time of a ring step ADDING: a wave that issues an
// LDS-DMA piece sits in its issue slot until the CU's address path has taken it (60-185 cycles per piece beside MFMAs and
// fragment reads), and an in-order wave issues no MFMA meanwhile.  A four-wave 256x64 tile - the 64-column layers: D.conv1,
// G.up2, every backward-data pass into 64 channels - issues 15 pieces per wave and step against 48 MFMAs (1 536 cycles): the
// matrix pipe starves on issue stalls (20-30 % busy), not on bytes (L2 at 7.7 of 34 TB/s).  Here the block has EIGHT waves:
// waves 0-3 multiply (one per SIMD, 64x64 accumulators each, fragments from LDS, NO vector-memory instruction in their K
// loop), waves 4-7 do nothing but issue the stage refills (a loader wave alone issues a piece in ~25 cycles:
// MI355X_MICROARCH.md, ldsdma-fill) and wait for them to land.  Same LDS image, same ring, same counted waits (kept by the
// loaders), ONE s_barrier per K step shared by both roles; the loaders match the epilogue's barriers and leave.
// Staging layout = the four-wave tile's (loader l owns the 16-row pieces l, l + 4, ... of every plane).
// M16: the multiplying waves use v_mfma_f32_16x16x32_bf16 (gconv_x3p16_kernel's fragments and LDS swizzle: less energy per FLOP on
// a chip that sits on its power cap under these kernels)
template <int WGM, int WGN, int WM, int WN, int NSTAGE, bool STATS, bool PREF, int MINW = 2, bool M16 = false>
__global__ __launch_bounds__(512, MINW) void gconv_x3ws_kernel(const GConvArgs a) {      // MINW = 4: two blocks per CU (128 registers)
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, NW = WGM * WGN;
  static_assert(NW == 4, "four multiplying waves + four loader waves");
  constexpr int RSA = BM / 16 / NW, RSB = BN / 16 / NW;
  constexpr int L = 3 * (RSA + RSB);
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, A_BYTES = 3 * A_PLANE, STAGE_BYTES = 3 * (BM + BN) * 64;
  static_assert(RSA >= 1 && RSB >= 1 && BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "every loader stages whole 16-row pieces");
  static_assert(NSTAGE >= 2 && NSTAGE <= 3 && L * (NSTAGE - 2) <= 63, "ring depth / vmcnt field");
  using G = EpiGeom<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, NW>;      // all eight waves share the epilogue's row passes
  static_assert(G::NH == 1, "one column round");
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const unsigned lq = lt / gridDim.y;
  const int pz = (int)(lq % gridDim.z);
  const int pM = a.ph[pz].M;
  const int m0 = (int)(lq / gridDim.z) * BM, n0 = (int)(lt % gridDim.y) * BN;
  if (m0 >= pM) {
    if (STATS) {
      for (int c = threadIdx.x; c < BN; c += 512)
        if (n0 + c < a.Ns) { a.stat_part[((size_t)lq * 2) * a.Ns + n0 + c] = 0.f; a.stat_part[((size_t)lq * 2 + 1) * a.Ns + n0 + c] = 0.f; }
    }
    return;
  }
  const int nt = a.ph[pz].steps;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* ldsc = (const char*)lds;

  if (wave8 >= NW) {
    // ================================================= loader waves =================================================
    const int wave = wave8 - NW;                                     // loader index = the staging role of wave `wave` of the 4-wave tile
    const int p_tw = a.ph[pz].tw, p_th = a.ph[pz].th;
    const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;
    const int p_dy0 = a.ph[pz].dy0, p_dx0 = a.ph[pz].dx0, p_dys = a.ph[pz].dys, p_dxs = a.ph[pz].dxs;
    const int p_wbase = a.ph[pz].wbase, p_wsy = a.ph[pz].wsy, p_wsx = a.ph[pz].wsx;
    const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
    const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg;
    const int IH = a.IH, IW = a.IW, Cs = a.Cs;
    const unsigned in_ps = a.in_ps, wt_ps = a.wt_ps;
    const int lrow = lane >> 2;
    // source chunk of this lane's LDS position: the swizzle that makes the multiplying waves' fragment reads conflict-free
    const unsigned sc = M16 ? (unsigned)((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 16u : (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    int aiy[RSA], aix[RSA];
    unsigned arow[RSA], wrow[RSB];
#pragma unroll
    for (int i = 0; i < RSA; ++i) {
      const int m = m0 + (i * NW + wave) * 16 + lrow;
      if (m < pM) {
        const int b = fdiv(m, d_plane);
        const int rem = m - b * plane;
        const int y = fdiv(rem, d_owg);
        const int x = rem - y * p_owg;
        aiy[i] = y * a.isy;
        aix[i] = x * a.isx;
        arow[i] = (unsigned)(((b * IH + aiy[i]) * IW + aix[i]) * Cs) * 2u + sc;
      } else {
        aiy[i] = ROW_INVALID; aix[i] = 0; arow[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < RSB; ++i) wrow[i] = (unsigned)((n0 + (i * NW + wave) * 16 + lrow) * a.Kp) * 2u + sc;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    u32x4 auxpf[PREF ? G::NIT : 1];
    if constexpr (PREF) pipe_aux_load<G>(a, pz, m0, n0, 0, auxpf);     // ahead of every LDS-DMA piece in this wave's vmcnt queue: the counted waits
                                                                       // ("at most N outstanding") never see it
    const bool chunk_fast = (a.korder & 1) != 0;
    const bool parity = (a.korder & 2) != 0 && a.isy == 2 && a.isx == 2 && p_th >= 2 && p_tw >= 2;
    int u_c = 0, u_ty = 0, u_tx = 0, q_cls = 0, q_dy = 0, q_dx = 0;
    auto tap_next = [&]() -> bool {
      bool wrapped = false;
      if (parity) {
        if (++q_dx == ((p_tw - (q_cls & 1) + 1) >> 1)) {
          q_dx = 0;
          if (++q_dy == ((p_th - (q_cls >> 1) + 1) >> 1)) { q_dy = 0; if (++q_cls == 4) { q_cls = 0; wrapped = true; } }
        }
        u_ty = (q_cls >> 1) + 2 * q_dy; u_tx = (q_cls & 1) + 2 * q_dx;
      } else if (++u_tx == p_tw) { u_tx = 0; if (++u_ty == p_th) { u_ty = 0; wrapped = true; } }
      return wrapped;
    };
    auto issue = [&](int buf) {
      const int w_dy = p_dy0 + u_ty * p_dys, w_dx = p_dx0 + u_tx * p_dxs;
      const int w_tapoff = ((w_dy * IW + w_dx) * Cs + u_c) * 2;
      const unsigned w_wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * 2u;
      const unsigned w_sbase = lds_base + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
#pragma unroll
      for (int q = 0; q < L; ++q) {
#ifdef IPRGAN_X3WS_HALFDMA      // debug build only (ws_phase_times.sh): 1 = every other piece is not issued - is the K step bound by DMA intake?
        if (IPRGAN_X3WS_HALFDMA == 1 && (q & 1)) continue;          // 2 = no weight pieces, 3 = no activation pieces (what does each kind cost?)
        if (IPRGAN_X3WS_HALFDMA == 2 && q >= 3 * RSA) continue;
        if (IPRGAN_X3WS_HALFDMA == 3 && q < 3 * RSA) continue;
#endif
        if (q < 3 * RSA) {
          const int p = q / RSA, i = q % RSA;
          const int iy = aiy[i] + w_dy, ix = aix[i] + w_dx;
          bool ok;
          unsigned off;
          if (reflect) {
            ok = aiy[i] != ROW_INVALID;
            const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
            off = arow[i] + (unsigned)((((ry - aiy[i]) * IW + (rx - aix[i])) * Cs + u_c) * 2);
          } else {
            ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
            off = arow[i] + (unsigned)w_tapoff;
          }
          dma16(rs_in, w_sbase + (unsigned)p * A_PLANE + (unsigned)(i * NW) * 1024u, ok ? off + (unsigned)p * in_ps : OOB_OFFSET);
        } else {
          const int p = (q - 3 * RSA) / RSB, i = (q - 3 * RSA) % RSB;
          dma16(rs_wt, w_sbase + A_BYTES + (unsigned)p * B_PLANE + (unsigned)(i * NW) * 1024u, wrow[i] + w_wk + (unsigned)p * wt_ps);
        }
      }
      if (chunk_fast) {
        u_c += 32;
        if (u_c == Cs) { u_c = 0; tap_next(); }
      } else if (tap_next()) u_c += 32;
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nt) issue(s);
    int nxt = NSTAGE - 1;
    for (int t = 0; t < nt; ++t) {
      const int rem = nt - 1 - t;
#ifdef IPRGAN_X3WS_HALFDMA
      wait_stages<(IPRGAN_X3WS_HALFDMA == 1 ? (L + 1) / 2 : IPRGAN_X3WS_HALFDMA == 2 ? 3 * RSA : 3 * RSB)>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);
#else
      wait_stages<L>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);   // this loader's share of stage t has landed
#endif
      __builtin_amdgcn_s_barrier();                            // ... and every multiplying wave is done with stage t - 1
      if (t + NSTAGE - 1 < nt) issue(nxt);
      nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
    __builtin_amdgcn_s_barrier();                              // "the epilogue reuses the ring"
    // the loaders take half of the epilogue's row passes (LDS -> split -> stores): same barriers, no accumulators
    if constexpr (M16) {
      f32x4 none[2 * WM][2 * WN];
      pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true, f32x4[2 * WM][2 * WN], NW, false>(a, none, (float*)lds, pz, lq, m0, n0, auxpf, nullptr, 0u);
    } else {
      f32x16 none[WM][WN];
      pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true, f32x16[WM][WN], NW, false>(a, none, (float*)lds, pz, lq, m0, n0, auxpf, nullptr, 0u);
    }
    return;
  }

  // =================================================== multiplying waves ===================================================
  const int wave = wave8;
  const int wm = wave / WGN, wn = wave % WGN;
  const unsigned a_wave = (unsigned)(wm * WM) * 2048u, b_wave = A_BYTES + (unsigned)(wn * WN) * 2048u;
  u32x4 auxpf[PREF ? G::NIT : 1];
  if constexpr (PREF) pipe_aux_load<G>(a, pz, m0, n0, 0, auxpf);       // (nothing else of this wave is ever in the vmcnt queue)
  // (s_setprio(3) on the multiplying waves: no effect on any layer - 8.125 / 8.136 ms per DCGAN step)
  if constexpr (M16) {
    f32x4 acc[2 * WM][2 * WN], accs[2 * WM][2 * WN];
#pragma unroll
    for (int i = 0; i < 2 * WM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * WN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = accs[i][j][r] = 0.f;
    const int l15 = lane & 15;
    const unsigned foff = (unsigned)l15 * 64u + (unsigned)((lane >> 4) ^ ((4 - ((l15 >> 2) & 3)) & 3)) * 16u;      // + 1024 per 16-row block
    int cur = 0;
    X3WS_STAMP(0);
#ifdef IPRGAN_X3WS_TIMING
    unsigned long long bar_wait = 0;               // cycles this wave spends between arriving at the K-loop barrier and leaving it
#endif
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef IPRGAN_X3WS_TIMING
      const unsigned long long tb = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      if (t > 0) bar_wait += __builtin_amdgcn_s_memtime() - tb;
#else
      __builtin_amdgcn_s_barrier();
#endif
      if (t == 0) X3WS_STAMP(1);
      const char* sb = ldsc + cur * STAGE_BYTES;
      // Order of the 16 product blocks of this wave: along the "staircase" (0,0) (0,1) (1,0) (1,1) (0,2) (1,2) (2,0) ... so that
      // every group of blocks needs ONE new fragment triple (three planes of one A block row or one B block column).  Walking
      // whole rows instead needs all twelve B fragments inside the first row: the four multiplying waves then pull 60 KB
      // out of LDS right behind the barrier (480 cycles of the LDS pipe) while their MFMAs wait for operands.
      bf16x8 af[2 * WM][3], bf[3][2 * WN];
      auto load_a = [&](int i) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) af[i][p] = *(const bf16x8*)(sb + a_wave + p * A_PLANE + i * 1024 + foff);
      };
      auto load_b = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p][j] = *(const bf16x8*)(sb + b_wave + p * B_PLANE + j * 1024 + foff);
      };
      auto block = [&](int i, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
          const int pa = tt == 0 ? 2 : (tt == 2 || tt == 3) ? 1 : 0;
          const int pb = tt == 1 ? 2 : (tt == 2 || tt == 4) ? 1 : 0;
          if (tt < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][pa], bf[pb][j], accs[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][pa], bf[pb][j], acc[i][j], 0, 0, 0);
        }
      };
      constexpr int NR = 2 * WM, NC = 2 * WN, ND = NR > NC ? NR : NC;
      load_a(0); load_b(0);
#pragma unroll
      for (int d = 0; d < ND; ++d) {          // ring d: column d against the rows above it, then row d up to the diagonal
        if (d + 1 < NC) load_b(d + 1);        // (the fragments of ring d + 1 are on their way while ring d multiplies)
        if (d + 1 < NR) load_a(d + 1);
        if (d < NC) {
#pragma unroll
          for (int i = 0; i < (d < NR ? d : NR); ++i) block(i, d);
        }
        if (d < NR) {
#pragma unroll
          for (int j = 0; j <= (d < NC ? d : NC - 1); ++j) block(d, j);
        }
      }
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }
    X3WS_STAMP(2);
#ifdef IPRGAN_X3WS_TIMING
    if (threadIdx.x == 0) { const unsigned bl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if (bl < 8192) g_x3ws_ts[bl * 8 + 7] = bar_wait; }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // the epilogue reuses the ring
    if constexpr (PREF) wait_vmcnt<0>();
#pragma unroll
    for (int i = 0; i < 2 * WM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * WN; ++j) acc[i][j] += accs[i][j];
    X3WS_STAMP(3);
    pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true, f32x4[2 * WM][2 * WN], NW>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf, nullptr, 0u);
    X3WS_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X3WS_STAMP(6);
    return;
  }
  f32x16 acc[WM][WN], accs[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = accs[i][j][r] = 0.f;
  const int half = lane >> 5, l31 = lane & 31;
  unsigned foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = (unsigned)l31 * 64u + (unsigned)((2 * kk + half) ^ ((l31 >> 2) & 3)) * 16u;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's fragment reads of step t - 1 are complete
    __builtin_amdgcn_s_barrier();
    const char* sb = ldsc + cur * STAGE_BYTES;
    constexpr int NFB = MINW <= 2 ? 2 : 1;       // fragment buffers: both sub-steps up front where 256 registers allow
    bf16x8 af[NFB][3][WM], bf[NFB][3][WN];
    auto frags = [&](int kk) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[kk % NFB][p][i] = *(const bf16x8*)(sb + a_wave + p * A_PLANE + i * 2048 + foff[kk]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[kk % NFB][p][j] = *(const bf16x8*)(sb + b_wave + p * B_PLANE + j * 2048 + foff[kk]);
      }
    };
    frags(0);
    if (NFB == 2) frags(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (NFB == 1 && kk == 1) frags(1);
#pragma unroll
      for (int tt = 0; tt < 6; ++tt) {
        const int pa = tt == 0 ? 2 : (tt == 2 || tt == 3) ? 1 : 0;
        const int pb = tt == 1 ? 2 : (tt == 2 || tt == 4) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (tt < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk % NFB][pa][i], bf[kk % NFB][pb][j], accs[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk % NFB][pa][i], bf[kk % NFB][pb][j], acc[i][j], 0, 0, 0);
          }
      }
    }
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // the epilogue reuses the ring
  if constexpr (PREF) wait_vmcnt<0>();           // the prefetched operand of the fused derivative
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] += accs[i][j];
  pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true, f32x16[WM][WN], NW>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf, nullptr, 0u);
}

// Measured and not kept (round 5): the loader-wave ring in PERSISTENT blocks (gconv_x3wp_kernel: a block stays on its CU and walks a
// tile list in the XCD map's order; the loader waves see one unbroken stream of K steps and refill across tile boundaries, so the
// next tile's first stages land while the multiplying waves are in the epilogue; the epilogue stages through the one ring buffer
// that is free between a tile's last step and the refill behind the next tile's first barrier - 128x128 in two column halves;
// past the last tile the loaders issue zero-fill dummies so that the counted wait is one instruction on every step).  Bit-identical
// to tiles 36 / 37 on multi-round, multi-phase, ragged layers - and exactly as fast: scripts/probe/tile_overhead.py fits
// (time per tile) = a + b (K steps) with a = 8.3 us, b = 1.15 us against a = 9.7 us, b = 1.10 us for tile 36 (north-star layer
// 218.6 vs 219.7 TFLOP/s, D.conv4 198 vs 201; 4 % faster at 18 K steps, 2 % slower at 72: its two-half epilogue costs what the
// hidden prologue saves, and its K step measured 3 % longer).  Where a tile's time goes, from s_memtime stamps inside the
// 128x128 tile (scripts/probe/ws_phase_times.sh, profiles/r05_ws_phase_cycles.jsonl; shader cycles): K step 1 900 (the six-MFMA
// block alone: 1 536 - the loaders need ~155 cycles per LDS-DMA piece beside the multiplying waves, 12 pieces per step; the 256x64
// tile has 15 and steps in 2 600), entry -> first stage landed 5 700 (64-channel inputs: 10 900), accumulators -> LDS + barrier
// 1 760, row passes 4 000 with all eight waves (was ~6 000 with four), store acknowledge 600-1 200: 12 500 cycles = 6.5 K steps per
// tile outside the K loop, 8 % of a 72-step tile, 16 % at 36 steps, 34 % at 18.  Tried on the epilogue without effect on `a`:
// every pass's LDS row read ahead of the first pass, a padded LDS pitch (the 16x16 accumulator layout writes 4-way bank
// conflicts), s_setprio on the multiplying waves, HIP_FORCE_DEV_KERNARG.  Hiding the epilogue needs the accumulators parked in
// LDS while the next tile multiplies (64 KB beside the ring: only a two-stage ring fits, and the two-stage 128x128 tile steps in
// 1.17 instead of 1.09 us) - not built.
// What bounds the K step (same stamps; IPRGAN_X3WS_HALFDMA: the loaders skip every other piece): the 128x128 tile steps in 1 894
// cycles with HALF the pieces as with all of them (1 905) - its K step is the multiplying waves' own (1 546 cycles of MFMA + operand
// fragments: 96 KB of LDS reads beside 48 KB of DMA writes per step are 1 152 cycles of the LDS pipe, bunched behind the barrier) -
// while the 256x64 tile drops from 2 600-2 900 to 1 900-2 200: THAT tile is bound by what its loaders move.  Pieces by kind (=2: no
// weight pieces, =3: no activation pieces): 256x64 steps in 2 590-2 640 with its 12 + 3 pieces per loader, 2 050-2 150 with the
// 12 activation pieces alone (3 120 for the stride-2 gather of D.conv1: 260 cycles per piece), 1 881 with the 3 weight pieces
// alone (= the multiplying waves' floor): ~170 cycles per piece of either kind, one 1 KB piece per ~40 cycles per CU from four
// loaders = 25 B/cycle = 43 GB/s at the 1.73 GHz these kernels clock at; two blocks per CU (eight loaders) reach 52 GB/s.
// MI355X_MICROARCH.md has the ceiling: rows gathered from the XCD's L2 at 66-73 GB/s per CU by a CU that does nothing else,
// 52-62 % of the stream-alone rate beside computing waves - the 64-column tiles sit ON it, and the 128x128 tile (48 KB per step
// against 60) is balanced between it and its multiplying waves.  Only fewer bytes per MAC move these tiles (the halo form reads
// a 3x3 layer's activations once instead of nine times, but steps 16 channels per barrier: 80-115 TFLOP/s on the same layers).
// Inside the K-loop barrier a multiplying wave of the 128x128 tile spends 60-73 cycles per step (256x64: 710-770, waiting for its
// loaders; 128x64 two per CU: 1 300-1 400): the 128x128 tile loses nothing to synchronisation - its ~290 cycles per step beyond
// the MFMAs are operand waits on an LDS pipe that is 66 % busy (a FULL / FREE word handshake instead of the barrier has nothing
// to win there).  Built on top of that and removed: every loader publishes a "landed" count in an LDS word as soon as its share of
// a stage is there, and a multiplying wave that finds all four counts ahead reads the first fragment triple of step t + 1 under the
// last ring of step t (correct: the words make the early read safe) - 7 % SLOWER (K step 1.17 -> 1.25 us): 255 registers
// instead of 196, and the flag read plus the conditional reads serialise the tail of the step.
// With half the DMA the same cycles pass 14 % faster in wall time (1 705 -> 1 465 us on the north-star layer): the DMA's watts
// come out of the clock.  Reading the next step's first fragments under the last MFMAs of a step (unsynchronised timing probe)
// made the step LONGER (2 090); 64x64 tiles at two or three blocks per CU: slower on every layer.
// Measured and not kept (round 5): the same ring as a PERSISTENT kernel (gconv_x3pp_kernel: grid = resident slots, each block walks
// its tile list through the same XCD map and never drains the ring - the LDS-DMA "fill" side runs NSTAGE - 1 steps ahead of the
// multiply side and switches to the next tile's rows when its walk is exhausted, the epilogue stages through the one stage that
// is free between a tile's last step and the next refill (EpiGeom with one stage's bytes: 128x128 in two column halves), so no
// block turnover and no exposed prologue).  128x128 / 3 stages, 128x64 / 2 stages and 64x64 / 3 stages on four waves, bit-identical
// to tiles 19 / 21 / 23 on 80-image layers with multi-phase geometries and ragged tiles; the 256x128 tile on eight waves needs 327
// spilled registers next to the fill state.  Slower than the plain tiles on every DCGAN / north-star layer but two
// (profiles/r05_persistent_ring_bench.jsonl: north-star 184 -> 177, 175 -> 167, 144 -> 134 TFLOP/s; D.conv1 forward 144 -> 112):
// the hardware's block dispatch balances tiles of uneven duration better than a static tile list, and the first wait behind an
// epilogue has to drain its stores (vmcnt retires in order).  One finding worth keeping: two instantiations of the step lambda
// (refilling / draining) inside ONE loop make the register allocator shuttle the accumulators between AGPRs and VGPRs (128
// v_accvgpr moves per 48 MFMAs, 15-35 % slower) - a single always-refilling instantiation whose slots past the end carry
// zero-fill dummies has none.

// ---- the same ring on v_mfma_f32_16x16x32_bf16 ------------------------------------------------------------------------
// These kernels are POWER-bound: under gconv_x3p_kernel the chip clocks at 1.74 GHz (GRBM_GUI_ACTIVE / time, PMC pass) with
// the matrix pipe 67 % busy - six MFMAs per product block leave no headroom.  MI355X_MICROARCH.md measures the 16x16x32 form
// at 1.12-1.15x the FLOP/s of the 32x32x16 form at equal cycles per FLOP (operands re-read from LDS): less energy per FLOP,
// higher clock.  Same stages, same DMA, same two accumulators; fragments are rows l & 15 x chunk l >> 4 (one MFMA covers
// the whole 32-channel step), the LDS swizzle is the one that makes THOSE reads conflict-free.
template <int WGM, int WGN, int WM, int WN, int NSTAGE, bool STATS, bool PREF>
__global__ __launch_bounds__(WGM * WGN * 64) void gconv_x3p16_kernel(const GConvArgs a) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, NW = WGM * WGN;
  constexpr bool FDB = NW <= 4;                  // one wave per SIMD: 512 registers
  constexpr int RSA = BM / 16 / NW, RSB = BN / 16 / NW;        // 16-row pieces per plane this wave stages
  constexpr int L = 3 * (RSA + RSB);                           // LDS-DMA instructions per wave and stage
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, A_BYTES = 3 * A_PLANE, STAGE_BYTES = 3 * (BM + BN) * 64;
  static_assert(RSA >= 1 && RSB >= 1 && BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "every wave stages whole 16-row pieces");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4 && L * (NSTAGE - 2) <= 63, "ring depth / vmcnt field");
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

  // logical tile order as in gconv_kernel: n tiles fastest, then the sub-pixel phases, then m tiles
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const unsigned lq = lt / gridDim.y;
  const int pz = (int)(lq % gridDim.z);
  const int pM = a.ph[pz].M;
  const int m0 = (int)(lq / gridDim.z) * BM, n0 = (int)(lt % gridDim.y) * BN;
  if (m0 >= pM) {
    if (STATS) {
      for (int c = threadIdx.x; c < BN; c += NW * 64)
        if (n0 + c < a.Ns) { a.stat_part[((size_t)lq * 2) * a.Ns + n0 + c] = 0.f; a.stat_part[((size_t)lq * 2 + 1) * a.Ns + n0 + c] = 0.f; }
    }
    return;
  }
  const int p_tw = a.ph[pz].tw, p_th = a.ph[pz].th;
  const int nt = a.ph[pz].steps;                     // 32-deep K steps (Cs % 32 == 0: a step lies inside one tap)
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;
  const int p_dy0 = a.ph[pz].dy0, p_dx0 = a.ph[pz].dx0, p_dys = a.ph[pz].dys, p_dxs = a.ph[pz].dxs;
  const int p_wbase = a.ph[pz].wbase, p_wsy = a.ph[pz].wsy, p_wsx = a.ph[pz].wsx;
  const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
  const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;
  const unsigned in_ps = a.in_ps, wt_ps = a.wt_ps;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // provably wave-uniform (LDS-DMA base, M0)
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 2;                                          // row of a 16-row piece this lane copies
  // 16x16 fragments read rows l & 15, chunk l >> 4: chunk c of row r sits at position c ^ ((4 - (r >> 2)) & 3) - the four
  // 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}: row quads 0, 3 at chunk 0 and 1, 2 at chunk 1) hit 16 slots
  const unsigned sc = (unsigned)((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 16u;   // source chunk of this lane's position

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

  int aiy[RSA], aix[RSA];
  unsigned arow[RSA], wrow[RSB];
#pragma unroll
  for (int i = 0; i < RSA; ++i) {
    const int r = (i * NW + wave) * 16 + lrow;
    const int m = m0 + r;
    if (m < pM) {
      const int b = fdiv(m, d_plane);
      const int rem = m - b * plane;
      const int y = fdiv(rem, d_owg);
      const int x = rem - y * p_owg;
      aiy[i] = y * a.isy;
      aix[i] = x * a.isx;
      arow[i] = (unsigned)(((b * IH + aiy[i]) * IW + aix[i]) * Cs) * 2u + sc;
    } else {
      aiy[i] = ROW_INVALID; aix[i] = 0; arow[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < RSB; ++i) {
    const int r = (i * NW + wave) * 16 + lrow;
    wrow[i] = (unsigned)((n0 + r) * a.Kp) * 2u + sc;
  }

  // TWO accumulators per block: `acc` takes h h' only, `accs` the five small terms (<= 2^-7 of the product each).  The bf16
  // MFMA aligns its 16 products to the exponent of the accumulator input and truncates: every MFMA into a LARGE accumulator
  // costs about an ulp of it, however small its products (measured on K = 1152: one accumulator, six MFMAs per 16 k: rms
  // error 5.1e-7 of the result, against 3.0e-7 for the fp32 MFMA).  With the small terms summed among themselves the large
  // accumulator sees one MFMA per 16 k, the small one's roundings are 2^-7 of that, and the two meet once, in fp32, below.
  f32x4 acc[2 * WM][2 * WN], accs[2 * WM][2 * WN];         // 16x16 blocks: v_mfma_f32_16x16x32_bf16
#pragma unroll
  for (int i = 0; i < 2 * WM; ++i)
#pragma unroll
    for (int j = 0; j < 2 * WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = accs[i][j][r] = 0.f;

  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  // wave-uniform walk over (channel chunk, tap): taps inside a 32-channel chunk, so that consecutive K steps re-read the
  // same pixels, shifted (they stay in L2)
  // K-walk order: GConvArgs::korder as in gconv_x3p_kernel (bit 0: channel chunks inside a tap; bit 1: the taps of a
  // stride-2 gather grouped by parity)
  const bool chunk_fast = (a.korder & 1) != 0;
  const bool parity = (a.korder & 2) != 0 && a.isy == 2 && a.isx == 2 && p_th >= 2 && p_tw >= 2;
  // taps of parity class (cy, cx): ty = cy, cy + 2, ... < th (k4: two per class and axis; k3: two even, one odd)
  int u_c = 0, u_ty = 0, u_tx = 0;
  int q_cls = 0, q_dy = 0, q_dx = 0;
  auto tap_next = [&]() -> bool {
    bool wrapped = false;
    if (parity) {
      if (++q_dx == ((p_tw - (q_cls & 1) + 1) >> 1)) {
        q_dx = 0;
        if (++q_dy == ((p_th - (q_cls >> 1) + 1) >> 1)) { q_dy = 0; if (++q_cls == 4) { q_cls = 0; wrapped = true; } }
      }
      u_ty = (q_cls >> 1) + 2 * q_dy; u_tx = (q_cls & 1) + 2 * q_dx;
    } else if (++u_tx == p_tw) { u_tx = 0; if (++u_ty == p_th) { u_ty = 0; wrapped = true; } }
    return wrapped;
  };
  int w_dy = 0, w_dx = 0, w_tapoff = 0;
  unsigned w_wk = 0, w_sbase = 0;
  auto walk_begin = [&](int buf) {            // address pieces of the K step the walk points at, into stage `buf`
    w_dy = p_dy0 + u_ty * p_dys; w_dx = p_dx0 + u_tx * p_dxs;
    w_tapoff = ((w_dy * IW + w_dx) * Cs + u_c) * 2;
    w_wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * 2u;
    w_sbase = lds_base + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
  };
  auto walk_next = [&]() {
    if (chunk_fast) {
      u_c += 32;
      if (u_c == Cs) { u_c = 0; tap_next(); }
    } else if (tap_next()) u_c += 32;
  };
  // piece q of the stage: q < 3 * RSA: plane q / RSA of the activation rows of row set q % RSA; then the weight rows
  auto piece = [&](int q) {
    if (q < 3 * RSA) {
      const int p = q / RSA, i = q % RSA;
      const int iy = aiy[i] + w_dy, ix = aix[i] + w_dx;
      bool ok;
      unsigned off;
      if (reflect) {              // wave-uniform branch: the mirrored pixel instead of a zero
        ok = aiy[i] != ROW_INVALID;
        const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
        off = arow[i] + (unsigned)((((ry - aiy[i]) * IW + (rx - aix[i])) * Cs + u_c) * 2);
      } else {
        ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        off = arow[i] + (unsigned)w_tapoff;
      }
      dma16(rs_in, w_sbase + (unsigned)p * A_PLANE + (unsigned)(i * NW) * 1024u, ok ? off + (unsigned)p * in_ps : OOB_OFFSET);
    } else {
      const int p = (q - 3 * RSA) / RSB, i = (q - 3 * RSA) % RSB;
      dma16(rs_wt, w_sbase + A_BYTES + (unsigned)p * B_PLANE + (unsigned)(i * NW) * 1024u, wrow[i] + w_wk + (unsigned)p * wt_ps);
    }
  };
  auto issue = [&](int buf) {
    walk_begin(buf);
#pragma unroll
    for (int q = 0; q < L; ++q) piece(q);
    walk_next();
  };

  // fragment read offsets inside a plane image: row (..) * 32 + l31, chunk (2 kk + half) ^ swz(l31)
  const int l15 = lane & 15;
  const unsigned foff = (unsigned)l15 * 64u + (unsigned)((lane >> 4) ^ ((4 - ((l15 >> 2) & 3)) & 3)) * 16u;      // + 1024 per 16-row block
  const unsigned a_wave = (unsigned)(wm * WM) * 2048u, b_wave = A_BYTES + (unsigned)(wn * WN) * 2048u;
  const char* ldsc = (const char*)lds;

  // one K step on stage `cb`; ISS: the refill of stage `nb` is woven into the MFMA stream (one LDS-DMA instruction after
  // every (SPREAD / L)-th MFMA), so that a wave waits for one slot of the CU's address path at a time while its (and its
  // SIMD partner's) MFMAs run.  The fragments of sub-step 1 are read while sub-step 0 is multiplied.
  // one K step (32 channels = ONE 16x16x32 MFMA per term and block pair) on stage `cb`; ISS: the refill of stage `nb` is
  // woven into the MFMA stream.  The B fragments of the step stay in registers, the A fragments of a 16-row block are
  // read right before its 6 x 2 WN MFMAs.
  auto step = [&](int cb, int nb, auto ISS) {
    constexpr bool iss = decltype(ISS)::value;
    constexpr int NMF = 6 * 4 * WM * WN;                               // MFMAs per step
    constexpr int SPREAD = NSTAGE >= 3 ? NMF : (3 * NMF) / 4;
    const char* sb = ldsc + cb * STAGE_BYTES;
    if constexpr (iss) walk_begin(nb);
    // the product blocks along the staircase of gconv_x3ws_kernel's multiplying waves (one new fragment triple per group of blocks)
    bf16x8 af[2 * WM][3], bf[3][2 * WN];
    int q = 0, mi = 0;
    auto load_a = [&](int i) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = *(const bf16x8*)(sb + a_wave + p * A_PLANE + i * 1024 + foff);
    };
    auto load_b = [&](int j) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p][j] = *(const bf16x8*)(sb + b_wave + p * B_PLANE + j * 1024 + foff);
    };
    auto block = [&](int i, int j) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {           // l h', h l', m m', m h', h m' into the small accumulator, h h' into the large one
        const int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
        const int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
        if (t < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][pa], bf[pb][j], accs[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][pa], bf[pb][j], acc[i][j], 0, 0, 0);
        ++mi;
        if constexpr (iss) {
          if (q < L && q * SPREAD < mi * L) {
            piece(q);
            ++q;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    constexpr int NR = 2 * WM, NC = 2 * WN, ND = NR > NC ? NR : NC;
    load_a(0); load_b(0);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      if (d + 1 < NC) load_b(d + 1);
      if (d + 1 < NR) load_a(d + 1);
      if (d < NC) {
#pragma unroll
        for (int i = 0; i < (d < NR ? d : NR); ++i) block(i, d);
      }
      if (d < NR) {
#pragma unroll
        for (int j = 0; j <= (d < NC ? d : NC - 1); ++j) block(d, j);
      }
    }
    if constexpr (iss) {
#pragma unroll
      for (; q < L; ++q) piece(q);
      walk_next();
    }
  };

  // PREF: the fused-derivative operand of this thread's stores, loaded FIRST: vmcnt retires in order, so the wait for
  // stage 0 covers these loads (the same latency, once per tile) and every later counted wait is unaffected
  using G = EpiGeom<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES>;
  u32x4 auxpf[PREF ? G::NIT : 1];
  if constexpr (PREF) pipe_aux_load<G>(a, pz, m0, n0, 0, auxpf);

  // ---- the ring
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nt) issue(s);
  int cur = 0, nxt = NSTAGE - 1;                 // stage read at step t, stage refilled at step t (= read at t - 1)
  int t = 0;
  for (; t < nt - (NSTAGE - 1); ++t) {           // steps that refill a stage
    wait_stages<L>(NSTAGE - 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of step t - 1 are complete
    __builtin_amdgcn_s_barrier();
    step(cur, nxt, std::true_type{});
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  for (; t < nt; ++t) {                          // the last NSTAGE - 1 steps drain the ring
    const int rem = nt - 1 - t;
    wait_stages<L>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    step(cur, nxt, std::false_type{});
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // the epilogue reuses the ring
#pragma unroll
  for (int i = 0; i < 2 * WM; ++i)
#pragma unroll
    for (int j = 0; j < 2 * WN; ++j) acc[i][j] += accs[i][j];

  pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf);
}


// ---- halo form ------------------------------------------------------------------------------------------------
// gconv_x3p_kernel above streams an im2col view of the activation: every tap of a K step re-fetches its 256 tile rows
// from L2, the XCD's L2 delivers ~29 B/clk per CU, and a 256x128 tile needs 72 KB per 3072 MFMA cycles: the DMA path is
// 80 % busy and its time ADDS to the matrix time (measured with K-loop probes: 216 TFLOP/s with both, 290 without the
// refills).  For stride-1 gathers (Conv2d k3 s1 forward and backward-data; the sub-pixel phases of k4 s2 backward-data /
// ConvTranspose2d forward, which are 2x2-tap stride-1 convolutions on the small grid) the taps of one channel chunk read
// SHIFTED copies of the same pixels.  Here a tile is a spatial patch (16x16 positions of one image, or four whole 8x8
// maps); per 16-channel chunk its halo ((16+2)^2 pixels x 3 planes = 31 KB) is staged ONCE, double-buffered a chunk
// ahead, and every tap reads it at a row shift; only the weight rows of a tap (12 KB for 128 columns) stream through a
// ring.  L2 -> LDS traffic per tap step: 12 + 31 / 9 = 16 KB per 1536 MFMA cycles (k3) - 10 B/clk instead of 24.
//   * K step = one tap x 16 channels: LDS rows are 32 bytes (16 channels of one pixel / weight row of one plane), 8 rows per
//     256-byte bank line: chunk c of row r sits at position c ^ ((r >> 3) & 1);
//   * every wave issues exactly L = LB + LH LDS-DMA instructions per step (weight rows of step s + NSB - 1; a share of the
//     next chunk's halo; dummies past the end go to a scratch row set) so that ONE counted vmcnt per step covers both rings:
//     vmcnt retires in order, the wait for the weight stage of step s also covers every halo piece issued before it;
//   * the epilogue is pipe_epilogue with a row table (tile row -> output element), since tile rows are not consecutive
//     positions of the phase grid.
// Measured and not kept (round 4): the same kernel for STRIDE-2 gathers of k4 p1 kernels (Conv2d k4 s2 forward, ConvT k4 s2
// backward-data) - the 16 taps are 2 x 2 taps on each of the four residue images of the input, so the K walk becomes
// (residue, 16-channel chunk) x (2 x 2 taps) with one (PH + 1) x (PW + 1) halo per residue and chunk.  Parity-green, but
// slower than the im2col ring on every DCGAN layer (D.conv1 101 vs 107, D.conv3 115 vs 160, D.conv5 60 vs 174, G.up0
// backward-data 119 vs 203 TFLOP/s): a 16-channel step of a 64-column tile is 12 MFMAs per wave between two barriers.
struct X3HGeom {
  int PH, PW, NI;            // patch rows x columns per image and images per tile: NI * PH * PW = 256
  int lpw, lpp;              // log2(PW), log2(PH * PW)
  int tpy, tpx;              // patches per image (phase grid / patch)
  FastDiv d_tpi, d_tpx;      // / (tpy * tpx), / tpx
};

template <int WGN_, int WN, int NSB, int LH, int HSTEPS, bool STATS>
__global__ __launch_bounds__(512) void gconv_x3h_kernel(const GConvArgs a, const X3HGeom g) {
  constexpr int WGM = 4, WGN = WGN_, WM = 2, NW = 8, BM = 256, BN = WGN * WN * 32;
  static_assert(WGM * WGN == NW, "eight waves");
  constexpr int HP_MAX = 13, HPB = HP_MAX * 1024, HBUF = 3 * HPB;           // halo: <= 416 rows of 32 bytes per plane
  constexpr int BPL = BN * 32, BSTAGE = 3 * BPL;                            // weight stage: [plane][BN rows][32 bytes]
  constexpr int NBP = 3 * BN / 32, LB = (NBP + NW - 1) / NW, L = LB + LH;   // weight pieces per stage, per wave; slots per step
  constexpr int OFF_B = 2 * HBUF, OFF_TRASH = OFF_B + NSB * BSTAGE, OFF_TAB = (OFF_TRASH + NW * 1024 > BM * BN * 4 ? OFF_TRASH + NW * 1024 : BM * BN * 4);
  static_assert(OFF_TAB + 1024 <= 160 * 1024 && (NSB - 2) * L <= 63 && NSB >= 3, "LDS budget / vmcnt field / ring depth");
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  char* ldsc = (char*)lds;
  unsigned* rowtab = (unsigned*)(ldsc + OFF_TAB);

  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const unsigned lq = lt / gridDim.y;
  const int pz = (int)(lq % gridDim.z), tile = (int)(lq / gridDim.z), n0 = (int)(lt % gridDim.y) * BN;
  const Phase& ph = a.ph[pz];
  const int th = ph.th, tw = ph.tw, ntap = th * tw;
  const int dymin = ph.dys < 0 ? ph.dy0 + (th - 1) * ph.dys : ph.dy0, dxmin = ph.dxs < 0 ? ph.dx0 + (tw - 1) * ph.dxs : ph.dx0;
  const int HHp = g.PH + th - 1, HWp = g.PW + tw - 1, hrows = g.NI * HHp * HWp, hp = (hrows + 31) >> 5;
  const int bimg = (int)fdiv((unsigned)tile, g.d_tpi), trem = tile - bimg * (g.tpy * g.tpx);
  const int tyy = (int)fdiv((unsigned)trem, g.d_tpx), txx = trem - tyy * g.tpx;
  const int b0 = bimg * g.NI, y0 = tyy * g.PH, x0 = txx * g.PW;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l31 = lane & 31;

  // ---- row table: tile row r = (image ni, patch row py, patch column px) -> element offset of the output pixel
  if (tid < BM) {
    const int ni = tid >> g.lpp, rr = tid & ((1 << g.lpp) - 1), py = rr >> g.lpw, px = rr & (g.PW - 1);
    const int b = b0 + ni, y = y0 + py, x = x0 + px;
    unsigned e = OOB_OFFSET;
    if (b < a.B && y < ph.ohg && x < ph.owg) e = (unsigned)((b * a.OH + y * a.osy + ph.ooy) * a.OW + x * a.osx + ph.oox) * (unsigned)a.Ns;
    rowtab[tid] = e;
  }

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const unsigned lds_base = (unsigned)(uintptr_t)lds;

  // ---- halo slots of this wave: slot t (t < HSLOTS) of a chunk carries piece id (t / LH * NW + wave) * LH + t % LH
  // = (plane, 32 rows); the lane's source pixel does not depend on the chunk: byte offset of its channel 0, or OOB
  constexpr int HSLOTS = HSTEPS * LH;            // the first HSTEPS = ntap - NSB + 1 steps of a chunk carry LH slots each
  unsigned hsrc[HSLOTS];
  int hdst[HSLOTS];                              // LDS byte offset inside a halo buffer, or -1 (dummy)
#pragma unroll
  for (int t = 0; t < HSLOTS; ++t) {
    const int id = ((t / LH) * NW + wave) * LH + t % LH;
    hsrc[t] = OOB_OFFSET; hdst[t] = -1;
    if (id < 3 * hp) {
      const int p = id / hp, pc = id - p * hp;
      hdst[t] = p * HPB + pc * 1024;
      const int hr = pc * 32 + (lane >> 1);
      if (hr < hrows) {
        const int ni = hr / (HHp * HWp), r2 = hr - ni * (HHp * HWp), hy = r2 / HWp, hx = r2 - hy * HWp;
        int y = y0 + hy + dymin, x = x0 + hx + dxmin;
        const int b = b0 + ni;
        bool ok = b < a.B;
        if (reflect) { y = reflect_idx(y, IH); x = reflect_idx(x, IW); }
        ok = ok && (unsigned)y < (unsigned)IH && (unsigned)x < (unsigned)IW;
        if (ok) hsrc[t] = (unsigned)(((b * IH + y) * IW + x) * Cs) * 2u + (unsigned)p * a.in_ps +
                          (unsigned)((lane & 1) ^ ((hr >> 3) & 1)) * 16u;
      }
    }
  }
  // ---- weight slots: slot j carries piece wave * LB + j = (plane, 32 rows) of the stage
  unsigned wsrc[LB];
  int wdst[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int id = wave * LB + j;
    wsrc[j] = OOB_OFFSET; wdst[j] = -1;
    if (id < NBP) {
      const int p = id / (BN / 32), rs = id % (BN / 32), row = rs * 32 + (lane >> 1);
      wdst[j] = p * BPL + rs * 1024;
      wsrc[j] = (unsigned)((n0 + row) * a.Kp) * 2u + (unsigned)p * a.wt_ps + (unsigned)((lane & 1) ^ ((row >> 3) & 1)) * 16u;
    }
  }
  const unsigned trash = lds_base + OFF_TRASH + (unsigned)wave * 1024u;

  // fragment rows: A block i of this wave: tile row (wm * 2 + i) * 32 + l31 -> halo row of tap offset 0; B block j: weight row
  int hbase[WM];
  unsigned boff[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    const int ni = r >> g.lpp, rr = r & ((1 << g.lpp) - 1), py = rr >> g.lpw, px = rr & (g.PW - 1);
    hbase[i] = (ni * HHp + py) * HWp + px;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int r = (wn * WN + j) * 32 + l31;
    boff[j] = (unsigned)(r * 32) + (unsigned)(half ^ ((r >> 3) & 1)) * 16u;
  }

  f32x16 acc[WM][WN], accs[WM][WN];              // h h' | the five small terms (see gconv_x3p_kernel)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = accs[i][j][r] = 0.f;

  const int nchunk = Cs / 16, nsteps = nchunk * ntap;
  const int hlast = HSTEPS - 1;                  // last step of a chunk that carries halo slots of the next one: its first
                                                 // fragments are read one step before that chunk starts (host: ntap = NSB + HSTEPS - 1)
  // weight walk (NSB - 1 steps ahead of the compute walk) and compute walk: (chunk, ty, tx)
  int wc = 0, wty = 0, wtx = 0, ws_ = 0;
  auto issue_w = [&](int j) __attribute__((always_inline)) {
    const bool live = ws_ < nsteps && wdst[j] >= 0;
    const unsigned wk = (unsigned)((ph.wbase + wty * ph.wsy + wtx * ph.wsx) * Cs + wc * 16) * 2u;
    const unsigned dst = live ? lds_base + OFF_B + (unsigned)((ws_ % NSB) * BSTAGE + wdst[j]) : trash;
    dma16(rs_wt, __builtin_amdgcn_readfirstlane(dst), live ? wsrc[j] + wk : OOB_OFFSET);
  };
  auto walk_w = [&]() __attribute__((always_inline)) {
    ++ws_;
    if (++wtx == tw) { wtx = 0; if (++wty == th) { wty = 0; ++wc; } }
  };
  auto issue_h = [&](int t, int chunk) __attribute__((always_inline)) {        // slot t of the halo of `chunk`
    const bool live = chunk < nchunk && hdst[t] >= 0;
    const unsigned dst = live ? lds_base + (unsigned)((chunk & 1) * HBUF + hdst[t]) : trash;
    dma16(rs_in, __builtin_amdgcn_readfirstlane(dst), live && hsrc[t] != OOB_OFFSET ? hsrc[t] + (unsigned)(chunk * 32) : OOB_OFFSET);
  };

  // ---- prologue: the first chunk's halo and the first NSB - 1 weight stages, drained once (2 us per 100+ us tile)
#pragma unroll
  for (int t = 0; t < HSLOTS; ++t) issue_h(t, 0);
#pragma unroll
  for (int s = 0; s < NSB; ++s) {
#pragma unroll
    for (int j = 0; j < LB; ++j) issue_w(j);
    walk_w();
  }
  wait_vmcnt<0>();
  lds_barrier();

  // Software pipeline: the fragments of step s + 1 are read (halo row shift + weight stage) while step s is multiplied, so
  // the barrier at the top of step s hands over the stage of step s + 1 - and, since every wave has the stage of step s in
  // registers by then, it is the slot of step s that is refilled (with step s + NSB).  The first version read its
  // fragments right behind the barrier: all eight waves stalled on the same LDS round trip once per 24 MFMAs (205 TFLOP/s
  // on the north-star shape, behind the im2col tile).
  int cc = 0, cty = 0, ctx = 0, ck = 0;          // walk of the step whose fragments are read next
  bf16x8 af[2][3][WM], bf[2][3][WN];
  auto frags = [&](int sidx, int fb) __attribute__((always_inline)) {
    const char* hb = ldsc + (cc & 1) * HBUF;
    const char* sb = ldsc + OFF_B + (sidx % NSB) * BSTAGE;
    const int tapoff = (ph.dy0 + cty * ph.dys - dymin) * HWp + (ph.dx0 + ctx * ph.dxs - dxmin);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row = hbase[i] + tapoff;
      const char* pa = hb + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
#pragma unroll
      for (int p = 0; p < 3; ++p) af[fb][p][i] = *(const bf16x8*)(pa + p * HPB);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[fb][p][j] = *(const bf16x8*)(sb + p * BPL + boff[j]);
    if (++ctx == tw) { ctx = 0; if (++cty == th) { cty = 0; ++cc; } }
  };
  int kc = 0, kk = 0;                            // chunk / step-in-chunk of the step being MULTIPLIED (halo slot bookkeeping)
  frags(0, 0);
  auto one_step = [&](int s, auto FB) __attribute__((always_inline)) {
    constexpr int fb = decltype(FB)::value;
    // stage s + 1 has landed: it left at step s + 1 - NSB, the slots of the NSB - 2 steps since then may be in flight
    wait_vmcnt<(NSB - 2) * L>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave holds the fragments of step s
    __builtin_amdgcn_s_barrier();
    if (s + 1 < nsteps) frags(s + 1, fb ^ 1);
    int q = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
      const int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (t < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[fb][pa][i], bf[fb][pb][j], accs[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[fb][pa][i], bf[fb][pb][j], acc[i][j], 0, 0, 0);
        }
      if (t < L) {                               // the step's L slots ride inside the MFMA stream
        if (q < LB) issue_w(q);
        else {
          const int hslot = kk * LH + (q - LB);
          bool done = false;
#pragma unroll
          for (int u = 0; u < HSLOTS; ++u)
            if (!done && u == hslot) { issue_h(u, kk <= hlast ? kc + 1 : nchunk); done = true; }
          if (!done) dma16(rs_in, __builtin_amdgcn_readfirstlane(trash), OOB_OFFSET);
        }
        ++q;
      }
    }
    walk_w();
    if (++kk == ntap) { kk = 0; ++kc; }
  };
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    one_step(s, std::integral_constant<int, 0>{});
    one_step(s + 1, std::integral_constant<int, 1>{});
  }
  if (s < nsteps) one_step(s, std::integral_constant<int, 0>{});
  wait_vmcnt<0>();                               // (dummy slots of the draining steps: nothing may be in flight into the epilogue's tile)
  lds_barrier();
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] += accs[i][j];

  u32x4 nopf[1];
  pipe_epilogue<WGM, WGN, WM, WN, BM * BN * 4, STATS, false, false, true>(a, acc, (float*)lds, pz, lq, 0, n0, nopf, rowtab);
}

// Measured and not kept (round 5): the halo form WITH LOADER WAVES (gconv_x3wh_kernel: 128 positions - 8x16, 16x8 or two 8x8 maps - x
// 128 / 64 columns, the halo staged once per 32-channel chunk and double-buffered, 64-byte LDS rows in gconv_x3p16_kernel's swizzle
// computed per lane from the shifted halo row, one v_mfma_f32_16x16x32_bf16 per block, plane pair and (tap, chunk) step = 96 per wave
// between two barriers, four loaders issuing L = LB + LH pieces per step under one counted vmcnt; 28 instead of 48 KB from L2 per
// step of a 128x128 tile).  Correct at the first run (as close to float64 as every other tile, statistics and fused derivative
// included) and NOT faster: north-star layer 216-219 against 225-227 TFLOP/s for the im2col loader-wave tile, D.conv4 178 / 197
// against 211 / 220, the 64-column form 184 against 190 (two blocks per CU) - the 128x128 K step is the multiplying waves' own
// (half-DMA probe above) and the shifted halo reads cost them more than the aligned stage reads; with 118-154 KB of LDS only one
// block fits a CU.  What the bytes saved do buy is clock (the reflect-padded north-star layer, where the im2col tile pays for its
// reflected gathers: 214 against 210).

// ---- storage conversion: fp32 <-> three planes (iprgan_cast with kind 2) -------------------------------------------
// One thread per 8 elements: two 16-byte loads, three 16-byte stores (or the reverse); plane p at element offset p * ps.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n8, size_t ps) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v0 = *(const f32x4*)(src + i * 8), v1 = *(const f32x4*)(src + i * 8 + 4);
    bf16x4 t0[3], t1[3];
    split3_bf16(v0, t0);
    split3_bf16(v1, t1);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x2 w0 = __builtin_bit_cast(u32x2, t0[p]), w1 = __builtin_bit_cast(u32x2, t1[p]);
      *(u32x4*)(dst + p * ps + i * 8) = u32x4{w0.x, w0.y, w1.x, w1.y};
    }
  }
}
__global__ __launch_bounds__(256) void join3_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, size_t n8, size_t ps) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 h0, h1, m0, m1, l0, l1;
    unpack_bf16x8(*(const u32x4*)(src + i * 8), h0, h1);
    unpack_bf16x8(*(const u32x4*)(src + ps + i * 8), m0, m1);
    unpack_bf16x8(*(const u32x4*)(src + 2 * ps + i * 8), l0, l1);
    *(f32x4*)(dst + i * 8) = h0 + (m0 + l0);
    *(f32x4*)(dst + i * 8 + 4) = h1 + (m1 + l1);
  }
}
// tail elements (n % 8) and unaligned tensors: one element per thread
__global__ void split3_tail_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t i0, size_t n, size_t ps) {
  const size_t i = i0 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = src[i];
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 m = (__bf16)r1;
  dst[i] = h; dst[ps + i] = m; dst[2 * ps + i] = (__bf16)(r1 - (float)m);
}
__global__ void join3_tail_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, size_t i0, size_t n, size_t ps) {
  const size_t i = i0 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  dst[i] = (float)src[i] + ((float)src[ps + i] + (float)src[2 * ps + i]);
}

int cast_planes(const void* src, void* dst, size_t n, size_t ps, bool to_planes, hipStream_t st) {
  if (!n) return 0;
  const bool al = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (ps % 8) == 0;
  const size_t n8 = al ? n / 8 : 0;
  if (n8) {
    const size_t want = (n8 + 255) / 256;
    const unsigned grid = (unsigned)(want < 16384 ? want : 16384);
    if (to_planes) hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, st, (const float*)src, (__bf16*)dst, n8, ps);
    else hipLaunchKernelGGL(join3_kernel, dim3(grid), dim3(256), 0, st, (const __bf16*)src, (float*)dst, n8, ps);
    IPR_LAUNCH_CHECK();
  }
  const size_t i0 = n8 * 8;
  if (i0 < n) {
    const unsigned grid = (unsigned)((n - i0 + 255) / 256);
    if (to_planes) hipLaunchKernelGGL(split3_tail_kernel, dim3(grid), dim3(256), 0, st, (const float*)src, (__bf16*)dst, i0, n, ps);
    else hipLaunchKernelGGL(join3_tail_kernel, dim3(grid), dim3(256), 0, st, (const __bf16*)src, (float*)dst, i0, n, ps);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

// ---- host side -----------------------------------------------------------------------------------------------
bool gconv_x3p_eligible(const GConvArgs& a, bool ksplit_ok = false) {
  if (a.in16 != 2 || (a.ksplit > 1 && !ksplit_ok) || a.wmod > 0 || a.planar_M) return false;
  auto simple = [](int act) { return act == IPRGAN_ACT_NONE || act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; };
  if ((a.Ns % 8) != 0 || (a.Cs % 32) != 0 || !simple(a.act) || (a.aux && !simple(a.aux_act))) return false;    // pipe_epilogue
  if (a.bn_mean) return false;
  if (a.aux && a.aux16 == 2) return false;        // (callers pass the h plane as a bf16 operand: aux16 == 1)
  return true;
}

template <void (*KERN)(const GConvArgs), int SLOT = 29>
static void x3p_go(const GConvArgs& a, dim3 grid, dim3 block, size_t smem, hipStream_t st) {
  static bool attr_set = false;            // per kernel instantiation
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
  prof_launch(KERN, grid, block, smem, st, SLOT, a.flops, a);      // bench.py's roofline slots: one per __global__ name
}

template <int WGM, int WGN, int WM, int WN, int NSTAGE>
static int launch_x3p_t(const GConvArgs& a_in, hipStream_t st, int* bm_out) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  int maxM = 0;
  for (int i = 0; i < a_in.nphase; ++i) maxM = a_in.ph[i].M > maxM ? a_in.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = (size_t)NSTAGE * 3 * (BM + BN) * 64;
  dim3 grid(cdiv(maxM, BM), cdiv(a_in.Ns, BN), a_in.ksplit > 1 ? a_in.ksplit : a_in.nphase);
  *bm_out = BM;
  static const int korder = getenv("IPRGAN_X3P_KORDER") ? atoi(getenv("IPRGAN_X3P_KORDER")) : -1;      // A/B override
  GConvArgs a = a_in;
  // K-walk order (gconv_x3p_kernel: korder).  Stride-2 gathers (Conv2d k4 s2 forward, ConvTranspose2d k4 s2 backward-data)
  // walk their taps grouped by parity with the channel chunks innermost: measured on D.conv1 (64 -> 64 k4 s2 @64x64, batch
  // 128; profiles/r05_dconv1_korder_tcc.txt) bytes fetched from beyond the XCD's L2 803 -> 138 MB per launch (the tensor is
  // 201 MB: every pixel used to arrive four times), forward 99 -> 144 TFLOP/s; G.up2's backward-data 135 -> 188.  Stride-1
  // gathers keep taps-inside-chunk (chunks innermost costs the 64-column k3 layers 5-30 %).
  a.korder = korder >= 0 ? korder : ((a.isy == 2 && a.isx == 2) ? 3 : 0);
  constexpr bool can_pf = EpiGeom<WGM, WGN, WM, WN, NSTAGE * 3 * (BM + BN) * 64>::PF_FIRST;
  const bool pref = a.aux && a.aux16 == 1 && can_pf;
  const dim3 block(WGM * WGN * 64);
  if (a.stat_part) {
    if constexpr (can_pf) { if (pref) { x3p_go<gconv_x3p_kernel<WGM, WGN, WM, WN, NSTAGE, true, true>>(a, grid, block, smem, st); IPR_LAUNCH_CHECK(); return 0; } }
    x3p_go<gconv_x3p_kernel<WGM, WGN, WM, WN, NSTAGE, true, false>>(a, grid, block, smem, st);
  } else {
    if constexpr (can_pf) { if (pref) { x3p_go<gconv_x3p_kernel<WGM, WGN, WM, WN, NSTAGE, false, true>>(a, grid, block, smem, st); IPR_LAUNCH_CHECK(); return 0; } }
    x3p_go<gconv_x3p_kernel<WGM, WGN, WM, WN, NSTAGE, false, false>>(a, grid, block, smem, st);
  }
  IPR_LAUNCH_CHECK();
  return 0;
}

// warp-specialized form (gconv_x3ws_kernel): variant 0 = 256x64 (2 stages, 120 KB), 1 = 128x128 (3 stages, 144 KB),
// 2 = 128x64 (2 stages, 72 KB, two blocks per CU)
template <int WGM, int WGN, int WM, int WN, int NSTAGE, int MINW = 2, bool M16 = false>
static int launch_x3ws_t(const GConvArgs& a_in, hipStream_t st, int* bm_out) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  int maxM = 0;
  for (int i = 0; i < a_in.nphase; ++i) maxM = a_in.ph[i].M > maxM ? a_in.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = (size_t)NSTAGE * 3 * (BM + BN) * 64;
  dim3 grid(cdiv(maxM, BM), cdiv(a_in.Ns, BN), a_in.nphase);
  *bm_out = BM;
  static const int korder = getenv("IPRGAN_X3P_KORDER") ? atoi(getenv("IPRGAN_X3P_KORDER")) : -1;
  GConvArgs a = a_in;
  a.korder = korder >= 0 ? korder : ((a.isy == 2 && a.isx == 2) ? 3 : 0);       // (launch_x3p_t)
  constexpr bool can_pf = EpiGeom<WGM, WGN, WM, WN, NSTAGE * 3 * (BM + BN) * 64, WGM * WGN>::PF_FIRST;
  const bool pref = a.aux && a.aux16 == 1 && can_pf;
  // (x3p_go is templated on the kernel: the dynamic-LDS attribute is set once per INSTANTIATION - a generic lambda taking the
  // kernel as a function pointer would share one `static` flag between the four STATS x PREF variants)
  const dim3 block(512);
  if (a.stat_part) {
    if constexpr (can_pf) { if (pref) { x3p_go<gconv_x3ws_kernel<WGM, WGN, WM, WN, NSTAGE, true, true, MINW, M16>, 33>(a, grid, block, smem, st); IPR_LAUNCH_CHECK(); return 0; } }
    x3p_go<gconv_x3ws_kernel<WGM, WGN, WM, WN, NSTAGE, true, false, MINW, M16>, 33>(a, grid, block, smem, st);
  } else {
    if constexpr (can_pf) { if (pref) { x3p_go<gconv_x3ws_kernel<WGM, WGN, WM, WN, NSTAGE, false, true, MINW, M16>, 33>(a, grid, block, smem, st); IPR_LAUNCH_CHECK(); return 0; } }
    x3p_go<gconv_x3ws_kernel<WGM, WGN, WM, WN, NSTAGE, false, false, MINW, M16>, 33>(a, grid, block, smem, st);
  }
  IPR_LAUNCH_CHECK();
  return 0;
}
int launch_gconv_x3ws(const GConvArgs& a, int variant, hipStream_t st, int* bm_out) {
  static const bool enabled = !getenv("IPRGAN_X3WS") || atoi(getenv("IPRGAN_X3WS")) != 0;       // A/B switch
  if (!enabled || !gconv_x3p_eligible(a)) return -1;
  switch (variant) {
    case 0: return launch_x3ws_t<4, 1, 2, 2, 2>(a, st, bm_out);
    case 1: return a.Ns >= 128 ? launch_x3ws_t<2, 2, 2, 2, 3>(a, st, bm_out) : -1;
    case 2: return launch_x3ws_t<2, 2, 2, 1, 2, 4>(a, st, bm_out);             // 128x64, 72 KB: two blocks (sixteen waves) per CU
    case 3: return launch_x3ws_t<4, 1, 2, 2, 2, 2, true>(a, st, bm_out);       // 3-5: the same tiles on v_mfma_f32_16x16x32_bf16
    case 4: return a.Ns >= 128 ? launch_x3ws_t<2, 2, 2, 2, 3, 2, true>(a, st, bm_out) : -1;
    case 5: return launch_x3ws_t<2, 2, 2, 1, 2, 4, true>(a, st, bm_out);
    // (64x64 tiles on this kernel, three or two blocks per CU so that co-resident blocks overlap each other's prologue and epilogue:
    // 68-78 registers, parity-green, slower on every layer - "SR 64->64 @24x24" 96-101 against 109 TFLOP/s for the 128x64 tile)
    default: return -1;
  }
}

// ---- halo form: host side ----------------------------------------------------------------------------------------
static bool x3h_geom(const GConvArgs& a, X3HGeom& g, int* ntap_out) {
  if (a.in16 != 2 || a.isy != 1 || a.isx != 1 || a.ksplit > 1 || a.wmod > 0 || a.planar_M || a.rs0 || a.bn_mean) return false;
  auto simple = [](int act) { return act == IPRGAN_ACT_NONE || act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; };
  if ((a.Ns % 8) != 0 || (a.Cs % 16) != 0 || !simple(a.act) || (a.aux && (!simple(a.aux_act) || a.aux16 == 2))) return false;
  const Phase& p0 = a.ph[0];
  if (p0.M <= 0) return false;
  for (int i = 0; i < a.nphase; ++i) {            // every phase: the same grid, the same (square) tap count, unit tap steps
    const Phase& p = a.ph[i];
    if (p.ohg != p0.ohg || p.owg != p0.owg || p.th != p0.th || p.tw != p0.tw || p.th != p.tw) return false;
    if ((p.dys != 1 && p.dys != -1) || (p.dxs != 1 && p.dxs != -1)) return false;
  }
  const int k = p0.th;
  if (k != 2 && k != 3) return false;
  if (a.pad_mode == IPRGAN_PAD_REFLECT && (a.nphase != 1 || a.IH < 2 || a.IW < 2)) return false;
  memset(&g, 0, sizeof(g));
  if ((p0.ohg % 16) == 0 && (p0.owg % 16) == 0) { g.PH = 16; g.PW = 16; g.NI = 1; g.lpw = 4; g.lpp = 8; }
  else if (p0.ohg == 8 && p0.owg == 8) { g.PH = 8; g.PW = 8; g.NI = 4; g.lpw = 3; g.lpp = 6; }
  else return false;
  g.tpy = p0.ohg / g.PH; g.tpx = p0.owg / g.PW;
  g.d_tpi = make_fastdiv(g.tpy * g.tpx); g.d_tpx = make_fastdiv(g.tpx);
  if (g.NI * (g.PH + k - 1) * (g.PW + k - 1) > 13 * 32) return false;
  *ntap_out = k * k;
  return true;
}

template <int WGN, int WN, int NSB, int LH, int HSTEPS>
static int launch_x3h_t(const GConvArgs& a, const X3HGeom& g, hipStream_t st, int* bm_out) {
  constexpr int BN = WGN * WN * 32, NBP = 3 * BN / 32, LB = (NBP + 7) / 8;
  constexpr int OFF_TRASH = 2 * 3 * 13 * 1024 + NSB * 3 * BN * 32;
  constexpr int OFF_TAB = OFF_TRASH + 8 * 1024 > 256 * BN * 4 ? OFF_TRASH + 8 * 1024 : 256 * BN * 4;
  (void)LB;
  const size_t smem = OFF_TAB + 1024;
  const int mtiles = cdiv(a.B, g.NI) * g.tpy * g.tpx;
  dim3 grid(mtiles, cdiv(a.Ns, BN), a.nphase);
  *bm_out = 256;
  // (one attribute flag per kernel instantiation: both STATS variants have the same function-pointer type)
#define X3H_GO(K) { static bool s = false; if (!s) { (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); s = true; } \
                    prof_launch(K, grid, dim3(512), smem, st, 31, a.flops, a, g); }
  if (a.stat_part) { auto k = gconv_x3h_kernel<WGN, WN, NSB, LH, HSTEPS, true>; X3H_GO(k) }
  else { auto k = gconv_x3h_kernel<WGN, WN, NSB, LH, HSTEPS, false>; X3H_GO(k) }
#undef X3H_GO
  IPR_LAUNCH_CHECK();
  return 0;
}

// variant 0: 256 positions x 128 columns, 1: x 64 columns; 3x3 taps: 4 weight stages, the first 6 steps of a chunk carry one
// halo slot per wave; 2x2 taps (sub-pixel phases of k4 s2): 3 stages, the first 2 steps carry three
int launch_gconv_x3h(const GConvArgs& a, int variant, hipStream_t st, int* bm_out) {
  X3HGeom g;
  int ntap = 0;
  if (!x3h_geom(a, g, &ntap)) return -1;
  if (variant == 0) {
    if (a.Ns < 128) return -1;
    return ntap == 9 ? launch_x3h_t<2, 2, 4, 1, 6>(a, g, st, bm_out) : launch_x3h_t<2, 2, 3, 3, 2>(a, g, st, bm_out);
  }
  if (variant == 1) return ntap == 9 ? launch_x3h_t<2, 1, 4, 1, 6>(a, g, st, bm_out) : launch_x3h_t<2, 1, 3, 3, 2>(a, g, st, bm_out);
  return -1;
}

template <int WGM, int WGN, int WM, int WN, int NSTAGE>
static int launch_x3p16_t(const GConvArgs& a_in, hipStream_t st, int* bm_out) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  int maxM = 0;
  for (int i = 0; i < a_in.nphase; ++i) maxM = a_in.ph[i].M > maxM ? a_in.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = (size_t)NSTAGE * 3 * (BM + BN) * 64;
  dim3 grid(cdiv(maxM, BM), cdiv(a_in.Ns, BN), a_in.nphase);
  *bm_out = BM;
  const dim3 block(WGM * WGN * 64);
  static const int korder = getenv("IPRGAN_X3P_KORDER") ? atoi(getenv("IPRGAN_X3P_KORDER")) : -1;
  GConvArgs a = a_in;
  a.korder = korder >= 0 ? korder : ((a.isy == 2 && a.isx == 2) ? 3 : 0);       // (launch_x3p_t)
  if (a.stat_part) x3p_go<gconv_x3p16_kernel<WGM, WGN, WM, WN, NSTAGE, true, false>, 32>(a, grid, block, smem, st);
  else x3p_go<gconv_x3p16_kernel<WGM, WGN, WM, WN, NSTAGE, false, false>, 32>(a, grid, block, smem, st);
  IPR_LAUNCH_CHECK();
  return 0;
}
// the tiles of launch_gconv_x3p's variants 0, 1, 2, 4 on the 16x16x32 MFMA (variant 0..3 here)
int launch_gconv_x3p16(const GConvArgs& a, int variant, hipStream_t st, int* bm_out) {
  if (!gconv_x3p_eligible(a)) return -1;
  switch (variant) {
    case 0: return a.Ns >= 128 ? launch_x3p16_t<4, 2, 2, 2, 2>(a, st, bm_out) : -1;
    case 1: return a.Ns >= 128 ? launch_x3p16_t<2, 2, 2, 2, 3>(a, st, bm_out) : -1;
    case 2: return launch_x3p16_t<2, 2, 2, 1, 3>(a, st, bm_out);
    case 3: return launch_x3p16_t<4, 1, 2, 2, 2>(a, st, bm_out);
    default: return -1;
  }
}

// Measured and not kept (round 4): the 128x128 tile with 16-channel K steps (32-byte LDS rows, half the stage: three stages in
// 72 KB, so that TWO blocks share a CU and one block's barriers, prologue and epilogue run under the other's MFMAs; 193-231
// registers, no spills, parity-green): 155 TFLOP/s on the north-star layer against 188 for the same tile with 32-channel steps
// and one block per CU (variant 1) and 218 for the 256x128 tile; slower than variant 1 on every DCGAN-64 layer too.  Twenty-four
// MFMAs per wave between barriers cost more than the co-resident block returns.
// variant: 0 = 256x128 (8 waves of 64x64, 2 stages, 144 KB), 1 = 128x128 (4 waves of 64x64, 3 stages, 144 KB),
//          2 = 128x64 (4 waves of 64x32, 3 stages, 108 KB), 3 = 128x64 (2 stages, 72 KB: two blocks per CU),
//          4 = 256x64 (4 waves of 64x64, 2 stages, 120 KB), 5 = 64x64 (4 waves of 32x32, 3 stages, 72 KB: two blocks per CU),
//          6 = 128x256 (4 waves of 64x128, 2 stages, 144 KB), 7 = 128x128 (2 stages, 96 KB)
// returns -1 when the variant does not apply to the geometry
int launch_gconv_x3p(const GConvArgs& a, int variant, hipStream_t st, int* bm_out) {
  if (!gconv_x3p_eligible(a, true)) return -1;          // (the 32x32x16 ring takes a K split: blockIdx.z, partial tiles into slabs)
  switch (variant) {
    case 0: return a.Ns >= 128 ? launch_x3p_t<4, 2, 2, 2, 2>(a, st, bm_out) : -1;
    case 1: return a.Ns >= 128 ? launch_x3p_t<2, 2, 2, 2, 3>(a, st, bm_out) : -1;
    case 2: return launch_x3p_t<2, 2, 2, 1, 3>(a, st, bm_out);
    case 3: return launch_x3p_t<2, 2, 2, 1, 2>(a, st, bm_out);
    case 4: return launch_x3p_t<4, 1, 2, 2, 2>(a, st, bm_out);
    case 5: return launch_x3p_t<2, 2, 1, 1, 3>(a, st, bm_out);
    case 6: return a.Ns >= 256 ? launch_x3p_t<2, 2, 2, 4, 2>(a, st, bm_out) : -1;
    case 7: return a.Ns >= 128 ? launch_x3p_t<2, 2, 2, 2, 2>(a, st, bm_out) : -1;
    default: return -1;
  }
}

}  // namespace iprgan

#ifdef IPRGAN_X3WS_TIMING
extern "C" int iprgan_debug_x3ws_ts(unsigned long long* out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(iprgan::g_x3ws_ts), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
