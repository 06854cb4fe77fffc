// bf16 gather-GEMM tiles staged by LDS-DMA through a multi-stage ring (math mode "bf16 activations", BASELINE config 5:
// DCGAN 128x128 batch 256; layers networks/conv_generator.py:8,21 and networks/sn_discriminator.py:9-18 at mg = md = 16).
//
// Same contraction, geometry (GConvArgs / Phase) and epilogue as gconv_kernel<.., IN16> of conv_igemm.hip:
//     out[b, y*osy+ooy, x*osx+oox, n] = sum_{tap,c} in[b, y*isy+dy(tap), x*isx+dx(tap), c] * Wt[n][tap*Cs+c]
// with both operands bf16 in HBM, Cs % 64 == 0 and zero padding, so that one 64-deep K step lies inside one tap.
// What differs is how the operand tiles reach LDS and how the K loop is synchronised:
//   * `buffer_load_dwordx4 ... lds` (LDS-DMA): a wave-instruction moves 8 tile rows x 128 bytes straight from L2/HBM
//     into LDS - no staging VGPRs, no ds_write pass.  The LDS destination is lane-linear (base + 16 * lane), so the
//     bank-conflict swizzle is applied to the per-lane SOURCE address: LDS position p of row r holds source chunk
//     p ^ ((r >> 1) & 7); the fragment reads apply the same XOR.  Rows outside the image (zero padding, ragged M) use an
//     out-of-range buffer offset: the DMA writes zeros.
//   * NSTAGE stage buffers form a ring; NSTAGE-1 K steps of loads are in flight behind the MFMAs.  Per K step: one
//     counted `s_waitcnt vmcnt(N)` (this wave's part of the oldest stage has landed), ONE raw `s_barrier` (everybody's
//     part has; everybody is done reading the stage that is about to be refilled), issue the loads of step t+NSTAGE-1,
//     then the fragment reads and MFMAs of step t.  hipcc's own `__syncthreads()` would drain vmcnt to 0.
//   * block tiles of 256 rows (8 waves, 64x64 / 64x32 / 128x64 per wave): half the L2 -> LDS bytes per FLOP of the
//     128-row tiles.
//   * the epilogue goes through LDS (pipe_epilogue): the common register-layout epilogue of conv_shared.h costs ~1400
//     vector instructions per wave (quad transposes, per-row-group pixel arithmetic, the general activation switch) -
//     with the bf16 MFMA 16x faster than the fp32 one that was 2.7x the matrix time of a short-K layer (D.conv1
//     backward-data: 46 VALU per MFMA measured).  Here the accumulators are written to the (now free) ring as an fp32
//     [rows][columns] tile and every thread then owns 8 consecutive channels of one pixel per pass: no transposes, one
//     pixel address per 8 values, 16-byte loads of the fused-derivative operand (issued before the K loop, so they
//     cost no latency) and 16-byte stores of whole contiguous rows.
#include "conv_pipe_shared.h"

namespace iprgan {

// F32: fp32 operands in HBM and the exact fp32 MFMA (v_mfma_f32_32x32x2_f32): a tile row is 32 channels x 4 bytes - the
// same 128-byte rows, the same LDS image and chunk swizzle, a 32-deep K step.  The fp32 MFMA is 16x slower per
// fragment, so a K step is 4096 cycles of matrix work per wave against the same 48 KB of DMA: the ring hides it entirely
// and the kernel's job is to keep the matrix pipe issuing (no staging registers, no ds_write pass, one barrier per step).
// ReflectionPad2d (fp32 workloads: CycleGAN) is folded into the DMA source offsets.
template <int WGM, int WGN, int WM, int WN, int NSTAGE, bool STATS, bool PREF, bool F32 = false, bool BNM = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gconv_pipe_kernel(const GConvArgs a) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, NW = WGM * WGN;
  constexpr int ESZ = F32 ? 4 : 2, KSTEP = F32 ? 32 : 64;             // bytes per operand element, channels per K step
  constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW, L = LA + LB;          // LDS-DMA instructions per wave and stage
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;        // 64 bf16 = 128 bytes per tile row
  static_assert(LA >= 1 && LB >= 1 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave stages whole 8-row pieces");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
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
  const bool korder = (a.korder & 1) != 0, krot = (a.korder & 2) != 0;
  const bool weave = (a.korder & 16) != 0, weave32 = (a.korder & 32) != 0;
  const int nt = F32 ? a.ph[pz].steps : a.ph[pz].steps / 2;          // steps counts 32-deep K steps (bf16: Cs % 64 == 0)
  const bool reflect = F32 && a.pad_mode == IPRGAN_PAD_REFLECT;
  const int p_dy0 = a.ph[pz].dy0, p_dx0 = a.ph[pz].dx0, p_dys = a.ph[pz].dys, p_dxs = a.ph[pz].dxs;
  const int p_wbase = a.ph[pz].wbase, p_wsy = a.ph[pz].wsy, p_wsx = a.ph[pz].wsx;
  const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
  const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // provably wave-uniform (LDS-DMA base, M0)
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 3, lchunk = lane & 7;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

  // rows this lane stages: piece (i * NW + wave) = tile rows 8 * piece .. 8 * piece + 7, this lane row 8 * piece + lrow,
  // LDS position lchunk <- source chunk lchunk ^ swz(row)
  int aiy[LA], aix[LA];
  unsigned arow[LA], wrow[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int m = m0 + r;
    const unsigned sc = (unsigned)(lchunk ^ ((r >> 1) & 7)) * 16u;
    if (m < pM) {
      const int b = fdiv(m, d_plane);
      const int rem = m - b * plane;
      const int y = fdiv(rem, d_owg);
      const int x = rem - y * p_owg;
      aiy[i] = y * a.isy;
      aix[i] = x * a.isx;
      arow[i] = (unsigned)(((b * IH + aiy[i]) * IW + aix[i]) * Cs) * (unsigned)ESZ + sc;
    } else {
      aiy[i] = ROW_INVALID; aix[i] = 0; arow[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    wrow[i] = (unsigned)((n0 + r) * a.Kp) * (unsigned)ESZ + (unsigned)(lchunk ^ ((r >> 1) & 7)) * 16u;
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds_base = (unsigned)(uintptr_t)lds;                  // LDS byte address of the ring
  // wave-uniform walk over (channel chunk, tap).  Tile rows start at different chunks (krot): at any moment the blocks of an
  // XCD then fetch different 128-byte columns of the NHWC rows instead of all camping on the L2 channels of one column
  const int nchunk = Cs / KSTEP;
  int u_c = krot ? (int)(lq % (unsigned)nchunk) * KSTEP : 0, u_ty = 0, u_tx = 0, u_n = 0;
  // korder bit 6: a stride-2 gather walks its taps grouped by parity, channel chunks innermost - the four taps through which
  // one input pixel meets its four output positions follow each other, so its re-reads are L2 hits (conv_x3.hip:
  // gconv_x3p_kernel; measured there: bytes from beyond L2 / 5.8 on D.conv1)
  const bool s2walk = (a.korder & 64) != 0 && a.isy == 2 && a.isx == 2 && p_th >= 2 && p_tw >= 2;
  int q_cls = 0, q_dy = 0, q_dx = 0;
  if (s2walk) u_c = 0;
  auto walk_adv = [&]() {
    if (s2walk) {
      u_c += KSTEP;
      if (u_c >= Cs) {
        u_c = 0;
        if (++q_dx == ((p_tw - (q_cls & 1) + 1) >> 1)) { q_dx = 0; if (++q_dy == ((p_th - (q_cls >> 1) + 1) >> 1)) { q_dy = 0; ++q_cls; } }
        u_ty = (q_cls >> 1) + 2 * q_dy; u_tx = (q_cls & 1) + 2 * q_dx;
      }
    } else if (korder) {          // taps inside a channel chunk: consecutive K steps re-read the same pixels, shifted
      if (++u_tx == p_tw) { u_tx = 0; if (++u_ty == p_th) { u_ty = 0; u_c += KSTEP; if (u_c >= Cs) u_c = 0; } }
    } else {
      u_c += KSTEP;
      if (u_c >= Cs) u_c = 0;
      if (++u_n == nchunk) { u_n = 0; if (++u_tx == p_tw) { u_tx = 0; ++u_ty; } }
    }
  };
  // issue the LDS-DMA of the next K step of the walk into stage buffer `buf`
  auto issue = [&](int buf) {
    const int dy = p_dy0 + u_ty * p_dys, dx = p_dx0 + u_tx * p_dxs;
    const int tapoff = ((dy * IW + dx) * Cs + u_c) * ESZ;
    const unsigned wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * (unsigned)ESZ;
    const unsigned sbase = lds_base + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int iy = aiy[i] + dy, ix = aix[i] + dx;
      bool ok;
      unsigned off;
      if (reflect) {              // wave-uniform branch: the mirrored pixel instead of a zero
        ok = aiy[i] != ROW_INVALID;
        const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
        off = arow[i] + (unsigned)((((ry - aiy[i]) * IW + (rx - aix[i])) * Cs + u_c) * ESZ);
      } else {
        ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        off = arow[i] + (unsigned)tapoff;
      }
      dma16(rs_in, sbase + (unsigned)(i * NW) * 1024u, ok ? off : OOB_OFFSET);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i)
      dma16(rs_wt, sbase + A_BYTES + (unsigned)(i * NW) * 1024u, wrow[i] + wk);
    walk_adv();
  };

  // fragment read offsets: row (wm*WM+i)*32 + l31 of A / (wn*WN+j)*32 + l31 of B, chunk (2 kk + half) ^ swz(l31)
  const int half = lane >> 5, l31 = lane & 31;
  unsigned foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = (unsigned)l31 * 128u + (unsigned)((2 * kk + half) ^ ((l31 >> 1) & 7)) * 16u;
  const unsigned a_wave = (unsigned)(wm * WM) * 4096u, b_wave = A_BYTES + (unsigned)(wn * WN) * 4096u;
  const char* ldsc = (const char*)lds;
  auto compute = [&](int buf) {
    const char* sb = ldsc + buf * STAGE_BYTES;
    if constexpr (F32) {
      // chunk 2 kq + half = four consecutive k of this lane's row; MFMA e of the group takes element e of both halves
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        f32x4 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const f32x4*)(sb + a_wave + i * 4096 + foff[kq]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *(const f32x4*)(sb + b_wave + j * 4096 + foff[kq]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *(const bf16x8*)(sb + a_wave + i * 4096 + foff[kk]);
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[j] = *(const bf16x8*)(sb + b_wave + j * 4096 + foff[kk]);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  // bf16 K step with the refill of stage `nb` woven into the MFMA stream of stage `cb`.  Issued as one burst behind the
  // barrier (issue(); compute()), the L pieces of all eight waves queue up in the CU's one address path and every wave
  // sits in its issue slot until the burst has drained: measured on the 256x256 tile, K loop with only the DMA 57 us,
  // with only the MFMAs 85 us, with both 134 us - the two did not overlap.  Here a piece follows every (SPREAD / L)-th
  // MFMA, so a wave waits for one queue slot at a time while its (and its SIMD partner's) MFMAs run.  Two-stage rings
  // keep the pieces in the first half of the step (the rest of it is landing time before the next barrier).  The
  // fragments of sub-step kk + 1 are read while those of kk are multiplied.
  auto step_woven = [&](int cb, int nb, auto ISS) {
    constexpr bool iss = decltype(ISS)::value;
    constexpr int NMF = (F32 ? 4 : 1) * WM * WN, SPREAD = NSTAGE >= 3 ? 3 * NMF : 2 * NMF;     // MFMAs per sub-step
    const char* sb = ldsc + cb * STAGE_BYTES;
    int dy = 0, dx = 0, tapoff = 0;
    unsigned wk = 0, sbase = 0;
    if constexpr (iss) {
      dy = p_dy0 + u_ty * p_dys; dx = p_dx0 + u_tx * p_dxs;
      tapoff = ((dy * IW + dx) * Cs + u_c) * ESZ;
      wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * (unsigned)ESZ;
      sbase = lds_base + (unsigned)nb * STAGE_BYTES + (unsigned)wave * 1024u;
    }
    auto piece = [&](int q) {
      if (q < LA) {
        const int iy = aiy[q] + dy, ix = aix[q] + dx;
        bool ok;
        unsigned off;
        if (reflect) {
          ok = aiy[q] != ROW_INVALID;
          const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
          off = arow[q] + (unsigned)((((ry - aiy[q]) * IW + (rx - aix[q])) * Cs + u_c) * ESZ);
        } else {
          ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
          off = arow[q] + (unsigned)tapoff;
        }
        dma16(rs_in, sbase + (unsigned)(q * NW) * 1024u, ok ? off : OOB_OFFSET);
      } else {
        dma16(rs_wt, sbase + A_BYTES + (unsigned)((q - LA) * NW) * 1024u, wrow[q - LA] + wk);
      }
    };
    using frag_t = typename std::conditional<F32, f32x4, bf16x8>::type;       // 16 bytes of a tile row either way
    frag_t af[2][WM], bf[2][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) af[0][i] = *(const frag_t*)(sb + a_wave + i * 4096 + foff[0]);
#pragma unroll
    for (int j = 0; j < WN; ++j) bf[0][j] = *(const frag_t*)(sb + b_wave + j * 4096 + foff[0]);
    int q = 0, mi = 0;
    auto after_mfma = [&]() {
      ++mi;
      if constexpr (iss) {
        if (q < L && q * SPREAD < mi * L) {
          piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[(kk + 1) & 1][i] = *(const frag_t*)(sb + a_wave + i * 4096 + foff[kk + 1]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[(kk + 1) & 1][j] = *(const frag_t*)(sb + b_wave + j * 4096 + foff[kk + 1]);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if constexpr (F32) {
            const f32x4 x = (const f32x4&)af[kk & 1][i], y = (const f32x4&)bf[kk & 1][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc[i][j], 0, 0, 0);
            after_mfma();
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc[i][j], 0, 0, 0);
            after_mfma();
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc[i][j], 0, 0, 0);
            after_mfma();
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc[i][j], 0, 0, 0);
            after_mfma();
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((const bf16x8&)af[kk & 1][i], (const bf16x8&)bf[kk & 1][j], acc[i][j], 0, 0, 0);
            after_mfma();
          }
        }
    }
    if constexpr (iss) {
#pragma unroll
      for (; q < L; ++q) piece(q);
      walk_adv();
    }
  };

  // PREF: the fused-derivative operand of this thread's stores, loaded FIRST: vmcnt retires in order, so the wait for
  // stage 0 covers these loads (the same latency, once per tile) and every later counted wait is unaffected
  using G = EpiGeom<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES>;
  u32x4 auxpf[G::NIT];
  if constexpr (PREF) pipe_aux_load<G>(a, pz, m0, n0, 0, auxpf);

  // ---- the ring
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nt) issue(s);
  int cur = 0, nxt = NSTAGE - 1;                 // stage read at step t, stage refilled at step t (= read at t-1)
  if ((F32 && !weave32) || !weave) {
    for (int t = 0; t < nt; ++t) {
      const int rem = nt - 1 - t;                  // K steps after this one
      wait_stages<L>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of step t-1 are complete
      __builtin_amdgcn_s_barrier();
      if (rem >= NSTAGE - 1) issue(nxt);
      compute(cur);
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
      nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
  } else {
    int t = 0;
    for (; t < nt - (NSTAGE - 1); ++t) {           // steps that refill a stage
      wait_stages<L>(NSTAGE - 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      step_woven(cur, nxt, std::true_type{});
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
      nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
    for (; t < nt; ++t) {                          // the last NSTAGE - 1 steps drain the ring
      const int rem = nt - 1 - t;
      wait_stages<L>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      step_woven(cur, nxt, std::false_type{});
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // the epilogue reuses the ring

  pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, BNM>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf);
}

// ---- 256x256 tile with a half-tile ring ---------------------------------------------------------------------------
// gconv_pipe_kernel's 256x256 tile refills a whole 64 KB stage per K step: with two stages at most one step of DMA is in
// flight, all of it issued in one burst, and every step ends in a wait for its slowest piece (measured, K loop only:
// DMA alone 57 us, MFMAs alone 85 us, together 134 us).  Here the two K-tile buffers are cut into half-tiles of 128 rows
// (A0 / A1: the first / second 64 rows of each wave row; B0 / B1: the first / second 32 columns of each wave column),
// a K tile is multiplied quadrant by quadrant (A0 B0, A0 B1, A1 B1, A1 B0; the fragments of a half are read from LDS
// once and stay in registers for both of its quadrants), and each of the four phases refills the half-tile slot that
// was last read one phase earlier.  The DMA stream is therefore in half-tile order A0 B0 B1 A1 of K tile 0, 1, 2, ...;
// phase p of K tile t issues half-tile 4t + 7 + p, waits (counted vmcnt, never 0) for half-tile 4t + 1 + p to have
// landed and leaves FIVE younger half-tiles (80 KB) in flight across the barrier.  One barrier per phase covers both
// hazards: every wave has waited for its own pieces of the half-tile read next, and every wave has finished the reads
// (they precede its MFMAs) of the half-tile that is overwritten next.
template <bool STATS, bool PREF>
__global__ __launch_bounds__(512) void gconv_pipe8_kernel(const GConvArgs a) {
  constexpr int WGM = 2, WGN = 4, WM = 4, WN = 2, NW = 8, BM = 256, BN = 256;
  constexpr int HT = 16384, BUF = 4 * HT;                // half-tile, K-tile buffer (slots in stream order A0 B0 B1 A1)
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

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
  const int nt = a.ph[pz].steps / 2;                   // 64-channel K tiles
  const int total = 4 * nt;                            // half-tiles of the stream
  const int p_dy0 = a.ph[pz].dy0, p_dx0 = a.ph[pz].dx0, p_dys = a.ph[pz].dys, p_dxs = a.ph[pz].dxs;
  const int p_wbase = a.ph[pz].wbase, p_wsy = a.ph[pz].wsy, p_wsx = a.ph[pz].wsx;
  const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
  const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 3, lchunk = lane & 7;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

  // what this lane stages: piece i of a half-tile = its rows 8 * (8 i + wave) .. + 7, this lane row rho = 8 (8 i + wave) + lrow,
  // 16-byte position lchunk <- source chunk lchunk ^ swz(rho).  A half h row rho = tile row (rho / 64) * 128 + 64 h + rho % 64,
  // B half h row rho = tile column (rho / 32) * 64 + 32 h + rho % 32.
  int aiy[2][2], aix[2][2];
  unsigned arow[2][2], wrow[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rho = (i * 8 + wave) * 8 + lrow;
      const unsigned sc = (unsigned)(lchunk ^ ((rho >> 1) & 7)) * 16u;
      const int m = m0 + (rho >> 6) * 128 + h * 64 + (rho & 63);
      if (m < pM) {
        const int b = fdiv(m, d_plane);
        const int rem = m - b * plane;
        const int y = fdiv(rem, d_owg);
        const int x = rem - y * p_owg;
        aiy[h][i] = y * a.isy;
        aix[h][i] = x * a.isx;
        arow[h][i] = (unsigned)(((b * IH + aiy[h][i]) * IW + aix[h][i]) * Cs) * 2u + sc;
      } else {
        aiy[h][i] = ROW_INVALID; aix[h][i] = 0; arow[h][i] = 0;
      }
      const int c = (rho >> 5) * 64 + h * 32 + (rho & 31);
      wrow[h][i] = (unsigned)((n0 + c) * a.Kp) * 2u + sc;
    }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  const bool korder = (a.korder & 1) != 0;
  const int nchunk = Cs / 64;
  struct Walk { int c, ty, tx, n; };
  auto advance = [&](Walk& w) {
    if (korder) {
      if (++w.tx == p_tw) { w.tx = 0; if (++w.ty == p_th) { w.ty = 0; w.c += 64; } }
    } else {
      w.c += 64;
      if (++w.n == nchunk) { w.n = 0; w.c = 0; if (++w.tx == p_tw) { w.tx = 0; ++w.ty; } }
    }
  };
  // the two pieces of half-tile X (0: A0, 1: B0, 2: B1, 3: A1) of the K tile at walk position w, into buffer `buf`;
  // piece i is issued by issue_piece<X>(.., i)
  auto issue_piece = [&](auto XC, int buf, const Walk& w, int i) {
    constexpr int X = decltype(XC)::value;
    const unsigned dst = lds_base + (unsigned)buf * BUF + (unsigned)X * HT + (unsigned)(i * 8 + wave) * 1024u;
    if constexpr (X == 0 || X == 3) {
      constexpr int h = X == 3;
      const int dy = p_dy0 + w.ty * p_dys, dx = p_dx0 + w.tx * p_dxs;
      const int tapoff = ((dy * IW + dx) * Cs + w.c) * 2;
      const int iy = aiy[h][i] + dy, ix = aix[h][i] + dx;
      const bool ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
      dma16(rs_in, dst, ok ? arow[h][i] + (unsigned)tapoff : OOB_OFFSET);
    } else {
      constexpr int h = X == 2;
      const unsigned wk = (unsigned)((p_wbase + w.ty * p_wsy + w.tx * p_wsx) * Cs + w.c) * 2u;
      dma16(rs_wt, dst, wrow[h][i] + wk);
    }
  };

  // fragment reads: A half h row wm * 64 + i * 32 + l31, B half h row wn * 32 + l31, chunk (2 kk + half) ^ swz(l31)
  const int half = lane >> 5, l31 = lane & 31;
  unsigned foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = (unsigned)l31 * 128u + (unsigned)((2 * kk + half) ^ ((l31 >> 1) & 7)) * 16u;
  const char* ldsc = (const char*)lds;
  const unsigned a_wave = (unsigned)(wm * 64) * 128u, b_wave = (unsigned)(wn * 32) * 128u;
  bf16x8 af[2][4], b0[4], b1[4];
  auto read_a = [&](int buf, int h) {
    const char* sb = ldsc + buf * BUF + (h ? 3 * HT : 0) + a_wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) af[i][kk] = *(const bf16x8*)(sb + i * 4096 + foff[kk]);
  };
  auto read_b = [&](int buf, int h, bf16x8 (&bf)[4]) {
    const char* sb = ldsc + buf * BUF + (h ? 2 * HT : HT) + b_wave;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bf[kk] = *(const bf16x8*)(sb + foff[kk]);
  };
  // quadrant (hA, hB) with the two pieces of the refill woven behind the first MFMAs
  auto quad = [&](int hA, int hB, const bf16x8 (&bf)[4], auto ISSUE) {
    int n = 0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[hA * 2 + i][hB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][kk], bf[kk], acc[hA * 2 + i][hB], 0, 0, 0);
        if (n < 2) { ISSUE(n); __builtin_amdgcn_sched_barrier(0); }
        ++n;
      }
  };
  auto wait_dyn = [&](int halves) {        // at most `halves` younger half-tiles (2 pieces each) still in flight
    if (halves >= 5) wait_vmcnt<10>();
    else if (halves == 4) wait_vmcnt<8>();
    else if (halves == 3) wait_vmcnt<6>();
    else if (halves == 2) wait_vmcnt<4>();
    else if (halves == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
  };
  auto barrier = [&]() {                   // (the fragment reads of the previous phase fed its MFMAs: long complete)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  using G = EpiGeom<WGM, WGN, WM, WN, 2 * BUF>;
  u32x4 auxpf[G::NIT];
  if constexpr (PREF) pipe_aux_load<G>(a, pz, m0, n0, 0, auxpf);

  // ---- prologue: half-tiles 0 .. 6 (K tile 0 and A0 B0 B1 of K tile 1)
  Walk w1{0, 0, 0, 0}, w2{0, 0, 0, 0};
  {
    issue_piece(std::integral_constant<int, 0>{}, 0, w1, 0); issue_piece(std::integral_constant<int, 0>{}, 0, w1, 1);
    issue_piece(std::integral_constant<int, 1>{}, 0, w1, 0); issue_piece(std::integral_constant<int, 1>{}, 0, w1, 1);
    issue_piece(std::integral_constant<int, 2>{}, 0, w1, 0); issue_piece(std::integral_constant<int, 2>{}, 0, w1, 1);
    issue_piece(std::integral_constant<int, 3>{}, 0, w1, 0); issue_piece(std::integral_constant<int, 3>{}, 0, w1, 1);
    advance(w1);                          // K tile 1
    if (nt > 1) {
      issue_piece(std::integral_constant<int, 0>{}, 1, w1, 0); issue_piece(std::integral_constant<int, 0>{}, 1, w1, 1);
      issue_piece(std::integral_constant<int, 1>{}, 1, w1, 0); issue_piece(std::integral_constant<int, 1>{}, 1, w1, 1);
      issue_piece(std::integral_constant<int, 2>{}, 1, w1, 0); issue_piece(std::integral_constant<int, 2>{}, 1, w1, 1);
    }
    w2 = w1;
    advance(w2);                          // K tile 2
  }
  int issued = nt > 1 ? 7 : 4;

  // one K tile; STEADY: all four refills exist and exactly five half-tiles are in flight behind each wait
  auto ktile = [&](int t, auto STEADY) {
    constexpr bool steady = decltype(STEADY)::value;
    const int cb = t & 1, ob = cb ^ 1, base = 4 * t;
    // phase 0: A0 B0
    if constexpr (steady) wait_vmcnt<10>(); else wait_dyn(issued - (base + 2));
    barrier();
    read_a(cb, 0);
    read_b(cb, 0, b0);
    {
      const bool go = steady || base + 7 < total;
      quad(0, 0, b0, [&](int i) { if (go) issue_piece(std::integral_constant<int, 3>{}, ob, w1, i); });
      if (!steady && go) ++issued;
    }
    // phase 1: A0 B1
    if constexpr (steady) wait_vmcnt<10>(); else wait_dyn(issued - (base + 3));
    barrier();
    read_b(cb, 1, b1);
    {
      const bool go = steady || base + 8 < total;
      quad(0, 1, b1, [&](int i) { if (go) issue_piece(std::integral_constant<int, 0>{}, cb, w2, i); });
      if (!steady && go) ++issued;
    }
    // phase 2: A1 B1
    if constexpr (steady) wait_vmcnt<10>(); else wait_dyn(issued - (base + 4));
    barrier();
    read_a(cb, 1);
    {
      const bool go = steady || base + 9 < total;
      quad(1, 1, b1, [&](int i) { if (go) issue_piece(std::integral_constant<int, 1>{}, cb, w2, i); });
      if (!steady && go) ++issued;
    }
    // phase 3: A1 B0 (registers only; the slot of B1 was last read in phase 1, two barriers ago)
    {
      const bool go = steady || base + 10 < total;
      quad(1, 0, b0, [&](int i) { if (go) issue_piece(std::integral_constant<int, 2>{}, cb, w2, i); });
      if (!steady && go) ++issued;
    }
    w1 = w2;
    advance(w2);
  };

  int t = 0;
  for (; t < nt - 2; ++t) ktile(t, std::true_type{});
  if (nt > 2) issued = 4 * t + 7;
  for (; t < nt; ++t) ktile(t, std::false_type{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // the epilogue reuses the ring

  pipe_epilogue<WGM, WGN, WM, WN, 2 * BUF, STATS, PREF>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf);
}

// ---- persistent form -------------------------------------------------------------------------------------------
// One block per CU walks a list of tiles.  What it buys over one-tile blocks (measured on the 9-step 64 -> 128 k3 layer,
// 256x128 tile, one block per CU: the block lived 34 k cycles for 9.2 k cycles of MFMA; 39 % of its wave cycles were
// parked in s_waitcnt / s_barrier): the first stages of the NEXT tile are issued before the epilogue of the current one
// (the ring is free by then; the epilogue works in its own LDS region), so the DMA latency of a tile's prologue and
// the HBM / L2 idle time of an epilogue disappear behind each other.
//   FWD (no fused derivative, no residual, bf16 output - every forward pass): bias, activation and the column sums
//   are applied in the accumulator layout, where a lane owns ONE channel (one bias value, two running sums per column
//   block); the tile goes to LDS as bf16 and leaves as a plain copy: 16-byte LDS read -> 16-byte store of whole rows.
//   !FWD: pipe_epilogue on a 64 KB fp32 region (256x128 tiles in two column halves), operand of the fused derivative
//   prefetched for the next tile while the current one is stored.
template <int WGM, int WGN, int WM, int WN, int NSTAGE, bool STATS, bool FWD, bool F32 = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gconv_pipe2_kernel(const GConvArgs a, int ntn, int nphz, int total) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, NW = WGM * WGN, NT = NW * 64;
  constexpr int ESZ = F32 ? 4 : 2, KSTEP = F32 ? 32 : 64;         // F32: fp32 operands, exact fp32 MFMA (as gconv_pipe_kernel)
  static_assert(!(F32 && FWD), "the accumulator-layout epilogue stores bf16");
  constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW, L = LA + LB;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128, RING = NSTAGE * STAGE_BYTES;
  constexpr int T_BYTES = 64 * 1024;
  static_assert(RING + T_BYTES <= 160 * 1024 && (!FWD || BM * BN * 2 <= T_BYTES), "LDS budget");
  using G = EpiGeom<WGM, WGN, WM, WN, T_BYTES>;
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  char* ldsc = (char*)lds;
  float* T = (float*)(ldsc + RING);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 3, lchunk = lane & 7, half = lane >> 5, l31 = lane & 31;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;
  const bool reflect = F32 && a.pad_mode == IPRGAN_PAD_REFLECT;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  const int myslot = (int)xcd_remap(blockIdx.x, gridDim.x);           // blocks of one XCD take neighbouring tiles of a round

  // ---- state of the tile whose operands are being loaded
  int l_pz = 0, l_m0 = 0, l_n0 = 0, l_nt = 0;
  unsigned l_lq = 0;
  int p_tw = 1, p_dy0 = 0, p_dx0 = 0, p_dys = 0, p_dxs = 0, p_wbase = 0, p_wsy = 0, p_wsx = 0;
  int aiy[LA], aix[LA];
  unsigned arow[LA], wrow[LB];
  int u_c = 0, u_ty = 0, u_tx = 0;
  // next tile of this block at or after round `it` that has rows (tiles past the end of a short phase only owe a row of
  // zero partials); returns false when the list is exhausted
  auto next_tile = [&](int& it) -> bool {
    for (;; ++it) {
      const long long lt = (long long)it * gridDim.x + myslot;
      if (lt >= total) return false;
      const unsigned lq = (unsigned)(lt / ntn);
      const int pz = (int)(lq % (unsigned)nphz), m0 = (int)(lq / (unsigned)nphz) * BM, n0 = (int)(lt % ntn) * BN;
      if (m0 >= a.ph[pz].M) {
        if (STATS)
          for (int c = tid; c < BN; c += NT)
            if (n0 + c < a.Ns) { a.stat_part[((size_t)lq * 2) * a.Ns + n0 + c] = 0.f; a.stat_part[((size_t)lq * 2 + 1) * a.Ns + n0 + c] = 0.f; }
        continue;
      }
      l_pz = pz; l_m0 = m0; l_n0 = n0; l_lq = lq; l_nt = F32 ? a.ph[pz].steps : a.ph[pz].steps / 2;
      return true;
    }
  };
  auto setup_rows = [&]() {
    const Phase& ph = a.ph[l_pz];
    p_tw = ph.tw; p_dy0 = ph.dy0; p_dx0 = ph.dx0; p_dys = ph.dys; p_dxs = ph.dxs;
    p_wbase = ph.wbase; p_wsy = ph.wsy; p_wsx = ph.wsx;
    const int plane = ph.ohg * ph.owg;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = (i * NW + wave) * 8 + lrow;
      const int m = l_m0 + r;
      const unsigned sc = (unsigned)(lchunk ^ ((r >> 1) & 7)) * 16u;
      if (m < ph.M) {
        const int b = fdiv(m, ph.d_plane);
        const int rem = m - b * plane;
        const int y = fdiv(rem, ph.d_owg);
        const int x = rem - y * ph.owg;
        aiy[i] = y * a.isy;
        aix[i] = x * a.isx;
        arow[i] = (unsigned)(((b * IH + aiy[i]) * IW + aix[i]) * Cs) * (unsigned)ESZ + sc;
      } else {
        aiy[i] = ROW_INVALID; aix[i] = 0; arow[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int r = (i * NW + wave) * 8 + lrow;
      wrow[i] = (unsigned)((l_n0 + r) * a.Kp) * (unsigned)ESZ + (unsigned)(lchunk ^ ((r >> 1) & 7)) * 16u;
    }
    u_c = 0; u_ty = 0; u_tx = 0;
  };
  auto issue = [&](int buf) {
    const int dy = p_dy0 + u_ty * p_dys, dx = p_dx0 + u_tx * p_dxs;
    const int tapoff = ((dy * IW + dx) * Cs + u_c) * ESZ;
    const unsigned wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * (unsigned)ESZ;
    const unsigned sbase = lds_base + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int iy = aiy[i] + dy, ix = aix[i] + dx;
      bool ok;
      unsigned off;
      if (reflect) {              // wave-uniform branch: the mirrored pixel instead of a zero
        ok = aiy[i] != ROW_INVALID;
        const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
        off = arow[i] + (unsigned)((((ry - aiy[i]) * IW + (rx - aix[i])) * Cs + u_c) * ESZ);
      } else {
        ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        off = arow[i] + (unsigned)tapoff;
      }
      dma16(rs_in, sbase + (unsigned)(i * NW) * 1024u, ok ? off : OOB_OFFSET);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i)
      dma16(rs_wt, sbase + A_BYTES + (unsigned)(i * NW) * 1024u, wrow[i] + wk);
    u_c += KSTEP;
    if (u_c >= Cs) { u_c = 0; if (++u_tx == p_tw) { u_tx = 0; ++u_ty; } }
  };
  auto prologue = [&]() {            // first NSTAGE - 1 stages of the loading tile into ring slots 0 ..
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < l_nt) issue(s);
  };

  unsigned foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = (unsigned)l31 * 128u + (unsigned)((2 * kk + half) ^ ((l31 >> 1) & 7)) * 16u;
  const unsigned a_wave = (unsigned)(wm * WM) * 4096u, b_wave = A_BYTES + (unsigned)(wn * WN) * 4096u;
  f32x16 acc[WM][WN];
  auto compute = [&](int buf) {
    const char* sb = ldsc + buf * STAGE_BYTES;
    if constexpr (F32) {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        f32x4 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const f32x4*)(sb + a_wave + i * 4096 + foff[kq]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *(const f32x4*)(sb + b_wave + j * 4096 + foff[kq]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *(const bf16x8*)(sb + a_wave + i * 4096 + foff[kk]);
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[j] = *(const bf16x8*)(sb + b_wave + j * 4096 + foff[kk]);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  constexpr int FWD_STORES = BM / (NT / (BN / 8));           // 16-byte output stores per lane and tile (FWD copy-out)
  bool first_tile = true;
  int it = 0;
  if (!next_tile(it)) return;
  setup_rows();
  // fused-derivative operand of the loading tile, prefetched a tile ahead when one column half covers the tile (the
  // 256x128 tile would need 64 more registers for two halves in flight twice: its epilogue loads per half instead)
  constexpr bool PFN = !FWD && G::NH == 1;
  u32x4 auxn[PFN ? G::NIT : 1];
  const bool use_aux = PFN && a.aux && a.aux16;
  if constexpr (PFN) {
    if (use_aux) pipe_aux_load<G>(a, l_pz, l_m0, l_n0, 0, auxn);
  }
  prologue();
  for (;;) {
    // ---- K loop of the current tile (its first stages are in flight)
    const int c_pz = l_pz, c_m0 = l_m0, c_n0 = l_n0, nt = l_nt;
    const unsigned c_lq = l_lq;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int cur = 0, nxt = NSTAGE - 1;
    for (int t = 0; t < nt; ++t) {
      const int rem = nt - 1 - t;
      // the previous tile's output stores were issued AFTER this tile's first stages: a counted wait for stage 0 has to
      // allow for them (FWD: exactly FWD_STORES per lane, + column-sum stores on some waves - a lower bound is safe), or
      // the tile would wait for its predecessor's stores to complete
      if (t == 0 && !first_tile) {
        if constexpr (FWD) {
          if (rem >= NSTAGE - 2) wait_vmcnt<(NSTAGE - 2) * L + FWD_STORES>(); else wait_stages<L>(0);
        } else {          // pipe_epilogue: >= NH * NIT stores per lane (+ the operand prefetched for this tile when one half covers it)
          if (rem < NSTAGE - 2) wait_stages<L>(0);
          else if (use_aux) wait_vmcnt<(NSTAGE - 2) * L + G::NH * G::NIT + G::NIT>();
          else wait_vmcnt<(NSTAGE - 2) * L + G::NH * G::NIT>();
        }
      } else {
        wait_stages<L>(rem < NSTAGE - 2 ? rem : NSTAGE - 2);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (rem >= NSTAGE - 1) issue(nxt);
      compute(cur);
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
      nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // every wave is done with the ring
    // ---- operands of the next tile: rows, prologue DMA, (operand of the fused derivative: after the current one is consumed)
    u32x4 auxc[PFN ? G::NIT : 1];
    if constexpr (PFN) {
#pragma unroll
      for (int i = 0; i < G::NIT; ++i) auxc[i] = auxn[i];
    }
    ++it;
    const bool more = next_tile(it);
    if (more) {
      setup_rows();
      if constexpr (PFN) {
        if (use_aux) pipe_aux_load<G>(a, l_pz, l_m0, l_n0, 0, auxn);
      }
      prologue();
    }
    // ---- epilogue of the current tile
    if constexpr (!FWD) {
      if (use_aux) pipe_epilogue<WGM, WGN, WM, WN, T_BYTES, STATS, true>(a, acc, T, c_pz, c_lq, c_m0, c_n0, auxc);
      else pipe_epilogue<WGM, WGN, WM, WN, T_BYTES, STATS, false>(a, acc, T, c_pz, c_lq, c_m0, c_n0, auxc);
      lds_barrier();                             // T is read out before the next tile's epilogue writes it
    } else {
      const int pM = a.ph[c_pz].M, halfM = pM >> 1;
      float rsc0 = 1.f, rsc1 = 1.f;
      if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
      const float neg_act = a.act == IPRGAN_ACT_NONE ? 1.f : a.act == IPRGAN_ACT_RELU ? 0.f : a.slope;
      __bf16* T16 = (__bf16*)T;
      float cs1[WN], cs2[WN], bias_l[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = c_n0 + (wn * WN + j) * 32 + l31;
        cs1[j] = cs2[j] = 0.f;
        bias_l[j] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float rsm = a.rs0 ? (c_m0 + row < halfM ? rsc0 : rsc1) : 1.f;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            float v = acc[i][j][r];
            if (a.rs0) v *= rsm;
            if (STATS) { cs1[j] += v; cs2[j] += v * v; }         // stat_mode 1: the accumulator before the bias (rows past M are zero)
            v += bias_l[j];
            if (a.act != IPRGAN_ACT_NONE) v = v > 0.f ? v : (neg_act == 0.f ? 0.f : v * neg_act);
            T16[row * BN + (wn * WN + j) * 32 + l31] = (__bf16)v;
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // (LDS-only barriers: __syncthreads() would wait for the output stores too)
      {                                          // copy-out: 8 channels (16 bytes) of one pixel per thread and pass
        constexpr int OCT = BN / 8, RPI = NT / OCT, NIT = BM / RPI;
        const int oct = tid % OCT, r0 = tid / OCT, n = c_n0 + oct * 8;
        const bool nok = n < a.Ns;
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
          const int row = q * RPI + r0;
          const unsigned e0 = pipe_row_elem(a, c_pz, c_m0 + row);
          const u32x4 w = *(const u32x4*)(T16 + row * BN + oct * 8);
          __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, (e0 != OOB_OFFSET && nok) ? (e0 + (unsigned)n) * 2u : OOB_OFFSET, 0, PIPE_NT);
        }
      }
      if (STATS) {
        // this lane: column (wn*WN+j)*32 + l31 over its 16 * WM rows; + the other half-wave; the WGM waves of a column through LDS
#pragma unroll
        for (int j = 0; j < WN; ++j) { cs1[j] += __shfl_xor(cs1[j], 32, 64); cs2[j] += __shfl_xor(cs2[j], 32, 64); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // the tile has been copied out
        if (half == 0) {
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int c = (wn * WN + j) * 32 + l31;
            T[(wm * BN + c) * 2] = cs1[j]; T[(wm * BN + c) * 2 + 1] = cs2[j];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int c = tid; c < BN; c += NT) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int w = 0; w < WGM; ++w) { s1 += T[(w * BN + c) * 2]; s2 += T[(w * BN + c) * 2 + 1]; }
          if (c_n0 + c < a.Ns) { a.stat_part[((size_t)c_lq * 2) * a.Ns + c_n0 + c] = s1; a.stat_part[((size_t)c_lq * 2 + 1) * a.Ns + c_n0 + c] = s2; }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // T is free for the next tile
    }
    if (!more) break;
    first_tile = false;
  }
}

// ---- four sub-pixel phases in one block ------------------------------------------------------------------------
// Backward-data of a k4 s2 p1 Conv2d and forward of the matching ConvTranspose2d (geom_bwd_form: four phases of 2x2 taps
// each).  As four independent tile sets (one per phase) every input pixel row crosses L2 -> LDS sixteen times (4 phases x
// 4 taps) for a K of only 4 x C per tile: at 64 channels the launch moved 160 KB of LDS-DMA per 256x64 tile and sat at
// the per-CU DMA rate (D.conv1 backward-data of DCGAN-128: 333 TFLOP/s).  Here a block owns 256 grid positions (whole
// rows of the small grid) x 64 output channels x ALL FOUR phases:
//   * the (R + 2) x (W + 2) halo of input pixels is staged ONCE per 64-channel chunk (LDS-DMA, zero padding = out of
//     range); the nine taps of the 3x3 neighbourhood are row shifts of that image (A fragments: ds_read_b128 at the
//     tap's halo row, XOR swizzle recomputed per tap);
//   * tap (dy, dx) feeds the phases that use it (corner 1, edge 2, centre 4): its A fragments are read once and
//     multiplied with one 64 x 64 weight tile per phase; weight tiles stream through an 8-slot ring two stages ahead;
//   * 4 phases x 256 x 64 accumulators = 128 registers per lane; each phase leaves through pipe_epilogue.
// L2 -> LDS bytes per output drop 3.6x (C = 64).  Same GConvArgs, same partial-row accounting as the phase tiles.
template <int HPW, bool DB, bool STATS>
__global__ __launch_bounds__(512) void gconv_phase4_kernel(const GConvArgs a) {
  constexpr int HALO_BYTES = HPW * 8 * 1024, RING_OFF = (DB ? 2 : 1) * HALO_BYTES, T_BYTES = 64 * 1024;
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  char* ldsc = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                    // this wave: tile rows 64 wr .., columns 32 wc ..
  const int half = lane >> 5, l31 = lane & 31, lrow = lane >> 3, lchunk = lane & 7;
  const int W = a.ph[0].owg, Hg = a.ph[0].ohg, R = 256 / W, HW = W + 2, HR = (R + 2) * HW;     // grid of every phase (even sizes)
  // logical tile order: n tiles fastest, then m tiles (the four phases are inside the block)
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int mt = (int)(lt / gridDim.y), n0 = (int)(lt % gridDim.y) * 64, m0 = mt * 256;
  const int b = m0 / (Hg * W), y0 = (m0 - b * Hg * W) / W;   // tile = rows y0 .. y0 + R - 1 of image b
  const int Cs = a.Cs, NC = Cs / 64;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const unsigned lds_base = (unsigned)(uintptr_t)lds;

  // halo slots of this lane: instruction j = i * 8 + wave covers halo pixels 8 j .. 8 j + 7, this lane pixel 8 j + lrow
  unsigned hoff[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int h = (i * 8 + wave) * 8 + lrow;
    const int hr = h / HW, hx = h - hr * HW, y = y0 - 1 + hr, x = hx - 1;
    const bool ok = h < HR && (unsigned)y < (unsigned)a.IH && (unsigned)x < (unsigned)a.IW;
    hoff[i] = ok ? (unsigned)(((b * a.IH + y) * a.IW + x) * Cs) * 2u + (unsigned)(lchunk ^ ((h >> 1) & 7)) * 16u : OOB_OFFSET;
  }
  auto issue_halo = [&](int c) {
    const unsigned dst = lds_base + (unsigned)((DB ? (c & 1) : 0) * HALO_BYTES) + (unsigned)wave * 1024u;
#pragma unroll
    for (int i = 0; i < HPW; ++i) dma16(rs_in, dst + (unsigned)i * 8192u, hoff[i] == OOB_OFFSET ? OOB_OFFSET : hoff[i] + (unsigned)c * 128u);
  };
  // weight tile of (tap t of the 3x3 neighbourhood, phase index q of that tap's list) for chunk c into ring slot `slot`
  const unsigned wrow = (unsigned)((n0 + wave * 8 + lrow) * a.Kp) * 2u + (unsigned)(lchunk ^ (((wave * 8 + lrow) >> 1) & 7)) * 16u;
  auto issue_w = [&](int c, int dyi, int dxi, int py, int px, int slot) {
    const Phase& ph = a.ph[py * 2 + px];
    const int ty = ph.dy0 - dyi, tx = ph.dx0 - dxi;                     // dys = dxs = -1
    const unsigned wk = (unsigned)((ph.wbase + ty * ph.wsy + tx * ph.wsx) * Cs + c * 64) * 2u;
    dma16(rs_wt, lds_base + RING_OFF + (unsigned)slot * 8192u + (unsigned)wave * 1024u, wrow + wk);
  };
  // stage t (tap) of chunk c: its NP weight tiles, compile-time slot list (prefix sums of 1,2,1,2,4,2,1,2,1 mod 8)
  auto issue_stage = [&](int c, auto TC) {
    constexpr int t = decltype(TC)::value;
    constexpr int dyi = t / 3 - 1, dxi = t % 3 - 1;
    constexpr int pre[9] = {0, 1, 3, 4, 6, 10, 12, 13, 15};
    int q = 0;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const bool uy = (dyi == -1 && py == 0) || dyi == 0 || (dyi == 1 && py == 1);
        const bool ux = (dxi == -1 && px == 0) || dxi == 0 || (dxi == 1 && px == 1);
        if (uy && ux) { issue_w(c, dyi, dxi, py, px, (pre[t] + q) & 7); ++q; }
      }
  };

  f32x16 acc[4][2][1];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][i][0][r] = 0.f;

  // halo row index of this lane's A rows for tap (0, 0): row m = 64 wr + 32 i + l31 of the tile = (m / W, m % W)
  int hidx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wr * 64 + i * 32 + l31, r = m / W, x = m - r * W;
    hidx[i] = (r + 1) * HW + x + 1;
  }
  const unsigned b_lane = (unsigned)(wc * 32 + l31) * 128u;
  const unsigned b_sw = (unsigned)((l31 >> 1) & 7);

  auto compute = [&](int c, auto TC) {
    constexpr int t = decltype(TC)::value;
    constexpr int dyi = t / 3 - 1, dxi = t % 3 - 1;
    constexpr int pre[9] = {0, 1, 3, 4, 6, 10, 12, 13, 15};
    const char* hb = ldsc + (DB ? (c & 1) : 0) * HALO_BYTES;
    unsigned aoff[2], asw[2];
    // (opaque to the optimiser: otherwise the nine taps' addresses are hoisted out of the chunk loop - 36 registers that
    // the 128 accumulators do not leave, reloaded from scratch in every step)
    asm volatile("" : "+v"(hidx[0]), "+v"(hidx[1]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = hidx[i] + dyi * HW + dxi;
      aoff[i] = (unsigned)idx * 128u;
      asw[i] = (unsigned)((idx >> 1) & 7);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {             // k outermost: two A fragments live at a time (128 accumulators leave little room)
      bf16x8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(hb + aoff[i] + (((unsigned)(2 * kk + half)) ^ asw[i]) * 16u);
      int q = 0;
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const bool uy = (dyi == -1 && py == 0) || dyi == 0 || (dyi == 1 && py == 1);
          const bool ux = (dxi == -1 && px == 0) || dxi == 0 || (dxi == 1 && px == 1);
          if (uy && ux) {
            const bf16x8 bf = *(const bf16x8*)(ldsc + RING_OFF + ((pre[t] + q) & 7) * 8192 + b_lane + (((unsigned)(2 * kk + half)) ^ b_sw) * 16u);
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[py * 2 + px][i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf, acc[py * 2 + px][i][0], 0, 0, 0);
            ++q;
          }
        }
    }
  };

  using G = EpiGeom<4, 2, 2, 1, T_BYTES>;
  const bool use_aux = a.aux && a.aux16;

  issue_halo(0);
  issue_stage(0, std::integral_constant<int, 0>{});
  issue_stage(0, std::integral_constant<int, 1>{});
  for (int c = 0; c < NC; ++c) {
    const bool more_c = c + 1 < NC;
    // one tap per step: wait for its weight tiles (and, in-order, everything older), barrier, issue two stages ahead
    // (+ the next chunk's halo right behind stage 2), multiply
    auto step = [&](auto TC) {
      constexpr int t = decltype(TC)::value;
      constexpr int NPv[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
      constexpr int nxt1 = t < 8 ? NPv[t + 1] : NPv[0];                 // instructions of the stage after this one
      const bool has1 = t < 8 || more_c;
      if (t == 1 || t == 2) {                    // the next chunk's halo (issued at t = 0, after stage 2) is younger than this stage
        if (more_c && DB) wait_vmcnt<nxt1 + HPW>(); else wait_vmcnt<nxt1>();
      } else if (has1) {
        wait_vmcnt<nxt1>();
      } else {
        wait_vmcnt<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 <= 8) issue_stage(c, std::integral_constant<int, (t + 2 <= 8 ? t + 2 : 0)>{});
      else if (more_c) issue_stage(c + 1, std::integral_constant<int, (t + 2 > 8 ? t + 2 - 9 : 0)>{});
      if (t == 0 && more_c && DB) issue_halo(c + 1);
      compute(c, TC);
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- four epilogues (phase p: rows m0 .. of its grid, partial row lq = 4 mt + p), next phase's operand loaded a phase ahead
  float* T = (float*)lds;
  auto epi = [&](int p, const u32x4* aux) {
    if (use_aux) pipe_epilogue<4, 2, 2, 1, T_BYTES, STATS, true>(a, acc[p], T, p, (unsigned)(mt * 4 + p), m0, n0, aux);
    else pipe_epilogue<4, 2, 2, 1, T_BYTES, STATS, false>(a, acc[p], T, p, (unsigned)(mt * 4 + p), m0, n0, aux);
    lds_barrier();
  };
  // (the fused-derivative operand of a phase is loaded one phase ahead; phase 0's here: 128 accumulators + the K loop's
  // working set leave no registers to carry it through the loop, and its latency is paid once per 4 x 256 x 64 outputs)
  u32x4 auxA[G::NIT], auxB[G::NIT];
  if (use_aux) { pipe_aux_load<G>(a, 0, m0, n0, 0, auxA); pipe_aux_load<G>(a, 1, m0, n0, 0, auxB); }
  epi(0, auxA);
  if (use_aux) pipe_aux_load<G>(a, 2, m0, n0, 0, auxA);
  epi(1, auxB);
  if (use_aux) pipe_aux_load<G>(a, 3, m0, n0, 0, auxB);
  epi(2, auxA);
  epi(3, auxB);
}

// ---- host side -----------------------------------------------------------------------------------------------
static int g_pipe_f32 = getenv("IPRGAN_PIPE_F32") ? atoi(getenv("IPRGAN_PIPE_F32")) : 1;     // A/B switch: LDS-DMA ring tiles for fp32 layers
// Measured and not kept (round 5): the bf16 ring with DEDICATED LOADER WAVES (four multiplying waves without vector-memory
// instructions + four waves that only issue the refills: what conv_x3.hip's gconv_x3ws_kernel does for three-plane operands,
// +10-27 % there).  256x128 (4 x 128x64, 3 stages), 128x128 (2 stages, two blocks per CU) and 256x64 tiles, parity-green:
// +2-7 % over the plain tiles of the same shape, behind the autotuner's picks on every DCGAN-128 layer, config 5 unchanged
// (17.3 ms; profiles/r05_bf16_loader_wave_tiles.jsonl).  A bf16 step moves the same bytes as a three-plane step for a sixth
// of the matrix work: these tiles are bound by the bytes a CU can pull from L2 (~70 GB/s), not by the issue slots.

bool gconv_pipe_eligible(const GConvArgs& a) {
  if (a.ksplit > 1 || a.wmod > 0 || a.planar_M || a.in16 == 2 || a.out16 == 2) return false;      // (three planes: conv_x3.hip)
  auto simple = [](int act) { return act == IPRGAN_ACT_NONE || act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; };
  if ((a.Ns % 8) != 0 || !simple(a.act) || (a.aux && !simple(a.aux_act))) return false;       // pipe_epilogue
  if (!a.in16) {          // fp32 operands and the fp32 MFMA (F32 instantiations): fp32 tensors throughout
    return g_pipe_f32 && (a.Cs % 32) == 0 && !a.out16 && !a.aux16;
  }
  if ((a.Cs % 64) != 0 || a.pad_mode != IPRGAN_PAD_ZERO) return false;
  for (int i = 0; i < a.nphase; ++i)
    if (a.ph[i].M > 0 && ((a.ph[i].steps & 1) || a.ph[i].steps < 2)) return false;
  return true;
}

template <void (*KERN)(const GConvArgs)>
static void pipe_go(const GConvArgs& a, dim3 grid, dim3 block, size_t smem, int slot, hipStream_t st) {
  static bool attr_set = false;            // per kernel instantiation
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
  prof_launch(KERN, grid, block, smem, st, slot, a.flops, a);
}

template <int WGM, int WGN, int WM, int WN, int NSTAGE>
static int launch_pipe_t(const GConvArgs& a, hipStream_t st, int* bm_out) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = (size_t)NSTAGE * (BM + BN) * 128;
  dim3 grid(cdiv(maxM, BM), cdiv(a.Ns, BN), a.nphase);
  *bm_out = BM;
  // (the 256x256 tile has no registers left for a prefetch ahead of the K loop: its epilogue loads per column half)
  constexpr bool can_pf = EpiGeom<WGM, WGN, WM, WN, NSTAGE * (BM + BN) * 128>::PF_FIRST;
  const bool pref = a.aux && a.aux16 && can_pf;
  const dim3 block(WGM * WGN * 64);
  if (a.bn_mean) {          // norm-backward mode of a backward-data pass (iprgan_conv_bwd_data_bn): its own instantiations
    constexpr bool bn_tile = !(BM == 256 && BN == 256);       // (the 256x256 tile keeps accumulators live across its column halves)
    if constexpr (bn_tile) {
      if (!a.stat_part) return -1;
      if (!a.in16) pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, true, false, true, true>>(a, grid, block, smem, 23, st);
      else if (pref && can_pf) pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, true, can_pf, false, true>>(a, grid, block, smem, BN >= 128 ? 19 : 20, st);
      else pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, true, false, false, true>>(a, grid, block, smem, BN >= 128 ? 19 : 20, st);
      IPR_LAUNCH_CHECK();
      return 0;
    }
    return -1;
  }
  if (!a.in16) {            // fp32 operands, exact fp32 MFMA
    if (a.stat_part) pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, true, false, true>>(a, grid, block, smem, 23, st);
    else pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, false, false, true>>(a, grid, block, smem, 23, st);
    IPR_LAUNCH_CHECK();
    return 0;
  }
  const int slot = BN >= 128 ? 19 : 20;
  if (a.stat_part) {
    if constexpr (can_pf) { if (pref) { pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, true, true>>(a, grid, block, smem, slot, st); IPR_LAUNCH_CHECK(); return 0; } }
    pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, true, false>>(a, grid, block, smem, slot, st);
  } else {
    if constexpr (can_pf) { if (pref) { pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, false, true>>(a, grid, block, smem, slot, st); IPR_LAUNCH_CHECK(); return 0; } }
    pipe_go<gconv_pipe_kernel<WGM, WGN, WM, WN, NSTAGE, false, false>>(a, grid, block, smem, slot, st);
  }
  IPR_LAUNCH_CHECK();
  return 0;
}

static int launch_pipe8(const GConvArgs& a, hipStream_t st, int* bm_out) {
  if (!a.in16 || a.Ns < 256 || a.bn_mean) return -1;
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = 128 * 1024;
  dim3 grid(cdiv(maxM, 256), cdiv(a.Ns, 256), a.nphase);
  *bm_out = 256;
  const dim3 block(512);
  if (a.stat_part) pipe_go<gconv_pipe8_kernel<true, false>>(a, grid, block, smem, 26, st);
  else pipe_go<gconv_pipe8_kernel<false, false>>(a, grid, block, smem, 26, st);
  IPR_LAUNCH_CHECK();
  return 0;
}

template <int WGM, int WGN, int WM, int WN, int NSTAGE>
static int launch_pipe2_t(const GConvArgs& a, hipStream_t st, int* bm_out) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  if (a.bn_mean) return -1;          // norm-backward mode: one-tile kernels only (launch_pipe_t)
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = (size_t)NSTAGE * (BM + BN) * 128 + 64 * 1024;
  const int ntn = cdiv(a.Ns, BN), total = cdiv(maxM, BM) * a.nphase * ntn;
  int ncu = 256;
  static int s_ncu = 0;
  if (!s_ncu) { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); if (hipGetDeviceProperties(&pr, dev) == hipSuccess) s_ncu = pr.multiProcessorCount; else s_ncu = 256; }
  ncu = s_ncu;
  dim3 grid(total < ncu ? total : ncu), block(WGM * WGN * 64);
  *bm_out = BM;
  const bool fwd = !a.aux && !a.res && a.out16;
  const int slot = !a.in16 ? 23 : BN >= 128 ? 19 : 20;
  auto go = [&](auto kern) { prof_launch(kern, grid, block, smem, st, slot, a.flops, a, ntn, a.nphase, total); };
#define PIPE2_ATTR(K) { static bool s = false; if (!s) { (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); s = true; } }
  if (!a.in16) {            // fp32 operands, exact fp32 MFMA
    if (a.stat_part) { auto k = gconv_pipe2_kernel<WGM, WGN, WM, WN, NSTAGE, true, false, true>; PIPE2_ATTR(k) go(k); }
    else { auto k = gconv_pipe2_kernel<WGM, WGN, WM, WN, NSTAGE, false, false, true>; PIPE2_ATTR(k) go(k); }
  } else
  if (a.stat_part) {
    if (fwd) { auto k = gconv_pipe2_kernel<WGM, WGN, WM, WN, NSTAGE, true, true>; PIPE2_ATTR(k) go(k); }
    else { auto k = gconv_pipe2_kernel<WGM, WGN, WM, WN, NSTAGE, true, false>; PIPE2_ATTR(k) go(k); }
  } else {
    if (fwd) { auto k = gconv_pipe2_kernel<WGM, WGN, WM, WN, NSTAGE, false, true>; PIPE2_ATTR(k) go(k); }
    else { auto k = gconv_pipe2_kernel<WGM, WGN, WM, WN, NSTAGE, false, false>; PIPE2_ATTR(k) go(k); }
  }
#undef PIPE2_ATTR
  IPR_LAUNCH_CHECK();
  return 0;
}

// the canonical k4 s2 p1 backward-data form: four phases of 2x2 taps on equal even grids whose rows tile 256 positions
static bool phase4_eligible(const GConvArgs& a) {
  if (!a.in16 || a.nphase != 4 || a.osy != 2 || a.osx != 2 || a.isy != 1 || a.isx != 1 || (a.Cs % 64) != 0) return false;
  if (a.pad_mode != IPRGAN_PAD_ZERO || a.ksplit > 1 || a.wmod > 0 || a.planar_M || (a.Ns % 64) != 0) return false;
  auto simple = [](int act) { return act == IPRGAN_ACT_NONE || act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; };
  if (!simple(a.act) || (a.aux && !simple(a.aux_act))) return false;
  const int W = a.ph[0].owg, H = a.ph[0].ohg;
  for (int p = 0; p < 4; ++p) {
    const Phase& ph = a.ph[p];
    if (ph.th != 2 || ph.tw != 2 || ph.dys != -1 || ph.dxs != -1 || ph.owg != W || ph.ohg != H) return false;
    if (ph.dy0 != p / 2 || ph.dx0 != p % 2 || ph.ooy != p / 2 || ph.oox != p % 2) return false;
  }
  if (W != a.IW || H != a.IH || W > 64 || W < 16 || (256 % W) != 0 || (H % (256 / W)) != 0) return false;
  if (W == 64 && a.Cs > 64) return false;          // two 56 KB halo buffers + the weight ring do not fit
  return true;
}

static int launch_phase4(const GConvArgs& a, hipStream_t st, int* bm_out) {
  if (!phase4_eligible(a) || a.bn_mean) return -1;
  const int W = a.ph[0].owg, mtiles = a.ph[0].M / 256;
  dim3 grid(mtiles, a.Ns / 64), block(512);
  *bm_out = 256;
  auto go = [&](auto kern, size_t smem) {
    prof_launch(kern, grid, block, smem, st, 24, a.flops, a);
  };
#define PH4_ATTR(K, S) { static bool s = false; if (!s) { (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(S)); s = true; } }
  if (W == 64) {
    const size_t smem = 7 * 8192 + 65536;          // one halo buffer (single channel chunk) + 8 weight slots; the epilogue's 64 KB alias them
    if (a.stat_part) { auto k = gconv_phase4_kernel<7, false, true>; PH4_ATTR(k, smem) go(k, smem); }
    else { auto k = gconv_phase4_kernel<7, false, false>; PH4_ATTR(k, smem) go(k, smem); }
  } else {
    const size_t smem = 2 * 6 * 8192 + 65536;
    if (a.stat_part) { auto k = gconv_phase4_kernel<6, true, true>; PH4_ATTR(k, smem) go(k, smem); }
    else { auto k = gconv_phase4_kernel<6, true, false>; PH4_ATTR(k, smem) go(k, smem); }
  }
#undef PH4_ATTR
  IPR_LAUNCH_CHECK();
  return 0;
}

// variant: 0 = 256x128 (8 waves of 64x64, 3 stages), 1 = 256x64 (8 waves of 64x32, 3 stages),
//          2 = 256x256 (8 waves of 128x64, 2 stages), 3 = 128x128 (4 waves of 64x64, 2 stages: two blocks per CU),
//          4 = 256x64 with 2 stages (80 KB: two blocks per CU), 5 = 128x64 (4 waves of 64x32, 3 stages, two blocks per CU),
//          6 / 7 = persistent 256x128 / 256x64 (gconv_pipe2_kernel: one block per CU walks the tile list),
//          8 = gconv_phase4_kernel (the four sub-pixel phases of a k4 s2 p1 backward-data form in one block)
// returns -1 when the variant does not apply to the geometry
// A/B switch (GConvArgs::korder): bit 0 = taps inside a channel chunk, bit 1 = tile rows start at different chunks (measured
// neutral: the L2 channels are not the limit), bit 4 = refill pieces woven into the MFMA stream (+5-10 % on the 8-wave bf16
// tiles), bit 5 = the same for the fp32 tiles (+1-3 %)
// bit 6 = stride-2 gathers walk their taps grouped by parity with the channel chunks innermost (round 5)
static int g_pipe_korder = getenv("IPRGAN_PIPE_KORDER") ? atoi(getenv("IPRGAN_PIPE_KORDER")) : 49 + 64;

int launch_gconv_pipe(const GConvArgs& a0, int variant, hipStream_t st, int* bm_out) {
  if (!gconv_pipe_eligible(a0)) return -1;
  GConvArgs a = a0;
  a.korder = g_pipe_korder;
  switch (variant) {
    case 0: return a.Ns >= 128 ? launch_pipe_t<4, 2, 2, 2, 3>(a, st, bm_out) : -1;
    case 1: return launch_pipe_t<4, 2, 2, 1, 3>(a, st, bm_out);
    case 2: return a.Ns >= 256 ? launch_pipe_t<2, 4, 4, 2, 2>(a, st, bm_out) : -1;
    case 3: return a.Ns >= 128 ? launch_pipe_t<2, 2, 2, 2, 2>(a, st, bm_out) : -1;
    case 4: return launch_pipe_t<4, 2, 2, 1, 2>(a, st, bm_out);
    case 5: return launch_pipe_t<2, 2, 2, 1, 3>(a, st, bm_out);
    case 6: return a.Ns >= 128 ? launch_pipe2_t<4, 2, 2, 2, 2>(a, st, bm_out) : -1;             // persistent 256x128, 2 stages + 64 KB
    case 7: return launch_pipe2_t<4, 2, 2, 1, 2>(a, st, bm_out);                                // persistent 256x64, 2 stages + 64 KB
    case 8: return launch_phase4(a, st, bm_out);                                                // four phases per block (k4 s2 p1)
    case 9: return launch_pipe8(a, st, bm_out);                                                 // 256x256, half-tile ring
    default: return -1;
  }
}

}  // namespace iprgan
