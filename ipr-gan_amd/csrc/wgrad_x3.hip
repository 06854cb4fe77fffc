// Backward-weight for three-plane tensors (storage kind 2: x = h + m + l exactly, include/iprgan.h) in the "halo" form of
// wgrad_halo.hip, on the bf16 matrix pipe with fp32-grade products (six MFMAs per block, math mode fp32x3).
//
//   dW[n][tap][c] = sum_m S[m][n] * L[m * stride + tap - pad][c]
// S = the tensor on the small grid (Conv2d: dy; ConvTranspose2d: x), L = the tensor gathered around it (Conv2d: x;
// ConvTranspose2d: dy).  The split-M GEMM (wgrad_kernel<.., IN3P>, conv_igemm.hip) streams an im2col view of L through
// registers: every L element crosses L2 once per tap (9x / 16x), the loader and the MFMAs take turns between two barriers
// per 32 rows, and it runs at 90-175 TFLOP/s fp32-equivalent.  Here, per step, a block stages ONE patch of TY x 8 pixels of
// S (64 channels) and the halo of L around it (CB x 32 channels) - all three planes of both, by LDS-DMA, once - and every
// tap is a row shift of the halo image in LDS.  Per 16 reduction pixels a wave multiplies 6 plane pairs per (tap, block):
// the L2 -> LDS traffic per MFMA is a sixth of the bf16 halo kernel's, which already ran at 0.85-1.0 PFLOP/s, so this form
// is bound by the matrix pipe (416.7 TFLOP/s fp32-equivalent), not by the staging.
//   * wave = (tap group, 32-channel half of S, 32-channel block of L): TPW taps x one 32x32 block, TWO accumulators per
//     (tap, block) - h h' into one, the five small terms into the other (conv_x3.hip: why) - merged at the slab store;
//   * LDS rows are 64 bytes (32 channels of one pixel of one plane); halo pixels of one stride residue are consecutive rows,
//     so the four pixel rows of a `ds_read_b64_tr_b16` block are 256 contiguous bytes for every tap shift (no swizzle);
//   * stages form a ring: counted `s_waitcnt vmcnt`, one raw `s_barrier` per patch; the DMA is issued from inline asm
//     (wgrad_halo.hip: hipcc would order every transposed read behind all outstanding LDS-DMA otherwise);
//   * zero padding / ragged edges are out-of-range DMA offsets (zeros); ReflectionPad2d is folded into the DMA offsets.
// Output: partial slabs ws[split][n][tap * Ls + c] in the layout of wgrad_kernel, summed in fixed order and scattered to
// PyTorch layout by wgrad_reduce_kernel (deterministic).
// Reference layers: networks/sn_discriminator.py:9-18, conv_generator.py:8, sr_resnet.py:22, resnet_generator.py:40-49.
#include "conv_shared.h"
#include <stdlib.h>
#include <type_traits>

namespace iprgan {

struct WX3Args {
  const void* S;        // [3][B][PH][PW][Ss] bf16 (plane stride s_ps bytes)
  const void* L;        // [3][B][QH][QW][Ls] bf16 (plane stride l_ps bytes)
  float* ws;            // slabs [nsplit][Nrows][Kw]
  int B, PH, PW, QH, QW, Ss, Ls, pad, reflect;
  int PTY, PTX;         // patches per image
  FastDiv d_ptx, d_ppi; // / PTX, / (PTY * PTX)
  int npatch, pps;      // patches in all, patches per split
  int Nrows, Kw;
  unsigned s_bytes, l_bytes, s_ps, l_ps;
  double flops;
};

typedef short wx_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wx_dma16(wx_u32x4 rs, unsigned lds_addr, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lds_addr), "s"(rs) : "memory");
#endif
}
__device__ __forceinline__ wx_u32x4 wx_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  wx_u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);       // stride 0
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
template <int N>
__device__ __forceinline__ void wx_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 wx_tr_read8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) wx_s16x4* lds_ptr;
  const wx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  const wx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
  union { wx_s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = lo; u.h[1] = hi;
  return u.v;
}

template <int KH, int KW, int STR, int TY, int TPW, int CB, int NSTAGE>
struct WX3Geom {
  static constexpr int TX = 8, NTAP = KH * KW, NTG = NTAP / TPW, NWAVE = NTG * 2 * CB, NPX = TY * TX;
  static constexpr int HH = (TY - 1) * STR + KH, HW = (TX - 1) * STR + KW;
  static constexpr int PHS = (HH + STR - 1) / STR, PWS = (HW + STR - 1) / STR, PP = PHS * PWS;   // rows of one stride-residue plane
  static constexpr int HROWS = STR * STR * PP;                 // halo pixel rows per 32-channel block and plane
  static constexpr int HPP = (HROWS + 15) / 16 * 16;
  static constexpr int Q_ROWS = 3 * CB * HPP, P_ROWS = 3 * 2 * NPX;       // 64-byte LDS rows: [plane][cb][halo pixel], [plane][nb][patch pixel]
  static constexpr int NINST = (Q_ROWS + P_ROWS) / 16;         // LDS-DMA wave-instructions per stage (16 rows each)
  static constexpr int LPW = (NINST + NWAVE - 1) / NWAVE;      // per wave (the last ones may be dummies into the pad)
  static constexpr int STAGE_BYTES = LPW * NWAVE * 1024;
  static constexpr int Q_BYTES = Q_ROWS * 64;
  static constexpr int SMEM = NSTAGE * STAGE_BYTES;
  static_assert(NTAP % TPW == 0 && (TY % 2) == 0 && (NPX % 16) == 0, "taps per wave / 16-pixel reduction sub-steps");
  static_assert(SMEM <= 160 * 1024 && NSTAGE >= 2 && NSTAGE <= 4 && LPW * (NSTAGE - 2) <= 63, "ring geometry");
};

constexpr int wx_threads(int ntap, int tpw, int cb) { return ntap / tpw * 2 * cb * 64; }

template <int KH, int KW, int STR, int TY, int TPW, int CB, int NSTAGE>
__global__ __launch_bounds__(wx_threads(KH * KW, TPW, CB)) void wgrad_x3h_kernel(const WX3Args a) {
  using G = WX3Geom<KH, KW, STR, TY, TPW, CB, NSTAGE>;
  constexpr int NWAVE = G::NWAVE, LPW = G::LPW;
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

  // logical block order: (c chunk, n tile) fastest, then the split: the blocks that read the same patches share an XCD's L2
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const int cx = (int)(lt % gridDim.x), ny = (int)((lt / gridDim.x) % gridDim.y), split = (int)(lt / (gridDim.x * gridDim.y));
  const int c0 = cx * (CB * 32), n0 = ny * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w_cb = wave % CB, w_nb = (wave / CB) & 1, w_tg = wave / (2 * CB);       // this wave's L block, S half, tap group

  int pbeg = split * a.pps, pend = pbeg + a.pps;
  if (pend > a.npatch) pend = a.npatch;
  const int np = pend - pbeg;

  const wx_u32x4 rs_s = wx_make_rsrc(a.S, a.s_bytes), rs_l = wx_make_rsrc(a.L, a.l_bytes);

  // what this lane moves in its LPW DMA slots of a stage: LDS row (16 * inst + lane / 4), 16-byte piece lane % 4.
  // rel = byte offset of (plane, channel block, piece); ryx = (y << 16) | x pixel offset from the patch origin (of L for halo
  // rows, of S for patch rows); pad rows and dummy instructions carry y = 0x7fff: always out of range
  unsigned rel[LPW];
  int ryx[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int inst = i * NWAVE + wave, row = inst * 16 + (lane >> 2), piece = lane & 3;
    rel[i] = 0; ryx[i] = 0x7fff << 16;
    if (row < G::Q_ROWS) {
      const int p = row / (CB * G::HPP), r2 = row % (CB * G::HPP), cb = r2 / G::HPP, hp = r2 % G::HPP;
      if (hp < G::HROWS) {
        const int plane = hp / G::PP, r = hp % G::PP, hyy = r / G::PWS, hxx = r % G::PWS;
        const int hy = hyy * STR + plane / STR, hx = hxx * STR + plane % STR;
        rel[i] = (unsigned)p * a.l_ps + (unsigned)((c0 + cb * 32) * 2 + piece * 16);
        ryx[i] = (hy << 16) | hx;
      }
    } else if (row < G::Q_ROWS + G::P_ROWS) {
      const int pr = row - G::Q_ROWS, p = pr / (2 * G::NPX), r2 = pr % (2 * G::NPX), nb = r2 / G::NPX, m = r2 % G::NPX;
      rel[i] = (unsigned)p * a.s_ps + (unsigned)((n0 + nb * 32) * 2 + piece * 16);
      ryx[i] = ((m >> 3) << 16) | (m & 7);
    }
  }

  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  // the patch a stage refill is for (wave-uniform), set by patch_begin(), consumed slot by slot by issue_slot()
  int i_b = 0, i_py0 = 0, i_px0 = 0, i_qy0 = 0, i_qx0 = 0;
  unsigned i_sb = 0;
  bool i_live = true;           // false past the split's last patch: the slots still leave (the counted waits rely on LPW
                                // DMA instructions per step) with out-of-range offsets, into a stage nobody reads again
  auto patch_begin = [&](int patch, int buf) __attribute__((always_inline)) {
    const int b = fdiv(patch, a.d_ppi);
    const int r = patch - b * (a.PTY * a.PTX);
    const int pty = fdiv(r, a.d_ptx), ptx = r - pty * a.PTX;
    i_live = patch < pend;
    i_b = b; i_py0 = pty * TY; i_px0 = ptx * 8;
    i_qy0 = i_py0 * STR - a.pad; i_qx0 = i_px0 * STR - a.pad;
    i_sb = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)buf * G::STAGE_BYTES + (unsigned)wave * 1024u);
  };
  auto issue_slot = [&](int i) __attribute__((always_inline)) {
    const int b = i_b, py0 = i_py0, px0 = i_px0, qy0 = i_qy0, qx0 = i_qx0;
    const unsigned sb = i_sb;
    {
      // every wave issues exactly LPW DMA instructions per stage (the counted vmcnt relies on it); which tensor a slot
      // reads is wave-uniform (Q_ROWS % 16 == 0): scalar selects, no lane-dependent control flow
      const bool isS = (i * NWAVE + wave) * 16 >= G::Q_ROWS;
      int y = (isS ? py0 : qy0) + (ryx[i] >> 16), x = (isS ? px0 : qx0) + (ryx[i] & 0xffff);
      bool ok;
      if (!isS && a.reflect) {             // mirrored pixel instead of a zero (dummy / pad rows stay out of range)
        ok = (ryx[i] >> 16) != 0x7fff && y < a.QH + a.pad && x < a.QW + a.pad;      // (rows below / right of a ragged last patch)
        y = reflect_idx(y, a.QH); x = reflect_idx(x, a.QW);
        ok = ok && (unsigned)y < (unsigned)a.QH && (unsigned)x < (unsigned)a.QW;
      } else {
        ok = ((unsigned)y < (unsigned)(isS ? a.PH : a.QH)) & ((unsigned)x < (unsigned)(isS ? a.PW : a.QW));
      }
      const unsigned pix = isS ? (unsigned)(((b * a.PH + y) * a.PW + x) * a.Ss) * 2u : (unsigned)(((b * a.QH + y) * a.QW + x) * a.Ls) * 2u;
      const unsigned off = (ok && i_live) ? pix + rel[i] : OOB_OFFSET;
      if (isS) wx_dma16(rs_s, sb + (unsigned)(i * NWAVE) * 1024u, off);
      else wx_dma16(rs_l, sb + (unsigned)(i * NWAVE) * 1024u, off);
    }
  };
  auto issue = [&](int patch, int buf) {
    patch_begin(patch, buf);
#pragma unroll
    for (int i = 0; i < LPW; ++i) issue_slot(i);
  };

  f32x16 acc[TPW], accs[TPW];          // h h' | the five small terms
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = accs[t][r] = 0.f;

  // fragment addressing (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of its 16
  // columns; lanes 16-31 take columns 16-31 of the 32-channel row; lanes 32-63 the second 8 of the 16 reduction rows)
  const int half = lane >> 5, gq = (lane & 15) >> 2, gp = lane & 3, gcol = (lane >> 4) & 1;
  const unsigned lane_col = (unsigned)(gcol * 32 + gp * 8);
  // S fragment of plane p, sub-step ks: + p * (2 * NPX * 64) + ks * 1024 (+ 256 for the second four rows)
  const unsigned a_lane = G::Q_BYTES + (unsigned)(w_nb * G::NPX * 64) + (unsigned)((8 * half + gq) * 64) + lane_col;
  // L fragment of plane p, tap tt, sub-step ks: + p * (CB * HPP * 64) + (2 * ks * PWS) * 64 (+ 4 * 64)
  unsigned b_tap[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = w_tg * TPW + tt;
    const int ty = tap / KW, tx = tap % KW;
    const int trow = ((ty % STR) * STR + tx % STR) * G::PP + (ty / STR) * G::PWS + tx / STR;
    b_tap[tt] = (unsigned)(w_cb * G::HPP * 64) + (unsigned)((trow + half * G::PWS + gq) * 64) + lane_col;
  }
  const char* ldsc = (const char*)lds;

  // REFILL: the LPW DMA slots of the next stage are woven into the MFMA stream (one slot after every NG / LPW-th group of
  // six MFMAs): issued as one burst behind the barrier, the 70 KB of a stage hold every wave of the block in its issue
  // slot until the CU's address path has drained (conv_pipe.hip measured the burst and the MFMA phase ADDING).  ONE
  // instantiation: with a refilling and a draining copy of this fully unrolled body side by side hipcc spilled 480 bytes
  // per lane (and ran at a quarter of the speed); the draining steps issue out-of-range slots instead (i_live).
  auto compute = [&](int buf, auto REFILL) __attribute__((always_inline)) {
    constexpr bool refill = decltype(REFILL)::value;
    constexpr int NG = (G::NPX / 16) * TPW;
    const char* sb = ldsc + buf * G::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < G::NPX / 16; ++ks) {
      bf16x8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        af[p] = wx_tr_read8(sb + a_lane + p * (2 * G::NPX * 64) + ks * 1024, sb + a_lane + p * (2 * G::NPX * 64) + ks * 1024 + 256);
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        bf16x8 bfr[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bfr[p] = wx_tr_read8(sb + b_tap[tt] + p * (CB * G::HPP * 64) + (2 * ks * G::PWS) * 64,
                               sb + b_tap[tt] + p * (CB * G::HPP * 64) + (2 * ks * G::PWS + 4) * 64);
        // (S term, L term): l h', h l', m m', m h', h m' into the small accumulator, h h' into the large one
        accs[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[0], accs[tt], 0, 0, 0);
        accs[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[2], accs[tt], 0, 0, 0);
        accs[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1], accs[tt], 0, 0, 0);
        accs[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], accs[tt], 0, 0, 0);
        accs[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], accs[tt], 0, 0, 0);
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], acc[tt], 0, 0, 0);
        if constexpr (refill) {          // slot q leaves behind group ceil((q + 1) NG / LPW) - 1 (all indices are constants here)
#pragma unroll
          for (int q = 0; q < LPW; ++q)
            if (((q + 1) * NG + LPW - 1) / LPW - 1 == ks * TPW + tt) {
              issue_slot(q);
            }
        }
      }
    }
  };

  // ---- the ring over this split's patches
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) issue(pbeg + s, s);            // (past the last patch: out-of-range slots, see i_live)
  int cur = 0, nxt = NSTAGE - 1;
  for (int t = 0; t < np; ++t) {
    wx_wait_vmcnt<(NSTAGE - 2) * LPW>();        // every step issues LPW slots: NSTAGE - 2 younger stages stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    patch_begin(pbeg + t + NSTAGE - 1, nxt);
    compute(cur, std::true_type{});
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }

  wx_wait_vmcnt<0>();             // (the draining steps' dummy slots: no LDS-DMA may be outstanding when the block ends)
  // ---- slab store: ws[split][n0 + nb*32 + row][tap * Ls + c0 + cb*32 + col], C layout row = (r&3) + 8*(r>>2) + 4*half
  const int l31 = lane & 31;
  float* slab = a.ws + (size_t)split * a.Nrows * a.Kw;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = w_tg * TPW + tt;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + w_nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      slab[(size_t)n * a.Kw + tap * a.Ls + c0 + w_cb * 32 + l31] = acc[tt][r] + accs[tt][r];
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------
bool wgrad_x3h_eligible(const iprgan_conv_desc* d) {
  if (d->x_bf16 != 2 || d->y_bf16 != 2) return false;
  if ((d->Cin % 64) != 0 || (d->Cout % 64) != 0 || d->KH != d->KW) return false;
  if (d->pad_mode == IPRGAN_PAD_REFLECT && (d->transposed || d->pad >= d->H || d->pad >= d->W)) return false;
  static const int k3s2 = getenv("IPRGAN_WX3_K3S2") ? atoi(getenv("IPRGAN_WX3_K3S2")) : 1;     // A/B switch (round 6)
  return (d->KH == 3 && d->stride == 1) || (d->KH == 4 && d->stride == 2) || (k3s2 && d->KH == 3 && d->stride == 2);
}
static int x3h_ty(const iprgan_conv_desc* d) { return d->KH == 3 && d->stride == 1 ? 8 : 4; }
static int x3h_cb(const iprgan_conv_desc* d) { return d->KH == 3 && d->stride == 1 ? 2 : 1; }

// nsplit for a target number of blocks (the caller sizes the slabs with the same function)
int wgrad_x3h_nsplit(const iprgan_conv_desc* d, int target_blocks) {
  const int PH = d->transposed ? d->H : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int PW = d->transposed ? d->W : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  const int npatch = d->B * cdiv(PH, x3h_ty(d)) * cdiv(PW, 8);
  const int Ls = d->transposed ? d->Cout : d->Cin, Ss = d->transposed ? d->Cin : d->Cout;
  const int tiles = (Ls / (32 * x3h_cb(d))) * (Ss / 64);
  int want = cdiv(target_blocks, tiles);
  if (want < 1) want = 1;
  if (want > npatch) want = npatch;
  const int pps = cdiv(npatch, want);
  return cdiv(npatch, pps);
}

template <int KH, int KW, int STR, int TY, int TPW, int CB, int NSTAGE>
static int launch_wx3(const WX3Args& a, dim3 grid, hipStream_t st) {
  using G = WX3Geom<KH, KW, STR, TY, TPW, CB, NSTAGE>;
  auto kern = wgrad_x3h_kernel<KH, KW, STR, TY, TPW, CB, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM); attr_set = true; }
  prof_launch(kern, grid, dim3(G::NWAVE * 64), (size_t)G::SMEM, st, 30, a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}

// S / L: see the top of the file.  Slabs: ws[nsplit][Nrows = S channels][Kw = taps * L channels].
int launch_wgrad_x3h(const iprgan_conv_desc* d, const void* x, const void* dy, float* ws, int target_blocks, hipStream_t st,
                     int* nsplit_out, int* Nrows_out, int* Kw_out) {
  if (!wgrad_x3h_eligible(d)) return -1;
  const int OH = d->transposed ? (d->H - 1) * d->stride - 2 * d->pad + d->KH + d->outpad : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int OW = d->transposed ? (d->W - 1) * d->stride - 2 * d->pad + d->KW + d->outpad : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  WX3Args a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.pad = d->pad; a.reflect = d->pad_mode == IPRGAN_PAD_REFLECT;
  long long sps, lps;          // plane strides in elements (0 = contiguous)
  if (d->transposed) { a.S = x; a.L = dy; a.PH = d->H; a.PW = d->W; a.QH = OH; a.QW = OW; a.Ss = d->Cin; a.Ls = d->Cout; sps = d->x_pstride; lps = d->y_pstride; }
  else { a.S = dy; a.L = x; a.PH = OH; a.PW = OW; a.QH = d->H; a.QW = d->W; a.Ss = d->Cout; a.Ls = d->Cin; sps = d->y_pstride; lps = d->x_pstride; }
  a.ws = ws;
  const int TY = x3h_ty(d);
  a.PTY = cdiv(a.PH, TY); a.PTX = cdiv(a.PW, 8);
  a.d_ptx = make_fastdiv(a.PTX); a.d_ppi = make_fastdiv(a.PTY * a.PTX);
  a.npatch = a.B * a.PTY * a.PTX;
  const int nsplit = wgrad_x3h_nsplit(d, target_blocks);
  a.pps = cdiv(a.npatch, nsplit);
  a.Nrows = a.Ss; a.Kw = d->KH * d->KW * a.Ls;
  const unsigned long long sb = (unsigned long long)a.B * a.PH * a.PW * a.Ss * 2, lb = (unsigned long long)a.B * a.QH * a.QW * a.Ls * 2;
  const unsigned long long sp = sps ? (unsigned long long)sps * 2 : sb, lp = lps ? (unsigned long long)lps * 2 : lb;
  IPR_CHECK(sb + 2 * sp < 0x7fffffffull && lb + 2 * lp < 0x7fffffffull, "conv_bwd_weight: three-plane tensor larger than 2 GiB");
  IPR_CHECK(a.QH < 32000 && a.QW < 32000, "conv_bwd_weight: image too large for the halo form");
  a.s_ps = (unsigned)sp; a.l_ps = (unsigned)lp;
  a.s_bytes = (unsigned)(sb + 2 * sp); a.l_bytes = (unsigned)(lb + 2 * lp);
  a.flops = 2.0 * a.B * (double)a.PH * a.PW * d->Cout * d->Cin * d->KH * d->KW;
  *nsplit_out = nsplit; *Nrows_out = a.Nrows; *Kw_out = a.Kw;
  dim3 grid(a.Ls / (32 * x3h_cb(d)), a.Ss / 64, nsplit);
  // (measured in round 6 and not kept: 16 waves = 8 tap groups of 2 taps, 106 registers, four waves per SIMD: -8 ... +5 % per
  // layer against run-to-run differences of the same size)
  // k3 s2 (round 6: the downsampling convolutions of Discriminator96 / ResnetGenerator and its ConvTranspose upsamplers ran on
  // the split-M GEMM tiles at 65-140 TFLOP/s): the same geometry class as k4 s2 - four residue planes of the halo, a 4 x 8 patch -
  // with 9 taps: 6 waves = 3 tap groups x 2 halves of S, 3 stages of 48 KB
  if (d->KH == 3 && d->stride == 2) return launch_wx3<3, 3, 2, 4, 3, 1, 3>(a, grid, st);
  if (d->KH == 4) return launch_wx3<4, 4, 2, 4, 4, 1, 3>(a, grid, st);       // 8 waves: 4 tap groups x 2 halves of S; 3 stages of 48 KB
  return launch_wx3<3, 3, 1, 8, 3, 2, 2>(a, grid, st);                       // 12 waves: 3 tap groups x 2 x 2; 2 stages of 72 KB
}

}  // namespace iprgan
