// Backward-weight for bf16 tensors with the gathered operand staged ONCE per pixel patch ("halo" form).
//
//   dW[n][tap][c] = sum_m S[m][n] * L[m * stride + tap - pad][c]
// S = the tensor on the small grid (Conv2d: dy on the output grid; ConvTranspose2d: x on the input grid), L = the
// tensor that is gathered around it tap by tap (Conv2d: x; ConvTranspose2d: dy).  The split-M GEMM of conv_igemm.hip
// (wgrad_kernel) streams an im2col view of L: every L element crosses L2 -> LDS once per tap that touches it (16x for
// k4, 9x for k3) and the kernel is bound by that traffic (DCGAN-128 D.conv1: 2.1 GB per launch for 0.67 GB of
// tensors; 320 TFLOP/s).  Here a block owns a 64-channel tile of S, a 64-channel chunk of L and ALL taps:
//   * per step it stages one 8x8 patch of S pixels (64 x 64 channels) and the halo of L pixels around it
//     (((8-1) stride + K)^2 pixels x 64 channels) by LDS-DMA (`buffer_load_dwordx4 ... lds`): L crosses L2 -> LDS once;
//     zero padding and ragged edges are out-of-range DMA offsets (zeros);
//   * every wave owns TPW taps x (64 x 64) of the accumulators; its B fragments for tap (ty, tx) are the halo rows shifted
//     by the tap - an immediate offset of the LDS address.  LDS rows are 64 bytes (32 channels of one pixel), pixels of
//     one stride residue are consecutive rows, so the four pixel rows of a `ds_read_b64_tr_b16` block are 256 contiguous
//     bytes: conflict-free for every tap shift without a swizzle;
//   * stages form a ring: counted `s_waitcnt vmcnt`, one raw `s_barrier` per patch (as conv_pipe.hip).
// Output: partial slabs ws[split][n][tap * Ls + c] in the layout of wgrad_kernel, summed in fixed order and scattered to
// PyTorch layout by wgrad_reduce_kernel (deterministic).
// Reference layers: networks/sn_discriminator.py:9-18 (k3 s1 / k4 s2 Conv2d), networks/conv_generator.py:8 (k4 s2 ConvT).
#include "conv_shared.h"

namespace iprgan {

struct WHaloArgs {
  const void* S;        // [B][PH][PW][Ss] bf16
  const void* L;        // [B][QH][QW][Ls] bf16
  float* ws;            // slabs [nsplit][Nrows][Kw]
  int B, PH, PW, QH, QW, Ss, Ls, pad;
  int PTY, PTX;         // patches per image
  FastDiv d_ptx, d_ppi; // / PTX, / (PTY * PTX)
  int npatch, pps;      // patches in all, patches per split
  int Nrows, Kw;
  unsigned s_bytes, l_bytes;
  double flops;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// One LDS-DMA wave-instruction, hidden from hipcc in inline asm: with the builtin form the compiler orders every later
// ds_read_b64_tr_b16 behind ALL outstanding LDS-DMA (an `s_waitcnt vmcnt(0)` at the top of each compute phase, which
// serialises the ring: measured 42 % of the wave cycles parked).  In asm the loads are invisible to its bookkeeping
// and are retired by the counted waits below only.  M0 (the LDS base of the DMA) is written in the same statement.
// rs: buffer descriptor words, wave-uniform (SGPRs), built once before the loop.
typedef unsigned int wh_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wh_dma16(wh_u32x4 rs, unsigned lds_addr, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lds_addr), "s"(rs) : "memory");
#endif
}
__device__ __forceinline__ wh_u32x4 wh_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  wh_u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);       // stride 0
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
template <int N>
__device__ __forceinline__ void wh_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 wh_tr_read8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
  union { s16x4_t h[2]; bf16x8 v; } u;
  u.h[0] = lo; u.h[1] = hi;
  return u.v;
}

template <int KH, int KW, int STR, int TPW, int NSTAGE, int CB>
struct WHGeom {
  static constexpr int TY = 8, TX = 8, NTAP = KH * KW;
  static constexpr int NWAVE = (NTAP + TPW - 1) / TPW;
  static constexpr int HH = (TY - 1) * STR + KH, HW = (TX - 1) * STR + KW;
  static constexpr int PHS = (HH + STR - 1) / STR, PWS = (HW + STR - 1) / STR, PP = PHS * PWS;   // rows of one stride-residue plane
  static constexpr int HROWS = STR * STR * PP;                 // halo pixel rows per 32-channel block
  static constexpr int HPP = (HROWS + 15) / 16 * 16;
  static constexpr int Q_ROWS = CB * HPP, P_ROWS = 2 * 64;      // 64-byte LDS rows: [cb][halo pixel], [nb][patch pixel]
  static constexpr int NINST = (Q_ROWS + P_ROWS) / 16;         // LDS-DMA wave-instructions per stage (16 rows each)
  static constexpr int LPW = (NINST + NWAVE - 1) / NWAVE;      // per wave (the last ones may be dummies into the pad)
  static constexpr int STAGE_BYTES = LPW * NWAVE * 1024;
  static constexpr int Q_BYTES = Q_ROWS * 64;
  static constexpr int SMEM = NSTAGE * STAGE_BYTES;
  static_assert(SMEM <= 160 * 1024, "ring larger than LDS");
  static_assert(Q_BYTES + P_ROWS * 64 < 65536, "immediate offsets of the fragment reads are 16 bits");
};

constexpr int wh_threads(int ntap, int tpw) { return (ntap + tpw - 1) / tpw * 64; }

template <int KH, int KW, int STR, int TPW, int NSTAGE, int CB>
__global__ __launch_bounds__(wh_threads(KH * KW, TPW)) void wgrad_halo_kernel(const WHaloArgs a) {
  using G = WHGeom<KH, KW, STR, TPW, NSTAGE, CB>;
  constexpr int NWAVE = G::NWAVE, LPW = G::LPW;
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

  // logical block order: (c chunk, n tile) fastest, then the split: the blocks that read the same patches share an XCD's L2
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const int cx = (int)(lt % gridDim.x), ny = (int)((lt / gridDim.x) % gridDim.y), split = (int)(lt / (gridDim.x * gridDim.y));
  const int c0 = cx * (CB * 32), n0 = ny * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int pbeg = split * a.pps, pend = pbeg + a.pps;
  if (pend > a.npatch) pend = a.npatch;
  const int np = pend - pbeg;

  const wh_u32x4 rs_s = wh_make_rsrc(a.S, a.s_bytes), rs_l = wh_make_rsrc(a.L, a.l_bytes);

  // what this lane moves in its LPW DMA slots of a stage: LDS row (16 * inst + lane / 4), 16-byte piece lane % 4.
  // rel = byte offset from the patch origin (of L for halo rows, of S for patch rows), ryx = (y << 16) | x pixel offset
  // from that origin for the bounds check; pad rows and dummy instructions carry y = 0x7fff: always out of range
  int rel[LPW], ryx[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int inst = i * NWAVE + wave, row = inst * 16 + (lane >> 2), piece = lane & 3;
    rel[i] = 0; ryx[i] = 0x7fff << 16;
    if (row < G::Q_ROWS) {
      const int cb = row / G::HPP, hp = row % G::HPP;
      if (hp < G::HROWS) {
        const int plane = hp / G::PP, r = hp % G::PP, hyy = r / G::PWS, hxx = r % G::PWS;
        const int hy = hyy * STR + plane / STR, hx = hxx * STR + plane % STR;
        rel[i] = ((hy * a.QW + hx) * a.Ls + c0 + cb * 32) * 2 + piece * 16;
        ryx[i] = (hy << 16) | hx;
      }
    } else if (row < G::Q_ROWS + G::P_ROWS) {
      const int pr = row - G::Q_ROWS, nb = pr >> 6, m = pr & 63, my = m >> 3, mx = m & 7;
      rel[i] = ((my * a.PW + mx) * a.Ss + n0 + nb * 32) * 2 + piece * 16;
      ryx[i] = (my << 16) | mx;
    }
  }

  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  auto issue = [&](int patch, int buf) {
    const int b = fdiv(patch, a.d_ppi);
    const int r = patch - b * (a.PTY * a.PTX);
    const int pty = fdiv(r, a.d_ptx), ptx = r - pty * a.PTX;
    const int py0 = pty * 8, px0 = ptx * 8;
    const int qy0 = py0 * STR - a.pad, qx0 = px0 * STR - a.pad;
    const int lbase = ((b * a.QH + qy0) * a.QW + qx0) * a.Ls * 2;          // may be negative: only used when in range
    const int sbase = ((b * a.PH + py0) * a.PW + px0) * a.Ss * 2;
    const unsigned sb = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)buf * G::STAGE_BYTES + (unsigned)wave * 1024u);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      // every wave issues exactly LPW DMA instructions per stage (the counted vmcnt relies on it); which tensor a slot
      // reads is wave-uniform (Q_ROWS % 16 == 0): scalar selects, no lane-dependent control flow
      const bool isS = (i * NWAVE + wave) * 16 >= G::Q_ROWS;
      const int y = (isS ? py0 : qy0) + (ryx[i] >> 16), x = (isS ? px0 : qx0) + (ryx[i] & 0xffff);
      const bool ok = ((unsigned)y < (unsigned)(isS ? a.PH : a.QH)) & ((unsigned)x < (unsigned)(isS ? a.PW : a.QW));
      const unsigned off = ok ? (unsigned)((isS ? sbase : lbase) + rel[i]) : OOB_OFFSET;
      if (isS) wh_dma16(rs_s, sb + (unsigned)(i * NWAVE) * 1024u, off);
      else wh_dma16(rs_l, sb + (unsigned)(i * NWAVE) * 1024u, off);
    }
  };

  f32x16 acc[TPW][2][CB];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  // fragment addressing (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of its 16
  // columns; lanes 16-31 take columns 16-31 of the 32-channel row; lanes 32-63 the second 8 of the 16 reduction rows)
  const int half = lane >> 5, gq = (lane & 15) >> 2, gp = lane & 3, gcol = (lane >> 4) & 1;
  const unsigned lane_col = (unsigned)(gcol * 32 + gp * 8);
  const unsigned a_lane = G::Q_BYTES + (unsigned)((8 * half + gq) * 64) + lane_col;            // + nb*4096 + ks*1024 + rd*256
  unsigned b_tap[TPW];                          // + cb*HPP*64 + (2*ks*PWS + 4*rd)*64: this wave's tap shift is a row offset
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = wave * TPW + tt < G::NTAP ? wave * TPW + tt : 0;
    const int ty = tap / KW, tx = tap % KW;
    const int trow = ((ty % STR) * STR + tx % STR) * G::PP + (ty / STR) * G::PWS + tx / STR;
    b_tap[tt] = (unsigned)((trow + half * G::PWS + gq) * 64) + lane_col;
  }
  const char* ldsc = (const char*)lds;

  auto compute = [&](int buf) {
    const char* sb = ldsc + buf * G::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = wh_tr_read8(sb + a_lane + i * 4096 + ks * 1024, sb + a_lane + i * 4096 + ks * 1024 + 256);
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        bf16x8 bfr[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j)
          bfr[j] = wh_tr_read8(sb + b_tap[tt] + j * (G::HPP * 64) + (2 * ks * G::PWS) * 64,
                               sb + b_tap[tt] + j * (G::HPP * 64) + (2 * ks * G::PWS + 4) * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            acc[tt][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[tt][i][j], 0, 0, 0);
      }
    }
  };

  // ---- the ring over this split's patches
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < np) issue(pbeg + s, s);
  int cur = 0, nxt = NSTAGE - 1;
  for (int t = 0; t < np; ++t) {
    const int rem = np - 1 - t;
    const int inflight = rem < NSTAGE - 2 ? rem : (NSTAGE - 2 < 3 ? NSTAGE - 2 : 3);
    static_assert(NSTAGE <= 5 && 3 * LPW <= 63, "ring depth / vmcnt field");
    if (inflight >= 3) wh_wait_vmcnt<(3 * LPW < 63 ? 3 * LPW : 63)>();
    else if (inflight == 2) wh_wait_vmcnt<2 * LPW>();
    else if (inflight == 1) wh_wait_vmcnt<LPW>();
    else wh_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (rem >= NSTAGE - 1) issue(pbeg + t + NSTAGE - 1, nxt);
    compute(cur);
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }

  // ---- slab store: ws[split][n0 + i*32 + row][tap * Ls + c0 + j*32 + col], C layout row = (r&3) + 8*(r>>2) + 4*half
  const int l31 = lane & 31;
  float* slab = a.ws + (size_t)split * a.Nrows * a.Kw;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = wave * TPW + tt;
    if (tap >= G::NTAP) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[(size_t)n * a.Kw + tap * a.Ls + c0 + j * 32 + l31] = acc[tt][i][j][r];
        }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The halo form for fp32 tensors (every fp32 workload: the headline DCGAN-64, SRGAN, CycleGAN): same staging - one
// 8x8 patch of the small grid + the halo of the gathered tensor per step, taps as LDS row shifts - on the exact fp32
// MFMA (v_mfma_f32_32x32x2_f32: k = 2 pixels per instruction).  An LDS row is 32 channels x 4 bytes = 128 bytes of one
// pixel; lane (l31, half) reads element l31 of pixel 2 ks + half with one ds_read_b32 (the 32 lanes of a half cover a
// row: conflict-free), so both operands are read straight in MFMA layout with immediate offsets per k step and tap.
// A block owns 64 S-channels x 32 L-channels x all taps (one tap per wave: 32 accumulator registers); the fp32 MFMA is
// 16x slower than the bf16 one, so the DMA (~57 KB per 16 k cycles) is nowhere near a bound and the ring of two stages
// hides it.  ReflectionPad2d (CycleGAN's residual convolutions) is folded into the DMA offsets.
struct WHalo32Args {
  const float* S;       // [B][PH][PW][Ss]
  const float* L;       // [B][QH][QW][Ls]
  float* ws;
  int B, PH, PW, QH, QW, Ss, Ls, pad, reflect;
  int PTY, PTX;
  FastDiv d_ptx, d_ppi;
  int npatch, pps, Nrows, Kw;
  unsigned s_bytes, l_bytes;
  double flops;
};

template <int KH, int KW, int STR, int NSTAGE>
struct WH32Geom {
  static constexpr int NTAP = KH * KW, NWAVE = NTAP;            // one tap per wave
  static constexpr int HH = 7 * STR + KH, HW = 7 * STR + KW;
  static constexpr int PHS = (HH + STR - 1) / STR, PWS = (HW + STR - 1) / STR, PP = PHS * PWS;
  static constexpr int HROWS = STR * STR * PP;                  // halo pixel rows (128 bytes each)
  static constexpr int HPP = (HROWS + 7) / 8 * 8;
  static constexpr int P_ROWS = 2 * 64;                         // [nb][patch pixel]
  static constexpr int NINST = (HPP + P_ROWS) / 8;              // 8 rows per DMA instruction
  static constexpr int LPW = (NINST + NWAVE - 1) / NWAVE;
  static constexpr int STAGE_BYTES = LPW * NWAVE * 1024;
  static constexpr int Q_BYTES = HPP * 128;
  static constexpr int SMEM = NSTAGE * STAGE_BYTES;
  static_assert(SMEM <= 160 * 1024 && Q_BYTES + P_ROWS * 128 < 65536 && LPW * (NSTAGE - 1) <= 63, "ring geometry");
};

template <int KH, int KW, int STR, int NSTAGE>
__global__ __launch_bounds__(KH * KW * 64) void wgrad_halo_f32_kernel(const WHalo32Args a) {
  using G = WH32Geom<KH, KW, STR, NSTAGE>;
  constexpr int NWAVE = G::NWAVE, LPW = G::LPW;
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const int cx = (int)(lt % gridDim.x), ny = (int)((lt / gridDim.x) % gridDim.y), split = (int)(lt / (gridDim.x * gridDim.y));
  const int c0 = cx * 32, n0 = ny * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int pbeg = split * a.pps, pend = pbeg + a.pps;
  if (pend > a.npatch) pend = a.npatch;
  const int np = pend - pbeg;
  const wh_u32x4 rs_s = wh_make_rsrc(a.S, a.s_bytes), rs_l = wh_make_rsrc(a.L, a.l_bytes);

  // DMA slots of this lane: LDS row 8 * inst + lane / 8, 16-byte piece lane % 8 (8 pieces = 32 fp32 channels)
  int rel[LPW], ryx[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int inst = i * NWAVE + wave, row = inst * 8 + (lane >> 3), piece = lane & 7;
    rel[i] = 0; ryx[i] = 0x7fff << 16;
    if (row < G::HPP) {
      if (row < G::HROWS) {
        const int plane = row / G::PP, r = row % G::PP, hyy = r / G::PWS, hxx = r % G::PWS;
        const int hy = hyy * STR + plane / STR, hx = hxx * STR + plane % STR;
        rel[i] = c0 * 4 + piece * 16;            // the pixel part depends on the (possibly mirrored) position: added per stage
        ryx[i] = (hy << 16) | hx;
      }
    } else if (row < G::HPP + G::P_ROWS) {
      const int pr = row - G::HPP, nb = pr >> 6, m = pr & 63, my = m >> 3, mx = m & 7;
      rel[i] = (n0 + nb * 32) * 4 + piece * 16;
      ryx[i] = (my << 16) | mx;
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  auto issue = [&](int patch, int buf) {
    const int b = fdiv(patch, a.d_ppi);
    const int r = patch - b * (a.PTY * a.PTX);
    const int pty = fdiv(r, a.d_ptx), ptx = r - pty * a.PTX;
    const int py0 = pty * 8, px0 = ptx * 8;
    const int qy0 = py0 * STR - a.pad, qx0 = px0 * STR - a.pad;
    const unsigned sb = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)buf * G::STAGE_BYTES + (unsigned)wave * 1024u);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const bool isS = (i * NWAVE + wave) * 8 >= G::HPP;               // wave-uniform: HPP % 8 == 0
      int y = (isS ? py0 : qy0) + (ryx[i] >> 16), x = (isS ? px0 : qx0) + (ryx[i] & 0xffff);
      const bool valid = (ryx[i] >> 16) != 0x7fff;
      if (a.reflect && !isS) { y = reflect_idx(y, a.QH); x = reflect_idx(x, a.QW); }
      const int H = isS ? a.PH : a.QH, W = isS ? a.PW : a.QW, Cs = isS ? a.Ss : a.Ls;
      const bool ok = valid & ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((b * H + y) * W + x) * Cs) * 4u + (unsigned)rel[i] : OOB_OFFSET;
      if (isS) wh_dma16(rs_s, sb + (unsigned)(i * NWAVE) * 1024u, off);
      else wh_dma16(rs_l, sb + (unsigned)(i * NWAVE) * 1024u, off);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int half = lane >> 5, l31 = lane & 31;
  const int tap = wave, ty = tap / KW, tx = tap % KW;
  const int trow = ((ty % STR) * STR + tx % STR) * G::PP + (ty / STR) * G::PWS + tx / STR;
  const unsigned a_lane = G::Q_BYTES + (unsigned)(half * 128 + l31 * 4);          // + nb * 8192 + ks * 256
  const unsigned b_lane = (unsigned)((trow + half) * 128 + l31 * 4);              // + ((ks >> 2) * PWS + 2 * (ks & 3)) * 128
  const char* ldsc = (const char*)lds;
  auto compute = [&](int buf) {
    const char* sb = ldsc + buf * G::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {            // pixels 2 ks, 2 ks + 1 of the patch = (ks >> 2, 2 (ks & 3) + half)
      const float a0 = *(const float*)(sb + a_lane + ks * 256);
      const float a1 = *(const float*)(sb + a_lane + 8192 + ks * 256);
      const float bv = *(const float*)(sb + b_lane + ((ks >> 2) * G::PWS + 2 * (ks & 3)) * 128);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < np) issue(pbeg + s, s);
  int cur = 0, nxt = NSTAGE - 1;
  for (int t = 0; t < np; ++t) {
    const int rem = np - 1 - t;
    const int inflight = rem < NSTAGE - 2 ? rem : NSTAGE - 2;
    if (inflight >= 1) wh_wait_vmcnt<(NSTAGE > 2 ? LPW : 0)>(); else wh_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (rem >= NSTAGE - 1) issue(pbeg + t + NSTAGE - 1, nxt);
    compute(cur);
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  float* slab = a.ws + (size_t)split * a.Nrows * a.Kw;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      slab[(size_t)n * a.Kw + tap * a.Ls + c0 + l31] = acc[i][r];
    }
}

bool wgrad_halo_f32_eligible(const iprgan_conv_desc* d) {
  if (d->x_bf16 || d->y_bf16 || d->KH != d->KW) return false;
  const int Ss = d->transposed ? d->Cin : d->Cout, Ls = d->transposed ? d->Cout : d->Cin;
  if ((Ss % 64) != 0 || (Ls % 32) != 0) return false;
  if (d->pad_mode == IPRGAN_PAD_REFLECT && (d->transposed || d->pad >= d->H || d->pad >= d->W)) return false;
  return (d->KH == 3 && (d->stride == 1 || d->stride == 2)) || (d->KH == 4 && (d->stride == 1 || d->stride == 2));
}
int wgrad_halo_f32_nsplit(const iprgan_conv_desc* d, int target_blocks) {
  const int PH = d->transposed ? d->H : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int PW = d->transposed ? d->W : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  const int npatch = d->B * cdiv(PH, 8) * cdiv(PW, 8);
  const int Ss = d->transposed ? d->Cin : d->Cout, Ls = d->transposed ? d->Cout : d->Cin;
  const int tiles = (Ls / 32) * (Ss / 64);
  int want = cdiv(target_blocks, tiles);
  if (want < 1) want = 1;
  if (want > npatch) want = npatch;
  return cdiv(npatch, cdiv(npatch, want));
}
template <int KH, int KW, int STR, int NSTAGE>
static int launch_wh32(const WHalo32Args& a, dim3 grid, hipStream_t st) {
  using G = WH32Geom<KH, KW, STR, NSTAGE>;
  auto kern = wgrad_halo_f32_kernel<KH, KW, STR, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM); attr_set = true; }
  prof_launch(kern, grid, dim3(G::NWAVE * 64), (size_t)G::SMEM, st, 25, a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}
int launch_wgrad_halo_f32(const iprgan_conv_desc* d, const float* x, const float* dy, float* ws, int target_blocks,
                          hipStream_t st, int* nsplit_out, int* Nrows_out, int* Kw_out) {
  if (!wgrad_halo_f32_eligible(d)) return -1;
  const int OH = d->transposed ? (d->H - 1) * d->stride - 2 * d->pad + d->KH + d->outpad : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int OW = d->transposed ? (d->W - 1) * d->stride - 2 * d->pad + d->KW + d->outpad : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  WHalo32Args a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.pad = d->pad; a.reflect = d->pad_mode == IPRGAN_PAD_REFLECT;
  if (d->transposed) { a.S = x; a.L = dy; a.PH = d->H; a.PW = d->W; a.QH = OH; a.QW = OW; a.Ss = d->Cin; a.Ls = d->Cout; }
  else { a.S = dy; a.L = x; a.PH = OH; a.PW = OW; a.QH = d->H; a.QW = d->W; a.Ss = d->Cout; a.Ls = d->Cin; }
  a.ws = ws;
  a.PTY = cdiv(a.PH, 8); a.PTX = cdiv(a.PW, 8);
  a.d_ptx = make_fastdiv(a.PTX); a.d_ppi = make_fastdiv(a.PTY * a.PTX);
  a.npatch = a.B * a.PTY * a.PTX;
  const int nsplit = wgrad_halo_f32_nsplit(d, target_blocks);
  a.pps = cdiv(a.npatch, nsplit);
  a.Nrows = a.Ss; a.Kw = d->KH * d->KW * a.Ls;
  const unsigned long long sb = (unsigned long long)a.B * a.PH * a.PW * a.Ss * 4, lb = (unsigned long long)a.B * a.QH * a.QW * a.Ls * 4;
  IPR_CHECK(sb < 0x7fffffffull && lb < 0x7fffffffull, "conv_bwd_weight: tensor larger than 2 GiB");
  IPR_CHECK(a.QH < 32000 && a.QW < 32000, "conv_bwd_weight: image too large for the halo form");
  a.s_bytes = (unsigned)sb; a.l_bytes = (unsigned)lb;
  a.flops = 2.0 * a.B * (double)a.PH * a.PW * d->Cout * d->Cin * d->KH * d->KW;
  *nsplit_out = nsplit; *Nrows_out = a.Nrows; *Kw_out = a.Kw;
  dim3 grid(a.Ls / 32, a.Ss / 64, nsplit);
  if (d->KH == 4) return d->stride == 2 ? launch_wh32<4, 4, 2, 2>(a, grid, st) : launch_wh32<4, 4, 1, 2>(a, grid, st);
  return d->stride == 2 ? launch_wh32<3, 3, 2, 2>(a, grid, st) : launch_wh32<3, 3, 1, 3>(a, grid, st);
}

// ------------------------------------------------------------------------------------------------------------------
// Backward-weight of the RGB layers (3 -> C stems, C -> 3 heads; k3 s1 p1): dW[n][c][tap] = sum_p T64[p][n] T4[p+tap-1][c]
// with T64 the bf16 tensor with many channels (Conv2d 3->C: dy; ConvTranspose2d C->3: x) and T4 the fp32 NHWC4 image
// (x / dy).  19 GFLOP over 0.6 GB at DCGAN-128 batch 256: bound by READING T64 once.  The split-M GEMM spent 330 us on it
// (1.8 TB/s: 64x64 tiles whose K = 36 columns fill one MFMA column block in two).  Here a block streams tiles of 256
// pixels (whole image rows): T64 by LDS-DMA into two [pixel][32 channel] planes (64-byte rows: the transposed reads of
// 4 consecutive pixels are conflict-free), the fp32 halo rows of T4 by LDS-DMA as well (zero padding = out-of-range
// offsets); B fragments [pixel][(tap, c)] are assembled from the halo with eight 4-byte LDS reads at the tap's shift and
// rounded to bf16 (the math mode's operand rounding).  Every wave reduces its 64 pixels of the tile into all 64 x 64
// (36 used) accumulators; waves and blocks are combined in fixed order (LDS, then the slab reduce).
// Layers: networks/sn_discriminator.py:9 (3 -> 64), networks/conv_generator.py:21 (64 -> 3).
struct WRgbArgs {
  const void* T64;      // [B][H][W][Cs] bf16
  const float* T4;      // [B][H][W][4] fp32
  float* ws;            // slabs [nsplit][Nrows][64]
  int B, H, W, Cs, R;   // R = image rows per tile (R * W = 256)
  int tpi, ntile, tps;  // tiles per image, tiles in all, tiles per split
  FastDiv d_tpi, d_w;
  int Nrows;
  unsigned t64_bytes, t4_bytes;
  double flops;
};

#define WRGB_TPIX 256
#define WRGB_NSTAGE 3

__global__ __launch_bounds__(256) void wgrad_rgb_kernel(const WRgbArgs a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  constexpr int A_BYTES = 2 * WRGB_TPIX * 64;                         // two 32-channel planes
  const int HROW = a.W + 2, HPIX = (a.R + 2) * HROW;                  // halo: R + 2 rows of W + 2 pixels, 16 bytes each
  const int HINST = (HPIX + 63) / 64;                                 // DMA instructions (64 pixels each)
  const int STAGE = A_BYTES + (HINST + 1) * 1024;                     // + one kilobyte that swallows the dummy instructions
  const int zero_off = WRGB_NSTAGE * STAGE;                           // 128 bytes of zeros behind the ring
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, n0 = blockIdx.y * 64;
  char* ldsc = (char*)lds;
  if (tid < 32) *(float*)(ldsc + zero_off + tid * 4) = 0.f;

  int tbeg = split * a.tps, tend = tbeg + a.tps;
  if (tend > a.ntile) tend = a.ntile;
  const int nt = tend - tbeg;
  const wh_u32x4 rs64 = wh_make_rsrc(a.T64, a.t64_bytes), rs4 = wh_make_rsrc(a.T4, a.t4_bytes);
  const unsigned lds_base = (unsigned)(uintptr_t)lds;

  // DMA slots of this wave per stage: 8 for T64 (32 instructions of 16 pixel rows x 64 B over 4 waves), then its share
  // of the halo instructions.  All waves issue the same count (dummies read out of range into the stage's pad).
  auto issue = [&](int tile, int buf) {
    const int b = fdiv(tile, a.d_tpi);
    const int r0 = (tile - b * a.tpi) * a.R;
    const unsigned sb = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)buf * STAGE);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int inst = i * 4 + wave;                                  // 0..31: plane = inst / 16, pixels 16 * (inst % 16) ..
      const int plane = inst >> 4, m = (inst & 15) * 16 + (lane >> 2), piece = lane & 3;
      const int r = fdiv(m, a.d_w), x = m - r * a.W;
      const bool ok = r0 + r < a.H;
      const unsigned off = ok ? (unsigned)((((b * a.H + r0 + r) * a.W + x) * a.Cs + n0 + plane * 32) * 2 + piece * 16) : OOB_OFFSET;
      wh_dma16(rs64, __builtin_amdgcn_readfirstlane(sb + (unsigned)inst * 1024u), off);
    }
    for (int i = 0; i < 6; ++i) {                                     // fixed trip count: the counted vmcnt needs it
      const int inst = i * 4 + wave, hp = inst * 64 + lane;           // halo pixel of this lane (16 bytes)
      const int hr = hp / HROW, hx = hp - hr * HROW;
      const int y = r0 - 1 + hr, x = hx - 1;
      const bool ok = inst < HINST && hp < HPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)(((b * a.H + y) * a.W + x) * 16) : OOB_OFFSET;
      // instructions past the halo (every wave issues six) write zeros into the stage's spare kilobyte
      const unsigned dst = sb + A_BYTES + (unsigned)(inst < HINST ? inst : HINST) * 1024u;
      wh_dma16(rs4, __builtin_amdgcn_readfirstlane(dst), off);
    }
  };
  constexpr int LPW = 8 + 6;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, gq = (lane & 15) >> 2, gp = lane & 3, gcol = (lane >> 4) & 1, l31 = lane & 31;
  const unsigned a_lane = (unsigned)((wave * 64 + 8 * half + gq) * 64 + gcol * 32 + gp * 8);     // + plane*16384 + ks*1024 + rd*256
  // B column of this lane in column block cf: col = cf * 32 + l31 = tap * 4 + c
  unsigned b_lane[2];
  bool b_ok[2];
#pragma unroll
  for (int cf = 0; cf < 2; ++cf) {
    const int col = cf * 32 + l31, tap = col >> 2, c = col & 3, ty = tap / 3, tx = tap - ty * 3;
    b_ok[cf] = col < 36;
    b_lane[cf] = (unsigned)(A_BYTES + ((ty * HROW + tx + 8 * half) * 16 + c * 4));
  }

  auto compute = [&](int buf) {
    const char* sb = ldsc + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = wh_tr_read8(sb + a_lane + i * 16384 + ks * 1024, sb + a_lane + i * 16384 + ks * 1024 + 256);
      const int m0 = wave * 64 + ks * 16;                             // first pixel of this k step: 16 pixels in one image row
      const int r = fdiv(m0, a.d_w), x0 = m0 - r * a.W;
      const unsigned srow = (unsigned)((r * HROW + x0) * 16);
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) {
        const char* bp = b_ok[cf] ? sb + b_lane[cf] + srow : ldsc + zero_off;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const float*)(bp + i * 16);
        bfr[cf] = bf16x8{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
          acc[i][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[cf], acc[i][cf], 0, 0, 0);
    }
  };

  __syncthreads();                                                   // the zero area is written
#pragma unroll
  for (int s = 0; s < WRGB_NSTAGE - 1; ++s)
    if (s < nt) issue(tbeg + s, s);
  int cur = 0, nxt = WRGB_NSTAGE - 1;
  for (int t = 0; t < nt; ++t) {
    const int rem = nt - 1 - t;
    if (rem >= 1) wh_wait_vmcnt<LPW>();
    else wh_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (rem >= WRGB_NSTAGE - 1) issue(tbeg + t + WRGB_NSTAGE - 1, nxt);
    compute(cur);
    cur = cur + 1 == WRGB_NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == WRGB_NSTAGE ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // combine the four waves (each reduced its own 64 pixels of every tile) in wave order, then one slab per block
  float* red = (float*)lds;                                           // [wave][64 n][64 col]: 64 KB of the ring
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[(wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + cf * 32 + l31] = acc[i][cf][r];
  __syncthreads();
  float* slab = a.ws + ((size_t)split * a.Nrows + n0) * 64;
  for (int e = tid; e < 64 * 64; e += 256) {
    const float s = ((red[e] + red[4096 + e]) + red[8192 + e]) + red[12288 + e];
    slab[e] = s;
  }
}

bool wgrad_rgb_eligible(const iprgan_conv_desc* d) {
  // T64 = dy (Conv2d 3->C) or x (ConvT C->3) must be bf16, the 3-channel side fp32 NHWC4
  if (d->pad_mode != IPRGAN_PAD_ZERO || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->outpad != 0) return false;
  const int n64 = d->transposed ? d->Cin : d->Cout, n4 = d->transposed ? d->Cout : d->Cin;
  const bool t64_16 = d->transposed ? d->x_bf16 : d->y_bf16, t4_16 = d->transposed ? d->y_bf16 : d->x_bf16;
  if (n4 > 4 || (n64 % 64) != 0 || !t64_16 || t4_16) return false;
  return (d->W % 16) == 0 && d->W <= 256 && (WRGB_TPIX % d->W) == 0;
}
int wgrad_rgb_nsplit(const iprgan_conv_desc* d, int target_blocks) {
  const int R = WRGB_TPIX / d->W, ntile = d->B * cdiv(d->H, R);
  const int n64 = d->transposed ? d->Cin : d->Cout;
  int want = cdiv(target_blocks, n64 / 64);
  if (want > ntile) want = ntile;
  if (want < 1) want = 1;
  return cdiv(ntile, cdiv(ntile, want));
}
int launch_wgrad_rgb(const iprgan_conv_desc* d, const void* x, const void* dy, float* ws, int target_blocks, hipStream_t st,
                     int* nsplit_out, int* Nrows_out, int* Kw_out) {
  if (!wgrad_rgb_eligible(d)) return -1;
  WRgbArgs a;
  memset(&a, 0, sizeof(a));
  a.T64 = d->transposed ? x : dy; a.T4 = (const float*)(d->transposed ? dy : x);
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cs = d->transposed ? d->Cin : d->Cout;
  a.R = WRGB_TPIX / d->W; a.tpi = cdiv(d->H, a.R); a.ntile = a.B * a.tpi;
  const int nsplit = wgrad_rgb_nsplit(d, target_blocks);
  a.tps = cdiv(a.ntile, nsplit);
  a.d_tpi = make_fastdiv(a.tpi); a.d_w = make_fastdiv(a.W);
  a.Nrows = a.Cs; a.ws = ws;
  const unsigned long long b64 = (unsigned long long)a.B * a.H * a.W * a.Cs * 2, b4 = (unsigned long long)a.B * a.H * a.W * 16;
  IPR_CHECK(b64 < 0x7fffffffull && b4 < 0x7fffffffull, "conv_bwd_weight: tensor larger than 2 GiB");
  a.t64_bytes = (unsigned)b64; a.t4_bytes = (unsigned)b4;
  a.flops = 2.0 * a.B * (double)a.H * a.W * d->Cout * d->Cin * 9;
  const int HINST = cdiv((a.R + 2) * (a.W + 2), 64);
  IPR_CHECK(HINST <= 24, "conv_bwd_weight: halo too large for the RGB kernel");
  const size_t smem = (size_t)WRGB_NSTAGE * (2 * WRGB_TPIX * 64 + (HINST + 1) * 1024) + 128;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)wgrad_rgb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  *nsplit_out = nsplit; *Nrows_out = a.Nrows; *Kw_out = 64;
  prof_launch(wgrad_rgb_kernel, dim3(nsplit, a.Cs / 64), dim3(256), smem, st, 22, a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}

// ---- host side -----------------------------------------------------------------------------------------------
bool wgrad_halo_eligible(const iprgan_conv_desc* d) {
  if (!d->x_bf16 || !d->y_bf16 || d->pad_mode != IPRGAN_PAD_ZERO) return false;
  if ((d->Cin % 64) != 0 || (d->Cout % 64) != 0 || d->KH != d->KW) return false;
  return (d->KH == 3 && d->stride == 1) || (d->KH == 4 && d->stride == 2);
}

// variant: 0 = 64-channel chunk of L per block (k4: 2 taps per wave, 2 stages; k3: 3 stages),
//          1 = 32-channel chunk, deeper ring (k4: 2 taps per wave, 4 stages; k3: 5 stages),
//          2 = 32-channel chunk, one tap per wave (k4: 16 waves, 4 stages; k3: as 1 with 4 stages)
static int halo_cb(int variant) { return variant == 0 ? 2 : 1; }

// nsplit for a target number of blocks (the caller sizes the slabs with the same function)
int wgrad_halo_nsplit(const iprgan_conv_desc* d, int variant, int target_blocks) {
  const int PH = d->transposed ? d->H : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int PW = d->transposed ? d->W : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  const int npatch = d->B * cdiv(PH, 8) * cdiv(PW, 8);
  const int Ls = d->transposed ? d->Cout : d->Cin, Ss = d->transposed ? d->Cin : d->Cout;
  const int tiles = (Ls / (32 * halo_cb(variant))) * (Ss / 64);
  int want = cdiv(target_blocks, tiles);
  if (want < 1) want = 1;
  if (want > npatch) want = npatch;
  const int pps = cdiv(npatch, want);
  return cdiv(npatch, pps);
}

template <int KH, int KW, int STR, int TPW, int NSTAGE, int CB>
static int launch_wh(const WHaloArgs& a, dim3 grid, hipStream_t st) {
  using G = WHGeom<KH, KW, STR, TPW, NSTAGE, CB>;
  auto kern = wgrad_halo_kernel<KH, KW, STR, TPW, NSTAGE, CB>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM); attr_set = true; }
  prof_launch(kern, grid, dim3(G::NWAVE * 64), (size_t)G::SMEM, st, 21, a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}

// S / L: see the top of the file.  Slabs: ws[nsplit][Nrows = S channels][Kw = taps * L channels].
int launch_wgrad_halo(const iprgan_conv_desc* d, const void* x, const void* dy, float* ws, int variant, int target_blocks,
                      hipStream_t st, int* nsplit_out, int* Nrows_out, int* Kw_out) {
  if (!wgrad_halo_eligible(d) || variant < 0 || variant > 2) return -1;
  const int OH = d->transposed ? (d->H - 1) * d->stride - 2 * d->pad + d->KH + d->outpad : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int OW = d->transposed ? (d->W - 1) * d->stride - 2 * d->pad + d->KW + d->outpad : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  WHaloArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.pad = d->pad;
  if (d->transposed) { a.S = x; a.L = dy; a.PH = d->H; a.PW = d->W; a.QH = OH; a.QW = OW; a.Ss = d->Cin; a.Ls = d->Cout; }
  else { a.S = dy; a.L = x; a.PH = OH; a.PW = OW; a.QH = d->H; a.QW = d->W; a.Ss = d->Cout; a.Ls = d->Cin; }
  a.ws = ws;
  a.PTY = cdiv(a.PH, 8); a.PTX = cdiv(a.PW, 8);
  a.d_ptx = make_fastdiv(a.PTX); a.d_ppi = make_fastdiv(a.PTY * a.PTX);
  a.npatch = a.B * a.PTY * a.PTX;
  const int nsplit = wgrad_halo_nsplit(d, variant, target_blocks);
  a.pps = cdiv(a.npatch, nsplit);
  a.Nrows = a.Ss; a.Kw = d->KH * d->KW * a.Ls;
  const unsigned long long sb = (unsigned long long)a.B * a.PH * a.PW * a.Ss * 2, lb = (unsigned long long)a.B * a.QH * a.QW * a.Ls * 2;
  IPR_CHECK(sb < 0x7fffffffull && lb < 0x7fffffffull, "conv_bwd_weight: tensor larger than 2 GiB");
  IPR_CHECK(a.QH < 32000 && a.QW < 32000, "conv_bwd_weight: image too large for the halo form");
  a.s_bytes = (unsigned)sb; a.l_bytes = (unsigned)lb;
  a.flops = 2.0 * a.B * (double)a.PH * a.PW * d->Cout * d->Cin * d->KH * d->KW;
  *nsplit_out = nsplit; *Nrows_out = a.Nrows; *Kw_out = a.Kw;
  dim3 grid(a.Ls / (32 * halo_cb(variant)), a.Ss / 64, nsplit);
  if (d->KH == 4) {
    if (variant == 0) return launch_wh<4, 4, 2, 2, 2, 2>(a, grid, st);
    if (variant == 1) return launch_wh<4, 4, 2, 2, 4, 1>(a, grid, st);
    return launch_wh<4, 4, 2, 1, 4, 1>(a, grid, st);
  }
  if (variant == 0) return launch_wh<3, 3, 1, 1, 3, 2>(a, grid, st);
  if (variant == 1) return launch_wh<3, 3, 1, 1, 5, 1>(a, grid, st);
  return launch_wh<3, 3, 1, 1, 4, 1>(a, grid, st);
}

}  // namespace iprgan
