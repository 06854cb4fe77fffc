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

// K-loop probes (A/B builds only: -DX3P_PROBE=1 no MFMAs, 2 no refills; results are wrong, only the time matters)
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
  int u_c = 0, u_ty = 0, u_tx = 0;
  int w_dy = 0, w_dx = 0, w_tapoff = 0;
  unsigned w_wk = 0, w_sbase = 0;
  auto walk_begin = [&](int buf) {            // address pieces of the K step the walk points at, into stage `buf`
    w_dy = p_dy0 + u_ty * p_dys; w_dx = p_dx0 + u_tx * p_dxs;
    w_tapoff = ((w_dy * IW + w_dx) * Cs + u_c) * 2;
    w_wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c) * 2u;
    w_sbase = lds_base + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
  };
  auto walk_next = [&]() {
    if (++u_tx == p_tw) { u_tx = 0; if (++u_ty == p_th) { u_ty = 0; u_c += 32; } }
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

  pipe_epilogue<WGM, WGN, WM, WN, NSTAGE * STAGE_BYTES, STATS, PREF, false, true>(a, acc, (float*)lds, pz, lq, m0, n0, auxpf);
}

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
bool gconv_x3p_eligible(const GConvArgs& a) {
  if (a.in16 != 2 || a.ksplit > 1 || a.wmod > 0 || a.planar_M) return false;
  auto simple = [](int act) { return act == IPRGAN_ACT_NONE || act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; };
  if ((a.Ns % 8) != 0 || (a.Cs % 32) != 0 || !simple(a.act) || (a.aux && !simple(a.aux_act))) return false;    // pipe_epilogue
  if (a.bn_mean) return false;
  if (a.aux && a.aux16 == 2) return false;        // (callers pass the h plane as a bf16 operand: aux16 == 1)
  return true;
}

template <void (*KERN)(const GConvArgs)>
static void x3p_go(const GConvArgs& a, dim3 grid, dim3 block, size_t smem, hipStream_t st) {
  static bool attr_set = false;            // per kernel instantiation
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
  prof_launch(KERN, grid, block, smem, st, 29, a.flops, a);
}

template <int WGM, int WGN, int WM, int WN, int NSTAGE>
static int launch_x3p_t(const GConvArgs& a, hipStream_t st, int* bm_out) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = (size_t)NSTAGE * 3 * (BM + BN) * 64;
  dim3 grid(cdiv(maxM, BM), cdiv(a.Ns, BN), a.nphase);
  *bm_out = BM;
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

// variant: 0 = 256x128 (8 waves of 64x64, 2 stages, 144 KB), 1 = 128x128 (4 waves of 64x64, 3 stages, 144 KB),
//          2 = 128x64 (4 waves of 64x32, 3 stages, 108 KB), 3 = 128x64 (2 stages, 72 KB: two blocks per CU),
//          4 = 256x64 (4 waves of 64x64, 2 stages, 120 KB), 5 = 64x64 (4 waves of 32x32, 3 stages, 72 KB: two blocks per CU),
//          6 = 128x256 (4 waves of 64x128, 2 stages, 144 KB), 7 = 128x128 (2 stages, 96 KB)
// returns -1 when the variant does not apply to the geometry
int launch_gconv_x3p(const GConvArgs& a, int variant, hipStream_t st, int* bm_out) {
  if (!gconv_x3p_eligible(a)) return -1;
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
