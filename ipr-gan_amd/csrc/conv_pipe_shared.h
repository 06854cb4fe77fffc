// Shared by the LDS-DMA ring kernels (conv_pipe.hip: bf16 / exact-fp32 tiles; conv_x3.hip: three-plane tiles): the LDS-DMA
// wave-instruction, counted waits, and the row-major epilogue through LDS.
#pragma once
#include "conv_shared.h"
#include <type_traits>

namespace iprgan {

typedef __attribute__((address_space(3))) void lds_void;

// Cache policy of the epilogue's streams (the tile's output, the operand of the fused derivative): each byte is touched
// once, while the operand rows of the K loop are re-read tap after tap and by the neighbouring tiles of the same XCD.
// aux bit 1 = nt: the streams pass through L2 without displacing those rows.
#ifndef PIPE_NT
#define PIPE_NT 0          // measured: nt on these streams is neutral to -5 % (DCGAN-128 layers), so the default policy stays
#endif

// One LDS-DMA wave-instruction: lane l copies the 16 bytes at buffer offset voff (out of range: zeros) to LDS byte
// address lds_addr + 16 * l (lds_addr wave-uniform: it travels in M0).  Device pass only: in the host pass the builtin
// is an error that clang defers silently and then drops the kernel's host stub.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(uintptr_t)lds_addr, 16, voff, 0, 0, 0);
#endif
}

// Workgroup barrier for LDS hand-offs only: this wave's LDS operations are complete, then s_barrier.  __syncthreads()
// additionally drains vmcnt - in the epilogues below that is a wait for the tile's own output stores (or the next tile's
// prefetch) to COMPLETE, paid once per phase / tile for nothing.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef IPRGAN_X3WS_TIMING      // debug build only (scripts/probe/ws_phase_times.sh): shader-clock stamps of one multiplying wave per block
static __device__ unsigned long long g_x3ws_ts[8192 * 8];
#define X3WS_STAMP(k) do { if (threadIdx.x == 0) { const unsigned bl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (bl < 8192) g_x3ws_ts[bl * 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define X3WS_STAMP(k) do { } while (0)
#endif
// wait until at most `stages` x L of this wave's LDS-DMA instructions are outstanding (stages: wave-uniform, 0..3)
template <int L>
__device__ __forceinline__ void wait_stages(int stages) {
  static_assert(3 * L <= 63, "vmcnt is a 6-bit field");
  if (stages >= 3) wait_vmcnt<3 * L>();
  else if (stages == 2) wait_vmcnt<2 * L>();
  else if (stages == 1) wait_vmcnt<L>();
  else wait_vmcnt<0>();
}

// ---- row-major epilogue through LDS ---------------------------------------------------------------------------
// XW: waves of the block beyond the WGM x WGN multiplying ones that take a share of the row passes (the loader waves of
// conv_x3.hip's gconv_x3ws_kernel: the passes are latency chains - LDS read, split, stores - that one wave per SIMD does not hide)
template <int WGM, int WGN, int WM, int WN, int RING_BYTES, int XW = 0>
struct EpiGeom {
  static constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, NT = (WGM * WGN + XW) * 64;
  static constexpr int NH = (BM * BN * 4 > RING_BYTES) ? 2 : 1;     // column halves (256x256: two rounds of 128 columns)
  static constexpr int CN = BN / NH;        // columns per round
  static constexpr int OCT = CN / 8;        // threads per tile row (8 channels each)
  static constexpr int RPI = NT / OCT;      // rows per pass of the block
  static constexpr int NIT = BM / RPI;      // passes
  static constexpr bool PF_FIRST = NH == 1 && NIT <= 8;    // registers for the operand prefetched ahead of the K loop
  static_assert(BM * CN * 4 <= RING_BYTES && WGN % NH == 0 && NT % OCT == 0 && BM % RPI == 0, "epilogue tile geometry");
};

// element offset (pixel * Ns; add the channel) of tile row m of phase pz, or OOB_OFFSET when the row is past the phase
__device__ __forceinline__ unsigned pipe_row_elem(const GConvArgs& a, int pz, int m) {
  if (m >= a.ph[pz].M) return OOB_OFFSET;
  unsigned opix = (unsigned)m;
  if (!a.linear_out) {
    const int plane = a.ph[pz].ohg * a.ph[pz].owg;
    const int b = fdiv(m, a.ph[pz].d_plane);
    const int rem = m - b * plane;
    const int y = fdiv(rem, a.ph[pz].d_owg);
    const int x = rem - y * a.ph[pz].owg;
    opix = (unsigned)((b * a.OH + y * a.osy + a.ph[pz].ooy) * a.OW + x * a.osx + a.ph[pz].oox);
  }
  return opix * (unsigned)a.Ns;
}

__device__ __forceinline__ void unpack_bf16x8(u32x4 r, f32x4& lo, f32x4& hi) {
  lo = f32x4{__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
             __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
  hi = f32x4{__builtin_bit_cast(float, r.z << 16), __builtin_bit_cast(float, r.z & 0xffff0000u),
             __builtin_bit_cast(float, r.w << 16), __builtin_bit_cast(float, r.w & 0xffff0000u)};
}

// the fused-derivative operand (bf16 storage) of this thread's stores of column half h: NIT 16-byte loads
// rowtab (optional, LDS): element offset (or OOB_OFFSET) of every tile row, for tiles whose rows are NOT consecutive
// positions of the phase grid (the spatial-patch tiles of conv_x3.hip's halo kernel)
template <class G>
__device__ __forceinline__ void pipe_aux_load(const GConvArgs& a, int pz, int m0, int n0, int h, u32x4 (&v)[G::NIT],
                                              const unsigned* rowtab = nullptr) {
  const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux_bytes, 0x00020000);
  const int tid = threadIdx.x, n = n0 + h * G::CN + (tid % G::OCT) * 8;
#pragma unroll
  for (int it = 0; it < G::NIT; ++it) {
    const unsigned e = rowtab ? rowtab[it * G::RPI + tid / G::OCT] : pipe_row_elem(a, pz, m0 + it * G::RPI + tid / G::OCT);
    v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_aux, (e != OOB_OFFSET && n < a.Ns) ? (e + (unsigned)n) * 2u : OOB_OFFSET, 0, PIPE_NT);
  }
}

// Same order of operations as gconv_epilogue (conv_shared.h): pair scale, [column sums of the accumulator], bias,
// activation, fused derivative, residual, [column sums of the stored value], store.  Activations: none / ReLU /
// LeakyReLU only (the launcher refuses the others).  T: the ring, free by now (all DMA landed, all fragment reads done).
// accumulators of one wave -> the fp32 [rows][CN] tile in LDS.  32x32 MFMA blocks (C layout: column = lane & 31, row =
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) or 16x16 blocks (v_mfma_f32_16x16x32: column = lane & 15, row = 4 (lane >> 4) + r;
// the wave tile is [2 WM][2 WN] such blocks)
template <int WM, int WN>
__device__ __forceinline__ void pipe_acc_to_lds(const f32x16 (&acc)[WM][WN], float* T, int CN, int wm, int wnh, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        T[((wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CN + (wnh * WN + j) * 32 + l31] = acc[i][j][r];
}
template <int WM, int WN>
__device__ __forceinline__ void pipe_acc_to_lds(const f32x4 (&acc)[2 * WM][2 * WN], float* T, int CN, int wm, int wnh, int lane) {
  const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int i = 0; i < 2 * WM; ++i)
#pragma unroll
    for (int j = 0; j < 2 * WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        T[((wm * WM * 2 + i) * 16 + 4 * g + r) * CN + (wnh * WN * 2 + j) * 16 + l15] = acc[i][j][r];
}

// X3: out (and res) may be a three-plane tensor (GConvArgs::out16 == 2): the 8 values of a store are split into x = h + m + l
// (exact) and leave as three 16-byte stores out_ps bytes apart; a three-plane residual is read back as h + (m + l) (exact).
// XW / HAS_ACC: see EpiGeom - the extra waves call this with HAS_ACC = false (they hold no accumulators) and run the same barriers
template <int WGM, int WGN, int WM, int WN, int RING_BYTES, bool STATS, bool PF, bool BNM = false, bool X3 = false, class ACC = f32x16[WM][WN],
          int XW = 0, bool HAS_ACC = true>
__device__ __forceinline__ void pipe_epilogue(const GConvArgs& a, ACC& acc, float* T, int pz, unsigned lq,
                                              int m0, int n0, const u32x4* auxpf,         // PF: [NH][NIT] prefetched
                                              const unsigned* rowtab = nullptr,
                                              unsigned slab_off = 0) {                    // split K: fp32 elements into a.out
  using G = EpiGeom<WGM, WGN, WM, WN, RING_BYTES, XW>;
  constexpr int CN = G::CN, OCT = G::OCT, RPI = G::RPI, NIT = G::NIT, NW = WGM * WGN + XW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN, half = lane >> 5, l31 = lane & 31;
  const int oct = tid % OCT, r0 = tid / OCT;
  const int halfM = a.ph[pz].M >> 1;
  float rsc0 = 1.f, rsc1 = 1.f;
  if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
  const float neg_act = a.act == IPRGAN_ACT_NONE ? 1.f : a.act == IPRGAN_ACT_RELU ? 0.f : a.slope;
  const float neg_aux = a.aux_act == IPRGAN_ACT_NONE ? 1.f : a.aux_act == IPRGAN_ACT_RELU ? 0.f : a.aux_slope;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)((float*)a.out + slab_off), 0, a.out_bytes - slab_off * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? a.aux_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? a.out_bytes : 0, 0x00020000);
  constexpr int WGN_H = WGN / G::NH;
#pragma unroll
  for (int h = 0; h < G::NH; ++h) {
    const int n = n0 + h * CN + oct * 8;
    const bool nok = n < a.Ns;                        // Ns % 8 == 0 (launcher)
    u32x4 auxl[NIT];
    if (!PF && a.aux && a.aux16) pipe_aux_load<G>(a, pz, m0, n0, h, auxl, rowtab);      // in flight across the LDS round trip
    if (h > 0) lds_barrier();                         // everybody is done reading the previous half
    if constexpr (HAS_ACC) { if (wn / WGN_H == h) pipe_acc_to_lds<WM, WN>(acc, T, CN, wm, wn % WGN_H, lane); }
    lds_barrier();
    if constexpr (X3 && HAS_ACC && XW > 0) X3WS_STAMP(4);
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { if (n + k < a.N) b0[k] = a.bias[n + k]; if (n + 4 + k < a.N) b1[k] = a.bias[n + 4 + k]; }
    }
    float cs1[8], cs2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cs1[k] = cs2[k] = 0.f;
    // norm-backward mode (GConvArgs::bn_mean; BNM instantiations only: the 32 registers of per-channel constants of this
    // thread's 8 channels spill in the 256x256 and four-phase tiles, which keep accumulators live across column halves)
    constexpr bool bn = STATS && BNM;
    const float bn_neg = a.bn_act == IPRGAN_ACT_NONE ? 1.f : a.bn_act == IPRGAN_ACT_RELU ? 0.f : a.bn_slope;
    float bnI[8], bnM[8], bnG[8], bnT[8];
    if (bn) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool nk = n + k < a.N;
        bnI[k] = nk ? a.bn_invstd[n + k] : 0.f;
        bnM[k] = nk ? a.bn_mean[n + k] : 0.f;
        bnG[k] = nk ? (a.bn_gamma ? a.bn_gamma[n + k] : 1.f) : 0.f;
        bnT[k] = nk ? (a.bn_beta ? a.bn_beta[n + k] : 0.f) : 0.f;
      }
    }
    // Rows of one thread are RPI apart.  When a pass covers whole grid rows (RPI % width == 0) inside one image and the
    // tile is full, the pixel offset advances by a constant per pass: one division chain per tile instead of one per row.
    const Phase& ph = a.ph[pz];
    const bool lin = !a.linear_out && (RPI % ph.owg) == 0 && ((ph.ohg * ph.owg) % G::BM) == 0 && m0 + G::BM <= ph.M;
    const unsigned e_first = pipe_row_elem(a, pz, m0 + r0 < ph.M ? m0 + r0 : 0);
    const unsigned e_step = a.linear_out ? (unsigned)(RPI * a.Ns) : (unsigned)((RPI / (ph.owg > 0 ? ph.owg : 1)) * a.osy * a.OW * a.Ns);
    const bool fast_rows = rowtab ? false : (a.linear_out ? m0 + G::BM <= ph.M : lin);
    const bool full = fast_rows && n0 + h * CN + CN <= a.Ns;         // no row or column of this tile is clipped
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = it * RPI + r0, m = m0 + r;
      const unsigned e0 = rowtab ? rowtab[r] : fast_rows ? e_first + (unsigned)it * e_step : pipe_row_elem(a, pz, m);
      const bool ok = e0 != OOB_OFFSET && nok;
      const unsigned e = e0 + (unsigned)n;
      f32x4 v0 = *(const f32x4*)(T + r * CN + oct * 8), v1 = *(const f32x4*)(T + r * CN + oct * 8 + 4);
      if (a.rs0) { const float rsm = m < halfM ? rsc0 : rsc1; v0 *= rsm; v1 *= rsm; }
      if (STATS && a.stat_mode == 1) {     // rows past M and columns past N hold zeros (zero-filled operands)
#pragma unroll
        for (int k = 0; k < 4; ++k) { cs1[k] += v0[k]; cs2[k] += v0[k] * v0[k]; cs1[4 + k] += v1[k]; cs2[4 + k] += v1[k] * v1[k]; }
      }
      v0 += b0; v1 += b1;
      if (a.act != IPRGAN_ACT_NONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v0[k] = v0[k] > 0.f ? v0[k] : (neg_act == 0.f ? 0.f : v0[k] * neg_act);
          v1[k] = v1[k] > 0.f ? v1[k] : (neg_act == 0.f ? 0.f : v1[k] * neg_act);
        }
      }
      if (a.aux) {
        f32x4 o0, o1;
        if (a.aux16) {
          unpack_bf16x8(PF ? auxpf[h * NIT + it] : auxl[it], o0, o1);
        } else {
          o0 = buf_load4(rs_aux, ok ? e * 4u : OOB_OFFSET);
          o1 = buf_load4(rs_aux, ok ? e * 4u + 16u : OOB_OFFSET);
        }
        if (bn) {               // (the forward's own expression for the mask: bit-identical to the stored activation's sign)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xh0 = (o0[k] - bnM[k]) * bnI[k], xh1 = (o1[k] - bnM[4 + k]) * bnI[4 + k];
            v0[k] *= (xh0 * bnG[k] + bnT[k]) > 0.f ? 1.f : bn_neg;
            v1[k] *= (xh1 * bnG[4 + k] + bnT[4 + k]) > 0.f ? 1.f : bn_neg;
            const float t0 = ok ? v0[k] : 0.f, t1 = ok ? v1[k] : 0.f;
            cs1[k] += t0; cs2[k] += t0 * xh0; cs1[4 + k] += t1; cs2[4 + k] += t1 * xh1;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) { v0[k] *= o0[k] > 0.f ? 1.f : neg_aux; v1[k] *= o1[k] > 0.f ? 1.f : neg_aux; }
        }
      }
      if (a.res) {
        if (X3 && a.out16 == 2) {
          f32x4 h0, h1, m0_, m1_, l0, l1;
          const unsigned eb = ok ? e * 2u : OOB_OFFSET;
          unpack_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rs_res, eb, 0, 0), h0, h1);
          unpack_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? eb + a.out_ps : OOB_OFFSET, 0, 0), m0_, m1_);
          unpack_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? eb + 2u * a.out_ps : OOB_OFFSET, 0, 0), l0, l1);
          v0 += h0 + (m0_ + l0); v1 += h1 + (m1_ + l1);
        } else if (a.out16) {
          f32x4 q0, q1;
          unpack_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? e * 2u : OOB_OFFSET, 0, 0), q0, q1);
          v0 += q0; v1 += q1;
        } else {
          v0 += buf_load4(rs_res, ok ? e * 4u : OOB_OFFSET);
          v1 += buf_load4(rs_res, ok ? e * 4u + 16u : OOB_OFFSET);
        }
      }
      if (STATS && a.stat_mode == 2) {     // column sums of what is stored (the bias gradient of the layer below): sums only
        if (full) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { cs1[k] += v0[k]; cs1[4 + k] += v1[k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) { cs1[k] += ok ? v0[k] : 0.f; cs1[4 + k] += ok ? v1[k] : 0.f; }
        }
      }
      if (X3 && a.out16 == 2) {
        bf16x4 t0[3], t1[3];
        split3_bf16(v0, t0);
        split3_bf16(v1, t1);
        const unsigned eb = ok ? e * 2u : OOB_OFFSET;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u32x2 w0 = __builtin_bit_cast(u32x2, t0[p]), w1 = __builtin_bit_cast(u32x2, t1[p]);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0.x, w0.y, w1.x, w1.y}, rs_out, ok ? eb + (unsigned)p * a.out_ps : OOB_OFFSET, 0, PIPE_NT);
        }
      } else if (a.out16) {
        const bf16x4 p0 = to_bf16x4(v0), p1 = to_bf16x4(v1);
        const u32x2 w0 = __builtin_bit_cast(u32x2, p0), w1 = __builtin_bit_cast(u32x2, p1);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0.x, w0.y, w1.x, w1.y}, rs_out, ok ? e * 2u : OOB_OFFSET, 0, PIPE_NT);
      } else {
        buf_store4(rs_out, ok ? e * 4u : OOB_OFFSET, v0);
        buf_store4(rs_out, ok ? e * 4u + 16u : OOB_OFFSET, v1);
      }
    }
    if (STATS) {
      // this thread: 8 channels x its NIT rows; the lanes of a wave with the same channel octet (lane % OCT) are combined
      // by butterflies, the NW waves through LDS in wave order: fixed order, deterministic
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int o = OCT; o < 64; o <<= 1) cs1[k] += __shfl_xor(cs1[k], o, 64);
      if (a.stat_mode != 2) {                        // second sums: forward statistics (1) and the norm backward (3)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int o = OCT; o < 64; o <<= 1) cs2[k] += __shfl_xor(cs2[k], o, 64);
      }
      lds_barrier();                                 // the tile has been consumed
      if (lane < OCT) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { T[(wave * CN + lane * 8 + k) * 2] = cs1[k]; T[(wave * CN + lane * 8 + k) * 2 + 1] = cs2[k]; }
      }
      lds_barrier();
      for (int c = tid; c < CN; c += G::NT) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { s1 += T[(w * CN + c) * 2]; s2 += T[(w * CN + c) * 2 + 1]; }
        const int nn = n0 + h * CN + c;
        if (nn < a.Ns) { a.stat_part[((size_t)lq * 2) * a.Ns + nn] = s1; a.stat_part[((size_t)lq * 2 + 1) * a.Ns + nn] = s2; }
      }
    }
  }
}

}  // namespace iprgan
