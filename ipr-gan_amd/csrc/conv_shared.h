// Shared between the conv translation units (conv_igemm.hip: fp32 / bf16 register-staged tiles, backward-weight;
// conv_pipe.hip: bf16 tiles staged by LDS-DMA through a multi-stage ring): vector types, the gather-GEMM argument
// block, buffer-descriptor helpers, the XCD-aware tile order and the common epilogue of the gather-GEMM kernels.
#pragma once
#include "common.h"
#include <hip/hip_ext.h>

namespace iprgan {

// ---- optional per-kernel timing (bench.py roofline), implemented in conv_igemm.hip -----------------------------
// prof_events(): when profiling is on, returns true and hands out the two events that ride on the dispatch packet.
bool prof_events(int slot, double flops, hipEvent_t* start, hipEvent_t* stop);
template <class Kern, class... Args>
static void prof_launch(Kern kern, dim3 grid, dim3 block, size_t smem, hipStream_t st, int slot, double flops,
                        const Args&... a) {
  hipEvent_t e0, e1;
  if (prof_events(slot, flops, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)smem, st, e0, e1, 0, a...);
  else hipLaunchKernelGGL(kern, grid, block, (unsigned)smem, st, a...);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 f) {      // round-to-nearest-even (v_cvt_pk_bf16_f32)
  const bf16x4 v = {(__bf16)f.x, (__bf16)f.y, (__bf16)f.z, (__bf16)f.w};
  return v;
}
// x = h + m + l, three bf16 terms (nearest-even each; the two subtractions are exact in fp32): t[0] = h, t[1] = m, t[2] = l
// (written on PAIRS: one v_cvt_pk_bf16_f32 per two elements and plane, widening = one shift / one mask per element;
// element-wise conversions cost the compiler a convert per element plus the re-packing: 60 instead of 44 instructions per 8)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {          // {bf16(a), bf16(b)}, nearest-even
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& hp, unsigned& mp, unsigned& lp) {
  hp = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, hp << 16), r1 = x1 - __builtin_bit_cast(float, hp & 0xffff0000u);
  mp = cvt_pk_bf16(r0, r1);
  lp = cvt_pk_bf16(r0 - __builtin_bit_cast(float, mp << 16), r1 - __builtin_bit_cast(float, mp & 0xffff0000u));
}
__device__ __forceinline__ void split3_bf16(f32x4 f, bf16x4 (&t)[3]) {
  typedef unsigned int u32pair __attribute__((ext_vector_type(2)));
  unsigned w0[3], w1[3];
  split3_pair(f.x, f.y, w0[0], w0[1], w0[2]);
  split3_pair(f.z, f.w, w1[0], w1[1], w1[2]);
#pragma unroll
  for (int p = 0; p < 3; ++p) t[p] = __builtin_bit_cast(bf16x4, u32pair{w0[p], w1[p]});
}
// 4 consecutive bf16 elements (element index idx of a tensor whose storage type is bf16) -> 4 floats
__device__ __forceinline__ f32x4 ld_bf16x4(const float* base, size_t idx) {
  const bf16x4 h = *(const bf16x4*)((const __bf16*)base + idx);
  const f32x4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
  return v;
}

struct Phase {
  int th, tw, ntap;
  int dy0, dx0, dys, dxs;
  int wbase, wsy, wsx;
  int ooy, oox, ohg, owg;
  int M, steps;
  FastDiv d_owg, d_plane, d_tw;
};

struct GConvArgs {
  const float* in;
  const float* wt;
  const float* bias;
  float* out;
  const float* aux;
  int B, IH, IW, Cs, c4n;
  FastDiv d_c4n;
  int Kp;
  int OH, OW, Ns, N;
  int isy, isx, osy, osx;
  int pad_mode, act;
  float slope;
  int aux_act;
  float aux_slope;
  unsigned in_bytes, wt_bytes, out_bytes, aux_bytes;   // buffer descriptor ranges (out_bytes also bounds the residual)
  int linear_out;
  int planar_M;        // > 0: store output channel n at plane n>>2 (tap-planar T of the small-N path)
  float* ws;           // host-side only: workspace for the small-N path (may be null)
  size_t ws_floats;
  int nphase;
  const float* res;    // added to the stored value (same layout as out): the gradient arriving over a skip connection
  const float* rs0;    // paired pass (two half-batches through one launch, each with its own spectral-norm sigma):
  const float* rs1;    //   rows of the first / second half of every phase are divided by *rs0 / *rs1 before the bias
  int in16;            // 1: both operands (activation and prepared weight) are bf16 in HBM -> IN16 kernels
                       // 2: both are THREE bf16 planes (x = h + m + l exactly, plane-major: plane p of the activation at byte
                       //    in_ps * p, of the prepared weight at wt_ps * p) -> the three-plane kernels (conv_x3.hip, gconv_kernel<.., IN3P>)
  int out16, aux16;    // storage of out (and res) / aux: 0 fp32, 1 bf16 (element offsets stay the same), out16 == 2: three
                       // bf16 planes out_ps bytes apart (aux16 == 1 then reads the h plane of such a tensor: same sign)
  unsigned in_ps, wt_ps, out_ps;     // plane strides in bytes (storage kind 2)
  float* stat_part;    // STATS kernels: per-tile column sums [tile rows][2][Ns] (see gconv_kernel)
  int stat_mode;
  int ksplit;          // > 1: blockIdx.z splits the K loop (single-phase geometries); partial tiles go to slabs of M*Ns floats
  int korder;          // LDS-DMA ring tiles: K-loop form bits, see g_pipe_korder (conv_pipe.hip)
  // Norm-backward mode of a backward-data pass (stat_mode 3; iprgan_conv_bwd_data_bn): the result is the gradient w.r.t.
  // the OUTPUT of a BatchNorm (+ReLU / LeakyReLU) whose INPUT x is `aux`.  Per element: xh = (x - mean) * invstd,
  // v = xh * gamma + beta, dz = acc * act'(v) is what is stored, and the STATS rows hold sum dz and sum dz * xh: the
  // two reductions of the norm backward, taken here instead of in a pass over (x, dy).
  const float* bn_mean;
  const float* bn_invstd;
  const float* bn_gamma;
  const float* bn_beta;
  int bn_act;
  float bn_slope;
  int wmod, wk1;       // > 0: operand row r lives at (r % wmod) * Kp + (r / wmod) * wk1 floats (full-map conv backward-data)
  Phase ph[4];
  double flops;   // algorithmic 2*MAC of this launch (host-side bookkeeping only)
};

#define ROW_INVALID (-(1 << 28))

__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define OOB_OFFSET 0x80000000u     // buffer voffset beyond any tensor (< 2 GiB): the load returns 0

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, unsigned voff, f32x4 v) {      // out-of-range: dropped
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, 0, 0);
}
__device__ __forceinline__ void buf_store_bf16x4(__amdgpu_buffer_rsrc_t rs, unsigned voff, bf16x4 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, 0, 0);
}
// 4 consecutive bf16 elements (8 bytes) widened to fp32; out-of-range offsets give zeros like buf_load4
__device__ __forceinline__ f32x4 buf_load4_bf16(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
  const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
  const f32x4 v = {__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                   __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
  return v;
}

// 4x4 transpose across the four lanes of a quad: afterwards register k of lane p holds what register p of
// lane k held.  Two butterfly stages (lane^1, lane^2) on the DPP quad_perm network, no LDS.
__device__ __forceinline__ float dpp_quad(float v, const int ctrl_is_xor2) {
  const int x = __builtin_bit_cast(int, v);
  const int r = ctrl_is_xor2 ? __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)    // quad_perm(2,3,0,1)
                             : __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);   // quad_perm(1,0,3,2)
  return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ void quad_transpose(float& a0, float& a1, float& a2, float& a3, int p) {
  const bool o1 = p & 1, o2 = p & 2;
  float t0 = dpp_quad(a1, 0), t1 = dpp_quad(a0, 0);
  float b0 = o1 ? t0 : a0, b1 = o1 ? a1 : t1;
  t0 = dpp_quad(a3, 0); t1 = dpp_quad(a2, 0);
  float b2 = o1 ? t0 : a2, b3 = o1 ? a3 : t1;
  t0 = dpp_quad(b2, 1); t1 = dpp_quad(b0, 1);
  a0 = o2 ? t0 : b0; a2 = o2 ? b2 : t1;
  t0 = dpp_quad(b3, 1); t1 = dpp_quad(b1, 1);
  a1 = o2 ? t0 : b1; a3 = o2 ? b3 : t1;
}

// The dispatcher hands workgroup i (x fastest) to XCD i % 8, each with its own 4 MB L2.  Blocks remap their
// id so that every XCD owns one contiguous run of logical tiles: neighbours in that order (which share
// operand rows) then hit the same L2 instead of fetching the rows once per XCD.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned total) {
  constexpr unsigned X = 8;
  const unsigned per = total / X, rem = total % X;      // XCD x owns per + (x < rem) tiles
  const unsigned x = id % X, j = id / X;
  return x * per + (x < rem ? x : rem) + j;
}

// Epilogue of a gather-GEMM tile: acc[i][j] = the 32x32 accumulators of this wave ((wm*WM+i), (wn*WN+j)) of the block
// tile at (m0, n0) of phase pz (zi = blockIdx.z after the remap: phase or K split; lq = logical tile row, the row of
// the STATS partials).  lds: the block's staging memory (free by now; >= WGM*BN*2 floats), used by STATS.
template <int WGM, int WGN, int WM, int WN, bool STATS>
__device__ __forceinline__ void gconv_epilogue(const GConvArgs& a, f32x16 (&acc)[WM][WN], f32x4* lds, int pz, int zi,
                                               unsigned lq, int m0, int n0) {
  constexpr int BN = WGN * WN * 32, NT = WGM * WGN * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int pM = a.ph[pz].M;
  const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
  const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg;
  // epilogue.  C/D layout of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  A 4x4
  // transpose inside each lane quad (2 DPP butterfly stages) turns registers 4g..4g+3 into ONE row with four
  // consecutive channels per lane, so the tile leaves as 16-byte stores (4x fewer store instructions: the
  // narrow-store epilogue was issue-bound on the layers with large outputs).
  const int half = lane >> 5, l31 = lane & 31;
  const int ooy = a.ph[pz].ooy, oox = a.ph[pz].oox;
  const int qp = lane & 3, qcol = l31 & ~3;
  float rsc0 = 1.f, rsc1 = 1.f;
  if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
  const int halfM = pM >> 1;
  float cs1[WN][4], cs2[WN][4];
  if (STATS) {
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) cs1[j][k] = cs2[j][k] = 0.f;
  }
  // The epilogue runs in two passes per 32-row tile so that its memory operations overlap: pass 1 turns the accumulators
  // into final pre-derivative values (transpose, pair scale, bias, activation) and computes one byte offset per 4-channel
  // store, with invalid rows / columns mapped to an out-of-range offset; pass 2 issues ALL loads of the fused derivative
  // and residual operands of the tile as buffer loads (out-of-range lanes read zeros, no exec-mask branches, so the
  // compiler keeps them in flight together instead of load - wait - store per store), then multiplies, adds and stores
  // through the buffer descriptor (out-of-range lanes are dropped).  The bias of this lane's channel quads is loaded
  // once; the common activations avoid the general switch, whose inlined tanh / sigmoid made each store several hundred
  // instructions of code.  On layers with short reductions (K = 576: 18 steps) the epilogue was a fifth of a wave's life.
  const bool aux_simple = a.aux_act == IPRGAN_ACT_NONE || a.aux_act == IPRGAN_ACT_RELU || a.aux_act == IPRGAN_ACT_LRELU;
  const float neg_aux = a.aux_act == IPRGAN_ACT_NONE ? 1.f : a.aux_act == IPRGAN_ACT_RELU ? 0.f : a.aux_slope;
  const unsigned esz_out = a.out16 ? 2u : 4u;
  const unsigned slab_off = a.ksplit > 1 ? (unsigned)zi * (unsigned)pM * (unsigned)a.Ns : 0u;     // elements (< 2^31: checked)
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? a.aux_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? a.out_bytes : 0, 0x00020000);
  f32x4 bias4[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + (wn * WN + j) * 32 + qcol;
    if (a.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (n + k < a.N) bias4[j][k] = a.bias[n + k];
    }
  }
  // norm-backward mode: xh = (o - bnB) * bnA, v = xh * bnG + bnT per channel of this lane's quads - the forward's own
  // expression ((x - mean) * invstd * gamma + beta, bn_apply_kernel), so that the mask agrees with the stored output bit for bit
  const bool bn = STATS && a.bn_mean != nullptr;
  const float bn_neg = a.bn_act == IPRGAN_ACT_NONE ? 1.f : a.bn_act == IPRGAN_ACT_RELU ? 0.f : a.bn_slope;
  f32x4 bnA[WN], bnB[WN], bnG[WN], bnT[WN];
  if (bn) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + (wn * WN + j) * 32 + qcol;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool nk = n + k < a.N;
        const float is = nk ? a.bn_invstd[n + k] : 0.f;
        bnA[j][k] = is;
        bnB[j][k] = nk ? a.bn_mean[n + k] : 0.f;
        bnG[j][k] = nk ? (a.bn_gamma ? a.bn_gamma[n + k] : 1.f) : 0.f;
        bnT[j][k] = nk ? (a.bn_beta ? a.bn_beta[n + k] : 0.f) : 0.f;
      }
    }
  }
  constexpr int GC = WN >= 2 ? 2 : 4;     // row groups per pass (4 stores in flight per lane: registers stay at the K loop's level)
#pragma unroll
  for (int ig = 0; ig < WM * (4 / GC); ++ig) {
    const int i = ig / (4 / GC), g0 = (ig % (4 / GC)) * GC;
    f32x4 val[GC][WN];
    unsigned eoff[GC][WN];            // element index of the store, or OOB_OFFSET
    // ---- pass 1: values and offsets
#pragma unroll
    for (int gg = 0; gg < GC; ++gg) {
      const int g = g0 + gg;
      const int m = m0 + (wm * WM + i) * 32 + 8 * g + 4 * half + qp;
      const bool mok = m < pM;
      unsigned opix = (unsigned)m;
      if (!a.linear_out) {
        const int mm = mok ? m : 0;
        const int b = fdiv(mm, d_plane);
        const int rem = mm - b * plane;
        const int y = fdiv(rem, d_owg);
        const int x = rem - y * p_owg;
        opix = (unsigned)((b * a.OH + y * a.osy + ooy) * a.OW + x * a.osx + oox);
      }
      const float rsm = a.rs0 ? (m < halfM ? rsc0 : rsc1) : 1.f;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        float c0 = acc[i][j][4 * g], c1 = acc[i][j][4 * g + 1], c2 = acc[i][j][4 * g + 2], c3 = acc[i][j][4 * g + 3];
        quad_transpose(c0, c1, c2, c3, qp);
        const int n = n0 + (wn * WN + j) * 32 + qcol;
        const bool ok = mok && n < a.Ns;
        f32x4 v = {c0, c1, c2, c3};
        if (a.rs0) v *= rsm;
        if (STATS && a.stat_mode == 1) {        // rows past M and columns past N accumulate zeros (zero-filled operands)
#pragma unroll
          for (int k = 0; k < 4; ++k) { cs1[j][k] += v[k]; cs2[j][k] += v[k] * v[k]; }
        }
        v += bias4[j];
        if (a.act == IPRGAN_ACT_LRELU) {        // uniform branches
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * a.slope;
        } else if (a.act == IPRGAN_ACT_RELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        } else if (a.act != IPRGAN_ACT_NONE) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], a.act, a.slope);
        }
        val[gg][j] = v;
        const unsigned e = (a.planar_M ? ((unsigned)(n >> 2) * (unsigned)a.planar_M + opix) * 4u : opix * (unsigned)a.Ns + (unsigned)n) + slab_off;
        eoff[gg][j] = ok ? e : OOB_OFFSET;
      }
    }
    // ---- pass 2: fused derivative, residual (all loads of the tile in flight together), statistics, stores
    if (a.aux) {
      f32x4 o[GC][WN];
#pragma unroll
      for (int g = 0; g < GC; ++g)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          o[g][j] = a.aux16 ? buf_load4_bf16(rs_aux, eoff[g][j] == OOB_OFFSET ? OOB_OFFSET : eoff[g][j] * 2u)
                            : buf_load4(rs_aux, eoff[g][j] == OOB_OFFSET ? OOB_OFFSET : eoff[g][j] * 4u);
#pragma unroll
      for (int g = 0; g < GC; ++g)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (bn) {
            const bool ok = eoff[g][j] != OOB_OFFSET;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float xh = (o[g][j][k] - bnB[j][k]) * bnA[j][k];
              const float dz = val[g][j][k] * ((xh * bnG[j][k] + bnT[j][k]) > 0.f ? 1.f : bn_neg);
              val[g][j][k] = dz;
              const float t = ok ? dz : 0.f;
              cs1[j][k] += t; cs2[j][k] += t * xh;
            }
          } else if (aux_simple) {
#pragma unroll
            for (int k = 0; k < 4; ++k) val[g][j][k] *= o[g][j][k] > 0.f ? 1.f : neg_aux;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) val[g][j][k] *= act_grad_from_out(o[g][j][k], a.aux_act, a.aux_slope);
          }
        }
    }
    if (a.res) {
      f32x4 r[GC][WN];
#pragma unroll
      for (int g = 0; g < GC; ++g)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          if (a.out16 == 2) {          // three-plane residual: h + (m + l), exact
            const unsigned rb = eoff[g][j] == OOB_OFFSET ? OOB_OFFSET : eoff[g][j] * 2u;
            r[g][j] = buf_load4_bf16(rs_res, rb) + (buf_load4_bf16(rs_res, rb == OOB_OFFSET ? rb : rb + a.out_ps) +
                                                    buf_load4_bf16(rs_res, rb == OOB_OFFSET ? rb : rb + 2u * a.out_ps));
          } else
          r[g][j] = a.out16 ? buf_load4_bf16(rs_res, eoff[g][j] == OOB_OFFSET ? OOB_OFFSET : eoff[g][j] * 2u)
                            : buf_load4(rs_res, eoff[g][j] == OOB_OFFSET ? OOB_OFFSET : eoff[g][j] * 4u);
#pragma unroll
      for (int g = 0; g < GC; ++g)
#pragma unroll
        for (int j = 0; j < WN; ++j) val[g][j] += r[g][j];
    }
#pragma unroll
    for (int g = 0; g < GC; ++g)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bool ok = eoff[g][j] != OOB_OFFSET;
        if (STATS && a.stat_mode == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { const float t = ok ? val[g][j][k] : 0.f; cs1[j][k] += t; cs2[j][k] += t * t; }
        }
        const unsigned boff = ok ? eoff[g][j] * esz_out : OOB_OFFSET;
        if (a.out16 == 2) {                // this tensor lives as three bf16 planes: split once, here
          bf16x4 t3[3];
          split3_bf16(val[g][j], t3);
#pragma unroll
          for (int p = 0; p < 3; ++p) buf_store_bf16x4(rs_out, ok ? boff + (unsigned)p * a.out_ps : OOB_OFFSET, t3[p]);
        } else if (a.out16) buf_store_bf16x4(rs_out, boff, to_bf16x4(val[g][j]));     // this tensor lives as bf16
        else buf_store4(rs_out, boff, val[g][j]);
      }
  }
  if (STATS) {
    // rows of one column quad live in the 8 lanes that differ in lane bits 0, 1 (row inside the transposed quad) and 5
    // (row group); then the WGM waves that share the columns are combined through LDS, in wave order
    float* red = (float*)lds;                    // the staging buffers are free: every wave is past its last compute()
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s1 = cs1[j][k], s2 = cs2[j][k];
        s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
        s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (qp == 0 && half == 0) {
          const int c = (wn * WN + j) * 32 + qcol + k;
          red[(wm * BN + c) * 2] = s1;
          red[(wm * BN + c) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) { s1 += red[(w * BN + c) * 2]; s2 += red[(w * BN + c) * 2 + 1]; }
      if (n0 + c < a.Ns) {
        a.stat_part[((size_t)lq * 2) * a.Ns + n0 + c] = s1;
        a.stat_part[((size_t)lq * 2 + 1) * a.Ns + n0 + c] = s2;
      }
    }
  }
}

}  // namespace iprgan
