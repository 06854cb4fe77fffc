// Linear(K -> C*HW) whose output is consumed as an NHWC map [B, HW, C] - the first layer of the reference's generators
// (networks/conv_generator.py:26-30: fc + ReLU + view(B, C, mg, mg)) and of the VAE decoder.
//
// Through the convolution family this layer cost the DCGAN-64 step 54 us forward (row permutation of the 16 MB weight into
// (hw, c) order, operand prep, split of z into planes, a 1x1 convolution) and 107 us backward (two joins of three-plane
// tensors, a stand-alone activation backward, the split-M GEMM, a 16 MB slab copy, a bias column sum, two un-permutes) for
// 1 GFLOP each way.  Here it is ONE launch each way, on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32) in every math mode:
//   forward   y[b][hw*C + c] = act(sum_k x[b][k] * W[c*HW + hw][k] + bias[c*HW + hw])        W, bias in PyTorch layout,
//   backward  dW[c*HW + hw][k] (+)= sum_b dz[b][hw*C + c] * x[b][k],  db[c*HW + hw] (+)= sum_b dz[b][hw*C + c],
//             dz = dy * act'(y)                                                               y, dy in any storage kind.
// A block owns 64 consecutive NHWC columns n' (one hw, 64 consecutive c: C % 64 == 0) and ALL rows b, 128 at a time:
// W rows / dW rows of the block are 64 rows of K floats, HW rows apart.  Bandwidth-bound: W (or dW) once, y / dy once.
// Summation orders are fixed (b ascending inside the MFMA's k walk, bias sums b ascending): deterministic.
#include "conv_shared.h"

namespace iprgan {

constexpr int FC_BN = 64, FC_BB = 128, FC_KMAX = 128;

template <int KIND>
__device__ __forceinline__ void fc_st8(void* base, size_t idx, size_t ps, f32x4 a, f32x4 b) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  if (KIND == 0) {
    *(f32x4*)((float*)base + idx) = a;
    *(f32x4*)((float*)base + idx + 4) = b;
  } else if (KIND == 1) {
    const u32x4 v = {cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(b.x, b.y), cvt_pk_bf16(b.z, b.w)};
    *(u32x4*)((__bf16*)base + idx) = v;
  } else {
    unsigned p0[3], p1[3], p2[3], p3[3];
    split3_pair(a.x, a.y, p0[0], p0[1], p0[2]);
    split3_pair(a.z, a.w, p1[0], p1[1], p1[2]);
    split3_pair(b.x, b.y, p2[0], p2[1], p2[2]);
    split3_pair(b.z, b.w, p3[0], p3[1], p3[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4 v = {p0[p], p1[p], p2[p], p3[p]};
      *(u32x4*)((__bf16*)base + idx + (size_t)p * ps) = v;
    }
  }
}
__device__ __forceinline__ void fc_widen8(const bf16x8 h, f32x4& a, f32x4& b) {
  a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  b = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
}
template <int KIND>
__device__ __forceinline__ void fc_ld8(const void* base, size_t idx, size_t ps, f32x4& a, f32x4& b) {
  if (KIND == 0) {
    a = *(const f32x4*)((const float*)base + idx);
    b = *(const f32x4*)((const float*)base + idx + 4);
  } else if (KIND == 1) {
    fc_widen8(*(const bf16x8*)((const __bf16*)base + idx), a, b);
  } else {                      // x = h + (m + l), exactly (conv_x3.hip: join3)
    f32x4 ha, hb, ma, mb, la, lb;
    fc_widen8(*(const bf16x8*)((const __bf16*)base + idx), ha, hb);
    fc_widen8(*(const bf16x8*)((const __bf16*)base + idx + ps), ma, mb);
    fc_widen8(*(const bf16x8*)((const __bf16*)base + idx + 2 * ps), la, lb);
    a = ha + (ma + la);
    b = hb + (mb + lb);
  }
}

struct FcArgs {
  const float* x;       // [B][K] fp32
  const float* w;       // [C*HW][K] fp32, PyTorch row order c*HW + hw
  const float* bias;    // [C*HW] or null
  void* y;              // [B][HW*C] storage kind of the instantiation (forward: written; backward: read)
  const void* dy;       // backward only
  float* dw;            // backward only: [C*HW][K]
  float* db;            // backward only: [C*HW] or null
  int B, K, C, HW, act;
  float slope, beta;
  size_t y_ps, dy_ps;   // plane strides (elements) of three-plane tensors
};

// LDS rows of the W tile are read by ds_read_b128 (4 consecutive k per lane): pitch K + 4 floats -> the 16 lanes of a read
// phase land on 16 distinct 16-byte bank groups (pitch mod 64 = 4).  The x operand never touches LDS: a lane's share of it
// (row b = its MFMA row, the k values of its lane half: K / 2 floats) is loaded straight into registers - x is 64 KB, every
// block reads all of it, and staging it cost 67 KB of LDS per block: one block per CU, load / multiply / store phases in
// series (39 us for this 1 GFLOP layer); with the W tile and the output tile alone two blocks share a CU.
__device__ __forceinline__ int fc_pitch(int K) { return K + 4; }

template <int KIND>
__global__ __launch_bounds__(256, 2) void fc_nhwc_fwd_kernel(const FcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fcl[];
  const int K = a.K, P = fc_pitch(K), N = a.C * a.HW;
  constexpr int OP = FC_BN + 4;
  float* Wl = fcl;                          // [64][P]
  float* Ol = Wl + FC_BN * P;               // output tile [128][OP]
  float* Bl = Ol + FC_BB * OP;              // [64] bias of the block's columns
  const int n0 = blockIdx.x * FC_BN;        // first NHWC column of the block
  const int hw = n0 / a.C, c0 = n0 - hw * a.C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int k4 = K >> 2, nk8 = K >> 3;      // 16-byte chunks per row; 8-deep k steps
  for (int i = tid; i < FC_BN * k4; i += 256) {
    const int r = i / k4, q = i - r * k4;
    *(f32x4*)(Wl + r * P + 4 * q) = *(const f32x4*)(a.w + ((size_t)(c0 + r) * a.HW + hw) * K + 4 * q);
  }
  if (tid < FC_BN) Bl[tid] = a.bias ? a.bias[(size_t)(c0 + tid) * a.HW + hw] : 0.f;
  for (int b0 = 0; b0 < a.B; b0 += FC_BB) {
    // this lane's x fragments: row b0 + 32 wave + lane % 32, k = 8 s + 4 half .. + 3 (lane half h supplies k + 4 h + j to MFMA j)
    f32x4 af[FC_KMAX / 8];
    const int brow = b0 + wave * 32 + l31;
#pragma unroll
    for (int s_ = 0; s_ < FC_KMAX / 8; ++s_) {
      af[s_] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (s_ < nk8 && brow < a.B) af[s_] = *(const f32x4*)(a.x + (size_t)brow * K + 8 * s_ + 4 * half);
    }
    __syncthreads();                        // W tile staged (first trip) / previous chunk's output tile stored
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* wb0 = Wl + l31 * P + 4 * half;
    const float* wb1 = Wl + (32 + l31) * P + 4 * half;
#pragma unroll
    for (int s_ = 0; s_ < FC_KMAX / 8; ++s_) {
      if (s_ < nk8) {
        const f32x4 b0f = *(const f32x4*)(wb0 + 8 * s_), b1f = *(const f32x4*)(wb1 + 8 * s_);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].x, b0f.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].x, b1f.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].y, b0f.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].y, b1f.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].z, b0f.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].z, b1f.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].w, b0f.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_].w, b1f.w, acc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)          // C layout: row = (r & 3) + 8 (r >> 2) + 4 half, column = lane % 32
        Ol[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * OP + j * 32 + l31] = acc[j][r];
    __syncthreads();
    // a thread owns 8 consecutive columns of one row: bias, activation, 16-byte stores (8 lanes = one row's 64 columns)
    const int oc = (tid & 7) * 8;
#pragma unroll
    for (int p = 0; p < FC_BB / 32; ++p) {
      const int r = (tid >> 3) + 32 * p;
      if (b0 + r >= a.B) continue;
      f32x4 v0 = *(const f32x4*)(Ol + r * OP + oc), v1 = *(const f32x4*)(Ol + r * OP + oc + 4);
      v0 += *(const f32x4*)(Bl + oc);
      v1 += *(const f32x4*)(Bl + oc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = act_apply(v0[e], a.act, a.slope); v1[e] = act_apply(v1[e], a.act, a.slope); }
      fc_st8<KIND>(a.y, (size_t)(b0 + r) * N + n0 + oc, a.y_ps, v0, v1);
    }
  }
}

// dz is read from LDS one float per lane and MFMA (the MFMA's k index = a batch row: lane half h supplies row b + 4 h): the
// pitch puts the two lane halves - 4 rows apart - 32 banks apart (4 * pitch mod 64 = 32).  x stays in registers: a lane's
// column k = 32 wave + lane % 32 at the 64 batch rows of its half (64 KB of x would otherwise make it one block per CU).
constexpr int FC_DZP = FC_BN + 8, FC_BW = 64;      // batch rows per chunk of the backward kernel (x: FC_BW / 2 registers per lane)

template <int KIND>
__global__ __launch_bounds__(256, 2) void fc_nhwc_bwd_kernel(const FcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fcl[];
  const int K = a.K, N = a.C * a.HW;
  float* Dz = fcl;                          // [64 b][FC_DZP]: dz of the block's 64 columns
  const int n0 = blockIdx.x * FC_BN;
  const int hw = n0 / a.C, c0 = n0 - hw * a.C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int nkb = K >> 5;                   // 32-wide k blocks of the dW tile: wave w owns block w (K <= 128)
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;
  for (int b0 = 0; b0 < a.B; b0 += FC_BW) {
    float xv[FC_BW / 2];                    // x[b0 + 8 (i / 4) + i % 4 + 4 half][32 wave + lane % 32]
    if (wave < nkb) {
#pragma unroll
      for (int i = 0; i < FC_BW / 2; ++i) {
        const int b = b0 + 8 * (i >> 2) + (i & 3) + 4 * half;
        xv[i] = b < a.B ? a.x[(size_t)b * K + wave * 32 + l31] : 0.f;
      }
    }
    __syncthreads();                        // (previous chunk's dz has been consumed)
    const int oc = (tid & 7) * 8;
#pragma unroll
    for (int p = 0; p < FC_BW / 32; ++p) {
      const int r = (tid >> 3) + 32 * p;
      f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
      if (b0 + r < a.B) {
        f32x4 y0, y1;
        const size_t idx = (size_t)(b0 + r) * N + n0 + oc;
        fc_ld8<KIND>(a.dy, idx, a.dy_ps, g0, g1);
        if (a.act != IPRGAN_ACT_NONE) {
          fc_ld8<KIND>(a.y, idx, a.y_ps, y0, y1);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            g0[e] *= act_grad_from_out(y0[e], a.act, a.slope);
            g1[e] *= act_grad_from_out(y1[e], a.act, a.slope);
          }
        }
      }
      *(f32x4*)(Dz + r * FC_DZP + oc) = g0;
      *(f32x4*)(Dz + r * FC_DZP + oc + 4) = g1;
    }
    __syncthreads();
    if (wave < nkb) {
      // dW[n'][k] += sum_b dz[b][n'] x[b][k]: MFMA rows = n' (two 32-blocks), columns = k (this wave's 32-block), k walk = b
      const float* ap = Dz + 4 * half * FC_DZP + l31;
#pragma unroll
      for (int i = 0; i < FC_BW / 2; ++i) { // MFMA i: batch rows 8 (i / 4) + i % 4 (lanes 0-31) and + 4 (lanes 32-63)
        const int b = 8 * (i >> 2) + (i & 3);
        const float d0 = ap[b * FC_DZP], d1 = ap[b * FC_DZP + 32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, xv[i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, xv[i], acc[1], 0, 0, 0);
      }
    }
    if (a.db && tid < FC_BN) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;         // (four chains: the 128 LDS reads overlap)
      for (int b = 0; b < FC_BW; b += 4) {
        s0 += Dz[b * FC_DZP + tid]; s1 += Dz[(b + 1) * FC_DZP + tid];
        s2 += Dz[(b + 2) * FC_DZP + tid]; s3 += Dz[(b + 3) * FC_DZP + tid];
      }
      bsum += (s0 + s1) + (s2 + s3);
    }
  }
  if (wave < nkb) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nl = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;       // column of the block = row of dW
        float* dst = a.dw + ((size_t)(c0 + nl) * a.HW + hw) * K + wave * 32 + l31;
        *dst = a.beta != 0.f ? a.beta * *dst + acc[j][r] : acc[j][r];
      }
  }
  if (a.db && tid < FC_BN) {
    float* dst = a.db + (size_t)(c0 + tid) * a.HW + hw;
    *dst = a.beta != 0.f ? a.beta * *dst + bsum : bsum;
  }
}

static bool fc_shape_ok(int B, int K, int C, int HW) {
  return B > 0 && K >= 32 && K <= FC_KMAX && (K % 32) == 0 && C > 0 && (C % FC_BN) == 0 && HW > 0 &&
         (long long)C * HW < (1ll << 30);
}

}  // namespace iprgan

using namespace iprgan;

extern "C" {

int iprgan_fc_nhwc_ok(int B, int K, int C, int HW) { return fc_shape_ok(B, K, C, HW) ? 1 : 0; }

int iprgan_fc_nhwc_fwd(const float* x, const float* w, const float* bias, void* y, int B, int K, int C, int HW, int act,
                       float slope, int y_kind, size_t y_pstride, void* stream) {
  IPR_CHECK(fc_shape_ok(B, K, C, HW), "fc_nhwc_fwd: unsupported shape B %d K %d C %d HW %d (iprgan_fc_nhwc_ok)", B, K, C, HW);
  IPR_CHECK(x && w && y, "fc_nhwc_fwd: null tensor");
  IPR_CHECK(y_kind >= 0 && y_kind <= 2, "fc_nhwc_fwd: storage kind %d", y_kind);
  FcArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.B = B; a.K = K; a.C = C; a.HW = HW; a.act = act; a.slope = slope;
  a.y_ps = y_pstride ? y_pstride : (size_t)B * C * HW;
  // [W tile 64 x (K + 4)] [output tile 128 x 68] [bias 64]
  const size_t smem = ((size_t)FC_BN * (K + 4) + (size_t)FC_BB * (FC_BN + 4) + FC_BN) * sizeof(float);
  const dim3 grid((unsigned)((size_t)C * HW / FC_BN)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FC_FWD(KIND)                                                                                                       \
  {                                                                                                                        \
    static bool s = false;                                                                                                 \
    if (!s) { (void)hipFuncSetAttribute((const void*)fc_nhwc_fwd_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)(((size_t)FC_BN * (FC_KMAX + 4) + (size_t)FC_BB * (FC_BN + 4) + FC_BN) * sizeof(float))); s = true; } \
    hipLaunchKernelGGL(fc_nhwc_fwd_kernel<KIND>, grid, block, (unsigned)smem, st, a);                                      \
  }
  if (y_kind == 0) FC_FWD(0) else if (y_kind == 1) FC_FWD(1) else FC_FWD(2)
#undef FC_FWD
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_fc_nhwc_bwd(const float* x, const void* y, const void* dy, float* dw, float* db, int B, int K, int C, int HW,
                       int act, float slope, int kind, size_t y_pstride, size_t dy_pstride, float beta, void* stream) {
  IPR_CHECK(x && dy && dw && (y || act == IPRGAN_ACT_NONE), "fc_nhwc_bwd: null tensor");
  IPR_CHECK(fc_shape_ok(B, K, C, HW), "fc_nhwc_bwd: unsupported shape B %d K %d C %d HW %d (iprgan_fc_nhwc_ok)", B, K, C, HW);
  IPR_CHECK(kind >= 0 && kind <= 2, "fc_nhwc_bwd: storage kind %d", kind);
  FcArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = const_cast<void*>(y); a.dy = dy; a.dw = dw; a.db = db; a.B = B; a.K = K; a.C = C; a.HW = HW; a.act = act;
  a.slope = slope; a.beta = beta;
  a.y_ps = y_pstride ? y_pstride : (size_t)B * C * HW;
  a.dy_ps = dy_pstride ? dy_pstride : (size_t)B * C * HW;
  const size_t smem = (size_t)FC_BW * FC_DZP * sizeof(float);
  const dim3 grid((unsigned)((size_t)C * HW / FC_BN)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FC_BWD(KIND)                                                                                                       \
  {                                                                                                                        \
    static bool s = false;                                                                                                 \
    if (!s) { (void)hipFuncSetAttribute((const void*)fc_nhwc_bwd_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)((size_t)FC_BW * FC_DZP * sizeof(float))); s = true; } \
    hipLaunchKernelGGL(fc_nhwc_bwd_kernel<KIND>, grid, block, (unsigned)smem, st, a);                                      \
  }
  if (kind == 0) FC_BWD(0) else if (kind == 1) FC_BWD(1) else FC_BWD(2)
#undef FC_BWD
  IPR_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
