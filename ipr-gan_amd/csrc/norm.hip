// BatchNorm2d (training / eval) and bias-gradient column sums over NHWC activations x[M][C]: HBM-bound.
//
// Column reductions share one shape: a 256-thread block = TC channel-quads (16-byte loads, coalesced
// along C) x TR row lanes; it walks a fixed slice of rows, reduces the row lanes through LDS and emits
// one partial per (slice, channel).  A second kernel sums the slices in slice order, so the result is
// deterministic (no float atomics).  Variance uses sums shifted by the tensor's first row
// (var = E[(x-s)^2] - E[x-s]^2 with s = x[0][c], s within a few sigma of the mean), which removes the
// E[x^2]-E[x]^2 cancellation while keeping a single pass over x.
// Semantics follow torch.nn.BatchNorm2d (biased variance to normalise, unbiased into running_var;
// reference networks/conv_generator.py:9, sr_resnet.py:23, discriminator_96.py:31).
#include "common.h"

namespace iprgan {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));
// Activation storage (ST = IPRGAN_ST_* of include/iprgan.h): fp32, bf16 ("bf16 activations"), or three bf16 planes ps
// elements apart (x = h + (m + l) exactly; split once per element when stored).  `e` is the ELEMENT index of 4 consecutive
// channels; arithmetic is fp32 in every case.
__device__ __forceinline__ f32x4 widen4(const nbf16x4 h) {
  const f32x4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
  return v;
}
__device__ __forceinline__ nbf16x4 narrow4(const f32x4& v) {
  const nbf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return h;
}
template <int ST>
__device__ __forceinline__ f32x4 ldv(const float* base, size_t e, size_t ps) {
  if (ST == 2) {
    const __bf16* b = (const __bf16*)base + e;
    return widen4(*(const nbf16x4*)b) + (widen4(*(const nbf16x4*)(b + ps)) + widen4(*(const nbf16x4*)(b + 2 * ps)));
  }
  if (ST == 1) return widen4(*(const nbf16x4*)((const __bf16*)base + e));
  return *(const f32x4*)(base + e);
}
template <int ST>
__device__ __forceinline__ void stv(float* base, size_t e, const f32x4& v, size_t ps) {
  if (ST == 2) {
    __bf16* b = (__bf16*)base + e;
    const nbf16x4 h = narrow4(v);
    const f32x4 r1 = v - widen4(h);
    const nbf16x4 m = narrow4(r1);
    *(nbf16x4*)b = h;
    *(nbf16x4*)(b + ps) = m;
    *(nbf16x4*)(b + 2 * ps) = narrow4(r1 - widen4(m));
  } else if (ST == 1) {
    *(nbf16x4*)((__bf16*)base + e) = narrow4(v);
  } else {
    *(f32x4*)(base + e) = v;
  }
}
// Storage kind 3 (norm entry points only, IPRGAN_ST_X3_XF32): the layer's INPUT x is fp32 - the convolution in front of a norm
// layer writes 4 instead of 6 bytes per element, and x is read three times (apply, backward reduction, backward apply) -
// while y, dy, dx and the residual are three-plane tensors.  SX / SY: the kind of x / of every other tensor.
// Storage kind 4 (backward entry points only, IPRGAN_ST_X3_XDF32, round 6): x AND dy are fp32 - the backward-data pass that
// feeds a norm backward writes 4 instead of 6 bytes per element too, and dy is read twice (reduction, apply) - while dx (and
// y) are three-plane tensors.  SG: the kind of dy.
#define SX ((ST == 3 || ST == 4) ? 0 : ST)
#define SY ((ST == 3 || ST == 4) ? 2 : ST)
#define SG (ST == 4 ? 0 : SY)
#define ST_PICK_BWD(kind, K, ...) ((kind) == 4 ? K<__VA_ARGS__, 4> : ST_PICK(kind, K, __VA_ARGS__))
// launch-time choice of a kernel instantiation by storage kind
#define ST_PICK(kind, K, ...) ((kind) == 3 ? K<__VA_ARGS__, 3> : (kind) == 2 ? K<__VA_ARGS__, 2> : (kind) == 1 ? K<__VA_ARGS__, 1> : K<__VA_ARGS__, 0>)

struct ColGeom {
  int TC, TR, gy, NB, rows_per_block;
};
static ColGeom col_geom(int M, int C) {
  ColGeom g;
  const int cq = C / 4;
  g.TC = cq < 64 ? cq : 64;
  // TC must divide 256 for the thread layout: round down to a power of two
  int tc = 1;
  while (tc * 2 <= g.TC) tc *= 2;
  g.TC = tc;
  g.TR = 256 / g.TC;
  g.gy = cdiv(cq, g.TC);
  int nb = cdiv(M, g.TR * 4);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  g.rows_per_block = rup(cdiv(M, nb), g.TR);
  g.NB = cdiv(M, g.rows_per_block);
  return g;
}

// MODE 0: s0 = sum x                     (bias gradient)
// MODE 1: s0 = sum (x-s), s1 = sum (x-s)^2            (BN statistics, s = x[0][c])
// MODE 2: s0 = sum dz, s1 = sum dz*xhat, dz = dy*act'(y), xhat = (x-mean)*invstd   (BN backward)
// Backward of norm + activation: the derivative mask of ReLU / LeakyReLU is recomputed from x (v = (x - mean) * invstd
// * gamma + beta, the forward's own expression: y > 0 <=> v > 0) instead of reading the saved output y - one tensor
// less per pass; with no activation y is not read either.  Other activations (never fused into a norm here) read y.
__host__ __device__ __forceinline__ bool act_from_x(int act) { return act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; }

template <int MODE, int ST = 0>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ y,
                                                        const float* __restrict__ dy,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        float* __restrict__ part, int M, int C, int TC,
                                                        int rows_per_block, int act, float slope,
                                                        const float* __restrict__ gamma = nullptr,
                                                        const float* __restrict__ beta = nullptr,
                                                        const float* __restrict__ slope_ptr = nullptr, size_t ps = 0) {
  if (slope_ptr) slope = *slope_ptr;          // PReLU fused into the norm layer: LeakyReLU with a learnable, device-resident slope
  // blockIdx.z = group (InstanceNorm: one group per sample; BatchNorm: a single group)
  const int grp = blockIdx.z;
  const size_t gofs = (size_t)grp * M * C;       // element offset of the group (pointer arithmetic below is per element)
  if (MODE == 2) { mean += (size_t)grp * C; invstd += (size_t)grp * C; }
  part += (size_t)grp * gridDim.x * 2 * C;
  __shared__ f32x4 sh[2][256];
  const int TR = 256 / TC;
  const int tc = threadIdx.x % TC, tr = threadIdx.x / TC;
  const int cq = blockIdx.y * TC + tc;
  const bool ok = cq * 4 < C;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {1.f, 1.f, 1.f, 1.f};
    if (MODE == 1) p0 = ldv<SX>(x, gofs + cq * 4, ps);
    f32x4 pg = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 2) {
      p0 = *(const f32x4*)(mean + cq * 4); p1 = *(const f32x4*)(invstd + cq * 4);
      if (gamma) pg = *(const f32x4*)(gamma + cq * 4);
      if (beta) pb = *(const f32x4*)(beta + cq * 4);
    }
    const bool from_x = MODE == 2 && act_from_x(act), no_act = MODE == 2 && act == IPRGAN_ACT_NONE;
    auto accumulate = [&](const f32x4 xv, const f32x4 gv, const f32x4 yv) {
      if (MODE == 0) {
        a0 += xv;
      } else if (MODE == 1) {
        const f32x4 d = xv - p0;
        a0 += d;
        a1 += d * d;
      } else {
        f32x4 dz = gv;
        if (from_x) {
#pragma unroll
          for (int k = 0; k < 4; ++k) dz[k] = gv[k] * act_grad_from_out((xv[k] - p0[k]) * p1[k] * pg[k] + pb[k], act, slope);
        } else if (!no_act) {
#pragma unroll
          for (int k = 0; k < 4; ++k) dz[k] = gv[k] * act_grad_from_out(yv[k], act, slope);
        }
        a0 += dz;
        a1 += dz * ((xv - p0) * p1);
      }
    };
    const bool need_y = MODE == 2 && !from_x && !no_act;
    int r = r0 + tr;
    // four rows per trip: all their loads are issued before the first is consumed (a thread streams 2-3 tensors with
    // nothing else to hide the latency behind; one row per trip ran the bf16 tensors of DCGAN-128 at 3 TB/s)
    for (; r + 3 * TR < r1; r += 4 * TR) {
      f32x4 xv[4], gv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t off = gofs + (size_t)(r + u * TR) * C + cq * 4;
        xv[u] = ldv<SX>(x, off, ps);
        if (MODE == 2) gv[u] = ldv<SG>(dy, off, ps);
        if (need_y) yv[u] = ldv<SY>(y, off, ps);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(xv[u], MODE == 2 ? gv[u] : xv[u], need_y ? yv[u] : xv[u]);   // same row order as below
    }
    for (; r < r1; r += TR) {
      const size_t off = gofs + (size_t)r * C + cq * 4;
      const f32x4 xv = ldv<SX>(x, off, ps);
      const f32x4 gv = MODE == 2 ? ldv<SG>(dy, off, ps) : xv;
      const f32x4 yv = need_y ? ldv<SY>(y, off, ps) : xv;
      accumulate(xv, gv, yv);
    }
  }
  sh[0][threadIdx.x] = a0;
  sh[1][threadIdx.x] = a1;
  __syncthreads();
  if (tr == 0 && ok) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < TR; ++i) { s0 += sh[0][i * TC + tc]; s1 += sh[1][i * TC + tc]; }
    float* p = part + ((size_t)blockIdx.x * 2) * C + cq * 4;
    *(f32x4*)p = s0;
    if (MODE != 0) *(f32x4*)(p + C) = s1;
  }
}

// Sums the NB partial rows of the 64 channels of a 1024-thread block: 64 row lanes x 16 channel quads, 16-byte loads (a
// lane walks rows lane, lane + 64, ... with two independent accumulators), then the row lanes are added in lane order by
// the first 64 threads.  Results are valid in the threads with lane (= threadIdx.x >> 6) == 0; thread t there owns
// channel blockIdx.x * 64 + t.  Deterministic: fixed row -> lane assignment, fixed addition order.  (The first version
// gave every channel 16 scalar lanes: beyond ~150 rows it was latency-bound and needed the compaction pre-pass below;
// this one takes 2048 rows directly - every norm layer of SRGAN and CycleGAN.)
#define FL 16
__device__ __forceinline__ void final_sums(const float* __restrict__ part, int NB, int C, int c, int lane,
                                           float (*sh)[FL][64], float& s0, float& s1, bool two) {
  (void)sh;
  __shared__ f32x4 shq[2][64][16];
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int cq = (c & ~63) + 4 * q;                 // first channel of this thread's quad (c & ~63 = blockIdx.x * 64)
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
  if (cq < C) {
    int r = rl;
    for (; r + 64 < NB; r += 128) {
      a0 += *(const f32x4*)(part + ((size_t)r * 2) * C + cq);
      b0 += *(const f32x4*)(part + ((size_t)(r + 64) * 2) * C + cq);
      if (two) {
        a1 += *(const f32x4*)(part + ((size_t)r * 2 + 1) * C + cq);
        b1 += *(const f32x4*)(part + ((size_t)(r + 64) * 2 + 1) * C + cq);
      }
    }
    if (r < NB) {
      a0 += *(const f32x4*)(part + ((size_t)r * 2) * C + cq);
      if (two) a1 += *(const f32x4*)(part + ((size_t)r * 2 + 1) * C + cq);
    }
  }
  shq[0][rl][q] = a0 + b0;
  shq[1][rl][q] = a1 + b1;
  __syncthreads();
  s0 = s1 = 0.f;
  if (lane == 0) {
    const int cl = threadIdx.x & 63;
    for (int l = 0; l < 64; ++l) { s0 += shq[0][l][cl >> 2][cl & 3]; s1 += shq[1][l][cl >> 2][cl & 3]; }
  }
}

// Convolution epilogues emit one row of column sums per OUTPUT TILE (thousands of rows for the largest layers); beyond
// COMPACT_ABOVE rows this pre-pass folds part[G][rows][W] (W = 2*Cs floats per row) down to out[G][NBC][W], slice by slice in row order.
#define COMPACT_ABOVE 2048
#define NBC 32
__global__ __launch_bounds__(256) void part_compact_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ out,
                                                           int rows, int W4, int rps) {
  __shared__ f32x4 sh[4][64];
  const int grp = blockIdx.z, q = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  part += (size_t)grp * rows * W4;
  const int r0 = blockIdx.x * rps;
  int r1 = r0 + rps;
  if (r1 > rows) r1 = rows;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (q < W4) {
    int r = r0 + rl;
    for (; r + 4 < r1; r += 8) { a += part[(size_t)r * W4 + q]; b += part[(size_t)(r + 4) * W4 + q]; }
    if (r < r1) a += part[(size_t)r * W4 + q];
  }
  sh[rl][threadIdx.x & 63] = a + b;
  __syncthreads();
  if (rl == 0 && q < W4)
    out[((size_t)grp * gridDim.x + blockIdx.x) * W4 + q] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// returns the partial pointer / row count the final kernel should read (compacted into the spare room behind the rows
// when there are many: iprgan_conv_stat_floats reserves it)
static int compact_partials(const float*& part, int& rows_per_group, int G, int Cs, hipStream_t st) {
  if (rows_per_group <= COMPACT_ABOVE) return 0;
  float* out = const_cast<float*>(part) + (size_t)G * rows_per_group * 2 * Cs;
  const int W4 = 2 * Cs / 4, rps = cdiv(rows_per_group, NBC), nb = cdiv(rows_per_group, rps);
  hipLaunchKernelGGL(part_compact_kernel, dim3(nb, cdiv(W4, 64), G), dim3(256), 0, st, (const f32x4*)part, (f32x4*)out,
                     rows_per_group, W4, rps);
  IPR_LAUNCH_CHECK();
  part = out;
  rows_per_group = nb;
  return 0;
}

__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int NB, int Cs,
                                                           int C, float* __restrict__ out, float beta) {
  __shared__ float sh[2][FL][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
  float s0, s1;
  final_sums(part, NB, Cs, c, lane, sh, s0, s1, false);
  if (lane == 0 && c < C) out[c] = beta != 0.f ? beta * out[c] + s0 : s0;
}

// the column sums of SEVERAL layers' partials in one launch (iprgan_colsum_partials_multi): a discriminator's backward pass
// owes one bias gradient per convolution - eight launches of 4-11 us with one to eight blocks each; every block does what its
// colsum_final_kernel block would have done (bit-identical)
#define COLSUM_MULTI_MAX 24
struct ColsumTable {
  const float* part[COLSUM_MULTI_MAX];
  float* out[COLSUM_MULTI_MAX];
  int NB[COLSUM_MULTI_MAX], Cs[COLSUM_MULTI_MAX], C[COLSUM_MULTI_MAX];
  float beta[COLSUM_MULTI_MAX];
  unsigned first[COLSUM_MULTI_MAX + 1];
  int n;
};
__global__ __launch_bounds__(1024) void colsum_final_multi_kernel(const ColsumTable t) {
  __shared__ float sh[2][FL][64];
  int e = 0;
  while (e + 1 < t.n && blockIdx.x >= t.first[e + 1]) ++e;
  const int blk = blockIdx.x - t.first[e];
  const int c = blk * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
  float s0, s1;
  final_sums(t.part[e], t.NB[e], t.Cs[e], c, lane, sh, s0, s1, false);
  float* out = t.out[e];
  if (lane == 0 && c < t.C[e]) out[c] = t.beta[e] != 0.f ? t.beta[e] * out[c] + s0 : s0;
}

__global__ __launch_bounds__(1024) void bn_stats_final_kernel(const float* __restrict__ part,
                                                             const float* __restrict__ x, int NB, int M,
                                                             int C, float eps, float momentum,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var,
                                                             float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd,
                                                             int shift_vec, long long* __restrict__ counter, int x_b16, size_t ps) {
  // shift_vec 0: the sums are about s = x[0][c] of the group (colreduce_kernel<1>); 1: about the per-channel vector
  // x[c] itself, or about zero when x is null (sums emitted by the producing convolution's epilogue, taken before
  // its bias was added: s = bias)
  const int grp = blockIdx.y;
  part += (size_t)grp * NB * 2 * C;
  const size_t xofs = shift_vec ? 0 : (size_t)grp * M * C;
  save_mean += (size_t)grp * C;
  save_invstd += (size_t)grp * C;
  if (counter && blockIdx.x == 0 && grp == 0 && threadIdx.x == 0) *counter += 1;     // num_batches_tracked
  __shared__ float sh[2][FL][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
  float s0, s1;
  final_sums(part, NB, C, c, lane, sh, s0, s1, true);
  if (lane != 0 || c >= C) return;
  const float invM = 1.0f / (float)M;
  const float d = s0 * invM;                 // E[x - s]
  float shift = 0.f;
  if (x) {
    const __bf16* xb = (const __bf16*)x + xofs + c;
    shift = shift_vec || !x_b16 ? x[xofs + c] : x_b16 == 2 ? (float)xb[0] + ((float)xb[ps] + (float)xb[2 * ps]) : (float)xb[0];
  }
  const float mean = shift + d;
  float var = s1 * invM - d * d;             // biased variance
  if (var < 0.f) var = 0.f;
  save_mean[c] = mean;
  save_invstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean,
                                     const float* __restrict__ running_var, float eps, int C,
                                     float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = running_mean[c];
  save_invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = act((x-mean)*invstd*gamma+beta), 16 B per lane over [M][C].
// The per-channel constants are vector loads; when the channel chunk of a thread never changes (FIXED: 256 % C4n
// == 0, so block base and grid stride are multiples of the row) and there is one group (BatchNorm) they are loaded
// ONCE per thread.  The first version fetched 16 scalars per 16 bytes of data and ran at a third of the streaming
// rate torch's elementwise kernels reach on the same tensors.
__device__ __forceinline__ f32x4 ld4(const float* p, int c, float dflt) {
  if (p) return *(const f32x4*)(p + c);
  const f32x4 d = {dflt, dflt, dflt, dflt};
  return d;
}
template <bool FIXED, int ST = 0>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       unsigned n4, int C4n, FastDiv d_c4n, FastDiv d_group4, int act,
                                                       float slope, const float* __restrict__ residual,
                                                       const float* __restrict__ slope_ptr, size_t ps) {
  if (slope_ptr) slope = *slope_ptr;          // PReLU as the norm layer's activation (networks/sr_resnet.py:7,13)
  // residual: y = act(norm(x)) + residual - the skip connection that closes a residual block right after its last norm
  // layer (networks/sr_resnet.py:37-38, resnet_generator.py:52-53), folded into this pass
  const unsigned stride = gridDim.x * blockDim.x;
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (FIXED) {               // 256 % C4n == 0: a thread keeps its channel quad; blockIdx.y = group (InstanceNorm: sample)
    const int c = (int)(threadIdx.x % (unsigned)C4n) * 4;
    const size_t go = (size_t)blockIdx.y * (size_t)C4n * 4;
    const f32x4 g = ld4(gamma, c, 1.f), b = ld4(beta, c, 0.f), m = ld4(mean + go, c, 0.f), is = ld4(invstd + go, c, 1.f);
    const size_t base = (size_t)blockIdx.y * n4;        // n4 = 16-byte quads per group here
#pragma unroll 4
    for (; i < n4; i += stride) {
      const f32x4 v = ldv<SX>(x, (base + i) * 4, ps);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = act_apply((v[k] - m[k]) * is[k] * g[k] + b[k], act, slope);
      if (residual) o += ldv<SY>(residual, (base + i) * 4, ps);
      stv<SY>(y, (base + i) * 4, o, ps);
    }
    return;
  }
  for (; i < n4; i += stride) {
    const int c = (int)(i - fdiv(i, d_c4n) * (unsigned)C4n) * 4;
    const size_t go = (size_t)fdiv(i, d_group4) * (size_t)C4n * 4;     // group offset into mean/invstd
    const f32x4 g = ld4(gamma, c, 1.f), b = ld4(beta, c, 0.f), m = ld4(mean + go, c, 0.f), is = ld4(invstd + go, c, 1.f);
    const f32x4 v = ldv<SX>(x, (size_t)i * 4, ps);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = act_apply((v[k] - m[k]) * is[k] * g[k] + b[k], act, slope);
    if (residual) o += ldv<SY>(residual, (size_t)i * 4, ps);
    stv<SY>(y, (size_t)i * 4, o, ps);
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const float* __restrict__ part, int NB, int C,
                                                           float* __restrict__ sums,
                                                           float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  const int grp = blockIdx.y;
  part += (size_t)grp * NB * 2 * C;
  sums += (size_t)grp * 2 * C;
  __shared__ float sh[2][FL][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
  float s1, s2;
  final_sums(part, NB, C, c, lane, sh, s1, s2, true);
  if (lane != 0 || c >= C) return;
  sums[c] = s1;
  sums[C + c] = s2;
  if (dgamma) dgamma[c] = s2;
  if (dbeta) dbeta[c] = s1;
}

template <bool FIXED, int ST = 0>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, float* __restrict__ dx,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ sums,
                                                           unsigned n4, int C4n, int C, FastDiv d_c4n, FastDiv d_group4,
                                                           float invM, int act, float slope, float* __restrict__ colpart,
                                                           const float* __restrict__ slope_ptr, float* __restrict__ apart, size_t ps) {
  // slope_ptr / apart: PReLU fused into the norm layer - the slope is read from the device and every block emits its share
  // of the slope gradient sum dy * min(v, 0), v = the normalised, affine value (one more partial array, summed in double)
  if (slope_ptr) slope = *slope_ptr;
  float asum = 0.f;
  // colpart (needs 256 % C4n == 0: a thread keeps its channel chunk): per-block column sums of dx, [block][2][C] like
  // every other partial buffer - the bias gradient of the convolution that feeds this norm layer, without another
  // pass over dx (colreduce_kernel<0>)
  const unsigned stride = gridDim.x * blockDim.x;
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool from_x = act_from_x(act), no_act = act == IPRGAN_ACT_NONE;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  auto body = [&](size_t idx, const f32x4& g, const f32x4& b, const f32x4& m, const f32x4& is, const f32x4& s1,
                  const f32x4& s2) {
    const f32x4 xv = ldv<SX>(x, idx * 4, ps), gv = ldv<SG>(dy, idx * 4, ps);
    f32x4 yv = {0.f, 0.f, 0.f, 0.f};
    if (!from_x && !no_act) yv = ldv<SY>(y, idx * 4, ps);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t = (xv[k] - m[k]) * is[k];
      const float v = t * g[k] + b[k];
      if (apart) asum += v > 0.f ? 0.f : gv[k] * v;
      const float dz = no_act ? gv[k] : gv[k] * act_grad_from_out(from_x ? v : yv[k], act, slope);
      o[k] = g[k] * is[k] * (dz - s1[k] * invM - t * s2[k] * invM);
    }
    stv<SY>(dx, idx * 4, o, ps);
    csum += o;
  };
  if (FIXED) {               // see bn_apply_kernel: blockIdx.y = group, n4 = quads per group
    const int c = (int)(threadIdx.x % (unsigned)C4n) * 4;
    const size_t grp = blockIdx.y;
    const f32x4 g = ld4(gamma, c, 1.f), b = ld4(beta, c, 0.f), m = ld4(mean + grp * C, c, 0.f), is = ld4(invstd + grp * C, c, 1.f);
    const f32x4 s1 = ld4(sums + grp * 2 * C, c, 0.f), s2 = ld4(sums + grp * 2 * C + C, c, 0.f);
    const size_t base = grp * n4;
#pragma unroll 2
    for (; i < n4; i += stride) body(base + i, g, b, m, is, s1, s2);
  } else {
    for (; i < n4; i += stride) {
      const int c = (int)(i - fdiv(i, d_c4n) * (unsigned)C4n) * 4;
      const size_t grp = fdiv(i, d_group4);
      const f32x4 g = ld4(gamma, c, 1.f), b = ld4(beta, c, 0.f), m = ld4(mean + grp * C, c, 0.f),
                  is = ld4(invstd + grp * C, c, 1.f);
      const f32x4 s1 = ld4(sums + grp * 2 * C, c, 0.f), s2 = ld4(sums + grp * 2 * C + C, c, 0.f);
      body(i, g, b, m, is, s1, s2);
    }
  }
  if (apart) {
    __shared__ float sha[16];
    const float sb = block_sum(asum, sha);
    if (threadIdx.x == 0) apart[blockIdx.y * gridDim.x + blockIdx.x] = sb;
  }
  if (colpart) {
    __shared__ f32x4 sh[256];
    sh[threadIdx.x] = csum;
    __syncthreads();
    if ((int)threadIdx.x < C4n) {
      f32x4 t = sh[threadIdx.x];
      for (int k = threadIdx.x + C4n; k < 256; k += C4n) t += sh[k];
      *(f32x4*)(colpart + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2) * C + threadIdx.x * 4) = t;
    }
  }
}

// InstanceNorm affine gradients: dgamma[c] = sum_g sums[g][C+c], dbeta[c] = sum_g sums[g][c]
__global__ void group_sum_kernel(const float* __restrict__ sums, int G, int C, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int g = 0; g < G; ++g) { s1 += sums[(size_t)g * 2 * C + c]; s2 += sums[(size_t)g * 2 * C + C + c]; }
  if (dgamma) dgamma[c] = s2;
  if (dbeta) dbeta[c] = s1;
}

// used by conv_igemm.hip for the bias gradient
size_t colsum_ws_floats(int M, int Cs) {
  const ColGeom g = col_geom(M, Cs);
  return (size_t)g.NB * 2 * Cs;
}
int colsum_launch(const float* x, float* out, float* ws, int M, int Cs, int C, hipStream_t st, float beta, int b16, size_t ps) {
  const ColGeom g = col_geom(M, Cs);
  auto kr0 = ST_PICK(b16, colreduce_kernel, 0);
  hipLaunchKernelGGL(kr0, dim3(g.NB, g.gy), dim3(256), 0, st, x,
                     nullptr, nullptr, nullptr, nullptr, ws, M, Cs, g.TC, g.rows_per_block, 0, 0.f, nullptr, nullptr, nullptr,
                     ps ? ps : (size_t)M * Cs);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 64)), dim3(64 * FL), 0, st, ws, g.NB, Cs, C, out, beta);
  IPR_LAUNCH_CHECK();
  return 0;
}

}  // namespace iprgan

using namespace iprgan;

namespace iprgan {
// sum of up to a few thousand block partials on one wave, in double, fixed order
__global__ void wave_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) *out = (float)s;
}
}  // namespace iprgan

static int norm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* save_mean, float* save_invstd, float* ws, int G, int M, int C,
                    float eps, float momentum, int use_running, int act, float slope, hipStream_t st,
                    const float* part = nullptr, int part_rows = 0, const float* shift = nullptr,
                    long long* counter = nullptr, const float* residual = nullptr, int b16 = 0,
                    const float* slope_ptr = nullptr) {
  IPR_CHECK(C % 4 == 0, "norm_fwd: C=%d must be a multiple of 4", C);
  IPR_CHECK(M > 0 && G > 0, "norm_fwd: empty input");
  const size_t ps = (size_t)G * M * C;         // three-plane tensors: contiguous, plane stride = the tensor
  if (use_running) {
    IPR_CHECK(running_mean && running_var && G == 1, "norm_fwd: eval mode needs running stats");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, running_mean,
                       running_var, eps, C, save_mean, save_invstd);
  } else if (part) {
    // column sums already emitted tile by tile by the convolution that produced x: no pass over x for statistics
    IPR_CHECK(part_rows > 0 && part_rows % G == 0, "norm_fwd: %d partial rows for %d groups", part_rows, G);
    int rpg = part_rows / G;
    if (compact_partials(part, rpg, G, C, st)) return 2;
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 64), G), dim3(64 * FL), 0, st, part, shift, rpg, M,
                       C, eps, momentum, running_mean, running_var, save_mean, save_invstd, 1, counter, 0, (size_t)0);
  } else {
    const ColGeom g = col_geom(M, C);
    auto kr1 = ST_PICK(b16, colreduce_kernel, 1);
    hipLaunchKernelGGL(kr1, dim3(g.NB, g.gy, G), dim3(256), 0, st,
                       x, nullptr, nullptr, nullptr, nullptr, ws, M, C, g.TC, g.rows_per_block, 0, 0.f, nullptr, nullptr, nullptr, ps);
    IPR_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 64), G), dim3(64 * FL), 0, st, ws, x, g.NB, M, C,
                       eps, momentum, running_mean, running_var, save_mean, save_invstd, 0, counter, b16 == 3 ? 0 : b16, ps);
  }
  IPR_LAUNCH_CHECK();
  const size_t n4 = (size_t)G * M * C / 4;
  IPR_CHECK(n4 < 0x7fffffffull, "norm_fwd: tensor of %zu elements is too large", n4 * 4);
  int blocks = (int)(cdivz(n4, 256) < 4096 ? cdivz(n4, 256) : 4096);
  const bool fixed = 256 % (C / 4) == 0 && G <= 65535;
  auto kern = fixed ? ST_PICK(b16, bn_apply_kernel, true) : ST_PICK(b16, bn_apply_kernel, false);
  const size_t n4g = (size_t)M * C / 4;              // quads per group
  if (fixed) blocks = (int)(cdivz(n4g, 256) < (size_t)cdiv(4096, G) ? cdivz(n4g, 256) : (size_t)cdiv(4096, G));
  hipLaunchKernelGGL(kern, dim3(blocks, fixed ? G : 1), dim3(256), 0, st, x, y, gamma, beta, save_mean, save_invstd,
                     (unsigned)(fixed ? n4g : n4), C / 4,
                     make_fastdiv(C / 4), make_fastdiv((uint32_t)n4g), act, slope, residual, slope_ptr, ps);
  IPR_LAUNCH_CHECK();
  return 0;
}

static int norm_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                    float* ws, int G, int M, int C, int act, float slope, hipStream_t st,
                    float* dbias_prev = nullptr, int dbias_n = 0, float dbias_beta = 0.f, int b16 = 0,
                    const float* pre_part = nullptr, int pre_rows = 0, const float* slope_ptr = nullptr,
                    float* dslope = nullptr) {
  // pre_part: the two reductions (sum dz, sum dz * xhat; dz = dy * act') were taken by the epilogue of the backward-data
  // pass that produced dy (iprgan_conv_bwd_data_bn), which stored dz in dy's place: no reduction pass, no mask here
  IPR_CHECK(C % 4 == 0, "norm_bwd: C=%d must be a multiple of 4", C);
  const size_t ps = (size_t)G * M * C;
  if (pre_part) act = IPRGAN_ACT_NONE;
  IPR_CHECK(act == IPRGAN_ACT_NONE || act_from_x(act) || y, "norm_bwd: this activation needs the saved output y");
  IPR_CHECK(!act_from_x(act) || !gamma == !beta, "norm_bwd: the ReLU mask is recomputed from x: gamma and beta are both needed (or both absent)");
  const ColGeom g = col_geom(M, C);
  float* sums = ws + (size_t)G * g.NB * 2 * C;
  if (pre_part) {
    IPR_CHECK(G == 1 && pre_rows > 0, "norm_bwd: epilogue partials are per batch (G = 1)");
    const float* pp = pre_part;
    int rows = pre_rows;
    if (compact_partials(pp, rows, 1, C, st)) return 2;
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 64), 1), dim3(64 * FL), 0, st, pp, rows, C, sums, dgamma, dbeta);
    IPR_LAUNCH_CHECK();
  } else {
    auto kr2 = ST_PICK_BWD(b16, colreduce_kernel, 2);
    hipLaunchKernelGGL(kr2, dim3(g.NB, g.gy, G), dim3(256), 0, st,
                       x, y, dy, save_mean, save_invstd, ws, M, C, g.TC, g.rows_per_block, act, slope, gamma, beta, slope_ptr, ps);
    IPR_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 64), G), dim3(64 * FL), 0, st, ws, g.NB, C, sums,
                       G == 1 ? dgamma : nullptr, G == 1 ? dbeta : nullptr);
    IPR_LAUNCH_CHECK();
  }
  if (G > 1 && (dgamma || dbeta)) {
    hipLaunchKernelGGL(group_sum_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, sums, G, C, dgamma, dbeta);
    IPR_LAUNCH_CHECK();
  }
  const size_t n4 = (size_t)G * M * C / 4;
  IPR_CHECK(n4 < 0x7fffffffull, "norm_bwd: tensor of %zu elements is too large", n4 * 4);
  int blocks = (int)(cdivz(n4, 256) < 4096 ? cdivz(n4, 256) : 4096);
  const bool fixed = 256 % (C / 4) == 0 && G <= 1024;
  const size_t n4g = (size_t)M * C / 4;              // quads per group
  int gy = 1;
  if (fixed) { gy = G; blocks = (int)(cdivz(n4g, 256) < (size_t)cdiv(4096, G) ? cdivz(n4g, 256) : (size_t)cdiv(4096, G)); }
  // column sums of dx (bias gradient of the producing convolution) ride on the apply pass when a thread keeps its
  // channel chunk; the per-block partials (at most 1024 rows) live behind the reduction workspace
  float* apart = sums + (size_t)G * 2 * C + (size_t)(1024 + NBC) * 2 * C;     // per-block slope-gradient partials (<= 4096)
  float* colpart = nullptr;
  if (dbias_prev && 256 % (C / 4) == 0) {
    const int cap = fixed ? (1024 / G > 0 ? 1024 / G : 1) : 1024;
    if (blocks > cap) blocks = cap;
    colpart = sums + (size_t)G * 2 * C;
  }
  auto kern = fixed ? ST_PICK_BWD(b16, bn_bwd_apply_kernel, true) : ST_PICK_BWD(b16, bn_bwd_apply_kernel, false);
  hipLaunchKernelGGL(kern, dim3(blocks, gy), dim3(256), 0, st, x, y, dy, dx, gamma, beta, save_mean, save_invstd, sums,
                     (unsigned)(fixed ? n4g : n4), C / 4, C,
                     make_fastdiv(C / 4), make_fastdiv((uint32_t)n4g), 1.0f / (float)M, act, slope,
                     colpart, slope_ptr, dslope ? apart : nullptr, ps);
  IPR_LAUNCH_CHECK();
  if (dslope) {
    hipLaunchKernelGGL(wave_sum_kernel, dim3(1), dim3(64), 0, st, apart, blocks * gy, dslope);
    IPR_LAUNCH_CHECK();
  }
  if (dbias_prev) {
    if (colpart) {
      const float* pp = colpart;
      int rows = blocks * gy;
      if (compact_partials(pp, rows, 1, C, st)) return 2;
      hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(dbias_n, 64)), dim3(64 * FL), 0, st, pp, rows, C, dbias_n,
                         dbias_prev, dbias_beta);
      IPR_LAUNCH_CHECK();
    } else {                 // channel counts that do not divide the block: the separate column-sum pass
      const int rc = colsum_launch(dx, dbias_prev, ws, G * M, C, dbias_n, st, dbias_beta, (b16 == 3 || b16 == 4) ? 2 : b16, ps);
      if (rc) return rc;
    }
  }
  return 0;
}

extern "C" {

// [reduction partials | sums] + [per-block column sums of dx (1024 blocks) | their compacted rows] for the bias gradient
size_t iprgan_bn_ws_floats(int M, int C) {
  const ColGeom g = col_geom(M, C);
  return (size_t)g.NB * 2 * C + 2 * (size_t)C + (size_t)(1024 + NBC) * 2 * C + 4096;
}
size_t iprgan_instnorm_ws_floats(int B, int HW, int C) {
  const ColGeom g = col_geom(HW, C);
  return (size_t)B * ((size_t)g.NB * 2 * C + 2 * (size_t)C) + (size_t)(1024 + NBC) * 2 * C;
}

int iprgan_colsum(const float* x, float* out, float* ws, int M, int Cs, int C, float beta, int x_bf16, void* stream) {
  IPR_CHECK(Cs % 4 == 0 && M > 0 && C <= Cs, "colsum: row length %d must be a multiple of 4 >= C=%d, M=%d positive", Cs, C, M);
  return colsum_launch(x, out, ws, M, Cs, C, (hipStream_t)stream, beta, x_bf16, 0);
}
size_t iprgan_colsum_ws_floats(int M, int C) { return colsum_ws_floats(M, C); }
int iprgan_colsum_partials_multi(const float* const* parts, const int* rows, const int* Cs, const int* C, float* const* outs,
                                 const float* betas, int n, void* stream) {
  IPR_CHECK(n >= 0 && (!n || (parts && rows && Cs && C && outs && betas)), "colsum_partials_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int i = 0;
  while (i < n) {
    ColsumTable t;
    memset(&t, 0, sizeof(t));
    unsigned blocks = 0;
    for (; i < n && t.n < COLSUM_MULTI_MAX; ++i) {
      IPR_CHECK(rows[i] > 0 && C[i] <= Cs[i] && parts[i] && outs[i], "colsum_partials_multi: entry %d: %d rows, C=%d, Cs=%d", i, rows[i], C[i], Cs[i]);
      const float* part = parts[i];
      int r = rows[i];
      if (compact_partials(part, r, 1, Cs[i], st)) return 2;       // (partials of thousands of rows: their own launch, as before)
      const int e = t.n++;
      t.part[e] = part; t.out[e] = outs[i]; t.NB[e] = r; t.Cs[e] = Cs[i]; t.C[e] = C[i]; t.beta[e] = betas[i];
      t.first[e] = blocks;
      blocks += (unsigned)cdiv(C[i], 64);
      t.first[e + 1] = blocks;
    }
    hipLaunchKernelGGL(colsum_final_multi_kernel, dim3(blocks), dim3(64 * FL), 0, st, t);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

int iprgan_colsum_partials(const float* part, int rows, int Cs, int C, float* out, float beta, void* stream) {
  IPR_CHECK(rows > 0 && C <= Cs, "colsum_partials: %d rows, C=%d, Cs=%d", rows, C, Cs);
  if (compact_partials(part, rows, 1, Cs, (hipStream_t)stream)) return 2;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 64)), dim3(64 * FL), 0, (hipStream_t)stream, part, rows, Cs, C, out, beta);
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_bn_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                  float* running_var, float* save_mean, float* save_invstd, float* ws, int M, int C,
                  float eps, float momentum, int use_running, int act, float slope, const float* conv_part,
                  int conv_part_rows, const float* conv_bias, long long* num_batches_tracked, const float* residual,
                  int act_bf16, void* stream) {
  return norm_fwd(x, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, ws, 1, M, C, eps,
                  momentum, use_running, act, slope, (hipStream_t)stream, conv_part, conv_part_rows, conv_bias,
                  use_running ? nullptr : num_batches_tracked, residual, act_bf16);
}
int iprgan_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                  const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                  float* dbeta, float* ws, int M, int C, int act, float slope, float* dbias_prev, int dbias_n,
                  float dbias_beta, int act_bf16, void* stream) {
  return norm_bwd(x, y, dy, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, ws, 1, M, C, act, slope,
                  (hipStream_t)stream, dbias_prev, dbias_n, dbias_beta, act_bf16);
}
// BatchNorm backward behind iprgan_conv_bwd_data_bn: dz = dy * act'(y) is already in `dz`, part[rows][2][C] holds the
// per-tile sums (sum dz, sum dz * xhat) of that pass's epilogue (room for the compacted rows behind them as for
// iprgan_conv_stat_floats).  Two launches (final + apply) and three tensor passes (x, dz -> dx) instead of five.
int iprgan_bn_bwd_pre(const float* x, const float* dz, const float* gamma, const float* save_mean,
                      const float* save_invstd, const float* part, int rows, float* dx, float* dgamma, float* dbeta,
                      float* ws, int M, int C, float* dbias_prev, int dbias_n, float dbias_beta, int act_bf16,
                      void* stream) {
  IPR_CHECK(part && rows > 0, "bn_bwd_pre: the epilogue partials are required");
  return norm_bwd(x, nullptr, dz, gamma, nullptr, save_mean, save_invstd, dx, dgamma, dbeta, ws, 1, M, C, IPRGAN_ACT_NONE,
                  0.f, (hipStream_t)stream, dbias_prev, dbias_n, dbias_beta, act_bf16, part, rows);
}
// BatchNorm2d + PReLU (one learnable slope, networks/sr_resnet.py:7,13: conv -> BatchNorm -> PReLU): the slope is read from
// the device (`slope`, one float), so the norm's apply pass IS the PReLU forward and its backward apply pass produces dx
// through both, with the slope gradient d_slope = sum dy * min(v, 0) summed on the same pass.
int iprgan_bn_prelu_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* save_mean, float* save_invstd, float* ws, int M, int C, float eps,
                        float momentum, int use_running, const float* slope, const float* conv_part, int conv_part_rows,
                        const float* conv_bias, long long* num_batches_tracked, const float* residual, int act_bf16,
                        void* stream) {
  IPR_CHECK(slope, "bn_prelu_fwd: the slope pointer is required");
  return norm_fwd(x, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, ws, 1, M, C, eps, momentum,
                  use_running, IPRGAN_ACT_LRELU, 0.f, (hipStream_t)stream, conv_part, conv_part_rows, conv_bias,
                  use_running ? nullptr : num_batches_tracked, residual, act_bf16, slope);
}
int iprgan_bn_prelu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, const float* slope, float* dx, float* dgamma, float* dbeta,
                        float* dslope, float* ws, int M, int C, float* dbias_prev, int dbias_n, float dbias_beta,
                        int act_bf16, void* stream) {
  IPR_CHECK(slope && dslope && gamma && beta, "bn_prelu_bwd: slope, its gradient output, gamma and beta are required");
  return norm_bwd(x, nullptr, dy, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, ws, 1, M, C, IPRGAN_ACT_LRELU, 0.f,
                  (hipStream_t)stream, dbias_prev, dbias_n, dbias_beta, act_bf16, nullptr, 0, slope, dslope);
}
int iprgan_instnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* save_mean,
                        float* save_invstd, float* ws, int B, int HW, int C, float eps, int act, float slope,
                        const float* conv_part, int conv_part_rows, const float* conv_bias, const float* residual,
                        int act_bf16, void* stream) {
  return norm_fwd(x, y, gamma, beta, nullptr, nullptr, save_mean, save_invstd, ws, B, HW, C, eps, 0.f, 0, act,
                  slope, (hipStream_t)stream, conv_part, conv_part_rows, conv_bias, nullptr, residual, act_bf16);
}
int iprgan_instnorm_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                        float* dbeta, float* ws, int B, int HW, int C, int act, float slope, float* dbias_prev,
                        int dbias_n, float dbias_beta, int act_bf16, void* stream) {
  return norm_bwd(x, y, dy, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, ws, B, HW, C, act, slope,
                  (hipStream_t)stream, dbias_prev, dbias_n, dbias_beta, act_bf16);
}

}  // extern "C"
