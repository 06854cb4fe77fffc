// HBM-/latency-bound helpers of the G+D step: layout changes at the NCHW API boundary, the
// GEMV head of SNDiscriminator, fused loss forward/gradient-seed kernels, the multi-tensor
// sign-loss / bit-error-rate kernels (tools/sign_model.py:42-60) and multi-tensor Adam.
#include <stdarg.h>

#include "common.h"

namespace iprgan {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// element access for activation tensors that may be stored as bf16 ("bf16 activations", include/iprgan.h)
__device__ __forceinline__ float lde(const float* p, size_t i, int b16) {
  return b16 ? (float)((const __bf16*)p)[i] : p[i];
}
__device__ __forceinline__ void ste(float* p, size_t i, float v, int b16) {
  if (b16) ((__bf16*)p)[i] = (__bf16)v;
  else p[i] = v;
}
__global__ void cast_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n, int src16, int dst16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    ste(dst, i, lde(src, i, src16), dst16);
}

// ---- layout --------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C,
                                    int HW, int Cs) {
  const size_t total = (size_t)B * HW * Cs;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cs);
    const size_t pix = i / Cs;
    const int p = (int)(pix % HW);
    const size_t b = pix / HW;
    dst[i] = c < C ? src[(b * C + c) * HW + p] : 0.f;
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C,
                                    int HW, int Cs) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const size_t bc = i / HW;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    dst[i] = src[(b * HW + p) * Cs + c];
  }
}
// dst[b][a][k] = src[a][b][k]
__global__ void permute_021_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int Bd,
                                   int K, float beta) {
  const size_t total = (size_t)A * Bd * K;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const size_t ba = i / K;
    const int a = (int)(ba % A);
    const size_t b = ba / A;
    const float v = src[((size_t)a * Bd + b) * K + k];
    dst[i] = beta != 0.f ? beta * dst[i] + v : v;
  }
}

__global__ void fill_kernel(float* p, float v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}
__global__ void axpy_kernel(float* y, const float* x, float a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}
// y_t += a * x_t for a table of tensors (small gradients - biases, norm scales - into their bucket views)
#define AXPY_MAX_TENSORS 64
struct AxpyTable {
  float* y[AXPY_MAX_TENSORS];
  const float* x[AXPY_MAX_TENSORS];
  long long n[AXPY_MAX_TENSORS];
};
__global__ void axpy_multi_kernel(const AxpyTable t, float a) {
  const int ti = blockIdx.y;
  const long long n = t.n[ti];
  float* __restrict__ y = t.y[ti];
  const float* __restrict__ x = t.x[ti];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}

// ---- GEMV head -----------------------------------------------------------------------------
// 4 consecutive elements of a tensor stored as fp32, bf16 or (b16 == 2) three bf16 planes ps elements apart
__device__ __forceinline__ float4 ld4e(const float* p, size_t i, int b16, size_t ps = 0) {
  if (b16 == 2) {
    typedef __bf16 h4 __attribute__((ext_vector_type(4)));
    const __bf16* b = (const __bf16*)p + i;
    const h4 h = *(const h4*)b, m = *(const h4*)(b + ps), l = *(const h4*)(b + 2 * ps);
    return make_float4((float)h.x + ((float)m.x + (float)l.x), (float)h.y + ((float)m.y + (float)l.y),
                       (float)h.z + ((float)m.z + (float)l.z), (float)h.w + ((float)m.w + (float)l.w));
  }
  if (b16) {
    typedef __bf16 h4 __attribute__((ext_vector_type(4)));
    const h4 h = *(const h4*)((const __bf16*)p + i);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  return *(const float4*)(p + i);
}
__device__ __forceinline__ void st4e(float* p, size_t i, float4 v, int b16, size_t ps = 0) {
  if (b16 == 2) {
    typedef __bf16 h4 __attribute__((ext_vector_type(4)));
    __bf16* b = (__bf16*)p + i;
    const h4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    const float4 r1 = make_float4(v.x - (float)h.x, v.y - (float)h.y, v.z - (float)h.z, v.w - (float)h.w);
    const h4 m = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
    const h4 l = {(__bf16)(r1.x - (float)m.x), (__bf16)(r1.y - (float)m.y), (__bf16)(r1.z - (float)m.z), (__bf16)(r1.w - (float)m.w)};
    *(h4*)b = h; *(h4*)(b + ps) = m; *(h4*)(b + 2 * ps) = l;
    return;
  }
  if (b16) {
    typedef __bf16 h4 __attribute__((ext_vector_type(4)));
    const h4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *(h4*)((__bf16*)p + i) = h;
  } else {
    *(float4*)(p + i) = v;
  }
}

// y[b] = <x[b, :], w> / sigma + bias: one block per sample; the K loop runs four 16-byte loads per thread per trip
// (K = 32768 .. 131072: at one load per trip a 256-thread block per row moved 0.8 TB/s)
__global__ __launch_bounds__(1024) void gemv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ inv_scale,
                                                        float* __restrict__ y, int K, int x16, size_t ps,
                                                        const float* __restrict__ inv_scale1 = nullptr, int half = 0) {
  // inv_scale1 / half (paired pass): rows >= half are divided by the second sigma
  __shared__ float sh[16];
  if (inv_scale1 && (int)blockIdx.x >= half) inv_scale = inv_scale1;
  const size_t row = (size_t)blockIdx.x * K;
  const int stride = blockDim.x * 4;
  float s = 0.f;
  int k = threadIdx.x * 4;
  for (; k + 3 * stride < K; k += 4 * stride) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = ld4e(x, row + k + u * stride, x16, ps); b[u] = *(const float4*)(w + k + u * stride); }
#pragma unroll
    for (int u = 0; u < 4; ++u) s += a[u].x * b[u].x + a[u].y * b[u].y + a[u].z * b[u].z + a[u].w * b[u].w;
  }
  for (; k < K; k += stride) {
    const float4 a = ld4e(x, row + k, x16, ps), b = *(const float4*)(w + k);
    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float sc = inv_scale ? *inv_scale : 1.f;
    y[blockIdx.x] = s / sc + (bias ? bias[0] : 0.f);
  }
}
// dx[b, k] = dy[b] * w[k] / sigma * act'(prev_out[b, k]): grid (k quads, samples), 16-byte accesses, no index division
__global__ __launch_bounds__(256) void gemv_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                   const float* __restrict__ inv_scale, float* __restrict__ dx,
                                   const float* __restrict__ prev_out, int prev_act, float prev_slope,
                                   int B, int K, int x16, size_t ps_dx, size_t ps_prev,
                                   const float* __restrict__ inv_scale1 = nullptr, int half = 0) {
  const int b = blockIdx.y;
  if (inv_scale1 && b >= half) inv_scale = inv_scale1;        // paired pass: the second half-batch's sigma
  const float sc = inv_scale ? *inv_scale : 1.f;
  const float g = dy[b];
  for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < K; k += gridDim.x * blockDim.x * 4) {
    const float4 wv = *(const float4*)(w + k);
    float4 v = make_float4(g * (wv.x / sc), g * (wv.y / sc), g * (wv.z / sc), g * (wv.w / sc));
    const size_t i = (size_t)b * K + k;
    if (prev_out) {
      const float4 o = ld4e(prev_out, i, x16, ps_prev);
      v.x *= act_grad_from_out(o.x, prev_act, prev_slope); v.y *= act_grad_from_out(o.y, prev_act, prev_slope);
      v.z *= act_grad_from_out(o.z, prev_act, prev_slope); v.w *= act_grad_from_out(o.w, prev_act, prev_slope);
    }
    st4e(dx, i, v, x16, ps_dx);
  }
}
// dw of a three-plane x: 4 columns per thread, four samples in flight
__global__ __launch_bounds__(64) void gemv_bwd_dw3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ db, int B, int K, size_t ps) {
  // blockIdx.y = group (paired pass: one weight / bias gradient per half-batch): rows [g B, (g + 1) B), dw + g K, db + g
  x = (const float*)((const __bf16*)x + (size_t)blockIdx.y * B * K);
  dy += (size_t)blockIdx.y * B;
  if (dw) dw += (size_t)blockIdx.y * K;
  if (db) db += blockIdx.y;
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (k < K && dw) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int b = 0;
    for (; b + 3 < B; b += 4) {
      float4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = ld4e(x, (size_t)(b + u) * K + k, 2, ps);
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float g = dy[b + u]; s.x += g * r[u].x; s.y += g * r[u].y; s.z += g * r[u].z; s.w += g * r[u].w; }
    }
    for (; b < B; ++b) {
      const float4 r = ld4e(x, (size_t)b * K + k, 2, ps);
      const float g = dy[b];
      s.x += g * r.x; s.y += g * r.y; s.z += g * r.z; s.w += g * r.w;
    }
    *(float4*)(dw + k) = s;
  }
  if (db && blockIdx.x == 0 && threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dy[b];
    db[0] = a;
  }
}
// dw[k] = sum_b dy[b] x[b, k] (sample order: deterministic).  One 16-byte load per thread and sample (8 bf16 / 4 fp32
// columns), eight samples in flight, one wave per block so that a 131072-column head still fills the chip.
template <bool X16>
__global__ __launch_bounds__(64) void gemv_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int K) {
  constexpr int CPT = X16 ? 8 : 4;
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) * CPT;
  if (k < K && dw) {
    if (k + CPT <= K && (K % CPT) == 0) {
      float s[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) s[c] = 0.f;
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      auto ld = [&](int b) { return *(const u4*)((const char*)x + ((size_t)b * K + k) * (X16 ? 2 : 4)); };
      auto fma_row = [&](const u4 rv, float g) {
        const unsigned r[4] = {rv.x, rv.y, rv.z, rv.w};
        if constexpr (X16) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            s[2 * c] += g * __builtin_bit_cast(float, r[c] << 16);
            s[2 * c + 1] += g * __builtin_bit_cast(float, r[c] & 0xffff0000u);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) s[c] += g * __builtin_bit_cast(float, r[c]);
        }
      };
      int b = 0;
      for (; b + 7 < B; b += 8) {
        u4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = ld(b + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) fma_row(r[u], dy[b + u]);
      }
      for (; b < B; ++b) fma_row(ld(b), dy[b]);
#pragma unroll
      for (int c = 0; c < CPT; ++c) dw[k + c] = s[c];
    } else {                                  // ragged K: scalar columns
      for (int c = 0; c < CPT && k + c < K; ++c) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) {
          const size_t i = (size_t)b * K + k + c;
          a += dy[b] * (X16 ? (float)((const __bf16*)x)[i] : x[i]);
        }
        dw[k + c] = a;
      }
    }
  }
  if (db && blockIdx.x == 0 && threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dy[b];
    db[0] = a;
  }
}

// ---- losses --------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ float loss_term(int kind, float x, float y) {
  switch (kind) {
    case IPRGAN_LOSS_HINGE_REAL: return fmaxf(1.f - x, 0.f);
    case IPRGAN_LOSS_HINGE_FAKE: return fmaxf(1.f + x, 0.f);
    case IPRGAN_LOSS_NEG_MEAN: return -x;
    case IPRGAN_LOSS_BCE_ONES: return fmaxf(-x, 0.f) + softplus_neg_abs(x);
    case IPRGAN_LOSS_BCE_ZEROS: return fmaxf(x, 0.f) + softplus_neg_abs(x);
    case IPRGAN_LOSS_MSE_ONES: return (x - 1.f) * (x - 1.f);
    case IPRGAN_LOSS_MSE_ZEROS: return x * x;
    case IPRGAN_LOSS_MSE: return (x - y) * (x - y);
    case IPRGAN_LOSS_BCE_PM1: {       // ATen binary_cross_entropy: (t - 1) * max(log(1 - p), -100) - t * max(log p, -100)
      const float p = (x + 1.f) / 2.f, t = (y + 1.f) / 2.f;
      return (t - 1.f) * fmaxf(logf(1.f - p), -100.f) - t * fmaxf(logf(p), -100.f);
    }
    case IPRGAN_LOSS_KL_MEAN: return x * x / 2.f;
    case IPRGAN_LOSS_KL_LOGVAR: return (expf(x) - 1.f - x) / 2.f;
    // tools/loss.py:15-18 with normalized=True: both arguments go through (v + 1) / 2 first, rounded as there
    case IPRGAN_LOSS_MSE_DENORM: { const float d = (x + 1.f) / 2.f - (y + 1.f) / 2.f; return d * d; }
    case IPRGAN_LOSS_L1_DENORM: return fabsf((x + 1.f) / 2.f - (y + 1.f) / 2.f);
    default: return fabsf(x - y);
  }
}
__device__ __forceinline__ float loss_grad(int kind, float x, float y) {
  switch (kind) {
    case IPRGAN_LOSS_HINGE_REAL: return (1.f - x) > 0.f ? -1.f : 0.f;
    case IPRGAN_LOSS_HINGE_FAKE: return (1.f + x) > 0.f ? 1.f : 0.f;
    case IPRGAN_LOSS_NEG_MEAN: return -1.f;
    case IPRGAN_LOSS_BCE_ONES: return 1.f / (1.f + expf(-x)) - 1.f;
    case IPRGAN_LOSS_BCE_ZEROS: return 1.f / (1.f + expf(-x));
    case IPRGAN_LOSS_MSE_ONES: return 2.f * (x - 1.f);
    case IPRGAN_LOSS_MSE_ZEROS: return 2.f * x;
    case IPRGAN_LOSS_MSE: return 2.f * (x - y);
    case IPRGAN_LOSS_BCE_PM1: {       // ATen backward: (p - t) / max((1 - p) p, 1e-12); dp/dx = 1/2
      const float p = (x + 1.f) / 2.f, t = (y + 1.f) / 2.f;
      return (p - t) / fmaxf((1.f - p) * p, 1e-12f) * 0.5f;
    }
    case IPRGAN_LOSS_KL_MEAN: return x;
    case IPRGAN_LOSS_KL_LOGVAR: return (expf(x) - 1.f) / 2.f;
    case IPRGAN_LOSS_MSE_DENORM: return (x + 1.f) / 2.f - (y + 1.f) / 2.f;          // 2 d * dp/dx, dp/dx = 1/2
    case IPRGAN_LOSS_L1_DENORM: {
      const float d = (x + 1.f) / 2.f - (y + 1.f) / 2.f;
      return d > 0.f ? 0.5f : (d < 0.f ? -0.5f : 0.f);
    }
    default: { const float d = x - y; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
  }
}
#define LOSS_BLOCKS 256
__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           float* __restrict__ part, size_t n) {
  __shared__ float sh[16];
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += loss_term(kind, x[i], y ? y[i] : 0.f);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// n <= 256 (the logits of a DCGAN / SRGAN discriminator pass): the one block's sum IS the loss - no partial array, no second
// launch; the same additions in the same order as loss_partial_kernel + loss_final_kernel with one block (bit-identical)
__global__ __launch_bounds__(256) void loss_small_kernel(int kind, const float* __restrict__ x, const float* __restrict__ y,
                                                         float* __restrict__ loss, size_t n, float scale) {
  __shared__ float sh[16];
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) s += loss_term(kind, x[i], y ? y[i] : 0.f);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    float t = 0.f;
    t += s;
    *loss = t * scale;
  }
}
// Two mean losses over the two halves of ONE vector and their sum, in one launch each way: the paired discriminator pass hands
// D(real) and D(fake) over as one [2B] tensor, and the reference's LossD = LossR + LossF (models/dcgan.py:33-37) on slices of it
// cost ten launches per step (two partial + final pairs, an ATen add, two gradient kernels, and the zero-fill + copy + add of
// autograd's two slice gradients).  n <= 256 per half: the additions of loss_small_kernel per half, then one fp32 add -
// bit-identical to the separate launches.
__global__ __launch_bounds__(256) void loss_pair_small_kernel(int kind_a, int kind_b, const float* __restrict__ x, int n,
                                                              float scale, float* __restrict__ out3) {
  __shared__ float sh[16];
  float sa = 0.f, sb = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) sa += loss_term(kind_a, x[i], 0.f);
  sa = block_sum(sa, sh);
  for (int i = threadIdx.x; i < n; i += blockDim.x) sb += loss_term(kind_b, x[n + i], 0.f);
  sb = block_sum(sb, sh);
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    ta += sa; tb += sb;
    const float la = ta * scale, lb = tb * scale;
    out3[0] = la; out3[1] = lb; out3[2] = la + lb;
  }
}
__global__ void loss_pair_bwd_kernel(int kind_a, int kind_b, const float* __restrict__ x, const float* __restrict__ gscale,
                                     float* __restrict__ dx, int n, float inv_n) {
  const float g = (gscale ? *gscale : 1.f) * inv_n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x)
    dx[i] = g * loss_grad(i < n ? kind_a : kind_b, x[i], 0.f);
}
__global__ void loss_final_kernel(const float* __restrict__ part, int nb, float inv_n, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < nb; ++i) s += part[i];
    *loss = s * inv_n;
  }
}
__global__ void loss_bwd_kernel(int kind, const float* __restrict__ x, const float* __restrict__ y,
                                const float* __restrict__ gscale, float* __restrict__ dx, size_t n,
                                float inv_n) {
  const float g = (gscale ? *gscale : 1.f) * inv_n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = g * loss_grad(kind, x[i], y ? y[i] : 0.f);
}

// ---- VAE reparameterisation (networks/encoder.py:24-28): std = exp(logvar * 0.5); z = eps * std + mean ----
__global__ void reparam_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                   const float* __restrict__ eps, float* __restrict__ z, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    z[i] = eps[i] * expf(logvar[i] * 0.5f) + mean[i];
}
__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ logvar,
                                   const float* __restrict__ eps, float* __restrict__ dmean,
                                   float* __restrict__ dlogvar, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = dz[i];
    dmean[i] = g;
    dlogvar[i] = g * eps[i] * expf(logvar[i] * 0.5f) * 0.5f;
  }
}

// ---- storage kinds of the elementwise kernels below: st = 0 fp32, st = 2 three bf16 planes ps elements apart (IPRGAN_ST_X3,
// include/iprgan.h: x = h + (m + l) exactly; split once per element when stored).  e = element index of 4 consecutive values.
typedef float ew_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 ew_b4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ew_f4 ew_wide(const ew_b4 h) {
  const ew_f4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
  return v;
}
__device__ __forceinline__ ew_b4 ew_narrow(const ew_f4& v) {
  const ew_b4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return h;
}
__device__ __forceinline__ ew_f4 ew_ld(const float* base, size_t e, int st, size_t ps) {
  if (st == 2) {
    const __bf16* b = (const __bf16*)base + e;
    return ew_wide(*(const ew_b4*)b) + (ew_wide(*(const ew_b4*)(b + ps)) + ew_wide(*(const ew_b4*)(b + 2 * ps)));
  }
  return *(const ew_f4*)(base + e);
}
__device__ __forceinline__ void ew_st(float* base, size_t e, const ew_f4& v, int st, size_t ps) {
  if (st == 2) {
    __bf16* b = (__bf16*)base + e;
    const ew_b4 h = ew_narrow(v);
    const ew_f4 r1 = v - ew_wide(h);
    const ew_b4 m = ew_narrow(r1);
    *(ew_b4*)b = h;
    *(ew_b4*)(b + ps) = m;
    *(ew_b4*)(b + 2 * ps) = ew_narrow(r1 - ew_wide(m));
  } else {
    *(ew_f4*)(base + e) = v;
  }
}

// ---- PReLU with ONE learnable slope (nn.PReLU(), networks/sr_resnet.py:7,14,43) ---------------------
__global__ void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                 float* __restrict__ y, size_t n, int st) {
  const float a = *alpha;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const size_t n4 = (st == 2 || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0) ? n / 4 : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (size_t i = i0; i < n4; i += stride) {              // 16-byte accesses over the aligned bulk
    const f4 v = ew_ld(x, i * 4, st, n);
    f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? v[k] : a * v[k];
    ew_st(y, i * 4, o, st, n);
  }
  if (st == 2) return;
  for (size_t i = n4 * 4 + i0; i < n; i += stride) {
    const float v = x[i];
    y[i] = v > 0.f ? v : a * v;
  }
}
// dx = dy * (x>0 ? 1 : alpha); part[block] = sum dy*x*[x<=0]
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ alpha, float* __restrict__ dx,
                                                        float* __restrict__ part, size_t n, int st) {
  __shared__ float sh[16];
  const float a = *alpha;
  // the slope gradient is ONE number summed over the whole tensor, with terms of both signs: this thread's running sum
  // and the final sum over the block partials are kept in double (against the float64 evaluation the fp32 chain was
  // 2e-5 off where torch's pairwise reduction is 2e-7; tests/test_gpu_models.py::test_net_accuracy_against_float64)
  double s = 0.0;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const size_t n4 = (st == 2 || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(dx)) & 15) == 0)
                        ? n / 4 : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (size_t i = i0; i < n4; i += stride) {              // 16-byte accesses (scalar ones ran at 0.8 TB/s)
    const f4 v = ew_ld(x, i * 4, st, n), g = ew_ld(dy, i * 4, st, n);
    f4 o;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = v[k] > 0.f ? g[k] : a * g[k];
      q += v[k] > 0.f ? 0.f : g[k] * v[k];
    }
    s += (double)q;
    ew_st(dx, i * 4, o, st, n);
  }
  for (size_t i = n4 * 4 + i0; i < n && st != 2; i += stride) {
    const float v = x[i], g = dy[i];
    dx[i] = v > 0.f ? g : a * g;
    s += v > 0.f ? 0.0 : (double)(g * v);
  }
  const float sb = block_sum((float)s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = sb;
}
// one wave: lane l adds partials l, l + 64, ... in double, then a fixed butterfly over the 64 lanes (deterministic; the
// single-thread walk over up to LOSS_BLOCKS partials took 12 us per PReLU layer)
__global__ void sum_partials_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  if (blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) s += (double)part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) *out = (float)s;
}

// ---- PixelShuffle(2) on NHWC: out[b,2h+i,2w+j,c] = in[b,h,w,c*4+i*2+j] (networks/sr_resnet.py:42) ----
// inverse=1 runs the permutation backwards (the gradient).
__global__ void pixel_shuffle2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H,
                                      int W, int C, int inverse) {
  const size_t total = (size_t)B * H * W * C * 4;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total;
       o += (size_t)gridDim.x * blockDim.x) {
    // o indexes the shuffled tensor [B,2H,2W,C]
    const int c = (int)(o % C);
    size_t t = o / C;
    const int ox = (int)(t % (2 * W));
    t /= 2 * W;
    const int oy = (int)(t % (2 * H));
    const size_t b = t / (2 * H);
    const size_t in = ((b * H + (oy >> 1)) * W + (ox >> 1)) * (size_t)(4 * C) + c * 4 + (oy & 1) * 2 + (ox & 1);
    if (inverse) dst[in] = src[o]; else dst[o] = src[in];
  }
}

// ---- PixelShuffle(2) + PReLU in one pass (the upsampling blocks of SRResNet, networks/sr_resnet.py:39-45: conv ->
// PixelShuffle -> PReLU; the PReLU has ONE slope, so it commutes with the permutation).  A thread owns pixel (b, h, w)
// and FOUR consecutive output channels c0..c0+3: the 16 source values x[b,h,w,4 c0 .. 4 c0 + 15] are four contiguous
// 16-byte loads, and each of the four sub-pixels (i, j) receives one 16-byte store {x[4 (c0+k) + 2 i + j]}, k = 0..3.
// Backward: the same walk reads dy at the four sub-pixels, writes dx contiguously and accumulates the slope gradient
// (double per thread, as prelu_bwd_kernel).  C % 4 == 0.
typedef float ps_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ps2_prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                            float* __restrict__ y, int B, int H, int W, int C, int st) {
  const float a = *alpha;
  const int cg = C / 4;
  const size_t total = (size_t)B * H * W * cg, ps = total * 16;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    size_t p = t / cg;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const size_t b = p / H;
    const size_t so = ((b * H + h) * W + w) * (size_t)(4 * C) + (size_t)g * 16;
    ps_f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ew_ld(x, so + 4 * k, st, ps);
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
      ps_f4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float e = v[k][ij]; o[k] = e > 0.f ? e : a * e; }
      ew_st(y, ((b * 2 * H + 2 * h + (ij >> 1)) * (size_t)(2 * W) + 2 * w + (ij & 1)) * (size_t)C + (size_t)g * 4, o, st, ps);
    }
  }
}
__global__ __launch_bounds__(256) void ps2_prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ alpha, float* __restrict__ dx,
                                                            float* __restrict__ part, int B, int H, int W, int C, int st) {
  __shared__ float sh[16];
  const float a = *alpha;
  const int cg = C / 4;
  const size_t total = (size_t)B * H * W * cg, ps = total * 16;
  double s = 0.0;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    size_t p = t / cg;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const size_t b = p / H;
    const size_t xo = ((b * H + h) * W + w) * (size_t)(4 * C) + (size_t)g * 16;
    ps_f4 v[4], d[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ew_ld(x, xo + 4 * k, st, ps);
#pragma unroll
    for (int ij = 0; ij < 4; ++ij)
      d[ij] = ew_ld(dy, ((b * 2 * H + 2 * h + (ij >> 1)) * (size_t)(2 * W) + 2 * w + (ij & 1)) * (size_t)C + (size_t)g * 4, st, ps);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int ij = 0; ij < 4; ++ij) {
        const float e = v[k][ij], gr = d[ij][k];
        o[k][ij] = e > 0.f ? gr : a * gr;
        q += e > 0.f ? 0.f : gr * e;
      }
    s += (double)q;
#pragma unroll
    for (int k = 0; k < 4; ++k) ew_st(dx, xo + 4 * k, o[k], st, ps);
  }
  const float sb = block_sum((float)s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = sb;
}

// ---- MaxPool2d(2,2) on NHWC (VGG19 features, networks/vgg.py) ---------------------------------------------
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                    int C) {
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)B * OH * OW * C;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total;
       o += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(o % C);
    size_t t = o / C;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const size_t b = t / OH;
    const float* p = x + ((b * H + 2 * oy) * W + 2 * ox) * (size_t)C + c;
    const float v0 = p[0], v1 = p[C], v2 = p[(size_t)W * C], v3 = p[(size_t)W * C + C];
    y[o] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
  }
}
// gradient goes to the FIRST maximum in window order (row-major), like aten::max_pool2d_with_indices
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    float* __restrict__ dx, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)B * H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int ix = (int)(t % W);
    t /= W;
    const int iy = (int)(t % H);
    const size_t b = t / H;
    const int oy = iy >> 1, ox = ix >> 1;
    float g = 0.f;
    if (oy < OH && ox < OW) {
      const float* p = x + ((b * H + 2 * oy) * W + 2 * ox) * (size_t)C + c;
      const float v[4] = {p[0], p[C], p[(size_t)W * C], p[(size_t)W * C + C]};
      int arg = 0;
      float m = v[0];
#pragma unroll
      for (int k = 1; k < 4; ++k) if (v[k] > m) { m = v[k]; arg = k; }
      if (arg == (iy & 1) * 2 + (ix & 1)) g = dy[((b * OH + oy) * OW + ox) * (size_t)C + c];
    }
    dx[i] = g;
  }
}

// the same pool on even maps with 4 channels per thread (16-byte accesses; fp32 or three-plane tensors): forward, and a
// backward that visits every 2x2 window once (the scalar kernels above re-read the window for each of its four pixels)
__global__ __launch_bounds__(256) void maxpool2_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                            int W, int C, int st) {
  const int OH = H / 2, OW = W / 2, cg = C / 4;
  const size_t total = (size_t)B * OH * OW * cg, psx = (size_t)B * H * W * C, psy = total * 4;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    size_t p = t / cg;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const size_t b = p / OH;
    const size_t i0 = ((b * H + 2 * oy) * W + 2 * ox) * (size_t)C + (size_t)g * 4;
    const ew_f4 v0 = ew_ld(x, i0, st, psx), v1 = ew_ld(x, i0 + C, st, psx), v2 = ew_ld(x, i0 + (size_t)W * C, st, psx),
                v3 = ew_ld(x, i0 + (size_t)W * C + C, st, psx);
    ew_f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
    ew_st(y, t * 4, o, st, psy);
  }
}
__global__ __launch_bounds__(256) void maxpool2_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int B, int H, int W, int C, int st) {
  const int OH = H / 2, OW = W / 2, cg = C / 4;
  const size_t total = (size_t)B * OH * OW * cg, psx = (size_t)B * H * W * C, psy = total * 4;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    size_t p = t / cg;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const size_t b = p / OH;
    const size_t i0 = ((b * H + 2 * oy) * W + 2 * ox) * (size_t)C + (size_t)g * 4;
    const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    ew_f4 v[4], o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ew_ld(x, i0 + off[q], st, psx);
    const ew_f4 gr = ew_ld(dy, t * 4, st, psy);
#pragma unroll
    for (int k = 0; k < 4; ++k) {                     // the FIRST maximum in window order takes the gradient
      int arg = 0;
      float m = v[0][k];
#pragma unroll
      for (int q = 1; q < 4; ++q) if (v[q][k] > m) { m = v[q][k]; arg = q; }
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q][k] = arg == q ? gr[k] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ew_st(dx, i0 + off[q], o[q], st, psx);
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           size_t n, int st) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const size_t n4 = (st == 2 || ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(out)) & 15) == 0)
                        ? n / 4 : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (size_t i = i0; i < n4; i += stride) ew_st(out, i * 4, ew_ld(a, i * 4, st, n) + ew_ld(b, i * 4, st, n), st, n);
  for (size_t i = n4 * 4 + i0; i < n && st != 2; i += stride) out[i] = a[i] + b[i];
}

// ---- ImagePool's swap branch (models/util.py:27-34) with the decisions read from DEVICE memory, so that a captured step
// replays with new draws: image i changes places with history slot index[i] when take[i] != 0.  The slots of one call are
// distinct (a prefix of a permutation), so the swaps are independent: blockIdx.y = image.
// a few integers from the host into device memory AS KERNEL ARGUMENTS: read at call time, ordered on the stream like
// any launch, no host buffer that a later call could overwrite under a copy still in flight
struct IntsArg { static constexpr int N = 64; int v[N]; };
__global__ void write_ints_kernel(int* __restrict__ dst, IntsArg t, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = t.v[threadIdx.x];
}
__global__ void pool_swap_kernel(float* __restrict__ images, float* __restrict__ pool, const int* __restrict__ index,
                                 const int* __restrict__ take, size_t n) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int i = blockIdx.y;
  if (!take[i]) return;
  float* a = images + (size_t)i * n;
  float* b = pool + (size_t)index[i] * n;
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t n4 = (((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b)) & 15) == 0) ? n / 4 : 0;
  for (size_t e = i0; e < n4; e += stride) {
    const f4 va = reinterpret_cast<f4*>(a)[e], vb = reinterpret_cast<f4*>(b)[e];
    reinterpret_cast<f4*>(a)[e] = vb;
    reinterpret_cast<f4*>(b)[e] = va;
  }
  for (size_t e = n4 * 4 + i0; e < n; e += stride) { const float va = a[e]; a[e] = b[e]; b[e] = va; }
}

// ---- ReflectionPad2d(p) backward: fold the gradient of the padded image back (resnet_generator.py:6,33,43,47)
__device__ __forceinline__ int refl_pre(int i, int n, int p, int* out) {
  // padded-coordinate preimages (0-based in the padded image) of interior index i
  int k = 0;
  out[k++] = i + p;
  if (i >= 1 && i <= p) out[k++] = p - i;
  if (i <= n - 2 && i >= n - 1 - p) out[k++] = p + 2 * (n - 1) - i;
  return k;
}
// one thread per pixel and 4 channels (16-byte accesses), 32-bit multiply-shift index arithmetic
typedef float rf_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void reflect_fold_kernel(const rf_f4* __restrict__ dxp, rf_f4* __restrict__ dx,
                                                           const rf_f4* __restrict__ prev_out, int prev_act,
                                                           float prev_slope, const rf_f4* __restrict__ residual,
                                                           unsigned total4, int H, int W, int C4n,
                                                           int p, FastDiv d_c4n, FastDiv d_w, FastDiv d_h, int okind, size_t ps) {
  // okind 2: dx and residual are three-plane tensors (plane stride ps elements), prev_out is the h plane of one (same sign)
  typedef __bf16 rf_b4 __attribute__((ext_vector_type(4)));
  auto ldb = [](const void* base, size_t e) {
    const rf_b4 h = *(const rf_b4*)((const __bf16*)base + e);
    const rf_f4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return v;
  };
  const int HP = H + 2 * p, WP = W + 2 * p;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const unsigned pix = fdiv(i, d_c4n), c = i - pix * (unsigned)C4n;
    const unsigned row = fdiv(pix, d_w);
    const int x = (int)(pix - row * (unsigned)W);
    const unsigned b = fdiv(row, d_h);
    const int y = (int)(row - b * (unsigned)H);
    int ys[3], xs[3];
    const int ny = refl_pre(y, H, p, ys), nx = refl_pre(x, W, p, xs);
    rf_f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
      for (int q = 0; q < nx; ++q) s += dxp[((size_t)(b * HP + ys[a]) * WP + xs[q]) * C4n + c];
    if (prev_out) {
      const rf_f4 o = okind == 2 ? ldb(prev_out, (size_t)i * 4) : prev_out[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] *= act_grad_from_out(o[k], prev_act, prev_slope);
    }
    if (okind == 2) {
      const size_t e = (size_t)i * 4;
      if (residual) s += ldb(residual, e) + (ldb(residual, e + ps) + ldb(residual, e + 2 * ps));
      const rf_b4 h = {(__bf16)s.x, (__bf16)s.y, (__bf16)s.z, (__bf16)s.w};
      const rf_f4 hf = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
      const rf_f4 r1 = s - hf;
      const rf_b4 m = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
      const rf_f4 mf = {(float)m.x, (float)m.y, (float)m.z, (float)m.w};
      const rf_f4 r2 = r1 - mf;
      const rf_b4 l = {(__bf16)r2.x, (__bf16)r2.y, (__bf16)r2.z, (__bf16)r2.w};
      __bf16* o = (__bf16*)dx + e;
      *(rf_b4*)o = h; *(rf_b4*)(o + ps) = m; *(rf_b4*)(o + 2 * ps) = l;
      continue;
    }
    if (residual) s += residual[i];
    dx[i] = s;
  }
}

// ---- ReflectionPad2d(p) backward without the padded grid (round 5; conv_igemm.hip: conv_bwd_data_reflect_direct) ----------
// The interior of the gradient is an ordinary zero-padded backward-data pass straight into dx (fused derivative, residual,
// three-plane split: the full-speed tile kernels, and a tile count that is not inflated by (H + 2p)(W + 2p) / HW - 274 tiles
// instead of 256 on 256 CUs made the batch-8 layers of CycleGAN 1.5x slower than their forward).  What the reflection adds
// lives on the four border STRIPS of the padded grid (rows / columns outside the image: 6 % of the positions), computed by one
// small launch into `dxp`; this kernel adds every strip position to the interior pixel it mirrors:
//   dx[y][x] += act'(prev[y][x]) * sum over the preimages (yq, xq) != (y, x) of dxp[yq][xq]
// for the "ring" pixels only (rows 1..p and H-1-p..H-2, columns likewise): 2p (W + H - 2p) pixels per image.
__global__ __launch_bounds__(256) void reflect_ring_fix_kernel(const rf_f4* __restrict__ dxp, rf_f4* __restrict__ dx,
                                                               const rf_f4* __restrict__ prev_out, int prev_act, float prev_slope,
                                                               unsigned total4, int H, int W, int C4n, int p, int ring,
                                                               FastDiv d_c4n, FastDiv d_ring, FastDiv d_w, FastDiv d_2p,
                                                               int okind, size_t ps) {
  typedef __bf16 rf_b4 __attribute__((ext_vector_type(4)));
  auto ldb = [](const void* base, size_t e) {
    const rf_b4 h = *(const rf_b4*)((const __bf16*)base + e);
    const rf_f4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return v;
  };
  const int HP = H + 2 * p, WP = W + 2 * p;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const unsigned rp = fdiv(i, d_c4n), c = i - rp * (unsigned)C4n;
    const unsigned b = fdiv(rp, d_ring);
    const int r = (int)(rp - b * (unsigned)ring);
    int y, x;
    if (r < 2 * p * W) {                       // the 2p ring ROWS, whole
      const int iy = (int)fdiv((unsigned)r, d_w);
      x = r - iy * W;
      y = iy < p ? 1 + iy : H - 1 - p + (iy - p);
    } else {                                   // the other H - 2p rows: their 2p ring COLUMNS
      const int r2 = r - 2 * p * W, j = (int)fdiv((unsigned)r2, d_2p), ix = r2 - j * 2 * p;
      y = j == 0 ? 0 : (j == H - 2 * p - 1 ? H - 1 : p + j);
      x = ix < p ? 1 + ix : W - 1 - p + (ix - p);
    }
    int ys[3], xs[3];
    const int ny = refl_pre(y, H, p, ys), nx = refl_pre(x, W, p, xs);
    rf_f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
      for (int q = 0; q < nx; ++q)
        if (a | q) s += dxp[((size_t)((int)b * HP + ys[a]) * WP + xs[q]) * C4n + c];      // (a, q) = (0, 0): the pixel itself
    const size_t e = ((size_t)((int)b * H + y) * W + x) * C4n * 4 + (size_t)c * 4;
    if (prev_out) {
      const rf_f4 o = okind == 2 ? ldb(prev_out, e) : prev_out[e / 4];
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] *= act_grad_from_out(o[k], prev_act, prev_slope);
    }
    if (okind == 2) {
      s += ldb(dx, e) + (ldb(dx, e + ps) + ldb(dx, e + 2 * ps));
      const rf_b4 h = {(__bf16)s.x, (__bf16)s.y, (__bf16)s.z, (__bf16)s.w};
      const rf_f4 hf = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
      const rf_f4 r1 = s - hf;
      const rf_b4 m = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
      const rf_f4 mf = {(float)m.x, (float)m.y, (float)m.z, (float)m.w};
      const rf_f4 r2 = r1 - mf;
      const rf_b4 l = {(__bf16)r2.x, (__bf16)r2.y, (__bf16)r2.z, (__bf16)r2.w};
      __bf16* o = (__bf16*)dx + e;
      *(rf_b4*)o = h; *(rf_b4*)(o + ps) = m; *(rf_b4*)(o + 2 * ps) = l;
    } else {
      dx[e / 4] += s;
    }
  }
}

// ---- sign loss / BER (multi-tensor: pointer table travels in the kernel arguments) ----------
#define SIGN_MAX_LAYERS 64
struct SignTable {
  const float* gamma[SIGN_MAX_LAYERS];
  const float* sign[SIGN_MAX_LAYERS];
  float* dgamma[SIGN_MAX_LAYERS];
  int size[SIGN_MAX_LAYERS];
  int nlayer;
};
__global__ __launch_bounds__(256) void sign_loss_fwd_kernel(const SignTable t, float gamma0,
                                                            float* __restrict__ loss, int accumulate) {
  __shared__ float sh[16];
  float total = 0.f;
  for (int l = 0; l < t.nlayer; ++l) {
    float s = 0.f;
    for (int i = threadIdx.x; i < t.size[l]; i += blockDim.x)
      s += fmaxf(gamma0 - t.gamma[l][i] * t.sign[l][i], 0.f);
    total += block_sum(s, sh) / (float)t.size[l];
  }
  if (threadIdx.x == 0) *loss = accumulate ? *loss + total : total;
}
__global__ void sign_loss_bwd_kernel(const SignTable t, float gamma0, const float* __restrict__ gscale,
                                     float beta) {
  const int l = blockIdx.x;
  const float g = (gscale ? *gscale : 1.f) / (float)t.size[l];
  for (int i = threadIdx.x; i < t.size[l]; i += blockDim.x) {
    const float b = t.sign[l][i];
    const float d = (gamma0 - t.gamma[l][i] * b) > 0.f ? -b * g : 0.f;
    t.dgamma[l][i] = beta != 0.f ? beta * t.dgamma[l][i] + d : d;
  }
}
__global__ __launch_bounds__(256) void sign_ber_kernel(const SignTable t, long long* __restrict__ counts,
                                                       int accumulate) {
  __shared__ int sh[4];
  int err = 0, tot = 0;
  for (int l = 0; l < t.nlayer; ++l) {
    for (int i = threadIdx.x; i < t.size[l]; i += blockDim.x) {
      const float g = t.gamma[l][i], b = t.sign[l][i];
      const bool same = (g > 0.f && b > 0.f) || (g < 0.f && b < 0.f);
      err += same ? 0 : 1;
    }
    tot += t.size[l];
  }
  // exact integer reduction: wave ballot-free shuffle sum, then 4 waves through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) err += __shfl_xor(err, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = err;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long e = (long long)sh[0] + sh[1] + sh[2] + sh[3];
    counts[0] = accumulate ? counts[0] + e : e;
    counts[1] = accumulate ? counts[1] + tot : tot;
  }
}

// ---- Adam (multi-tensor) ---------------------------------------------------------------------
#define ADAM_MAX_TENSORS 48
struct AdamTable {
  float* p[ADAM_MAX_TENSORS];
  const float* g[ADAM_MAX_TENSORS];
  float* m[ADAM_MAX_TENSORS];
  float* v[ADAM_MAX_TENSORS];
  long long n[ADAM_MAX_TENSORS];
};
// step count on the device (iprgan_adam_step_dev: a captured HIP graph replays the same kernel arguments every step, so
// the bias corrections cannot be host-computed arguments): one thread advances the counter and writes
// coef[0] = lr / (1 - beta1^t), coef[1] = sqrt(1 - beta2^t), computed in double like the host path
__global__ void adam_prep_kernel(int* __restrict__ step, double lr, double beta1, double beta2, float* __restrict__ coef) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int t = *step + 1;
    *step = t;
    coef[0] = (float)(lr / (1.0 - pow(beta1, (double)t)));
    coef[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
  }
}
__global__ __launch_bounds__(256) void adam_kernel(const AdamTable t, float omb1, float beta2, float omb2,
                                                   float eps, float weight_decay, float step_size,
                                                   float bc2_sqrt, float grad_scale, const float* __restrict__ coef) {
  if (coef) { step_size = coef[0]; bc2_sqrt = coef[1]; }
  const int ti = blockIdx.y;
  const long long n = t.n[ti];
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  // 16-byte accesses where the four arrays allow it (the scalar loop moved 2.7 TB/s on the 19 M-element tensors of D96
  // and of the DCGAN-128 generator; the arithmetic per element is unchanged)
  typedef float f4 __attribute__((ext_vector_type(4)));
  long long n4 = 0;
  if (((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v)) & 15) == 0)
    n4 = n / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f4 g4 = ((const f4*)g)[i], p4 = ((const f4*)p)[i], m4 = ((const f4*)m)[i], v4 = ((const f4*)v)[i];
    f4 po, mo, vo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gi = g4[k] * grad_scale;
      if (weight_decay != 0.f) gi += weight_decay * p4[k];
      const float mi = m4[k] + (gi - m4[k]) * omb1;
      const float vi = v4[k] * beta2 + omb2 * gi * gi;
      mo[k] = mi; vo[k] = vi;
      po[k] = p4[k] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
    ((f4*)m)[i] = mo; ((f4*)v)[i] = vo; ((f4*)p)[i] = po;
  }
  for (long long i = n4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * grad_scale;       // 1/world of the summed data-parallel gradient (1 on a single GPU)
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = m[i] + (gi - m[i]) * omb1;               // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + omb2 * gi * gi;           // mul_(beta2).addcmul_(g, g, 1-beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

static inline int grid_for(size_t n, int cap) {
  size_t b = cdivz(n, 256);
  if (b < 1) b = 1;
  return (int)(b < (size_t)cap ? b : (size_t)cap);
}

}  // namespace iprgan

using namespace iprgan;

extern "C" {

const char* iprgan_last_error(void) { return g_err; }
int iprgan_version(void) { return IPRGAN_VERSION; }

int iprgan_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
  const int Cs = (C + 3) & ~3;
  const size_t n = (size_t)B * H * W * Cs;
  if (!n) return 0;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, B, C, H * W, Cs);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
  const int Cs = (C + 3) & ~3;
  const size_t n = (size_t)B * H * W * C;
  if (!n) return 0;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, B, C, H * W, Cs);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_permute_021(const float* src, float* dst, int A, int Bd, int K, float beta, void* stream) {
  const size_t n = (size_t)A * Bd * K;
  if (!n) return 0;
  hipLaunchKernelGGL(permute_021_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, A, Bd, K, beta);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_fill(float* p, float v, size_t n, void* stream) {
  if (!n) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_axpy(float* y, const float* x, float a, size_t n, void* stream) {
  if (!n) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_axpy_multi(float* const* y, const float* const* x, const long long* sizes, int n, float a, void* stream) {
  for (int b = 0; b < n; b += AXPY_MAX_TENSORS) {
    AxpyTable t;
    memset(&t, 0, sizeof(t));
    int cnt = n - b;
    if (cnt > AXPY_MAX_TENSORS) cnt = AXPY_MAX_TENSORS;
    long long maxn = 0;
    for (int i = 0; i < cnt; ++i) {
      t.y[i] = y[b + i]; t.x[i] = x[b + i]; t.n[i] = sizes[b + i];
      if (t.n[i] > maxn) maxn = t.n[i];
    }
    if (maxn == 0) continue;
    hipLaunchKernelGGL(axpy_multi_kernel, dim3(grid_for((size_t)maxn, 256), cnt), dim3(256), 0, (hipStream_t)stream, t, a);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

int iprgan_cast(const float* src, float* dst, size_t n, int src_bf16, int dst_bf16, void* stream) {
  if (!n) return 0;
  if (src_bf16 == 2 || dst_bf16 == 2) {        // fp32 <-> three planes (contiguous: plane stride n); conv_x3.hip
    IPR_CHECK((src_bf16 == 2 && dst_bf16 == 0) || (src_bf16 == 0 && dst_bf16 == 2), "cast: three planes convert from / to fp32 only");
    return cast_planes(src, dst, n, n, dst_bf16 == 2, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, n, src_bf16, dst_bf16);
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_cast_planes(const void* src, void* dst, size_t n, size_t pstride, int to_planes, void* stream) {
  IPR_CHECK(pstride >= n, "cast_planes: plane stride %zu smaller than the tensor (%zu elements)", pstride, n);
  return cast_planes(src, dst, n, pstride, to_planes != 0, (hipStream_t)stream);
}

int iprgan_gemv_fwd(const float* x, const float* w, const float* bias, const float* inv_scale, float* y,
                    int B, int K, int x_bf16, size_t x_pstride, void* stream) {
  IPR_CHECK(K % 4 == 0, "gemv_fwd: K=%d must be a multiple of 4", K);
  if (B == 0) return 0;
  const size_t ps = x_pstride ? x_pstride : (size_t)B * K;
  hipLaunchKernelGGL(gemv_fwd_kernel, dim3(B), dim3(K >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, w, bias, inv_scale, y, K, x_bf16, ps);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_gemv_fwd_pair(const float* x, const float* w, const float* bias, const float* inv_scale0, const float* inv_scale1,
                         float* y, int B, int K, int x_bf16, size_t x_pstride, void* stream) {
  IPR_CHECK(K % 4 == 0 && (B & 1) == 0 && inv_scale0 && inv_scale1, "gemv_fwd_pair: K=%d, B=%d (even), two sigmas", K, B);
  if (B == 0) return 0;
  const size_t ps = x_pstride ? x_pstride : (size_t)B * K;
  hipLaunchKernelGGL(gemv_fwd_kernel, dim3(B), dim3(K >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, x, w, bias, inv_scale0, y, K,
                     x_bf16, ps, inv_scale1, B / 2);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_gemv_bwd_pair(const float* x, const float* w, const float* dy, const float* inv_scale0, const float* inv_scale1,
                         float* dx, float* dw2, float* db2, const float* prev_out, int prev_act, float prev_slope, int B, int K,
                         int x_bf16, size_t x_pstride, size_t dx_pstride, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(K % 4 == 0 && (B & 1) == 0 && inv_scale0 && inv_scale1, "gemv_bwd_pair: K=%d, B=%d (even), two sigmas", K, B);
  IPR_CHECK(x_bf16 == 2 || !(dw2 || db2), "gemv_bwd_pair: the per-half weight gradients are built for three-plane activations");
  if (B == 0) return 0;
  const size_t ps = x_pstride ? x_pstride : (size_t)B * K, ps_dx = dx_pstride ? dx_pstride : (size_t)B * K;
  if (dx) {
    const int gx = cdiv(K / 4, 256) < 64 ? cdiv(K / 4, 256) : 64;
    hipLaunchKernelGGL(gemv_bwd_dx_kernel, dim3(gx, B), dim3(256), 0, st, w, dy, inv_scale0, dx, prev_out, prev_act, prev_slope, B, K,
                       x_bf16, ps_dx, ps, inv_scale1, B / 2);
    IPR_LAUNCH_CHECK();
  }
  if (dw2 || db2) {          // dw2 [2][K], db2 [2]: gradient of the first / second half-batch
    hipLaunchKernelGGL(gemv_bwd_dw3_kernel, dim3(cdiv(K, 256), 2), dim3(64), 0, st, x, dy, dw2, db2, B / 2, K, ps);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}
int iprgan_gemv_bwd(const float* x, const float* w, const float* dy, const float* inv_scale, float* dx,
                    float* dw, float* db, const float* prev_out, int prev_act, float prev_slope, int B,
                    int K, int x_bf16, size_t x_pstride, size_t dx_pstride, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return 0;
  const size_t ps = x_pstride ? x_pstride : (size_t)B * K, ps_dx = dx_pstride ? dx_pstride : (size_t)B * K;
  if (dx) {
    IPR_CHECK(K % 4 == 0, "gemv_bwd: K=%d must be a multiple of 4", K);
    const int gx = cdiv(K / 4, 256) < 64 ? cdiv(K / 4, 256) : 64;
    hipLaunchKernelGGL(gemv_bwd_dx_kernel, dim3(gx, B), dim3(256), 0, st, w, dy,
                       inv_scale, dx, prev_out, prev_act, prev_slope, B, K, x_bf16, ps_dx, ps);
    IPR_LAUNCH_CHECK();
  }
  if (dw || db) {
    if (x_bf16 == 2) {
      IPR_CHECK(K % 4 == 0, "gemv_bwd: K=%d must be a multiple of 4", K);
      hipLaunchKernelGGL(gemv_bwd_dw3_kernel, dim3(cdiv(K, 256)), dim3(64), 0, st, x, dy, dw, db, B, K, ps);
    } else
    if (x_bf16) hipLaunchKernelGGL(gemv_bwd_dw_kernel<true>, dim3(cdiv(K, 512)), dim3(64), 0, st, x, dy, dw, db, B, K);
    else hipLaunchKernelGGL(gemv_bwd_dw_kernel<false>, dim3(cdiv(K, 256)), dim3(64), 0, st, x, dy, dw, db, B, K);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}


#define IPR_ST_CHECK(st, n, what) \
  IPR_CHECK((st) == 0 || ((st) == 2 && ((n) % 4) == 0), what ": storage kind %d (fp32 or three planes; three planes need n %% 4 == 0)", (int)(st))
int iprgan_prelu_fwd(const float* x, const float* alpha, float* y, size_t n, int act_st, void* stream) {
  if (!n) return 0;
  IPR_ST_CHECK(act_st, n, "prelu_fwd");
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, alpha, y, n, act_st);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_prelu_bwd(const float* x, const float* dy, const float* alpha, float* dx, float* dalpha, float* ws,
                     size_t n, int act_st, void* stream) {
  if (!n) return 0;
  IPR_ST_CHECK(act_st, n, "prelu_bwd");
  const int nb = grid_for(n, LOSS_BLOCKS);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, dy, alpha, dx, ws, n, act_st);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, dalpha);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_pixel_shuffle2(const float* src, float* dst, int B, int H, int W, int C, int inverse, void* stream) {
  const size_t n = (size_t)B * H * W * C * 4;
  if (!n) return 0;
  hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     B, H, W, C, inverse);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_pixel_shuffle2_prelu_fwd(const float* x, const float* alpha, float* y, int B, int H, int W, int C, int act_st,
                                    void* stream) {
  IPR_CHECK(C > 0 && (C % 4) == 0, "pixel_shuffle2_prelu: %d output channels (a multiple of 4 is required)", C);
  const size_t n = (size_t)B * H * W * (C / 4);
  if (!n) return 0;
  IPR_ST_CHECK(act_st, 4, "pixel_shuffle2_prelu_fwd");
  hipLaunchKernelGGL(ps2_prelu_fwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, alpha, y, B, H, W, C, act_st);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_pixel_shuffle2_prelu_bwd(const float* x, const float* dy, const float* alpha, float* dx, float* dalpha, float* ws,
                                    int B, int H, int W, int C, int act_st, void* stream) {
  IPR_CHECK(C > 0 && (C % 4) == 0, "pixel_shuffle2_prelu: %d output channels (a multiple of 4 is required)", C);
  const size_t n = (size_t)B * H * W * (C / 4);
  if (!n) return 0;
  IPR_ST_CHECK(act_st, 4, "pixel_shuffle2_prelu_bwd");
  const int nb = grid_for(n, LOSS_BLOCKS);
  hipLaunchKernelGGL(ps2_prelu_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, dy, alpha, dx, ws, B, H, W, C, act_st);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, dalpha);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, int act_st, void* stream) {
  const size_t n = (size_t)B * (H / 2) * (W / 2) * C;
  if (!n) return 0;
  const bool vec = (C % 4) == 0 && (H % 2) == 0 && (W % 2) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  IPR_CHECK(act_st == 0 || (act_st == 2 && vec), "maxpool2_fwd: three-plane tensors need even maps and C %% 4 == 0");
  if (vec) {
    hipLaunchKernelGGL(maxpool2_fwd4_kernel, dim3(grid_for(n / 4, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, act_st);
    IPR_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int act_st, void* stream) {
  const size_t n = (size_t)B * H * W * C;
  if (!n) return 0;
  const bool vec = (C % 4) == 0 && (H % 2) == 0 && (W % 2) == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  IPR_CHECK(act_st == 0 || (act_st == 2 && vec), "maxpool2_bwd: three-plane tensors need even maps and C %% 4 == 0");
  if (vec) {
    hipLaunchKernelGGL(maxpool2_bwd4_kernel, dim3(grid_for(n / 16, 8192)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, B, H, W, C, act_st);
    IPR_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, dy, dx,
                     B, H, W, C);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_add(const float* a, const float* b, float* out, size_t n, int act_st, void* stream) {
  if (!n) return 0;
  IPR_ST_CHECK(act_st, n, "add");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, act_st);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_write_ints(int* dst, const int* values, int n, void* stream) {
  for (int off = 0; off < n; off += IntsArg::N) {
    IntsArg t;
    const int cnt = n - off < IntsArg::N ? n - off : IntsArg::N;
    for (int i = 0; i < cnt; ++i) t.v[i] = values[off + i];
    hipLaunchKernelGGL(write_ints_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst + off, t, cnt);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}
int iprgan_pool_swap(float* images, float* pool, const int* index, const int* take, int count, size_t n, void* stream) {
  if (!n || count <= 0) return 0;
  IPR_CHECK(count <= 65535, "pool_swap: %d images in one call (limit 65535)", count);
  hipLaunchKernelGGL(pool_swap_kernel, dim3(grid_for(n, 8192), count), dim3(256), 0, (hipStream_t)stream, images, pool, index, take, n);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_reflect_fold(const float* dxp, float* dx, const float* prev_out, int prev_act, float prev_slope,
                        const float* residual, int B,
                        int H, int W, int C, int pad, void* stream) {
  return reflect_fold_launch(dxp, dx, prev_out, prev_act, prev_slope, residual, B, H, W, C, pad, 0, 0, (hipStream_t)stream);
}
}  // extern "C"
namespace iprgan {
int reflect_fold_launch(const float* dxp, float* dx, const float* prev_out, int prev_act, float prev_slope,
                        const float* residual, int B, int H, int W, int C, int pad, int okind, size_t ps, hipStream_t stream) {
  const size_t n = (size_t)B * H * W * C;
  if (okind == 2 && !ps) ps = n;
  IPR_CHECK(pad < H && pad < W, "reflect_fold: pad %d must be smaller than the image", pad);
  IPR_CHECK(C % 4 == 0 && n / 4 < 0x7fffffffull, "reflect_fold: C=%d must be a multiple of 4 (and < 2^33 elements)", C);
  if (!n) return 0;
  hipLaunchKernelGGL(reflect_fold_kernel, dim3(grid_for(n / 4, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const rf_f4*)dxp, (rf_f4*)dx, (const rf_f4*)prev_out, prev_act, prev_slope, (const rf_f4*)residual,
                     (unsigned)(n / 4), H, W,
                     C / 4, pad, make_fastdiv(C / 4), make_fastdiv(W), make_fastdiv(H), okind, ps);
  IPR_LAUNCH_CHECK();
  return 0;
}
int reflect_ring_fix_launch(const float* dxp, float* dx, const float* prev_out, int prev_act, float prev_slope, int B, int H,
                            int W, int C, int pad, int okind, size_t ps, hipStream_t stream) {
  IPR_CHECK(pad >= 1 && 2 * pad + 2 <= H && 2 * pad + 2 <= W && C % 4 == 0, "reflect_ring_fix: pad %d on a %dx%d image", pad, H, W);
  const int ring = 2 * pad * W + (H - 2 * pad) * 2 * pad;
  const size_t n4 = (size_t)B * ring * (C / 4);
  IPR_CHECK(n4 < 0x7fffffffull, "reflect_ring_fix: too many elements");
  if (okind == 2 && !ps) ps = (size_t)B * H * W * C;
  if (!n4) return 0;
  hipLaunchKernelGGL(reflect_ring_fix_kernel, dim3(grid_for(n4, 8192)), dim3(256), 0, stream, (const rf_f4*)dxp, (rf_f4*)dx,
                     (const rf_f4*)prev_out, prev_act, prev_slope, (unsigned)n4, H, W, C / 4, pad, ring, make_fastdiv(C / 4),
                     make_fastdiv(ring), make_fastdiv(W), make_fastdiv(2 * pad), okind, ps);
  IPR_LAUNCH_CHECK();
  return 0;
}
}  // namespace iprgan
extern "C" {

static bool loss_needs_y(int kind) {
  return kind == IPRGAN_LOSS_MSE || kind == IPRGAN_LOSS_L1 || kind == IPRGAN_LOSS_BCE_PM1 ||
         kind == IPRGAN_LOSS_MSE_DENORM || kind == IPRGAN_LOSS_L1_DENORM;
}
size_t iprgan_loss_ws_floats(size_t n) { (void)n; return LOSS_BLOCKS; }
int iprgan_loss_sum_fwd(int kind, const float* x, const float* y, float* loss, float* ws, size_t n,
                        float scale, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(kind >= 0 && kind <= IPRGAN_LOSS_L1_DENORM, "loss_fwd: bad kind %d", kind);
  IPR_CHECK(n > 0, "loss_fwd: empty input");
  IPR_CHECK(!loss_needs_y(kind) || y, "loss_fwd: kind %d needs a second input", kind);
  const int nb = grid_for(n, LOSS_BLOCKS);
  if (nb == 1) {
    hipLaunchKernelGGL(loss_small_kernel, dim3(1), dim3(256), 0, st, kind, x, y, loss, n, scale);
    IPR_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, st, kind, x, y, ws, n);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, ws, nb, scale, loss);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_loss_sum_bwd(int kind, const float* x, const float* y, const float* gscale, float* dx, size_t n,
                        float scale, void* stream) {
  IPR_CHECK(kind >= 0 && kind <= IPRGAN_LOSS_L1_DENORM, "loss_bwd: bad kind %d", kind);
  IPR_CHECK(!loss_needs_y(kind) || y, "loss_bwd: kind %d needs a second input", kind);
  if (!n) return 0;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, kind, x, y,
                     gscale, dx, n, scale);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_loss_pair_fwd(int kind_a, int kind_b, const float* x, size_t n_half, float* out3, void* stream) {
  IPR_CHECK(kind_a >= 0 && kind_a <= IPRGAN_LOSS_L1_DENORM && kind_b >= 0 && kind_b <= IPRGAN_LOSS_L1_DENORM &&
            !loss_needs_y(kind_a) && !loss_needs_y(kind_b), "loss_pair_fwd: kinds %d / %d (one-input losses only)", kind_a, kind_b);
  IPR_CHECK(n_half > 0 && n_half <= 256, "loss_pair_fwd: %zu elements per half (1 .. 256)", n_half);
  hipLaunchKernelGGL(loss_pair_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kind_a, kind_b, x, (int)n_half,
                     1.0f / (float)n_half, out3);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_loss_pair_bwd(int kind_a, int kind_b, const float* x, const float* gscale, float* dx, size_t n_half, void* stream) {
  IPR_CHECK(kind_a >= 0 && kind_a <= IPRGAN_LOSS_L1_DENORM && kind_b >= 0 && kind_b <= IPRGAN_LOSS_L1_DENORM &&
            !loss_needs_y(kind_a) && !loss_needs_y(kind_b), "loss_pair_bwd: kinds %d / %d (one-input losses only)", kind_a, kind_b);
  IPR_CHECK(n_half > 0 && n_half <= 256, "loss_pair_bwd: %zu elements per half (1 .. 256)", n_half);
  hipLaunchKernelGGL(loss_pair_bwd_kernel, dim3(cdiv((int)(2 * n_half), 256)), dim3(256), 0, (hipStream_t)stream, kind_a, kind_b, x,
                     gscale, dx, (int)n_half, 1.0f / (float)n_half);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_loss_fwd(int kind, const float* x, const float* y, float* loss, float* ws, size_t n,
                    void* stream) {
  IPR_CHECK(n > 0, "loss_fwd: empty input");
  return iprgan_loss_sum_fwd(kind, x, y, loss, ws, n, 1.0f / (float)n, stream);
}
int iprgan_loss_bwd(int kind, const float* x, const float* y, const float* gscale, float* dx, size_t n,
                    void* stream) {
  if (!n) return 0;
  return iprgan_loss_sum_bwd(kind, x, y, gscale, dx, n, 1.0f / (float)n, stream);
}

int iprgan_reparam_fwd(const float* mean, const float* logvar, const float* eps, float* z, size_t n,
                       void* stream) {
  if (!n) return 0;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, mean,
                     logvar, eps, z, n);
  IPR_LAUNCH_CHECK();
  return 0;
}
int iprgan_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmean, float* dlogvar,
                       size_t n, void* stream) {
  if (!n) return 0;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, dz,
                     logvar, eps, dmean, dlogvar, n);
  IPR_LAUNCH_CHECK();
  return 0;
}

static int fill_sign_table(SignTable& t, const float* const* gammas, const float* const* signs,
                           float* const* dgammas, const int* sizes, int begin, int nlayer) {
  int cnt = nlayer - begin;
  if (cnt > SIGN_MAX_LAYERS) cnt = SIGN_MAX_LAYERS;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < cnt; ++i) {
    t.gamma[i] = gammas[begin + i];
    t.sign[i] = signs[begin + i];
    t.dgamma[i] = dgammas ? dgammas[begin + i] : nullptr;
    t.size[i] = sizes[begin + i];
  }
  t.nlayer = cnt;
  return cnt;
}

int iprgan_sign_loss_fwd(const float* const* gammas, const float* const* signs, const int* sizes, int nlayer,
                         float gamma0, float* loss, void* stream) {
  IPR_CHECK(nlayer > 0, "sign_loss_fwd: no layers");
  for (int b = 0; b < nlayer; b += SIGN_MAX_LAYERS) {
    SignTable t;
    fill_sign_table(t, gammas, signs, nullptr, sizes, b, nlayer);
    hipLaunchKernelGGL(sign_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, gamma0, loss,
                       b > 0 ? 1 : 0);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}
int iprgan_sign_loss_bwd(const float* const* gammas, const float* const* signs, float* const* dgammas,
                         const int* sizes, int nlayer, float gamma0, const float* gscale, float beta, void* stream) {
  for (int b = 0; b < nlayer; b += SIGN_MAX_LAYERS) {
    SignTable t;
    const int cnt = fill_sign_table(t, gammas, signs, dgammas, sizes, b, nlayer);
    hipLaunchKernelGGL(sign_loss_bwd_kernel, dim3(cnt), dim3(256), 0, (hipStream_t)stream, t, gamma0, gscale, beta);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}
int iprgan_sign_ber(const float* const* gammas, const float* const* signs, const int* sizes, int nlayer,
                    long long* counts, void* stream) {
  IPR_CHECK(nlayer > 0, "sign_ber: no layers");
  for (int b = 0; b < nlayer; b += SIGN_MAX_LAYERS) {
    SignTable t;
    fill_sign_table(t, gammas, signs, nullptr, sizes, b, nlayer);
    hipLaunchKernelGGL(sign_ber_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, counts, b > 0 ? 1 : 0);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

static int adam_launch(float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const long long* sizes, int n, double beta1, double beta2, double eps,
                       double weight_decay, float step_size, float bc2_sqrt, const float* coef, double grad_scale,
                       void* stream);

int iprgan_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const long long* sizes, int n, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
  IPR_CHECK(step >= 1, "adam_step: step must be >= 1");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  return adam_launch(params, grads, exp_avg, exp_avg_sq, sizes, n, beta1, beta2, eps, weight_decay, (float)(lr / bc1),
                     (float)sqrt(bc2), nullptr, grad_scale, stream);
}

int iprgan_adam_step_dev(float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const long long* sizes, int n, double lr, double beta1,
                         double beta2, double eps, double weight_decay, int* step_dev, float* coef, double grad_scale,
                         void* stream) {
  IPR_CHECK(step_dev && coef, "adam_step_dev: the device step counter and the two-float coefficient buffer are required");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr, beta1, beta2, coef);
  IPR_LAUNCH_CHECK();
  return adam_launch(params, grads, exp_avg, exp_avg_sq, sizes, n, beta1, beta2, eps, weight_decay, 0.f, 0.f, coef,
                     grad_scale, stream);
}

static int adam_launch(float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const long long* sizes, int n, double beta1, double beta2, double eps,
                       double weight_decay, float step_size, float bc2_sqrt, const float* coef, double grad_scale,
                       void* stream) {
  for (int b = 0; b < n; b += ADAM_MAX_TENSORS) {
    AdamTable t;
    memset(&t, 0, sizeof(t));
    int cnt = n - b;
    if (cnt > ADAM_MAX_TENSORS) cnt = ADAM_MAX_TENSORS;
    long long maxn = 0;
    for (int i = 0; i < cnt; ++i) {
      t.p[i] = params[b + i]; t.g[i] = grads[b + i]; t.m[i] = exp_avg[b + i]; t.v[i] = exp_avg_sq[b + i];
      t.n[i] = sizes[b + i];
      if (t.n[i] > maxn) maxn = t.n[i];
    }
    const int gx = grid_for((size_t)maxn / 4 + 1, 1024);          // (16-byte accesses: four elements per thread and round)
    hipLaunchKernelGGL(adam_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, t,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)weight_decay, step_size, bc2_sqrt, (float)grad_scale, coef);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
