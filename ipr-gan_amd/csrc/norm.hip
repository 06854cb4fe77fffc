// BatchNorm2d (training / eval) over NHWC activations x[M][C], HBM-bound.
// Statistics are reduced in two deterministic stages: each block takes a fixed slice of rows and
// emits (local mean, local M2) per channel; the finalize kernel merges the slices in slice order
// with Chan's parallel-variance formula (stable: no E[x^2]-E[x]^2 cancellation), writes
// mean/invstd and updates the running statistics exactly as torch.nn.BatchNorm2d does
// (biased variance to normalise, unbiased into running_var; reference networks/conv_generator.py:9).
#include "common.h"

namespace iprgan {

#define BN_ROWS_PER_BLOCK 512
#define BN_CH_PER_BLOCK 64   // threads = 64 channels x 4 row lanes

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x,
                                                               float* __restrict__ part, int M, int C) {
  __shared__ float sh[4][BN_CH_PER_BLOCK];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.y * BN_CH_PER_BLOCK + cl;
  const int r0 = blockIdx.x * BN_ROWS_PER_BLOCK;
  int r1 = r0 + BN_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  const int cnt = r1 - r0;
  const bool okc = c < C;
  float s = 0.f;
  if (okc)
    for (int r = r0 + rg; r < r1; r += 4) s += x[(size_t)r * C + c];
  sh[rg][cl] = s;
  __syncthreads();
  const float lmean = (sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) / (float)cnt;
  __syncthreads();
  float m2 = 0.f;
  if (okc)
    for (int r = r0 + rg; r < r1; r += 4) {
      const float d = x[(size_t)r * C + c] - lmean;
      m2 += d * d;
    }
  sh[rg][cl] = m2;
  __syncthreads();
  if (rg == 0 && okc) {
    float* p = part + ((size_t)blockIdx.x * 2) * C;
    p[c] = lmean;
    p[C + c] = sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl];
  }
}

__global__ void bn_stats_final_kernel(const float* __restrict__ part, int nblk, int M, int C,
                                      float eps, float momentum, float* __restrict__ running_mean,
                                      float* __restrict__ running_var, float* __restrict__ save_mean,
                                      float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean = 0.f, m2 = 0.f;
  int n = 0;
  for (int b = 0; b < nblk; ++b) {
    int cnt = M - b * BN_ROWS_PER_BLOCK;
    if (cnt > BN_ROWS_PER_BLOCK) cnt = BN_ROWS_PER_BLOCK;
    const float lm = part[((size_t)b * 2) * C + c], lm2 = part[((size_t)b * 2 + 1) * C + c];
    const float delta = lm - mean;
    const int nn = n + cnt;
    mean += delta * ((float)cnt / (float)nn);
    m2 += lm2 + delta * delta * ((float)n * (float)cnt / (float)nn);
    n = nn;
  }
  const float var = m2 / (float)M;
  save_mean[c] = mean;
  save_invstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unb = M > 1 ? m2 / (float)(M - 1) : var;
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

// y = act((x-mean)*invstd*gamma+beta), float4 over [M][C] (C % 4 == 0)
__global__ void bn_apply_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                size_t n4, int C4n, int act, float slope) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4n) * 4;
    const float4 v = x[i];
    float in[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = gamma ? gamma[c + k] : 1.f, b = beta ? beta[c + k] : 0.f;
      o[k] = act_apply((in[k] - mean[c + k]) * invstd[c + k] * g + b, act, slope);
    }
    y[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// backward stage 1: per block slice, s1 = sum dz, s2 = sum dz*xhat, dz = dy*act'(y)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             float* __restrict__ part, int M, int C,
                                                             int act, float slope) {
  __shared__ float sh[2][4][BN_CH_PER_BLOCK];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.y * BN_CH_PER_BLOCK + cl;
  const int r0 = blockIdx.x * BN_ROWS_PER_BLOCK;
  int r1 = r0 + BN_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    for (int r = r0 + rg; r < r1; r += 4) {
      const size_t i = (size_t)r * C + c;
      const float dz = dy[i] * act_grad_from_out(y[i], act, slope);
      s1 += dz;
      s2 += dz * (x[i] - mu) * is;
    }
  }
  sh[0][rg][cl] = s1;
  sh[1][rg][cl] = s2;
  __syncthreads();
  if (rg == 0 && c < C) {
    float* p = part + ((size_t)blockIdx.x * 2) * C;
    p[c] = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
    p[C + c] = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
  }
}

__global__ void bn_bwd_final_kernel(const float* __restrict__ part, int nblk, int C,
                                    float* __restrict__ sums, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < nblk; ++b) {
    s1 += part[((size_t)b * 2) * C + c];
    s2 += part[((size_t)b * 2 + 1) * C + c];
  }
  sums[c] = s1;
  sums[C + c] = s2;
  if (dgamma) dgamma[c] = s2;
  if (dbeta) dbeta[c] = s1;
}

__global__ void bn_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ y,
                                    const float4* __restrict__ dy, float4* __restrict__ dx,
                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ sums,
                                    size_t n4, int C4n, int C, float invM, int act, float slope) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4n) * 4;
    const float4 xv = x[i], yv = y[i], gv = dy[i];
    const float xi[4] = {xv.x, xv.y, xv.z, xv.w}, yi[4] = {yv.x, yv.y, yv.z, yv.w},
                gi[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float is = invstd[c + k];
      const float xh = (xi[k] - mean[c + k]) * is;
      const float dz = gi[k] * act_grad_from_out(yi[k], act, slope);
      const float g = gamma ? gamma[c + k] : 1.f;
      o[k] = g * is * (dz - sums[c + k] * invM - xh * sums[C + c + k] * invM);
    }
    dx[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace iprgan

using namespace iprgan;

extern "C" {

size_t iprgan_bn_ws_floats(int M, int C) {
  return (size_t)cdiv(M, BN_ROWS_PER_BLOCK) * 2 * C + 2 * (size_t)C;
}

int iprgan_bn_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                  float* running_var, float* save_mean, float* save_invstd, float* ws, int M, int C,
                  float eps, float momentum, int use_running, int act, float slope, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(C % 4 == 0, "bn_fwd: C=%d must be a multiple of 4", C);
  IPR_CHECK(M > 0, "bn_fwd: empty batch");
  if (use_running) {
    IPR_CHECK(running_mean && running_var, "bn_fwd: eval mode needs running stats");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, running_mean,
                       running_var, eps, C, save_mean, save_invstd);
  } else {
    const int nblk = cdiv(M, BN_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nblk, cdiv(C, BN_CH_PER_BLOCK)), dim3(256), 0, st,
                       x, ws, M, C);
    IPR_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, ws, nblk, M, C, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd);
  }
  IPR_LAUNCH_CHECK();
  const size_t n4 = (size_t)M * C / 4;
  const int blocks = (int)(cdivz(n4, 256) < 4096 ? cdivz(n4, 256) : 4096);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)x, (float4*)y,
                     gamma, beta, save_mean, save_invstd, n4, C / 4, act, slope);
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma,
                  const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                  float* dbeta, float* ws, int M, int C, int act, float slope, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(C % 4 == 0, "bn_bwd: C=%d must be a multiple of 4", C);
  const int nblk = cdiv(M, BN_ROWS_PER_BLOCK);
  float* sums = ws + (size_t)nblk * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk, cdiv(C, BN_CH_PER_BLOCK)), dim3(256), 0, st, x, y,
                     dy, save_mean, save_invstd, ws, M, C, act, slope);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, ws, nblk, C, sums, dgamma,
                     dbeta);
  IPR_LAUNCH_CHECK();
  const size_t n4 = (size_t)M * C / 4;
  const int blocks = (int)(cdivz(n4, 256) < 4096 ? cdivz(n4, 256) : 4096);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)x,
                     (const float4*)y, (const float4*)dy, (float4*)dx, gamma, save_mean, save_invstd,
                     sums, n4, C / 4, C, 1.0f / (float)M, act, slope);
  IPR_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
