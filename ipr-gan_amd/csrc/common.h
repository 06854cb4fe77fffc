// Shared host/device helpers for libiprgan_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "iprgan.h"

namespace iprgan {

void set_error(const char* fmt, ...);
// norm.hip: deterministic column sums of x[M][Cs] -> out[C] (bias gradients)
size_t colsum_ws_floats(int M, int Cs);
int colsum_launch(const float* x, float* out, float* ws, int M, int Cs, int C, hipStream_t st, float beta = 0.f, int b16 = 0,
                  size_t ps = 0);     // b16: storage kind of x; ps: plane stride (elements) of a three-plane x, 0 = M * Cs
// conv_x3.hip: fp32 <-> three bf16 planes (plane stride ps elements)
int cast_planes(const void* src, void* dst, size_t n, size_t ps, bool to_planes, hipStream_t st);
// elementwise.hip: gradient of ReflectionPad2d (iprgan_reflect_fold) with a result of storage kind okind (2: plane stride ps)
int reflect_fold_launch(const float* dxp, float* dx, const float* prev_out, int prev_act, float prev_slope,
                        const float* residual, int B, int H, int W, int C, int pad, int okind, size_t ps, hipStream_t stream);

// elementwise.hip: the mirror terms of a ReflectionPad2d gradient (border strips of the padded grid, fp32 `dxp`) added to the
// ring pixels of dx, which already holds the zero-padded backward-data result (conv_igemm.hip: conv_bwd_data_reflect_direct)
int reflect_ring_fix_launch(const float* dxp, float* dx, const float* prev_out, int prev_act, float prev_slope, int B, int H,
                            int W, int C, int pad, int okind, size_t ps, hipStream_t stream);

#define IPR_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) {                           \
      ::iprgan::set_error(__VA_ARGS__);      \
      return 1;                              \
    }                                        \
  } while (0)

#define IPR_LAUNCH_CHECK()                                                       \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ::iprgan::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                          hipGetErrorString(e__));                               \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int rup(int a, int b) { return cdiv(a, b) * b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// Exact unsigned division by a runtime-constant divisor for n < 2^31 (Granlund-Montgomery):
// q = (umulhi(n, mul) + n) >> shift.
struct FastDiv {
  uint32_t mul, shift, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  return f;
}
__host__ __device__ static inline uint32_t fdiv(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t = __umulhi(n, f.mul);
#else
  uint32_t t = (uint32_t)(((uint64_t)n * f.mul) >> 32);
#endif
  return (t + n) >> f.shift;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024; every thread gets the result. smem: >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];   // fixed order: deterministic
  return r;
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case IPRGAN_ACT_RELU: return v > 0.f ? v : 0.f;
    case IPRGAN_ACT_LRELU: return v > 0.f ? v : v * slope;
    case IPRGAN_ACT_TANH: return tanhf(v);
    case IPRGAN_ACT_SIGMOID_PM1: return (1.f / (1.f + expf(-v))) * 2.f - 1.f;
    default: return v;
  }
}
// derivative of the activation expressed through its OUTPUT o
__device__ __forceinline__ float act_grad_from_out(float o, int act, float slope) {
  switch (act) {
    case IPRGAN_ACT_RELU: return o > 0.f ? 1.f : 0.f;
    case IPRGAN_ACT_LRELU: return o > 0.f ? 1.f : slope;
    case IPRGAN_ACT_TANH: return 1.f - o * o;
    case IPRGAN_ACT_SIGMOID_PM1: return (1.f + o) * (1.f - o) * 0.5f;     // 2 s (1 - s), s = (o + 1) / 2
    default: return 1.f;
  }
}

}  // namespace iprgan
