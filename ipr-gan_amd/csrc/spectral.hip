// Spectral-norm power iteration and its backward for W_mat[rows][cols] (row-major view of the
// layer weight, dim=0), one iteration per training forward, eps = 1e-12 in both normalisations:
//   v <- normalize(W^T u);  u <- normalize(W v);  sigma = u . (W v)
// (torch.nn.utils.spectral_norm as used at reference networks/sn_discriminator.py:9,11,18,21).
// HBM/latency-bound (W <= 4.7 MB); all reductions are fixed-order (deterministic).
#include "common.h"

namespace iprgan {

#define SN_MAX_RSPLIT 8

// t_part[rs][j] = sum_{i in row split rs} W[i][j]*u[i]
__global__ void sn_wtu_partial_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                      float* __restrict__ tpart, int rows, int cols, int rows_per_split) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  const int r0 = blockIdx.y * rows_per_split;
  int r1 = r0 + rows_per_split;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  for (int i = r0; i < r1; ++i) s += w[(size_t)i * cols + j] * u[i];
  tpart[(size_t)blockIdx.y * cols + j] = s;
}

// single block: t = sum of partials; v = t / max(||t||, eps)
__global__ __launch_bounds__(1024) void sn_v_final_kernel(const float* __restrict__ tpart, int nsplit,
                                                           int cols, float eps, float* __restrict__ v) {
  __shared__ float sh[16];
  float ss = 0.f;
  for (int j = threadIdx.x; j < cols; j += blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += tpart[(size_t)s * cols + j];
    ss += t * t;
  }
  const float nrm = sqrtf(block_sum(ss, sh));
  const float inv = 1.f / fmaxf(nrm, eps);
  for (int j = threadIdx.x; j < cols; j += blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += tpart[(size_t)s * cols + j];
    v[j] = t * inv;
  }
}

// s[i] = sum_j W[i][j]*v[j]: one 256-thread block per row (16-B loads when cols % 4 == 0)
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                                    float* __restrict__ s, int rows, int cols) {
  __shared__ float sh[16];
  const int row = blockIdx.x;
  const float* wr = w + (size_t)row * cols;
  float acc = 0.f;
  if ((cols & 3) == 0) {
    for (int j = threadIdx.x * 4; j < cols; j += 1024) {
      const float4 a = *(const float4*)(wr + j), b = *(const float4*)(v + j);
      acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (int j = threadIdx.x; j < cols; j += 256) acc += wr[j] * v[j];
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) s[row] = acc;
}

// single block: u = s / max(||s||, eps); sigma = u . s  (training) or sigma = u_old . s (eval)
__global__ __launch_bounds__(1024) void sn_u_final_kernel(const float* __restrict__ s, int rows, float eps,
                                                           float* __restrict__ u, float* __restrict__ sigma,
                                                           int training) {
  __shared__ float sh[16];
  if (training) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) ss += s[i] * s[i];
    const float nrm = sqrtf(block_sum(ss, sh));
    const float inv = 1.f / fmaxf(nrm, eps);
    float d = 0.f;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
      const float un = s[i] * inv;
      u[i] = un;
      d += un * s[i];
    }
    d = block_sum(d, sh);
    if (threadIdx.x == 0) *sigma = d;
  } else {
    float d = 0.f;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) d += u[i] * s[i];
    d = block_sum(d, sh);
    if (threadIdx.x == 0) *sigma = d;
  }
}

__global__ __launch_bounds__(256) void sn_dot_partial_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ b,
                                                             float* __restrict__ part, size_t n) {
  __shared__ float sh[16];
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += a[i] * b[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ dwsn,
                                                           const float* __restrict__ u,
                                                           const float* __restrict__ v,
                                                           const float* __restrict__ sigma,
                                                           const float* __restrict__ part, int npart,
                                                           float* __restrict__ dw, int rows, int cols) {
  float dot = 0.f;
  for (int i = 0; i < npart; ++i) dot += part[i];   // same fixed order in every thread
  const float sg = *sigma;
  const float coef = dot / sg;
  const size_t n = (size_t)rows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    dw[i] = (dwsn[i] - coef * u[r] * v[c]) / sg;
  }
}


// ---- multi-layer form: the power iterations of ALL spectrally-normalised layers of a network depend only
// on the weights, so one pass runs them together: 4 launches per discriminator pass instead of 4 per layer.
#define SN_MAX_LAYERS 16
struct SNTable {
  const float* w[SN_MAX_LAYERS];
  float* u[SN_MAX_LAYERS];
  float* v[SN_MAX_LAYERS];
  float* u_out[SN_MAX_LAYERS];      // this pass's copies (later passes overwrite u, v)
  float* v_out[SN_MAX_LAYERS];
  int rows[SN_MAX_LAYERS], cols[SN_MAX_LAYERS], nsplit[SN_MAX_LAYERS], rps[SN_MAX_LAYERS];
  int csplit[SN_MAX_LAYERS];        // > 1: W v of this (few-row, wide) layer is summed over column chunks by blockIdx.z
  long long ws_off[SN_MAX_LAYERS];  // per layer: tpart [SN_MAX_RSPLIT*cols] then s [rows]
  int n;
};

__global__ void snm_wtu_kernel(const SNTable t, float* __restrict__ ws) {
  const int l = blockIdx.z;
  const int cols = t.cols[l], rows = t.rows[l];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols || (int)blockIdx.y >= t.nsplit[l]) return;
  const int r0 = blockIdx.y * t.rps[l];
  int r1 = r0 + t.rps[l];
  if (r1 > rows) r1 = rows;
  const float* w = t.w[l];
  const float* u = t.u[l];
  // four independent accumulators: the row loads overlap (one accumulator serialised them: 27 us for 12 MB)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = r0;
  for (; i + 3 < r1; i += 4) {
    s0 += w[(size_t)i * cols + j] * u[i];
    s1 += w[(size_t)(i + 1) * cols + j] * u[i + 1];
    s2 += w[(size_t)(i + 2) * cols + j] * u[i + 2];
    s3 += w[(size_t)(i + 3) * cols + j] * u[i + 3];
  }
  for (; i < r1; ++i) s0 += w[(size_t)i * cols + j] * u[i];
  ws[t.ws_off[l] + (size_t)blockIdx.y * cols + j] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(1024) void snm_v_kernel(const SNTable t, float* __restrict__ ws, float eps) {
  __shared__ float sh[16];
  const int l = blockIdx.x;
  const int cols = t.cols[l], nsplit = t.nsplit[l];
  const float* tpart = ws + t.ws_off[l];
  auto colsum = [&](int j) {                    // all split partials of a column are loaded together
    float p[SN_MAX_RSPLIT];
#pragma unroll
    for (int s = 0; s < SN_MAX_RSPLIT; ++s) p[s] = s < nsplit ? tpart[(size_t)s * cols + j] : 0.f;
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < SN_MAX_RSPLIT; ++s) a += p[s];
    return a;
  };
  // up to 32 columns per thread stay in registers between the norm pass and the write pass, loaded together
  // (one block per layer: the serial column walk was the long pole, 19 us for the 32768-column head)
  constexpr int KEEP = 32;
  float* v = t.v[l];
  float* vo = t.v_out[l];
  float keep[KEEP];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int j = threadIdx.x + q * blockDim.x;
    keep[q] = j < cols ? colsum(j) : 0.f;
    ss += keep[q] * keep[q];
  }
  for (int j = threadIdx.x + KEEP * blockDim.x; j < cols; j += blockDim.x) {
    const float a = colsum(j);
    ss += a * a;
  }
  const float inv = 1.f / fmaxf(sqrtf(block_sum(ss, sh)), eps);
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int j = threadIdx.x + q * blockDim.x;
    if (j < cols) {
      const float a = keep[q] * inv;
      v[j] = a;
      if (vo) vo[j] = a;
    }
  }
  for (int j = threadIdx.x + KEEP * blockDim.x; j < cols; j += blockDim.x) {
    float a = colsum(j);
    a *= inv;
    v[j] = a;
    if (vo) vo[j] = a;
  }
}
// Wide layers (the 131072-column Linear head of the 128x128 discriminator): one block per layer walks the columns for
// 80 us.  Split form: blocks of SNV_COLS columns write the raw column sums and one partial of the squared norm each;
// the scale pass adds the partials in block order (deterministic) and normalises its columns in place.
constexpr int SNV_COLS = 4096, SNV_MAX_BLOCKS = 64;
__device__ __forceinline__ float* snv_part(const SNTable& t, float* ws, int l) {
  return ws + t.ws_off[l] + (size_t)SN_MAX_RSPLIT * t.cols[l] + t.rows[l] + 16;
}
__global__ __launch_bounds__(1024) void snm_vsum_kernel(const SNTable t, float* __restrict__ ws) {
  __shared__ float sh[16];
  const int l = blockIdx.y;
  const int cols = t.cols[l], nsplit = t.nsplit[l];
  const int j0 = blockIdx.x * SNV_COLS;
  if (j0 >= cols) return;
  const float* tpart = ws + t.ws_off[l];
  float* v = t.v[l];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < SNV_COLS / 1024; ++q) {
    const int j = j0 + threadIdx.x + q * 1024;
    if (j < cols) {
      float p[SN_MAX_RSPLIT];
#pragma unroll
      for (int s = 0; s < SN_MAX_RSPLIT; ++s) p[s] = s < nsplit ? tpart[(size_t)s * cols + j] : 0.f;
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < SN_MAX_RSPLIT; ++s) a += p[s];
      v[j] = a;
      ss += a * a;
    }
  }
  ss = block_sum(ss, sh);
  if (threadIdx.x == 0) snv_part(t, ws, l)[blockIdx.x] = ss;
}
__global__ __launch_bounds__(1024) void snm_vscale_kernel(const SNTable t, float* __restrict__ ws, float eps) {
  const int l = blockIdx.y;
  const int cols = t.cols[l];
  const int j0 = blockIdx.x * SNV_COLS;
  if (j0 >= cols) return;
  const float* part = snv_part(t, ws, l);
  const int nblk = (cols + SNV_COLS - 1) / SNV_COLS;
  float ss = 0.f;
  for (int b = 0; b < nblk; ++b) ss += part[b];
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  float* v = t.v[l];
  float* vo = t.v_out[l];
#pragma unroll
  for (int q = 0; q < SNV_COLS / 1024; ++q) {
    const int j = j0 + threadIdx.x + q * 1024;
    if (j < cols) {
      const float a = v[j] * inv;
      v[j] = a;
      if (vo) vo[j] = a;
    }
  }
}
// column chunks of a few-row, wide layer (the 1 x 131072 Linear head at 128x128: one block walked the row for 38 us):
// partial dot products per (row, chunk), summed in chunk order by snm_u_kernel
constexpr int SNW_MAX_CSPLIT = 16, SNW_SPLIT_ROWS = 8;
__device__ __forceinline__ float* snw_part(const SNTable& t, float* ws, int l) {
  return ws + t.ws_off[l] + (size_t)SN_MAX_RSPLIT * t.cols[l] + t.rows[l] + 16 + SNV_MAX_BLOCKS;
}
__global__ __launch_bounds__(256) void snm_wv_kernel(const SNTable t, float* __restrict__ ws) {
  __shared__ float sh[16];
  const int l = blockIdx.y, row = blockIdx.x;
  const int cols = t.cols[l], cs = t.csplit[l];
  if (row >= t.rows[l] || (int)blockIdx.z >= cs) return;
  const float* wr = t.w[l] + (size_t)row * cols;
  const float* v = t.v[l];
  // chunk of this block: multiples of 1024 columns so that the 16-byte path keeps its alignment
  const int per = cs > 1 ? ((cols + cs - 1) / cs + 1023) / 1024 * 1024 : cols;
  const int j0 = (int)blockIdx.z * per, j1 = j0 + per < cols ? j0 + per : cols;
  float acc = 0.f;
  if ((cols & 3) == 0) {
    for (int j = j0 + threadIdx.x * 4; j < j1; j += 1024) {
      const float4 a = *(const float4*)(wr + j), b = *(const float4*)(v + j);
      acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (int j = j0 + threadIdx.x; j < j1; j += 256) acc += wr[j] * v[j];
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    if (cs > 1) snw_part(t, ws, l)[row * SNW_MAX_CSPLIT + blockIdx.z] = acc;
    else ws[t.ws_off[l] + (size_t)SN_MAX_RSPLIT * cols + row] = acc;
  }
}
__global__ __launch_bounds__(1024) void snm_u_kernel(const SNTable t, const float* __restrict__ ws, float eps,
                                                      float* __restrict__ sigma, int training) {
  __shared__ float sh[16];
  const int l = blockIdx.x;
  const int rows = t.rows[l], cols = t.cols[l];
  float* s = const_cast<float*>(ws) + t.ws_off[l] + (size_t)SN_MAX_RSPLIT * cols;
  if (t.csplit[l] > 1) {               // column-chunk partials of W v -> s[row], in chunk order
    if ((int)threadIdx.x < rows) {
      const float* part = snw_part(t, const_cast<float*>(ws), l) + threadIdx.x * SNW_MAX_CSPLIT;
      float a = 0.f;
      for (int z = 0; z < t.csplit[l]; ++z) a += part[z];
      s[threadIdx.x] = a;
    }
    __syncthreads();
  }
  float* u = t.u[l];
  float* uo = t.u_out[l];
  float d = 0.f;
  if (training) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) ss += s[i] * s[i];
    const float inv = 1.f / fmaxf(sqrtf(block_sum(ss, sh)), eps);
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
      const float un = s[i] * inv;
      u[i] = un;
      if (uo) uo[i] = un;
      d += un * s[i];
    }
  } else {
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
      d += u[i] * s[i];
      if (uo) uo[i] = u[i];
    }
    if (t.v_out[l]) for (int j = threadIdx.x; j < cols; j += blockDim.x) t.v_out[l][j] = t.v[l][j];
  }
  d = block_sum(d, sh);
  if (threadIdx.x == 0) sigma[l] = d;
}

// ---- multi-layer backward: dW_orig = (dWsn - (sum dWsn*W)/sigma * u v^T)/sigma for n layers in 2 launches
struct SNBwdTable {
  const float* dwsn[SN_MAX_LAYERS];
  const float* w[SN_MAX_LAYERS];
  const float* u[SN_MAX_LAYERS];
  const float* v[SN_MAX_LAYERS];
  const float* sigma[SN_MAX_LAYERS];
  float* dw[SN_MAX_LAYERS];
  int rows[SN_MAX_LAYERS], cols[SN_MAX_LAYERS];
};
#define SNB_BLOCKS 64
__global__ __launch_bounds__(256) void snm_dot_kernel(const SNBwdTable t, float* __restrict__ part) {
  __shared__ float sh[16];
  const int l = blockIdx.y;
  const size_t n = (size_t)t.rows[l] * t.cols[l];
  const float* a = t.dwsn[l];
  const float* b = t.w[l];
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if ((n & 3) == 0) {          // 16-byte loads, two of each array in flight
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* a4 = (const f4*)a;
    const f4* b4 = (const f4*)b;
    const size_t n4 = n >> 2;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (; i + stride < n4; i += 2 * stride) {
      acc0 += a4[i] * b4[i];
      acc1 += a4[i + stride] * b4[i + stride];
    }
    if (i < n4) acc0 += a4[i] * b4[i];
    const f4 t4 = acc0 + acc1;
    s = (t4.x + t4.y) + (t4.z + t4.w);
  } else {
    for (; i < n; i += stride) s += a[i] * b[i];
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[l * SNB_BLOCKS + blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void snm_bwd_apply_kernel(const SNBwdTable t, const float* __restrict__ part,
                                                            float beta) {
  const int l = blockIdx.y;
  float dot = 0.f;
  for (int i = 0; i < SNB_BLOCKS; ++i) dot += part[l * SNB_BLOCKS + i];
  const float sg = *t.sigma[l];
  const float coef = dot / sg;
  const int cols = t.cols[l];
  const size_t n = (size_t)t.rows[l] * cols;
  const float* dwsn = t.dwsn[l];
  const float* u = t.u[l];
  const float* v = t.v[l];
  float* dw = t.dw[l];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    const float g = (dwsn[i] - coef * u[r] * v[c]) / sg;
    dw[i] = beta != 0.f ? beta * dw[i] + g : g;
  }
}

}  // namespace iprgan

using namespace iprgan;

#define SN_DOT_BLOCKS 128

extern "C" {

size_t iprgan_sn_ws_floats(int rows, int cols) {
  return (size_t)SN_MAX_RSPLIT * cols + rows + SN_DOT_BLOCKS + 16;
}

int iprgan_sn_power_iter(const float* w, float* u, float* v, float* sigma, float* ws, int rows, int cols,
                         float eps, int training, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float* tpart = ws;
  float* s = ws + (size_t)SN_MAX_RSPLIT * cols;
  if (training) {
    int nsplit = rows / 32;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > SN_MAX_RSPLIT) nsplit = SN_MAX_RSPLIT;
    const int rps = cdiv(rows, nsplit);
    nsplit = cdiv(rows, rps);
    hipLaunchKernelGGL(sn_wtu_partial_kernel, dim3(cdiv(cols, 256), nsplit), dim3(256), 0, st, w, u, tpart,
                       rows, cols, rps);
    IPR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_v_final_kernel, dim3(1), dim3(1024), 0, st, tpart, nsplit, cols, eps, v);
    IPR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(sn_wv_kernel, dim3(rows), dim3(256), 0, st, w, v, s, rows, cols);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sn_u_final_kernel, dim3(1), dim3(1024), 0, st, s, rows, eps, u, sigma, training);
  IPR_LAUNCH_CHECK();
  return 0;
}


size_t iprgan_sn_multi_ws_floats(const int* rows, const int* cols, int n) {
  size_t t = 0;
  for (int i = 0; i < n; ++i) t += (size_t)SN_MAX_RSPLIT * cols[i] + rows[i] + 16 + SNV_MAX_BLOCKS + SNW_SPLIT_ROWS * SNW_MAX_CSPLIT;
  return t;
}

int iprgan_sn_power_iter_multi(const float* const* w, float* const* u, float* const* v, float* const* u_out,
                               float* const* v_out, float* sigma, float* ws, const int* rows, const int* cols,
                               int n, float eps, int training, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(n >= 1 && n <= SN_MAX_LAYERS, "sn_power_iter_multi: %d layers (max %d)", n, SN_MAX_LAYERS);
  SNTable t;
  memset(&t, 0, sizeof(t));
  t.n = n;
  long long off = 0;
  int max_cols = 0, max_rows = 0, max_split = 1, max_csplit = 1;
  for (int i = 0; i < n; ++i) {
    t.w[i] = w[i]; t.u[i] = u[i]; t.v[i] = v[i];
    t.u_out[i] = u_out ? u_out[i] : nullptr; t.v_out[i] = v_out ? v_out[i] : nullptr;
    t.rows[i] = rows[i]; t.cols[i] = cols[i];
    int ns = rows[i] / 32;
    if (ns < 1) ns = 1;
    if (ns > SN_MAX_RSPLIT) ns = SN_MAX_RSPLIT;
    t.rps[i] = cdiv(rows[i], ns);
    t.nsplit[i] = cdiv(rows[i], t.rps[i]);
    t.ws_off[i] = off;
    off += (long long)SN_MAX_RSPLIT * cols[i] + rows[i] + 16 + SNV_MAX_BLOCKS + SNW_SPLIT_ROWS * SNW_MAX_CSPLIT;
    if (cols[i] > max_cols) max_cols = cols[i];
    if (rows[i] > max_rows) max_rows = rows[i];
    if (t.nsplit[i] > max_split) max_split = t.nsplit[i];
    t.csplit[i] = 1;
    if (rows[i] <= SNW_SPLIT_ROWS && cols[i] >= 16384) {
      int cs = cols[i] / 8192;
      t.csplit[i] = cs > SNW_MAX_CSPLIT ? SNW_MAX_CSPLIT : cs;
    }
    if (t.csplit[i] > max_csplit) max_csplit = t.csplit[i];
  }
  if (training) {
    hipLaunchKernelGGL(snm_wtu_kernel, dim3(cdiv(max_cols, 256), max_split, n), dim3(256), 0, st, t, ws);
    IPR_LAUNCH_CHECK();
    if (max_cols > 32768 && max_cols <= SNV_COLS * SNV_MAX_BLOCKS) {
      const dim3 g(cdiv(max_cols, SNV_COLS), n);
      hipLaunchKernelGGL(snm_vsum_kernel, g, dim3(1024), 0, st, t, ws);
      IPR_LAUNCH_CHECK();
      hipLaunchKernelGGL(snm_vscale_kernel, g, dim3(1024), 0, st, t, ws, eps);
    } else {
      hipLaunchKernelGGL(snm_v_kernel, dim3(n), dim3(1024), 0, st, t, ws, eps);
    }
    IPR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(snm_wv_kernel, dim3(max_rows, n, max_csplit), dim3(256), 0, st, t, ws);
  IPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(snm_u_kernel, dim3(n), dim3(1024), 0, st, t, ws, eps, sigma, training);
  IPR_LAUNCH_CHECK();
  return 0;
}


int iprgan_sn_bwd_multi(const float* const* dwsn, const float* const* w, const float* const* u,
                        const float* const* v, const float* const* sigma, float* const* dw, float* ws,
                        const int* rows, const int* cols, int n, float beta, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(n >= 1 && n <= SN_MAX_LAYERS, "sn_bwd_multi: %d layers (max %d)", n, SN_MAX_LAYERS);
  SNBwdTable t;
  memset(&t, 0, sizeof(t));
  size_t maxn = 0;
  for (int i = 0; i < n; ++i) {
    t.dwsn[i] = dwsn[i]; t.w[i] = w[i]; t.u[i] = u[i]; t.v[i] = v[i]; t.sigma[i] = sigma[i]; t.dw[i] = dw[i];
    t.rows[i] = rows[i]; t.cols[i] = cols[i];
    const size_t e = (size_t)rows[i] * cols[i];
    if (e > maxn) maxn = e;
  }
  hipLaunchKernelGGL(snm_dot_kernel, dim3(SNB_BLOCKS, n), dim3(256), 0, st, t, ws);
  IPR_LAUNCH_CHECK();
  const int bx = (int)(cdivz(maxn, 1024) < 512 ? cdivz(maxn, 1024) : 512);
  hipLaunchKernelGGL(snm_bwd_apply_kernel, dim3(bx > 0 ? bx : 1, n), dim3(256), 0, st, t, ws, beta);
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_sn_bwd(const float* dwsn, const float* w, const float* u, const float* v, const float* sigma,
                  float* dw, float* ws, int rows, int cols, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)rows * cols;
  float* part = ws + (size_t)SN_MAX_RSPLIT * cols + rows;
  int nb = (int)(cdivz(n, 1024) < SN_DOT_BLOCKS ? cdivz(n, 1024) : SN_DOT_BLOCKS);
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sn_dot_partial_kernel, dim3(nb), dim3(256), 0, st, dwsn, w, part, n);
  IPR_LAUNCH_CHECK();
  const int blocks = (int)(cdivz(n, 256) < 2048 ? cdivz(n, 256) : 2048);
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, dwsn, u, v, sigma, part, nb, dw,
                     rows, cols);
  IPR_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
