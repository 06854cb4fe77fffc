// SSIM loss of the black-box watermark objective (reference tools/loss.py:82-85: 1 - SSIM(data_range=1),
// optionally on (x+1)/2, (y+1)/2).  The arithmetic is the third-party pytorch-msssim 0.2.1 `ssim`
// (absent offline, restated): 11-tap Gaussian (sigma 1.5) "valid" window per channel,
//   S = (2 mx my + C1)(2 sxy + C2) / ((mx^2 + my^2 + C1)(sx^2 + sy^2 + C2)),  C1 = 0.01^2, C2 = 0.03^2,
// mean over window positions, channels and batch.  HBM/latency-bound (images of 32..256 pixels): one tile
// kernel per direction, tiles staged through LDS, deterministic two-stage reduction of the mean.
//
// forward : per 16x16 tile of window positions, the 26x26 patch of x and y goes to LDS once; every thread
//           accumulates the five windowed moments (x, y, xx, yy, xy) of its position, forms S and - for the
//           backward pass - the three sensitivities dS/dE[x], dS/dE[xx], dS/dE[xy] (gmaps).
// backward: dL/dx_p = sum over windows q containing p of w(p-q) (A_q + 2 x_p B_q + y_p C_q): the transposed
//           ("full") window applied to gmaps, again tile + halo through LDS.  y carries no gradient
//           (the wrapper detaches it, models/wrappers.py:50-52).
#include "common.h"

namespace iprgan {

#define SSIM_WIN 11
#define SSIM_T 16
#define SSIM_P (SSIM_T + SSIM_WIN - 1)

struct SsimWin {
  float w[SSIM_WIN];
};

static SsimWin ssim_window() {          // pytorch_msssim._fspecial_gauss_1d(11, 1.5) in fp32
  SsimWin g;
  float s = 0.f;
  for (int i = 0; i < SSIM_WIN; ++i) {
    const float c = (float)(i - SSIM_WIN / 2);
    g.w[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
    s += g.w[i];
  }
  for (int i = 0; i < SSIM_WIN; ++i) g.w[i] /= s;
  return g;
}

__device__ __forceinline__ float ssim_in(float v, int denorm) { return denorm ? (v + 1.f) / 2.f : v; }

__global__ __launch_bounds__(SSIM_T* SSIM_T) void ssim_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ y,
                                                                  float* __restrict__ gmaps,
                                                                  float* __restrict__ part, int H, int W,
                                                                  int OH, int OW, int denorm, SsimWin g, int which) {
  // which 0: S = SSIM map; 1: S = contrast-structure map cs = (2 sxy + C2) / (sx^2 + sy^2 + C2) - the per-scale factor of
  // MS-SSIM (pytorch_msssim.ms_ssim: cs at scales 1-4, ssim at scale 5)
  __shared__ float sx[SSIM_P][SSIM_P + 1], sy[SSIM_P][SSIM_P + 1];
  __shared__ float red[16];
  const int plane = blockIdx.z;
  const int oy0 = blockIdx.y * SSIM_T, ox0 = blockIdx.x * SSIM_T;
  const float* xp = x + (size_t)plane * H * W;
  const float* yp = y + (size_t)plane * H * W;
  for (int i = threadIdx.x; i < SSIM_P * SSIM_P; i += blockDim.x) {
    const int r = i / SSIM_P, c = i - r * SSIM_P;
    const int iy = oy0 + r, ix = ox0 + c;
    const bool ok = iy < H && ix < W;
    sx[r][c] = ok ? ssim_in(xp[(size_t)iy * W + ix], denorm) : 0.f;
    sy[r][c] = ok ? ssim_in(yp[(size_t)iy * W + ix], denorm) : 0.f;
  }
  __syncthreads();
  const int ty = threadIdx.x / SSIM_T, tx = threadIdx.x % SSIM_T;
  const int oy = oy0 + ty, ox = ox0 + tx;
  float S = 0.f;
  if (oy < OH && ox < OW) {
    // rows first, then columns: the order of pytorch_msssim.gaussian_filter (dim 2, then dim 3)
    float mx = 0.f, my = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
    for (int j = 0; j < SSIM_WIN; ++j) {
      float cmx = 0.f, cmy = 0.f, cxx = 0.f, cyy = 0.f, cxy = 0.f;
#pragma unroll
      for (int i = 0; i < SSIM_WIN; ++i) {
        const float a = sx[ty + i][tx + j], b = sy[ty + i][tx + j], w = g.w[i];
        cmx += w * a; cmy += w * b; cxx += w * a * a; cyy += w * b * b; cxy += w * a * b;
      }
      const float w = g.w[j];
      mx += w * cmx; my += w * cmy; xx += w * cxx; yy += w * cyy; xy += w * cxy;
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float sxx = xx - mx * mx, syy = yy - my * my, sxy = xy - mx * my;
    const float a1 = 2.f * mx * my + C1, a2 = 2.f * sxy + C2;
    const float b1 = mx * mx + my * my + C1, b2 = sxx + syy + C2;
    S = which ? a2 / b2 : (a1 / b1) * (a2 / b2);
    if (gmaps) {
      const size_t n = (size_t)gridDim.z * OH * OW, o = ((size_t)plane * OH + oy) * OW + ox;
      if (which) {               // cs = a2 / b2
        gmaps[o] = (2.f * mx * S - 2.f * my) / b2;                                      // dcs/dE[x]
        gmaps[n + o] = -S / b2;                                                         // dcs/dE[xx]
        gmaps[2 * n + o] = 2.f / b2;                                                    // dcs/dE[xy]
      } else {
        // S = a1 a2 / (b1 b2); written without divisions by a1, a2 (which may vanish; b1 >= C1, b2 ~>= C2)
        const float ib = 1.f / (b1 * b2);
        gmaps[o] = 2.f * my * (a2 - a1) * ib + 2.f * mx * S * (1.f / b2 - 1.f / b1);   // dS/dE[x]
        gmaps[n + o] = -S / b2;                                                        // dS/dE[xx]
        gmaps[2 * n + o] = 2.f * a1 * ib;                                              // dS/dE[xy]
      }
    }
  }
  const float s = block_sum(S, red);
  if (threadIdx.x == 0) part[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

__global__ void ssim_final_kernel(const float* __restrict__ part, int nb, float inv_n, float* __restrict__ loss) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = 1.f - s * inv_n;
}

__global__ __launch_bounds__(SSIM_T* SSIM_T) void ssim_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ y,
                                                                  const float* __restrict__ gmaps,
                                                                  const float* __restrict__ gscale,
                                                                  float* __restrict__ dx, int H, int W, int OH,
                                                                  int OW, int denorm, float inv_n, SsimWin g,
                                                                  const float* __restrict__ pscale, int accumulate) {
  // pscale (MS-SSIM): per-plane factor d loss / d (plane mean of this scale's map) instead of the global -1/n;
  // accumulate: dx += (the gradient arriving from the coarser scales is already there)
  __shared__ float sa[SSIM_P][SSIM_P + 1], sb[SSIM_P][SSIM_P + 1], sc[SSIM_P][SSIM_P + 1];
  const int plane = blockIdx.z;
  const int iy0 = blockIdx.y * SSIM_T, ix0 = blockIdx.x * SSIM_T;
  const size_t n = (size_t)gridDim.z * OH * OW;
  const float* ga = gmaps + (size_t)plane * OH * OW;
  // windows q = p - (i, j), i, j in 0..10: the tile needs q rows iy0-10 .. iy0+15
  for (int i = threadIdx.x; i < SSIM_P * SSIM_P; i += blockDim.x) {
    const int r = i / SSIM_P, c = i - r * SSIM_P;
    const int qy = iy0 - (SSIM_WIN - 1) + r, qx = ix0 - (SSIM_WIN - 1) + c;
    const bool ok = qy >= 0 && qy < OH && qx >= 0 && qx < OW;
    const size_t o = ok ? (size_t)qy * OW + qx : 0;
    sa[r][c] = ok ? ga[o] : 0.f;
    sb[r][c] = ok ? ga[n + o] : 0.f;
    sc[r][c] = ok ? ga[2 * n + o] : 0.f;
  }
  __syncthreads();
  const int ty = threadIdx.x / SSIM_T, tx = threadIdx.x % SSIM_T;
  const int iy = iy0 + ty, ix = ix0 + tx;
  if (iy >= H || ix >= W) return;
  const size_t p = ((size_t)plane * H + iy) * W + ix;
  const float xv = ssim_in(x[p], denorm), yv = ssim_in(y[p], denorm);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < SSIM_WIN; ++i) {
    float row = 0.f;
#pragma unroll
    for (int j = 0; j < SSIM_WIN; ++j) {
      // window q = p - (i, j) sits at LDS [ty + 10 - i][tx + 10 - j]
      const int r = ty + (SSIM_WIN - 1) - i, c = tx + (SSIM_WIN - 1) - j;
      row += g.w[j] * (sa[r][c] + 2.f * xv * sb[r][c] + yv * sc[r][c]);
    }
    acc += g.w[i] * row;
  }
  // loss = 1 - mean S  ->  dL/dS = -1/n ; (x+1)/2 contributes 1/2
  const float v = acc * (gscale ? *gscale : 1.f) * (pscale ? pscale[plane] * inv_n : -inv_n) * (denorm ? 0.5f : 1.f);
  dx[p] = accumulate ? dx[p] + v : v;
}

// ---- MS-SSIM (tools/loss.py:78-80 -> pytorch_msssim.MS_SSIM(data_range=1), third-party, restated) ----------------------
// five scales; between scales X and Y are average-pooled 2x2 (zero padding of one row / column on odd sizes, the pad
// counted in the average); per (image, channel) plane: prod_{l<5} relu(mean cs_l)^w_l * relu(mean ssim_5)^w_5.
__global__ void avgpool2_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W, int OH,
                                    int OW, int py, int px, int denorm) {
  const size_t total = (size_t)planes * OH * OW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const size_t pl = i / ((size_t)OW * OH);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int iy = 2 * oy - py + a, ix = 2 * ox - px + b;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) s += ssim_in(x[(pl * H + iy) * W + ix], denorm);
      }
    y[i] = 0.25f * s;
  }
}
// dx[p] += 0.25 * dy[pooled cell of p] (* 1/2 when the pooled image was read through (x+1)/2)
__global__ void avgpool2_pad_bwd_add_kernel(const float* __restrict__ dy, float* __restrict__ dx, int planes, int H, int W,
                                            int OH, int OW, int py, int px, float scale) {
  const size_t total = (size_t)planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W), iy = (int)((i / W) % H);
    const size_t pl = i / ((size_t)W * H);
    const int oy = (iy + py) >> 1, ox = (ix + px) >> 1;
    dx[i] += scale * dy[(pl * OH + oy) * OW + ox];
  }
}
// part[(plane, tile)] -> means[plane]; one block per plane, fixed order
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ part, int tiles, float inv_n,
                                                         float* __restrict__ means) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < tiles; i += blockDim.x) s += part[(size_t)blockIdx.x * tiles + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) means[blockIdx.x] = s * inv_n;
}
#define MSSSIM_LEVELS 5
struct MsWeights { float w[MSSSIM_LEVELS]; };
// means[l][plane] -> loss = 1 - mean_p prod_l relu(m)^w_l and coef[l][plane] = d loss / d means[l][plane]
__global__ __launch_bounds__(256) void msssim_combine_kernel(const float* __restrict__ means, int planes, MsWeights w,
                                                             float* __restrict__ loss, float* __restrict__ coef) {
  __shared__ float red[16];
  float s = 0.f;
  for (int p = threadIdx.x; p < planes; p += blockDim.x) {
    float val = 1.f;
    float t[MSSSIM_LEVELS];
#pragma unroll
    for (int l = 0; l < MSSSIM_LEVELS; ++l) {
      t[l] = fmaxf(means[(size_t)l * planes + p], 0.f);
      val *= powf(t[l], w.w[l]);
    }
    s += val;
    if (coef) {
#pragma unroll
      for (int l = 0; l < MSSSIM_LEVELS; ++l)
        coef[(size_t)l * planes + p] = t[l] > 0.f ? -val * w.w[l] / (t[l] * (float)planes) : 0.f;
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = 1.f - s / (float)planes;
}

}  // namespace iprgan

using namespace iprgan;

extern "C" {

size_t iprgan_ssim_ws_floats(int planes, int H, int W) {
  if (H < SSIM_WIN || W < SSIM_WIN) return 0;
  const int OH = H - SSIM_WIN + 1, OW = W - SSIM_WIN + 1;
  return (size_t)planes * cdiv(OH, SSIM_T) * cdiv(OW, SSIM_T);
}
size_t iprgan_ssim_gmap_floats(int planes, int H, int W) {
  if (H < SSIM_WIN || W < SSIM_WIN) return 0;
  return (size_t)3 * planes * (H - SSIM_WIN + 1) * (W - SSIM_WIN + 1);
}

int iprgan_ssim_fwd(const float* x, const float* y, float* loss, float* gmaps, float* ws, int planes, int H,
                    int W, int denorm, void* stream) {
  IPR_CHECK(planes > 0 && planes < 65536, "ssim_fwd: %d image planes unsupported", planes);
  IPR_CHECK(H >= SSIM_WIN && W >= SSIM_WIN, "ssim_fwd: image %dx%d smaller than the 11x11 window", H, W);
  hipStream_t st = (hipStream_t)stream;
  const int OH = H - SSIM_WIN + 1, OW = W - SSIM_WIN + 1;
  dim3 grid(cdiv(OW, SSIM_T), cdiv(OH, SSIM_T), planes);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(SSIM_T * SSIM_T), 0, st, x, y, gmaps, ws, H, W, OH, OW, denorm,
                     ssim_window(), 0);
  IPR_LAUNCH_CHECK();
  const int nb = (int)(grid.x * grid.y * grid.z);
  hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(256), 0, st, ws, nb, 1.0f / ((float)planes * OH * OW), loss);
  IPR_LAUNCH_CHECK();
  return 0;
}

int iprgan_ssim_bwd(const float* x, const float* y, const float* gmaps, const float* gscale, float* dx,
                    int planes, int H, int W, int denorm, void* stream) {
  IPR_CHECK(planes > 0 && planes < 65536, "ssim_bwd: %d image planes unsupported", planes);
  IPR_CHECK(H >= SSIM_WIN && W >= SSIM_WIN, "ssim_bwd: image %dx%d smaller than the 11x11 window", H, W);
  const int OH = H - SSIM_WIN + 1, OW = W - SSIM_WIN + 1;
  dim3 grid(cdiv(W, SSIM_T), cdiv(H, SSIM_T), planes);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(SSIM_T * SSIM_T), 0, (hipStream_t)stream, x, y, gmaps, gscale,
                     dx, H, W, OH, OW, denorm, 1.0f / ((float)planes * OH * OW), ssim_window(), (const float*)nullptr, 0);
  IPR_LAUNCH_CHECK();
  return 0;
}

// ---- MS-SSIM loss: 1 - MS_SSIM(data_range=1)(x, y) (tools/loss.py:78-80) -------------------------------------------
// Workspace layout (floats), all sizes from iprgan_msssim_sizes():
//   pyr   : pooled images of scales 2..5 for x and y (scale 1 = the inputs)
//   gmaps : 3 sensitivity maps per scale (cs for scales 1-4, ssim for scale 5), kept for the backward pass
//   small : tile partials, means[5][planes], coef[5][planes]
struct MsGeom { int H[MSSSIM_LEVELS], W[MSSSIM_LEVELS], OH[MSSSIM_LEVELS], OW[MSSSIM_LEVELS]; size_t pyr_off[MSSSIM_LEVELS], gmap_off[MSSSIM_LEVELS], pyr, gmaps, part; };
static MsGeom ms_geom(int planes, int H, int W) {
  MsGeom g;
  size_t po = 0, go = 0, part = 0;
  for (int l = 0; l < MSSSIM_LEVELS; ++l) {
    g.H[l] = H; g.W[l] = W;
    g.OH[l] = H - SSIM_WIN + 1; g.OW[l] = W - SSIM_WIN + 1;
    g.pyr_off[l] = po;
    if (l > 0) po += (size_t)2 * planes * H * W;         // x then y
    g.gmap_off[l] = go;
    go += (size_t)3 * planes * g.OH[l] * g.OW[l];
    const size_t t = (size_t)planes * cdiv(g.OH[l], SSIM_T) * cdiv(g.OW[l], SSIM_T);
    if (t > part) part = t;
    H = (H + 2 * (H & 1) - 2) / 2 + 1; W = (W + 2 * (W & 1) - 2) / 2 + 1;
  }
  g.pyr = po; g.gmaps = go; g.part = part;
  return g;
}
static const MsWeights kMsW = {{0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f}};

int iprgan_msssim_sizes(int planes, int H, int W, size_t* pyr_floats, size_t* gmap_floats, size_t* small_floats) {
  IPR_CHECK((H < W ? H : W) > (SSIM_WIN - 1) * 16, "ms_ssim: the smaller image side (%d) must exceed %d (five scales of an 11x11 window)",
            H < W ? H : W, (SSIM_WIN - 1) * 16);
  const MsGeom g = ms_geom(planes, H, W);
  *pyr_floats = g.pyr; *gmap_floats = g.gmaps; *small_floats = g.part + (size_t)2 * MSSSIM_LEVELS * planes;
  return 0;
}

int iprgan_msssim_fwd(const float* x, const float* y, float* loss, float* pyr, float* gmaps, float* small, int planes,
                      int H, int W, int denorm, int want_grad, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IPR_CHECK(planes > 0 && planes < 65536, "msssim_fwd: %d image planes unsupported", planes);
  IPR_CHECK((H < W ? H : W) > (SSIM_WIN - 1) * 16, "msssim_fwd: image %dx%d too small for five scales", H, W);
  const MsGeom g = ms_geom(planes, H, W);
  float* part = small;
  float* means = small + g.part;
  float* coef = means + (size_t)MSSSIM_LEVELS * planes;
  const float* xl = x;
  const float* yl = y;
  for (int l = 0; l < MSSSIM_LEVELS; ++l) {
    const int dn = (l == 0) ? denorm : 0;
    dim3 grid(cdiv(g.OW[l], SSIM_T), cdiv(g.OH[l], SSIM_T), planes);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(SSIM_T * SSIM_T), 0, st, xl, yl, want_grad ? gmaps + g.gmap_off[l] : nullptr,
                       part, g.H[l], g.W[l], g.OH[l], g.OW[l], dn, ssim_window(), l < MSSSIM_LEVELS - 1 ? 1 : 0);
    IPR_LAUNCH_CHECK();
    hipLaunchKernelGGL(plane_mean_kernel, dim3(planes), dim3(256), 0, st, part, (int)(grid.x * grid.y),
                       1.0f / ((float)g.OH[l] * g.OW[l]), means + (size_t)l * planes);
    IPR_LAUNCH_CHECK();
    if (l + 1 < MSSSIM_LEVELS) {
      float* xn = pyr + g.pyr_off[l + 1];
      float* yn = xn + (size_t)planes * g.H[l + 1] * g.W[l + 1];
      const size_t tot = (size_t)planes * g.H[l + 1] * g.W[l + 1];
      const int blocks = (int)(cdivz(tot, 256) < 4096 ? cdivz(tot, 256) : 4096);
      hipLaunchKernelGGL(avgpool2_pad_kernel, dim3(blocks), dim3(256), 0, st, xl, xn, planes, g.H[l], g.W[l], g.H[l + 1],
                         g.W[l + 1], g.H[l] & 1, g.W[l] & 1, dn);
      hipLaunchKernelGGL(avgpool2_pad_kernel, dim3(blocks), dim3(256), 0, st, yl, yn, planes, g.H[l], g.W[l], g.H[l + 1],
                         g.W[l + 1], g.H[l] & 1, g.W[l] & 1, dn);
      IPR_LAUNCH_CHECK();
      xl = xn; yl = yn;
    }
  }
  hipLaunchKernelGGL(msssim_combine_kernel, dim3(1), dim3(256), 0, st, means, planes, kMsW, loss, want_grad ? coef : nullptr);
  IPR_LAUNCH_CHECK();
  return 0;
}

// dx (the gradient w.r.t. x, same shape) out; pyr/gmaps/small as left by iprgan_msssim_fwd(want_grad = 1); ws: scratch
// of 2 * planes * H2 * W2 floats (two ping-pong gradient images of scale 2, the largest pooled scale)
int iprgan_msssim_bwd(const float* x, const float* y, const float* pyr, const float* gmaps, const float* small,
                      const float* gscale, float* dx, float* ws, int planes, int H, int W, int denorm, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const MsGeom g = ms_geom(planes, H, W);
  const float* coef = small + g.part + (size_t)MSSSIM_LEVELS * planes;
  float* buf[2] = {ws, ws + (size_t)planes * g.H[1] * g.W[1]};
  float* coarse = nullptr;           // gradient w.r.t. the image of scale l + 1
  for (int l = MSSSIM_LEVELS - 1; l >= 0; --l) {
    const float* xl = l ? pyr + g.pyr_off[l] : x;
    const float* yl = l ? xl + (size_t)planes * g.H[l] * g.W[l] : y;
    float* out = l ? buf[l & 1] : dx;
    const int dn = (l == 0) ? denorm : 0;
    dim3 grid(cdiv(g.W[l], SSIM_T), cdiv(g.H[l], SSIM_T), planes);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(SSIM_T * SSIM_T), 0, st, xl, yl, gmaps + g.gmap_off[l], gscale, out,
                       g.H[l], g.W[l], g.OH[l], g.OW[l], dn, 1.0f / ((float)g.OH[l] * g.OW[l]), ssim_window(),
                       coef + (size_t)l * planes, 0);
    IPR_LAUNCH_CHECK();
    if (coarse) {
      const size_t tot = (size_t)planes * g.H[l] * g.W[l];
      const int blocks = (int)(cdivz(tot, 256) < 4096 ? cdivz(tot, 256) : 4096);
      hipLaunchKernelGGL(avgpool2_pad_bwd_add_kernel, dim3(blocks), dim3(256), 0, st, coarse, out, planes, g.H[l], g.W[l],
                         g.H[l + 1], g.W[l + 1], g.H[l] & 1, g.W[l] & 1, dn ? 0.125f : 0.25f);
      IPR_LAUNCH_CHECK();
    }
    coarse = out;
  }
  return 0;
}

}  // extern "C"
