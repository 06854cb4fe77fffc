// Implicit-GEMM convolution family for gfx950 on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One "gather-GEMM" kernel serves Conv2d forward, Conv2d backward-data (sub-pixel phases for
// stride > 1, no zero insertion), ConvTranspose2d forward (= backward-data form) and
// ConvTranspose2d backward-data (= forward form):
//     out[b, y*osy+ooy, x*osx+oox, n] = sum_{tap,c} in[b, y*isy+dy(tap), x*isx+dx(tap), c] * Wt[n][tap*Cs+c]
// A (activations, NHWC, channels contiguous) and B (tap-major weight K-vectors) tiles are staged
// through LDS in 16-byte chunks with an XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free; each wave owns (32*WM)x(32*WN) of the block tile and issues 4 MFMAs per
// fragment pair (k = 8 per ds_read_b128 pair: lane half h supplies k = 4h+j to MFMA j).
// A second kernel computes backward-weight as a split-M GEMM into partial slabs that a
// fixed-order reduce sums (deterministic) and scatters into PyTorch's weight layout.
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "conv_shared.h"

namespace iprgan {

// ---- optional per-kernel timing (bench.py roofline): HIP events on the launch stream ------------
struct ProfSlot {
  const char* name;
  long long launches;
  double ms, flops;
};
static ProfSlot g_slots[] = {
    {"gconv_kernel<128x128>", 0, 0, 0}, {"gconv_kernel<128x64>", 0, 0, 0},
    {"gconv_kernel<64x64>", 0, 0, 0},   {"gconv_kernel<128x32>", 0, 0, 0},
    {"wgrad_kernel<128x128>", 0, 0, 0}, {"wgrad_kernel<128x64>", 0, 0, 0},
    {"wgrad_kernel<64x64>", 0, 0, 0},   {"wgrad_kernel<32x128>", 0, 0, 0},
    {"gconv_kernel<64x128>", 0, 0, 0},  {"gconv_kernel<128x128,8w>", 0, 0, 0},
    {"gconv_kernel<128x64,8w>", 0, 0, 0}, {"wgrad_kernel<128x128,8w>", 0, 0, 0},
    {"gconv_bf16_kernel", 0, 0, 0},     {"wgrad_bf16_kernel", 0, 0, 0},
    {"wgrad_t_kernel<128x128>", 0, 0, 0}, {"wgrad_t_kernel<128x64>", 0, 0, 0},
    {"wgrad_t_kernel<64x64>", 0, 0, 0},   {"wgrad_t_kernel<128x128,8w>", 0, 0, 0},
    {"fewin_conv_kernel", 0, 0, 0},
    {"gconv_pipe_kernel", 0, 0, 0},     {"gconv_pipe_kernel<256x64>", 0, 0, 0},
    {"wgrad_halo_kernel", 0, 0, 0},     {"wgrad_rgb_kernel", 0, 0, 0},
    {"gconv_pipe_f32_kernel", 0, 0, 0}, {"gconv_phase4_kernel", 0, 0, 0},
    {"wgrad_halo_f32_kernel", 0, 0, 0}, {"gconv_pipe8_kernel", 0, 0, 0},
    {"gconv_x3_kernel", 0, 0, 0},       {"wgrad_x3_kernel", 0, 0, 0},
    {"gconv_x3p_kernel", 0, 0, 0},      {"wgrad_x3h_kernel", 0, 0, 0},
    {"gconv_x3h_kernel", 0, 0, 0},
    {"gconv_x3p16_kernel", 0, 0, 0},    {"gconv_x3ws_kernel", 0, 0, 0},
};
static const int g_nslots = sizeof(g_slots) / sizeof(g_slots[0]);
struct ProfRec { hipEvent_t a, b; int slot; double flops; int tag; };
static bool g_prof_on = false;
// per-layer view of the same records: the entry points name the layer (pass + geometry) before they launch
struct LayerAgg { long long launches; double ms, flops; };
static std::vector<std::string> g_tag_names;
static std::map<std::string, int> g_tag_index;
static std::vector<LayerAgg> g_tag_agg;
static int g_cur_tag = -1;
static void prof_tag(const char* pass, const iprgan_conv_desc* d) {
  if (!g_prof_on) return;
  char buf[160];
  // (x<kind> y<kind>: storage kinds of the layer's input / output side - bench.py prices a layer's algorithmic bytes from them)
  snprintf(buf, sizeof(buf), "%-5s B%d %dx%d %d->%d k%dx%d s%d p%d%s%s x%d y%d", pass, d->B, d->H, d->W, d->Cin, d->Cout, d->KH,
           d->KW, d->stride, d->pad, d->transposed ? " T" : "", d->pad_mode ? " reflect" : "", d->x_bf16, d->y_bf16);
  auto it = g_tag_index.find(buf);
  if (it == g_tag_index.end()) {
    g_cur_tag = (int)g_tag_names.size();
    g_tag_index[buf] = g_cur_tag;
    g_tag_names.push_back(buf);
    g_tag_agg.push_back(LayerAgg{0, 0, 0});
  } else {
    g_cur_tag = it->second;
  }
}
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free_events;
static hipEvent_t prof_event() {
  hipEvent_t e;
  if (!g_free_events.empty()) { e = g_free_events.back(); g_free_events.pop_back(); return e; }
  (void)hipEventCreate(&e);
  return e;
}
bool prof_events(int slot, double flops, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof_on || g_recs.size() >= 65536) return false;
  ProfRec r;
  r.a = prof_event(); r.b = prof_event(); r.slot = slot; r.flops = flops; r.tag = g_cur_tag;
  g_recs.push_back(r);
  *start = r.a; *stop = r.b;
  return true;
}


// FAST: zero padding and Cs % 32 == 0, so every 32-wide K step lies inside ONE tap: the tap walk is
// wave-uniform (scalar registers), borders are handled by the buffer bounds check (no branches).
// !FAST: reflect padding and/or Cs in {4,8,16} (taps change inside a K step; RGB layers).
// BF16 (math mode 1, BASELINE config "DCGAN 128x128 bf16"): tensors stay fp32 in HBM (fp32 master weights and
// activations); the tiles are rounded to bf16 on their way into LDS and multiplied by v_mfma_f32_32x32x16_bf16
// with fp32 accumulation - 16x the fp32 MFMA rate, half the LDS bytes, the same loader and epilogue.  An LDS row
// is then 32 k x 2 B = 64 B = four 16-byte chunks of 8 consecutive k; lane half h of MFMA kk reads chunk
// 2 kk + h; chunk c of row r sits at c ^ ((r >> 2) & 3), which makes the 16-lane groups of ds_read_b128 hit 16
// distinct slots of the 256-B bank line (4 rows).
// STATS: the epilogue also emits per-tile COLUMN sums (s1 = sum t, s2 = sum t^2 over the tile's valid rows) into
// a.stat_part[(tile row lq)][2][Ns], combined afterwards in fixed order (deterministic).  stat_mode 1: t = the
// accumulator before bias and activation - the batch / instance statistics of the norm layer that follows the
// convolution come for free instead of from another pass over its output (mean = bias + s1/M, var = s2/M - (s1/M)^2);
// stat_mode 2: t = the value stored - column sums of a backward-data result = the bias gradient of the layer below.
// IN16 (math mode 2, "bf16 activations"): both operands ARE bf16 in HBM (activations with C4 % 32 == 0 are stored as
// bf16 by every producer, prepared weights are emitted as bf16): a 16-byte load is 8 K-elements and goes to the LDS
// image as it is - half the loader bytes of the fp32-in-HBM form, no conversion in the staging path.
// SPLIT (math mode 2, "fp32 as three bf16 terms"): fp32 operands, fp32-grade products, on the bf16 matrix pipe.  Every
// element is split on its way into LDS into x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 3 x 8
// mantissa bits, the subtractions are exact), three bf16 images per operand, and a product a * b is accumulated as the
// six terms l*h' + h*l' + m*m' + m*h' + h*m' + h*h' (small first; m*l', l*m', l*l' are below 2^-26 of the product and
// dropped).  Six v_mfma_f32_32x32x16_bf16 (8 passes for 16 k) replace eight v_mfma_f32_32x32x2_f32 (16 passes for 2 k
// each): 6 x 8 = 48 matrix-pipe passes per 16 k instead of 128, with the accumulation in fp32 as before.
template <int WGM, int WGN, int WM, int WN, bool FAST, int NBUF, int BK, bool BF16 = false, bool STATS = false, bool IN16 = false, bool SPLIT = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gconv_kernel(const GConvArgs a) {
  static_assert(!BF16 || BK == 32 || BK == 64 || BK == 128, "bf16 tiles: K steps of 32, 64 or 128");
  static_assert(!IN16 || (BF16 && FAST), "bf16 operands in HBM: bf16 tiles on the one-tap-per-step path");
  static_assert(!SPLIT || (BF16 && FAST), "split tiles: the bf16 image, one tap per K step");
  constexpr bool IN3P = IN16 && SPLIT;         // operands ARE three bf16 planes in HBM (storage kind 2): loaded plane by plane, no split arithmetic
  constexpr int NLD = IN3P ? 3 : 1;
  constexpr int NPL = SPLIT ? 3 : 1;            // bf16 images per operand
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;            // threads: one wave per (32*WM)x(32*WN) sub-tile
  constexpr int CH = IN16 ? BK / 8 : BK / 4;    // 16-byte chunks per tile row the LOADER handles (K step = BK elements)
  constexpr int ESZ = IN16 ? 2 : 4;             // bytes per operand element in HBM
  constexpr int KG = BK / 4;                    // 4-channel groups per K step (the unit of the tap walk)
  constexpr int RP = NT / CH;                   // tile rows loaded per pass of the block
  constexpr int RA = BM / RP, RB = BN / RP;
  constexpr int CHB = BK / 8;                   // bf16 mode: 16-byte chunks per tile row
  constexpr int PLANE4 = (BM + BN) * (BF16 ? CHB : CH);        // 16-byte chunks of one image pair (A rows, then B rows)
  constexpr int TILE4 = NPL * PLANE4;                          // 16-byte chunks per stage buffer
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];

  // logical tile order (see xcd_remap): n tiles fastest, then the sub-pixel phases, then m tiles, so the
  // blocks that read the same input rows run together on one XCD and share its L2
  const unsigned lt = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                gridDim.x * gridDim.y * gridDim.z);
  const unsigned lq = lt / gridDim.y;
  const int zi = (int)(lq % gridDim.z);
  const int pz = a.ksplit > 1 ? 0 : zi;          // blockIdx.z: sub-pixel phase, or K split of a single-phase geometry
  const int pM = a.ph[pz].M;
  const int m0 = (int)(lq / gridDim.z) * BM, n0 = (int)(lt % gridDim.y) * BN;
  if (m0 >= pM) {
    if (STATS) {                 // a tile past the end of a short phase still owns a row of partials: zeros
      for (int c = threadIdx.x; c < BN; c += NT)
        if (n0 + c < a.Ns) { a.stat_part[((size_t)lq * 2) * a.Ns + n0 + c] = 0.f; a.stat_part[((size_t)lq * 2 + 1) * a.Ns + n0 + c] = 0.f; }
    }
    return;
  }
  // phase constants into scalars once (the K loop must not re-read the kernel arguments)
  const int p_ntap = a.ph[pz].ntap, p_tw = a.ph[pz].tw;
  int p_steps = (a.ph[pz].steps * 32 + BK - 1) / BK, s_begin = 0;
  if (a.ksplit > 1) {
    const int per = (p_steps + a.ksplit - 1) / a.ksplit;
    s_begin = zi * per;
    p_steps = s_begin + per < p_steps ? s_begin + per : p_steps;      // = end step of this split
  }
  const int p_dy0 = a.ph[pz].dy0, p_dx0 = a.ph[pz].dx0, p_dys = a.ph[pz].dys, p_dxs = a.ph[pz].dxs;
  const int p_wbase = a.ph[pz].wbase, p_wsy = a.ph[pz].wsy, p_wsx = a.ph[pz].wsx;
  const int p_owg = a.ph[pz].owg, plane = a.ph[pz].ohg * a.ph[pz].owg;
  const FastDiv d_plane = a.ph[pz].d_plane, d_owg = a.ph[pz].d_owg, d_tw = a.ph[pz].d_tw;
  const int IH = a.IH, IW = a.IW, Cs = a.Cs;
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int chunk = tid % CH, lrow = tid / CH;
  // LDS chunk swizzle: conflict-free ds_read_b128 fragments (row stride 128 B: 2 rows per bank row; 256 B: 1)
  auto swz = [](int r) { return CH == 8 ? ((r >> 1) & 7) : (r & 15); };
  // bf16 image: a row holds BK bf16 = CHB 16-byte chunks (64 / 128 / 256 bytes): 4 / 2 / 1 rows per 256-byte bank line
  auto swzb = [](int r) { return CHB == 4 ? ((r >> 2) & 3) : CHB == 8 ? ((r >> 1) & 7) : (r & 15); };

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

  // per-row gather state (rows are fixed for the whole K loop): byte offset of (b, y*isy, x*isx, c=chunk*4)
  int aiy[RA], aix[RA];
  unsigned arow[RA], wrow[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + lrow + RP * i;
    if (m < pM) {
      const int b = fdiv(m, d_plane);
      const int rem = m - b * plane;
      const int y = fdiv(rem, d_owg);
      const int x = rem - y * p_owg;
      aiy[i] = y * a.isy;
      aix[i] = x * a.isx;
      arow[i] = (unsigned)(((b * IH + aiy[i]) * IW + aix[i]) * Cs) * (unsigned)ESZ + (FAST ? chunk * 16u : 0u);
    } else {
      aiy[i] = ROW_INVALID; aix[i] = 0; arow[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int r = n0 + lrow + RP * i;
    const unsigned base = a.wmod > 0 ? (unsigned)((r % a.wmod) * a.Kp + (r / a.wmod) * a.wk1) : (unsigned)(r * a.Kp);
    wrow[i] = base * (unsigned)ESZ + (FAST ? chunk * 16u : 0u);
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // split tiles: the five small terms of a product block get their own accumulator (conv_x3.hip: every bf16 MFMA into a
  // large accumulator costs about an ulp of it; summed among themselves they cost 2^-7 of that), merged before the epilogue
  f32x16 accs[SPLIT ? WM : 1][SPLIT ? WN : 1];
  if (SPLIT) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[SPLIT ? i : 0][SPLIT ? j : 0][r] = 0.f;
  }

  f32x4 ra[NLD][RA], rb[NLD][RB];
  // wave-uniform tap walk for the FAST path
  int u_c4 = 0, u_ty = 0, u_tx = 0;
  if (FAST && s_begin > 0) {                      // K split: start the walk at step s_begin
    const int q0 = s_begin * KG, t0 = q0 / a.c4n;
    u_c4 = q0 - t0 * a.c4n;
    u_ty = t0 / p_tw;
    u_tx = t0 - u_ty * p_tw;
  }

  auto gload = [&](int step) {
    if (FAST) {
      const int dy = p_dy0 + u_ty * p_dys, dx = p_dx0 + u_tx * p_dxs;
      const int tapoff = ((dy * IW + dx) * Cs + u_c4 * 4) * ESZ;                       // bytes, uniform
      const unsigned wk = (unsigned)((p_wbase + u_ty * p_wsy + u_tx * p_wsx) * Cs + u_c4 * 4) * (unsigned)ESZ;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int iy = aiy[i] + dy, ix = aix[i] + dx;
        bool ok;
        unsigned off;
        if (reflect) {            // wave-uniform branch: ReflectionPad2d folded into the gather
          ok = aiy[i] != ROW_INVALID;
          const int ry = reflect_idx(iy, IH), rx = reflect_idx(ix, IW);
          off = arow[i] + (unsigned)((((ry - aiy[i]) * IW + (rx - aix[i])) * Cs + u_c4 * 4) * ESZ);
        } else {
          ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
          off = arow[i] + (unsigned)tapoff;
        }
#pragma unroll
        for (int p = 0; p < NLD; ++p) ra[p][i] = buf_load4(rs_in, ok ? off + (unsigned)p * a.in_ps : OOB_OFFSET);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int p = 0; p < NLD; ++p) rb[p][i] = buf_load4(rs_wt, wrow[i] + wk + (unsigned)p * a.wt_ps);
      u_c4 += KG;
      if (u_c4 >= a.c4n) { u_c4 = 0; if (++u_tx == p_tw) { u_tx = 0; ++u_ty; } }
    } else {
      const int q = step * CH + chunk;
      const int t = fdiv(q, a.d_c4n);
      const int c4 = q - t * a.c4n;
      const bool valid = t < p_ntap;
      const int ty = fdiv(t, d_tw);
      const int tx = t - ty * p_tw;
      const int dy = p_dy0 + ty * p_dys, dx = p_dx0 + tx * p_dxs;
      const unsigned wk = (unsigned)((p_wbase + ty * p_wsy + tx * p_wsx) * Cs + c4 * 4) * 4u;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int iy = aiy[i] + dy, ix = aix[i] + dx;
        bool ok = valid && aiy[i] != ROW_INVALID;
        if (a.pad_mode == IPRGAN_PAD_REFLECT) {
          iy = reflect_idx(iy, IH);
          ix = reflect_idx(ix, IW);
        } else {
          ok = ok && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        }
        const int tapoff = (((iy - aiy[i]) * IW + (ix - aix[i])) * Cs + c4 * 4) * 4;
        ra[0][i] = buf_load4(rs_in, ok ? arow[i] + (unsigned)tapoff : OOB_OFFSET);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) rb[0][i] = buf_load4(rs_wt, valid ? wrow[i] + wk : OOB_OFFSET);
    }
  };
  auto lstore = [&](int buf) {
    if (IN16) {                  // the 16 bytes loaded ARE chunk `chunk` of the bf16 row image (IN3P: of each plane's image)
      f32x4* A16 = lds + buf * TILE4;
      f32x4* B16 = A16 + BM * CHB;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int r = lrow + RP * i;
#pragma unroll
        for (int p = 0; p < NLD; ++p) A16[p * PLANE4 + r * CHB + (chunk ^ swzb(r))] = ra[p][i];
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int r = lrow + RP * i;
#pragma unroll
        for (int p = 0; p < NLD; ++p) B16[p * PLANE4 + r * CHB + (chunk ^ swzb(r))] = rb[p][i];
      }
      return;
    }
    if (SPLIT) {                 // three images (h, m, l) of the bf16 layout below
      bf16x4* A8 = (bf16x4*)(lds + buf * TILE4);
      bf16x4* B8 = A8 + BM * CHB * 2;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int r = lrow + RP * i;
        bf16x4 t3[3];
        split3_bf16(ra[0][i], t3);
#pragma unroll
        for (int p = 0; p < 3; ++p) A8[p * PLANE4 * 2 + (r * CHB + ((chunk >> 1) ^ swzb(r))) * 2 + (chunk & 1)] = t3[p];
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int r = lrow + RP * i;
        bf16x4 t3[3];
        split3_bf16(rb[0][i], t3);
#pragma unroll
        for (int p = 0; p < 3; ++p) B8[p * PLANE4 * 2 + (r * CHB + ((chunk >> 1) ^ swzb(r))) * 2 + (chunk & 1)] = t3[p];
      }
      return;
    }
    if (BF16) {                  // this thread's 4 floats are half of 16-byte chunk (chunk >> 1)
      bf16x4* A8 = (bf16x4*)(lds + buf * TILE4);
      bf16x4* B8 = A8 + BM * CHB * 2;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int r = lrow + RP * i;
        A8[(r * CHB + ((chunk >> 1) ^ swzb(r))) * 2 + (chunk & 1)] = to_bf16x4(ra[0][i]);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int r = lrow + RP * i;
        B8[(r * CHB + ((chunk >> 1) ^ swzb(r))) * 2 + (chunk & 1)] = to_bf16x4(rb[0][i]);
      }
      return;
    }
    f32x4* A4 = lds + buf * TILE4;
    f32x4* B4 = A4 + BM * CH;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int r = lrow + RP * i;
      A4[r * CH + (chunk ^ swz(r))] = ra[0][i];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int r = lrow + RP * i;
      B4[r * CH + (chunk ^ swz(r))] = rb[0][i];
    }
  };
  auto compute = [&](int buf) {
    const int half = lane >> 5, l31 = lane & 31;
    if (SPLIT) {
      const bf16x8* A16 = (const bf16x8*)(lds + buf * TILE4);
      const bf16x8* B16 = A16 + BM * CHB;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 af[3][WM], bf[3][WN];
        const int c = 2 * kk + half;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            const int r = (wm * WM + i) * 32 + l31;
            af[p][i] = A16[p * PLANE4 + r * CHB + (c ^ swzb(r))];
          }
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int r = (wn * WN + j) * 32 + l31;
            bf[p][j] = B16[p * PLANE4 + r * CHB + (c ^ swzb(r))];
          }
        }
        // (A term, B term): l h', h l', m m', m h', h m', h h' - consecutive MFMAs go to different accumulators
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
          const int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              if (t < 5) accs[SPLIT ? i : 0][SPLIT ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bf[pb][j], accs[SPLIT ? i : 0][SPLIT ? j : 0], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bf[pb][j], acc[i][j], 0, 0, 0);
            }
        }
      }
      return;
    }
    if (BF16) {
      const bf16x8* A16 = (const bf16x8*)(lds + buf * TILE4);
      const bf16x8* B16 = A16 + BM * CHB;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 af[WM], bf[WN];
        const int c = 2 * kk + half;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const int r = (wm * WM + i) * 32 + l31;
          af[i] = A16[r * CHB + (c ^ swzb(r))];
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int r = (wn * WN + j) * 32 + l31;
          bf[j] = B16[r * CHB + (c ^ swzb(r))];
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      return;
    }
    const f32x4* A4 = lds + buf * TILE4;
    const f32x4* B4 = A4 + BM * CH;
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      f32x4 af[WM], bf[WN];
      const int c = 2 * kq + half;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int r = (wm * WM + i) * 32 + l31;
        af[i] = A4[r * CH + (c ^ swz(r))];
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int r = (wn * WN + j) * 32 + l31;
        bf[j] = B4[r * CH + (c ^ swz(r))];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  if (p_steps > s_begin) {
    gload(s_begin);
    lstore(0);
    __syncthreads();
    for (int s = s_begin; s < p_steps; ++s) {
      const bool more = s + 1 < p_steps;
      if (more) gload(s + 1);
      if (NBUF == 2) {           // double-buffered LDS: one barrier per K step, 2 blocks per CU
        compute((s - s_begin) & 1);
        if (more) lstore((s + 1 - s_begin) & 1);
        __syncthreads();
      } else {                   // single LDS buffer (half the LDS -> 3-4 blocks per CU), two barriers
        compute(0);
        __syncthreads();
        if (more) lstore(0);
        __syncthreads();
      }
    }
  }

  if (SPLIT) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] += accs[SPLIT ? i : 0][SPLIT ? j : 0];
  }
  gconv_epilogue<WGM, WGN, WM, WN, STATS>(a, acc, lds, pz, zi, lq, m0, n0);
}

// ------------------------------------------------------------------------------------------
// Few-INPUT-channel convolutions (RGB stems 3 -> 64: Conv2d forward; and the backward-data of RGB heads, which has the
// same form): K = taps * 4 is far too short for the MFMA tiles (36 padded to 64: two K steps, the launch is all
// prologue and epilogue) and the layer is bound by WRITING its 64-channel output.  Direct form on the vector ALU:
// a block owns a 16 x 16 tile of output pixels and 64 output channels; the input halo (4 channels = 16 bytes per
// pixel) and the block's weights [tap][n][c] sit in LDS; lane = (channel quad q, pixel column): every thread
// accumulates 16 pixels (one column of the tile) x 4 channels, reading one 16-byte input vector per (pixel, tap)
// and four 16-byte weight vectors per tap.  Stores: a wave writes 4 pixels x 256 contiguous bytes per instruction.
// ------------------------------------------------------------------------------------------
// Epilogue of one 4-channel group of the few-input-channel kernels (same order of operations as gconv_kernel's: pair
// scale, bias, activation, fused derivative, residual).  SIMPLE: activation and fused derivative are none / ReLU /
// LeakyReLU, folded into one select-and-multiply each (neg = factor of the non-positive side); the general form inlines
// tanh / sigmoid per element, and 64 copies of that made these short kernels 11 k instructions long (94 KB of code,
// more than the instruction cache) - their K loop is 27..196 deep, so the epilogue IS the kernel.
template <bool SIMPLE>
__device__ __forceinline__ void fewin_store(const GConvArgs& a, f32x4 v, const f32x4 bias4, float rs, float neg_act,
                                            float neg_aux, size_t idx, int n) {
  if (a.rs0) v *= rs;
  v += bias4;
  if (SIMPLE) {                  // neg_act = 1 (none), 0 (ReLU: select, so that -inf and NaN behave as in act_apply) or the slope
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : (neg_act == 0.f ? 0.f : v[k] * neg_act);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], a.act, a.slope);
  }
  if (a.aux) {
    const f32x4 o = a.aux16 ? ld_bf16x4(a.aux, idx) : *(const f32x4*)(a.aux + idx);
    if (SIMPLE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= o[k] > 0.f ? 1.f : neg_aux;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= act_grad_from_out(o[k], a.aux_act, a.aux_slope);
    }
  }
  if (a.out16 == 2) {              // three-plane output (and residual): plane stride out_ps bytes
    const size_t ps = a.out_ps / 2;
    if (a.res) v += ld_bf16x4(a.res, idx) + (ld_bf16x4(a.res, idx + ps) + ld_bf16x4(a.res, idx + 2 * ps));
    bf16x4 t3[3];
    split3_bf16(v, t3);
#pragma unroll
    for (int p = 0; p < 3; ++p) *(bf16x4*)((__bf16*)a.out + idx + p * ps) = t3[p];
    return;
  }
  if (a.res) v += a.out16 ? ld_bf16x4(a.res, idx) : *(const f32x4*)(a.res + idx);
  if (a.out16) *(bf16x4*)((__bf16*)a.out + idx) = to_bf16x4(v);
  else *(f32x4*)(a.out + idx) = v;
}
__device__ __forceinline__ f32x4 fewin_bias4(const GConvArgs& a, int n) {
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (n + k < a.N) b[k] = a.bias[n + k];
  }
  return b;
}
static bool fewin_simple_act(int act) { return act == IPRGAN_ACT_NONE || act == IPRGAN_ACT_RELU || act == IPRGAN_ACT_LRELU; }
static float fewin_neg(int act, float slope) { return act == IPRGAN_ACT_NONE ? 1.f : act == IPRGAN_ACT_RELU ? 0.f : slope; }

#define FEWIN_T 16
#define FEWIN_TAPS 16          // taps of weights resident in LDS at a time (fewin_conv_kernel): 16 KB
#ifndef FEWIN_WAVES
#define FEWIN_WAVES 5          // waves per SIMD the register allocation of fewin_mfma_kernel must allow
#endif
template <bool SIMPLE>
__global__ __launch_bounds__(256) void fewin_conv_kernel(const GConvArgs a, int lds_w, int lds_h, float neg_act, float neg_aux) {
  extern __shared__ __attribute__((aligned(16))) f32x4 flds[];
  const Phase& ph = a.ph[0];
  const int ntap = ph.ntap, tw = ph.tw, th = ph.th;
  f32x4* X = flds;                                   // [lds_h][lds_w] input pixels (4 channels each)
  f32x4* Wl = flds + lds_h * lds_w;                  // [ntap][4 input channels][16 quads]: weights of 4 output channels
  const int tiles_x = (a.OW + FEWIN_T - 1) / FEWIN_T, tiles_y = (a.OH + FEWIN_T - 1) / FEWIN_T;
  const int tile = blockIdx.x, b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * FEWIN_T, x0 = (tr % tiles_x) * FEWIN_T;
  const int n0 = blockIdx.y * 64;
  const int tid = threadIdx.x, q = tid & 15, pg = tid >> 4;
  // input window of the tile: rows iy = y * isy + dy0 + t * dys for y in [y0, y0 + 15], t in [0, th)
  const int dyl = ph.dys < 0 ? (th - 1) * ph.dys : 0, dxl = ph.dxs < 0 ? (tw - 1) * ph.dxs : 0;
  const int iy_lo = y0 * a.isy + ph.dy0 + dyl, ix_lo = x0 * a.isx + ph.dx0 + dxl;
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;
  for (int i = tid; i < lds_h * lds_w; i += 256) {
    const int r = i / lds_w, c = i - r * lds_w;
    int iy = iy_lo + r, ix = ix_lo + c;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool ok = true;
    if (reflect) { iy = reflect_idx(iy, a.IH); ix = reflect_idx(ix, a.IW); ok = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW; }
    else ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
    if (ok) v = *(const f32x4*)(a.in + ((size_t)(b * a.IH + iy) * a.IW + ix) * 4);
    X[i] = v;
  }
  f32x4 acc[FEWIN_T];
#pragma unroll
  for (int i = 0; i < FEWIN_T; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cx = pg * a.isx - dxl;                  // LDS column of this thread's pixel for tap offset 0
  // The weights pass through LDS in chunks of FEWIN_TAPS taps (k7 / k9 stems: 49 / 81 taps x 1 KB would leave one block
  // per CU - one wave per SIMD and nothing to hide a latency behind; the k9 backward-data of SRGAN ran at 26 TFLOP/s).
  for (int t0 = 0; t0 < ntap; t0 += FEWIN_TAPS) {
    const int tn = ntap - t0 < FEWIN_TAPS ? ntap - t0 : FEWIN_TAPS;
    if (t0) __syncthreads();                          // everyone is done with the previous chunk
    for (int i = tid; i < tn * 64; i += 256) {        // transposed while staged: Wl[t][c][quad] = (W[4 quad + k][t][c])_k
      const int tl = i >> 6, n = i & 63, t = t0 + tl;
      const int ty = t / tw, tx = t - ty * tw;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + n < a.Ns) v = *(const f32x4*)(a.wt + (size_t)(n0 + n) * a.Kp + (size_t)(ph.wbase + ty * ph.wsy + tx * ph.wsx) * 4);
      float* wf = (float*)Wl;
#pragma unroll
      for (int c = 0; c < 4; ++c) wf[((tl * 4 + c) * 16 + (n >> 2)) * 4 + (n & 3)] = v[c];
    }
    __syncthreads();                                  // (also covers the input window on the first trip)
    for (int tl = 0; tl < tn; ++tl) {
      const int t = t0 + tl;
      const int ty = t / tw, tx = t - ty * tw;
      // wc: this thread's 4 output channels for input channel c (vector x scalar FMAs: v_pk_fma_f32 pairs)
      const f32x4 w0 = Wl[(tl * 4 + 0) * 16 + q], w1 = Wl[(tl * 4 + 1) * 16 + q], w2 = Wl[(tl * 4 + 2) * 16 + q], w3 = Wl[(tl * 4 + 3) * 16 + q];
      const f32x4* xp = X + (ty * ph.dys - dyl) * lds_w + cx + tx * ph.dxs;
      const int xstep = a.isy * lds_w;
#pragma unroll
      for (int i = 0; i < FEWIN_T; ++i) {
        const f32x4 xin = xp[i * xstep];
        acc[i] += w0 * xin.x;
        acc[i] += w1 * xin.y;
        acc[i] += w2 * xin.z;
        acc[i] += w3 * xin.w;
      }
    }
  }
  const int n = n0 + 4 * q, ox = x0 + pg;
  if (n >= a.Ns || ox >= a.OW) return;
  float rsc0 = 1.f, rsc1 = 1.f;
  if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
  const float rs = b < (a.B >> 1) ? rsc0 : rsc1;
  const f32x4 bias4 = fewin_bias4(a, n);
#pragma unroll
  for (int i = 0; i < FEWIN_T; ++i) {
    const int oy = y0 + i;
    if (oy >= a.OH) break;
    fewin_store<SIMPLE>(a, acc[i], bias4, rs, neg_act, neg_aux, ((size_t)(b * a.OH + oy) * a.OW + ox) * a.Ns + n, n);
  }
}

// Three-plane output form of fewin_conv_kernel (round 6): the stem writes 6 bytes per element and is bound by that write - with
// a thread owning 4 channels a plane store is 8 bytes per lane, 512 contiguous bytes per wave-instruction, and the kernel moved
// 3.9 TB/s of output.  Here a thread owns EIGHT consecutive channels of one pixel column (tile = 32 pixels wide x 8 rows, the same
// 64 accumulator floats): a plane store is 16 bytes per lane and a wave-instruction covers 8 consecutive pixels x 128 bytes = 1 KB
// of one plane; the fused-derivative operand (h plane) is one 16-byte load.  SIMPLE activations, no residual, Ns % 64 == 0.
#define FEWIN8_W 32
#define FEWIN8_H 8
__global__ __launch_bounds__(256) void fewin_conv8_kernel(const GConvArgs a, int lds_w, int lds_h, float neg_act, float neg_aux) {
  extern __shared__ __attribute__((aligned(16))) f32x4 flds[];
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const Phase& ph = a.ph[0];
  const int ntap = ph.ntap, tw = ph.tw, th = ph.th;
  f32x4* X = flds;                                   // [lds_h][lds_w] input pixels (4 channels each)
  f32x4* Wl = flds + lds_h * lds_w;                  // [ntap chunk][4 input channels][16 quads]
  const int tiles_x = (a.OW + FEWIN8_W - 1) / FEWIN8_W, tiles_y = (a.OH + FEWIN8_H - 1) / FEWIN8_H;
  const int tile = blockIdx.x, b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * FEWIN8_H, x0 = (tr % tiles_x) * FEWIN8_W;
  const int n0 = blockIdx.y * 64;
  const int tid = threadIdx.x, o = tid & 7, pc = tid >> 3;
  const int dyl = ph.dys < 0 ? (th - 1) * ph.dys : 0, dxl = ph.dxs < 0 ? (tw - 1) * ph.dxs : 0;
  const int iy_lo = y0 * a.isy + ph.dy0 + dyl, ix_lo = x0 * a.isx + ph.dx0 + dxl;
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;
  for (int i = tid; i < lds_h * lds_w; i += 256) {
    const int r = i / lds_w, c = i - r * lds_w;
    int iy = iy_lo + r, ix = ix_lo + c;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool ok = true;
    if (reflect) { iy = reflect_idx(iy, a.IH); ix = reflect_idx(ix, a.IW); ok = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW; }
    else ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
    if (ok) v = *(const f32x4*)(a.in + ((size_t)(b * a.IH + iy) * a.IW + ix) * 4);
    X[i] = v;
  }
  f32x4 acc[FEWIN8_H][2];
#pragma unroll
  for (int i = 0; i < FEWIN8_H; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cx = pc * a.isx - dxl;
  for (int t0 = 0; t0 < ntap; t0 += FEWIN_TAPS) {
    const int tn = ntap - t0 < FEWIN_TAPS ? ntap - t0 : FEWIN_TAPS;
    if (t0) __syncthreads();
    for (int i = tid; i < tn * 64; i += 256) {        // Wl[t][c][quad] = (W[4 quad + k][t][c])_k, as fewin_conv_kernel
      const int tl = i >> 6, n = i & 63, t = t0 + tl;
      const int ty = t / tw, tx = t - ty * tw;
      const f32x4 v = *(const f32x4*)(a.wt + (size_t)(n0 + n) * a.Kp + (size_t)(ph.wbase + ty * ph.wsy + tx * ph.wsx) * 4);
      float* wf = (float*)Wl;
#pragma unroll
      for (int c = 0; c < 4; ++c) wf[((tl * 4 + c) * 16 + (n >> 2)) * 4 + (n & 3)] = v[c];
    }
    __syncthreads();
    for (int tl = 0; tl < tn; ++tl) {
      const int t = t0 + tl;
      const int ty = t / tw, tx = t - ty * tw;
      f32x4 w[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c) { w[c][0] = Wl[(tl * 4 + c) * 16 + 2 * o]; w[c][1] = Wl[(tl * 4 + c) * 16 + 2 * o + 1]; }
      const f32x4* xp = X + (ty * ph.dys - dyl) * lds_w + cx + tx * ph.dxs;
      const int xstep = a.isy * lds_w;
#pragma unroll
      for (int i = 0; i < FEWIN8_H; ++i) {
        const f32x4 xin = xp[i * xstep];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc[i][h] += w[0][h] * xin.x;
          acc[i][h] += w[1][h] * xin.y;
          acc[i][h] += w[2][h] * xin.z;
          acc[i][h] += w[3][h] * xin.w;
        }
      }
    }
  }
  const int n = n0 + 8 * o, ox = x0 + pc;
  if (ox >= a.OW) return;
  float rsc0 = 1.f, rsc1 = 1.f;
  if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
  const float rs = b < (a.B >> 1) ? rsc0 : rsc1;
  const f32x4 bias0 = fewin_bias4(a, n), bias1 = fewin_bias4(a, n + 4);
  const size_t ps = a.out_ps / 2;
#pragma unroll
  for (int i = 0; i < FEWIN8_H; ++i) {
    const int oy = y0 + i;
    if (oy >= a.OH) break;
    const size_t idx = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.Ns + n;
    f32x4 v0 = acc[i][0], v1 = acc[i][1];
    if (a.rs0) { v0 *= rs; v1 *= rs; }
    v0 += bias0; v1 += bias1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v0[k] = v0[k] > 0.f ? v0[k] : (neg_act == 0.f ? 0.f : v0[k] * neg_act);
      v1[k] = v1[k] > 0.f ? v1[k] : (neg_act == 0.f ? 0.f : v1[k] * neg_act);
    }
    if (a.aux) {             // fused derivative of the producer's activation: its sign, from the h plane (aux16 = 1) or fp32
      f32x4 o0, o1;
      if (a.aux16) {
        const u32x4_t r = *(const u32x4_t*)((const __bf16*)a.aux + idx);
        o0 = f32x4{__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                   __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
        o1 = f32x4{__builtin_bit_cast(float, r.z << 16), __builtin_bit_cast(float, r.z & 0xffff0000u),
                   __builtin_bit_cast(float, r.w << 16), __builtin_bit_cast(float, r.w & 0xffff0000u)};
      } else {
        o0 = *(const f32x4*)(a.aux + idx); o1 = *(const f32x4*)(a.aux + idx + 4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { v0[k] *= o0[k] > 0.f ? 1.f : neg_aux; v1[k] *= o1[k] > 0.f ? 1.f : neg_aux; }
    }
    unsigned p0[3], p1[3], p2[3], p3[3];
    split3_pair(v0.x, v0.y, p0[0], p0[1], p0[2]);
    split3_pair(v0.z, v0.w, p1[0], p1[1], p1[2]);
    split3_pair(v1.x, v1.y, p2[0], p2[1], p2[2]);
    split3_pair(v1.z, v1.w, p3[0], p3[1], p3[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) *(u32x4_t*)((__bf16*)a.out + idx + (size_t)p * ps) = u32x4_t{p0[p], p1[p], p2[p], p3[p]};
  }
}

// The same layers under IPRGAN_MATH_BF16: the 16x16 pixel tile is a 256-row M tile, (tap, channel) is the K index
// (4 channels per tap, K padded to 16), 64 output channels are two 32-column N tiles: 4 waves x (2 x 2) tiles of
// v_mfma_f32_32x32x16_bf16.  A fragments come straight out of the halo image in LDS (8 consecutive k = two taps x 4
// channels = two 8-byte reads at the taps' pixel offsets), so the gather costs no address arithmetic in HBM space and
// the layer runs at its output-write bound instead of the fp32 vector-ALU bound of fewin_conv_kernel.
// SIMPLE instantiation (every stem / head of the GANs): PERSISTENT - a block walks tiles blockIdx.x, + gridDim.x, ...:
// the 64 x K weights are converted and laid out once per block instead of once per 16 x 16 tile (704 16-byte loads per
// tile before), and the halo pixels of the next tile are loaded into registers while the current tile is multiplied and
// stored (a tile's life was 23 k cycles, 55 % of them parked behind two dependent rounds of global loads).
constexpr int fewin_waves(bool simple) { return simple ? 3 : FEWIN_WAVES; }      // persistent form: ~41 KB of LDS, three blocks per CU
template <bool SIMPLE>
__global__ __launch_bounds__(256, fewin_waves(SIMPLE)) void fewin_mfma_kernel(const GConvArgs a, int lds_w, int lds_h, int kpad, float neg_act, float neg_aux, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) f32x4 flds[];
  const Phase& ph = a.ph[0];
  const int ntap = ph.ntap, tw = ph.tw, th = ph.th;
  const int ksteps = (ntap * 4 + 15) / 16;
  bf16x4* X = (bf16x4*)flds;                                        // [lds_h][lds_w] halo pixels, 4 channels each
  __bf16* Wl = (__bf16*)(X + ((lds_h * lds_w + 1) & ~1));             // [64][kpad] weights, k = tap * 4 + c, zero past ntap * 4
  int* toff = (int*)(Wl + 64 * kpad);                               // [ksteps * 4] LDS pixel offset of every tap (clamped)
  u32x4* Tst = (u32x4*)(((uintptr_t)(toff + ksteps * 4) + 15) & ~(uintptr_t)15);      // [256 pixels][8 pieces]: staged bf16 output
  const int tiles_x = (a.OW + FEWIN_T - 1) / FEWIN_T, tiles_y = (a.OH + FEWIN_T - 1) / FEWIN_T;
  const int n0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dyl = ph.dys < 0 ? (th - 1) * ph.dys : 0, dxl = ph.dxs < 0 ? (tw - 1) * ph.dxs : 0;
  const bool reflect = a.pad_mode == IPRGAN_PAD_REFLECT;
  for (int i = tid; i < 64 * (kpad / 4); i += 256) {
    const int n = i / (kpad / 4), t = i - n * (kpad / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < ntap && n0 + n < a.Ns) {
      const int ty = t / tw, tx = t - ty * tw;
      v = *(const f32x4*)(a.wt + (size_t)(n0 + n) * a.Kp + (size_t)(ph.wbase + ty * ph.wsy + tx * ph.wsx) * 4);
    }
    *(bf16x4*)(Wl + n * kpad + t * 4) = to_bf16x4(v);
  }
  for (int t = tid; t < ksteps * 4; t += 256) {
    const int tc = t < ntap ? t : ntap - 1;                         // padding taps read a valid pixel against zero weights
    const int ty = tc / tw, tx = tc - ty * tw;
    toff[t] = (ty * ph.dys - dyl) * lds_w + tx * ph.dxs - dxl;
  }
  // halo pixels of this thread (i = tid + 256 q): position inside the halo, fixed for every tile
  constexpr int HQ = 3;                                             // up to 768 halo pixels (k9: 24 x 24)
  int hr[HQ], hc[HQ];
#pragma unroll
  for (int q = 0; q < HQ; ++q) {
    const int i = tid + 256 * q;
    hr[q] = i < lds_h * lds_w ? i / lds_w : -(1 << 20);
    hc[q] = i - (i / lds_w) * lds_w;
  }
  f32x4 hv[HQ];
  auto load_halo = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
    const int y0 = (tr / tiles_x) * FEWIN_T, x0 = (tr % tiles_x) * FEWIN_T;
    const int iy_lo = y0 * a.isy + ph.dy0 + dyl, ix_lo = x0 * a.isx + ph.dx0 + dxl;
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
      int iy = iy_lo + hr[q], ix = ix_lo + hc[q];
      bool ok;
      if (reflect) { iy = reflect_idx(iy, a.IH); ix = reflect_idx(ix, a.IW); ok = hr[q] >= 0 && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW; }
      else ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;          // (hr < 0: iy hugely negative)
      hv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) hv[q] = *(const f32x4*)(a.in + ((size_t)(b * a.IH + iy) * a.IW + ix) * 4);
    }
  };
  int tile = blockIdx.x;
  load_halo(tile);
  for (;;) {
  const int b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * FEWIN_T, x0 = (tr % tiles_x) * FEWIN_T;
#pragma unroll
  for (int q = 0; q < HQ; ++q)
    if (hr[q] >= 0) X[tid + 256 * q] = to_bf16x4(hv[q]);
  // LDS-only barriers in this loop: __syncthreads() would also drain vmcnt, i.e. wait for the previous tile's output
  // stores to COMPLETE (microseconds per tile, which a one-tile block never pays: it simply exits)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int nxt = tile + gridDim.x;
  if (SIMPLE && nxt < ntiles) load_halo(nxt);                       // in flight behind this tile's MFMAs and stores
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // row l31 of M tile i of this wave = pixel (4 * wave + 2 * i + (l31 >> 4), l31 & 15) of the 16 x 16 tile
  int pbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) pbase[i] = (4 * wave + 2 * i + (l31 >> 4)) * a.isy * lds_w + (l31 & 15) * a.isx;
  for (int kk = 0; kk < ksteps; ++kk) {
    const int o0 = toff[kk * 4 + 2 * half], o1 = toff[kk * 4 + 2 * half + 1];
    bf16x8 af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16x4 lo = X[pbase[i] + o0], hi = X[pbase[i] + o1];
      af[i] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = *(const bf16x8*)(Wl + (j * 32 + l31) * kpad + kk * 16 + half * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)           // SIMPLE: operand roles swapped (weights as A): the tile comes out transposed
        acc[i][j] = SIMPLE ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  }
  if constexpr (SIMPLE) {
    // Transposed accumulators (rows = channels, columns = pixels): lane (l31, half) holds pixel l31 of M tile i and, in
    // registers 4g..4g+3, channels 8g + 4 half + 0..3 of column block j.  One v_permlane32_swap per register pairs group
    // g of the lower half-wave with group g of the upper one: afterwards a lane owns 8 CONSECUTIVE channels of its pixel
    // (lower half: 16 gp .. +7, upper half: 16 gp + 8 .. +15) - no quad transposes, one pixel address per M tile, 16-byte
    // stores and 16-byte loads of the fused-derivative operand.  (The row-per-lane form retired ~35 vector instructions
    // per 4-channel store and ran at 2.7 TB/s of output; this form about half of that.)
    float rsc0 = 1.f, rsc1 = 1.f;
    if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
    const float rs = b < (a.B >> 1) ? rsc0 : rsc1;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? a.aux_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? a.out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
    // bf16 output: the 16-byte pieces go through LDS ([pixel][8 pieces], piece slot XOR pixel & 7) and leave as whole
    // 128-byte pixel rows (8 consecutive lanes = one pixel): a lane-per-pixel store touched 32 lines per instruction
    // with 32 bytes each and ran at 2.7 TB/s of output.  (fp32 output would need 64 KB of LDS per block: stored directly.)
    const bool staged = a.out16 != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = y0 + 4 * wave + 2 * i + (l31 >> 4), ox = x0 + (l31 & 15);
      const bool mok = oy < a.OH && ox < a.OW;
      const unsigned pix = (unsigned)((b * a.OH + oy) * a.OW + ox) * (unsigned)a.Ns;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          f32x4 v0, v1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // (inline asm: with __builtin_amdgcn_permlane32_swap hipcc 7.2 produced a tile whose 32 channels all carried
            // channel 0 here - the two v_nop are the VALU-write -> permlane wait states the compiler would have added)
            float lo = acc[i][j][8 * gp + k], hi = acc[i][j][8 * gp + 4 + k];
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
            v0[k] = lo;
            v1[k] = hi;
          }
#ifdef FEWIN_DBG_NOSWAP
          {
            const int nA = n0 + j * 32 + 16 * gp + 4 * half, nB = nA + 8;
            f32x4 qa, qb;
            for (int k = 0; k < 4; ++k) { qa[k] = acc[i][j][8 * gp + k]; qb[k] = acc[i][j][8 * gp + 4 + k]; }
            buf_store4(rs_out, mok ? (pix + nA) * 4u : OOB_OFFSET, qa);
            buf_store4(rs_out, mok ? (pix + nB) * 4u : OOB_OFFSET, qb);
            continue;
          }
#endif
          const int n = n0 + j * 32 + 16 * gp + 8 * half;
          const bool ok = mok && n < a.Ns;                      // Ns % 8 == 0 (launcher)
          const unsigned e = pix + (unsigned)n;
          if (a.rs0) { v0 *= rs; v1 *= rs; }
          if (a.bias) {      // the bias vector has N floats: dword loads through a descriptor of exactly that size (zeros past it)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v0[k] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bias, (unsigned)(n + k) * 4u, 0, 0));
              v1[k] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bias, (unsigned)(n + 4 + k) * 4u, 0, 0));
            }
          }
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
              const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_aux, ok ? e * 2u : OOB_OFFSET, 0, 0);
              o0 = f32x4{__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                         __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
              o1 = f32x4{__builtin_bit_cast(float, r.z << 16), __builtin_bit_cast(float, r.z & 0xffff0000u),
                         __builtin_bit_cast(float, r.w << 16), __builtin_bit_cast(float, r.w & 0xffff0000u)};
            } else {
              o0 = buf_load4(rs_aux, ok ? e * 4u : OOB_OFFSET);
              o1 = buf_load4(rs_aux, ok ? e * 4u + 16u : OOB_OFFSET);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { v0[k] *= o0[k] > 0.f ? 1.f : neg_aux; v1[k] *= o1[k] > 0.f ? 1.f : neg_aux; }
          }
          if (a.res) {
            if (a.out16) {
              const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? e * 2u : OOB_OFFSET, 0, 0);
              v0 += f32x4{__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                          __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
              v1 += f32x4{__builtin_bit_cast(float, r.z << 16), __builtin_bit_cast(float, r.z & 0xffff0000u),
                          __builtin_bit_cast(float, r.w << 16), __builtin_bit_cast(float, r.w & 0xffff0000u)};
            } else {
              v0 += buf_load4(rs_res, ok ? e * 4u : OOB_OFFSET);
              v1 += buf_load4(rs_res, ok ? e * 4u + 16u : OOB_OFFSET);
            }
          }
          if (a.out16) {
            const bf16x4 p0 = to_bf16x4(v0), p1 = to_bf16x4(v1);
            const u32x2 w0 = __builtin_bit_cast(u32x2, p0), w1 = __builtin_bit_cast(u32x2, p1);
            const int pl = (4 * wave + 2 * i + (l31 >> 4)) * 16 + (l31 & 15), piece = j * 4 + gp * 2 + half;
            Tst[pl * 8 + (piece ^ (pl & 7))] = u32x4{w0.x, w0.y, w1.x, w1.y};
          } else {
            buf_store4(rs_out, ok ? e * 4u : OOB_OFFSET, v0);
            buf_store4(rs_out, ok ? e * 4u + 16u : OOB_OFFSET, v1);
          }
        }
    }
    if (staged) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = q * 256 + tid, pl = idx >> 3, piece = idx & 7;
        const int oy = y0 + (pl >> 4), ox = x0 + (pl & 15), n = n0 + piece * 8;
        const bool ok = oy < a.OH && ox < a.OW && n < a.Ns;
        const u32x4 w = Tst[pl * 8 + (piece ^ (pl & 7))];
        __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, ok ? ((unsigned)((b * a.OH + oy) * a.OW + ox) * (unsigned)a.Ns + (unsigned)n) * 2u : OOB_OFFSET, 0, 0);
      }
    }
    if (nxt >= ntiles) return;
    tile = nxt;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // the halo image and the staged tile are free for the next tile
    continue;
  }
  // epilogue: quad transpose to 4 consecutive channels per lane (as in gconv_kernel), then fewin_store.  Row
  // ml = 8 g + 4 half + qp of M tile i is pixel (4 wave + 2 i + (g >> 1), 8 (g & 1) + 4 half + qp) of the 16 x 16 tile.
  const int qp = lane & 3, qcol = l31 & ~3;
  float rsc0 = 1.f, rsc1 = 1.f;
  if (a.rs0) { rsc0 = 1.f / *a.rs0; rsc1 = 1.f / *a.rs1; }
  const float rs = b < (a.B >> 1) ? rsc0 : rsc1;
  f32x4 bias4[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias4[j] = fewin_bias4(a, n0 + j * 32 + qcol);
  const int oyb = y0 + 4 * wave, oxb = x0 + 4 * half + qp;
  const size_t base = ((size_t)(b * a.OH + oyb) * a.OW + oxb) * a.Ns + n0 + qcol;
  if (SIMPLE && y0 + FEWIN_T <= a.OH && x0 + FEWIN_T <= a.OW && n0 + 64 <= a.Ns && !a.aux && !a.res) {
    // whole tile, plain epilogue (every stem forward of the four workloads): no bounds, no per-store flag tests; this
    // kernel retires ~40 instructions per 4-channel store, and at 64 stores of 8 bytes per wave-lane that IS its time
    const int rstep = a.OW * a.Ns;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float c0 = acc[i][j][4 * g], c1 = acc[i][j][4 * g + 1], c2 = acc[i][j][4 * g + 2], c3 = acc[i][j][4 * g + 3];
          quad_transpose(c0, c1, c2, c3, qp);
          f32x4 v = f32x4{c0, c1, c2, c3} * rs + bias4[j];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : (neg_act == 0.f ? 0.f : v[k] * neg_act);
          const size_t idx = base + (size_t)((2 * i + (g >> 1)) * rstep + 8 * (g & 1) * a.Ns + j * 32);
          if (a.out16) *(bf16x4*)((__bf16*)a.out + idx) = to_bf16x4(v);
          else *(f32x4*)(a.out + idx) = v;
        }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int oy = oyb + 2 * i + (g >> 1), ox = oxb + 8 * (g & 1);
      const bool mok = oy < a.OH && ox < a.OW;
      const size_t pix = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.Ns;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float c0 = acc[i][j][4 * g], c1 = acc[i][j][4 * g + 1], c2 = acc[i][j][4 * g + 2], c3 = acc[i][j][4 * g + 3];
        quad_transpose(c0, c1, c2, c3, qp);
        const int n = n0 + j * 32 + qcol;
        if (!mok || n >= a.Ns) continue;
        fewin_store<SIMPLE>(a, f32x4{c0, c1, c2, c3}, bias4[j], rs, neg_act, neg_aux, pix + n, n);
      }
    }
  }
  return;                                        // (general epilogue: one tile per block)
  }
}

// ------------------------------------------------------------------------------------------
// Tiny-Cout convolutions (RGB heads: 64->3 k3/k7/k9, and the data gradient of RGB stems): on the MFMA tile
// the 3 output channels would use 3 of 32 columns.  Instead:  T[tap][pix][n] = sum_c in[pix][c] W[n][tap][c]
// is ONE dense 1x1 GEMM with N' = taps*4 columns (full MFMA tiles), and out[pix][n] = sum_tap T[tap][pix+tap][n]
// is a coalesced gather over tap planes (HBM-bound).  Stride-1 forward- or backward-form geometries only.
// ------------------------------------------------------------------------------------------
__global__ void smalln_weight_kernel(const float* __restrict__ wt, float* __restrict__ w2, int Kp, int Cs,
                                     int ntap, int tw, int wbase, int wsy, int wsx, int rows_alloc, int bf16) {
  const int total = rows_alloc * Cs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % Cs, r = i / Cs;
    const int t = r >> 2, n = r & 3;
    float v = 0.f;
    if (t < ntap) {
      const int ty = t / tw, tx = t - ty * tw;
      const size_t src = (size_t)n * Kp + (size_t)(wbase + ty * wsy + tx * wsx) * Cs + c;
      if (bf16 == 2) {           // three planes: copied plane by plane (source plane stride 128 rows, destination rows_alloc)
        const size_t sps = (size_t)128 * Kp;
#pragma unroll
        for (int p = 0; p < 3; ++p) ((__bf16*)w2)[(size_t)p * total + i] = ((const __bf16*)wt)[p * sps + src];
        continue;
      }
      v = bf16 ? (float)((const __bf16*)wt)[src] : wt[src];
    }
    if (bf16 == 2) {
#pragma unroll
      for (int p = 0; p < 3; ++p) ((__bf16*)w2)[(size_t)p * total + i] = (__bf16)0.f;
    } else if (bf16) ((__bf16*)w2)[i] = (__bf16)v;       // (exact: v already is a bf16 value)
    else w2[i] = v;
  }
}

struct TapGatherArgs {
  const float* T;
  const float* bias;
  const float* aux;
  float* out;
  int B, IH, IW, OH, OW, N;
  int th, tw, dy0, dx0, dys, dxs;
  int pad_mode, act, aux_act;
  float slope, aux_slope;
  long long plane;     // floats per tap plane = B*IH*IW*4
};
__global__ void tap_gather_kernel(const TapGatherArgs a) {
  const long long total = (long long)a.B * a.OH * a.OW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.OW);
    const long long t1 = i / a.OW;
    const int y = (int)(t1 % a.OH);
    const long long b = t1 / a.OH;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int ty = 0; ty < a.th; ++ty) {
      int sy = y + a.dy0 + ty * a.dys;
      if (a.pad_mode == IPRGAN_PAD_REFLECT) sy = reflect_idx(sy, a.IH);
      else if ((unsigned)sy >= (unsigned)a.IH) continue;
      for (int tx = 0; tx < a.tw; ++tx) {
        int sx = x + a.dx0 + tx * a.dxs;
        if (a.pad_mode == IPRGAN_PAD_REFLECT) sx = reflect_idx(sx, a.IW);
        else if ((unsigned)sx >= (unsigned)a.IW) continue;
        s += *(const f32x4*)(a.T + (size_t)(ty * a.tw + tx) * a.plane + ((b * a.IH + sy) * a.IW + sx) * 4);
      }
    }
    f32x4 o;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float v = s[n];
      if (a.bias && n < a.N) v += a.bias[n];
      v = act_apply(v, a.act, a.slope);
      if (a.aux) v *= act_grad_from_out(a.aux[i * 4 + n], a.aux_act, a.aux_slope);
      o[n] = n < a.N ? v : 0.f;
    }
    *(f32x4*)(a.out + i * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------
// backward-weight: ws[split][n][k] = sum_{m in split} P[m][n] * Qg[m][k],  k = tap*Qs + c
// ------------------------------------------------------------------------------------------
struct WGradArgs {
  const float* P;
  const float* Q;
  float* ws;
  int Ps, Pvalid;            // channel stride of P, number of readable channels (= Ps)
  int QH, QW, Qs, c4n;
  FastDiv d_c4n, d_pw, d_plane, d_tw;
  int PH, PW, M;
  int isy, isx, pad, tw, ntap, pad_mode;
  int flip;                  // +1: Q pixel = P pixel * stride + (tap - pad);  -1 (swapped roles): P pixel - (tap - pad)
  int Kw, Nrows;             // slab row length / rows
  int chunks_per_split;
  unsigned p_bytes, q_bytes;
  int dx32, dy32;            // 32 rows of m = db32 images + dy32 rows + dx32 pixels (mixed radix of PH x PW)
  int xcd;                   // wgrad_t_kernel: XCD-contiguous tile order
  int in16;                  // 1: P and Q are bf16 tensors and the bf16 image applies (host-side: picks the IN16 kernel)
                             // 2: P and Q are three-plane tensors (plane strides p_ps / q_ps bytes): the IN3P split kernel
  int p16, q16;              // storage kind of P / Q (kernels without IN16 widen on arrival: 1 = bf16 chunks; 2 = three
                             // planes, summed h + (m + l) exactly - the 64-channel side of an RGB stem / head in fp32x3 mode,
                             // whose other operand is an fp32 image: wgrad_kernel only)
  unsigned p_ps, q_ps;
  unsigned bstep0, bstep1;   // byte step of the image base for db32 / db32+1 images
  double flops;
};

// BF16 (math mode 1; 128x128 tiles only, so both LDS images have 256-byte rows): the chunk rows are rounded to
// bf16 on the way into LDS, stored row-major [m][n] / [m][k] exactly as they arrive, and consumed along m - the
// MFMA's K dimension - through ds_read_b64_tr_b16 (the CDNA4 transposing read: a 16-lane group fetches a 4-row x
// 16-column block and every lane receives one column).  Two such reads give a lane its 8 consecutive m for
// v_mfma_f32_32x32x16_bf16.  16-byte chunk ch of row r sits at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)): conflict-free
// for the 8-byte stores and for the transposed reads (cdna_hip_programming.md T10, image (b)).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned wg_bf16_off(int row, int ch) {
  return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ bf16x8 wg_tr_read8(const char* base, unsigned off0, unsigned off1) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + off1));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = lo; u.h[1] = hi;
  return u.v;
}

// SPLIT (math mode 2): three bf16 images (h, m, l) per operand and six MFMAs per product block, as in gconv_kernel.
template <int WGM, int WGN, int WM, int WN, int NBUF, bool REFLECT, bool BF16 = false, bool IN16 = false, bool SPLIT = false>
__global__ __launch_bounds__(WGM * WGN * 64) void wgrad_kernel(const WGradArgs a) {
  static_assert(!BF16 || (WGM * WM == 4 && WGN * WN == 4), "the bf16 wgrad image is written for 128x128 tiles");
  static_assert(!IN16 || BF16, "bf16 operands in HBM need the bf16 image");
  static_assert(!SPLIT || BF16, "split tiles: the bf16 image");
  constexpr bool IN3P = IN16 && SPLIT;       // P and Q ARE three bf16 planes in HBM (storage kind 2): loaded plane by plane
  constexpr int NLD = IN3P ? 3 : 1;
  constexpr int BN = WGM * WM * 32;   // tile over n (P channels)  -> MFMA rows
  constexpr int BK = WGN * WN * 32;   // tile over k (tap,c)       -> MFMA cols
  constexpr int EPC = IN16 ? 8 : 4;                // elements per loader chunk (IN16: P and Q are bf16 tensors, 16-byte chunks of 8)
  constexpr int CP = BN / EPC, CQ = BK / EPC;      // 16-B chunks per tile row
  constexpr int NT = WGM * WGN * 64;               // threads: one wave per (32*WM)x(32*WN) sub-tile
  constexpr int RPP = NT / CP, RPQ = NT / CQ;      // rows per pass
  constexpr int NP = 32 / RPP, NQ = 32 / RPQ;      // passes per 32-row chunk
  constexpr int STAGE = 32 * (BN + BK);            // floats per stage
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  float* ldsf = (float*)lds;

  // (no xcd_remap here: keeping the blocks of one split on one XCD cuts the L2 miss traffic 4x but measured
  //  2-3 % slower - the blocks of a split then queue on the same L2 channels)
  const int k0 = blockIdx.x * BK, n0 = blockIdx.y * BN, split = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int cp = tid % CP, rp = tid / CP;
  const int cq = tid % CQ, rq = tid / CQ;

  // this thread's fixed k-chunk of the Q tile (IN16: chunks of 8 channels)
  const int q = k0 / EPC + cq;
  const int cpn = IN16 ? a.c4n / 2 : a.c4n;            // chunks per pixel
  const int t = IN16 ? q / cpn : (int)fdiv(q, a.d_c4n);
  const int c4 = q - t * cpn;
  const bool qvalid = t < a.ntap;
  const int ty = fdiv(t, a.d_tw), tx = t - ty * a.tw;
  const int dy = (ty - a.pad) * a.flip, dx = (tx - a.pad) * a.flip;
  const bool pvalid = (n0 + cp * EPC) < a.Pvalid;
  const int plane = a.PH * a.PW;

  const int chunk_begin = split * a.chunks_per_split;
  int chunk_end = chunk_begin + a.chunks_per_split;
  const int total_chunks = (a.M + 31) / 32;
  if (chunk_end > total_chunks) chunk_end = total_chunks;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // split tiles: the five small terms of a product block get their own accumulator (conv_x3.hip: every bf16 MFMA into a
  // large accumulator costs about an ulp of it; summed among themselves they cost 2^-7 of that), merged before the epilogue
  f32x16 accs[SPLIT ? WM : 1][SPLIT ? WN : 1];
  if (SPLIT) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[SPLIT ? i : 0][SPLIT ? j : 0][r] = 0.f;
  }

  f32x4 rP[NLD][NP], rQ[NLD][NQ];

  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.P, 0, a.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)a.Q, 0, a.q_bytes, 0x00020000);
  const int M = a.M, PW = a.PW, PH = a.PH, QH = a.QH, QW = a.QW, isy = a.isy, isx = a.isx;
  // element sizes in HBM: IN16 kernels read bf16 chunks of 8 channels; the other kernels read chunks of 4 channels that
  // are fp32 (16 bytes) or, for a bf16 tensor (a.p16 / a.q16), 8 bytes widened to fp32 on arrival
  const unsigned pes = IN16 ? 2u : (a.p16 ? 2u : 4u), qes = IN16 ? 2u : (a.q16 ? 2u : 4u);
  const int Qs4 = a.Qs * (int)qes, dx32 = a.dx32, dy32 = a.dy32;          // bytes per Q pixel
  const unsigned pstep = 32u * (unsigned)a.Ps * pes, bstep0 = a.bstep0, bstep1 = a.bstep1;
  const unsigned p_plane = (unsigned)a.M * (unsigned)a.Ps * 2u;           // bytes of one plane of a three-plane P

  // Running state of this thread's rows, advanced by 32 rows of m per chunk with adds and selects only.
  // The integer work of the loader competes with the MFMAs for issue slots (the waves of a block are
  // in the same phase between barriers), so there are no divisions and only 24-bit (full-rate) multiplies
  // in the loop: pixel (y, x) and the image base are carried, taps and borders are applied per load.
  // A thread whose channel chunk is outside the tensor carries OOB_OFFSET in its base (rows past M of P
  // fall outside the buffer by themselves: p_bytes = M * Ps * 4).
  unsigned po[NP], qb[NQ];
  int qm[NQ], qy[NQ], qx[NQ];
  int pm[IN3P ? NP : 1];            // three planes: rows past M of plane h are plane m's memory, not the end of the buffer
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    po[i] = pvalid ? (unsigned)((chunk_begin * 32 + rp + RPP * i) * a.Ps) * pes + (unsigned)(n0 + cp * EPC) * pes
                   : OOB_OFFSET;
    if (IN3P) pm[IN3P ? i : 0] = chunk_begin * 32 + rp + RPP * i;
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int m = chunk_begin * 32 + rq + RPQ * i;
    const int b = fdiv(m, a.d_plane);
    const int rem = m - b * plane;
    qy[i] = fdiv(rem, a.d_pw);
    qx[i] = rem - qy[i] * PW;
    qm[i] = m;
    qb[i] = (unsigned)b * (unsigned)(QH * QW) * (unsigned)Qs4 + (qvalid ? (unsigned)c4 * (unsigned)EPC * qes : OOB_OFFSET);
  }

  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (IN3P) {
        const bool okp = po[i] != OOB_OFFSET && pm[IN3P ? i : 0] < M;
#pragma unroll
        for (int p = 0; p < NLD; ++p) rP[p][i] = buf_load4(rs_p, okp ? po[i] + (unsigned)p * a.p_ps : OOB_OFFSET);
        pm[IN3P ? i : 0] += 32;
        if (po[i] != OOB_OFFSET) po[i] += pstep;
        continue;
      }
      if (!IN16 && a.p16 == 2) {       // (rows past M of plane h are plane m's memory: po < bytes of one plane)
        const bool okp = po[i] < p_plane;
        rP[0][i] = buf_load4_bf16(rs_p, okp ? po[i] : OOB_OFFSET) + (buf_load4_bf16(rs_p, okp ? po[i] + a.p_ps : OOB_OFFSET) +
                                                                     buf_load4_bf16(rs_p, okp ? po[i] + 2u * a.p_ps : OOB_OFFSET));
        if (po[i] != OOB_OFFSET) po[i] += pstep;
        continue;
      }
      rP[0][i] = (!IN16 && a.p16) ? buf_load4_bf16(rs_p, po[i]) : buf_load4(rs_p, po[i]);
      po[i] += pstep;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      int iy = __mul24(qy[i], isy) + dy, ix = __mul24(qx[i], isx) + dx;
      bool ok = qm[i] < M;
      if (REFLECT) {
        iy = reflect_idx(iy, QH);
        ix = reflect_idx(ix, QW);
      } else {
        ok = ok && (unsigned)iy < (unsigned)QH && (unsigned)ix < (unsigned)QW;
      }
      const unsigned off = qb[i] + (unsigned)__mul24(__mul24(iy, QW) + ix, Qs4);
      if (IN3P) {
#pragma unroll
        for (int p = 0; p < NLD; ++p) rQ[p][i] = buf_load4(rs_q, ok && qb[i] < OOB_OFFSET ? off + (unsigned)p * a.q_ps : OOB_OFFSET);
      } else if (!IN16 && a.q16 == 2) {
        const bool okq = ok && qb[i] < OOB_OFFSET;
        rQ[0][i] = buf_load4_bf16(rs_q, okq ? off : OOB_OFFSET) + (buf_load4_bf16(rs_q, okq ? off + a.q_ps : OOB_OFFSET) +
                                                                   buf_load4_bf16(rs_q, okq ? off + 2u * a.q_ps : OOB_OFFSET));
      } else
      rQ[0][i] = (!IN16 && a.q16) ? buf_load4_bf16(rs_q, ok ? off : OOB_OFFSET) : buf_load4(rs_q, ok ? off : OOB_OFFSET);
      int x = qx[i] + dx32, y = qy[i] + dy32;
      const bool cx = x >= PW;
      x -= cx ? PW : 0;
      y += cx ? 1 : 0;
      const bool cy = y >= PH;
      y -= cy ? PH : 0;
      qb[i] += cy ? bstep1 : bstep0;
      qx[i] = x; qy[i] = y; qm[i] += 32;
    }
  };
  auto lstore = [&](int buf) {
    if (IN16) {               // the 16 bytes loaded are chunk cp of row r of the [32][128] bf16 image (IN3P: of each plane's image)
      char* Pb = (char*)lds + buf * (IN3P ? 49152 : 16384);
      char* Qb = Pb + (IN3P ? 24576 : 8192);
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int p = 0; p < NLD; ++p) *(f32x4*)(Pb + p * 8192 + wg_bf16_off(rp + RPP * i, cp)) = rP[p][i];
#pragma unroll
      for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int p = 0; p < NLD; ++p) *(f32x4*)(Qb + p * 8192 + wg_bf16_off(rq + RPQ * i, cq)) = rQ[p][i];
      return;
    }
    if (SPLIT) {              // stage = 3 P images, then 3 Q images, of 8 KB each
      char* Pb = (char*)lds + buf * 49152;
      char* Qb = Pb + 24576;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        bf16x4 t3[3];
        split3_bf16(rP[0][i], t3);
#pragma unroll
        for (int p = 0; p < 3; ++p) *(bf16x4*)(Pb + p * 8192 + wg_bf16_off(rp + RPP * i, cp >> 1) + 8 * (cp & 1)) = t3[p];
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        bf16x4 t3[3];
        split3_bf16(rQ[0][i], t3);
#pragma unroll
        for (int p = 0; p < 3; ++p) *(bf16x4*)(Qb + p * 8192 + wg_bf16_off(rq + RPQ * i, cq >> 1) + 8 * (cq & 1)) = t3[p];
      }
      return;
    }
    if (BF16) {               // stage = two [32][128] bf16 images of 8 KB
      char* Pb = (char*)lds + buf * 16384;
      char* Qb = Pb + 8192;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        *(bf16x4*)(Pb + wg_bf16_off(rp + RPP * i, cp >> 1) + 8 * (cp & 1)) = to_bf16x4(rP[0][i]);
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        *(bf16x4*)(Qb + wg_bf16_off(rq + RPQ * i, cq >> 1) + 8 * (cq & 1)) = to_bf16x4(rQ[0][i]);
      return;
    }
    float* Pt = ldsf + buf * STAGE;
    float* Qt = Pt + 32 * BN;
#pragma unroll
    for (int i = 0; i < NP; ++i) *(f32x4*)(Pt + (rp + RPP * i) * BN + cp * 4) = rP[0][i];
#pragma unroll
    for (int i = 0; i < NQ; ++i) *(f32x4*)(Qt + (rq + RPQ * i) * BK + cq * 4) = rQ[0][i];
  };
  auto compute = [&](int buf) {
    const int half = lane >> 5, l31 = lane & 31;
    if (SPLIT) {
      const char* Pb = (const char*)lds + buf * 49152;
      const char* Qb = Pb + 24576;
      const int gq = (lane & 15) >> 2, gp = lane & 3, gcol = (lane >> 4) & 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int r0 = 16 * kk + 8 * half + gq;
        bf16x8 af[3][WM], bf[3][WN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            const int ch = (wm * WM + i) * 4 + 2 * gcol + (gp >> 1);
            af[p][i] = wg_tr_read8(Pb + p * 8192, wg_bf16_off(r0, ch) + 8 * (gp & 1), wg_bf16_off(r0 + 4, ch) + 8 * (gp & 1));
          }
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int ch = (wn * WN + j) * 4 + 2 * gcol + (gp >> 1);
            bf[p][j] = wg_tr_read8(Qb + p * 8192, wg_bf16_off(r0, ch) + 8 * (gp & 1), wg_bf16_off(r0 + 4, ch) + 8 * (gp & 1));
          }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {           // l h', h l', m m', m h', h m', h h'
          const int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
          const int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              if (t < 5) accs[SPLIT ? i : 0][SPLIT ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bf[pb][j], accs[SPLIT ? i : 0][SPLIT ? j : 0], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bf[pb][j], acc[i][j], 0, 0, 0);
            }
        }
      }
      return;
    }
    if (BF16) {
      const char* Pb = (const char*)lds + buf * 16384;
      const char* Qb = Pb + 8192;
      // 16-lane group: lane 4q+p addresses row r0+q, columns 4p..4p+3 of the group's 16 columns
      const int gq = (lane & 15) >> 2, gp = lane & 3, gcol = (lane >> 4) & 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int r0 = 16 * kk + 8 * half + gq;            // rows r0 (first read) and r0 + 4 (second read)
        bf16x8 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const int ch = (wm * WM + i) * 4 + 2 * gcol + (gp >> 1);
          af[i] = wg_tr_read8(Pb, wg_bf16_off(r0, ch) + 8 * (gp & 1), wg_bf16_off(r0 + 4, ch) + 8 * (gp & 1));
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int ch = (wn * WN + j) * 4 + 2 * gcol + (gp >> 1);
          bf[j] = wg_tr_read8(Qb, wg_bf16_off(r0, ch) + 8 * (gp & 1), wg_bf16_off(r0 + 4, ch) + 8 * (gp & 1));
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      return;
    }
    const float* Pt = ldsf + buf * STAGE;
    const float* Qt = Pt + 32 * BN;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int row = 2 * kk + half;
      float af[WM], bf[WN];
      // a wave with two tiles per side reads 8 bytes per lane: channels 2*l31, 2*l31+1 feed tiles 0 and 1
      // (tile t owns rows/cols base + 2*i + t), halving the LDS read count; one tile per side reads 4 bytes
      if (WM == 2) {
        const f32x2 v = *(const f32x2*)(Pt + row * BN + wm * 64 + 2 * l31);
        af[0] = v.x; af[WM - 1] = v.y;
      } else {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = Pt[row * BN + (wm * WM + i) * 32 + l31];
      }
      if (WN == 2) {
        const f32x2 v = *(const f32x2*)(Qt + row * BK + wn * 64 + 2 * l31);
        bf[0] = v.x; bf[WN - 1] = v.y;
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = Qt[row * BK + (wn * WN + j) * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  if (chunk_begin < chunk_end) {
    gload();
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
      const bool more = ch + 1 < chunk_end;
      if (more) gload();
      if (NBUF == 2) {
        compute(buf);
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
      } else {
        compute(0);
        __syncthreads();
        if (more) lstore(0);
        __syncthreads();
      }
    }
  }

  if (SPLIT) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] += accs[SPLIT ? i : 0][SPLIT ? j : 0];
  }
  // slab store: quad transpose (see gconv epilogue) so that every lane writes 16 contiguous bytes
  const int half = lane >> 5, l31 = lane & 31;
  const int qp = lane & 3, q4 = (l31 >> 2);
  float* slab = a.ws + (size_t)split * a.Nrows * a.Kw;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ri = 8 * g + 4 * half + qp;
      const int n = (WM == 2 && !BF16) ? n0 + wm * 64 + 2 * ri + i : n0 + (wm * WM + i) * 32 + ri;
      float t[WN][4];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        t[j][0] = acc[i][j][4 * g]; t[j][1] = acc[i][j][4 * g + 1]; t[j][2] = acc[i][j][4 * g + 2]; t[j][3] = acc[i][j][4 * g + 3];
        quad_transpose(t[j][0], t[j][1], t[j][2], t[j][3], qp);
      }
      if (n >= a.Nrows) continue;
      float* row = slab + (size_t)n * a.Kw;
      if (WN == 2 && !BF16) {  // tile tn owns columns base + 2*col + tn: interleave the two tiles -> 8 contiguous floats
        const int k = k0 + wn * 64 + 8 * q4;
        if (k < a.Kw) {
          const f32x4 v0 = {t[0][0], t[WN - 1][0], t[0][1], t[WN - 1][1]};
          const f32x4 v1 = {t[0][2], t[WN - 1][2], t[0][3], t[WN - 1][3]};
          *(f32x4*)(row + k) = v0;
          *(f32x4*)(row + k + 4) = v1;
        }
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int k = k0 + (wn * WN + j) * 32 + 4 * q4;
          if (k < a.Kw) {
            const f32x4 v = {t[j][0], t[j][1], t[j][2], t[j][3]};
            *(f32x4*)(row + k) = v;
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// backward-weight, second form ("transposed image"): the same GEMM  ws[split][n][k] = sum_m P[m][n] Q[m][k], but
// the LDS tiles have the layout of gconv_kernel - one 128-byte row per OUTPUT row (n for P, k for Q) holding the 32
// reduction indices m of the stage as eight 16-byte chunks of 4 consecutive m - so that the fragment reads are the
// conflict-free ds_read_b128 pairs of gconv_kernel (one pair feeds 4 MFMAs; 16 LDS reads per 64 MFMAs instead of 32)
// and the inner loop IS gconv's.  The data arrive from HBM the other way round (a 16-byte load = 4 channels of ONE
// pixel), so every loader thread owns a 4 pixel x 4 channel block: four 16-byte loads (consecutive pixels), a 4x4
// transpose in registers, four 16-byte stores (one per channel).  Lanes 0-7 of a store group cover 4 channel chunks x
// 2 pixel groups: their 8 slots of the 128-byte bank line are distinct (chunk c of row r sits at c ^ ((r >> 1) & 7)).
// With PW % 4 == 0 (every map of the three GANs except the PatchGAN's 31/30-pixel ones) the four pixels of a block
// lie in one image row: one (y, x) state per thread instead of four, and the tap / border arithmetic is shared.
// ------------------------------------------------------------------------------------------
template <int WGM, int WGN, int WM, int WN, bool REFLECT, bool ROW4>
__global__ __launch_bounds__(WGM * WGN * 64) void wgrad_t_kernel(const WGradArgs a) {
  constexpr int BN = WGM * WM * 32, BK = WGN * WN * 32, NT = WGM * WGN * 64;
  constexpr int CH = 8;
  constexpr int NPB = BN / 4 * 8, NQB = BK / 4 * 8;        // 4x4 blocks per operand and stage
  constexpr bool SPLIT = NPB + NQB <= NT;                  // enough threads: disjoint loaders for P and Q
  static_assert(NPB <= NT && NQB <= NT, "one block per thread and operand");
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  f32x4* A4 = lds;
  f32x4* B4 = lds + BN * CH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  // logical tile order: with a.xcd the blocks of one split (which read the same P / Q rows) run on one XCD
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (a.xcd) lin = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
  const int tiles = gridDim.x * gridDim.y;
  const int split = (int)(lin / tiles), tile = (int)(lin % tiles);
  const int k0 = (tile % (int)gridDim.x) * BK, n0 = (tile / (int)gridDim.x) * BN;

  // loader roles: block index -> (channel chunk c, pixel group g); lane bits [c0 c1 | g0 | c2.. | g1 g2]
  const bool doP = tid < NPB;
  const int qt = SPLIT ? tid - (NT - NQB) : tid;
  const bool doQ = qt >= 0 && qt < NQB;
  auto decode = [](int idx, int C, int& c, int& g) {
    const int chi = (idx >> 3) & (C / 4 - 1);
    c = chi * 4 + (idx & 3);
    g = ((idx >> 3) / (C / 4)) * 2 + ((idx >> 2) & 1);
  };
  int pc = 0, pg = 0, qc = 0, qg = 0;
  decode(doP ? tid : 0, BN / 4, pc, pg);
  decode(doQ ? qt : 0, BK / 4, qc, qg);

  // this thread's fixed k-chunk of the Q tile
  const int q = k0 / 4 + qc;
  const int t = fdiv(q, a.d_c4n);
  const int c4 = q - t * a.c4n;
  const bool qvalid = doQ && t < a.ntap;
  const int ty = fdiv(t, a.d_tw), tx = t - ty * a.tw;
  const int dy = (ty - a.pad) * a.flip, dx = (tx - a.pad) * a.flip;
  const bool pvalid = doP && (n0 + pc * 4) < a.Pvalid;
  const int plane = a.PH * a.PW;

  const int chunk_begin = split * a.chunks_per_split;
  int chunk_end = chunk_begin + a.chunks_per_split;
  const int total_chunks = (a.M + 31) / 32;
  if (chunk_end > total_chunks) chunk_end = total_chunks;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.P, 0, a.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)a.Q, 0, a.q_bytes, 0x00020000);
  const int M = a.M, PW = a.PW, PH = a.PH, QH = a.QH, QW = a.QW, isy = a.isy, isx = a.isx;
  const int Qs4 = a.Qs * 4, dx32 = a.dx32, dy32 = a.dy32;
  const unsigned prow = (unsigned)a.Ps * 4u, pstep = 32u * prow, bstep0 = a.bstep0, bstep1 = a.bstep1;

  f32x4 rP[4], rQ[4];
  // P: rows m = 32 chunk + 4 pg + i; rows past M fall outside the buffer (p_bytes = M * Ps * 4)
  // Q: running pixel state (see wgrad_kernel), one per block (ROW4) or one per pixel
  constexpr int NS = ROW4 ? 1 : 4;
  unsigned po = OOB_OFFSET;
  unsigned qb[NS];
  int qm[NS], qy[NS], qx[NS];
  auto init_state = [&](int chunk) {
    po = pvalid ? (unsigned)((chunk * 32 + 4 * pg) * a.Ps) * 4u + (unsigned)(n0 + pc * 4) * 4u : OOB_OFFSET;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int m = chunk * 32 + 4 * qg + i;
      const int b = fdiv(m, a.d_plane);
      const int rem = m - b * plane;
      qy[i] = fdiv(rem, a.d_pw);
      qx[i] = rem - qy[i] * PW;
      qm[i] = m;
      qb[i] = (unsigned)b * (unsigned)(QH * QW) * (unsigned)Qs4 + (qvalid ? (unsigned)c4 * 16u : OOB_OFFSET);
    }
  };
  // XCD-contiguous order (a.xcd): the tiles of a split run together on one XCD and share its L2 - but they would all
  // ask for the same P / Q rows at the same moment.  Each tile therefore starts its walk over the split's chunks at
  // its own offset and wraps around (the summation order of a tile is rotated, still fixed: deterministic).
  const int nch = chunk_end > chunk_begin ? chunk_end - chunk_begin : 0;
  int cur = chunk_begin + ((a.xcd && nch > 0) ? (tile * 5) % nch : 0);
  init_state(cur);

  auto gload = [&]() {
    if (doP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rP[i] = buf_load4(rs_p, pvalid ? po + (unsigned)i * prow : OOB_OFFSET);
      po += pstep;
    }
    if (doQ) {
      if (ROW4) {
        int iy = __mul24(qy[0], isy) + dy;
        const int ix0 = __mul24(qx[0], isx) + dx;
        bool oky = qm[0] < M;
        if (REFLECT) iy = reflect_idx(iy, QH); else oky = oky && (unsigned)iy < (unsigned)QH;
        const unsigned rowoff = qb[0] + (unsigned)__mul24(__mul24(iy, QW), Qs4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int ix = ix0 + i * isx;
          bool ok = oky;
          if (REFLECT) ix = reflect_idx(ix, QW); else ok = ok && (unsigned)ix < (unsigned)QW;
          rQ[i] = buf_load4(rs_q, ok ? rowoff + (unsigned)__mul24(ix, Qs4) : OOB_OFFSET);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int iy = __mul24(qy[i], isy) + dy, ix = __mul24(qx[i], isx) + dx;
          bool ok = qm[i] < M;
          if (REFLECT) {
            iy = reflect_idx(iy, QH);
            ix = reflect_idx(ix, QW);
          } else {
            ok = ok && (unsigned)iy < (unsigned)QH && (unsigned)ix < (unsigned)QW;
          }
          rQ[i] = buf_load4(rs_q, ok ? qb[i] + (unsigned)__mul24(__mul24(iy, QW) + ix, Qs4) : OOB_OFFSET);
        }
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        int x = qx[i] + dx32, y = qy[i] + dy32;
        const bool cx = x >= PW;
        x -= cx ? PW : 0;
        y += cx ? 1 : 0;
        const bool cy = y >= PH;
        y -= cy ? PH : 0;
        qb[i] += cy ? bstep1 : bstep0;
        qx[i] = x; qy[i] = y; qm[i] += 32;
      }
    }
  };
  auto lstore = [&]() {
    if (doP) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * pc + j;
        const f32x4 v = {rP[0][j], rP[1][j], rP[2][j], rP[3][j]};
        A4[r * CH + (pg ^ ((r >> 1) & 7))] = v;
      }
    }
    if (doQ) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * qc + j;
        const f32x4 v = {rQ[0][j], rQ[1][j], rQ[2][j], rQ[3][j]};
        B4[r * CH + (qg ^ ((r >> 1) & 7))] = v;
      }
    }
  };
  auto compute = [&]() {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      f32x4 af[WM], bf[WN];
      const int c = 2 * kq + half;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int r = (wm * WM + i) * 32 + l31;
        af[i] = A4[r * CH + (c ^ ((r >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int r = (wn * WN + j) * 32 + l31;
        bf[j] = B4[r * CH + (c ^ ((r >> 1) & 7))];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  if (nch > 0) {
    gload();
    lstore();
    __syncthreads();
    for (int it = 0; it < nch; ++it) {
      const bool more = it + 1 < nch;
      if (more) {
        if (++cur == chunk_end) { cur = chunk_begin; init_state(cur); }      // wrap (rotated start only)
        gload();
      }
      compute();
      __syncthreads();
      if (more) lstore();
      __syncthreads();
    }
  }

  // slab store: quad transpose (see gconv epilogue) so that every lane writes 16 contiguous bytes
  const int half = lane >> 5, l31 = lane & 31;
  const int qp = lane & 3, q4 = (l31 >> 2);
  float* slab = a.ws + (size_t)split * a.Nrows * a.Kw;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + (wm * WM + i) * 32 + 8 * g + 4 * half + qp;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        float t0 = acc[i][j][4 * g], t1 = acc[i][j][4 * g + 1], t2 = acc[i][j][4 * g + 2], t3 = acc[i][j][4 * g + 3];
        quad_transpose(t0, t1, t2, t3, qp);
        const int k = k0 + (wn * WN + j) * 32 + 4 * q4;
        if (n < a.Nrows && k < a.Kw) {
          const f32x4 v = {t0, t1, t2, t3};
          *(f32x4*)(slab + (size_t)n * a.Kw + k) = v;
        }
      }
    }
}

// out[m][n] = act(sum_z slab[z][m][n] + bias[n]) * act'(aux[m][n]) + res[m][n] in fixed split order (deterministic);
// pad channels stay zero.  The same epilogue as gconv_kernel's, applied once to the reduced tile.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const f32x4* __restrict__ slab, const float* __restrict__ bias,
                                                            f32x4* __restrict__ out, unsigned n4, unsigned Ns4, int N,
                                                            int ksplit, int act, float slope, const f32x4* __restrict__ aux,
                                                            int aux_act, float aux_slope, const f32x4* __restrict__ res,
                                                            int okind = 0, size_t ps = 0) {
  // okind 2: out (and res) are three-plane tensors with plane stride ps elements, aux is the h plane of one (bf16)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    f32x4 v = slab[i];
    for (int z = 1; z < ksplit; ++z) v += slab[(size_t)z * n4 + i];
    const int n = (int)(i % Ns4) * 4;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (aux) o = okind == 2 ? ld_bf16x4((const float*)aux, (size_t)i * 4) : aux[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = v[k];
      if (bias && n + k < N) t += bias[n + k];
      t = act_apply(t, act, slope);
      if (aux) t *= act_grad_from_out(o[k], aux_act, aux_slope);
      v[k] = n + k < N ? t : 0.f;
    }
    if (okind == 2) {
      const size_t e = (size_t)i * 4;
      if (res) v += ld_bf16x4((const float*)res, e) + (ld_bf16x4((const float*)res, e + ps) + ld_bf16x4((const float*)res, e + 2 * ps));
      bf16x4 t3[3];
      split3_bf16(v, t3);
#pragma unroll
      for (int p = 0; p < 3; ++p) *(bf16x4*)((__bf16*)out + e + p * ps) = t3[p];
      continue;
    }
    if (res) v += res[i];
    out[i] = v;
  }
}

// x[B][H][W][C4] -> xp[B][H+2p][W+2p][C4] with ReflectionPad2d borders (only for the swapped-role wgrad of the
// reflect-padded RGB heads, see WGeom)
__global__ void reflect_pad_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ xp, int B, int H, int W,
                                   int C4n, int pad, size_t x_ps = 0) {        // x_ps != 0: x is a three-plane tensor
  const int HP = H + 2 * pad, WP = W + 2 * pad;
  const size_t total = (size_t)B * HP * WP * C4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4n);
    size_t r = i / C4n;
    const int xq = (int)(r % WP); r /= WP;
    const int yq = (int)(r % HP);
    const int b = (int)(r / HP);
    const int sy = reflect_idx(yq - pad, H), sx = reflect_idx(xq - pad, W);
    const size_t e = (((size_t)b * H + sy) * W + sx) * C4n + c;
    if (x_ps) {
      const size_t e4 = e * 4;
      xp[i] = ld_bf16x4((const float*)x, e4) + (ld_bf16x4((const float*)x, e4 + x_ps) + ld_bf16x4((const float*)x, e4 + 2 * x_ps));
    } else {
      xp[i] = x[e];
    }
  }
}

// sum the split slabs in fixed order and scatter into PyTorch weight layout.
// block = 16 element quads (64 consecutive slab floats, 16-byte loads) x 16 split lanes: lane l sums splits
// l, l+16, ... with four loads in flight, then the 16 lanes are combined in lane order (deterministic).  The first
// version (4 lanes, scalar loads, one load in flight) ran at 0.5 TB/s on the slabs: latency-bound.
__device__ __forceinline__ void wgrad_reduce_block(unsigned block, const float* __restrict__ ws, float* __restrict__ dw,
                                                   int nsplit, int Nrows, int Kw, int N, int C, int Qs, int ntap, FastDiv d_qs,
                                                   FastDiv d_row, long long sn, long long sc, float beta, f32x4 (*sh)[16]) {
  const int qd = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int rowlen = ntap * Qs;                     // multiple of 4 (Qs is), rows are 16-byte aligned (Kw % 64 == 0)
  const long long total = (long long)N * rowlen;
  const long long i = ((long long)block * 16 + qd) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int n = 0, kk = 0;
  if (i < total) {
    n = fdiv((uint32_t)i, d_row);
    kk = (int)(i - (long long)n * rowlen);
    const float* p = ws + (size_t)n * Kw + kk;
    const size_t slab = (size_t)Nrows * Kw;
    f32x4 a0 = s, a1 = s, a2 = s, a3 = s;
    int sp = lane;
    for (; sp + 48 < nsplit; sp += 64) {
      a0 += *(const f32x4*)(p + (size_t)sp * slab);
      a1 += *(const f32x4*)(p + (size_t)(sp + 16) * slab);
      a2 += *(const f32x4*)(p + (size_t)(sp + 32) * slab);
      a3 += *(const f32x4*)(p + (size_t)(sp + 48) * slab);
    }
    for (; sp < nsplit; sp += 16) a0 += *(const f32x4*)(p + (size_t)sp * slab);
    s = (a0 + a1) + (a2 + a3);
  }
  sh[lane][qd] = s;
  __syncthreads();
  if (lane == 0 && i < total) {
    f32x4 t = sh[0][qd];
#pragma unroll
    for (int l = 1; l < 16; ++l) t += sh[l][qd];
    const int tap = fdiv(kk, d_qs);
    const int c = kk - tap * Qs;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c + j < C) {
        float* o = dw + n * sn + (c + j) * sc + tap;
        *o = beta != 0.f ? beta * *o + t[j] : t[j];      // beta = 1: accumulate into a gradient bucket view
      }
  }
}
// Row form (round 6): the block above finishes with 16 of its 256 threads scattering single floats, tap-strided, into dw (and
// reading them back first when it accumulates) - the slab reduces of a DCGAN step moved their 0.5 GB at 2.4 TB/s.  Here a block
// owns ONE weight row n and `cw` consecutive channels for ALL taps: its slab reads are runs of cw floats per tap (the whole
// row [tap][c] when cw == Qs), its output is cw * ntap CONSECUTIVE floats of dw ([c][tap] order: PyTorch's), transposed through
// LDS and written (or accumulated) as 16-byte vectors by every thread.  P = ntap * cw / 4 float4 positions x SL = 256 / P
// split lanes; a lane sums its splits sl, sl + SL, ... into eight accumulators in a fixed pattern, the lanes are combined in
// lane order: deterministic.  For Conv2d / ConvTranspose2d weight layouts with unpadded channels (wgrad_reduce_cw).
__device__ __forceinline__ void wgrad_reduce_rows(unsigned block, const float* __restrict__ ws, float* __restrict__ dw, int nsplit,
                                                  int Nrows, int Kw, int Qs, int ntap, int cw, long long sn, float beta,
                                                  f32x4 (*sh)[16], float* tile) {
  const int cq = cw >> 2, P = ntap * cq, SL = 256 / P, groups = Qs / cw;
  const int n = (int)(block / groups), c0 = (int)(block % groups) * cw;
  const int tid = threadIdx.x, p = tid % P, sl = tid / P;
  f32x4* part = &sh[0][0];                   // [SL][P] float4
  if (sl < SL) {
    const int tap = p / cq, q = p - tap * cq;
    const float* src = ws + (size_t)n * Kw + (size_t)tap * Qs + c0 + 4 * q;
    const size_t slab = (size_t)Nrows * Kw;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = z, a1 = z, a2 = z, a3 = z, a4 = z, a5 = z, a6 = z, a7 = z;
    int sp = sl;
    for (; sp + 7 * SL < nsplit; sp += 8 * SL) {
      a0 += *(const f32x4*)(src + (size_t)sp * slab);
      a1 += *(const f32x4*)(src + (size_t)(sp + SL) * slab);
      a2 += *(const f32x4*)(src + (size_t)(sp + 2 * SL) * slab);
      a3 += *(const f32x4*)(src + (size_t)(sp + 3 * SL) * slab);
      a4 += *(const f32x4*)(src + (size_t)(sp + 4 * SL) * slab);
      a5 += *(const f32x4*)(src + (size_t)(sp + 5 * SL) * slab);
      a6 += *(const f32x4*)(src + (size_t)(sp + 6 * SL) * slab);
      a7 += *(const f32x4*)(src + (size_t)(sp + 7 * SL) * slab);
    }
    for (; sp < nsplit; sp += SL) a0 += *(const f32x4*)(src + (size_t)sp * slab);
    part[sl * P + p] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  __syncthreads();
  if (tid < P) {
    f32x4 t = part[tid];
    for (int l = 1; l < SL; ++l) t += part[l * P + tid];
    const int tap = tid / cq, q = tid - tap * cq;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[tap * (cw + 1) + 4 * q + j] = t[j];
  }
  __syncthreads();
  if (tid < P) {                              // cw * ntap / 4 = P output vectors, consecutive in dw
    float* dst = dw + (size_t)n * sn + (size_t)c0 * ntap + 4 * tid;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = 4 * tid + j, c = o / ntap, tap = o - c * ntap;
      v[j] = tile[tap * (cw + 1) + c];
    }
    if (beta != 0.f) v += beta * *(const f32x4*)dst;       // beta = 1: accumulate into a gradient bucket view
    *(f32x4*)dst = v;
  }
}
// channels per block of the row form, 0 = the layout needs the generic block (padded channel counts, the role-swapped
// layers' strides, more than 16 taps)
static int wgrad_reduce_cw(const float* dw, int N, int C, int Qs, int ntap, long long sn, long long sc) {
  if (((uintptr_t)dw & 15) != 0) return 0;         // (a gradient-bucket view behind a parameter with an odd element count)
  if (C != Qs || sc != ntap || sn != (long long)C * ntap || ntap > 16 || (Qs % 16) != 0 || ((C * ntap) % 4) != 0) return 0;
  static const int off = getenv("IPRGAN_WGRAD_REDUCE_ROWS") ? atoi(getenv("IPRGAN_WGRAD_REDUCE_ROWS")) == 0 : 0;     // A/B switch
  if (off) return 0;
  const int cands[3] = {64, 32, 16};
  for (int i = 0; i < 3; ++i) {
    const int cw = cands[i];
    if ((Qs % cw) != 0 || ntap * cw / 4 > 256) continue;
    if ((long long)N * (Qs / cw) >= 512 || cw == 16) return cw;
  }
  return 0;
}
static unsigned wgrad_reduce_blocks(int N, int Qs, int ntap, int cw) {
  return cw ? (unsigned)(N * (Qs / cw)) : (unsigned)(((long long)N * ntap * Qs + 63) / 64);
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws,
                                                           float* __restrict__ dw, int nsplit, int Nrows,
                                                           int Kw, int N, int C, int Qs, int ntap,
                                                           FastDiv d_qs, FastDiv d_row, long long sn,
                                                           long long sc, float beta, int cw) {
  __shared__ f32x4 sh[16][16];
  __shared__ float tile[16 * 65];
  if (cw) wgrad_reduce_rows(blockIdx.x, ws, dw, nsplit, Nrows, Kw, Qs, ntap, cw, sn, beta, sh, tile);
  else wgrad_reduce_block(blockIdx.x, ws, dw, nsplit, Nrows, Kw, N, C, Qs, ntap, d_qs, d_row, sn, sc, beta, sh);
}
// The slab reduces of SEVERAL layers in one launch (iprgan_wgrad_reduce_multi): a backward pass owes one per layer - 19 launches
// of 5-17 us per DCGAN step, each behind a kernel boundary - and nothing reads a weight gradient before the pass flushes its
// deferred writes.  Every block does exactly what its wgrad_reduce_kernel block would have done (same lanes, same order of
// additions): the results are bit-identical to the one-launch-per-layer form.
#define WGRAD_MULTI_MAX 24
struct WGradReduceTable {
  const float* ws[WGRAD_MULTI_MAX];
  float* dw[WGRAD_MULTI_MAX];
  long long sn[WGRAD_MULTI_MAX], sc[WGRAD_MULTI_MAX];
  FastDiv d_qs[WGRAD_MULTI_MAX], d_row[WGRAD_MULTI_MAX];
  int nsplit[WGRAD_MULTI_MAX], Nrows[WGRAD_MULTI_MAX], Kw[WGRAD_MULTI_MAX], N[WGRAD_MULTI_MAX], C[WGRAD_MULTI_MAX], Qs[WGRAD_MULTI_MAX],
      ntap[WGRAD_MULTI_MAX];
  float beta[WGRAD_MULTI_MAX];
  int cw[WGRAD_MULTI_MAX];                    // > 0: row form with this many channels per block (wgrad_reduce_cw)
  unsigned first[WGRAD_MULTI_MAX + 1];        // first block of entry e; first[n] = grid size
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WGradReduceTable t) {
  __shared__ f32x4 sh[16][16];
  __shared__ float tile[16 * 65];
  int e = 0;
  while (e + 1 < t.n && blockIdx.x >= t.first[e + 1]) ++e;       // (block-uniform: scalar loads of the kernel arguments)
  if (t.cw[e])
    wgrad_reduce_rows(blockIdx.x - t.first[e], t.ws[e], t.dw[e], t.nsplit[e], t.Nrows[e], t.Kw[e], t.Qs[e], t.ntap[e], t.cw[e],
                      t.sn[e], t.beta[e], sh, tile);
  else
    wgrad_reduce_block(blockIdx.x - t.first[e], t.ws[e], t.dw[e], t.nsplit[e], t.Nrows[e], t.Kw[e], t.N[e], t.C[e], t.Qs[e], t.ntap[e],
                       t.d_qs[e], t.d_row[e], t.sn[e], t.sc[e], t.beta[e], sh);
}

// prepared-weight builder: w[D0][D1][ntap] (PyTorch) -> ot[R0][K0] (row d0, k = tap*C4(D1)+d1)
//                                                      and it[R1][K1] (row d1, k = tap*C4(D0)+d0)
__global__ void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale,
                                   float* __restrict__ dst, int rows_alloc, int Kp, int Drow,
                                   int Dcol, int Cs, int ntap, int row_is_d0, int out16) {
  const float sc = inv_scale ? *inv_scale : 1.f;
  const long long total = (long long)rows_alloc * Kp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const int r = (int)(i / Kp);
    const int tap = k / Cs, c = k - tap * Cs;
    float v = 0.f;
    if (r < Drow && tap < ntap && c < Dcol) {
      const long long src = row_is_d0 ? ((long long)r * Dcol + c) * ntap + tap
                                      : ((long long)c * Drow + r) * ntap + tap;
      v = w[src];
      if (inv_scale) v = v / sc;
    }
    if (out16 == 2) {                              // operand of a three-plane activation: split once here (x = h + m + l)
      const __bf16 h = (__bf16)v;
      const float r1 = v - (float)h;
      const __bf16 m = (__bf16)r1;
      ((__bf16*)dst)[i] = h; ((__bf16*)dst)[total + i] = m; ((__bf16*)dst)[2 * total + i] = (__bf16)(r1 - (float)m);
    } else if (out16) ((__bf16*)dst)[i] = (__bf16)v;      // operand of a bf16 activation: rounded once here (nearest-even)
    else dst[i] = v;
  }
}


// all layers of a network in one launch: blockIdx.y = table entry
#define PREP_MAX 32
struct PrepEntry {
  const float* w;
  const float* inv_scale;
  float* dst;
  int rows_alloc, Kp, Drow, Dcol, Cs, ntap, row_is_d0, out16;
  FastDiv d_ntap;
};
struct PrepTable {
  PrepEntry e[PREP_MAX];
};
// One block per (row r, 64-channel slice): the slice of the PyTorch weight - 64 x ntap floats, contiguous for
// row_is_d0 layouts, ntap-float runs otherwise - is read tap-fastest (coalesced) into LDS and written back
// channel-fastest, i.e. transposed to the tap-major operand.  The first version read the source with a stride of
// ntap floats per lane (and 64-bit divisions per element): 217 us for the 20 M-parameter VGG / D96 tables.
#define PREP_CCH 64
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const PrepTable t) {
  extern __shared__ float psh[];                     // [PREP_CCH][ntap + 1]
  const PrepEntry e = t.e[blockIdx.z];
  const int r = blockIdx.x;
  if (r >= e.rows_alloc) return;
  const float sc = e.inv_scale ? *e.inv_scale : 1.f;
  const int c0 = blockIdx.y * PREP_CCH;
  float* out = e.dst + (size_t)r * e.Kp;
  __bf16* out16 = (__bf16*)e.dst + (size_t)r * e.Kp;
  const int pitch = e.ntap + 1;
  if (c0 < e.Cs) {
    const bool live = r < e.Drow;
    const int n_el = PREP_CCH * e.ntap;
    for (int i = threadIdx.x; i < n_el; i += blockDim.x) {
      const int cl = (int)fdiv((uint32_t)i, e.d_ntap), tap = i - cl * e.ntap;
      const int c = c0 + cl;
      float v = 0.f;
      if (live && c < e.Dcol) {
        const size_t src = e.row_is_d0 ? ((size_t)r * e.Dcol + c) * e.ntap + tap
                                       : ((size_t)c * e.Drow + r) * e.ntap + tap;
        v = e.w[src];
        if (e.inv_scale) v = v / sc;
      }
      psh[cl * pitch + tap] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_el; i += blockDim.x) {
      const int tap = i / PREP_CCH, cl = i % PREP_CCH;
      if (c0 + cl < e.Cs) {
        const float v = psh[cl * pitch + tap];
        if (e.out16 == 2) {
          const size_t ps = (size_t)e.rows_alloc * e.Kp;
          const __bf16 h = (__bf16)v;
          const float r1 = v - (float)h;
          const __bf16 m = (__bf16)r1;
          __bf16* o = out16 + tap * e.Cs + c0 + cl;
          o[0] = h; o[ps] = m; o[2 * ps] = (__bf16)(r1 - (float)m);
        } else if (e.out16) out16[tap * e.Cs + c0 + cl] = (__bf16)v;
        else out[tap * e.Cs + c0 + cl] = v;
      }
    }
  }
  // K padding behind the last tap (Kp is a multiple of 32): the first channel slice clears it
  if (blockIdx.y == 0)
    for (int k = e.ntap * e.Cs + threadIdx.x; k < e.Kp; k += blockDim.x) {
      if (e.out16 == 2) {
        const size_t ps = (size_t)e.rows_alloc * e.Kp;
        out16[k] = (__bf16)0.f; out16[ps + k] = (__bf16)0.f; out16[2 * ps + k] = (__bf16)0.f;
      } else if (e.out16) out16[k] = (__bf16)0.f;
      else out[k] = 0.f;
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                               float* __restrict__ dz, size_t n, int act, float slope, int b16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    if (b16) {
      const __bf16* d = (const __bf16*)dy;
      const __bf16* o = (const __bf16*)out;
      ((__bf16*)dz)[i] = (__bf16)((float)d[i] * act_grad_from_out((float)o[i], act, slope));
    } else {
      dz[i] = dy[i] * act_grad_from_out(out[i], act, slope);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline int c4(int c) { return (c + 3) & ~3; }

struct Shape {
  int OH, OW;
};
static Shape out_shape(const iprgan_conv_desc* d) {
  Shape s;
  if (d->transposed) {
    s.OH = (d->H - 1) * d->stride - 2 * d->pad + d->KH + d->outpad;
    s.OW = (d->W - 1) * d->stride - 2 * d->pad + d->KW + d->outpad;
  } else {
    s.OH = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    s.OW = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  }
  return s;
}

static void finish_phase(Phase& p, int B, int Cs) {
  p.ntap = p.th * p.tw;
  p.M = B * p.ohg * p.owg;
  p.steps = cdiv(p.ntap * Cs, 32);
  p.d_owg = make_fastdiv(p.owg > 0 ? p.owg : 1);
  p.d_plane = make_fastdiv(p.ohg * p.owg > 0 ? p.ohg * p.owg : 1);
  p.d_tw = make_fastdiv(p.tw > 0 ? p.tw : 1);
}

// "forward form": out grid = full output, gather with stride; used by Conv2d fwd and ConvT bwd-data
static void geom_forward_form(GConvArgs& a, int B, int IH, int IW, int Cred, int OH, int OW, int N,
                              int KH, int KW, int stride, int pad) {
  a.B = B; a.IH = IH; a.IW = IW; a.Cs = c4(Cred); a.c4n = a.Cs / 4; a.d_c4n = make_fastdiv(a.c4n);
  a.Kp = rup(KH * KW * a.Cs, 32);
  a.OH = OH; a.OW = OW; a.Ns = c4(N); a.N = N;
  a.isy = a.isx = stride; a.osy = a.osx = 1;
  a.nphase = 1;
  Phase& p = a.ph[0];
  p.th = KH; p.tw = KW; p.dy0 = -pad; p.dx0 = -pad; p.dys = p.dxs = 1;
  p.wbase = 0; p.wsy = KW; p.wsx = 1; p.ooy = p.oox = 0; p.ohg = OH; p.owg = OW;
  finish_phase(p, B, a.Cs);
}

// "backward-data form": out = the larger (input-side) image, gather from the smaller one, one
// phase per output residue mod stride; used by Conv2d bwd-data and ConvT fwd.
// (OHs, OWs) = dims of the gathered (small) image; (H, W) = dims of the produced image.
static void geom_bwd_form(GConvArgs& a, int B, int OHs, int OWs, int Cred, int H, int W, int N,
                          int KH, int KW, int stride, int pad) {
  a.B = B; a.IH = OHs; a.IW = OWs; a.Cs = c4(Cred); a.c4n = a.Cs / 4; a.d_c4n = make_fastdiv(a.c4n);
  a.Kp = rup(KH * KW * a.Cs, 32);
  a.OH = H; a.OW = W; a.Ns = c4(N); a.N = N;
  a.isy = a.isx = 1; a.osy = a.osx = stride;
  a.nphase = stride * stride;
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      Phase& p = a.ph[py * stride + px];
      const int ky0 = (py + pad) % stride, kx0 = (px + pad) % stride;
      p.th = ky0 < KH ? (KH - ky0 + stride - 1) / stride : 0;
      p.tw = kx0 < KW ? (KW - kx0 + stride - 1) / stride : 0;
      p.dy0 = (py + pad - ky0) / stride; p.dx0 = (px + pad - kx0) / stride;
      p.dys = p.dxs = -1;
      p.wbase = ky0 * KW + kx0; p.wsy = stride * KW; p.wsx = stride;
      p.ooy = py; p.oox = px;
      p.ohg = H > py ? (H - py + stride - 1) / stride : 0;
      p.owg = W > px ? (W - px + stride - 1) / stride : 0;
      if (p.th == 0 || p.tw == 0) { p.th = p.tw = 0; }
      finish_phase(p, B, a.Cs);
    }
}

struct TuneKey {
  int v[16];
  bool operator<(const TuneKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
static std::map<TuneKey, int> g_tune;
static std::mutex g_tune_mutex;                 // the tune table and its cache file: several host threads may launch
// IPRGAN_TUNE_CACHE=<file>: tuning decisions are appended to the file and read back by later processes, which
// then launch no tuning trials (start-up time; and a profiler run sees only the launches of the step itself).
// Under a multi-process launcher (RANK set, WORLD_SIZE > 1) every rank uses its own file <file>.<RANK>: ranks tune
// independently and must not interleave their appends.
static bool g_tune_loaded = false;
static std::string tune_path() {
  const char* path = getenv("IPRGAN_TUNE_CACHE");
  if (!path) return std::string();
  std::string p(path);
  const char* ws = getenv("WORLD_SIZE");
  const char* rk = getenv("RANK");
  if (ws && rk && atoi(ws) > 1) p += std::string(".") + rk;
  return p;
}
static void tune_load() {              // caller holds g_tune_mutex
  if (g_tune_loaded) return;
  g_tune_loaded = true;
  const std::string path = tune_path();
  FILE* f = path.empty() ? nullptr : fopen(path.c_str(), "r");
  if (!f) return;
  TuneKey k;
  int v;
  for (;;) {
    int n = 0;
    for (int i = 0; i < 16; ++i) n += fscanf(f, "%d", &k.v[i]);
    n += fscanf(f, "%d", &v);
    if (n != 17) break;
    g_tune[k] = v;
  }
  fclose(f);
}
static bool tune_lookup(const TuneKey& k, int* v) {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  tune_load();
  auto it = g_tune.find(k);
  if (it == g_tune.end()) return false;
  *v = it->second;
  return true;
}
static void tune_store(const TuneKey& k, int v) {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  g_tune[k] = v;
  const std::string path = tune_path();
  FILE* f = path.empty() ? nullptr : fopen(path.c_str(), "a");
  if (!f) return;
  for (int i = 0; i < 16; ++i) fprintf(f, "%d ", k.v[i]);
  fprintf(f, "%d\n", v);
  fclose(f);
}
static int g_force_tile = -1, g_force_wgrad = -1;     // test hook: iprgan_debug_force_tiles
static int g_autotune = getenv("IPRGAN_AUTOTUNE") ? atoi(getenv("IPRGAN_AUTOTUNE")) : 1;
static int g_smalln = getenv("IPRGAN_SMALLN") ? atoi(getenv("IPRGAN_SMALLN")) : 1;
static int g_fewin = getenv("IPRGAN_FEWIN") ? atoi(getenv("IPRGAN_FEWIN")) : 1;      // A/B switch: direct few-input-channel kernel
static int g_math = IPRGAN_MATH_FP32;                 // iprgan_set_math_mode
static int g_bf16_bk = getenv("IPRGAN_BF16_BK") ? atoi(getenv("IPRGAN_BF16_BK")) : 64;   // K step of the bf16 gconv tiles
static int g_nbuf = getenv("IPRGAN_LDS_BUFS") ? atoi(getenv("IPRGAN_LDS_BUFS")) : 1;  // wgrad_kernel only: 1 = single LDS buffer (measured faster: 3-4 blocks/CU)

static thread_local int t_last_bm = 0;      // M tile of the last gconv launch of this thread (partial-row count of STATS launches)

template <int WGM, int WGN, int WM, int WN, bool FAST, int NBUF, int BK, bool BF16 = false, bool STATS = false, bool IN16 = false, bool SPLIT = false>
static int launch_gconv_tfnk(const GConvArgs& a, hipStream_t st) {
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  if (maxM == 0) return 0;
  const size_t smem = NBUF * (size_t)(BM + BN) * (BF16 ? BK / 8 : BK / 4) * sizeof(f32x4) * (SPLIT ? 3 : 1);
  auto kern = gconv_kernel<WGM, WGN, WM, WN, FAST, NBUF, BK, BF16, STATS, IN16, SPLIT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid(cdiv(maxM, BM), cdiv(a.Ns, BN), a.ksplit > 1 ? a.ksplit : a.nphase);
  t_last_bm = BM;
  prof_launch(kern, grid, dim3(WGM * WGN * 64), smem, st,
              SPLIT ? 27 : BF16 ? 12 : WGM * WGN == 8 ? (BN == 128 ? 9 : 10) : BM == 128 ? (BN == 128 ? 0 : (BN == 64 ? 1 : 3)) : (BN == 128 ? 8 : 2),
              a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}

template <int WGM, int WGN, int WM, int WN>
static int launch_gconv_t(const GConvArgs& a, hipStream_t st) {
  // K step 32 everywhere (a 64-deep step was measured neutral-to-slower, -15 % on the N = 64 layers) and ONE LDS buffer
  // (two barriers per step but 3-4 blocks per CU: measured faster than double buffering, which is no longer built)
  const bool fast = (a.Cs % 32) == 0;
  // bf16 math: layers whose K step lies in one tap (C4 % 32 == 0); RGB stems / heads keep the fp32 kernel
  if (a.in16 == 2) { // three bf16 planes per operand in HBM: plane-by-plane loader, the split tiles' LDS image and MFMA terms
    if constexpr (WGM * WM * 32 >= WGM * WGN * 16 && WGN * WN * 32 >= WGM * WGN * 16)      // (a loader pass covers threads / 4 rows)
      return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, true, true, true>(a, st)
                         : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, false, true, true>(a, st);
    return -1;
  }
  if (a.in16)        // bf16 operands in HBM: 64-deep K steps (8 loader chunks per row)
    return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, true, true>(a, st)
                       : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, false, true>(a, st);
  if (g_math == IPRGAN_MATH_FP32X3 && fast)       // fp32 operands as three bf16 terms each, six bf16 MFMAs per product block
    return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, true, false, true>(a, st)
                       : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, false, false, true>(a, st);
  if (g_math == IPRGAN_MATH_BF16 && fast) {
    // the bf16 MFMA retires a 32-deep K step in a quarter of the fp32 time, so the two barriers per step dominate:
    // deeper steps (more MFMAs per barrier pair) when the channel count allows a step to stay inside one tap
    // (measured on DCGAN-128: 32 -> 64 deep +2 %, 128 deep -10 %: the staging path, not the barrier count, is the limit)
    const int bk = (g_bf16_bk >= 64 && a.Cs % 64 == 0) ? 64 : 32;
    if (bk == 64)
      return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, true>(a, st)
                         : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, false>(a, st);
    return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, true>(a, st)
                       : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, false>(a, st);
  }
  if (a.stat_part)
    return fast ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, false, true>(a, st)
                : launch_gconv_tfnk<WGM, WGN, WM, WN, false, 1, 32, false, true>(a, st);
  return fast ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, false, false>(a, st)
              : launch_gconv_tfnk<WGM, WGN, WM, WN, false, 1, 32, false, false>(a, st);
}

// bf16 tiles with 128x64 / 128x128 per wave.  A 64x64 wave tile reads 4 operand fragments from LDS per 4 MFMAs of 32
// cycles: with one wave per SIMD that is 4 x 32 LDS cycles per 128 MFMA cycles - the LDS port is as busy as the matrix
// core, and the staging stores come on top.  128x64 per wave reads 6 fragments per 8 MFMAs (75 %), 128x128 reads 8 per
// 16 (50 %).  (The fp32 MFMA is 4x slower per fragment, so the fp32 tiles are nowhere near this limit.)
template <int WGM, int WGN, int WM, int WN>
static int launch_gconv_bf16big(const GConvArgs& a, hipStream_t st) {
  if constexpr (WM * WN <= 8) {
    if (a.in16 == 2 && a.Ns >= WGN * WN * 32)
      return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, true, true, true>(a, st)
                         : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, false, true, true>(a, st);
    if (a.in16 == 2) return -1;
    if (g_math == IPRGAN_MATH_FP32X3 && (a.Cs % 32) == 0 && a.Ns >= WGN * WN * 32)
      return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, true, false, true>(a, st)
                         : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 32, true, false, false, true>(a, st);
  }
  if (a.in16 == 2 || g_math != IPRGAN_MATH_BF16 || (a.Cs % 64) != 0 || a.Ns < WGN * WN * 32) return -1;
  if (a.in16)
    return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, true, true>(a, st)
                       : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, false, true>(a, st);
  return a.stat_part ? launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, true>(a, st)
                     : launch_gconv_tfnk<WGM, WGN, WM, WN, true, 1, 64, true, false>(a, st);
}

static int launch_gconv(const GConvArgs& ain, hipStream_t st);
int launch_gconv_pipe(const GConvArgs& a, int variant, hipStream_t st, int* bm_out);     // conv_pipe.hip
int launch_gconv_x3p(const GConvArgs& a, int variant, hipStream_t st, int* bm_out);      // conv_x3.hip
int launch_gconv_x3h(const GConvArgs& a, int variant, hipStream_t st, int* bm_out);      // conv_x3.hip: halo form
int launch_gconv_x3p16(const GConvArgs& a, int variant, hipStream_t st, int* bm_out);    // conv_x3.hip: 16x16x32 MFMA form
int launch_gconv_x3ws(const GConvArgs& a, int variant, hipStream_t st, int* bm_out);     // conv_x3.hip: loader waves + multiplying waves

static bool smalln_eligible(const GConvArgs& a) {
  return g_smalln && !a.rs0 && !a.stat_part && a.Ns == 4 && a.nphase == 1 && a.isy == 1 && a.isx == 1 && a.osy == 1 && a.osx == 1 &&
         (a.Cs % 32) == 0 && a.ph[0].ntap >= 2 && !a.planar_M;
}
static size_t smalln_ws_floats(const GConvArgs& a) {
  return (size_t)2 * rup(a.ph[0].ntap * 4, 128) * a.Cs + (size_t)a.ph[0].ntap * a.B * a.IH * a.IW * 4;     // (operand: up to three bf16 planes)
}
static int launch_smalln(const GConvArgs& a, hipStream_t st) {
  const Phase& p = a.ph[0];
  const int ntap = p.ntap, rows = rup(ntap * 4, 128);
  float* w2 = a.ws;
  float* T = a.ws + (size_t)2 * rows * a.Cs;
  hipLaunchKernelGGL(smalln_weight_kernel, dim3(cdiv(rows * a.Cs, 256)), dim3(256), 0, st, a.wt, w2, a.Kp, a.Cs,
                     ntap, p.tw, p.wbase, p.wsy, p.wsx, rows, a.in16);
  IPR_LAUNCH_CHECK();
  GConvArgs g;
  memset(&g, 0, sizeof(g));
  geom_forward_form(g, a.B, a.IH, a.IW, a.Cs, a.IH, a.IW, ntap * 4, 1, 1, 1, 0);
  g.in = a.in; g.wt = w2; g.out = T; g.in16 = a.in16; g.in_ps = a.in_ps;
  g.planar_M = a.B * a.IH * a.IW;
  g.flops = a.flops;                       // the algorithmic FLOPs of the convolution are accounted here
  int rc = launch_gconv(g, st);
  if (rc) return rc;
  TapGatherArgs t;
  memset(&t, 0, sizeof(t));
  t.T = T; t.bias = a.bias; t.aux = a.aux; t.out = a.out;
  t.B = a.B; t.IH = a.IH; t.IW = a.IW; t.OH = a.OH; t.OW = a.OW; t.N = a.N;
  t.th = p.th; t.tw = p.tw; t.dy0 = p.dy0; t.dx0 = p.dx0; t.dys = p.dys; t.dxs = p.dxs;
  t.pad_mode = a.pad_mode; t.act = a.act; t.slope = a.slope; t.aux_act = a.aux_act; t.aux_slope = a.aux_slope;
  t.plane = (long long)a.B * a.IH * a.IW * 4;
  const long long total = (long long)a.B * a.OH * a.OW;
  hipLaunchKernelGGL(tap_gather_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)),
                     dim3(256), 0, st, t);
  IPR_LAUNCH_CHECK();
  return 0;
}

// Times candidates 0..ncand-1 on the caller's stream and returns the fastest.  run(c) returns 0 on success,
// -1 if c does not apply to this geometry (skipped), anything else is an error (returned through *err).
// Two passes: 3 launches of every candidate, then 10 launches of those within 8 % of the best - with ~20
// candidates a single short timing picks a noise winner often enough to cost 1-2 % of a step.
template <class Run>
static int tune_pick(int ncand, Run run, hipStream_t st, int fallback, float* best_us, int* err) {
  *err = 0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<float> ms(ncand, -1.f);
  auto time_one = [&](int c, int reps) -> float {
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) run(c);
    (void)hipEventRecord(e1, st);
    float t = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) return -1.f;
    return t / reps;
  };
  float best = 1e30f;
  int pick = fallback;
  for (int c = 0; c < ncand && !*err; ++c) {
    const int rc = run(c);                    // warm-up (also sets the LDS attribute)
    if (rc == -1) continue;
    if (rc) { *err = rc; break; }
    ms[c] = time_one(c, 3);
    if (ms[c] >= 0.f && ms[c] < best) { best = ms[c]; pick = c; }
  }
  if (!*err && best < 1e30f) {
    const float cut = best * 1.08f;
    best = 1e30f;
    for (int c = 0; c < ncand; ++c) {
      if (ms[c] < 0.f || ms[c] > cut) continue;
      const float t = time_one(c, 10);
      if (t >= 0.f && t < best) { best = t; pick = c; }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (best_us) *best_us = best * 1000.f;
  return pick;
}

static int launch_gconv(const GConvArgs& ain, hipStream_t st) {
  GConvArgs a = ain;
  {
    const unsigned long long esz = a.in16 ? 2ull : 4ull;
    unsigned long long inb = (unsigned long long)a.B * a.IH * a.IW * a.Cs * esz;
    unsigned long long wtb = (unsigned long long)rup(a.wmod > 0 ? a.wmod : a.N, 128) * a.Kp * esz;
    IPR_CHECK(a.in16 != 1 || (g_math == IPRGAN_MATH_BF16 && (a.Cs % 64) == 0 && !a.wmod && a.ksplit <= 1),
              "conv: bf16 activations need IPRGAN_MATH_BF16, a channel count that is a multiple of 64 and a regular convolution");
    IPR_CHECK(a.in16 != 2 || (g_math == IPRGAN_MATH_FP32X3 && (a.Cs % 32) == 0),
              "conv: three-plane activations need IPRGAN_MATH_FP32X3 and a channel count that is a multiple of 32");
    if (a.in16 == 2) {           // plane strides: the caller's (a batch slice of a larger tensor) or the contiguous ones
      IPR_CHECK(inb < 0x2fffffffull && wtb < 0x2fffffffull, "conv: three-plane tensor larger than 2 GiB (%llu / %llu bytes per plane)", inb, wtb);
      if (!a.in_ps) a.in_ps = (unsigned)inb;
      a.wt_ps = (unsigned)wtb;
      inb += 2ull * a.in_ps; wtb *= 3ull;
    }
    IPR_CHECK(inb < 0x7fffffffull && wtb < 0x7fffffffull, "conv: tensor larger than 2 GiB (%llu / %llu bytes)", inb, wtb);
    a.in_bytes = (unsigned)inb; a.wt_bytes = (unsigned)wtb;
    // output (and the same-shaped fused-derivative / residual operands): addressed through buffer descriptors too
    const unsigned long long oel = a.planar_M ? (unsigned long long)a.Ns * a.planar_M
                                   : a.ksplit > 1 ? (unsigned long long)a.ksplit * a.ph[0].M * a.Ns
                                                  : (unsigned long long)a.B * a.OH * a.OW * a.Ns;
    unsigned long long outb = oel * (a.out16 ? 2ull : 4ull);
    const unsigned long long auxb = a.aux ? oel * (a.aux16 ? 2ull : 4ull) : 0ull;      // (no operand: nothing to bound)
    if (a.out16 == 2) {
      IPR_CHECK(!a.planar_M && a.ksplit <= 1, "conv: three-plane output on a workspace pass");
      if (!a.out_ps) a.out_ps = (unsigned)outb;
      outb += 2ull * a.out_ps;
    }
    IPR_CHECK(a.aux16 != 2, "conv: the fused-derivative operand of a three-plane tensor is passed as its h plane (aux16 = 1)");
    IPR_CHECK(outb < 0x7fffffffull && auxb < 0x7fffffffull, "conv: output tensor larger than 2 GiB (%llu bytes)", outb);
    a.out_bytes = (unsigned)outb; a.aux_bytes = (unsigned)auxb;
    a.linear_out = (a.nphase == 1 && a.osy == 1 && a.osx == 1 && a.ph[0].ooy == 0 && a.ph[0].oox == 0) ? 1 : 0;
  }
  IPR_CHECK(a.nphase >= 1 && a.nphase <= 4, "conv: stride %d unsupported (max 2)", a.osy);
  if (g_fewin && a.Cs == 4 && a.nphase == 1 && a.osy == 1 && a.osx == 1 && a.ph[0].ooy == 0 && a.ph[0].oox == 0 &&
      !a.stat_part && a.ksplit <= 1 && !a.planar_M && !a.wmod && !a.in16 && a.Ns >= 32) {
    // few input channels (RGB stems; backward-data of RGB heads): direct form on the vector ALU (fewin_conv_kernel)
    const Phase& p = a.ph[0];
    const int lw = (FEWIN_T - 1) * a.isx + (p.tw - 1) * (p.dxs < 0 ? -p.dxs : p.dxs) + 1;
    const int lh = (FEWIN_T - 1) * a.isy + (p.th - 1) * (p.dys < 0 ? -p.dys : p.dys) + 1;
    const bool simple = fewin_simple_act(a.act) && (!a.aux || fewin_simple_act(a.aux_act));
    const float neg_act = fewin_neg(a.act, a.slope), neg_aux = fewin_neg(a.aux_act, a.aux_slope);
    if (g_math == IPRGAN_MATH_BF16) {          // bf16 math: the same tile on the matrix cores (fewin_mfma_kernel)
      const int ksteps = cdiv(p.ntap * 4, 16), kpad = ksteps * 16 + 8;
      size_t smem16 = (size_t)((lw * lh + 1) & ~1) * 8 + (size_t)64 * kpad * 2 + (size_t)ksteps * 4 * sizeof(int) + 16;
      if (a.out16) smem16 += 32768;                          // + the staged bf16 tile (its own region: the weights persist)
      if (smem16 <= 150 * 1024) {
        static bool attr16_set = false;
        if (!attr16_set) {
          (void)hipFuncSetAttribute((const void*)fewin_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
          (void)hipFuncSetAttribute((const void*)fewin_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
          attr16_set = true;
        }
        const int ntiles = a.B * cdiv(a.OH, FEWIN_T) * cdiv(a.OW, FEWIN_T);
        dim3 grid((unsigned)ntiles, (unsigned)cdiv(a.Ns, 64));
        if (simple && (a.Ns % 8) == 0 && lw * lh <= 768) {       // persistent: ~4 blocks per CU walk the tiles
          const int per_cu = 3;
          if (ntiles > 256 * per_cu) grid.x = 256 * per_cu;
          prof_launch(fewin_mfma_kernel<true>, grid, dim3(256), smem16, st, 18, a.flops, a, lw, lh, kpad, neg_act, neg_aux, ntiles);
        } else {
          prof_launch(fewin_mfma_kernel<false>, grid, dim3(256), smem16, st, 18, a.flops, a, lw, lh, kpad, neg_act, neg_aux, ntiles);
        }
        IPR_LAUNCH_CHECK();
        return 0;
      }
    }
    static const int g_fewin8 = getenv("IPRGAN_FEWIN8") ? atoi(getenv("IPRGAN_FEWIN8")) : 1;      // A/B switch
    if (g_fewin8 && simple && a.out16 == 2 && !a.res && (a.Ns % 64) == 0) {      // three-plane output: 8 channels per thread, 16-byte plane stores
      const int lw8 = (FEWIN8_W - 1) * a.isx + (p.tw - 1) * (p.dxs < 0 ? -p.dxs : p.dxs) + 1;
      const int lh8 = (FEWIN8_H - 1) * a.isy + (p.th - 1) * (p.dys < 0 ? -p.dys : p.dys) + 1;
      const size_t smem8 = ((size_t)lw8 * lh8 + (size_t)(p.ntap < FEWIN_TAPS ? p.ntap : FEWIN_TAPS) * 64) * sizeof(f32x4);
      if (smem8 <= 64 * 1024) {
        static bool attr8_set = false;
        if (!attr8_set) { (void)hipFuncSetAttribute((const void*)fewin_conv8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr8_set = true; }
        dim3 grid((unsigned)(a.B * cdiv(a.OH, FEWIN8_H) * cdiv(a.OW, FEWIN8_W)), (unsigned)(a.Ns / 64));
        prof_launch(fewin_conv8_kernel, grid, dim3(256), smem8, st, 18, a.flops, a, lw8, lh8, neg_act, neg_aux);
        IPR_LAUNCH_CHECK();
        return 0;
      }
    }
    const size_t smem = ((size_t)lw * lh + (size_t)(p.ntap < FEWIN_TAPS ? p.ntap : FEWIN_TAPS) * 64) * sizeof(f32x4);
    if (smem <= 150 * 1024) {
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fewin_conv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)fewin_conv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
      }
      dim3 grid((unsigned)(a.B * cdiv(a.OH, FEWIN_T) * cdiv(a.OW, FEWIN_T)), (unsigned)cdiv(a.Ns, 64));
      if (simple) prof_launch(fewin_conv_kernel<true>, grid, dim3(256), smem, st, 18, a.flops, a, lw, lh, neg_act, neg_aux);
      else prof_launch(fewin_conv_kernel<false>, grid, dim3(256), smem, st, 18, a.flops, a, lw, lh, neg_act, neg_aux);
      IPR_LAUNCH_CHECK();
      return 0;
    }
  }
  if (smalln_eligible(a) && a.ws && a.ws_floats >= smalln_ws_floats(a)) return launch_smalln(a, st);
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  const int N = a.Ns;
  if (N <= 32 && a.in16 != 2) return launch_gconv_t<4, 1, 1, 1>(a, st);
  auto run = [&](int tile) {
    switch (tile) {
      case 0: return launch_gconv_t<2, 2, 2, 2>(a, st);
      case 1: return launch_gconv_t<2, 2, 2, 1>(a, st);
      case 3: return launch_gconv_t<2, 2, 1, 2>(a, st);
      case 4: return launch_gconv_t<2, 4, 2, 1>(a, st);      // 128x128, 8 waves of 64x32
      case 5: return launch_gconv_t<4, 2, 1, 1>(a, st);      // 128x64, 8 waves of 32x32
      case 6: return launch_gconv_bf16big<2, 2, 4, 2>(a, st);   // bf16 only: 256x128, 4 waves of 128x64
      case 7: return launch_gconv_bf16big<2, 2, 4, 4>(a, st);   // bf16 only: 256x256, 4 waves of 128x128
      case 8: case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: case 17:   // LDS-DMA ring tiles (conv_pipe.hip)
        return launch_gconv_pipe(a, tile - 8, st, &t_last_bm);
      case 18: case 19: case 20: case 21: case 22: case 23: case 24: case 25:                   // three-plane ring tiles (conv_x3.hip)
        return launch_gconv_x3p(a, tile - 18, st, &t_last_bm);
      case 26: case 27:                                                                         // ... their halo form (stride-1 gathers)
        return launch_gconv_x3h(a, tile - 26, st, &t_last_bm);
      case 28: case 29: case 30: case 31:                                                       // ... on v_mfma_f32_16x16x32_bf16
        return launch_gconv_x3p16(a, tile - 28, st, &t_last_bm);
      case 32: case 33: case 34: case 35: case 36: case 37:                                     // ... with dedicated loader waves (round 5)
        return launch_gconv_x3ws(a, tile - 32, st, &t_last_bm);
      default: return launch_gconv_t<2, 2, 1, 1>(a, st);
    }
  };
  // heuristic: widest tile that still yields >= 1.5 blocks per CU (256 CUs)
  auto blocks = [&](int bm, int bn) { return (long long)cdiv(maxM, bm) * cdiv(N, bn) * a.nphase; };
  int tile = 2;
  if (N >= 128 && blocks(128, 128) >= 384) tile = 0;
  else if (blocks(128, 64) >= 384) tile = 1;
  if (g_force_tile >= 0) {
    if ((g_force_tile == 0 || g_force_tile == 3 || g_force_tile == 4) && N < 128) return run(2);
    const int rc = run(g_force_tile);
    return rc == -1 ? run(2) : rc;          // a bf16-only tile forced on an fp32 launch
  }
  if (!g_autotune) return run(tile);

  // autotune (the reference trains with cudnn.benchmark = True, train.py:44-45): the first launch of a new
  // geometry times every tile on the caller's stream and keeps the fastest.  Tiles only change the summation
  // order, results stay within fp32 rounding of each other.
  TuneKey key = {{a.B, a.IH, a.IW, a.Cs, a.OH, a.OW, a.Ns, a.isy, a.osy, a.nphase, a.ph[0].th, a.ph[0].tw,
                  a.ph[0].ohg, a.ph[0].owg, a.pad_mode + 16 * g_math + 64 * a.ksplit + 8192 * (a.wmod > 0) + 16384 * (a.rs0 != nullptr) +
                      32768 * (a.stat_part != nullptr) + 65536 * a.in16 + 262144 * a.out16 + 1048576 * (a.bn_mean != nullptr), a.Kp}};
  {
    int cached;
    if (tune_lookup(key, &cached)) return run(cached);
  }
  const bool prof_was = g_prof_on;
  g_prof_on = false;
  float best_us = 0.f;
  int err = 0;
  const int best = tune_pick(38, [&](int cand) -> int {
    if ((cand == 0 || cand == 3 || cand == 4) && N < 128) return -1;
    if ((cand == 6 || cand == 7) && (long long)cdiv(maxM, 256) * cdiv(N, 128) * a.nphase < 256) return -1;    // not even one block per CU
    return run(cand);
  }, st, tile, &best_us, &err);
  g_prof_on = prof_was;
  if (err) return err;
  tune_store(key, best);
  if (getenv("IPRGAN_TUNE_LOG"))
    fprintf(stderr, "[iprgan tune] gconv B%d in %dx%dx%d out %dx%dx%d taps %dx%d phases %d -> tile %d (%.1f us)\n", a.B,
            a.IH, a.IW, a.Cs, a.OH, a.OW, a.Ns, a.ph[0].th, a.ph[0].tw, a.nphase, best, best_us);
  return run(best);
}

// ---- split-K for regular convolutions with few output tiles.  A 3x3 512->512 layer on a 6x6 map at batch 64 (VGG
// block 5 of the SRGAN content loss, networks/vgg.py) has 2304 output rows: 288 tiles of 64x64 for 256 CUs, a round and
// an eighth, at 62 TFLOP/s, while its reduction is 4608 long.  Such layers split the K loop over blockIdx.z like the
// full-map path below (slabs of M*Ns partial sums, fixed-order reduce that applies the whole epilogue once); the split
// count 1..4 is autotuned per geometry together with the tile.
#define SPLITK_MAX 4
static int g_force_splitk = -1;
static bool splitk_geom_ok(const GConvArgs& a) {
  if (a.nphase != 1 || a.osy != 1 || a.osx != 1 || a.ph[0].ooy || a.ph[0].oox) return false;
  const long long M = a.ph[0].M, blocks = (long long)cdiv((int)M, 64) * cdiv(a.Ns, 64);
  static const long long lim_blocks = getenv("IPRGAN_SPLITK_BLOCKS") ? atoll(getenv("IPRGAN_SPLITK_BLOCKS")) : 1280;     // (read once)
  static const long long lim_out = getenv("IPRGAN_SPLITK_OUT") ? atoll(getenv("IPRGAN_SPLITK_OUT")) : (5ll << 20);
  return a.Ns >= 64 && (a.Cs % 32) == 0 && blocks <= lim_blocks && a.ph[0].steps >= 32 && M * a.Ns <= lim_out;
}
static size_t splitk_ws_floats(const GConvArgs& a) {
  return splitk_geom_ok(a) ? (size_t)SPLITK_MAX * (size_t)a.ph[0].M * a.Ns : 0;
}
static bool splitk_eligible(const GConvArgs& a, const float* ws) {
  // storage kinds: all fp32, or all three-plane (the fused-derivative operand then arrives as its h plane: aux16 == 1)
  const bool kinds = a.in16 == 2 ? (a.out16 == 2 && (!a.aux || a.aux16 == 1)) : (!a.in16 && !a.out16 && !a.aux16);
  return ws && splitk_geom_ok(a) && !a.stat_part && !a.rs0 && kinds && !a.planar_M && !a.wmod &&
         a.ksplit <= 1 && a.res != a.out && a.aux != a.out && g_force_tile < 0;
}
static int gconv_splitk(const GConvArgs& a0, float* ws, hipStream_t st) {
  const int M = a0.ph[0].M;
  auto run = [&](int cand) -> int {
    const int ks = cand + 1;
    if (ks == 1) return launch_gconv(a0, st);
    GConvArgs a = a0;
    a.ksplit = ks; a.out = ws; a.bias = nullptr; a.act = IPRGAN_ACT_NONE; a.aux = nullptr; a.res = nullptr;
    a.out16 = 0; a.aux16 = 0; a.out_ps = 0;              // the slabs are fp32 whatever the tensors' storage kind
    const int rc = launch_gconv(a, st);
    if (rc) return rc;
    const unsigned n4 = (unsigned)((size_t)M * a0.Ns / 4);
    const size_t ps = a0.out_ps ? (size_t)a0.out_ps / 2 : (size_t)M * a0.Ns;       // plane stride of the output, elements
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((int)n4, 256) < 2048 ? cdiv((int)n4, 256) : 2048), dim3(256), 0, st,
                       (const f32x4*)ws, a0.bias, (f32x4*)a0.out, n4, (unsigned)(a0.Ns / 4), a0.N, ks, a0.act, a0.slope,
                       (const f32x4*)a0.aux, a0.aux_act, a0.aux_slope, (const f32x4*)a0.res, a0.out16 == 2 ? 2 : 0, ps);
    IPR_LAUNCH_CHECK();
    return 0;
  };
  const long long blocks = (long long)cdiv(M, 64) * cdiv(a0.Ns, 64);
  int guess = blocks <= 384 ? (int)((768 + blocks / 2) / blocks) : 1;        // ~3 blocks of 4 waves per CU
  guess = guess > SPLITK_MAX ? SPLITK_MAX : guess < 1 ? 1 : guess;
  if (g_force_splitk >= 1) return run((g_force_splitk > SPLITK_MAX ? SPLITK_MAX : g_force_splitk) - 1);
  if (!g_autotune) return run(guess - 1);
  TuneKey key = {{a0.B, a0.IH, a0.IW, a0.Cs, a0.OH, a0.OW, a0.Ns, a0.isy, a0.osy, a0.nphase, a0.ph[0].th, a0.ph[0].tw,
                  a0.ph[0].ohg, a0.ph[0].owg, a0.pad_mode + 16 * g_math + (1 << 20) + (1 << 21) * (a0.aux != nullptr) +
                      (1 << 22) * (a0.res != nullptr), a0.Kp}};
  int cached;
  if (tune_lookup(key, &cached)) return run(cached);
  const bool prof_was = g_prof_on;
  g_prof_on = false;
  float best_us = 0.f;
  int err = 0;
  const int best = tune_pick(SPLITK_MAX, run, st, guess - 1, &best_us, &err);
  g_prof_on = prof_was;
  if (err) return err;
  tune_store(key, best);
  if (getenv("IPRGAN_TUNE_LOG"))
    fprintf(stderr, "[iprgan tune] split-K B%d in %dx%dx%d out %dx%dx%d taps %dx%d -> %d split(s) (%.1f us)\n", a0.B, a0.IH,
            a0.IW, a0.Cs, a0.OH, a0.OW, a0.Ns, a0.ph[0].th, a0.ph[0].tw, best + 1, best_us);
  return run(best);
}

// ---- full-map convolutions (kernel = whole input map, output 1x1: the "FC" layer of networks/discriminator_96.py:20)
// are skinny GEMMs, M = batch rows: Y[B][Cout] = X[B][H*W*Cin] W^T.  As a convolution they fill 1/16 of the GPU
// (B = 64: one 64-row tile per 64 output channels) and their backward-data multiplies 35 of 36 taps by zero.  Here the
// forward pass is a 1x1 convolution over the flattened NHWC map (its (tap, channel) order IS the K order of the
// prepared operand) with the K loop split over blockIdx.z and a fixed-order slab reduce; backward-data is a 1x1
// convolution with (tap, channel) output columns whose operand rows are read out of the backward operand in place.
static bool fullmap_conv(const iprgan_conv_desc* d) {
  return !d->transposed && d->KH == d->H && d->KW == d->W && d->pad == 0 && d->stride == 1 && d->KH * d->KW > 1 &&
         (d->Cin % 32) == 0 && d->pad_mode == IPRGAN_PAD_ZERO;
}
static int fullmap_ksplit(const iprgan_conv_desc* d) {
  const int steps = d->KH * d->KW * d->Cin / 32;
  const int ntile = cdiv(c4(d->Cout), 64) * cdiv(d->B, 64);
  int ks = cdiv(576, ntile);                  // ~2 blocks of 4 waves per CU
  if (ks > steps / 4) ks = steps / 4;         // at least 4 K steps per block
  if (ks > 64) ks = 64;
  return ks < 1 ? 1 : ks;
}

// ---- backward-weight ------------------------------------------------------------------------
struct WGradPlan {
  int N, Cq, Ps, Qs, ntap, Kw, Nrows, bn, bk, tiles, nsplit, cps, M, w8;
  int variant;      // 0: wgrad_kernel ([m][n] LDS image), 1: wgrad_t_kernel (transposed image), 2: the same, XCD-contiguous splits
};
// cand = 4 * target + shape.  shape: 0 = 128x128 tiles, 1 = 64x64, 2 = 128x64, 3 = 128x128 on 8 waves;
// target blocks {768, 1536, 3072, 6144, 384}.  768 blocks are ONE round of 3 blocks per CU; on long reductions
// more, shorter splits balance better across CUs (north-star shape: 113 -> 129 TFLOP/s at 3072) until the slab
// traffic of the extra splits costs more.  (Starting the blocks of a round out of phase with s_sleep did nothing.)
#define WGRAD_NSHAPE 4
#define WGRAD_NTARGET 5
#define WGRAD_NBASE (WGRAD_NSHAPE * WGRAD_NTARGET)
#define WGRAD_NVARIANT 3
#define WGRAD_NCAND (WGRAD_NBASE * WGRAD_NVARIANT)     // cand = WGRAD_NBASE * variant + 4 * target + shape
#define WGRAD_MAX_SLAB_FLOATS ((size_t)128 << 20)      // candidates needing more than 512 MB of slabs are skipped
// Roles of the two tensors.  P is walked on its own pixel grid (the GEMM's reduction index m), Q is gathered
// around it tap by tap; the slab is [P channel][(tap, Q channel)].
//   Conv2d          : P = dy on the output grid, Q = x at  o * stride + (tap - pad)
//   ConvTranspose2d : P = x  on the input grid,  Q = dy at i * stride + (tap - pad)
//   Conv2d, swapped : P = x  on the input grid,  Q = dy at i - (tap - pad)          (stride 1 only)
// The swapped form serves layers with a handful of output channels (RGB heads 64->3 k7/k9, patch logits
// 512->1): with P = dy the MFMA tile has 3 useful rows of 32; swapped, the 4-padded output channels sit in the K
// dimension (3 useful of 4) and the rows are the 64+ input channels.  With ReflectionPad2d the input is first
// copied into its padded form (reflect_pad_kernel) and the layer becomes a pad-0 convolution of that image.
struct WGeom {
  int N, Cq, PH, PW, QH, QW, pad, flip, swap, padded;
};
static WGeom wgrad_geom(const iprgan_conv_desc* d) {
  const Shape s = out_shape(d);
  WGeom g;
  memset(&g, 0, sizeof(g));
  g.flip = 1; g.pad = d->pad;
  if (d->transposed) {
    g.N = d->Cin; g.Cq = d->Cout; g.PH = d->H; g.PW = d->W; g.QH = s.OH; g.QW = s.OW;
  } else if (d->stride == 1 && d->Cout <= 16 && d->Cin >= 32) {
    g.swap = 1; g.flip = -1;
    g.padded = d->pad_mode == IPRGAN_PAD_REFLECT && d->pad > 0;
    g.N = d->Cin; g.Cq = d->Cout; g.QH = s.OH; g.QW = s.OW;
    g.PH = d->H + (g.padded ? 2 * d->pad : 0); g.PW = d->W + (g.padded ? 2 * d->pad : 0);
    if (g.padded) g.pad = 0;
  } else {
    g.N = d->Cout; g.Cq = d->Cin; g.PH = s.OH; g.PW = s.OW; g.QH = d->H; g.QW = d->W;
  }
  return g;
}
static size_t rup4(size_t n) { return (n + 3) & ~(size_t)3; }
static size_t wgrad_padded_floats(const iprgan_conv_desc* d) {
  const WGeom g = wgrad_geom(d);
  return g.padded ? (size_t)d->B * g.PH * g.PW * c4(d->Cin) : 0;
}

// both operands bf16 in HBM and the bf16 [32][128] image applicable: the IN16 kernel runs on the 128x128 tiles
static bool wgrad_in16(const iprgan_conv_desc* d) {
  return d->x_bf16 == 1 && d->y_bf16 == 1 && g_math == IPRGAN_MATH_BF16 && d->pad_mode == IPRGAN_PAD_ZERO;
}
// both operands three-plane tensors: the split kernel on the 128x128 tiles reads them plane by plane (any padding mode)
static bool wgrad_in3p(const iprgan_conv_desc* d) {
  return d->x_bf16 == 2 && d->y_bf16 == 2 && g_math == IPRGAN_MATH_FP32X3;
}
static bool wgrad_has_planes(const iprgan_conv_desc* d) { return d->x_bf16 == 2 || d->y_bf16 == 2; }
static bool wgrad_plan_c(const iprgan_conv_desc* d, int cand, WGradPlan& p) {
  const WGeom g = wgrad_geom(d);
  if (cand < 0 || cand >= WGRAD_NCAND) return false;
  p.variant = cand / WGRAD_NBASE;
  cand %= WGRAD_NBASE;
  if (p.variant && (g.N <= 32 || g_math == IPRGAN_MATH_BF16)) return false;   // the 32-row tile and the bf16 image exist in the first form only
  if (p.variant && (d->x_bf16 || d->y_bf16)) return false;      // the transposed-image kernel reads fp32 tensors only
  if (wgrad_in3p(d) && (g.swap || (cand % WGRAD_NSHAPE != 0 && cand % WGRAD_NSHAPE != 3))) return false;
  // one three-plane operand next to an fp32 one (RGB stems / heads in fp32x3 mode): the fp32 tiles of the first form
  // widen it on arrival (WGradArgs::p16 / q16 = 2); a bf16 operand next to it has no form
  if (wgrad_has_planes(d) && !wgrad_in3p(d) && (d->x_bf16 == 1 || d->y_bf16 == 1)) return false;
  p.N = g.N;
  p.Cq = g.Cq;
  p.Ps = c4(p.N); p.Qs = c4(p.Cq);
  p.ntap = d->KH * d->KW;
  p.M = d->B * g.PH * g.PW;
  const int K = p.ntap * p.Qs;
  static const int targets[WGRAD_NTARGET] = {768, 1536, 3072, 6144, 384};
  const int shape = (cand % WGRAD_NSHAPE) == 3 ? 0 : cand % WGRAD_NSHAPE;
  const int target = targets[cand / WGRAD_NSHAPE];
  p.w8 = (cand % WGRAD_NSHAPE) == 3 ? 1 : 0;
  const bool half_ok = (wgrad_in16(d) || wgrad_in3p(d)) && p.N >= 64;     // 64 rows of P on the 128-row image: the upper half reads zeros
  if (p.w8 && ((p.N < 128 && !half_ok) || K < 128)) return false;
  if (p.N <= 32) {
    if (shape != 0) return false;
    p.bn = 32; p.bk = 128;
  } else if (shape == 0) {
    // (a 64-row P of bf16 tensors still takes the 128x128 bf16 tile: the upper half of the rows reads zeros)
    if ((p.N < 128 && !half_ok) || K < 128) return false;
    p.bn = 128; p.bk = 128;
  } else if (shape == 1) {
    p.bn = 64; p.bk = 64;
  } else {
    if (p.N < 128) return false;
    p.bn = 128; p.bk = 64;
  }
  p.Kw = rup(K, p.bk);
  p.Nrows = rup(p.N, p.bn);
  p.tiles = (p.Kw / p.bk) * (p.Nrows / p.bn);
  const int chunks = cdiv(p.M, 32);
  int want = cdiv(target, p.tiles);
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  p.cps = cdiv(chunks, want);
  p.nsplit = cdiv(chunks, p.cps);
  if (cand >= WGRAD_NSHAPE && (size_t)p.nsplit * p.Nrows * p.Kw > WGRAD_MAX_SLAB_FLOATS) return false;
  return true;
}
static WGradPlan wgrad_plan(const iprgan_conv_desc* d) {       // the un-tuned default
  WGradPlan p;
  if (!wgrad_plan_c(d, 0, p)) wgrad_plan_c(d, 1, p);
  return p;
}
// the halo form (wgrad_halo.hip): candidates WGRAD_NCAND + i, i = index into these block targets
bool wgrad_halo_eligible(const iprgan_conv_desc* d);
int wgrad_halo_nsplit(const iprgan_conv_desc* d, int variant, int target_blocks);
int launch_wgrad_halo(const iprgan_conv_desc* d, const void* x, const void* dy, float* ws, int variant, int target_blocks,
                      hipStream_t st, int* nsplit_out, int* Nrows_out, int* Kw_out);
bool wgrad_rgb_eligible(const iprgan_conv_desc* d);
int wgrad_rgb_nsplit(const iprgan_conv_desc* d, int target_blocks);
int launch_wgrad_rgb(const iprgan_conv_desc* d, const void* x, const void* dy, float* ws, int target_blocks, hipStream_t st,
                     int* nsplit_out, int* Nrows_out, int* Kw_out);
bool wgrad_x3h_eligible(const iprgan_conv_desc* d);
int wgrad_x3h_nsplit(const iprgan_conv_desc* d, int target_blocks);
int launch_wgrad_x3h(const iprgan_conv_desc* d, const void* x, const void* dy, float* ws, int target_blocks, hipStream_t st,
                     int* nsplit_out, int* Nrows_out, int* Kw_out);
bool wgrad_halo_f32_eligible(const iprgan_conv_desc* d);
int wgrad_halo_f32_nsplit(const iprgan_conv_desc* d, int target_blocks);
int launch_wgrad_halo_f32(const iprgan_conv_desc* d, const float* x, const float* dy, float* ws, int target_blocks,
                          hipStream_t st, int* nsplit_out, int* Nrows_out, int* Kw_out);
#define WGRAD_NH32 3                        // candidates WGRAD_NCAND + WGRAD_NHALO + WGRAD_NRGB + {0, 1, 2}: fp32 halo form
static const int g_h32_targets[WGRAD_NH32] = {256, 512, 1024};
// the next candidates: halo form for three-plane tensors (wgrad_x3.hip) at a target block count.  64 / 128 blocks (round 6,
// appended: the indices of the first three are unchanged) are for layers whose whole weight is ONE tile - SRResNet's 33
// 64 -> 64 k3 layers at 24x24: 256 blocks of 2.25 patches each wrote 256 slabs of 147 KB for a 147 KB gradient, 38 MB per layer
// to write and to read back (the slab reduces were 1.2 ms of the 26 ms SRGAN step)
#define WGRAD_NX3H 5
static const int g_x3h_targets[WGRAD_NX3H] = {256, 512, 1024, 128, 64};
#define WGRAD_X3H0 (WGRAD_NCAND + WGRAD_NHALO + WGRAD_NRGB + WGRAD_NH32)
#define WGRAD_NALL (WGRAD_X3H0 + WGRAD_NX3H)
#define WGRAD_NRGB 2                        // candidates WGRAD_NCAND + WGRAD_NHALO + {0, 1}: 256 / 512 blocks
static const int g_rgb_targets[WGRAD_NRGB] = {256, 512};
#define WGRAD_NHALO 9                       // candidate WGRAD_NCAND + 3 * variant + target index
static const int g_halo_targets[3] = {128, 256, 512};
static bool wgrad_halo_ok(const iprgan_conv_desc* d) { return g_math == IPRGAN_MATH_BF16 && wgrad_halo_eligible(d); }
static bool wgrad_rgb_ok(const iprgan_conv_desc* d) { return g_math == IPRGAN_MATH_BF16 && wgrad_rgb_eligible(d); }
static bool wgrad_x3h_ok(const iprgan_conv_desc* d) { return g_math == IPRGAN_MATH_FP32X3 && wgrad_x3h_eligible(d); }
static bool wgrad_h32_ok(const iprgan_conv_desc* d) { return g_math != IPRGAN_MATH_BF16 && !wgrad_has_planes(d) && wgrad_halo_f32_eligible(d); }

static size_t wgrad_slab_floats(const iprgan_conv_desc* d) {   // workspace that fits every candidate
  size_t m = 0;
  if (wgrad_halo_ok(d)) {
    const WGeom g = wgrad_geom(d);
    for (int i = 0; i < WGRAD_NHALO; ++i) {
      const size_t n = (size_t)wgrad_halo_nsplit(d, i / 3, g_halo_targets[i % 3]) * c4(g.N) * d->KH * d->KW * c4(g.Cq);
      if (n > m) m = n;
    }
  }
  if (wgrad_x3h_ok(d)) {
    const WGeom g = wgrad_geom(d);
    for (int i = 0; i < WGRAD_NX3H; ++i) {
      const size_t n = (size_t)wgrad_x3h_nsplit(d, g_x3h_targets[i]) * c4(g.N) * d->KH * d->KW * c4(g.Cq);
      if (n > m && n <= WGRAD_MAX_SLAB_FLOATS) m = n;
    }
  }
  if (wgrad_h32_ok(d)) {
    const WGeom g = wgrad_geom(d);
    for (int i = 0; i < WGRAD_NH32; ++i) {
      const size_t n = (size_t)wgrad_halo_f32_nsplit(d, g_h32_targets[i]) * c4(g.N) * d->KH * d->KW * c4(g.Cq);
      if (n > m && n <= WGRAD_MAX_SLAB_FLOATS) m = n;
    }
  }
  if (wgrad_rgb_ok(d)) {
    const WGeom g = wgrad_geom(d);
    for (int i = 0; i < WGRAD_NRGB; ++i) {
      const size_t n = (size_t)wgrad_rgb_nsplit(d, g_rgb_targets[i]) * c4(g.N) * 64;
      if (n > m) m = n;
    }
  }
  for (int c = 0; c < WGRAD_NCAND; ++c) {
    WGradPlan p;
    if (!wgrad_plan_c(d, c, p)) continue;
    const size_t n = (size_t)p.nsplit * p.Nrows * p.Kw;
    if (n > m) m = n;
  }
  return m;
}

template <int WGM, int WGN, int WM, int WN, int NBUF, bool REFLECT, bool BF16 = false, bool IN16 = false, bool SPLIT = false>
static int launch_wgrad_tn(const WGradArgs& a, const WGradPlan& p, hipStream_t st) {
  constexpr int BN = WGM * WM * 32, BK = WGN * WN * 32;
  const size_t smem = NBUF * (size_t)32 * (BN + BK) * (SPLIT ? 6 : BF16 ? 2 : sizeof(float));
  auto kern = wgrad_kernel<WGM, WGN, WM, WN, NBUF, REFLECT, BF16, IN16, SPLIT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid(p.Kw / BK, p.Nrows / BN, p.nsplit);
  prof_launch(kern, grid, dim3(WGM * WGN * 64), smem, st,
              (SPLIT && IN16) ? 30 : SPLIT ? 28 : BF16 ? 13 : WGM * WGN == 8 ? 11 : BN == 128 ? (BK == 128 ? 4 : 5) : (BN == 64 ? 6 : 7), a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}

template <int WGM, int WGN, int WM, int WN>
static int launch_wgrad_t(const WGradArgs& a, const WGradPlan& p, hipStream_t st) {
  if constexpr (WGM * WM == 4 && WGN * WN == 4) {       // bf16 math: the 128x128 tiles (4 and 8 waves)
    if (a.in16 == 2)
      return a.pad_mode == IPRGAN_PAD_REFLECT ? launch_wgrad_tn<WGM, WGN, WM, WN, 1, true, true, true, true>(a, p, st)
                                              : launch_wgrad_tn<WGM, WGN, WM, WN, 1, false, true, true, true>(a, p, st);
    if (a.in16) return launch_wgrad_tn<WGM, WGN, WM, WN, 1, false, true, true>(a, p, st);
    if (g_math == IPRGAN_MATH_BF16)
      return a.pad_mode == IPRGAN_PAD_REFLECT ? launch_wgrad_tn<WGM, WGN, WM, WN, 1, true, true>(a, p, st)
                                              : launch_wgrad_tn<WGM, WGN, WM, WN, 1, false, true>(a, p, st);
    if (g_math == IPRGAN_MATH_FP32X3 && !a.p16 && !a.q16)
      return a.pad_mode == IPRGAN_PAD_REFLECT ? launch_wgrad_tn<WGM, WGN, WM, WN, 1, true, true, false, true>(a, p, st)
                                              : launch_wgrad_tn<WGM, WGN, WM, WN, 1, false, true, false, true>(a, p, st);
  }
  if (a.in16 == 2) return -1;            // three-plane tensors: the 128x128 split tiles only
  if (a.pad_mode == IPRGAN_PAD_REFLECT)
    return g_nbuf == 1 ? launch_wgrad_tn<WGM, WGN, WM, WN, 1, true>(a, p, st)
                       : launch_wgrad_tn<WGM, WGN, WM, WN, 2, true>(a, p, st);
  return g_nbuf == 1 ? launch_wgrad_tn<WGM, WGN, WM, WN, 1, false>(a, p, st)
                     : launch_wgrad_tn<WGM, WGN, WM, WN, 2, false>(a, p, st);
}

template <int WGM, int WGN, int WM, int WN, bool REFLECT, bool ROW4>
static int launch_wgrad_t2r(const WGradArgs& a, const WGradPlan& p, hipStream_t st) {
  constexpr int BN = WGM * WM * 32, BK = WGN * WN * 32;
  const size_t smem = (size_t)(BN + BK) * 8 * sizeof(f32x4);
  auto kern = wgrad_t_kernel<WGM, WGN, WM, WN, REFLECT, ROW4>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid(p.Kw / BK, p.Nrows / BN, p.nsplit);
  prof_launch(kern, grid, dim3(WGM * WGN * 64), smem, st,
              WGM * WGN == 8 ? 17 : BN == 128 ? (BK == 128 ? 14 : 15) : 16, a.flops, a);
  IPR_LAUNCH_CHECK();
  return 0;
}
template <int WGM, int WGN, int WM, int WN>
static int launch_wgrad_t2(const WGradArgs& a, const WGradPlan& p, hipStream_t st) {
  const bool row4 = (a.PW % 4) == 0;
  if (a.pad_mode == IPRGAN_PAD_REFLECT)
    return row4 ? launch_wgrad_t2r<WGM, WGN, WM, WN, true, true>(a, p, st)
                : launch_wgrad_t2r<WGM, WGN, WM, WN, true, false>(a, p, st);
  return row4 ? launch_wgrad_t2r<WGM, WGN, WM, WN, false, true>(a, p, st)
              : launch_wgrad_t2r<WGM, WGN, WM, WN, false, false>(a, p, st);
}

// partial rows written by the last STATS launch of geometry a: (M tiles of the tile that ran) x (phases)
static int stat_rows_of(const GConvArgs& a) {
  int maxM = 0;
  for (int i = 0; i < a.nphase; ++i) maxM = a.ph[i].M > maxM ? a.ph[i].M : maxM;
  return cdiv(maxM, t_last_bm) * a.nphase;
}

}  // namespace iprgan

using namespace iprgan;

extern "C" {

size_t iprgan_conv_stat_floats(const iprgan_conv_desc* d, int backward) {
  // worst case: 64-row tiles.  forward: output grid of the layer; backward-data: its input grid.  Phases (stride 2
  // transposed / backward forms) split the grid, each phase has its own tile rows.
  const Shape s = out_shape(d);
  const bool fwd_form = backward ? (d->transposed != 0) : (d->transposed == 0);
  const int gh = backward ? d->H : s.OH, gw = backward ? d->W : s.OW, st = fwd_form ? 1 : d->stride;
  const int nph = st * st;
  const long long rows = ((long long)d->B * cdiv(gh, st) * cdiv(gw, st) + 63) / 64 + 1;
  // + room for the compacted rows of the final reduction (norm.hip: compact_partials; 32 rows per group, <= B groups)
  return ((size_t)rows * nph + 32 * (size_t)d->B) * 2 * c4(backward ? d->Cin : d->Cout);
}

size_t iprgan_conv_wfwd_floats(const iprgan_conv_desc* d) {
  // Conv2d fwd uses rows=Cout,k=(tap,Cin); ConvT fwd uses rows=Cout,k=(tap,Cin) as well
  // (the operand of a three-plane x is three bf16 planes: 6 bytes per element)
  const size_t n = (size_t)rup(d->Cout, 128) * rup(d->KH * d->KW * c4(d->Cin), 32);
  return d->x_bf16 == 2 ? n + n / 2 : n;
}
size_t iprgan_conv_wbwd_floats(const iprgan_conv_desc* d) {
  const size_t n = (size_t)rup(d->Cin, 128) * rup(d->KH * d->KW * c4(d->Cout), 32);
  return d->y_bf16 == 2 ? n + n / 2 : n;
}

int iprgan_conv_weight_prep(const iprgan_conv_desc* d, const float* w, const float* inv_scale,
                            float* wfwd, float* wbwd, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int ntap = d->KH * d->KW;
  // PyTorch weight dims: Conv2d [D0=Cout][D1=Cin], ConvT [D0=Cin][D1=Cout]
  // fwd operand: rows = Cout, reduce = Cin ; bwd operand: rows = Cin, reduce = Cout
  for (int which = 0; which < 2; ++which) {
    float* dst = which == 0 ? wfwd : wbwd;
    if (!dst) continue;
    const int rows = which == 0 ? d->Cout : d->Cin, red = which == 0 ? d->Cin : d->Cout;
    // which==0: conv -> rows=Cout=D0 (1); convT -> rows=Cout=D1 (0)
    // which==1: conv -> rows=Cin=D1 (0);  convT -> rows=Cin=D0 (1)
    const int row_is_d0 = ((which == 0) != (d->transposed != 0)) ? 1 : 0;
    const int Cs = c4(red), Kp = rup(ntap * Cs, 32), R = rup(rows, 128);
    const long long total = (long long)R * Kp;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(weight_prep_kernel, dim3(blocks), dim3(256), 0, st, w, inv_scale, dst, R, Kp,
                       rows, red, Cs, ntap, row_is_d0, which == 0 ? d->x_bf16 : d->y_bf16);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

static size_t smalln_ws_for(const iprgan_conv_desc* d, bool fwd) {
  // forward of a layer with <= 4 output channels, or backward-data of a layer with <= 4 input channels
  const int n_out = fwd ? d->Cout : d->Cin, c_red = fwd ? d->Cin : d->Cout;
  if (!g_smalln || n_out > 4 || d->stride != 1 || (c4(c_red) % 32) != 0 || d->KH * d->KW < 2) return 0;
  const Shape s = out_shape(d);
  // source image of the gather: fwd of Conv2d / bwd of ConvT read x-shaped [H,W]; the other two read y-shaped
  const bool src_is_in = fwd;
  const long long pix = (long long)d->B * (src_is_in ? (long long)d->H * d->W : (long long)s.OH * s.OW);
  const int ntap = d->KH * d->KW;
  return (size_t)2 * rup(ntap * 4, 128) * c4(c_red) + (size_t)ntap * pix * 4;
}
// geometry of the regular (not full-map) forward / backward-data pass of a layer
static GConvArgs conv_fwd_geom(const iprgan_conv_desc* d) {
  GConvArgs a;
  memset(&a, 0, sizeof(a));
  const Shape s = out_shape(d);
  if (!d->transposed) {
    geom_forward_form(a, d->B, d->H, d->W, d->Cin, s.OH, s.OW, d->Cout, d->KH, d->KW, d->stride, d->pad);
    a.pad_mode = d->pad_mode;
  } else {
    geom_bwd_form(a, d->B, d->H, d->W, d->Cin, s.OH, s.OW, d->Cout, d->KH, d->KW, d->stride, d->pad);
  }
  return a;
}
static GConvArgs conv_bwd_geom(const iprgan_conv_desc* d) {          // zero-padded layers only
  GConvArgs a;
  memset(&a, 0, sizeof(a));
  const Shape s = out_shape(d);
  if (!d->transposed) geom_bwd_form(a, d->B, s.OH, s.OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad);
  else geom_forward_form(a, d->B, s.OH, s.OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad);
  return a;
}
size_t iprgan_conv_fwd_ws_floats(const iprgan_conv_desc* d) {
  if (fullmap_conv(d)) {
    const int ks = fullmap_ksplit(d);
    return ks > 1 ? (size_t)ks * d->B * c4(d->Cout) : 0;
  }
  const size_t sk = splitk_ws_floats(conv_fwd_geom(d)), sn = smalln_ws_for(d, true);
  return sk > sn ? sk : sn;
}


int iprgan_conv_weight_prep_multi(const iprgan_conv_desc* descs, const float* const* w,
                                  const float* const* inv_scale, float* const* wfwd, float* const* wbwd, int n,
                                  void* stream) {
  hipStream_t st = (hipStream_t)stream;
  PrepTable t;
  int cnt = 0;
  int maxrows = 0, maxcs = 0, maxtap = 0;
  auto flush = [&]() -> int {
    if (!cnt) return 0;
    hipLaunchKernelGGL(weight_prep_multi_kernel, dim3(maxrows, cdiv(maxcs, PREP_CCH), cnt), dim3(256),
                       (size_t)PREP_CCH * (maxtap + 1) * sizeof(float), st, t);
    IPR_LAUNCH_CHECK();
    cnt = 0; maxrows = 0; maxcs = 0; maxtap = 0;
    return 0;
  };
  for (int l = 0; l < n; ++l) {
    const iprgan_conv_desc* d = descs + l;
    const int ntap = d->KH * d->KW;
    for (int which = 0; which < 2; ++which) {
      float* dst = which == 0 ? (wfwd ? wfwd[l] : nullptr) : (wbwd ? wbwd[l] : nullptr);
      if (!dst) continue;
      const int rows = which == 0 ? d->Cout : d->Cin, red = which == 0 ? d->Cin : d->Cout;
      PrepEntry& e = t.e[cnt++];
      e.w = w[l]; e.inv_scale = inv_scale ? inv_scale[l] : nullptr; e.dst = dst;
      e.Cs = c4(red); e.Kp = rup(ntap * e.Cs, 32); e.rows_alloc = rup(rows, 128);
      e.Drow = rows; e.Dcol = red; e.ntap = ntap; e.d_ntap = make_fastdiv(ntap);
      if (e.Cs > maxcs) maxcs = e.Cs;
      if (ntap > maxtap) maxtap = ntap;
      e.row_is_d0 = ((which == 0) != (d->transposed != 0)) ? 1 : 0;
      e.out16 = which == 0 ? d->x_bf16 : d->y_bf16;
      if (e.rows_alloc > maxrows) maxrows = e.rows_alloc;
      if (cnt == PREP_MAX) { const int rc = flush(); if (rc) return rc; }
    }
  }
  return flush();
}

int iprgan_conv_fwd(const iprgan_conv_desc* d, const float* x, const float* wfwd, const float* bias,
                    float* y, float* ws, const float* pair_sigma0, const float* pair_sigma1, float* stat_part,
                    int* stat_rows, void* stream) {
  IPR_CHECK(!stat_part || (stat_rows && !fullmap_conv(d) && c4(d->Cout) > 4),
            "conv_fwd: column statistics need the row-count output and a regular convolution with more than 4 channels");
  IPR_CHECK(!pair_sigma0 == !pair_sigma1 && (!pair_sigma0 || ((d->B & 1) == 0 && !fullmap_conv(d))),
            "conv_fwd: a paired pass needs both sigmas, an even batch and a regular convolution");
  IPR_CHECK(d->stride >= 1 && d->stride <= 2, "conv_fwd: stride %d unsupported", d->stride);
  prof_tag("fwd", d);
  GConvArgs a;
  memset(&a, 0, sizeof(a));
  const Shape s = out_shape(d);
  if (fullmap_conv(d)) {
    const int K = d->KH * d->KW * d->Cin, ks = fullmap_ksplit(d);
    geom_forward_form(a, d->B, 1, 1, K, 1, 1, d->Cout, 1, 1, 1, 0);
    a.in = x; a.wt = wfwd; a.flops = 2.0 * d->B * (double)d->Cout * K;
    IPR_CHECK(d->x_bf16 != 1 && d->y_bf16 != 1, "conv_fwd: bf16 activations on a full-map convolution are not built");
    a.in16 = d->x_bf16; a.in_ps = (unsigned)(d->x_pstride * 2);
    if (ks > 1) {
      IPR_CHECK(ws, "conv_fwd: the full-map path needs its workspace (iprgan_conv_fwd_ws_floats)");
      a.ksplit = ks; a.out = ws; a.act = IPRGAN_ACT_NONE;
      int rc = launch_gconv(a, (hipStream_t)stream);
      if (rc) return rc;
      const unsigned n4 = (unsigned)d->B * (unsigned)(c4(d->Cout) / 4);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((int)n4, 256) < 1024 ? cdiv((int)n4, 256) : 1024), dim3(256), 0,
                         (hipStream_t)stream, (const f32x4*)ws, bias, (f32x4*)y, n4, (unsigned)(c4(d->Cout) / 4), d->Cout,
                         ks, d->act, d->slope, (const f32x4*)nullptr, 0, 0.f, (const f32x4*)nullptr, d->y_bf16,
                         d->y_pstride ? (size_t)d->y_pstride : (size_t)d->B * c4(d->Cout));
      IPR_LAUNCH_CHECK();
      return 0;
    }
    a.bias = bias; a.out = y; a.act = d->act; a.slope = d->slope;
    a.out16 = d->y_bf16; a.out_ps = (unsigned)(d->y_pstride * 2);
    return launch_gconv(a, (hipStream_t)stream);
  }
  if (!d->transposed) {
    geom_forward_form(a, d->B, d->H, d->W, d->Cin, s.OH, s.OW, d->Cout, d->KH, d->KW, d->stride, d->pad);
    a.pad_mode = d->pad_mode;
  } else {
    IPR_CHECK(d->pad_mode == IPRGAN_PAD_ZERO, "conv_transpose: reflect pad unsupported");
    geom_bwd_form(a, d->B, d->H, d->W, d->Cin, s.OH, s.OW, d->Cout, d->KH, d->KW, d->stride, d->pad);
  }
  a.in = x; a.wt = wfwd; a.bias = bias; a.out = y; a.aux = nullptr;
  a.rs0 = pair_sigma0; a.rs1 = pair_sigma1;
  a.in16 = d->x_bf16; a.out16 = d->y_bf16;
  a.in_ps = (unsigned)(d->x_pstride * 2); a.out_ps = (unsigned)(d->y_pstride * 2);
  a.flops = d->transposed ? 2.0 * d->B * (double)d->H * d->W * d->Cout * d->Cin * d->KH * d->KW : 2.0 * d->B * (double)s.OH * s.OW * d->Cout * d->Cin * d->KH * d->KW;
  a.act = d->act; a.slope = d->slope;
  a.ws = ws; a.ws_floats = ws ? iprgan_conv_fwd_ws_floats(d) : 0;
  a.stat_part = stat_part; a.stat_mode = 1;
  if (splitk_eligible(a, ws)) return gconv_splitk(a, ws, (hipStream_t)stream);
  const int rc = launch_gconv(a, (hipStream_t)stream);
  if (stat_part && !rc) *stat_rows = stat_rows_of(a);
  return rc;
}

static size_t reflect_padded_floats(const iprgan_conv_desc* d) {
  return rup4((size_t)d->B * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * c4(d->Cin));
}
size_t iprgan_conv_bwd_data_ws_floats(const iprgan_conv_desc* d) {
  if (d->pad_mode != IPRGAN_PAD_REFLECT) {
    const size_t sk = fullmap_conv(d) ? 0 : splitk_ws_floats(conv_bwd_geom(d)), sn = smalln_ws_for(d, false);
    return sk > sn ? sk : sn;
  }
  // [gradient w.r.t. the reflection-padded image | workspace of the few-channel path on that padded grid]
  return reflect_padded_floats(d) + smalln_ws_for(d, false);
}

struct BnAux { const float *mean, *invstd, *gamma, *beta; int act; float slope; };
static int conv_bwd_data_impl(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dx, float* ws,
                              const float* prev_out, int prev_act, float prev_slope, const float* pair_sigma0,
                              const float* pair_sigma1, float* stat_part, int* stat_rows, const float* residual,
                              void* stream, const BnAux* bn);

int iprgan_conv_bwd_data(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dx, float* ws,
                         const float* prev_out, int prev_act, float prev_slope, const float* pair_sigma0,
                         const float* pair_sigma1, float* stat_part, int* stat_rows, const float* residual,
                         void* stream) {
  return conv_bwd_data_impl(d, dy, wbwd, dx, ws, prev_out, prev_act, prev_slope, pair_sigma0, pair_sigma1, stat_part,
                            stat_rows, residual, stream, nullptr);
}

// 1 when iprgan_conv_bwd_data_bn applies to the layer: a zero-padded regular (not full-map) convolution whose input has
// more than 4 channels - the tile kernels' epilogue then takes the norm backward's two reductions
int iprgan_conv_bwd_data_bn_ok(const iprgan_conv_desc* d) {
  // (three-plane tensors: the mask operand of a fused derivative is the h plane only, the norm backward needs all of x)
  return !fullmap_conv(d) && d->pad_mode != IPRGAN_PAD_REFLECT && c4(d->Cin) > 32 && d->stride >= 1 && d->stride <= 2 && d->x_bf16 != 2;
}

int iprgan_conv_bwd_data_bn(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dz, const float* bn_x,
                            const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                            int bn_act, float bn_slope, float* stat_part, int* stat_rows, void* stream) {
  IPR_CHECK(iprgan_conv_bwd_data_bn_ok(d), "conv_bwd_data_bn: layer not eligible (iprgan_conv_bwd_data_bn_ok)");
  IPR_CHECK(bn_x && bn_mean && bn_invstd && stat_part && stat_rows && !bn_gamma == !bn_beta,
            "conv_bwd_data_bn: x, mean, invstd and the partial-sum buffer are required; gamma and beta come together");
  IPR_CHECK(bn_act == IPRGAN_ACT_NONE || bn_act == IPRGAN_ACT_RELU || bn_act == IPRGAN_ACT_LRELU,
            "conv_bwd_data_bn: activation %d has no mask form", bn_act);
  const BnAux bn = {bn_mean, bn_invstd, bn_gamma, bn_beta, bn_act, bn_slope};
  return conv_bwd_data_impl(d, dy, wbwd, dz, nullptr, bn_x, IPRGAN_ACT_NONE, 0.f, nullptr, nullptr, stat_part, stat_rows,
                            nullptr, stream, &bn);
}

// ReflectionPad2d(p) + Conv2d(k = 2p + 1, stride 1) backward-data WITHOUT the padded grid (round 5).
// The padded form (a zero-pad pass over (H + 2p) x (W + 2p) into an fp32 workspace, then iprgan_reflect_fold) costs the extra
// rows, 8 bytes per element of workspace traffic and - what hurt most - the tile count of the padded grid: 34 848 rows instead
// of 32 768 at batch 8 are 274 tiles instead of 256 on 256 CUs (dgrad 140 TFLOP/s where the forward pass runs at 217).  Here:
//   1. the border STRIPS of the padded gradient - positions outside the image, the only ones that fold onto other pixels - as
//      four phases of one small launch into `ws` (top / bottom: p rows of W + 2p; left / right: p columns of H), each with
//      only the taps that can reach the image (p of the 2p + 1 rows or columns of the kernel);
//   2. the image itself as an ordinary zero-padded backward-data pass straight into dx (every epilogue feature);
//   3. reflect_ring_fix_kernel adds the strips to the ring pixels they mirror, inside the activation derivative.
// (acc + mirror) * act' + residual becomes (acc * act' + residual) + mirror * act': one fp32 rounding apart.
static int g_reflect_direct = getenv("IPRGAN_REFLECT_DIRECT") ? atoi(getenv("IPRGAN_REFLECT_DIRECT")) : 1;
static bool reflect_direct_ok(const iprgan_conv_desc* d) {
  const int P = d->pad;
  return g_reflect_direct && d->pad_mode == IPRGAN_PAD_REFLECT && !d->transposed && d->stride == 1 && P >= 1 && d->KH == 2 * P + 1 &&
         d->KW == 2 * P + 1 && 2 * P + 2 <= d->H && 2 * P + 2 <= d->W && (c4(d->Cout) % 32) == 0 && c4(d->Cin) >= 32 &&
         d->x_bf16 != 1 && d->y_bf16 != 1;
}
static int conv_bwd_data_reflect_direct(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dx, float* ws,
                                        const float* prev_out, int prev_act, float prev_slope, const float* residual,
                                        hipStream_t st) {
  const Shape s = out_shape(d);
  const int P = d->pad, H = d->H, W = d->W, HP = H + 2 * P, WP = W + 2 * P;
  {
    GConvArgs b;
    memset(&b, 0, sizeof(b));
    geom_bwd_form(b, d->B, s.OH, s.OW, d->Cout, HP, WP, d->Cin, d->KH, d->KW, 1, 0);     // source pixel = padded position - tap
    const Phase base = b.ph[0];
    b.nphase = 4;
    for (int i = 0; i < 4; ++i) b.ph[i] = base;
    Phase& top = b.ph[0];        // rows 0 .. P-1 of the padded grid: kernel rows 0 .. P-1 reach the image
    top.th = P; top.ohg = P; top.owg = WP; top.ooy = 0; top.oox = 0; top.dy0 = 0; top.dx0 = 0; top.wbase = 0;
    Phase& bot = b.ph[1];        // rows H+P .. H+2P-1: kernel rows P+1 .. 2P
    bot.th = P; bot.ohg = P; bot.owg = WP; bot.ooy = H + P; bot.oox = 0; bot.dy0 = H - 1; bot.dx0 = 0; bot.wbase = (P + 1) * d->KW;
    Phase& lef = b.ph[2];        // columns 0 .. P-1 of the rows in between: kernel columns 0 .. P-1
    lef.tw = P; lef.ohg = H; lef.owg = P; lef.ooy = P; lef.oox = 0; lef.dy0 = P; lef.dx0 = 0; lef.wbase = 0;
    Phase& rig = b.ph[3];        // columns W+P .. W+2P-1: kernel columns P+1 .. 2P
    rig.tw = P; rig.ohg = H; rig.owg = P; rig.ooy = P; rig.oox = W + P; rig.dy0 = P; rig.dx0 = W - 1; rig.wbase = P + 1;
    for (int i = 0; i < 4; ++i) finish_phase(b.ph[i], d->B, b.Cs);
    b.in = dy; b.wt = wbwd; b.out = ws;
    b.in16 = d->y_bf16; b.in_ps = (unsigned)(d->y_pstride * 2);
    b.flops = 0.0;               // (the layer's algorithmic FLOPs are accounted once, by the pass over the image)
    const int rc = launch_gconv(b, st);
    if (rc) return rc;
  }
  GConvArgs a;
  memset(&a, 0, sizeof(a));
  geom_bwd_form(a, d->B, s.OH, s.OW, d->Cout, H, W, d->Cin, d->KH, d->KW, 1, P);
  a.in = dy; a.wt = wbwd; a.out = dx;
  a.in16 = d->y_bf16; a.out16 = d->x_bf16; a.aux16 = d->x_bf16 != 0;
  a.in_ps = (unsigned)(d->y_pstride * 2); a.out_ps = (unsigned)(d->x_pstride * 2);
  a.aux = prev_out; a.aux_act = prev_act; a.aux_slope = prev_slope; a.res = residual;
  a.flops = 2.0 * d->B * (double)s.OH * s.OW * d->Cout * d->Cin * d->KH * d->KW;
  const int rc = launch_gconv(a, st);
  if (rc) return rc;
  return reflect_ring_fix_launch(ws, dx, prev_out, prev_act, prev_slope, d->B, H, W, c4(d->Cin), P, d->x_bf16, (size_t)d->x_pstride, st);
}

static int conv_bwd_data_impl(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dx, float* ws,
                              const float* prev_out, int prev_act, float prev_slope, const float* pair_sigma0,
                              const float* pair_sigma1, float* stat_part, int* stat_rows, const float* residual,
                              void* stream, const BnAux* bn) {
  IPR_CHECK(!stat_part || (stat_rows && !fullmap_conv(d) && d->pad_mode != IPRGAN_PAD_REFLECT && c4(d->Cin) > 4),
            "conv_bwd_data: column sums need the row-count output and a zero-padded regular convolution with more than 4 input channels");
  IPR_CHECK(!pair_sigma0 == !pair_sigma1 &&
            (!pair_sigma0 || ((d->B & 1) == 0 && !fullmap_conv(d) && d->pad_mode != IPRGAN_PAD_REFLECT)),
            "conv_bwd_data: a paired pass needs both sigmas, an even batch and a zero-padded regular convolution");
  IPR_CHECK(d->stride >= 1 && d->stride <= 2, "conv_bwd_data: stride %d unsupported", d->stride);
  prof_tag("dgrad", d);
  GConvArgs a;
  memset(&a, 0, sizeof(a));
  const Shape s = out_shape(d);
  const bool reflect = d->pad_mode == IPRGAN_PAD_REFLECT;
  if (fullmap_conv(d) && (d->Cout % 32) == 0) {
    // dx[b][(tap, c)] = sum_n dy[b][n] W[n][c][tap]: operand row (tap, c) = row c of the backward operand at k offset tap*Cout
    const int ntap = d->KH * d->KW;
    geom_forward_form(a, d->B, 1, 1, d->Cout, 1, 1, ntap * d->Cin, 1, 1, 1, 0);
    a.Kp = rup(ntap * c4(d->Cout), 32);                 // row pitch of the backward operand
    a.wmod = d->Cin; a.wk1 = c4(d->Cout);
    a.in = dy; a.wt = wbwd; a.out = dx; a.act = IPRGAN_ACT_NONE;
    a.aux = prev_out; a.aux_act = prev_act; a.aux_slope = prev_slope; a.res = residual;
    IPR_CHECK(d->x_bf16 != 1 && d->y_bf16 != 1, "conv_bwd_data: bf16 activations on a full-map convolution are not built");
    a.in16 = d->y_bf16; a.in_ps = (unsigned)(d->y_pstride * 2);
    a.out16 = d->x_bf16; a.out_ps = (unsigned)(d->x_pstride * 2); a.aux16 = d->x_bf16 != 0;
    a.flops = 2.0 * d->B * (double)d->Cout * d->Cin * ntap;
    return launch_gconv(a, (hipStream_t)stream);
  }
  if (reflect && ws && !bn && reflect_direct_ok(d))
    return conv_bwd_data_reflect_direct(d, dy, wbwd, dx, ws, prev_out, prev_act, prev_slope, residual, (hipStream_t)stream);
  if (reflect) {
    // gradient w.r.t. the reflection-padded image (a zero-pad conv with pad 0 over H+2p), then fold
    IPR_CHECK(!d->transposed && ws, "conv_bwd_data: reflect pad needs a Conv2d and a workspace");
    geom_bwd_form(a, d->B, s.OH, s.OW, d->Cout, d->H + 2 * d->pad, d->W + 2 * d->pad, d->Cin, d->KH, d->KW,
                  d->stride, 0);
  } else if (!d->transposed) {
    geom_bwd_form(a, d->B, s.OH, s.OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad);
  } else {
    geom_forward_form(a, d->B, s.OH, s.OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad);
  }
  a.in = dy; a.wt = wbwd; a.bias = nullptr; a.out = reflect ? ws : dx;
  a.rs0 = pair_sigma0; a.rs1 = pair_sigma1;
  a.in16 = d->y_bf16; a.out16 = reflect ? 0 : d->x_bf16; a.aux16 = d->x_bf16 != 0;     // (a three-plane prev_out: its h plane)
  a.in_ps = (unsigned)(d->y_pstride * 2); a.out_ps = (unsigned)(d->x_pstride * 2);
  IPR_CHECK(!(reflect && (a.in16 == 1 || d->x_bf16 == 1)), "conv_bwd_data: bf16 activations with reflection padding are not built");
  a.flops = d->transposed ? 2.0 * d->B * (double)d->H * d->W * d->Cout * d->Cin * d->KH * d->KW : 2.0 * d->B * (double)s.OH * s.OW * d->Cout * d->Cin * d->KH * d->KW;
  a.act = IPRGAN_ACT_NONE; a.slope = 0.f;
  if (!reflect) { a.aux = prev_out; a.aux_act = prev_act; a.aux_slope = prev_slope; a.ws = ws; a.ws_floats = ws ? smalln_ws_for(d, false) : 0; }
  else { a.ws = ws + reflect_padded_floats(d); a.ws_floats = smalln_ws_for(d, false); }    // RGB stems: few-channel path
  a.stat_part = stat_part; a.stat_mode = 2;
  if (bn) {
    a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta;
    a.bn_act = bn->act; a.bn_slope = bn->slope; a.stat_mode = 3;
  }
  if (!reflect) a.res = residual;
  IPR_CHECK(!residual || reflect || !(smalln_eligible(a) && ws), "conv_bwd_data: no residual input on the few-channel path");
  if (!reflect && splitk_eligible(a, ws)) return gconv_splitk(a, ws, (hipStream_t)stream);
  const int rc = launch_gconv(a, (hipStream_t)stream);
  if (stat_part && !rc) *stat_rows = stat_rows_of(a);
  if (rc || !reflect) return rc;
  return reflect_fold_launch(ws, dx, prev_out, prev_act, prev_slope, residual, d->B, d->H, d->W, c4(d->Cin), d->pad,
                             d->x_bf16, (size_t)d->x_pstride, (hipStream_t)stream);
}

// ---- deferred slab reduce (iprgan_conv_bwd_weight_deferred / iprgan_wgrad_reduce_multi) ---------------------------------
static thread_local iprgan_wgrad_reduce_rec* t_wgrad_defer = nullptr;
static bool wgrad_reduce_deferred(const float* ws, float* dw, int nsplit, int Nrows, int Kw, int N, int C, int Qs, int ntap,
                                  long long sn, long long sc, float beta) {
  iprgan_wgrad_reduce_rec* r = t_wgrad_defer;
  if (!r) return false;
  r->ws = ws; r->dw = dw; r->nsplit = nsplit; r->Nrows = Nrows; r->Kw = Kw; r->N = N; r->C = C; r->Qs = Qs; r->ntap = ntap;
  r->sn = sn; r->sc = sc; r->beta = beta; r->pending = 1;
  return true;
}

static size_t wgrad_weight_floats(const iprgan_conv_desc* d) { return (size_t)d->Cout * d->Cin * d->KH * d->KW; }
size_t iprgan_conv_wgrad_ws_floats(const iprgan_conv_desc* d) {
  const Shape s = out_shape(d);
  // [slabs | bias partials] [reflect-padded copy of x] [scratch dw for the autotuner's trial launches]
  return rup4(wgrad_slab_floats(d) + colsum_ws_floats(d->B * s.OH * s.OW, c4(d->Cout))) + rup4(wgrad_padded_floats(d)) +
         wgrad_weight_floats(d);
}

int iprgan_conv_wgrad_takes_bf16(const iprgan_conv_desc* d) {
  // bf16 x and / or dy (desc flags) are read directly by every backward-weight kernel except the reflect-padded
  // swapped form (which copies x into a padded fp32 image first): there the caller hands an fp32 x (iprgan_cast)
  const WGeom g = wgrad_geom(d);
  if (wgrad_in3p(d)) {            // three planes, both tensors: a regular (not role-swapped) layer with a 128x128 split tile
    WGradPlan p;
    return !g.swap && (wgrad_plan_c(d, 0, p) || wgrad_plan_c(d, 3, p) || wgrad_x3h_ok(d)) ? 1 : 0;
  }
  if (wgrad_has_planes(d)) {      // one three-plane operand, the other fp32 (RGB stems / heads): widened on arrival by the
    WGradPlan p;                  // fp32 tiles; the reflect-padded swapped form pads an fp32 copy of x, joined on the way
    return d->x_bf16 != 1 && d->y_bf16 != 1 && (wgrad_plan_c(d, 0, p) || wgrad_plan_c(d, 1, p) || wgrad_plan_c(d, 2, p)) ? 1 : 0;
  }
  return !(g.padded && d->x_bf16);
}

int iprgan_conv_bwd_weight(const iprgan_conv_desc* d, const float* x, const float* dy, float* dw,
                           float* db, float* ws, float beta, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  // a deferred call (iprgan_conv_bwd_weight_deferred) owes the reduce of its FINAL launch only: the autotuner's trial
  // launches reduce into their scratch dw at once
  iprgan_wgrad_reduce_rec* const defer_rec = t_wgrad_defer;
  t_wgrad_defer = nullptr;
  IPR_CHECK(iprgan_conv_wgrad_takes_bf16(d), "conv_bwd_weight: this layer needs an fp32 x (iprgan_conv_wgrad_takes_bf16)");
  prof_tag("wgrad", d);
  const Shape s = out_shape(d);
  const WGeom g = wgrad_geom(d);
  const float* xin = x;
  if (g.padded) {             // reflect-padded copy of x behind the slabs and the bias partials
    float* xp = ws + rup4(wgrad_slab_floats(d) + colsum_ws_floats(d->B * s.OH * s.OW, c4(d->Cout)));
    const size_t n4 = wgrad_padded_floats(d) / 4;
    const size_t xps = d->x_bf16 == 2 ? (d->x_pstride ? (size_t)d->x_pstride : (size_t)d->B * d->H * d->W * c4(d->Cin)) : 0;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256)),
                       dim3(256), 0, st, (const f32x4*)x, (f32x4*)xp, d->B, d->H, d->W, c4(d->Cin) / 4, d->pad, xps);
    IPR_LAUNCH_CHECK();
    xin = xp;
  }
  // the autotuner's trial launches must not touch dw (with beta = 1 it holds other passes' gradients)
  float* dw_trial = ws + rup4(wgrad_slab_floats(d) + colsum_ws_floats(d->B * s.OH * s.OW, c4(d->Cout))) +
                    rup4(wgrad_padded_floats(d));
  auto run_to = [&](int cand, float* dw_out, float beta_out) -> int {
    if (cand >= WGRAD_NCAND) {           // halo / RGB forms: same slabs, same fixed-order reduce
      int nsplit = 0, Nrows = 0, Kw = 0, rc;
      if (cand >= WGRAD_X3H0) {
        const int i = cand - WGRAD_X3H0;
        if (i >= WGRAD_NX3H || !wgrad_x3h_ok(d) || g.swap || g.padded) return -1;
        if ((size_t)wgrad_x3h_nsplit(d, g_x3h_targets[i]) * c4(g.N) * d->KH * d->KW * c4(g.Cq) > WGRAD_MAX_SLAB_FLOATS) return -1;
        rc = launch_wgrad_x3h(d, x, dy, ws, g_x3h_targets[i], st, &nsplit, &Nrows, &Kw);
      } else
      if (cand >= WGRAD_NCAND + WGRAD_NHALO + WGRAD_NRGB) {
        const int i = cand - WGRAD_NCAND - WGRAD_NHALO - WGRAD_NRGB;
        if (i >= WGRAD_NH32 || !wgrad_h32_ok(d) || g.swap || g.padded) return -1;
        if ((size_t)wgrad_halo_f32_nsplit(d, g_h32_targets[i]) * c4(g.N) * d->KH * d->KW * c4(g.Cq) > WGRAD_MAX_SLAB_FLOATS) return -1;
        rc = launch_wgrad_halo_f32(d, x, dy, ws, g_h32_targets[i], st, &nsplit, &Nrows, &Kw);
      } else if (cand >= WGRAD_NCAND + WGRAD_NHALO) {
        if (!wgrad_rgb_ok(d)) return -1;
        rc = launch_wgrad_rgb(d, x, dy, ws, g_rgb_targets[cand - WGRAD_NCAND - WGRAD_NHALO], st, &nsplit, &Nrows, &Kw);
      } else {
        if (!wgrad_halo_ok(d)) return -1;
        rc = launch_wgrad_halo(d, x, dy, ws, (cand - WGRAD_NCAND) / 3, g_halo_targets[(cand - WGRAD_NCAND) % 3], st, &nsplit, &Nrows, &Kw);
      }
      if (rc) return rc;
      const int ntap = d->KH * d->KW, Qs = c4(g.Cq);
      const long long total = (long long)g.N * ntap * Qs;
      IPR_CHECK(total < (1ll << 31), "conv_bwd_weight: weight too large");
      if (wgrad_reduce_deferred(ws, dw_out, nsplit, Nrows, Kw, g.N, g.Cq, Qs, ntap, (long long)g.Cq * ntap, (long long)ntap, beta_out)) return 0;
      const int cw = wgrad_reduce_cw(dw_out, g.N, g.Cq, Qs, ntap, (long long)g.Cq * ntap, (long long)ntap);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(g.N, Qs, ntap, cw)), dim3(256), 0, st, ws, dw_out, nsplit,
                         Nrows, Kw, g.N, g.Cq, Qs, ntap, make_fastdiv(Qs), make_fastdiv(ntap * Qs),
                         (long long)g.Cq * ntap, (long long)ntap, beta_out, cw);
      IPR_LAUNCH_CHECK();
      return 0;
    }
    WGradPlan p;
    if (!wgrad_plan_c(d, cand, p)) return -1;
    WGradArgs a;
    memset(&a, 0, sizeof(a));
    a.P = (d->transposed || g.swap) ? xin : dy;
    a.Q = (d->transposed || g.swap) ? dy : xin;
    a.ws = ws;
    a.Ps = p.Ps; a.Pvalid = p.Ps;
    a.QH = g.QH; a.QW = g.QW;
    a.Qs = p.Qs; a.c4n = p.Qs / 4; a.d_c4n = make_fastdiv(a.c4n);
    a.PH = g.PH; a.PW = g.PW;
    a.M = p.M;
    a.d_pw = make_fastdiv(a.PW); a.d_plane = make_fastdiv(a.PH * a.PW); a.d_tw = make_fastdiv(d->KW);
    a.isy = a.isx = d->stride; a.pad = g.pad; a.tw = d->KW; a.ntap = p.ntap; a.flip = g.flip;
    a.pad_mode = g.swap ? IPRGAN_PAD_ZERO : d->pad_mode;      // swapped + reflect runs on the padded copy
    a.Kw = p.Kw; a.Nrows = p.Nrows; a.chunks_per_split = p.cps;
    const bool p_is_x = d->transposed || g.swap;
    a.p16 = p_is_x ? (g.padded ? 0 : d->x_bf16) : d->y_bf16;       // (the padded copy of x is fp32 whatever x is)
    a.q16 = p_is_x ? d->y_bf16 : d->x_bf16;
    a.in16 = wgrad_in3p(d) ? 2 : (wgrad_in16(d) && p.bn == 128 && p.bk == 128) ? 1 : 0;
    if (!a.in16 && (a.p16 == 2 || a.q16 == 2) && p.variant) return -1;        // widened on arrival by wgrad_kernel only
    const unsigned long long pesz = a.p16 ? 2ull : 4ull, esz = a.q16 ? 2ull : 4ull;
    unsigned long long pb = (unsigned long long)a.M * a.Ps * pesz;
    unsigned long long qb = (unsigned long long)d->B * a.QH * a.QW * a.Qs * esz;
    {
      const long long pps = p_is_x ? d->x_pstride : d->y_pstride, qps = p_is_x ? d->y_pstride : d->x_pstride;
      IPR_CHECK((a.p16 != 2 || pb < 0x2fffffffull) && (a.q16 != 2 || qb < 0x2fffffffull), "conv_bwd_weight: three-plane tensor larger than 2 GiB");
      if (a.p16 == 2) { a.p_ps = pps ? (unsigned)(pps * 2) : (unsigned)pb; pb += 2ull * a.p_ps; }
      if (a.q16 == 2) { a.q_ps = qps ? (unsigned)(qps * 2) : (unsigned)qb; qb += 2ull * a.q_ps; }
    }
    IPR_CHECK(pb < 0x7fffffffull && qb < 0x7fffffffull, "conv_bwd_weight: tensor larger than 2 GiB");
    a.p_bytes = (unsigned)pb; a.q_bytes = (unsigned)qb;
    IPR_CHECK(a.QH * (long long)a.QW < (1 << 24) && a.PH * (long long)a.PW < (1 << 24),
              "conv_bwd_weight: image larger than 2^24 pixels");
    {
      const int plane = a.PH * a.PW, db = 32 / plane, r = 32 % plane;
      a.dy32 = r / a.PW; a.dx32 = r % a.PW;
      const unsigned img = (unsigned)a.QH * a.QW * a.Qs * (unsigned)esz;
      a.bstep0 = (unsigned)db * img; a.bstep1 = (unsigned)(db + 1) * img;
    }
    a.flops = d->transposed ? 2.0 * d->B * (double)d->H * d->W * d->Cout * d->Cin * d->KH * d->KW
                            : 2.0 * d->B * (double)s.OH * s.OW * d->Cout * d->Cin * d->KH * d->KW;
    int rc;
    a.xcd = p.variant == 2;
    if (p.variant) {
      if (p.w8) rc = launch_wgrad_t2<2, 4, 2, 1>(a, p, st);
      else if (p.bn == 128 && p.bk == 128) rc = launch_wgrad_t2<2, 2, 2, 2>(a, p, st);
      else if (p.bn == 64) rc = launch_wgrad_t2<2, 2, 1, 1>(a, p, st);
      else rc = launch_wgrad_t2<2, 2, 2, 1>(a, p, st);
    } else
    if (p.w8) rc = launch_wgrad_t<2, 4, 2, 1>(a, p, st);        // 128x128, 8 waves of 64x32
    else if (p.bn == 128 && p.bk == 128) rc = launch_wgrad_t<2, 2, 2, 2>(a, p, st);
    else if (p.bn == 64) rc = launch_wgrad_t<2, 2, 1, 1>(a, p, st);
    else if (p.bn == 128) rc = launch_wgrad_t<2, 2, 2, 1>(a, p, st);
    else rc = launch_wgrad_t<1, 4, 1, 1>(a, p, st);     // bn == 32, bk == 128
    if (rc) return rc;
    const long long total = (long long)p.N * p.ntap * p.Qs;
    IPR_CHECK(total < (1ll << 31), "conv_bwd_weight: weight too large");
    // dw[row * sn + qchannel * sc + tap]: Conv2d [Cout][Cin][taps], ConvTranspose2d [Cin][Cout][taps];
    // swapped roles: rows are Cin and the Q channel is Cout of a Conv2d weight
    const long long sn = g.swap ? p.ntap : (long long)p.Cq * p.ntap, sc = g.swap ? (long long)p.N * p.ntap : p.ntap;
    if (wgrad_reduce_deferred(ws, dw_out, p.nsplit, p.Nrows, p.Kw, p.N, p.Cq, p.Qs, p.ntap, sn, sc, beta_out)) return 0;
    const int cw = wgrad_reduce_cw(dw_out, p.N, p.Cq, p.Qs, p.ntap, sn, sc);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(p.N, p.Qs, p.ntap, cw)), dim3(256), 0, st, ws, dw_out,
                       p.nsplit, p.Nrows, p.Kw, p.N, p.Cq, p.Qs, p.ntap, make_fastdiv(p.Qs),
                       make_fastdiv(p.ntap * p.Qs), sn, sc, beta_out, cw);
    IPR_LAUNCH_CHECK();
    return 0;
  };
  auto run = [&](int cand) -> int { return run_to(cand, dw_trial, 0.f); };
  int cand = 0;
  {
    WGradPlan p0;
    if (!wgrad_plan_c(d, 0, p0)) cand = !wgrad_in3p(d) ? 1 : wgrad_plan_c(d, 3, p0) ? 3 : WGRAD_X3H0;
  }
  if (g_force_wgrad >= 0) {
    WGradPlan pf;
    if (g_force_wgrad >= WGRAD_X3H0 ? (g_force_wgrad < WGRAD_NALL && wgrad_x3h_ok(d) && !g.swap && !g.padded)
        : g_force_wgrad >= WGRAD_NCAND + WGRAD_NHALO + WGRAD_NRGB ? (wgrad_h32_ok(d) && !g.swap && !g.padded)
        : g_force_wgrad >= WGRAD_NCAND + WGRAD_NHALO ? wgrad_rgb_ok(d)
        : g_force_wgrad >= WGRAD_NCAND ? wgrad_halo_ok(d) : wgrad_plan_c(d, g_force_wgrad, pf))
      cand = g_force_wgrad;
  } else if (g_autotune) {     // same scheme as the forward/backward-data tiles: time every candidate once per geometry
    TuneKey key = {{d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->outpad,
                    d->transposed, d->pad_mode, -7, g_math, d->x_bf16, d->y_bf16}};
    int cached;
    if (tune_lookup(key, &cached)) {
      cand = cached;
    } else {
      const bool prof_was = g_prof_on;
      g_prof_on = false;
      float best_us = 0.f;
      int err = 0;
      cand = tune_pick(WGRAD_NALL, run, st, cand, &best_us, &err);
      g_prof_on = prof_was;
      if (err) return err;
      if (getenv("IPRGAN_TUNE_LOG"))
        fprintf(stderr, "[iprgan tune] wgrad B%d %dx%d %d->%d k%d s%d t%d -> cand %d (%.1f us)\n", d->B, d->H, d->W,
                d->Cin, d->Cout, d->KH, d->stride, d->transposed, cand, best_us);
      tune_store(key, cand);
    }
  }
  {
    t_wgrad_defer = defer_rec;
    const int rc = run_to(cand, dw, beta);
    t_wgrad_defer = nullptr;
    IPR_CHECK(rc != -1, "conv_bwd_weight: candidate %d does not apply to this layer (storage kinds x %d / dy %d)", cand, d->x_bf16, d->y_bf16);
    if (rc) return rc;
  }
  if (db) {
    const int Cs = c4(d->Cout), M = d->B * s.OH * s.OW;
    float* part = ws + wgrad_slab_floats(d);
    const int rc2 = colsum_launch(dy, db, part, M, Cs, d->Cout, st, beta, d->y_bf16, (size_t)d->y_pstride);
    if (rc2) return rc2;
  }
  return 0;
}

int iprgan_conv_bwd_weight_deferred(const iprgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, float* ws,
                                    float beta, void* stream, iprgan_wgrad_reduce_rec* rec) {
  IPR_CHECK(rec, "conv_bwd_weight_deferred: null record");
  memset(rec, 0, sizeof(*rec));
  t_wgrad_defer = rec;
  const int rc = iprgan_conv_bwd_weight(d, x, dy, dw, db, ws, beta, stream);
  t_wgrad_defer = nullptr;
  return rc;
}

int iprgan_wgrad_reduce_multi(const iprgan_wgrad_reduce_rec* recs, int n, void* stream) {
  IPR_CHECK(n >= 0 && (recs || !n), "wgrad_reduce_multi: bad arguments");
  int i = 0;
  while (i < n) {                             // (tables of WGRAD_MULTI_MAX entries: one launch each)
    WGradReduceTable t;
    memset(&t, 0, sizeof(t));
    unsigned blocks = 0;
    for (; i < n && t.n < WGRAD_MULTI_MAX; ++i) {
      const iprgan_wgrad_reduce_rec& r = recs[i];
      if (!r.pending) continue;
      const long long total = (long long)r.N * r.ntap * r.Qs;
      IPR_CHECK(r.ws && r.dw && total > 0 && total < (1ll << 31), "wgrad_reduce_multi: bad record %d", i);
      const int e = t.n++;
      t.ws[e] = r.ws; t.dw[e] = r.dw; t.sn[e] = r.sn; t.sc[e] = r.sc;
      t.d_qs[e] = make_fastdiv(r.Qs); t.d_row[e] = make_fastdiv(r.ntap * r.Qs);
      t.nsplit[e] = r.nsplit; t.Nrows[e] = r.Nrows; t.Kw[e] = r.Kw; t.N[e] = r.N; t.C[e] = r.C; t.Qs[e] = r.Qs; t.ntap[e] = r.ntap;
      t.beta[e] = r.beta;
      t.cw[e] = wgrad_reduce_cw(r.dw, r.N, r.C, r.Qs, r.ntap, r.sn, r.sc);
      t.first[e] = blocks;
      blocks += wgrad_reduce_blocks(r.N, r.Qs, r.ntap, t.cw[e]);
      t.first[e + 1] = blocks;
    }
    if (!t.n) continue;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
    IPR_LAUNCH_CHECK();
  }
  return 0;
}

int iprgan_set_math_mode(int mode) {
  IPR_CHECK(mode == IPRGAN_MATH_FP32 || mode == IPRGAN_MATH_BF16 || mode == IPRGAN_MATH_FP32X3, "set_math_mode: unknown mode %d", mode);
  g_math = mode;
  return 0;
}
int iprgan_get_math_mode(void) { return g_math; }

int iprgan_debug_force_tiles(int gconv_tile, int wgrad_cand) {
  g_force_tile = gconv_tile;
  g_force_wgrad = wgrad_cand;
  return 0;
}

int iprgan_debug_force_splitk(int splits) {
  g_force_splitk = splits;
  return 0;
}

// The autotuner's table as flat records of IPRGAN_TUNE_RECORD_INTS ints (16 key ints + the choice), so that the ranks of a
// data-parallel job can adopt ONE rank's choices (iprgan/parallel.py: sync_autotune): every replica then runs the same tiles
// - the same summation orders - whatever its own timings said.
int iprgan_tune_export(int* records, size_t cap_records, size_t* count) {
  IPR_CHECK(count, "tune_export: null count");
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  tune_load();
  *count = g_tune.size();
  if (!records) return 0;
  size_t i = 0;
  for (auto it = g_tune.begin(); it != g_tune.end() && i < cap_records; ++it, ++i) {
    memcpy(records + i * IPRGAN_TUNE_RECORD_INTS, it->first.v, sizeof(it->first.v));
    records[i * IPRGAN_TUNE_RECORD_INTS + 16] = it->second;
  }
  *count = i;           // the records actually WRITTEN (another host thread may have tuned a geometry since the sizing call)
  return 0;
}
int iprgan_tune_import(const int* records, size_t n_records, int replace) {
  IPR_CHECK(records || !n_records, "tune_import: null records");
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  tune_load();
  if (replace) g_tune.clear();
  for (size_t i = 0; i < n_records; ++i) {
    TuneKey k;
    memcpy(k.v, records + i * IPRGAN_TUNE_RECORD_INTS, sizeof(k.v));
    g_tune[k] = records[i * IPRGAN_TUNE_RECORD_INTS + 16];
  }
  // the adopted table is what this process runs from now on: with a cache file configured it is rewritten, so that a restarted
  // rank reloads the COMMON picks instead of its own earlier ones (and does not re-tune into a divergent table)
  const std::string path = tune_path();
  FILE* f = path.empty() ? nullptr : fopen(path.c_str(), "w");
  if (f) {
    for (auto& kv : g_tune) {
      for (int j = 0; j < 16; ++j) fprintf(f, "%d ", kv.first.v[j]);
      fprintf(f, "%d\n", kv.second);
    }
    fclose(f);
  }
  return 0;
}

int iprgan_prof_enable(int on) {
  if (on) {
    for (int i = 0; i < g_nslots; ++i) { g_slots[i].launches = 0; g_slots[i].ms = 0; g_slots[i].flops = 0; }
    for (auto& t : g_tag_agg) t = LayerAgg{0, 0, 0};
  }
  g_prof_on = on != 0;
  return 0;
}
int iprgan_prof_collect(void) {
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_slots[r.slot].launches += 1; g_slots[r.slot].ms += ms; g_slots[r.slot].flops += r.flops;
      if (r.tag >= 0 && r.tag < (int)g_tag_agg.size()) {
        g_tag_agg[r.tag].launches += 1; g_tag_agg[r.tag].ms += ms; g_tag_agg[r.tag].flops += r.flops;
      }
    }
    g_free_events.push_back(r.a); g_free_events.push_back(r.b);
  }
  g_recs.clear();
  return 0;
}
int iprgan_prof_num_kernels(void) { return g_nslots; }
int iprgan_prof_num_layers(void) { return (int)g_tag_names.size(); }
int iprgan_prof_get_layer(int i, char* name, int name_len, long long* launches, double* ms, double* flops) {
  IPR_CHECK(i >= 0 && i < (int)g_tag_names.size(), "prof_get_layer: bad index %d", i);
  snprintf(name, name_len, "%s", g_tag_names[i].c_str());
  *launches = g_tag_agg[i].launches; *ms = g_tag_agg[i].ms; *flops = g_tag_agg[i].flops;
  return 0;
}
int iprgan_prof_get(int i, char* name, int name_len, long long* launches, double* ms, double* flops) {
  IPR_CHECK(i >= 0 && i < g_nslots, "prof_get: bad index %d", i);
  snprintf(name, name_len, "%s", g_slots[i].name);
  *launches = g_slots[i].launches; *ms = g_slots[i].ms; *flops = g_slots[i].flops;
  return 0;
}

int iprgan_act_bwd(const float* dy, const float* out, float* dz, size_t n, int act, float slope, int act_bf16,
                   void* stream) {
  if (n == 0) return 0;
  const int blocks = (int)(cdivz(n, 256) < 8192 ? cdivz(n, 256) : 8192);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, out, dz, n,
                     act, slope, act_bf16);
  IPR_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
