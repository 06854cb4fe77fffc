/* libiprgan_hip.so — C ABI of the MI355X (gfx950) kernels behind ipr-gan's G+D training step.
 *
 * Drop-in boundary.  The reference has no native code: its
 * "kernel layer" is torch ATen/cuDNN reached through torch.nn modules
 * (networks/<net>.py) and torch.optim / torch.nn.functional (models/<model>.py).  Each entry point
 * below replaces the ATen op named in its comment at the cited reference call site.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory unless noted
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); no allocation.  No synchronisation either,
 *     with ONE exception: the convolution entry points autotune their tile (the reference trains with
 *     cudnn.benchmark = True, train.py:44-45) - the FIRST call with a new layer geometry times the candidate tiles
 *     on the caller's stream and blocks the calling host thread on HIP events until they have run (a few ms per
 *     geometry, once per process; outputs are written normally).  Every later call with that geometry only enqueues.
 *     IPRGAN_AUTOTUNE=0 in the environment disables it (a block-count heuristic picks the tile, nothing ever blocks);
 *     IPRGAN_TUNE_CACHE=<file> stores the choices and replays them in later processes (per rank: <file>.<RANK>).
 *   - limits: every tensor handed to a convolution entry point must be smaller than 2 GiB (the kernels address them
 *     through 32-bit buffer offsets, which is also how out-of-image taps are zero-filled without branches) and an
 *     image smaller than 2^24 pixels; larger inputs are refused with an error, never truncated
 *   - process-global state: the conv tile table (mutex-guarded), the math mode, the per-kernel timer and the RCCL
 *     communicator.  Everything else is re-entrant; concurrent calls from several host threads are safe as long as
 *     they use different streams and do not flip the math mode under each other
 *   - return 0 on success, non-zero on error; text via iprgan_last_error() (thread-local)
 *   - activations are fp32 NHWC with the channel count padded to a multiple of 4
 *     ("C4" = (C+3)&~3; RGB images are NHWC4 with a zero 4th channel)
 *   - "prepared" conv weights are tap-major K-vectors, produced by iprgan_conv_weight_prep
 */
#ifndef IPRGAN_H
#define IPRGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPRGAN_VERSION 226

enum { IPRGAN_ACT_NONE = 0, IPRGAN_ACT_RELU = 1, IPRGAN_ACT_LRELU = 2, IPRGAN_ACT_TANH = 3,
       IPRGAN_ACT_SIGMOID_PM1 = 4 };   /* sigmoid(x)*2-1: nn.Sigmoid + Decoder32.Normalize (networks/decoder.py:14-16,31-32) */
enum { IPRGAN_PAD_ZERO = 0, IPRGAN_PAD_REFLECT = 1 };

/* One 2-D convolution / transposed convolution layer.
 * Conv2d:          x[B,H,W,Cin]  -> y[B,OH,OW,Cout], OH=(H+2*pad-KH)/stride+1
 * ConvTranspose2d: x[B,H,W,Cin]  -> y[B,OH,OW,Cout], OH=(H-1)*stride-2*pad+KH+outpad */
typedef struct {
  int32_t B, H, W, Cin, Cout;
  int32_t KH, KW, stride, pad, outpad;
  int32_t transposed;     /* 0 = Conv2d, 1 = ConvTranspose2d */
  int32_t pad_mode;       /* IPRGAN_PAD_*: reflect = ReflectionPad2d(pad) folded into the gather */
  int32_t act;            /* IPRGAN_ACT_* fused into the forward epilogue */
  float   slope;          /* LeakyReLU negative slope */
  int32_t x_bf16;         /* storage kind (IPRGAN_ST_*) of the layer's INPUT activation x (and of dx, prev_out, residual) */
  int32_t y_bf16;         /* storage kind of the layer's OUTPUT activation y (and of dy) */
  int64_t x_pstride;      /* IPRGAN_ST_X3 only: distance between the planes of x / y in ELEMENTS; 0 = the tensor is     */
  int64_t y_pstride;      /*   contiguous (B*H*W*C4).  A batch slice of a larger tensor keeps the larger tensor's stride. */
} iprgan_conv_desc;
/* Storage kinds of an activation tensor.  F32: fp32.  BF16: see below.  X3 ("three planes", math mode IPRGAN_MATH_FP32X3,
 * padded channel count % 32 == 0): the tensor is stored as THREE bf16 tensors of its shape, plane-major - h = bf16(x),
 * m = bf16(x - h), l = bf16(x - h - m), all round-to-nearest-even; x = h + (m + l) EXACTLY (3 x 8 mantissa bits cover
 * the 24 of fp32; the subtractions are exact), so this is an fp32 tensor in 6 bytes per element whose first plane alone is
 * its bf16 rounding (same sign: activation masks read only that plane).  The pointer handed to an entry point is plane h.
 * Producers split ONCE per element; the convolution kernels then move the planes to LDS by DMA and multiply plane pairs
 * on the bf16 matrix pipe (csrc/conv_x3.hip) with no vector-ALU work in their K loops. */
/* RANGE of the three-plane storage (tests/test_gpu_x3.py: "the ends of fp32's exponent range"; measured on gfx950):
 *   - 2^-110 <= |x| <= 0x7F7F0000 (3.3895e38, the largest finite bf16): the round trip fp32 -> planes -> fp32 is BIT-EXACT;
 *   - |x| < 2^-110: the planes run out of exponent range before fp32 does.  The stored value is x rounded to the bf16 subnormal
 *     grid (absolute error <= 2^-133): 24 significant bits at 2^-110, 8 at 2^-126 (only h is left); signs are kept, nothing
 *     becomes non-finite.  fp32 subnormals flush the same way;
 *   - NaN stays NaN.  +-inf and finite |x| > 0x7F7F7FFF (h rounds to inf, the residual planes are inf - inf) come back as NaN:
 *     a non-finite or over-range value NEVER turns into a finite one, so a diverged run trips the same finiteness checks as
 *     in the fp32 mode - but an inf is not preserved as an inf;
 *   - products: a convolution whose operands are in the exact range is as close to the float64 result as the fp32 MFMA's over
 *     the whole range the PRODUCT survives in (operands at 2^-110 x 2^100, 2^100 x 2^-80, ...: test_conv_at_the_ends_of_the_
 *     exponent_range), also when the leading h x h' terms cancel and the result is made of the small terms alone; an output whose
 *     receptive field contains a non-finite input is non-finite in both modes (NaN here where the fp32 mode gives +-inf). */
enum { IPRGAN_ST_F32 = 0, IPRGAN_ST_BF16 = 1, IPRGAN_ST_X3 = 2,
       /* the `act_bf16` argument of the norm entry points (iprgan_bn_*, iprgan_instnorm_*, iprgan_bn_prelu_*) only: the layer's
        * INPUT x is fp32 while y, dy, dx and the residual are three-plane tensors - the convolution in front of a norm layer
        * then writes 4 instead of 6 bytes per element (y_bf16 = IPRGAN_ST_F32 in its descriptor) and x is read three times */
       IPRGAN_ST_X3_XF32 = 3,
       /* the BACKWARD norm entry points (iprgan_bn_bwd, iprgan_bn_prelu_bwd, iprgan_instnorm_bwd) only: x AND dy are fp32, dx (and y) are
        * three-plane tensors - the backward-data pass that produced dy wrote 4 instead of 6 bytes per element as well
        * (x_bf16 = IPRGAN_ST_F32 in its descriptor), and dy is read twice (reduction, apply) */
       IPRGAN_ST_X3_XDF32 = 4 };
/* bf16 activations ("bf16 in HBM", BASELINE config 5): only with IPRGAN_MATH_BF16 and only for tensors whose padded
 * channel count is a multiple of 64; such a tensor is bf16 for EVERY entry point that touches it (element offsets and
 * shapes are unchanged, the `float*` in the signatures is then a bf16 buffer).  Prepared weights follow the operand
 * they are multiplied with: iprgan_conv_weight_prep emits bf16 operands for a bf16 x (forward) / bf16 y (backward).
 * Weight gradients, statistics, losses, master weights and Adam stay fp32. */

const char* iprgan_last_error(void);
int iprgan_version(void);

/* ---- layout ------------------------------------------------------------------------ */
/* NCHW [B,C,H,W] -> NHWC [B,H,W,C4] (pad channels zero) and back (drops pad channels).   */
int iprgan_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream);
int iprgan_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream);
/* dst[b][a][k] = src[a][b][k] (NCHW-flatten <-> NHWC-flatten order of Linear weights,
 * networks/conv_generator.py:26 and sn_discriminator.py:32) */
/* beta: 0 overwrites dst, 1 accumulates (dst = beta*dst + permuted src) - see "gradient accumulation" below */
int iprgan_permute_021(const float* src, float* dst, int A, int Bd, int K, float beta, void* stream);

/* ---- Linear(K -> C*HW) feeding an NHWC map (replaces aten::addmm + view + relu and their backward for the first layer of
 *      the generators: networks/conv_generator.py:26-30 `self.fc(z).view(-1, C, mg, mg)`, the VAE decoder's first layer).
 * x [B][K] fp32; w [C*HW][K] and bias [C*HW] fp32 in PyTorch's row order c*HW + hw (the parameters themselves: no permuted
 * copy, no prepared operand); y / dy [B][HW][C] in storage kind `kind` (IPRGAN_ST_*, plane stride in elements, 0 = B*HW*C).
 *   fwd:  y[b][hw][c] = act(sum_k x[b][k] w[c*HW + hw][k] + bias[c*HW + hw])
 *   bwd:  dz = dy * act'(y);  dw[c*HW + hw][k] = beta*dw + sum_b dz[b][hw][c] x[b][k];  db likewise (db may be NULL)
 * One launch each, exact fp32 MFMA in every math mode, fixed summation order.  Shapes: K % 32 == 0, K <= 128, C % 64 == 0
 * (iprgan_fc_nhwc_ok); other shapes go through the convolution family as a 1x1 layer.  dx is not produced here (the
 * generators' latent needs none; a caller that does uses iprgan_conv_bwd_data on the 1x1 form). */
int iprgan_fc_nhwc_ok(int B, int K, int C, int HW);
int iprgan_fc_nhwc_fwd(const float* x, const float* w, const float* bias, void* y, int B, int K, int C, int HW, int act,
                       float slope, int y_kind, size_t y_pstride, void* stream);
int iprgan_fc_nhwc_bwd(const float* x, const void* y, const void* dy, float* dw, float* db, int B, int K, int C, int HW,
                       int act, float slope, int kind, size_t y_pstride, size_t dy_pstride, float beta, void* stream);

/* ---- convolution (replaces aten::conv2d / conv_transpose2d + their backward;
 *      networks/sn_discriminator.py:9-18, conv_generator.py:8,21, sr_resnet.py:22,
 *      discriminator_96.py:7-21, resnet_generator.py:7-34, conv_discriminator.py:6-21) ---- */
/* sizes (in floats) of the two prepared-weight buffers and of the wgrad workspace */
size_t iprgan_conv_wfwd_floats(const iprgan_conv_desc* d);
size_t iprgan_conv_wbwd_floats(const iprgan_conv_desc* d);
size_t iprgan_conv_wgrad_ws_floats(const iprgan_conv_desc* d);
/* w: PyTorch layout (Conv2d [Cout,Cin,KH,KW]; ConvTranspose2d [Cin,Cout,KH,KW]).
 * inv_scale: optional device scalar; weights are multiplied by 1/(*inv_scale) (spectral-norm sigma).
 * wfwd / wbwd (either may be NULL): tap-major operands for forward and for backward-data. */
int iprgan_conv_weight_prep(const iprgan_conv_desc* d, const float* w, const float* inv_scale,
                            float* wfwd, float* wbwd, void* stream);
/* The same for n layers in one launch: descs[n] (only Cin/Cout/KH/KW/transposed are read); w / inv_scale /
 * wfwd / wbwd are HOST arrays of n DEVICE pointers (entries, or whole arrays, may be NULL). */
int iprgan_conv_weight_prep_multi(const iprgan_conv_desc* descs, const float* const* w,
                                  const float* const* inv_scale, float* const* wfwd, float* const* wbwd, int n,
                                  void* stream);
/* y = act(conv(x, w) + bias); bias may be NULL (length Cout).  ws: optional workspace of
 * iprgan_conv_fwd_ws_floats(d) floats (non-zero only for layers with <= 4 output channels, which then run as
 * one dense 1x1 GEMM over tap planes + a gather instead of a 3/32-full MFMA tile); NULL = generic kernel. */
size_t iprgan_conv_fwd_ws_floats(const iprgan_conv_desc* d);
int iprgan_conv_fwd(const iprgan_conv_desc* d, const float* x, const float* wfwd, const float* bias,
                    float* y, float* ws, const float* pair_sigma0, const float* pair_sigma1, float* stat_part,
                    int* stat_rows, void* stream);
/* Column statistics from the epilogue (stat_part non-NULL, iprgan_conv_stat_floats(d, 0) floats; *stat_rows, HOST int,
 * receives the number of partial rows written): every output tile also stores the column sums s1 = sum acc and
 * s2 = sum acc^2 of its rows, acc = the accumulator BEFORE bias and activation, as stat_part[row][2][C4(Cout)]; rows of
 * one sample (and of one sub-pixel phase) are contiguous when its pixel count is a multiple of 128.  A BatchNorm /
 * InstanceNorm that follows the convolution takes them instead of reading the activation once more for its statistics
 * (iprgan_bn_fwd / iprgan_instnorm_fwd: conv_part, conv_part_rows, conv_bias).  iprgan_conv_bwd_data has the same
 * pair of arguments: there the sums are over the STORED values (s1 = column sums of dx after the fused activation
 * derivative = the bias gradient of the layer below, combined with iprgan_colsum_partials). */
size_t iprgan_conv_stat_floats(const iprgan_conv_desc* d, int backward);
/* dx = conv_bwd_data(dy, w) [* act'(x_out_prev)]: if prev_out != NULL the result is multiplied by the
 * derivative of activation prev_act evaluated from the saved OUTPUT prev_out of the previous layer
 * (same shape as dx), i.e. the previous layer's activation backward is fused into this epilogue. */
size_t iprgan_conv_bwd_data_ws_floats(const iprgan_conv_desc* d);   /* reflect padding, or <= 4 input channels */
int iprgan_conv_bwd_data(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dx, float* ws,
                         const float* prev_out, int prev_act, float prev_slope, const float* pair_sigma0,
                         const float* pair_sigma1, float* stat_part, int* stat_rows, const float* residual,
                         void* stream);
/* Backward-data INTO a BatchNorm (networks/conv_generator.py:8-10, discriminator_96.py:27-31: conv -> BatchNorm -> ReLU /
 * LeakyReLU -> THIS conv): the result is the gradient w.r.t. the norm layer's OUTPUT, and the norm backward's two
 * reductions over (x, dy) can be taken in this epilogue.  bn_x = the norm layer's INPUT (same shape and storage type as
 * the result), bn_mean / bn_invstd = its saved statistics, bn_gamma / bn_beta its affine parameters (both or neither),
 * bn_act / bn_slope the activation behind it (none, ReLU, LeakyReLU).  Stored: dz = backward-data(dy, w) * act'(v),
 * v = (x - mean) * invstd * gamma + beta (the forward's own expression); stat_part[row][2][C4(Cin)] receives per tile
 * s1 = sum dz and s2 = sum dz * (x - mean) * invstd.  iprgan_bn_bwd_pre finishes the norm backward from dz and these
 * rows: three tensor passes instead of five, one launch less.  iprgan_conv_bwd_data_bn_ok: 1 if the layer qualifies
 * (zero-padded, not a full-map convolution, more than 32 input channels). */
int iprgan_conv_bwd_data_bn_ok(const iprgan_conv_desc* d);
int iprgan_conv_bwd_data_bn(const iprgan_conv_desc* d, const float* dy, const float* wbwd, float* dz, const float* bn_x,
                            const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                            int bn_act, float bn_slope, float* stat_part, int* stat_rows, void* stream);
/* residual (optional, same shape as dx): dx = backward-data(...) * act'(prev_out) + residual - the gradient that arrives
 * over a skip connection at the input of a residual block (networks/sr_resnet.py:37-38, resnet_generator.py:52-53) is
 * added in the epilogue instead of by a separate pass. */
/* Paired pass (pair_sigma0/1 non-NULL, device scalars; B even): the batch holds TWO half-batches that the reference
 * sends through a spectrally normalised network one after the other (models/dcgan.py:47-48: D(real), D(fake)) - between
 * them the power iteration advances, so the halves see W/sigma0 and W/sigma1.  The operands are prepared from the
 * un-normalised W (inv_scale NULL) and the rows of the first / second half are divided by sigma0 / sigma1 in the
 * epilogue (before the bias; conv is linear in W).  One launch of twice the size fills the GPU better than two. */
/* out[c] = beta*out[c] + sum_m x[m][c], c < C, over x[M][Cs] (bias gradients); ws: iprgan_colsum_ws_floats(M, Cs) floats */
size_t iprgan_colsum_ws_floats(int M, int Cs);
int iprgan_colsum(const float* x, float* out, float* ws, int M, int Cs, int C, float beta, int x_bf16, void* stream);
/* the same from per-tile partials part[rows][2][Cs] (first of the two sums) written by a convolution epilogue */
int iprgan_colsum_partials(const float* part, int rows, int Cs, int C, float* out, float beta, void* stream);
/* the same for n layers in one launch (HOST arrays of device pointers / sizes): the bias gradients a backward pass owes,
 * flushed together by the executor; bit-identical to n single calls */
int iprgan_colsum_partials_multi(const float* const* parts, const int* rows, const int* Cs, const int* C, float* const* outs,
                                 const float* betas, int n, void* stream);
/* dw (PyTorch layout) = beta*dw + conv_bwd_weight(x, dy); db (optional, length Cout) = beta*db + sum dy.
 * beta = 0 overwrites; beta = 1 accumulates straight into a gradient bucket (what autograd's AccumulateGrad
 * add plus DDP's bucket copy do in two extra passes).  ws: workspace of iprgan_conv_wgrad_ws_floats(d) floats.
 * Deterministic (fixed-order split reduce). */
int iprgan_conv_bwd_weight(const iprgan_conv_desc* d, const float* x, const float* dy, float* dw,
                           float* db, float* ws, float beta, void* stream);
/* dz = dy * act'(out) elementwise (activation backward from the saved output), n floats. */
int iprgan_act_bwd(const float* dy, const float* out, float* dz, size_t n, int act, float slope, int act_bf16,
                   void* stream);

/* ---- GEMV head: SN-Linear 512*md*md -> 1 (networks/sn_discriminator.py:21) -------------- */
/* y[b] = dot(x[b,:], w)/(*inv_scale) + bias[0];  x [B,K] */
/* x_bf16: storage kind of x (and dx, prev_out); x_pstride / dx_pstride: plane strides in elements of three-plane x
 * (= prev_out) / dx, 0 = contiguous (B*K) - a half-batch of a paired pass keeps the whole batch's stride */
int iprgan_gemv_fwd(const float* x, const float* w, const float* bias, const float* inv_scale, float* y,
                    int B, int K, int x_bf16, size_t x_pstride, void* stream);
/* dx[b,k] = dy[b]*w[k]/(*inv_scale) [* act'(prev_out)];  dw[k] = sum_b dy[b]*x[b,k] (gradient w.r.t.
 * the NORMALISED weight w/sigma; feed it to iprgan_sn_bwd);  db[0] = sum dy.  dx/dw/db may be NULL. */
int iprgan_gemv_bwd(const float* x, const float* w, const float* dy, const float* inv_scale, float* dx,
                    float* dw, float* db, const float* prev_out, int prev_act, float prev_slope, int B,
                    int K, int x_bf16, size_t x_pstride, size_t dx_pstride, void* stream);
/* Paired pass (two half-batches of B / 2 rows, each with its own sigma: DESIGN.md section 4): the two iprgan_gemv_fwd / _bwd
 * calls of the halves as one launch per kernel.  dw2 [2][K] and db2 [2] receive the gradients of the first / second half
 * (three-plane x only; x_pstride as for the whole tensor).  Bit-identical to the per-half calls. */
int iprgan_gemv_fwd_pair(const float* x, const float* w, const float* bias, const float* inv_scale0, const float* inv_scale1,
                         float* y, int B, int K, int x_bf16, size_t x_pstride, void* stream);
int iprgan_gemv_bwd_pair(const float* x, const float* w, const float* dy, const float* inv_scale0, const float* inv_scale1,
                         float* dx, float* dw2, float* db2, const float* prev_out, int prev_act, float prev_slope, int B, int K,
                         int x_bf16, size_t x_pstride, size_t dx_pstride, void* stream);

/* ---- BatchNorm2d (networks/conv_generator.py:9, sr_resnet.py:23, discriminator_96.py:31) -- */
size_t iprgan_bn_ws_floats(int M, int C);
/* training forward: batch statistics over M=B*H*W rows of x[M,C]; y = act((x-mean)*invstd*g+b);
 * save_mean/save_invstd [C] out; running stats updated in place (momentum, unbiased var) when
 * running_mean != NULL.  eval forward: use_running=1 normalises with the running stats.
 * conv_part / conv_part_rows / conv_bias: column sums emitted by the convolution that produced x (iprgan_conv_fwd:
 * stat_part), taken before its bias (conv_bias, may be NULL) was added - the statistics then cost no pass over x.
 * num_batches_tracked (int64 device scalar, may be NULL) is incremented by the statistics kernel in training mode. */
int iprgan_bn_fwd(const float* x, float* y, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                  float* ws, int M, int C, float eps, float momentum, int use_running, int act,
                  float slope, const float* conv_part, int conv_part_rows, const float* conv_bias,
                  long long* num_batches_tracked, const float* residual, int act_bf16, void* stream);
/* residual (optional, same shape as y): y = act(norm(x)) + residual, the skip connection closing a residual block. */
/* backward through act + BN: inputs x (pre-norm), dy, and y (post-act output; NOT read for no activation and for
 * ReLU / LeakyReLU, whose derivative mask is recomputed from x with gamma / beta - may be NULL then).  dx, dgamma, dbeta
 * out.  dbias_prev (optional, dbias_n floats): dbias_prev = dbias_beta*dbias_prev + column sums of dx, i.e. the bias
 * gradient of the convolution that produced x, accumulated on the apply pass itself. */
int iprgan_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                  const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                  float* dbeta, float* ws, int M, int C, int act, float slope, float* dbias_prev, int dbias_n,
                  float dbias_beta, int act_bf16, void* stream);
/* BatchNorm2d + PReLU with one learnable slope (networks/sr_resnet.py:7,13: conv -> BatchNorm -> PReLU) in the norm layer's
 * own passes: `slope` is the PReLU parameter ON THE DEVICE (one float); forward = iprgan_bn_fwd with the activation
 * v > 0 ? v : slope * v; backward also returns dslope = sum dy * min(v, 0) (v = the normalised, affine value) - no
 * separate PReLU forward / backward passes over the tensor. */
int iprgan_bn_prelu_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* save_mean, float* save_invstd, float* ws, int M, int C, float eps,
                        float momentum, int use_running, const float* slope, const float* conv_part, int conv_part_rows,
                        const float* conv_bias, long long* num_batches_tracked, const float* residual, int act_bf16,
                        void* stream);
int iprgan_bn_prelu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, const float* slope, float* dx, float* dgamma, float* dbeta,
                        float* dslope, float* ws, int M, int C, float* dbias_prev, int dbias_n, float dbias_beta,
                        int act_bf16, void* stream);
/* the norm backward behind iprgan_conv_bwd_data_bn: dz and its per-tile sums part[rows][2][C] come from that pass
 * (the buffer holds iprgan_conv_stat_floats(d, 1) floats: the rows are compacted in place when there are many) */
int iprgan_bn_bwd_pre(const float* x, const float* dz, const float* gamma, const float* save_mean,
                      const float* save_invstd, const float* part, int rows, float* dx, float* dgamma, float* dbeta,
                      float* ws, int M, int C, float* dbias_prev, int dbias_n, float dbias_beta, int act_bf16,
                      void* stream);

/* ---- InstanceNorm2d (networks/resnet_generator.py:8-49 affine, conv_discriminator.py:10-18 plain):
 * per-(sample, channel) statistics over HW rows of x[B,HW,C]; gamma/beta may be NULL; never tracks
 * running stats (same in eval).  save_mean/save_invstd are [B,C]. */
size_t iprgan_instnorm_ws_floats(int B, int HW, int C);
int iprgan_instnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* save_mean,
                        float* save_invstd, float* ws, int B, int HW, int C, float eps, int act, float slope,
                        const float* conv_part, int conv_part_rows, const float* conv_bias, const float* residual,
                        int act_bf16, void* stream);
int iprgan_instnorm_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                        float* dbeta, float* ws, int B, int HW, int C, int act, float slope, float* dbias_prev,
                        int dbias_n, float dbias_beta, int act_bf16, void* stream);

/* ---- PReLU / PixelShuffle / MaxPool / residual add / reflection-pad backward ----------------------------
 * nn.PReLU() with one slope (sr_resnet.py:7,14,43): y = x>0 ? x : alpha*x; dalpha = sum dy*x*[x<=0];
 * ws >= iprgan_loss_ws_floats(n) floats.  PixelShuffle(2) (sr_resnet.py:42) on NHWC: src [B,H,W,4C] ->
 * dst [B,2H,2W,C] (inverse=1: the other way, i.e. its gradient).  MaxPool2d(2,2) (VGG19 features,
 * vgg.py:33): gradient to the first maximum of each window.  reflect_fold: gradient of
 * ReflectionPad2d(pad) - dxp [B,H+2p,W+2p,C] summed back onto dx [B,H,W,C] (times act'(prev_out) if given). */
/* act_st (here and below): storage kind of the activation tensors of the call (IPRGAN_ST_F32, or IPRGAN_ST_X3 for
 * contiguous three-plane tensors: plane stride = the tensor's element count) */
int iprgan_prelu_fwd(const float* x, const float* alpha, float* y, size_t n, int act_st, void* stream);
int iprgan_prelu_bwd(const float* x, const float* dy, const float* alpha, float* dx, float* dalpha, float* ws,
                     size_t n, int act_st, void* stream);
int iprgan_pixel_shuffle2(const float* src, float* dst, int B, int H, int W, int C, int inverse, void* stream);
/* conv -> PixelShuffle(2) -> PReLU (the upsampling blocks, sr_resnet.py:39-45) in one pass each way: x [B,H,W,4C] is the
 * convolution's output, y / dy [B,2H,2W,C]; y = prelu(shuffle(x)), dx = unshuffle(dy) * prelu'(x), dalpha as above
 * (ws >= iprgan_loss_ws_floats(B*H*W*C) floats).  C % 4 == 0. */
int iprgan_pixel_shuffle2_prelu_fwd(const float* x, const float* alpha, float* y, int B, int H, int W, int C, int act_st,
                                    void* stream);
int iprgan_pixel_shuffle2_prelu_bwd(const float* x, const float* dy, const float* alpha, float* dx, float* dalpha, float* ws,
                                    int B, int H, int W, int C, int act_st, void* stream);
int iprgan_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, int act_st, void* stream);
int iprgan_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int act_st, void* stream);
int iprgan_add(const float* a, const float* b, float* out, size_t n, int act_st, void* stream);
int iprgan_reflect_fold(const float* dxp, float* dx, const float* prev_out, int prev_act, float prev_slope,
                        const float* residual, int B, int H, int W, int C, int pad, void* stream);
/* ImagePool's swap branch (models/util.py:27-34: `pool_images = self.images[index[prob]].clone(); self.images[index[prob]] =
 * images[prob]; images[prob] = pool_images`) with the draws in device memory, so that a captured training step replays
 * with new ones: image i of images[count][n] changes places with row index[i] of pool[..][n] when take[i] != 0.  The
 * entries of index must be distinct (the reference takes a prefix of a permutation); fp32, any layout, n elements per
 * image; both operands are updated in place. */
/* n integers from HOST memory `values` into device memory `dst`, carried as kernel arguments: they are read before the
 * call returns and land in stream order (the per-step draws of iprgan_pool_swap: no pinned buffer to keep alive). */
int iprgan_write_ints(int* dst, const int* values, int n, void* stream);
int iprgan_pool_swap(float* images, float* pool, const int* index, const int* take, int count, size_t n, void* stream);

/* ---- spectral norm (torch.nn.utils.spectral_norm at networks/sn_discriminator.py:9,11,18,21) */
size_t iprgan_sn_ws_floats(int rows, int cols);
/* one power iteration on W_mat[rows,cols] (row-major = weight.view(Cout,-1)): updates u[rows], v[cols]
 * in place (skipped when training==0) and writes sigma = u.(W v) to *sigma (device). eps = 1e-12. */
int iprgan_sn_power_iter(const float* w, float* u, float* v, float* sigma, float* ws, int rows,
                         int cols, float eps, int training, void* stream);
/* The same for n layers at once (HOST arrays of DEVICE pointers / sizes, n <= 16): sigma[l] out; u_out/v_out
 * (optional) receive this pass's u, v copies that the backward needs. 4 launches for a whole network. */
size_t iprgan_sn_multi_ws_floats(const int* rows, const int* cols, int n);
int iprgan_sn_power_iter_multi(const float* const* w, float* const* u, float* const* v, float* const* u_out,
                               float* const* v_out, float* sigma, float* ws, const int* rows, const int* cols,
                               int n, float eps, int training, void* stream);
/* dW_orig = (dW_sn - (sum dW_sn*W)/sigma * u v^T) / sigma   (autograd through sigma, u,v constant) */
int iprgan_sn_bwd(const float* dwsn, const float* w, const float* u, const float* v,
                  const float* sigma, float* dw, float* ws, int rows, int cols, void* stream);
/* n layers at once (HOST arrays of DEVICE pointers; sigma[l] points at that layer's device scalar);
 * ws >= 64*16 floats. */
int iprgan_sn_bwd_multi(const float* const* dwsn, const float* const* w, const float* const* u,
                        const float* const* v, const float* const* sigma, float* const* dw, float* ws,
                        const int* rows, const int* cols, int n, float beta, void* stream);   /* dw = beta*dw + ... */

/* ---- losses (models/dcgan.py:33-40, srgan.py:36-59, cyclegan.py:122-142) ------------------ */
enum { IPRGAN_LOSS_HINGE_REAL = 0,   /* mean(relu(1-x)) */
       IPRGAN_LOSS_HINGE_FAKE = 1,   /* mean(relu(1+x)) */
       IPRGAN_LOSS_NEG_MEAN = 2,     /* -mean(x) */
       IPRGAN_LOSS_BCE_ONES = 3,     /* BCE-with-logits vs 1 */
       IPRGAN_LOSS_BCE_ZEROS = 4,    /* BCE-with-logits vs 0 */
       IPRGAN_LOSS_MSE_ONES = 5, IPRGAN_LOSS_MSE_ZEROS = 6,
       IPRGAN_LOSS_MSE = 7, IPRGAN_LOSS_L1 = 8,    /* two-input forms use y */
       /* VAE terms (models/vae.py:36-48), used with iprgan_loss_sum_* (sum * scale, scale = 1/batch): */
       IPRGAN_LOSS_BCE_PM1 = 9,      /* F.binary_cross_entropy((x+1)/2, (y+1)/2): logs clamped at -100 like ATen */
       IPRGAN_LOSS_KL_MEAN = 10,     /* x^2 / 2                      (x = mean)   */
       IPRGAN_LOSS_KL_LOGVAR = 11,   /* (exp(x) - 1 - x) / 2         (x = logvar) */
       IPRGAN_LOSS_MSE_DENORM = 12,  /* tools/loss.py:15-18 normalized=True: MSE of ((x+1)/2, (y+1)/2) */
       IPRGAN_LOSS_L1_DENORM = 13 }; /*                                    : L1  of ((x+1)/2, (y+1)/2) */
size_t iprgan_loss_ws_floats(size_t n);
int iprgan_loss_fwd(int kind, const float* x, const float* y, float* loss, float* ws, size_t n,
                    void* stream);
/* dx = (*gscale) * dloss/dx ; gscale is a device scalar (upstream gradient), may be NULL (=1). */
int iprgan_loss_bwd(int kind, const float* x, const float* y, const float* gscale, float* dx,
                    size_t n, void* stream);
/* Two one-input mean losses over the two halves of one vector x[0 .. n_half) / x[n_half .. 2 n_half) and their sum, one launch each
 * way (the paired discriminator pass: models/dcgan.py:33-37 LossR, LossF, LossD = LossR + LossF).  out3 = {loss_a, loss_b,
 * loss_a + loss_b}; bwd: dx[2 n_half] = gscale * d(loss_a + loss_b)/dx.  n_half <= 256; bit-identical to two
 * iprgan_loss_fwd / _bwd calls and an fp32 add. */
int iprgan_loss_pair_fwd(int kind_a, int kind_b, const float* x, size_t n_half, float* out3, void* stream);
int iprgan_loss_pair_bwd(int kind_a, int kind_b, const float* x, const float* gscale, float* dx, size_t n_half, void* stream);
/* same kernels with an explicit reduction: loss = scale * sum_i term_i (reduction='sum' / N, models/vae.py:41-47) */
int iprgan_loss_sum_fwd(int kind, const float* x, const float* y, float* loss, float* ws, size_t n,
                        float scale, void* stream);
int iprgan_loss_sum_bwd(int kind, const float* x, const float* y, const float* gscale, float* dx,
                        size_t n, float scale, void* stream);

/* ---- VAE reparameterisation (networks/encoder.py:24-28): z = eps * exp(logvar/2) + mean ---- */
int iprgan_reparam_fwd(const float* mean, const float* logvar, const float* eps, float* z, size_t n,
                       void* stream);
/* dmean = dz ; dlogvar = dz * eps * exp(logvar/2) / 2 */
int iprgan_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmean,
                       float* dlogvar, size_t n, void* stream);

/* ---- SSIM loss of the black-box objective (tools/loss.py:82-85 -> third-party pytorch-msssim 0.2.1 `ssim`,
 * restated: 11-tap Gaussian sigma 1.5 valid window per channel, C1 = 0.01^2, C2 = 0.03^2, data_range 1).
 * x, y: `planes` = B*C contiguous H x W images (NCHW tensors); denorm != 0 applies (v+1)/2 to both first
 * (tools/loss.py:15-18).  loss = 1 - mean SSIM.  gmaps (iprgan_ssim_gmap_floats, may be NULL when no gradient
 * is needed) receives the per-window sensitivities the backward pass consumes; ws: iprgan_ssim_ws_floats.
 * Only x receives a gradient (the wrapper detaches y, models/wrappers.py:50-52). */
size_t iprgan_ssim_ws_floats(int planes, int H, int W);
size_t iprgan_ssim_gmap_floats(int planes, int H, int W);
int iprgan_ssim_fwd(const float* x, const float* y, float* loss, float* gmaps, float* ws, int planes, int H,
                    int W, int denorm, void* stream);
int iprgan_ssim_bwd(const float* x, const float* y, const float* gmaps, const float* gscale, float* dx,
                    int planes, int H, int W, int denorm, void* stream);

/* ---- MS-SSIM loss: 1 - MS_SSIM(data_range=1)(x, y) (tools/loss.py:78-80; pytorch-msssim 0.2.1 restated: five scales,
 * 2x2 average pooling with zero padding on odd sizes, cs at scales 1-4 and ssim at scale 5, weights 0.0448 0.2856 0.3001
 * 0.2363 0.1333, per (image, channel) product of relu(mean)^w, mean over planes).  x, y: NCHW planes [planes][H][W]
 * with min(H, W) > 160.  iprgan_msssim_sizes gives the three buffer sizes (floats); the forward pass fills them, the
 * backward pass (gradient w.r.t. x only) reads them and needs ws = 2 * planes * ceil(H/2) * ceil(W/2) floats more. */
int iprgan_msssim_sizes(int planes, int H, int W, size_t* pyr_floats, size_t* gmap_floats, size_t* small_floats);
int iprgan_msssim_fwd(const float* x, const float* y, float* loss, float* pyr, float* gmaps, float* small, int planes,
                      int H, int W, int denorm, int want_grad, void* stream);
int iprgan_msssim_bwd(const float* x, const float* y, const float* pyr, const float* gmaps, const float* small,
                      const float* gscale, float* dx, float* ws, int planes, int H, int W, int denorm, void* stream);

/* ---- sign-loss watermark (tools/sign_model.py:42-60) --------------------------------------- */
/* gammas/signs/dgammas: HOST arrays of nlayer DEVICE pointers, sizes: HOST array of channel counts.
 * loss = sum_l mean(relu(gamma0 - gamma_l*sign_l)).  Pointer tables are copied into the launch. */
int iprgan_sign_loss_fwd(const float* const* gammas, const float* const* signs, const int* sizes,
                         int nlayer, float gamma0, float* loss, void* stream);
/* dgamma_l = beta*dgamma_l + (*gscale) * (-sign/n_l) * [gamma0 - gamma*sign > 0] */
int iprgan_sign_loss_bwd(const float* const* gammas, const float* const* signs,
                         float* const* dgammas, const int* sizes, int nlayer, float gamma0,
                         const float* gscale, float beta, void* stream);
/* counts[0] = #(sign(gamma) != sign_bit) (sign(0)=0 counts as an error), counts[1] = total bits;
 * int64 device output, bit-exact. */
int iprgan_sign_ber(const float* const* gammas, const float* const* signs, const int* sizes,
                    int nlayer, long long* counts, void* stream);

/* ---- Adam (torch.optim.Adam.step at models/dcgan.py:69,78) --------------------------------- */
/* multi-tensor: HOST arrays of n DEVICE pointers; step is the 1-based step count after increment.
 * Hyper-parameters are doubles: 1-beta and the bias corrections are formed in double like torch does.
 * grad_scale multiplies every gradient as it is read: 1/world_size turns the all-reduced SUM of the data-parallel
 * gradients into their mean without a separate pass over the buckets (1.0 on a single GPU: exact). */
int iprgan_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const long long* sizes, int n, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);
/* the same step with the step COUNT on the device: *step_dev (int, device) is incremented by the call and the bias
 * corrections are computed from it on the device into coef (two floats, device) - the form a captured HIP graph needs,
 * whose kernel arguments are frozen at capture time (iprgan/graphs.py) */
int iprgan_adam_step_dev(float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const long long* sizes, int n, double lr, double beta1,
                         double beta2, double eps, double weight_decay, int* step_dev, float* coef, double grad_scale,
                         void* stream);

/* ---- measurement (bench.py roofline): when enabled, every conv-family launch is bracketed by HIP
 * events on its own stream; collect() waits for them and accumulates per-kernel launch count, device
 * milliseconds and algorithmic FLOPs (2*B*OH*OW*Cout*Cin*KH*KW per forward / dgrad / wgrad launch). */
int iprgan_prof_enable(int on);
int iprgan_prof_collect(void);
int iprgan_prof_num_kernels(void);
int iprgan_prof_get(int i, char* name, int name_len, long long* launches, double* ms, double* flops);
/* the same records grouped by layer: name = pass (fwd / dgrad / wgrad) + descriptor geometry */
int iprgan_prof_num_layers(void);
int iprgan_prof_get_layer(int i, char* name, int name_len, long long* launches, double* ms, double* flops);

/* ---- math mode of the conv family (process-wide).  FP32 (default): v_mfma_f32_32x32x2_f32 on fp32 tiles.
 * BF16 (BASELINE config "DCGAN 128x128 bs256 bf16"): tensors and master weights stay fp32 in HBM, tiles are
 * rounded to bf16 (nearest-even) when staged into LDS and multiplied by v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation; layers with fewer than 32 (padded) input channels keep the fp32 kernel.  Norms, losses, spectral
 * norm and Adam are fp32 in both modes.
 * FP32X3: fp32 tensors, fp32-grade products on the bf16 matrix pipe: layers with (padded) input channels % 32 == 0 split
 * every operand element into three bf16 terms while staging it into LDS (x = h + m + l, exact) and accumulate a product
 * block from six bf16 MFMAs (l h' + h l' + m m' + m h' + h m' + h h', fp32 accumulator); the dropped terms are below
 * 2^-26 of a product, the distance to a float64 convolution is that of the fp32 MFMA or smaller (tests/test_gpu_x3.py).
 * All other layers and kernels run as in FP32. */
enum { IPRGAN_MATH_FP32 = 0, IPRGAN_MATH_BF16 = 1, IPRGAN_MATH_FP32X3 = 2 };
int iprgan_set_math_mode(int mode);
int iprgan_get_math_mode(void);

/* test hook: force one tile configuration so that the parity tests can exercise every variant, not only the one the
 * autotuner picks; -1 = autotune / heuristic.  A candidate that does not apply to a geometry falls back to the default.
 *   gconv_tile 0..7   register-staged tiles of conv_igemm.hip (6, 7: bf16 modes only)
 *              8..13  LDS-DMA ring tiles of conv_pipe.hip: 256x128, 256x64, 256x256, 128x128, 256x64 (two blocks per
 *                     CU), 128x64; bf16 operands in HBM, or fp32 operands with the exact fp32 MFMA
 *              18..25 three-plane ring tiles of conv_x3.hip (fp32x3 mode, three-plane operands): 256x128, 128x128,
 *                     128x64 (3 / 2 stages), 256x64, 64x64, 128x256, 128x128 (2 stages)
 *              26, 27 their halo form for stride-1 gathers (k3 s1; the 2x2-tap sub-pixel phases of k4 s2): 256 positions x
 *                     128 / 64 columns, the tile's halo staged once per 16-channel chunk, taps as row shifts
 *              28..31 tiles 18, 19, 20, 22 on v_mfma_f32_16x16x32_bf16 (half the accumulator rows per instruction; measured
 *                     on par, kept as autotune candidates)
 *              32..34 the three-plane ring with DEDICATED LOADER WAVES (round 5): 256x64, 128x128, 128x64 (two blocks per CU) -
 *                     four waves multiply, four waves only issue the stage refills; 35..37 the same on v_mfma_f32_16x16x32_bf16
 *              14, 15 the persistent 256x128 / 256x64 form (bf16 operands)
 *              16     four sub-pixel phases per block (k4 s2 p1 backward-data forms, bf16 operands)
 *              17     256x256 with a half-tile ring: quadrant phases, five half-tiles of DMA in flight (bf16 operands)
 *   wgrad_cand 0..59  = 20 * variant + 4 * block target + tile shape (split-M GEMM of conv_igemm.hip)
 *              60..68 halo form for bf16 tensors (wgrad_halo.hip): 3 * variant + block target {128, 256, 512}
 *              69, 70 RGB-layer streaming form (block targets 256 / 512)
 *              71..73 halo form on the exact fp32 MFMA for fp32 tensors (block targets 256 / 512 / 1024)
 *              74..76 halo form for three-plane tensors, k3 s1 / k4 s2 (wgrad_x3.hip; block targets 256 / 512 / 1024) */
int iprgan_debug_force_tiles(int gconv_tile, int wgrad_cand);
/* test hook: force the split count (1..4) of the split-K path that convolutions with few output tiles take when their
 * workspace is passed (iprgan_conv_fwd_ws_floats / iprgan_conv_bwd_data_ws_floats); -1 = autotuned. */
int iprgan_debug_force_splitk(int splits);
/* The autotuner's table (process-global, mutex-guarded) as flat records of IPRGAN_TUNE_RECORD_INTS ints each (16 ints of
 * geometry key + the chosen tile / candidate), HOST memory.  export: *count = records in the table; up to cap_records are
 * written when `records` is not NULL.  import: inserts / overwrites (replace != 0: the table is cleared first).  The ranks of
 * a data-parallel job adopt rank 0's table after the first step (iprgan/parallel.py: sync_autotune), so that every replica
 * runs the same tiles; the reference's counterpart is cudnn.benchmark = True deciding per process (train.py:44-45). */
/* export: records == NULL -> *count = table size; otherwise up to cap_records records are written and *count = the number
 * written.  import: with IPRGAN_TUNE_CACHE set, the (per-rank) cache file is rewritten with the table as adopted. */
#define IPRGAN_TUNE_RECORD_INTS 17
int iprgan_tune_export(int* records, size_t cap_records, size_t* count);
int iprgan_tune_import(const int* records, size_t n_records, int replace);

/* ---- misc elementwise ----------------------------------------------------------------------- */
int iprgan_fill(float* p, float v, size_t n, void* stream);
/* dst[i] = src[i] for n elements between the fp32 and bf16 storage types (round-to-nearest-even) */
int iprgan_cast(const float* src, float* dst, size_t n, int src_bf16, int dst_bf16, void* stream);
/* fp32 <-> three planes (IPRGAN_ST_X3; exact both ways): to_planes != 0 splits n fp32 elements of src into planes
 * dst + p * pstride (bf16 elements, p = 0, 1, 2); otherwise src is the h plane of a three-plane tensor with that plane
 * stride and dst receives h + (m + l).  pstride >= n (a batch slice of a larger tensor keeps the larger stride). */
int iprgan_cast_planes(const void* src, void* dst, size_t n, size_t pstride, int to_planes, void* stream);
/* 1 if iprgan_conv_bwd_weight consumes x / dy of this layer in the storage kinds the descriptor names (bf16; three planes on
 * both sides; or ONE three-plane operand next to an fp32 one - the 64-channel side of an RGB stem / head, summed h + (m + l)
 * as it is loaded), 0 if it wants fp32 copies */
int iprgan_conv_wgrad_takes_bf16(const iprgan_conv_desc* d);
/* Backward-weight with its slab reduce DEFERRED.  iprgan_conv_bwd_weight runs two kernels: the tiles, which leave partial
 * weight gradients in slabs of `ws`, and a fixed-order reduce + scatter into dw (PyTorch layout; dw = beta * dw + sum).  The
 * deferred form enqueues the tiles (and the bias gradient, if asked for) and fills *rec (HOST memory) with what the reduce needs
 * instead of launching it; iprgan_wgrad_reduce_multi then runs the reduces of up to 24 records per launch - one launch for all
 * layers of a backward pass instead of one per layer (19 per DCGAN-64 step), bit-identical to the undeferred results (every block
 * performs the same additions in the same order).  The caller keeps `ws` and `dw` of every pending record alive and unmodified
 * until the multi-reduce has been enqueued on the SAME stream; rec->pending == 0 after the call means nothing is owed (the layer
 * took a form without slabs).  Autotuning trial launches of a first call reduce at once into their scratch area. */
typedef struct {
  const float* ws; float* dw;
  int32_t nsplit, Nrows, Kw, N, C, Qs, ntap, pending;
  int64_t sn, sc;
  float beta;
} iprgan_wgrad_reduce_rec;
int iprgan_conv_bwd_weight_deferred(const iprgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, float* ws,
                                    float beta, void* stream, iprgan_wgrad_reduce_rec* rec);
int iprgan_wgrad_reduce_multi(const iprgan_wgrad_reduce_rec* recs, int n, void* stream);
int iprgan_axpy(float* y, const float* x, float a, size_t n, void* stream);   /* y += a*x */
/* y_t += a*x_t for n tensors in one launch (HOST arrays of DEVICE pointers / element counts): the small
 * gradients of a pass (biases, norm scales, PReLU slopes) into their gradient-bucket views */
int iprgan_axpy_multi(float* const* y, const float* const* x, const long long* sizes, int n, float a, void* stream);

/* ---- data-parallel gradient exchange: RCCL all-reduce over xGMI ---------------
 * Replaces torch.nn.DataParallel's per-forward replicate/scatter/gather (models/dcgan.py:16-17, srgan.py:17-19,
 * cyclegan.py:19-23) by one in-place SUM per gradient bucket and optimizer step; one process per GPU.
 *   rank 0:     iprgan_comm_unique_id(id)      (128 bytes, HOST memory; ship it to the other ranks out of band)
 *   every rank: iprgan_comm_init(rank, nranks, id)   binds the CURRENT HIP device; collective, blocks until all joined
 *   per bucket: iprgan_allreduce_bucket(buf, n, IPRGAN_DTYPE_F32, side_stream)   enqueued, returns at once; the
 *               caller orders it against its compute stream with events (record after the bucket's last producer,
 *               wait before the optimizer reads the bucket)
 *   shutdown:   iprgan_comm_destroy()
 * RCCL is bound at run time (the copy already loaded in the process, else ROCm's; IPRGAN_RCCL_LIB overrides).
 * The communicator is the only process-global handle of this group. */
#define IPRGAN_COMM_ID_BYTES 128
enum { IPRGAN_DTYPE_F32 = 0, IPRGAN_DTYPE_BF16 = 1 };
int iprgan_comm_unique_id(void* id128);
int iprgan_comm_init(int rank, int nranks, const void* id128);
int iprgan_allreduce_bucket(void* buf, size_t n, int dtype, void* stream);
int iprgan_comm_nranks(void);          /* 0 when no communicator exists */
int iprgan_comm_rank(void);            /* -1 when no communicator exists */
int iprgan_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* IPRGAN_H */
