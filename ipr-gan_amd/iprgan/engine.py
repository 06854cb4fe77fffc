"""Chain executor: a network is a list of layer ops whose forward/backward enqueue HIP kernels.

One ``torch.autograd.Function`` (``ChainFn``) runs a whole network so PyTorch's autograd sees a
single node per G / D pass (it is used for the graph only: which pass feeds which loss, and
``.grad`` accumulation); everything inside is ours: NHWC activations, fused bias/activation
epilogues, the previous layer's activation derivative fused into the next dgrad epilogue,
spectral-norm power iteration, BatchNorm statistics, deterministic split reductions.

Parameters and buffers stay ordinary tensors owned by the nn.Module tree (state_dict / .to() /
optimizers keep working, SURVEY.md section 8b "State/ownership").
"""
import torch

import os

from . import _lib as L
from . import ops, parallel

_BATCH_PREP = os.environ.get('IPRGAN_BATCH_PREP', '1') != '0'
_FUSE_STATS = os.environ.get('IPRGAN_FUSE_STATS', '1') != '0'       # A/B switch: column sums from conv epilogues
# BatchNorm-backward sums from the consumer's dgrad epilogue (iprgan_conv_bwd_data_bn / iprgan_bn_bwd_pre): built, parity-
# tested, and OFF by default - it removes the reduction pass over (x, dy) (norm-backward traffic 1.67x -> 1.0x of the
# algorithmic bytes) but puts an operand read and the mask arithmetic into epilogues that are already the slow part of the
# short-K layers, and keeps the 256x256 / four-phase / persistent tiles out of those passes: DCGAN-128 bf16act 17.15 ->
# 17.68 ms, SRGAN 33.30 -> 33.53 ms, DCGAN-64 11.52 -> 11.53 ms (round 3, same box, back to back).
_FUSE_BN_BWD = os.environ.get('IPRGAN_FUSE_BN_BWD', '0') != '0'
_GEMV_PAIR = os.environ.get('IPRGAN_GEMV_PAIR', '1') != '0'        # A/B switch: the head of a paired pass in one launch per kernel


class Op:
    """One layer.  ``params`` lists the nn.Parameters it reads (order = gradient order)."""
    params = ()
    out_act = (L.ACT_NONE, 0.0)     # activation fused in this op's forward (derivative from output)
    fuses_prev_act = False          # can multiply its dx by the producer's act'(x) in the epilogue

    def forward(self, x, st, train):
        raise NotImplementedError

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        """dy: gradient w.r.t. this op's output (pre-activation if ``st['dy_is_preact']``).
        prev_act: (act, slope) of the producer to fuse into dx, or None.  Returns (dx, [param grads]).
        sink: the GradReducer that owns this pass's gradient buckets, or None.  With a sink, large gradients are
        accumulated straight into ``sink.view_of(param)`` and reported as ``DIRECT``; small ones are still returned
        as tensors and added to their views in one multi-tensor launch by the executor."""
        raise NotImplementedError


DIRECT = object()     # marker: the gradient has already been accumulated into its bucket view


class ToNHWC(Op):
    """API boundary: NCHW image -> NHWC4 (networks take/return NCHW like the reference)."""

    def __init__(self, channels):
        self.channels = channels

    def forward(self, x, st, train):
        return ops.nchw_to_nhwc(x)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        return (ops.nhwc_to_nchw(dy, self.channels) if need_dx else None), []


class ToNCHW(Op):
    def __init__(self, channels):
        self.channels = channels

    def forward(self, x, st, train):
        return ops.nhwc_to_nchw(x, self.channels)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        return (ops.nchw_to_nhwc(dy) if need_dx else None), []


class View(Op):
    """Reinterpret a flat [B, H*W*C] activation as NHWC [B,H,W,C] (no data movement)."""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.fuses_prev_act = False

    def forward(self, x, st, train):
        st['in_shape'] = tuple(x.shape)
        return x.view(x.shape[0], *self.shape)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        return dy.view(st['in_shape']), []


class Conv(Op):
    """Conv2d / ConvTranspose2d (+bias, +activation), optionally spectrally normalised
    (weight_orig / weight_u / weight_v of torch.nn.utils.spectral_norm)."""
    fuses_prev_act = True

    def __init__(self, spec, module, sn=False):
        self.spec, self.m, self.is_sn = spec, module, sn
        self.out_act = (spec.act, spec.slope)

    # tensors are fetched from the module at call time: .to()/load_state_dict may replace buffers.  Straight from the module's
    # parameter / buffer dicts: nn.Module.__getattr__ (the slow path every `m.weight` takes) was ~550 calls = 0.4 ms of host
    # time per eager DCGAN step
    @property
    def weight(self):
        return self.m._parameters['weight_orig' if self.is_sn else 'weight']

    @property
    def bias(self):
        return self.m._parameters.get('bias')

    @property
    def sn(self):
        b = self.m._buffers
        return (b['weight_u'], b['weight_v']) if self.is_sn else None

    @property
    def params(self):
        w, b = self.weight, self.bias
        return (w,) if b is None else (w, b)

    # prepared operands ('wf' forward, 'wb' backward-data) of layers WITHOUT spectral norm, keyed on the weight's
    # version counter and storage (optimizer steps, load_state_dict and .to() all change the key)
    def _operand_key(self):
        w = self.weight
        return (w._version, w.data_ptr(), L.get_math_cached())

    def cached_operand(self, which, st):
        if self.is_sn or st.get('sigma') is not None:
            return False
        ent = self.__dict__.get('_op_' + which)
        if ent is not None and ent[0] == self._operand_key():
            st[which] = ent[1]
            return True
        return False

    def keep_operand(self, which, tensor, st):
        if not self.is_sn and st.get('sigma') is None:
            self.__dict__['_op_' + which] = (self._operand_key(), tensor)

    def forward(self, x, st, train):
        sp = self.spec
        B, H, W, _ = x.shape
        # y_f32 (set by the chain): the consumer is a norm layer - in 'fp32x3' mode the output leaves as fp32 (4 instead of 6
        # bytes per element) and the norm layer, which reads it three times, is where the tensor is split (ops._norm_fwd_kinds)
        d = sp.desc(B, H, W, y16=ops.ST_F32 if st.get('y_f32') else None)
        sigma = st.get('sigma')             # set by Chain's batched spectral-norm pre-pass
        if self.sn is not None and sigma is None and st.get('pair') is None:
            u, v = self.sn
            sigma = ops.sn_power_iter(self.weight, u, v, train)
            st['u'], st['v'] = u.clone(), v.clone()     # this pass's u, v (later passes overwrite the buffers)
        wf = st.pop('wf', None)             # set by Chain's batched weight-prep pre-pass
        pair = st.get('pair')               # paired pass: (sigma of the first half-batch, of the second); W un-normalised
        if pair is not None:
            sigma = None
        if wf is None:
            wf, _ = ops.conv_prep(sp, d, self.weight, sigma, fwd=True, bwd=False)
        if st.get('emit_stats'):            # the norm layer that follows takes its statistics from this epilogue
            y, stats = ops.conv_fwd(sp, d, x, wf, self.bias, pair=pair, stats=True)
            st['ctx']['conv_stats'] = (stats, self.bias)
        else:
            y = ops.conv_fwd(sp, d, x, wf, self.bias, pair=pair)
        st.update(x=x, y=y, d=d, sigma=sigma)
        return y

    def can_emit_stats(self, H, W, per_instance):
        """Column statistics from the epilogue: any BatchNorm; InstanceNorm when every sample's rows (per sub-pixel
        phase) form whole tiles (include/iprgan.h)."""
        sp = self.spec
        if ops.c4(sp.cout) <= 4 or (sp.k == H and sp.k == W and sp.pad == 0 and not sp.transposed and sp.k > 1):
            return False
        if not per_instance:
            return True
        OH, OW = sp.out_hw(H, W)
        s_ = sp.stride if sp.transposed else 1
        # tile rows must never straddle two samples: the autotuner may pick a 256-row tile in every math mode (the LDS-DMA
        # ring tiles are candidates for fp32 layers too)
        rows = 256
        return OH % s_ == 0 and OW % s_ == 0 and ((OH // s_) * (OW // s_)) % rows == 0

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        sp, d, sigma = self.spec, st['d'], st['sigma']
        if d.y_bf16 == ops.ST_F32 and ops.is16(dy) == ops.ST_X3:    # (y_f32: the forward output left as fp32, its gradient
            d = ops._desc_with(d, y_bf16=ops.ST_X3)                  # arrives from the norm layer as a three-plane tensor)
        if sp.act != L.ACT_NONE and not st.get('dy_is_preact', False):
            dy = ops.act_bwd(dy, st['y'], sp.act, sp.slope)
        grads = []
        pair = st.get('pair')
        if need_w and pair is not None:
            # the two half-batches have their own spectral-norm factors: one weight gradient per half (the SN backward
            # is not linear across them), one bias gradient over the whole batch
            has_b = self.bias is not None
            B2 = d.B // 2
            dh = sp.desc(B2, d.H, d.W)
            x_ = st['x']
            dwa, _ = ops.conv_bwd_weight(sp, dh, x_[:B2], dy[:B2], self.weight.shape, False, defer=st.get('wg_defer'))
            dwb, _ = ops.conv_bwd_weight(sp, dh, x_[B2:], dy[B2:], self.weight.shape, False, defer=st.get('wg_defer'))
            st['dwsn'] = [(dwa, st['uv'][0][0], st['uv'][0][1], pair[0]), (dwb, st['uv'][1][0], st['uv'][1][1], pair[1])]
            dbp = st.get('db_part')
            db = None
            if has_b and dbp is not None and sink is not None and st.get('cs_defer') is not None:
                # straight into the bucket view, with the pass's other bias gradients in one launch (ChainFn.backward: flush)
                st['cs_defer'].append((dbp[0], dbp[1], dy.shape[-1], sp.cout, sink.view_of(self.bias), 1.0))
                db = DIRECT
            elif has_b:
                db = ops.colsum_partials(dbp[0], dbp[1], dy.shape[-1], sp.cout) if dbp is not None else ops.colsum(dy, sp.cout)
            grads = [dwa] + ([db] if has_b else [])
        elif need_w:
            has_b = self.bias is not None
            dbp = st.get('db_part') if has_b else None     # column sums of dy from the epilogue that produced it
            db_done = st.get('db_done') if has_b else None      # ... or already accumulated by the norm layer above
            if db_done is None and dbp is not None:
                if sink is not None:
                    cs = st.get('cs_defer')
                    if cs is not None:      # one launch for the pass's bias gradients (ChainFn.backward: flush)
                        cs.append((dbp[0], dbp[1], dy.shape[-1], sp.cout, sink.view_of(self.bias), 1.0))
                    else:
                        ops.colsum_partials(dbp[0], dbp[1], dy.shape[-1], sp.cout, out=sink.view_of(self.bias), beta=1.0)
                    db_done = DIRECT
                else:
                    db_done = ops.colsum_partials(dbp[0], dbp[1], dy.shape[-1], sp.cout)
            want_b = has_b and db_done is None
            if sink is not None and self.sn is None:
                ops.conv_bwd_weight(sp, d, st['x'], dy, self.weight.shape, want_b, dw=sink.view_of(self.weight),
                                    db=sink.view_of(self.bias) if want_b else None, beta=1.0, defer=st.get('wg_defer'))
                grads = [DIRECT] + ([DIRECT] if has_b else [])
            else:
                # (spectral norm: dW_sn is a temporary - the batched SN backward accumulates dW_orig into the view -
                # and the bias gradient joins the pass's small-gradient add)
                dw, db = ops.conv_bwd_weight(sp, d, st['x'], dy, self.weight.shape, want_b, defer=st.get('wg_defer'))
                grads = [dw] + ([db if want_b else db_done] if has_b else [])
            if self.sn is not None:         # spectral-norm backward of all layers is batched by ChainFn.backward
                st['dwsn'] = [(dw, st['u'], st['v'], st['sigma'])]
        dx = None
        if need_dx:
            wb = st.pop('wb', None)
            if wb is None:
                _, wb = ops.conv_prep(sp, d, self.weight, sigma, fwd=False, bwd=True)
            want_cs = st.get('want_dx_colsums', False)
            pa = (st['x'], prev_act[0], prev_act[1]) if prev_act is not None else (None, L.ACT_NONE, 0.0)
            res = None
            if st.get('open_skip'):         # this conv opens a residual branch: the skip gradient is added in its epilogue
                res = st['ctx']['skip_grads'][-1]
                st['ctx']['skip_grad_fused'] = True
            dd = d
            if st.get('dx_f32') and res is None and pair is None and prev_act is None and d.x_bf16 == ops.ST_X3:
                # the layer below is a norm layer: its backward reads this gradient twice - it leaves as fp32 (4 instead of 6
                # bytes per element) and the norm backward emits the three-plane dx (ops._norm_bwd_kinds: ST_X3_XDF32)
                dd = ops._desc_with(d, x_bf16=ops.ST_F32)
            bnf = st.pop('bn_fuse', None)
            if bnf is not None and res is None and pair is None and prev_act is None:
                # the layer below is a BatchNorm (+ReLU / LeakyReLU): its activation derivative and both reductions of its
                # backward are taken in this epilogue (include/iprgan.h: iprgan_conv_bwd_data_bn)
                dx, st['bn_partials'] = ops.conv_bwd_data_bn(sp, d, dy, wb, *bnf)
                return dx, grads
            dx = ops.conv_bwd_data(sp, dd, dy, wb, pa[0], pa[1], pa[2], pair=pair, colsums=want_cs, residual=res)
            if want_cs:
                dx, st['dx_colsums'] = dx
        return dx, grads

    def can_emit_dx_colsums(self, H, W):
        sp = self.spec
        fullmap = sp.k == H and sp.k == W and sp.pad == 0 and not sp.transposed and sp.k > 1
        return ops.c4(sp.cin) > 4 and sp.pad_mode == L.PAD_ZERO and not fullmap


class LinearNHWC(Op):
    """nn.Linear(K -> C*H*W) whose output is consumed as NHWC [B,H,W,C] although the weight rows
    are in the reference's NCHW-flatten order (networks/conv_generator.py:26): rows are permuted
    (c,hw)->(hw,c) on the fly, then it is a 1x1 conv on [B,1,1,K]."""
    fuses_prev_act = False

    def __init__(self, module, channels, hw, act=L.ACT_RELU):
        self.m, self.C, self.HW = module, channels, hw
        self.spec = ops.ConvSpec(module.in_features, channels * hw, 1, act=act)
        self.out_act = (act, 0.0)

    @property
    def weight(self):
        return self.m._parameters['weight']

    @property
    def bias(self):
        return self.m._parameters['bias']

    @property
    def params(self):
        p = self.m._parameters
        return (p['weight'], p['bias'])

    def forward(self, x, st, train):
        B, K = x.shape
        if x.dtype == torch.float32 and ops.fc_nhwc_ok(B, K, self.C, self.HW):
            # one launch on the parameters as they are (csrc/fc.hip): no permuted copy, no prepared operand, no split of x
            x = x.contiguous()
            y = ops.fc_nhwc_fwd(x, self.weight, self.bias, self.C, self.HW, self.spec.act, self.spec.slope)
            st.update(x=x, y=y, direct=True)
            return y
        d = self.spec.desc(B, 1, 1)
        # the row-permuted copies are kept until the weight changes (the optimizer bumps the version counter)
        key = (self.weight._version, self.weight.data_ptr(), self.bias._version, self.bias.data_ptr())
        if getattr(self, '_perm_key', None) != key:
            self._perm = (ops.permute_021(self.weight, self.C, self.HW, K), ops.permute_021(self.bias, self.C, self.HW, 1))
            self._perm_key = key
        wp, bp = self._perm
        x4 = x.contiguous().view(B, 1, 1, K)
        if self.spec.is_identity_prep:
            wf = wp
        else:
            wf, _ = ops.conv_prep(self.spec, d, wp.view(self.C * self.HW, K, 1, 1))
        y = ops.conv_fwd(self.spec, d, x4, wf, bp).view(B, self.C * self.HW)
        st.update(x=x4, y=y, d=d, wp=wp)
        return y

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        sp = self.spec
        if st.get('direct'):
            act = L.ACT_NONE if st.get('dy_is_preact', False) else sp.act
            grads = []
            if need_w:
                if sink is not None:       # straight into the bucket views, PyTorch layout (the kernel addresses rows c*HW + hw)
                    ops.fc_nhwc_bwd(st['x'], st['y'], dy, self.weight.shape, self.C, self.HW, act, sp.slope,
                                    dw=sink.view_of(self.weight), db=sink.view_of(self.bias), beta=1.0)
                    grads = [DIRECT, DIRECT]
                else:
                    grads = list(ops.fc_nhwc_bwd(st['x'], st['y'], dy, self.weight.shape, self.C, self.HW, act, sp.slope))
            dx = None
            if need_dx:                     # (the VAE decoder: its latent comes from the encoder) - the 1x1 form of rounds 1-5
                B, K = st['x'].shape
                d = sp.desc(B, 1, 1)
                dz = ops.act_bwd(dy, st['y'], sp.act, sp.slope) if act != L.ACT_NONE else dy
                wp = ops.permute_021(self.weight, self.C, self.HW, K)
                _, wb = ops.conv_prep(sp, d, wp.view(sp.cout, K, 1, 1), fwd=False, bwd=True)
                dx = ops.f32(ops.conv_bwd_data(sp, d, dz.view(B, 1, 1, sp.cout), wb)).view(B, K)
            return dx, grads
        d = st['d']
        B, K = d.B, sp.cin
        if not st.get('dy_is_preact', False):
            dy = ops.act_bwd(dy, st['y'], sp.act, 0.0)
        dy4 = dy.view(B, 1, 1, sp.cout)
        grads = []
        if need_w:
            dwp, dbp = ops.conv_bwd_weight(sp, d, st['x'], dy4, (sp.cout, K, 1, 1), True)
            if sink is not None:        # un-permute straight into the bucket views
                ops.permute_021(dwp, self.HW, self.C, K, out=sink.view_of(self.weight), beta=1.0)
                ops.permute_021(dbp, self.HW, self.C, 1, out=sink.view_of(self.bias), beta=1.0)
                grads = [DIRECT, DIRECT]
            else:
                dw = ops.permute_021(dwp, self.HW, self.C, K).view(sp.cout, K)
                db = ops.permute_021(dbp, self.HW, self.C, 1)
                grads = [dw, db]
        dx = None
        if need_dx:
            _, wb = ops.conv_prep(sp, d, st['wp'].view(sp.cout, K, 1, 1), fwd=False, bwd=True)
            dx = ops.conv_bwd_data(sp, d, dy4, wb).view(B, K)
        return dx, grads


class BatchNorm(Op):
    """BatchNorm2d (+activation) over NHWC; owns nothing, mutates the module's running stats."""

    def __init__(self, bn_module, act=L.ACT_NONE, slope=0.0, prelu=None):
        self.m = bn_module
        self.act, self.slope = act, slope
        # prelu: the nn.PReLU() (one slope) that follows the norm layer (networks/sr_resnet.py:7,13), folded into the norm's
        # apply passes (iprgan_bn_prelu_fwd / _bwd): no separate pass over the tensor for its forward or backward
        self.prelu = prelu
        self.out_act = (L.ACT_NONE, 0.0)     # handles its own activation derivative

    @property
    def params(self):
        p = self.m._parameters
        return (p['weight'], p['bias']) + ((self.prelu._parameters['weight'],) if self.prelu is not None else ())

    def forward(self, x, st, train):
        m = self.m
        use_batch = train or not m.track_running_stats
        track = train and m.track_running_stats
        mom = m.momentum
        counter = None
        if track and m.num_batches_tracked is not None:
            if mom is None:                 # cumulative moving average: the host needs the count (never in the reference)
                m.num_batches_tracked.add_(1)
                mom = 1.0 / float(m.num_batches_tracked)
            else:
                counter = m.num_batches_tracked       # incremented by the statistics kernel
        cs = st['ctx'].pop('conv_stats', None) if use_batch else None
        res = None
        if st.get('close_skip'):            # the residual block ends right behind this layer: its add rides on the apply
            res = st['ctx']['skips'][-1]
            st['ctx']['skip_fused'] = True
        rm = m.running_mean if (track or not use_batch) else None
        rv = m.running_var if (track or not use_batch) else None
        if self.prelu is not None:
            y, mean, invstd = ops.bn_prelu_fwd(x, m.weight, m.bias, rm, rv, m.eps, mom if mom is not None else 0.0,
                                               use_batch, self.prelu.weight, conv_stats=cs[0] if cs else None,
                                               conv_bias=cs[1] if cs else None, counter=counter, residual=res)
        else:
            y, mean, invstd = ops.bn_fwd(x, m.weight, m.bias, rm, rv,
                                         m.eps, mom if mom is not None else 0.0, use_batch, self.act, self.slope,
                                         conv_stats=cs[0] if cs else None, conv_bias=cs[1] if cs else None,
                                         counter=counter, residual=res)
        st.update(x=x, y=y, mean=mean, invstd=invstd, use_batch=use_batch)
        return y

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        tgt = st.get('dbias_prev')          # (tensor, beta): bias gradient of the convolution below, see ChainFn.backward
        pre = st.pop('dz_partials', None)
        if pre is not None:                 # dy is dz: the consumer's dgrad epilogue applied act' and took both reductions
            dx, dg, db = ops.bn_bwd_pre(st['x'], dy, self.m.weight, st['mean'], st['invstd'], pre,
                                        dbias=tgt[0] if tgt else None, dbias_beta=tgt[1] if tgt else 0.0)
            return dx, [dg, db]
        if self.prelu is not None:
            dx, dg, db, da = ops.bn_prelu_bwd(st['x'], dy, self.m.weight, self.m.bias, st['mean'], st['invstd'],
                                              self.prelu.weight, dbias=tgt[0] if tgt else None,
                                              dbias_beta=tgt[1] if tgt else 0.0)
            return dx, [dg, db, da]
        dx, dg, db = ops.bn_bwd(st['x'], st['y'], dy, self.m.weight, st['mean'], st['invstd'],
                                self.act, self.slope, beta=self.m.bias, dbias=tgt[0] if tgt else None,
                                dbias_beta=tgt[1] if tgt else 0.0)
        return dx, [dg, db]


class GemvHead(Op):
    """flatten (NCHW order in the reference, sn_discriminator.py:27-32) + SN Linear(K -> 1) + view(-1).
    Our activations are NHWC, so the weight vector is permuted (c,hw)->(hw,c) per pass."""
    fuses_prev_act = True

    def __init__(self, module, channels, hw):
        self.m, self.C, self.HW = module, channels, hw

    @property
    def weight(self):
        return self.m._parameters['weight_orig']

    @property
    def bias(self):
        return self.m._parameters['bias']

    @property
    def sn(self):
        b = self.m._buffers
        return (b['weight_u'], b['weight_v'])

    @property
    def u(self):
        return self.m._buffers['weight_u']

    @property
    def v(self):
        return self.m._buffers['weight_v']

    @property
    def params(self):
        p = self.m._parameters
        return (p['weight_orig'], p['bias'])

    def forward(self, x, st, train):
        B = x.shape[0]
        K = self.C * self.HW
        sigma = st.get('sigma')
        pair = st.get('pair')
        if sigma is None and pair is None:
            sigma = ops.sn_power_iter(self.weight, self.u, self.v, train)
            st['u'], st['v'] = self.u.clone(), self.v.clone()
        key = (self.weight._version, self.weight.data_ptr())
        if getattr(self, '_perm_key', None) != key:
            self._perm, self._perm_key = ops.permute_021(self.weight, self.C, self.HW, 1), key
        wp = self._perm
        x2 = x.view(B, K)
        if pair is not None:                # paired pass: each half-batch with its own sigma
            if _GEMV_PAIR:                  # both halves in one launch (rows >= B / 2 take the second sigma): bit-identical
                y = ops.gemv_fwd_pair(x2, wp, self.bias, pair[0], pair[1])
            else:
                B2 = B // 2
                y = ops.empty((B,), x2)
                ops.gemv_fwd(x2[:B2], wp, self.bias, pair[0], out=y[:B2])
                ops.gemv_fwd(x2[B2:], wp, self.bias, pair[1], out=y[B2:])
        else:
            y = ops.gemv_fwd(x2, wp, self.bias, sigma)
        st.update(x=x2, xshape=tuple(x.shape), wp=wp, sigma=sigma)
        return y

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        pa, ps = prev_act if prev_act is not None else (L.ACT_NONE, 0.0)
        pair = st.get('pair')
        dy = dy.contiguous()
        unperm = lambda t: ops.permute_021(t, self.HW, self.C, 1).view(1, -1)
        if pair is not None and _GEMV_PAIR and ops.is16(st['x']) == ops.ST_X3 and ops.pstride(st['x']) == st['x'].numel():
            x2 = st['x']
            dx, dw2, db2 = ops.gemv_bwd_pair(x2, st['wp'], dy, pair[0], pair[1], need_dx, need_w,
                                             x2 if prev_act is not None else None, pa, ps)
            grads = []
            if need_w:
                st['dwsn'] = [(unperm(dw2[h]), st['uv'][h][0], st['uv'][h][1], pair[h]) for h in (0, 1)]
                grads = [st['dwsn'][0][0], ops.add(db2[0:1], db2[1:2])]
            return (dx.view(st['xshape']) if dx is not None else None), grads
        if pair is not None:
            x2 = st['x']
            B2 = x2.shape[0] // 2
            dx = ops._empty_like(x2) if need_dx else None          # (same storage kind as x: fp32, bf16 or three planes)
            entries, dbs = [], []
            for h, sl in enumerate((slice(0, B2), slice(B2, None))):
                _, dwp, db = ops.gemv_bwd(x2[sl], st['wp'], dy[sl], pair[h], need_dx, need_w,
                                          x2[sl] if prev_act is not None else None, pa, ps,
                                          dx_out=dx[sl] if need_dx else None)
                if need_w:
                    entries.append((unperm(dwp), st['uv'][h][0], st['uv'][h][1], pair[h]))
                    dbs.append(db)
            grads = []
            if need_w:
                st['dwsn'] = entries
                grads = [entries[0][0], ops.add(dbs[0], dbs[1])]
            return (dx.view(st['xshape']) if dx is not None else None), grads
        dx, dwp, db = ops.gemv_bwd(st['x'], st['wp'], dy, st['sigma'], need_dx, need_w,
                                   st['x'] if prev_act is not None else None, pa, ps)
        grads = []
        if need_w:
            dwsn = unperm(dwp)
            st['dwsn'] = [(dwsn, st['u'], st['v'], st['sigma'])]
            grads = [dwsn, db]
        return (dx.view(st['xshape']) if dx is not None else None), grads


class InstanceNorm(Op):
    """InstanceNorm2d (affine or not, never tracks running stats) + optional activation."""

    def __init__(self, module, act=L.ACT_NONE, slope=0.0):
        self.m, self.act, self.slope = module, act, slope

    @property
    def params(self):
        return (self.m.weight, self.m.bias) if self.m.weight is not None else ()

    def forward(self, x, st, train):
        cs = st['ctx'].pop('conv_stats', None)
        res = None
        if st.get('close_skip'):
            res = st['ctx']['skips'][-1]
            st['ctx']['skip_fused'] = True
        y, mean, invstd = ops.instnorm_fwd(x, self.m.weight, self.m.bias, self.m.eps, self.act, self.slope,
                                           conv_stats=cs[0] if cs else None, conv_bias=cs[1] if cs else None,
                                           residual=res)
        st.update(x=x, y=y, mean=mean, invstd=invstd)
        return y

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        tgt = st.get('dbias_prev')
        dx, dg, db = ops.instnorm_bwd(st['x'], st['y'], dy, self.m.weight, st['mean'], st['invstd'],
                                      self.act, self.slope, beta=self.m.bias, dbias=tgt[0] if tgt else None,
                                      dbias_beta=tgt[1] if tgt else 0.0)
        return dx, ([dg, db] if self.m.weight is not None else [])


class PReLU(Op):
    """nn.PReLU() with a single learnable slope."""

    def __init__(self, module):
        self.m = module

    @property
    def params(self):
        return (self.m.weight,)

    def forward(self, x, st, train):
        st['x'] = x
        return ops.prelu_fwd(x, self.m.weight)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        dx, dalpha = ops.prelu_bwd(st['x'], dy.contiguous(), self.m.weight)
        return dx, [dalpha]


class PixelShuffle2(Op):
    def forward(self, x, st, train):
        return ops.pixel_shuffle2(x)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        return ops.pixel_shuffle2(dy.contiguous(), inverse=True), []


class PixelShufflePReLU(Op):
    """nn.PixelShuffle(2) followed by nn.PReLU() (one slope) in one pass each way (sr_resnet.py:39-45)."""

    def __init__(self, module):
        self.m = module

    @property
    def params(self):
        return (self.m.weight,)

    def forward(self, x, st, train):
        st['x'] = x
        return ops.pixel_shuffle2_prelu_fwd(x, self.m.weight)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        dx, dalpha = ops.pixel_shuffle2_prelu_bwd(st['x'], dy.contiguous(), self.m.weight)
        return dx, [dalpha]


class MaxPool2(Op):
    def forward(self, x, st, train):
        st['x'] = x
        return ops.maxpool2_fwd(x)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        return ops.maxpool2_bwd(st['x'], dy.contiguous()), []


class SkipStart(Op):
    """Marks the input of a residual branch: y = x + f(x) is [SkipStart, f..., SkipEnd]."""

    def forward(self, x, st, train):
        st['ctx']['skips'].append(x)
        return x

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        g = st['ctx']['skip_grads'].pop()
        if st['ctx'].pop('skip_grad_fused', False):      # already added in the epilogue of the branch's first dgrad
            return dy, []
        return ops.add(dy, g), []


class SkipEnd(Op):
    def forward(self, x, st, train):
        skip = st['ctx']['skips'].pop()
        if st['ctx'].pop('skip_fused', False):           # the norm layer in front of this op added it on its apply pass
            return x
        return ops.add(x, skip)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        st['ctx']['skip_grads'].append(dy)
        return dy, []


class Squeeze(Op):
    """`.squeeze()` of a [B,1,1,C4>=1] logit map to [B] (networks/discriminator_96.py:24-25) or
    NHWC [B,H,W,1(+pad)] -> NCHW [B,1,H,W] handled by ToNCHW; this op only flattens B x 1 x 1."""

    def forward(self, x, st, train):
        st['shape'] = tuple(x.shape)
        B = x.shape[0]
        return ops.nhwc_to_nchw(x, 1).view(B)

    def backward(self, dy, st, need_dx, need_w, prev_act, sink=None):
        B = st['shape'][0]
        return ops.nchw_to_nhwc(dy.contiguous().view(B, 1, 1, 1)), []


def drop_operand_caches(root):
    """Forget every version-keyed operand copy (prepared conv operands of layers without spectral norm, the row-permuted
    weights of LinearNHWC / GemvHead) of every engine network under ``root`` (an nn.Module).  graphs.GraphedStep calls it
    right before a capture: a cache HIT during the capture would bake the address of a tensor computed by an earlier,
    eager pass into the graph, and later replays would read that stale copy (observed: the discriminator head's permuted
    weight is computed in update_g's pass and reused by the next step's update_d pass - same weights, same version)."""
    n = 0
    for mod in root.modules():
        ch = mod.__dict__.get('_chain')
        if ch is None:
            continue
        for op in ch.ops:
            for k in ('_op_wf', '_op_wb', '_perm_key', '_perm'):
                if op.__dict__.pop(k, None) is not None:
                    n += 1
    return n


class Chain:
    """A sequential network plan.  ``ops`` run in order; parameters are collected in op order."""

    def __init__(self, ops_list):
        self.ops = list(ops_list)

    @property
    def params(self):
        return [p for op in self.ops for p in op.params]

    def __call__(self, x, train, pair=False):
        """pair=True: ``x`` holds two half-batches that the reference sends through this (spectrally normalised) network
        one after the other; the power iteration runs twice and each half is scaled by its own sigma
        (include/iprgan.h: paired pass)."""
        if not x.is_cuda:
            raise RuntimeError('iprgan networks run on the HIP kernels only: input must be a GPU tensor '
                               '(there is no CPU fallback; use the reference/oracle for CPU runs)')
        L.load()
        params = self.params
        red = None
        if torch.is_grad_enabled():
            # a recorded pass that will produce parameter gradients: its optimizer's reducer counts it, so that the
            # backward executor knows which pass of the step is the last one (parallel.GradReducer)
            for p in params:
                if p.requires_grad:
                    red = parallel.owner_of(p)
                    break
            if red is not None:
                red.note_forward()
        return ChainFn.apply(self, (bool(train), red, bool(pair)), x, *params)


class ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain, opts, x, *params):
        train, red, pair = opts
        h = x.detach()
        if h.dtype != torch.float32:
            raise RuntimeError('iprgan networks take float32 inputs')
        shared = {'skips': [], 'skip_grads': []}
        stash = [{'ctx': shared} for _ in chain.ops]
        # spectral norm depends on the weights only: run every layer's power iteration up front, together
        sn_idx = [i for i, op in enumerate(chain.ops) if getattr(op, 'sn', None) is not None]
        if sn_idx:
            ws_, us_, vs_ = zip(*[(chain.ops[i].weight,) + tuple(chain.ops[i].sn) for i in sn_idx])
            sig, uo, vo = ops.sn_power_iter_multi(list(ws_), list(us_), list(vs_), train)
            if pair:                        # the second half-batch sees the NEXT power iteration, as two calls would
                sig2, uo2, vo2 = ops.sn_power_iter_multi(list(ws_), list(us_), list(vs_), train)
                for k, i in enumerate(sn_idx):
                    stash[i].update(pair=(sig[k:k + 1], sig2[k:k + 1]), uv=((uo[k], vo[k]), (uo2[k], vo2[k])))
            else:
                for k, i in enumerate(sn_idx):
                    stash[i].update(sigma=sig[k:k + 1], u=uo[k], v=vo[k])
        elif pair:
            raise RuntimeError('a paired pass is for spectrally normalised networks')
        # ... and every conv layer's forward operand (tap-major copy, divided by sigma) in one launch.  Operands of
        # layers without spectral norm depend on the weight only: they are kept until the weight changes (the optimizer
        # bumps the version counter), so the second and third pass of a step - and a frozen network like VGG19 for the
        # whole run - reuse them.
        cv = [i for i, op in enumerate(chain.ops) if isinstance(op, Conv)] if _BATCH_PREP else []
        todo = [i for i in cv if not chain.ops[i].cached_operand('wf', stash[i])]
        if todo:
            wfs = ops.conv_prep_multi([chain.ops[i].spec for i in todo], [chain.ops[i].weight for i in todo],
                                      [stash[i].get('sigma') for i in todo])
            for i, wf in zip(todo, wfs):
                stash[i]['wf'] = wf
                chain.ops[i].keep_operand('wf', wf, stash[i])
        n_ops = len(chain.ops)
        for i, (op, st) in enumerate(zip(chain.ops, stash)):
            if _FUSE_STATS and isinstance(op, Conv) and i + 1 < n_ops and not pair and h.dim() == 4:
                nxt = chain.ops[i + 1]
                per_inst = isinstance(nxt, InstanceNorm)
                if (per_inst or (isinstance(nxt, BatchNorm) and (train or not nxt.m.track_running_stats))) \
                        and op.can_emit_stats(h.shape[1], h.shape[2], per_inst):
                    st['emit_stats'] = True
            if isinstance(op, Conv) and i + 1 < n_ops and isinstance(chain.ops[i + 1], (BatchNorm, InstanceNorm)) and not pair \
                    and h.dim() == 4 and ops._NORM_XF32 and op.spec.act == L.ACT_NONE and ops.act_kind(op.spec.cout) == ops.ST_X3:
                st['y_f32'] = True
            if _FUSE_STATS and isinstance(op, (BatchNorm, InstanceNorm)) and i + 1 < n_ops and isinstance(chain.ops[i + 1], SkipEnd):
                st['close_skip'] = True
            h = op.forward(h, st, train)
            shared.pop('conv_stats', None) if not isinstance(op, Conv) else None
        ctx.chain, ctx.stash, ctx.red = chain, stash, red
        ctx.plist = params                  # (the parameter objects of this pass, in op order: backward walks them again)
        # the backward pass reads the live parameters (weights for the data gradients, spectral-norm factors):
        # remember their versions so that a step taken between this forward and its backward is an error, as it
        # is in PyTorch for tensors saved by autograd
        ctx.versions = [p._version for p in params]
        return ops.f32(h)                   # (autograd sees fp32 tensors only: a three-plane result is joined here)

    @staticmethod
    def backward(ctx, dy):
        chain, stash, red = ctx.chain, ctx.stash, ctx.red
        if stash is None:
            raise RuntimeError('this network pass has already been back-propagated: the engine frees a pass\'s '
                               'activations in backward (retain_graph / double backward are not supported)')
        need_x = ctx.needs_input_grad[2]
        need_p = ctx.needs_input_grad[3:]
        ops_list = chain.ops
        plist = ctx.plist
        for p, v in zip(plist, ctx.versions):
            if p._version != v:
                raise RuntimeError('a parameter of this network was modified in place (optimizer step?) between the '
                                   'forward pass and its backward pass: the gradients would be computed with the '
                                   'new weights')
        # which ops need to produce dx: everything after the first op that has a trainable parameter
        # needing a gradient, or all of them when the input itself needs one
        first_needed = 0 if need_x else None
        pi = 0
        op_need_w, op_need_p = [], []
        for i, op in enumerate(ops_list):
            n = len(op.params)
            w = any(need_p[pi:pi + n]) if n else False
            op_need_w.append(w)
            op_need_p.append(need_p[pi:pi + n])
            if w and first_needed is None:
                first_needed = i
            pi += n
        # gradient sink: the optimizer's reducer when it is armed (bucket views are the .grad tensors); otherwise
        # gradients are returned to autograd
        wants = any(op_need_w)
        sink = red if (wants and red is not None and red.armed
                       and all(red.owns(p) for p, n in zip(plist, need_p) if n)) else None
        counted = wants and red is not None
        final = red.begin_pass() if counted else False
        if sink is not None and final:          # parameters of the optimizer's OTHER networks are final already
            mine = set(plist)
            red.params_done([p for p in red.params if p not in mine])
        grads_per_op = [None] * len(ops_list)
        small_dst, small_src, sn_wait = [], [], []
        wg_wait = []        # slab reduces the convolutions of this pass owe (ops.conv_bwd_weight(defer=...)): one launch in flush()
        cs_wait = []        # bias gradients from dgrad-epilogue partials, straight into bucket views: one launch in flush()

        small_seen = set()

        def flush():
            """deferred writes into the bucket views: spectral-norm backward of the layers seen so far (batched; a paired
            pass has two (dW_sn, u, v, sigma) entries per layer: two rounds, the second accumulating) and the small
            gradients (one multi-tensor add); first of all the slab reduces of the weight gradients, which the spectral-norm
            backward reads"""
            ops.wgrad_reduce_flush(wg_wait)
            ops.colsum_partials_flush(cs_wait)
            if sn_wait:
                out_of = {i: (red.view_of(ops_list[i].weight) if sink is not None else torch.empty_like(ops_list[i].weight))
                          for i in sn_wait}
                for r in range(max(len(stash[i]['dwsn']) for i in sn_wait)):
                    idx = [i for i in sn_wait if len(stash[i]['dwsn']) > r]
                    ent = [stash[i]['dwsn'][r] for i in idx]
                    ops.sn_bwd_multi([e[0] for e in ent], [ops_list[i].weight for i in idx], [e[1] for e in ent],
                                     [e[2] for e in ent], [e[3] for e in ent], outs=[out_of[i].view_as(ops_list[i].weight)
                                                                                      for i in idx],
                                     beta=1.0 if (sink is not None or r > 0) else 0.0)
                for i in sn_wait:
                    grads_per_op[i][0] = DIRECT if sink is not None else out_of[i]
                del sn_wait[:]
            if small_dst:
                ops.axpy_multi(small_dst, small_src)
                del small_dst[:], small_src[:]
                small_seen.clear()

        if first_needed is not None and _BATCH_PREP:        # backward-data operands of every conv that must produce dx: one launch
            cv = [i for i, op in enumerate(ops_list) if isinstance(op, Conv) and (i > first_needed or (need_x and i == 0))]
            todo = [i for i in cv if not ops_list[i].cached_operand('wb', stash[i])]
            if todo:
                wbs = ops.conv_prep_multi([ops_list[i].spec for i in todo], [ops_list[i].weight for i in todo],
                                          [stash[i].get('sigma') for i in todo], bwd=True)
                for i, wb in zip(todo, wbs):
                    stash[i]['wb'] = wb
                    ops_list[i].keep_operand('wb', wb, stash[i])
        g = dy.contiguous()
        for i in range(len(ops_list) - 1, -1, -1):
            op, st = ops_list[i], stash[i]
            if first_needed is None or i < first_needed:
                break
            need_dx = (i > first_needed) or (need_x and i == 0)
            prev = ops_list[i - 1] if i > 0 else None
            fuse = None
            if need_dx and prev is not None and op.fuses_prev_act and prev.out_act[0] != L.ACT_NONE:
                fuse = prev.out_act
                stash[i - 1]['dy_is_preact'] = True
            if (_FUSE_STATS and need_dx and isinstance(op, Conv) and isinstance(prev, Conv) and prev.bias is not None
                    and op_need_w[i - 1] and (fuse is not None or prev.out_act[0] == L.ACT_NONE)
                    and op.can_emit_dx_colsums(st['d'].H, st['d'].W)):
                st['want_dx_colsums'] = True      # the column sums of this dx are the bias gradient of the layer below
            if (_FUSE_STATS and isinstance(op, (BatchNorm, InstanceNorm)) and isinstance(prev, Conv)
                    and prev.bias is not None and op_need_w[i - 1] and 'pair' not in stash[i - 1]):
                # the column sums of this norm layer's dx are the bias gradient of the convolution below: they ride on the
                # norm's apply pass (straight into the bucket view when there is one)
                if sink is not None:
                    st['dbias_prev'] = (red.view_of(prev.bias), 1.0)
                    stash[i - 1]['db_done'] = DIRECT
                else:
                    t = ops.empty((prev.bias.numel(),), g)
                    st['dbias_prev'] = (t, 0.0)
                    stash[i - 1]['db_done'] = t
            if (_FUSE_STATS and need_dx and isinstance(op, Conv) and isinstance(prev, SkipStart)
                    and not (ops.c4(op.spec.cin) <= 4 and op.spec.stride == 1)):
                st['open_skip'] = True
            if (_FUSE_BN_BWD and _FUSE_STATS and need_dx and isinstance(op, Conv) and isinstance(prev, BatchNorm)
                    and prev.prelu is None and stash[i - 1].get('use_batch') and prev.act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU)
                    and 'pair' not in st and not st.get('open_skip') and ops.conv_bwd_data_bn_ok(st['d'])
                    and i - 1 >= first_needed):
                pst = stash[i - 1]
                st['bn_fuse'] = (pst['x'], pst['mean'], pst['invstd'], prev.m.weight, prev.m.bias, prev.act, prev.slope)
            if (isinstance(op, Conv) and need_dx and isinstance(prev, (BatchNorm, InstanceNorm)) and ops._NORM_DYF32
                    and ops._NORM_XF32 and i - 1 >= first_needed and 'x' in stash[i - 1]
                    and ops.is16(stash[i - 1]['x']) == ops.ST_F32 and ops._norm_splits_here(stash[i - 1]['x'])):
                st['dx_f32'] = True         # (the norm layer's saved input is fp32: the norm backward takes an fp32 dy beside it)
            if isinstance(op, Conv):
                st['wg_defer'] = wg_wait
                st['cs_defer'] = cs_wait
            g, pg = op.backward(g, st, need_dx, op_need_w[i], fuse, sink)
            if 'bn_partials' in st:
                stash[i - 1]['dz_partials'] = st.pop('bn_partials')
            if 'dx_colsums' in st:
                stash[i - 1]['db_part'] = st.pop('dx_colsums')
            grads_per_op[i] = pg
            if op_need_w[i] and 'dwsn' in st:
                sn_wait.append(i)
            if sink is not None and op.params:
                if red.trace is not None and op_need_w[i]:
                    red.trace.append(('wgrad', i, next(parallel._seq)))
                for k, p in enumerate(op.params):
                    gk = pg[k] if k < len(pg) else None
                    if gk is None or not op_need_p[i][k]:
                        continue
                    red.touch(p)
                    if gk is not DIRECT and not (k == 0 and 'dwsn' in st):
                        if id(p) in small_seen:          # one destination per launch
                            flush()
                        small_seen.add(id(p))
                        small_dst.append(red.view_of(p))
                        small_src.append(gk.reshape(p.shape) if gk.shape != p.shape else gk)
                if final:
                    if red.bucket_would_complete(op.params):
                        flush()
                    red.params_done(op.params)
        flush()
        if counted:
            red.end_pass()                      # last pass: everything that has not left yet is sent now
        out = []
        pi = 0
        for i, op in enumerate(ops_list):
            n = len(op.params)
            pg = grads_per_op[i] or [None] * n
            for k in range(n):
                gk = pg[k] if (k < len(pg) and need_p[pi + k]) else None
                out.append(None if (gk is DIRECT or sink is not None) else gk)
            pi += n
        ctx.stash = None
        return (None, None, ops.f32(g) if need_x else None, *out)
