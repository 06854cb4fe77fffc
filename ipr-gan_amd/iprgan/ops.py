"""Functional wrappers over the C ABI: allocate outputs/workspaces from PyTorch's caching
allocator (plumbing) and enqueue the HIP kernels on the current stream.  No autograd here and
no ATen arithmetic; activations are fp32 NHWC tensors [B,H,W,C4] (C4 = channels padded to 4).
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import ConvDesc, call, ptr, ptr32, query, stream


def c4(c):
    return (c + 3) & ~3


def _int_table(sizes):
    return (C.c_int * len(sizes))(*sizes)


_POISON = os.environ.get('IPRGAN_DEBUG_POISON') == '1'     # debugging aid: every workspace / output starts as NaN


def empty(shape, like, dtype=torch.float32):
    if _POISON:
        return torch.full(shape, float('nan'), dtype=dtype, device=like.device)
    return torch.empty(shape, dtype=dtype, device=like.device)


def scratch(n, like):
    """Workspace of ``n`` floats (slabs, partials of a reduction, padded copies): allocated like an output, but left out of the
    algorithmic byte count of bench.py's accounting step (_lib.acct_scratch)."""
    return L.acct_scratch(empty((n,), like))


def _empty_like(t):
    if t.dtype == torch.bfloat16 and L.act_x3():
        return empty_kind(t.shape, t, ST_X3)
    return torch.full_like(t, float('nan')) if (_POISON and t.is_floating_point()) else torch.empty_like(t)


# Storage kinds of an activation (include/iprgan.h: IPRGAN_ST_*).  X3 = three bf16 planes (x = h + m + l exactly): the
# tensor object IS its h plane (a contiguous bf16 tensor of the activation's shape) inside a storage that holds all three,
# plane-major; plane stride = storage elements / 3 (a batch slice keeps the stride of the tensor it was cut from).
ST_F32, ST_BF16, ST_X3 = 0, 1, 2


def act_kind(channels):
    """Storage kind of an NHWC activation with ``channels`` channels: three planes in 'fp32x3' mode when the padded channel
    count is a multiple of 32, bf16 in 'bf16act' mode when it is a multiple of 64 (include/iprgan.h), fp32 otherwise."""
    if L.act_x3() and c4(channels) % 32 == 0:
        return ST_X3
    if L.act_bf16() and c4(channels) % 64 == 0:
        return ST_BF16
    return ST_F32


def act_dtype(channels):
    return torch.float32 if act_kind(channels) == ST_F32 else torch.bfloat16


def is16(t):
    """Storage kind of a tensor: a bf16 tensor is a three-plane tensor in 'fp32x3' mode (nothing else is bf16 there)."""
    if t is None or t.dtype != torch.bfloat16:
        return ST_F32
    return ST_X3 if L.act_x3() else ST_BF16


kind = is16


def pstride(t):
    """Plane stride (elements) of a three-plane tensor; 0 for the other kinds (the descriptors' "contiguous" value)."""
    return t.untyped_storage().nbytes() // 6 if is16(t) == ST_X3 else 0


def span_ok(t):
    """A three-plane tensor is addressed through ONE 32-bit buffer descriptor from its h plane: the l plane of a batch slice
    ends at 2 * plane stride + its own element count (a slice keeps its parent's stride).  False when that span leaves the
    range - the convolution wrappers then hand the kernels an fp32 copy instead of failing inside a launch."""
    return is16(t) != ST_X3 or (2 * pstride(t) + t.numel()) * 2 < 0x7fffffff


def empty_kind(shape, like, k):
    """Uninitialised activation of storage kind ``k`` on ``like``'s device."""
    if k != ST_X3:
        return empty(tuple(shape), like, torch.bfloat16 if k == ST_BF16 else torch.float32)
    buf = torch.empty((3,) + tuple(shape), dtype=torch.bfloat16, device=like.device)      # plane-major: [h | m | l]
    if _POISON:
        buf.fill_(float('nan'))
    return buf[0]                           # the tensor IS its h plane; plane stride = the storage's size / 3 (pstride)


def to_kind(t, k):
    """Copy of an activation in storage kind ``k`` (fp32 <-> bf16: round-to-nearest-even; fp32 <-> three planes: exact)."""
    kt = is16(t)
    if kt == k:
        return t
    if ST_X3 in (kt, k):
        if ST_BF16 in (kt, k):
            raise RuntimeError('no conversion between bf16 and three-plane activations')
        out = empty_kind(t.shape, t, k)
        call('iprgan_cast_planes', ptr(t), ptr(out), t.numel(), pstride(t) if kt == ST_X3 else t.numel(),
             1 if k == ST_X3 else 0, stream())
        return out
    out = torch.empty(t.shape, dtype=torch.bfloat16 if k == ST_BF16 else torch.float32, device=t.device)
    call('iprgan_cast', ptr(t), ptr(out), t.numel(), kt, k, stream())
    return out


def cast(t, dtype):
    """fp32 <-> the 2-byte storage kind of the current math mode (bf16, or three planes in 'fp32x3' mode)."""
    if dtype == torch.float32:
        return to_kind(t, ST_F32)
    return to_kind(t, ST_X3 if L.act_x3() else ST_BF16)


def f32(t):
    """fp32 view / copy of an activation of any kind (ops without a form for the 2-byte kinds go through it)."""
    return t if (t is None or t.dtype == torch.float32) else to_kind(t, ST_F32)


_IO_F32 = False


def io_f32(on):
    """Test switch: the convolution wrappers hand back fp32 copies of three-plane results (the kernels run on three-plane
    operands and write three-plane results all the same), so that op-level parity tests written against fp32 tensors run
    unchanged in 'fp32x3' mode."""
    global _IO_F32
    _IO_F32 = bool(on)


def _out(t):
    return f32(t) if (_IO_F32 and t is not None) else t


def _f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError('this op has no bf16-activation form yet (bf16act mode covers the DCGAN-family layers)')


def _desc_with(d, **kw):
    """Copy of a conv descriptor with some fields replaced (storage kinds / plane strides of this call's tensors)."""
    c = ConvDesc.from_buffer_copy(d)
    for k_, v in kw.items():
        setattr(c, k_, v)
    return c


# Size / capability queries of the library are pure functions of (descriptor, math mode): an eager step asked ~100 of them
# through ctypes; they are answered from a dict keyed on the descriptor's bytes (VERDICT r05 next #5b: host cost of an eager step)
_qcache = {}


def dquery(name, d, *extra):
    k = (name, bytes(d), extra, L._math_cached)
    v = _qcache.get(k)
    if v is None:
        v = _qcache[k] = query(name, C.byref(d), *extra)
    return v


# ---- layout ---------------------------------------------------------------------------------
def nchw_to_nhwc(x):
    _f32(x)
    B, Cc, H, W = x.shape
    y = empty((B, H, W, c4(Cc)), x)
    call('iprgan_nchw_to_nhwc', ptr(x.contiguous()), ptr(y), B, Cc, H, W, stream())
    return y


def nhwc_to_nchw(x, channels):
    if x.dtype != torch.float32:
        x = cast(x, torch.float32)
    B, H, W, _ = x.shape
    y = empty((B, channels, H, W), x)
    call('iprgan_nhwc_to_nchw', ptr(x), ptr(y), B, channels, H, W, stream())
    return y


def permute_021(src, A, Bd, K, out=None, beta=0.0):
    """dst[b][a][k] = src[a][b][k]; returns a flat tensor of A*Bd*K floats (or ``out = beta*out + permuted``)."""
    dst = empty((A * Bd * K,), src) if out is None else out
    call('iprgan_permute_021', ptr(src), ptr(dst), A, Bd, K, float(beta), stream())
    return dst


def act_bwd(dy, out, act, slope=0.0):
    if ST_X3 in (is16(dy), is16(out)):          # (no three-plane form: a stand-alone activation backward is rare)
        dy, out = f32(dy), f32(out)
    if dy.dtype != out.dtype:
        dy = cast(dy, out.dtype)
    dz = _empty_like(dy)
    call('iprgan_act_bwd', ptr(dy), ptr(out), ptr(dz), dy.numel(), act, float(slope), is16(dy), stream())
    return dz


# ---- convolution ------------------------------------------------------------------------------
class ConvSpec:
    """Static description of one Conv2d / ConvTranspose2d layer (include/iprgan.h: iprgan_conv_desc)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, outpad=0, transposed=False,
                 pad_mode=L.PAD_ZERO, act=L.ACT_NONE, slope=0.0):
        self.cin, self.cout, self.k = cin, cout, k
        self.stride, self.pad, self.outpad = stride, pad, outpad
        self.transposed, self.pad_mode, self.act, self.slope = transposed, pad_mode, act, slope

    def out_hw(self, H, W):
        if self.transposed:
            f = lambda n: (n - 1) * self.stride - 2 * self.pad + self.k + self.outpad
        else:
            f = lambda n: (n + 2 * self.pad - self.k) // self.stride + 1
        return f(H), f(W)

    def desc(self, B, H, W, x16=None, y16=None):
        """x16 / y16: storage type of the layer's input / output activation (default: the bf16act rule)."""
        key = (B, H, W, x16, y16, L._math_cached, L._act_bf16)
        hit = self.__dict__.setdefault('_descs', {}).get(key)
        if hit is not None:
            return hit                     # (shared: callers that change a field take a copy first, _desc_with)
        x16 = act_kind(self.cin) if x16 is None else x16
        y16 = act_kind(self.cout) if y16 is None else y16
        # a three-plane tensor is 6 bytes per element behind ONE 32-bit buffer descriptor (include/iprgan.h: tensors < 2 GiB):
        # past 357 M elements it stays fp32 (4 bytes: the limit of the fp32 mode, 536 M) instead of failing inside a launch;
        # the kind depends on the tensor's own shape only, so producer and consumer agree
        OH, OW = self.out_hw(H, W)
        if x16 == ST_X3 and B * H * W * c4(self.cin) * 6 >= 0x7fffffff:
            x16 = ST_F32
        if y16 == ST_X3 and B * OH * OW * c4(self.cout) * 6 >= 0x7fffffff:
            y16 = ST_F32
        d = self._descs[key] = ConvDesc(B, H, W, self.cin, self.cout, self.k, self.k, self.stride, self.pad, self.outpad,
                                        int(self.transposed), self.pad_mode, self.act, float(self.slope), int(x16), int(y16))
        return d

    @property
    def is_identity_prep(self):
        """1x1 convs whose PyTorch weight already is the forward operand (rows=Cout, k=Cin); not with a bf16 input
        (the operand is then emitted as bf16)."""
        return (self.k == 1 and not self.transposed and self.cin % 32 == 0 and self.cout % 128 == 0
                and act_kind(self.cin) == ST_F32)


def conv_flops(spec, d):
    """2 * MACs of one pass (forward, backward-data or backward-weight) of the layer ``d`` describes."""
    OH, OW = (d.H, d.W) if spec.transposed else spec.out_hw(d.H, d.W)
    return 2.0 * d.B * OH * OW * spec.cin * spec.cout * spec.k * spec.k


def conv_prep(spec, d, w, sigma=None, fwd=True, bwd=False):
    wf = wb = None
    if fwd:
        wf = empty((dquery('iprgan_conv_wfwd_floats', d),), w)
    if bwd:
        wb = empty((dquery('iprgan_conv_wbwd_floats', d),), w)
    call('iprgan_conv_weight_prep', C.byref(d), ptr(w), ptr(sigma), ptr(wf), ptr(wb), stream())
    return wf, wb


def conv_prep_multi(specs, weights, sigmas, bwd=False):
    """Prepared operands (forward ones, or backward-data ones with bwd=True) of several layers in one launch.
    sigmas[i] is that layer's device scalar or None."""
    n = len(specs)
    descs = (ConvDesc * n)(*[sp.desc(1, 1, 1) for sp in specs])
    q = 'iprgan_conv_wbwd_floats' if bwd else 'iprgan_conv_wfwd_floats'
    outs = [empty((query(q, C.byref(descs[i])),), weights[i]) for i in range(n)]
    sig = (C.c_void_p * n)(*[ptr(s) for s in sigmas])
    tab = L.ptr_table(outs)
    call('iprgan_conv_weight_prep_multi', descs, L.ptr_table(weights), sig, None if bwd else tab,
         tab if bwd else None, n, stream())
    return outs


def conv_fwd(spec, d, x, wfwd, bias, pair=None, stats=False):
    """pair = (sigma0, sigma1): paired pass, see include/iprgan.h (rows of the two half-batches divided by their sigma).
    stats=True: also returns (partials, rows): per-tile column sums of the pre-bias accumulator for the norm layer that
    follows (include/iprgan.h: column statistics from the epilogue)."""
    OH, OW = spec.out_hw(d.H, d.W)
    if not span_ok(x):
        x, d = f32(x), _desc_with(d, x_bf16=ST_F32)
    if is16(x) != d.x_bf16:
        x = to_kind(x, d.x_bf16)
    y = empty_kind((d.B, OH, OW, c4(spec.cout)), x, d.y_bf16)
    if d.x_bf16 == ST_X3:
        d = _desc_with(d, x_pstride=pstride(x), y_pstride=0)
    nws = dquery('iprgan_conv_fwd_ws_floats', d)
    ws = scratch(nws, x) if nws else None
    p0, p1 = (ptr(pair[0]), ptr(pair[1])) if pair is not None else (None, None)
    part, rows = None, C.c_int(0)
    if stats:
        part = empty((dquery('iprgan_conv_stat_floats', d, 0),), x)
    L.acct_flops(conv_flops(spec, d))
    call('iprgan_conv_fwd', C.byref(d), ptr(x), ptr(wfwd), ptr(bias), ptr(y), ptr(ws), p0, p1, ptr(part),
         C.byref(rows), stream())
    y = _out(y)
    return (y, (part, rows.value)) if stats else y


def conv_bwd_data(spec, d, dy, wbwd, prev_out=None, prev_act=L.ACT_NONE, prev_slope=0.0, pair=None, colsums=False,
                  residual=None):
    """colsums=True: also returns (partials, rows): per-tile column sums of dx (after the fused activation derivative),
    i.e. the bias gradient of the layer that produced this layer's input, up to ``colsum_partials``."""
    if not span_ok(dy):
        dy, d = f32(dy), _desc_with(d, y_bf16=ST_F32)
    if is16(dy) != d.y_bf16:
        dy = to_kind(dy, d.y_bf16)
    dx = empty_kind((d.B, d.H, d.W, c4(spec.cin)), dy, d.x_bf16)
    if prev_out is not None and is16(prev_out) != d.x_bf16:
        if ST_X3 not in (is16(prev_out), d.x_bf16):
            raise RuntimeError('conv_bwd_data: prev_out must have the storage type of the layer input')
        prev_out = to_kind(prev_out, d.x_bf16)
    if residual is not None and is16(residual) != d.x_bf16:
        residual = to_kind(residual, d.x_bf16)
    if d.y_bf16 == ST_X3:
        d = _desc_with(d, y_pstride=pstride(dy), x_pstride=0)
    nws = dquery('iprgan_conv_bwd_data_ws_floats', d)
    ws = scratch(nws, dy) if nws else None
    p0, p1 = (ptr(pair[0]), ptr(pair[1])) if pair is not None else (None, None)
    part, rows = None, C.c_int(0)
    if colsums:
        part = empty((dquery('iprgan_conv_stat_floats', d, 1),), dy)
    L.acct_flops(conv_flops(spec, d))
    call('iprgan_conv_bwd_data', C.byref(d), ptr(dy), ptr(wbwd), ptr(dx), ptr(ws), ptr(prev_out), prev_act,
         float(prev_slope), p0, p1, ptr(part), C.byref(rows), ptr(residual), stream())
    dx = _out(dx)
    return (dx, (part, rows.value)) if colsums else dx


def conv_bwd_data_bn_ok(d):
    return bool(dquery('iprgan_conv_bwd_data_bn_ok', d))


def conv_bwd_data_bn(spec, d, dy, wbwd, bn_x, mean, invstd, gamma, beta, act, slope=0.0):
    """Backward-data into a BatchNorm (+ReLU / LeakyReLU): returns (dz, (partials, rows)) - the gradient w.r.t. the norm
    layer's output with the activation derivative applied, and the per-tile sums (sum dz, sum dz * xhat) that
    ``bn_bwd_pre`` finishes the norm backward from (include/iprgan.h: iprgan_conv_bwd_data_bn)."""
    if is16(dy) != d.y_bf16:
        dy = cast(dy, torch.bfloat16 if d.y_bf16 else torch.float32)
    if is16(bn_x) != d.x_bf16:
        raise RuntimeError('conv_bwd_data_bn: the norm input must have the storage type of the layer input')
    dz = empty((d.B, d.H, d.W, c4(spec.cin)), dy, torch.bfloat16 if d.x_bf16 else torch.float32)
    part, rows = empty((dquery('iprgan_conv_stat_floats', d, 1),), dy), C.c_int(0)
    L.acct_flops(conv_flops(spec, d))
    call('iprgan_conv_bwd_data_bn', C.byref(d), ptr(dy), ptr(wbwd), ptr(dz), ptr(bn_x), ptr(mean), ptr(invstd), ptr(gamma),
         ptr(beta), act, float(slope), ptr(part), C.byref(rows), stream())
    return dz, (part, rows.value)


def colsum_partials(part, rows, Cs, channels, out=None, beta=0.0):
    res = empty((channels,), part) if out is None else out
    call('iprgan_colsum_partials', ptr(part), int(rows), int(Cs), int(channels), ptr(res), float(beta), stream())
    return res


def colsum_partials_flush(pending):
    """The bias gradients that ``pending`` = [(partials, rows, Cs, channels, out, beta)] owe, in one launch (bit-identical to
    one colsum_partials call each); the partials are kept alive by the list until the launch is enqueued."""
    n = len(pending)
    if not n:
        return
    if n == 1:
        colsum_partials(*pending[0][:4], out=pending[0][4], beta=pending[0][5])
    else:
        call('iprgan_colsum_partials_multi', L.ptr_table([p_[0] for p_ in pending]), _int_table([int(p_[1]) for p_ in pending]),
             _int_table([int(p_[2]) for p_ in pending]), _int_table([int(p_[3]) for p_ in pending]),
             L.ptr_table([p_[4] for p_ in pending]), (C.c_float * n)(*[float(p_[5]) for p_ in pending]), n, stream())
    del pending[:]


def colsum(x2d_like, channels, out=None, beta=0.0):
    """Column sums of an activation tensor [..., C4] over all leading dims -> [channels] (bias gradient)."""
    if is16(x2d_like) == ST_X3 and pstride(x2d_like) != x2d_like.numel():     # a batch slice of a three-plane tensor
        x2d_like = f32(x2d_like)
    C_ = x2d_like.shape[-1]
    M = x2d_like.numel() // C_
    res = empty((channels,), x2d_like) if out is None else out
    ws = scratch(query('iprgan_colsum_ws_floats', M, C_), x2d_like)
    call('iprgan_colsum', ptr(x2d_like), ptr(res), ptr(ws), M, C_, channels, float(beta), is16(x2d_like), stream())
    return res


_DEFER_WGRAD = os.environ.get('IPRGAN_DEFER_WGRAD_REDUCE', '1') != '0'     # A/B switch: one slab reduce launch per backward pass
_WGRAD_PENDING_BYTES = int(os.environ.get('IPRGAN_WGRAD_PENDING_MB', '6144')) << 20


def wgrad_reduce_flush(pending):
    """The slab reduces that ``conv_bwd_weight(..., defer=pending)`` calls of a backward pass owe, in ONE launch per 24 layers
    (iprgan_wgrad_reduce_multi; bit-identical to the per-layer launches).  ``pending`` = [(record, workspace, dw)]: the
    tensors are kept alive until the launch is enqueued."""
    if not pending:
        return
    n = len(pending)
    arr = (L.WGradReduceRec * n)(*[p[0] for p in pending])
    call('iprgan_wgrad_reduce_multi', arr, n, stream())
    del pending[:]


def conv_bwd_weight(spec, d, x, dy, w_shape, want_bias, dw=None, db=None, beta=0.0, defer=None):
    """dw, db given (gradient-bucket views): ``dw = beta*dw + grad`` written in place, no temporary.
    bf16 x / dy are consumed directly where the 128x128 bf16 tile applies, through fp32 copies elsewhere.
    ``defer`` (a list): the slab reduce into dw is NOT launched - it is appended to the list and runs with the other layers'
    in ``wgrad_reduce_flush(defer)``; until then dw is not valid."""
    d = ConvDesc.from_buffer_copy(d)
    if not (span_ok(x) and span_ok(dy)):
        x, dy = f32(x), f32(dy)
    if ST_X3 in (is16(x), is16(dy)) and is16(x) != is16(dy):
        if c4(spec.cin) % 32 == 0 and c4(spec.cout) % 32 == 0 and ST_BF16 not in (is16(x), is16(dy)):
            x, dy = to_kind(x, ST_X3), to_kind(dy, ST_X3)      # (a gradient that arrived as fp32: split, not the other side joined)
        # (else RGB stems / heads: one side is an fp32 image, the kernel widens the three-plane side on arrival)
    d.x_bf16, d.y_bf16 = is16(x), is16(dy)            # the kernels read either storage type (include/iprgan.h)
    d.x_pstride, d.y_pstride = pstride(x), pstride(dy)
    if (d.x_bf16 or d.y_bf16 == ST_X3) and not dquery('iprgan_conv_wgrad_takes_bf16', d):
        x, dy = (f32(x), f32(dy)) if ST_X3 in (d.x_bf16, d.y_bf16) else (cast(x, torch.float32), dy)
        d.x_bf16, d.y_bf16 = is16(x), is16(dy)
        d.x_pstride = d.y_pstride = 0
    if dw is None:
        dw = empty(tuple(w_shape), x)
    if db is None and want_bias:
        db = empty((spec.cout,), x)
    ws = scratch(dquery('iprgan_conv_wgrad_ws_floats', d), x)
    L.acct_flops(conv_flops(spec, d))
    if defer is not None and _DEFER_WGRAD:
        rec = L.WGradReduceRec()
        call('iprgan_conv_bwd_weight_deferred', C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db) if want_bias else None, ptr(ws),
             float(beta), stream(), C.byref(rec))
        if rec.pending:
            # one destination per multi-launch (two records adding into the same dw view would race), and a bound on the slab
            # workspaces a pass keeps alive until its flush (IPRGAN_WGRAD_PENDING_MB, default 6144 of the 288 GB: a workspace is
            # sized for the layer's LARGEST candidate - 0.1-0.3 GB per DCGAN layer, a discriminator pass of DCGAN-64 holds
            # ~2.5 GB; 1024 cut that pass into six flushes instead of its two bucket boundaries); ADVICE r05
            if any(p_[2].data_ptr() == dw.data_ptr() for p_ in defer) or \
                    sum(p_[1].numel() for p_ in defer) * 4 + ws.numel() * 4 > _WGRAD_PENDING_BYTES:
                wgrad_reduce_flush(defer)
            defer.append((rec, ws, dw))
        return dw, (db if want_bias else None)
    call('iprgan_conv_bwd_weight', C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db) if want_bias else None, ptr(ws),
         float(beta), stream())
    return dw, (db if want_bias else None)


# ---- Linear feeding an NHWC map (csrc/fc.hip) ------------------------------------------------------------
_FC_DIRECT = os.environ.get('IPRGAN_FC_DIRECT', '1') != '0'      # A/B switch: 0 = the layer as a 1x1 convolution (rounds 1-5)


def fc_nhwc_ok(B, K, C, HW):
    return _FC_DIRECT and bool(query('iprgan_fc_nhwc_ok', int(B), int(K), int(C), int(HW)))


def fc_nhwc_fwd(x, w, bias, C, HW, act, slope=0.0):
    """y[b, hw*C + c] = act(x @ w[c*HW + hw] + bias[c*HW + hw]): [B, HW*C] in the storage kind of a C-channel activation."""
    B, K = x.shape
    k = act_kind(C)
    if k == ST_X3 and B * C * HW * 6 >= 0x7fffffff:
        k = ST_F32
    y = empty_kind((B, C * HW), x, k)
    L.acct_flops(2.0 * B * K * C * HW)
    call('iprgan_fc_nhwc_fwd', ptr32(x), ptr32(w), ptr32(bias), ptr(y), B, K, C, HW, act, float(slope), k, pstride(y), stream())
    return y


def fc_nhwc_bwd(x, y, dy, w_shape, C, HW, act, slope=0.0, dw=None, db=None, beta=0.0, want_bias=True):
    """(dw, db) of fc_nhwc_fwd in PyTorch layout; dw / db given (gradient-bucket views): accumulated in place with ``beta``."""
    B, K = x.shape
    k = is16(y)
    if is16(dy) != k:
        dy = to_kind(f32(dy), k)
    if dw is None:
        dw = empty(tuple(w_shape), x)
    if db is None and want_bias:
        db = empty((C * HW,), x)
    L.acct_flops(2.0 * B * K * C * HW)
    call('iprgan_fc_nhwc_bwd', ptr32(x), ptr(y), ptr(dy), ptr32(dw), ptr32(db) if want_bias else None, B, K, C, HW, act,
         float(slope), k, pstride(y), pstride(dy), float(beta), stream())
    return dw, (db if want_bias else None)


# ---- GEMV head ----------------------------------------------------------------------------------
def gemv_fwd(x2d, w, bias, sigma, out=None):
    B, K = x2d.shape
    y = empty((B,), x2d) if out is None else out
    call('iprgan_gemv_fwd', ptr(x2d), ptr(w), ptr(bias), ptr(sigma), ptr(y), B, K, is16(x2d), pstride(x2d), stream())
    return y


def gemv_bwd(x2d, w, dy, sigma, need_dx, need_dw, prev_out=None, prev_act=L.ACT_NONE, prev_slope=0.0, dx_out=None):
    """prev_out, when given, is x2d itself (the fused derivative's operand is this layer's input)."""
    B, K = x2d.shape
    dx = (_empty_like(x2d) if dx_out is None else dx_out) if need_dx else None
    if dx is not None and is16(dx) != is16(x2d):
        raise RuntimeError('gemv_bwd: dx must have the storage kind of x')
    dw = empty((K,), x2d) if need_dw else None
    db = empty((1,), x2d) if need_dw else None
    call('iprgan_gemv_bwd', ptr(x2d), ptr(w), ptr(dy), ptr(sigma), ptr(dx), ptr(dw), ptr(db),
         ptr(prev_out), prev_act, float(prev_slope), B, K, is16(x2d), pstride(x2d), pstride(dx) if dx is not None else 0, stream())
    return dx, dw, db


def gemv_fwd_pair(x2d, w, bias, sigma0, sigma1):
    """gemv_fwd of the two half-batches of a paired pass (rows [:B/2] / [B/2:] divided by sigma0 / sigma1) in one launch."""
    B, K = x2d.shape
    y = empty((B,), x2d)
    call('iprgan_gemv_fwd_pair', ptr(x2d), ptr(w), ptr(bias), ptr(sigma0), ptr(sigma1), ptr(y), B, K, is16(x2d), pstride(x2d), stream())
    return y


def gemv_bwd_pair(x2d, w, dy, sigma0, sigma1, need_dx, need_dw, prev_out=None, prev_act=L.ACT_NONE, prev_slope=0.0):
    """gemv_bwd of both halves in one launch per kernel: (dx [B, K], dw2 [2, K], db2 [2]); three-plane x for the weight side."""
    B, K = x2d.shape
    dx = _empty_like(x2d) if need_dx else None
    dw2 = empty((2, K), x2d) if need_dw else None
    db2 = empty((2,), x2d) if need_dw else None
    call('iprgan_gemv_bwd_pair', ptr(x2d), ptr(w), ptr(dy), ptr(sigma0), ptr(sigma1), ptr(dx), ptr(dw2), ptr(db2), ptr(prev_out),
         prev_act, float(prev_slope), B, K, is16(x2d), pstride(x2d), pstride(dx) if dx is not None else 0, stream())
    return dx, dw2, db2


# ---- batch norm -----------------------------------------------------------------------------------
ST_X3_XF32 = 3          # norm entry points only (include/iprgan.h): x fp32; y, dy, dx, residual three planes
_NORM_XF32 = os.environ.get('IPRGAN_NORM_XF32', '1') != '0'
_NORM_DYF32 = os.environ.get('IPRGAN_NORM_DYF32', '1') != '0'     # A/B switch: fp32 gradients on the conv -> norm edges of the backward pass


def _whole3(t):
    """A three-plane tensor whose planes are NOT its own element count apart (a batch slice of a larger tensor keeps the
    parent's plane stride) re-packed into a contiguous one: the norm kernels address the planes at G*M*C (csrc/norm.hip),
    so a sliced operand would have its m and l planes read / written at the wrong offsets."""
    if t is not None and is16(t) == ST_X3 and pstride(t) != t.numel():
        return to_kind(f32(t), ST_X3)
    return t


ST_X3_XDF32 = 4         # backward norm entry points only: x and dy fp32, dx (and y) three planes


def _norm_splits_here(x):
    """True when this fp32 norm input belongs to a layer whose output is three planes (the norm layer is where the tensor is
    split): 'fp32x3' mode, an NHWC tensor whose channel count allows planes, under the 2 GiB size rule of ConvSpec.desc."""
    return (_NORM_XF32 and L.act_x3() and x.dim() == 4 and x.shape[-1] % 32 == 0 and x.numel() * 6 < 0x7fffffff)


def _norm_fwd_kinds(x):
    """(storage-kind argument of the norm entry point, kind of its output y).  In 'fp32x3' mode an fp32 input whose channel
    count allows three planes gets a three-plane output: the convolution in front of a norm layer writes fp32 (4 instead
    of 6 bytes per element; engine.Conv: y_f32) and the norm layer is where the tensor is split."""
    k = is16(x)
    if k == ST_F32 and _norm_splits_here(x):
        return ST_X3_XF32, ST_X3
    return k, k


def _norm_bwd_kinds(x, dy, y=None):
    """(storage-kind argument, dy, y) of a norm backward: the saved fp32 x of a layer whose output is three planes stays
    fp32 (ST_X3_XF32); otherwise dy (and y) follow x."""
    x_, dy, y = x, _whole3(dy), _whole3(y)
    k = is16(x)
    if k == ST_X3 and pstride(x_) != x_.numel():
        raise RuntimeError('norm backward: the saved input is a batch slice of a three-plane tensor (plane stride != element count)')
    if k == ST_F32 and is16(dy) == ST_X3:
        if y is not None and is16(y) != ST_X3:
            y = to_kind(y, ST_X3)
        return ST_X3_XF32, dy, y
    if k == ST_F32 and is16(dy) == ST_F32 and _NORM_DYF32 and _norm_splits_here(x) and dy.shape == x.shape:
        # the backward-data pass above wrote an fp32 gradient (engine.Conv.backward: dx_f32): read it as it is, emit dx as
        # three planes for the convolution below
        if y is not None and is16(y) != ST_X3:
            y = to_kind(y, ST_X3)
        return ST_X3_XDF32, dy, y
    if is16(dy) != k:
        dy = to_kind(dy, k)
    if y is not None and is16(y) != k:
        y = to_kind(y, k)
    return k, dy, y


def bn_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, training, act, slope=0.0, conv_stats=None,
           conv_bias=None, counter=None, residual=None):
    """conv_stats = (partials, rows) from conv_fwd(stats=True): the statistics are taken from them instead of a pass
    over x.  counter: the module's int64 num_batches_tracked, incremented on the device."""
    C_ = x.shape[-1]
    M = x.numel() // C_
    x = _whole3(x)
    st, ky = _norm_fwd_kinds(x)
    y = empty_kind(x.shape, x, ky)
    if residual is not None and is16(residual) != ky:
        residual = to_kind(residual, ky)
    residual = _whole3(residual)
    mean, invstd = empty((C_,), x), empty((C_,), x)
    part, rows = conv_stats if conv_stats is not None else (None, 0)
    ws = None if part is not None else scratch(query('iprgan_bn_ws_floats', M, C_), x)
    call('iprgan_bn_fwd', ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
         ptr(mean), ptr(invstd), ptr(ws), M, C_, float(eps), float(momentum), 0 if training else 1,
         act, float(slope), ptr(part), int(rows), ptr(conv_bias) if part is not None else None,
         counter.data_ptr() if counter is not None else None, ptr(residual), st, stream())
    return _out(y), mean, invstd


def bn_bwd(x, y, dy, gamma, mean, invstd, act, slope=0.0, beta=None, dbias=None, dbias_beta=0.0):
    """dbias: a tensor that receives (beta-accumulates) the column sums of dx = the bias gradient of the convolution
    that produced x."""
    C_ = x.shape[-1]
    M = x.numel() // C_
    st, dy, y = _norm_bwd_kinds(x, dy, y)
    dx = empty_kind(dy.shape, dy, ST_X3) if st == ST_X3_XDF32 else _empty_like(dy)
    dgamma, dbeta = empty((C_,), x), empty((C_,), x)
    ws = scratch(query('iprgan_bn_ws_floats', M, C_), x)
    call('iprgan_bn_bwd', ptr(x), ptr(y), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dx),
         ptr(dgamma), ptr(dbeta), ptr(ws), M, C_, act, float(slope), ptr(dbias), dbias.numel() if dbias is not None else 0,
         float(dbias_beta), st, stream())
    return _out(dx), dgamma, dbeta


def bn_prelu_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, training, slope_t, conv_stats=None,
                 conv_bias=None, counter=None, residual=None):
    """BatchNorm + PReLU (``slope_t``: the PReLU parameter, one float on the device) in the norm's own passes."""
    C_ = x.shape[-1]
    M = x.numel() // C_
    x = _whole3(x)
    st, ky = _norm_fwd_kinds(x)
    y = empty_kind(x.shape, x, ky)
    if residual is not None and is16(residual) != ky:
        residual = to_kind(residual, ky)
    residual = _whole3(residual)
    mean, invstd = empty((C_,), x), empty((C_,), x)
    part, rows = conv_stats if conv_stats is not None else (None, 0)
    ws = None if part is not None else scratch(query('iprgan_bn_ws_floats', M, C_), x)
    call('iprgan_bn_prelu_fwd', ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(mean),
         ptr(invstd), ptr(ws), M, C_, float(eps), float(momentum), 0 if training else 1, ptr(slope_t), ptr(part), int(rows),
         ptr(conv_bias) if part is not None else None, counter.data_ptr() if counter is not None else None, ptr(residual),
         st, stream())
    return _out(y), mean, invstd


def bn_prelu_bwd(x, dy, gamma, beta, mean, invstd, slope_t, dbias=None, dbias_beta=0.0):
    C_ = x.shape[-1]
    M = x.numel() // C_
    st, dy, _ = _norm_bwd_kinds(x, dy)
    dx = empty_kind(dy.shape, dy, ST_X3) if st == ST_X3_XDF32 else _empty_like(dy)
    dgamma, dbeta, dslope = empty((C_,), x), empty((C_,), x), empty((1,), x)
    ws = scratch(query('iprgan_bn_ws_floats', M, C_), x)
    call('iprgan_bn_prelu_bwd', ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(slope_t), ptr(dx),
         ptr(dgamma), ptr(dbeta), ptr(dslope), ptr(ws), M, C_, ptr(dbias), dbias.numel() if dbias is not None else 0,
         float(dbias_beta), st, stream())
    return _out(dx), dgamma, dbeta, dslope


def bn_bwd_pre(x, dz, gamma, mean, invstd, partials, dbias=None, dbias_beta=0.0):
    """The norm backward from ``conv_bwd_data_bn``'s outputs: dz (activation derivative applied) and its per-tile sums."""
    C_ = x.shape[-1]
    M = x.numel() // C_
    part, rows = partials
    dx = _empty_like(x)
    dgamma, dbeta = empty((C_,), x), empty((C_,), x)
    ws = scratch(query('iprgan_bn_ws_floats', M, C_), x)
    call('iprgan_bn_bwd_pre', ptr(x), ptr(dz), ptr(gamma), ptr(mean), ptr(invstd), ptr(part), int(rows), ptr(dx),
         ptr(dgamma), ptr(dbeta), ptr(ws), M, C_, ptr(dbias), dbias.numel() if dbias is not None else 0,
         float(dbias_beta), is16(x), stream())
    return dx, dgamma, dbeta


# ---- spectral norm ----------------------------------------------------------------------------------
def sn_power_iter(w_orig, u, v, training, eps=1e-12):
    rows = w_orig.shape[0]
    cols = w_orig.numel() // rows
    sigma = empty((1,), w_orig)
    ws = scratch(query('iprgan_sn_ws_floats', rows, cols), w_orig)
    call('iprgan_sn_power_iter', ptr(w_orig), ptr(u), ptr(v), ptr(sigma), ptr(ws), rows, cols,
         float(eps), 1 if training else 0, stream())
    return sigma


def sn_power_iter_multi(weights, us, vs, training, eps=1e-12):
    """One power iteration for every (weight_orig, u, v) triple in 4 launches.  Returns (sigmas [n] device
    tensor, u copies, v copies): the copies are this pass's vectors for its backward."""
    n = len(weights)
    rows = [w.shape[0] for w in weights]
    cols = [w.numel() // w.shape[0] for w in weights]
    sig = empty((n,), weights[0])
    u_out = [_empty_like(u) for u in us]
    v_out = [_empty_like(v) for v in vs]
    r, c = _int_table(rows), _int_table(cols)
    ws = scratch(query('iprgan_sn_multi_ws_floats', r, c, n), weights[0])
    call('iprgan_sn_power_iter_multi', L.ptr_table(weights), L.ptr_table(us), L.ptr_table(vs),
         L.ptr_table(u_out), L.ptr_table(v_out), ptr(sig), ptr(ws), r, c, n, float(eps),
         1 if training else 0, stream())
    return sig, u_out, v_out


def sn_bwd(dwsn, w_orig, u, v, sigma):
    rows = w_orig.shape[0]
    cols = w_orig.numel() // rows
    dw = _empty_like(w_orig)
    ws = scratch(query('iprgan_sn_ws_floats', rows, cols), w_orig)
    call('iprgan_sn_bwd', ptr(dwsn), ptr(w_orig), ptr(u), ptr(v), ptr(sigma), ptr(dw), ptr(ws), rows,
         cols, stream())
    return dw


def sn_bwd_multi(dwsns, weights, us, vs, sigmas, outs=None, beta=0.0):
    """Spectral-norm backward of several layers in two launches; returns the list of dW_orig
    (``outs[i] = beta*outs[i] + dW_orig`` when given)."""
    n = len(weights)
    rows = [w.shape[0] for w in weights]
    cols = [w.numel() // w.shape[0] for w in weights]
    dws = [_empty_like(w) for w in weights] if outs is None else outs
    ws = scratch(64 * 16, weights[0])
    call('iprgan_sn_bwd_multi', L.ptr_table(dwsns), L.ptr_table(weights), L.ptr_table(us), L.ptr_table(vs),
         L.ptr_table(sigmas), L.ptr_table(dws), ptr(ws), _int_table(rows), _int_table(cols), n, float(beta), stream())
    return dws


# ---- losses ---------------------------------------------------------------------------------------
def loss_fwd(kind, x, y=None):
    out = empty((), x)
    ws = scratch(query('iprgan_loss_ws_floats', x.numel()), x)
    call('iprgan_loss_fwd', kind, ptr(x), ptr(y), ptr(out), ptr(ws), x.numel(), stream())
    return out


def loss_bwd(kind, x, y, gscale):
    dx = _empty_like(x)
    call('iprgan_loss_bwd', kind, ptr(x), ptr(y), ptr(gscale), ptr(dx), x.numel(), stream())
    return dx


def loss_pair_fwd(kind_a, kind_b, x, n_half):
    """[mean_a(x[:n]), mean_b(x[n:]), their sum] as one 3-float tensor (n <= 256 per half)."""
    out = empty((3,), x)
    call('iprgan_loss_pair_fwd', kind_a, kind_b, ptr(x), int(n_half), ptr(out), stream())
    return out


def loss_pair_bwd(kind_a, kind_b, x, n_half, gscale):
    dx = _empty_like(x)
    call('iprgan_loss_pair_bwd', kind_a, kind_b, ptr(x), ptr(gscale), ptr(dx), int(n_half), stream())
    return dx


def loss_sum_fwd(kind, x, y, scale):
    """scale * sum(term): the reduction='sum' / N losses of models/vae.py:36-48."""
    out = empty((), x)
    ws = scratch(query('iprgan_loss_ws_floats', x.numel()), x)
    call('iprgan_loss_sum_fwd', kind, ptr(x), ptr(y), ptr(out), ptr(ws), x.numel(), float(scale), stream())
    return out


def loss_sum_bwd(kind, x, y, gscale, scale):
    dx = _empty_like(x)
    call('iprgan_loss_sum_bwd', kind, ptr(x), ptr(y), ptr(gscale), ptr(dx), x.numel(), float(scale), stream())
    return dx


# ---- SSIM loss (black-box watermark objective) --------------------------------------------------------
def ssim_fwd(x, y, denorm, want_grad):
    """x, y: contiguous NCHW fp32.  Returns (loss = 1 - mean SSIM, gmaps for the backward pass or None)."""
    B, Cc, H, W = x.shape
    planes = B * Cc
    out = empty((), x)
    ws = scratch(query('iprgan_ssim_ws_floats', planes, H, W), x)
    gm = empty((query('iprgan_ssim_gmap_floats', planes, H, W),), x) if want_grad else None
    call('iprgan_ssim_fwd', ptr(x), ptr(y), ptr(out), ptr(gm), ptr(ws), planes, H, W, int(bool(denorm)), stream())
    return out, gm


def ssim_bwd(x, y, gmaps, gscale, denorm):
    B, Cc, H, W = x.shape
    dx = _empty_like(x)
    call('iprgan_ssim_bwd', ptr(x), ptr(y), ptr(gmaps), ptr(gscale), ptr(dx), B * Cc, H, W, int(bool(denorm)),
         stream())
    return dx


def msssim_fwd(x, y, denorm, want_grad):
    """x, y: contiguous NCHW fp32, min(H, W) > 160.  Returns (loss = 1 - MS-SSIM, state for msssim_bwd or None)."""
    B, Cc, H, W = x.shape
    planes = B * Cc
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    call('iprgan_msssim_sizes', planes, H, W, C.byref(a), C.byref(b), C.byref(c))
    out = empty((), x)
    pyr, small = empty((max(1, a.value),), x), empty((c.value,), x)
    gm = empty((b.value,), x) if want_grad else None
    call('iprgan_msssim_fwd', ptr(x), ptr(y), ptr(out), ptr(pyr), ptr(gm), ptr(small), planes, H, W, int(bool(denorm)),
         1 if want_grad else 0, stream())
    return out, ((pyr, gm, small) if want_grad else None)


def msssim_bwd(x, y, state, gscale, denorm):
    B, Cc, H, W = x.shape
    pyr, gm, small = state
    dx = _empty_like(x)
    ws = scratch(2 * B * Cc * ((H + 1) // 2) * ((W + 1) // 2), x)
    call('iprgan_msssim_bwd', ptr(x), ptr(y), ptr(pyr), ptr(gm), ptr(small), ptr(gscale), ptr(dx), ptr(ws), B * Cc, H, W,
         int(bool(denorm)), stream())
    return dx


# ---- VAE reparameterisation -----------------------------------------------------------------------
def reparam_fwd(mean, logvar, eps):
    z = _empty_like(mean)
    call('iprgan_reparam_fwd', ptr(mean), ptr(logvar), ptr(eps), ptr(z), mean.numel(), stream())
    return z


def reparam_bwd(dz, logvar, eps):
    dmean, dlogvar = _empty_like(dz), _empty_like(dz)
    call('iprgan_reparam_bwd', ptr(dz), ptr(logvar), ptr(eps), ptr(dmean), ptr(dlogvar), dz.numel(), stream())
    return dmean, dlogvar


# ---- sign loss ------------------------------------------------------------------------------------


def sign_loss_fwd(gammas, signs, gamma0):
    out = empty((), gammas[0])
    call('iprgan_sign_loss_fwd', L.ptr_table(gammas), L.ptr_table(signs),
         _int_table([g.numel() for g in gammas]), len(gammas), float(gamma0), ptr(out), stream())
    return out


def sign_loss_bwd(gammas, signs, gamma0, gscale, outs=None, beta=0.0):
    grads = [_empty_like(g) for g in gammas] if outs is None else outs
    call('iprgan_sign_loss_bwd', L.ptr_table(gammas), L.ptr_table(signs), L.ptr_table(grads),
         _int_table([g.numel() for g in gammas]), len(gammas), float(gamma0), ptr(gscale), float(beta), stream())
    return grads


def sign_ber_counts(gammas, signs):
    counts = torch.empty((2,), dtype=torch.int64, device=gammas[0].device)
    call('iprgan_sign_ber', L.ptr_table(gammas), L.ptr_table(signs),
         _int_table([g.numel() for g in gammas]), len(gammas), counts.data_ptr(), stream())
    return counts


# ---- Adam -----------------------------------------------------------------------------------------
def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    n = len(params)
    sizes = (C.c_longlong * n)(*[p.numel() for p in params])
    call('iprgan_adam_step', L.ptr_table(params), L.ptr_table(grads), L.ptr_table(exp_avg),
         L.ptr_table(exp_avg_sq), sizes, n, float(lr), float(beta1), float(beta2), float(eps),
         float(weight_decay), int(step), float(grad_scale), stream())


def adam_step_tables(ptab, grads, mtab, vtab, sizes, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """adam_step with the parameter / moment pointer tables and sizes built once by the caller (optim.Adam)."""
    if L.acct_on():         # parameters and both moments are read and written (their tables are cached: not seen by ptr())
        L.acct_bytes(24 * sum(sizes[i] for i in range(n)))
    call('iprgan_adam_step', ptab, L.ptr_table(grads), mtab, vtab, sizes, n, float(lr), float(beta1), float(beta2),
         float(eps), float(weight_decay), int(step), float(grad_scale), stream())


def adam_step_tables_dev(ptab, grads, mtab, vtab, sizes, n, lr, beta1, beta2, eps, weight_decay, step_dev, coef,
                         grad_scale=1.0):
    """adam_step_tables with the step count on the device (``step_dev`` int32[1] is incremented by the call; ``coef``
    float32[2] receives the bias corrections): nothing in the launch depends on the host's step number, so a captured
    HIP graph can replay it (graphs.py)."""
    if L.acct_on():
        L.acct_bytes(24 * sum(sizes[i] for i in range(n)))
    call('iprgan_adam_step_dev', ptab, L.ptr_table(grads), mtab, vtab, sizes, n, float(lr), float(beta1), float(beta2),
         float(eps), float(weight_decay), C.c_void_p(step_dev.data_ptr()), ptr(coef), float(grad_scale), stream())


def axpy_multi(dsts, srcs, alpha=1.0):
    """dsts[i] += alpha * srcs[i] for a list of tensors in one launch."""
    n = len(dsts)
    if not n:
        return
    sizes = (C.c_longlong * n)(*[t.numel() for t in dsts])
    call('iprgan_axpy_multi', L.ptr_table(dsts), L.ptr_table(srcs), sizes, n, float(alpha), stream())


def fill(t, value=0.0):
    call('iprgan_fill', ptr(t), float(value), t.numel(), stream())


# ---- instance norm / PReLU / pixel shuffle / max pool / residual add ---------------------------------
def instnorm_fwd(x, gamma, beta, eps, act, slope=0.0, conv_stats=None, conv_bias=None, residual=None):
    B, H, W, C_ = x.shape
    x = _whole3(x)
    st, ky = _norm_fwd_kinds(x)
    y = empty_kind(x.shape, x, ky)
    if residual is not None and is16(residual) != ky:
        residual = to_kind(residual, ky)
    residual = _whole3(residual)
    mean, invstd = empty((B, C_), x), empty((B, C_), x)
    part, rows = conv_stats if conv_stats is not None else (None, 0)
    ws = None if part is not None else scratch(query('iprgan_instnorm_ws_floats', B, H * W, C_), x)
    call('iprgan_instnorm_fwd', ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(ws),
         B, H * W, C_, float(eps), act, float(slope), ptr(part), int(rows),
         ptr(conv_bias) if part is not None else None, ptr(residual), st, stream())
    return _out(y), mean, invstd


def instnorm_bwd(x, y, dy, gamma, mean, invstd, act, slope=0.0, beta=None, dbias=None, dbias_beta=0.0):
    B, H, W, C_ = x.shape
    st, dy, y = _norm_bwd_kinds(x, dy, y)
    dx = empty_kind(dy.shape, dy, ST_X3) if st == ST_X3_XDF32 else _empty_like(dy)
    dgamma = empty((C_,), x) if gamma is not None else None
    dbeta = empty((C_,), x) if gamma is not None else None
    ws = scratch(query('iprgan_instnorm_ws_floats', B, H * W, C_), x)
    call('iprgan_instnorm_bwd', ptr(x), ptr(y), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dx),
         ptr(dgamma), ptr(dbeta), ptr(ws), B, H * W, C_, act, float(slope), ptr(dbias),
         dbias.numel() if dbias is not None else 0, float(dbias_beta), st, stream())
    return _out(dx), dgamma, dbeta


def _same_kind(*ts):
    """The tensors of one elementwise call in ONE storage kind (fp32 or, when every one of them is, three planes)."""
    ks = {is16(t) for t in ts}
    if ks == {ST_X3} and all(pstride(t) == t.numel() for t in ts):
        return ST_X3, ts
    return ST_F32, tuple(f32(t) for t in ts)


def prelu_fwd(x, alpha):
    st, (x,) = _same_kind(x)
    y = _empty_like(x)
    call('iprgan_prelu_fwd', ptr(x), ptr(alpha), ptr(y), x.numel(), st, stream())
    return y


def prelu_bwd(x, dy, alpha):
    st, (x, dy) = _same_kind(x, dy)
    dx = _empty_like(x)
    dalpha = empty((1,), x)
    ws = scratch(query('iprgan_loss_ws_floats', x.numel()), x)
    call('iprgan_prelu_bwd', ptr(x), ptr(dy), ptr(alpha), ptr(dx), ptr(dalpha), ptr(ws), x.numel(), st, stream())
    return dx, dalpha


def pixel_shuffle2(x, inverse=False):
    """forward: [B,H,W,4C] -> [B,2H,2W,C]; inverse: [B,2H,2W,C] -> [B,H,W,4C]."""
    x = f32(x) if L.act_x3() else x
    _f32(x)
    if not inverse:
        B, H, W, C4_ = x.shape
        Cc = C4_ // 4
        y = empty((B, 2 * H, 2 * W, Cc), x)
    else:
        B, H2, W2, Cc = x.shape
        H, W = H2 // 2, W2 // 2
        y = empty((B, H, W, 4 * Cc), x)
    call('iprgan_pixel_shuffle2', ptr(x), ptr(y), B, H, W, Cc, 1 if inverse else 0, stream())
    return y


def pixel_shuffle2_prelu_fwd(x, alpha):
    """prelu(pixel_shuffle(x, 2)) in one pass: [B,H,W,4C] -> [B,2H,2W,C] (C % 4 == 0)."""
    st, (x,) = _same_kind(x)
    B, H, W, C4_ = x.shape
    Cc = C4_ // 4
    y = empty_kind((B, 2 * H, 2 * W, Cc), x, st)
    call('iprgan_pixel_shuffle2_prelu_fwd', ptr(x), ptr(alpha), ptr(y), B, H, W, Cc, st, stream())
    return y


def pixel_shuffle2_prelu_bwd(x, dy, alpha):
    st, (x, dy) = _same_kind(x, dy)
    B, H, W, C4_ = x.shape
    Cc = C4_ // 4
    dx = _empty_like(x)
    dalpha = empty((1,), x)
    ws = scratch(query('iprgan_loss_ws_floats', x.numel()), x)
    call('iprgan_pixel_shuffle2_prelu_bwd', ptr(x), ptr(dy), ptr(alpha), ptr(dx), ptr(dalpha), ptr(ws), B, H, W, Cc, st, stream())
    return dx, dalpha


def maxpool2_fwd(x):
    st, (x,) = _same_kind(x)
    B, H, W, C_ = x.shape
    if st == ST_X3 and (H % 2 or W % 2):
        st, x = ST_F32, f32(x)
    y = empty_kind((B, H // 2, W // 2, C_), x, st)
    call('iprgan_maxpool2_fwd', ptr(x), ptr(y), B, H, W, C_, st, stream())
    return y


def maxpool2_bwd(x, dy):
    st, (x, dy) = _same_kind(x, dy)
    B, H, W, C_ = x.shape
    if st == ST_X3 and (H % 2 or W % 2):
        st, x, dy = ST_F32, f32(x), f32(dy)
    dx = _empty_like(x)
    call('iprgan_maxpool2_bwd', ptr(x), ptr(dy), ptr(dx), B, H, W, C_, st, stream())
    return dx


def add(a, b):
    st, (a, b) = _same_kind(a, b)
    out = _empty_like(a)
    call('iprgan_add', ptr(a), ptr(b), ptr(out), a.numel(), st, stream())
    return out


def write_ints(dst, values):
    """A short list of Python ints into the int32 device tensor ``dst`` in stream order (travels as kernel arguments)."""
    assert dst.is_cuda and dst.dtype == torch.int32 and dst.is_contiguous() and dst.numel() >= len(values)
    arr = (C.c_int * len(values))(*[int(v) for v in values])
    call('iprgan_write_ints', C.c_void_p(dst.data_ptr()), C.cast(arr, C.c_void_p), len(values), stream())


def pool_swap(images, pool, index, take):
    """ImagePool's swap branch (models/util.py:27-34) in place on both tensors, the draws read from device memory:
    images[i] <-> pool[index[i]] where take[i] != 0 (index: distinct int32 rows of ``pool``)."""
    assert images.is_contiguous() and pool.is_contiguous() and images.dtype == pool.dtype == torch.float32
    assert index.dtype == take.dtype == torch.int32 and index.numel() >= images.shape[0] <= take.numel()
    n = images[0].numel()
    assert pool[0].numel() == n
    assert index.is_cuda and take.is_cuda and index.is_contiguous() and take.is_contiguous()
    call('iprgan_pool_swap', ptr(images), ptr(pool), C.c_void_p(index.data_ptr()), C.c_void_p(take.data_ptr()), images.shape[0], n,
         stream())
