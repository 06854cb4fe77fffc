"""ctypes binding of libiprgan_hip.so (the C ABI declared in include/iprgan.h).

There is no CPU or ATen fallback behind these calls: if the shared library is missing,
or an op is handed a non-GPU tensor, the call raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C ipr-gan_amd/csrc`` and travels with the repo.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IPRGAN_LIB', os.path.join(_HERE, 'libiprgan_hip.so'))   # override: A/B builds

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID_PM1 = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1
(LOSS_HINGE_REAL, LOSS_HINGE_FAKE, LOSS_NEG_MEAN, LOSS_BCE_ONES, LOSS_BCE_ZEROS,
 LOSS_MSE_ONES, LOSS_MSE_ZEROS, LOSS_MSE, LOSS_L1, LOSS_BCE_PM1, LOSS_KL_MEAN, LOSS_KL_LOGVAR,
 LOSS_MSE_DENORM, LOSS_L1_DENORM) = range(14)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('B', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'pad', 'outpad',
                 'transposed', 'pad_mode', 'act')] + [('slope', C.c_float), ('x_bf16', C.c_int32), ('y_bf16', C.c_int32),
                                                     ('x_pstride', C.c_int64), ('y_pstride', C.c_int64)]


class WGradReduceRec(C.Structure):
    """iprgan_wgrad_reduce_rec (include/iprgan.h): what a deferred backward-weight call owes."""
    _fields_ = [('ws', C.c_void_p), ('dw', C.c_void_p)] + [(n, C.c_int32) for n in
                ('nsplit', 'Nrows', 'Kw', 'N', 'C', 'Qs', 'ntap', 'pending')] + [('sn', C.c_int64), ('sc', C.c_int64), ('beta', C.c_float)]


_P, _F, _I, _Z, _LL = C.c_void_p, C.c_float, C.c_int, C.c_size_t, C.c_longlong
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes).  Every symbol declared in include/iprgan.h must be listed here;
# tests/test_abi.py cross-checks the header, this table and the built library.
SIGNATURES = {
    'iprgan_last_error': (C.c_char_p, []),
    'iprgan_version': (_I, []),
    'iprgan_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'iprgan_nhwc_to_nchw': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'iprgan_permute_021': (_I, [_P, _P, _I, _I, _I, _F, _P]),
    'iprgan_fc_nhwc_ok': (_I, [_I, _I, _I, _I]),
    'iprgan_fc_nhwc_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _Z, _P]),
    'iprgan_fc_nhwc_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _Z, _Z, _F, _P]),
    'iprgan_conv_wfwd_floats': (_Z, [_D]),
    'iprgan_conv_wbwd_floats': (_Z, [_D]),
    'iprgan_conv_wgrad_ws_floats': (_Z, [_D]),
    'iprgan_conv_weight_prep': (_I, [_D, _P, _P, _P, _P, _P]),
    'iprgan_conv_weight_prep_multi': (_I, [_P, _P, _P, _P, _P, _I, _P]),
    'iprgan_conv_fwd_ws_floats': (_Z, [_D]),
    'iprgan_conv_fwd': (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int), _P]),
    'iprgan_conv_stat_floats': (_Z, [_D, _I]),
    'iprgan_colsum_partials': (_I, [_P, _I, _I, _I, _P, _F, _P]),
    'iprgan_colsum_partials_multi': (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    'iprgan_conv_bwd_data_ws_floats': (_Z, [_D]),
    'iprgan_conv_bwd_data': (_I, [_D, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, C.POINTER(C.c_int), _P, _P]),
    'iprgan_conv_bwd_data_bn_ok': (_I, [_D]),
    'iprgan_conv_bwd_data_bn': (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, C.POINTER(C.c_int), _P]),
    'iprgan_colsum_ws_floats': (_Z, [_I, _I]),
    'iprgan_colsum': (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _P]),
    'iprgan_conv_bwd_weight': (_I, [_D, _P, _P, _P, _P, _P, _F, _P]),
    'iprgan_act_bwd': (_I, [_P, _P, _P, _Z, _I, _F, _I, _P]),
    'iprgan_cast': (_I, [_P, _P, _Z, _I, _I, _P]),
    'iprgan_cast_planes': (_I, [_P, _P, _Z, _Z, _I, _P]),
    'iprgan_conv_wgrad_takes_bf16': (_I, [_D]),
    'iprgan_conv_bwd_weight_deferred': (_I, [_D, _P, _P, _P, _P, _P, _F, _P, _P]),
    'iprgan_wgrad_reduce_multi': (_I, [_P, _I, _P]),
    'iprgan_gemv_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _Z, _P]),
    'iprgan_gemv_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _Z, _Z, _P]),
    'iprgan_gemv_fwd_pair': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _Z, _P]),
    'iprgan_gemv_bwd_pair': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _Z, _Z, _P]),
    'iprgan_bn_ws_floats': (_Z, [_I, _I]),
    'iprgan_bn_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _F, _P, _I, _P, _P, _P, _I, _P]),
    'iprgan_bn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _F, _I, _P]),
    'iprgan_bn_prelu_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _P, _P, _I, _P, _P, _P, _I, _P]),
    'iprgan_bn_prelu_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _F, _I, _P]),
    'iprgan_bn_bwd_pre': (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _F, _I, _P]),
    'iprgan_instnorm_ws_floats': (_Z, [_I, _I, _I]),
    'iprgan_instnorm_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _P, _I, _P, _P, _I, _P]),
    'iprgan_instnorm_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _I, _F, _I, _P]),
    'iprgan_prelu_fwd': (_I, [_P, _P, _P, _Z, _I, _P]),
    'iprgan_prelu_bwd': (_I, [_P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    'iprgan_pixel_shuffle2': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_pixel_shuffle2_prelu_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_pixel_shuffle2_prelu_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_maxpool2_fwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_maxpool2_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_add': (_I, [_P, _P, _P, _Z, _I, _P]),
    'iprgan_pool_swap': (_I, [_P, _P, _P, _P, _I, _Z, _P]),
    'iprgan_write_ints': (_I, [_P, _P, _I, _P]),
    'iprgan_reflect_fold': (_I, [_P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_sn_ws_floats': (_Z, [_I, _I]),
    'iprgan_sn_power_iter': (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    'iprgan_sn_multi_ws_floats': (_Z, [_P, _P, _I]),
    'iprgan_sn_power_iter_multi': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _I, _P]),
    'iprgan_sn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'iprgan_sn_bwd_multi': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P]),
    'iprgan_loss_ws_floats': (_Z, [_Z]),
    'iprgan_loss_fwd': (_I, [_I, _P, _P, _P, _P, _Z, _P]),
    'iprgan_loss_bwd': (_I, [_I, _P, _P, _P, _P, _Z, _P]),
    'iprgan_loss_pair_fwd': (_I, [_I, _I, _P, _Z, _P, _P]),
    'iprgan_loss_pair_bwd': (_I, [_I, _I, _P, _P, _P, _Z, _P]),
    'iprgan_loss_sum_fwd': (_I, [_I, _P, _P, _P, _P, _Z, _F, _P]),
    'iprgan_loss_sum_bwd': (_I, [_I, _P, _P, _P, _P, _Z, _F, _P]),
    'iprgan_reparam_fwd': (_I, [_P, _P, _P, _P, _Z, _P]),
    'iprgan_ssim_ws_floats': (_Z, [_I, _I, _I]),
    'iprgan_ssim_gmap_floats': (_Z, [_I, _I, _I]),
    'iprgan_ssim_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'iprgan_ssim_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'iprgan_msssim_sizes': (_I, [_I, _I, _I, C.POINTER(_Z), C.POINTER(_Z), C.POINTER(_Z)]),
    'iprgan_msssim_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'iprgan_msssim_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'iprgan_reparam_bwd': (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    'iprgan_sign_loss_fwd': (_I, [_P, _P, _P, _I, _F, _P, _P]),
    'iprgan_sign_loss_bwd': (_I, [_P, _P, _P, _P, _I, _F, _P, _F, _P]),
    'iprgan_sign_ber': (_I, [_P, _P, _P, _I, _P, _P]),
    'iprgan_adam_step': (_I, [_P, _P, _P, _P, _P, _I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _I,
                         C.c_double, _P]),
    'iprgan_adam_step_dev': (_I, [_P, _P, _P, _P, _P, _I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P,
                             _P, C.c_double, _P]),
    'iprgan_debug_force_tiles': (_I, [_I, _I]),
    'iprgan_debug_force_splitk': (_I, [_I]),
    'iprgan_tune_export': (_I, [_P, _Z, _P]),
    'iprgan_tune_import': (_I, [_P, _Z, _I]),
    'iprgan_set_math_mode': (_I, [_I]),
    'iprgan_get_math_mode': (_I, []),
    'iprgan_prof_enable': (_I, [_I]),
    'iprgan_prof_collect': (_I, []),
    'iprgan_prof_num_kernels': (_I, []),
    'iprgan_prof_get': (_I, [_I, C.c_char_p, _I, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'iprgan_prof_num_layers': (_I, []),
    'iprgan_prof_get_layer': (_I, [_I, C.c_char_p, _I, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'iprgan_fill': (_I, [_P, _F, _Z, _P]),
    'iprgan_axpy': (_I, [_P, _P, _F, _Z, _P]),
    'iprgan_axpy_multi': (_I, [_P, _P, _P, _I, _F, _P]),
    'iprgan_comm_unique_id': (_I, [_P]),
    'iprgan_comm_init': (_I, [_I, _I, _P]),
    'iprgan_allreduce_bucket': (_I, [_P, _Z, _I, _P]),
    'iprgan_comm_nranks': (_I, []),
    'iprgan_comm_rank': (_I, []),
    'iprgan_comm_destroy': (_I, []),
}

_lib = None


def load():
    """Load the library (once).  Raises if it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C ipr-gan_amd/csrc`. iprgan has no CPU/ATen fallback for its kernels.')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
        if _FP32_VIA_X3:
            lib.iprgan_set_math_mode(MATH_MODES['fp32x3'])
    return _lib


def call(name, *args):
    lib = _lib or load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib.iprgan_last_error().decode()}')
    if _acct is not None:
        _acct_close(name)


# ---- algorithmic work accounting (bench.py: `step_roofline`) -----------------------------------------------------------
# While acct_begin() ... acct_end() is open, every entry-point call is recorded as (name, flops, bytes): bytes = the sizes of
# the distinct tensors handed to it through ptr() / ptr_table() - each operand and result counted ONCE, whatever the kernels
# behind the entry point re-read - minus the tensors the wrappers marked as scratch (acct_scratch: slabs, partials, padded
# copies); flops = what the wrapper declared with acct_flops() (2 * MACs of the convolution family; nothing else is priced).
# Host-side bookkeeping only: nothing is launched, timed or synchronised here, and it is off (None) outside bench.py's
# accounting step.
_acct = None
_acct_pending = {}
_acct_scratch = set()
_acct_f = 0.0


def acct_begin():
    global _acct, _acct_f
    _acct, _acct_f = [], 0.0
    _acct_pending.clear()
    _acct_scratch.clear()


def acct_end():
    """-> [(entry point, flops, bytes)] since acct_begin()."""
    global _acct
    out, _acct = _acct, None
    _acct_pending.clear()
    _acct_scratch.clear()
    return out or []


def acct_on():
    return _acct is not None


def acct_flops(f):
    global _acct_f
    if _acct is not None:
        _acct_f += float(f)


def acct_bytes(n):
    """Bytes of tensors that reach the entry point through a pointer table built earlier (optim.Adam's cached tables)."""
    if _acct is not None:
        _acct_pending[-1] = _acct_pending.get(-1, 0) + int(n)


def acct_scratch(t):
    if _acct is not None and t is not None:
        _acct_scratch.add(t.data_ptr())
    return t


def _acct_note(t):
    p = t.data_ptr()
    if p in _acct_scratch:
        return
    n = t.numel() * t.element_size()
    if t.dtype == torch.bfloat16 and act_x3():       # a three-plane tensor is handed over as its h plane
        n *= 3
    if n > _acct_pending.get(p, 0):
        _acct_pending[p] = n


def _acct_close(name):
    global _acct_f
    _acct.append((name, _acct_f, float(sum(_acct_pending.values()))))
    _acct_pending.clear()
    _acct_f = 0.0


def query(name, *args):
    return getattr(load(), name)(*args)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses anything that is not a contiguous fp32
    GPU tensor so that a CPU tensor can never silently reach a kernel."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('iprgan kernels need GPU tensors (got a CPU tensor); there is no CPU path')
    if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous():
        raise RuntimeError(f'iprgan kernels need contiguous float32 (or, for activations in bf16-activation mode, '
                           f'bfloat16) tensors (got {t.dtype}, contiguous={t.is_contiguous()})')
    if _acct is not None:
        _acct_note(t)
    return t.data_ptr()


def ptr32(t):
    """ptr() for arguments that are ALWAYS fp32 (weights, statistics, gradients of parameters, RGB images)."""
    if t is not None and t.dtype != torch.float32:
        raise RuntimeError(f'this iprgan kernel argument must be float32 (got {t.dtype})')
    return ptr(t)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """Raw hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (11 us per call, measured: a third of the host time of a DCGAN-64 step, which
    launches ~150 kernels); the two C accessors below return the same handle in well under a microsecond."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def ptr_table(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr


def prof_enable(on):
    call('iprgan_prof_enable', 1 if on else 0)


def prof_results():
    """[{name, launches, ms, flops}] accumulated since prof_enable(True); waits for pending events."""
    call('iprgan_prof_collect')
    out = []
    for i in range(query('iprgan_prof_num_kernels')):
        name = C.create_string_buffer(96)
        n, ms, fl = C.c_longlong(0), C.c_double(0), C.c_double(0)
        call('iprgan_prof_get', i, name, 96, C.byref(n), C.byref(ms), C.byref(fl))
        out.append(dict(name=name.value.decode(), launches=n.value, ms=ms.value, flops=fl.value))
    return out


def prof_layers():
    """prof_results() grouped by layer (pass + geometry); call after prof_results() (which collects)."""
    out = []
    for i in range(query('iprgan_prof_num_layers')):
        name = C.create_string_buffer(160)
        n, ms, fl = C.c_longlong(0), C.c_double(0), C.c_double(0)
        call('iprgan_prof_get_layer', i, name, 160, C.byref(n), C.byref(ms), C.byref(fl))
        if n.value:
            out.append(dict(name=name.value.decode(), launches=n.value, ms=ms.value, flops=fl.value))
    return out


MATH_MODES = {'fp32': 0, 'bf16': 1, 'bf16act': 1, 'fp32x3': 2}
# validation switch: every 'fp32' request (and the library default) runs as 'fp32x3', so that the whole GPU suite, with its
# fp32 tolerances, can be pointed at the split-operand tiles: IPRGAN_FP32_VIA_X3=1 python -m pytest tests -m gpu
_FP32_VIA_X3 = os.environ.get('IPRGAN_FP32_VIA_X3', '0') == '1'
_act_bf16 = False


def set_math(mode):
    """Process-wide math mode of the conv family (include/iprgan.h: iprgan_set_math_mode): 'fp32' (default), 'bf16'
    (bf16 MFMA tiles, fp32 accumulation, fp32 tensors and master weights in HBM) or 'bf16act' (the same tiles, and
    activations whose padded channel count is a multiple of 64 LIVE as bf16 in HBM: half the activation traffic, operands
    reach LDS without a conversion; DCGAN-family layers only this round).  'fp32x3': fp32 tensors and fp32-grade
    products on the bf16 matrix pipe - every operand element is split into three bf16 terms while it is staged into LDS and
    a product block is six bf16 MFMAs (conv_igemm.hip: SPLIT); layers whose channel count is not a multiple of 32 keep the
    fp32 MFMA."""
    global _math_cached, _act_bf16
    if _FP32_VIA_X3 and mode in ('fp32', 0):
        mode = 'fp32x3'
    _act_bf16 = mode == 'bf16act'
    _math_cached = MATH_MODES[mode] if isinstance(mode, str) else int(mode)
    call('iprgan_set_math_mode', _math_cached)


def act_bf16():
    """True in 'bf16act' mode (host-side allocation rule; the kernels are told per tensor through the descriptors)."""
    return _act_bf16


# 'fp32x3' mode: activations whose padded channel count is a multiple of 32 live as three bf16 planes (include/iprgan.h:
# IPRGAN_ST_X3).  IPRGAN_X3_PLANES=0: round 3's form (fp32 tensors, operands split while they are staged into LDS) for A/B.
_X3_PLANES = os.environ.get('IPRGAN_X3_PLANES', '1') != '0'


def act_x3():
    """True in 'fp32x3' mode with three-plane activations (host-side allocation rule, like act_bf16)."""
    return _math_cached == 2 and _X3_PLANES


_math_cached = 2 if _FP32_VIA_X3 else 0


def get_math_cached():
    """The math mode last set through set_math (no library call: part of per-pass cache keys)."""
    return _math_cached + (2 if _act_bf16 else 0)


def get_math():
    return 'bf16act' if _act_bf16 else {0: 'fp32', 1: 'bf16', 2: 'fp32x3'}[load().iprgan_get_math_mode()]
