"""ctypes binding of ``include/deephumor_hip.h`` (the C-ABI of the gfx950 kernels).

PyTorch is plumbing here: it owns device memory and the stream; every operation on the product
path goes through ``libdeephumor_hip.so``.  There is NO CPU or eager-torch fallback: if the
library cannot be loaded, importing a kernel raises ``RuntimeError``.
"""
import ctypes
import os

import torch

from . import _build

from ._abi import (ABI_VERSION, BF16, BF16_OUT_F32, ERR_ALL_FILTERED, ERR_NONFINITE, ERR_OVERFLOW, ERR_TOO_FEW, F16, F16_OUT_F32, F32,  # noqa: F401
                   HALF_DTYPES, MAX_BEAMS, SIGNATURES, LnFold, LstmLayer, LstmModel, LstmScratch, TrLayer, TrModel, TrScratch)

_c = ctypes
_P, _I, _F, _U64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_uint64


_lib = None


def load():
    """Loads the shared library.  A single-process run with a compiler present rebuilds it first when the sources are
    newer (developer convenience; the build itself is serialised by a file lock and replaces the library atomically).
    Under a multi-process launcher (WORLD_SIZE > 1) nothing is ever built here -- every rank would race on the same
    files: run ``python __graft_entry__.py build`` first."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    solo = int(os.environ.get("WORLD_SIZE", "1")) == 1 and int(os.environ.get("LOCAL_RANK", "0")) == 0
    if solo and os.path.exists(_build._hipcc()) and (not os.path.exists(path) or _build.needs_build()):
        try:
            _build.build()
        except Exception as e:  # pragma: no cover - depends on toolchain presence
            raise RuntimeError(f"deephumor_amd: cannot build {path}: {e}") from e
    if not os.path.exists(path):
        raise RuntimeError(f"deephumor_amd: HIP extension missing at {path}; run `python __graft_entry__.py build`. "
                           "There is no CPU fallback on the product path.")
    lib = ctypes.CDLL(os.environ.get("DEEPHUMOR_HIP_LIB") or path)      # (developer A/B switch: another build of the same ABI)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)             # AttributeError if the ABI lost a symbol
        fn.argtypes = argtypes
        fn.restype = _I
    lib.dh_error_string.argtypes = [_I]
    lib.dh_error_string.restype = _c.c_char_p
    lib.dh_prof_tag.argtypes = [_c.c_char_p]
    lib.dh_prof_tag.restype = None
    for name in ("dh_option_name", "dh_option_env"):
        getattr(lib, name).argtypes = [_I]
        getattr(lib, name).restype = _c.c_char_p
    if lib.dh_abi_version() != ABI_VERSION:
        raise RuntimeError(f"deephumor_amd: {path} has ABI v{lib.dh_abi_version()}, this binding expects v{ABI_VERSION}; "
                           "rebuild with `python __graft_entry__.py build`")
    _lib = lib
    return lib


def option(key):
    """Current value of a run-time option (``dh_get_option``): the library's ONE table of kernel-selection switches -- default
    from the option's environment variable, else built in; see ``options()`` and INTEGRATION.md."""
    got = _opt_cache.get(key)
    if got is None:          # (cached on the Python side: the models ask per layer; every change goes through set_option below)
        v = _I()
        _check(load().dh_get_option(key.encode(), _c.byref(v)), f"dh_get_option({key!r})")
        got = _opt_cache[key] = v.value
    return got


def set_option(key, value):
    """``dh_set_option``: overrides an option from now on.  Dispatch options of the native step drivers (``decode_wreg``,
    ``decode_wreg_min_rows``, ``qkv_fusion_max_rows``, ``cross_qproj``, ``lstm_wreg``, ``vocab_wreg``, tile choices) take effect at
    the next launch; the options the Python layer reads while it builds a model's weight plan (``conv1x1_wreg`` ... ``qproj_fusion``,
    ``f32_split``) are part of every plan's cache key (``options_epoch``): models rebuild their plans -- and re-capture their
    hipGraphs -- at the next call.  Returns the previous value."""
    global options_epoch
    old = option(key)
    _check(load().dh_set_option(key.encode(), int(value)), f"dh_set_option({key!r})")
    _opt_cache.pop(key, None)
    if int(value) != old:
        options_epoch += 1
    return old


_opt_cache = {}
options_epoch = 0       # bumped by every set_option that changes a value; part of the models' plan keys


class option_scope:
    """``with hip.option_scope(decode_wreg=0, s3_tail=0): ...`` -- options set for the block, previous values restored after it
    (A/B tests of results-identical kernels)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)


def options():
    """{key: (value, environment variable)} of every option."""
    lib = load()
    return {lib.dh_option_name(i).decode(): (option(lib.dh_option_name(i).decode()), lib.dh_option_env(i).decode())
            for i in range(lib.dh_option_count())}


def _check(code, name):
    if code != 0:
        raise RuntimeError(f"{name} failed: {load().dh_error_string(code).decode()} (code {code})")


from ._prof import Profiler  # noqa: E402  (the in-library event profiler; needs load / _check above)


_prof = None
_fns = {}


def profile(watch=None, stride=1):
    """``with hip.profile(watch={...}) as prof: ...; prof.summary()`` -- per-launch-key timing from HIP events inside the library."""
    return Profiler(watch, stride)


def _launch(name, *args, tag=None):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    if tag is not None and _prof is not None:
        load().dh_prof_tag(tag.encode())
    code = fn(*args)
    if code != 0:
        _check(code, name)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("deephumor_amd kernels need CUDA(HIP) tensors; there is no CPU path "
                               "(move the model and inputs to 'cuda')")


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}: the kernels compute in fp32 (parity path) or with bf16 / fp16 operands on "
                    "the matrix cores (fp32 accumulation) -- use model.float(), model.bfloat16() or model.half()")


# ------------------------------------------------------------------------------------------------
# thin tensor-level wrappers (shape checks live here; the library only sees pointers and sizes)
# ------------------------------------------------------------------------------------------------
def conv2d_bn_act(x, w, scale, shift, residual=None, relu=True, stride=1, pad=0, out=None):
    _dev(x, w, scale, shift, residual)
    n, cin, h, wd = x.shape
    cout, cin2, kh, kw = w.shape
    assert cin == cin2 and x.is_contiguous() and w.is_contiguous()
    ho, wo = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device)
    _launch("dh_conv2d_bn_act", _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
                                   n, cin, h, wd, cout, kh, kw, stride, pad, int(relu), _dt(x), _stream(),
            tag=f"{kh}x{kw}")
    return out


def stem_conv_nhwc(x, w, scale, shift, stride=2, pad=3, relu=True, out_dtype=torch.bfloat16):
    """x NCHW fp32 image, w fp32 [Cout,Cin,KS,KS] -> channels-last bf16 / fp16 [N,Ho,Wo,Cout]."""
    _dev(x, w, scale, shift)
    n, cin, h, wd = x.shape
    cout, _, ks, _ = w.shape
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    out = torch.empty((n, ho, wo, cout), dtype=out_dtype, device=x.device)
    _launch("dh_stem_conv_nhwc", _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(out), n, cin, h, wd, cout, ks,
            stride, pad, int(relu), _dt(out), _stream())
    return out


def pack_nchw_to_nhwc8(x, out_dtype=torch.bfloat16):
    """x NCHW fp32 [N,C<=8,H,W] -> channels-last bf16 / fp16 [N,H,W,8] (zero-padded channels)."""
    _dev(x)
    n, c, h, w = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((n, h, w, 8), dtype=out_dtype, device=x.device)
    _launch("dh_pack_nchw_to_nhwc8", _ptr(x), _ptr(out), n, c, h, w, _dt(out), _stream())
    return out


def round16_keep_nonzero(x, out_dtype):
    """fp32 tensor -> bf16 / fp16 tensor of the same shape, rounding to nearest even but never flushing a non-zero value to zero."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and out_dtype in HALF_DTYPES
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _launch("dh_round16_keep_nonzero", _ptr(x), _ptr(out), x.numel(), _dt(out), _stream())
    return out


def maxpool3x3s2_nhwc(x):
    _dev(x)
    n, h, w, c = x.shape
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=x.dtype, device=x.device)
    _launch("dh_maxpool3x3s2_nhwc", _ptr(x), _ptr(out), n, h, w, c, _dt(x), _stream())
    return out


def avgpool_nhwc(x):
    """x [N, H, W, C] channels-last -> [N, C]"""
    _dev(x)
    n, h, w, c = x.shape
    out = torch.empty((n, c), dtype=x.dtype, device=x.device)
    _launch("dh_avgpool_nhwc", _ptr(x), _ptr(out), n, h * w, c, _dt(x), _stream())
    return out


def maxpool3x3s2(x):
    _dev(x)
    n, c, h, w = x.shape
    out = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=x.dtype, device=x.device)
    _launch("dh_maxpool3x3s2", _ptr(x), _ptr(out), n, c, h, w, _dt(x), _stream())
    return out


def avgpool_rows(x):
    """x [N, C, H, W] -> [N, C]"""
    _dev(x)
    n, c, h, w = x.shape
    out = torch.empty((n, c), dtype=x.dtype, device=x.device)
    _launch("dh_avgpool_rows", _ptr(x), _ptr(out), n * c, h * w, _dt(x), _stream())
    return out


def nchw_to_rows(x):
    """x [N, C, H, W] -> [N, H*W, C]"""
    _dev(x)
    n, c, h, w = x.shape
    out = torch.empty((n, h * w, c), dtype=x.dtype, device=x.device)
    _launch("dh_nchw_to_rows", _ptr(x), _ptr(out), n, c, h * w, _dt(x), _stream())
    return out


def label_mean(emb, labels, out):
    """emb [V, E], labels int64 [N, L] -> out [N, E] view (row stride may exceed E)."""
    _dev(emb, labels, out)
    n, l = labels.shape
    assert labels.dtype == torch.int64 and labels.is_contiguous() and out.stride(1) == 1
    _launch("dh_label_mean", _ptr(emb), _ptr(labels), _ptr(out), out.stride(0), n, l, emb.shape[1], _dt(emb),
                                _stream())
    return out


def split_f32x(w):
    """fp32 ``w [N, K]`` -> the fp16 planes ``[2, N, Kp]`` of ``dh_split_f32x`` (hi, lo * 2^11; ``Kp`` = K rounded up to 32): the
    weight operand of ``linear_f32x`` / ``conv2d_nhwc_f32x``.  Made once per weight version (the models' plans)."""
    _dev(w)
    assert w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1
    n, k = w.shape
    kp = (k + 31) // 32 * 32
    planes = torch.empty((2, n, kp), dtype=torch.float16, device=w.device)
    _launch("dh_split_f32x", _ptr(w), w.stride(0), _ptr(planes), n, k, kp, _stream())
    return planes


def pack_f32x_fragments(planes):
    """The two planes of ``split_f32x`` (``[2, N, Kp]`` fp16), each through ``dh_pack_mfma_fragments`` -> ``[2, Kp / 32, N / 16, 64, 8]``:
    the weight operand of ``linear_f32x_wreg`` (fragments loaded straight into registers).  None when the shape is not one it takes."""
    _dev(planes)
    two, n, kp = planes.shape
    if two != 2 or n % 64 or not load().dh_linear_f32x_wreg_supported(40, n, kp):
        return None
    out = torch.empty((2, kp // 32, n // 16, 64, 8), dtype=torch.float16, device=planes.device)
    for i in range(2):
        _launch("dh_pack_mfma_fragments", _ptr(planes[i]), _ptr(out[i]), n, kp, _stream())
    return out


def linear_f32x_wreg(a, packed, bias, relu=False, residual=None, out=None, tag=None):
    """``dh_linear_f32x_wreg``: ``linear_f32x`` for a decode position's rows with the split weights stationary in registers
    (``packed = pack_f32x_fragments(split_f32x(w))``); bit-identical to ``linear_f32x``."""
    _dev(a, packed, bias, residual, out)
    m, k = a.shape
    n = packed.shape[2] * 16
    assert a.dtype == torch.float32 and a.stride(1) == 1 and packed.shape[1] * 32 == k
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _launch("dh_linear_f32x_wreg", _ptr(a), a.stride(0), _ptr(packed), _ptr(bias), _ptr(residual), residual.stride(0) if residual is not None else 0,
            _ptr(out), out.stride(0), m, n, k, int(relu), _stream(), tag=tag)
    return out


def f32x_take_overflow(device=None):
    """``dh_f32x_take_overflow`` on the current stream -> bool (a host read: synchronises).  True when an f32x launch on this stream
    since the last call split an ACTIVATION outside the fp16 range: its results hold inf / NaN and the caller repeats on the exact
    path (``f32x_guarded``)."""
    flag = torch.zeros((1,), dtype=torch.int32, device=device or torch.device("cuda", torch.cuda.current_device()))
    _launch("dh_f32x_take_overflow", _ptr(flag), _stream())
    return bool(int(flag.item()))


def f32_split_ok(w):
    """Plan-time range check of the f32x path (fp16 planes): every weight below the fp16 maximum."""
    return bool(w.detach().abs().max() < 6.0e4)


def linear_f32x(a, planes, bias=None, scale=None, shift=None, relu=False, out=None, tag=None, residual=None):
    """``dh_linear_f32x``: fp32 ``a [M, K] @ w[N, K].T`` as three fp16 MFMAs on split operands (fp32-class result)."""
    _dev(a, planes, bias, scale, shift, out, residual)
    m, k = a.shape
    n, kp = planes.shape[1], planes.shape[2]
    assert a.dtype == torch.float32 and planes.dtype == torch.float16 and a.stride(1) == 1 and kp == (k + 31) // 32 * 32
    assert k % 4 == 0, "dh_linear_f32x takes K % 4 == 0 (16-byte fp32 chunks); hip.linear() keeps the exact-fp32 kernel otherwise"
    if a.stride(0) % 4 or a.data_ptr() % 16:
        a = a.contiguous()
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    assert out.shape == (m, n) and out.stride(1) == 1 and out.dtype == torch.float32
    _launch("dh_linear_f32x", _ptr(a), a.stride(0), _ptr(planes), kp, _ptr(bias), _ptr(scale), _ptr(shift), _ptr(residual),
            residual.stride(0) if residual is not None else 0, _ptr(out), out.stride(0), m, n, k, int(relu), _stream(), tag=tag)
    return out


def conv2d_nhwc_f32x(x, planes, ks, scale, shift, residual=None, relu=True, stride=1, pad=0):
    """``dh_conv2d_nhwc_f32x``: channels-last fp32 convolution + BatchNorm affine (+ residual) (+ ReLU); ``planes`` =
    ``split_f32x`` of the ``[Cout, KS*KS*Cin]`` weight (ci fastest)."""
    _dev(x, planes, scale, shift, residual)
    n, h, w, cin = x.shape
    cout, kp = planes.shape[1], planes.shape[2]
    assert x.dtype == torch.float32 and x.is_contiguous() and kp == (ks * ks * cin + 31) // 32 * 32
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    _launch("dh_conv2d_nhwc_f32x", _ptr(x), _ptr(planes), kp, _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), n, h, w, cin, cout,
            ks, stride, pad, int(relu), _stream())
    return y


def nchw_to_nhwc_f32(x, cp=4):
    _dev(x)
    n, c, h, w = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device)
    _launch("dh_nchw_to_nhwc_f32", _ptr(x), _ptr(y), n, c, h, w, cp, _stream())
    return y


def maxpool3x3s2_nhwc_f32(x):
    _dev(x)
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, c), dtype=torch.float32, device=x.device)
    _launch("dh_maxpool3x3s2_nhwc_f32", _ptr(x), _ptr(y), n, h, w, c, _stream())
    return y


def avgpool_nhwc_f32(x):
    """x [N, H, W, C] fp32 -> [N, C]."""
    _dev(x)
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _launch("dh_avgpool_nhwc_f32", _ptr(x), _ptr(y), n, h * w, c, _stream())
    return y


def linear(a, w, bias=None, scale=None, shift=None, relu=False, out=None, tag=None, residual=None,
           out_dtype=None, w_x=None):
    """a [M, K] (row stride may exceed K), w [N, K] -> [M, N].  bf16 / fp16 operands may produce fp32
    (``out_dtype=torch.float32``: logits).  ``w_x``: ``split_f32x(w)`` -- an fp32 product then runs on the 16-bit matrix cores
    with split operands when option ``f32_split`` is on (``linear_f32x``)."""
    _dev(a, w, bias, scale, shift, out, residual)
    if w_x is not None and a.dtype == torch.float32 and a.shape[1] % 4 == 0 and option("f32_split"):
        return linear_f32x(a, w_x, bias, scale, shift, relu, out, tag, residual)
    m, k = a.shape
    n, k2 = w.shape
    assert k == k2 and a.stride(1) == 1 and w.stride(1) == 1 and a.dtype == w.dtype
    if out is None:
        if a.dtype in HALF_DTYPES and out_dtype == torch.float32 and (n % 4) and m * n >= (1 << 24):
            # large fp32 logits: rows padded to 64 floats (16-byte aligned rows) so that the persistent classifier kernel
            # with its 16-byte register stores applies; the result is a [m, n] view of the padded buffer
            out = torch.empty((m, (n + 63) // 64 * 64), dtype=torch.float32, device=a.device)[:, :n]
        else:
            out = torch.empty((m, n), dtype=out_dtype or a.dtype, device=a.device)
    assert out.shape == (m, n) and out.stride(1) == 1
    dt = _dt(a)
    if dt in (BF16, F16) and out.dtype == torch.float32:
        dt = BF16_OUT_F32 if dt == BF16 else F16_OUT_F32
    else:
        assert out.dtype == a.dtype
    _launch("dh_linear", _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(scale), _ptr(shift),
            _ptr(residual), residual.stride(0) if residual is not None else 0,
            _ptr(out), out.stride(0), m, n, k, int(relu), dt, _stream(), tag=tag)
    return out


def linear_ln(a, w, bias, out=None, residual=None, relu=False, a_ln=None, r_ln=None, want_stats=False, tag=None):
    """``dh_linear_ln``: 16-bit ``a [M, K] @ w[N, K].T + bias (+ residual)`` with deferred LayerNorm.
    ``a_ln = (stats [M, K/64, 2], eps, colsum [N])``: ``a`` is pre-LayerNorm, ``w`` / ``bias`` have gamma / beta folded in;
    ``r_ln = (stats [M, N/64, 2], eps, gamma, beta)``: the residual rows are pre-LayerNorm;
    ``want_stats``: also returns the partial statistics ``[M, N/64, 2]`` of the output rows."""
    _dev(a, w, bias, out, residual)
    m, k = a.shape
    n = w.shape[0]
    assert a.dtype in HALF_DTYPES and a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    f = LnFold()
    if a_ln is not None:
        st, eps, colsum = a_ln
        f.a_stats, f.a_tiles, f.a_eps, f.a_colsum = _ptr(st), k // 64, float(eps), _ptr(colsum)
    if r_ln is not None:
        st, eps, gamma, beta = r_ln
        f.r_stats, f.r_tiles, f.r_eps, f.r_gamma, f.r_beta = _ptr(st), n // 64, float(eps), _ptr(gamma), _ptr(beta)
    stats = torch.empty((m, n // 64, 2), dtype=torch.float32, device=a.device) if want_stats else None
    f.o_stats = _ptr(stats)
    _launch("dh_linear_ln", _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(residual),
            residual.stride(0) if residual is not None else 0, _ptr(out), out.stride(0), m, n, k, int(relu), _c.byref(f),
            _dt(a), _stream(), tag=tag)
    return (out, stats) if want_stats else out


def linear_ln_wreg_supported(n, k, with_residual_stats):
    return bool(load().dh_linear_ln_wreg_supported(int(n), int(k), int(bool(with_residual_stats))))


def linear_ln_wreg(a, w_packed, n, bias, out=None, residual=None, relu=False, a_ln=None, r_ln=None, tag=None):
    """``dh_linear_ln_wreg``: ``linear_ln`` on ``w_packed = pack_mfma_fragments(w [n, k])`` -- the register-stationary decode
    GEMM (bit-identical results).  With ``residual`` the partial statistics of the output rows are always produced and
    returned: ``(out, stats)``."""
    _dev(a, w_packed, bias, out, residual)
    m, k = a.shape
    assert a.dtype in HALF_DTYPES and a.dtype == w_packed.dtype and a.stride(1) == 1 and w_packed.numel() == n * k
    if out is None:
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    f = LnFold()
    if a_ln is not None:
        st, eps, colsum = a_ln
        f.a_stats, f.a_tiles, f.a_eps, f.a_colsum = _ptr(st), k // 64, float(eps), _ptr(colsum)
    if r_ln is not None:
        st, eps, gamma, beta = r_ln
        f.r_stats, f.r_tiles, f.r_eps, f.r_gamma, f.r_beta = _ptr(st), n // 64, float(eps), _ptr(gamma), _ptr(beta)
    stats = torch.empty((m, n // 64, 2), dtype=torch.float32, device=a.device) if residual is not None else None
    f.o_stats = _ptr(stats)
    _launch("dh_linear_ln_wreg", _ptr(a), a.stride(0), _ptr(w_packed), _ptr(bias), _ptr(residual),
            residual.stride(0) if residual is not None else 0, _ptr(out), out.stride(0), m, n, k, int(relu), _c.byref(f),
            _dt(a), _stream(), tag=tag)
    return (out, stats) if residual is not None else out


def attn_cross_pack(kv, n_img, s, d, n_heads, dperm=False):
    """kv [n_img*S, 2D] (16-bit) -> (kp, vt) [n_img, n_heads, 64, 64] each: the matrix-core cross-attention layout
    (``dperm``: K's head-dim slots permuted for ``attn_cross_qproj_decode``)."""
    _dev(kv)
    kp = torch.empty((n_img, n_heads, 64, 64), dtype=kv.dtype, device=kv.device)
    vt = torch.empty_like(kp)
    _launch("dh_attn_cross_pack", _ptr(kv), _ptr(kp), _ptr(vt), n_img, s, d, n_heads, int(dperm), _dt(kv), _stream())
    return kp, vt


def attn_cross_decode_packed(q, kp, vt, keymask, out, n_img, rows_per_img, s, d, n_heads, scale, dperm=False):
    _dev(q, kp, vt, keymask, out)
    _launch("dh_attn_cross_decode_packed", _ptr(q), q.stride(0), _ptr(kp), _ptr(vt), _ptr(keymask), _ptr(out), n_img,
            rows_per_img, s, d, n_heads, float(scale), int(dperm), _dt(q), _stream())
    return out


def attn_cross_prefill_packed(q, kp, vt, keymask, n_img, n_pos, s, d, n_heads, scale, dperm=False):
    _dev(q, kp, vt, keymask)
    out = torch.empty((n_img * n_pos, d), dtype=q.dtype, device=q.device)
    _launch("dh_attn_cross_prefill_packed", _ptr(q), q.stride(0), _ptr(kp), _ptr(vt), _ptr(keymask), _ptr(out), n_img, n_pos,
            s, d, n_heads, float(scale), int(dperm), _dt(q), _stream())
    return out


def conv2d_nhwc_bn_act(x, w, scale, shift, residual=None, relu=True, stride=1, pad=0):
    """Channels-last bf16 convolution: x [N,H,W,Cin], w [Cout,KS,KS,Cin] -> [N,Ho,Wo,Cout]."""
    _dev(x, w, scale, shift, residual)
    n, h, wd, cin = x.shape
    cout, ks, ks2, cin2 = w.shape
    assert cin == cin2 and ks == ks2 and x.is_contiguous() and w.is_contiguous()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    out = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
    _launch("dh_conv2d_nhwc_bn_act", _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
            n, h, wd, cin, cout, ks, stride, pad, int(relu), _dt(x), _stream(), tag=f"{ks}x{ks}")
    return out


def conv2d_nhwc_bn_relu_maxpool(x, w, scale, shift, stride=2, pad=3):
    """The ResNet stem in one launch: x [N,H,W,Cin] 16-bit (Cin % 8 == 0), w [Cout,KS,KS,Cin] -> pooled [N,Ho/2,Wo/2,Cout]."""
    _dev(x, w, scale, shift)
    n, h, wd, cin = x.shape
    cout, ks, _, cin2 = w.shape
    assert cin == cin2 and x.is_contiguous() and w.is_contiguous()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    out = torch.empty((n, ho // 2, wo // 2, cout), dtype=x.dtype, device=x.device)
    _launch("dh_conv2d_nhwc_bn_relu_maxpool", _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(out), n, h, wd, cin, cout, ks,
            stride, pad, _dt(x), _stream())
    return out


_c3_ok = {}


def conv3x3_direct_supported(h, w, cin, cout):
    key = (int(h), int(w), int(cin), int(cout))
    if key not in _c3_ok:
        _c3_ok[key] = bool(load().dh_conv3x3_direct_supported(*key))
    return _c3_ok[key]


def conv3x3_direct_nhwc(x, w, scale, shift):
    """3x3 / stride 1 / pad 1 convolution + BN + ReLU as a direct matrix-core convolution (stage-1 / stage-2 bottleneck conv2)."""
    _dev(x, w, scale, shift)
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    assert tuple(w.shape) == (cout, 3, 3, cin) and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype
    out = torch.empty((n, h, wd, cout), dtype=x.dtype, device=x.device)
    _launch("dh_conv3x3_direct_nhwc", _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(out), n, h, wd, cin, cout, 1, _dt(x), _stream(),
            tag="3x3")
    return out


def bottleneck_tail_nhwc(y1, w2, scale2, shift2, w3, scale3, shift3, residual):
    """relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(y1))))) + residual) in one launch (shapes of ``conv3x3_direct_supported``)."""
    _dev(y1, w2, scale2, shift2, w3, scale3, shift3, residual)
    n, h, wd, c = y1.shape
    assert tuple(w2.shape) == (c, 3, 3, c) and w3.shape[0] == 4 * c and w3.numel() == 4 * c * c and tuple(residual.shape) == (n, h, wd, 4 * c)
    assert y1.is_contiguous() and w2.is_contiguous() and w3.is_contiguous() and residual.is_contiguous() and y1.dtype == w2.dtype == w3.dtype == residual.dtype
    out = torch.empty_like(residual)
    _launch("dh_bottleneck_tail_nhwc", _ptr(y1), _ptr(w2), _ptr(scale2), _ptr(shift2), _ptr(w3), _ptr(scale3), _ptr(shift3), _ptr(residual),
            _ptr(out), n, h, wd, c, _dt(y1), _stream())
    return out


def bottleneck_tail_s3_supported(h, w, c):
    return bool(load().dh_bottleneck_tail_s3_supported(int(h), int(w), int(c)))


def pack_mfma_fragments(w):
    """16-bit weight matrix ``[R, K]`` (or a conv weight ``[Cout, kh, kw, Cin]`` viewed as ``[Cout, kh*kw*Cin]``) -> the
    fragment-ordered copy the register-streaming kernels load (one coalesced 1 KB load per 16 rows x 32 k)."""
    _dev(w)
    assert w.dtype in HALF_DTYPES and w.is_contiguous()
    r, k = w.shape[0], w.numel() // w.shape[0]
    out = torch.empty((r * k,), dtype=w.dtype, device=w.device)
    _launch("dh_pack_mfma_fragments", _ptr(w), _ptr(out), r, k, _stream())
    return out


def bottleneck_tail_s1_supported(h, w, c, n1=0):
    return bool(load().dh_bottleneck_tail_s1_supported(int(h), int(w), int(c), int(n1)))


def bottleneck_tail_s1_nhwc(y1, w2p, scale2, shift2, w3p, scale3, shift3, residual, w1p=None, scale1=None, shift1=None, n1=0):
    """``dh_bottleneck_tail_s1_nhwc``: the stage-1 bottleneck tail on fragment-packed weights; with ``w1p`` also the NEXT bottleneck's
    conv1 + bn1 + relu on the output tile while it is in LDS.  Returns ``out`` or ``(out, y1_next)``."""
    _dev(y1, w2p, scale2, shift2, w3p, scale3, shift3, residual, w1p, scale1, shift1)
    n, h, w, c = y1.shape
    out = torch.empty((n, h, w, 4 * c), dtype=y1.dtype, device=y1.device)
    y1n = torch.empty((n, h, w, n1), dtype=y1.dtype, device=y1.device) if w1p is not None else None
    _launch("dh_bottleneck_tail_s1_nhwc", _ptr(y1), _ptr(w2p), _ptr(scale2), _ptr(shift2), _ptr(w3p), _ptr(scale3), _ptr(shift3),
            _ptr(residual), _ptr(out), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(y1n), n1, n, h, w, c, _dt(y1), _stream())
    return out if w1p is None else (out, y1n)


def bottleneck_tail_s2_supported(h, w, c):
    return bool(load().dh_bottleneck_tail_s2_supported(int(h), int(w), int(c)))


def bottleneck_tail_s2_nhwc(y1, w2p, scale2, shift2, w3p, scale3, shift3, residual, w1p=None, scale1=None, shift1=None, n1=0):
    """``dh_bottleneck_tail_s2_nhwc``: the stage-2 bottleneck tail on fragment-packed weights; bit-identical to ``bottleneck_tail_nhwc``.
    With ``w1p`` also the NEXT bottleneck's conv1 + bn1 + relu (512 -> 128) on the output chunks while they are in LDS: returns
    ``(out, y1_next)``."""
    _dev(y1, w2p, scale2, shift2, w3p, scale3, shift3, residual, w1p, scale1, shift1)
    n, h, w, c = y1.shape
    out = torch.empty((n, h, w, 4 * c), dtype=y1.dtype, device=y1.device)
    y1n = torch.empty((n, h, w, n1), dtype=y1.dtype, device=y1.device) if w1p is not None else None
    _launch("dh_bottleneck_tail_s2_nhwc", _ptr(y1), _ptr(w2p), _ptr(scale2), _ptr(shift2), _ptr(w3p), _ptr(scale3), _ptr(shift3),
            _ptr(residual), _ptr(out), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(y1n), n1, n, h, w, c, _dt(y1), _stream())
    return out if w1p is None else (out, y1n)


def conv3x3_s4_supported(h, w, c):
    return bool(load().dh_conv3x3_s4_supported(int(h), int(w), int(c)))


def conv3x3_s4_nhwc(x, w_packed, scale, shift):
    """``dh_conv3x3_s4_nhwc``: 3x3 stride-1 convolution + BatchNorm + ReLU of ``x [N, 7, 7, 512]`` on fragment-packed weights
    (``pack_mfma_fragments(w [512, 3, 3, 512])``); bit-identical to ``conv2d_nhwc_bn_act``."""
    _dev(x, w_packed, scale, shift)
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    _launch("dh_conv3x3_s4_nhwc", _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(y), n, h, w, c, _dt(x), _stream())
    return y


def conv1x1_wreg_supported(m, cin, cout):
    return bool(load().dh_conv1x1_wreg_supported(int(m), int(cin), int(cout)))


def conv1x1_wreg_nhwc(x, w_packed, cout, scale, shift, relu=True, residual=None):
    """``dh_conv1x1_wreg_nhwc``: 1x1 convolution + BatchNorm (+ residual, Cin = 512 only) (+ ReLU) of channels-last ``x [N, H, W, Cin]`` on
    fragment-packed weights (``pack_mfma_fragments(w [Cout, Cin])``): weights stationary in registers, pixels streamed
    (csrc/conv1x1_wreg.hip).  Bit-identical to ``conv2d_nhwc_bn_act``."""
    _dev(x, w_packed, scale, shift, residual)
    n, h, w, cin = x.shape
    out = torch.empty((n, h, w, cout), dtype=x.dtype, device=x.device)
    _launch("dh_conv1x1_wreg_nhwc", _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out), n * h * w, cin, cout,
            int(relu), _dt(x), _stream())
    return out


def bottleneck_tail_s3_nhwc(y1, w2p, scale2, shift2, w3p=None, scale3=None, shift3=None, residual=None):
    """Stage-3 bottleneck tail (14 x 14 x 256 -> 1024) in one launch on fragment-packed weights (``pack_mfma_fragments``);
    ``w3p=None``: only relu(bn2(conv2_3x3(y1)))."""
    _dev(y1, w2p, scale2, shift2, w3p, scale3, shift3, residual)
    n, h, wd, c = y1.shape
    assert y1.is_contiguous() and w2p.numel() == 9 * c * c and y1.dtype == w2p.dtype
    if w3p is None:
        out = torch.empty_like(y1)
    else:
        assert w3p.numel() == 4 * c * c and tuple(residual.shape) == (n, h, wd, 4 * c) and residual.is_contiguous()
        out = torch.empty_like(residual)
    _launch("dh_bottleneck_tail_s3_nhwc", _ptr(y1), _ptr(w2p), _ptr(scale2), _ptr(shift2), _ptr(w3p), _ptr(scale3), _ptr(shift3),
            _ptr(residual), _ptr(out), n, h, wd, c, _dt(y1), _stream())
    return out


def stem_conv7_bn_relu_maxpool(x, wpk, scale, shift, out_dtype=None):
    """The ResNet stem as a direct matrix-core convolution (conv 7x7/2/3 + BN + ReLU + maxpool 3/2/1, one launch).
    x: fp32 NCHW [N,3,H,W] or 16-bit channels-last [N,H,W,8]; wpk [64,7,8,4] 16-bit (``pack_stem_weight``)."""
    _dev(x, wpk, scale, shift)
    assert x.is_contiguous() and wpk.is_contiguous() and tuple(wpk.shape) == (64, 7, 8, 4) and wpk.dtype in HALF_DTYPES
    if x.dtype == torch.float32:
        n, c, h, wd = x.shape
        assert c == 3
        fmt = 0
    else:
        n, h, wd, c = x.shape
        assert c == 8 and x.dtype == wpk.dtype
        fmt = 1
    ho, wo = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
    out = torch.empty((n, ho // 2, wo // 2, 64), dtype=wpk.dtype if out_dtype is None else out_dtype, device=x.device)
    _launch("dh_stem_conv7_bn_relu_maxpool", _ptr(x), fmt, _ptr(wpk), _ptr(scale), _ptr(shift), _ptr(out), n, h, wd, _dt(out), _stream())
    return out


def pack_stem_weight(w, dtype):
    """Checkpoint stem weight [64,3,7,7] -> [64, 7 kh, 8 kw slots, 4 channels] (slot 7 / channel 3 zero) in ``dtype``."""
    assert tuple(w.shape) == (64, 3, 7, 7)
    out = torch.zeros((64, 7, 8, 4), dtype=dtype, device=w.device)
    out[:, :, :7, :3] = w.detach().permute(0, 2, 3, 1).to(dtype)
    return out.contiguous()


def embed_rows(tok_emb, pos_emb, start_emb, tokens, x, rows, rows_per_img, row_mult, pos, scale):
    _dev(tok_emb, pos_emb, start_emb, tokens, x)
    d = tok_emb.shape[1]
    _launch("dh_embed_rows", _ptr(tok_emb), _ptr(pos_emb), _ptr(start_emb), _ptr(tokens),
                                tokens.stride(0) if tokens is not None else 0, _ptr(x), rows, rows_per_img,
                                row_mult, pos, d, float(scale), _dt(tok_emb), _stream())
    return x


def add_layernorm(x, y, gamma, beta, out=None, eps=1e-5):
    _dev(x, y, gamma, beta)
    rows, d = x.shape
    if out is None:
        out = torch.empty_like(x)
    _launch("dh_add_layernorm", _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(out), rows, d, float(eps),
                                   _dt(x), _stream())
    return out


def attn_self_decode(qkv, kcache, vcache, src, tokens, out, n_img, rows_per_img, row_mult, rows_total, t, d,
                     n_heads, scale, pad_index):
    _dev(qkv, kcache, vcache, src, tokens, out)
    _launch("dh_attn_self_decode", _ptr(qkv), _ptr(kcache), _ptr(vcache), _ptr(src), src.stride(0),
                                      _ptr(tokens), tokens.stride(0), _ptr(out), n_img, rows_per_img, row_mult,
                                      rows_total, t, d, n_heads, float(scale), pad_index, _dt(qkv), _stream())
    return out


def attn_cross_decode(q, kv, keymask, out, n_img, rows_per_img, s, d, n_heads, scale):
    _dev(q, kv, keymask, out)
    _launch("dh_attn_cross_decode", _ptr(q), q.stride(0), _ptr(kv), _ptr(keymask), _ptr(out), n_img,
                                       rows_per_img, s, d, n_heads, float(scale), _dt(q), _stream())
    return out


def embed_prefill(tok_emb, pos_emb, start_emb, tokens, n_seq, n_pos, scale):
    """Sequence-major rows [n_seq * n_pos, D] of all positions (teacher forcing)."""
    _dev(tok_emb, pos_emb, start_emb, tokens)
    d = tok_emb.shape[1]
    x = torch.empty((n_seq * n_pos, d), dtype=tok_emb.dtype, device=tok_emb.device)
    _launch("dh_embed_prefill", _ptr(tok_emb), _ptr(pos_emb), _ptr(start_emb), _ptr(tokens), tokens.stride(0), _ptr(x), n_seq,
            n_pos, d, float(scale), _dt(tok_emb), _stream())
    return x


def attn_self_prefill(qkv, tokens, n_seq, n_pos, d, n_heads, scale, pad_index):
    _dev(qkv, tokens)
    out = torch.empty((n_seq * n_pos, d), dtype=qkv.dtype, device=qkv.device)
    _launch("dh_attn_self_prefill", _ptr(qkv), _ptr(tokens), tokens.stride(0), _ptr(out), n_seq, n_pos, d, n_heads,
            float(scale), pad_index, _dt(qkv), _stream())
    return out


def attn_cross_prefill(q, kv, keymask, n_img, n_pos, s, d, n_heads, scale):
    _dev(q, kv, keymask)
    out = torch.empty((n_img * n_pos, d), dtype=q.dtype, device=q.device)
    _launch("dh_attn_cross_prefill", _ptr(q), q.stride(0), _ptr(kv), _ptr(keymask), _ptr(out), n_img, n_pos, s, d, n_heads,
            float(scale), _dt(q), _stream())
    return out


def enc_key_mask(enc_out):
    """enc_out [rows, D] -> uint8 [rows]"""
    _dev(enc_out)
    rows, d = enc_out.shape
    out = torch.empty((rows,), dtype=torch.uint8, device=enc_out.device)
    _launch("dh_enc_key_mask", _ptr(enc_out), _ptr(out), rows, d, _dt(enc_out), _stream())
    return out


def pad_mask(query, key, pad_index=0):
    """Reference ``get_pad_mask`` (transformers.py:12-26): bool [bs, query_len, key_len], True where key == pad_index."""
    _dev(query, key)
    bs, lq = query.shape[:2]
    lk = key.shape[1]
    key = key.to(torch.int64).contiguous()
    mask = torch.empty((bs, lq, lk), dtype=torch.uint8, device=key.device)
    _launch("dh_pad_mask", _ptr(key), _ptr(mask), bs, lq, lk, int(pad_index), _stream())
    return mask.view(torch.bool)


def autoregressive_mask(seq):
    """Reference ``get_autoregressive_mask`` (transformers.py:29-40): bool [bs, L, L], True above the diagonal."""
    _dev(seq)
    bs, l = seq.shape[:2]
    mask = torch.empty((bs, l, l), dtype=torch.uint8, device=seq.device)
    _launch("dh_autoregressive_mask", _ptr(mask), bs, l, _stream())
    return mask.view(torch.bool)


def mask_or(a, b):
    """a | b for two boolean masks of the same shape (new tensor)."""
    _dev(a, b)
    assert a.shape == b.shape
    out = a.contiguous().view(torch.uint8).clone()
    _launch("dh_mask_or", _ptr(out), _ptr(b.contiguous().view(torch.uint8)), out.numel(), _stream())
    return out.view(torch.bool)


def enc_nonzero_rows(enc_out):
    """enc_out [bs, L, D] -> int64 [bs, L]: 1 where no element of the row is 0 (transformers.py:480)."""
    _dev(enc_out)
    bs, l, d = enc_out.shape
    out = torch.empty((bs, l), dtype=torch.int64, device=enc_out.device)
    _launch("dh_enc_nonzero_rows", _ptr(enc_out.contiguous()), _ptr(out), bs * l, d, _dt(enc_out), _stream())
    return out


def attn_masked(q, k, v, mask, bs, l, d, n_heads, scale):
    """q, k, v projected rows [bs*l, d]; mask bool/uint8 [bs, l, l] or None -> [bs*l, d]."""
    _dev(q, k, v, mask)
    out = torch.empty((bs * l, d), dtype=q.dtype, device=q.device)
    if mask is not None:
        mask = mask.contiguous()
        mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        assert tuple(mask.shape) == (bs, l, l)
    _launch("dh_attn_masked", _ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(mask), _ptr(out),
            bs, l, d, n_heads, float(scale), _dt(q), _stream())
    return out


def lstm_prepare(emb, img_emb, tokens, tok_pos, hparent, h_prev, c_prev, xcat0, xcatl, c_cur, rows, rows_per_img,
                 row_mult, rows_total, n_layers, e, hh):
    _dev(emb, img_emb, tokens, hparent, h_prev, c_prev, xcat0, xcatl, c_cur)
    _launch("dh_lstm_prepare", _ptr(emb), _ptr(img_emb), _ptr(tokens),
                                  tokens.stride(0) if tokens is not None else 0, tok_pos, _ptr(hparent),
                                  _ptr(h_prev), _ptr(c_prev), _ptr(xcat0), _ptr(xcatl), _ptr(c_cur), rows,
                                  rows_per_img, row_mult, rows_total, n_layers, e, hh, _dt(xcat0), _stream())


def lstm_cell(gates, c_cur, h_new, c_new, h_out, ld_out, rows, row_mult, hh):
    _dev(gates, c_cur, h_new, c_new, h_out)
    _launch("dh_lstm_cell", _ptr(gates), _ptr(c_cur), _ptr(h_new), _ptr(c_new), _ptr(h_out), ld_out, rows,
                               row_mult, hh, _dt(h_new), _stream())


def beam_row_sample(logits, v, rows, rows_per_img, beam, top_k, temperature, unk_index, noise, seed, img0, step,
                    pick_idx, pick_val, err, seed_ptr=None, exact=False):
    """``exact``: the general kernel only (``dh_beam_row_sample_exact``): flat rows with more than 1,024 survivors are drawn over the
    whole row instead of flagging ERR_OVERFLOW."""
    _dev(logits, noise, pick_idx, pick_val, err)
    assert logits.dtype == torch.float32
    _launch("dh_beam_row_sample_exact" if exact else "dh_beam_row_sample", _ptr(logits), logits.stride(0), v, rows, rows_per_img, beam, top_k,
                                     float(temperature), unk_index, _ptr(noise), seed, _ptr(seed_ptr), img0, step,
                                     _ptr(pick_idx), _ptr(pick_val), _ptr(err), _stream())


def vocab_logits(a, w, bias, logits, group_max):
    """bf16 a [M,K], w [V,K] -> fp32 logits [M,V] + group_max [M, n_groups(V)] (max of every 64-column group)."""
    _dev(a, w, bias, logits, group_max)
    m, k = a.shape
    v = w.shape[0]
    _launch("dh_vocab_logits", _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(logits),
            logits.stride(0) if logits is not None else 0,
            _ptr(group_max), group_max.stride(0), m, v, k, _dt(a), _stream())


def pack_vocab_weights(w, bias):
    """Classifier weights ``[V, 512]`` (16-bit) and bias -> (fragment-packed weights padded to whole 256-column chunks with copies of
    row V - 1, fp32 bias padded likewise): the operands of ``vocab_logits_wreg``; made once per weight version."""
    _dev(w, bias)
    v = w.shape[0]
    vpad = (v + 255) // 256 * 256
    idx = torch.clamp(torch.arange(vpad, device=w.device), max=v - 1)
    frag = pack_mfma_fragments(w.index_select(0, idx).contiguous())           # [k-step][vpad / 16 tiles][64 lanes x 8 elements]
    k32 = w.shape[1] // 32
    frag = frag.view(k32, vpad // 256, 16, 512).permute(1, 0, 2, 3).contiguous().view(-1)      # chunk-major: 256 KB contiguous per 256 columns
    return frag, (bias.float().index_select(0, idx).contiguous() if bias is not None else None)


def vocab_logits_wreg_supported(m, v, k, ldl, gm_ld):
    return bool(load().dh_vocab_logits_wreg_supported(int(m), int(v), int(k), int(ldl), int(gm_ld)))


def vocab_logits_wreg(a, w_packed, bias_padded, v, logits, group_max):
    """``vocab_logits`` on the operands of ``pack_vocab_weights`` (csrc/vocab_wreg.hip): same logits and group maxima, bit for bit."""
    _dev(a, w_packed, bias_padded, logits, group_max)
    m, k = a.shape
    _launch("dh_vocab_logits_wreg", _ptr(a), a.stride(0), _ptr(w_packed), _ptr(bias_padded), _ptr(logits),
            logits.stride(0) if logits is not None else 0, _ptr(group_max), group_max.stride(0), m, v, k, _dt(a), _stream())


def vocab_logprob(a, w, bias, targets):
    """bf16 a [M,K], w [V,K], int64 targets [M] -> fp32 log_softmax(a @ w.T + bias)[targets] [M]; the [M,V] logits are
    never written (per-group log-sum-exp partials in the classifier GEMM's epilogue)."""
    _dev(a, w, bias, targets)
    m, k = a.shape
    v = w.shape[0]
    ng = n_groups(v)
    scratch = torch.empty((2, m, ng), dtype=torch.float32, device=a.device)
    tgt = torch.empty((m,), dtype=torch.float32, device=a.device)
    logp = torch.empty((m,), dtype=torch.float32, device=a.device)
    _launch("dh_vocab_logprob", _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(targets.contiguous()), _ptr(logp),
            _ptr(scratch[0]), _ptr(scratch[1]), _ptr(tgt), ng, m, v, k, _dt(a), _stream())
    return logp


def beam_row_sample_groups(logits, v, group_max, rows, rows_per_img, beam, top_k, temperature, unk_index, noise, seed,
                           img0, step, pick_idx, pick_val, err, seed_ptr=None):
    _dev(logits, group_max, noise, pick_idx, pick_val, err)
    _launch("dh_beam_row_sample_groups", _ptr(logits), logits.stride(0), v, _ptr(group_max), group_max.stride(0),
            n_groups(v), GROUP_COLS, rows, rows_per_img, beam, top_k, float(temperature), unk_index, _ptr(noise), seed,
            _ptr(seed_ptr), img0, step, _ptr(pick_idx), _ptr(pick_val), _ptr(err), _stream())




def beam_select(pick_idx, pick_val, tokens, vals, ended, src, parent, hparent, done, end_step, n_img, beam, first,
                first_sets_ended, write_pos, t, step_index, temperature, eos_index, noise, seed, img0, seed_ptr=None):
    _dev(pick_idx, pick_val, tokens, vals, ended, src, parent, hparent, done, end_step, noise)
    _launch("dh_beam_select", _ptr(pick_idx), _ptr(pick_val), _ptr(tokens), tokens.stride(0), _ptr(vals),
                                 _ptr(ended), _ptr(src), src.stride(0) if src is not None else 0, _ptr(parent),
                                 _ptr(hparent), _ptr(done), _ptr(end_step), n_img, beam, int(first),
                                 int(first_sets_ended), write_pos, t, step_index, float(temperature), eos_index,
                                 _ptr(noise), seed, _ptr(seed_ptr), img0, _stream())


def beam_filter_top_k(logits, top_k, unk_index):
    """beam.py:32-37 in place on fp32 ``logits [rows, V]`` (unit column stride; the row stride may exceed V)."""
    _dev(logits)
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    _launch("dh_beam_filter_top_k", _ptr(logits), logits.stride(0), logits.shape[1], logits.shape[0], top_k, unk_index, _stream())


def beam_sample_k(x, k, temperature, noise, seed, stream_id, draw, out, err, seed_ptr=None):
    """beam.py:39-48: ``out [rows, k]`` int64 = multinomial(softmax(x / T), k) as an Exp(1) race."""
    _dev(x, noise, out, err)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.int64
    _launch("dh_beam_sample_k", _ptr(x), x.stride(0), x.shape[1], x.shape[0], k, float(temperature), _ptr(noise),
            0 if noise is None else noise.stride(0), seed, _ptr(seed_ptr), stream_id, draw, _ptr(out), _ptr(err), _stream())


def beam_gather(values, indices, out, err=None):
    """beam.py:50-53: ``out[r, j] = values[r, indices[r, j]]``."""
    _dev(values, indices, out, err)
    assert values.dtype == torch.float32 and values.stride(1) == 1 and indices.dtype == torch.int64 and indices.is_contiguous()
    _launch("dh_beam_gather", _ptr(values), values.stride(0), values.shape[1], _ptr(indices), indices.shape[1], _ptr(out),
            values.shape[0], _ptr(err), _stream())


def beam_expand(new_ind, gathered, ended, seqs, vals, beam, eos_index, prev_seqs, prev_vals, out_ind, out_val, out_ended):
    """beam.py:78-106 (log-softmax of the picks + candidate expansion by the has_ended flags)."""
    _dev(new_ind, gathered, ended, seqs, vals, prev_seqs, prev_vals, out_ind, out_val, out_ended)
    n = ended.shape[0]
    _launch("dh_beam_expand", _ptr(new_ind), _ptr(gathered), _ptr(ended), _ptr(seqs), seqs.shape[1], _ptr(vals),
            vals.numel() // n, n, beam, eos_index, _ptr(prev_seqs), _ptr(prev_vals), _ptr(out_ind), _ptr(out_val), _ptr(out_ended),
            _stream())


def beam_finalize(tokens, vals, done, end_step, out, out_len, n_img, beam, len_bias_done, full_len, pad_index,
                  temperature, noise, seed, img0, seed_ptr=None):
    _dev(tokens, vals, done, end_step, out, out_len, noise)
    _launch("dh_beam_finalize", _ptr(tokens), tokens.stride(0), _ptr(vals), _ptr(done), _ptr(end_step),
                                   _ptr(out), out.stride(0), _ptr(out_len), n_img, beam, len_bias_done, full_len,
                                   pad_index, float(temperature), _ptr(noise), seed, _ptr(seed_ptr), img0, _stream())


GROUP_COLS = 64     # column-group width of dh_vocab_logits' group maxima


def n_groups(v):
    return 2 * ((v + 127) // 128)


def transformer_decode_position(model, scratch, start_emb, tokens, src, n_img, rows_per_img, row_mult, rows_total, t,
                                x_out=None, logits=None, group_max=None):
    _launch("dh_transformer_decode_position", _c.byref(model), _c.byref(scratch), _ptr(start_emb), _ptr(tokens),
            tokens.stride(0), _ptr(src), src.stride(0), n_img, rows_per_img, row_mult, rows_total, t, _ptr(x_out),
            _ptr(logits), logits.stride(0) if logits is not None else 0, _ptr(group_max),
            group_max.stride(0) if group_max is not None else 0, _stream())


def decode_layers_supported(model, rows_per_img, t):
    """``dh_decode_layers_supported`` for a ``TrModel`` description."""
    return bool(load().dh_decode_layers_supported(_c.byref(model), int(rows_per_img), int(t)))


def decode_layers_table(model, device):
    """The device-resident per-layer table of ``dh_decode_layers`` for this ``TrModel`` description (``dh_decode_layers_table``)."""
    nbytes = load().dh_decode_layers_table_bytes(model.n_layers)
    table = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    _launch("dh_decode_layers_table", _c.byref(model), _ptr(table), _stream())
    return table


def conv1x1_dual_nhwc(y, x, w_cat, shift, stride, relu=True):
    """relu(y (*) W3' + x[strided] (*) Wd' + shift): conv3 + downsample of a stage's first bottleneck as one GEMM
    (bf16, NHWC; ``w_cat`` [Cout, C1 + C2] has the BatchNorm scales folded in)."""
    _dev(y, x, w_cat, shift)
    n, ho, wo, c1 = y.shape
    _, h, w_, c2 = x.shape
    cout = w_cat.shape[0]
    out = torch.empty((n, ho, wo, cout), dtype=y.dtype, device=y.device)
    _launch("dh_conv1x1_dual_nhwc", _ptr(y), _ptr(x), _ptr(w_cat), _ptr(shift), _ptr(out), n, ho, wo, c1, h, w_, c2, stride,
            cout, int(relu), _dt(y), _stream())
    return out


def conv1x1_dual_wreg_supported(y_shape, x_shape, cout):
    """Whether ``conv1x1_dual_wreg_nhwc`` takes ``y [N, Ho, Wo, C1]`` + ``x [N, H, W, C2]`` -> ``Cout`` channels."""
    n, ho, wo, c1 = y_shape
    _, h, w, c2 = x_shape
    return bool(load().dh_conv1x1_dual_wreg_supported(int(n), int(ho), int(wo), int(h), int(w), int(c1), int(c2), int(cout)))


def conv1x1_dual_wreg_nhwc(y, x, w_packed, cout, shift, stride, relu=True, w1p=None, scale1=None, shift1=None, n1=0):
    """``conv1x1_dual_nhwc`` with the weights stationary in registers and the pixels of both sources streamed
    (``w_packed = pack_mfma_fragments(w_cat)``; csrc/conv1x1_wreg.hip).  Bit-identical to ``conv1x1_dual_nhwc``.  With ``w1p``
    (stage 1: K = 128, ``cout`` = 256, ``n1`` = 64) also the NEXT bottleneck's conv1 + bn1 + relu on the outputs while they are in
    LDS: returns ``(out, y1_next)``."""
    _dev(y, x, w_packed, shift, w1p, scale1, shift1)
    n, ho, wo, c1 = y.shape
    _, h, w_, c2 = x.shape
    out = torch.empty((n, ho, wo, cout), dtype=y.dtype, device=y.device)
    y1n = torch.empty((n, ho, wo, n1), dtype=y.dtype, device=y.device) if w1p is not None else None
    _launch("dh_conv1x1_dual_wreg_nhwc", _ptr(y), _ptr(x), _ptr(w_packed), _ptr(shift), _ptr(out), n, ho, wo, c1, h, w_, c2, stride,
            cout, int(relu), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(y1n), n1, _dt(y), _stream())
    return out if w1p is None else (out, y1n)


def normalize_u8_hwc(x, mean, std):
    """uint8 [N,H,W,C] -> fp32 [N,C,H,W] (x / 255 - mean) / std, as torchvision ToTensor + Normalize."""
    _dev(x, mean, std)
    assert x.dtype == torch.uint8 and x.is_contiguous() and x.dim() == 4
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    _launch("dh_normalize_u8_hwc", _ptr(x), _ptr(mean), _ptr(std), _ptr(y), n, h, w, c, _stream())
    return y


def resize_u8_hwc(x, out_h, out_w, coeff_x, coeff_y):
    """uint8 [N,H,W,C] -> uint8 [N,out_h,out_w,C]: Pillow's BILINEAR resample (``transforms.Resize``), bit-exact.
    ``coeff_*`` = (bounds int32 [n_out,2], weights int32 [n_out,ksize]) device tensors, or None when that size is unchanged."""
    _dev(x)
    assert x.dtype == torch.uint8 and x.is_contiguous() and x.dim() == 4
    n, h, w, c = x.shape
    dst = torch.empty((n, out_h, out_w, c), dtype=torch.uint8, device=x.device)
    tmp = (torch.empty((n * max(h * out_w, out_h * w) * c,), dtype=torch.uint8, device=x.device)      # either pass order (header)
           if (h != out_h and w != out_w) else None)
    bx, kx = coeff_x if coeff_x is not None else (None, None)
    by, ky = coeff_y if coeff_y is not None else (None, None)
    _launch("dh_resize_u8_hwc", _ptr(x), _ptr(tmp), _ptr(dst), _ptr(bx), _ptr(kx), kx.shape[1] if kx is not None else 0,
            _ptr(by), _ptr(ky), ky.shape[1] if ky is not None else 0, n, h, w, out_h, out_w, c, _stream())
    return dst


def normalize_pack_u8(x, mean, std, out_dtype=torch.bfloat16):
    """uint8 [N,H,W,C<=8] -> normalised 16-bit channels-last [N,H,W,8] (the stem convolution's packed input)."""
    _dev(x, mean, std)
    assert x.dtype == torch.uint8 and x.is_contiguous() and x.dim() == 4
    n, h, w, c = x.shape
    y = torch.empty((n, h, w, 8), dtype=out_dtype, device=x.device)
    _launch("dh_normalize_pack_u8", _ptr(x), _ptr(mean), _ptr(std), _ptr(y), n, h, w, c, _dt(y), _stream())
    return y


def lstm_layer_fused(x_rows, x_div, emb, tokens, tok_pos, h_prev, c_prev, hparent, h_next, c_next, h_out, w_il, b_il,
                     rows, row_mult, e, hh):
    """One LSTM layer time step in one launch (bf16, gate-interleaved weights); see include/deephumor_hip.h."""
    _dev(h_next, c_next, h_out, w_il, b_il)
    _launch("dh_lstm_layer_fused", _ptr(x_rows), x_rows.stride(0) if x_rows is not None else 0, x_div, _ptr(emb),
            _ptr(tokens), tokens.stride(0) if tokens is not None else 0, tok_pos, _ptr(h_prev), _ptr(c_prev),
            _ptr(hparent), _ptr(h_next), _ptr(c_next), _ptr(h_out), h_out.stride(0), _ptr(w_il), _ptr(b_il), rows,
            row_mult, e, hh, _dt(w_il), _stream())


def lstm_layer_wreg_supported(e, hh):
    return bool(load().dh_lstm_layer_wreg_supported(int(e), int(hh)))


def lstm_layer_wreg(x_rows, x_div, emb, tokens, tok_pos, h_prev, c_prev, hparent, h_next, c_next, h_out, w_pk, b_il,
                    rows, row_mult, e, hh):
    """``lstm_layer_fused`` with the gate weights stationary in registers (``w_pk = pack_mfma_fragments(w_il)``)."""
    _dev(h_next, c_next, h_out, w_pk, b_il)
    _launch("dh_lstm_layer_wreg", _ptr(x_rows), x_rows.stride(0) if x_rows is not None else 0, x_div, _ptr(emb),
            _ptr(tokens), tokens.stride(0) if tokens is not None else 0, tok_pos, _ptr(h_prev), _ptr(c_prev),
            _ptr(hparent), _ptr(h_next), _ptr(c_next), _ptr(h_out), h_out.stride(0), _ptr(w_pk), _ptr(b_il), rows,
            row_mult, e, hh, _dt(w_pk), _stream())


def lstm_decode_step(model, scratch, img_emb, tokens, tok_pos, hparent, started, rows, rows_per_img, row_mult,
                     rows_total, h_out=None, logits=None, group_max=None):
    _launch("dh_lstm_decode_step", _c.byref(model), _c.byref(scratch), _ptr(img_emb), _ptr(tokens),
            tokens.stride(0) if tokens is not None else 0, tok_pos, _ptr(hparent), int(started), rows, rows_per_img,
            row_mult, rows_total, _ptr(h_out), h_out.stride(0) if h_out is not None else 0, _ptr(logits),
            logits.stride(0) if logits is not None else 0, _ptr(group_max),
            group_max.stride(0) if group_max is not None else 0, _stream())


def token_logprob(logits, targets):
    """logits fp32 [rows, V], targets int64 [rows] -> log_softmax(logits)[targets] fp32 [rows]."""
    _dev(logits, targets)
    assert logits.dtype == torch.float32 and targets.dtype == torch.int64 and logits.stride(1) == 1
    rows, v = logits.shape
    out = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    _launch("dh_token_logprob", _ptr(logits), logits.stride(0), v, _ptr(targets.contiguous()), _ptr(out), rows, _stream())
    return out


def seq_perplexity(logp, targets, lengths, pad_index=0):
    """logp fp32 [n, L], targets int64 [n, L], lengths int64 [n] -> per-sequence perplexity fp32 [n]."""
    _dev(logp, targets, lengths)
    n, l = targets.shape
    out = torch.empty((n,), dtype=torch.float32, device=logp.device)
    _launch("dh_seq_perplexity", _ptr(logp.contiguous()), _ptr(targets.contiguous()), _ptr(lengths.contiguous()),
            _ptr(out), n, l, pad_index, _stream())
    return out
