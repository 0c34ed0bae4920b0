"""Python side of the split-operand fp32 kernels that take activations STORED as their two fp16 planes (``csrc/gemm_f32xp.hip``,
``csrc/linear_f32x_wreg.hip``, the planes outputs of LayerNorm and the decode attentions).

A planes tensor is ``[2, ...]`` fp16: index 0 = hi = fp16(x), index 1 = lo = fp16((x - hi) * 2^11) -- what ``dh_linear_f32x`` makes of
an fp32 operand in registers, here made ONCE by the producer.  Same products, same sums: bit-identical results
(``tests/test_f32x_gpu.py``).  Replaces the same reference call sites as ``hip.linear_f32x`` / ``hip.linear_f32x_wreg``
(rnn_models.py:45; transformers.py:97,127,162-163,489).
"""
import torch

from . import hip


def split_act(a, tag=None):
    """fp32 ``a [M, K]`` -> planes ``[2, M, Kp]`` (``dh_split_act_f32x``; range-guarded like the kernels' own splits)."""
    hip._dev(a)
    assert a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
    if a.stride(0) % 4 or a.data_ptr() % 16:
        a = a.contiguous()
    m, k = a.shape
    kp = (k + 31) // 32 * 32
    planes = torch.empty((2, m, kp), dtype=torch.float16, device=a.device)
    hip._launch("dh_split_act_f32x", hip._ptr(a), a.stride(0), hip._ptr(planes), m, k, kp, hip._stream(), tag=tag)
    return planes


def join(planes):
    """planes -> fp32 (hi + lo * 2^-11): test helper / debugging; exact for values that were produced by a split."""
    return planes[0].float() + planes[1].float() * (1.0 / 2048.0)


def linear(a_planes, w_planes, bias=None, scale=None, shift=None, relu=False, residual=None, out=None, out_planes=False, group_max=None,
           tag=None):
    """``dh_linear_f32xp``: ``a_planes [2, M, Kp]`` x ``w_planes [2, N, Kp]`` -> fp32 ``[M, N]`` (``out`` may be a wider pre-allocated
    buffer), planes ``[2, M, N]`` with ``out_planes=True`` (``"only"``: no fp32 output), and the 64-column group maxima into ``group_max``."""
    hip._dev(a_planes, w_planes, bias, scale, shift, residual, out, group_max)
    two, m, kp = a_planes.shape
    n = w_planes.shape[1]
    assert two == 2 and w_planes.shape[2] == kp and a_planes.dtype == torch.float16 and w_planes.dtype == torch.float16
    assert a_planes.is_contiguous() and w_planes.is_contiguous()
    if out is None and out_planes != "only":
        out = torch.empty((m, n), dtype=torch.float32, device=a_planes.device)
    cp = torch.empty((2, m, n), dtype=torch.float16, device=a_planes.device) if out_planes else None
    hip._launch("dh_linear_f32xp", hip._ptr(a_planes), hip._ptr(w_planes), kp, hip._ptr(bias), hip._ptr(scale), hip._ptr(shift),
                hip._ptr(residual), residual.stride(0) if residual is not None else 0, hip._ptr(out), out.stride(0) if out is not None else 0,
                hip._ptr(cp), hip._ptr(group_max), group_max.stride(0) if group_max is not None else 0, m, n, int(relu), hip._stream(), tag=tag)
    return (out, cp) if out_planes is True else cp if out_planes else out


def linear_wreg(a_planes, packed, bias, relu=False, residual=None, out=None, want="f32", tag=None):
    """``dh_linear_f32xp_wreg``: ``hip.linear_f32x_wreg`` for an activation stored as planes ``[2, M, K]``; ``want`` = "f32", "planes"
    (``[2, M, N]``, no fp32 output) or "both"."""
    hip._dev(a_planes, packed, bias, residual, out)
    two, m, k = a_planes.shape
    n = packed.shape[2] * 16
    assert two == 2 and a_planes.is_contiguous() and a_planes.dtype == torch.float16 and packed.shape[1] * 32 == k
    if out is None and want != "planes":
        out = torch.empty((m, n), dtype=torch.float32, device=a_planes.device)
    cp = torch.empty((2, m, n), dtype=torch.float16, device=a_planes.device) if want != "f32" else None
    hip._launch("dh_linear_f32xp_wreg", hip._ptr(a_planes), hip._ptr(packed), hip._ptr(bias), hip._ptr(residual),
                residual.stride(0) if residual is not None else 0, hip._ptr(out), out.stride(0) if out is not None else 0, hip._ptr(cp),
                m, n, k, int(relu), hip._stream(), tag=tag)
    return (out, cp) if want == "both" else cp if want == "planes" else out


def add_layernorm(x, y, gamma, beta, eps=1e-5):
    """``dh_add_layernorm_f32x``: ``LayerNorm(x + y)`` of fp32 rows -> (fp32 ``[rows, D]``, planes ``[2, rows, D]``)."""
    hip._dev(x, y, gamma, beta)
    rows, d = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and (y is None or y.is_contiguous())
    out = torch.empty_like(x)
    planes = torch.empty((2, rows, d), dtype=torch.float16, device=x.device)
    hip._launch("dh_add_layernorm_f32x", hip._ptr(x), hip._ptr(y), hip._ptr(gamma), hip._ptr(beta), hip._ptr(out), hip._ptr(planes), rows, d,
                float(eps), hip._stream())
    return out, planes


DH_F32_OUT_PLANES = 5


def attn_self_decode_planes(qkv, kcache, vcache, src, tokens, n_img, rows_per_img, row_mult, rows_total, t, d, n_heads, scale, pad_index):
    """``dh_attn_self_decode`` with dtype ``DH_F32_OUT_PLANES``: fp32 operands, the result as planes ``[2, rows, D]``."""
    hip._dev(qkv, kcache, vcache, src, tokens)
    out = torch.empty((2, n_img * rows_per_img, d), dtype=torch.float16, device=qkv.device)
    hip._launch("dh_attn_self_decode", hip._ptr(qkv), hip._ptr(kcache), hip._ptr(vcache), hip._ptr(src), src.stride(0), hip._ptr(tokens),
                tokens.stride(0), hip._ptr(out), n_img, rows_per_img, row_mult, rows_total, t, d, n_heads, float(scale), pad_index,
                DH_F32_OUT_PLANES, hip._stream())
    return out


def attn_cross_decode_planes(q, kv, keymask, n_img, rows_per_img, s, d, n_heads, scale):
    hip._dev(q, kv, keymask)
    out = torch.empty((2, n_img * rows_per_img, d), dtype=torch.float16, device=q.device)
    hip._launch("dh_attn_cross_decode", hip._ptr(q), q.stride(0), hip._ptr(kv), hip._ptr(keymask), hip._ptr(out), n_img, rows_per_img, s, d,
                n_heads, float(scale), DH_F32_OUT_PLANES, hip._stream())
    return out


def pack_conv1x1(w_planes):
    """The planes of ``hip.split_f32x(w [Cout, Cin])`` in MFMA fragment order ``[2, Cin / 32, Cout / 16, 64, 8]`` for
    ``conv1x1_stream``; None when the layer is not one the streaming kernel takes at any row count."""
    hip._dev(w_planes)
    two, n, kp = w_planes.shape
    if two != 2 or not hip.load().dh_conv1x1_f32x_stream_supported(1 << 30, kp, n):
        return None
    out = torch.empty((2, kp // 32, n // 16, 64, 8), dtype=torch.float16, device=w_planes.device)
    for i in range(2):
        hip._launch("dh_pack_mfma_fragments", hip._ptr(w_planes[i]), hip._ptr(out[i]), n, kp, hip._stream())
    return out


def conv1x1_stream_supported(m, cin, cout):
    return bool(hip.load().dh_conv1x1_f32x_stream_supported(m, cin, cout))


def conv1x1_stream(x, packed, scale, shift, residual=None, relu=True):
    """``dh_conv1x1_f32x_stream``: channels-last fp32 ``x [N, H, W, Cin]`` -> ``[N, H, W, Cout]``; bit-identical to
    ``hip.conv2d_nhwc_f32x`` of the same 1 x 1 / stride 1 layer."""
    hip._dev(x, packed, scale, shift, residual)
    n, h, w, cin = x.shape
    cout = packed.shape[2] * 16
    assert x.dtype == torch.float32 and x.is_contiguous() and packed.shape[1] * 32 == cin and (residual is None or residual.is_contiguous())
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    hip._launch("dh_conv1x1_f32x_stream", hip._ptr(x), hip._ptr(packed), hip._ptr(scale), hip._ptr(shift), hip._ptr(residual), hip._ptr(y),
                n * h * w, cin, cout, int(relu), hip._stream())
    return y


def conv2d_nhwc_planes_out(x, w_planes, ks, scale, shift, relu=True, stride=1, pad=0):
    """``dh_conv2d_nhwc_f32x_planes_out``: ``hip.conv2d_nhwc_f32x`` (fp32 ``x [N, H, W, Cin]``) with the result as planes
    ``[2, N, Ho, Wo, Cout]``."""
    hip._dev(x, w_planes, scale, shift)
    n, h, w, cin = x.shape
    cout, kp = w_planes.shape[1], w_planes.shape[2]
    assert x.dtype == torch.float32 and x.is_contiguous() and kp == (ks * ks * cin + 31) // 32 * 32
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    yp = torch.empty((2, n, ho, wo, cout), dtype=torch.float16, device=x.device)
    hip._launch("dh_conv2d_nhwc_f32x_planes_out", hip._ptr(x), hip._ptr(w_planes), kp, hip._ptr(scale), hip._ptr(shift), hip._ptr(yp), n, h, w,
                cin, cout, ks, stride, pad, int(relu), hip._stream())
    return yp


def conv2d_nhwc(x_planes, w_planes, ks, scale, shift, residual=None, relu=True, stride=1, pad=0, want="f32"):
    """``dh_conv2d_nhwc_f32xp``: ``x_planes [2, N, H, W, Cin]`` -> ``want`` = "f32" (fp32 ``[N, Ho, Wo, Cout]``), "planes", or "both"."""
    hip._dev(x_planes, w_planes, scale, shift, residual)
    two, n, h, w, cin = x_planes.shape
    cout, kp = w_planes.shape[1], w_planes.shape[2]
    assert two == 2 and x_planes.is_contiguous() and kp == ks * ks * cin and cin % 32 == 0
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x_planes.device) if want != "planes" else None
    yp = torch.empty((2, n, ho, wo, cout), dtype=torch.float16, device=x_planes.device) if want != "f32" else None
    hip._launch("dh_conv2d_nhwc_f32xp", hip._ptr(x_planes), hip._ptr(w_planes), hip._ptr(scale), hip._ptr(shift), hip._ptr(residual),
                hip._ptr(y), hip._ptr(yp), n, h, w, cin, cout, ks, stride, pad, int(relu), hip._stream())
    return (y, yp) if want == "both" else yp if want == "planes" else y
