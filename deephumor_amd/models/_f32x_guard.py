"""Range guard of the split-operand fp32 path (option ``f32_split``): the decorator the models put on ``forward`` /
``generate_batch``."""
import threading

import torch

from .. import hip

_f32x_tls = threading.local()
_f32x_warned = [False]


def f32x_guarded(fn):
    """Decorator of the models' ``forward`` / ``generate_batch``: with option ``f32_split`` on an fp32 CUDA model, the OUTERMOST guarded
    call reads the stream's range word once it returns (one host read -- the opt-in path only) and, if an activation left the fp16
    range, repeats the call with the option off: the exact-fp32 kernels, same RNG state.  ADVICE r5: the split silently returned
    inf / NaN for |x| >= 65504."""
    import functools
    import warnings

    @functools.wraps(fn)
    def wrapped(self, *args, **kw):
        par = next(self.parameters(), None)
        if (getattr(_f32x_tls, "depth", 0) or par is None or par.dtype != torch.float32 or not par.is_cuda or not hip.option("f32_split")
                or torch.cuda.is_current_stream_capturing()):
            return fn(self, *args, **kw)
        _f32x_tls.depth = 1
        rng = torch.get_rng_state()
        try:
            out = fn(self, *args, **kw)
            with torch.cuda.device(par.device):
                over = hip.f32x_take_overflow(par.device)
        finally:
            _f32x_tls.depth = 0
        if not over:
            return out
        if not _f32x_warned[0]:
            _f32x_warned[0] = True
            warnings.warn("deephumor_amd: an activation left the fp16 range of the split-operand fp32 path (option f32_split); this call "
                          "is repeated on the exact-fp32 kernels", RuntimeWarning)
        torch.set_rng_state(rng)
        with hip.option_scope(f32_split=0):
            return fn(self, *args, **kw)
    return wrapped
