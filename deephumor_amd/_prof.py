"""The in-library launch profiler's Python face (``dh_prof_begin`` / ``dh_prof_end`` / ``dh_prof_get``), used through ``hip.profile``."""
import ctypes

from . import hip


class Profiler:
    """Per-entry-point timing with HIP events recorded INSIDE the library on the launch stream
    (``dh_prof_begin`` / ``dh_prof_end``), so launches made by the native step drivers are seen too.
    ``with hip.profile(watch={...}) as prof`` ... ``prof.summary()`` -> {"entry[tag]": calls, ms, flops, bytes}
    with the algorithmic flops/bytes the library attaches to each launch.  Off otherwise."""

    def __init__(self, watch=None, stride=1):
        self.watch = None if watch is None else sorted(watch)
        self.stride = stride
        self._summary = None

    def __enter__(self):
        self._prev, hip._prof = hip._prof, self
        hip._check(hip.load().dh_prof_set_stride(self.stride), "dh_prof_set_stride")
        hip._check(hip.load().dh_prof_begin(",".join(self.watch).encode() if self.watch else None), "dh_prof_begin")
        return self

    def __exit__(self, *exc):
        hip._prof = self._prev
        self.summary()

    def summary(self):
        if self._summary is None:
            lib = hip.load()
            hip._check(lib.dh_prof_end(), "dh_prof_end")
            out = {}
            name = ctypes.create_string_buffer(96)
            calls, ms, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            for i in range(lib.dh_prof_num()):
                hip._check(lib.dh_prof_get(i, name, 96, ctypes.byref(calls), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), "dh_prof_get")
                out[name.value.decode()] = dict(calls=calls.value, ms=ms.value, flops=fl.value, bytes=by.value)
            self._summary = out
        return self._summary
