"""deephumor_amd: MI355X-native (gfx950) implementation of deephumor's image -> caption hot path.

``deephumor_amd.models`` mirrors ``deephumor.models``; the arithmetic runs in the hand-written HIP
kernels behind the C-ABI of ``include/deephumor_hip.h`` (``deephumor_amd.hip``).
"""
__version__ = "0.1.0"
