"""Build-container-only stand-in so that ``/root/reference`` (which does
``from torchvision import models``, encoders.py:4) can be imported where torchvision is absent.

TEST INFRASTRUCTURE ONLY.  Used solely by ``oracle/make_golden.py`` to run the real reference
and record golden vectors.  Never imported by the product package, never needed on the GPU box.
"""
from . import models  # noqa: F401
