"""svi_mapper_amd — MI355X-native hot path of svi_mapper (descriptor matcher + LM bundle adjustment).

The package is a host-side mirror of the reference's interfaces for this path over the C ABI of
include/svi_hot.h (libsvi_hot.so: hand-written HIP kernels for gfx950). Importing it loads the
library; there is no CPU fallback (a missing library is an ImportError, a missing GPU makes every
compute call raise SviError(SVI_ERR_NO_DEVICE)).
"""
from ._capi import SviError, load_library  # noqa: F401

load_library()

from .matcher import DMatch, HammingMatcher, NoMatchFound, Triangulator  # noqa: E402,F401
from .optimizer import BundleAdjuster  # noqa: E402,F401
