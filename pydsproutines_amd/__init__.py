"""pydsproutines_amd: MI355X-native CAF / matched-filter engine behind the call signatures of
icyveins7/pydsproutines' correlation routines.  Python host -> ctypes C-ABI (libcaf.so) ->
hand-written gfx950 HIP kernels + batched rocFFT.  No CPU fallback, no cupy/Triton."""

from . import _lib  # noqa: F401
from .caf import CAFPlan, CAFResult  # noqa: F401
from .devarray import DeviceArray, asarray, asnumpy, empty, zeros  # noqa: F401

__version__ = "0.1.0"
