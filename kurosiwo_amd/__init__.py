"""kurosiwo_amd — MI355X-native (gfx950) training hot path of Kuro Siwo.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all math runs in hand-written HIP kernels behind the C-ABI of include/ksmi.h
(libksmi.so).  There is no CPU fallback: ops raise if the library is missing or
a tensor is not on the GPU.
"""
from ._lib import KsmiError, load as load_library  # noqa: F401

__all__ = ["KsmiError", "load_library"]
