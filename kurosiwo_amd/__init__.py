"""kurosiwo_amd — MI355X-native (gfx950) training hot path of Kuro Siwo.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all math runs in hand-written HIP kernels behind the C-ABI of include/ksmi.h
(libksmi.so).  There is no CPU fallback: ops raise if the library is missing or
a tensor is not on the GPU.
"""
import os as _os

# Data-parallel runs (torchrun: WORLD_SIZE > 1; KSMI_DP_FORCE: the one-rank RCCL smoke configuration) add two streams to the three of
# the train step (the bucket issue stream and ProcessGroupNCCL's own).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4): with five streams on four queues two streams of the step share a queue and lose their concurrency -- measured on
# MI355X, SNUNet bs 32, one-rank RCCL: +7 % step time at 4 or 5 queues, +2 % at 6 or 7, +14 % at 8 (profiles/r05_dp_hw_queues.txt).
# The runtime reads the variable when it initialises, i.e. at the first device call after this import.
if int(_os.environ.get("WORLD_SIZE", "1")) > 1 or _os.environ.get("KSMI_DP_FORCE"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")

from ._lib import KsmiError, load as load_library  # noqa: F401,E402

__all__ = ["KsmiError", "load_library"]
