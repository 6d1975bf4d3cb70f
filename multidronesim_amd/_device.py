"""Device-tensor plumbing (PyTorch-ROCm is used for buffers and streams only)."""
from __future__ import annotations

import numpy as np
import torch

from . import _capi as capi

TORCH_DTYPE = {capi.MDS_F32: torch.float32, capi.MDS_F64: torch.float64, capi.MDS_F16: torch.float16, capi.MDS_F32C: torch.float32}
DTYPE_BY_NAME = {"float32": capi.MDS_F32, "fp32": capi.MDS_F32, "f32": capi.MDS_F32, torch.float32: capi.MDS_F32,
                 "float64": capi.MDS_F64, "fp64": capi.MDS_F64, "f64": capi.MDS_F64, torch.float64: capi.MDS_F64,
                 "float32c": capi.MDS_F32C, "fp32c": capi.MDS_F32C, "f32c": capi.MDS_F32C,      # fp32 with compensated state accumulation
                 "float16": capi.MDS_F16, "fp16": capi.MDS_F16, "f16": capi.MDS_F16, torch.float16: capi.MDS_F16}


def require_gpu(device_index: int) -> torch.device:
    if not torch.cuda.is_available():
        raise capi.MdsError(-3, "multidronesim_amd", "no HIP device visible: the batched step only runs on a GPU "
                            "(there is no CPU fallback)")
    return torch.device("cuda", device_index)


def default_device_index() -> int:
    """The GPU a handle is created on when the caller names none: this rank's own (LOCAL_RANK, one process per GPU) when the
    process sees several, else the thread's current torch device."""
    import os
    if not torch.cuda.is_available():
        return 0
    if torch.cuda.device_count() > 1 and "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    return torch.cuda.current_device()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def to_device(x, device, dtype) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    a = np.asarray(x)
    if not (a.flags.c_contiguous and a.flags.writeable):
        a = np.array(a, order="C")          # broadcast views etc.: torch wants a writable buffer
    return torch.as_tensor(a, dtype=dtype).to(device)
