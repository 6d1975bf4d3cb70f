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
    """The GPU a handle is created on when the caller names none.  The thread's current torch device when the caller has chosen
    one (torch.cuda.set_device / a device context: anything but device 0), so that helper handles (trajectory evaluation, DSLPID)
    land beside an env created with an explicit ``device=``; else this rank's own GPU (LOCAL_RANK, one process per GPU) when the
    process sees several; else device 0.  An env created with an explicit ``device=`` takes precedence over LOCAL_RANK."""
    import os
    if not torch.cuda.is_available():
        return 0
    cur = torch.cuda.current_device()
    if cur != 0:
        return cur
    if _last_env_device is not None:          # the GPU of the env created last with an explicit device= (helpers serve that env)
        return _last_env_device
    if torch.cuda.device_count() > 1 and "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    return cur


_last_env_device = None


def note_env_device(index: int) -> None:
    """Called by the env constructor when the caller named its GPU: helper handles created afterwards without a device of their own
    (trajectory evaluation, DSLPID) follow it instead of LOCAL_RANK (a rehearsal that puts every rank on device 0, or a process that
    drives several GPUs, would otherwise get cross-device tensors)."""
    global _last_env_device
    _last_env_device = int(index)


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def to_device(x, device, dtype) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=dtype).contiguous()
        return t.clone() if t.data_ptr() % 16 else t      # a view that starts in mid-allocation (e.g. odd rows of an fp16 log): the C-ABI wants 16 bytes
    a = np.asarray(x)
    if not (a.flags.c_contiguous and a.flags.writeable):
        a = np.array(a, order="C")          # broadcast views etc.: torch wants a writable buffer
    return torch.as_tensor(a, dtype=dtype).to(device)
