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
    land beside an env created with an explicit ``device=``; else the GPU of the LIVE envs created with an explicit ``device=`` when
    they all sit on one GPU (a process that drives several GPUs gives no such hint: its helpers take the fallback below, or the
    caller selects the device with ``torch.cuda.device``); else this rank's own GPU (LOCAL_RANK, one process per GPU) when the
    process sees several; else device 0."""
    import os
    if not torch.cuda.is_available():
        return 0
    cur = torch.cuda.current_device()
    if cur != 0:
        return cur
    hinted = set(_env_devices.values())
    if len(hinted) == 1:                      # every live env with an explicit device= is on this GPU (helpers serve those envs)
        return next(iter(hinted))
    if torch.cuda.device_count() > 1 and "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    return cur


_env_devices = {}          # id(env) -> GPU index, for live envs created with an explicit device=


def note_env_device(env_key: int, index: int) -> None:
    """Called by the env constructor when the caller named its GPU: helper handles created afterwards without a device of their own
    (trajectory evaluation, DSLPID) follow it instead of LOCAL_RANK (a rehearsal that puts every rank on device 0 would otherwise get
    cross-device tensors).  Keyed per env and dropped by ``forget_env_device`` on ``env.close()``: a closed env leaves no hint behind,
    and envs on different GPUs cancel the hint instead of sending every helper to the GPU of the env created last."""
    _env_devices[int(env_key)] = int(index)


def forget_env_device(env_key: int) -> None:
    _env_devices.pop(int(env_key), None)


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
