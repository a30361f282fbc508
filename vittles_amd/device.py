"""PyTorch-ROCm plumbing: device memory ownership, streams, type round-trips.

torch is used for tensors/streams/torch.distributed only; all arithmetic of the
hot path happens in the HIP kernels behind ``_lib``.
"""
import numpy as np
import torch

from . import _lib


def require_gpu():
    """Fail loudly when the HIP path cannot run (no silent CPU fallback)."""
    lib = _lib.load()
    if not torch.cuda.is_available() or lib.vt_device_count() <= 0:
        raise _lib.VittlesHipError(
            "vittles_amd needs an AMD GPU (gfx950): no HIP device is visible and there is no CPU fallback.")
    return lib


def is_torch(x):
    return isinstance(x, torch.Tensor)


def to_device(x, device=None, dtype=torch.float64):
    """numpy / sequence / torch (any device) -> contiguous-or-strided CUDA tensor."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(device)


def like_input(result, template):
    """Return ``result`` (a CUDA tensor) in the container type of ``template``."""
    if isinstance(template, torch.Tensor):
        return result
    return result.cpu().numpy()


def ld(t):
    """Leading dimension (row stride in elements) of a row-major 2-D tensor for the C-ABI.  torch
    reports arbitrary strides (0, 1) for tensors with fewer than two rows; any value >= the row
    length is right for them."""
    return t.stride(0) if t.shape[0] > 1 else max(int(t.shape[1]), 1)


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def empty(shape, device, dtype=torch.float64):
    return torch.empty(shape, dtype=dtype, device=device)
