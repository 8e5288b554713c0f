"""Host-side mirror of the reference operator interface for the hot path, over the C-ABI.

Host arrays (numpy) go through the drop-in entry points ``rdgpu_<op>_<dtype>`` (H2D, compute, D2H into
the same buffer); HBM-resident torch tensors go through ``rdgpu_<op>_dev_<dtype>``.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import RdgpuError, check, lib

_SUFFIX = {
    np.dtype(np.uint8): "u8",
    np.dtype(np.int16): "i16",
    np.dtype(np.uint16): "u16",
    np.dtype(np.int32): "i32",
    np.dtype(np.uint32): "u32",
    np.dtype(np.float32): "f32",
}
_TOPO = {"D8": 8, "D4": 4, 8: 8, 4: 4}


def _suffix(dtype) -> str:
    try:
        return _SUFFIX[np.dtype(dtype)]
    except KeyError:
        raise RdgpuError(f"unsupported elevation dtype {dtype} (supported: {sorted(str(k) for k in _SUFFIX)})") from None


def _topo(topology) -> int:
    try:
        return _TOPO[topology]
    except KeyError:
        raise RdgpuError("Unknown topology!") from None  # depressions.hpp:19-20


def FillDepressions(dem: np.ndarray, epsilon: bool = False, in_place: bool = False, topology="D8"):
    """Fill all depressions of ``dem`` (reference: ``rd.FillDepressions``,
    wrappers/pyrichdem/richdem/__init__.py:381-422 -> FillDepressions<topo>, depressions.hpp:13-21).
    Returns the filled array (or None when ``in_place``)."""
    if epsilon:
        raise RdgpuError("FillDepressions(epsilon=True) is not part of this round's hot path")
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("FillDepressions: expected a 2-D numpy array")
    out = dem if in_place else dem.copy()
    if not out.flags["C_CONTIGUOUS"]:
        if in_place:
            raise RdgpuError("FillDepressions(in_place=True) needs a C-contiguous array")
        out = np.ascontiguousarray(out)
    h, w = out.shape
    fn = getattr(lib(), f"rdgpu_fill_{_suffix(out.dtype)}")
    check(fn(out.ctypes.data_as(ctypes.c_void_p), w, h, _topo(topology)), "rdgpu_fill")
    return None if in_place else out


# ---- HBM-resident variants (torch tensors on the GPU) ---------------------------------------
def _torch_suffix(t) -> str:
    import torch

    m = {torch.uint8: "u8", torch.int16: "i16", torch.int32: "i32", torch.float32: "f32"}
    if t.dtype not in m:
        raise RdgpuError(f"unsupported tensor dtype {t.dtype}")
    return m[t.dtype]


def _stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def fill_depressions_dev(dem, topology="D8") -> None:
    """In-place fill of a 2-D contiguous CUDA(HIP) tensor, on torch's current stream."""
    if not (dem.is_cuda and dem.dim() == 2 and dem.is_contiguous()):
        raise RdgpuError("fill_depressions_dev: expected a contiguous 2-D tensor on the GPU")
    h, w = dem.shape
    fn = getattr(lib(), f"rdgpu_fill_dev_{_torch_suffix(dem)}")
    check(fn(ctypes.c_void_p(dem.data_ptr()), w, h, _topo(topology), _stream_ptr()), "rdgpu_fill_dev")


def synth_dem_dev(out, seed: int, x0: int = 0, y0: int = 0, tilt: float = 0.0) -> None:
    """Fill a float32 CUDA tensor [h, w] with the seeded fractal DEM G(seed) (bench/test input)."""
    import torch

    if not (out.is_cuda and out.dim() == 2 and out.is_contiguous() and out.dtype == torch.float32):
        raise RdgpuError("synth_dem_dev: expected a contiguous 2-D float32 tensor on the GPU")
    h, w = out.shape
    check(
        lib().rdgpu_synth_dem_dev_f32(
            ctypes.c_void_p(out.data_ptr()), w, h, int(seed), int(x0), int(y0), ctypes.c_float(tilt), _stream_ptr()
        ),
        "rdgpu_synth_dem_dev_f32",
    )
