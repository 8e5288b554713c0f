"""Position-dependent 64-bit band digests of a raster, identical for numpy arrays (the reference's outputs,
computed once in the build container) and torch tensors (the engine's outputs, in HBM).  TEST INFRASTRUCTURE.

digest(band) = sum over the band's cells of mix(bits(value), linear cell index)  (mod 2^64), with

    h = bits * K1 + index * K2;  h ^= h >> 32 (arithmetic);  h *= K3

in wrapping int64 arithmetic -- numpy and torch agree on every one of these operations.  A transposition,
a shifted row or a single differing cell changes the band's digest; the band tells where to look.
"""
from __future__ import annotations

import numpy as np

BAND_ROWS = 1000
_K1 = np.int64(np.uint64(0x9E3779B97F4A7C15).astype(np.int64))
_K2 = np.int64(np.uint64(0xC2B2AE3D27D4EB4F).astype(np.int64))
_K3 = np.int64(np.uint64(0xD6E8FEB86659FD93).astype(np.int64))


def _bits_np(a: np.ndarray) -> np.ndarray:
    if a.dtype == np.float32:
        return a.view(np.int32).astype(np.int64)
    if a.dtype == np.float64:
        return a.view(np.int64)
    return a.astype(np.int64)


def band_digests_np(a: np.ndarray, band_rows: int = BAND_ROWS) -> np.ndarray:
    """uint64 digest of every band of ``band_rows`` rows of the 2-D array ``a``."""
    h, w = a.shape
    out = []
    with np.errstate(over="ignore"):
        for y0 in range(0, h, band_rows):
            band = np.ascontiguousarray(a[y0:y0 + band_rows])
            v = _bits_np(band).ravel()
            idx = np.arange(y0 * w, y0 * w + v.size, dtype=np.int64)
            x = v * _K1 + idx * _K2
            x ^= x >> np.int64(32)
            x *= _K3
            out.append(np.uint64(x.sum(dtype=np.int64).astype(np.uint64)))
    return np.array(out, dtype=np.uint64)


def band_digests_torch(t, band_rows: int = BAND_ROWS) -> np.ndarray:
    """the same digests of a 2-D torch tensor (any device), band by band (0.3 GB of temporaries per band at 40000 columns)."""
    import torch

    h, w = t.shape
    k1, k2, k3 = int(_K1), int(_K2), int(_K3)
    out = []
    for y0 in range(0, h, band_rows):
        band = t[y0:y0 + band_rows].contiguous()
        if band.dtype == torch.float32:
            v = band.view(torch.int32).to(torch.int64)
        elif band.dtype == torch.float64:
            v = band.view(torch.int64)
        else:
            v = band.to(torch.int64)
        v = v.reshape(-1)
        idx = torch.arange(y0 * w, y0 * w + v.numel(), dtype=torch.int64, device=t.device)
        x = v * k1 + idx * k2
        x = x ^ (x >> 32)
        x = x * k3
        out.append(x.sum(dtype=torch.int64))
    d = torch.stack(out).cpu().numpy()
    return d.view(np.uint64)
