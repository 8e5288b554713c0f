"""Seeded synthetic DEM generators (generator G(seed)).

Fractal value noise built from a 24-bit integer lattice hash: deterministic, no libm, f32 arithmetic
in a fixed order (no FMA), so this numpy version and the HIP generator in ``csrc/synth.hip``
(``rdgpu_synth_dem_f32_dev``) produce the same bits.  These are test/bench INPUTS, nothing else.

    z(x,y) = 1000 * sum_{o=0..8} 2^-o * vnoise(x*2^o/512, y*2^o/512, seed+o)
"""
from __future__ import annotations

import numpy as np

OCTAVES = 9
BASE_PERIOD = 512


def _hash24(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    """32-bit integer mix of (ix, iy, seed) -> 24-bit lattice value (uint32 arithmetic, wraps)."""
    with np.errstate(over="ignore"):
        h = ix.astype(np.uint32) * np.uint32(0x9E3779B1)
        h = h ^ (iy.astype(np.uint32) * np.uint32(0x85EBCA77))
        h = h ^ (np.uint32(seed & 0xFFFFFFFF) * np.uint32(0xC2B2AE3D))
        h = h ^ (h >> np.uint32(15))
        h = h * np.uint32(0x2C1B3C6D)
        h = h ^ (h >> np.uint32(12))
        h = h * np.uint32(0x297A2D39)
        h = h ^ (h >> np.uint32(15))
    return h >> np.uint32(8)


def fractal_dem(width: int, height: int, seed: int, x0: int = 0, y0: int = 0, tilt: float = 0.0) -> np.ndarray:
    """float32 [height, width] fractal value-noise DEM, G(seed).  ``tilt`` adds tilt*(x+y) (the
    'tilted plane + noise' best-case variant).  (x0, y0) offsets the window so row-block shards of one
    big DEM can be generated independently."""
    f32 = np.float32
    xs = (np.arange(width, dtype=np.int64) + x0)
    ys = (np.arange(height, dtype=np.int64) + y0)
    z = np.zeros((height, width), f32)
    inv24 = f32(1.0 / 16777216.0)
    for o in range(OCTAVES):
        period = BASE_PERIOD >> o  # 512 .. 2 cells
        ix, fx = np.divmod(xs, period)
        iy, fy = np.divmod(ys, period)
        tx = (fx.astype(f32) / f32(period)).astype(f32)
        ty = (fy.astype(f32) / f32(period)).astype(f32)
        sx = (tx * tx * (f32(3.0) - f32(2.0) * tx)).astype(f32)[None, :]
        sy = (ty * ty * (f32(3.0) - f32(2.0) * ty)).astype(f32)[:, None]
        IX, IY = ix[None, :], iy[:, None]
        v00 = _hash24(IX, IY, seed + o).astype(f32) * inv24
        v10 = _hash24(IX + 1, IY, seed + o).astype(f32) * inv24
        v01 = _hash24(IX, IY + 1, seed + o).astype(f32) * inv24
        v11 = _hash24(IX + 1, IY + 1, seed + o).astype(f32) * inv24
        a = (v00 + sx * (v10 - v00)).astype(f32)
        b = (v01 + sx * (v11 - v01)).astype(f32)
        v = (a + sy * (b - a)).astype(f32)
        z = (z + v * f32(1.0 / (1 << o))).astype(f32)
    z = (z * f32(1000.0)).astype(f32)
    if tilt:
        z = (z + f32(tilt) * (xs[None, :] + ys[:, None]).astype(f32)).astype(f32)
    return z


def fractal_dem_int(width: int, height: int, seed: int, scale: float = 1.0, dtype=np.int32) -> np.ndarray:
    """G_int: floor(z*scale) as an integer DEM -- large flats, stresses flat resolution."""
    z = fractal_dem(width, height, seed)
    return np.floor(z * np.float32(scale)).astype(dtype)
