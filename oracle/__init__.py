"""oracle -- CPU checker for the hot path.  TEST INFRASTRUCTURE ONLY.

Two back ends with one numpy API:

* ``oracle.port``  -> ``oracle/liboracle.so`` (``oracle.c``: the plain-C restatement,
  every function cites the reference file:line it follows);
* ``oracle.ref``   -> ``oracle/_ref/libref.so`` (``ref_wrap.cpp``: the UNMODIFIED reference
  headers compiled in place from ``/root/reference/include``; present wherever it was
  built -- it travels to the GPU box as a prebuilt file).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  ``richdem_amd`` (the product) never imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_SUFFIX = {
    np.dtype(np.int8): "i8",
    np.dtype(np.uint8): "u8",
    np.dtype(np.int16): "i16",
    np.dtype(np.uint16): "u16",
    np.dtype(np.int32): "i32",
    np.dtype(np.uint32): "u32",
    np.dtype(np.float32): "f32",
    np.dtype(np.float64): "f64",
    np.dtype(np.int64): "i64",
    np.dtype(np.uint64): "u64",
}
_CT = {
    "i8": ctypes.c_int8,
    "u8": ctypes.c_uint8,
    "i16": ctypes.c_int16,
    "u16": ctypes.c_uint16,
    "i32": ctypes.c_int32,
    "u32": ctypes.c_uint32,
    "f32": ctypes.c_float,
    "f64": ctypes.c_double,
    "i64": ctypes.c_int64,
    "u64": ctypes.c_uint64,
}
#: variants of the reference fill (oracle/ref_wrap.cpp ref_fill)
ZHOU2016, BARNES2014_D8, BARNES2014_D4, WEI2018, ORIGINAL_D8 = 0, 1, 2, 3, 4


def build(force: bool = False) -> None:
    """Compile liboracle.so (always) and _ref/libref.so (only where /root/reference exists)."""
    if force or not os.path.exists(os.path.join(_HERE, "liboracle.so")) or (
        os.path.getmtime(os.path.join(_HERE, "liboracle.so"))
        < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("oracle.c", "oracle_impl.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref.so")
    if os.path.isdir("/root/reference/include/richdem") and (
        force or not os.path.exists(ref_so) or os.path.getmtime(ref_so) < os.path.getmtime(os.path.join(_HERE, "ref_wrap.cpp"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _suf(a: np.ndarray) -> str:
    try:
        return _SUFFIX[a.dtype]
    except KeyError:
        raise TypeError(f"oracle: unsupported elevation dtype {a.dtype}") from None


class _Backend:
    """numpy front end shared by the port and the compiled reference."""

    def __init__(self, path: str, prefix: str):
        self.path = path
        self.prefix = prefix
        self._lib = None

    @property
    def available(self) -> bool:
        return os.path.exists(self.path)

    @property
    def lib(self):
        if self._lib is None:
            if not self.available:
                raise FileNotFoundError(f"{self.path} is not built (run oracle.build())")
            self._lib = ctypes.CDLL(self.path)
        return self._lib

    def _fn(self, name):
        f = getattr(self.lib, f"{self.prefix}_{name}")
        f.restype = None
        return f

    # ---- fill -------------------------------------------------------------------------------
    def fill(self, dem: np.ndarray, topo: int = 8, variant: int | None = None) -> np.ndarray:
        """Filled copy of ``dem`` (FillDepressions<D8/D4>, depressions.hpp:13-21)."""
        out = np.ascontiguousarray(dem).copy()
        h, w = out.shape
        s = _suf(out)
        if self.prefix == "ref":
            if variant is None:
                variant = ZHOU2016 if topo == 8 else BARNES2014_D4
            self._fn(f"fill_{s}")(_ptr(out), w, h, int(variant))
        else:
            self._fn(f"fill_{s}")(_ptr(out), w, h, int(topo))
        return out

    def fill_wei2018(self, dem: np.ndarray, nodata) -> np.ndarray:
        """PriorityFlood_Wei2018 (depressions/Wei2018.hpp:154-202) with the raster's NoData value set."""
        out = np.ascontiguousarray(dem).copy()
        h, w = out.shape
        s = _suf(out)
        self._fn(f"fill_wei2018_{s}")(_ptr(out), _CT[s](nodata), w, h)
        return out

    def fill_original(self, dem: np.ndarray, topo: int = 8) -> np.ndarray:
        """PriorityFlood_Original<topo> (depressions/Barnes2014.hpp:136-198); the restatement's fill returns the same surface."""
        out = np.ascontiguousarray(dem).copy()
        h, w = out.shape
        s = _suf(out)
        self._fn(f"fill_original_{s}" if self.prefix == "ref" else f"fill_{s}")(_ptr(out), w, h, int(topo))
        return out

    def has_depressions(self, dem: np.ndarray, topo: int = 8) -> bool:
        """HasDepressions<topo> (depressions/Barnes2014.hpp:44-103)."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        f = getattr(self.lib, f"{self.prefix}_has_depressions_{_suf(dem)}")
        f.restype = ctypes.c_int
        return bool(f(_ptr(dem), w, h, int(topo)))

    # ---- directions -------------------------------------------------------------------------
    def d8_flowdirs(self, dem: np.ndarray, nodata) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w), np.uint8)
        self._fn(f"d8_flowdirs_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(out))
        return out

    def resolve_flats(self, dem: np.ndarray, nodata):
        """(dirs_before_flats, flat_mask, labels) of resolve_flats_barnes (flat_resolution.hpp:447)."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        dirs = np.empty((h, w), np.uint8)
        mask = np.empty((h, w), np.int32)
        labels = np.empty((h, w), np.int32)
        if self.prefix == "ref":
            self._fn(f"resolve_flats_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(dirs), _ptr(mask), _ptr(labels))
        else:
            self._fn(f"d8_flowdirs_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(dirs))
            self._fn(f"resolve_flats_{s}")(_ptr(dem), w, h, _ptr(dirs), _ptr(mask), _ptr(labels))
        return dirs, mask, labels

    def flat_resolution(self, dem: np.ndarray, nodata) -> np.ndarray:
        """barnes_flat_resolution_d8(dem, dirs, alter=false) (flat_resolution.hpp:587-605)."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w), np.uint8)
        if self.prefix == "ref":
            tmp = dem.copy()
            self._fn(f"flat_resolution_{s}")(_ptr(tmp), _CT[s](nodata), w, h, _ptr(out), 0)
        else:
            self._fn(f"flat_resolution_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(out))
        return out

    def flat_resolution_alter(self, dem: np.ndarray, nodata):
        """(altered dem, dirs) of barnes_flat_resolution_d8(dem, dirs, alter=true)."""
        out_dem = np.ascontiguousarray(dem).copy()
        h, w = out_dem.shape
        s = _suf(out_dem)
        dirs = np.empty((h, w), np.uint8)
        if self.prefix == "ref":
            self._fn(f"flat_resolution_{s}")(_ptr(out_dem), _CT[s](nodata), w, h, _ptr(dirs), 1)
        else:
            self._fn(f"flat_resolution_alter_{s}")(_ptr(out_dem), _CT[s](nodata), w, h, _ptr(dirs))
        return out_dem, dirs

    # ---- accumulation -----------------------------------------------------------------------
    def d8_flow_accum(self, dirs: np.ndarray, nodata: int = 255, dtype=np.float64) -> np.ndarray:
        dirs = np.ascontiguousarray(dirs, dtype=np.uint8)
        h, w = dirs.shape
        s = {np.dtype(np.int32): "i32", np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]
        out = np.empty((h, w), dtype)
        self._fn(f"d8_flow_accum_{s}")(_ptr(dirs), ctypes.c_uint8(nodata), w, h, _ptr(out))
        return out

    def fm_d8(self, dem: np.ndarray, nodata) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w, 9), np.float32)
        self._fn(f"fm_d8_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(out))
        return out

    MFD = {"Holmgren": 0, "Freeman": 1, "Quinn": 2, "D4": 3}

    def fm_mfd(self, dem: np.ndarray, nodata, method: str, xparam: float = 1.0) -> np.ndarray:
        """FM_Holmgren / FM_Freeman / FM_Quinn / FM_D4 proportions [h, w, 9]."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w, 9), np.float32)
        self._fn(f"fm_mfd_{s}")(_ptr(dem), _CT[s](nodata), w, h, self.MFD[method], ctypes.c_double(xparam), _ptr(out))
        return out

    def fa_mfd(self, dem: np.ndarray, nodata, method: str, xparam: float = 1.0, weights: np.ndarray | None = None) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        acc = np.ones((h, w), np.float64) if weights is None else np.ascontiguousarray(weights, dtype=np.float64).copy()
        self._fn(f"fa_mfd_{s}")(_ptr(dem), _CT[s](nodata), w, h, self.MFD[method], ctypes.c_double(xparam), _ptr(acc))
        return acc

    def resolve_flats_epsilon(self, dem: np.ndarray, nodata) -> np.ndarray:
        """ResolveFlatsEpsilon (flats/flats.hpp:21-28): returns the altered DEM."""
        dem = np.ascontiguousarray(dem).copy()
        h, w = dem.shape
        s = _suf(dem)
        self._fn(f"resolve_flats_epsilon_{s}")(_ptr(dem), _CT[s](nodata), w, h)
        return dem

    def pit_mask(self, dem: np.ndarray, nodata, topo: int = 8) -> np.ndarray:
        """pit_mask<topo> (depressions/Barnes2014.hpp:593-676): 1 in a depression, 0 not, 3 NoData."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w), np.uint8)
        self._fn(f"pit_mask_{s}")(_ptr(dem), _CT[s](nodata), w, h, int(topo), _ptr(out))
        return out

    # ---- the other outputs of the Priority-Flood sweep ---------------------------
    def fill_epsilon(self, dem: np.ndarray, nodata, topo: int = 8) -> np.ndarray:
        """PriorityFloodEpsilon_Barnes2014<topo> (depressions/Barnes2014.hpp:335-420): float32 / float64 only."""
        out = np.ascontiguousarray(dem).copy()
        h, w = out.shape
        s = _suf(out)
        if s not in ("f32", "f64"):
            raise TypeError("Priority-Flood+Epsilon is only available for floating-point data types!")
        self._fn(f"fill_epsilon_{s}")(_ptr(out), _CT[s](nodata), w, h, int(topo))
        return out

    def watersheds(self, dem: np.ndarray, nodata, topo: int = 8, alter: bool = False):
        """PriorityFloodWatersheds_Barnes2014<topo> (:713-807): (labels int32, dem -- filled when alter)."""
        out = np.ascontiguousarray(dem).copy()
        h, w = out.shape
        s = _suf(out)
        labels = np.empty((h, w), np.int32)
        self._fn(f"watersheds_{s}")(_ptr(out), _CT[s](nodata), w, h, int(topo), int(bool(alter)), _ptr(labels))
        return labels, out

    def fill_max_dep(self, dem: np.ndarray, max_dep_size: int, topo: int = 8) -> np.ndarray:
        """PriorityFlood_Barnes2014_max_dep<topo> (:844-931): only depressions of <= max_dep_size cells are filled."""
        out = np.ascontiguousarray(dem).copy()
        h, w = out.shape
        s = _suf(out)
        self._fn(f"fill_max_dep_{s}")(_ptr(out), w, h, int(topo), ctypes.c_uint64(int(max_dep_size)))
        return out

    def pf_flowdirs(self, dem: np.ndarray, nodata) -> np.ndarray:
        """PriorityFloodFlowdirs_Barnes2014 (depressions/Barnes2014.hpp:483-555): uint8 D8 directions, every cell flows
        to the cell that closed it; NoData cells 0."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w), np.uint8)
        self._fn(f"pf_flowdirs_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(out))
        return out

    def dinf_flowdirs(self, dem: np.ndarray, nodata) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w), np.float32)
        self._fn(f"dinf_flowdirs_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(out))
        return out

    def fm_tarboton(self, dem: np.ndarray, nodata) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        out = np.empty((h, w, 9), np.float32)
        self._fn(f"fm_tarboton_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(out))
        return out

    def fa_tarboton(self, dem: np.ndarray, nodata, weights: np.ndarray | None = None) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        acc = np.ones((h, w), np.float64) if weights is None else np.ascontiguousarray(weights, dtype=np.float64).copy()
        self._fn(f"fa_tarboton_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(acc))
        return acc

    def flow_accumulation(self, props9: np.ndarray, weights: np.ndarray | None = None) -> np.ndarray:
        props9 = np.ascontiguousarray(props9, dtype=np.float32)
        h, w, _ = props9.shape
        acc = np.ones((h, w), np.float64) if weights is None else np.ascontiguousarray(weights, dtype=np.float64).copy()
        self._fn("flow_accumulation_f64")(_ptr(props9), w, h, _ptr(acc))
        return acc

    def fa_d8(self, dem: np.ndarray, nodata, weights: np.ndarray | None = None) -> np.ndarray:
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        acc = np.ones((h, w), np.float64) if weights is None else np.ascontiguousarray(weights, dtype=np.float64).copy()
        self._fn(f"fa_d8_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(acc))
        return acc


    def fa_d8_lean(self, dem: np.ndarray, nodata, weights: np.ndarray | None = None) -> np.ndarray:
        """FA_D8 through one receiver byte per cell (oracle port only; see oracle_impl.h orc_fa_d8_lean)."""
        dem = np.ascontiguousarray(dem)
        h, w = dem.shape
        s = _suf(dem)
        acc = np.ones((h, w), np.float64) if weights is None else np.ascontiguousarray(weights, dtype=np.float64).copy()
        self._fn(f"fa_d8_lean_{s}")(_ptr(dem), _CT[s](nodata), w, h, _ptr(acc))
        return acc


port = _Backend(os.path.join(_HERE, "liboracle.so"), "orc")
ref = _Backend(os.path.join(_HERE, "_ref", "libref.so"), "ref")


def read_ascii_grid(path: str, dtype=np.float64):
    """ArcInfo-ASCII reader for the reference's text fixtures.
    Returns (array[h, w] cast to dtype, nodata as float)."""
    with open(path) as f:
        toks = f.read().split()
    hdr = {}
    i = 0
    while i < len(toks) and toks[i][0].isalpha():
        hdr[toks[i].lower()] = toks[i + 1]
        i += 2
    w, h = int(hdr["ncols"]), int(hdr["nrows"])
    nodata = float(hdr.get("nodata_value", -9999))
    vals = np.array([float(t) for t in toks[i : i + w * h]], dtype=np.float64)
    return vals.reshape(h, w).astype(dtype), nodata
