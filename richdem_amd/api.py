"""Host-side mirror of the reference operator interface for the hot path, over the C-ABI.

Host arrays (numpy) go through the drop-in entry points ``rdgpu_<op>_<dtype>`` (H2D, compute, D2H into
the same buffer); HBM-resident torch tensors go through ``rdgpu_<op>_dev_<dtype>``.
"""
from __future__ import annotations

import ctypes
import os
import warnings

import numpy as np

from ._lib import RdgpuError, check, lib

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
_TOPO = {"D8": 8, "D4": 4, 8: 8, 4: 4}
_CT = {"i8": ctypes.c_int8, "u8": ctypes.c_uint8, "i16": ctypes.c_int16, "u16": ctypes.c_uint16, "i32": ctypes.c_int32,
       "u32": ctypes.c_uint32, "f32": ctypes.c_float, "f64": ctypes.c_double, "i64": ctypes.c_int64, "u64": ctypes.c_uint64}
_NP = {"i8": np.int8, "u8": np.uint8, "i16": np.int16, "u16": np.uint16, "i32": np.int32, "u32": np.uint32, "f32": np.float32,
       "f64": np.float64, "i64": np.int64, "u64": np.uint64}


def _scalar(s: str, nodata):
    """ctypes scalar of the DEM's element type from the user's no_data value, converted the way the reference's
    setNoData(double) does (a C cast to T: fractions truncate towards zero, out-of-range integers wrap as they do
    on x86-64): rdarray(int16_arr, no_data=-9999.0) works, and so does the common uint8 DEM with no_data=-9999.
    NaN / infinite values for an integer DEM are an error."""
    if s in ("f32", "f64"):
        return _CT[s](float(nodata))
    try:
        v = int(float(nodata))
    except (TypeError, ValueError, OverflowError):
        raise RdgpuError(f"no_data value {nodata!r} is not representable in the DEM's element type {_NP[s].__name__}") from None
    bits = 8 * np.dtype(_NP[s]).itemsize
    v &= (1 << bits) - 1
    if np.issubdtype(_NP[s], np.signedinteger) and v >= 1 << (bits - 1):
        v -= 1 << bits
    return _CT[s](v)


_ELEV_SUFFIX = dict(_SUFFIX)   # d8 directions, flat resolution, ResolveFlatsEpsilon, FA_D8 / FM_D8: every element type
_MFD_SUFFIX = {k: v for k, v in _SUFFIX.items() if v not in ("i64", "u64")}   # D-infinity and the MFD families
_ACC_SUFFIX = {np.dtype(np.int32): "i32", np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}


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


def FillDepressions(dem: np.ndarray, epsilon: bool = False, in_place: bool = False, topology="D8", shards: int = 1,
                    nodata=-9999):
    """Fill all depressions of ``dem`` (reference: ``rd.FillDepressions``,
    wrappers/pyrichdem/richdem/__init__.py:381-422 -> FillDepressions<topo>, depressions.hpp:13-21; ``epsilon=True``
    -> PriorityFloodEpsilon_Barnes2014<topo>, depressions/Barnes2014.hpp:335-420, floating point only, ``nodata`` cells
    are left alone).  Returns the filled array (or None when ``in_place``).

    Ties (``epsilon=True`` only; the plain fill is exact for every input): the reference's result depends on the order
    in which ``std::priority_queue`` pops cells of EQUAL elevation -- a plateau or lake entered through several cells
    of one level gets its gradient from whichever pops first.  This engine returns the order-free surface
    ``E(c) = max(z(c), nextafter(min over neighbours E))``: identical to the reference when no two gradient sources
    share an elevation, a cell-wise lower bound of it otherwise (quantised / integer-valued DEMs).  The number of such
    sources is counted on the device (``epsilon_stats()["tie_sources"]``) and a ``RuntimeWarning`` is raised when it is
    not zero."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("FillDepressions: expected a 2-D numpy array")
    out = dem if in_place else dem.copy()
    if not out.flags["C_CONTIGUOUS"]:
        if in_place:
            raise RdgpuError("FillDepressions(in_place=True) needs a C-contiguous array")
        out = np.ascontiguousarray(out)
    h, w = out.shape
    if epsilon:
        if shards > 1:
            raise RdgpuError("FillDepressions(epsilon=True, shards=...): the epsilon fill is not sharded; use shards=1")
        if out.dtype not in (np.float32, np.float64):
            raise RdgpuError("Priority-Flood+Epsilon is only available for floating-point data types!")   # Barnes2014.hpp:424-451
        s = _suffix(out.dtype)
        check(getattr(lib(), f"rdgpu_fill_epsilon_{s}")(out.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h,
                                                        _topo(topology)), "rdgpu_fill_epsilon")
        ties = epsilon_stats()["tie_sources"]
        if ties:
            warnings.warn(f"FillDepressions(epsilon=True): {ties} gradient sources share their elevation with another one; "
                          "the reference's Priority-Flood+Epsilon resolves such ties by std::priority_queue's pop order, "
                          "this engine returns the order-free surface (a cell-wise lower bound of the reference's)",
                          RuntimeWarning, stacklevel=2)
        return None if in_place else out
    if shards > 1:   # the multi-GPU row-block protocol, shard after shard on one GPU
        if _suffix(out.dtype) in ("f64", "i64", "u64"):
            raise RdgpuError("FillDepressions(shards=...): the row-block shard engine takes the 32-bit element types")
        fn = getattr(lib(), f"rdgpu_fill_sharded_{_suffix(out.dtype)}")
        check(fn(out.ctypes.data_as(ctypes.c_void_p), w, h, _topo(topology), int(shards)), "rdgpu_fill_sharded")
    else:
        fn = getattr(lib(), f"rdgpu_fill_{_suffix(out.dtype)}")
        check(fn(out.ctypes.data_as(ctypes.c_void_p), w, h, _topo(topology)), "rdgpu_fill")
    return None if in_place else out


def has_depressions(dem: np.ndarray, topology="D8") -> bool:
    """HasDepressions<topo> (depressions/Barnes2014.hpp:44-103; apps/rd_depressions_has.cpp): True when the fill would
    raise at least one cell of ``dem``."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("has_depressions: expected a 2-D numpy array")
    dem = np.ascontiguousarray(dem)
    h, w = dem.shape
    if w == 0 or h == 0:
        return False
    out = ctypes.c_int(0)
    check(getattr(lib(), f"rdgpu_has_depressions_{_suffix(dem.dtype)}")(dem.ctypes.data_as(ctypes.c_void_p), w, h, _topo(topology),
                                                                          ctypes.byref(out)), "rdgpu_has_depressions")
    return bool(out.value)


def fill_wei2018(dem: np.ndarray, nodata=-9999, in_place: bool = False):
    """PriorityFlood_Wei2018 (depressions/Wei2018.hpp:154-202): the D8 fill in which NoData cells stay as they are and the
    data cells next to them drain like the raster's edge cells (InitPriorityQue, :14-50)."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("fill_wei2018: expected a 2-D numpy array")
    out = dem if in_place else dem.copy()
    if not out.flags["C_CONTIGUOUS"]:
        if in_place:
            raise RdgpuError("fill_wei2018(in_place=True) needs a C-contiguous array")
        out = np.ascontiguousarray(out)
    s = _suffix(out.dtype)
    h, w = out.shape
    if w and h:
        check(getattr(lib(), f"rdgpu_fill_wei2018_{s}")(out.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h),
              "rdgpu_fill_wei2018")
    return None if in_place else out


def fill_max_dep(dem: np.ndarray, max_dep_size: int, topology="D8", in_place: bool = False):
    """PriorityFlood_Barnes2014_max_dep<topo> (depressions/Barnes2014.hpp:844-931; rd_depressions_flood's third
    argument): only depressions of at most ``max_dep_size`` cells are filled."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("fill_max_dep: expected a 2-D numpy array")
    out = dem if in_place else dem.copy()
    if not out.flags["C_CONTIGUOUS"]:
        if in_place:
            raise RdgpuError("fill_max_dep(in_place=True) needs a C-contiguous array")
        out = np.ascontiguousarray(out)
    s = _suffix(out.dtype)
    if int(max_dep_size) < 0:
        raise RdgpuError("fill_max_dep: max_dep_size must not be negative")
    h, w = out.shape
    check(getattr(lib(), f"rdgpu_fill_max_dep_{s}")(out.ctypes.data_as(ctypes.c_void_p), w, h, _topo(topology),
                                                    ctypes.c_uint64(int(max_dep_size))), "rdgpu_fill_max_dep")
    st = max_dep_stats()   # (64-bit element types run on dense value ranks through the same engine: equal values, equal ranks)
    if st["tie_cluster_cells"]:   # (cells whose fate the order can actually decide; pockets with two candidates alone are common)
        warnings.warn(f"fill_max_dep: {st['tie_pockets']} of {st['pockets']} pockets can be flooded by two or more cells of their "
                      f"spill elevation; in their clusters ({st['tie_cluster_cells']} of {st['pocket_cells']} pocket cells) the "
                      "reference's grouping follows its heap's pop order, this engine's the lowest cell index", RuntimeWarning)
    return None if in_place else out


def watersheds(dem: np.ndarray, nodata=-9999, topology="D8", alter: bool = False):
    """PriorityFloodWatersheds_Barnes2014<topo> (depressions/Barnes2014.hpp:713-807): int32 watershed labels (1.. in the
    order the watersheds' first cells are flooded, -1 for NoData connected to the border); with ``alter`` also the filled
    DEM: returns labels, or (labels, filled)."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("watersheds: expected a 2-D numpy array")
    work = np.ascontiguousarray(dem).copy()
    s = _suffix(work.dtype)
    h, w = work.shape
    labels = np.empty((h, w), np.int32)
    check(getattr(lib(), f"rdgpu_watersheds_{s}")(work.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h, _topo(topology),
                                                  1 if alter else 0, labels.ctypes.data_as(ctypes.c_void_p)), "rdgpu_watersheds")
    return (labels, work) if alter else labels


def pf_flowdirs_dev(dem, nodata, dirs) -> None:
    """PriorityFloodFlowdirs_Barnes2014 of a contiguous 2-D CUDA tensor (8 / 16 / 32-bit element type) into a uint8 CUDA
    tensor of the same shape, on torch's current stream (the call synchronises: one read-back per nesting level)."""
    import torch

    h, w = _dev2d(dem, "pf_flowdirs_dev")
    if dirs.dtype != torch.uint8 or tuple(dirs.shape) != (h, w) or not dirs.is_contiguous() or not dirs.is_cuda:
        raise RdgpuError("pf_flowdirs_dev: dirs must be a contiguous uint8 CUDA tensor of the DEM's shape")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_pf_flowdirs_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h,
                                                       ctypes.c_void_p(dirs.data_ptr()), _stream_ptr()), "rdgpu_pf_flowdirs_dev")


class _PfdStats(ctypes.Structure):
    _fields_ = [("levels", ctypes.c_uint32), ("twins", ctypes.c_uint32), ("unresolved", ctypes.c_uint64),
                ("tie_passes", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


def pf_flowdirs_stats() -> dict:
    st = _PfdStats()
    check(lib().rdgpu_pf_flowdirs_get_stats(ctypes.byref(st)), "rdgpu_pf_flowdirs_get_stats")
    return {"levels": st.levels, "twins": st.twins, "unresolved": st.unresolved, "tie_passes": st.tie_passes}


def pf_flowdirs(dem: np.ndarray, nodata=-9999) -> np.ndarray:
    """PriorityFloodFlowdirs_Barnes2014 (depressions/Barnes2014.hpp:483-555): uint8 D8 directions in which every cell
    points at the neighbour the (non-raising) flood reached first; NoData cells 0.  Equal to the reference, equal
    elevations included: the reference's stable queue pops equal elevations in order of insertion, and that order is found
    as a fixed point -- the flood runs on the raster's unique ranks of (elevation, discovery time) until the ranks
    reproduce themselves (pf_flowdirs_stats(): "twins" cells with an equal elsewhere, "tie_passes" extra floods; a
    RuntimeWarning only if the passes ran out, "unresolved" != 0)."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("pf_flowdirs: expected a 2-D numpy array")
    dem = np.ascontiguousarray(dem)
    s = _suffix(dem.dtype)
    h, w = dem.shape
    out = np.empty((h, w), np.uint8)
    check(getattr(lib(), f"rdgpu_pf_flowdirs_{s}")(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h,
                                                   out.ctypes.data_as(ctypes.c_void_p)), "rdgpu_pf_flowdirs")
    st = pf_flowdirs_stats()
    if st["unresolved"]:
        import warnings

        if os.environ.get("RDGPU_PFD_RANKS", "1")[:1] == "0":
            why = (f"ties at {st['unresolved']} cells were decided by neighbour number (RDGPU_PFD_RANKS=0, the fast path without the "
                   "reference's insertion order)")
        elif st["tie_passes"] == 0:
            why = "no tie-order pass ran (RDGPU_PFD_TIE_PASSES=0): equal cells were taken in a first-guess order"
        else:
            why = (f"the order of {st['unresolved']} of them among their equals was still moving when the tie-order passes were stopped "
                   f"after {st['tie_passes']} (RDGPU_PFD_TIE_PASSES / RDGPU_PFD_TIE_SECONDS)")
        warnings.warn(f"pf_flowdirs: {st['twins']} cells share their elevation with another cell and {why}: the result is the "
                      "reference's only where those ties do not decide", RuntimeWarning)
    return out


def pit_mask(dem: np.ndarray, nodata, topology="D8") -> np.ndarray:
    """uint8 mask of the cells lying in depressions: 1 = the fill would raise the cell, 0 = not, 3 = NoData
    (reference pit_mask<topo>, depressions/Barnes2014.hpp:593-676; apps/rd_depressions_mask.cpp)."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("pit_mask: expected a 2-D numpy array")
    dem = np.ascontiguousarray(dem)
    s = _suffix(dem.dtype)
    h, w = dem.shape
    out = np.empty((h, w), np.uint8)
    check(getattr(lib(), f"rdgpu_pit_mask_{s}")(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h, _topo(topology),
                                                out.ctypes.data_as(ctypes.c_void_p)), "rdgpu_pit_mask")
    return out


def _elev(dem, who, mfd: bool = False):
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError(f"{who}: expected a 2-D numpy array")
    dem = np.ascontiguousarray(dem)
    try:
        return dem, (_MFD_SUFFIX if mfd else _ELEV_SUFFIX)[dem.dtype]
    except KeyError:
        raise RdgpuError(f"{who}: unsupported elevation dtype {dem.dtype}") from None


def d8_flow_directions(dem: np.ndarray, nodata) -> np.ndarray:
    """uint8 D8 directions (reference d8_flow_directions, flowmet/d8_flowdirs.hpp:96-123)."""
    dem, s = _elev(dem, "d8_flow_directions")
    h, w = dem.shape
    out = np.empty((h, w), np.uint8)
    fn = getattr(lib(), f"rdgpu_d8_flowdirs_{s}")
    check(fn(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h, out.ctypes.data_as(ctypes.c_void_p)),
          "rdgpu_d8_flowdirs")
    return out


def barnes_flat_resolution_d8(dem: np.ndarray, nodata, alter: bool = False) -> np.ndarray:
    """Flat-resolved uint8 D8 directions (reference barnes_flat_resolution_d8(elev, flowdirs, alter=false),
    flats/flat_resolution.hpp:587-605)."""
    if alter:
        if not (isinstance(dem, np.ndarray) and dem.ndim == 2 and dem.flags["C_CONTIGUOUS"]):
            raise RdgpuError("barnes_flat_resolution_d8(alter=True): needs a C-contiguous 2-D array (it is altered in place)")
        h, w = dem.shape
        s = _suffix(dem.dtype)   # integer element types step towards zero, as the reference's nextafterf(e, 0) does
        out = np.empty((h, w), np.uint8)
        fn = getattr(lib(), f"rdgpu_flat_resolution_d8_alter_{s}")
        check(fn(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h, out.ctypes.data_as(ctypes.c_void_p)),
              "rdgpu_flat_resolution_d8_alter")
        return out
    dem, s = _elev(dem, "barnes_flat_resolution_d8")
    h, w = dem.shape
    out = np.empty((h, w), np.uint8)
    fn = getattr(lib(), f"rdgpu_flat_resolution_d8_{s}")
    check(fn(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h, out.ctypes.data_as(ctypes.c_void_p)),
          "rdgpu_flat_resolution_d8")
    return out


def resolve_flats(dem: np.ndarray, nodata):
    """(flat-resolved dirs, flat_mask, flat partition labels) -- exposes resolve_flats_barnes's
    intermediate arrays (flats/flat_resolution.hpp:447-517) for parity tests."""
    dem, s = _elev(dem, "resolve_flats")
    h, w = dem.shape
    dirs = np.empty((h, w), np.uint8)
    mask = np.empty((h, w), np.int32)
    labels = np.empty((h, w), np.int32)
    fn = getattr(lib(), f"rdgpu_resolve_flats_{s}")
    check(fn(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h, dirs.ctypes.data_as(ctypes.c_void_p),
             mask.ctypes.data_as(ctypes.c_void_p), labels.ctypes.data_as(ctypes.c_void_p)), "rdgpu_resolve_flats")
    return dirs, mask, labels


def resolve_flats_epsilon(dem: np.ndarray, nodata, in_place: bool = False):
    """ResolveFlatsEpsilon (reference flats/flats.hpp:21-28, what ``rd.ResolveFlats`` calls): the DEM altered so
    that every flat with an outlet drains.  Returns the altered array (None when ``in_place``)."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("ResolveFlats: expected a 2-D numpy array")
    out = dem if in_place else dem.copy()
    if not out.flags["C_CONTIGUOUS"]:
        if in_place:
            raise RdgpuError("ResolveFlats(in_place=True) needs a C-contiguous array")
        out = np.ascontiguousarray(out)
    try:
        s = _ELEV_SUFFIX[out.dtype]
    except KeyError:
        raise RdgpuError(f"ResolveFlats: unsupported elevation dtype {out.dtype}") from None
    h, w = out.shape
    check(getattr(lib(), f"rdgpu_resolve_flats_epsilon_{s}")(out.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h),
          "rdgpu_resolve_flats_epsilon")
    return None if in_place else out


def d8_flow_accum(dirs: np.ndarray, nodata: int = 255, dtype=np.float64) -> np.ndarray:
    """Cells draining through each cell from uint8 D8 directions (reference d8_flow_accum,
    methods/d8_methods.hpp:47-139)."""
    if not isinstance(dirs, np.ndarray) or dirs.ndim != 2 or dirs.dtype != np.uint8:
        raise RdgpuError("d8_flow_accum: expected a 2-D uint8 array of D8 directions")
    dirs = np.ascontiguousarray(dirs)
    h, w = dirs.shape
    try:
        s = _ACC_SUFFIX[np.dtype(dtype)]
    except KeyError:
        raise RdgpuError(f"d8_flow_accum: unsupported accumulation dtype {dtype}") from None
    out = np.empty((h, w), dtype)
    fn = getattr(lib(), f"rdgpu_d8_flow_accum_{s}")
    check(fn(dirs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint8(nodata), w, h, out.ctypes.data_as(ctypes.c_void_p)),
          "rdgpu_d8_flow_accum")
    return out


def dinf_flow_directions(dem: np.ndarray, nodata) -> np.ndarray:
    """float32 D-infinity angles (reference dinf_flow_directions, flowmet/dinf_flowdirs.hpp:128-152)."""
    dem, s = _elev(dem, "dinf_flow_directions", mfd=True)
    h, w = dem.shape
    out = np.empty((h, w), np.float32)
    check(getattr(lib(), f"rdgpu_dinf_flowdirs_{s}")(dem.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), w, h,
                                                     out.ctypes.data_as(ctypes.c_void_p)), "rdgpu_dinf_flowdirs")
    return out


# method name -> (kind, code).  Names and aliases as in the reference's Python wrapper
# (wrappers/pyrichdem/richdem/__init__.py:532-549, 693-710).
_METHODS = {
    "Tarboton": ("tarboton", None), "Dinf": ("tarboton", None),
    "D8": ("d8", None), "OCallaghanD8": ("d8", None),
    "Holmgren": ("mfd", 0), "Freeman": ("mfd", 1), "Quinn": ("mfd", 2),
    "D4": ("mfd", 3), "OCallaghanD4": ("mfd", 3),
}
_NEEDS_EXPONENT = ("Holmgren", "Freeman")
_RANDOM_METHODS = ("Rho8", "Rho4", "FairfieldLeymarieD8", "FairfieldLeymarieD4")
VALID_METHODS = ["Tarboton", "Dinf", "Quinn", "FairfieldLeymarieD8", "FairfieldLeymarieD4", "Rho8", "Rho4",
                 "OCallaghanD8", "OCallaghanD4", "D8", "D4", "Freeman", "Holmgren"]


def _method(who: str, method, exponent):
    if method in _RANDOM_METHODS:
        raise RdgpuError(f"{who}: {method} draws from the reference's process-global random engine in raster order; "
                         "its output cannot be reproduced by a parallel engine and is not provided")
    if method not in _METHODS:
        raise RdgpuError(f"Invalid {who} method. Valid methods are: " + ", ".join(VALID_METHODS))
    if method in _NEEDS_EXPONENT and exponent is None:
        raise RdgpuError(f'{who} method "{method}" requires an exponent!')
    kind, code = _METHODS[method]
    return kind, code, float(exponent) if (exponent is not None and method in _NEEDS_EXPONENT) else 1.0


def FlowProportions(dem: np.ndarray, method: str = "Dinf", nodata=-9999, exponent=None) -> np.ndarray:
    """[h, w, 9] float32 flow proportions (reference ``rd.FlowProportions`` -> FM_Tarboton / FM_D8 /
    FM_Holmgren / FM_Freeman / FM_Quinn / FM_D4, flowmet/*.hpp)."""
    kind, code, xp = _method("FlowProportions", method, exponent)
    dem, s = _elev(dem, "FlowProportions", mfd=kind != "d8")
    h, w = dem.shape
    out = np.empty((h, w, 9), np.float32)
    pd, po = dem.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)
    if kind == "mfd":
        check(getattr(lib(), f"rdgpu_fm_mfd_{s}")(pd, _scalar(s, nodata), w, h, code, ctypes.c_double(xp), po), "rdgpu_fm_mfd")
    else:
        check(getattr(lib(), f"rdgpu_fm_d8_{s}" if kind == "d8" else f"rdgpu_fm_tarboton_{s}")(pd, _scalar(s, nodata), w, h, po),
              "rdgpu_fm_" + kind)
    return out


def FlowAccumFromProps(props: np.ndarray, weights: np.ndarray | None = None) -> np.ndarray:
    """Generic accumulation over a 9-float proportions array (reference ``rd.FlowAccumFromProps`` ->
    FlowAccumulation, methods/flow_accumulation_generic.hpp:33-100)."""
    if not isinstance(props, np.ndarray) or props.ndim != 3 or props.shape[2] != 9:
        raise RdgpuError("FlowAccumFromProps: expected an [h, w, 9] array")
    props = np.ascontiguousarray(props, dtype=np.float32)
    h, w, _ = props.shape
    if weights is None:
        acc = np.ones((h, w), np.float64)
    else:
        if weights.shape != (h, w):
            raise RdgpuError("Accumulation array must have same dimensions as proportions array!")
        acc = np.ascontiguousarray(weights, dtype=np.float64).copy()
    check(lib().rdgpu_flow_accumulation_f64(props.ctypes.data_as(ctypes.c_void_p), w, h,
                                            acc.ctypes.data_as(ctypes.c_void_p)), "rdgpu_flow_accumulation_f64")
    return acc


def flow_accumulation_into(dem: np.ndarray, method, nodata, acc: np.ndarray, exponent=None) -> None:
    """FA_<method>(dem, acc): acc (float64, C-contiguous, dem's shape) holds the flow generated per cell on
    entry and the accumulation on return (methods/flow_accumulation.hpp:16-28)."""
    kind, code, xp = _method("FlowAccumulation", method, exponent)
    dem, s = _elev(dem, "FlowAccumulation", mfd=kind != "d8")
    h, w = dem.shape
    if acc.shape != dem.shape:                     # flow_accumulation_generic.hpp:42-43
        raise RdgpuError("Accumulation array must have same dimensions as proportions array!")
    if acc.dtype != np.float64 or not acc.flags["C_CONTIGUOUS"]:
        raise RdgpuError("Accumulation array must be of type 'float64'!")
    pd, pa = dem.ctypes.data_as(ctypes.c_void_p), acc.ctypes.data_as(ctypes.c_void_p)
    if kind == "mfd":
        check(getattr(lib(), f"rdgpu_fa_mfd_{s}")(pd, _scalar(s, nodata), w, h, code, ctypes.c_double(xp), pa), "rdgpu_fa_mfd")
    else:
        check(getattr(lib(), f"rdgpu_fa_d8_{s}" if kind == "d8" else f"rdgpu_fa_tarboton_{s}")(pd, _scalar(s, nodata), w, h, pa),
              "rdgpu_fa_" + kind)


def FlowAccumulation(dem: np.ndarray, method: str = "D8", nodata=-9999, weights: np.ndarray | None = None, exponent=None):
    """Flow accumulation (reference ``rd.FlowAccumulation(dem, method, exponent, weights)``,
    wrappers/pyrichdem/richdem/__init__.py:490-597 -> FA_D8 etc., methods/flow_accumulation.hpp:16-28).
    Returns float64 accumulation; NoData cells get -1."""
    if not isinstance(dem, np.ndarray) or dem.ndim != 2:
        raise RdgpuError("FlowAccumulation: expected a 2-D numpy array")
    if weights is None:
        kind, _, _ = _method("FlowAccumulation", method, exponent)
        if kind == "d8":                           # every cell generates 1 (__init__.py:560-563) and THIS code knows it:
            d, s = _elev(dem, "FlowAccumulation", mfd=False)   # the unit-weights entry neither reads nor uploads an array of ones
            acc = np.empty(d.shape, np.float64)
            check(getattr(lib(), f"rdgpu_fa_d8_unit_{s}")(d.ctypes.data_as(ctypes.c_void_p), _scalar(s, nodata), d.shape[1], d.shape[0],
                                                          acc.ctypes.data_as(ctypes.c_void_p)), "rdgpu_fa_d8_unit")
            return acc
        acc = np.ones(dem.shape, np.float64)
    else:
        if weights.shape != dem.shape:             # flow_accumulation_generic.hpp:42-43
            raise RdgpuError("Accumulation array must have same dimensions as proportions array!")
        acc = np.ascontiguousarray(weights, dtype=np.float64).copy()
    flow_accumulation_into(dem, method, nodata, acc, exponent)
    return acc


# ---- HBM-resident variants (torch tensors on the GPU) ---------------------------------------
def _torch_suffix(t) -> str:
    import torch

    m = {torch.int8: "i8", torch.uint8: "u8", torch.int16: "i16", torch.int32: "i32", torch.float32: "f32",
         torch.float64: "f64", torch.int64: "i64"}
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


def _dev2d(t, who, dtype=None):
    if not (t.is_cuda and t.dim() == 2 and t.is_contiguous()):
        raise RdgpuError(f"{who}: expected a contiguous 2-D tensor on the GPU")
    if dtype is not None and t.dtype != dtype:
        raise RdgpuError(f"{who}: expected dtype {dtype}, got {t.dtype}")
    return t.shape


def _torch_elev_suffix(t) -> str:
    import torch

    m = {torch.int8: "i8", torch.uint8: "u8", torch.int16: "i16", torch.int32: "i32", torch.float32: "f32",
         torch.float64: "f64", torch.int64: "i64"}
    if t.dtype not in m:
        raise RdgpuError(f"unsupported tensor dtype {t.dtype}")
    return m[t.dtype]


def d8_flow_directions_dev(dem, nodata, dirs, flats: bool = False) -> None:
    """dirs (uint8 CUDA tensor) <- D8 directions of dem; flats=True also resolves flats
    (barnes_flat_resolution_d8, alter=false)."""
    import torch

    h, w = _dev2d(dem, "d8_flow_directions_dev")
    if _dev2d(dirs, "d8_flow_directions_dev", torch.uint8) != (h, w):
        raise RdgpuError("d8_flow_directions_dev: shape mismatch")
    s = _torch_elev_suffix(dem)
    name = f"rdgpu_flat_resolution_d8_dev_{s}" if flats else f"rdgpu_d8_flowdirs_dev_{s}"
    check(getattr(lib(), name)(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h, ctypes.c_void_p(dirs.data_ptr()),
                               _stream_ptr()), name)


def d8_flow_accum_dev(dirs, area, nodata: int = 255) -> None:
    import torch

    h, w = _dev2d(dirs, "d8_flow_accum_dev", torch.uint8)
    if _dev2d(area, "d8_flow_accum_dev") != (h, w):
        raise RdgpuError("d8_flow_accum_dev: shape mismatch")
    s = {torch.int32: "i32", torch.float32: "f32", torch.float64: "f64"}.get(area.dtype)
    if s is None:
        raise RdgpuError(f"d8_flow_accum_dev: unsupported accumulation dtype {area.dtype}")
    check(getattr(lib(), f"rdgpu_d8_flow_accum_dev_{s}")(ctypes.c_void_p(dirs.data_ptr()), ctypes.c_uint8(nodata), w, h,
                                                         ctypes.c_void_p(area.data_ptr()), _stream_ptr()),
          "rdgpu_d8_flow_accum_dev")


def fa_d8_dev(dem, nodata, accum, unit_weights: bool = False) -> None:
    """accum (float64 CUDA tensor, pre-loaded with per-cell weights) <- FA_D8 accumulation.  unit_weights=True: the caller
    guarantees that every weight is 1 (accum is then output only; the engine does not read the weights to find out)."""
    import torch

    h, w = _dev2d(dem, "fa_d8_dev")
    if _dev2d(accum, "fa_d8_dev", torch.float64) != (h, w):
        raise RdgpuError("Accumulation array must have same dimensions as proportions array!")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_fa_d8_unit_dev_{s}" if unit_weights else f"rdgpu_fa_d8_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h,
                                                 ctypes.c_void_p(accum.data_ptr()), _stream_ptr()), "rdgpu_fa_d8_dev")


def resolve_flats_epsilon_dev(dem, nodata) -> None:
    """In-place ResolveFlatsEpsilon (flats/flats.hpp:21-28) of a contiguous 2-D CUDA tensor, on torch's current stream."""
    h, w = _dev2d(dem, "resolve_flats_epsilon_dev")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_resolve_flats_epsilon_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h,
                                                               _stream_ptr()), "rdgpu_resolve_flats_epsilon_dev")


class _FlatStats(ctypes.Structure):
    _fields_ = [("low", ctypes.c_uint64), ("high", ctypes.c_uint64), ("noflow", ctypes.c_uint64),
                ("away", ctypes.c_uint32), ("towards", ctypes.c_uint32)]


class _FlatAsyncStats(ctypes.Structure):
    _fields_ = [("visits", ctypes.c_uint64), ("launches", ctypes.c_uint32), ("failures", ctypes.c_uint32),
                ("live_tiles", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


def flat_stats() -> dict:
    """rdgpu_flat_get_stats + rdgpu_flat_get_async_stats of the last flat resolution on this thread."""
    st = _FlatStats()
    check(lib().rdgpu_flat_get_stats(ctypes.byref(st)), "rdgpu_flat_get_stats")
    a = _FlatAsyncStats()
    check(lib().rdgpu_flat_get_async_stats(ctypes.byref(a)), "rdgpu_flat_get_async_stats")
    return {"low_edges": st.low, "high_edges": st.high, "noflow": st.noflow, "away": st.away, "towards": st.towards,
            "tail_visits": a.visits, "tail_live_tiles": a.live_tiles, "tail_launches": a.launches, "tail_failures": a.failures}


def release_workspace() -> None:
    """Free the grow-only device workspace cached between calls (rdgpu_release_workspace)."""
    check(lib().rdgpu_release_workspace(), "rdgpu_release_workspace")


def fill_epsilon_dev(dem, nodata, topology="D8") -> None:
    """In-place PriorityFloodEpsilon of a contiguous 2-D float32 / float64 CUDA tensor, on torch's current stream."""
    import torch

    h, w = _dev2d(dem, "fill_epsilon_dev")
    if dem.dtype not in (torch.float32, torch.float64):
        raise RdgpuError("Priority-Flood+Epsilon is only available for floating-point data types!")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_fill_epsilon_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h,
                                                        _topo(topology), _stream_ptr()), "rdgpu_fill_epsilon_dev")


def dinf_flow_directions_dev(dem, nodata, angles) -> None:
    """dinf_flow_directions (flowmet/dinf_flowdirs.hpp:128-152) of a contiguous 2-D CUDA tensor into a float32 CUDA tensor
    of the same shape, on torch's current stream."""
    import torch

    h, w = _dev2d(dem, "dinf_flow_directions_dev")
    if angles.dtype != torch.float32 or tuple(angles.shape) != (h, w) or not angles.is_contiguous() or not angles.is_cuda:
        raise RdgpuError("dinf_flow_directions_dev: angles must be a contiguous float32 CUDA tensor of the DEM's shape")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_dinf_flowdirs_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h,
                                                         ctypes.c_void_p(angles.data_ptr()), _stream_ptr()), "rdgpu_dinf_flowdirs_dev")


def fa_tarboton_dev(dem, nodata, accum) -> None:
    """FA_Tarboton / FA_Dinfinity (methods/flow_accumulation.hpp:16-17): accum (float64 CUDA tensor, in: the cells' weights,
    out: the accumulation) from the DEM, on torch's current stream."""
    import torch

    h, w = _dev2d(dem, "fa_tarboton_dev")
    if accum.dtype != torch.float64 or tuple(accum.shape) != (h, w) or not accum.is_contiguous() or not accum.is_cuda:
        raise RdgpuError("fa_tarboton_dev: accum must be a contiguous float64 CUDA tensor of the DEM's shape")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_fa_tarboton_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h,
                                                       ctypes.c_void_p(accum.data_ptr()), _stream_ptr()), "rdgpu_fa_tarboton_dev")


class _MaxDepStats(ctypes.Structure):
    _fields_ = [("pockets", ctypes.c_uint64), ("tie_pockets", ctypes.c_uint64), ("tie_cluster_cells", ctypes.c_uint64),
                ("pocket_cells", ctypes.c_uint64)]


def max_dep_stats() -> dict:
    """Tie census of the calling thread's last max_dep fill: pockets, pockets that several cells of their spill elevation
    can flood (there the reference's heap order decides the grouping), and the cells of the clusters they touch."""
    st = _MaxDepStats()
    check(lib().rdgpu_fill_max_dep_get_stats(ctypes.byref(st)), "rdgpu_fill_max_dep_get_stats")
    return {k: int(getattr(st, k)) for k, _ in _MaxDepStats._fields_}


def fill_max_dep_ties_dev(dem, max_dep_size: int, tie_mask, topology="D8") -> None:
    """fill_max_dep_dev + the cells of tie-flagged pocket clusters as a uint8 CUDA tensor (1 = the reference's heap order can
    decide this cell, 0 = order free)."""
    import torch

    h, w = _dev2d(dem, "fill_max_dep_ties_dev")
    if tie_mask.dtype != torch.uint8 or tuple(tie_mask.shape) != (h, w) or not tie_mask.is_contiguous() or not tie_mask.is_cuda:
        raise RdgpuError("fill_max_dep_ties_dev: tie_mask must be a contiguous uint8 CUDA tensor of the DEM's shape")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_fill_max_dep_ties_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), w, h, _topo(topology),
                                                             ctypes.c_uint64(int(max_dep_size)), ctypes.c_void_p(tie_mask.data_ptr()),
                                                             _stream_ptr()), "rdgpu_fill_max_dep_ties_dev")


def fill_max_dep_dev(dem, max_dep_size: int, topology="D8") -> None:
    """In-place PriorityFlood_Barnes2014_max_dep of a contiguous 2-D CUDA tensor, on torch's current stream."""
    h, w = _dev2d(dem, "fill_max_dep_dev")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_fill_max_dep_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), w, h, _topo(topology),
                                                        ctypes.c_uint64(int(max_dep_size)), _stream_ptr()), "rdgpu_fill_max_dep_dev")


def watersheds_dev(dem, nodata, labels, topology="D8", alter: bool = False) -> None:
    """PriorityFloodWatersheds_Barnes2014 of a contiguous 2-D CUDA tensor into an int32 CUDA tensor of the same shape
    (``alter``: the DEM is filled in place), on torch's current stream."""
    import torch

    h, w = _dev2d(dem, "watersheds_dev")
    if labels.dtype != torch.int32 or tuple(labels.shape) != (h, w) or not labels.is_contiguous() or not labels.is_cuda:
        raise RdgpuError("watersheds_dev: labels must be a contiguous int32 CUDA tensor of the DEM's shape")
    s = _torch_elev_suffix(dem)
    check(getattr(lib(), f"rdgpu_watersheds_dev_{s}")(ctypes.c_void_p(dem.data_ptr()), _scalar(s, nodata), w, h, _topo(topology),
                                                      1 if alter else 0, ctypes.c_void_p(labels.data_ptr()), _stream_ptr()),
          "rdgpu_watersheds_dev")


class _EpsStats(ctypes.Structure):
    _fields_ = [("rounds", ctypes.c_uint32), ("attempts", ctypes.c_uint32), ("tile_relaxations", ctypes.c_uint64),
                ("slack", ctypes.c_uint64), ("max_lift", ctypes.c_uint64), ("tie_sources", ctypes.c_uint64)]


def epsilon_stats() -> dict:
    st = _EpsStats()
    check(lib().rdgpu_fill_epsilon_get_stats(ctypes.byref(st)), "rdgpu_fill_epsilon_get_stats")
    return {k: getattr(st, k) for k, _ in _EpsStats._fields_}
