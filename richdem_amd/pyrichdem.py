"""The reference's Python interface for the hot path (wrappers/pyrichdem/richdem/__init__.py), over the GPU
library: ``rdarray`` / ``rd3array`` (numpy subclasses that carry no_data, geotransform, projection and
metadata), ``FillDepressions``, ``FlowAccumulation``, ``FlowProportions``, ``FlowAccumFromProps`` with the
reference's argument names, defaults, return conventions and error messages, so that

    import richdem_amd as rd
    dem = rd.rdarray(np.load("dem.npy"), no_data=-9999)
    rd.FillDepressions(dem, in_place=True)
    acc = rd.FlowAccumulation(dem, method="D8")

reads like the reference's own examples (docs/flow_accumulation.rst).  Every function also accepts a plain
``numpy.ndarray`` (then ``nodata=`` names the NoData value and plain arrays come back).  GDAL I/O
(LoadGDAL / SaveGDAL), breaching, terrain attributes and the depression hierarchy are outside this engine;
``ResolveFlats`` and the native raster format (``LoadNative`` / ``SaveNative``) are provided.
"""
from __future__ import annotations

import copy
import datetime

import numpy as np

from . import api as _api
from ._lib import RdgpuError

__all__ = ["rdarray", "rd3array", "FillDepressions", "FlowAccumulation", "FlowProportions", "FlowAccumFromProps",
           "ResolveFlats", "BreachDepressions", "TerrainAttribute", "LoadGDAL", "SaveGDAL", "LoadNative", "SaveNative"]

_META = ("metadata", "no_data", "projection", "geotransform")
_META_DEFAULT = {"metadata": dict, "no_data": lambda: None, "projection": str, "geotransform": lambda: None}


def _inherit(dst, src) -> None:
    for name in _META:
        setattr(dst, name, copy.deepcopy(getattr(src, name, _META_DEFAULT[name]())))


class rdarray(np.ndarray):
    """2-D raster with RichDEM's metadata (reference class of the same name, __init__.py:155-223).
    ``no_data`` is mandatory, directly or through ``meta_obj``."""

    def __new__(cls, array, meta_obj=None, no_data=None, dtype=None, order=None, geotransform=None, **kwargs):
        obj = np.asarray(array, dtype=dtype, order=order)
        if kwargs.get("copy"):
            obj = obj.copy()
        obj = obj.view(cls)
        if meta_obj is not None:
            _inherit(obj, meta_obj)
        elif geotransform is not None:
            obj.geotransform = geotransform
        if no_data is not None:
            obj.no_data = no_data
        if no_data is None:   # (sic) the reference insists on the argument even when meta_obj carries one
            raise Exception("A no_data value must be specified!")
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            _inherit(self, obj)


class rd3array(np.ndarray):
    """[h, w, 9] float32 flow proportions with the same metadata (reference __init__.py:226-279)."""

    def __new__(cls, array, meta_obj=None, no_data=None, order=None, **kwargs):
        obj = np.asarray(array, dtype=np.float32, order=order).view(cls)
        if meta_obj is not None:
            _inherit(obj, meta_obj)
        if no_data is not None:
            obj.no_data = no_data
        if no_data is None:
            raise Exception("A no_data value must be specified!")
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            _inherit(self, obj)


def _version() -> str:
    return "RichDEM-rdgpu (MI355X engine, C-ABI include/rdgpu.h)"


def _add_analysis(rda, analysis: str) -> None:
    """PROCESSING_HISTORY line, as the reference's _AddAnalysis (__init__.py:34-48)."""
    if type(rda) not in (rdarray, rd3array):
        raise Exception("An rdarray or rd3array is required!")
    stamp = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M:%S.%f UTC")
    if rda.metadata is None:
        rda.metadata = dict()
    rda.metadata["PROCESSING_HISTORY"] = rda.metadata.get("PROCESSING_HISTORY", "") + f"\n{stamp} | {_version()} | {analysis}"


def _nodata_of(rda):
    if rda.no_data is None:
        print("Warning! no_data was None. Setting it to -9999!")   # rdarray.wrap, __init__.py:203-205
        return -9999
    return rda.no_data


def _plain(a) -> np.ndarray:
    return np.asarray(a).view(np.ndarray)


def FillDepressions(dem, epsilon: bool = False, in_place: bool = False, topology: str = "D8", shards: int = 1, nodata=None):
    """Fills all depressions in a DEM (reference __init__.py:381-422 -> rdFillDepressionsD8/D4 =
    PriorityFlood_Zhou2016 / PriorityFlood_Barnes2014<D4>, pywrapper.hpp:32-33).
    rdarray in: returns a new rdarray, or None when ``in_place``.  ``epsilon=True`` -> rdPFepsilonD8/D4 =
    PriorityFloodEpsilon_Barnes2014<topo> (pywrapper.hpp:34-35), floating-point DEMs only.  With equal elevations among
    the cells the reference's heap holds, its epsilon surface follows the heap's pop order; this engine returns the
    order-free surface (a cell-wise lower bound, identical when there are no such ties), warns, and says so in
    PROCESSING_HISTORY -- see ``richdem_amd.api.FillDepressions``."""
    if type(dem) is not rdarray:
        if isinstance(dem, np.ndarray) and type(dem) is np.ndarray:
            return _api.FillDepressions(dem, epsilon=epsilon, in_place=in_place, topology=topology, shards=shards,
                                        nodata=-9999 if nodata is None else nodata)
        raise Exception("A richdem.rdarray or numpy.ndarray is required!")
    if topology not in ["D8", "D4"]:
        raise Exception("Unknown topology!")
    if not in_place:
        dem = dem.copy()
    elif not dem.flags["C_CONTIGUOUS"]:
        raise RdgpuError("FillDepressions(in_place=True) needs a C-contiguous array")
    _add_analysis(dem, f"FillDepressions(dem, epsilon={epsilon})")
    work = _plain(dem)
    nd = (_nodata_of(dem) if nodata is None else nodata) if epsilon else -9999
    if not work.flags["C_CONTIGUOUS"]:
        filled = _api.FillDepressions(np.ascontiguousarray(work), epsilon=epsilon, in_place=False, topology=topology,
                                      shards=shards, nodata=nd)
        work[...] = filled
    else:
        _api.FillDepressions(work, epsilon=epsilon, in_place=True, topology=topology, shards=shards, nodata=nd)
    if epsilon:
        ties = _api.epsilon_stats()["tie_sources"]
        if ties:
            _add_analysis(dem, f"FillDepressions: {ties} gradient sources of equal elevation -- order-free epsilon surface "
                               "(lower bound of the reference's heap-order-dependent one)")
    if not in_place:
        return dem
    return None


def _accum_array(like, weights, in_place: bool):
    """The accumulation array the reference builds (__init__.py:551-567): the weights themselves when
    ``in_place``, a copy of them otherwise, ones when there are none; float64 or an exception."""
    if weights is not None and in_place:
        accum = rdarray(weights, no_data=-1)
    elif weights is not None:
        accum = rdarray(weights, copy=True, meta_obj=like, no_data=-1)
    else:
        accum = rdarray(np.ones(shape=like.shape[0:2], dtype="float64"), meta_obj=like, no_data=-1)
    if accum.dtype != "float64":
        raise Exception("Accumulation array must be of type 'float64'!")
    return accum


def FlowAccumulation(dem, method=None, exponent=None, weights=None, in_place: bool = False, nodata=None):
    """Flow accumulation by ``method`` (reference __init__.py:490-597).  Methods: D8 / OCallaghanD8, D4 /
    OCallaghanD4, Dinf / Tarboton, Quinn, Holmgren(E), Freeman(E); the randomised Rho8/Rho4 family is refused.
    Returns the accumulation rdarray (a view of ``weights`` when ``in_place``)."""
    if type(dem) is not rdarray:
        if isinstance(dem, np.ndarray) and type(dem) is np.ndarray:
            return _api.FlowAccumulation(dem, method="D8" if method is None else method,
                                         nodata=-9999 if nodata is None else nodata, weights=weights, exponent=exponent)
        raise Exception("A richdem.rdarray or numpy.ndarray is required!")
    accum = _accum_array(dem, weights, in_place)
    _add_analysis(accum, "FlowAccumulation(dem, method={method}, exponent={exponent}, weights={weights}, in_place={in_place})".format(
        method=method, exponent=exponent, weights="None" if weights is None else "weights", in_place=in_place))
    try:
        _api._method("FlowAccumulation", method, exponent)
    except RdgpuError as e:
        raise Exception(str(e)) from None
    acc = _plain(accum)
    if acc.flags["C_CONTIGUOUS"]:
        _api.flow_accumulation_into(_plain(dem), method, _nodata_of(dem), acc, exponent)
    else:
        tmp = np.ascontiguousarray(acc)
        _api.flow_accumulation_into(_plain(dem), method, _nodata_of(dem), tmp, exponent)
        acc[...] = tmp
    accum.no_data = -1.0     # copyFromWrapped: accum.setNoData(ACCUM_NO_DATA), flow_accumulation_generic.hpp:40
    return accum


def FlowProportions(dem, method=None, exponent=None, nodata=None):
    """Flow proportions [h, w, 9] by ``method`` (reference __init__.py:650-732)."""
    if type(dem) is not rdarray:
        if isinstance(dem, np.ndarray) and type(dem) is np.ndarray:
            return _api.FlowProportions(dem, method="Dinf" if method is None else method,
                                        nodata=-9999 if nodata is None else nodata, exponent=exponent)
        raise Exception("A richdem.rdarray or numpy.ndarray is required!")
    try:
        _api._method("FlowProportions", method, exponent)
    except RdgpuError as e:
        raise Exception(str(e)) from None
    props = _api.FlowProportions(_plain(dem), method=method, nodata=_nodata_of(dem), exponent=exponent)
    fprops = rd3array(props, meta_obj=dem, no_data=-2)
    _add_analysis(fprops, f"FlowProportions(dem, method={method}, exponent={exponent})")
    fprops.no_data = -2.0
    return fprops


def FlowAccumFromProps(props, weights=None, in_place: bool = False):
    """Flow accumulation from flow proportions (reference __init__.py:599-647 -> FlowAccumulation(Array3D))."""
    if type(props) is not rd3array:
        if isinstance(props, np.ndarray) and type(props) is np.ndarray:
            return _api.FlowAccumFromProps(props, weights)
        raise Exception("A richdem.rd3array or numpy.ndarray is required!")
    accum = _accum_array(props, weights, in_place)
    _add_analysis(accum, "FlowAccumFromProps(dem, weights={weights}, in_place={in_place})".format(
        weights="None" if weights is None else "weights", in_place=in_place))
    res = _api.FlowAccumFromProps(_plain(props), _plain(accum))
    _plain(accum)[...] = res
    accum.no_data = -1.0
    return accum


_NATIVE_GT = (1000.0, 1.0, 0.0, 1000.0, 0.0, -1.0)   # the reference's fallback geotransform (Array2D.hpp:149)


def SaveNative(filename: str, rda) -> None:
    """Write a raster in the reference's native format (``Array2D::saveToCache``, common/Array2D.hpp:209-246,
    uncompressed build): loads in the reference with ``Array2D<T>(filename, true)`` and in
    ``rdgpu::Array2D<T>(filename)``.  GDAL formats are outside this engine."""
    if type(rda) is not rdarray:
        raise Exception("A richdem.rdarray or numpy.ndarray is required!")
    a = np.ascontiguousarray(_plain(rda))
    if a.ndim != 2:
        raise RdgpuError("SaveNative: expected a 2-D raster")
    gt = tuple(rda.geotransform) if rda.geotransform is not None and len(rda.geotransform) >= 6 else _NATIVE_GT
    proj = (rda.projection or "").encode()
    with open(filename, "wb") as f:
        f.write(np.array([a.shape[0], a.shape[1], 0, 0], np.int32).tobytes())
        f.write(np.array([0xFFFFFFFF], np.uint32).tobytes())          # data cells: not counted
        f.write(np.array([_nodata_of(rda)], a.dtype).tobytes())
        f.write(np.array(gt[:6], np.float64).tobytes())
        f.write(np.array([len(proj)], np.uint64).tobytes())
        f.write(proj)
        f.write(a.tobytes())


def LoadNative(filename: str, dtype) -> "rdarray":
    """Read a raster in the reference's native format (``Array2D::loadNative``, common/Array2D.hpp:251-281).
    The element type is not stored in the file: ``dtype`` must be the one it was written with."""
    dt = np.dtype(dtype)
    with open(filename, "rb") as f:
        buf = f.read()
    if len(buf) < 20 + dt.itemsize + 56:
        raise RdgpuError(f"Failed to load native file '{filename}'!")
    h, w, _xoff, _yoff = np.frombuffer(buf, np.int32, 4, 0)
    pos = 20
    no_data = np.frombuffer(buf, dt, 1, pos)[0]; pos += dt.itemsize
    gt = np.frombuffer(buf, np.float64, 6, pos); pos += 48
    plen = int(np.frombuffer(buf, np.uint64, 1, pos)[0]); pos += 8
    if h < 0 or w < 0 or len(buf) < pos + plen + int(h) * int(w) * dt.itemsize:
        raise RdgpuError(f"Failed to load native file '{filename}'!")
    proj = buf[pos:pos + plen].decode(errors="replace"); pos += plen
    data = np.frombuffer(buf, dt, int(h) * int(w), pos).reshape(int(h), int(w)).copy()
    out = rdarray(data, no_data=no_data.item(), geotransform=tuple(gt.tolist()))
    out.projection = proj
    return out


def _outside(name: str):
    def fn(*args, **kwargs):
        raise RdgpuError(f"{name} is outside the GPU engine's scope (depression filling -> D8 directions / flat "
                         "resolution -> flow accumulation); use the reference implementation for it")
    fn.__name__ = name
    fn.__doc__ = f"{name}: not part of this engine (see DESIGN.md section 10)."
    return fn


def ResolveFlats(dem, in_place: bool = False, nodata=None):
    """Attempts to resolve flats by imposing a local gradient (reference __init__.py:461-487 ->
    rdResolveFlatsEpsilon = ResolveFlatsEpsilon, flats/flats.hpp:21-28).  rdarray in: returns a new rdarray, or
    None when ``in_place``."""
    if type(dem) is not rdarray:
        if isinstance(dem, np.ndarray) and type(dem) is np.ndarray:
            return _api.resolve_flats_epsilon(dem, -9999 if nodata is None else nodata, in_place=in_place)
        raise Exception("A richdem.rdarray or numpy.ndarray is required!")
    if not in_place:
        dem = dem.copy()
    elif not dem.flags["C_CONTIGUOUS"]:
        raise RdgpuError("ResolveFlats(in_place=True) needs a C-contiguous array")
    _add_analysis(dem, f"ResolveFlats(dem, in_place={in_place})")
    work = _plain(dem)
    if work.flags["C_CONTIGUOUS"]:
        _api.resolve_flats_epsilon(work, _nodata_of(dem), in_place=True)
    else:
        work[...] = _api.resolve_flats_epsilon(np.ascontiguousarray(work), _nodata_of(dem))
    if not in_place:
        return dem
    return None


BreachDepressions = _outside("BreachDepressions")
TerrainAttribute = _outside("TerrainAttribute")
LoadGDAL = _outside("LoadGDAL")
SaveGDAL = _outside("SaveGDAL")
