"""The reference's Python interface over the GPU engine (richdem_amd/pyrichdem.py mirrors
wrappers/pyrichdem/richdem/__init__.py): metadata semantics on CPU, results on the GPU."""
import numpy as np
import pytest


def test_rdarray_metadata_semantics(rd):
    a = rd.rdarray(np.arange(12, dtype=np.float32).reshape(3, 4), no_data=-9999, geotransform=(0, 2, 0, 0, 0, -2))
    assert type(a) is rd.rdarray and a.no_data == -9999 and a.metadata == {} and a.projection == ""
    assert tuple(a.geotransform) == (0, 2, 0, 0, 0, -2)
    b = a[1:]                                   # views and copies inherit the metadata (deep copies)
    assert type(b) is rd.rdarray and b.no_data == -9999
    c = a.copy(); c.metadata["k"] = 1
    assert a.metadata == {}
    d = rd.rdarray(np.zeros((3, 4)), meta_obj=a, no_data=-1)
    assert d.no_data == -1 and tuple(d.geotransform) == (0, 2, 0, 0, 0, -2)
    with pytest.raises(Exception, match="no_data value must be specified"):
        rd.rdarray(np.zeros((2, 2)))
    with pytest.raises(Exception, match="no_data value must be specified"):
        rd.rd3array(np.zeros((2, 2, 9)))
    p = rd.rd3array(np.zeros((2, 2, 9), np.float64), meta_obj=a, no_data=-2)
    assert p.dtype == np.float32 and p.no_data == -2


def test_argument_errors_without_gpu(rd):
    z = rd.rdarray(np.zeros((4, 4), np.float32), no_data=-1)
    with pytest.raises(Exception, match="rdarray or numpy.ndarray is required"):
        rd.FillDepressions([[1, 2], [3, 4]])
    with pytest.raises(Exception, match="Unknown topology"):
        rd.FillDepressions(z, topology="D6")
    with pytest.raises(rd.RdgpuError, match="only available for floating-point"):    # Barnes2014.hpp:424-451
        rd.FillDepressions(rd.rdarray(np.zeros((4, 4), np.int32), no_data=-1), epsilon=True)
    with pytest.raises(Exception, match="Invalid FlowAccumulation method"):
        rd.FlowAccumulation(z, method="Foo")
    with pytest.raises(Exception, match="Invalid FlowAccumulation method"):
        rd.FlowAccumulation(z)                                   # the reference's default method=None is invalid too
    with pytest.raises(Exception, match='"Holmgren" requires an exponent'):
        rd.FlowAccumulation(z, method="Holmgren")
    with pytest.raises(Exception, match="must be of type 'float64'"):
        rd.FlowAccumulation(z, method="D8", weights=rd.rdarray(np.ones((4, 4), np.float32), no_data=-1))
    with pytest.raises(Exception, match="rd3array or numpy.ndarray is required"):
        rd.FlowAccumFromProps(z)
    for name in ("BreachDepressions", "TerrainAttribute", "LoadGDAL", "SaveGDAL"):
        with pytest.raises(rd.RdgpuError, match="outside"):
            getattr(rd, name)(z)


@pytest.mark.gpu
def test_reference_style_session(rd, orc):
    """docs/flow_accumulation.rst-style use: fill in place, accumulate with every method family."""
    from richdem_amd.synth import fractal_dem

    raw = fractal_dem(240, 180, 77)
    nd = np.float32(-9999)
    raw[50:54, 60:70] = nd
    dem = rd.rdarray(raw.copy(), no_data=-9999, geotransform=(10, 1, 0, 20, 0, -1))
    filled = rd.FillDepressions(dem)                                        # returns a new rdarray
    assert type(filled) is rd.rdarray and filled.no_data == -9999 and np.array_equal(dem, raw)
    assert np.array_equal(np.asarray(filled), orc.port.fill(raw))
    assert "FillDepressions(dem, epsilon=False)" in filled.metadata["PROCESSING_HISTORY"]
    assert rd.FillDepressions(dem, in_place=True) is None and np.array_equal(dem, filled)
    d4 = rd.FillDepressions(rd.rdarray(raw.copy(), no_data=-9999), topology="D4")
    assert np.array_equal(np.asarray(d4), orc.port.fill(raw, 4))

    acc = rd.FlowAccumulation(dem, method="D8")
    assert type(acc) is rd.rdarray and acc.no_data == -1 and acc.dtype == np.float64
    assert tuple(acc.geotransform) == (10, 1, 0, 20, 0, -1)
    assert np.array_equal(np.asarray(acc), orc.port.fa_d8(np.asarray(dem), nd))
    w = rd.rdarray(np.random.default_rng(1).integers(0, 5, dem.shape).astype(np.float64), no_data=-1)
    w0 = np.asarray(w).copy()
    acc2 = rd.FlowAccumulation(dem, method="D8", weights=w)                # weights untouched
    assert np.array_equal(w, w0) and np.array_equal(np.asarray(acc2), orc.port.fa_d8(np.asarray(dem), nd, w0))
    acc3 = rd.FlowAccumulation(dem, method="D8", weights=w, in_place=True)  # a view of the modified weights
    assert np.shares_memory(acc3, w) and np.array_equal(np.asarray(w), np.asarray(acc2))
    for method, x in (("Dinf", None), ("Tarboton", None), ("Quinn", None), ("Holmgren", 1.5), ("Freeman", 1.1), ("D4", None),
                      ("OCallaghanD8", None), ("OCallaghanD4", None)):
        a = np.asarray(rd.FlowAccumulation(dem, method=method, exponent=x))
        if method in ("Dinf", "Tarboton"):
            e = orc.port.fa_tarboton(np.asarray(dem), nd)
        elif method == "OCallaghanD8":
            e = orc.port.fa_d8(np.asarray(dem), nd)
        else:
            e = orc.port.fa_mfd(np.asarray(dem), nd, method.replace("OCallaghan", ""), 1.0 if x is None else x)
        assert np.allclose(a, e, rtol=1e-12, atol=0), method   # (end to end: the f64 sums differ in the last bits only, tests/test_mfd_gpu.py)

    props = rd.FlowProportions(dem, method="D8")
    assert type(props) is rd.rd3array and props.no_data == -2 and props.shape == dem.shape + (9,)
    assert np.array_equal(np.asarray(props), orc.port.fm_d8(np.asarray(dem), nd))
    acc4 = rd.FlowAccumFromProps(props)
    assert type(acc4) is rd.rdarray and np.array_equal(np.asarray(acc4), np.asarray(acc))
    acc5 = rd.FlowAccumFromProps(props, weights=rd.rdarray(w0, no_data=-1))
    assert np.array_equal(np.asarray(acc5), np.asarray(acc2))


@pytest.mark.gpu
def test_non_contiguous_and_integer_rdarrays(rd, orc):
    from richdem_amd.synth import fractal_dem_int

    base = fractal_dem_int(90, 140, 5, 0.3)
    view = rd.rdarray(base, no_data=-9999)[:, ::2]                       # a strided view
    got = rd.FillDepressions(view)
    assert np.array_equal(np.asarray(got), orc.port.fill(np.ascontiguousarray(base[:, ::2])))
    with pytest.raises(rd.RdgpuError):
        rd.FillDepressions(view, in_place=True)
    acc = rd.FlowAccumulation(rd.rdarray(base.astype(np.int16), no_data=-9999), method="D8")
    assert np.array_equal(np.asarray(acc), orc.port.fa_d8(base.astype(np.int16), np.int16(-9999)))


def _ref_native(orc):
    import ctypes

    if not orc.ref.available:
        pytest.skip("oracle/_ref/libref.so not built here")
    return orc.ref.lib, ctypes


@pytest.mark.parametrize("dtype,suf", [(np.float32, "f32"), (np.int32, "i32"), (np.uint8, "u8"), (np.float64, "f64")])
def test_native_format_matches_the_reference(rd, orc, tmp_path, dtype, suf):
    """SaveNative / LoadNative vs the reference's own saveToCache / Array2D(filename, native=true)
    (common/Array2D.hpp:209-281): byte-identical files, and each side reads the other's."""
    L, ctypes = _ref_native(orc)
    rng = np.random.default_rng(4)
    a = (rng.random((37, 53)) * 200).astype(dtype)
    gt = (100.5, 2.0, 0.0, 900.25, 0.0, -2.0)
    r = rd.rdarray(a, no_data=dtype(7), geotransform=gt)
    r.projection = "PROJCS[\"made up\"]"
    mine, theirs = str(tmp_path / "mine.rd"), str(tmp_path / "theirs.rd")
    rd.SaveNative(mine, r)
    gtc = (ctypes.c_double * 6)(*gt)
    ct = {"f32": ctypes.c_float, "i32": ctypes.c_int32, "u8": ctypes.c_uint8, "f64": ctypes.c_double}[suf]
    save = getattr(L, f"ref_native_save_{suf}")
    save.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ct, ctypes.c_void_p, ctypes.c_char_p]
    assert save(theirs.encode(), a.ctypes.data_as(ctypes.c_void_p), 53, 37, ct(7), gtc, r.projection.encode()) == 0
    assert open(mine, "rb").read() == open(theirs, "rb").read()
    back = rd.LoadNative(theirs, dtype)
    assert type(back) is rd.rdarray and np.array_equal(back, a) and back.no_data == 7
    assert tuple(back.geotransform) == gt and back.projection == r.projection
    # the reference reads ours
    load = getattr(L, f"ref_native_load_{suf}")
    out = np.zeros_like(a)
    w, h, nd = ctypes.c_int(53), ctypes.c_int(37), ct(0)
    gto, proj = (ctypes.c_double * 6)(), ctypes.create_string_buffer(256)
    assert load(mine.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(w), ctypes.byref(h), ctypes.byref(nd), gto,
                proj, 256) == 0
    assert np.array_equal(out, a) and nd.value == 7 and tuple(gto) == gt and proj.value.decode() == r.projection


def test_native_format_errors(rd, tmp_path):
    with pytest.raises(Exception):
        rd.LoadNative(str(tmp_path / "missing.rd"), np.float32)
    (tmp_path / "short.rd").write_bytes(b"\x01\x02")
    with pytest.raises(rd.RdgpuError):
        rd.LoadNative(str(tmp_path / "short.rd"), np.float32)
    with pytest.raises(Exception, match="rdarray"):
        rd.SaveNative(str(tmp_path / "x.rd"), np.zeros((2, 2)))
