"""`_richdem` on the engine, on the GPU: what the reference's richdem/__init__.py calls (pywrapper.hpp:27-82,
pywrapper.cpp:50) against the oracle, through the pybind11 module and the C++ shim."""
import numpy as np
import pytest

from richdem_amd.synth import fractal_dem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from richdem_amd import compat

    if compat.module_path() is None:   # normally built by __graft_entry__.build() and shipped in-tree
        compat.build()
    return compat.load()               # no fallback


def wrap(R, arr, nodata):
    name = {"float32": "float", "float64": "double", "int16": "int16_t", "int32": "int32_t", "uint8": "uint8_t",
            "uint16": "uint16_t", "uint32": "uint32_t", "int64": "int64_t", "uint64": "uint64_t"}[str(arr.dtype)]
    w = getattr(R, "Array2D_" + name)(arr)
    w.setNoData(nodata)
    w.geotransform = np.array([0, 1, 0, 0, 0, -1], dtype="float64")
    return w


def test_fill_in_place_every_type(R, orc):
    z = fractal_dem(300, 220, seed=5)
    for dt, scale in ((np.float32, 1.0), (np.float64, 1.0), (np.int32, 0.2), (np.int16, 0.1), (np.uint8, 0.03), (np.uint16, 0.5),
                      (np.uint32, 2.0), (np.int64, 0.2), (np.uint64, 0.2)):
        dem = (np.floor((z - z.min()) * scale) if np.issubdtype(dt, np.integer) else z * scale).astype(dt)
        for fn, topo in ((R.rdFillDepressionsD8, 8), (R.rdFillDepressionsD4, 4)):
            a = dem.copy()
            assert fn(wrap(R, a, -9999 if np.issubdtype(dt, np.signedinteger) or dt in (np.float32, np.float64) else 0)) is None
            assert np.array_equal(a, orc.port.fill(dem, topo)), (dt, topo)   # the numpy array itself was filled


def test_the_sweeps_other_names(R, orc):
    """r06: PriorityFlood_Original<D8/D4>, PriorityFlood_Wei2018 (NoData holes drain) and HasDepressions<D8/D4> through the
    module (the reference binds only the default fill; these are extras under `rd...` names)."""
    dem = fractal_dem(260, 190, seed=8).copy()
    for fn, topo in ((R.rdPriorityFloodOriginalD8, 8), (R.rdPriorityFloodOriginalD4, 4)):
        a = dem.copy()
        fn(wrap(R, a, -9999))
        assert np.array_equal(a, orc.port.fill(dem, topo))
    holes = dem.copy()
    holes[70:80, 100:140] = -9999
    a = holes.copy()
    R.rdPriorityFloodWei2018(wrap(R, a, -9999))
    assert np.array_equal(a, orc.port.fill_wei2018(holes, np.float32(-9999)))
    assert not np.array_equal(a, orc.port.fill(holes, 8))
    assert R.rdHasDepressionsD8(wrap(R, dem.copy(), -9999)) is True
    assert R.rdHasDepressionsD4(wrap(R, orc.port.fill(dem, 4), -9999)) is False
    i16 = np.floor((dem - dem.min()) * 0.1).astype(np.int16)
    assert R.rdHasDepressionsD8(wrap(R, i16, -9999)) == orc.port.has_depressions(i16, 8)


def test_accumulation_families(R, orc):
    raw = fractal_dem(240, 180, 77)
    nd = np.float32(-9999)
    raw[50:54, 60:70] = nd
    dem = orc.port.fill(raw)
    demw = wrap(R, dem, -9999)
    wts = np.random.default_rng(1).integers(0, 5, dem.shape).astype(np.float64)

    def run(fn, *extra, weights=None):
        acc = np.ones(dem.shape, np.float64) if weights is None else weights.copy()
        accw = R.Array2D_double(acc)
        fn(demw, accw, *extra)
        assert accw.noData() == -1                                      # ACCUM_NO_DATA
        return acc

    assert np.array_equal(run(R.FA_D8), orc.port.fa_d8(dem, nd))
    assert np.array_equal(run(R.FA_OCallaghanD8), orc.port.fa_d8(dem, nd))
    assert np.array_equal(run(R.FA_D8, weights=wts), orc.port.fa_d8(dem, nd, wts))
    for fn, method, x in ((R.FA_Quinn, "Quinn", None), (R.FA_Holmgren, "Holmgren", 1.5), (R.FA_Freeman, "Freeman", 1.1),
                          (R.FA_D4, "D4", None), (R.FA_OCallaghanD4, "D4", None)):
        got = run(fn, *(() if x is None else (x,)))
        assert np.allclose(got, orc.port.fa_mfd(dem, nd, method, 1.0 if x is None else x), rtol=1e-12, atol=0), method   # (end to end: the f64 sums differ in the last bits only, tests/test_mfd_gpu.py)
    for fn in (R.FA_Tarboton, R.FA_Dinfinity):
        assert np.allclose(run(fn), orc.port.fa_tarboton(dem, nd), rtol=1e-12, atol=0)

    # proportions (Array3D_float: nine slots per cell) and the generic accumulation over them
    props = np.zeros(dem.shape + (9,), np.float32)
    pw = R.Array3D_float(props)
    R.FM_D8(demw, pw)
    assert pw.noData() == -2 and np.array_equal(props, orc.port.fm_d8(dem, nd))
    acc = np.ones(dem.shape, np.float64)
    R.FlowAccumulation(pw, R.Array2D_double(acc))
    assert np.array_equal(acc, orc.port.fa_d8(dem, nd))
    R.FM_OCallaghanD8(demw, pw)
    assert np.array_equal(props, orc.port.fm_d8(dem, nd))
    for fn, method, x in ((R.FM_Quinn, "Quinn", None), (R.FM_Holmgren, "Holmgren", 2.0), (R.FM_Freeman, "Freeman", 1.1),
                          (R.FM_D4, "D4", None), (R.FM_OCallaghanD4, "D4", None)):
        fn(demw, pw, *(() if x is None else (x,)))
        exp = orc.port.fm_mfd(dem, nd, method, 1.0 if x is None else x)
        assert np.allclose(props, exp, rtol=3e-7, atol=0), method
    for fn in (R.FM_Tarboton, R.FM_Dinfinity):
        fn(demw, pw)
        assert np.allclose(props, orc.port.fm_tarboton(dem, nd), rtol=3e-7, atol=1e-7)
    with pytest.raises(RuntimeError, match="dimensions"):
        R.FM_D8(demw, R.Array3D_float(np.zeros((5, 5, 9), np.float32)))


def test_resolve_flats_epsilon(R, orc):
    z = np.floor(fractal_dem(200, 160, seed=9) * 0.05).astype(np.float32)
    dem = orc.port.fill(z)
    a = dem.copy()
    R.rdResolveFlatsEpsilon(wrap(R, a, -9999))
    assert np.array_equal(a, orc.port.resolve_flats_epsilon(dem, np.float32(-9999)))


def test_epsilon_fill_warns_with_the_tie_count(R, orc):
    """rd.FillDepressions(epsilon=True) through `_richdem`: on an integer-valued DEM the reference's surface follows its heap's
    pop order; the binding says how many gradient sources shared an elevation (a RuntimeWarning with the device's count),
    and stays silent -- and equal to the restatement -- on a tie-free DEM."""
    import warnings

    import richdem_amd as rd

    z = np.floor(fractal_dem(180, 140, seed=12) * 0.05).astype(np.float32)
    with pytest.warns(RuntimeWarning, match=r"\d+ gradient sources share their elevation"):
        R.rdPFepsilonD8(wrap(R, z.copy(), -9999))
    assert rd.epsilon_stats()["tie_sources"] > 0
    rng = np.random.default_rng(4)
    free = (rng.permutation(120 * 90).reshape(90, 120) * 0.25).astype(np.float32)
    a = free.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        R.rdPFepsilonD8(wrap(R, a, -9999))
    assert np.array_equal(a, orc.port.fill_epsilon(free, np.float32(-9999), 8))
