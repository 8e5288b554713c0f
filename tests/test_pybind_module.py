"""`_richdem` on the engine (wrappers/pyrichdem_gpu): the binding surface of the reference's extension module
(wrappers/pyrichdem/src/pywrapper.cpp, pywrapper.hpp), checked without a GPU -- builds, imports, wraps numpy memory
without copying, carries NoData / georeferencing, raises for what is out of scope -- and the reference's own
`richdem/__init__.py` imports and drives it unchanged (this container only: it reads /root/reference)."""
import gc
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

REF_PKG = "/root/reference/wrappers/pyrichdem"

# every _richdem.<name> the reference's richdem/__init__.py touches (kept here so the check also runs where the
# reference is absent; test_surface_matches_the_reference_package re-derives it from the source when it is present)
SURFACE = """Array2D_double Array2D_float Array2D_int16_t Array2D_int32_t Array2D_int64_t Array2D_int8_t Array2D_uint16_t
Array2D_uint32_t Array2D_uint64_t Array2D_uint8_t Array3D_float FA_D4 FA_D8 FA_FairfieldLeymarieD4 FA_FairfieldLeymarieD8
FA_Freeman FA_Holmgren FA_OCallaghanD4 FA_OCallaghanD8 FA_Quinn FA_Rho4 FA_Rho8 FA_Tarboton FM_D4 FM_D8
FM_FairfieldLeymarieD4 FM_FairfieldLeymarieD8 FM_Freeman FM_Holmgren FM_OCallaghanD4 FM_OCallaghanD8 FM_Quinn FM_Rho4
FM_Rho8 FM_Tarboton FlowAccumulation NO_FLOW TA_CTI TA_SPI TA_aspect TA_curvature TA_planform_curvature
TA_profile_curvature TA_slope_degrees TA_slope_percentage TA_slope_radians TA_slope_riserun generate_perlin_terrain
rdBreachDepressionsD4 rdBreachDepressionsD8 rdCompileTime rdFillDepressionsD4 rdFillDepressionsD8 rdHash rdPFepsilonD4
rdPFepsilonD8 rdResolveFlatsEpsilon depression_hierarchy FA_Dinfinity FM_Dinfinity MapStringString""".split()


@pytest.fixture(scope="module")
def R():
    from richdem_amd import compat

    compat.build()
    return compat.load()


def test_surface_is_complete(R):
    missing = [n for n in SURFACE if not hasattr(R, n)]
    assert not missing, missing
    assert R.NO_FLOW == 0 and R.engine == "rdgpu"
    dh = R.depression_hierarchy
    assert (dh.OCEAN, dh.NO_DEP, dh.NO_PARENT, dh.NO_VALUE) == (0, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)
    d = dh.Depression()   # defaults as the reference's own Python test expects (wrappers/pyrichdem/tests/tests.py:8-22)
    assert d.parent == dh.NO_PARENT and d.pit_elev == float("inf") and d.out_elev == float("inf") and d.ocean_linked == []
    assert d.out_cell == d.pit_cell == d.odep == d.geolink == d.lchild == d.rchild == dh.NO_VALUE
    assert d.ocean_parent is False and (d.dep_label, d.cell_count, d.dep_vol, d.water_vol, d.total_elevation) == (0, 0, 0, 0, 0)
    assert isinstance(R.rdHash(), str) and isinstance(R.rdCompileTime(), str)


@pytest.mark.skipif(not os.path.isdir(REF_PKG), reason="the reference tree is only present in the build container")
def test_surface_matches_the_reference_package(R):
    src = open(os.path.join(REF_PKG, "richdem", "__init__.py")).read()
    used = set(re.findall(r"_richdem\.([A-Za-z_0-9]+)", src))
    assert used, "no _richdem uses found"
    assert not [n for n in sorted(used) if not hasattr(R, n)]
    assert used <= set(SURFACE)
    for n in set(re.findall(r"depression_hierarchy\.([A-Za-z_]+)", src)):
        assert hasattr(R.depression_hierarchy, n), n


@pytest.mark.parametrize("name,dt", [("float", np.float32), ("double", np.float64), ("int8_t", np.int8), ("int16_t", np.int16),
                                      ("int32_t", np.int32), ("int64_t", np.int64), ("uint8_t", np.uint8),
                                      ("uint16_t", np.uint16), ("uint32_t", np.uint32), ("uint64_t", np.uint64)])
def test_array2d_wraps_numpy_memory(R, name, dt):
    cls = getattr(R, "Array2D_" + name)
    a = (np.arange(12) % 7).astype(dt).reshape(3, 4)
    w = cls(a)
    assert (w.width(), w.height(), w.size(), w.empty()) == (4, 3, 12, False)
    assert repr(w) == f"<RichDEM array: type={name}, width=4, height=3, owned=0>"
    assert w(1, 2) == a[2, 1] and w(9) == a.flat[9]        # (x, y) and flat index, pywrapper.hpp:176-185
    a[2, 1] = 5
    assert w(1, 2) == 5                                       # zero copy
    w.setNoData(6)
    assert w.noData() == 6 and w.max() == 5 and w.min() == 0  # extremes skip NoData (Array2D.hpp:516-535)
    w.setNoData(3.0)
    assert w.noData() == 3
    w.geotransform = np.array([10, 2, 0, 50, 0, -2], dtype="float64")
    w.projection = "EPSG:32633"
    w.metadata = {"k": "v"}
    assert w.geotransform == [10, 2, 0, 50, 0, -2] and w.projection == "EPSG:32633" and w.metadata == {"k": "v"}
    c = w.copy()                                              # owning deep copy with the metadata
    a[0, 0] = 4
    assert c(0, 0) == 0 and w(0, 0) == 4 and "owned=1" in repr(c)
    assert c.noData() == 3 and c.projection == "EPSG:32633" and c.metadata == {"k": "v"}
    with pytest.raises(IndexError):
        w(4, 0)
    with pytest.raises(IndexError):
        w(12)
    fresh = cls(5, 2, 1)
    assert (fresh.width(), fresh.height(), fresh(4, 1)) == (5, 2, 1)
    assert cls().empty()


def test_wrapping_never_copies_and_keeps_the_array_alive(R):
    a = np.zeros((4, 6), np.float32)
    for bad in (a.astype(np.float64), a.T, a[:, ::2], [[1.0, 2.0]], np.zeros(5, np.float32), np.zeros((2, 2, 2), np.float32)):
        with pytest.raises(RuntimeError):
            R.Array2D_float(bad)
    with pytest.raises(RuntimeError):
        R.Array3D_float(np.zeros((3, 4, 8), np.float32))      # nine slots per cell

    def make():
        return R.Array2D_double(np.full((4, 4), 3.5))

    w = make()
    gc.collect()
    assert w(3, 3) == 3.5


def test_array3d(R):
    p = np.arange(2 * 3 * 9, dtype=np.float32).reshape(2, 3, 9)
    w = R.Array3D_float(p)
    assert (w.width(), w.height(), w.size()) == (3, 2, 6)       # cells, not slots (Array3D.hpp:168)
    assert w(2, 1, 4) == p[1, 2, 4] and w.getIN(4, 7) == p.reshape(6, 9)[4, 7]
    w.setNoData(-2)
    assert w.noData() == -2 and repr(w) == "<RichDEM 3D array: type=float, width=3, height=2, owned=0>"
    p[1, 2, 4] = -5
    assert w(2, 1, 4) == -5
    with pytest.raises(IndexError):
        w(0, 0, 9)


def test_out_of_scope_functions_raise(R):
    dem = R.Array2D_float(np.zeros((4, 4), np.float32))
    for name in ("rdBreachDepressionsD8", "TA_slope_degrees", "FA_Rho8", "FM_Rho4", "generate_perlin_terrain"):
        with pytest.raises(RuntimeError, match="outside the scope"):
            getattr(R, name)(dem)
    with pytest.raises(RuntimeError, match="outside the scope"):
        R.depression_hierarchy.get_depression_hierarchy(dem, dem)
    with pytest.raises(RuntimeError, match="only available for floating-point"):   # Barnes2014.hpp:424-451
        R.rdPFepsilonD8(R.Array2D_int32_t(np.zeros((3, 3), np.int32)))
    with pytest.raises(RuntimeError, match="element type not supported"):   # bound for every type, as in the reference
        R.FA_Quinn(R.Array2D_int64_t(np.zeros((3, 3), np.int64)), R.Array2D_double(np.ones((3, 3))))
    with pytest.raises(RuntimeError, match="same dimensions"):              # flow_accumulation_generic.hpp:42-43
        R.FA_D8(dem, R.Array2D_double(np.ones((4, 5))))
    with pytest.raises(TypeError):                                            # accumulation is Array2D<double> only
        R.FA_D8(dem, R.Array2D_float(np.ones((4, 4), np.float32)))


@pytest.mark.skipif(not os.path.isdir(REF_PKG), reason="the reference tree is only present in the build container")
def test_the_reference_package_runs_on_it():
    """`import richdem` of the reference's own Python package binds to the GPU module after compat.install(); its
    drivers get as far as the engine (which needs a GPU: here the HIP error, or a result on a GPU box)."""
    code = textwrap.dedent(f"""
        import sys
        import numpy as np
        import richdem_amd.compat as compat
        m = compat.install()
        sys.path.insert(0, {REF_PKG!r})
        import richdem as rd
        import _richdem
        assert _richdem is m and rd._richdem is m
        rd._RichDEMVersion = lambda: "RichDEM (not installed as a distribution)"
        dem = rd.rdarray(np.random.default_rng(0).random((20, 30)).astype(np.float32), no_data=-9999,
                         geotransform=[0, 1, 0, 0, 0, -1])
        w = dem.wrap()
        assert (w.width(), w.height(), w.noData()) == (30, 20, -9999)
        reached = 0
        for fn, kw in ((rd.FillDepressions, {{}}), (rd.FlowAccumulation, {{"method": "D8"}}),
                       (rd.FlowAccumulation, {{"method": "Holmgren", "exponent": 2.0}}),
                       (rd.FlowProportions, {{"method": "Quinn"}}), (rd.ResolveFlats, {{}}),
                       (rd.FillDepressions, {{"epsilon": True}})):
            try:
                out = fn(dem, **kw)
                assert out.shape[:2] == dem.shape
                reached += 1
            except RuntimeError as e:
                assert "hip" in str(e).lower() or "rocm" in str(e).lower(), e
                reached += 1
        assert reached == 6
        for fn, kw in ((rd.BreachDepressions, {{}}),
                       (rd.TerrainAttribute, {{"attrib": "slope_degrees"}})):
            try:
                fn(dem, **kw)
                raise SystemExit("should have raised")
            except RuntimeError as e:
                assert "outside the scope" in str(e), e
        print("ok")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
