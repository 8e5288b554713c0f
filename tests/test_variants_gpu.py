"""HasDepressions / PriorityFlood_Original / PriorityFlood_Wei2018 on the GPU engine (csrc/variants.hip) against the
compiled reference's vectors (tests/golden/ref_variants.npz), the C restatement and, where it travelled, the live compiled
reference -- reference tests/tests.cpp:233-271 runs the same three against testdem1."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_variants.npz")


def test_reference_vectors(rd):
    g = np.load(GOLD)
    for n in sorted({k.split("/")[0] for k in g.files}):
        dem, nd = g[f"{n}/dem"], g[f"{n}/nodata"].item()
        assert np.array_equal(rd.fill_wei2018(dem, nd), g[f"{n}/wei2018"]), n
        for topo in ("D8", "D4"):
            assert rd.has_depressions(dem, topo) == bool(g[f"{n}/has_depressions_d{topo[1]}"]), (n, topo)
            assert np.array_equal(rd.FillDepressions(dem, topology=topo), g[f"{n}/original_d{topo[1]}"]), (n, topo)


def test_reference_golden_testdem1(rd, fixtures):
    dem, exp = fixtures["fill/testdem1/dem"], fixtures["fill/testdem1/all_out"]   # tests.cpp:259-262: Wei2018 == all.out
    assert np.array_equal(rd.fill_wei2018(dem, int(fixtures["fill/testdem1/nodata"])), exp)
    assert rd.has_depressions(dem) == bool((exp != dem).any())


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64, np.int64,
                                   np.uint64])
def test_every_element_type_with_nodata_holes(rd, orc, dtype):
    rng = np.random.default_rng(7)
    from richdem_amd.synth import fractal_dem

    z = fractal_dem(201, 167, 71)
    z = (z - z.min()) / (z.max() - z.min())
    if np.issubdtype(dtype, np.floating):
        dem = (z * 900).astype(dtype)
        if dtype == np.float64:
            dem += 1e-9 * np.arange(dem.size).reshape(dem.shape)   # not representable in f32: the rank path
        nd = dtype(-9999)
    else:
        top = min(np.iinfo(dtype).max, 5000)
        dem = (1 + z * (top - 2)).astype(dtype)
        if dtype in (np.int64, np.uint64):
            dem = dem * dtype(1 << 33)
        nd = dtype(0)
    dem[rng.random(dem.shape) < 0.004] = nd
    dem[60:75, 90:130] = nd
    exp = orc.port.fill_wei2018(dem, nd)
    got = rd.fill_wei2018(dem, nd)
    assert np.array_equal(got, exp)
    assert (got != orc.port.fill(dem, 8)).any()            # the holes do drain something
    assert np.array_equal(got == nd, dem == nd)            # NoData cells stay NoData, nothing becomes NoData
    if orc.ref.available:
        assert np.array_equal(orc.ref.fill_wei2018(dem, nd), exp)
    for topo, t in (("D8", 8), ("D4", 4)):
        assert rd.has_depressions(dem, topo) == orc.port.has_depressions(dem, t)


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (9, 1), (2, 2), (3, 3), (64, 64), (65, 63), (130, 257)])
def test_shapes_around_the_tiles(rd, orc, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    dem = rng.integers(1, 9, shape).astype(np.int32)
    dem[rng.random(shape) < 0.05] = -1
    assert np.array_equal(rd.fill_wei2018(dem, -1), orc.port.fill_wei2018(dem, -1))
    for topo, t in (("D8", 8), ("D4", 4)):
        assert rd.has_depressions(dem, topo) == orc.port.has_depressions(dem, t)


def test_has_depressions_is_false_on_filled_and_flat_rasters(rd):
    from richdem_amd.synth import fractal_dem

    dem = fractal_dem(300, 260, 5)
    assert rd.has_depressions(dem)
    assert not rd.has_depressions(rd.FillDepressions(dem))
    assert rd.has_depressions(rd.FillDepressions(dem, topology="D8"), "D4") in (True, False)   # (D4 may still find some)
    assert not rd.has_depressions(rd.FillDepressions(dem, topology="D4"), "D4")
    assert not rd.has_depressions(np.zeros((40, 50), np.float32))
    assert not rd.has_depressions(np.empty((0, 5), np.float32))


def test_wei2018_without_nodata_is_the_plain_fill(rd):
    from richdem_amd.synth import fractal_dem

    dem = fractal_dem(500, 400, 9)
    assert np.array_equal(rd.fill_wei2018(dem, -9999), rd.FillDepressions(dem))   # tests/wei2018-test/main.cpp:55-77


def test_s2_window_against_the_compiled_reference(rd, orc):
    if not orc.ref.available:
        pytest.skip("oracle/_ref/libref.so did not travel")
    from richdem_amd.synth import fractal_dem

    dem = fractal_dem(2000, 1500, 2).copy()
    rng = np.random.default_rng(1)
    for _ in range(40):   # lakes of NoData inside the raster
        y, x = int(rng.integers(0, 1450)), int(rng.integers(0, 1950))
        dem[y:y + int(rng.integers(1, 50)), x:x + int(rng.integers(1, 50))] = -9999
    assert np.array_equal(rd.fill_wei2018(dem, -9999), orc.ref.fill_wei2018(dem, np.float32(-9999)))
    assert rd.has_depressions(dem) == orc.ref.has_depressions(dem, 8)
