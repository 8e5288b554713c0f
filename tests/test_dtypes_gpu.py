"""Element types the reference binds (wrappers/pyrichdem/src/pywrapper.cpp:25-45) beyond the 8/16/32-bit unsigned and
float ones: int8 through every entry point, int64 / uint64 through the entry points that compare elevations in their
own type (d8_flow_directions, flat resolution, ResolveFlatsEpsilon, FA_D8) and the fill (value ranks)."""
import numpy as np
import pytest

from richdem_amd.synth import fractal_dem

pytestmark = pytest.mark.gpu


def _dem(dtype, seed=3, shape=(150, 130), scale=0.05):
    z = fractal_dem(shape[1], shape[0], seed=seed)
    q = np.floor((z - z.mean()) * scale)
    if dtype == np.int8:
        return np.clip(q, -120, 120).astype(np.int8), np.int8(-128)
    if dtype == np.int64:
        return (q.astype(np.int64) * 3_000_000_007), np.int64(-(2 ** 62))
    return ((q - q.min()).astype(np.uint64) * np.uint64(5_000_000_011)), np.uint64(2 ** 63)


@pytest.mark.parametrize("dtype", [np.int8, np.int64, np.uint64])
def test_stencil_chain_all_element_types(rd, orc, dtype):
    dem, nd = _dem(dtype)
    filled = rd.FillDepressions(dem)
    assert filled.dtype == dem.dtype and filled.tobytes() == orc.port.fill(dem, 8).tobytes()
    assert rd.FillDepressions(dem, topology="D4").tobytes() == orc.port.fill(dem, 4).tobytes()
    for src in (dem, filled):
        assert np.array_equal(rd.d8_flow_directions(src, nd), orc.port.d8_flowdirs(src, nd))
        dirs, mask, _ = rd.resolve_flats(src, nd)
        odirs, omask, _ = orc.port.resolve_flats(src, nd)
        fr = rd.barnes_flat_resolution_d8(src, nd)
        assert np.array_equal(mask, omask) and np.array_equal(fr, orc.port.flat_resolution(src, nd))
        assert np.array_equal(rd.FlowAccumulation(src, "D8", nodata=nd), orc.port.fa_d8(src, nd))
        assert np.array_equal(rd.FlowProportions(src, "D8", nodata=nd), orc.port.fm_d8(src, nd))
        got = rd.resolve_flats_epsilon(src, nd)
        assert got.dtype == dem.dtype and got.tobytes() == orc.port.resolve_flats_epsilon(src, nd).tobytes()


def test_int8_everything_else(rd, orc):
    dem, nd = _dem(np.int8, seed=5)
    assert np.array_equal(rd.pit_mask(dem, nd), orc.port.pit_mask(dem, nd, 8))
    assert np.array_equal(rd.fill_max_dep(dem, 10 ** 7), orc.port.fill(dem, 8))
    assert np.array_equal(rd.fill_max_dep(dem, 0), dem)
    lab = rd.watersheds(dem, nd)
    assert lab.shape == dem.shape and lab.min() >= 1
    assert rd.FillDepressions(dem, shards=4).tobytes() == orc.port.fill(dem, 8).tobytes()
    for method, x in (("Quinn", None), ("Holmgren", 2.0), ("D4", None), ("Dinf", None)):
        got = rd.FlowProportions(dem, method, nodata=nd, exponent=x)
        exp = orc.port.fm_tarboton(dem, nd) if method == "Dinf" else orc.port.fm_mfd(dem, nd, method, x or 1.0)
        assert np.allclose(got, exp, rtol=2e-7, atol=0), method
    with pytest.raises(rd.RdgpuError):
        rd.FlowProportions(_dem(np.int64)[0], "Quinn")
    with pytest.raises(rd.RdgpuError, match="floating-point"):
        rd.FillDepressions(dem, epsilon=True)
