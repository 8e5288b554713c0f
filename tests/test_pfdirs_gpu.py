"""PriorityFloodFlowdirs_Barnes2014 (depressions/Barnes2014.hpp:483-555) on the GPU: one fill per nesting level of the
depressions (csrc/pfdirs.hip) against the oracle's restatement of the reference's stable-queue sweep, which is pinned to
the compiled reference (tests/test_oracle_pinning.py).  Equal on DEMs without equal elevations; with ties the cells that
stay ambiguous are counted."""
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN

from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu


def _distinct(dem, rng):
    out = dem.astype(np.float32).copy()
    flat = out.ravel()
    for _ in range(60):
        _, first = np.unique(flat, return_index=True)
        dup = np.setdiff1d(np.arange(flat.size), first)
        if dup.size == 0:
            return out
        flat[dup] = np.nextafter(flat[dup], np.float32(np.inf)) + rng.random(dup.size).astype(np.float32) * np.float32(1e-3)
    raise AssertionError("could not make the DEM tie free")


@pytest.mark.parametrize("shape", [(3, 3), (4, 7), (9, 9), (33, 65), (64, 64), (130, 97), (257, 300)])
def test_random_permutations(rd, orc, shape):
    """every elevation distinct, no structure at all: deep nesting on small rasters"""
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    dem = rng.permutation(h * w).reshape(h, w).astype(np.int32)
    got = rd.pf_flowdirs(dem, nodata=np.int32(-9999))
    assert np.array_equal(got, orc.port.pf_flowdirs(dem, np.int32(-9999)))
    assert rd.pf_flowdirs_stats()["unresolved"] == 0 and rd.pf_flowdirs_stats()["twins"] == 0


@pytest.mark.parametrize("seed,shape", [(1, (200, 260)), (2, (333, 190)), (3, (512, 512))])
def test_fractal_terrain_without_ties(rd, orc, seed, shape):
    rng = np.random.default_rng(seed)
    dem = _distinct(fractal_dem(shape[1], shape[0], seed=40 + seed), rng)
    got = rd.pf_flowdirs(dem, nodata=np.float32(-9999))
    exp = orc.port.pf_flowdirs(dem, np.float32(-9999))
    assert np.array_equal(got, exp), int((got != exp).sum())
    st = rd.pf_flowdirs_stats()
    assert st["unresolved"] == 0 and st["levels"] >= 3


def test_rank_permutation_of_a_3000_square_terrain(rd, orc):
    """Many 64 x 64 tiles, lakes hundreds of cells across, every elevation distinct (the terrain's ranks): 9 million cells
    against the stable-queue sweep, a few hundred nesting levels."""
    n = 3000
    z = fractal_dem(n, n, seed=3).astype(np.float64)
    rng = np.random.default_rng(12)
    order = np.argsort(z.ravel() + rng.random(n * n) * 1e-9, kind="stable")
    ranks = np.empty(n * n, np.int32)
    ranks[order] = np.arange(n * n, dtype=np.int32)
    dem = ranks.reshape(n, n)
    got = rd.pf_flowdirs(dem, nodata=np.int32(-9999))
    exp = orc.port.pf_flowdirs(dem, np.int32(-9999))
    st = rd.pf_flowdirs_stats()
    assert np.array_equal(got, exp), int((got != exp).sum())
    assert st["twins"] == 0 and st["unresolved"] == 0 and st["levels"] > 30


def test_nodata_cells_and_other_dtypes(rd, orc):
    rng = np.random.default_rng(7)
    dem = rng.permutation(90 * 70).reshape(70, 90).astype(np.float32)
    dem[rng.random(dem.shape) < 0.03] = -9999.0          # NoData cells flood like any cell and get direction 0 (:545-548)
    nd = np.float32(-9999)
    holes = dem == nd
    dem[holes] = -9999.0 - np.arange(holes.sum(), dtype=np.float32)   # (distinct values below the data ...)
    exp = orc.port.pf_flowdirs(dem, np.float32(-9999))
    assert np.array_equal(rd.pf_flowdirs(dem, nodata=nd), exp)
    for dt in (np.uint16, np.int16, np.uint32):
        d = rng.permutation(60 * 50).reshape(50, 60).astype(dt)
        assert np.array_equal(rd.pf_flowdirs(d, nodata=dt(0)), orc.port.pf_flowdirs(d, dt(0))), dt
    for dt in (np.float64, np.int64, np.uint64):          # the 64-bit types run on the dense value ranks
        d = (rng.permutation(64 * 45).reshape(45, 64).astype(np.float64) * 1e12 + (1 << 60 if dt == np.uint64 else 0)).astype(dt)
        assert np.unique(d).size == d.size
        assert np.array_equal(rd.pf_flowdirs(d, nodata=dt(0)), orc.port.pf_flowdirs(d, dt(0))), dt
    small = rng.permutation(200).reshape(10, 20).astype(np.uint8)
    assert np.array_equal(rd.pf_flowdirs(small, nodata=np.uint8(255)), orc.port.pf_flowdirs(small, np.uint8(255)))


def test_equal_elevations_equal_the_reference(rd, orc, monkeypatch):
    """Equal elevations: the reference's stable queue (GridCellZk_low_pq, common/grid_cell.hpp:101-122) pops them in order of
    insertion, so its directions are a function of the DEM -- and the engine's are THE SAME: the tie order is found as the
    fixed point of "discovery time under the flood of (elevation, discovery time)" (csrc/pfdirs.hip).  The reference's own
    tie-heavy vectors (tests/golden/ref_pf_flowdirs_ties.npz, compiled reference), random few-level rasters, a flat raster, a
    quantised terrain with large plateaus, NoData regions."""
    t = np.load(os.path.join(GOLDEN, "ref_pf_flowdirs_ties.npz"))
    for name in ("ties_i32", "ties_nodata_f32"):
        dem = t[f"{name}/dem"]
        got = rd.pf_flowdirs(dem, nodata=dem.dtype.type(-9999))
        st = rd.pf_flowdirs_stats()
        assert st["unresolved"] == 0 and st["twins"] > 0 and st["tie_passes"] >= 1, st
        assert np.array_equal(got, t[f"{name}/pf_flowdirs"]), (name, int((got != t[f"{name}/pf_flowdirs"]).sum()), st)
    rng = np.random.default_rng(9)
    cases = {"6 levels": rng.integers(0, 6, (80, 100)).astype(np.int32),
             "2 levels": rng.integers(0, 2, (50, 70)).astype(np.uint8),
             "flat": np.zeros((33, 47), np.int16),
             "plateaus": fractal_dem_int(300, 260, 32, 0.05),
             "terrain, integer": fractal_dem_int(400, 300, 31, 1.0),
             "float32 terrain (local ties)": fractal_dem(700, 500, seed=3)}
    holes = rng.integers(0, 40, (90, 120)).astype(np.float32)
    holes[rng.random(holes.shape) < 0.06] = -9999.0
    holes[30:45, 50:80] = -9999.0
    holes[:, :4] = -9999.0
    cases["NoData regions"] = holes
    for name, dem in cases.items():
        nd = dem.dtype.type(-9999) if dem.dtype.kind in "if" and dem.dtype.itemsize > 1 else dem.dtype.type(255)
        got = rd.pf_flowdirs(dem, nodata=nd)
        st = rd.pf_flowdirs_stats()
        exp = orc.port.pf_flowdirs(dem, nd)
        assert st["unresolved"] == 0, (name, st)
        assert np.array_equal(got, exp), (name, int((got != exp).sum()), dem.size, st)
    # r05, the tree iteration is the default: one level flood, then (tree of directions, ranks) iterated to "every cell points at
    # its first-popped neighbour and the order reproduces itself", a level flood again once the ranks rest.  The same
    # directions from a level flood per pass (RDGPU_PFD_TREE_ITER=0, r04), from refloods at other moments, and without them
    monkeypatch.setenv("RDGPU_PFD_TREE_ITER", "0")
    for name, dem in cases.items():
        nd = dem.dtype.type(-9999) if dem.dtype.kind in "if" and dem.dtype.itemsize > 1 else dem.dtype.type(255)
        got = rd.pf_flowdirs(dem, nodata=nd)
        st = rd.pf_flowdirs_stats()
        assert st["unresolved"] == 0, (name, st)
        assert np.array_equal(got, orc.port.pf_flowdirs(dem, nd)), ("tree iteration", name, st)
    monkeypatch.delenv("RDGPU_PFD_TREE_ITER")
    for var, val in (("RDGPU_PFD_LEVEL_TREE", "0"),            # the record tree by the ancestor search also after an exact flood
                     ("RDGPU_PFD_TREE_REFLOOD", "3"),          # a fresh level flood every third iteration
                     ("RDGPU_PFD_TREE_REFLOOD_MOVED", "100000000"),   # ... after every iteration (the threshold always holds)
                     ("RDGPU_PFD_TREE_REFLOOD", "1000000")):   # ... only once the ranks rest: the tree alone iterated to its fixed point
        monkeypatch.setenv(var, val)
        if val == "1000000":
            monkeypatch.setenv("RDGPU_PFD_TREE_REFLOOD_MOVED", "0")
        for name in ("plateaus", "6 levels", "float32 terrain (local ties)"):
            dem = cases[name]
            got = rd.pf_flowdirs(dem, nodata=dem.dtype.type(-9999))
            st = rd.pf_flowdirs_stats()
            assert st["unresolved"] == 0 and np.array_equal(got, orc.port.pf_flowdirs(dem, dem.dtype.type(-9999))), (var, val, name, st)
        monkeypatch.delenv(var)
    monkeypatch.delenv("RDGPU_PFD_TREE_REFLOOD_MOVED")
    for shape in [(3, 3), (3, 4), (4, 4), (5, 3), (3, 9), (2, 7), (1, 5), (6, 2)]:      # the smallest rasters, all ties
        for dem in (np.zeros(shape, np.int32), rng.integers(0, 2, shape).astype(np.int32)):
            assert np.array_equal(rd.pf_flowdirs(dem, nodata=np.int32(-9999)), orc.port.pf_flowdirs(dem, np.int32(-9999))), (shape, dem)
    # the two earlier tie rules stay selectable, and say what they are
    dem = cases["6 levels"]
    exp = orc.port.pf_flowdirs(dem, np.int32(-9999))
    monkeypatch.setenv("RDGPU_PFD_TIE_PASSES", "0")          # equal cells in raster order: one exact flood of the unique ranks
    monkeypatch.setenv("RDGPU_PFD_TIE_INIT", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        raster_order = rd.pf_flowdirs(dem, nodata=np.int32(-9999))
    st0 = rd.pf_flowdirs_stats()
    assert st0["tie_passes"] == 0 and st0["unresolved"] == st0["twins"] > 0, st0    # no re-rank pass ran: every twin is undecided
    order = np.argsort(dem.ravel(), kind="stable")
    ranks = np.empty(dem.size, np.int32)
    ranks[order] = np.arange(dem.size, dtype=np.int32)
    assert np.array_equal(raster_order, orc.port.pf_flowdirs(ranks.reshape(dem.shape), np.int32(-9999)))
    monkeypatch.delenv("RDGPU_PFD_TIE_PASSES")
    assert np.array_equal(rd.pf_flowdirs(dem, nodata=np.int32(-9999)), exp)     # (any first order converges: here raster order)
    monkeypatch.delenv("RDGPU_PFD_TIE_INIT")
    # the passes are bounded (count and wall time): a stopped iteration says so -- stats and a RuntimeWarning
    flat = np.zeros((33, 47), np.int16)
    for var, val in (("RDGPU_PFD_TIE_PASSES", "2"), ("RDGPU_PFD_TIE_SECONDS", "0")):
        monkeypatch.setenv(var, val)
        with pytest.warns(RuntimeWarning, match="tie-order passes were stopped"):
            rd.pf_flowdirs(flat, nodata=np.int16(-9999))
        st1 = rd.pf_flowdirs_stats()
        assert 1 <= st1["tie_passes"] <= 2 and st1["unresolved"] > 0, (var, st1)
        monkeypatch.delenv(var)
    monkeypatch.setenv("RDGPU_PFD_RANKS", "0")               # r03: ties decided inside the levels, by neighbour number
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        late = rd.pf_flowdirs(dem, nodata=np.int32(-9999))
    assert rd.pf_flowdirs_stats()["unresolved"] > 0
    warnings.warn(f"pf_flowdirs, 6 elevations on 80 x 100: fixed point 0.000 of the cells differ from the reference, raster-order "
                  f"ties {float((raster_order != exp).mean()):.3f}, ties inside the levels {float((late != exp).mean()):.3f}")


def test_ties_with_nodata_cells(rd, orc):
    """NoData cells flood with their NoData value as elevation (two of them are already a tie) and come out with direction 0
    where they are interior cells: equal to the reference."""
    rng = np.random.default_rng(10)
    dem = (rng.integers(0, 50, (70, 90))).astype(np.float32)
    dem[rng.random(dem.shape) < 0.05] = -9999.0
    dem[20:30, 40:55] = -9999.0
    got = rd.pf_flowdirs(dem, nodata=np.float32(-9999))
    assert np.array_equal(got, orc.port.pf_flowdirs(dem, np.float32(-9999)))


def test_sparse_levels_give_the_same_directions(rd, orc, monkeypatch):
    """From the second level on the fills skip the 64 x 64 tiles that hold nothing but walls (RDGPU_PFD_SPARSE=0: every level
    over the whole raster): same directions, with and without equal elevations."""
    rng = np.random.default_rng(21)
    tie_free = _distinct(fractal_dem(1400, 1100, seed=77), rng)
    with_ties = np.floor(fractal_dem(1400, 1100, seed=78) * np.float32(2.0)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a1, b1 = rd.pf_flowdirs(tie_free, nodata=np.float32(-9999)), rd.pf_flowdirs(with_ties, nodata=np.float32(-9999))
        monkeypatch.setenv("RDGPU_PFD_SPARSE", "0")
        a0, b0 = rd.pf_flowdirs(tie_free, nodata=np.float32(-9999)), rd.pf_flowdirs(with_ties, nodata=np.float32(-9999))
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    assert np.array_equal(a1, orc.port.pf_flowdirs(tie_free, np.float32(-9999)))


def test_border_only_rasters_and_errors(rd, orc):
    for shape in [(1, 1), (1, 5), (2, 2), (2, 9)]:
        dem = np.arange(shape[0] * shape[1], dtype=np.float32).reshape(shape)
        assert np.array_equal(rd.pf_flowdirs(dem, nodata=np.float32(-1)), orc.port.pf_flowdirs(dem, np.float32(-1))), shape
    with pytest.raises(rd.RdgpuError):
        rd.pf_flowdirs(np.zeros((4, 4), np.complex64))


def test_the_types_highest_value_in_the_dem(rd, orc):
    """The levels' walls are the element type's highest value (+inf for floats); a DEM that holds that value itself -- 255 in
    a uint8 raster -- is flooded on its unique ranks, where no cell reaches the wall."""
    rng = np.random.default_rng(12)
    dem = (16 + rng.permutation(240)).reshape(15, 16).astype(np.uint8)     # distinct values 16..255: 255 is in the DEM
    assert dem.max() == 255 and np.unique(dem).size == dem.size
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rd.pf_flowdirs(dem, nodata=np.uint8(0))
    assert np.array_equal(got, orc.port.pf_flowdirs(dem, np.uint8(0)))      # tie free: the reference's answer
    f = np.full((40, 50), 5.0, np.float32) + rng.random((40, 50)).astype(np.float32)
    f[10:14, 20:24] = np.inf          # a mesa of +inf: a wall inside the DEM
    got = rd.pf_flowdirs(f, nodata=np.float32(-9999))
    assert np.array_equal(got, orc.port.pf_flowdirs(f, np.float32(-9999)))
