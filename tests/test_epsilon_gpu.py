"""PriorityFloodEpsilon_Barnes2014<topo> (depressions/Barnes2014.hpp:335-420) on the GPU, through the C-ABI.

Bit-for-bit (`==` on every cell) against the compiled reference and the C restatement on DEMs without equal
elevations; with equal elevations the reference's own output depends on the order in which std::priority_queue
returns them -- there the GPU surface (the unique fixed point) must be a lower bound of the reference, and every
mismatching case must really contain a tie (detected, not assumed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ND = -9999.0


def _ulp_dem(rng, h, w, base=100.0, spread=2, dtype=np.float32):
    """distinct values a few representable steps apart: every k*epsilon coincidence the reference can meet"""
    it = np.uint32 if dtype == np.float32 else np.uint64
    b = np.array(base, dtype).view(it)
    ks = rng.permutation(h * w * spread)[: h * w].astype(it)
    return (b + ks).view(dtype).reshape(h, w)


def _cases(rng):
    out = []
    for i in range(24):
        h, w = (int(v) for v in rng.integers(3, 150, 2))
        out.append((f"rand{i}", (rng.random((h, w)) * 100).astype(np.float32)))
    for i in range(8):
        h, w = (int(v) for v in rng.integers(3, 120, 2))
        out.append((f"ulp{i}", _ulp_dem(rng, h, w, spread=1 + i % 3)))
        out.append((f"ulpneg{i}", -_ulp_dem(rng, h, w, spread=1 + i % 3)))
    out.append(("around_zero", (_ulp_dem(rng, 40, 50, base=0.0, spread=1).astype(np.float32) - np.float32(1.4e-42))))
    out.append(("ragged", (rng.random((65, 33)) * 10).astype(np.float32)))
    out.append(("tile_edges", (rng.random((64, 128)) * 10).astype(np.float32)))
    out.append(("thin", (rng.random((3, 200)) * 10).astype(np.float32)))
    out.append(("tiny", (rng.random((1, 1))).astype(np.float32)))
    out.append(("two_rows", (rng.random((2, 9))).astype(np.float32)))
    return out


def _unique(z):
    return np.unique(z).size == z.size


@pytest.mark.parametrize("topo", [8, 4])
def test_epsilon_equals_reference_on_tie_free_dems(rd, orc, topo):
    rng = np.random.default_rng(7)
    name_topo = "D8" if topo == 8 else "D4"
    for name, z in _cases(rng):
        if not _unique(z):
            continue
        got = rd.FillDepressions(z, epsilon=True, topology=name_topo, nodata=ND)
        assert rd.epsilon_stats()["tie_sources"] == 0, name      # no equal elevations: the detector stays silent
        exp = orc.port.fill_epsilon(z, ND, topo)
        assert np.array_equal(got, exp), (name, topo, int((got != exp).sum()))
        if orc.ref.available:
            assert np.array_equal(got, orc.ref.fill_epsilon(z, ND, topo)), (name, topo)


def test_epsilon_float64(rd, orc):
    rng = np.random.default_rng(11)
    for i in range(10):
        h, w = (int(v) for v in rng.integers(3, 130, 2))
        z = rng.random((h, w)) * 1000 if i % 2 else _ulp_dem(rng, h, w, base=1000.0, spread=2, dtype=np.float64)
        for topo, nm in ((8, "D8"), (4, "D4")):
            got = rd.FillDepressions(z, epsilon=True, topology=nm, nodata=ND)
            assert got.dtype == np.float64 and np.array_equal(got, orc.port.fill_epsilon(z, ND, topo)), (i, topo)


def test_epsilon_nodata_regions_on_the_border(rd, orc):
    """NoData connected to the raster border: processed first by the reference, never altered, its data neighbours keep
    their elevation (Barnes2014.hpp:401-402)."""
    rng = np.random.default_rng(3)
    for i in range(12):
        h, w = (int(v) for v in rng.integers(6, 120, 2))
        z = (rng.random((h, w)) * 100).astype(np.float32)
        m = np.zeros((h, w), bool)
        m[: rng.integers(1, h // 2 + 1), : rng.integers(1, w // 2 + 1)] = True
        if i % 2:
            m[-1, :] = True
        if i % 3 == 0:
            m[:, -2:] = True
        z[m] = ND
        for topo, nm in ((8, "D8"), (4, "D4")):
            got = rd.FillDepressions(z, epsilon=True, topology=nm, nodata=ND)
            exp = orc.port.fill_epsilon(z, ND, topo)
            assert np.array_equal(got, exp), (i, topo, int((got != exp).sum()))
            assert (got[m] == ND).all()


def _eps_model(z, nd, topo, orc):
    """numpy restatement of the engine's DEFINITION (csrc/epsilon.hip): E = z on the raster border and on NoData that is
    connected to it; an interior NoData hole passes the level on (floor -infinity, one step per cell, never written);
    everything else E(c) = max(z(c), nextafter(min over neighbours E)).  Relaxed from above to the unique fixed point."""
    h, w = z.shape
    isnd = z == nd
    fixed = np.zeros((h, w), bool)
    fixed[0, :] = fixed[-1, :] = fixed[:, 0] = fixed[:, -1] = True
    fixed |= isnd & (orc.port.fill(z, topo) == nd)          # the plain fill leaves border-connected NoData at NoData
    floor = np.where(isnd & ~fixed, -np.inf, z).astype(z.dtype)
    D = np.full((h, w), np.inf, z.dtype)
    D[fixed] = z[fixed]
    inf = z.dtype.type(np.inf)
    offs = [(dy, dx) for dy in (0, 1, 2) for dx in (0, 1, 2) if (dy, dx) != (1, 1) and (topo == 8 or dy == 1 or dx == 1)]
    while True:
        P = np.pad(D, 1, constant_values=np.inf)
        nb = np.full((h, w), np.inf, z.dtype)
        for dy, dx in offs:
            nb = np.minimum(nb, P[dy:dy + h, dx:dx + w])
        new = np.where(fixed, D, np.minimum(D, np.maximum(floor, np.nextafter(nb, inf))))
        if np.array_equal(new, D):
            break
        D = new
    out = D.copy()
    out[isnd] = nd
    return out, isnd & ~fixed


def test_epsilon_interior_nodata_hole_inside_a_lake(rd, orc):
    """A NoData hole in the middle of a lake.  The reference floods the lake, hands the flood on through the hole, and
    leaves at their own elevation those ring cells that a hole cell happens to close before the breadth-first front does
    (queue order).  The engine fills the lake too (the level passes through the hole) and raises every ring cell:
    bit-exact against the numpy restatement of its definition, equal to the compiled reference everywhere except around
    the hole, and nowhere near the hole-as-a-drain surface a fixed NoData cell would give."""
    rng = np.random.default_rng(21)
    ref = orc.ref if orc.ref.available else orc.port
    nd = np.float32(ND)
    for trial in range(8):
        h, w = 40 + 3 * trial, 46
        z = (rng.permutation(h * w).reshape(h, w) * 0.01 + 60).astype(np.float32)      # no equal elevations
        z[6:h - 6, 6:w - 6] -= 45                                                        # a deep lake inside a rim
        hy, hx, hs = 14 + trial, 18, 2 + trial % 3
        z[hy:hy + hs, hx:hx + hs] = nd                                                   # the hole
        if trial % 2:
            z[:3, 25:] = nd                                                              # and NoData on the border
        for topo, nm in ((8, "D8"), (4, "D4")):
            got = rd.FillDepressions(z, epsilon=True, topology=nm, nodata=ND)
            model, hole = _eps_model(z, nd, topo, orc)
            assert hole.sum() == hs * hs
            assert np.array_equal(got, model), (trial, topo, int((got != model).sum()))
            exp = ref.fill_epsilon(z, nd, topo)
            diff = got != exp
            # differences: the ring cells of the hole (raised here, some left at their own elevation by the reference), and
            # behind those a shadow where the reference's breadth-first front had to walk around them: a few steps apart
            ring = np.zeros((h, w), bool)
            ring[hy - 1:hy + hs + 1, hx - 1:hx + hs + 1] = True
            steps = np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64))
            near = steps <= 2 * hs + 2
            assert (near | (ring & (got >= exp)))[diff].all(), (trial, topo)
            assert diff.sum() <= 0.1 * h * w
            lake = np.zeros((h, w), bool)
            lake[6:h - 6, 6:w - 6] = True
            lake &= z != nd
            assert (got[lake] > 50).all()                       # the lake is filled (a fixed hole would leave it near 15)


def test_epsilon_with_ties_is_a_lower_bound_and_ties_are_detected(rd, orc):
    """Integer-valued and plateau DEMs: the reference depends on its heap's pop order.  Every mismatching case is
    checked to contain equal elevations among the cells that are NOT raised (the cells of the reference's heap), and the
    GPU surface -- the fixed point -- never lies above the reference."""
    rng = np.random.default_rng(5)
    ref = orc.ref if orc.ref.available else orc.port
    n_equal = n_diff = 0
    for i in range(30):
        h, w = (int(v) for v in rng.integers(4, 90, 2))
        z = rng.integers(0, 6 + i, (h, w)).astype(np.float32)
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = rd.FillDepressions(z, epsilon=True, nodata=ND)
        ties = rd.epsilon_stats()["tie_sources"]
        exp = ref.fill_epsilon(z, ND, 8)
        assert (got <= exp).all() and (got >= z).all()
        if np.array_equal(got, exp):
            n_equal += 1
        else:
            n_diff += 1
            assert ties > 0, "a mismatch the device-side tie detector did not announce"
            assert any("gradient sources share their elevation" in str(c.message) for c in caught)
            unraised = z[got == z]
            assert np.unique(unraised).size < unraised.size, "a mismatch without a tie among the heap's cells"
        # whatever the ties, the surface is the fixed point: every interior cell equals max(z, nextafter(min neighbour))
        e = got.astype(np.float32)
        if h > 2 and w > 2:
            nb = np.full((h - 2, w - 2), np.inf, np.float32)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dy or dx:
                        nb = np.minimum(nb, e[1 + dy : h - 1 + dy, 1 + dx : w - 1 + dx])
            want = np.maximum(z[1:-1, 1:-1], np.nextafter(nb, np.float32(np.inf)))
            assert np.array_equal(e[1:-1, 1:-1], want)
    assert n_equal + n_diff == 30


def test_epsilon_drains_everything(rd):
    """The point of the function: afterwards every interior cell has a strictly lower neighbour (no pits, no flats),
    at a size well past one tile, with the slack retry path exercised by a tiny RDGPU_EPS_SLACK."""
    import os

    from richdem_amd.synth import fractal_dem

    z = fractal_dem(1500, 1100, seed=4)
    e = rd.FillDepressions(z, epsilon=True, nodata=ND)
    st = rd.epsilon_stats()
    assert st["attempts"] >= 1 and st["max_lift"] > 0
    nb = np.full((1098, 1498), np.inf, np.float32)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy or dx:
                nb = np.minimum(nb, e[1 + dy : 1099 + dy, 1 + dx : 1499 + dx])
    assert (nb < e[1:-1, 1:-1]).all()
    os.environ["RDGPU_EPS_SLACK"] = "1"          # far too small: the proof must fail and the retry must fix it
    try:
        e2 = rd.FillDepressions(z, epsilon=True, nodata=ND)
        assert rd.epsilon_stats()["attempts"] > 1
    finally:
        del os.environ["RDGPU_EPS_SLACK"]
    assert np.array_equal(e, e2)


def test_epsilon_device_entry_and_errors(rd):
    import torch

    from richdem_amd.synth import fractal_dem

    z = fractal_dem(300, 200, seed=9)
    t = torch.from_numpy(z).cuda()
    rd.fill_epsilon_dev(t, ND)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), rd.FillDepressions(z, epsilon=True, nodata=ND))
    with pytest.raises(rd.RdgpuError, match="floating-point"):
        rd.FillDepressions(np.zeros((4, 4), np.int16), epsilon=True)
    a = rd.rdarray(z.copy(), no_data=ND)
    out = rd.FillDepressions(a, epsilon=True)
    assert "FillDepressions(dem, epsilon=True)" in out.metadata["PROCESSING_HISTORY"] and np.array_equal(np.asarray(out), t.cpu().numpy())


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_epsilon_behind_walls_of_infinity(rd, orc, dt):
    """nextafter(+inf) == +inf: cells whose only way out leads over cells of elevation +inf end at +inf however many steps
    they are from the wall (the relaxation counts its steps unsaturated and saturates when it stores, csrc/epsilon.hip);
    the largest finite values step up to +inf too."""
    rng = np.random.default_rng(5)
    h, w = 70, 90
    z = (rng.permutation(h * w).reshape(h, w) * 0.25 + 10).astype(dt)
    z[10:40, 20] = z[10:40, 60] = z[10, 20:61] = z[39, 20:61] = np.inf          # a walled yard, 28 x 39 cells inside
    z[50:60, 5:15] = np.finfo(dt).max                                           # a plateau of the largest finite value ...
    z[53:57, 8:12] = 1.0                                                        # ... around a pit
    got = rd.FillDepressions(z, epsilon=True, nodata=dt(-9999))
    exp = orc.port.fill_epsilon(z, dt(-9999), 8)
    assert np.array_equal(got, exp), int((got != exp).sum())
    assert np.isinf(got[11:39, 21:60]).all() and np.isinf(got[53:57, 8:12]).all()
    assert np.array_equal(got[0], z[0]) and np.array_equal(got[:, 0], z[:, 0])
