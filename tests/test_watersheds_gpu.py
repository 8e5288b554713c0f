"""PriorityFloodWatersheds_Barnes2014<topo> (depressions/Barnes2014.hpp:713-807) on the GPU, through the C-ABI: labels
AND their numbering equal the compiled reference / the C restatement on DEMs without equal elevations; with ties the
partition is compared after canonical relabelling and mismatches are counted."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ND = -9999.0


@pytest.fixture(scope="module")
def f2():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_f2.npz"))


def canon(lab):
    """labels renumbered in order of first appearance (-1 kept): equal partitions give equal arrays"""
    flat = lab.ravel()
    _, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))
    out = order[inv].astype(np.int64)
    out[flat == -1] = -1
    return out.reshape(lab.shape)


def test_watersheds_equal_the_compiled_reference_outputs(rd, f2):
    names = sorted({k.split("/")[0] for k in f2.files if not k.startswith("max_dep/")})
    for name in names:
        dem = f2[f"{name}/dem"]
        if dem.dtype != np.float32:
            continue
        for topo, nm in ((8, "D8"), (4, "D4")):
            lab = rd.watersheds(dem, ND, nm)
            assert lab.dtype == np.int32 and np.array_equal(lab, f2[f"{name}/watersheds_d{topo}"]), (name, topo)


def test_watersheds_random_tie_free(rd, orc):
    rng = np.random.default_rng(19)
    for i in range(30):
        h, w = (int(v) for v in rng.integers(3, 150, 2))
        z = (rng.random((h, w)) * 100).astype(np.float32)
        if np.unique(z).size != z.size:
            continue
        if i % 3 == 0 and h > 6 and w > 6:      # a NoData region on the border and a NoData hole inside
            z[: h // 3, : w // 4] = ND
            z[h // 2 : h // 2 + 2, w // 2 : w // 2 + 3] = ND
        for topo, nm in ((8, "D8"), (4, "D4")):
            exp, filled = orc.port.watersheds(z, ND, topo, True)
            got, gfill = rd.watersheds(z, ND, nm, alter=True)
            assert np.array_equal(got, exp), (i, topo, int((got != exp).sum()))
            assert np.array_equal(gfill, filled)


def test_watersheds_larger_and_integer(rd, orc):
    from richdem_amd.synth import fractal_dem

    z = fractal_dem(900, 700, seed=8)
    zz = z + (np.arange(z.size, dtype=np.float32).reshape(z.shape) * np.float32(1e-3))
    if np.unique(zz).size == zz.size:
        assert np.array_equal(rd.watersheds(zz, ND), orc.port.watersheds(zz, ND, 8)[0])
    # integer DEMs (ties everywhere): a valid partition -- every label class is connected to the border cell that starts
    # it -- and the count of mismatches against the C restatement after canonical relabelling is reported
    rng = np.random.default_rng(2)
    differ = 0
    for i in range(10):
        q = rng.integers(0, 30, (50, 60)).astype(np.int32)
        got = rd.watersheds(q, -1)
        exp = orc.port.watersheds(q, -1, 8)[0]
        assert got.min() >= 1 and got.max() <= 2 * (50 + 60)
        # numbering: one label per border cell at most, all labels used
        assert np.unique(got).size == got.max()
        differ += not np.array_equal(canon(got), canon(exp))
    print(f"watersheds on integer DEMs: {differ} of 10 partitions differ from the C restatement (ties)")


def test_watersheds_64bit_element_types(rd, orc):
    """f64 / i64 / u64 through the dense value ranks: labels, numbering and the altered DEM equal the C restatement on
    tie-free DEMs, NoData regions and holes included."""
    rng = np.random.default_rng(23)
    for dt in (np.float64, np.int64, np.uint64):
        h, w = 70, 85
        if dt is np.float64:
            z = (rng.permutation(h * w).reshape(h, w) * 1e-3 + 5.0) * (1 + 2.0 ** -40)
            nd = dt(-9999)
        else:
            z = (rng.permutation(h * w).reshape(h, w).astype(np.int64) << 34).astype(dt) + dt(1 << 40)
            nd = dt(5)
        z[: h // 3, : w // 4] = nd
        z[h // 2 : h // 2 + 2, w // 2 : w // 2 + 3] = nd
        for topo, nm in ((8, "D8"), (4, "D4")):
            exp, filled = orc.port.watersheds(z, nd, topo, True)
            got, gfill = rd.watersheds(z, nd, nm, alter=True)
            assert np.array_equal(got, exp), (dt, topo, int((got != exp).sum()))
            assert gfill.dtype == z.dtype and np.array_equal(gfill, filled)
