"""PriorityFlood_Barnes2014_max_dep<topo> (depressions/Barnes2014.hpp:844-931) on the GPU, through the C-ABI: the
reference's own goldens (tests/depressions/testdem1.{1,2}.out, tests/tests.cpp:273-287), the committed outputs of the
compiled reference on tie-free DEMs, and random DEMs against the C restatement."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def f2():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_f2.npz"))


def test_max_dep_reference_goldens(rd, fixtures, f2):
    dem = fixtures["fill/testdem1/dem"]
    for k in (1, 2):
        assert rd.fill_max_dep(dem, k).tobytes() == f2[f"max_dep/testdem1/{k}"].tobytes(), k
    assert rd.fill_max_dep(dem, 10 ** 9).tobytes() == fixtures["fill/testdem1/all_out"].tobytes()
    assert rd.fill_max_dep(dem, 0).tobytes() == dem.tobytes()


def test_max_dep_equals_compiled_reference_outputs(rd, f2):
    names = sorted({k.split("/")[0] for k in f2.files if not k.startswith("max_dep/") and not k.endswith("f64/dem")})
    for name in names:
        dem = f2[f"{name}/dem"]
        if dem.dtype != np.float32:
            continue
        for topo, nm in ((8, "D8"), (4, "D4")):
            for md in (0, 3, 40, 100000):
                got = rd.fill_max_dep(dem, md, nm)
                assert np.array_equal(got, f2[f"{name}/max_dep{md}_d{topo}"]), (name, topo, md)


def test_max_dep_random_tie_free(rd, orc):
    rng = np.random.default_rng(13)
    for i in range(30):
        h, w = (int(v) for v in rng.integers(3, 140, 2))
        z = (rng.random((h, w)) * 100).astype(np.float32)
        if np.unique(z).size != z.size:
            continue
        for topo, nm in ((8, "D8"), (4, "D4")):
            for md in (1, 2, 7, 30, 500):
                assert np.array_equal(rd.fill_max_dep(z, md, nm), orc.port.fill_max_dep(z, md, topo)), (i, topo, md)


def test_max_dep_integer_dems(rd, orc):
    """Integer DEMs are full of equal elevations, and then which cell of elevation L floods a pocket -- hence how pockets
    group into depressions and which groups pass the size test -- follows the pop order of the reference's heap: on
    these noisy DEMs the C restatement itself (same algorithm, its own heap) differs from the compiled reference in
    about half of the cases.  What holds whatever the order: every cell is either left alone or raised to the plain
    fill's level, a raised cell's whole pocket is raised with it, and the two limits (0: nothing, huge: the plain fill)
    are exact.  Mismatching cases are counted and printed, not hidden."""
    rng = np.random.default_rng(17)
    differ = total = 0
    for i in range(40):
        h, w = (int(v) for v in rng.integers(4, 80, 2))
        z = rng.integers(0, 5 + i, (h, w)).astype(np.int32)
        W = orc.port.fill(z, 8)
        for md in (1, 3, 10):
            got = rd.fill_max_dep(z, md)
            exp = orc.port.fill_max_dep(z, md, 8)
            assert ((got == z) | (got == W)).all()
            # all-or-nothing per pocket: two adjacent cells that the plain fill raises are either both raised or both left
            r = W > z
            up = got > z
            assert not (r[:, 1:] & r[:, :-1] & (up[:, 1:] != up[:, :-1])).any()
            assert not (r[1:, :] & r[:-1, :] & (up[1:, :] != up[:-1, :])).any()
            assert not (r[1:, 1:] & r[:-1, :-1] & (up[1:, 1:] != up[:-1, :-1])).any()
            assert not (r[1:, :-1] & r[:-1, 1:] & (up[1:, :-1] != up[:-1, 1:])).any()
            total += 1
            differ += not np.array_equal(got, exp)
        assert np.array_equal(rd.fill_max_dep(z, 0), z) and np.array_equal(rd.fill_max_dep(z, 10 ** 8), W)
    print(f"max_dep integer DEMs: {differ} of {total} cases differ from the C restatement (ties)")


def test_max_dep_tie_detector(rd, orc):
    """r05: WHERE the heap's order can decide is detected on the device -- pockets that two or more cells of their spill
    elevation can flood, and the clusters of pockets sharing a possible flooding cell with them (csrc/fill.hip k_md_ties).
    Outside the flagged clusters the output is a function of the DEM alone, so BOTH CPU implementations (the compiled
    reference on libstdc++'s heap, the C restatement on its own heap -- they disagree with each other on about half of these
    rasters) must agree with the engine on every unflagged cell.  Tie-free rasters flag nothing."""
    import torch

    rng = np.random.default_rng(23)
    checked = differing = flagged_cells = pocket_cells = 0
    backends = [orc.port] + ([orc.ref] if orc.ref.available else [])
    for i in range(60):
        h, w = (int(v) for v in rng.integers(4, 120, 2))
        if i % 3 == 0:
            z = rng.integers(0, 4 + i, (h, w)).astype(np.int32)
        elif i % 3 == 1:
            z = np.floor(rng.random((h, w)) * (6 + i)).astype(np.float32)
        else:
            z = rng.integers(0, 3 + i // 2, (h, w)).astype(np.uint8)
        for topo, nm in ((8, "D8"), (4, "D4")):
            for md in (1, 4, 25):
                t = torch.from_numpy(z).cuda()
                mask = torch.empty((h, w), dtype=torch.uint8, device="cuda")
                rd.fill_max_dep_ties_dev(t, md, mask, nm)
                torch.cuda.synchronize()
                got, m = t.cpu().numpy(), mask.cpu().numpy().astype(bool)
                st = rd.max_dep_stats()
                assert st["tie_cluster_cells"] == int(m.sum()) and st["pocket_cells"] == int((orc.port.fill(z, topo) > z).sum()), st
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    assert np.array_equal(got, rd.fill_max_dep(z, md, nm))      # (the host entry warns about the same pockets)
                for be in backends:
                    exp = be.fill_max_dep(z, md, topo)
                    d = got != exp
                    assert not (d & ~m).any(), (i, nm, md, int((d & ~m).sum()), st)
                    differing += int(d.any())
                    checked += 1
                flagged_cells += int(m.sum())
                pocket_cells += st["pocket_cells"]
    assert differing > 0          # (the sweep does contain rasters where the heaps disagree: the assertion above had work to do)
    print(f"max_dep tie detector: {differing} of {checked} comparisons differ from a CPU heap, always inside flagged clusters "
          f"({flagged_cells} of {pocket_cells} pocket cells flagged)")
    for seed in range(6):         # tie-free: nothing flagged, equal to the restatement
        z = (rng.permutation(90 * 70).reshape(70, 90) * 0.5).astype(np.float32)
        t = torch.from_numpy(z).cuda()
        mask = torch.empty(z.shape, dtype=torch.uint8, device="cuda")
        rd.fill_max_dep_ties_dev(t, 12, mask)
        torch.cuda.synchronize()
        st = rd.max_dep_stats()
        assert st["tie_pockets"] == 0 and st["tie_cluster_cells"] == 0 and not mask.any().item(), st
        assert np.array_equal(t.cpu().numpy(), orc.port.fill_max_dep(z, 12, 8))


def test_max_dep_sizes_and_types(rd, orc):
    from richdem_amd.synth import fractal_dem

    z = fractal_dem(700, 500, seed=5)
    zz = z + (np.arange(z.size, dtype=np.float32).reshape(z.shape) * np.float32(1e-3))   # spread the ties of the generator
    if np.unique(zz).size == zz.size:
        for md in (5, 200, 5000):
            assert np.array_equal(rd.fill_max_dep(zz, md), orc.port.fill_max_dep(zz, md, 8)), md
    for dt in (np.uint8, np.int16, np.uint16, np.uint32):
        q = (rng_dem := np.random.default_rng(3).integers(0, 200, (40, 50))).astype(dt)
        assert np.array_equal(rd.fill_max_dep(q, 10 ** 7), orc.port.fill(q, 8))
    # 64-bit element types (dense value ranks): values no float32 holds, tie free
    z64 = zz.astype(np.float64) * (1 + 2.0 ** -40) + 1e-9
    i64 = (np.random.default_rng(4).permutation(60 * 70).reshape(60, 70).astype(np.int64) << 33) - (1 << 44)
    for md in (5, 200, 10 ** 7):
        assert np.array_equal(rd.fill_max_dep(z64, md), orc.port.fill_max_dep(z64, md, 8)), md
        assert np.array_equal(rd.fill_max_dep(i64, md, "D4"), orc.port.fill_max_dep(i64, md, 4)), md
        assert np.array_equal(rd.fill_max_dep(i64.astype(np.uint64) + np.uint64(1 << 63), md),
                              orc.port.fill_max_dep(i64.astype(np.uint64) + np.uint64(1 << 63), md, 8)), md
