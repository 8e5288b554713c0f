"""BASELINE configs[2] and [4] at FULL size, inside the driver-run suite: the engine's outputs on the 40000 x 40000
bench DEM G(seed=3) against digests of the COMPILED REFERENCE's outputs (tests/golden/ref_s3_digests.npz, made once by
`tests/golden/make_golden.py --s3-digests`: PriorityFlood_Zhou2016 -> barnes_flat_resolution_d8 ->
d8_flow_accum<uint8,double>, and fill -> ResolveFlatsEpsilon -> FA_D8).  One 64-bit position-dependent digest per band
of 1000 rows (tests/golden/digest.py): any differing cell fails its band.  Seconds on the GPU box, so `pytest -m gpu`
itself proves full-size parity of whatever HEAD is.  A 3000 x 3000 set of the same chain runs first (quick, and the one
that still fits a small device)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from digest import band_digests_torch  # noqa: E402

pytestmark = pytest.mark.gpu


def _bands(name, got, exp):
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, f"{name}: {bad.size} of {exp.size} bands differ from the reference, first bands {bad[:8].tolist()}"


def _chain(rd, g):
    import torch

    n, seed, rows = int(g["size"]), int(g["seed"]), int(g["band_rows"])
    nodata = -9999.0
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=seed)
    _bands("bench DEM", band_digests_torch(Z, rows), g["dem"])                     # the INPUT is the same raster
    W = Z.clone()
    rd.fill_depressions_dev(W)
    torch.cuda.synchronize()
    _bands("FillDepressions<D8>", band_digests_torch(W, rows), g["fill"])
    assert int((W != Z).sum().item()) == int(g["fill_cells_raised"])
    del Z
    dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    rd.d8_flow_directions_dev(W, nodata, dirs, flats=True)
    torch.cuda.synchronize()
    _bands("barnes_flat_resolution_d8", band_digests_torch(dirs, rows), g["flat_dirs"])
    assert int((dirs == 0).sum().item()) == int(g["flat_dirs_noflow_left"])
    area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    rd.d8_flow_accum_dev(dirs, area)
    torch.cuda.synchronize()
    _bands("d8_flow_accum<u8,f64>", band_digests_torch(area, rows), g["d8_flow_accum"])
    assert float(area.max().item()) == float(g["d8_flow_accum_max"])
    del dirs
    E = W.clone()
    rd.resolve_flats_epsilon_dev(E, nodata)
    torch.cuda.synchronize()
    _bands("ResolveFlatsEpsilon", band_digests_torch(E, rows), g["resolve_flats_epsilon"])
    assert int((E != W).sum().item()) == int(g["resolve_flats_epsilon_cells_changed"])
    del W
    area.fill_(1.0)
    rd.fa_d8_dev(E, nodata, area)
    torch.cuda.synchronize()
    _bands("FA_D8", band_digests_torch(area, rows), g["fa_d8"])
    assert float(area.max().item()) == float(g["fa_d8_max"])
    del area, E
    rd.release_workspace()
    torch.cuda.empty_cache()


def test_s3_chain_3000(rd):
    _chain(rd, np.load(os.path.join(GOLDEN, "ref_s3_digests_3000.npz")))


def test_s3_chain_full_size_equals_the_reference(rd):
    """40000 x 40000: every one of the 1.6e9 cells of the fill, the flat-resolved directions, d8_flow_accum, the
    epsilon-resolved DEM and FA_D8 enters a band digest that must equal the compiled reference's."""
    path = os.path.join(GOLDEN, "ref_s3_digests.npz")
    assert os.path.exists(path), "tests/golden/ref_s3_digests.npz missing (make_golden.py --s3-digests)"
    g = np.load(path)
    assert int(g["size"]) == 40000 and g["fill"].size == 40
    _chain(rd, g)
