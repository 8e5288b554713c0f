"""The band digests of tests/golden/digest.py: numpy (the reference side) and torch (the engine side) agree, and a
single moved or changed cell changes its band.  CPU only."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from digest import band_digests_np, band_digests_torch  # noqa: E402


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8, np.int32])
def test_numpy_and_torch_digests_agree(dtype):
    import torch

    rng = np.random.default_rng(5)
    a = (rng.random((2300, 130)) * 1000 - 200).astype(dtype)
    d = band_digests_np(a)
    assert d.dtype == np.uint64 and d.size == 3
    assert np.array_equal(d, band_digests_torch(torch.from_numpy(a)))
    b = a.copy()
    b[1500, 7], b[1500, 8] = a[1500, 8], a[1500, 7] + 1
    assert (band_digests_np(b) != d).tolist() == [False, True, False]


def test_committed_s3_digests_are_well_formed():
    for name, n in (("ref_s3_digests_3000.npz", 3000), ("ref_s3_digests.npz", 40000)):
        g = np.load(os.path.join(GOLDEN, name))
        assert int(g["size"]) == n and int(g["seed"]) == 3
        for k in ("dem", "fill", "flat_dirs", "d8_flow_accum", "resolve_flats_epsilon", "fa_d8"):
            assert g[k].dtype == np.uint64 and g[k].size == (n + 999) // 1000, (name, k)


def test_small_digest_set_matches_the_oracle_port(orc):
    """the 3000 x 3000 set again, from the C restatement (pins the generator, the digest and the chain's wiring on the CPU)"""
    from richdem_amd.synth import fractal_dem

    g = np.load(os.path.join(GOLDEN, "ref_s3_digests_3000.npz"))
    z = fractal_dem(3000, 3000, 3)
    assert np.array_equal(band_digests_np(z), g["dem"])
    W = orc.port.fill(z, 8)
    assert np.array_equal(band_digests_np(W), g["fill"])
    dirs = orc.port.flat_resolution(W, np.float32(-9999.0))
    assert np.array_equal(band_digests_np(dirs), g["flat_dirs"])
    assert np.array_equal(band_digests_np(orc.port.d8_flow_accum(dirs, 255, np.float64)), g["d8_flow_accum"])
    E = orc.port.resolve_flats_epsilon(W, np.float32(-9999.0))
    assert np.array_equal(band_digests_np(E), g["resolve_flats_epsilon"])
    assert np.array_equal(band_digests_np(orc.port.fa_d8_lean(E, np.float32(-9999.0))), g["fa_d8"])
    assert np.array_equal(orc.port.fa_d8_lean(E[:700, :900], np.float32(-9999.0)), orc.port.fa_d8(E[:700, :900], np.float32(-9999.0)))
