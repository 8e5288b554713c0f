"""BASELINE configs[3] and [4] at FULL size on ONE GPU: the 40000 x 40000 bench DEM as 8 ROW BLOCKS through the product's
shard entries -- the tile protocol of programs/parallel_priority_flood (main.cpp:276-330, Zhou2016pf.hpp:142-227) and of
programs/parallel_d8_accum (main.cpp:270-464) as the ranks of an 8-GPU run execute it, block after block -- against the
SAME digests of the compiled reference's single-core outputs that pin the single-block engines
(tests/golden/ref_s3_digests.npz).  That is the reference's own acceptance idea for its distributed programs: tiling
invariance against the single-core answer (programs/parallel_priority_flood/test.py:44-118).

    fill        rdgpu_fill_shard_begin / _export_dev / rdgpu_fill_graph_solve_dev / _finish_dev   (sharded.fill_depressions_blocks)
                rdgpu_fill_sharded_f32(host, 8)  and  rdgpu_fill_multi_f32(host, devices = [0] * 8)
    directions  rdgpu_flat_shard_* over 8 blocks with ghost rows                                     (sharded.flat_resolution_blocks)
    accumulation rdgpu_accum_shard_begin_local / _links / _add_paths, ONE exchange                   (sharded.d8_flow_accum_blocks)

A 3000 x 3000 set of the same chain runs first."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from digest import band_digests_torch  # noqa: E402

pytestmark = pytest.mark.gpu

BLOCKS = 8


def _bands(name, got, exp):
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, f"{name}: {bad.size} of {exp.size} bands differ from the reference, first bands {bad[:8].tolist()}"


def _chain(rd, g, host_entries):
    import torch

    from richdem_amd._lib import check
    from richdem_amd.sharded import d8_flow_accum_blocks, fill_depressions_blocks, flat_resolution_blocks

    n, seed, rows = int(g["size"]), int(g["seed"]), int(g["band_rows"])
    nodata = -9999.0
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=seed)
    _bands("bench DEM", band_digests_torch(Z, rows), g["dem"])
    # ---- configs[3]: the fill over 8 row blocks, HBM-resident shard protocol --------------------------------------------
    W = Z.clone()
    fill_depressions_blocks(W, BLOCKS)
    torch.cuda.synchronize()
    _bands(f"fill, {BLOCKS} row blocks (shard protocol)", band_digests_torch(W, rows), g["fill"])
    assert int((W != Z).sum().item()) == int(g["fill_cells_raised"])
    if host_entries:
        # the C-ABI's own drivers of the same protocol, on a host raster (what rd_depressions_flood hands over)
        L = rd.lib()
        host = Z.cpu().numpy()
        check(L.rdgpu_fill_sharded_f32(host.ctypes.data_as(ctypes.c_void_p), n, n, 8, BLOCKS), "rdgpu_fill_sharded_f32")
        _bands("rdgpu_fill_sharded_f32(8)", band_digests_torch(torch.from_numpy(host).cuda(), rows), g["fill"])
        host = Z.cpu().numpy()
        devs = (ctypes.c_int * BLOCKS)(*([0] * BLOCKS))
        check(L.rdgpu_fill_multi_f32(host.ctypes.data_as(ctypes.c_void_p), n, n, 8, devs, BLOCKS), "rdgpu_fill_multi_f32")
        _bands("rdgpu_fill_multi_f32(devices=[0]*8)", band_digests_torch(torch.from_numpy(host).cuda(), rows), g["fill"])
        del host
    del Z
    # ---- configs[4]: directions + flat resolution, then the accumulation, over the same 8 blocks --------------------------
    dirs, exchanges = flat_resolution_blocks(W, nodata, BLOCKS)
    torch.cuda.synchronize()
    _bands(f"barnes_flat_resolution_d8, {BLOCKS} row blocks", band_digests_torch(dirs, rows), g["flat_dirs"])
    assert int((dirs == 0).sum().item()) == int(g["flat_dirs_noflow_left"])
    del W
    dirs = dirs.contiguous()
    area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ex = d8_flow_accum_blocks(dirs, area, BLOCKS)
    torch.cuda.synchronize()
    assert ex == 1, f"the accumulation over {BLOCKS} blocks took {ex} exchanges (a DEM's directions are loop free: 1)"
    _bands(f"d8_flow_accum<u8,f64>, {BLOCKS} row blocks, one exchange", band_digests_torch(area, rows), g["d8_flow_accum"])
    assert float(area.max().item()) == float(g["d8_flow_accum_max"])
    del dirs, area
    rd.release_workspace()
    torch.cuda.empty_cache()
    return exchanges


def test_s3_blocks_3000(rd):
    _chain(rd, np.load(os.path.join(GOLDEN, "ref_s3_digests_3000.npz")), host_entries=True)


def test_s3_in_8_row_blocks_equals_the_reference(rd):
    """40000 x 40000 in 8 blocks of 5000 rows: every cell of the sharded fill (three drivers), of the sharded flat-resolved
    directions and of the one-exchange accumulation enters a band digest that must equal the compiled reference's."""
    path = os.path.join(GOLDEN, "ref_s3_digests.npz")
    assert os.path.exists(path), "tests/golden/ref_s3_digests.npz missing (make_golden.py --s3-digests)"
    g = np.load(path)
    assert int(g["size"]) == 40000 and g["fill"].size == 40
    _chain(rd, g, host_entries=True)
