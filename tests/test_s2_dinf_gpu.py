"""D-infinity on the record at 10000 x 10000 (north_star names D-infinity beside D8): dinf_flow_directions
(flowmet/dinf_flowdirs.hpp:128-152) and FA_Tarboton (methods/flow_accumulation.hpp:16 = FM_Tarboton, flowmet/Tarboton1997.hpp:14-144,
+ FlowAccumulation) of the FILLED BASELINE configs[1] raster against the COMPILED REFERENCE (tests/golden/ref_s2_dinf.npz,
`make_golden.py --s2-dinf`): band digests -- bitwise equality of the float32 angles / of the accumulation cast to float32,
counted per band -- and a ULP histogram over a fixed sample of 262 144 cells.  Tolerance (north_star): 1 float32 ULP; the
angles come from device atan2 / sqrt in double against glibc's, the accumulation sums in a different order than the
reference's FIFO."""
import json
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, GOLDEN)
from digest import band_digests_torch  # noqa: E402

pytestmark = pytest.mark.gpu


def _ulps32(a, b):
    """distance in float32 representable steps (both finite, same sign or zero)"""
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


def test_dinf_10k_against_the_compiled_reference(rd):
    import torch

    path = os.path.join(GOLDEN, "ref_s2_dinf.npz")
    assert os.path.exists(path), "tests/golden/ref_s2_dinf.npz missing (make_golden.py --s2-dinf)"
    g = np.load(path)
    n, rows = int(g["size"]), int(g["band_rows"])
    W = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(W, seed=int(g["seed"]))
    rd.fill_depressions_dev(W)
    torch.cuda.synchronize()
    assert np.array_equal(band_digests_torch(W, rows), g["fill"])          # the same input as the reference's
    k, stride = int(g["sample_k"]), int(g["sample_stride"])
    pos = (torch.arange(k, dtype=torch.int64, device="cuda") * stride) % (n * n)
    report = {}
    # ---- the angle raster ---------------------------------------------------------------------------------------------
    ang = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.dinf_flow_directions_dev(W, -9999.0, ang)
    torch.cuda.synchronize()
    bands = band_digests_torch(ang, rows)
    got = ang.reshape(-1)[pos].cpu().numpy()
    u = _ulps32(got, g["dinf_sample"])
    report["dinf_flow_directions"] = {"bands_bitwise_equal": int((bands == g["dinf_bands"]).sum()), "bands": int(bands.size),
                                      "sample_cells": k, "ulp_histogram": {str(i): int((u == i).sum()) for i in range(3)},
                                      "max_ulp": int(u.max())}
    assert u.max() <= 1, report
    del ang
    # ---- FA_Tarboton, unit weights -------------------------------------------------------------------------------------
    acc = torch.ones((n, n), dtype=torch.float64, device="cuda")
    rd.fa_tarboton_dev(W, -9999.0, acc)
    torch.cuda.synchronize()
    a32 = acc.to(torch.float32)
    bands = band_digests_torch(a32, rows)
    got64 = acc.reshape(-1)[pos].cpu().numpy()
    ref64 = g["fa_tarboton_sample"]
    u = _ulps32(got64.astype(np.float32), ref64.astype(np.float32))
    rel = np.abs(got64 - ref64) / np.maximum(np.abs(ref64), 1e-300)
    report["fa_tarboton"] = {"bands_bitwise_equal_after_f32_cast": int((bands == g["fa_tarboton_f32_bands"]).sum()), "bands": int(bands.size),
                             "sample_cells": k, "ulp32_histogram": {str(i): int((u == i).sum()) for i in range(3)},
                             "max_ulp32": int(u.max()), "max_relative_difference_f64": float(rel.max()),
                             "max_accum": float(acc.max().item()), "reference_max_accum": float(g["fa_tarboton_max"])}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "s2_dinf.json"), "w") as f:
        json.dump(report, f, indent=1)
    warnings.warn("D-infinity at 10000^2 vs the compiled reference: " + json.dumps(report), UserWarning)
    assert u.max() <= 1 and rel.max() <= 1e-12, report
    del acc, a32, W
    rd.release_workspace()
    torch.cuda.empty_cache()
