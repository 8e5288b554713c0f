"""SURVEY 8(f2) at FULL size, quantified: the other outputs of the Priority-Flood sweep on the UNFILLED 40000 x 40000 bench
DEM against the COMPILED REFERENCE (tests/golden/ref_s3_f2_<output>.npz, made once by `make_golden.py --s3-f2 <output>`:
PriorityFloodFlowdirs_Barnes2014 838 s, PriorityFlood_Barnes2014_max_dep(100) 809 s, PriorityFloodEpsilon_Barnes2014 and
PriorityFloodWatersheds_Barnes2014 of one core each).  Per output the file holds one digest per 1000-row band, one per
1000 x 1000 block, and the reference's VALUES at a fixed quasi-uniform sample of cells (cell (j * 982451653) mod 1.6e9).

* max_dep: the pockets are order free; which of them one flooding cell joins into one run is not.  The engine flags the
  pockets where that can happen on the device, the reference's cells of the blocks that differ are committed, and the test
  asserts that every differing cell lies in a tie-flagged cluster and that all other cells are equal.
* PriorityFloodFlowdirs runs on the reference's STABLE queue: its output is a function of the DEM, ties included, and the
  engine reproduces it: every band and block digest must equal the reference's.
* The epsilon fill and the watershed labels follow std::priority_queue's pop order among equal elevations -- and a float32
  raster of 1.6e9 cells cannot avoid equal elevations.  For these the test COUNTS: bands and blocks whose digest differs, and
  the sampled cells that differ (an estimate of the differing fraction; +- 3 / sqrt(hits) relative).  The numbers go to the
  test's warning line, to gpurun_out/s3_f2.json (-> profiles/) and bench.py repeats the sample count beside the two f2 stage
  times (`cells_differing_from_reference`).  Bounds asserted: the fractions measured in r04 with a factor of safety."""
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


def block_digests_torch(t, rows=1000, cols=1000):
    """tests/golden/make_golden.py::_block_digests on a CUDA tensor"""
    import torch

    from digest import _K1, _K2, _K3

    h, w = t.shape
    k1, k2, k3 = int(_K1), int(_K2), int(_K3)
    out = np.zeros((-(-h // rows), -(-w // cols)), np.uint64)
    for by, y0 in enumerate(range(0, h, rows)):
        band = t[y0:y0 + rows]
        if band.dtype == torch.float32:
            v = band.contiguous().view(torch.int32).to(torch.int64)
        else:
            v = band.to(torch.int64)
        idx = (torch.arange(y0, y0 + band.shape[0], dtype=torch.int64, device=t.device)[:, None] * w
               + torch.arange(w, dtype=torch.int64, device=t.device)[None, :])
        x = v * k1 + idx * k2
        x = x ^ (x >> 32)
        x = x * k3
        nb = -(-w // cols)
        pad = nb * cols - w
        if pad:
            x = torch.nn.functional.pad(x, (0, pad))
        out[by] = x.view(band.shape[0], nb, cols).sum(dim=(0, 2), dtype=torch.int64).cpu().numpy().view(np.uint64)
    return out


def sample_of(t, g):
    import torch

    k, stride = int(g["sample_k"]), int(g["sample_stride"])
    pos = (torch.arange(k, dtype=torch.int64, device=t.device) * stride) % t.numel()
    return t.reshape(-1)[pos], pos


def compare(name, out, g, report, same=None):
    """digests + sample of `out` against the reference file g; `same(got, ref)` -> bool tensor for the sample (default ==)"""
    import torch

    rows = int(g["band_rows"])
    bands = band_digests_torch(out, rows)
    blocks = block_digests_torch(out)
    got, _ = sample_of(out, g)
    ref = torch.from_numpy(g["sample"]).to(out.device)
    eq = (got == ref) if same is None else same(got, ref)
    k = int(g["sample_k"])
    ndiff = int((~eq).sum().item())
    report[name] = {"bands_differing": int((bands != g["bands"]).sum()), "bands": int(g["bands"].size),
                    "blocks_differing": int((blocks != g["blocks"]).sum()), "blocks": int(g["blocks"].size),
                    "sample_cells": k, "sample_differing": ndiff, "fraction": ndiff / k,
                    "estimated_cells_differing": int(round(ndiff / k * out.numel())),
                    "reference_seconds_one_core": float(g["ref_seconds"])}
    return report[name]


def _load(which):
    path = os.path.join(GOLDEN, f"ref_s3_f2_{which}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (make_golden.py --s3-f2 {which})")
    g = np.load(path)
    assert int(g["size"]) == 40000
    return g


def _write(report):
    path = os.path.join(ROOT, "gpurun_out", "s3_f2.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = {}
    if os.path.exists(path):
        with open(path) as f:
            old = json.load(f)
    old.update(report)
    with open(path, "w") as f:
        json.dump(old, f, indent=1)


def _dem(rd, g):
    import torch

    n = int(g["size"])
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=int(g["seed"]))
    return Z


def test_s3_max_dep_differs_only_in_tie_flagged_clusters(rd):
    """PriorityFlood_Barnes2014_max_dep(100) at 40000^2 against the compiled reference.  The pockets are order free; WHICH
    pockets one flooding cell unites into one run (and so whether the run stays under the size limit) follows the heap's order
    among equal elevations.  r04 counted 6 of 1600 blocks holding a difference and asserted that it was a tie effect; r05
    PROVES it cell by cell: the engine flags on the device every pocket that two or more cells of its spill elevation can
    flood, with the cluster of pockets sharing a possible flooding cell with it (k_md_ties, csrc/fill.hip) -- outside those
    clusters the output does not depend on the pop order -- and the reference's cells of every differing block are committed
    (tests/golden/ref_s3_f2_maxdep_blocks.npz: its raised-cell mask, cut from the full-size run by make_golden.py
    --s3-f2-maxdep-blocks; a raised cell sits at the plain fill's level, which reproduces the block exactly).  Asserted:
    every block whose digest differs is one of the committed ones, reconstructing the reference's block gives the reference's
    digest, EVERY differing cell lies in a tie-flagged cluster, and every other cell of the raster is equal (digests)."""
    import torch

    from digest import _K1, _K2, _K3

    g = _load("maxdep")
    Z = _dem(rd, g)
    W = Z.clone()
    mask = torch.empty(Z.shape, dtype=torch.uint8, device="cuda")
    rd.fill_max_dep_ties_dev(W, 100, mask)
    torch.cuda.synchronize()
    st = rd.max_dep_stats()
    rep = {}
    r = compare("max_dep_100", W, g, rep)
    r["cells_changed"] = int((W != Z).sum().item())
    r["reference_cells_changed"] = int(g["cells_changed"])
    r.update({k: int(v) for k, v in st.items()})
    blocks = block_digests_torch(W)
    ids = [int(b) for b in np.flatnonzero((blocks != g["blocks"]).ravel())]
    r["blocks_differing_ids"] = ids
    _write(rep)
    F = Z.clone()
    rd.fill_depressions_dev(F)                     # the level a raised cell is raised to
    torch.cuda.synchronize()
    path = os.path.join(GOLDEN, "ref_s3_f2_maxdep_blocks.npz")
    nb = blocks.shape[1]
    have = np.load(path) if os.path.exists(path) else None
    known = set(int(b) for b in have["block_ids"]) if have is not None else set()
    cells_differing = outside = 0
    per_block = {}
    for b in ids:
        by, bx = divmod(b, nb)
        sl = (slice(by * 1000, (by + 1) * 1000), slice(bx * 1000, (bx + 1) * 1000))
        if b not in known:
            continue
        raised = torch.from_numpy(np.unpackbits(have[f"raised/{b}"], axis=1)[:, :1000].astype(bool)).cuda()
        ref = torch.where(raised, F[sl], Z[sl])
        # the reconstruction IS the reference's block: its digest equals the committed one
        idx = (torch.arange(by * 1000, (by + 1) * 1000, dtype=torch.int64, device="cuda")[:, None] * Z.shape[1]
               + torch.arange(bx * 1000, (bx + 1) * 1000, dtype=torch.int64, device="cuda")[None, :])
        x = ref.contiguous().view(torch.int32).to(torch.int64) * int(_K1) + idx * int(_K2)
        x = (x ^ (x >> 32)) * int(_K3)
        assert np.uint64(x.sum(dtype=torch.int64).cpu().numpy().view(np.uint64)) == g["blocks"][by, bx], ("reconstruction", b)
        d = W[sl] != ref
        nd, no = int(d.sum().item()), int((d & (mask[sl] == 0)).sum().item())
        per_block[str(b)] = {"cells_differing": nd, "outside_tie_flagged_clusters": no,
                             "tie_flagged_cells_in_block": int(mask[sl].sum().item())}
        cells_differing += nd
        outside += no
    r.update({"cells_differing_in_committed_blocks": cells_differing, "of_them_outside_tie_flagged_clusters": outside,
              "per_block": per_block, "blocks_without_committed_reference_cells": sorted(set(ids) - known)})
    _write(rep)
    # what the next `make_golden.py --s3-f2-maxdep-blocks` needs, and the engine's own cells there (builder's cross-check)
    torch.save({"ids": ids, "raised": {b: (W != Z)[divmod(b, nb)[0] * 1000:(divmod(b, nb)[0] + 1) * 1000,
                                                 divmod(b, nb)[1] * 1000:(divmod(b, nb)[1] + 1) * 1000].cpu() for b in ids},
                "mask": {b: mask[divmod(b, nb)[0] * 1000:(divmod(b, nb)[0] + 1) * 1000,
                              divmod(b, nb)[1] * 1000:(divmod(b, nb)[1] + 1) * 1000].cpu() for b in ids}},
               os.path.join(ROOT, "gpurun_out", "s3_maxdep_blocks.pt"))
    warnings.warn(f"PriorityFlood_Barnes2014_max_dep(100) at 40000^2 vs the compiled reference: {len(ids)} of {r['blocks']} blocks hold a "
                  f"difference ({ids}), {cells_differing} cells, {outside} of them outside tie-flagged clusters; "
                  f"{st['tie_pockets']} of {st['pockets']} pockets have two or more possible flooding cells, their clusters hold "
                  f"{st['tie_cluster_cells']} of {st['pocket_cells']} pocket cells; {r['cells_changed']} cells raised vs "
                  f"{r['reference_cells_changed']}", UserWarning)
    if have is None:
        pytest.xfail(f"tests/golden/ref_s3_f2_maxdep_blocks.npz not generated yet: make_golden.py --s3-f2-maxdep-blocks {','.join(map(str, ids))}")
    assert set(ids) <= known, ("a block differs whose reference cells are not committed", sorted(set(ids) - known))
    assert outside == 0, r                          # a difference outside the flagged clusters would be a bug at scale, not a tie
    assert r["sample_differing"] <= 2 and len(ids) <= 24 and cells_differing <= 4000, r
    del Z, W, F, mask
    rd.release_workspace()
    torch.cuda.empty_cache()


def test_s3_pf_flowdirs_equals_the_reference(rd):
    import torch

    g = _load("flowdirs")
    Z = _dem(rd, g)
    dirs = torch.empty(Z.shape, dtype=torch.uint8, device="cuda")
    rd.pf_flowdirs_dev(Z, -9999.0, dirs)
    torch.cuda.synchronize()
    rep = {}
    r = compare("priority_flood_flowdirs", dirs, g, rep)
    r.update({k: int(v) for k, v in rd.pf_flowdirs_stats().items()})
    _write(rep)
    warnings.warn(f"PriorityFloodFlowdirs at 40000^2 vs the compiled reference: {r['sample_differing']} of {r['sample_cells']} sampled "
                  f"cells differ (~{r['estimated_cells_differing']} cells, fraction {r['fraction']:.2e}); {r['blocks_differing']} of "
                  f"{r['blocks']} blocks hold a difference; twins {r['twins']}, unresolved {r['unresolved']}", UserWarning)
    # r04: the stable queue's tie order is reproduced (csrc/pfdirs.hip: fixed point of the discovery order) -- every band and
    # every block digest of the 1.6e9 directions equals the compiled reference's, although 1.58e9 cells have a twin
    assert r["bands_differing"] == 0 and r["blocks_differing"] == 0 and r["sample_differing"] == 0 and r["unresolved"] == 0, r
    del Z, dirs
    rd.release_workspace()
    torch.cuda.empty_cache()


def test_s3_epsilon_difference_is_counted_and_bounded(rd):
    import torch

    g = _load("epsilon")
    Z = _dem(rd, g)
    E = Z.clone()
    rd.fill_epsilon_dev(E, -9999.0)
    torch.cuda.synchronize()
    rep = {}
    r = compare("priority_flood_epsilon", E, g, rep)
    got, _ = sample_of(E, g)
    ref = torch.from_numpy(g["sample"]).cuda()
    steps = (ref.view(torch.int32).to(torch.int64) - got.view(torch.int32).to(torch.int64))
    r["sample_above_reference"] = int((got > ref).sum().item())       # the order-free surface is a lower bound
    r["max_steps_below_reference"] = int(steps.max().item())
    r["cells_changed"] = int((E != Z).sum().item())
    r["reference_cells_changed"] = int(g["cells_changed"])
    r["tie_sources"] = int(rd.epsilon_stats()["tie_sources"])
    _write(rep)
    warnings.warn(f"PriorityFloodEpsilon at 40000^2 vs the compiled reference: {r['sample_differing']} of {r['sample_cells']} sampled "
                  f"cells differ (fraction {r['fraction']:.4f}, at most {r['max_steps_below_reference']} representable steps below, "
                  f"{r['sample_above_reference']} above); tie sources {r['tie_sources']}", UserWarning)
    assert r["sample_above_reference"] == 0, r
    # bounded (r05): 13 of 524 288 sampled cells in r04 = 2.5e-5; a regression to 1e-4 of the sample fails
    assert r["sample_differing"] <= 52 and r["fraction"] <= 1e-4, r
    assert r["max_steps_below_reference"] <= 4096, r
    del Z, E
    rd.release_workspace()
    torch.cuda.empty_cache()


def test_s3_watersheds_difference_is_counted(rd):
    import torch

    g = _load("watersheds")
    Z = _dem(rd, g)
    lab = torch.empty(Z.shape, dtype=torch.int32, device="cuda")
    rd.watersheds_dev(Z, -9999.0, lab)
    torch.cuda.synchronize()
    got = sample_of(lab, g)[0].cpu().numpy().astype(np.int64)
    ref = g["sample"].astype(np.int64)
    # The numbering follows the pop order too, so the PARTITIONS are compared, on the sample: of the pairs of sampled cells
    # that share a watershed in one result (neighbours in the sample sorted by that result's label), how many are
    # separated in the other.  The same cells are unlabelled (-1: NoData connected to the border) in both.
    k = int(g["sample_k"])
    assert np.array_equal(got == -1, ref == -1)

    def split(a, b):
        o = np.argsort(a, kind="stable")
        sa, sb = a[o], b[o]
        pair = (sa[1:] == sa[:-1]) & (sa[1:] != -1)
        return int(pair.sum()), int((pair & (sb[1:] != sb[:-1])).sum())

    pr, sr = split(ref, got)
    pg, sg = split(got, ref)
    rep = {"priority_flood_watersheds": {"sample_cells": k, "labels": int(lab.max().item()), "reference_labels": int(g["labels"]),
                                         "pairs_in_one_reference_watershed": pr, "of_them_separated_here": sr,
                                         "pairs_in_one_watershed_here": pg, "of_them_separated_in_the_reference": sg,
                                         "fraction": (sr + sg) / max(pr + pg, 1),
                                         "reference_seconds_one_core": float(g["ref_seconds"])}}
    _write(rep)
    r = rep["priority_flood_watersheds"]
    warnings.warn(f"PriorityFloodWatersheds at 40000^2 vs the compiled reference: of {pr} sampled pairs inside one reference "
                  f"watershed {sr} are separated here, of {pg} pairs inside one watershed here {sg} are separated in the reference "
                  f"(fraction {r['fraction']:.4f}); labels {r['labels']} vs {r['reference_labels']}", UserWarning)
    del Z, lab
    rd.release_workspace()
    torch.cuda.empty_cache()
