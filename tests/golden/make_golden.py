#!/usr/bin/env python3
"""Generates the committed golden vectors.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

* ref_fixtures.npz   -- the reference's OWN golden vectors, transcribed from its ArcInfo-ASCII test files
                        (tests/depressions/testdem1.{dem,all.out}, tests/flow_accum/*.{d8,out};
                        reference tests/tests.cpp:135-146, :233-271) plus its un-asserted inputs data/*.dem.
* ref_f2.npz         -- SURVEY 8(f2): the reference's max_dep goldens (tests/depressions/testdem1.{1,2}.out,
                        tests/tests.cpp:273-287) and outputs of the compiled reference's PriorityFloodEpsilon /
                        PriorityFloodWatersheds / PriorityFlood_Barnes2014_max_dep on seeded DEMs WITHOUT equal
                        elevations (with ties those three depend on std::priority_queue's pop order).
* ref_variants.npz   -- (--variants) PriorityFlood_Wei2018 (NoData holes as outlets), PriorityFlood_Original<D8/D4>,
                        HasDepressions<D8/D4> of the compiled reference on seeded rasters.
* ref_generated.npz  -- outputs of the UNMODIFIED reference headers (oracle/_ref/libref.so) on seeded
                        inputs, for the functions the reference has no golden file for
                        (d8_flow_directions, barnes_flat_resolution_d8, FA_D8, fill on float/int DEMs).
                        Inputs are stored too, so the tests do not depend on the generator.
* ref_s3_digests.npz -- (--s3-digests) per-1000-row-band digests of the compiled reference's outputs on the
                        40000 x 40000 bench DEM: BASELINE configs[2] / [4] at full size (see s3_digests()).
/root/reference does not exist on the GPU box; tests read only these .npz files.
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from richdem_amd.synth import fractal_dem, fractal_dem_int  # noqa: E402

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    oracle.build()
    assert oracle.ref.available, "oracle/_ref/libref.so missing (needs /root/reference)"
    fx = {}
    dem, nd = oracle.read_ascii_grid(f"{REF}/tests/depressions/testdem1.dem", np.int32)
    out, _ = oracle.read_ascii_grid(f"{REF}/tests/depressions/testdem1.all.out", np.int32)
    fx["fill/testdem1/dem"], fx["fill/testdem1/nodata"], fx["fill/testdem1/all_out"] = dem, np.int32(nd), out
    for f in sorted(glob.glob(f"{REF}/tests/flow_accum/*.d8")):
        name = os.path.basename(f)[:-3]
        dirs, nd = oracle.read_ascii_grid(f, np.int32)
        exp, end = oracle.read_ascii_grid(f[:-3] + ".out", np.int32)
        fx[f"accum/{name}/d8"] = dirs.astype(np.uint8)  # the reference loads these as uint8 (tests.cpp:138)
        fx[f"accum/{name}/nodata"] = np.uint8(int(nd) & 0xFF)
        fx[f"accum/{name}/out"] = exp
    for f in sorted(glob.glob(f"{REF}/data/*.dem")):
        name = os.path.basename(f)[:-4]
        d, nd = oracle.read_ascii_grid(f, np.float32)
        fx[f"data/{name}/dem"], fx[f"data/{name}/nodata"] = d, np.float32(nd)
    np.savez_compressed(os.path.join(HERE, "ref_fixtures.npz"), **fx)

    gen = {}
    R = oracle.ref
    cases = {
        "frac_f32": fractal_dem(96, 80, 11),
        "frac_i32": fractal_dem_int(96, 80, 12, 1.0),
        "flats_i32": fractal_dem_int(90, 70, 13, 0.05),
        "steps_i16": fractal_dem_int(64, 64, 14, 0.02, np.int16),
        "tilt_f32": fractal_dem(70, 50, 15, tilt=3.0),
    }
    hole = fractal_dem(80, 60, 16).copy()
    hole[20:30, 30:45] = -9999.0  # interior NoData block + a NoData edge strip
    hole[:, :3] = -9999.0
    cases["nodata_f32"] = hole
    for name, d in fx.items():
        if name.startswith("data/") and name.endswith("/dem"):
            cases["data_" + name.split("/")[1]] = d
    for name, dem in cases.items():
        nd = dem.dtype.type(fx[f"data/{name[5:]}/nodata"]) if name.startswith("data_") else dem.dtype.type(-9999)
        gen[f"{name}/dem"] = dem
        gen[f"{name}/nodata"] = nd
        filled = R.fill(dem, 8)
        gen[f"{name}/fill_d8"] = filled
        gen[f"{name}/fill_d4"] = R.fill(dem, 4)
        assert (filled == R.fill(dem, 8, oracle.BARNES2014_D8)).all()
        for tag, src in (("raw", dem), ("filled", filled)):
            gen[f"{name}/{tag}/d8_flowdirs"] = R.d8_flowdirs(src, nd)
            _, mask, labels = R.resolve_flats(src, nd)
            gen[f"{name}/{tag}/flat_mask"] = mask
            gen[f"{name}/{tag}/flat_labels"] = labels
            fr = R.flat_resolution(src, nd)
            gen[f"{name}/{tag}/flat_resolved_dirs"] = fr
            gen[f"{name}/{tag}/d8_flow_accum_f64"] = R.d8_flow_accum(fr, 255, np.float64)
            gen[f"{name}/{tag}/fa_d8"] = R.fa_d8(src, nd)
    np.savez_compressed(os.path.join(HERE, "ref_generated.npz"), **gen)
    print("wrote", len(fx), "fixture arrays and", len(gen), "generated arrays")


def variant_cases():
    """Seeded rasters for PriorityFlood_Wei2018 / PriorityFlood_Original / HasDepressions: NoData holes inside the raster
    (Wei2018 drains into them), NoData on the border, none at all, a raster without depressions."""
    rng = np.random.default_rng(6)
    cases = {}
    z = fractal_dem(130, 97, 61).copy()
    cases["plain_f32"] = (z, np.float32(-9999))
    holes = z.copy()
    holes[40:47, 50:61] = -9999
    holes[80, 20] = -9999
    holes[10:12, 100:103] = -9999
    holes[0:5, 0:9] = -9999
    cases["holes_f32"] = (holes, np.float32(-9999))
    q = np.floor((fractal_dem(83, 140, 62) + 1200) * 0.05).astype(np.int32)
    q[rng.random(q.shape) < 0.01] = -1
    cases["holes_i32"] = (q, np.int32(-1))
    u = np.floor((fractal_dem(70, 66, 63) + 1200) * 0.02).astype(np.uint8)
    u[30:33, 30:36] = 0          # NoData = 0, lower than everything, as the reference's precondition has it
    cases["holes_u8"] = (u, np.uint8(0))
    d = fractal_dem(64, 71, 64).astype(np.float64) + 1e-9 * np.arange(64 * 71).reshape(71, 64)
    d[20:24, 5:15] = -9999
    cases["holes_f64"] = (d, np.float64(-9999))
    yy, xx = np.mgrid[0:50, 0:60]
    cases["cone_f32"] = ((np.hypot(yy - 25, xx - 30) * -1.0).astype(np.float32), np.float32(-9999))   # no depression
    cases["bowl_i16"] = ((np.hypot(yy - 25, xx - 30) * 3).astype(np.int16), np.int16(-9999))           # one big one
    return cases


def variants():
    oracle.build()
    R = oracle.ref
    g = {}
    for name, (dem, nd) in variant_cases().items():
        g[f"{name}/dem"], g[f"{name}/nodata"] = dem, nd
        g[f"{name}/wei2018"] = R.fill_wei2018(dem, nd)
        for topo in (8, 4):
            g[f"{name}/original_d{topo}"] = R.fill_original(dem, topo)
            g[f"{name}/has_depressions_d{topo}"] = np.bool_(R.has_depressions(dem, topo))
    np.savez_compressed(os.path.join(HERE, "ref_variants.npz"), **g)
    print("wrote", len(g), "variant arrays")


def tie_free(dem, nodata=-9999.0):
    """the same DEM with every repeated data value nudged up by single representable steps until all differ"""
    out = dem.copy()
    for _ in range(64):
        flat = out.ravel()
        data = np.flatnonzero(flat != nodata)
        _, first = np.unique(flat[data], return_index=True)
        dup = np.setdiff1d(np.arange(data.size), first)
        if dup.size == 0:
            return out
        flat[data[dup]] = np.nextafter(flat[data[dup]], np.inf, dtype=flat.dtype)
    raise AssertionError("could not make the DEM tie free")


def f2():
    oracle.build()
    R = oracle.ref
    g = {}
    for k in (1, 2):
        out, _ = oracle.read_ascii_grid(f"{REF}/tests/depressions/testdem1.{k}.out", np.int32)
        g[f"max_dep/testdem1/{k}"] = out
    rng = np.random.default_rng(2)
    cases = {"frac_f32": tie_free(fractal_dem(96, 80, 21)), "tilt_f32": tie_free(fractal_dem(70, 50, 22, tilt=3.0)),
             "rand_f32": (rng.random((60, 75)) * 50).astype(np.float32),
             "ulps_f32": (np.float32(100.0).view(np.uint32) + rng.permutation(2 * 57 * 49)[: 57 * 49].astype(np.uint32)).view(np.float32).reshape(57, 49),
             "rand_f64": rng.random((50, 64)) * 1000}
    edge = fractal_dem(80, 60, 23).copy()
    edge[:12, :30] = -9999.0     # NoData region touching the raster border
    edge[:, -2:] = -9999.0
    cases["nodata_border_f32"] = tie_free(edge)
    for name, dem in cases.items():
        data = dem[dem != -9999.0]
        assert np.unique(data).size == data.size, name
        g[f"{name}/dem"] = dem
        for topo in (8, 4):
            g[f"{name}/epsilon_d{topo}"] = R.fill_epsilon(dem, -9999.0, topo)
            lab, filled = R.watersheds(dem, -9999.0, topo, True)
            g[f"{name}/watersheds_d{topo}"] = lab
            assert np.array_equal(filled, R.fill(dem, topo))
            for md in (0, 3, 40, 100000):
                g[f"{name}/max_dep{md}_d{topo}"] = R.fill_max_dep(dem, md, topo)
        g[f"{name}/pf_flowdirs"] = R.pf_flowdirs(dem, -9999.0)
    ties = {}
    # PriorityFloodFlowdirs breaks ties by insertion order (a total order): its output is a function of the DEM even
    # with equal elevations, so tie-heavy integer DEMs are pinned too
    for name, dem in {"ties_i32": rng.integers(0, 6, (47, 61)).astype(np.int32),
                      "ties_nodata_f32": np.where(rng.random((40, 53)) < 0.06, -9999.0, rng.integers(0, 40, (40, 53))).astype(np.float32)}.items():
        ties[f"{name}/dem"] = dem
        ties[f"{name}/pf_flowdirs"] = R.pf_flowdirs(dem, dem.dtype.type(-9999))
    np.savez_compressed(os.path.join(HERE, "ref_pf_flowdirs_ties.npz"), **ties)
    np.savez_compressed(os.path.join(HERE, "ref_f2.npz"), **g)
    print("wrote", len(g), "f2 arrays")


def s3_digests(size=40000, seed=3, out=None, keep=None):
    """BASELINE configs[2] / [4] at FULL size: the compiled reference runs ONCE here on the bench DEM G(seed=3)
    40000 x 40000 (fill -> barnes_flat_resolution_d8 -> d8_flow_accum<u8,f64>; fill -> ResolveFlatsEpsilon -> FA_D8)
    and only per-1000-row-band 64-bit digests (tests/golden/digest.py) plus a few counts are committed
    (ref_s3_digests.npz).  tests/test_s3_digests_gpu.py computes the same digests from the engine's outputs in HBM:
    seconds on the GPU box, so the driver's own `pytest -m gpu` run proves full-size parity of whatever HEAD is.
    FA_D8 proper needs ~78 GB of host memory at this size (its 36 B/cell proportions array); its digests come from
    oracle.port.fa_d8_lean -- the same rule and order on one receiver byte per cell, pinned to the compiled
    reference's FA_D8 in tests/test_oracle_pinning.py.  ~20 minutes of one host core, ~30 GB."""
    import time
    from digest import BAND_ROWS, band_digests_np
    oracle.build()
    R = oracle.ref
    assert R.available
    n = size
    g = {"size": np.int64(n), "seed": np.int64(seed), "band_rows": np.int64(BAND_ROWS)}
    times = {}
    keep = keep or os.environ.get("S3_KEEP")          # directory for the full arrays (debugging aid, not committed)
    if keep:
        os.makedirs(keep, exist_ok=True)

    def stash(name, a):
        if keep:
            np.save(os.path.join(keep, name + ".npy"), a)

    z = np.empty((n, n), np.float32)
    for y0 in range(0, n, 2000):
        z[y0:y0 + 2000] = fractal_dem(n, min(2000, n - y0), seed, y0=y0)
    g["dem"] = band_digests_np(z)
    t0 = time.perf_counter()
    W = R.fill(z, 8)                                   # PriorityFlood_Zhou2016 = FillDepressions<D8>
    times["fill"] = time.perf_counter() - t0
    g["fill"] = band_digests_np(W)
    g["fill_cells_raised"] = np.int64((W != z).sum())
    del z
    stash("fill", W)
    print("fill", times, int(g["fill_cells_raised"]), flush=True)
    t0 = time.perf_counter()
    dirs = R.flat_resolution(W, np.float32(-9999.0))   # barnes_flat_resolution_d8(alter=false)
    times["flat_resolution"] = time.perf_counter() - t0
    g["flat_dirs"] = band_digests_np(dirs)
    g["flat_dirs_noflow_left"] = np.int64((dirs == 0).sum())
    stash("flat_dirs", dirs)
    print("flats", times, int(g["flat_dirs_noflow_left"]), flush=True)
    t0 = time.perf_counter()
    area = R.d8_flow_accum(dirs, 255, np.float64)      # d8_flow_accum<uint8_t,double>, one thread
    times["d8_flow_accum"] = time.perf_counter() - t0
    g["d8_flow_accum"] = band_digests_np(area)
    g["d8_flow_accum_max"] = np.float64(area.max())
    stash("d8_flow_accum", area)
    del area, dirs
    print("accum", times, float(g["d8_flow_accum_max"]), flush=True)
    t0 = time.perf_counter()
    E = R.resolve_flats_epsilon(W, np.float32(-9999.0))   # ResolveFlatsEpsilon
    times["resolve_flats_epsilon"] = time.perf_counter() - t0
    g["resolve_flats_epsilon"] = band_digests_np(E)
    g["resolve_flats_epsilon_cells_changed"] = np.int64((E != W).sum())
    del W
    stash("resolve_flats_epsilon", E)
    print("rfe", times, flush=True)
    t0 = time.perf_counter()
    fa = oracle.port.fa_d8_lean(E, np.float32(-9999.0))
    times["fa_d8_lean_port"] = time.perf_counter() - t0
    g["fa_d8"] = band_digests_np(fa)
    g["fa_d8_max"] = np.float64(fa.max())
    stash("fa_d8", fa)
    for k, v in times.items():
        g["ref_seconds/" + k] = np.float64(round(v, 2))
    out = out or os.path.join(HERE, "ref_s3_digests.npz" if n == 40000 else f"ref_s3_digests_{n}.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, {k: float(v) for k, v in times.items()})


S3_SAMPLE_STRIDE = 982451653            # prime, coprime to 40000^2: sample j sits at cell (j * stride) mod (n*n)


def s3_sample_positions(n_cells: int, k: int) -> np.ndarray:
    """k distinct, quasi-uniform cell indices, reproducible in numpy and torch without wrapping arithmetic"""
    j = np.arange(k, dtype=np.int64)
    return (j * np.int64(S3_SAMPLE_STRIDE)) % np.int64(n_cells)


def _block_digests(a, rows=1000, cols=1000):
    """one digest per rows x cols block (the digest's position term is the cell's index in the WHOLE raster)"""
    from digest import _K1, _K2, _K3, _bits_np
    h, w = a.shape
    out = np.zeros((-(-h // rows), -(-w // cols)), np.uint64)
    with np.errstate(over="ignore"):
        for by, y0 in enumerate(range(0, h, rows)):
            band = np.ascontiguousarray(a[y0:y0 + rows])
            v = _bits_np(band)
            idx = (np.arange(y0, y0 + band.shape[0], dtype=np.int64)[:, None] * np.int64(w)
                   + np.arange(w, dtype=np.int64)[None, :])
            x = v * _K1 + idx * _K2
            x ^= x >> np.int64(32)
            x *= _K3
            for bx, x0 in enumerate(range(0, w, cols)):
                out[by, bx] = np.uint64(x[:, x0:x0 + cols].sum(dtype=np.int64).astype(np.uint64))
    return out


def s3_f2(which, size=40000, seed=3):
    """SURVEY 8(f2) at FULL size: the compiled reference's PriorityFloodFlowdirs / PriorityFloodEpsilon /
    PriorityFloodWatersheds / PriorityFlood_Barnes2014_max_dep(100) on the UNFILLED 40000 x 40000 bench DEM, one
    function per invocation (`--s3-f2 flowdirs|epsilon|watersheds|maxdep`, 10-20 GB and 5-15 minutes of one core
    each).  Committed per function (tests/golden/ref_s3_f2_<which>.npz): band digests, one digest per 1000 x 1000 block,
    and the reference's VALUES at a fixed quasi-uniform sample of cells (s3_sample_positions) -- the float32 bench DEM
    cannot avoid equal elevations, so for the three tie-sensitive outputs the GPU test COUNTS the sampled cells that
    differ (an estimate of the differing fraction, with the blocks that hold a difference) instead of asserting zero;
    max_dep is asserted equal."""
    import time
    from digest import BAND_ROWS, band_digests_np
    oracle.build()
    R = oracle.ref
    assert R.available
    n = size
    z = np.empty((n, n), np.float32)
    for y0 in range(0, n, 2000):
        z[y0:y0 + 2000] = fractal_dem(n, min(2000, n - y0), seed, y0=y0)
    nd = np.float32(-9999.0)
    k = {"flowdirs": 1 << 22, "epsilon": 1 << 19, "watersheds": 1 << 19, "maxdep": 1 << 16}[which]
    pos = s3_sample_positions(n * n, k)
    t0 = time.perf_counter()
    if which == "flowdirs":
        out = R.pf_flowdirs(z, nd)
    elif which == "epsilon":
        out = R.fill_epsilon(z, nd, 8)
    elif which == "watersheds":
        out = R.watersheds(z, nd, 8, False)[0]
    elif which == "maxdep":
        out = R.fill_max_dep(z, 100, 8)
    else:
        raise SystemExit("unknown output " + which)
    secs = time.perf_counter() - t0
    g = {"size": np.int64(n), "seed": np.int64(seed), "band_rows": np.int64(BAND_ROWS), "sample_k": np.int64(k),
         "sample_stride": np.int64(S3_SAMPLE_STRIDE), "ref_seconds": np.float64(round(secs, 2)),
         "bands": band_digests_np(out), "blocks": _block_digests(out), "sample": out.ravel()[pos]}
    if which in ("epsilon", "maxdep"):
        g["cells_changed"] = np.int64((out != z).sum())
    if which == "maxdep":
        # r05: the reference's RAISED cells of the whole raster as a bit mask (200 MB, scratch/ -- not committed); the blocks the
        # GPU test reports as differing are cut out of it by `--s3-f2-maxdep-blocks` and committed, so that the test can assert
        # that EVERY differing cell lies in a tie-flagged cluster of pockets (a raised cell holds its pocket's fill level, so the
        # mask and the plain fill reproduce the reference's output exactly)
        os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
        raised = out != z
        assert np.array_equal(out[raised], R.fill(z, 8)[raised])          # raised cells sit at the plain fill's level
        np.save(os.path.join(ROOT, "scratch", f"ref_s3_maxdep_raised_{n}.npy"), np.packbits(raised, axis=1))
    if which == "watersheds":
        g["labels"] = np.int64(out.max())
    path = os.path.join(HERE, f"ref_s3_f2_{which}.npz" if n == 40000 else f"ref_s3_f2_{which}_{n}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, round(secs, 1), "s", os.path.getsize(path), "bytes", flush=True)


def s3_f2_maxdep_blocks(blocks, size=40000):
    """Cuts the 1000 x 1000 blocks `blocks` (flat indices by * 40 + bx, as the GPU test reports them) out of the reference's
    raised-cell mask left in scratch/ by `--s3-f2 maxdep` and commits them: tests/golden/ref_s3_f2_maxdep_blocks.npz."""
    packed = np.load(os.path.join(ROOT, "scratch", f"ref_s3_maxdep_raised_{size}.npy"), mmap_mode="r")
    nb = -(-size // 1000)
    g = {"size": np.int64(size), "block_ids": np.array(sorted(blocks), np.int64)}
    for b in sorted(blocks):
        by, bx = divmod(int(b), nb)
        rows = np.unpackbits(np.asarray(packed[by * 1000:(by + 1) * 1000]), axis=1)[:, :size]
        g[f"raised/{b}"] = np.packbits(rows[:, bx * 1000:(bx + 1) * 1000], axis=1)
    path = os.path.join(HERE, "ref_s3_f2_maxdep_blocks.npz" if size == 40000 else f"ref_s3_f2_maxdep_blocks_{size}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def s2_dinf(size=10000, seed=2):
    """D-infinity on the record (north_star names it beside D8): the compiled reference's dinf_flow_directions
    (flowmet/dinf_flowdirs.hpp:128-152) and FA_Tarboton (methods/flow_accumulation.hpp:16 = FM_Tarboton,
    flowmet/Tarboton1997.hpp:14-144, + FlowAccumulation) on the FILLED 10000 x 10000 DEM G(seed=2) (BASELINE configs[1]'s
    raster).  Committed (ref_s2_dinf.npz): band digests of the float32 angles and of the accumulation cast to float32 --
    exact equality is the common case, reported per band -- and both outputs at a fixed sample of cells, from which the GPU
    test builds the ULP histogram (tolerance: 1 float32 ULP, north_star)."""
    import time
    from digest import BAND_ROWS, band_digests_np
    oracle.build()
    R = oracle.ref
    assert R.available
    n = size
    z = np.empty((n, n), np.float32)
    for y0 in range(0, n, 2000):
        z[y0:y0 + 2000] = fractal_dem(n, min(2000, n - y0), seed, y0=y0)
    nd = np.float32(-9999.0)
    W = R.fill(z, 8)
    del z
    k = 1 << 18
    pos = s3_sample_positions(n * n, k)
    t0 = time.perf_counter()
    ang = R.dinf_flowdirs(W, nd)
    t_ang = time.perf_counter() - t0
    t0 = time.perf_counter()
    fa = R.fa_tarboton(W, nd)
    t_fa = time.perf_counter() - t0
    fa32 = fa.astype(np.float32)
    g = {"size": np.int64(n), "seed": np.int64(seed), "band_rows": np.int64(BAND_ROWS), "sample_k": np.int64(k),
         "sample_stride": np.int64(S3_SAMPLE_STRIDE), "fill": band_digests_np(W),
         "dinf_bands": band_digests_np(ang), "dinf_sample": ang.ravel()[pos],
         "fa_tarboton_f32_bands": band_digests_np(fa32), "fa_tarboton_sample": fa.ravel()[pos], "fa_tarboton_max": np.float64(fa.max()),
         "ref_seconds/dinf_flow_directions": np.float64(round(t_ang, 2)), "ref_seconds/fa_tarboton": np.float64(round(t_fa, 2))}
    path = os.path.join(HERE, "ref_s2_dinf.npz" if n == 10000 else f"ref_s2_dinf_{n}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, t_ang, t_fa, os.path.getsize(path), flush=True)


if __name__ == "__main__":
    if "--s3-digests" in sys.argv:
        sys.path.insert(0, HERE)
        size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 40000
        s3_digests(size)
    elif "--s2-dinf" in sys.argv:
        sys.path.insert(0, HERE)
        s2_dinf(int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 10000)
    elif "--s3-f2-maxdep-blocks" in sys.argv:
        size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 40000
        s3_f2_maxdep_blocks([int(b) for b in sys.argv[sys.argv.index("--s3-f2-maxdep-blocks") + 1].split(",")], size)
    elif "--s3-f2" in sys.argv:
        sys.path.insert(0, HERE)
        size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 40000
        s3_f2(sys.argv[sys.argv.index("--s3-f2") + 1], size)
    elif "--f2" in sys.argv:
        f2()
    elif "--variants" in sys.argv:
        variants()
    else:
        main()
        f2()
        variants()
