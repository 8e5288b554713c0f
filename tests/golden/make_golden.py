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
* ref_generated.npz  -- outputs of the UNMODIFIED reference headers (oracle/_ref/libref.so) on seeded
                        inputs, for the functions the reference has no golden file for
                        (d8_flow_directions, barnes_flat_resolution_d8, FA_D8, fill on float/int DEMs).
                        Inputs are stored too, so the tests do not depend on the generator.
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
    np.savez_compressed(os.path.join(HERE, "ref_f2.npz"), **g)
    print("wrote", len(g), "f2 arrays")


if __name__ == "__main__":
    if "--f2" in sys.argv:
        f2()
    else:
        main()
        f2()
