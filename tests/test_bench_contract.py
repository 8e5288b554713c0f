"""bench.py's contract pieces that do not need a GPU: the roofline object (richdem_amd/roofline.py) and the rule that
stdout carries nothing but rank 0's JSON line (library chatter goes to stderr)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_object(tmp_path):
    from richdem_amd.roofline import FILL_ALG_BYTES_PER_CELL, HBM_PEAK_GBS, fill_roofline

    cells, steps = 1_600_000_000, 3
    stats = {"scan_tiles": 781250, "tile_cells": 2048}
    prof = {"fill.scan": (19.5, 3), "fill.descent": (18.6, 3), "fill.finalize": (10.2, 3), "fill.edge_round": (5.6, 24)}
    r = fill_roofline(prof, stats, cells, steps, step_seconds=0.0222)
    assert r["kernel"] == "fill.scan" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == HBM_PEAK_GBS
    alg = 781250 * 2048 * FILL_ALG_BYTES_PER_CELL                       # one launch visits every tile
    assert abs(r["achieved"] - alg / (19.5 / 3 / 1e3) / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert r["launches_per_step"] == 1.0 and abs(r["alg_GB_per_launch"] - 12.8) < 1e-9 and r["traffic"] is None
    assert abs(r["whole_fill_alg_GBps"] - cells * 8 / 0.0222 / 1e9) < 0.1
    # another raster kernel dominating: its launches visit every cell
    prof["fill.descent"] = (30.0, 3)
    r = fill_roofline(prof, stats, cells, steps)
    assert r["kernel"] == "fill.descent" and abs(r["achieved"] - cells * 8 / 0.010 / 1e9) < 0.1 and "whole_fill_alg_GBps" not in r
    # raster fallback rounds: several scan launches, only the visited tiles count
    r = fill_roofline({"fill.scan": (57.0, 27)}, {"scan_tiles": 3_500_000, "tile_cells": 2048}, cells, steps)
    assert r["launches_per_step"] == 9.0 and abs(r["alg_GB_per_launch"] - 3_500_000 * 2048 * 8 / 9 / 1e9) < 1e-3
    assert fill_roofline({}, stats, cells, steps) is None
    # traffic comes from a PMC summary, per kernel, only for the size AND the fill.hip it was measured on
    from richdem_amd.roofline import engine_sha

    pt = str(tmp_path / "pmc_traffic.json")
    per = {"size": 40000, "GB_per_launch": {"fill.scan": 15.1}, "GB_per_fill": 59.0, "engine_sha": engine_sha()}
    json.dump(per, open(pt, "w"))
    r = fill_roofline({"fill.scan": (19.5, 3)}, stats, cells, steps, step_seconds=0.0222, traffic_file=pt, size=40000)
    assert r["traffic"] == 15.1 and r["whole_fill_traffic_GB"] == 59.0 and abs(r["whole_fill_pass_count"] - 59.0 / 12.8) < 0.01
    assert list(r)[:3] == ["bound", "whole_fill_alg_GBps", "whole_fill_frac"]          # the whole fill leads
    assert abs(r["whole_fill_frac"] - cells * 8 / 0.0222 / 1e9 / 8000.0) < 1e-4
    assert fill_roofline({"fill.scan": (19.5, 3)}, stats, cells, steps, traffic_file=pt, size=123)["traffic"] is None
    per["engine_sha"] = "0123456789abcdef"                                             # measured on other kernels: stale
    json.dump(per, open(pt, "w"))
    assert fill_roofline({"fill.scan": (19.5, 3)}, stats, cells, steps, traffic_file=pt, size=40000)["traffic"] is None
    committed = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    assert set(json.load(open(committed))) >= {"size", "GB_per_launch", "engine_sha"}


def test_stage_entry():
    sys.path.insert(0, ROOT)
    import bench

    e = bench.stage_entry(0.181, 1_600_000_000, 6)
    assert e["ms"] == 181.0 and e["alg_bytes_per_cell"] == 6 and abs(e["alg_GBps"] - 9.6 / 0.181) < 0.1
    assert abs(e["frac_of_peak"] - 9.6 / 0.181 / 8000) < 1e-4 and abs(e["Mcells_s"] - 1600 / 0.181) < 0.1
    assert bench.STAGE_BYTES == {"d8_flow_directions": 5, "directions_plus_flat_resolution": 6, "d8_flow_accum": 9,
                                 "resolve_flats_epsilon": 8, "fa_d8": 20, "priority_flood_epsilon": 8,
                                 "priority_flood_flowdirs": 5,
                                 "dinf_flow_directions": 8, "fa_tarboton": 20}   # SURVEY.md section 8d; D-infinity: a11 / VERDICT r03 #8


def test_bench_stdout_is_reserved_for_the_json_line():
    """Anything bench.py (or a library under it) prints goes to stderr; here the run stops at the launch check."""
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0 and r.stdout == "" and "WORLD_SIZE=1" in r.stderr


def test_full_size_reference_child_leg(tmp_path):
    """bench.py's cpu_baseline at the metric's own configuration: the compiled reference on the WHOLE DEM runs in a child
    process (`bench.py --cpu-full-child <raw f32> <n> <json>`) beside the GPU stages and hands back its time, the number of
    cells it raised and the band digests of its output -- which the parent compares with the GPU fill's.  Here: the child on
    a small raster; its digests equal those of the C restatement's fill (the digests are the S3 parity tests' own)."""
    import numpy as np
    import pytest

    import oracle
    from richdem_amd.synth import fractal_dem

    oracle.build()
    if not oracle.ref.available:
        pytest.skip("oracle/_ref/libref.so (the compiled reference) is not on this machine")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from digest import band_digests_np

    n = 1200
    z = fractal_dem(n, n, seed=3)
    raw, res = str(tmp_path / "dem.f32"), str(tmp_path / "out.json")
    z.tofile(raw)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-full-child", raw, str(n), res])
    got = json.load(open(res))
    exp = oracle.port.fill(z, 8)
    assert got["kind"] == "reference" and got["seconds"] > 0 and got["cells_raised"] == int((exp != z).sum())
    assert np.array_equal(np.array(got["digests"], dtype=np.uint64), band_digests_np(exp))
