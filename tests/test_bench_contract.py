"""bench.py's contract pieces that do not need a GPU: the roofline object (richdem_amd/roofline.py) and the rule that
stdout carries nothing but rank 0's JSON line (library chatter goes to stderr)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_object():
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
    # traffic comes from the committed PMC summary, per kernel, only for the size it was measured at
    pt = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    per = json.load(open(pt))
    r = fill_roofline({"fill.scan": (19.5, 3)}, stats, cells, steps, traffic_file=pt, size=per["size"])
    assert r["traffic"] == per["GB_per_launch"]["fill.scan"] and r["traffic"] > r["alg_GB_per_launch"] * 0.9
    assert fill_roofline({"fill.scan": (19.5, 3)}, stats, cells, steps, traffic_file=pt, size=123)["traffic"] is None


def test_bench_stdout_is_reserved_for_the_json_line():
    """Anything bench.py (or a library under it) prints goes to stderr; here the run stops at the launch check."""
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0 and r.stdout == "" and "WORLD_SIZE=1" in r.stderr
