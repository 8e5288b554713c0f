"""roofline object of bench.py's JSON line for the fill (8 algorithmic bytes per cell).

Every raster kernel of the fill is one pass over the DEM (or the rank's row block), so a launch's algorithmic bytes
are 8 B x the cells it visits; the kernel with the largest share of the step is reported as the dominant one."""
from __future__ import annotations

import json
import os

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FILL_ALG_BYTES_PER_CELL = 8.0  # read z (4 B) + write W (4 B)
RASTER_KERNELS = ("fill.scan", "fill.descent", "fill.tile_label", "fill.finalize")


def fill_roofline(prof: dict, stats: dict, cells: int, steps: int, step_seconds: float | None = None,
                  traffic_file: str | None = None, size: int | None = None) -> dict | None:
    """prof: {kernel: (total_ms, launches)} over `steps` fills of `cells` cells each (HIP events on the launch
    stream); stats: rdgpu_fill_stats of the last fill.  traffic_file: profiles/pmc_traffic.json (rocprofv3 PMC)."""
    dominant = max(RASTER_KERNELS, key=lambda k: prof.get(k, (0.0, 0))[0])
    k_ms, k_n = prof.get(dominant, (0.0, 0))
    if not k_n:
        return None
    total_kernel_ms = sum(v[0] for v in prof.values())
    launches = k_n / steps
    if dominant == "fill.scan":   # tiles visited x cells per tile (raster fallback rounds skip dead tiles)
        visited_cells = stats["scan_tiles"] * stats["tile_cells"]
    else:
        visited_cells = cells * launches
    achieved = visited_cells * FILL_ALG_BYTES_PER_CELL / (k_ms / steps / 1e3) / 1e9
    traffic = None
    if traffic_file:
        try:
            with open(traffic_file) as f:
                pt = json.load(f)
            if pt.get("size") == size:
                traffic = pt.get("GB_per_launch", {}).get(dominant)
        except OSError:
            pass
    out = {
        "bound": "hbm",
        "kernel": dominant,
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_unit": "GB per launch (rocprofv3 PMC, profiles/)" if traffic is not None else None,
        "avg_launch_ms": round(k_ms / k_n, 4),
        "launches_per_step": launches,
        "alg_GB_per_launch": round(visited_cells * FILL_ALG_BYTES_PER_CELL / launches / 1e9, 3),
        "share_of_kernel_time": round(k_ms / total_kernel_ms, 3) if total_kernel_ms else None,
    }
    if step_seconds:
        out["whole_fill_alg_GBps"] = round(cells * FILL_ALG_BYTES_PER_CELL / step_seconds / 1e9, 1)
    return out
