"""roofline object of bench.py's JSON line for the fill (8 algorithmic bytes per cell).

The fill as a whole reads z and writes W once: 8 B/cell for ALL of its kernels together -- that is the figure the object
leads with (`whole_fill_*`).  Below it the dominant raster kernel is reported the way the bench contract defines a
kernel's roofline: every raster kernel of the fill is one pass over the DEM (or the rank's row block), so a launch's
algorithmic bytes are 8 B x the cells it visits, divided by that kernel's own launch time (HIP events)."""
from __future__ import annotations

import hashlib
import json
import os

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FILL_ALG_BYTES_PER_CELL = 8.0  # read z (4 B) + write W (4 B)
RASTER_KERNELS = ("fill.scan", "fill.descent", "fill.tile_label", "fill.finalize")
_FILL_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "fill.hip")


def engine_sha() -> str | None:
    """Fingerprint of the fill's kernel source: PMC traffic figures are only valid for the kernels they were taken on."""
    try:
        with open(_FILL_SRC, "rb") as f:
            return hashlib.sha1(f.read()).hexdigest()[:16]
    except OSError:
        return None


def load_traffic(traffic_file: str | None, size: int | None) -> dict | None:
    """profiles/pmc_traffic.json if it was measured at this raster size AND on the current fill.hip (else None:
    a figure of another engine state is not reported)."""
    if not traffic_file:
        return None
    try:
        with open(traffic_file) as f:
            pt = json.load(f)
    except (OSError, ValueError):
        return None
    if pt.get("size") != size:
        return None
    if pt.get("engine_sha") is not None and pt.get("engine_sha") != engine_sha():
        return None
    return pt


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
    pt = load_traffic(traffic_file, size)
    traffic = pt.get("GB_per_launch", {}).get(dominant) if pt else None
    out = {"bound": "hbm"}
    if step_seconds:
        whole = cells * FILL_ALG_BYTES_PER_CELL / step_seconds / 1e9
        out["whole_fill_alg_GBps"] = round(whole, 1)
        out["whole_fill_frac"] = round(whole / HBM_PEAK_GBS, 4)
        if pt and pt.get("GB_per_fill") is not None:
            out["whole_fill_traffic_GB"] = pt["GB_per_fill"]
            out["whole_fill_pass_count"] = round(pt["GB_per_fill"] / (cells * FILL_ALG_BYTES_PER_CELL / 1e9), 2)
    out.update({
        "kernel": dominant,
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_unit": "GB per launch (rocprofv3 PMC of this fill.hip, profiles/)" if traffic is not None else None,
        "avg_launch_ms": round(k_ms / k_n, 4),
        "launches_per_step": launches,
        "alg_GB_per_launch": round(visited_cells * FILL_ALG_BYTES_PER_CELL / launches / 1e9, 3),
        "share_of_kernel_time": round(k_ms / total_kernel_ms, 3) if total_kernel_ms else None,
    })
    return out
