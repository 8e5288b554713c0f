"""richdem_amd -- MI355X-native depression filling / flow routing behind RichDEM's API.

The product is ``librdgpu.so`` (hand-written HIP for gfx950, C-ABI in ``include/rdgpu.h``) and the
C++ header shim ``include/rdgpu/richdem_gpu.hpp`` that keeps the reference's ``Array2D<T>`` signatures.
This Python package is thin plumbing over the same C-ABI (ctypes) for tests, ``bench.py`` and
multi-GPU sharding with ``torch.distributed``; function names follow the reference's Python wrapper
(``wrappers/pyrichdem/richdem/__init__.py``: ``FillDepressions``, ``FlowAccumulation``).

There is NO CPU fallback: if ``librdgpu.so`` is missing or a HIP call fails, calls raise.
"""
from __future__ import annotations

from ._lib import (  # noqa: F401
    RdgpuError,
    lib,
    lib_path,
    build,
    fill_stats,
    profile_enable,
    profile_collect,
    profile_reset,
    profile_totals,
)
from .pyrichdem import (  # noqa: F401
    rdarray,
    rd3array,
    FillDepressions,
    FlowAccumulation,
    FlowProportions,
    FlowAccumFromProps,
    ResolveFlats,
    BreachDepressions,
    TerrainAttribute,
    LoadGDAL,
    SaveGDAL,
    LoadNative,
    SaveNative,
)
from .api import (  # noqa: F401
    pf_flowdirs,
    pf_flowdirs_dev,
    pf_flowdirs_stats,
    pit_mask,
    fill_max_dep,
    watersheds,
    dinf_flow_directions,
    d8_flow_directions,
    d8_flow_accum,
    barnes_flat_resolution_d8,
    resolve_flats,
    resolve_flats_epsilon,
    fill_depressions_dev,
    d8_flow_directions_dev,
    d8_flow_accum_dev,
    fa_d8_dev,
    synth_dem_dev,
    resolve_flats_epsilon_dev,
    fill_epsilon_dev,
    fill_max_dep_dev,
    dinf_flow_directions_dev,
    fa_tarboton_dev,
    watersheds_dev,
    epsilon_stats,
    flat_stats,
    release_workspace,
)

__all__ = [
    "RdgpuError",
    "lib",
    "lib_path",
    "build",
    "rdarray",
    "rd3array",
    "LoadNative",
    "SaveNative",
    "FillDepressions",
    "FlowAccumulation",
    "FlowProportions",
    "FlowAccumFromProps",
    "dinf_flow_directions",
    "d8_flow_directions",
    "d8_flow_accum",
    "barnes_flat_resolution_d8",
    "resolve_flats",
    "fill_depressions_dev",
    "d8_flow_directions_dev",
    "d8_flow_accum_dev",
    "fa_d8_dev",
    "synth_dem_dev",
    "resolve_flats_epsilon_dev",
    "flat_stats",
    "release_workspace",
    "fill_stats",
    "profile_enable",
    "profile_collect",
    "profile_reset",
    "profile_totals",
    "ResolveFlats",
    "pf_flowdirs",
    "pf_flowdirs_dev",
    "pf_flowdirs_stats",
    "pit_mask",
    "fill_max_dep",
    "watersheds",
    "resolve_flats_epsilon",
    "fill_epsilon_dev",
    "epsilon_stats",
]
