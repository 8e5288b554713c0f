"""ctypes loader for librdgpu.so.  Fails loudly: no fallback of any kind."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RdgpuError(RuntimeError):
    """Raised when a C-ABI call returns non-zero (mirrors the reference's std::runtime_error)."""


def lib_path() -> str:
    # RDGPU_LIB: another build of the same library (tools/probes: librdgpu_probe.so, `make -C richdem_amd/csrc probe`)
    return os.environ.get("RDGPU_LIB") or os.path.join(_HERE, "librdgpu.so")


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into richdem_amd/librdgpu.so (in-tree)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return lib_path()


class rdgpu_fill_stats(ctypes.Structure):
    _fields_ = [
        ("cells", ctypes.c_uint64),
        ("basins", ctypes.c_uint64),
        ("rounds", ctypes.c_uint32),
        ("jump_passes", ctypes.c_uint32),
        ("scan_tiles", ctypes.c_uint64),
        ("tile_cells", ctypes.c_uint32),
        ("edge_records", ctypes.c_uint32),
        ("host_syncs", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RdgpuError(
                f"{p} is missing: the HIP extension is not built (run richdem_amd.build() / "
                "python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback."
            )
        # torch bundles its own HIP runtime; when both live in one process it must be the first one
        # loaded (otherwise torch.cuda reports "No HIP GPUs are available").  torch is plumbing here:
        # device memory, streams, torch.distributed.
        try:
            import torch  # noqa: F401
        except ImportError:  # pure C-ABI use without torch is fine
            pass
        L = ctypes.CDLL(p)
        L.rdgpu_last_error.restype = ctypes.c_char_p
        L.rdgpu_version.restype = ctypes.c_char_p
        L.rdgpu_profile_name.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().rdgpu_last_error().decode(errors="replace")
        err = RdgpuError(f"{what}: {msg} (code {rc})" if what else f"{msg} (code {rc})")
        err.code = rc
        raise err


def fill_stats() -> dict:
    st = rdgpu_fill_stats()
    check(lib().rdgpu_fill_get_stats(ctypes.byref(st)), "rdgpu_fill_get_stats")
    return {"cells": st.cells, "basins": st.basins, "rounds": st.rounds, "jump_passes": st.jump_passes,
            "scan_tiles": st.scan_tiles, "tile_cells": st.tile_cells, "edge_records": st.edge_records,
            "host_syncs": st.host_syncs}


def profile_enable(on: bool = True) -> None:
    lib().rdgpu_profile_enable(1 if on else 0)


def profile_collect() -> None:
    check(lib().rdgpu_profile_collect(), "rdgpu_profile_collect")


def profile_reset() -> None:
    check(lib().rdgpu_profile_reset(), "rdgpu_profile_reset")


def profile_totals() -> dict:
    """{kernel name: (total_ms, launches)} measured with HIP events on the launch stream."""
    L = lib()
    profile_collect()
    out = {}
    i = 0
    while True:
        nm = L.rdgpu_profile_name(i)
        if nm is None:
            break
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        L.rdgpu_profile_get(nm, ctypes.byref(ms), ctypes.byref(n))
        out[nm.decode()] = (ms.value, n.value)
        i += 1
    return out
