"""`_richdem` on the MI355X engine: the extension module the reference's Python package imports
(wrappers/pyrichdem/richdem/__init__.py: ``import _richdem``), built from wrappers/pyrichdem_gpu/pywrapper_gpu.cpp
over the C++ shim and librdgpu.so.

    import richdem_amd.compat as compat
    compat.install()          # sys.modules['_richdem'] = the GPU module
    import richdem as rd      # the reference's own package, unchanged: rd.FillDepressions(...) now runs on the GPU

There is no CPU fallback: without the built module `install()` raises."""
from __future__ import annotations

import glob
import importlib.util
import os
import subprocess
import sys

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wrappers", "pyrichdem_gpu")


def module_path() -> str | None:
    hits = sorted(glob.glob(os.path.join(_DIR, "_richdem*.so")))
    return hits[0] if hits else None


def build() -> str:
    """g++ + pybind11, in-tree (wrappers/pyrichdem_gpu/Makefile).  librdgpu.so must exist (richdem_amd.build())."""
    subprocess.check_call(["make", "-C", _DIR], stdout=subprocess.DEVNULL)
    p = module_path()
    if p is None:
        raise RuntimeError("wrappers/pyrichdem_gpu: the _richdem module was not produced")
    return p


def load():
    """The GPU `_richdem` module (not yet registered under that name)."""
    p = module_path()
    if p is None:
        raise ImportError(f"{_DIR}/_richdem*.so is missing: run richdem_amd.compat.build() "
                          "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    try:   # torch bundles its own HIP runtime; when both live in one process it must be loaded first (see _lib.lib)
        import torch  # noqa: F401
    except ImportError:
        pass
    spec = importlib.util.spec_from_file_location("_richdem", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install():
    """Registers the GPU module as `_richdem` (and its depression_hierarchy submodule), so that a later
    `import richdem` of the reference's package binds to it.  Returns the module."""
    mod = sys.modules.get("_richdem")
    if mod is not None and getattr(mod, "engine", None) == "rdgpu":
        return mod
    mod = load()
    sys.modules["_richdem"] = mod
    sys.modules["_richdem.depression_hierarchy"] = mod.depression_hierarchy
    return mod
