"""The C-ABI library loads and exports every symbol include/rdgpu.h declares (no compute: no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rdgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(rdgpu_[a-z0-9_]+)\s*\(", src))
    # entry points declared through the RDGPU_DECL_MFD(SUF, T) macro
    for suf in re.findall(r"RDGPU_DECL_MFD\((\w+),", src):
        if suf != "SUF":
            names |= {f"rdgpu_dinf_flowdirs_{suf}", f"rdgpu_dinf_flowdirs_dev_{suf}", f"rdgpu_fm_tarboton_{suf}",
                      f"rdgpu_fa_tarboton_{suf}", f"rdgpu_fa_tarboton_dev_{suf}"}
    return sorted(n for n in names if "##" not in n and not n.endswith("_"))


def test_library_exports_every_declared_symbol(rd):
    L = rd.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert b"gfx950" in L.rdgpu_version()


def test_argument_errors_do_not_need_a_gpu(rd):
    with pytest.raises(rd.RdgpuError, match="topology"):
        rd.FillDepressions(np.zeros((4, 4), np.float32), topology="D6")
    with pytest.raises(rd.RdgpuError, match="dtype"):
        rd.FillDepressions(np.zeros((4, 4), np.complex64))
    with pytest.raises(rd.RdgpuError):
        rd.FillDepressions(np.zeros(4, np.float32))


def test_no_cpu_fallback_without_gpu(rd):
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rd.RdgpuError, match="(?i)hip|device"):
        rd.FillDepressions(np.zeros((8, 8), np.float32))


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for top in ("richdem_amd", "include", "apps", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in txt and "liboracle" not in txt and "libref" not in txt, (top, f)
