#!/usr/bin/env python3
"""ULP histogram (after an f32 cast) of FA_Tarboton / FA_Holmgren / FA_Freeman / FA_Quinn END TO END against the oracle:
how far the device libm's last-ulp differences in the proportions travel down the accumulation."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import richdem_amd as rd
from test_mfd_gpu import dems, ulp_diff_f32, MFD_METHODS
oracle.build()
out = {}
for name, dem in dems(oracle):
    nd = dem.dtype.type(250 if dem.dtype == np.uint8 else -9999)
    for method, x in [("Dinf", None)] + MFD_METHODS:
        got = rd.FlowAccumulation(dem, method, nodata=nd, exponent=x)
        exp = oracle.port.fa_tarboton(dem, nd) if method == "Dinf" else oracle.port.fa_mfd(dem, nd, method, 1.0 if x is None else x)
        u = ulp_diff_f32(got, exp)
        rel = float(np.abs(got / np.where(exp == 0, 1, exp) - 1).max())
        hist = np.bincount(np.minimum(u, 8).ravel(), minlength=9).tolist()
        out[f"{name}/{method}/{x}"] = {"max_ulp_f32": int(u.max()), "hist_0..8+": hist, "max_rel_f64": rel}
print(json.dumps(out, indent=1))
