#!/usr/bin/env python3
"""Per-stage timing of the whole hot path on one GPU, HBM-resident (not the headline bench):
fill -> d8 directions -> flat resolution -> d8_flow_accum -> FA_D8, with per-kernel HIP-event totals."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import torch

    import richdem_amd as rd

    n = args.size
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    W = Z.clone()
    dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    res = {"size": n, "cells": n * n}

    def timed(name, fn, reps=args.reps):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[name + "_ms"] = round(best * 1e3, 2)
        res[name + "_Mcells_s"] = round(n * n / 1e6 / best, 1)

    def fill():
        W.copy_(Z)
        rd.fill_depressions_dev(W)
    timed("fill_incl_copy", fill)
    timed("d8_flowdirs", lambda: rd.d8_flow_directions_dev(W, -9999.0, dirs))
    flat = lambda: rd.d8_flow_directions_dev(W, -9999.0, dirs, flats=True)
    flat()                                     # first call grows the workspace (hipMalloc): not timed
    timed("flat_resolution_incl_d8", flat, reps=2)
    rd.profile_reset(); rd.profile_enable(True)
    flat()                                     # per-kernel times from a separate, instrumented run
    rd.profile_enable(False)
    res["flat_kernels_ms"] = {k: round(v[0], 2) for k, v in sorted(rd.profile_totals().items(), key=lambda kv: -kv[1][0])[:12]}
    class S(ctypes.Structure):
        _fields_ = [("low", ctypes.c_uint64), ("high", ctypes.c_uint64), ("noflow", ctypes.c_uint64), ("away", ctypes.c_uint32), ("towards", ctypes.c_uint32)]
    st = S(); rd.lib().rdgpu_flat_get_stats(ctypes.byref(st))
    res["flat_stats"] = {"low_edges": st.low, "high_edges": st.high, "noflow": st.noflow, "away_levels": st.away, "towards_levels": st.towards}
    def fa():
        area.fill_(1.0)
        rd.fa_d8_dev(W, -9999.0, area)
    rd.d8_flow_accum_dev(dirs, area); fa()     # workspace growth, not timed
    timed("d8_flow_accum_f64", lambda: rd.d8_flow_accum_dev(dirs, area), reps=2)
    timed("fa_d8_incl_fill", fa, reps=2)
    rd.profile_reset(); rd.profile_enable(True)
    rd.d8_flow_accum_dev(dirs, area); fa()
    rd.profile_enable(False)
    res["accum_kernels_ms"] = {k: round(v[0], 2) for k, v in sorted(rd.profile_totals().items(), key=lambda kv: -kv[1][0])[:8]}
    res["max_area"] = float(area.max().item())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
