#!/usr/bin/env python3
"""Times the REFERENCE (unmodified headers, oracle/_ref/libref.so) on this box's host cores, stage by stage, on a
window of the bench DEM -- the CPU figures quoted next to the GPU ones in DESIGN.md / README.md.
The fill variants are single-threaded in the reference; the stencil stages use its OpenMP loops (all host cores)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8000)
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()
    import numpy as np

    import oracle
    from richdem_amd.synth import fractal_dem

    R = oracle.ref
    assert R.available, "oracle/_ref/libref.so is not built"
    n = args.size
    z = fractal_dem(n, n, args.seed)
    nd = np.float32(-9999)
    out = {"size": n, "cells": n * n, "host_cores": os.cpu_count()}

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        out[name] = {"s": round(dt, 2), "Mcells_s": round(n * n / 1e6 / dt, 2)}
        return r

    filled = timed("fill_zhou2016", lambda: R.fill(z, 8, oracle.ZHOU2016))
    timed("fill_barnes2014", lambda: R.fill(z, 8, oracle.BARNES2014_D8))
    timed("fill_wei2018", lambda: R.fill(z, 8, oracle.WEI2018))
    timed("d8_flow_directions", lambda: R.d8_flowdirs(filled, nd))
    dirs = timed("barnes_flat_resolution_d8", lambda: R.flat_resolution(filled, nd))
    timed("d8_flow_accum_f64", lambda: R.d8_flow_accum(dirs, 255, np.float64))
    timed("fa_d8", lambda: R.fa_d8(filled, nd))
    timed("fa_tarboton", lambda: R.fa_tarboton(filled, nd))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
