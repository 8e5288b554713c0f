#!/usr/bin/env python3
"""BASELINE configs[2] and [4] pinned at FULL size against the compiled reference, cell for cell.

    python tests/tools/parity40k.py [--size 40000] [--out gpurun_out/r02_parity40k.json]

On the GPU box: the 40000 x 40000 float32 DEM G(seed=3) of bench.py is filled by the unmodified reference
(PriorityFlood_Zhou2016 = FillDepressions<D8>, oracle/_ref, one host thread) and by rdgpu_fill_dev_f32; then
barnes_flat_resolution_d8(alter=false) and d8_flow_accum<uint8,double> of the reference run on the filled DEM and on
the resulting directions, and the GPU stages run on the same inputs.  Every cell is compared (`==`); the JSON holds the
mismatch counts and the reference's wall times -- which are also the real full-size CPU baseline.  Test
infrastructure: uses oracle/ as the checker.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r02_parity40k.json"))
    ap.add_argument("--skip-flats", action="store_true")
    ap.add_argument("--skip-accum", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch

    import oracle
    import richdem_amd as rd

    assert oracle.ref.available, "oracle/_ref/libref.so (the compiled reference) must travel with the snapshot"
    n = args.size
    res = {"size": n, "cells": n * n, "seed": args.seed, "host_cores": os.cpu_count(),
           "reference": "unmodified reference headers, oracle/_ref/libref.so (g++ -O3 -fopenmp -DNDEBUG), fill and d8_flow_accum on 1 thread"}
    with open("/proc/meminfo") as f:
        res["host_mem_GB"] = round(int(f.readline().split()[1]) / 1e6, 1)

    def save():
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)

    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    zh = Z.cpu().numpy()

    # ---- fill ------------------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    ref_fill = oracle.ref.fill(zh, 8)
    res["ref_fill_s"] = round(time.perf_counter() - t0, 2)
    res["ref_fill_Mcells_s"] = round(n * n / 1e6 / res["ref_fill_s"], 2)
    W = Z.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rd.fill_depressions_dev(W)
    torch.cuda.synchronize()
    res["gpu_fill_ms_first_call"] = round((time.perf_counter() - t0) * 1e3, 2)
    R = torch.from_numpy(ref_fill).cuda()
    res["fill_mismatches"] = int((W != R).sum().item())
    res["fill_cells_raised"] = int((R != Z).sum().item())
    del R, Z
    save()
    print("fill", res, flush=True)

    # ---- directions + flat resolution ----------------------------------------------------------------------------
    dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    if not args.skip_flats:
        t0 = time.perf_counter()
        ref_dirs = oracle.ref.flat_resolution(ref_fill, -9999.0)
        res["ref_flat_resolution_s"] = round(time.perf_counter() - t0, 2)
        res["ref_flat_resolution_Mcells_s"] = round(n * n / 1e6 / res["ref_flat_resolution_s"], 2)
        rd.d8_flow_directions_dev(W, -9999.0, dirs, flats=True)
        torch.cuda.synchronize()
        RD = torch.from_numpy(ref_dirs).cuda()
        res["flat_dirs_mismatches"] = int((dirs != RD).sum().item())
        res["noflow_cells_left"] = int((RD == 0).sum().item())
        del RD
        save()
        print("flats", res, flush=True)
    else:
        rd.d8_flow_directions_dev(W, -9999.0, dirs, flats=True)
        ref_dirs = dirs.cpu().numpy()
    del ref_fill

    # ---- d8_flow_accum ---------------------------------------------------------------------------------------------
    if not args.skip_accum:
        t0 = time.perf_counter()
        ref_area = oracle.ref.d8_flow_accum(ref_dirs, 255, np.float64)
        res["ref_d8_flow_accum_s"] = round(time.perf_counter() - t0, 2)
        res["ref_d8_flow_accum_Mcells_s"] = round(n * n / 1e6 / res["ref_d8_flow_accum_s"], 2)
        area = torch.empty((n, n), dtype=torch.float64, device="cuda")
        RDd = torch.from_numpy(ref_dirs).cuda()
        rd.d8_flow_accum_dev(RDd, area)
        torch.cuda.synchronize()
        RA = torch.from_numpy(ref_area).cuda()
        res["d8_flow_accum_mismatches"] = int((area != RA).sum().item())
        res["max_area"] = float(RA.max().item())
        save()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
