#!/usr/bin/env python3
"""bench.py -- Mcells/s of the Priority-Flood-equivalent fill on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S]

A step = one fill (rdgpu_fill_dev_f32 through the C-ABI) of one synthetic float32 DEM, G(seed) of
SURVEY.md section 8d, already resident in HBM when the timed region starts.  At N=1 the workload is
BASELINE config "40000x40000 float32 DEM, Priority-Flood fill on 1 MI355X".  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)



def cpu_baseline(Z, sample: int):
    """Times the reference's FillDepressions<D8> (or the C port) on a bounded window of the SAME DEM,
    on this box's host cores (1 thread: the reference fill has no OpenMP).  Reported baseline only."""
    import numpy as np

    import oracle  # checker, used here only for the reported CPU baseline

    s = min(sample, Z.shape[0], Z.shape[1])
    win = Z[:s, :s].cpu().numpy().copy()
    if oracle.ref.available:
        be, kind, what = oracle.ref, "reference", "PriorityFlood_Zhou2016 (unmodified reference headers, oracle/_ref)"
    else:
        if not oracle.port.available:
            oracle.build()
        be, kind, what = oracle.port, "port", "oracle/oracle.c Priority-Flood (Barnes2014 improved)"
    t0 = time.perf_counter()
    out = be.fill(win, 8)
    dt = time.perf_counter() - t0
    assert out.shape == win.shape and np.isfinite(out).all()
    return {
        "value": round(s * s / 1e6 / dt, 3),
        "unit": "Mcells/s",
        "cores": 1,
        "kind": kind,
        "sample": f"{s}x{s} top-left window of the bench DEM, {what}, {dt:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=40000, help="DEM is size x size cells")
    ap.add_argument("--cpu-sample", type=int, default=10000, help="window edge for the CPU baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()

    # stdout must carry exactly ONE line, the JSON of rank 0.  Libraries print there too (RCCL: "Librccl path : ...",
    # flushed from the C stdio buffer at exit, on every rank), so file descriptor 1 is pointed at stderr for the
    # whole run and the JSON line is written to the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch

    import richdem_amd as rd
    from richdem_amd.roofline import fill_roofline

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("RDGPU_BENCH_FORCE_SHARDED") == "1":   # the env switch runs the N>1 code path on 1 rank
        import torch.distributed as dist

        # RCCL logs to stdout by default; route them to a file so stdout carries exactly one JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rdgpu_rccl_%h_%p.log")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from richdem_amd.sharded import bench_sharded

        out = bench_sharded(args, rank, world)
        if out is not None:   # rank 0
            emit(out)
        return

    n = args.size
    cells = n * n
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    bufs = [Z.clone() for _ in range(args.steps)]   # fill is in place: one pristine copy per timed step
    scratch = Z.clone()
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        scratch.copy_(Z)
        rd.fill_depressions_dev(scratch)
    torch.cuda.synchronize()

    rd.profile_reset()
    rd.profile_enable(True)   # HIP events around every kernel, on the stream the kernel is launched on
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        rd.fill_depressions_dev(bufs[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rd.profile_enable(False)
    prof = rd.profile_totals()
    stats = rd.fill_stats()

    changed = float((bufs[0] != Z).float().mean())
    ms_step = dt * 1e3 / args.steps
    value = cells / 1e6 / (dt / args.steps)

    roofline = fill_roofline(prof, stats, cells, args.steps, dt / args.steps,
                             os.path.join(ROOT, "profiles", "pmc_traffic.json"), n)
    out = {
        "metric": "Mcells/s Priority-Flood fill, 40k x 40k f32 DEM",
        "value": round(value, 2),
        "unit": "Mcells/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{n}x{n} float32 fractal value-noise DEM G(seed={args.seed}), FillDepressions<D8>, HBM-resident",
            "cells": cells,
            "basins": stats["basins"],
            "boruvka_rounds": stats["rounds"],
            "pair_records": stats["edge_records"],
            "jump_passes": stats["jump_passes"],
            "cells_raised_frac": round(changed, 4),
            "parallelism": "1 GPU",
        },
        "roofline": roofline,
        "kernels_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
    }
    if args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(Z, args.cpu_sample)
    emit(out)


if __name__ == "__main__":
    main()
