#!/usr/bin/env python3
"""A/B timing of one stage under several environment switches, in ONE process on one GPU (a gpurun call is expensive):

    python tools/ab_stage.py --stage flats --size 40000 --cfg "" --cfg "RDGPU_FLAT_SUPER=0" --cfg "RDGPU_FLAT_SUPER_OCC=5"

Every configuration's output is compared with the first one's (the engine's switches must not change results); the
best of --reps wall times and the per-kernel HIP-event totals of one extra instrumented run are reported as one JSON line.
Stages: fill, flats (directions + flat resolution), accum (d8_flow_accum f64 on the flat-resolved directions),
rfe (ResolveFlatsEpsilon), fa_d8 (on the epsilon-resolved DEM), eps (PriorityFloodEpsilon)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="flats")
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cfg", action="append", default=[], help='"K=V K2=V2" (empty string: defaults)')
    ap.add_argument("--kernels", type=int, default=10, help="kernels listed per configuration")
    args = ap.parse_args()
    import torch

    import richdem_amd as rd

    n = args.size
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    W = Z.clone()
    rd.fill_depressions_dev(W)
    nodata = -9999.0
    stage = args.stage
    dirs = area = E = None
    if stage in ("flats", "accum"):
        dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    if stage == "accum":
        rd.d8_flow_directions_dev(W, nodata, dirs, flats=True)
        area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    if stage in ("rfe", "fa_d8", "eps"):
        E = W.clone()
    if stage == "fa_d8":
        rd.resolve_flats_epsilon_dev(E, nodata)
        area = torch.empty((n, n), dtype=torch.float64, device="cuda")
    scratch = Z.clone() if stage in ("fill", "eps") else None

    def prep():
        if stage == "fill":
            scratch.copy_(Z)
        elif stage == "rfe":
            E.copy_(W)
        elif stage == "eps":
            scratch.copy_(Z)
        elif stage == "fa_d8":
            area.fill_(1.0)

    def run():
        if stage == "fill":
            rd.fill_depressions_dev(scratch)
            return scratch
        if stage == "flats":
            rd.d8_flow_directions_dev(W, nodata, dirs, flats=True)
            return dirs
        if stage == "accum":
            rd.d8_flow_accum_dev(dirs, area)
            return area
        if stage == "rfe":
            rd.resolve_flats_epsilon_dev(E, nodata)
            return E
        if stage == "fa_d8":
            rd.fa_d8_dev(E, nodata, area)
            return area
        if stage == "eps":
            rd.fill_epsilon_dev(scratch, nodata)
            return scratch
        raise SystemExit("unknown stage " + stage)

    results = []
    ref = None
    cfgs = args.cfg or [""]
    for cfg in cfgs:
        env = dict(kv.split("=", 1) for kv in cfg.split()) if cfg.strip() else {}
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            prep()
            out = run()                      # workspace growth / first use: not timed
            torch.cuda.synchronize()
            same = None
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(out, ref))
            best = 1e9
            for _ in range(args.reps):
                prep()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            rd.profile_reset()
            rd.profile_enable(True)
            prep()
            run()
            torch.cuda.synchronize()
            rd.profile_enable(False)
            prof = rd.profile_totals()
            r = {"cfg": cfg, "ms": round(best * 1e3, 3), "same_as_first": same,
                 "kernels_ms": {k: [round(v[0], 3), int(v[1])] for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:args.kernels]}}
            if stage in ("flats", "rfe"):
                r["flat_stats"] = rd.flat_stats()
            if stage == "fill":
                r["fill_stats"] = rd.fill_stats()
            results.append(r)
            print(json.dumps(r), file=sys.stderr, flush=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    print(json.dumps({"stage": stage, "size": n, "results": results}))


if __name__ == "__main__":
    main()
