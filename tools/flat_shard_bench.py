#!/usr/bin/env python3
"""Sharded flat resolution on ONE GPU, all row blocks driven in lockstep (what the ranks of an N-GPU run do in
parallel): wall time, exchanges per phase, equality with the single-block result."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()
    import torch

    import richdem_amd as rd
    from richdem_amd.sharded import flat_resolution_blocks

    n = args.size
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    rd.fill_depressions_dev(Z)
    exp = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    out = {}
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rd.d8_flow_directions_dev(Z, -9999.0, exp, flats=True)
        torch.cuda.synchronize(); out["single_block_ms"] = (time.perf_counter() - t0) * 1e3
    for rep in range(2):
        rd.profile_reset(); rd.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got, ex = flat_resolution_blocks(Z, -9999.0, args.shards)
        torch.cuda.synchronize(); out["sharded_lockstep_ms"] = (time.perf_counter() - t0) * 1e3
        rd.profile_enable(False)
        tot = rd.profile_totals()
    out["exchanges"] = ex
    out["equal"] = bool((got == exp).all())
    out["kernel_ms_all_shards"] = {k: [round(v[0], 2), v[1]] for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:10]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
