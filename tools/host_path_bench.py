#!/usr/bin/env python3
"""End-to-end time of the drop-in (host pointer) entry: rdgpu_fill_f32 on a pageable numpy array, H2D + fill + D2H."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    args = ap.parse_args()
    import numpy as np
    import torch

    import richdem_amd as rd

    n = args.size
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=3)
    host = Z.cpu().numpy()
    del Z
    torch.cuda.empty_cache()
    out = {"size": n, "GB": round(host.nbytes / 1e9, 2)}
    for rep in range(3):
        a = host.copy()
        t0 = time.perf_counter()
        rd.FillDepressions(a, in_place=True)
        out[f"fill_host_s_{rep}"] = round(time.perf_counter() - t0, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
