#!/usr/bin/env python3
"""Wall times of the f2 outputs at --size on an HBM-resident DEM (second call of each)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=40000); a = ap.parse_args()
import numpy as np
import torch
import richdem_amd as rd
n = a.size
Z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(Z, seed=3)
out = {"size": n}
def timed(name, fn):
    best = 1e9
    for _ in range(2):
        W = Z.clone(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(W); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out[name + "_ms"] = round(best * 1e3, 1)
timed("fill", lambda W: rd.fill_depressions_dev(W))
timed("fill_epsilon", lambda W: rd.fill_epsilon_dev(W, -9999.0))
out["epsilon_stats"] = rd.epsilon_stats()
print(json.dumps(out))
