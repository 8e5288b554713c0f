#!/usr/bin/env python3
"""max flat_mask value (~ 2 x the deepest towards level) and relaxation rounds at --size"""
import argparse, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=10000); a = ap.parse_args()
import torch
import richdem_amd as rd
n = a.size
Z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(Z, seed=3)
rd.fill_depressions_dev(Z)
z = Z.cpu().numpy()
dirs, mask, labels = rd.resolve_flats(z, -9999.0)
print("size", n, "max mask", int(mask.max()), "flat cells", int((mask > 0).sum()), "flats", int(labels.max()))
