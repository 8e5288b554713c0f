#!/usr/bin/env python3
"""visit statistics of the bitmap search at --size (RDGPU_FLAT_TRACE: the STATS instantiation of k_relax_bits)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=40000); a = ap.parse_args()
import torch
import richdem_amd as rd
n = a.size
Z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(Z, seed=3)
rd.fill_depressions_dev(Z)
dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
rd.d8_flow_directions_dev(Z, -9999.0, dirs, flats=True)
torch.cuda.synchronize()
os.environ["RDGPU_FLAT_TRACE"] = "1"
rd.d8_flow_directions_dev(Z, -9999.0, dirs, flats=True)
torch.cuda.synchronize()
