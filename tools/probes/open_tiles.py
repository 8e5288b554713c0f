#!/usr/bin/env python3
"""How many 64 x 64 tiles of the filled bench DEM are entirely NO_FLOW (open water) / partly / not at all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import richdem_amd as rd
n = 40000
Z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(Z, seed=3)
rd.fill_depressions_dev(Z)
dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
rd.d8_flow_directions_dev(Z, -9999.0, dirs)
m = (dirs == 0)
t = m[: n // 64 * 64, : n // 64 * 64].reshape(n // 64, 64, n // 64, 64).permute(0, 2, 1, 3).reshape(-1, 4096)
s = t.sum(1)
print("tiles", s.numel(), "all NO_FLOW", int((s == 4096).sum()), "none", int((s == 0).sum()), "partly", int(((s > 0) & (s < 4096)).sum()),
      "cells in full tiles", int((s == 4096).sum()) * 4096, "of NO_FLOW cells", int(m.sum()))
