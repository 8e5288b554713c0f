#!/usr/bin/env python3
"""Fill time per element type on the bench DEM (quantised for the integer types), HBM resident."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import richdem_amd as rd

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=3)
    zmin = float(Z.min())
    out = {"size": n}
    for name, dt, scale in (("f32", torch.float32, 1.0), ("i32", torch.int32, 10.0), ("i16", torch.int16, 10.0), ("u8", torch.uint8, 0.1)):
        src = Z if dt == torch.float32 else ((Z - zmin) * scale).floor().clamp(0, 250 if dt == torch.uint8 else 32000).to(dt)
        best = 1e9
        for rep in range(3):
            W = src.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rd.fill_depressions_dev(W)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        out[name + "_ms"] = round(best * 1e3, 2)
        out[name + "_basins"] = rd.fill_stats()["basins"]
        del W, src
    print(json.dumps(out))


if __name__ == "__main__":
    main()
