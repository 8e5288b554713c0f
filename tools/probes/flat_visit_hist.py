#!/usr/bin/env python3
"""How many times the two level searches of flat resolution visit each 64 x 64 tile, and how many of those visits ran AHEAD of
the tile's order (a later visit started at a lower level).  Needs the probe build:
    make -C richdem_amd/csrc probe && RDGPU_LIB=richdem_amd/librdgpu_probe.so python tools/probes/flat_visit_hist.py --size 40000"""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=40000); ap.add_argument("--out", default=None)
a = ap.parse_args()
import torch
import richdem_amd as rd
n = a.size
Z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(Z, seed=3)
rd.fill_depressions_dev(Z)
dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")
rd.d8_flow_directions_dev(Z, -9999.0, dirs, flats=True)
torch.cuda.synchronize()
tiles = ((n + 63) // 64) ** 2
hist = torch.zeros((2, 5, tiles), dtype=torch.int32, device="cuda")
assert rd.lib().rdgpu_probe_flat_hist(ctypes.c_void_p(hist.data_ptr())) == 0
rd.d8_flow_directions_dev(Z, -9999.0, dirs, flats=True)
torch.cuda.synchronize()
rd.lib().rdgpu_probe_flat_hist(ctypes.c_void_p(0))
out = {"size": n, "tiles": tiles, "flat_stats": rd.flat_stats()}
for f, name in enumerate(("towards", "away")):
    v, wk, st, inv = (hist[f, k].long() for k in range(4))
    q = torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda", dtype=torch.float64)
    vv = v[v > 0].double()
    out[name] = {"tiles_visited": int((v > 0).sum()), "visits": int(v.sum()), "working_visits": int(wk.sum()),
                 "tiles_with_work": int((wk > 0).sum()), "levels_stepped": int(st.sum()), "visits_that_ran_ahead": int(inv.sum()),
                 "tiles_with_a_visit_that_ran_ahead": int((inv > 0).sum()),
                 "working_visits_per_tile_quantiles": [float(x) for x in torch.quantile(wk[wk > 0].double()[:16000000], q)],
                 "max_working_visits": int(wk.max()), "visits_per_tile_mean": float(vv.mean()),
                 "working_visit_histogram_1_2_3_4_5to8_9to16_17plus": [int(((wk >= lo) & (wk <= hi)).sum()) for lo, hi in
                                                                        ((1, 1), (2, 2), (3, 3), (4, 4), (5, 8), (9, 16), (17, 1 << 30))]}
print(json.dumps(out, indent=1))
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
