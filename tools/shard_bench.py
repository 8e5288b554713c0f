#!/usr/bin/env python3
"""Times the phases of the row-block shard protocol on ONE GPU, shard after shard (what each rank of an
N-GPU run does in parallel): local phase + export per shard, host graph solve, finish per shard."""
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
    import numpy as np
    import torch

    import richdem_amd as rd
    from richdem_amd.sharded import GpuShardEngine, graph_solve, row_split

    n, S = args.size, args.shards
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    ref = Z.clone()
    rd.fill_depressions_dev(ref)
    torch.cuda.synchronize()
    for rep in range(2):
        W = Z.clone()
        blocks = [W[a:b] for a, b in row_split(n, S)]
        engs, keys, edges, t_local = [], [], [], []
        for s, blk in enumerate(blocks):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e = GpuShardEngine()
            k, ed = e.begin(blk, s > 0, s + 1 < S, 8)
            torch.cuda.synchronize(); t_local.append(time.perf_counter() - t0)
            engs.append(e); keys.append(k); edges.append(ed)
        t0 = time.perf_counter()
        levels = graph_solve(np.stack(keys), edges, 8)
        t_solve = time.perf_counter() - t0
        t_fin = []
        for s, e in enumerate(engs):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.finish(levels[s])
            torch.cuda.synchronize(); t_fin.append(time.perf_counter() - t0)
    # device-resident variant
    import torch as _t
    from richdem_amd.sharded import graph_solve_dev
    W2 = Z.clone()
    blocks = [W2[a:b] for a, b in row_split(n, S)]
    for rep in range(2):
        W2.copy_(Z)
        engs, keys, edges = [], [], []
        for s, blk in enumerate(blocks):
            e = GpuShardEngine(); k, ed = e.begin_dev(blk, s > 0, s + 1 < S, 8)
            engs.append(e); keys.append(k); edges.append(ed)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cap = max(int(ed.shape[0]) for ed in edges)
        edges_all = _t.zeros((S, cap, 3), dtype=_t.int32, device="cuda")
        for s, ed in enumerate(edges):
            edges_all[s, : ed.shape[0]] = ed
        counts = _t.tensor([int(ed.shape[0]) for ed in edges], dtype=_t.int32, device="cuda")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        lv = graph_solve_dev(_t.stack(keys), edges_all, counts, 8)
        torch.cuda.synchronize(); t_gsolve = time.perf_counter() - t1
        for s, e in enumerate(engs):
            e.finish_dev(lv[s].contiguous())
        torch.cuda.synchronize()
    ok2 = bool((W2 == ref).all())
    ok = bool((W == ref).all()) and ok2
    print(json.dumps({"size": n, "shards": S, "identical_to_single_block": ok,
                      "local_ms_per_shard": [round(t * 1e3, 2) for t in t_local],
                      "edges_per_shard": [int(len(e)) for e in edges],
                      "graph_solve_ms": round(t_solve * 1e3, 2), "graph_solve_gpu_ms": round(t_gsolve * 1e3, 2),
                      "finish_ms_per_shard": [round(t * 1e3, 2) for t in t_fin]}))


if __name__ == "__main__":
    main()
