#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random shapes / element types / terrain styles / NoData, every entry
point through the C-ABI against the oracle.  Prints one JSON line {"cases": n, "mismatches": [...]}.
    python tests/tools/fuzz_parity.py --seconds 120 --seed 1"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ulp32(a, b):
    import numpy as np

    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return int(np.abs(ia - ib).max()) if a.size else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--trace", action="store_true", help="name every check on stderr as it completes (to locate a crash)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import oracle
    import richdem_amd as rd
    from richdem_amd.sharded import GpuAccumShard, accum_link_solve, flat_resolution_blocks, row_split

    def accum_one_exchange(dirs_np, world, adt):
        """the one-exchange protocol block after block on this GPU; None when it reports a loop"""
        tdt = {np.float64: torch.float64, np.float32: torch.float32, np.int32: torch.int32}[adt]
        dirs = torch.from_numpy(dirs_np).cuda()
        blocks = [dirs[a:b].contiguous() for a, b in row_split(dirs.shape[0], world)]
        shards, boxes, links, pend = [], [], [], []
        for k, blk in enumerate(blocks):
            sh = GpuAccumShard()
            sh.begin_local(blk, 255, blocks[k - 1][-1] if k > 0 else None, blocks[k + 1][0] if k + 1 < world else None)
            shards.append(sh)
            boxes.append(sh.outbox())
            lk, pn = sh.links()
            links.append(lk)
            pend.append(pn)
        inflow = None
        if int(torch.cat(pend).sum().item()) == 0:
            inflow = accum_link_solve(torch.stack(boxes), torch.stack(links), world, dirs.shape[1])
        if inflow is None:
            for sh in shards:
                sh.abort()
            return None
        out = []
        for k, (sh, blk) in enumerate(zip(shards, blocks)):
            sh.add_paths(inflow[k, 0] if k > 0 else None, inflow[k, 1] if k + 1 < world else None)
            a = torch.empty(blk.shape, dtype=tdt, device="cuda")
            sh.finish(a)
            out.append(a)
        return torch.cat(out, 0).cpu().numpy()
    from richdem_amd.synth import fractal_dem

    oracle.build()
    P = oracle.port
    rng = np.random.default_rng(args.seed)
    bad, cases, t_end = [], 0, time.time() + args.seconds
    dtypes = [np.float32, np.float32, np.int32, np.int16, np.uint8, np.uint16, np.float64, np.uint32]
    while time.time() < t_end:
        h, w = int(rng.integers(1, 260)), int(rng.integers(1, 300))
        if rng.random() < 0.2:
            h, w = int(rng.integers(1, 6)), int(rng.integers(1, 200))
        elif rng.random() < 0.12:   # many 64 x 64 tiles: the tile-to-tile paths of the flat search and the tile links
            h, w = int(rng.integers(260, 700)), int(rng.integers(300, 900))
        dt = dtypes[int(rng.integers(len(dtypes)))]
        style = int(rng.integers(5))
        if style == 0:
            z = rng.integers(0, int(rng.integers(2, 9)), (h, w)).astype(np.float64) * 3
        elif style == 1:
            z = fractal_dem(w, h, int(rng.integers(1 << 20))).astype(np.float64)
            z = z - z.min()
        elif style == 2:
            z = np.floor((fractal_dem(w, h, int(rng.integers(1 << 20))).astype(np.float64) + 500) * rng.choice([0.02, 0.1, 0.5]))
        elif style == 3:
            yy, xx = np.mgrid[0:h, 0:w]
            z = np.floor(np.hypot(yy - h / 2, xx - w / 2) * rng.choice([0.3, 1.0])) + rng.integers(0, 2, (h, w))
        else:
            z = rng.random((h, w)) * 100
        if np.issubdtype(dt, np.integer):
            info = np.iinfo(dt)
            z = np.clip(np.floor(z) + (0 if info.min == 0 else -20), info.min, min(info.max, 60000))
        dem = z.astype(dt)
        nd = dt(rng.choice([0, 5, 250])) if dt == np.uint8 else dt(-9999) if np.issubdtype(dt, np.floating) or np.iinfo(dt).min < 0 else dt(0)
        if rng.random() < 0.4 and h > 2 and w > 2:
            dem[rng.random((h, w)) < rng.choice([0.01, 0.1])] = nd
        tag = f"{h}x{w} {np.dtype(dt).name} style{style}"

        def chk(name, ok):
            nonlocal cases
            cases += 1
            if args.trace:
                print(cases, name, tag, file=sys.stderr, flush=True)
            if not ok:
                bad.append(f"{name} {tag}")

        try:
            topo = int(rng.choice([8, 4]))
            filled = rd.FillDepressions(dem, topology="D8" if topo == 8 else "D4")
            chk(f"fill{topo}", filled.tobytes() == P.fill(dem, topo).tobytes())
            if np.issubdtype(dt, np.floating) and np.unique(dem).size == dem.size:   # (the epsilon fill: defined by the DEM only without ties)
                chk(f"fill_epsilon{topo}", rd.FillDepressions(dem, epsilon=True, topology="D8" if topo == 8 else "D4", nodata=nd).tobytes()
                    == P.fill_epsilon(dem, nd, topo).tobytes())
            # r06: the sweep's other names -- HasDepressions, Wei2018 with its NoData-as-outlet seeds
            chk(f"has_depressions{topo}", rd.has_depressions(dem, "D8" if topo == 8 else "D4") == P.has_depressions(dem, topo))
            chk("wei2018", rd.fill_wei2018(dem, nd).tobytes() == P.fill_wei2018(dem, nd).tobytes())
            if dt in (np.uint32,):
                continue
            src = filled if rng.random() < 0.7 else dem
            chk("d8dirs", np.array_equal(rd.d8_flow_directions(src, nd), P.d8_flowdirs(src, nd)))
            exp_dirs = P.flat_resolution(src, nd)
            dirs = rd.barnes_flat_resolution_d8(src, nd)
            chk("flats", np.array_equal(dirs, exp_dirs))
            chk("epsilon", rd.ResolveFlats(src, nodata=nd).tobytes() == P.resolve_flats_epsilon(src, nd).tobytes())
            if h >= 4:
                world = int(rng.integers(2, max(3, min(9, h // 2 + 1))))
                if h // world >= 2:
                    got, _ = flat_resolution_blocks(torch.from_numpy(np.ascontiguousarray(src)).cuda(), nd, world)
                    chk(f"flats_sharded{world}", np.array_equal(got.cpu().numpy(), exp_dirs))
                    if dt != np.float64:
                        chk(f"fill_sharded{world}", rd.FillDepressions(dem, shards=world).tobytes() == P.fill(dem, 8).tobytes())
            adt = [np.float64, np.float32, np.int32][int(rng.integers(3))]
            d2 = dirs.copy()
            if rng.random() < 0.3:
                d2[rng.random(d2.shape) < 0.05] = rng.integers(0, 9)      # arbitrary directions, loops included
            exp_acc = P.d8_flow_accum(d2, 255, adt)
            chk("d8_accum", np.array_equal(rd.d8_flow_accum(d2, 255, adt), exp_acc))
            if h >= 2:
                world = int(rng.integers(2, min(h, 12) + 1))
                got = accum_one_exchange(d2, world, adt)
                if got is not None:   # (None: a direction loop was reported -- the iterated protocol's case)
                    chk(f"accum_one_exchange{world}", np.array_equal(got, exp_acc))
                elif d2 is dirs or np.array_equal(d2, dirs):
                    chk("accum_one_exchange_loopfree_reported_loop", False)
            chk("fa_d8_unit", np.array_equal(rd.FlowAccumulation(src, "D8", nodata=nd), P.fa_d8(src, nd)))
            wts = rng.integers(0, 7, (h, w)).astype(np.float64)
            chk("fa_d8", np.array_equal(rd.FlowAccumulation(src, "D8", nodata=nd, weights=wts), P.fa_d8(src, nd, wts)))
            m, x = [("Quinn", None), ("Holmgren", 2.0), ("Freeman", 1.1), ("D4", None), ("Holmgren", 0.7)][int(rng.integers(5))]
            gp, ep = rd.FlowProportions(src, m, nodata=nd, exponent=x), P.fm_mfd(src, nd, m, 1.0 if x is None else x)
            chk(f"fm_{m}", np.array_equal(gp > 0, ep > 0) and ulp32(gp, ep) <= 1)
            ga, ea = rd.FlowAccumulation(src, m, nodata=nd, exponent=x), P.fa_mfd(src, nd, m, 1.0 if x is None else x)
            chk(f"fa_{m}", np.array_equal(ga == -1, ea == -1) and np.allclose(ga, ea, rtol=2e-6, atol=0))
            ga, ea = rd.FlowAccumulation(src, "Dinf", nodata=nd), P.fa_tarboton(src, nd)
            chk("fa_dinf", np.array_equal(ga == -1, ea == -1) and np.allclose(ga, ea, rtol=2e-6, atol=0))
            # PriorityFloodFlowdirs on the same shape with every elevation distinct (a permutation ranked by the terrain)
            order = np.argsort(dem.astype(np.float64).ravel() + rng.random(h * w) * 0.5, kind="stable")
            perm = np.empty(h * w, np.int32)
            perm[order] = np.arange(h * w, dtype=np.int32)
            pd = perm.reshape(h, w)
            chk("pf_flowdirs", np.array_equal(rd.pf_flowdirs(pd, nodata=np.int32(-9999)), P.pf_flowdirs(pd, np.int32(-9999))))
            # ... and on the terrain itself, equal elevations and NoData cells as they are (r04: the stable queue's tie order;
            # small rasters only -- a plateau takes as many floods as it is deep)
            if h * w <= 20000 and dem.dtype != np.float64 and dem.dtype.itemsize <= 4:
                import warnings

                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    got = rd.pf_flowdirs(dem, nodata=nd)
                chk("pf_flowdirs_ties", np.array_equal(got, P.pf_flowdirs(dem, nd)) and rd.pf_flowdirs_stats()["unresolved"] == 0)
        except Exception as e:   # noqa: BLE001
            bad.append(f"EXC {tag}: {type(e).__name__}: {e}")
    kinds = {}
    for b in bad:
        kinds[b.split(" ")[0]] = kinds.get(b.split(" ")[0], 0) + 1
    print(json.dumps({"cases": cases, "n_mismatches": len(bad), "by_kind": kinds, "first": bad[:12]}))


if __name__ == "__main__":
    main()
